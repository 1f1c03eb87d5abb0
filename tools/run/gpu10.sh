cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_index.py tests/test_product_kats.py -x -q -m gpu 2>&1 | grep -v "^$" | grep -i "passed\|failed\|error" | tail -3
bash tools/run/ab.sh tools/ab/lib_cur.so strling_amd/lib/libstrling_amd.so
