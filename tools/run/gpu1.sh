cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bgzf_device.py tests/test_front_device.py -x -q 2>&1 | tail -5
timeout 900 python tools/inflate_bench.py 524288 32768 2>&1 | tail -4
