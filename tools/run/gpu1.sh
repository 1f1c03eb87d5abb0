cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bgzf_device.py tests/test_front_device.py -x -q 2>&1 | tail -2
for w in 4 5; do echo "== waves $w"; STRL_LIB=tools/ab/lib_w$w.so timeout 900 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s"; done
echo "== waves 6 (default build)"; timeout 900 python tools/inflate_bench.py 524288 32768 2>&1 | grep "GB/s"
