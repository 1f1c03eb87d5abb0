#!/bin/bash
# GPU box, the round's last run: gpu suite + both inflate-touching fuzzers + the inflate bench on the final build (loop bounded again).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final6
mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt | tail -2
timeout 60 python tests/fuzz/fuzz_inflate.py 40 101 > $O/fuzz_inflate.log 2>&1; tail -1 $O/fuzz_inflate.log
timeout 60 python tests/fuzz/fuzz_call.py 40 102 > $O/fuzz_call.log 2>&1; tail -1 $O/fuzz_call.log
timeout 200 python tools/inflate_bench.py 524288 32768 > $O/inflate.log 2>&1; tail -1 $O/inflate.log > $O/inflate.json; grep "GB/s" $O/inflate.log
