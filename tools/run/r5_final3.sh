#!/bin/bash
# GPU box, last run of round 5: the gpu suite, both fuzzers of the front end for a short while, the inflate bench -- of the final build.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final3
mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt | tail -2
timeout 150 python tests/fuzz/fuzz_call.py 120 81 > $O/fuzz_call.log 2>&1; tail -1 $O/fuzz_call.log
timeout 100 python tests/fuzz/fuzz_inflate.py 75 31 > $O/fuzz_inflate.log 2>&1; tail -1 $O/fuzz_inflate.log
timeout 300 python tools/inflate_bench.py 524288 32768 > $O/inflate.log 2>&1; tail -1 $O/inflate.log > $O/inflate.json; grep "GB/s" $O/inflate.log
