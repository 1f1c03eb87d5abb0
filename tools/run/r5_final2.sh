#!/bin/bash
# GPU box, end of round 5 (second session): the driver's bench command, the inflate bench, the gpu test suite -- of the build with the long codes in the loop.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final2
mkdir -p $O
cd $R && export TMPDIR=/tmp
timeout 1100 python bench.py > $O/bench_driver_command.json 2> $O/bench_driver_command.err
tail -c 600 $O/bench_driver_command.err
timeout 300 python tools/inflate_bench.py 524288 32768 > $O/inflate.log 2>&1; tail -1 $O/inflate.log > $O/inflate.json; cat $O/inflate.log | grep "GB/s"
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt | tail -2
