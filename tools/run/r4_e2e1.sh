#!/bin/bash
# round 4, first pass of the file-sized end to end: launch test, 2^24-read and 2^26-read level-6 files through extract -> call / merge
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_launch.py tests/test_comm_native.py tests/test_cli.py -m gpu -x -q > gpurun_out/r4/t1.txt 2>&1; tail -3 gpurun_out/r4/t1.txt
timeout 900 python tools/e2e_bench.py $((1<<23)) --check-slabs 6 --out gpurun_out/r4/e2e_23.json > gpurun_out/r4/e2e_23.log 2>&1; tail -c 3000 gpurun_out/r4/e2e_23.log
timeout 1500 python tools/e2e_bench.py $((1<<25)) --check-slabs 8 --repeats 2 --out gpurun_out/r4/e2e_25.json > gpurun_out/r4/e2e_25.log 2>&1; tail -c 3000 gpurun_out/r4/e2e_25.log
