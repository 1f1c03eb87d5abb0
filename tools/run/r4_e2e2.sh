#!/bin/bash
# round 4: CLI + call suites with the threaded evidence pass and the limit route, then 2^26 and 2^28-read files end to end
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cli.py tests/test_call.py tests/test_bench_launch.py -m gpu -x -q > gpurun_out/r4/t2.txt 2>&1; tail -5 gpurun_out/r4/t2.txt
timeout 900 python tools/e2e_bench.py $((1<<25)) --check-slabs 8 --out gpurun_out/r4/e2e_25b.json > gpurun_out/r4/e2e_25b.log 2>&1; tail -c 2500 gpurun_out/r4/e2e_25b.log
timeout 1500 python tools/e2e_bench.py $((1<<27)) --check-slabs 16 --out gpurun_out/r4/e2e_27.json > gpurun_out/r4/e2e_27.log 2>&1; tail -c 2500 gpurun_out/r4/e2e_27.log
