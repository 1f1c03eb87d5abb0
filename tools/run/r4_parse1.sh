#!/bin/bash
# round 4: rec_parse with the wave-cooperative SEQ copy -- byte identity tests, then the 6.7e7-read e2e with kernel stats
mkdir -p gpurun_out/r4
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_front_device.py tests/test_cli.py tests/test_bgzf_device.py -m gpu -x -q > gpurun_out/r4/parse1_tests.log 2>&1
tail -5 gpurun_out/r4/parse1_tests.log
N=33554432
timeout 900 python tools/e2e_bench.py $N --dir /tmp --repeats 3 --check-slabs 4 --keep --out gpurun_out/r4/parse1_e2e.json > gpurun_out/r4/parse1_e2e.log 2>&1
tail -c 1500 gpurun_out/r4/parse1_e2e.log
CLI=$R/strling_amd/lib/strling
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/parse1_kt -o run -- $CLI extract -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof.bin > $R/gpurun_out/r4/parse1_kt.log 2>&1
f=$(find $R/gpurun_out/r4/parse1_kt -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r4/parse1_kernel_stats.csv
find $R/gpurun_out/r4/parse1_kt -name 'run_kernel_trace.csv' -delete; find $R/gpurun_out/r4/parse1_kt -name '*agent_info*' -delete
head -8 $R/gpurun_out/r4/parse1_kernel_stats.csv | cut -c1-150
