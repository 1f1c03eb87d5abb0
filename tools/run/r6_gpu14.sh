#!/bin/bash
# side streams made at first use, the front end's streams beside its allocations, one table copy: the suite, the bench step
# (which overlaps batches on the side streams), the context's laps on a small file, wall on the 1.3e8-read file
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6r; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 900 python -m pytest tests/test_cli.py tests/test_front_device.py tests/test_call.py tests/test_multi_device.py -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E 'passed|failed' $O/gpu_tests.txt | tail -2
python - > $O/make.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
e2e_bench.make_input(4194304, d='/dev/shm')
e2e_bench.make_input(67108864, d='/dev/shm')
PY
B=$(ls /dev/shm/e2e_4194304_*.bam | head -1); S=${B%.bam}.str
{
for rep in 1 2 3; do sleep 2; echo "== extract, 4.2e6 pairs, run $rep"; ( time STRL_CTX_TIMING=1 timeout 120 $CLI extract -v -g $S $B /dev/shm/x.bin ) 2>&1 | grep -E 'strl_ctx_create|seconds before|real' | cut -c1-400; done
for rep in 1 2 3; do sleep 2; echo "== merge, run $rep"; ( time timeout 120 $CLI merge -v -o /dev/shm/m /dev/shm/x.bin ) 2>&1 | grep -E 'seconds|real' | cut -c1-400; done
B=$(ls /dev/shm/e2e_67108864_*.bam | head -1); S=${B%.bam}.str
for rep in 1 2 3 4; do sleep 3; echo "== extract, 6.7e7 pairs, run $rep"; ( time timeout 300 $CLI extract -v -g $S $B /dev/shm/y.bin ) 2>&1 | grep -E 'seconds before|process:|real' | cut -c1-500; done
for rep in 1 2 3; do sleep 3; echo "== call, run $rep"; ( time timeout 300 $CLI call -v -o /dev/shm/c $B /dev/shm/y.bin ) 2>&1 | grep -E 'seconds:|real' | cut -c1-330; done
} > $O/ctx_laps_lazy_streams.log 2>&1
cat $O/ctx_laps_lazy_streams.log
