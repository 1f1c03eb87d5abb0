#!/bin/bash
# the laps of strl_ctx_create inside `strling extract` (small file)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
python - > $O/make_small.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
inp = e2e_bench.make_input(4194304, d='/dev/shm')
PY
B=$(ls /dev/shm/e2e_4194304_*.bam | head -1); S=${B%.bam}.str
{
for rep in 1 2 3; do sleep 2; echo "== extract, 4.2e6 pairs, run $rep"; ( time STRL_CTX_TIMING=1 STRL_ALLOC_TIMING=1 timeout 120 strling_amd/lib/strling extract -v -g $S $B /dev/shm/x.bin ) 2>&1 | grep -E 'strl_ctx_create|seconds before|real' | cut -c1-400; done
for rep in 1 2; do sleep 2; echo "== merge, run $rep"; ( time STRL_CTX_TIMING=1 timeout 120 strling_amd/lib/strling merge -v -o /dev/shm/m /dev/shm/x.bin ) 2>&1 | grep -E 'strl_ctx_create|seconds|real' | cut -c1-400; done
} > $O/ctx_laps.log 2>&1
cat $O/ctx_laps.log
