#!/bin/bash
# what the exit of a process costs by what it holds: device memory, page-locked memory, both (tools/ubench/exit_probe.hip);
# the laps of strl_ctx_create inside `strling extract`
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
python - > $O/exit_probe.log 2>&1 <<'PY'
import subprocess, time
P = 'tools/ubench/bin/exit_probe'
for rep in range(3):
    for cfg in (["0", "0"], ["70", "0"], ["0", "1300"], ["0", "2600"], ["70", "1300"], ["70", "1300", "unpin"], ["4", "2300"], ["4", "2300", "unpin"]):
        time.sleep(2)
        t0 = time.time()
        r = subprocess.run([P] + cfg, capture_output=True, text=True, timeout=120)
        w = time.time() - t0
        print(r.stdout.strip(), "| whole process %.3f s" % w, flush=True)
PY
cat $O/exit_probe.log
python - > $O/make_small.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
inp = e2e_bench.make_input(4194304, d='/dev/shm')
PY
B=$(ls /dev/shm/e2e_4194304_*.bam | head -1); S=${B%.bam}.str
{
for rep in 1 2 3 4; do sleep 2; echo "== extract, 4.2e6 pairs, run $rep"; ( time STRL_CTX_TIMING=1 timeout 120 strling_amd/lib/strling extract -v -g $S $B /dev/shm/x.bin ) 2>&1 | grep -E 'strl_ctx_create|seconds before|process:|real' | cut -c1-500; done
} > $O/ctx_laps.log 2>&1
cat $O/ctx_laps.log
