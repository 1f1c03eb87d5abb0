#!/bin/bash
# round 6, fourth GPU call: why the per-read state of a whole genome takes 0.5 s to allocate since the bring-up moved onto threads
# (0.04 s in round 5): every large hipMalloc timed, on the bring-up thread and on the main thread, twice each, on a small file
# with the whole genome's read-count hint; then stage A without its k = 3, 4 recounts (ceiling of pooling them)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
CLI=$R/strling_amd/lib/strling
python - > /dev/null 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from strling_amd import synth, bamio
rec, g = synth.synth_wgs(20000, seed=5, contig_len=2_000_000)
bamio.write_bam('/tmp/st.bam', rec); bamio.write_genome_bed('/tmp/st.str', g, rec.targets)
PY
{
for rep in 1 2 3; do
  echo "== bring-up thread, run $rep"; STRL_ALLOC_TIMING=1 STRL_READS_HINT=649792981 $CLI extract -v -g /tmp/st.str /tmp/st.bam /tmp/a.bin 2>&1 | grep -E 'hipMalloc|seconds before'; sleep 4
  echo "== main thread, run $rep"; STRL_STATE_ON_MAIN=1 STRL_ALLOC_TIMING=1 STRL_READS_HINT=649792981 $CLI extract -v -g /tmp/st.str /tmp/st.bam /tmp/a.bin 2>&1 | grep -E 'hipMalloc|seconds before'; sleep 4
done
echo "== main thread, serial contexts"; STRL_SERIAL_CTX=1 STRL_STATE_ON_MAIN=1 STRL_ALLOC_TIMING=1 STRL_READS_HINT=649792981 $CLI extract -v -g /tmp/st.str /tmp/st.bam /tmp/a.bin 2>&1 | grep -E 'hipMalloc|seconds before'
} > $O/state_alloc_diag.log 2>&1
cat $O/state_alloc_diag.log | cut -c1-420
# stage A without the k = 3, 4 recounts
python bench.py --cache /tmp --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > $O/stage_a_default.json 2>/dev/null
STRL_LIB=$R/strling_amd/lib/libstrling_amd_norc34.so python bench.py --cache /tmp --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > $O/stage_a_norc34.json 2> $O/stage_a_norc34.err
python - <<'PY'
import json
for f in ('stage_a_default', 'stage_a_norc34'):
    try:
        j = json.loads(open(f'gpurun_out/r6/{f}.json').read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['roofline']['kernel_ms']['score_kernel<whole,A>'], j['roofline']['kernel_ms']['score_kernel<segment,A+B>'])
    except Exception as e:
        print(f, 'failed', e)
PY
