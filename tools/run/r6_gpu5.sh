#!/bin/bash
# round 6, fifth GPU call: the suite, `python bench.py` as the driver runs it (full size; the input stays cached), then on the cached
# 57 GB file: the feed by pread against the mapping's memcpy, shares on the one device, the host feed alone; the fuzzers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests_5.txt 2>&1; grep -E 'passed|failed' $O/gpu_tests_5.txt | tail -2
timeout 2400 python bench.py > $O/bench_default_full_size.json 2> $O/bench_default_full_size.err
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/r6/bench_default_full_size.json').read().strip().splitlines()[-1]); e = j['end_to_end']
    print('bench: value', j['value'], 'ms', j['ms_per_step'], 'frac', j['roofline']['frac'], '| e2e reads', e['reads'], 'extract_s', e.get('extract_s'), 'first', e.get('first_run_wall_s'), 'call', e.get('call_s'), 'merge', e.get('merge_s'), 'x+c', e.get('extract_plus_call_s'), 'check', (e.get('check') or {}).get('ok'), (e.get('check') or {}).get('slabs'), 'vs cpu', e.get('vs_cpu_baseline_e2e_wall'), e.get('vs_cpu_baseline_e2e_extract_plus_call'))
    for r in e['runs']: print('  run', r['wall_s'], r['loop_s'], r['outside_the_loop'][:330])
except Exception as ex:
    print('bench failed', ex); print(open('gpurun_out/r6/bench_default_full_size.err').read()[-1500:])
PY
B=$(ls /tmp/e2e_268435456_6.bam /dev/shm/e2e_268435456_6.bam 2>/dev/null | head -1); S=${B%.bam}.str
if [ -n "$B" ]; then
  D=$(dirname $B)
  {
  for rep in 1 2; do
    for how in mmap pread; do
      sleep 5; echo "== extract, one context, STRL_FEED=$how, run $rep"
      ( time STRL_FEED=$how STRL_FRONT_TIMING=1 timeout 300 $CLI extract -v -g $S $B $D/x_$how.bin ) 2>&1 | grep -E 'seconds: total|seconds before|process:|real|device front end, ms' | cut -c1-600
    done
  done
  cmp $D/x_mmap.bin $D/x_pread.bin && echo ".bin identical (mmap feed, pread feed)"
  for g in 4 8; do
    for how in mmap pread; do
      sleep 5; echo "== feed only, $g shares, STRL_FEED=$how"
      ( time STRL_FEED=$how STRL_FEED_ONLY=1 timeout 300 $CLI extract -v -g $S --gpus $g $B $D/f.bin ) 2>&1 | grep -E 'feed only|real'
    done
  done
  for g in 4 8; do
    sleep 5; echo "== extract --gpus $g on the one device"
    ( time timeout 400 $CLI extract -v -g $S --gpus $g $B $D/x_g$g.bin ) 2>&1 | grep -E 'seconds: total|seconds before|process:|real|gathered|share [0-9]' | cut -c1-500
    cmp $D/x_mmap.bin $D/x_g$g.bin && echo ".bin identical (--gpus $g)"
  done
  } > $O/full_size_feed_and_shares.log 2>&1
  cat $O/full_size_feed_and_shares.log | cut -c1-420
  rm -f $D/x_*.bin $D/f.bin
fi
timeout 400 python tests/fuzz/fuzz_parity.py 240 606 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log | cut -c1-300
timeout 500 python tests/fuzz/fuzz_call.py 300 88 > $O/fuzz_call.log 2>&1; tail -1 $O/fuzz_call.log | cut -c1-300
timeout 300 python tests/fuzz/fuzz_inflate.py 120 > $O/fuzz_inflate.log 2>&1; tail -1 $O/fuzz_inflate.log | cut -c1-300
