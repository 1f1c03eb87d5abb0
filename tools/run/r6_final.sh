#!/bin/bash
# round 6, the final build: the GPU suite, `python bench.py` as the driver runs it (full size; the input stays cached in the work
# directory), then on the cached 57 GB file: one context with both feeds, shares on the one device, the host feed alone, two
# replicas on the one device, and the front end's kernel trace (as it runs, and with every launch alone on the device)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests_final.txt 2>&1; grep -E 'passed|failed' $O/gpu_tests_final.txt | tail -2
timeout 2400 python bench.py > $O/bench_default_full_size.json 2> $O/bench_default_full_size.err
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/r6u/bench_default_full_size.json').read().strip().splitlines()[-1]); e = j['end_to_end']
    print('bench: value', j['value'], 'ms', j['ms_per_step'], 'frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'], '| e2e reads', e['reads'], 'extract_s', e.get('extract_s'), 'first', e.get('first_run_wall_s'), 'call', e.get('call_s'), 'merge', e.get('merge_s'), 'x+c', e.get('extract_plus_call_s'), 'check', (e.get('check') or {}).get('ok'), (e.get('check') or {}).get('slabs'), 'vs cpu', e.get('vs_cpu_baseline_e2e_wall'), e.get('vs_cpu_baseline_e2e_extract_plus_call'))
    for r in e['runs']: print('  run', r['wall_s'], r['loop_s'], r['outside_the_loop'][:330])
except Exception as ex:
    print('bench failed', ex); print(open('gpurun_out/r6u/bench_default_full_size.err').read()[-1500:])
PY
B=$(ls /tmp/e2e_268435456_6.bam /dev/shm/e2e_268435456_6.bam 2>/dev/null | head -1); S=${B%.bam}.str
if [ -n "$B" ]; then
  D=$(dirname $B)
  {
  for rep in 1 2; do
    for how in mmap pread; do
      sleep 5; echo "== extract, one context, STRL_FEED=$how, run $rep"
      ( time STRL_FEED=$how STRL_FRONT_TIMING=1 STRL_BIN_TIMING=1 timeout 300 $CLI extract -v -g $S $B $D/x_$how.bin ) 2>&1 | grep -E 'seconds: total|seconds before|process:|real|device front end, ms|\.bin:' | cut -c1-600
    done
  done
  cmp $D/x_mmap.bin $D/x_pread.bin && echo ".bin identical (mmap feed, pread feed)"
  for how in device host device; do
    sleep 5; echo "== call, STRL_CALL_FRAG=$how"
    ( time STRL_CALL_FRAG=$how STRL_FRAG_TIMING=1 STRL_CLUSTER_TIMING=1 STRL_BIN_TIMING=1 timeout 300 $CLI call -v -o $D/c_$how $B $D/x_mmap.bin ) 2>&1 | grep -E 'seconds:|real|strl_bin_read|cluster_collect\]|strl_cluster\]|on the host|fragment lengths\]' | cut -c1-700
  done
  for f in bounds genotype unplaced; do cmp $D/c_device-$f.txt $D/c_host-$f.txt && echo "call $f identical (sample on the device / on the host)"; done
  for rep in 1 2; do sleep 5; echo "== merge, run $rep"; ( time STRL_CLUSTER_TIMING=1 STRL_BIN_TIMING=1 timeout 300 $CLI merge -v -o $D/m $D/x_mmap.bin ) 2>&1 | grep -E 'seconds:|real|strl_bin_read' | cut -c1-400; done
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
  cat $O/full_size_feed_and_shares.log | grep -E '^==|real|feed only|identical|seconds before|seconds: device context' | cut -c1-420
  rm -f $D/x_g*.bin $D/f.bin
  # two per-sample replicas on the one device (the leg bench.py runs at N > 1 with a device each)
  python - > $O/replicas_two_on_one_device.json 2> $O/replicas.err <<PY
import sys, json; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
from strling_amd import build
inp = e2e_bench.make_input(268435456, d='$D')
inp['out'] = '$D/x_mmap.bin'
print(json.dumps(e2e_bench.replicas(inp, build.CLI, 2, 1)))
PY
  cut -c1-700 $O/replicas_two_on_one_device.json
  # kernel trace of the front end on the full-size file
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/e2e_kt -o run -- $CLI extract -g $S $B $D/p1.bin > $R/$O/e2e_kt.log 2>&1
  f=$(find $R/$O/e2e_kt -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/e2e_kernel_stats.csv
  STRL_FRONT_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/e2e_kt_serial -o run -- $CLI extract -v -g $S $B $D/p2.bin > $R/$O/e2e_kt_serial.log 2>&1
  f=$(find $R/$O/e2e_kt_serial -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/e2e_kernel_stats_serial.csv
  cmp $D/p1.bin $D/p2.bin && echo "serial .bin identical" >> $R/$O/e2e_kt_serial.log
  find $R/$O -name 'run_kernel_trace.csv' -delete; find $R/$O -name '*agent_info*' -delete; rm -rf $R/$O/e2e_kt $R/$O/e2e_kt_serial
  cd $R; head -6 $O/e2e_kernel_stats_serial.csv | cut -c1-140
  rm -f $D/p1.bin $D/p2.bin $D/x_*.bin
fi
cd $R
timeout 400 python tests/fuzz/fuzz_call.py 300 99 > $O/fuzz_call.log 2>&1; tail -1 $O/fuzz_call.log | cut -c1-300
timeout 300 python tests/fuzz/fuzz_parity.py 180 707 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log | cut -c1-300
