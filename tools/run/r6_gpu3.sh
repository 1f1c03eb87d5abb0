#!/bin/bash
# round 6, third GPU call: the suite again, the N > 1 bench legs on one device (dry run), the inflate's concurrency sweep, the feed
# probe with the user-space copy, the bench step's profiles (kernel stats + PMC) for this build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests_3.txt 2>&1; tail -4 $O/gpu_tests_3.txt | cut -c1-300
# -- N > 1 legs of bench.py on the one device: two ranks over gloo, the file leg with --gpus 2, feed only, two replicas (--device k)
timeout 1500 python bench.py --gpus 2 --steps 3 --warmup 1 --reads-per-gpu 1048576 --no-cpu-baseline --e2e-pairs 16777216 > $O/bench_gpus2_one_device.json 2> $O/bench_gpus2_one_device.err
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/r6/bench_gpus2_one_device.json').read().strip().splitlines()[-1]); e = j['end_to_end']
    print('gpus2 dry run: n_gpus', j['n_gpus'], 'exchange', j['exchange'], 'e2e extract_s', e.get('extract_s'), 'loop', e.get('reads_per_s_loop'), 'feed', e.get('feed_only'), 'replicas', {k: v for k, v in (e.get('replicas') or {}).items() if k != 'what'}, 'check', (e.get('check') or {}).get('ok'), 'scaling', e.get('strong_scaling_vs_n1'), 'build', j.get('build'))
except Exception as ex:
    print('gpus2 dry run failed', ex); print(open('gpurun_out/r6/bench_gpus2_one_device.err').read()[-1500:])
PY
# -- inflate: is concurrency what it lacks?  (occupancy lowered by unused dynamic LDS; a build held to 8 waves per SIMD)
python tools/inflate_bench.py 524288 32768 > $O/inflate_default.log 2>&1; tail -1 $O/inflate_default.log > $O/inflate.json; tail -2 $O/inflate_default.log | cut -c1-400
for pad in 1024 2048 4096 8192; do echo "== STRL_INFLATE_LDS_PAD=$pad"; STRL_INFLATE_LDS_PAD=$pad python tools/inflate_bench.py 524288 32768 2>&1 | tail -1 | cut -c1-300; done > $O/inflate_concurrency.txt 2>&1
python tools/build_variant.py w8 bgzf.hip -DSTRL_INFLATE_WAVES=8 > /dev/null 2>&1 && { echo "== built for 8 waves per SIMD (64 registers)"; STRL_LIB=$R/strling_amd/lib/libstrling_amd_w8.so python tools/inflate_bench.py 524288 32768 2>&1 | tail -1 | cut -c1-300; } >> $O/inflate_concurrency.txt 2>&1
python tools/build_variant.py w5 bgzf.hip -DSTRL_INFLATE_WAVES=5 > /dev/null 2>&1 && { echo "== built for 5 waves per SIMD"; STRL_LIB=$R/strling_amd/lib/libstrling_amd_w5.so python tools/inflate_bench.py 524288 32768 2>&1 | tail -1 | cut -c1-300; } >> $O/inflate_concurrency.txt 2>&1
cat $O/inflate_concurrency.txt
# -- feed probe with the user-space copy, on the cached 3.4e7-read file the dry run left in the work directory
B=$(ls /tmp/e2e_16777216_6.bam /dev/shm/e2e_16777216_6.bam 2>/dev/null | head -1)
[ -n "$B" ] && { timeout 300 tools/ubench/bin/feed_probe $B 320 4 12 pread,mmapcopy > $O/feed_probe_mmapcopy.log 2>&1; timeout 300 tools/ubench/bin/feed_probe $B 320 4 24 pread,mmapcopy >> $O/feed_probe_mmapcopy.log 2>&1; cat $O/feed_probe_mmapcopy.log; }
# -- profiles of the bench step for this build
STEPS=20 bash tools/collect_profiles.sh r6 > $O/collect.log 2>&1; tail -3 $O/collect.log | cut -c1-400
