#!/bin/bash
# round 6, second GPU call: long-read tests, the suite, the side-by-side bring-up under stress, the 1.3e8-read file with 1 / 4 / 8
# contexts (bring-up lines), the feed probe, process start
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 900 python -m pytest tests/test_long_reads.py -x -q -m gpu > $O/long_reads.txt 2>&1; tail -4 $O/long_reads.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests_2.txt 2>&1; tail -12 $O/gpu_tests_2.txt
# -- stress: 40 x `extract --gpus 8` (contexts side by side) on a small file, every process under a timeout
python - > $O/stress_setup.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
from strling_amd import synth, bamio
rec, g = synth.synth_wgs(20000, seed=5, contig_len=2_000_000)
bamio.write_bam('/tmp/st.bam', rec); bamio.write_genome_bed('/tmp/st.str', g, rec.targets)
PY
$CLI extract -g /tmp/st.str /tmp/st.bam /tmp/st1.bin 2> /dev/null
hang=0; diff=0
for i in $(seq 1 40); do
  timeout 60 $CLI extract -g /tmp/st.str --gpus 8 /tmp/st.bam /tmp/st8.bin 2> /tmp/st8.err; rc=$?
  [ $rc -ne 0 ] && { hang=$((hang+1)); echo "run $i rc $rc"; tail -3 /tmp/st8.err; }
  cmp -s /tmp/st1.bin /tmp/st8.bin || diff=$((diff+1))
done > $O/stress_parallel_ctx.log 2>&1
echo "stress: 40 runs of extract --gpus 8 (bring-up side by side): $hang failed or timed out, $diff .bin differ" | tee -a $O/stress_parallel_ctx.log
# -- process start
( time $CLI extract > /dev/null ) 2> $O/process_start.log; ( time $CLI extract > /dev/null ) 2>> $O/process_start.log; cat $O/process_start.log | grep real
# -- the 1.3e8-read file
python tools/e2e_bench.py 67108864 --dir /dev/shm --keep --check-slabs 4 --repeats 2 --out $O/e2e_2p27.json > $O/e2e_2p27.log 2>&1; tail -c 1500 $O/e2e_2p27.log
B=/dev/shm/e2e_67108864_6
for g in 1 4 8; do
  STRL_BIN_TIMING=1 timeout 300 $CLI extract -v -g $B.str --gpus $g $B.bam /dev/shm/g$g.bin 2> $O/shares_g$g.log; echo "g$g rc $?"
  grep -E 'seconds before the loop|seconds: total|\.bin:' $O/shares_g$g.log | cut -c1-700
done
cmp /dev/shm/g1.bin /dev/shm/g4.bin && cmp /dev/shm/g1.bin /dev/shm/g8.bin && echo "g1 g4 g8 .bin identical"
STRL_SERIAL_CTX=1 timeout 300 $CLI extract -v -g $B.str --gpus 8 $B.bam /dev/shm/g8s.bin 2> $O/shares_g8_serial.log; grep -E 'seconds before the loop' $O/shares_g8_serial.log | cut -c1-700
for g in 4 8; do ( time STRL_FEED_ONLY=1 timeout 300 $CLI extract -v -g $B.str --gpus $g $B.bam /dev/shm/x.bin ) 2> $O/feed_only_g$g.log; grep -E 'feed only|real' $O/feed_only_g$g.log; done
# -- the feed probe: page cache (shm) and a disk file
timeout 600 tools/ubench/bin/feed_probe $B.bam 320 14 12 > $O/feed_probe_shm.log 2>&1; cat $O/feed_probe_shm.log
head -c 6000000000 $B.bam > /tmp/part.bam; sync
timeout 600 tools/ubench/bin/feed_probe /tmp/part.bam 320 6 12 > $O/feed_probe_tmp.log 2>&1; cat $O/feed_probe_tmp.log
rm -f /dev/shm/*.bin /tmp/part.bam $B.bam $B.bam.bai $B.str
