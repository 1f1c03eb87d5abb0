#!/bin/bash
# the translations dropped by a background thread of the feed: mapped feed against pread on the 1.3e8-read file, feed only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 600 python -m pytest tests/test_cli.py tests/test_front_device.py -q -m gpu -x > $O/cli_tests_9.txt 2>&1; tail -2 $O/cli_tests_9.txt | cut -c1-200
python - > $O/make_2p27.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
inp = e2e_bench.make_input(67108864, d='/dev/shm')
PY
B=/dev/shm/e2e_67108864_6
{
for rep in 1 2 3; do
  for how in mmap pread; do
    sleep 3; echo "== STRL_FEED=$how, run $rep"
    ( time STRL_FEED=$how timeout 300 $CLI extract -v -g $B.str $B.bam /dev/shm/x_$how.bin ) 2>&1 | grep -E 'seconds: total|process:|real' | cut -c1-330
  done
done
cmp /dev/shm/x_mmap.bin /dev/shm/x_pread.bin && echo ".bin identical"
for how in mmap pread mmap pread; do sleep 3; echo "== feed only, 8 shares, STRL_FEED=$how"; ( time STRL_FEED=$how STRL_FEED_ONLY=1 timeout 300 $CLI extract -v -g $B.str --gpus 8 $B.bam /dev/shm/f.bin ) 2>&1 | grep -E 'feed only|real'; done
} > $O/zapper_2p27.log 2>&1
cat $O/zapper_2p27.log
