#!/bin/bash
# the unused rings un-registered beside the .bin write instead of behind the process' last line: whole process and its own clock,
# with and without (STRL_NO_EARLY_UNPIN=1), on the full-size file
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6x; mkdir -p $O
CLI=$R/strling_amd/lib/strling
timeout 600 python -m pytest tests/test_cli.py tests/test_front_device.py tests/test_multi_device.py -q -m gpu -x > $O/gpu_tests.txt 2>&1; grep -E 'passed|failed' $O/gpu_tests.txt | tail -1
python - > $O/make_full.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
e2e_bench.make_input(268435456, d='/tmp')
PY
B=$(ls /tmp/e2e_268435456_6.bam /dev/shm/e2e_268435456_6.bam 2>/dev/null | head -1); S=${B%.bam}.str; D=$(dirname $B)
{
for rep in 1 2 3 4 5; do
  for how in A=1 STRL_NO_EARLY_UNPIN=1; do
    sleep 6
    s=$(date +%s.%N)
    env $how timeout 300 $CLI extract -v -g $S $B $D/x_$how.bin > /tmp/x.err 2>&1
    e=$(date +%s.%N)
    python3 - "$how" $s $e <<'PY'
import re, sys
t = open('/tmp/x.err').read()
m = re.findall(r'now ([0-9.]+) s after exec', t)
w = re.findall(r'writing the .bin ([0-9.]+)', t)
st = re.findall(r'front-end buffers ([0-9.]+)', t)
print("%-22s whole process %.3f s, its own clock at the last line %s s, .bin %s s, state alloc %s s" % (sys.argv[1], float(sys.argv[3]) - float(sys.argv[2]), m[-1] if m else '?', w[-1] if w else '?', st[-1] if st else '?'))
PY
  done
done
cmp "$D/x_A=1.bin" "$D/x_STRL_NO_EARLY_UNPIN=1.bin" && echo ".bin identical"
} > $O/early_unpin_full_size.log 2>&1
cat $O/early_unpin_full_size.log
rm -f $D/x_*.bin
