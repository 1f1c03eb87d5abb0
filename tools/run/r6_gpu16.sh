#!/bin/bash
# `strling call` on the full-size file by the size of its region batches (= of its page-locked buffers: 4 sets of 1.25 x the batch):
# wall against the process' own clock at its last line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O
CLI=$R/strling_amd/lib/strling
python - > $O/make_full.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
e2e_bench.make_input(268435456, d='/tmp')
PY
B=$(ls /tmp/e2e_268435456_6.bam /dev/shm/e2e_268435456_6.bam 2>/dev/null | head -1); S=${B%.bam}.str; D=$(dirname $B)
timeout 300 $CLI extract -g $S $B $D/x.bin > /dev/null 2>&1
{
for rep in 1 2 3; do
  for mb in 384 256 192 128; do
    sleep 4
    s=$(date +%s.%N)
    STRL_CALL_BATCH_MB=$mb timeout 300 $CLI call -v -o $D/c_$mb $B $D/x.bin > /tmp/call.err 2>&1
    e=$(date +%s.%N)
    python3 - $mb $s $e <<'PY'
import re, sys
t = open('/tmp/call.err').read()
m = re.findall(r'since the start ([0-9.]+)', t)
ev = re.findall(r'evidence \+ genotypes of \d+ bounds on \d+ threads ([0-9.]+)', t)
print("batch %4s MB: whole process %.3f s, since the start at the last line %s s, evidence %s s" % (sys.argv[1], float(sys.argv[3]) - float(sys.argv[2]), m[-1] if m else '?', ev[-1] if ev else '?'))
PY
  done
done
for mb in 256 192 128; do for f in bounds genotype unplaced; do cmp $D/c_384-$f.txt $D/c_$mb-$f.txt || echo "DIFFERENT: $mb $f"; done; done; echo "outputs compared"
} > $O/call_batch_mb.log 2>&1
cat $O/call_batch_mb.log
rm -f $D/x.bin $D/c_*
