#!/bin/bash
# `strling call` on the e2e file of $1 pairs: batch size of the device evidence reads against wall time
N=${1:-67108864}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
python tools/e2e_bench.py $N --dir /tmp --check-slabs 0 --repeats 1 --keep --out $O/e2e_c.json > $O/e2e_c.log 2>&1
CLI=$R/strling_amd/lib/strling
B=/tmp/e2e_${N}_6
TIMEFORMAT="real %R s"
for mb in 768 384 256 128 768 384 256 128; do
  sleep 3
  echo "batch $mb MB" >> $O/call_batch.txt
  { time STRL_CALL_BATCH_MB=$mb $CLI call -v -o /tmp/devcall $B.bam $B.bin ; } 2>&1 | grep "seconds:\|real" | sed 's/.*evidence + genotypes/evidence/' | cut -c1-420 >> $O/call_batch.txt
done
sleep 3
echo host >> $O/call_batch.txt
{ time STRL_CALL_REGIONS=host $CLI call -v -o /tmp/hostcall $B.bam $B.bin ; } 2>&1 | grep "real" >> $O/call_batch.txt
sleep 3
echo merge >> $O/call_batch.txt
{ time $CLI merge -v -o /tmp/mrg $B.bin ; } 2>&1 | grep "seconds:\|real" >> $O/call_batch.txt
exit 0
