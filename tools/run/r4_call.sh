#!/bin/bash
# `strling call` with the evidence reads on the device against the host reader, on the e2e file of $1 pairs (default 2^26)
N=${1:-67108864}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
python tools/e2e_bench.py $N --dir /tmp --check-slabs 4 --repeats 1 --keep --out $O/e2e_b.json > $O/e2e_b.log 2>&1
tail -2 $O/e2e_b.log | cut -c1-300
CLI=$R/strling_amd/lib/strling
B=/tmp/e2e_${N}_6
for k in 1 2; do
  sleep 3
  ( time STRL_CALL_REGIONS=host $CLI call -v -o /tmp/hostcall $B.bam $B.bin ) 2>&1 | grep "seconds:\|real" >> $O/call_host.txt
  sleep 3
  ( time $CLI call -v -o /tmp/devcall $B.bam $B.bin ) 2>&1 | grep "seconds:\|real" >> $O/call_dev.txt
done
cmp /tmp/hostcall-bounds.txt /tmp/devcall-bounds.txt && cmp /tmp/hostcall-genotype.txt /tmp/devcall-genotype.txt && cmp /tmp/hostcall-unplaced.txt /tmp/devcall-unplaced.txt && echo SAME >> $O/call_dev.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/call_kt -o run -- $CLI call -o /tmp/profcall $B.bam $B.bin > $O/call_kt.log 2>&1
f=$(find $O/call_kt -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/call_kernel_stats.csv
rm -rf $O/call_kt
