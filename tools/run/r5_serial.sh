#!/bin/bash
# round 5: the front end's kernels one at a time (STRL_FRONT_SERIAL=1: inflate, CRC, record scan, parse, scorer of every chunk on ONE stream)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r5e
mkdir -p $O
cd $R && export TMPDIR=/tmp
N=33554432
python tools/e2e_bench.py $N --dir /tmp --check-slabs 0 --repeats 1 --keep --out $O/e2e_raw2.json > $O/e2e_raw2.log 2>&1
CLI=$R/strling_amd/lib/strling
cd /tmp
$CLI extract -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof.bin > /dev/null 2>&1
STRL_FRONT_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_kt_serial -o run -- $CLI extract -v -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof2.bin > $O/e2e_kt_serial.log 2>&1
f=$(find $O/e2e_kt_serial -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/e2e_kernel_stats_serial.csv
cmp /tmp/e2e_prof.bin /tmp/e2e_prof2.bin && echo "serial .bin identical" >> $O/e2e_kt_serial.log
find $O -name 'run_kernel_trace.csv' -delete; find $O -name '*agent_info*' -delete
head -16 $O/e2e_kernel_stats_serial.csv | cut -c1-70,140-250; grep 'seconds: total\|serial' $O/e2e_kt_serial.log | cut -c1-300
