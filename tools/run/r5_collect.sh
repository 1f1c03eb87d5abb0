#!/bin/bash
# round 5 evidence in one call: the bench step's rocprofv3 kernel stats + PMC passes + calibration; the front end's kernel stats as it runs (overlapped) and with
# STRL_FRONT_SERIAL=1 (every launch alone on the device: what each front-end kernel costs by itself); the inflate kernel alone
bash tools/collect_profiles.sh r5a > gpurun_out/r5_collect_a.log 2>&1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r5e
mkdir -p $O
cd $R && export TMPDIR=/tmp
N=33554432
python tools/e2e_bench.py $N --dir /tmp --check-slabs 8 --repeats 3 --keep --out $O/e2e_raw.json > $O/e2e_raw.log 2>&1
CLI=$R/strling_amd/lib/strling
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_kt -o run -- $CLI extract -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof.bin > $O/e2e_kt.log 2>&1
f=$(find $O/e2e_kt -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/e2e_kernel_stats.csv
STRL_FRONT_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_kt_serial -o run -- $CLI extract -v -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof2.bin > $O/e2e_kt_serial.log 2>&1
f=$(find $O/e2e_kt_serial -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/e2e_kernel_stats_serial.csv
cmp /tmp/e2e_prof.bin /tmp/e2e_prof2.bin && echo "serial .bin identical" >> $O/e2e_kt_serial.log
find $O -name 'run_kernel_trace.csv' -delete; find $O -name '*agent_info*' -delete
cd $R
python tools/inflate_bench.py 524288 32768 > $O/inflate.log 2>&1
tail -1 $O/inflate.log > $O/inflate.json
tail -5 gpurun_out/r5_collect_a.log | cut -c1-600; tail -3 $O/e2e_raw.log | cut -c1-800; cat $O/inflate.log | tail -3; head -14 $O/e2e_kernel_stats_serial.csv | cut -c1-160; tail -4 $O/e2e_kt_serial.log | cut -c1-400
