#!/bin/bash
# One gpurun call for the front-end evidence: strling extract -> call / merge end to end on a level-6 file (three extract runs), rocprofv3 kernel stats of one CLI run,
# the inflate kernel alone on two inputs, its PMC pass.  Writes gpurun_out/prof_$1/.
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R && export TMPDIR=/tmp
N=${PAIRS:-33554432}
python tools/e2e_bench.py $N --dir /tmp --check-slabs 8 --repeats 3 --keep --out $O/e2e_raw.json > $O/e2e_raw.log 2>&1
CLI=$R/strling_amd/lib/strling
[ -x $CLI ] || CLI=$(python -c "from strling_amd import build; print(build.CLI)")
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_kt -o run -- $CLI extract -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof.bin > $O/e2e_kt.log 2>&1
f=$(find $O/e2e_kt -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/e2e_kernel_stats.csv
find $O -name 'run_kernel_trace.csv' -delete; find $O -name '*agent_info*' -delete
cd $R
python tools/inflate_bench.py 524288 32768 > $O/inflate.log 2>&1
tail -1 $O/inflate.log > $O/inflate.json
bash tools/prof_inflate.sh 524288 32768 > $O/inflate_pmc.txt 2>&1
tail -3 $O/e2e_raw.log | cut -c1-1500; cat $O/inflate.log | tail -4; head -12 $O/e2e_kernel_stats.csv | cut -c1-150
