#!/bin/bash
# One gpurun call: bench line + rocprofv3 kernel stats + PMC passes (each in its own run, never combined with tracing)
# + the FETCH_SIZE / WRITE_SIZE calibration for the current build.  Writes gpurun_out/prof_$1/; copy what is to be
# judged into profiles/<round>/.
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R && python bench.py --cache /tmp --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
run() { d=$1; shift; STEPS=${STEPS:-30} timeout 900 rocprofv3 "$@" --output-format csv -d $O/$d -o run -- python $R/tools/prof_run.py > $O/$d.log 2>&1; }
run kt --kernel-trace --stats
STRL_NO_OVERLAP=1 run kt_serial --kernel-trace --stats     # every launch alone on the device: comparable with bench.py's kernel_ms
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU
run pmc2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
# calibration of the traffic counters on classify's access shapes
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib $R/tools/calib_traffic.hip > $O/calib_build.log 2>&1
/tmp/calib > $O/calib_bytes.json 2>> $O/calib_build.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o run -- /tmp/calib > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o run -- /tmp/calib > /dev/null 2>&1
find $O -name '*agent_info*' -delete; find $O -name '*.log' -size +200k -delete
cd $R
python $R/tools/summarise_profiles.py $O
find $O -name 'run_counter_collection.csv' -delete; find $O -name 'run_kernel_trace.csv' -delete
ls -la $O
cut -c1-600 $O/bench.json
