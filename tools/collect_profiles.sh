#!/bin/bash
# One gpurun call: bench line + rocprofv3 kernel stats + PMC passes (each in its own run, never combined with tracing)
# for the current build.  Writes gpurun_out/prof_$1/; tools/summarise_profiles.py turns that into profiles/<round>/<tag>_*.
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
run() { d=$1; shift; timeout 600 rocprofv3 "$@" --output-format csv -d $O/$d -o run -- python $R/tools/prof_run.py > $O/$d.log 2>&1; }
run kt --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU
run pmc2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/idx -o run -- python $R/tests/fuzz/index_bench.py 100000000 > $O/index_bench.json 2> $O/idx.log
find $O -name '*agent_info*' -delete; find $O -name '*.log' -size +200k -delete
# keep only the per-kernel averages of the big counter CSVs
python $R/tools/summarise_profiles.py $O
find $O -name 'run_counter_collection.csv' -delete; find $O -name 'run_kernel_trace.csv' -delete
ls -la $O
cut -c1-400 $O/bench.json
