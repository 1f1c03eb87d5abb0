# quick PMC look at the scorer kernels of the current build (two counter passes, a few steps)     usage: pmc.sh TAG [kernel substring]
TAG=${1:-q}; SUB=${2:-score_kernel<10, 64, 0, 0}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd $R && python bench.py --cache /tmp --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
run() { d=$1; shift; STEPS=4 timeout 600 rocprofv3 "$@" --output-format csv -d $O/$d -o run -- python $R/tools/prof_run.py > $O/$d.log 2>&1; }
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU
run pmc2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
cd $R && python tools/summarise_profiles.py $O > /dev/null 2>&1
find $O -name 'run_counter_collection.csv' -delete; find $O -name '*agent_info*' -delete
python - <<PY
import json
d=json.load(open("$O/pmc_counters.json"))
for k,v in d.items():
    if "$SUB" in k or "score_kernel<10, 64, 1, 0" in k:
        print(k); print({a:b for a,b in v.items() if a[0]!='_'})
PY
