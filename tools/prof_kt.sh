#!/bin/bash
# kernel-trace profile of the bench step: gpurun_out/prof_$1/kernel_stats.txt  (usage: tools/prof_kt.sh TAG [STEPS])
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R && python bench.py --cache /tmp --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
STEPS=${2:-20} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $R/tools/prof_run.py > $O/kt.log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$O/kt/**/run_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("$O/kernel_stats.txt", "w") as out:
        for r in rows[:60]:
            line = f'{r["Name"][:100].ljust(100)} calls {r["Calls"]:>6} total_ns {r["TotalDurationNs"]:>12} avg_ns {r["AverageNs"]:>12} pct {r["Percentage"]}'
            print(line); out.write(line + "\n")
PY
find $O -name "run_kernel_trace.csv" -delete; find $O -name "*agent_info*" -delete
python -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step']); print(d['roofline']['kernel_ms']); print(d['roofline']['group_ms'])"
