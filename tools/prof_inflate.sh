#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/ibp -o run -- python $R/tools/inflate_bench.py ${1:-524288} ${2:-32768} > /tmp/ibp.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d /tmp/ibq -o run -- python $R/tools/inflate_bench.py ${1:-524288} ${2:-32768} > /tmp/ibq.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("/tmp/ibp", "/tmp/ibq"):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/run_counter_collection.csv", recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "inflate_" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        # (two inputs, three launches each: the first three dispatches are the level-1 blocks, the last three the level-6 ones)
        disps = sorted({int(d) for d, _ in per})
        for (disp, c), v in per.items():
            acc[("L1 " if disps.index(int(disp)) < len(disps) // 2 else "L6 ") + c].append(v)
    for c, v in sorted(acc.items()):
        print(c.ljust(24), "%.4g" % (sum(v) / len(v)))
PY
