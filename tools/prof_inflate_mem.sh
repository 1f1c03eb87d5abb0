#!/bin/bash
# WARNING: the first pass (five TCC_* counters at once) hung on the GPU box for the whole timeout in round 5; split the counters before running this again.
# GPU box: memory-side counters of the inflate kernel (L2 hits / misses, fabric requests = FETCH_SIZE / WRITE_SIZE) for one form at one launch size.
# usage: STRL_INFLATE_FORM=wave|group bash tools/prof_inflate_mem.sh [n_pairs] [min_blocks]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ibm1 /tmp/ibm2
STRL_BENCH_NOCHECK=1 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/ibm1 -o run -- python $R/tools/inflate_bench.py ${1:-262144} ${2:-16384} > /tmp/ibm1.log 2>&1
STRL_BENCH_NOCHECK=1 timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_BUSY_CYCLES --output-format csv -d /tmp/ibm2 -o run -- python $R/tools/inflate_bench.py ${1:-262144} ${2:-16384} > /tmp/ibm2.log 2>&1
grep binned /tmp/ibm1.log
python - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("/tmp/ibm1", "/tmp/ibm2"):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/run_counter_collection.csv", recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "inflate_" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        disps = sorted({int(d) for d, _ in per})
        for (disp, c), v in per.items():
            acc[("L1 " if disps.index(int(disp)) < len(disps) // 2 else "L6 ") + c].append(v)
    for c, v in sorted(acc.items()):
        print(c.ljust(28), "%.4g" % (sum(v) / len(v)))
PY
