cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import sys, json
sys.path.insert(0, "tools")
import e2e_bench
inp = e2e_bench.make_input(8388608)
json.dump(inp, open("/tmp/e2e_inp.json", "w"))
print(inp)
PY
BAM=/tmp/e2e_8388608.bam; BED=/tmp/e2e_8388608.str; OUT=/tmp/e2e_8388608.bin
CLI=strling_amd/lib/strling
for i in 1 2; do STRL_FRONT_TIMING=1 $CLI extract -v -g $BED $BAM $OUT 2>&1 | grep -v "reads/sec" | tail -8; done
mkdir -p gpurun_out/prof_e2e
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o e2e -- $GRAFT_REPO_ROOT/$CLI extract -g $BED $BAM $OUT > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_e2e -name "*kernel_stats*" | head
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_e2e/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:25]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.3f ms total" % (float(r["TotalDurationNs"]) / 1e6)).rjust(18), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(16), r["Percentage"])
PY
