cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import sys, json
sys.path.insert(0, "tools")
import e2e_bench
inp = e2e_bench.make_input(8388608)
print(inp)
PY
BAM=/tmp/e2e_8388608.bam; BED=/tmp/e2e_8388608.str; OUT=/tmp/e2e_8388608.bin
CLI=strling_amd/lib/strling
for cb in 8192 8192 6144; do echo "== chunk blocks $cb"; python - $cb <<'PY'
import subprocess, sys, time, os
env = dict(os.environ, STRL_CHUNK_BLOCKS=sys.argv[1], STRL_FRONT_TIMING="1")
t = time.time()
r = subprocess.run(["strling_amd/lib/strling", "extract", "-v", "-g", "/tmp/e2e_8388608.str", "/tmp/e2e_8388608.bam", "/tmp/e2e_8388608.bin"], capture_output=True, text=True, env=env)
w = time.time() - t
print("\n".join(l for l in r.stderr.splitlines() if "seconds" in l or "device front" in l))
print("wall %.3f s  -> %.3g reads/s wall" % (w, 16777216 / w))
PY
done
mkdir -p gpurun_out/prof_e2e
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o e2e -- $GRAFT_REPO_ROOT/$CLI extract -g $BED $BAM $OUT > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_e2e/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:22]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), ("%.3f ms" % (float(r["TotalDurationNs"]) / 1e6)).rjust(12), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(16), r["Percentage"])
PY
find gpurun_out/prof_e2e -name "*kernel_trace.csv" -delete; find gpurun_out/prof_e2e -name "*agent_info*" -delete
