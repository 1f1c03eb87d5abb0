cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=$(date +%s)
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
E=$(date +%s)
echo "bench.py default run: $((E-S)) s, rc $?"
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_default.json") if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "traffic") if k in d["roofline"]})
print("cpu_baseline", d["cpu_baseline"]["value"], "| nproc", d["cpu_baseline_nproc"], "| e2e", d["cpu_baseline_e2e"])
e = d["end_to_end"]
print("end_to_end", {k: e.get(k) for k in ("value", "vs_cpu_baseline_e2e_wall", "vs_cpu_baseline_e2e_loop")}, e["runs"][0]["reads_per_s_loop"], e["runs"][0]["device_front_end"])
PY
