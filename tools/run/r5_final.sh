#!/bin/bash
# round 5, final build: the whole -m gpu suite, then the driver's own bench command
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 1500 python -u -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r5/final_tests.txt 2>&1; tail -4 gpurun_out/r5/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1750 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err ) 2> gpurun_out/r5/bench_final.time
tail -3 gpurun_out/r5/bench_final.time
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r5/bench_final.json") if l.startswith("{")][-1])
e = j["end_to_end"]
print({k: j[k] for k in ("value", "ms_per_step", "n_gpus")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["traffic"], j["cpu_baseline"]["value"])
print({k: e.get(k) for k in ("reads", "make_s", "value", "extract_s", "call_s", "merge_s", "extract_plus_call_s", "vs_cpu_baseline_e2e_extract_plus_call", "vs_cpu_baseline_e2e_wall", "error")})
print(e.get("check", {}).get("ok"), [ (r["wall_s"], r["loop_s"]) for r in e.get("runs", [])])
print([k for k in j if "full_size" in k])
PY
