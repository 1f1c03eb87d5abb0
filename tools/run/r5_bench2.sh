#!/bin/bash
# round 5: `python bench.py --gpus 2` on the one device (two ranks over gloo: a dry run of the N > 1 path) WITH its file leg: rank 0 runs `strling extract --gpus 2`
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --reads-per-gpu 1048576 --no-cpu-baseline --e2e-pairs 4194304 > gpurun_out/r5/bench_gpus2.json 2> gpurun_out/r5/bench_gpus2.err ) 2> gpurun_out/r5/bench_gpus2.time
tail -3 gpurun_out/r5/bench_gpus2.time; tail -5 gpurun_out/r5/bench_gpus2.err | cut -c1-300
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r5/bench_gpus2.json") if l.startswith("{")][-1])
e = j["end_to_end"]
print({k: j[k] for k in ("value", "ms_per_step", "n_gpus", "exchange", "backend", "devices_visible")})
print({k: e.get(k) for k in ("gpus", "reads", "extract_s", "call_s", "merge_s", "error")}, e.get("check", {}).get("ok"))
print(e["runs"][0].get("shares"), e["runs"][0].get("gather"))
PY
