#!/bin/bash
# round 5: the driver's own command (python bench.py, no flags) -- its end_to_end leg now writes the 2^28-pair file when the box has the room
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
( time timeout 1750 python bench.py > gpurun_out/r5/bench_full.json 2> gpurun_out/r5/bench_full.err ) 2> gpurun_out/r5/bench_full.time
tail -3 gpurun_out/r5/bench_full.time
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r5/bench_full.json") if l.startswith("{")][-1])
e = j["end_to_end"]
print({k: j[k] for k in ("value", "ms_per_step", "n_gpus")})
print({k: e.get(k) for k in ("reads", "bam_MB", "make_s", "value", "extract_s", "call_s", "merge_s", "extract_plus_call_s", "vs_cpu_baseline_e2e_extract_plus_call", "vs_cpu_baseline_e2e_wall", "error")})
print(e.get("check"))
for r in e.get("runs", []):
    print(r["wall_s"], r["loop_s"], r["device_mem_GB"], r.get("device_front_end"))
PY
tail -5 gpurun_out/r5/bench_full.err
