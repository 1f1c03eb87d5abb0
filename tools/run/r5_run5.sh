#!/bin/bash
# round 5: tests of the out-of-memory route, CRAM 3.1, the background allocation; CRAM decode / extract rates; one-GPU e2e again
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_cli.py tests/test_cram.py tests/test_front_device.py tests/test_abi.py tests/test_multi_device.py tests/test_call.py -m gpu -x -q > gpurun_out/r5/t5.txt 2>&1; tail -8 gpurun_out/r5/t5.txt
timeout 900 python tools/cram_bench.py --records 40000 --repeat 125 --extract --qualities > gpurun_out/r5/cram_bench.log 2>&1; cat gpurun_out/r5/cram_bench.log | cut -c1-400
timeout 900 python tools/e2e_bench.py $((1<<26)) --check-slabs 4 --repeats 3 --out gpurun_out/r5/e2e_26.json > gpurun_out/r5/e2e_26.log 2>&1
python - <<'PY'
import json
e = json.load(open("gpurun_out/r5/e2e_26.json"))
for r in e["runs"]:
    print(r["wall_s"], r["loop_s"], r["outside_the_loop"])
print({k: e.get(k) for k in ("extract_s", "call_s", "merge_s", "check")})
PY
