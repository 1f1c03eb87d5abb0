#!/bin/bash
# GPU box: host inflate A/B (zlib vs cli/fast_inflate.cpp): single-thread block bench, the reader alone, and extract end to end.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<'PY'
import sys, json
sys.path.insert(0, "tools")
import e2e_bench
inp = e2e_bench.make_input(2 ** 23)
json.dump(inp, open("/tmp/e2e_inp.json", "w"))
print(inp)
PY
BAM=$(python -c "import json; print(json.load(open('/tmp/e2e_inp.json'))['bam'])")
g++ -O3 -std=c++17 -Istrling_amd/csrc/cli tools/host_inflate_bench.cpp strling_amd/csrc/cli/fast_inflate.cpp -lz -o /tmp/hib && /tmp/hib $BAM
for m in zlib fast zlib fast; do STRL_INFLATE=$m STRL_DECODE_TIMING=1 strling_amd/lib/strling _decode $BAM 2>&1 | tail -2; done
for m in zlib fast; do
STRL_INFLATE=$m python - <<'PY'
import sys, json, os
sys.path.insert(0, "tools")
import e2e_bench
from strling_amd import build
inp = json.load(open("/tmp/e2e_inp.json"))
r = e2e_bench.run(inp, build.CLI, (0, 0))
print(os.environ["STRL_INFLATE"], json.dumps(r["runs"]))
PY
done
