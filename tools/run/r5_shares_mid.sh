#!/bin/bash
# shares + the background allocation at a size where both are active (1.3e8 reads: hint per context > 2^25), every process under a timeout
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
python - 2>&1 <<'PY' | head -c 400000 > gpurun_out/r5/shares_mid.log
import json, os, subprocess, sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import e2e_bench
from strling_amd import build
inp = e2e_bench.make_input(1 << 26)
print("made", inp["bam_MB"], "MB in", inp["make_s"], "s", flush=True)
cli = build.CLI
def run(tag, cmd, env=None, limit=150):
    e = dict(os.environ, STRL_FRONT_TIMING="1"); e.update(env or {})
    time.sleep(3)
    t = time.time()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=limit)
        rc, err = r.returncode, r.stderr
    except subprocess.TimeoutExpired as x:
        rc, err = 124, (x.stderr.decode(errors="replace") if isinstance(x.stderr, bytes) else (x.stderr or "")) + "\n*** TIMEOUT ***"
    w = time.time() - t
    keep = [l for l in err.splitlines() if not l.strip().endswith("reads/sec")]
    print(f"==== {tag} rc {rc} wall {w:.3f} s", flush=True)
    print("\n".join(l[:600] for l in keep[-14:]), flush=True)
ex = [cli, "extract", "-v", "-g", inp["bed"]]
run("g1", ex + [inp["bam"], inp["out"]])
run("g4", ex + ["--gpus", "4", inp["bam"], inp["out"] + "g4"])
run("g8", ex + ["--gpus", "8", inp["bam"], inp["out"] + "g8"])
run("g4 sync alloc", ex + ["--gpus", "4", inp["bam"], inp["out"] + "g4s"], {"STRL_SYNC_ALLOC": "1"})
for t in ("g4", "g8", "g4s"):
    p = inp["out"] + t
    print("bin", t, "identical:", os.path.exists(p) and subprocess.run(["cmp", inp["out"], p]).returncode == 0, flush=True)
run("call", [cli, "call", "-v", "-o", inp["prefix"], inp["bam"], inp["out"]], {"STRL_CLUSTER_TIMING": "1"})
run("merge", [cli, "merge", "-v", "-o", inp["prefix"] + "-joint", inp["out"]], {"STRL_CLUSTER_TIMING": "1"})
run("merge --gpus 4", [cli, "merge", "-v", "--gpus", "4", "-o", inp["prefix"] + "-joint4", inp["out"]])
for t in ("g4", "g8", "g4s"):
    try: os.remove(inp["out"] + t)
    except OSError: pass
e2e_bench.cleanup(inp)
PY
grep -n '^====\|identical\|TIMEOUT\|seconds: total\|gathered\|strl_cluster\|cluster_collect\|^made' gpurun_out/r5/shares_mid.log | cut -c1-300
