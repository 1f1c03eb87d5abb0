#!/bin/bash
# round 5: the 57 GB / 5.4e8-read file through `extract` on 1 context (twice), then by shares on 4 and 8 contexts (ONE device: what it
# shows is that shares of 7 - 14 GB, hundreds of chunks and the in-place gather of 5.4e8 reads give the same .bin), the host feed
# alone at that size, `call` with the clustering's host phases, `merge`, 16 slabs against the oracle.  Every process under a timeout.
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
python - 2>&1 <<'PY' | head -c 600000 > gpurun_out/r5/full_shares.log
import json, os, subprocess, sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import e2e_bench
from strling_amd import build
T0 = time.time()
inp = e2e_bench.make_input(1 << 28)
print("made", inp["bam_MB"], "MB in", inp["make_s"], "s; at", round(time.time() - T0), flush=True)
cli = build.CLI
def run(tag, cmd, env=None, limit=240):
    e = dict(os.environ, STRL_FRONT_TIMING="1"); e.update(env or {})
    time.sleep(6)
    t = time.time()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=limit)
        rc, err = r.returncode, r.stderr
    except subprocess.TimeoutExpired as x:
        rc, err = 124, (x.stderr.decode(errors="replace") if isinstance(x.stderr, bytes) else (x.stderr or "")) + "\n*** TIMEOUT ***"
    w = time.time() - t
    lines = err.splitlines()
    keep = [l for l in lines if not l.strip().endswith("reads/sec")] if rc != 124 else lines
    print(f"==== {tag} rc {rc} wall {w:.3f} s (at {time.time() - T0:.0f} s)", flush=True)
    print("\n".join(l[:700] for l in keep[-(40 if rc == 124 else 16):]), flush=True)
    return rc
ex = [cli, "extract", "-v", "-g", inp["bed"]]
run("g1", ex + [inp["bam"], inp["out"]])
run("g1", ex + [inp["bam"], inp["out"]])
run("g1 sync alloc", ex + [inp["bam"], inp["out"] + "s"], {"STRL_SYNC_ALLOC": "1"})
run("g4", ex + ["--gpus", "4", inp["bam"], inp["out"] + "g4"])
run("g8", ex + ["--gpus", "8", inp["bam"], inp["out"] + "g8"])
for t in ("s", "g4", "g8"):
    p = inp["out"] + t
    print("bin", t, "identical to the one-context run:", os.path.exists(p) and subprocess.run(["cmp", inp["out"], p]).returncode == 0, flush=True)
    if os.path.exists(p): os.remove(p)
run("feed8", ex + ["--gpus", "8", inp["bam"], inp["out"] + "f"], {"STRL_FEED_ONLY": "1"})
run("feed4", ex + ["--gpus", "4", inp["bam"], inp["out"] + "f"], {"STRL_FEED_ONLY": "1"})
run("call", [cli, "call", "-v", "-o", inp["prefix"], inp["bam"], inp["out"]], {"STRL_CLUSTER_TIMING": "1"})
run("merge", [cli, "merge", "-v", "-o", inp["prefix"] + "-joint", inp["out"]], {"STRL_CLUSTER_TIMING": "1"})
if time.time() - T0 < 1500:
    chk = e2e_bench.check_in_subprocess(inp, e2e_bench.pick_slabs(inp["n_slabs"], 16), call=True)
    print("check", json.dumps(chk), flush=True)
e2e_bench.cleanup(inp)
print("done at", round(time.time() - T0), flush=True)
PY
grep -n '^====\|identical\|TIMEOUT\|feed only\|seconds: total\|gathered\|^check\|strl_cluster\|cluster_collect\|^made\|^done' gpurun_out/r5/full_shares.log | cut -c1-330
