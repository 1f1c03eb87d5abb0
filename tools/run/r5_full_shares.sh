#!/bin/bash
# round 5: the 57 GB / 5.4e8-read file through `extract` on 1 context (twice), then by shares on 4 and 8 contexts (ONE device: what it
# shows is that shares of 7 - 14 GB, hundreds of chunks and the in-place gather of 5.4e8 reads give the same .bin), the host feed
# alone at that size, `call` with the clustering's host phases, `merge`, 16 slabs against the oracle
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
python - > gpurun_out/r5/full_shares.log 2>&1 <<'PY'
import json, os, subprocess, sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import e2e_bench
from strling_amd import build
inp = e2e_bench.make_input(1 << 28, progress=True)
print("made", inp["bam_MB"], "MB in", inp["make_s"], "s", flush=True)
cli = build.CLI
def go(tag, args, env=None, out=None):
    e = dict(os.environ, STRL_FRONT_TIMING="1"); e.update(env or {})
    time.sleep(6)
    t = time.time()
    r = subprocess.run([cli, "extract", "-v", "-g", inp["bed"]] + args + [inp["bam"], out or (inp["out"] + tag)], capture_output=True, text=True, env=e, timeout=900)
    w = time.time() - t
    keep = [l for l in r.stderr.splitlines() if not l.strip().endswith("reads/sec")]
    print(f"==== {tag} rc {r.returncode} wall {w:.3f} s", flush=True)
    print("\n".join(l[:700] for l in keep[-16:]), flush=True)
    return w
go("g1", [], out=inp["out"])
go("g1", [], out=inp["out"])
go("g4", ["--gpus", "4"])
go("g8", ["--gpus", "8"])
for t in ("g4", "g8"):
    a, b = inp["out"], inp["out"] + t
    same = subprocess.run(["cmp", a, b]).returncode == 0
    print("bin", t, "identical to the one-context run:", same, flush=True)
    os.remove(b)
go("feed8", ["--gpus", "8"], {"STRL_FEED_ONLY": "1"})
go("feed4", ["--gpus", "4"], {"STRL_FEED_ONLY": "1"})
time.sleep(6)
t = time.time()
r = subprocess.run([cli, "call", "-v", "-o", inp["prefix"], inp["bam"], inp["out"]], capture_output=True, text=True, env=dict(os.environ, STRL_CLUSTER_TIMING="1"))
print(f"==== call rc {r.returncode} wall {time.time() - t:.3f} s\n" + "\n".join(l[:900] for l in r.stderr.splitlines() if "strl_cluster" in l or "seconds" in l), flush=True)
time.sleep(6)
t = time.time()
r = subprocess.run([cli, "merge", "-v", "-o", inp["prefix"] + "-joint", inp["out"]], capture_output=True, text=True, env=dict(os.environ, STRL_CLUSTER_TIMING="1"))
print(f"==== merge rc {r.returncode} wall {time.time() - t:.3f} s\n" + "\n".join(l[:900] for l in r.stderr.splitlines() if "strl_cluster" in l or "seconds" in l), flush=True)
chk = e2e_bench.check(inp, e2e_bench.pick_slabs(inp["n_slabs"], 16), call=True)
print("check", json.dumps(chk), flush=True)
e2e_bench.cleanup(inp)
PY
grep -n '^====\|identical\|feed only\|seconds: total\|gathered\|^check\|strl_cluster\|^made' gpurun_out/r5/full_shares.log | cut -c1-330
