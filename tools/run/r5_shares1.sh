#!/bin/bash
# round 5: shares (extract --gpus N, a contiguous share of the file per context) -- the tests, then a 2^26-pair file through 1 / 2 / 4 contexts on the one device,
# chunk by chunk for comparison, and the host feed alone (no device stage) at 4 and 8 shares
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_cli.py tests/test_multi_device.py tests/test_front_device.py tests/test_abi.py -m gpu -x -q > gpurun_out/r5/t1.txt 2>&1; tail -15 gpurun_out/r5/t1.txt
python - > gpurun_out/r5/shares_e2e.log 2>&1 <<'PY'
import json, os, subprocess, sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import e2e_bench
from strling_amd import build
inp = e2e_bench.make_input(1 << 26)
print("made", inp["bam_MB"], "MB in", inp["make_s"], "s", flush=True)
cli = build.CLI
def go(tag, args, env=None):
    e = dict(os.environ, STRL_FRONT_TIMING="1"); e.update(env or {})
    time.sleep(3)
    t = time.time()
    r = subprocess.run([cli, "extract", "-v", "-g", inp["bed"]] + args + [inp["bam"], inp["out"] + tag], capture_output=True, text=True, env=e)
    w = time.time() - t
    keep = [l for l in r.stderr.splitlines() if not l.strip().endswith("reads/sec")]
    print(f"==== {tag} rc {r.returncode} wall {w:.3f} s", flush=True)
    print("\n".join(keep[-24:]), flush=True)
    return r
for rep in range(2):
    go("g1", [])
go("g2", ["--gpus", "2"])
go("g4", ["--gpus", "4"])
go("g4", ["--gpus", "4"])
go("g8", ["--gpus", "8"])
go("g4rr", ["--gpus", "4"], {"STRL_SHARES": "0"})
for t in ("g2", "g4", "g8", "g4rr"):
    same = open(inp["out"] + "g1", "rb").read() == open(inp["out"] + t, "rb").read()
    print("bin", t, "identical to g1:", same, flush=True)
for g in ("4", "8"):
    go("feed" + g, ["--gpus", g], {"STRL_FEED_ONLY": "1"})
for t in ("g1", "g2", "g4", "g8", "g4rr"):
    try: os.remove(inp["out"] + t)
    except OSError: pass
e2e_bench.cleanup(inp)
PY
tail -c 9000 gpurun_out/r5/shares_e2e.log
