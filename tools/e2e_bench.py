"""End-to-end `strling extract` rate (BAM bytes -> .bin) on a synthetic BAM, with the CLI's own phase breakdown.
usage: python tools/e2e_bench.py [n_pairs] [threads...]     (GPU box)"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from strling_amd import bamio, build, synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
repeat = int(os.environ.get("REPEAT", "1"))
threads = [int(x) for x in sys.argv[2:]] or [1]
d = os.environ.get("TMPDIR", "/tmp")
bam, bed, out = f"{d}/e2e.bam", f"{d}/e2e.str", f"{d}/e2e.bin"
t0 = time.time()
rec, g = synth.synth_wgs(n_pairs, seed=99)
bamio.write_bam(bam, rec, level=6, repeat=repeat)
bamio.write_genome_bed(bed, g, rec.targets)
res = {"reads": rec.n * repeat, "bam_MB": round(os.path.getsize(bam) / 1e6, 1), "make_s": round(time.time() - t0, 1), "nproc": os.cpu_count(), "runs": []}
for t in threads:
    env = dict(os.environ, STRL_THREADS=str(t), STRL_DECODE_TIMING="1")
    t1 = time.time()
    r = subprocess.run([build.CLI, "extract", "-v", "-g", bed, bam, out], capture_output=True, text=True, env=env)
    wall = time.time() - t1
    line = [l for l in r.stderr.splitlines() if "seconds: total" in l]
    dec = [l.split("decode seconds:")[1].strip() for l in r.stderr.splitlines() if "decode seconds:" in l]
    res["runs"].append({"threads": t, "decode": dec, "rc": r.returncode, "wall_s": round(wall, 3), "reads_per_s": round(rec.n * repeat / wall), "phases": line[-1].split("seconds:")[1].strip() if line else r.stderr[-300:]})
print(json.dumps(res))
