#!/bin/bash
# where do the seconds in front of the extract loop go on a whole-genome file?  small file, whole-genome sized hint / contig count
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python - <<'P' > gpurun_out/r4/setup_probe.txt 2>&1
import os, subprocess, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import e2e_bench
from strling_amd import build, bamio
inp = e2e_bench.make_input(1 << 21, level=1)
for hint in ("", "1190000000", "600000000"):
    env = dict(os.environ)
    if hint: env["STRL_READS_HINT"] = hint
    r = subprocess.run([build.CLI, "extract", "-v", "-g", inp["bed"], inp["bam"], inp["out"]], capture_output=True, text=True, env=env)
    print("hint", hint or "default", [l for l in r.stderr.splitlines() if "before the loop" in l or "device memory" in l])
# many contigs: 2048 slabs of 2^10 pairs
r = bamio.write_bam_slabs("/tmp/many.bam", 1024, 1 << 11, seed=5, level=1, bed="/tmp/many.str", index=False)
rr = subprocess.run([build.CLI, "extract", "-v", "-g", "/tmp/many.str", "/tmp/many.bam", "/tmp/many.bin"], capture_output=True, text=True)
print("2048 contigs", r["reads"], [l for l in rr.stderr.splitlines() if "before the loop" in l or "seconds: total" in l])
P
cat gpurun_out/r4/setup_probe.txt
