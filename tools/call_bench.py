"""Wall time of `strling extract` + `strling call` on a synthetic indexed BAM.  usage: python tools/call_bench.py [n_pairs]"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strling_amd import bamio, build, synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
d = os.environ.get("TMPDIR", "/tmp")
rec, g = synth.synth_wgs(n_pairs, seed=5, n_contigs=25, contig_len=1_000_000, str_frac=0.02)
bam, bed, binp, pre = (os.path.join(d, x) for x in ("cb.bam", "cb.str", "cb.bin", "cb"))
bamio.write_bam(bam, rec, level=6)
bamio.write_genome_bed(bed, g, rec.targets)
t0 = time.time(); r1 = subprocess.run([build.CLI, "extract", "-g", bed, bam, binp], capture_output=True, text=True); t1 = time.time()
r2 = subprocess.run([build.CLI, "call", "-o", pre, bam, binp], capture_output=True, text=True); t2 = time.time()
assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-300:], r2.stderr[-300:])
print(json.dumps({"reads": rec.n, "bam_MB": round(os.path.getsize(bam) / 1e6, 1), "extract_s": round(t1 - t0, 2), "call_s": round(t2 - t1, 2),
                  "bounds_rows": sum(1 for _ in open(pre + "-bounds.txt")) - 1, "genotype_rows": sum(1 for _ in open(pre + "-genotype.txt")) - 1}))
