"""Decode rate of the CRAM reader on a file whose reads look like a sequencer's (a read feature or two per record -- the files of
tests/test_cram.py disagree with their reference at most positions: ~100 features per record, the worst case).  A sample of
N records is written once by the Python writer (strling_amd/cramio.py) and its data containers are repeated K times; the
same records as a BAM beside it.  `strling _decode` = the reader alone (all host threads); with --extract on a GPU box
`strling extract` on both.        usage: python tools/cram_bench.py [--records 40000] [--repeat 200] [--extract]"""
import argparse
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import numpy as np
    from strling_amd import bamio, build, cramio, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=40000)
    ap.add_argument("--repeat", type=int, default=200)
    ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--extract", action="store_true")
    ap.add_argument("--qualities", action="store_true")
    ap.add_argument("--threads", default="")
    a = ap.parse_args()
    t0 = time.time()
    rec, g = synth.synth_wgs(a.records, seed=11, n_contigs=2, contig_len=max(200_000, a.records * 4), indel_frac=0.01, soft_frac=0.03)
    rng = np.random.default_rng(3)
    refs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).astype(np.uint8).tobytes() for _, ln in rec.targets]
    cramio.reads_from_reference(rec, refs)
    d = a.dir
    fa, cram, bam, bed = f"{d}/cb.fa", f"{d}/cb.cram", f"{d}/cb.bam", f"{d}/cb.str"
    cramio.write_fasta(fa, rec.targets, refs)
    st = {}
    cramio.write_cram(cram, rec, refs, records_per_slice=10000, slices_per_container=1, index=False, repeat=a.repeat, qualities=a.qualities, tags=True, stats=st)
    bamio.write_bam(bam, rec, repeat=a.repeat, level=6)
    bamio.write_genome_bed(bed, g, rec.targets)
    n = rec.n * a.repeat
    print(f"[cram_bench] {rec.n} records x {a.repeat}: {os.path.getsize(cram) / 1e6:.1f} MB CRAM, {os.path.getsize(bam) / 1e6:.1f} MB BAM, written in {time.time() - t0:.1f} s; mate chains {st}", flush=True)
    env = dict(os.environ, STRL_CRAM_FASTA=fa)
    if a.threads:
        env["STRL_THREADS"] = a.threads
    for path in (cram, bam):
        for _ in range(2):
            r = subprocess.run([build.CLI, "_decode", path, "1048576", "nosum"], capture_output=True, text=True, env=env)
            m = re.search(r"decoded (\d+) records in ([\d.]+) s with (\d+) threads", r.stderr)
            print(f"[cram_bench] _decode {os.path.basename(path)}: rc {r.returncode} {r.stderr.strip()[-200:]}", flush=True)
            assert r.returncode == 0 and m and int(m.group(1)) == n, r.stderr[-400:]
    if a.extract:
        for path in (cram, bam):
            for _ in range(2):
                t = time.time()
                r = subprocess.run([build.CLI, "extract", "-v", "-f", fa, "-g", bed, path, path + ".bin"], capture_output=True, text=True, env=env)
                w = time.time() - t
                tail = [l for l in r.stderr.splitlines() if "seconds" in l or "reads," in l]
                print(f"[cram_bench] extract {os.path.basename(path)}: rc {r.returncode} wall {w:.3f} s = {n / w:.3e} records/s\n   " + "\n   ".join(tail[-3:]), flush=True)
    for p in (fa, fa + ".fai", cram, bam, bed, cram + ".bin", bam + ".bin"):
        if os.path.exists(p):
            os.remove(p)


if __name__ == "__main__":
    main()
