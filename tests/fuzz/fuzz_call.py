"""Randomised end-to-end sweep of the CLI (extract -> call [-l/-b], extract x3 -> merge [-l]) against the oracle.
usage: python tests/fuzz/fuzz_call.py [seconds]     (test infrastructure: uses the oracle; GPU box)"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from strling_amd import bamio, build, synth
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
master = int(sys.argv[2]) if len(sys.argv) > 2 else 77
rng = np.random.default_rng(master)
CLI = build.CLI
print(f"fuzz_call: master seed {master}, budget {budget:.0f} s (every case's own seed and parameters are in its assertion message; a progress line per 25 cases)", flush=True)


def run(args):
    env = dict(os.environ)
    if args[0] == "call":        # the evidence reads through the device in batches of one region / 1 MB / the default, or on the host
        mode = int(rng.integers(0, 4))
        if mode == 3:
            env["STRL_CALL_REGIONS"] = "host"
        elif mode < 2:
            env["STRL_CALL_BATCH_MB"] = str(mode)
    r = subprocess.run([CLI] + args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, (args, r.stderr[-600:])
    return r


def loci_bed(rows, targets, k0):
    """loci the reference accepts (parse_bed asserts left_most <= right_most, i.e. the locus must start inside its contig)"""
    length = dict(targets)
    out = []
    for k, r in enumerate(rows):
        f = r.split("\t")
        if (k + k0) % 2 == 0:
            out.append(f"{f[0]}\t{max(0, int(f[1]) - 3)}\t{int(f[2]) + 5}\t{f[3]}\tlocus{k}")
        elif (k + k0) % 3 == 0 and int(f[1]) + 1540 < length[f[0]]:
            out.append(f"{f[0]} {int(f[1]) + 1500} {int(f[1]) + 1540} {f[3]}")
    out.append(f"{targets[0][0]}\t100\t1500\tAC\twide")
    return "\n".join(out) + "\n"


t0 = time.time()
n_call = n_merge = n_rows = n_cram = n_gpus_runs = 0
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        seed = int(rng.integers(1, 1 << 30))
        nc = int(rng.choice([2, 3, 5]))
        kw = dict(n_contigs=nc, contig_len=int(rng.choice([20_000, 40_000, 80_000, 160_000])), str_frac=float(rng.choice([0.01, 0.05, 0.15])),
                  soft_frac=float(rng.choice([0.03, 0.15])), unmapped_frac=float(rng.choice([0.005, 0.03])))
        n_pairs = int(rng.choice([3000, 6000, 12000]))
        m, q = int(rng.choice([2, 3, 5])), int(rng.choice([0, 20, 40]))
        c, t = int(rng.choice([0, 0, 1])), int(rng.choice([0, 0, 2]))
        tag = f"seed={seed} pairs={n_pairs} m={m} q={q} c={c} t={t} {kw}"
        if rng.random() < 0.7:          # ---- extract -> call ----
            rec, g = synth.synth_wgs(n_pairs, seed=seed, **kw)
            bam, bed, binp, pre = (os.path.join(d, x) for x in ("s.bam", "g.str", "s.bin", "o"))
            fa_args = []
            if n_pairs == 3000 and rng.random() < 0.5:     # the same records as a CRAM (random slice / container shapes): same files expected
                from strling_amd import cramio
                bam = os.path.join(d, "s.cram")
                refs = cramio.make_reference(rec, seed=seed)
                cramio.write_fasta(os.path.join(d, "ref.fa"), rec.targets, refs)
                cramio.write_cram(bam, rec, refs, records_per_slice=int(rng.choice([97, 500, 4000])), slices_per_container=int(rng.choice([1, 3])),
                                  ap_delta=bool(rng.random() < 0.7), version=(3, 1) if rng.random() < 0.5 else (3, 0), qualities=bool(rng.random() < 0.3),
                                  tags=bool(rng.random() < 0.3))
                fa_args = ["-f", os.path.join(d, "ref.fa")]
                n_cram += 1
            else:
                bamio.write_bam(bam, rec, block=int(rng.choice([0xFF00, 0xFF00, 3000, 700])), level=int(rng.choice([1, 6])))   # (small blocks: many per 16 KiB index window, records across block ends)
            bamio.write_genome_bed(bed, g, rec.targets)
            gp = int(rng.choice([1, 1, 2, 3, 5])) if not fa_args else 1      # BAM input: also over several contexts (shares cut at the .bai's record starts)
            run(["extract"] + fa_args + (["--gpus", str(gp)] if gp > 1 else []) + ["-g", bed, "-q", str(q), bam, binp])
            n_gpus_runs += gp > 1
            frag = synth.frag_hist(rec)
            tr = O.extract(rec, g, O.make_opts(O.median(frag), 0.8, q))
            kwc = dict(min_support=m, min_mapq=q, min_clip=c, min_clip_total=t)
            eb, eg, eu = O.call(tr, rec, frag, **kwc)
            args = ["call"] + fa_args + ["-m", str(m), "-q", str(q), "-c", str(c), "-t", str(t), "-o", pre]
            extra = {}
            rows = ["\t".join(l.split("\t")[:11]) for l in eb.splitlines()[1:]]
            if rows and rng.random() < 0.5:
                bt = "#h\n" + "\n".join(rows[: max(1, len(rows) - 1)]) + "\n"
                open(os.path.join(d, "b.txt"), "w").write(bt)
                args += ["-b", os.path.join(d, "b.txt")]
                extra["bounds_text"] = bt
            if rows and rng.random() < 0.5:
                lt = loci_bed(rows, rec.targets, int(rng.integers(0, 3)))
                open(os.path.join(d, "l.bed"), "w").write(lt)
                args += ["-l", os.path.join(d, "l.bed")]
                extra["loci_text"] = lt
            if extra:
                eb, eg, eu = O.call(tr, rec, frag, **kwc, **extra)
            run(args + [bam, binp])
            for suf, exp in (("-bounds.txt", eb), ("-genotype.txt", eg), ("-unplaced.txt", eu)):
                assert open(pre + suf).read() == exp, (suf, tag, extra.keys())
            n_call += 1
            if n_call % 25 == 0:
                print(f"  {n_call} extract->call runs, {n_merge} merges, {time.time() - t0:.0f} s, last: {tag}", flush=True)
            n_rows += eb.count("\n") - 1
        else:                            # ---- three samples -> merge ----
            bins, parts, frags, targets = [], [], [], None
            for s_i in range(3):
                rec, g = synth.synth_wgs(n_pairs // 2, seed=seed + s_i, **kw)
                targets = rec.targets
                frag = synth.frag_hist(rec)
                tr = O.extract(rec, g, O.make_opts(O.median(frag), 0.8, 40))
                p = os.path.join(d, f"m{s_i}.bin")
                open(p, "wb").write(O.bin_write(0.8, 40, frag, bamio.sam_header(rec.targets), tr, rec.qname_off, rec.qnames))
                bins.append(p)
                tr = tr[tr["tid"] >= 0].copy()
                tr["qname_id"] = s_i
                parts.append(tr)
                frags.append(frag)
            merged = np.concatenate(parts)
            frag = np.sum(frags, axis=0).astype(np.uint32)
            window, mcd = O.median(frag, 0.98), int(0.5 * O.median(frag, 0.5))
            exp = O.merge_text(merged, window, targets, min_support=m, min_clip=c, min_clip_total=t, max_clip_dist=mcd)
            args = ["merge", "-m", str(m), "-c", str(c), "-t", str(t), "-o", os.path.join(d, "j")]
            rows = exp.splitlines()[1:]
            if rows and rng.random() < 0.5:
                lt = loci_bed(rows, targets, int(rng.integers(0, 3)))
                open(os.path.join(d, "ml.bed"), "w").write(lt)
                args += ["-l", os.path.join(d, "ml.bed")]
                exp = O.merge_text(merged, window, targets, min_support=m, min_clip=c, min_clip_total=t, max_clip_dist=mcd, loci_text=lt)
            run(args + bins)
            assert open(os.path.join(d, "j-bounds.txt")).read() == exp, ("merge", tag)
            n_merge += 1
            n_rows += exp.count("\n") - 1
print(f"fuzz_call ok (master seed {master}): {n_call} extract->call runs ({n_cram} of them from CRAM 3.0 / 3.1, {n_gpus_runs} with --gpus 2 / 3 / 5), {n_merge} merges, {n_rows} bounds rows identical to the oracle in {time.time() - t0:.0f} s")
