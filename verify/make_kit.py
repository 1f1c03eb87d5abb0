#!/usr/bin/env python
"""Builds the verification kit for the assumptions NO reference test pins (SURVEY section 8c / DESIGN section 2): small inputs on
which each assumption changes output bytes, and the outputs the oracle (oracle/strling_oracle.c, the CPU restatement the HIP
path is checked against) expects.  A maintainer with Nim 1.6 + htslib runs verify/run_reference.sh against a real `strling`
binary: every PASS pins an assumption, every FAIL names the one function to correct on each side.

    python verify/make_kit.py          # regenerates verify/cases/* (deterministic; the files are committed)

case            assumption under test                                       changes
iupac           kmer's code of a base that is not A/C/G/T (assumed: 'A')      repeat unit / count of reads with <= 20 N or IUPAC codes
widths          msgpack4nim writes the smallest integer / string encoding     .bin bytes (tid, position, flag, counts, qname headers)
ties            CountTable.largest = first maximum in Nim 1.6 slot order      left / right of a bound whose clip positions tie
manygroups      Table[(tid, repeat)] iteration order after it grows           row order of -bounds.txt with > 8192 groups
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import oracle as O                     # noqa: E402
from strling_amd import bamio                      # noqa: E402
from strling_amd.records import RecordBatch        # noqa: E402

CASES = os.path.join(HERE, "cases")
N_CONTIGS = 130
TARGETS = [("chr000", 70000)] + [("chr%03d" % i, 1000) for i in range(1, N_CONTIGS)]


def write_reference():
    rng = np.random.default_rng(20260929)
    fa, fai, off = [], [], 0
    for name, ln in TARGETS:
        seq = "".join(rng.choice(list("ACGT"), ln))
        hdr = ">%s\n" % name
        off += len(hdr)
        fai.append("%s\t%d\t%d\t60\t61\n" % (name, ln, off))
        body = "".join(seq[i:i + 60] + "\n" for i in range(0, ln, 60))
        off += len(body)
        fa.append(hdr + body)
    open(os.path.join(CASES, "ref.fa"), "w").write("".join(fa))
    open(os.path.join(CASES, "ref.fa.fai"), "w").write("".join(fai))
    open(os.path.join(CASES, "ref.fa.str"), "w").write("chr000\t10\t40\tAC\n")     # given with -g: no index step involved


def pair(f, qname, tid, pos, mpos, seq1, seq2, cig1="10S140M", cig2="150M", flag1=99, flag2=147, mapq=60, mtid=None, isize=300):
    mtid = tid if mtid is None else mtid
    for (p, mp, s, c, fl, t, mt) in ((pos, mpos, seq1, cig1, flag1, tid, mtid), (mpos, pos, seq2, cig2, flag2, mtid, tid)):
        f["qnames"].append(qname); f["flag"].append(fl); f["tid"].append(t); f["pos"].append(p); f["mapq"].append(mapq)
        f["cigars"].append(c); f["mtid"].append(mt); f["mpos"].append(mp); f["isize"].append(isize if p <= mp else -isize); f["seqs"].append(s)


def sorted_batch(f):
    n = len(f["seqs"])
    key = [(t if t >= 0 else 1 << 30, p, i) for i, (t, p) in enumerate(zip(f["tid"], f["pos"]))]
    order = [k[2] for k in sorted(key)]
    g = {k: [v[i] for i in order] for k, v in f.items()}
    return RecordBatch.from_fields(targets=TARGETS, **g)


def flank(rng, n=150):
    return "".join(rng.choice(list("ACGT"), n))


def mutate(seq, positions, letters):
    s = list(seq)
    for p, c in zip(positions, letters):
        s[p] = c
    return "".join(s)


def case_iupac():
    """kmer's code of a nibble that is not A/C/G/T decides which windows collide in the k-mer histograms.  A homopolymer of base B
    with the code X at every 7th position (20 of them: still scored, utils.nim:238) tells X -> B from X -> anything else: if X counts
    as B, k = 3 sees 50 windows of BBB and the literal recount lifts the result to 3 x 44; if not, k = 3 stays below k = 2's score
    and the result is 2 x 64.  All 12 non-ACGT nibble codes x 4 bases."""
    rng = np.random.default_rng(1)
    f = dict(tid=[], pos=[], mtid=[], mpos=[], flag=[], mapq=[], cigars=[], seqs=[], qnames=[], isize=[])
    cag = "CAG" * 50
    pos = 1000
    for code in "=MRSVWYHKDBN":
        for base in "ACGT":
            pair(f, "%s_%s" % ("eq" if code == "=" else code, base), 0, pos, pos + 200, mutate(base * 150, range(6, 146, 7), code * 20), flank(rng))
            pos += 400
    pair(f, "n21", 0, pos, pos + 200, mutate("A" * 150, range(3, 150, 7), "N" * 21), flank(rng))               # > 20 N: not scored
    pair(f, "clipN", 0, pos + 400, pos + 600, mutate(cag, [2, 5, 8, 40], "NNNR"), flank(rng), cig1="60S90M")      # the soft-clip scan sees them too
    pair(f, "cagN", 0, pos + 800, pos + 1000, mutate(cag, [10, 70, 130], "NNN"), flank(rng))
    return sorted_batch(f)


def case_widths():
    rng = np.random.default_rng(2)
    f = dict(tid=[], pos=[], mtid=[], mpos=[], flag=[], mapq=[], cigars=[], seqs=[], qnames=[], isize=[])
    cag = "CAG" * 50
    k = 0
    for tid, pos in ((0, 77), (0, 127), (0, 128), (0, 255), (0, 256), (0, 65535 - 60), (0, 65535), (0, 65536), (127, 100), (128, 100), (129, 500)):
        for flag1, flag2 in ((99, 147), (1123, 1171)):          # 1024 (duplicate) pushes the flag over 255
            qn = ("q%02d" % k).ljust((31, 32, 200, 5)[k % 4], "x")
            pair(f, qn, tid, pos, pos + 200 if pos + 400 < TARGETS[tid][1] else pos, cag, ("GCA" * 50 if k % 3 else flank(rng)), cig1="50S100M",
                 cig2="150M" if k % 2 else "100M50S", flag1=flag1, flag2=flag2, mapq=(60, 0, 130)[k % 3])
            k += 1
    # both mates unmapped, both STR: tid -1 in the .bin (negative fixint), and the twice-visited tail (extract.nim:326)
    pair(f, "unplaced", -1, -1, -1, cag, "AGC" * 50, cig1="*", cig2="*", flag1=77, flag2=141, mapq=0)
    return sorted_batch(f)


def make_treads(rows):
    """rows: (tid, position, repeat, flag, split, mapq, repeat_count, align_length, qname) -> (tread array, qname_off, qnames)"""
    t = np.zeros(len(rows), O.TREAD_DTYPE)
    qo, qb = [0], bytearray()
    for i, (tid, pos, rep, flag, split, mq, rc, al, qn) in enumerate(rows):
        t[i]["tid"] = tid; t[i]["position"] = pos; t[i]["repeat"] = rep.encode(); t[i]["flag"] = flag; t[i]["split"] = split
        t[i]["mapping_quality"] = mq; t[i]["repeat_count"] = rc; t[i]["align_length"] = al; t[i]["qname_id"] = i
        qb += qn.encode(); qo.append(len(qb))
    return t, np.asarray(qo, np.uint64), bytes(qb)


LEFT, RIGHT, NONE = 0, 1, 3      # cluster.nim:14-20 Soft


def case_ties():
    """clusters whose left / right clip positions tie in count: bounds() takes CountTable.largest (cluster.nim:204-211), i.e. the
    first maximum in slot order -- which position that is depends on Nim's hash of the uint32 key and the table's growth"""
    samples = []
    for s in range(2):
        rows, q = [], 0
        for c, (tid, base, unit) in enumerate(((0, 10000, "CAG"), (0, 30000, "AAAG"), (3, 300, "AT"), (5, 500, "CCG"))):
            for j in range(6):                                  # anchors
                rows.append((tid, base - 150 + 40 * j + s, unit, 99, NONE, 60, 30, 150, "s%da%d" % (s, q))); q += 1
            for off in ((0, 0, 7, 7), (3, 3, 12, 12, 25), (1, 1, 2, 2, 3, 3))[c % 3]:        # two / three positions with equal counts
                rows.append((tid, base + off + 100 * s, unit, 99, LEFT, 60, 20, 70, "s%dl%d" % (s, q))); q += 1
            for off in ((60, 60, 64, 64), (70, 70, 90, 90), (55, 55, 56, 56, 57, 57))[c % 3]:
                rows.append((tid, base + off + 100 * s, unit, 147, RIGHT, 60, 20, 70, "s%dr%d" % (s, q))); q += 1
        samples.append(make_treads(rows))
    return samples


def case_manygroups():
    """9000 (tid, repeat) groups of two reads each: the Table that groups the reads (merge.nim:92,121) grows past its initial
    size several times; the rows of -bounds.txt come out in its iteration order"""
    units = []
    for a in "ACGT":
        for b in "ACGT":
            for c in "ACGT":
                for d in "ACGT":
                    u = a + b + c + d
                    if len(set(u)) > 1:
                        units.append(u)
    rows, q = [], 0
    for g in range(9000):
        tid, unit = g % 60, units[(g // 60) % len(units)]
        base = 100 + 7 * (g // 60)
        for j in range(2):
            rows.append((tid, min(base + 3 * j, TARGETS[tid][1] - 1), unit, 99, NONE, 60, 30, 150, "g%d" % q)); q += 1
    rng = np.random.default_rng(4)
    order = rng.permutation(len(rows))                          # first appearance of the groups in no particular order
    return [make_treads([rows[i] for i in order])]


def frag_of(rec):
    from strling_amd import synth
    return synth.frag_hist(rec)


def main():
    os.makedirs(CASES, exist_ok=True)
    write_reference()
    from strling_amd.records import GenomeStr
    g = GenomeStr.from_lists(len(TARGETS), {0: [(10, 40)]})
    report = []
    for name, rec in (("iupac", case_iupac()), ("widths", case_widths())):
        hdr = bamio.write_bam(os.path.join(CASES, name + ".bam"), rec, level=6, index=True)
        frag = frag_of(rec)
        med = O.median(frag)
        t = O.extract(rec, g, O.make_opts(med, 0.8, 40))
        blob = O.bin_write(0.8, 40, frag, hdr.rstrip("\0"), t, rec.qname_off, rec.qnames)
        open(os.path.join(CASES, name + ".expected.bin"), "wb").write(blob)
        with open(os.path.join(CASES, name + ".expected.treads.tsv"), "w") as f:      # the same, readable
            f.write("tid\tposition\trepeat\tflag\tsplit\tmapq\trepeat_count\talign_length\tqname\n")
            for x in t:
                f.write("%d\t%d\t%s\t%d\t%d\t%d\t%d\t%d\t%s\n" % (x["tid"], x["position"], bytes(x["repeat"]).rstrip(b"\0").decode(), x["flag"], x["split"],
                                                               x["mapping_quality"], x["repeat_count"], x["align_length"], rec.qname(int(x["qname_id"])).decode()))
        report.append((name, rec.n, len(t)))
    hdr = bamio.sam_header(TARGETS)
    frag = np.zeros(4096, np.uint32)
    frag[300:400] = 1000
    for name, samples, m in (("ties", case_ties(), 2), ("manygroups", case_manygroups(), 2)):
        all_t = []
        for s, (t, qo, qn) in enumerate(samples):
            blob = O.bin_write(0.8, 40, frag, hdr, t, qo, qn)
            open(os.path.join(CASES, "%s.%d.bin" % (name, s)), "wb").write(blob)
            k = t[t["tid"] >= 0].copy()
            k["qname_id"] = s
            all_t.append(k)
        fs = (frag.astype(np.uint64) * len(samples)).astype(np.uint32)
        window, mcd = O.median(fs, 0.98), int(0.5 * O.median(fs, 0.5))
        b, _ = O.call_bounds(np.concatenate(all_t), 0, window, min_support=m, max_clip_dist=mcd)
        with open(os.path.join(CASES, name + ".expected-bounds.txt"), "w") as f:
            f.write("#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total\n")
            for x in b:
                f.write(O.bounds_row(x, TARGETS[int(x["tid"])][0]) + "\n")
        report.append((name, sum(len(s[0]) for s in samples), len(b)))
    for r in report:
        print("%-12s %6d inputs -> %6d expected rows" % r)


if __name__ == "__main__":
    main()
