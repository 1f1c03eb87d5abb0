"""Reads of more than STRL_DEVICE_READ_LEN (510) bases: the reference scores any length with uint8 histograms that wrap
(extract.nim:36-40, utils.nim:192-195).  The kernels pass such a read by; the library's host twin of the scorer
(csrc/host_score.cpp) scores it and its words are merged into the device's results by record index.

CPU: the host twin against the oracle on reads whose bins wrap.  GPU: batches and BAMs that mix 150-base reads with a few long
ones, through the C ABI, the device front end and the host reader, against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from strling_amd import api, bamio, build, synth
from strling_amd.records import CIGAR_OPS, RecordBatch, unpack_result
from helpers import oracle_words, soft_items_expected, treads_equal

CLI = build.CLI


def _host_word(L, read, p):
    w = C.c_uint32(0)
    assert L.strl_score_read_host(read.encode(), len(read), p, C.byref(w)) == 0
    return unpack_result(w.value)


def _tract(rng, unit, n_bases, purity):
    s = np.frombuffer((unit * (n_bases // len(unit) + 2)).encode(), np.uint8)[int(rng.integers(0, len(unit))):][:n_bases].copy()
    hit = rng.random(n_bases) > purity
    s[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    return s.tobytes().decode()


def _random_read(rng, L):
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].tobytes().decode()


def _long_read(rng, L):
    """background + one or two repeat tracts; some carry N / IUPAC letters"""
    kind = int(rng.integers(0, 6))
    if kind == 0:
        s = _random_read(rng, L)
    else:
        k = int(rng.integers(1, 7))
        unit = _random_read(rng, k)
        frac = float(rng.choice([0.3, 0.55, 0.75, 0.9, 1.0]))
        t = int(L * frac)
        a = int(rng.integers(0, L - t + 1))
        s = _random_read(rng, a) + _tract(rng, unit, t, float(rng.choice([0.8, 0.95, 1.0]))) + _random_read(rng, L - a - t)
        if kind == 5 and L > 40:       # a second tract of another unit on top
            u2 = _random_read(rng, int(rng.integers(2, 7)))
            t2 = int(rng.integers(10, L // 2))
            b = int(rng.integers(0, L - t2))
            s = s[:b] + _tract(rng, u2, t2, 1.0) + s[b + t2:]
    n_odd = int(rng.choice([0, 0, 3, 19, 21, 40]))
    if n_odd and L:
        s = list(s)
        for j in rng.integers(0, L, n_odd):
            s[int(j)] = "N" if rng.random() < 0.8 else "R"
        s = "".join(s)
    return s


def test_host_twin_equals_the_oracle_where_the_bins_wrap(oracle):
    """strl_score_read_host (the product's host scorer) == the oracle's get_repeat on 1500 reads of 0 .. 6000 bases: a class seen
    more than 255 times wraps its uint8 bin, and the running arg-max follows the wrapped values (utils.nim:192-195)"""
    L = api.load()
    rng = np.random.default_rng(77)
    wrapped = kept = 0
    for it in range(1500):
        n = int(rng.choice([0, 1, 2, 5, 6, 11, 150, 511, 512, 600, 777, 1000, 1536, 3000, 6000]))
        read = _long_read(rng, n)
        p = float(rng.choice([0.8, 0.73, 0.6, 0.5]))
        unit, count = oracle.get_repeat(read, p)
        if count >= 65536:
            continue
        gu, gc, gs = _host_word(L, read, p)
        assert (gu, gc, gs) == (unit, count, False), (it, n, p, read[:60])
        kept += count > 0
        wrapped += n >= 1024
    assert kept > 150 and wrapped > 300
    # a pure dinucleotide tract of 1200 bases: the class bin wraps four times (600 windows), the literal recount does not
    read = "AC" * 600
    assert oracle.get_repeat(read, 0.8) == ("CA", 599) and _host_word(L, read, 0.8) == ("CA", 599, False)     # (C < A: the minimum rotation)
    # homopolymer: reduce_repeat multiplies (utils.nim:271)
    assert _host_word(L, "A" * 700, 0.8)[:2] == oracle.get_repeat("A" * 700, 0.8)
    # more than 20 N: nothing (utils.nim:238)
    assert _host_word(L, "N" * 21 + "CAG" * 300, 0.8) == ("", 0, False)


def _mix_long(rec, rng, n_long, lengths=(511, 600, 1000, 2500, 7000)):
    """rec with n_long of its records replaced by long ones (same placement, flags, mates; new SEQ + cigar)"""
    n = rec.n
    seqs = [rec.sequence(i) for i in range(n)]
    cigs = [[int(c) for c in rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]] for i in range(n)]
    mapq = rec.mapq.copy()
    pick = rng.choice(n, n_long, replace=False)
    for i in pick:
        L = int(rng.choice(lengths))
        unit = _random_read(rng, int(rng.integers(2, 7)))
        shape = int(rng.integers(0, 7))
        S, M = CIGAR_OPS.index("S"), CIGAR_OPS.index("M")
        if shape == 0:       # plain match, random bases
            seqs[i], cigs[i] = _random_read(rng, L), [(L << 4) | M]
        elif shape == 1:     # a repeat tract over most of the read (count < 256 by the tract's size)
            t = min(L * 7 // 8, 250 * len(unit))
            a = int(rng.integers(0, L - t + 1))
            seqs[i], cigs[i] = _random_read(rng, a) + _tract(rng, unit, t, 0.97) + _random_read(rng, L - a - t), [(L << 4) | M]
        elif shape in (2, 3, 4):   # clipped on the left / right / both, the clip a repeat
            cl = int(rng.choice([5, 17, 80, 300, min(600, L - 20)])) if shape in (2, 4) else 0
            cr = int(rng.choice([9, 17, 120, 400, min(700, L - cl - 10)])) if shape in (3, 4) else 0
            cl, cr = min(cl, 250 * len(unit), L // 2 - 5), min(cr, 250 * len(unit), L // 2 - 5)
            mid = L - cl - cr
            seqs[i] = _tract(rng, unit, cl, 0.98) + _random_read(rng, mid) + _tract(rng, unit, cr, 0.98)
            cigs[i] = ([(cl << 4) | S] if cl else []) + [(mid << 4) | M] + ([(cr << 4) | S] if cr else [])
            mapq[i] = 60 if rng.random() < 0.8 else 3
        elif shape == 5:     # one single soft-clip op (both of add_soft's iterations look at it)
            seqs[i], cigs[i] = _tract(rng, unit, L, 0.9)[: min(L, 200 * len(unit))].ljust(L, "A")[:L], [(L << 4) | S]
            seqs[i] = _random_read(rng, L - min(L, 200 * len(unit))) + _tract(rng, unit, min(L, 200 * len(unit)), 0.97)
        else:                # a class seen more than 255 times in a read that does not pass: wrapped bins, no unit
            seqs[i], cigs[i] = _tract(rng, "AC", L * 55 // 100, 1.0) + _random_read(rng, L - L * 55 // 100), [(L << 4) | M]
    qn = [rec.qname(i) for i in range(n)]
    out = RecordBatch.from_fields(rec.tid, rec.pos, rec.mtid, rec.mpos, rec.flag, mapq, cigs, seqs, qn, isize=rec.isize, targets=rec.targets)
    return out, np.sort(pick)


@pytest.mark.gpu
def test_batches_with_long_reads_through_the_abi(ctx, oracle):
    """strl_score_reads and the whole extract on batches of 150-base reads with 40 long ones among them == the oracle: the scorer
    words of every read, the soft-clip records, the treads"""
    rng = np.random.default_rng(5)
    rec0, g = synth.synth_wgs(3000, seed=19, contig_len=600_000)
    rec, pick = _mix_long(rec0, rng, 40)
    assert int(rec.l_seq.max()) > 2000
    med = oracle.median(synth.frag_hist(rec))
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    opts = oracle.make_opts(med, 0.8, 40)
    whole, soft, st = ctx.score_reads(rec)
    exp_whole, exp_soft = oracle_words(oracle, rec, g, opts)
    assert np.array_equal(whole, exp_whole), [(int(i), int(rec.l_seq[i]), unpack_result(whole[i]), unpack_result(exp_whole[i])) for i in np.nonzero(whole != exp_whole)[0][:5]]
    items = soft_items_expected(rec, exp_whole, 40)
    assert soft["read_side"].tolist() == [(i << 1) | s for i, s in items]
    assert soft["res_first"].tolist() == [exp_soft[it][0] for it in items]
    assert soft["res_after"].tolist() == [exp_soft[it][1] for it in items]
    assert soft["seg_len"].tolist() == [min(int(rec.cigar[int(rec.cigar_off[i]) if s == 0 else int(rec.cigar_off[i + 1]) - 1]) >> 4, int(rec.l_seq[i])) for i, s in items]
    longs = set(pick.tolist())
    assert sum(1 for i in pick if whole[i] >> 16) >= 3 and sum(1 for i, s in items if i in longs) >= 5
    got, _ = ctx.extract(rec)
    exp = oracle.extract(rec, g, opts)
    ok, why = treads_equal(got, exp)
    assert ok and len(exp) > 40, why
    # the same in chunks (strl_extract_begin / add / finish: what the CLI's host reader drives), chunk borders between mates
    edges = [0, 1001, 1002, 4000, rec.n]
    keep, chunks = [], []
    for a, b in zip(edges[:-1], edges[1:]):
        part = rec.slice(a, b)
        soa = api.Soa(part)
        rows, qh = soa.pair_rows()
        keep.append((part, soa, rows, qh))
        chunks.append((soa.c_struct(), api.CPairSoa(rows.ctypes.data, qh.ctypes.data)))
    ctx.extract_chunks(chunks, int((rec.tid < 0).sum()))
    got2, _ = ctx.treads_fetch()
    ok, why = treads_equal(got2, exp)
    assert ok, why


@pytest.mark.gpu
def test_a_long_read_whose_count_does_not_fit_a_byte_is_the_references_assert(ctx, oracle):
    """doAssert repeat_count < 256 (extract.nim:72): a 700-base homopolymer ends the reference; here STRL_ERR_ASSERT"""
    rec0, g = synth.synth_wgs(200, seed=3, contig_len=100_000)
    seqs = [rec0.sequence(i) for i in range(rec0.n)]
    cigs = [[int(c) for c in rec0.cigar[int(rec0.cigar_off[i]):int(rec0.cigar_off[i + 1])]] for i in range(rec0.n)]
    seqs[0], cigs[0] = "A" * 700, [(700 << 4) | 0]
    rec = RecordBatch.from_fields(rec0.tid, rec0.pos, rec0.mtid, rec0.mpos, rec0.flag, rec0.mapq, cigs, seqs, [rec0.qname(i) for i in range(rec0.n)],
                                  isize=rec0.isize, targets=rec0.targets)
    ctx.set_opts(0.8, 40, 350)
    ctx.set_genome(None)
    whole, _, _ = ctx.score_reads(rec)
    assert unpack_result(whole[0]) == ("A", 700, False)
    with pytest.raises(api.StrlingError) as e:
        ctx.extract(rec)
    assert "256" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("front", ["device", "host"])
def test_extract_of_a_bam_with_long_records(tmp_path, oracle, front):
    """`strling extract` on a BAM of 150-base reads with 600 .. 7000-base records among them, through the device front end
    (inflate, record scan, parse on the GPU; tiny chunks so that several chunks hold long records) and through the host
    reader: the .bin is the oracle's, byte for byte"""
    rng = np.random.default_rng(11)
    rec0, g = synth.synth_wgs(4000, seed=23, contig_len=700_000)
    rec, pick = _mix_long(rec0, rng, 60)
    bam = str(tmp_path / "long.bam")
    hdr = bamio.write_bam(bam, rec)
    bed = str(tmp_path / "ref.fa.str")
    bamio.write_genome_bed(bed, g, rec.targets)
    out = str(tmp_path / "long.bin")
    env = dict(os.environ, STRL_CHUNK_BLOCKS="7")
    if front == "host":
        env["STRL_FRONT"] = "host"
    r = subprocess.run([CLI, "extract", "-g", bed, "-v", bam, out], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    frag = synth.frag_hist(rec)
    med = oracle.median(frag)
    exp_t = oracle.extract(rec, g, oracle.make_opts(med, 0.8, 40))
    assert len(exp_t) > 60
    exp = oracle.bin_write(0.8, 40, frag, hdr.rstrip("\0"), exp_t, rec.qname_off, rec.qnames)
    assert open(out, "rb").read() == exp
    long_ids = set(pick.tolist())
    assert sum(1 for t in exp_t if int(t["qname_id"]) in long_ids) >= 5      # treads that come from the host twin's words
