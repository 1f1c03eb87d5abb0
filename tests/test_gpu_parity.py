"""Parity of the HIP path against the oracle, through the C ABI.  Needs a real MI355X (-m gpu)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from strling_amd import api, synth
from strling_amd.records import RecordBatch, unpack_result
from helpers import oracle_words, soft_items_expected, treads_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _as_api_treads(t):
    out = np.zeros(len(t), api.TREAD_DTYPE)
    for f in out.dtype.names:
        out[f] = t[f]
    return out


def test_native_library_is_loaded(ctx):
    maps = open("/proc/self/maps").read()
    assert "libstrling_amd.so" in maps


@pytest.mark.parametrize("k", KATS["get_repeat"], ids=lambda k: k["source"])
def test_reference_kats_through_the_kernels(ctx, k):
    rec = RecordBatch.from_sam(k["sam_header"], [k["sam"]])
    ctx.set_opts(k["proportion_repeat"], k["min_mapq"], 500)
    ctx.set_genome(None)
    whole, soft, st = ctx.score_reads(rec)
    unit, count, skipped = unpack_result(whole[0])
    assert (unit, count, skipped) == (k["expect_unit"], k["expect_count"], False)


@pytest.mark.parametrize("n_pairs,seed,p,q", [(30000, 1234, 0.8, 40), (8000, 7, 0.6, 20), (8000, 11, 0.9, 0)])
def test_score_reads_matches_oracle(ctx, oracle, n_pairs, seed, p, q):
    rec, g = synth.synth_wgs(n_pairs, seed=seed, contig_len=3_000_000)
    opts = oracle.make_opts(350, p, q)
    ctx.set_opts(p, q, 350)
    ctx.set_genome(g)
    whole, soft, st = ctx.score_reads(rec)
    exp_whole, exp_soft = oracle_words(oracle, rec, g, opts)
    assert np.array_equal(whole, exp_whole), np.nonzero(whole != exp_whole)[0][:10]
    items = soft_items_expected(rec, exp_whole, q)
    assert soft["read_side"].tolist() == [(i << 1) | s for i, s in items]
    assert soft["res_first"].tolist() == [exp_soft[it][0] for it in items]
    assert soft["res_after"].tolist() == [exp_soft[it][1] for it in items]
    assert st.n_reads == rec.n and st.n_skipped == int(((exp_whole & 0x8000) != 0).sum())
    assert st.n_scored + st.n_skipped == rec.n


def test_ragged_lengths_and_edge_cases(ctx, oracle):
    rng = np.random.default_rng(5)
    seqs, cig = [], []
    for L in [0, 1, 2, 3, 5, 6, 7, 15, 16, 17, 31, 32, 33, 64, 100, 149, 150, 151, 160, 161, 200, 250, 255, 256, 257, 300, 400, 510]:
        for kind in range(4):
            if kind == 0:
                s = "".join(rng.choice(list("ACGT"), L))
            elif kind == 1:
                u = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 7))))
                s = (u * (L + 6))[:L]
            elif kind == 2:
                s = "".join(rng.choice(list("ACGTN"), L))
            else:
                s = "N" * L
            seqs.append(s)
            if L >= 40 and kind < 2:
                c = int(rng.integers(1, L // 2))
                cig.append(f"{c}S{L - c}M" if rng.random() < 0.5 else f"{L - c}M{c}S")
            elif L > 0:
                cig.append(f"{L}M")
            else:
                cig.append("*")
    n = len(seqs)
    rec = RecordBatch.from_fields(tid=[0] * n, pos=list(range(100, 100 + n)), mtid=[0] * n, mpos=[5] * n, flag=[99] * n,
                                  mapq=[60] * n, cigars=cig, seqs=seqs, qnames=[f"r{i}" for i in range(n)])
    opts = oracle.make_opts(350, 0.8, 40)
    ctx.set_opts(0.8, 40, 350)
    ctx.set_genome(None)
    whole, soft, st = ctx.score_reads(rec)
    exp_whole, exp_soft = oracle_words(oracle, rec, None, opts)
    assert np.array_equal(whole, exp_whole), [(seqs[i], unpack_result(whole[i]), unpack_result(exp_whole[i])) for i in np.nonzero(whole != exp_whole)[0][:5]]
    items = soft_items_expected(rec, exp_whole, 40)
    assert soft["read_side"].tolist() == [(i << 1) | s for i, s in items]
    assert soft["res_first"].tolist() == [exp_soft[it][0] for it in items]
    assert soft["res_after"].tolist() == [exp_soft[it][1] for it in items]


@pytest.mark.parametrize("frac", [1.0, 0.3])
def test_reads_with_bases_that_are_not_acgt(ctx, oracle, frac):
    """The masks of such bases live in per-wave LDS slots (16 lanes) with a global spill behind them: with every read of a
    wave carrying one, both are in use; the repeats make the literal recount (which the masks feed) decide the result."""
    rng = np.random.default_rng(23)
    seqs, cig = [], []
    n = 6000
    for i in range(n):
        L = int(rng.choice([100, 150, 150, 150, 151, 250]))
        u = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 7))))
        a = int(rng.integers(0, L))
        s = list("".join(rng.choice(list("ACGT"), a)) + (u * L)[:L - a])
        if rng.random() < frac:
            for _ in range(int(rng.choice([1, 1, 1, 2, 5, 19, 22]))):
                s[int(rng.integers(0, L))] = str(rng.choice(list("NNNNRYKMSWBDHV")))
        seqs.append("".join(s))
        c = int(rng.integers(17, L // 2))
        cig.append([f"{L}M", f"{c}S{L - c}M", f"{L - c}M{c}S"][int(rng.integers(0, 3))])
    rec = RecordBatch.from_fields(tid=[0] * n, pos=list(range(100, 100 + n)), mtid=[0] * n, mpos=[5] * n, flag=[99] * n,
                                  mapq=[60] * n, cigars=cig, seqs=seqs, qnames=[f"r{i}" for i in range(n)])
    opts = oracle.make_opts(350, 0.8, 40)
    ctx.set_opts(0.8, 40, 350)
    ctx.set_genome(None)
    whole, soft, st = ctx.score_reads(rec)
    exp_whole, exp_soft = oracle_words(oracle, rec, None, opts)
    assert np.array_equal(whole, exp_whole), [(seqs[i], unpack_result(whole[i]), unpack_result(exp_whole[i])) for i in np.nonzero(whole != exp_whole)[0][:5]]
    items = soft_items_expected(rec, exp_whole, 40)
    assert soft["read_side"].tolist() == [(i << 1) | s for i, s in items]
    assert soft["res_first"].tolist() == [exp_soft[it][0] for it in items]
    assert soft["res_after"].tolist() == [exp_soft[it][1] for it in items]


def test_fresh_context_without_genome(oracle):
    """no strl_ctx_set_genome at all, and an explicit empty table, on a context that never held one: nothing is skipped"""
    rec, _ = synth.synth_wgs(300, seed=3, contig_len=100_000, n_contigs=2)
    opts = oracle.make_opts(350, 0.8, 40)
    exp_whole, _ = oracle_words(oracle, rec, None, opts)
    for explicit in (False, True):
        c = api.Context(0)
        c.set_opts(0.8, 40, 350)
        if explicit:
            c.set_genome(None)
        whole, soft, st = c.score_reads(rec)
        assert np.array_equal(whole, exp_whole) and st.n_skipped == 0


def test_empty_batch(ctx):
    rec = RecordBatch.from_fields([], [], [], [], [], [], [], [], [])
    ctx.set_opts(0.8, 40, 350)
    whole, soft, st = ctx.score_reads(rec)
    assert whole.size == 0 and soft.size == 0 and st.n_reads == 0
    b, u, cs = ctx.cluster(np.zeros(0, api.TREAD_DTYPE), api.MODE_MERGE, 500)
    assert b.size == 0


def test_read_longer_than_the_kernels_take_is_scored_and_one_beyond_the_columns_is_refused(ctx):
    """more than STRL_DEVICE_READ_LEN (510) bases: the host twin of the scorer (tests/test_long_reads.py); more than
    STRL_MAX_READ_LEN (65534, what the 16-bit length columns hold): STRL_ERR_ARG"""
    ctx.set_opts(0.8, 40, 350)
    ctx.set_genome(None)
    rec = RecordBatch.from_fields([0], [1], [0], [1], [99], [60], ["600M"], ["AC" * 300], ["x"])
    whole, _, _ = ctx.score_reads(rec)
    assert unpack_result(whole[0]) == ("CA", 299, False)
    rec = RecordBatch.from_fields([0], [1], [0], [1], [99], [60], ["65535M"], ["A" * 65535], ["x"])
    with pytest.raises(api.StrlingError):
        ctx.score_reads(rec)


@pytest.mark.parametrize("n_pairs,seed,p,q", [(30000, 4321, 0.8, 40), (10000, 3, 0.7, 30)])
def test_extract_matches_oracle(ctx, oracle, n_pairs, seed, p, q):
    """tread records identical to the oracle's extract (extract.nim:308-329), same order."""
    rec, g = synth.synth_wgs(n_pairs, seed=seed, contig_len=3_000_000)
    med = oracle.median(synth.frag_hist(rec))
    ctx.set_opts(p, q, med)
    ctx.set_genome(g)
    got, st = ctx.extract(rec)
    exp = oracle.extract(rec, g, oracle.make_opts(med, p, q))
    assert len(exp) > 300
    ok, why = treads_equal(got, exp)
    assert ok, why


@pytest.mark.parametrize("k", KATS["cluster"], ids=lambda k: k["source"])
def test_cluster_kats_through_the_kernels(ctx, oracle, k):
    t = np.zeros(len(k["positions"]), api.TREAD_DTYPE)
    t["tid"] = k["tid"]
    t["repeat"] = k["repeat"].encode() or b"A"
    t["position"] = k["positions"]
    t["split"] = k["splits"]
    b, u, st = ctx.cluster(t, api.MODE_CALL, k["max_dist"], min_support=k["min_supporting_reads"])
    assert [int(x) for x in b["n_total"]] == [e["n"] for e in k["expect"]]


@pytest.mark.parametrize("k", KATS["bounds"], ids=lambda k: k["source"])
def test_bounds_kats_through_the_kernels(ctx, k):
    t = np.zeros(len(k["positions"]), api.TREAD_DTYPE)
    t["tid"] = 1
    t["repeat"] = b"ATG"
    t["position"] = k["positions"]
    t["split"] = k["splits"]
    # one cluster holding every read: window larger than the span, min_support 1
    b, u, st = ctx.cluster(t, api.MODE_CALL, 2_000_000, min_support=1, max_clip_dist=k["max_clip_dist"])
    assert len(b) == 1
    for f, v in k.get("expect", {}).items():
        if f in ("left_most", "right_most"):
            continue     # Cluster.left_most/right_most come from the sweep here, not from a bare Cluster literal
        assert int(b[f][0]) == v, f
    assert b["left"][0] < b["right"][0]


@pytest.mark.parametrize("mode,n_samples,n_loci,seed,min_support", [(api.MODE_MERGE, 6, 500, 1000, 5), (api.MODE_CALL, 1, 800, 2000, 3),
                                                                     (api.MODE_MERGE, 3, 300, 5, 2)])
def test_cluster_matches_oracle(ctx, oracle, mode, n_samples, n_loci, seed, min_support):
    """-bounds rows identical to the oracle's merge/call clustering, same row order."""
    t = synth.synth_treads(n_samples=n_samples, n_loci=n_loci, seed=seed, contig_len=2_000_000)
    if mode == api.MODE_CALL:   # add unplaced groups
        t["tid"][::97] = -1
        t["position"][::97] = 0
    ot = np.zeros(len(t), oracle.TREAD_DTYPE)
    for f in t.dtype.names:
        ot[f] = t[f]
    exp_b, exp_u = oracle.call_bounds(ot, mode, 560, min_support=min_support, max_clip_dist=175)
    b, u, st = ctx.cluster(t, mode, 560, min_support=min_support, max_clip_dist=175)
    assert len(exp_b) > 20
    rows = [api.bounds_row(x, f"chr{int(x['tid']) + 1}") for x in b]
    exp_rows = [oracle.bounds_row(x, f"chr{int(x['tid']) + 1}") for x in exp_b]
    assert sorted(rows) == sorted(exp_rows)          # same set of loci
    assert rows == exp_rows                          # and the reference's row order (Nim Table slot order)
    assert [(x["repeat"].decode(), int(x["count"])) for x in u] == exp_u


def test_cluster_large_positions_and_ties(ctx, oracle):
    """uint32 wrap in left_most (cluster.nim:344), ties between modal clip positions (CountTable.largest slot order)."""
    rng = np.random.default_rng(17)
    rows = []
    for locus in range(300):
        base = int(rng.integers(0, 400)) if locus % 3 == 0 else int(rng.integers(1000, 4_000_000_000))
        k = int(rng.integers(2, 5))
        for j in range(k):   # k tied left-clip positions and k tied right-clip positions
            for _ in range(3):
                rows.append((base + 40 + j, 0))
                rows.append((base + j, 1))
        for _ in range(6):
            rows.append((max(0, base + int(rng.integers(-300, 300))), 3))
    t = np.zeros(len(rows), api.TREAD_DTYPE)
    perm = rng.permutation(len(rows))
    t["position"] = np.array([r[0] for r in rows], np.uint64)[perm].astype(np.uint32)
    t["split"] = np.array([r[1] for r in rows])[perm]
    t["tid"] = 2
    t["repeat"] = b"AC"
    t["qname_id"] = rng.integers(0, 3, len(rows))
    ot = np.zeros(len(t), oracle.TREAD_DTYPE)
    for f in t.dtype.names:
        ot[f] = t[f]
    for mode in (api.MODE_MERGE, api.MODE_CALL):
        exp_b, _ = oracle.call_bounds(ot, mode, 500, min_support=3, max_clip_dist=200)
        b, _, _ = ctx.cluster(t, mode, 500, min_support=3, max_clip_dist=200)
        assert len(exp_b) > 100
        for f in ("left", "right", "left_most", "right_most", "center_mass", "n_left", "n_right", "n_total"):
            assert np.array_equal(b[f], exp_b[f]), f


@pytest.mark.parametrize("force_two", [False, True])
def test_cluster_many_contigs_two_pass_sort(ctx, oracle, monkeypatch, force_two):
    """tid >= 2^17 makes the (tid, unit) key wider than 32 bits.  One 64-bit composite key still holds it; when
    group key + position bits exceed 64 (forced here) the single sort is replaced by the position sort + stable group
    sort; rows must not change."""
    if force_two:
        monkeypatch.setenv("STRL_CLUSTER_TWO_SORTS", "1")
    t = synth.synth_treads(n_samples=2, n_loci=200, seed=9, contig_len=1_000_000)
    t["tid"] = t["tid"] + 140000
    ot = np.zeros(len(t), oracle.TREAD_DTYPE)
    for f in t.dtype.names:
        ot[f] = t[f]
    exp_b, _ = oracle.call_bounds(ot, api.MODE_MERGE, 560, min_support=3, max_clip_dist=175)
    b, _, st = ctx.cluster(t, api.MODE_MERGE, 560, min_support=3, max_clip_dist=175)
    assert len(exp_b) > 10
    for f in ("tid", "left", "right", "left_most", "right_most", "center_mass", "n_left", "n_right", "n_total", "repeat"):
        assert np.array_equal(b[f], exp_b[f]), f


def _pair_soa(rec, soa):
    """the pairing arrays of strl_pair_soa for a host batch"""
    keep = soa.pair_rows()
    return api.CPairSoa(keep[0].ctypes.data, keep[1].ctypes.data), keep


@pytest.mark.parametrize("n_pairs,seed,p,q", [(30000, 4321, 0.8, 40), (6000, 17, 0.65, 10)])
def test_extract_device_and_resident_clustering_match_oracle(ctx, oracle, n_pairs, seed, p, q):
    """scoring + pair logic + clustering all on the device, nothing but the final rows crossing to the host"""
    rec, g = synth.synth_wgs(n_pairs, seed=seed, contig_len=3_000_000)
    frag = synth.frag_hist(rec)
    med = oracle.median(frag)
    ctx.set_opts(p, q, med)
    ctx.set_genome(g)
    soa = api.Soa(rec)
    cp, keep = _pair_soa(rec, soa)
    n_tail = int((rec.tid < 0).sum())
    ctx.extract_device(soa.c_struct(), cp, n_tail)
    got, st = ctx.treads_fetch()
    exp = oracle.extract(rec, g, oracle.make_opts(med, p, q))
    ok, why = treads_equal(got, exp)
    assert ok and len(exp) > 100, why
    assert st.n_reads == rec.n and st.n_scored + st.n_skipped == rec.n
    window = api.frag_median(frag, 0.99)
    mcd = int(0.5 * api.frag_median(frag, 0.5))
    b, u, cs = ctx.cluster_resident(len(rec.targets), window, min_support=3, max_clip_dist=mcd, pos_bits=24)
    eb, eu = oracle.call_bounds(exp, 1, window, min_support=3, max_clip_dist=mcd)
    assert [api.bounds_row(x, "c") for x in b] == [oracle.bounds_row(x, "c") for x in eb] and len(eb) > 5
    assert [(x["repeat"].decode(), int(x["count"])) for x in u] == [(r, int(k)) for r, k in eu]
    # the same rows from the host-array entry point
    b2, u2, _ = ctx.cluster(got, api.MODE_CALL, window, min_support=3, max_clip_dist=mcd)
    assert np.array_equal(b, b2) and np.array_equal(u, u2)


@pytest.mark.parametrize("resident", [False, True])
def test_overlapped_clustering_of_one_batch_and_extract_of_the_next(ctx, oracle, resident):
    """(resident: the batches' arrays live on the device, so the pair logic of a batch runs on a side stream as well and the
    context rotates through its buffer sets.)  The pipeline the bench times: batch A's clustering runs asynchronously on the context's side stream while batch B's
    extract is enqueued; collecting A's rows after B's extract was issued gives A's rows, B's treads are B's, and B's own
    clustering afterwards is B's -- three rounds, alternating batches, against the oracle."""
    batches = []
    for seed, n_pairs in ((7001, 20000), (7002, 26000)):
        rec, g = synth.synth_wgs(n_pairs, seed=seed, contig_len=2_000_000)
        frag = synth.frag_hist(rec)
        med = oracle.median(frag)
        soa = api.Soa(rec)
        cp, keep = _pair_soa(rec, soa)
        window, mcd = api.frag_median(frag, 0.99), int(0.5 * api.frag_median(frag, 0.5))
        exp = oracle.extract(rec, g, oracle.make_opts(med, 0.8, 40))
        eb, eu = oracle.call_bounds(exp, 1, window, min_support=3, max_clip_dist=mcd)
        cs = soa.c_struct()
        if resident:
            import torch
            from helpers import device_batch
            # (one batch with the packed strl_read_meta rows, one with the five columns only: both gathers of the skip-predicate pass)
            cs, cp, keep_dev = device_batch(torch, torch.device("cuda", 0), soa, keep[0], keep[1], with_meta=len(batches) % 2 == 0)
            keep = (keep, keep_dev)
        batches.append(dict(rec=rec, g=g, med=med, soa=soa, cs=cs, cp=cp, keep=keep, window=window, mcd=mcd, exp=exp,
                            rows=[oracle.bounds_row(x, "c") for x in eb], unpl=[(r, int(k)) for r, k in eu]))
    assert all(len(b["rows"]) > 5 for b in batches) and batches[0]["rows"] != batches[1]["rows"]

    def extract(b):
        ctx.set_opts(0.8, 40, b["med"])
        ctx.set_genome(b["g"])
        ctx.extract_device(b["cs"], b["cp"], int((b["rec"].tid < 0).sum()))
        if resident:
            assert ctx.tail_stream() != ctx.stream      # this batch's pair logic runs on a side stream

    def cluster_async(b):
        ctx.cluster_resident(len(b["rec"].targets), b["window"], min_support=3, max_clip_dist=b["mcd"], pos_bits=24, fetch=False)

    def check_rows(b, got):
        rows, unpl, _ = got
        assert [api.bounds_row(x, "c") for x in rows] == b["rows"]
        assert [(x["repeat"].decode(), int(x["count"])) for x in unpl] == b["unpl"]

    extract(batches[0])
    for r in range(5):
        cur, nxt = batches[r % 2], batches[(r + 1) % 2]
        cluster_async(cur)              # side stream
        extract(nxt)                    # main stream: scorer of the next batch overlaps; its pair logic waits on the device
        check_rows(cur, ctx.cluster_collect())
        got, _ = ctx.treads_fetch()
        ok, why = treads_equal(got, nxt["exp"])
        assert ok, why
    # ... and the way the bench drives it: many steps enqueued back to back without a single synchronisation in between
    # (buffer sets are reused while earlier steps are still in flight), only the last step's results looked at
    for rounds in (7, 8):
        last = None
        for r in range(rounds):
            last = batches[r % 2]
            extract(last)
            cluster_async(last)
        check_rows(last, ctx.cluster_collect())
        got, _ = ctx.treads_fetch()
        ok, why = treads_equal(got, last["exp"])
        assert ok, why


def test_resident_clustering_before_and_after_the_bin_order_sort(ctx, oracle):
    """strl_cluster_resident straight behind strl_extract_device clusters the treads as the pair logic emitted them
    (unordered, first appearance from the emission keys); after a fetch it clusters the ordered array: same rows either way.
    The batch is nearly all STR reads: most qname groups emit."""
    rec, g = synth.synth_wgs(6000, seed=91, contig_len=400_000, str_frac=0.9, n_contigs=4)
    frag = synth.frag_hist(rec)
    med = oracle.median(frag)
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    soa = api.Soa(rec)
    cp, keep = _pair_soa(rec, soa)
    window, mcd = api.frag_median(frag, 0.99), int(0.5 * api.frag_median(frag, 0.5))
    ctx.extract_device(soa.c_struct(), cp, int((rec.tid < 0).sum()))
    b1, u1, _ = ctx.cluster_resident(len(rec.targets), window, min_support=3, max_clip_dist=mcd)          # unordered treads
    got, _ = ctx.treads_fetch()                                                                             # orders them
    b2, u2, _ = ctx.cluster_resident(len(rec.targets), window, min_support=3, max_clip_dist=mcd)          # ordered treads
    exp = oracle.extract(rec, g, oracle.make_opts(med, 0.8, 40))
    ok, why = treads_equal(got, exp)
    assert ok and len(exp) > 3000, why
    eb, eu = oracle.call_bounds(exp, 1, window, min_support=3, max_clip_dist=mcd)
    rows = [oracle.bounds_row(x, "c") for x in eb]
    assert [api.bounds_row(x, "c") for x in b1] == rows and [api.bounds_row(x, "c") for x in b2] == rows and len(rows) > 5
    assert [(x["repeat"].decode(), int(x["count"])) for x in u1] == [(r, int(k)) for r, k in eu]
    assert np.array_equal(u1, u2)


def test_pair_logic_corner_cases_on_the_device(ctx, oracle):
    """qname groups the reference treats specially: a third record with a qname already paired, a one-op soft clip
    (both loop iterations of add_soft look at cigar[0]), secondary / supplementary records, an unpaired read, equal
    start positions, the unmapped tail that is visited twice."""
    A, C_, T = "A" * 150, "CAG" * 50, "ACGT" * 37 + "AC"
    rows = [
        # tid pos mtid mpos flag mapq cigar seq qname
        (0, 100, 0, 400, 99, 60, "150M", C_, "p1"), (0, 400, 0, 100, 147, 60, "150M", T, "p1"),
        (0, 120, 0, 120, 99, 60, "150M", C_, "same"), (0, 120, 0, 120, 147, 60, "150M", A, "same"),
        (0, 130, 0, 500, 99, 60, "150S", C_, "oneop"), (0, 500, 0, 130, 147, 60, "150M", T, "oneop"),
        (0, 140, 0, 600, 99, 60, "100M50S", T[:100] + "AC" * 25, "clip"), (0, 600, 0, 140, 147, 60, "40S110M", "TG" * 20 + T[:110], "clip"),
        (0, 150, 0, 700, 99 | 0x100, 60, "150M", C_, "p1"), (0, 160, 0, 700, 99 | 0x800, 60, "150M", C_, "p1"),
        (0, 170, 0, 100, 99, 60, "150M", C_, "p1"),           # third primary record of a qname
        (0, 180, -1, -1, 0, 60, "150M", C_, "lonely"),
        (1, 50, 0, 90, 147, 30, "150M", A, "xchrom"), (0, 90, 1, 50, 99, 10, "150M", C_, "xchrom"),
        (-1, -1, -1, -1, 77, 0, "*", C_, "un1"), (-1, -1, -1, -1, 141, 0, "*", C_, "un1"),
        (-1, -1, -1, -1, 77, 0, "*", A, "un2"), (-1, -1, -1, -1, 141, 0, "*", T, "un2"),
    ]
    rows.sort(key=lambda r: ((1 << 40) if r[0] < 0 else (r[0] << 32) + max(r[1], 0)))
    rec = RecordBatch.from_fields(*[[r[j] for r in rows] for j in range(9)])
    for p, q in ((0.8, 40), (0.5, 0)):
        ctx.set_opts(p, q, 350)
        ctx.set_genome(None)
        got, _ = ctx.extract(rec)
        exp = oracle.extract(rec, None, oracle.make_opts(350, p, q))
        ok, why = treads_equal(got, exp)
        assert ok and len(exp) >= 8, why
        soa = api.Soa(rec)
        cp, keep = _pair_soa(rec, soa)
        ctx.extract_device(soa.c_struct(), cp, int((rec.tid < 0).sum()))
        got2, _ = ctx.treads_fetch()
        ok, why = treads_equal(got2, exp)
        assert ok, why


@pytest.mark.parametrize("n", [16, 200, 700])
def test_many_records_under_one_qname(ctx, oracle, n):
    """one qname on n primary records: up to 512 join items the device replays the run (a block of its own, pair_long_kernel);
    beyond, it reports the run and strl_extract repeats the batch on the host's string-keyed Cache"""
    C_ = "CAG" * 50
    rec = RecordBatch.from_fields([0] * n, list(range(100, 100 + n)), [0] * n, [5000] * n, [99] * n, [60] * n, ["150M"] * n, [C_] * n, ["dup"] * n)
    ctx.set_opts(0.8, 40, 350)
    ctx.set_genome(None)
    exp = oracle.extract(rec, None, oracle.make_opts(350, 0.8, 40))
    soa = api.Soa(rec)
    cp, keep = _pair_soa(rec, soa)
    ctx.extract_device(soa.c_struct(), cp, 0)
    if n <= 512:
        got, _ = ctx.treads_fetch()
        ok, why = treads_equal(got, exp)
        assert ok, why
    else:
        with pytest.raises(api.StrlingError):
            ctx.treads_fetch()
    got, _ = ctx.extract(rec)
    ok, why = treads_equal(got, exp)
    assert ok, why


@pytest.mark.parametrize("cuts", [[], [1], [5000, 5001, 20000, 43000], [59999]])
def test_chunked_extract_matches_oracle(ctx, oracle, cuts):
    """strl_extract_begin / _add / _finish: chunks scored as they arrive, the pair logic once over the whole input --
    pairs that straddle chunk boundaries, empty chunks and the twice-visited tail included"""
    rec, g = synth.synth_wgs(30000, seed=77, contig_len=2_000_000)
    med = oracle.median(synth.frag_hist(rec))
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    edges = [0] + cuts + [rec.n]
    keep, chunks = [], []
    for a, b in zip(edges[:-1], edges[1:]):
        part = rec.slice(a, b)
        soa = api.Soa(part)
        rows, qh = soa.pair_rows()
        keep.append((part, soa, rows, qh))
        chunks.append((soa.c_struct(), api.CPairSoa(rows.ctypes.data, qh.ctypes.data)))
    ctx.extract_chunks(chunks, int((rec.tid < 0).sum()))
    got, st = ctx.treads_fetch()
    exp = oracle.extract(rec, g, oracle.make_opts(med, 0.8, 40))
    ok, why = treads_equal(got, exp)
    assert ok and len(exp) > 300, why
    assert st.n_reads == rec.n and st.n_scored + st.n_skipped == rec.n


def test_segment_scorer_fused_and_split_launches_agree(tmp_path):
    """Soft-clip segments of the short-read class are scored by ONE launch (both stages); STRL_SPLIT_SEGMENTS=1 selects the three
    launches (stage A, survivor compaction, stage B) the long-read classes still use.  Same records out of both, in a
    process each (the switch is read once)."""
    import subprocess, sys
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from strling_amd import api, synth\n"
        "rec, g = synth.synth_wgs(20000, seed=77, contig_len=3_000_000)\n"
        "c = api.Context(0); c.set_opts(0.8, 40, 350); c.set_genome(g)\n"
        "whole, soft, st = c.score_reads(rec)\n"
        "np.save(sys.argv[1] + '_w.npy', whole); np.save(sys.argv[1] + '_s.npy', soft)\n"
    ) % os.path.dirname(HERE)
    outs = []
    for name, env in (("fused", {}), ("split", {"STRL_SPLIT_SEGMENTS": "1"})):
        base = str(tmp_path / name)
        e = dict(os.environ, **env)
        e.pop("STRL_SPLIT_SEGMENTS", None) if not env else None
        subprocess.run([sys.executable, "-c", script, base], check=True, env=e)
        outs.append((np.load(base + "_w.npy"), np.load(base + "_s.npy")))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert len(outs[0][1]) > 100 and outs[0][1].tobytes() == outs[1][1].tobytes()
