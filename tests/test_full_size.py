"""BASELINE.json's full batch size (2^25 reads per GPU, HBM-resident) checked through size-independent properties:
the batch is T tiles of one base sample, so every tile must reproduce the oracle's answer for the base sample."""
import numpy as np
import pytest

from strling_amd import api, synth
from helpers import oracle_words, soft_items_expected

pytestmark = pytest.mark.gpu


def test_full_size_batch_is_tilewise_identical_to_the_oracle(oracle):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    base_pairs, n_total = 2 ** 16, 2 ** 25
    rec, g = synth.synth_wgs(base_pairs, seed=1234)
    soa = api.Soa(rec)
    nb = soa.n
    tiles = n_total // nb
    n = nb * tiles
    stride16 = int(soa.seq_off[1] - soa.seq_off[0])
    seq_bytes = nb * stride16 * 16

    def tile(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev).view(dt).repeat(tiles)

    d = dict(tid=tile(soa.tid, torch.int32), pos=tile(soa.pos, torch.int32), end=tile(soa.end, torch.int32),
             l_seq=tile(soa.l_seq.view(np.int16), torch.int16), clip_l=tile(soa.clip_l.view(np.int16), torch.int16),
             clip_r=tile(soa.clip_r.view(np.int16), torch.int16), mapq=tile(soa.mapq, torch.uint8), cig=tile(soa.cig, torch.uint8))
    so = torch.from_numpy(soa.seq_off.astype(np.int64)).to(dev)
    d["seq_off"] = (so[None, :] + (torch.arange(tiles, device=dev, dtype=torch.int64) * (seq_bytes // 16))[:, None]).reshape(-1).to(torch.int32)
    d["seq4"] = torch.cat([torch.from_numpy(soa.seq4[:seq_bytes]).to(dev).repeat(tiles), torch.zeros(64, dtype=torch.uint8, device=dev)])
    whole = torch.zeros(n, dtype=torch.int32, device=dev)
    soft_cap = n // 8
    soft = torch.zeros((soft_cap, 4), dtype=torch.int32, device=dev)
    cs = api.CReadSoa(n, d["tid"].data_ptr(), d["pos"].data_ptr(), d["end"].data_ptr(), d["seq_off"].data_ptr(), d["l_seq"].data_ptr(),
                      d["clip_l"].data_ptr(), d["clip_r"].data_ptr(), d["mapq"].data_ptr(), d["cig"].data_ptr(), d["seq4"].data_ptr(),
                      d["seq4"].numel(), soa.max_l_seq, api.MEM_DEVICE)
    torch.cuda.synchronize()
    ctx = api.Context(0)
    med = api.frag_median(synth.frag_hist(rec))
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    n_soft, st = ctx.score_device(cs, whole.data_ptr(), soft.data_ptr(), soft_cap, sync=True)

    # the oracle on the base sample (1.3e5 reads)
    opts = oracle.make_opts(med, 0.8, 40)
    exp_whole, exp_soft = oracle_words(oracle, rec, g, opts)
    items = soft_items_expected(rec, exp_whole, 40)

    # 1. every tile carries the oracle's whole-read words
    w = whole.view(tiles, nb)
    exp_w = torch.from_numpy(exp_whole.view(np.int32)).to(dev)
    assert bool((w == exp_w[None, :]).all())
    # 2. counters add up
    assert st.n_reads == n and st.n_skipped + st.n_scored == n
    assert st.n_skipped == tiles * int(((exp_whole & 0x8000) != 0).sum())
    assert n_soft == tiles * len(items) and st.n_soft_items == n_soft
    # 3. the soft records are, tile by tile, the oracle's records (order on the device is unspecified: sort by read_side)
    s = soft[:n_soft].cpu().numpy().view(np.uint32)
    s = s[np.argsort(s[:, 0], kind="stable")]
    rs = s[:, 0].astype(np.int64)
    assert np.array_equal(rs, (np.repeat(np.arange(tiles, dtype=np.int64) * nb, len(items)) * 2 +
                               np.tile(np.array([(i << 1) | sd for i, sd in items], np.int64), tiles)))
    assert np.array_equal(s[:, 1], np.tile(np.array([exp_soft[it][0] for it in items], np.uint32), tiles))
    assert np.array_equal(s[:, 2], np.tile(np.array([exp_soft[it][1] for it in items], np.uint32), tiles))
    # 4. idempotence: a second pass over the same resident batch changes nothing
    w1 = whole.clone()
    torch.cuda.synchronize()      # the clone runs on torch's stream, the kernels on the context's own
    n_soft2, st2 = ctx.score_device(cs, whole.data_ptr(), soft.data_ptr(), soft_cap, sync=True)
    assert n_soft2 == n_soft and bool((whole == w1).all())

    # 5. clustering at the size such a batch yields: tiles of the base sample's treads on their own contigs
    base_t = oracle.extract(rec, g, opts)
    exp_b, exp_u = oracle.call_bounds(base_t, 1, api.frag_median(synth.frag_hist(rec), 0.99), min_support=5, max_clip_dist=int(0.5 * med))
    t = np.zeros(len(base_t) * tiles, api.TREAD_DTYPE)
    for f in t.dtype.names:
        t[f] = np.tile(base_t[f], tiles)
    n_contigs = len(rec.targets)
    shift = np.repeat(np.arange(tiles, dtype=np.int32) * n_contigs, len(base_t))
    t["tid"] = np.where(t["tid"] >= 0, t["tid"] + shift, -1)
    b, u, cst = ctx.cluster(t, api.MODE_CALL, api.frag_median(synth.frag_hist(rec), 0.99), min_support=5, max_clip_dist=int(0.5 * med))
    assert len(b) == tiles * len(exp_b) and len(exp_b) > 0
    key = lambda x: (int(x["tid"]) % n_contigs, int(x["left"]), int(x["right"]), bytes(x["repeat"]), int(x["left_most"]), int(x["right_most"]),
                     int(x["center_mass"]), int(x["n_left"]), int(x["n_right"]), int(x["n_total"]))
    exp_keys = sorted(key(x) for x in exp_b)
    per_tile = {}
    for x in b:
        per_tile.setdefault(int(x["tid"]) // n_contigs, []).append(key(x))
    assert len(per_tile) == tiles and all(sorted(v) == exp_keys for v in per_tile.values())
    # unplaced reads of all tiles fall into the same (tid = -1, unit) groups
    assert sorted((x["repeat"].decode(), int(x["count"])) for x in u) == sorted((r, c * tiles) for r, c in exp_u)


def test_joint_merge_at_s2_scale(ctx, oracle):
    """BASELINE.json configs[4] (S2: 50 samples, ~5x10^7 STR reads through `strling merge` semantics): 64 tiles of a
    50-sample base set on their own contigs; every tile must give the oracle's rows for the base set, in the
    reference's row order within the tile's groups.  19 200 (tid, unit) groups also push the emulated Nim table
    through its enlarge path (8192 initial slots)."""
    import time
    base = synth.synth_treads(n_samples=50, n_loci=1200, seed=77, n_contigs=5, contig_len=2_000_000)
    tiles, n_contigs = 64, 5
    ot = np.zeros(len(base), oracle.TREAD_DTYPE)
    for f in base.dtype.names:
        ot[f] = base[f]
    exp_b, _ = oracle.call_bounds(ot, api.MODE_MERGE, 560, min_support=5, max_clip_dist=175)
    assert len(exp_b) > 1000
    t = np.tile(base, tiles)
    t["tid"] += np.repeat(np.arange(tiles, dtype=np.int32) * n_contigs, len(base))
    assert len(t) > 4.5e7
    t0 = time.time()
    b, u, st = ctx.cluster(t, api.MODE_MERGE, 560, min_support=5, max_clip_dist=175)
    dt = time.time() - t0
    assert len(b) == tiles * len(exp_b) and st.n_groups == tiles * len({(int(x["tid"]), bytes(x["repeat"])) for x in base})
    fields = ("left", "right", "left_most", "right_most", "center_mass", "n_left", "n_right", "n_total", "repeat")
    tile_of = b["tid"] // n_contigs
    exp_sorted = np.sort(np.array([tuple([int(x["tid"])] + [x[f] for f in fields]) for x in exp_b],
                                  dtype=[("tid", "i8")] + [(f, exp_b.dtype[f]) for f in fields]))
    for k in range(tiles):
        bk = b[tile_of == k]
        got = np.sort(np.array([tuple([int(x["tid"]) - k * n_contigs] + [x[f] for f in fields]) for x in bk], dtype=exp_sorted.dtype))
        assert np.array_equal(got, exp_sorted), k
    print(f"S2 scale: {len(t)} treads, {st.n_groups} groups, {st.n_clusters} clusters, {len(b)} bounds in {dt:.2f} s (host keys + device pass + row order)")


def test_full_size_distinct_reads_whole_path_matches_the_oracle(oracle):
    """BASELINE.json configs[1]+[2] at full size without tiling: 2^25 DISTINCT reads (64 independent sub-samples merged into
    one coordinate-sorted batch) through the whole device path -- scorer, soft-clip scan, pair logic, clustering -- and the
    oracle over the same 2^25 records: identical treads (same order) and identical -bounds rows.  Plus the scorer words of
    16 random 2^16-read slices, read by read."""
    rng = np.random.default_rng(99)
    rec, g = synth.synth_wgs_30x(64, 2 ** 18, seed=4321)
    assert rec.n == 2 ** 25
    frag = synth.frag_hist(rec)
    med = api.frag_median(frag)
    window, mcd = api.frag_median(frag, 0.99), int(0.5 * api.frag_median(frag, 0.5))
    ctx = api.Context(0)
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    soa = api.Soa(rec)
    rows, qh = soa.pair_rows()
    n_tail = int((rec.tid < 0).sum())
    ctx.extract_device(soa.c_struct(), api.CPairSoa(rows.ctypes.data, qh.ctypes.data), n_tail)
    got, st = ctx.treads_fetch()
    b, u, cst = ctx.cluster_resident(len(rec.targets), window, min_support=5, max_clip_dist=mcd, pos_bits=22)
    opts = oracle.make_opts(med, 0.8, 40)
    exp = oracle.extract(rec, g, opts)
    assert len(exp) > 400_000
    for f in ("tid", "position", "repeat", "flag", "split", "mapping_quality", "repeat_count", "align_length", "qname_id"):
        assert np.array_equal(np.asarray(got[f]), np.asarray(exp[f])), f
    eb, eu = oracle.call_bounds(exp, 1, window, min_support=5, max_clip_dist=mcd)
    assert len(eb) > 3000 and len(b) == len(eb)
    for f in ("tid", "left", "right", "left_most", "right_most", "center_mass", "n_left", "n_right", "n_total", "repeat"):
        assert np.array_equal(b[f], eb[f]), f                       # same rows in the reference's row order
    assert [(x["repeat"].decode(), int(x["count"])) for x in u] == [(r, int(k)) for r, k in eu]
    assert st.n_reads == rec.n and st.n_scored + st.n_skipped == rec.n
    # scorer words of random slices, every read
    for a in rng.integers(0, rec.n - 2 ** 16, size=16):
        part = rec.slice(int(a), int(a) + 2 ** 16)
        whole, soft, _ = ctx.score_reads(part)
        exp_whole, exp_soft = oracle_words(oracle, part, g, opts)
        assert np.array_equal(whole, exp_whole), int(a)
        items = soft_items_expected(part, exp_whole, 40)
        assert soft["read_side"].tolist() == [(i << 1) | s for i, s in items]
        assert soft["res_first"].tolist() == [exp_soft[it][0] for it in items]
        assert soft["res_after"].tolist() == [exp_soft[it][1] for it in items]
    ctx.close()
