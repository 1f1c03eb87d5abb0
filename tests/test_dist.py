"""World-size-2 gloo run of the sharded extract (strling_amd/dist.py): both ranks score their own record range
(here with the oracle's words, since there is no GPU), exchange only the hot qname groups, replay the pair logic
and must reproduce the single-process treads exactly.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from strling_amd import api, dist as sdist, synth
    from oracle import oracle as O
    from helpers import oracle_words, soft_items_expected
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec, g = synth.synth_wgs(5000, seed=77, contig_len=600_000)
        med = O.median(synth.frag_hist(rec))
        opts = O.make_opts(med, 0.8, 40)
        lo, hi = sdist.shard_bounds(rec.n, world)[rank]
        shard = sdist.slice_records(rec, lo, hi)

        def score_fn(r):   # stands in for api.Context.score_reads (same packed outputs)
            whole, softd = oracle_words(O, r, g, opts)
            items = soft_items_expected(r, whole, 40)
            soft = np.zeros(len(items), api.SOFT_DTYPE)
            for j, (i, side) in enumerate(items):
                soft[j] = ((i << 1) | side, softd[(i, side)][0], softd[(i, side)][1], 0)
            return whole, soft

        tail_start = rec.n - int((rec.tid[::-1] < 0).cumprod().sum())
        t = sdist.extract_sharded(score_fn, shard, lo, rec.n, tail_start, (0.8, 40, med))
        exp = O.extract(rec, g, opts)
        ok = len(t) == len(exp) and all(np.array_equal(t[f], exp[f]) for f in
                                        ("tid", "position", "repeat", "flag", "split", "mapping_quality", "repeat_count", "align_length", "qname_id"))
        q.put((rank, bool(ok), len(t), len(exp)))
    finally:
        dist.destroy_process_group()


def test_sharded_extract_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    for rank, ok, n, ne in res:
        assert ok and n == ne and n > 100, (rank, ok, n, ne)


def _cluster_worker(rank, world, port, q, use_gpu):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from strling_amd import api, dist as sdist, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for mode, seed in ((api.MODE_MERGE, 31), (api.MODE_CALL, 32)):
            t = synth.synth_treads(n_samples=4, n_loci=300, seed=seed, n_contigs=6, contig_len=500_000)
            if mode == api.MODE_CALL:
                t["tid"][::53] = -1
                t["position"][::53] = 0

            def to_oracle(x):
                o = np.zeros(len(x), O.TREAD_DTYPE)
                for f in x.dtype.names:
                    o[f] = x[f]
                return o

            if use_gpu:
                ctx = api.Context(0)
                fn = lambda x: ctx.cluster(x, mode, 560, min_support=3, max_clip_dist=175)
            else:     # the oracle stands in for the device clustering of a rank's groups (same row layout)
                def fn(x):
                    b, u = O.call_bounds(to_oracle(x), mode, 560, min_support=3, max_clip_dist=175)
                    bb = np.zeros(len(b), api.BOUNDS_DTYPE)
                    for f in bb.dtype.names:
                        bb[f] = b[f]
                    uu = np.zeros(len(u), api.UNPLACED_DTYPE)
                    uu["repeat"] = [r.encode() for r, _ in u]
                    uu["count"] = [c for _, c in u]
                    return bb, uu
            lo, hi = sdist.shard_bounds(len(t), world)[rank]
            b, u = sdist.cluster_sharded(fn, t[lo:hi], mode)
            eb, eu = O.call_bounds(to_oracle(t), mode, 560, min_support=3, max_clip_dist=175)
            rows = [api.bounds_row(x, f"chr{int(x['tid']) + 1}") for x in b]
            exp_rows = [O.bounds_row(x, f"chr{int(x['tid']) + 1}") for x in eb]
            ok = ok and rows == exp_rows and len(rows) > 20
            ok = ok and [(x["repeat"].decode(), int(x["count"])) for x in u] == eu
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_cluster(use_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_cluster_worker, args=(r, 2, port, q, use_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)], res


def test_sharded_clustering_row_order_matches_single_process():
    """two ranks cluster disjoint sets of (tid, unit) groups; the gathered rows come back in the reference's order"""
    _run_cluster(False)


@pytest.mark.gpu
def test_sharded_clustering_on_the_device():
    _run_cluster(True)


def _device_exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from strling_amd import api, dist as sdist, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        recs = [synth.synth_wgs(12000, seed=500 + r, contig_len=1_500_000) for r in range(world)]     # every rank could make all of them
        frag = synth.frag_hist(recs[0][0])
        med, window, mcd = O.median(frag), O.median(frag, 0.99), int(0.5 * O.median(frag, 0.5))
        rec, g = recs[rank]
        ctx = api.Context(0)
        ctx.set_opts(0.8, 40, med)
        ctx.set_genome(g)
        soa = api.Soa(rec)
        rows, qh = soa.pair_rows()
        ctx.extract_device(soa.c_struct(), api.CPairSoa(rows.ctypes.data, qh.ctypes.data), int((rec.tid < 0).sum()))
        mine, _ = ctx.treads_fetch()
        ex = sdist.DeviceClusterExchange(ctx, world, rank, len(mine), dev)
        b, u = sdist.cluster_sharded_device(ex, dict(window=window, min_support=3, max_clip_dist=mcd, pos_bits=22), len(rec.targets))
        # single-process answer: the treads of all ranks in rank order through the oracle
        opts = O.make_opts(med, 0.8, 40)
        all_t = np.concatenate([O.extract(r, gg, opts) for r, gg in recs])
        eb, eu = O.call_bounds(all_t, 1, window, min_support=3, max_clip_dist=mcd)
        rows_got = [api.bounds_row(x, f"chr{int(x['tid']) + 1}") for x in b]
        rows_exp = [O.bounds_row(x, f"chr{int(x['tid']) + 1}") for x in eb]
        ok = rows_got == rows_exp and len(rows_exp) > 10 and [(x["repeat"].decode(), int(x["count"])) for x in u] == [(r, int(k)) for r, k in eu]
        # the asynchronous form (what bench.py times) leaves the same share on the device
        ex.step(len(rec.targets), window, 3, mcd, 22, fetch=False)
        ctx.sync()
        # ... and on DEVICE-RESIDENT input, where the pair logic, the .bin-order sort, the gather and the clustering of a batch
        # all run on a side stream of the context while the next batch's scorer runs on the main stream: two batches back
        # to back, my share of each collected afterwards, against the synchronous share
        share_b, share_u, _ = ex.step(len(rec.targets), window, 3, mcd, 22, fetch=True)

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d = {k: up(getattr(soa, k)) for k in ("tid", "pos", "end", "seq_off", "l_seq", "clip_l", "clip_r", "mapq", "cig", "seq4")}
        drows, dqh = up(rows.view(np.uint8)), up(qh)
        cs = api.CReadSoa(soa.n, d["tid"].data_ptr(), d["pos"].data_ptr(), d["end"].data_ptr(), d["seq_off"].data_ptr(), d["l_seq"].data_ptr(),
                          d["clip_l"].data_ptr(), d["clip_r"].data_ptr(), d["mapq"].data_ptr(), d["cig"].data_ptr(), d["seq4"].data_ptr(),
                          d["seq4"].numel(), soa.max_l_seq, api.MEM_DEVICE)
        cp = api.CPairSoa(drows.data_ptr(), dqh.data_ptr())
        torch.cuda.synchronize()
        n_tail = int((rec.tid < 0).sum())
        side_ok = True
        for _ in range(3):
            ctx.extract_device(cs, cp, n_tail)
            side_ok = side_ok and ctx.tail_stream() != ctx.stream          # the tail of this batch runs on a side stream
            ex.step(len(rec.targets), window, 3, mcd, 22, fetch=False)
            ctx.extract_device(cs, cp, n_tail)                            # the next batch's scorer overlaps the exchange
            b2, u2, _ = ctx.cluster_collect()
            side_ok = side_ok and np.array_equal(b2, share_b) and np.array_equal(u2, share_u)
        ctx.sync()
        q.put((rank, bool(ok and side_ok), len(rows_got), len(rows_exp)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_device_tread_exchange_and_owned_clustering():
    """world size 2 on one GPU: both ranks extract their own sample on the device, all-gather the resident tread buffers
    (gloo here, RCCL in bench.py -- same DeviceClusterExchange object), cluster the groups they own on the device; the merged
    rows equal the oracle's rows for the union, in the reference's order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_device_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, ok, n, ne in res:
        assert ok and n == ne, (rank, ok, n, ne)
