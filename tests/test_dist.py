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
