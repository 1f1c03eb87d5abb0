"""Randomised parity sweep of the HIP path against the oracle (longer than the test suite allows).
usage: python tests/fuzz/fuzz_parity.py [seconds]     (test infrastructure: uses the oracle; GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from strling_amd import api, synth
from oracle import oracle as O
from helpers import oracle_words, soft_items_expected, treads_equal

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
master = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
ctx = api.Context(0)
rng = np.random.default_rng(master)
print(f"fuzz_parity: master seed {master}, budget {budget:.0f} s (every case's own seed and parameters are in its assertion message; a progress line per 100 cases)", flush=True)
t0 = time.time()
n_cases = n_reads = n_treads = n_bounds = n_asserts = n_long_cases = 0
while time.time() - t0 < budget:
    L = int(rng.choice([76, 100, 101, 125, 150, 151, 160, 161, 200, 250, 256, 300, 400]))
    p = float(rng.choice([0.55, 0.6, 0.7, 0.8, 0.85, 0.9, 0.95]))
    q = int(rng.choice([0, 10, 20, 40, 60]))
    seed = int(rng.integers(1, 1 << 30))
    kw = dict(read_len=L, n_contigs=int(rng.choice([2, 3, 25])), contig_len=int(rng.choice([200_000, 3_000_000])),
              str_frac=float(rng.choice([0.01, 0.1, 0.3])), soft_frac=float(rng.choice([0.03, 0.2])),
              indel_frac=0.02, unmapped_frac=float(rng.choice([0.005, 0.05])), genome_overlap=float(rng.choice([0.01, 0.03, 0.3])))
    try:
        rec, g = synth.synth_wgs(6000, seed=seed, **kw)
    except (ValueError, IndexError):      # a parameter draw the generator cannot realise
        continue
    n_long = 0
    if rng.random() < 0.25:               # a few records of 511 .. 7000 bases among them: the host twin of the scorer (csrc/host_score.cpp)
        from test_long_reads import _mix_long
        n_long = int(rng.integers(1, 40))
        rec, _ = _mix_long(rec, np.random.default_rng(seed), n_long)
        n_long_cases += 1
    med = O.median(synth.frag_hist(rec))
    opts = O.make_opts(med, p, q)
    ctx.set_opts(p, q, med)
    ctx.set_genome(g)
    whole, soft, st = ctx.score_reads(rec)
    exp_whole, exp_soft = oracle_words(O, rec, g, opts)
    tag = f"L={L} p={p} q={q} seed={seed} long={n_long} {kw}"
    assert np.array_equal(whole, exp_whole), ("whole", tag, np.nonzero(whole != exp_whole)[0][:5])
    items = soft_items_expected(rec, exp_whole, q)
    assert soft["read_side"].tolist() == [(i << 1) | s for i, s in items], ("soft items", tag)
    assert soft["res_first"].tolist() == [exp_soft[it][0] for it in items], ("soft first", tag)
    assert soft["res_after"].tolist() == [exp_soft[it][1] for it in items], ("soft after", tag)
    try:
        got, _ = ctx.extract(rec)
    except api.StrlingError as e:              # a homopolymer read of >= 256 bases: the reference's doAssert (extract.nim:72)
        assert "extract.nim:72" in str(e) and (L >= 256 or n_long), (tag, str(e))
        n_asserts += 1
        continue
    exp = O.extract(rec, g, opts)
    ok, why = treads_equal(got, exp)
    assert ok, ("extract", tag, why)
    # clustering of those treads, both modes
    for mode in (api.MODE_CALL, api.MODE_MERGE):
        t = got.copy()
        if mode == api.MODE_MERGE:
            t["qname_id"] = rng.integers(0, 4, len(t))
        ot = np.zeros(len(t), O.TREAD_DTYPE)
        for f in t.dtype.names:
            ot[f] = t[f]
        w = int(rng.choice([300, 560, 900]))
        ms = int(rng.choice([2, 3, 5]))
        eb, eu = O.call_bounds(ot, mode, w, min_support=ms, max_clip_dist=int(0.5 * med))
        b, u, _ = ctx.cluster(t, mode, w, min_support=ms, max_clip_dist=int(0.5 * med))
        assert len(b) == len(eb), ("bounds count", tag, mode)
        for f in ("tid", "left", "right", "left_most", "right_most", "center_mass", "n_left", "n_right", "n_total", "repeat"):
            assert np.array_equal(b[f], eb[f]), ("bounds", f, tag, mode)
        n_bounds += len(b)
    n_cases += 1
    if n_cases % 100 == 0:
        print(f"  {n_cases} cases, {time.time() - t0:.0f} s, last: {tag}", flush=True)
    n_reads += rec.n
    n_treads += len(exp)
print(f"fuzz ok (master seed {master}): {n_asserts} reference-assert cases (count >= 256), {n_cases} cases ({n_long_cases} drawn with long records), {n_reads} reads, {n_treads} treads, {n_bounds} bounds in {time.time() - t0:.0f} s")
