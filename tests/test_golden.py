"""Committed golden vectors (tests/golden/s1_small.npz, written by tests/golden/make_golden.py with the oracle):
the oracle must still reproduce them (CPU), and the HIP path must match them through the C ABI (GPU)."""
import os
import sys

import numpy as np
import pytest

from strling_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import PARAMS  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "s1_small.npz"))
FIELDS = [("tid", "tread_tid"), ("position", "tread_pos"), ("repeat", "tread_repeat"), ("flag", "tread_flag"), ("split", "tread_split"),
          ("mapping_quality", "tread_mapq"), ("repeat_count", "tread_count"), ("align_length", "tread_alen"), ("qname_id", "tread_qid")]


def _batch():
    return synth.synth_wgs(PARAMS["n_pairs"], seed=PARAMS["seed"], contig_len=PARAMS["contig_len"])


def test_oracle_reproduces_the_fixture(oracle):
    rec, g = _batch()
    med = oracle.median(synth.frag_hist(rec))
    assert med == int(G["frag_median"])
    t = oracle.extract(rec, g, oracle.make_opts(med, PARAMS["p"], PARAMS["min_mapq"]))
    for f, k in FIELDS:
        assert np.array_equal(t[f], G[k]), f
    b, u = oracle.call_bounds(t, 1, int(G["window"]), min_support=PARAMS["min_support"], max_clip_dist=int(G["max_clip_dist"]))
    assert [oracle.bounds_row(x, rec.targets[int(x["tid"])][0]) for x in b] == G["bounds_rows"].tolist()
    assert [f"{a}\t{c}" for a, c in u] == G["unplaced"].tolist()


@pytest.mark.gpu
def test_hip_path_matches_the_fixture(ctx):
    rec, g = _batch()
    ctx.set_opts(PARAMS["p"], PARAMS["min_mapq"], int(G["frag_median"]))
    ctx.set_genome(g)
    whole, soft, st = ctx.score_reads(rec)
    assert np.array_equal(whole, G["whole"])
    t, _ = ctx.extract(rec)
    for f, k in FIELDS:
        assert np.array_equal(t[f], G[k]), f
    b, u, _ = ctx.cluster(t, api.MODE_CALL, int(G["window"]), min_support=PARAMS["min_support"], max_clip_dist=int(G["max_clip_dist"]))
    assert [api.bounds_row(x, rec.targets[int(x["tid"])][0]) for x in b] == G["bounds_rows"].tolist()
    assert [f"{x['repeat'].decode()}\t{int(x['count'])}" for x in u] == G["unplaced"].tolist()
