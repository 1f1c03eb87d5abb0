"""The reference's own known-answer vectors (tests/golden/reference_kats.json, transcribed from the Nim test suite) run
through the PRODUCT's code, not the oracle: the host functions of libstrling_amd.so here (no GPU needed), and the device
functions the kernels call in the -m gpu half.  The oracle is a twin by the same author; agreement between the two says
little about the real Nim, these vectors do."""
import json
import os

import numpy as np
import pytest

from strling_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _tread(tid=0, position=0, repeat="", mapping_quality=0, repeat_count=0, align_length=0, split=3, flag=0):
    t = np.zeros(1, api.TREAD_DTYPE)[0]
    t["tid"], t["position"], t["repeat"], t["mapping_quality"] = tid, position, repeat.encode(), mapping_quality
    t["repeat_count"], t["align_length"], t["split"], t["flag"] = repeat_count % 256, align_length, split, flag
    return t


def _check_pair_rules(ctx):
    for k in KATS["canonical_repeat"]:
        _, a = api.pair_rule(ctx, 2, _tread(repeat=k["in"]), _tread(), (0.8, 40, 0))
        assert a["repeat"].decode() == k["out"], k
    for k in KATS["unplaced_pair"]:
        A = _tread(tid=0, position=222, **k["A"])
        B = _tread(tid=0, position=222, **k["B"])
        res, _ = api.pair_rule(ctx, 1, A, B, (k["p"], k["min_mapq"], 500))
        assert bool(res) == k["expect"], k
    for k in KATS["adjust_by"]:
        A, B = _tread(**k["A"]), _tread(**k["B"])
        res, a = api.pair_rule(ctx, 0, A, B, (k["p"], k["min_mapq"], 0), B_position=int(B["position"]))
        assert bool(res) == k["expect_return"] and int(a["position"]) == k["expect_position"] == int(B["position"]) + int(B["align_length"])
        assert int(a["tid"]) == int(B["tid"]) and int(a["split"]) == 3 and int(a["mapping_quality"]) == 60   # extract.nim:168-171


def test_pair_rules_host_twins():
    """adjust_by / unplaced_pair / canonical_repeat of host_logic.cpp (what the streaming pairer runs)"""
    _check_pair_rules(None)


def test_canonical_repeat_host():
    L = api.load()
    for k in KATS["canonical_repeat"]:
        out = (bytes(6) + b"\0")
        import ctypes as C
        buf = C.create_string_buffer(7)
        L.strl_canonical_repeat(k["in"].encode().ljust(6, b"\0"), buf)
        assert buf.raw[:6].rstrip(b"\0").decode() == k["out"]


def test_frag_median_host():
    frag = np.ones(4096, np.uint32)
    assert api.frag_median(frag, 0.5) == 2047


@pytest.mark.gpu
def test_pair_rules_device(ctx):
    """the same vectors through the device functions pair_groups_kernel calls (pair.hip)"""
    _check_pair_rules(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("k", KATS["bounds"], ids=lambda k: k["source"])
def test_bounds_kats_on_a_bare_cluster_including_left_most_right_most(ctx, k):
    """bounds() vectors of tests/test_cluster.nim through the device function on exactly the KAT's bare Cluster (left_most =
    right_most = 0), every stated column compared -- left_most / right_most included"""
    order = np.argsort(np.asarray(k["positions"]), kind="stable")
    pos = np.asarray(k["positions"], np.uint32)[order]
    spl = np.asarray(k["splits"], np.uint8)[order]
    b, good = ctx.bounds_bare(pos, spl, k["max_clip_dist"])
    for f, v in k.get("expect", {}).items():
        assert int(b[f]) == v, (f, int(b[f]), v)
    assert int(b["left"]) < int(b["right"])


def test_bin_writer_packs_large_tread_sets_on_several_threads(tmp_path):
    """more than 2^18 treads: strl_bin_write packs ranges of them on a few threads -- the bytes stay the sequential writer's (the oracle's)"""
    import numpy as np
    from oracle import oracle as O
    from strling_amd import api
    rng = np.random.default_rng(11)
    n = 300_000
    t = np.zeros(n, api.TREAD_DTYPE)
    t["tid"] = rng.integers(-1, 3000, n)
    t["position"] = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    units = np.array([b"A", b"AC", b"CAG", b"AAAG", b"AACCT", b"ACGTCA"])
    t["repeat"] = units[rng.integers(0, 6, n)]
    t["flag"] = rng.integers(0, 4096, n)
    t["split"] = rng.integers(0, 6, n)
    t["mapping_quality"] = rng.integers(0, 61, n)
    t["repeat_count"] = rng.integers(0, 256, n)
    t["align_length"] = rng.integers(0, 256, n)
    t["qname_id"] = np.arange(n)
    names = [b"q%d" % int(x) for x in rng.integers(0, 10 ** 9, n)]
    qoff = np.zeros(n + 1, np.uint64)
    qoff[1:] = np.cumsum([len(x) for x in names])
    qn = b"".join(names)
    frag = rng.integers(0, 1000, 4096).astype(np.uint32)
    p = str(tmp_path / "big.bin")
    api.bin_write(p, 0.8, 40, frag, "@HD\tVN:1.6\n", t, qoff, qn)
    ot = np.zeros(n, O.TREAD_DTYPE)
    for f in ot.dtype.names:
        if f in t.dtype.names:
            ot[f] = t[f]
    assert open(p, "rb").read() == O.bin_write(0.8, 40, frag, "@HD\tVN:1.6\n", ot, qoff, qn)
    back = api.bin_read(p)
    assert np.array_equal(back["treads"]["position"], t["position"]) and back["qnames"] == qn
