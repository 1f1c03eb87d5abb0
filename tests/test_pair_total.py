"""The device pair logic is total: hash runs longer than the in-block replay takes (one qname on 40 primary records, three
qname groups whose mixed hashes share their low 32 bits), secondary / supplementary records that share a hot qname, and --
through the device front end, where the names are on the device -- the qname check behind the 64-bit hash."""
import numpy as np
import pytest

from strling_amd import api, bamio, synth
from strling_amd.records import RecordBatch

pytestmark = pytest.mark.gpu
FIELDS = ("tid", "position", "repeat", "flag", "split", "mapping_quality", "repeat_count", "align_length", "qname_id")
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _mixed_hash(names):
    """fmix64(strl_qname_hash(name)) of equal-length names, vectorised (host_logic.cpp hash_bytes, common.h fmix64)"""
    a = np.frombuffer(b"".join(names), np.uint8).reshape(len(names), -1)
    with np.errstate(over="ignore"):
        h = np.full(len(names), 0xcbf29ce484222325, np.uint64)
        for j in range(a.shape[1]):
            h = (h ^ a[:, j].astype(np.uint64)) * np.uint64(0x100000001b3)
        h = h ^ (h >> np.uint64(29))
        h ^= h >> np.uint64(33); h *= np.uint64(0xff51afd7ed558ccd)
        h ^= h >> np.uint64(33); h *= np.uint64(0xc4ceb9fe1a85ec53)
        h ^= h >> np.uint64(33)
    return h


def colliding_names(k=3, n=9_000_000):
    """k names whose mixed hashes share their low 32 bits"""
    names = [b"c%08d" % i for i in range(n)]
    lo = (_mixed_hash(names) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    order = np.argsort(lo, kind="stable")
    s = lo[order]
    run = np.flatnonzero(s[k - 1:] == s[:len(s) - k + 1])
    assert run.size, "no %d-fold collision among %d names" % (k, n)
    return [names[int(order[run[0] + j])] for j in range(k)]


def take(rec, idx, flags=None, names=None):
    """records rec[idx] (repeats allowed) with optional per-output flag / qname overrides -> RecordBatch"""
    cig, cig_off, seq_parts, seq_off, qn, qoff = [], [0], [], [], bytearray(), [0]
    so = 0
    for o, i in enumerate(idx):
        i = int(i)
        c0, c1 = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
        cig.extend(rec.cigar[c0:c1].tolist())
        cig_off.append(len(cig))
        nb = (int(rec.l_seq[i]) + 1) // 2
        s0 = int(rec.seq_off[i])
        pad = (nb + 15) // 16 * 16
        buf = np.zeros(pad, np.uint8)
        buf[:nb] = rec.seq4[s0:s0 + nb]
        seq_parts.append(buf)
        seq_off.append(so)
        so += pad
        qn += names[o] if names is not None and names[o] is not None else rec.qname(i)
        qoff.append(len(qn))
    idx = np.asarray(idx)
    fl = rec.flag[idx].copy() if flags is None else np.asarray(flags, np.uint16)
    seq4 = np.concatenate(seq_parts + [np.zeros(32, np.uint8)])
    return RecordBatch(rec.tid[idx], rec.pos[idx], rec.mtid[idx], rec.mpos[idx], fl, rec.mapq[idx], np.asarray(cig_off, np.uint32),
                       np.asarray(cig, np.uint32), np.asarray(seq_off, np.uint64), rec.l_seq[idx], seq4, np.asarray(qoff, np.uint64), bytes(qn),
                       None if rec.isize is None else rec.isize[idx], rec.targets)


def _build(oracle):
    rec, g = synth.synth_wgs(6000, seed=31, contig_len=900_000, soft_frac=0.2)
    med = oracle.median(synth.frag_hist(rec))
    opts = oracle.make_opts(med, 0.8, 40)
    base = oracle.extract(rec, g, opts)
    # qname groups (pairs) that emit treads, both mates placed: candidates to be multiplied / renamed
    by_name = {}
    for i in range(rec.n):
        by_name.setdefault(rec.qname(i), []).append(i)
    hot = []
    for i in np.unique(base["qname_id"]):
        q = rec.qname(int(i))
        m = by_name[q]
        if len(m) == 2 and all(rec.tid[j] >= 0 for j in m) and q not in [h[0] for h in hot]:
            hot.append((q, m))
    assert len(hot) >= 8
    names3 = colliding_names(3)
    idx, flags, names = [], [], []
    dup = {hot[0][1][0]: 20, hot[0][1][1]: 20}              # 40 primary records under one qname
    sec = {hot[1][1][0]: 0x100, hot[2][1][1]: 0x800, hot[0][1][0]: 0x100}   # secondary / supplementary copies beside hot records
    rename = {}
    for k, (q, m) in enumerate([hot[0], hot[3], hot[4]]):   # three groups (one of them the 40-record one) under one 32-bit prefix
        for j in m:
            rename[j] = names3[k]
    for i in range(rec.n):
        for r in range(dup.get(i, 1)):
            idx.append(i); flags.append(int(rec.flag[i])); names.append(rename.get(i))
        if i in sec:
            idx.append(i); flags.append(int(rec.flag[i]) | sec[i]); names.append(rename.get(i))
    rec2 = take(rec, idx, flags, names)
    return rec2, g, med, opts


@pytest.fixture(scope="module")
def case():
    from oracle import oracle as O
    rec2, g, med, opts = _build(O)
    exp = O.extract(rec2, g, opts)
    return rec2, g, med, exp


def test_long_runs_and_prefix_collisions_on_the_device(ctx, case):
    """strl_extract_begin/_add/_finish: the DEVICE join itself (no host fallback in this entry point) replays a 44-item run"""
    rec2, g, med, exp = case
    assert len(exp) > 100
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    soa = api.Soa(rec2)
    rows, qh = soa.pair_rows()
    n_tail = 0
    while n_tail < rec2.n and rec2.tid[rec2.n - 1 - n_tail] < 0:
        n_tail += 1
    ctx.extract_chunks([(soa.c_struct(), api.CPairSoa(rows.ctypes.data, qh.ctypes.data))], n_tail)
    got, st = ctx.treads_fetch()
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), f
    # the three renamed groups really share the low 32 bits, and one of them has 40 + 1 records
    names = {rec2.qname(i) for i in range(rec2.n) if rec2.qname(i).startswith(b"c")}
    assert len(names) == 3
    lo = _mixed_hash(sorted(names)) & np.uint64(0xFFFFFFFF)
    assert len(set(lo.tolist())) == 1
    assert max(sum(1 for i in range(rec2.n) if rec2.qname(i) == q) for q in names) >= 41


def test_the_same_file_through_the_device_front_end(ctx, case, tmp_path):
    """... and with the BAM front end on the device, where every hash group's records are checked to carry ONE qname"""
    rec2, g, med, exp = case
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    path = str(tmp_path / "t.bam")
    bamio.write_bam(path, rec2, level=1, index=False)
    got = ctx.extract_bam_device(path, chunk_blocks=4)
    for f in FIELDS:
        assert np.array_equal(got["treads"][f], exp[f]), f
    assert got["qnames"] == [rec2.qname(int(i)) for i in exp["qname_id"]]
