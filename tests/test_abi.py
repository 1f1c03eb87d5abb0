"""C-ABI library: loads, exports every symbol include/strling_amd.h declares, fails loudly without
a device, and its host-only entry points (SoA derivation, pair logic, .bin, bounds rows, fragment
statistics) agree with the oracle.  CPU only -- no compute kernels are launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from strling_amd import api, synth
from strling_amd.records import RecordBatch
from helpers import oracle_words, soft_items_expected, treads_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    L = api.load()
    hdr = open(os.path.join(ROOT, "include", "strling_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(strl_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/strling_amd.h but not exported"
    assert sorted(api.EXPORTS) == declared


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = api.load()
    h = C.c_void_p()
    rc = L.strl_ctx_create(0, C.byref(h))
    assert rc == -1 and not h.value
    assert b"no CPU fallback" in L.strl_last_error()
    with pytest.raises(api.StrlingError):
        api.Context(0)


@pytest.fixture(scope="module")
def batch():
    rec, g = synth.synth_wgs(6000, seed=99, contig_len=1_000_000)
    return rec, g


def test_soa_from_records(batch, oracle):
    rec, g = batch
    soa = api.Soa(rec)
    assert soa.max_l_seq == 150
    for i in range(0, rec.n, 37):
        a, b = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
        ops = [(int(c) & 15, int(c) >> 4) for c in rec.cigar[a:b]]
        ref_len = sum(l for o, l in ops if o in (0, 2, 3, 7, 8)) if not (rec.flag[i] & 4) else 0
        assert soa.end[i] == rec.pos[i] + (ref_len or 1)
        single_m = len(ops) == 1 and ops[0][0] == 0      # clip_l carries the M length of a single-M cigar (extract.nim:33)
        assert soa.clip_l[i] == (ops[0][1] if ops and (ops[0][0] == 4 or single_m) else 0)
        assert soa.clip_r[i] == (ops[-1][1] if ops and ops[-1][0] == 4 else 0)
        assert bool(soa.cig[i] & 1) == (len(ops) == 1 and ops[0][0] == 0)
        assert bool(soa.cig[i] & 16) == (len(ops) == 0)
        assert soa.seq_off[i] * 16 == rec.seq_off[i]


@pytest.mark.parametrize("p,q", [(0.8, 40), (0.6, 20)])
def test_pair_logic_matches_oracle(batch, oracle, p, q):
    """strl_pair_reads (host state machine of extract.nim:192-248) fed with scorer words == oracle extract."""
    rec, g = batch
    med = oracle.median(synth.frag_hist(rec))
    opts = oracle.make_opts(med, p, q)
    whole, softd = oracle_words(oracle, rec, g, opts)
    items = soft_items_expected(rec, whole, q)
    soft = np.zeros(len(items), api.SOFT_DTYPE)
    for j, (i, side) in enumerate(items):
        soft[j] = ((i << 1) | side, softd[(i, side)][0], softd[(i, side)][1], 0)
    exp = oracle.extract(rec, g, opts)
    L = api.load()
    rv = api._RecView(rec)
    o = api.Opts(med, p, q)
    out = np.zeros(len(exp) + 16, api.TREAD_DTYPE)
    no = C.c_uint64(0)
    rc = L.strl_pair_reads(C.byref(rv.c), C.byref(o), whole.ctypes.data, soft.ctypes.data, soft.size, -1, out.ctypes.data, out.size,
                           C.byref(no))
    assert rc == 0, L.strl_last_error()
    assert len(exp) > 100
    ok, why = treads_equal(out[:no.value], exp)
    assert ok, why


def test_unmapped_tail_is_replayed(oracle):
    """extract.nim:308 reads to EOF (tail included) and :326 query("*") visits the tail again: a both-STR
    unmapped pair is emitted twice."""
    seq = "CAG" * 50
    rec = RecordBatch.from_fields(tid=[-1, -1], pos=[-1, -1], mtid=[-1, -1], mpos=[-1, -1], flag=[77, 141], mapq=[0, 0],
                                  cigars=["*", "*"], seqs=[seq, seq], qnames=["x", "x"])
    opts = oracle.make_opts(350, 0.8, 40)
    t = oracle.extract(rec, None, opts)
    assert len(t) == 4 and set(t["tid"]) == {-1}
    assert set(t["repeat"]) == {oracle.canonical_repeat(oracle.get_repeat(seq, 0.8)[0]).encode()}


def test_frag_median(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        f = rng.integers(0, 50, 4096).astype(np.uint32)
        f[rng.integers(0, 4096, 300)] += rng.integers(0, 10000, 300).astype(np.uint32)
        for pct in (0.5, 0.1, 0.9, 0.98, 0.99):
            assert api.frag_median(f, pct) == oracle.median(f, pct)


def test_bin_roundtrip_and_bytes(tmp_path, batch, oracle):
    import msgpack
    rec, g = batch
    opts = oracle.make_opts(350, 0.8, 40)
    exp = oracle.extract(rec, g, opts)
    frag = synth.frag_hist(rec)
    hdr = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in rec.targets)
    t = np.zeros(len(exp), api.TREAD_DTYPE)
    for f in t.dtype.names:
        t[f] = exp[f]
    path = str(tmp_path / "s.bin")
    api.bin_write(path, 0.8, 40, frag, hdr, t, rec.qname_off, rec.qnames)
    raw = open(path, "rb").read()
    assert raw == oracle.bin_write(0.8, 40, frag, hdr, exp, rec.qname_off, rec.qnames)      # byte-identical writers
    # header layout of extract.nim:336-344
    assert raw[:3] == b"STR" and raw[3:5] == b"\0\0" and raw[5:14] == b"0.6.0\0\0\0\0"
    assert np.frombuffer(raw[14:18], "<f4")[0] == np.float32(0.8) and raw[18] == 40
    o = 19 + 4096 * 4
    hl = int(np.frombuffer(raw[o:o + 4], "<i4")[0])
    assert raw[o + 4:o + 4 + hl].decode() == hdr
    n = int(np.frombuffer(raw[o + 4 + hl:o + 8 + hl], "<i4")[0])
    assert n == len(exp)
    # the records are plain msgpack values in pack_type order (cluster.nim:38-50)
    up = msgpack.Unpacker(raw=True)
    up.feed(raw[o + 8 + hl:])
    vals = list(up)
    assert len(vals) == 10 * n
    for i in (0, n // 2, n - 1):
        v = vals[10 * i:10 * i + 10]
        assert v[0] == exp["tid"][i] and v[1] == exp["position"][i] and bytes(v[2]).rstrip(b"\0") == exp["repeat"][i]
        assert v[3] == exp["flag"][i] and v[4] == exp["split"][i] and v[8] == len(v[9]) and v[9] == rec.qname(int(exp["qname_id"][i]))
    back = api.bin_read(path)
    assert back["header"] == hdr and back["min_mapq"] == 40 and np.array_equal(back["frag"], frag)
    ok, why = treads_equal(back["treads"], t, fields=("tid", "position", "repeat", "flag", "split", "mapping_quality", "repeat_count",
                                                      "align_length"))
    assert ok, why
    qn = [back["qnames"][int(back["qname_off"][i]):int(back["qname_off"][i + 1])] for i in range(n)]
    assert qn == [rec.qname(int(i)) for i in exp["qname_id"]]


def test_bounds_row(oracle):
    b = np.zeros(1, api.BOUNDS_DTYPE)
    b["tid"], b["left"], b["right"], b["left_most"], b["right_most"], b["center_mass"] = 0, 990, 1010, 500, 1500, 1000
    b["n_left"], b["n_right"], b["n_total"], b["repeat"] = 3, 1, 50, b"CAG"
    row = api.bounds_row(b[0], "chr1")
    assert row == "chr1\t990\t1010\tCAG\t\t500\t1500\t1000\t3\t1\t50"      # tests/test_cluster.nim:171 line layout
    ob = np.zeros(1, oracle.BOUNDS_DTYPE)
    for f in b.dtype.names:
        ob[f] = b[f]
    assert oracle.bounds_row(ob[0], "chr1") == row
    assert len(row.split("\t")) == 11                                    # parse_boundsline requires 11 fields (cluster.nim:146)


def test_bin_writer_threads_bytes(tmp_path, batch, oracle):
    """past 2^18 treads strl_bin_write packs and writes in parts on several threads: the bytes are the sequential writer's"""
    rec, g = batch
    exp = oracle.extract(rec, g, oracle.make_opts(350, 0.8, 40))
    assert len(exp) > 20
    reps = (300_000 + len(exp) - 1) // len(exp)
    big = np.concatenate([exp] * reps)
    frag = synth.frag_hist(rec)
    hdr = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in rec.targets)
    t = np.zeros(len(big), api.TREAD_DTYPE)
    for f in t.dtype.names:
        t[f] = big[f]
    path = str(tmp_path / "big.bin")
    api.bin_write(path, 0.8, 40, frag, hdr, t, rec.qname_off, rec.qnames)
    assert open(path, "rb").read() == oracle.bin_write(0.8, 40, frag, hdr, big, rec.qname_off, rec.qnames)
    back = api.bin_read(path)
    assert len(back["treads"]) == len(big) and np.array_equal(back["treads"]["position"], big["position"])


def test_bin_reader_parts_equal_sequential(tmp_path, batch, oracle):
    """a .bin past 32 MB is parsed in parts by several threads (record starts guessed, then verified by linking the parts):
    the same arrays as the sequential walk, for ragged names too; a damaged file gets the sequential walk's verdict"""
    import subprocess
    import sys
    rec, g = batch
    exp = oracle.extract(rec, g, oracle.make_opts(350, 0.8, 40))
    reps = (1_800_000 + len(exp) - 1) // len(exp)
    big = np.concatenate([exp] * reps)
    t = np.zeros(len(big), api.TREAD_DTYPE)
    for f in t.dtype.names:
        t[f] = big[f]
    rng = np.random.default_rng(5)
    t["position"] = rng.integers(0, 1 << 32, len(t), dtype=np.uint64).astype(np.uint32)      # every width of msgpack integer
    t["tid"] = rng.integers(-1, 70000, len(t)).astype(np.int32)
    frag = synth.frag_hist(rec)
    path = str(tmp_path / "big.bin")
    api.bin_write(path, 0.8, 40, frag, "@HD\tVN:1.6\n", t, rec.qname_off, rec.qnames)
    assert os.path.getsize(path) > (36 << 20)
    a = api.bin_read(path)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from strling_amd import api; b = api.bin_read(%r); "
            "np.save(%r, b['treads']); np.save(%r, b['qname_off']); open(%r, 'wb').write(b['qnames'])")
    outs = [str(tmp_path / x) for x in ("t.npy", "q.npy", "n.bin")]
    r = subprocess.run([sys.executable, "-c", code % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, *outs)],
                       env=dict(os.environ, STRL_BIN_READ="seq"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(a["treads"], np.load(outs[0])) and np.array_equal(a["qname_off"], np.load(outs[1])) and a["qnames"] == open(outs[2], "rb").read()
    assert np.array_equal(a["treads"]["position"], t["position"]) and np.array_equal(a["treads"]["tid"], t["tid"])
    assert np.array_equal(a["treads"]["qname_id"], np.arange(len(t)))
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x55                                       # somewhere inside a record
    open(path, "wb").write(raw[: len(raw) - 7])                      # ... and a truncated tail
    with pytest.raises(api.StrlingError):
        api.bin_read(path)


def test_bin_reader_parts_survive_a_made_up_first_record(tmp_path):
    """a share that starts on the last two bytes of a record's 4-byte position parses them as tid + position, finds the array
    marker and is in step from there: eight strict records, the first made up.  The parts still link (round 6: this guess sent
    whole-genome files down the sequential walk); the file is built so that share 1 starts exactly there"""
    import subprocess
    import sys
    hw = len(os.sched_getaffinity(0))
    if hw < 2:
        pytest.skip("one CPU: the reader does not split the file")
    header = "@HD\tVN:1.6\n"
    body0 = 3 + 2 + 9 + 4 + 1 + 4096 * 4 + 4 + len(header) + 4
    S = 42                                                           # every record: 1 + 5 + 1 + 6 + 3 + 4 + 1 + 1 + 20 bytes
    K = min(hw, 16, (S * 900_000) >> 22)                             # the reader's number of parts (host_logic.cpp)
    n = next(m for m in range(900_000, 905_000) if any(1 <= ((S * m) // K * k) % S <= 4 for k in range(1, K)))   # a share starts on 0xce .. b1
    rng = np.random.default_rng(9)
    t = np.zeros(n, api.TREAD_DTYPE)
    t["tid"] = rng.integers(0, 100, n)
    t["position"] = (rng.integers(1, 65536, n) << 16) | (rng.integers(0, 128, n) << 8) | rng.integers(0, 128, n)   # 0xce + 4 bytes, the last two < 0x80
    t["repeat"] = np.array([b"AC", b"AGC", b"AAAAG", b"AACCCT"], "S6")[rng.integers(0, 4, n)]
    t["flag"] = rng.integers(256, 4096, n)
    t["split"] = rng.integers(0, 3, n); t["mapping_quality"] = rng.integers(0, 61, n)
    t["repeat_count"] = rng.integers(0, 100, n); t["align_length"] = rng.integers(0, 128, n)
    t["qname_id"] = np.arange(n)
    qo = np.arange(n + 1, dtype=np.uint64) * 20
    qn = rng.integers(97, 123, 20 * n, dtype=np.uint8).tobytes()
    path = str(tmp_path / "made_up.bin")
    api.bin_write(path, 0.8, 40, np.zeros(4096, np.uint32), header, t, qo, qn)
    assert os.path.getsize(path) == body0 + S * n
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from strling_amd import api; b = api.bin_read(%r); "
            "np.save(%r, b['treads']); np.save(%r, b['qname_off']); open(%r, 'wb').write(b['qnames'])")
    outs = [str(tmp_path / x) for x in ("t.npy", "q.npy", "n.bin")]
    r = subprocess.run([sys.executable, "-c", code % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, *outs)],
                       env=dict(os.environ, STRL_BIN_TIMING="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "records parsed into place" in r.stderr, r.stderr          # the parts linked: no sequential walk
    assert np.array_equal(np.load(outs[0]), t) and np.array_equal(np.load(outs[1]), qo) and open(outs[2], "rb").read() == qn


def test_bin_reader_parts_on_small_files_equal_sequential(tmp_path, capfd, monkeypatch):
    """the guess-and-link reader forced onto small files (STRL_BIN_READ=parts=K): every width of integer, names of 1..40 bytes,
    shares of a few records -- always the sequential walk's arrays, and the parts link in most files (the rest fall back)"""
    rng = np.random.default_rng(77)
    linked = cases = 0
    for seed in range(60):
        n = int(rng.integers(50, 3000))
        t = np.zeros(n, api.TREAD_DTYPE)
        t["tid"] = rng.integers(-1, [3, 200, 70000][seed % 3], n)
        hi = [1 << 7, 1 << 16, 1 << 32][seed % 3]
        t["position"] = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
        t["repeat"] = np.array([b"A", b"AC", b"AGC", b"AAAG", b"AAAAG", b"AACCCT"], "S6")[rng.integers(0, 6, n)]
        t["flag"] = rng.integers(0, 4096, n)
        t["split"] = rng.integers(0, 3, n); t["mapping_quality"] = rng.integers(0, 255, n)
        t["repeat_count"] = rng.integers(0, 256, n); t["align_length"] = rng.integers(0, 256, n)
        t["qname_id"] = np.arange(n)
        lens = rng.integers(1, 41, n).astype(np.uint64)   # (an empty name: written as the reference writes it, refused as it refuses it)
        qo = np.zeros(n + 1, np.uint64); qo[1:] = np.cumsum(lens)
        qn = rng.integers(33, 127, int(qo[-1]), dtype=np.uint8).tobytes()
        path = str(tmp_path / f"s{seed}.bin")
        api.bin_write(path, 0.8, 40, np.zeros(4096, np.uint32), "@HD\tVN:1.6\n", t, qo, qn)
        monkeypatch.setenv("STRL_BIN_READ", "seq")
        a = api.bin_read(path)
        assert np.array_equal(a["treads"], t) and np.array_equal(a["qname_off"], qo) and a["qnames"] == qn
        for K in (2, int(rng.integers(3, 17))):
            monkeypatch.setenv("STRL_BIN_READ", f"parts={K}")
            monkeypatch.setenv("STRL_BIN_TIMING", "1")
            capfd.readouterr()
            b = api.bin_read(path)
            err = capfd.readouterr().err
            monkeypatch.delenv("STRL_BIN_TIMING")
            cases += 1
            linked += "records parsed into place" in err
            assert np.array_equal(b["treads"], t) and np.array_equal(b["qname_off"], qo) and b["qnames"] == qn, (seed, K)
    assert linked >= cases * 0.8, (linked, cases)


def test_bin_header_count_the_file_cannot_hold_is_refused_by_the_peek(tmp_path):
    """callers size their arrays from the header's read count (strl_bin_peek): 2^31 - 1 reads in a 17 KB file is a format error
    there, not a 64 GB allocation followed by the reader's `expected N got M`"""
    import ctypes as C
    t = np.zeros(3, api.TREAD_DTYPE)
    t["repeat"] = b"AC"; t["qname_id"] = np.arange(3)
    path = str(tmp_path / "few.bin")
    hdr = "@HD\tVN:1.6\n"
    api.bin_write(path, 0.8, 40, np.zeros(4096, np.uint32), hdr, t, np.array([0, 2, 4, 6], np.uint64), b"aabbcc")
    info = api.BinInfo()
    assert api.load().strl_bin_peek(path.encode(), C.byref(info)) == 0 and info.n_reads == 3
    raw = bytearray(open(path, "rb").read())
    at = 3 + 2 + 9 + 4 + 1 + 4096 * 4 + 4 + len(hdr)
    assert int.from_bytes(raw[at:at + 4], "little") == 3
    raw[at:at + 4] = (0x7fffffff).to_bytes(4, "little")
    open(path, "wb").write(raw)
    assert api.load().strl_bin_peek(path.encode(), C.byref(info)) != 0 and b"expected 2147483647" in api.load().strl_last_error()
    with pytest.raises(api.StrlingError):
        api.bin_read(path)


def test_bin_reader_parts_and_sequential_agree_on_damaged_files(tmp_path, monkeypatch):
    """a flipped byte, a dropped byte or a cut tail anywhere in a .bin: the guess-and-link parse (forced, 2..12 parts) gives the
    sequential walk's verdict -- the same error, or the same arrays where the damage happens to parse"""
    rng = np.random.default_rng(123)
    n = 1200
    t = np.zeros(n, api.TREAD_DTYPE)
    t["tid"] = rng.integers(-1, 300, n)
    t["position"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    t["repeat"] = np.array([b"A", b"AC", b"AGC", b"AAAG", b"AAAAG", b"AACCCT"], "S6")[rng.integers(0, 6, n)]
    t["flag"] = rng.integers(0, 4096, n); t["split"] = rng.integers(0, 3, n); t["mapping_quality"] = rng.integers(0, 255, n)
    t["repeat_count"] = rng.integers(0, 256, n); t["align_length"] = rng.integers(0, 256, n); t["qname_id"] = np.arange(n)
    lens = rng.integers(1, 41, n).astype(np.uint64)
    qo = np.zeros(n + 1, np.uint64); qo[1:] = np.cumsum(lens)
    qn = rng.integers(33, 127, int(qo[-1]), dtype=np.uint8).tobytes()
    hdr = "@HD\tVN:1.6\n"
    good = str(tmp_path / "good.bin")
    api.bin_write(good, 0.8, 40, np.zeros(4096, np.uint32), hdr, t, qo, qn)
    raw = open(good, "rb").read()
    body0 = 3 + 2 + 9 + 4 + 1 + 4096 * 4 + 4 + len(hdr) + 4
    errors = same = 0
    for case in range(120):
        b = bytearray(raw)
        at = int(rng.integers(body0, len(b)))
        kind = case % 3
        if kind == 0:
            b[at] ^= int(rng.integers(1, 256))
        elif kind == 1:
            del b[at]
        else:
            b = b[:at]
        path = str(tmp_path / "bad.bin")
        open(path, "wb").write(b)
        out = []
        for mode in ("seq", f"parts={int(rng.integers(2, 13))}"):
            monkeypatch.setenv("STRL_BIN_READ", mode)
            try:
                r = api.bin_read(path)
                out.append((r["treads"], r["qname_off"], r["qnames"]))
            except api.StrlingError as e:
                out.append("error")
        assert isinstance(out[0], str) == isinstance(out[1], str), (case, kind, at)
        if isinstance(out[0], str):
            errors += 1
        else:     # (field by field: numpy's copy of a padded record type leaves the padding to chance)
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2], (case, kind, at)
            same += 1
    assert errors >= 60 and same >= 5, (errors, same)        # (a flipped name byte or count still parses: those must agree too)
