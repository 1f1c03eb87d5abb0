"""ctypes binding of the parity oracle (oracle/strling_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py.  Nothing under strling_amd/ may import it.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "strling_oracle.c")
    hdr = os.path.join(_HERE, "strling_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class Tread(C.Structure):
    _fields_ = [("tid", C.c_int32), ("position", C.c_uint32), ("repeat", C.c_char * 6), ("flag", C.c_uint16),
                ("split", C.c_uint8), ("mapping_quality", C.c_uint8), ("repeat_count", C.c_uint8),
                ("align_length", C.c_uint8), ("qname_id", C.c_int64), ("src", C.c_int64)]


TREAD_DTYPE = np.dtype([("tid", "<i4"), ("position", "<u4"), ("repeat", "S6"), ("flag", "<u2"), ("split", "u1"),
                        ("mapping_quality", "u1"), ("repeat_count", "u1"), ("align_length", "u1"),
                        ("qname_id", "<i8"), ("src", "<i8")], align=True)
assert TREAD_DTYPE.itemsize == C.sizeof(Tread), (TREAD_DTYPE.itemsize, C.sizeof(Tread))


class Opts(C.Structure):
    _fields_ = [("median_fragment_length", C.c_int), ("proportion_repeat", C.c_double), ("min_mapq", C.c_uint8)]


class Records(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("mtid", C.c_void_p), ("mpos", C.c_void_p),
                ("flag", C.c_void_p), ("mapq", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("l_seq", C.c_void_p), ("seq4", C.c_void_p), ("qname_off", C.c_void_p),
                ("qnames", C.c_void_p)]


class GenomeStr(C.Structure):
    _fields_ = [("n_tid", C.c_int32), ("has_chrom", C.c_void_p), ("iv_off", C.c_void_p), ("iv_start", C.c_void_p),
                ("iv_stop", C.c_void_p), ("max_len", C.c_void_p)]


class SegResult(C.Structure):
    _fields_ = [("rep", C.c_char * 6), ("count", C.c_int32), ("align_length", C.c_int32)]


class Bounds(C.Structure):
    _fields_ = [("tid", C.c_int32), ("left", C.c_uint32), ("left_most", C.c_uint32), ("right", C.c_uint32),
                ("right_most", C.c_uint32), ("center_mass", C.c_uint32), ("n_left", C.c_uint16),
                ("n_right", C.c_uint16), ("n_total", C.c_uint16), ("repeat", C.c_char * 7)]


BOUNDS_DTYPE = np.dtype([("tid", "<i4"), ("left", "<u4"), ("left_most", "<u4"), ("right", "<u4"), ("right_most", "<u4"),
                         ("center_mass", "<u4"), ("n_left", "<u2"), ("n_right", "<u2"), ("n_total", "<u2"),
                         ("repeat", "S7")], align=True)
assert BOUNDS_DTYPE.itemsize == C.sizeof(Bounds), (BOUNDS_DTYPE.itemsize, C.sizeof(Bounds))


class Support(C.Structure):
    _fields_ = [("type", C.c_uint8), ("repeat_count", C.c_uint8), ("cigar_ins", C.c_uint8), ("cigar_del", C.c_uint8),
                ("frag_len", C.c_uint32), ("frag_pct", C.c_double), ("rec", C.c_int64)]


SUPPORT_DTYPE = np.dtype([("type", "u1"), ("repeat_count", "u1"), ("cigar_ins", "u1"), ("cigar_del", "u1"), ("frag_len", "<u4"),
                          ("frag_pct", "<f8"), ("rec", "<i8")], align=True)
assert SUPPORT_DTYPE.itemsize == C.sizeof(Support)
SUPPORT_TYPES = ["SpanningFragment", "SpanningRead", "OverlappingRead"]


class Gt(C.Structure):
    _fields_ = [("tid", C.c_int32), ("start", C.c_uint32), ("stop", C.c_uint32), ("repeat", C.c_char * 7), ("allele1", C.c_double),
                ("allele2", C.c_double), ("overlapping_reads", C.c_uint32), ("anchored_reads", C.c_uint32), ("spanning_reads", C.c_uint32),
                ("spanning_pairs", C.c_uint32), ("left_clips", C.c_uint32), ("right_clips", C.c_uint32), ("sum_str_counts", C.c_uint32),
                ("expected_spanning_fragments", C.c_float), ("pctile", C.c_float), ("unplaced_reads", C.c_int32), ("depth", C.c_double),
                ("is_large", C.c_int)]


class Locus(C.Structure):
    _fields_ = [("b", Bounds), ("name", C.c_char * 128)]


class Unplaced(C.Structure):
    _fields_ = [("repeat", C.c_char * 7), ("count", C.c_int64)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_get_repeat.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_char * 6, C.POINTER(C.c_int)]
        L.orc_slide_by.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_slide_by.restype = C.c_int
        L.orc_reduce_repeat.argtypes = [C.c_char * 6]
        L.orc_reduce_repeat.restype = C.c_int
        L.orc_canonical_repeat.argtypes = [C.c_char * 6, C.c_char * 6]
        L.orc_min_rev_complement.argtypes = [C.c_char * 6]
        L.orc_median.argtypes = [C.c_void_p, C.c_double]
        L.orc_median.restype = C.c_int
        L.orc_p_repeat.argtypes = [C.POINTER(Tread)]
        L.orc_p_repeat.restype = C.c_double
        L.orc_adjust_by.argtypes = [C.POINTER(Tread), C.POINTER(Tread), C.POINTER(Opts), C.c_uint32]
        L.orc_adjust_by.restype = C.c_int
        L.orc_unplaced_pair.argtypes = [C.POINTER(Tread), C.POINTER(Tread), C.POINTER(Opts)]
        L.orc_unplaced_pair.restype = C.c_int
        L.orc_extract.argtypes = [C.POINTER(Records), C.c_int64, C.c_void_p, C.POINTER(Opts), C.c_void_p, C.c_int64,
                                  C.POINTER(C.c_int64)]
        L.orc_extract.restype = C.c_int64
        L.orc_score_record.argtypes = [C.POINTER(Records), C.c_int64, C.c_void_p, C.POINTER(Opts), C.POINTER(SegResult),
                                       C.POINTER(SegResult), C.POINTER(C.c_int)]
        L.orc_bounds_of.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint16, C.POINTER(Bounds)]
        L.orc_call_bounds.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_int, C.c_uint16, C.c_uint16,
                                      C.c_uint16, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_call_bounds.restype = C.c_int64
        L.orc_nim_hash_int.argtypes = [C.c_uint64]
        L.orc_nim_hash_int.restype = C.c_uint64
        L.orc_nim_hash_bytes.argtypes = [C.c_char_p, C.c_int]
        L.orc_nim_hash_bytes.restype = C.c_uint64
        L.orc_nim_hash_tidrep.argtypes = [C.c_int32, C.c_char * 6]
        L.orc_nim_hash_tidrep.restype = C.c_uint64
        L.orc_counttable_largest.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_bin_write.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_uint8, C.c_void_p, C.c_char_p, C.c_int32,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_bin_write.restype = C.c_int64
        L.orc_bounds_row.argtypes = [C.c_char_p, C.c_int, C.POINTER(Bounds), C.c_char_p]
        L.orc_bounds_row.restype = C.c_int
        L.orc_index_chrom.argtypes = [C.c_char_p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_index_chrom.restype = C.c_int64
        L.orc_cumulative.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_expected_spanning_probability.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int64]
        L.orc_expected_spanning_probability.restype = C.c_double
        L.orc_percentile.argtypes = [C.c_void_p, C.c_int64]
        L.orc_percentile.restype = C.c_double
        L.orc_median_depth.argtypes = [C.c_void_p, C.c_int64]
        L.orc_overlapping_read.argtypes = [C.POINTER(Records), C.c_int64, C.POINTER(Bounds), C.POINTER(Support)]
        L.orc_spanning_fragment.argtypes = [C.POINTER(Records), C.c_void_p, C.c_int64, C.c_int64, C.POINTER(Bounds), C.POINTER(Support), C.c_void_p]
        L.orc_spanners.argtypes = [C.POINTER(Records), C.c_void_p, C.POINTER(Bounds), C.c_int, C.c_void_p, C.c_uint8, C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.orc_spanners.restype = C.c_int64
        L.orc_spanning_read_est.argtypes = [C.c_void_p, C.c_int64] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_uint32)]
        L.orc_genotype.argtypes = [C.POINTER(Bounds), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint16,
                                   C.c_uint16, C.c_int, C.c_double, C.POINTER(Gt)]
        L.orc_call_row.argtypes = [C.c_char_p, C.c_int, C.POINTER(Gt), C.c_char_p]
        L.orc_call.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(Records), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_uint16, C.c_uint16, C.c_uint8, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                               C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        L.orc_call_members.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_uint16, C.c_uint16, C.c_uint16, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.POINTER(C.c_int64)]
        L.orc_call_members.restype = C.c_int64
        L.orc_parse_bed.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_int64]
        L.orc_parse_bed.restype = C.c_int64
        L.orc_parse_bounds.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.orc_parse_bounds.restype = C.c_int64
        L.orc_merge_text.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_uint16, C.c_uint16, C.c_uint16, C.c_char_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_cluster_group.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    return _LIB


def _rep6(s):
    if isinstance(s, str):
        s = s.encode()
    return (C.c_char * 6)(*s.ljust(6, b"\0"))


def get_repeat(read, p):
    """utils.nim:236 -> (unit str, count)"""
    if isinstance(read, str):
        read = read.encode()
    rep = (C.c_char * 6)()
    cnt = C.c_int(0)
    lib().orc_get_repeat(read, len(read), p, rep, C.byref(cnt))
    return rep.raw.rstrip(b"\0").decode(), cnt.value


def slide_by(s, k):
    if isinstance(s, str):
        s = s.encode()
    out = np.zeros(max(1, len(s)), dtype=np.uint64)
    n = lib().orc_slide_by(s, len(s), k, out.ctypes.data)
    return out[:n].copy()


def reduce_repeat(rep):
    r = _rep6(rep)
    m = lib().orc_reduce_repeat(r)
    return m, r.raw.rstrip(b"\0").decode()


def canonical_repeat(rep):
    out = (C.c_char * 6)()
    lib().orc_canonical_repeat(_rep6(rep), out)
    return out.raw.rstrip(b"\0").decode()


def min_rev_complement(rep):
    r = _rep6(rep)
    lib().orc_min_rev_complement(r)
    return r.raw.rstrip(b"\0").decode()


def median(frag, pct=0.5):
    frag = np.ascontiguousarray(frag, dtype=np.uint32)
    assert frag.size == 4096
    return lib().orc_median(frag.ctypes.data, pct)


def make_tread(**kw):
    t = Tread()
    t.split = 3
    for k, v in kw.items():
        if k == "repeat":
            v = v.encode() if isinstance(v, str) else v
        setattr(t, k, v)
    return t


def make_opts(median_fragment_length=0, proportion_repeat=0.8, min_mapq=40):
    return Opts(median_fragment_length, proportion_repeat, min_mapq)


class RecordsView:
    """Keeps numpy arrays alive behind an orc_records struct."""

    def __init__(self, rec):
        # rec: strling_amd.records.RecordBatch-like object with the numpy fields below
        self.keep = dict(
            tid=np.ascontiguousarray(rec.tid, np.int32), pos=np.ascontiguousarray(rec.pos, np.int32),
            mtid=np.ascontiguousarray(rec.mtid, np.int32), mpos=np.ascontiguousarray(rec.mpos, np.int32),
            flag=np.ascontiguousarray(rec.flag, np.uint16), mapq=np.ascontiguousarray(rec.mapq, np.uint8),
            cigar_off=np.ascontiguousarray(rec.cigar_off, np.uint32), cigar=np.ascontiguousarray(rec.cigar, np.uint32),
            seq_off=np.ascontiguousarray(rec.seq_off, np.uint64), l_seq=np.ascontiguousarray(rec.l_seq, np.int32),
            seq4=np.ascontiguousarray(rec.seq4, np.uint8), qname_off=np.ascontiguousarray(rec.qname_off, np.uint64),
            qnames=np.frombuffer(bytes(rec.qnames) + b"\0", dtype=np.uint8))
        k = self.keep
        self.n = int(k["tid"].size)
        self.c = Records(self.n, *[k[f].ctypes.data for f in
                                   ("tid", "pos", "mtid", "mpos", "flag", "mapq", "cigar_off", "cigar", "seq_off",
                                    "l_seq", "seq4", "qname_off", "qnames")])


class GenomeView:
    def __init__(self, g):
        # g: object with n_tid, has_chrom(u8[n_tid]), iv_off(i64[n_tid+1]), iv_start(i32), iv_stop(i32)
        off = np.ascontiguousarray(g.iv_off, np.int64)
        st = np.array(g.iv_start, np.int32)
        en = np.array(g.iv_stop, np.int32)
        mx = np.zeros(int(g.n_tid), np.int32)
        for t in range(int(g.n_tid)):          # lapify(): sort by start, remember the longest interval
            a, b = int(off[t]), int(off[t + 1])
            if b > a:
                o = np.argsort(st[a:b], kind="stable")
                st[a:b], en[a:b] = st[a:b][o], en[a:b][o]
                mx[t] = int((en[a:b] - st[a:b]).max())
        self.keep = dict(has=np.ascontiguousarray(g.has_chrom, np.uint8), off=off, st=st, en=en, mx=mx)
        k = self.keep
        self.c = GenomeStr(int(g.n_tid), k["has"].ctypes.data, k["off"].ctypes.data,
                           k["st"].ctypes.data if k["st"].size else None, k["en"].ctypes.data if k["en"].size else None,
                           k["mx"].ctypes.data)


def extract(rec, genome, opts, n_tail=-1):
    """extract.nim:308-329 over a record batch -> structured array of treads (TREAD_DTYPE)."""
    rv = RecordsView(rec)
    gv = GenomeView(genome) if genome is not None else None
    cap = max(1024, rv.n // 4)
    while True:
        out = np.zeros(cap, dtype=TREAD_DTYPE)
        need = C.c_int64(0)
        n = lib().orc_extract(C.byref(rv.c), n_tail, C.byref(gv.c) if gv else None, C.byref(opts), out.ctypes.data, cap,
                              C.byref(need))
        if need.value <= cap:
            return out[:n].copy()
        cap = need.value


def score_records(rec, genome, opts, idx=None):
    """Per-record scorer outputs: list of (skipped, whole(unit,count,al), [4 soft (unit,count,al)])"""
    rv = RecordsView(rec)
    gv = GenomeView(genome) if genome is not None else None
    res = []
    whole = SegResult()
    soft = (SegResult * 4)()
    sk = C.c_int(0)
    it = range(rv.n) if idx is None else idx
    for i in it:
        lib().orc_score_record(C.byref(rv.c), int(i), C.byref(gv.c) if gv else None, C.byref(opts), C.byref(whole), soft,
                               C.byref(sk))
        res.append((sk.value, (whole.rep.rstrip(b"\0").decode(), whole.count, whole.align_length),
                    [(s.rep.rstrip(b"\0").decode(), s.count, s.align_length) for s in soft]))
    return res


def score_records_packed(rec, genome, opts):
    """Vector form used by parity tests: returns (skipped u8[n], whole_unit S6[n], whole_count i32[n],
    soft_unit S6[n,4], soft_count i32[n,4])."""
    rv = RecordsView(rec)
    gv = GenomeView(genome) if genome is not None else None
    n = rv.n
    skipped = np.zeros(n, np.uint8)
    wu = np.zeros(n, "S6")
    wc = np.zeros(n, np.int32)
    su = np.zeros((n, 4), "S6")
    sc = np.zeros((n, 4), np.int32)
    whole = SegResult()
    soft = (SegResult * 4)()
    sk = C.c_int(0)
    f = lib().orc_score_record
    rp, gp, op = C.byref(rv.c), (C.byref(gv.c) if gv else None), C.byref(opts)
    for i in range(n):
        f(rp, i, gp, op, C.byref(whole), soft, C.byref(sk))
        skipped[i] = sk.value
        wu[i] = whole.rep
        wc[i] = whole.count
        for j in range(4):
            su[i, j] = soft[j].rep
            sc[i, j] = soft[j].count
    return skipped, wu, wc, su, sc


def bounds_of(treads, left_most=0, right_most=0, max_clip_dist=200):
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    b = Bounds()
    lib().orc_bounds_of(t.ctypes.data, t.size, left_most, right_most, max_clip_dist, C.byref(b))
    return b


def call_bounds(treads, mode, window, min_support=5, min_clip=0, min_clip_total=0, max_clip_dist=200):
    """merge.nim:172-187 (mode 0) / call.nim:223-235 (mode 1) -> (bounds array, [(unit, count)] unplaced)"""
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    cap = max(16, t.size)
    out = np.zeros(cap, dtype=BOUNDS_DTYPE)
    unpl = (Unplaced * 8192)()
    nu = C.c_int64(0)
    n = lib().orc_call_bounds(t.ctypes.data, t.size, mode, window, min_support, min_clip, min_clip_total, max_clip_dist,
                              out.ctypes.data, cap, unpl, 8192, C.byref(nu))
    return out[:n].copy(), [(unpl[i].repeat.decode(), unpl[i].count) for i in range(min(nu.value, 8192))]


_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32)


def cluster_group(treads, max_dist, min_supporting_reads):
    """cluster.nim:364-374 on one sorted group -> list of (reads array, left_most, right_most)"""
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    res = []

    def cb(ud, reads, n, lm, rm):
        buf = (C.c_char * (n * TREAD_DTYPE.itemsize)).from_address(reads)
        res.append((np.frombuffer(buf, dtype=TREAD_DTYPE, count=n).copy(), lm, rm))

    lib().orc_cluster_group(t.ctypes.data, t.size, max_dist, min_supporting_reads, _CB(cb), None)
    return res


def index_chrom(seq, p=0.8, window=100, step=60):
    """genome_strs.nim:61-92 on one (upper-cased) chromosome -> [(start, stop, unit)]"""
    if isinstance(seq, str):
        seq = seq.encode()
    cap = max(16, len(seq) // step + 2)
    st = np.zeros(cap, np.int64)
    en = np.zeros(cap, np.int64)
    un = np.zeros(cap, "S7")
    n = lib().orc_index_chrom(seq, len(seq), p, window, step, st.ctypes.data, en.ctypes.data, un.ctypes.data, cap)
    if n < 0:
        raise AssertionError("doAssert of genome_strs.trim would fire")
    return [(int(st[i]), int(en[i]), un[i].decode()) for i in range(n)]


def make_bounds(tid, left, right, repeat, **kw):
    b = Bounds()
    b.tid, b.left, b.right, b.repeat = tid, left, right, repeat.encode()
    for k, v in kw.items():
        setattr(b, k, v)
    return b


def _as_bounds(b):
    return Bounds.from_buffer_copy(b.tobytes()) if isinstance(b, np.void) else b


def median_depth(depths):
    d = np.ascontiguousarray(depths, np.int64)
    return lib().orc_median_depth(d.ctypes.data, d.size)


def overlapping_read(rec, i, b):
    """collect.nim:97-119 -> None or Support"""
    rv = RecordsView(rec)
    s = Support()
    return s if lib().orc_overlapping_read(C.byref(rv.c), i, C.byref(_as_bounds(b)), C.byref(s)) else None


def spanning_fragment(rec, l, r, b, frag):
    rv = RecordsView(rec)
    s = Support()
    frag = np.ascontiguousarray(frag, np.uint32)
    isz = np.ascontiguousarray(rec.isize if rec.isize is not None else np.zeros(rec.n), np.int32)
    ok = lib().orc_spanning_fragment(C.byref(rv.c), isz.ctypes.data, l, r, C.byref(_as_bounds(b)), C.byref(s), frag.ctypes.data)
    return s if ok else None


def spanners(rec, b, window, frag, min_mapq=20):
    """collect.nim:132-182 over all records of `rec` -> (support array, median_depth, expected_spanners)"""
    rv = RecordsView(rec)
    frag = np.ascontiguousarray(frag, np.uint32)
    isz = np.ascontiguousarray(rec.isize if rec.isize is not None else np.zeros(rec.n), np.int32)
    cap = 2 * rec.n + 16
    out = np.zeros(cap, SUPPORT_DTYPE)
    md, es = C.c_int(0), C.c_float(0)
    n = lib().orc_spanners(C.byref(rv.c), isz.ctypes.data, C.byref(_as_bounds(b)), window, frag.ctypes.data, min_mapq, out.ctypes.data, cap,
                           C.byref(md), C.byref(es))
    return out[:n].copy(), md.value, es.value


def spanning_read_est(supports):
    s = np.ascontiguousarray(supports, SUPPORT_DTYPE)
    v = [C.c_double(0) for _ in range(4)]
    sup = C.c_uint32(0)
    lib().orc_spanning_read_est(s.ctypes.data, s.size, *[C.byref(x) for x in v], C.byref(sup))
    return dict(allele1_bp=v[0].value, allele2_bp=v[1].value, allele1_ru=v[2].value, allele2_ru=v[3].value, supporting_reads=sup.value)


def _targets(targets):
    names = (C.c_char_p * len(targets))(*[n.encode() for n, _ in targets])
    lens = np.array([l for _, l in targets], np.uint32)
    return names, lens


def parse_bed(text, targets, window):
    names, lens = _targets(targets)
    out = (Locus * 4096)()
    n = lib().orc_parse_bed(text.encode(), names, lens.ctypes.data, len(targets), window, out, 4096)
    if n < 0:
        raise ValueError("parse_bed: the reference quits on this input")
    return [out[i] for i in range(n)]


def parse_bounds(text, targets):
    names, lens = _targets(targets)
    out = (Locus * 4096)()
    n = lib().orc_parse_bounds(text.encode(), names, len(targets), out, 4096)
    if n < 0:
        raise ValueError("parse_bounds: the reference quits on this input")
    return [out[i] for i in range(n)]


def merge_text(treads, window, targets, min_support=5, min_clip=0, min_clip_total=0, max_clip_dist=200, loci_text=None):
    """merge.nim:154-187 -> text of -bounds.txt (qname_id = sample index), with optional -l loci"""
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    names, lens = _targets(targets)
    cap = 1 << 22
    buf = C.create_string_buffer(cap)
    need = C.c_int64(0)
    rc = lib().orc_merge_text(t.ctypes.data, t.size, window, min_support, min_clip, min_clip_total, max_clip_dist,
                              loci_text.encode() if loci_text is not None else None, names, lens.ctypes.data, len(targets), buf, cap, C.byref(need))
    assert rc == 0 and need.value < cap
    return buf.raw[:need.value].decode()


def call(treads, rec, frag, min_support=5, min_clip=0, min_clip_total=0, min_mapq=40, loci_text=None, bounds_text=None):
    """call.nim:111-285: (bounds.txt, genotype.txt, unplaced.txt) texts.  tread.qname_id indexes rec's qnames."""
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    rv = RecordsView(rec)
    frag = np.ascontiguousarray(frag, np.uint32)
    isz = np.ascontiguousarray(rec.isize if rec.isize is not None else np.zeros(rec.n), np.int32)
    names, lens = _targets(rec.targets)
    caps = [1 << 22, 1 << 22, 1 << 16]
    bufs = [C.create_string_buffer(c) for c in caps]
    ns = [C.c_int64(0) for _ in range(3)]
    lib().orc_call(t.ctypes.data, t.size, rv.keep["qname_off"].ctypes.data, rv.keep["qnames"].ctypes.data, C.byref(rv.c), isz.ctypes.data,
                   frag.ctypes.data, names, min_support, min_clip, min_clip_total, min_mapq, bufs[0], caps[0], bufs[1], caps[1], bufs[2], caps[2],
                   C.byref(ns[0]), C.byref(ns[1]), C.byref(ns[2]), loci_text.encode() if loci_text is not None else None,
                   bounds_text.encode() if bounds_text is not None else None, lens.ctypes.data, len(rec.targets))
    assert all(n.value < c for n, c in zip(ns, caps))
    return tuple(b.raw[:n.value].decode() for b, n in zip(bufs, ns))


def cluster_members_call(treads, window, min_support=5, max_clip_dist=200, min_clip=0, min_clip_total=0):
    """per bound of call_bounds(mode 1): index array into treads (c.reads, call.nim:246)"""
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    off = np.zeros(t.size + 2, np.int64)
    mem = np.zeros(t.size + 1, np.int64)
    nm = C.c_int64(0)
    nb = lib().orc_call_members(t.ctypes.data, t.size, window, min_support, min_clip, min_clip_total, max_clip_dist, off.ctypes.data,
                                mem.ctypes.data, mem.size, C.byref(nm))
    return [mem[off[j]:off[j + 1]].copy() for j in range(nb)]


def bounds_row(b, chrom):
    buf = C.create_string_buffer(512)
    if isinstance(b, np.void):
        bb = Bounds.from_buffer_copy(b.tobytes())
    else:
        bb = b
    lib().orc_bounds_row(buf, 512, C.byref(bb), chrom.encode())
    return buf.value.decode()


def bin_write(proportion_repeat, min_mapq, frag, sam_header, treads, qname_off, qnames):
    t = np.ascontiguousarray(treads, dtype=TREAD_DTYPE)
    frag = np.ascontiguousarray(frag, np.uint32)
    qo = np.ascontiguousarray(qname_off, np.uint64)
    qn = np.frombuffer(bytes(qnames) + b"\0", dtype=np.uint8)
    hdr = sam_header.encode() if isinstance(sam_header, str) else sam_header
    need = lib().orc_bin_write(None, 0, proportion_repeat, min_mapq, frag.ctypes.data, hdr, len(hdr), t.ctypes.data, t.size,
                               qo.ctypes.data, qn.ctypes.data)
    buf = np.zeros(need, np.uint8)
    lib().orc_bin_write(buf.ctypes.data, need, proportion_repeat, min_mapq, frag.ctypes.data, hdr, len(hdr), t.ctypes.data,
                        t.size, qo.ctypes.data, qn.ctypes.data)
    return buf.tobytes()
