"""Host-side containers for the BAM-native record batches the C ABI consumes (numpy SoA).

Mirrors what hts-nim hands to src/strpkg/extract.nim: tid/pos/mate fields, flag, mapq, cigar,
4-bit SEQ (BAM nibble packing, every read 16-byte aligned so the device can issue aligned
16-byte loads) and qname.
"""
from dataclasses import dataclass, field
import re
import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
_NT16_LUT = np.full(256, 15, dtype=np.uint8)
for _i, _c in enumerate(NT16):
    _NT16_LUT[ord(_c)] = _i
    _NT16_LUT[ord(_c.lower())] = _i
CIGAR_OPS = "MIDNSHP=X"
_CIG_RE = re.compile(r"(\d+)([MIDNSHP=X])")


def encode_cigar(s):
    if s == "*" or s == "":
        return []
    return [(int(n) << 4) | CIGAR_OPS.index(op) for n, op in _CIG_RE.findall(s)]


def pack_seq4(seqs):
    """list of ASCII sequences -> (seq4 uint8 array with 32 B slack, seq_off uint64[n], l_seq int32[n])"""
    n = len(seqs)
    l_seq = np.fromiter((len(s) for s in seqs), dtype=np.int32, count=n)
    nbytes = ((l_seq + 1) // 2 + 15) // 16 * 16
    seq_off = np.zeros(n, dtype=np.uint64)
    if n:
        seq_off[1:] = np.cumsum(nbytes[:-1], dtype=np.uint64)
    total = int(nbytes.sum()) + 32
    seq4 = np.zeros(total, dtype=np.uint8)
    for i, s in enumerate(seqs):
        if not s:
            continue
        b = s if isinstance(s, bytes) else s.encode()
        codes = _NT16_LUT[np.frombuffer(b, dtype=np.uint8)]
        if codes.size & 1:
            codes = np.append(codes, 0)
        o = int(seq_off[i])
        seq4[o:o + codes.size // 2] = (codes[0::2] << 4) | codes[1::2]
    return seq4, seq_off, l_seq


def pack_codes4(codes, l_seq):
    """codes: uint8 [n, Lmax] nibble values (0..15), l_seq int32[n] -> fixed-stride packed SEQ.
    Vectorised path used by the synthetic generators."""
    n, lmax = codes.shape
    stride = ((lmax + 1) // 2 + 15) // 16 * 16
    c = np.zeros((n, stride * 2), dtype=np.uint8)
    c[:, :lmax] = codes
    mask = np.arange(stride * 2)[None, :] < l_seq[:, None]
    c &= mask.astype(np.uint8) * 0xF
    packed = (c[:, 0::2] << 4) | c[:, 1::2]
    seq4 = np.zeros(n * stride + 32, dtype=np.uint8)
    seq4[: n * stride] = packed.reshape(-1)
    seq_off = (np.arange(n, dtype=np.uint64) * np.uint64(stride))
    return seq4, seq_off


@dataclass
class GenomeStr:
    """ref.fasta.str intervals flattened per tid (genome_strs.nim:107-141 / read_bed.nim:30-50)."""
    n_tid: int
    has_chrom: np.ndarray
    iv_off: np.ndarray
    iv_start: np.ndarray
    iv_stop: np.ndarray

    @staticmethod
    def from_lists(n_tid, per_tid):
        """per_tid: dict tid -> list of (start, stop). A tid present (even empty) is a key of the table."""
        has = np.zeros(n_tid, np.uint8)
        off = np.zeros(n_tid + 1, np.int64)
        st, en = [], []
        for t in range(n_tid):
            ivs = per_tid.get(t)
            if ivs is not None:
                has[t] = 1
                for a, b in ivs:
                    st.append(a)
                    en.append(b)
            off[t + 1] = len(st)
        return GenomeStr(n_tid, has, off, np.asarray(st, np.int32), np.asarray(en, np.int32))


@dataclass
class RecordBatch:
    tid: np.ndarray
    pos: np.ndarray
    mtid: np.ndarray
    mpos: np.ndarray
    flag: np.ndarray
    mapq: np.ndarray
    cigar_off: np.ndarray
    cigar: np.ndarray
    seq_off: np.ndarray
    l_seq: np.ndarray
    seq4: np.ndarray
    qname_off: np.ndarray
    qnames: bytes
    isize: np.ndarray = None
    targets: list = field(default_factory=list)  # [(name, length)]

    @property
    def n(self):
        return int(self.tid.size)

    def slice(self, a, b):
        """records [a, b) as a batch of their own (offsets rebased; SEQ keeps its 16-byte alignment)"""
        c0, c1 = int(self.cigar_off[a]), int(self.cigar_off[b])
        q0, q1 = int(self.qname_off[a]), int(self.qname_off[b])
        s0 = int(self.seq_off[a]) if b > a else 0
        s1 = (int(self.seq_off[b - 1]) + (int(self.l_seq[b - 1]) + 1) // 2 + 15) // 16 * 16 if b > a else 0
        seq4 = np.zeros(s1 - s0 + 32, np.uint8)
        seq4[: s1 - s0] = self.seq4[s0:s1]
        return RecordBatch(self.tid[a:b], self.pos[a:b], self.mtid[a:b], self.mpos[a:b], self.flag[a:b], self.mapq[a:b],
                           (self.cigar_off[a:b + 1] - np.uint32(c0)).astype(np.uint32), self.cigar[c0:c1],
                           (self.seq_off[a:b] - np.uint64(s0)).astype(np.uint64), self.l_seq[a:b], seq4,
                           (self.qname_off[a:b + 1] - np.uint64(q0)).astype(np.uint64), bytes(self.qnames[q0:q1]),
                           None if self.isize is None else self.isize[a:b], self.targets)

    def qname(self, i):
        return self.qnames[int(self.qname_off[i]):int(self.qname_off[i + 1])]

    def sequence(self, i):
        o, l = int(self.seq_off[i]), int(self.l_seq[i])
        b = self.seq4[o:o + (l + 1) // 2]
        out = np.empty(2 * b.size, np.uint8)
        out[0::2] = b >> 4
        out[1::2] = b & 15
        return "".join(NT16[c] for c in out[:l])

    @staticmethod
    def from_fields(tid, pos, mtid, mpos, flag, mapq, cigars, seqs, qnames, isize=None, targets=None):
        n = len(seqs)
        cig_off = np.zeros(n + 1, np.uint32)
        cig = []
        for i, c in enumerate(cigars):
            ops = encode_cigar(c) if isinstance(c, str) else list(c)
            cig.extend(ops)
            cig_off[i + 1] = len(cig)
        seq4, seq_off, l_seq = pack_seq4(seqs)
        qo = np.zeros(n + 1, np.uint64)
        qb = bytearray()
        for i, q in enumerate(qnames):
            qb += q if isinstance(q, bytes) else q.encode()
            qo[i + 1] = len(qb)
        return RecordBatch(np.asarray(tid, np.int32), np.asarray(pos, np.int32), np.asarray(mtid, np.int32),
                           np.asarray(mpos, np.int32), np.asarray(flag, np.uint16), np.asarray(mapq, np.uint8), cig_off,
                           np.asarray(cig, np.uint32), seq_off, l_seq, seq4, qo, bytes(qb),
                           None if isize is None else np.asarray(isize, np.int32), targets or [])

    @staticmethod
    def from_sam(header_text, lines):
        """Minimal SAM text -> RecordBatch (what hts-nim's from_string does in the reference tests)."""
        targets = []
        for h in header_text.splitlines():
            if h.startswith("@SQ"):
                d = dict(f.split(":", 1) for f in h.split("\t")[1:])
                targets.append((d["SN"], int(d["LN"])))
        names = {t[0]: i for i, t in enumerate(targets)}
        f = dict(tid=[], pos=[], mtid=[], mpos=[], flag=[], mapq=[], cigars=[], seqs=[], qnames=[], isize=[])
        for ln in lines:
            c = ln.rstrip("\n").split("\t")
            tid = names.get(c[2], -1)
            f["qnames"].append(c[0]); f["flag"].append(int(c[1])); f["tid"].append(tid)
            f["pos"].append(int(c[3]) - 1); f["mapq"].append(int(c[4])); f["cigars"].append(c[5])
            f["mtid"].append(tid if c[6] == "=" else names.get(c[6], -1)); f["mpos"].append(int(c[7]) - 1)
            f["isize"].append(int(c[8])); f["seqs"].append("" if c[9] == "*" else c[9])
        return RecordBatch.from_fields(targets=targets, **f)


UNIT_BASES = "CATG"  # kmer module code order


def unpack_result(w):
    """packed scorer word -> (unit str, count, skipped)"""
    w = int(w)
    k = (w >> 12) & 7
    code = w & 0xFFF
    unit = "".join(UNIT_BASES[(code >> (2 * (k - 1 - j))) & 3] for j in range(k))
    return unit, w >> 16, bool(w & 0x8000)
