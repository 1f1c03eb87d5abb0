"""CRAM 3.0 writer for tests (pure Python; htslib / samtools are not available here).

Writes a spec-conformant CRAM v3.0 (CRAMv3.pdf: file definition, containers, slices, blocks, the compression header's
preservation / data-series / tag maps, the record layout of section 10, rANS 4x8 of section 13) from a records.RecordBatch
and a reference, so that the `strling` CLI's CRAM reader (csrc/cli/cram_reader.cpp: `strling extract -f FASTA x.cram`,
extract.nim:253,278-279) can be exercised end to end.  Deliberately VARIED rather than compact: data series go through
EXTERNAL blocks stored raw / gzip / rANS order 0 / rANS order 1 in rotation and through the core bit stream (HUFFMAN with one
and with several symbols, BETA, GAMMA, SUBEXP), pairs inside a slice are linked "mate downstream", pairs across slices are
"detached", bases come from the reference plus substitution / base / insertion / soft-clip / deletion / skip features.
Test infrastructure: nothing here is used by the product.
"""
import gzip
import hashlib
import struct
import zlib

import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
EOF_V3 = bytes.fromhex("0f000000ffffffff0fe0454f4600000000010005bdd94f0001000606010001000100ee63014b")


def itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80:
        return bytes([v])
    if v < 0x4000:
        return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000:
        return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000:
        return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | ((v >> 28) & 0x0F), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v):
    v &= (1 << 64) - 1
    if v < 0x80:
        return bytes([v])
    for n in range(1, 8):                      # n extra bytes
        if v < (1 << (7 * (n + 1))):
            lead = (0xFF << (8 - n)) & 0xFF
            return bytes([lead | (v >> (8 * n))]) + (v & ((1 << (8 * n)) - 1)).to_bytes(n, "big")
    return bytes([0xFF]) + v.to_bytes(8, "big")


# ---- rANS 4x8 (CRAMv3 section 13; the encoder mirrors the decoder's interleaving: states 0..3 take symbols i, i+1, i+2, i+3) ----
RANS_L = 1 << 23
TF = 12


def _norm_freqs(cnt):
    """counts -> frequencies summing to 4096, every present symbol >= 1"""
    tot = int(cnt.sum())
    F = np.zeros(256, np.int64)
    if tot == 0:
        return F
    nz = np.nonzero(cnt)[0]
    F[nz] = np.maximum(1, (cnt[nz].astype(np.int64) * 4096) // tot)
    d = 4096 - int(F.sum())
    big = int(nz[np.argmax(F[nz])])
    F[big] += d
    if F[big] < 1:                                # (pathological: hundreds of rare symbols) take from others instead
        F[big] = 1
        d = 4096 - int(F.sum())
        for s in sorted(nz, key=lambda s: -F[s]):
            take = min(int(F[s]) - 1, -d)
            F[s] -= take
            d += take
            if d == 0:
                break
    assert F.sum() == 4096 and (F[nz] >= 1).all()
    return F


def _freq_table(F):
    """the run-length coded symbol / frequency list of one context"""
    out = bytearray()
    syms = [int(s) for s in np.nonzero(F)[0]]
    rle = 0
    for k, s in enumerate(syms):
        if rle:
            rle -= 1
        else:
            out.append(s)
            if k and syms[k - 1] == s - 1:
                run = 0
                while k + 1 + run < len(syms) and syms[k + 1 + run] == s + 1 + run:
                    run += 1
                out.append(run)
                rle = run
        f = int(F[s])
        if f < 128:
            out.append(f)
        else:
            out += bytes([128 | (f >> 8), f & 0xFF])
    out.append(0)
    return bytes(out)


def _put(x, emit, start, freq):
    x_max = ((RANS_L >> TF) << 8) * freq
    while x >= x_max:
        emit.append(x & 0xFF)
        x >>= 8
    return ((x // freq) << TF) + (x % freq) + start


def rans_encode(data, order):
    data = bytes(data)
    n = len(data)
    if order == 1 and n < 4:
        order = 0
    a = np.frombuffer(data, np.uint8)
    emit = []                                         # bytes in the order the (backwards writing) encoder produces them
    R = [RANS_L] * 4
    if order == 0:
        F = _norm_freqs(np.bincount(a, minlength=256))
        C = np.concatenate([[0], np.cumsum(F)[:-1]])
        table = _freq_table(F)
        tail = n & 3
        for k in range(tail - 1, -1, -1):             # the last n & 3 symbols belong to states 0 .. tail-1
            s = data[n - tail + k]
            R[k] = _put(R[k], emit, int(C[s]), int(F[s]))
        for i in range(n - tail, 0, -4):
            for k in (3, 2, 1, 0):
                s = data[i - 4 + k]
                R[k] = _put(R[k], emit, int(C[s]), int(F[s]))
    else:
        q = n >> 2
        cnt = np.zeros((256, 256), np.int64)
        prev = np.concatenate([[0], a[:-1]]).astype(np.int64)
        for k in range(4):                            # every quarter starts in context 0
            prev[k * q] = 0
        np.add.at(cnt, (prev, a), 1)
        Fs = {c: _norm_freqs(cnt[c]) for c in np.nonzero(cnt.sum(axis=1))[0]}
        Cs = {c: np.concatenate([[0], np.cumsum(F)[:-1]]) for c, F in Fs.items()}
        table = bytearray()
        ctxs = sorted(int(c) for c in Fs)
        rle = 0
        for k, c in enumerate(ctxs):
            if rle:
                rle -= 1
            else:
                table.append(c)
                if k and ctxs[k - 1] == c - 1:
                    run = 0
                    while k + 1 + run < len(ctxs) and ctxs[k + 1 + run] == c + 1 + run:
                        run += 1
                    table.append(run)
                    rle = run
            table += _freq_table(Fs[c])
        table.append(0)
        table = bytes(table)
        # the remainder behind the four quarters belongs to state 3
        for i in range(n - 1, 4 * q - 1, -1):
            c = int(prev[i])
            R[3] = _put(R[3], emit, int(Cs[c][data[i]]), int(Fs[c][data[i]]))
        for j in range(q - 1, -1, -1):
            for k in (3, 2, 1, 0):
                i = k * q + j
                c = int(prev[i])
                R[k] = _put(R[k], emit, int(Cs[c][data[i]]), int(Fs[c][data[i]]))
    for k in (3, 2, 1, 0):                            # flush: state 0 ends up first in the stream
        x = R[k]
        emit += [(x >> 24) & 0xFF, (x >> 16) & 0xFF, (x >> 8) & 0xFF, x & 0xFF]
    body = table + bytes(reversed(emit))
    return bytes([order]) + struct.pack("<II", len(body), n) + body



# ---- CRAM 3.1 codecs (CRAMcodecs sections 3 and 5): test encoders for cli/cram_codecs.cpp ----------------------------
def u7(v):
    """variable-length integer, most significant 7-bit group first"""
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    return bytes(reversed(out))


def _alphabet(syms):
    """the run-length coded list of the symbols that occur (as in _freq_table, without the frequencies)"""
    out = bytearray()
    syms = sorted(int(x) for x in syms)
    rle = 0
    for k, x in enumerate(syms):
        if rle:
            rle -= 1
            continue
        out.append(x)
        if k and syms[k - 1] == x - 1:
            run = 0
            while k + 1 + run < len(syms) and syms[k + 1 + run] == x + 1 + run:
                run += 1
            out.append(run)
            rle = run
    out.append(0)
    return bytes(out)


def _norm_to(cnt, bits):
    """counts -> frequencies summing to 2^bits (every present symbol >= 1)"""
    F = _norm_freqs(np.asarray(cnt, np.int64))
    if bits == 12:
        return F
    tot, want = 4096, 1 << bits
    nz = np.nonzero(F)[0]
    G = np.zeros(256, np.int64)
    G[nz] = np.maximum(1, F[nz] * want // tot)
    d = want - int(G.sum())
    order = sorted(nz, key=lambda x: -G[x])
    k = 0
    while d:
        x = order[k % len(order)]
        step = 1 if d > 0 else (-1 if G[x] > 1 else 0)
        G[x] += step
        d -= step
        k += 1
    return G


def _put16(x, emit, start, freq, shift):
    x_max = (((1 << 15) >> shift) << 16) * freq
    while x >= x_max:
        emit.append(x & 0xFFFF)
        x >>= 16
    return ((x // freq) << shift) + (x % freq) + start


def _nx16_order0(data, N):
    a = np.frombuffer(data, np.uint8)
    n = len(data)
    F = _norm_freqs(np.bincount(a, minlength=256))
    C = np.concatenate([[0], np.cumsum(F)[:-1]])
    present = np.nonzero(F)[0]
    table = _alphabet(present) + b"".join(u7(int(F[x])) for x in present)
    emit, R = [], [1 << 15] * N
    for i in range(n - 1, -1, -1):
        k = i % N
        x = data[i]
        R[k] = _put16(R[k], emit, int(C[x]), int(F[x]), 12)
    return table + b"".join(struct.pack("<I", R[k]) for k in range(N)) + b"".join(struct.pack("<H", w) for w in reversed(emit))


def _nx16_order1(data, N, shift=12, compress_table=False):
    a = np.frombuffer(data, np.uint8)
    n = len(data)
    seg = n // N
    prev = np.concatenate([[0], a[:-1]]).astype(np.int64)
    for k in range(N):
        if k * seg < n:
            prev[k * seg] = 0
    cnt = np.zeros((256, 256), np.int64)
    np.add.at(cnt, (prev, a.astype(np.int64)), 1)
    alpha = sorted(set(int(x) for x in np.nonzero(cnt.sum(axis=1))[0]) | set(int(x) for x in np.nonzero(cnt.sum(axis=0))[0]) | {0})
    Fs, Cs = {}, {}
    tab = bytearray(_alphabet(alpha))
    for c in alpha:
        F = _norm_to(cnt[c], shift) if cnt[c].sum() else np.zeros(256, np.int64)
        Fs[c], Cs[c] = F, np.concatenate([[0], np.cumsum(F)[:-1]])
        run = 0
        for k, x in enumerate(alpha):
            if run:
                run -= 1
                continue
            tab += u7(int(F[x]))
            if not F[x]:
                while k + 1 + run < len(alpha) and F[alpha[k + 1 + run]] == 0 and run < 255:
                    run += 1
                tab.append(run)
    emit, R = [], [1 << 15] * N
    for i in range(n - 1, seg * N - 1, -1):
        c = int(prev[i])
        R[N - 1] = _put16(R[N - 1], emit, int(Cs[c][data[i]]), int(Fs[c][data[i]]), shift)
    for j in range(seg - 1, -1, -1):
        for k in range(N - 1, -1, -1):
            i = k * seg + j
            c = int(prev[i])
            R[k] = _put16(R[k], emit, int(Cs[c][data[i]]), int(Fs[c][data[i]]), shift)
    if compress_table:
        ct = _nx16_order0(bytes(tab), 4)
        head = bytes([(shift << 4) | 1]) + u7(len(tab)) + u7(len(ct)) + ct
    else:
        head = bytes([shift << 4]) + bytes(tab)
    return head + b"".join(struct.pack("<I", R[k]) for k in range(N)) + b"".join(struct.pack("<H", w) for w in reversed(emit))


def rans_nx16_encode(data, order=0, x32=False, pack=False, rle=False, stripe=0, cat=False, nosz=False, shift=12, compress_table=False):
    """one rANS Nx16 stream: flags | size | [PACK map, packed size] | [RLE metadata, literal count] | payload"""
    data = bytes(data)
    flags = (order & 1) | (4 if x32 else 0) | (8 if stripe else 0) | (0x10 if nosz else 0) | (0x20 if cat else 0)
    out = bytearray()
    if stripe:
        subs = [rans_nx16_encode(data[j::stripe], order=order, x32=x32, nosz=True, pack=pack, rle=rle, shift=shift) for j in range(stripe)]
        out = bytearray([flags]) + (b"" if nosz else u7(len(data))) + bytes([stripe]) + b"".join(u7(len(x)) for x in subs) + b"".join(subs)
        return bytes(out)
    cur = data
    meta = bytearray()
    if pack:
        syms = sorted(set(cur))
        if len(syms) <= 16:
            flags |= 0x80
            idx = {x: k for k, x in enumerate(syms)}
            bits = 0 if len(syms) <= 1 else 1 if len(syms) <= 2 else 2 if len(syms) <= 4 else 4
            if bits:
                per = 8 // bits
                packed = bytearray()
                for i in range(0, len(cur), per):
                    v = 0
                    for k, x in enumerate(cur[i:i + per]):
                        v |= idx[x] << (bits * k)
                    packed.append(v)
                cur = bytes(packed)
            else:
                cur = b""
            meta += bytes([len(syms)]) + bytes(syms) + u7(len(cur))
    rep = set()                                    # symbols worth a run length: those that ever repeat (none: no RLE -- a count of 0 means 256)
    for i in range(1, len(cur)):
        if cur[i] == cur[i - 1]:
            rep.add(cur[i])
    if rle and rep:
        flags |= 0x40
        lits, runs = bytearray(), bytearray()
        i = 0
        while i < len(cur):
            x = cur[i]
            j = i + 1
            if x in rep:
                while j < len(cur) and cur[j] == x:
                    j += 1
                runs += u7(j - i - 1)
            lits.append(x)
            i = j
        rsyms = sorted(rep)
        rmeta = bytes([len(rsyms) & 0xFF]) + bytes(rsyms) + bytes(runs)
        if len(rmeta) > 64:
            crm = _nx16_order0(rmeta, 4)
            meta += u7(len(rmeta) * 2) + u7(len(lits)) + u7(len(crm)) + crm
        else:
            meta += u7(len(rmeta) * 2 + 1) + u7(len(lits)) + rmeta
        cur = bytes(lits)
    N = 32 if x32 else 4
    if cat or not cur:
        flags |= 0x20
        body = cur
    elif order & 1 and len(cur) >= N:
        body = _nx16_order1(cur, N, shift, compress_table)
    else:
        flags &= ~1
        body = _nx16_order0(cur, N)
    return bytes([flags]) + (b"" if nosz else u7(len(data))) + bytes(meta) + body


T_TYPE, T_ALPHA, T_CHAR, T_DIGITS0, T_DZLEN, T_DUP, T_DIFF, T_DIGITS, T_DELTA, T_DELTA0, T_MATCH, T_NOP, T_END = range(13)


def _tokens(name):
    """a name cut into runs of digits (at most 9, so the value fits 32 bits) and the text between them"""
    out, i = [], 0
    while i < len(name):
        if 48 <= name[i] <= 57:
            j = i
            while j < len(name) and 48 <= name[j] <= 57 and j - i < 9:
                j += 1
            out.append(("d", name[i:j]))
        else:
            j = i
            while j < len(name) and not 48 <= name[j] <= 57:
                j += 1
            out.append(("a", name[i:j]))
        i = j
    return out


def tok3_encode(names, stream_kw=None, dup_streams=True):
    """the name tokeniser's layout (CRAMcodecs section 5): names (bytes, no NUL) -> block payload.  Every name is coded against
    the previous one: whole-name duplicates, per-position MATCH / DELTA / DELTA0, otherwise its own token."""
    stream_kw = stream_kw or {}
    S = {}                                         # (position, type) -> bytearray

    def put(pos, typ, b):
        S.setdefault((pos, typ), bytearray()).extend(b)

    prev_tok, prev_name = None, None
    for c, name in enumerate(names):
        if prev_name is not None and name == prev_name:
            put(0, T_TYPE, bytes([T_DUP]))
            put(0, T_DUP, struct.pack("<I", 1))
            continue
        put(0, T_TYPE, bytes([T_DIFF]))
        put(0, T_DIFF, struct.pack("<I", 1 if c else 0))
        toks = _tokens(name)
        for k, (kind, txt) in enumerate(toks):
            pos = k + 1
            pt = prev_tok[k] if prev_tok is not None and k < len(prev_tok) else None
            if pt is not None and pt == (kind, txt):
                put(pos, T_TYPE, bytes([T_MATCH]))
            elif kind == "d":
                v = int(txt)
                lead0 = len(txt) > 1 and txt[0] == 48
                p_is_d = pt is not None and pt[0] == "d"
                p_lead0 = p_is_d and len(pt[1]) > 1 and pt[1][0] == 48
                near = p_is_d and 0 <= v - int(pt[1]) < 256
                if near and lead0 and p_lead0 and len(pt[1]) == len(txt):
                    put(pos, T_TYPE, bytes([T_DELTA0])); put(pos, T_DELTA0, bytes([v - int(pt[1])]))
                elif near and not lead0 and not p_lead0:
                    put(pos, T_TYPE, bytes([T_DELTA])); put(pos, T_DELTA, bytes([v - int(pt[1])]))
                elif lead0:
                    put(pos, T_TYPE, bytes([T_DIGITS0])); put(pos, T_DIGITS0, struct.pack("<I", v)); put(pos, T_DZLEN, bytes([len(txt)]))
                else:
                    put(pos, T_TYPE, bytes([T_DIGITS])); put(pos, T_DIGITS, struct.pack("<I", v))
            elif len(txt) == 1:
                put(pos, T_TYPE, bytes([T_CHAR])); put(pos, T_CHAR, txt)
            else:
                put(pos, T_TYPE, bytes([T_ALPHA])); put(pos, T_ALPHA, txt + b"\0")
        put(len(toks) + 1, T_TYPE, bytes([T_END]))
        prev_tok, prev_name = toks, name
    ulen = sum(len(x) + 1 for x in names)
    out = bytearray(struct.pack("<IIB", ulen, len(names), 0))
    seen = {}
    npos = max(p for p, _ in S) + 1
    for pos in range(npos):
        first = True
        for typ in range(13):
            if (pos, typ) not in S:
                continue
            data = bytes(S[(pos, typ)])
            tt = typ | (0x80 if first else 0)
            first = False
            if dup_streams and data in seen and seen[data] != (pos, typ):
                out += bytes([tt | 0x40, seen[data][0], seen[data][1]])
                continue
            seen.setdefault(data, (pos, typ))
            comp = rans_nx16_encode(data, **stream_kw)
            out += bytes([tt]) + u7(len(comp)) + comp
    return bytes(out)


# ---- blocks, encodings ----------------------------------------------------------------------------------------------
RAW, GZIP, BZIP2, LZMA, RANS = 0, 1, 2, 3, 4
RANSNX16, TOK3 = 5, 8
FILE_HEADER, COMPRESSION_HEADER, SLICE_HEADER, EXTERNAL_DATA, CORE_DATA = 0, 1, 2, 4, 5


def block(method, content_type, content_id, raw):
    raw = bytes(raw)
    if method == GZIP:
        data = gzip.compress(raw, 6, mtime=0)
    elif method == BZIP2:
        import bz2
        data = bz2.compress(raw, 9)
    elif method == LZMA:
        import lzma
        data = lzma.compress(raw, format=lzma.FORMAT_XZ)
    elif method == (RANS, 0):
        data, method = rans_encode(raw, 0), RANS
    elif method == (RANS, 1):
        data, method = rans_encode(raw, 1), RANS
    elif isinstance(method, tuple) and method[0] == RANSNX16:
        data, method = rans_nx16_encode(raw, **method[1]), RANSNX16
    elif method == TOK3:
        names = raw.split(b"\0")[:-1] if raw else []
        data, method = (tok3_encode(names), TOK3) if names else (raw, RAW)
    else:
        data, method = raw, RAW
    b = bytes([method, content_type]) + itf8(content_id) + itf8(len(data)) + itf8(len(raw)) + data
    return b + struct.pack("<I", zlib.crc32(b) & 0xFFFFFFFF)


def enc_external(cid):
    return itf8(1) + itf8(len(itf8(cid))) + itf8(cid)


def enc_stop(stop, cid):
    p = bytes([stop]) + itf8(cid)
    return itf8(5) + itf8(len(p)) + p


def enc_len(len_enc, val_enc):
    p = len_enc + val_enc
    return itf8(4) + itf8(len(p)) + p


def enc_huffman(symbols, lengths):
    p = itf8(len(symbols)) + b"".join(itf8(s) for s in symbols) + itf8(len(lengths)) + b"".join(itf8(l) for l in lengths)
    return itf8(3) + itf8(len(p)) + p


def enc_beta(offset, nbits):
    p = itf8(offset) + itf8(nbits)
    return itf8(6) + itf8(len(p)) + p


def enc_gamma(offset):
    p = itf8(offset)
    return itf8(9) + itf8(len(p)) + p


def enc_subexp(offset, k):
    p = itf8(offset) + itf8(k)
    return itf8(7) + itf8(len(p)) + p


class Bits:
    """the core data block: most significant bit first"""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, nbits):
        for k in range(nbits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((v >> k) & 1)
            self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                self.acc, self.n = 0, 0

    def done(self):
        if self.n:
            self.out.append(self.acc << (8 - self.n))
            self.acc, self.n = 0, 0
        return bytes(self.out)


def canonical_codes(symbols, lengths):
    """CRAM / DEFLATE-style canonical codes: by (length, symbol value)"""
    order = sorted(range(len(symbols)), key=lambda i: (lengths[i], symbols[i]))
    codes, code, prev = {}, 0, lengths[order[0]]
    for i in order:
        code <<= lengths[i] - prev
        prev = lengths[i]
        codes[symbols[i]] = (code, lengths[i])
        code += 1
    return codes


def huffman_lengths(freq):
    """code lengths of a Huffman code over {symbol: count} (package-free: plain tree merge)"""
    import heapq
    if len(freq) == 1:
        return {next(iter(freq)): 0}
    h = [(c, i, (s,)) for i, (s, c) in enumerate(sorted(freq.items()))]
    heapq.heapify(h)
    depth = {s: 0 for s in freq}
    k = len(h)
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        for s in a[2] + b[2]:
            depth[s] += 1
        heapq.heappush(h, (a[0] + b[0], k, a[2] + b[2]))
        k += 1
    return depth


# content ids of the external blocks
CID = dict(BF=1, RL=3, AP=4, RN=6, MF=7, NS=8, NP=9, TS=10, NF=11, FC=14, FP=15, BS=17, IN=18, IN_LEN=19, SC=20, HC=21, PD=22, RS=23, BA=25, QS=26, RI=27)
METHODS = [RAW, GZIP, (RANS, 0), (RANS, 1)]
METHODS_31 = [(RANSNX16, dict(order=0)), (RANSNX16, dict(order=1)), (RANSNX16, dict(order=1, x32=True)), (RANSNX16, dict(order=0, pack=True, rle=True)),
              (RANSNX16, dict(order=0, stripe=4)), (RANSNX16, dict(order=1, compress_table=True, shift=10)), (RANSNX16, dict(cat=True))]
SUBST_ALT = {"A": "CGTN", "C": "AGTN", "G": "ACTN", "T": "ACGN", "N": "ACGT"}


def _ref_len(cig):
    return sum(int(c) >> 4 for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8))


def _seq(rec, i):
    so, L = int(rec.seq_off[i]), int(rec.l_seq[i])
    b = rec.seq4[so:so + (L + 1) // 2]
    s = np.empty(2 * b.size, np.uint8)
    s[0::2], s[1::2] = b >> 4, b & 15
    return "".join(NT16[int(x)] for x in s[:L])


def make_reference(rec, seed=1):
    """a reference the synthetic reads partly agree with: every position takes the base of the first mapped read that covers it
    with an M operation (other reads disagree there and get substitution features), the rest is random"""
    rng = np.random.default_rng(seed)
    refs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).astype(np.uint8) for _, ln in rec.targets]
    done = [np.zeros(ln, bool) for _, ln in rec.targets]
    for i in range(rec.n):
        t = int(rec.tid[i])
        if t < 0 or int(rec.flag[i]) & 4:
            continue
        s = _seq(rec, i)
        rp, qp = int(rec.pos[i]), 0
        for c in rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]:
            op, ln = int(c) & 15, int(c) >> 4
            if op in (0, 7, 8):
                for k in range(ln):
                    if 0 <= rp + k < refs[t].size and not done[t][rp + k] and s[qp + k] in "ACGT":
                        refs[t][rp + k] = ord(s[qp + k])
                        done[t][rp + k] = True
                rp += ln; qp += ln
            elif op in (1, 4):
                qp += ln
            elif op in (2, 3):
                rp += ln
    return [r.tobytes() for r in refs]


def reads_from_reference(rec, refs, mismatch=0.004, seed=1):
    """rewrites the bases of `rec` (in place) so that its mapped reads READ LIKE A SEQUENCER'S against `refs`: every base under an
    M / = / X operation is the reference's, except a fraction `mismatch` of them; clipped and inserted bases stay what they were.
    What a real CRAM holds -- a read feature or two per record, not a hundred -- for decode-rate measurements."""
    rng = np.random.default_rng(seed)
    code = np.zeros(256, np.uint8)
    for k, ch in enumerate(b"=ACMGRSVTWYHKDBN"):
        code[ch] = k
    acgt = np.frombuffer(b"ACGT", np.uint8)
    refa = [np.frombuffer(r, np.uint8) for r in refs]
    for i in range(rec.n):
        t = int(rec.tid[i])
        if t < 0 or int(rec.flag[i]) & 4:
            continue
        L, so = int(rec.l_seq[i]), int(rec.seq_off[i])
        nib = np.empty(L + (L & 1), np.uint8)
        packed = rec.seq4[so:so + (L + 1) // 2]
        nib[0::2] = packed >> 4
        nib[1::2] = packed & 15
        rp, qp = int(rec.pos[i]), 0
        for c in rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]:
            op, ln = int(c) & 15, int(c) >> 4
            if op in (0, 7, 8):
                a, b = max(rp, 0), min(rp + ln, refa[t].size)
                if b > a:
                    seg = refa[t][a:b].copy()
                    mm = np.nonzero(rng.random(b - a) < mismatch)[0]
                    if mm.size:
                        seg[mm] = acgt[(np.searchsorted(acgt, seg[mm]) + 1 + rng.integers(0, 3, mm.size)) % 4]
                    nib[qp + (a - rp):qp + (b - rp)] = code[seg]
                rp += ln; qp += ln
            elif op in (1, 4):
                qp += ln
            elif op in (2, 3):
                rp += ln
        rec.seq4[so:so + (L + 1) // 2] = (nib[0::2] << 4) | nib[1::2]
        if L & 1:
            rec.seq4[so + L // 2] &= 0xF0
    return rec


def write_fasta(path, targets, refs, width=60):
    with open(path, "wb") as f, open(path + ".fai", "w") as fai:
        for (name, ln), seq in zip(targets, refs):
            f.write(b">" + name.encode() + b"\n")
            off = f.tell()
            for o in range(0, len(seq), width):
                f.write(seq[o:o + width] + b"\n")
            fai.write(f"{name}\t{ln}\t{off}\t{width}\t{width + 1}\n")


def _pair_tlen(a, b):
    """template length of two records of one slice as the reader derives it: span from the leftmost start to the rightmost end;
    positive for the record that starts first (the one flagged first-in-pair when they start together)"""
    left, right = min(a["pos"], b["pos"]), max(a["end"], b["end"])
    t = right - left + 1
    if a["pos"] < b["pos"] or (a["pos"] == b["pos"] and a["flag"] & 0x40):
        return t, -t
    return -t, t


def _chain_fields(xs):
    """what the reader derives for the records of ONE template linked into a chain inside a slice (htslib, cram_decode_slice_xref):
    every record's mate is the next of the chain, the last one's the first; the template length spans the leftmost start to the
    rightmost end of all of them, positive for a leftmost record (several leftmost: the first-in-pair one), zero when the chain
    touches two references or the record or its mate is unmapped.  -> [(mtid, mpos, tlen, mate reverse, mate unmapped)]"""
    left = min(x["pos"] for x in xs)
    left_cnt = sum(1 for x in xs if x["pos"] == left)
    right = max(x["end"] for x in xs)
    one_ref = len({x["tid"] for x in xs}) == 1
    t = right - left + 1
    out = []
    for j, x in enumerate(xs):
        m = xs[(j + 1) % len(xs)]
        tl = 0 if not one_ref else (t if x["pos"] == left and (left_cnt == 1 or x["flag"] & 0x40) else -t)
        if (m["flag"] & 4) or (x["flag"] & 4):
            tl = 0
        out.append((m["tid"], m["pos"], tl, bool(m["flag"] & 0x10), bool(m["flag"] & 0x4)))
    return out


def write_cram(path, rec, refs, header_text=None, records_per_slice=300, slices_per_container=2, read_names=True, index=True, ap_delta=True,
               multi_ref=False, qualities=False, tags=False, slice_md5=True, stats=None, repeat=1, version=(3, 0), block_methods=None, embed_ref=False):
    """records of `rec` (coordinate sorted, unmapped tail last) -> CRAM 3.0 + .crai.  refs[tid] = reference bytes (ACGTN).
    multi_ref: slices run across reference boundaries (slice reference id -2, RI per record); qualities: every record carries its
    quality array (CF bit 1, QS per base); tags: every record carries NM:C and MD:Z (tag dictionary + tag encoding map: values
    the reader must walk past)."""
    from .bamio import sam_header
    text = (header_text if header_text is not None else sam_header(rec.targets)).encode()
    out = bytearray(b"CRAM" + bytes(version) + b"strling-test".ljust(20, b"\0"))
    # version 3.1: the external blocks rotate through the rANS Nx16 variants too, the read names go through the name tokeniser
    methods = METHODS if tuple(version) == (3, 0) else METHODS + METHODS_31
    if block_methods:                   # (tests: the methods the external blocks rotate through, e.g. bzip2 / lzma)
        methods = list(block_methods)

    def container(ref_id, start, span, n_rec, counter, bases, blocks, landmarks):
        body = b"".join(blocks)
        h = itf8(ref_id) + itf8(start) + itf8(span) + itf8(n_rec) + ltf8(counter) + ltf8(bases) + itf8(len(blocks)) + itf8(len(landmarks)) + \
            b"".join(itf8(x) for x in landmarks)
        h = struct.pack("<i", len(body)) + h
        return h + struct.pack("<I", zlib.crc32(h) & 0xFFFFFFFF) + body

    out += container(0, 0, 0, 0, 0, 0, [block(RAW, FILE_HEADER, 0, struct.pack("<i", len(text)) + text)], [0])

    data_at = len(out)
    # slices: runs of records on one reference (unmapped tail: -1)
    slices, i = [], 0
    while i < rec.n:
        t = int(rec.tid[i])
        j = i
        while j < rec.n and j - i < records_per_slice and (multi_ref or int(rec.tid[j]) == t):
            j += 1
        slices.append((i, j))
        i = j
    crai, counter, method_rot = [], 0, 0
    for c0 in range(0, len(slices), slices_per_container):
        group = slices[c0:c0 + slices_per_container]
        # ---- records of the container as dictionaries ----
        recs = []
        for (a, b) in group:
            rs = []
            for i in range(a, b):
                cig = rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]
                unm = bool(int(rec.flag[i]) & 4)
                pos1 = int(rec.pos[i]) + 1 if int(rec.tid[i]) >= 0 else 0
                rs.append(dict(i=i, tid=int(rec.tid[i]), pos=pos1, end=pos1 if unm else pos1 + max(_ref_len(cig), 1) - 1,
                               flag=int(rec.flag[i]), mapq=int(rec.mapq[i]), L=int(rec.l_seq[i]), cig=cig, seq=_seq(rec, i), name=rec.qname(i),
                               mtid=int(rec.mtid[i]), mpos=int(rec.mpos[i]) + 1 if int(rec.mtid[i]) >= 0 else 0,
                               tlen=int(rec.isize[i]) if rec.isize is not None else 0, cf=0, nf=0))
            # pairs inside the slice whose mate fields are what the reader would derive: "mate downstream"; everything else detached
            by_name = {}
            for k, r in enumerate(rs):
                by_name.setdefault(r["name"], []).append(k)
            for r in rs:
                r["cf"] = (2 if r["flag"] & 1 else 0) | (1 if qualities else 0)
            for ks in by_name.values():
                if len(ks) < 2:
                    continue
                xs = [rs[k] for k in ks]
                if not all(x["flag"] & 1 for x in xs) or xs[0]["tid"] < 0:
                    continue
                if len(ks) == 2 and (xs[0]["flag"] | xs[1]["flag"]) & 0x900:
                    continue
                ok = all((x["mtid"], x["mpos"], x["tlen"], bool(x["flag"] & 0x20), bool(x["flag"] & 0x8)) == d for x, d in zip(xs, _chain_fields(xs)))
                if ok:
                    if stats is not None:
                        stats[len(xs)] = stats.get(len(xs), 0) + 1       # chains written, by length
                    q1 = 1 if qualities else 0
                    for j, x in enumerate(xs):
                        if j + 1 < len(xs):
                            x["cf"], x["nf"] = 4 | q1, ks[j + 1] - ks[j] - 1
                        else:
                            x["cf"] = q1
            recs.append(rs)
        # ---- compression header ----
        cf_freq = {}
        for rs in recs:
            for r in rs:
                cf_freq[r["cf"]] = cf_freq.get(r["cf"], 0) + 1
        cf_len = huffman_lengths(cf_freq)
        cf_syms = sorted(cf_len)
        cf_codes = canonical_codes(cf_syms, [cf_len[s] for s in cf_syms]) if len(cf_syms) > 1 else {cf_syms[0]: (0, 0)}
        pres = b"RN" + bytes([1 if read_names else 0]) + b"AP" + bytes([1 if ap_delta else 0]) + b"RR" + bytes([1]) + b"SM" + bytes([0x1B] * 5) + \
            b"TD" + (itf8(7) + b"NMCMDZ\0" if tags else itf8(1) + b"\0")
        pres = itf8(5) + pres
        ds = {
            "BF": enc_external(CID["BF"]), "CF": enc_huffman(cf_syms, [cf_len[s] for s in cf_syms]), "RI": enc_external(CID["RI"]),
            "RL": enc_external(CID["RL"]), "AP": enc_external(CID["AP"]), "RG": enc_huffman([-1], [0]),
            "RN": enc_stop(0, CID["RN"]), "MF": enc_external(CID["MF"]), "NS": enc_external(CID["NS"]), "NP": enc_external(CID["NP"]),
            "TS": enc_external(CID["TS"]), "NF": enc_external(CID["NF"]), "TL": enc_huffman([0], [0]), "FN": enc_gamma(1),
            "FC": enc_external(CID["FC"]), "FP": enc_external(CID["FP"]), "DL": enc_subexp(0, 2), "BS": enc_external(CID["BS"]),
            "IN": enc_len(enc_external(CID["IN_LEN"]), enc_external(CID["IN"])), "SC": enc_stop(0, CID["SC"]), "HC": enc_external(CID["HC"]),
            "PD": enc_external(CID["PD"]), "RS": enc_external(CID["RS"]), "MQ": enc_beta(0, 8), "BA": enc_external(CID["BA"]), "QS": enc_external(CID["QS"]),
        }
        dsm = itf8(len(ds)) + b"".join(k.encode() + v for k, v in ds.items())
        NM_KEY, MD_KEY = (ord("N") << 16) | (ord("M") << 8) | ord("C"), (ord("M") << 16) | (ord("D") << 8) | ord("Z")
        tagm = itf8(2) + itf8(NM_KEY) + enc_len(enc_huffman([1], [0]), enc_external(40)) + itf8(MD_KEY) + enc_stop(9, 41) if tags else itf8(0)
        comp_hdr = block(GZIP if c0 % 2 else RAW, COMPRESSION_HEADER, 0, itf8(len(pres)) + pres + itf8(len(dsm)) + dsm + itf8(len(tagm)) + tagm)
        # ---- slices ----
        blocks, landmarks, at = [comp_hdr], [], len(comp_hdr)
        n_rec_c, bases_c = 0, 0
        c_start, c_end = None, 0
        slice_meta = []
        for (a, b), rs in zip(group, recs):
            ext = {k: bytearray() for k in set(CID.values()) | {40, 41}}
            bits = Bits()
            tid = rs[0]["tid"] if len({r["tid"] for r in rs}) == 1 else -2
            mapped = [r for r in rs if r["tid"] >= 0]
            s_start = min((r["pos"] for r in mapped), default=0) if tid != -2 else 0
            s_end = max((max(r["end"], r["pos"]) for r in mapped), default=0) if tid != -2 else 0
            prev_pos = s_start
            for r in rs:
                ext[CID["BF"]] += itf8(r["flag"])
                code, nb = cf_codes[r["cf"]]
                bits.put(code, nb)
                if tid == -2:
                    ext[CID["RI"]] += itf8(r["tid"])
                ext[CID["RL"]] += itf8(r["L"])
                ext[CID["AP"]] += itf8(r["pos"] - prev_pos if ap_delta else r["pos"])
                prev_pos = r["pos"] if ap_delta else prev_pos
                if read_names:
                    ext[CID["RN"]] += r["name"] + b"\0"
                if r["cf"] & 2:
                    mf = (1 if r["flag"] & 0x20 else 0) | (2 if r["flag"] & 0x8 else 0)
                    ext[CID["MF"]] += itf8(mf)
                    if not read_names:
                        ext[CID["RN"]] += r["name"] + b"\0"
                    ext[CID["NS"]] += itf8(r["mtid"])
                    ext[CID["NP"]] += itf8(r["mpos"])
                    ext[CID["TS"]] += itf8(r["tlen"])
                elif r["cf"] & 4:
                    ext[CID["NF"]] += itf8(r["nf"])
                # (TL: a one-symbol code, no bits: every record uses tag line 0)
                if tags:
                    ext[40].append(r["L"] & 0x7F)                       # NM:C, one byte (its length through a one-symbol HUFFMAN code)
                    ext[41] += str(r["L"]).encode() + b"\t"             # MD:Z ... stop byte 9
                if not r["flag"] & 4:
                    feats = []
                    ref = refs[r["tid"]]
                    rp, qp = r["pos"] - 1, 0
                    for c in r["cig"]:
                        op, ln = int(c) & 15, int(c) >> 4
                        if op in (0, 7, 8):
                            for k in range(ln):
                                rb = chr(ref[rp + k]).upper() if 0 <= rp + k < len(ref) else "N"
                                qb = r["seq"][qp + k]
                                if qb != rb:
                                    if qb in "ACGTN" and rb in "ACGTN" and qb in SUBST_ALT[rb]:
                                        feats.append((qp + k + 1, "X", SUBST_ALT[rb].index(qb)))
                                    else:
                                        feats.append((qp + k + 1, "B", qb))
                            rp += ln; qp += ln
                        elif op == 1:
                            feats.append((qp + 1, "I", r["seq"][qp:qp + ln])); qp += ln
                        elif op == 4:
                            feats.append((qp + 1, "S", r["seq"][qp:qp + ln])); qp += ln
                        elif op == 2:
                            feats.append((qp + 1, "D", ln)); rp += ln
                        elif op == 3:
                            feats.append((qp + 1, "N", ln)); rp += ln
                        elif op == 5:
                            feats.append((qp + 1, "H", ln))
                        elif op == 6:
                            feats.append((qp + 1, "P", ln))
                    # FN: gamma, offset 1
                    v = len(feats) + 1
                    nbv = v.bit_length()
                    bits.put(0, nbv - 1); bits.put(v, nbv)
                    last = 0
                    for fp, code, val in feats:
                        ext[CID["FC"]].append(ord(code))
                        ext[CID["FP"]] += itf8(fp - last)
                        last = fp
                        if code == "X":
                            ext[CID["BS"]].append(val)
                        elif code == "B":
                            ext[CID["BA"]].append(ord(val)); ext[CID["QS"]].append(0xFF)
                        elif code == "I":
                            ext[CID["IN_LEN"]] += itf8(len(val)); ext[CID["IN"]] += val.encode()
                        elif code == "S":
                            ext[CID["SC"]] += val.encode() + b"\0"
                        elif code == "D":                     # SUBEXP, offset 0, k = 2
                            v = val
                            if v < 4:
                                bits.put(0, 1); bits.put(v, 2)
                            else:
                                b_ = v.bit_length() - 1
                                u = b_ - 2 + 1
                                bits.put((1 << u) - 1, u); bits.put(0, 1); bits.put(v & ((1 << b_) - 1), b_)
                        elif code == "N":
                            ext[CID["RS"]] += itf8(val)
                        elif code == "H":
                            ext[CID["HC"]] += itf8(val)
                        elif code == "P":
                            ext[CID["PD"]] += itf8(val)
                    bits.put(r["mapq"], 8)                    # MQ: BETA, 8 bits
                else:
                    ext[CID["BA"]] += r["seq"].encode()
                if r["cf"] & 1:
                    ext[CID["QS"]] += bytes((7 * k + r["L"]) % 41 for k in range(r["L"]))
                bases_c += r["L"]
            used = sorted(k for k, v in ext.items() if v)
            eblocks = []
            for k in used:
                m = methods[method_rot % len(methods)]
                if tuple(version) != (3, 0) and k == CID["RN"] and read_names:
                    m = TOK3
                eblocks.append(block(m, EXTERNAL_DATA, k, ext[k]))
                method_rot += 1
            core = block(RAW, CORE_DATA, 0, bits.done())
            span = s_end - s_start + 1 if tid >= 0 and s_start else 0
            if tid == -2:
                mp = [r for r in rs if r["tid"] >= 0]          # the index lists a multi-reference slice once per reference it holds
                multi_lines = []
                for t2 in sorted({r["tid"] for r in mp}):
                    rr = [r for r in mp if r["tid"] == t2]
                    a2, b2 = min(r["pos"] for r in rr), max(max(r["end"], r["pos"]) for r in rr)
                    multi_lines.append((t2, a2, b2 - a2 + 1))
            # the MD5 of the reference bases the slice spans (CRAMv3 section 8.5); multi-reference and unmapped slices carry zeros
            md5 = hashlib.md5(bytes(refs[tid][s_start - 1:s_start - 1 + span]).upper()).digest() if tid >= 0 and span and slice_md5 else bytes(16)
            emb_id = -1
            if embed_ref and tid >= 0 and span:      # samtools' embed_ref: the bases the slice spans travel in an external block of their own
                emb_id = 250
                eblocks.append(block(GZIP, EXTERNAL_DATA, emb_id, bytes(refs[tid][s_start - 1:s_start - 1 + span])))
                used = used + [emb_id]
            sh = itf8(tid) + itf8(s_start if tid >= 0 else 0) + itf8(span) + itf8(len(rs)) + ltf8(counter) + itf8(1 + len(eblocks)) + \
                itf8(len(used)) + b"".join(itf8(k) for k in used) + itf8(emb_id) + md5
            shb = block(RAW, SLICE_HEADER, 0, sh)
            landmarks.append(at)
            sl_bytes = shb + core + b"".join(eblocks)
            if tid == -2:
                for t2, a2, sp2 in multi_lines:
                    slice_meta.append((t2, a2, sp2, at, len(sl_bytes)))
            else:
                slice_meta.append((tid, s_start if tid >= 0 else 0, span, at, len(sl_bytes)))
            blocks += [shb, core] + eblocks
            at += len(sl_bytes)
            counter += len(rs)
            n_rec_c += len(rs)
            if tid >= 0:
                c_start = s_start if c_start is None else min(c_start, s_start)
                c_end = max(c_end, s_end)
        tids = {m[0] for m in slice_meta}
        c_tid = tids.pop() if len(tids) == 1 and not multi_ref else -2
        coff = len(out)
        out += container(c_tid, (c_start or 0) if c_tid >= 0 else 0, (c_end - c_start + 1) if c_tid >= 0 and c_start else 0, n_rec_c, counter - n_rec_c, bases_c,
                         blocks, landmarks)
        for tid, st, sp, lm, sz in slice_meta:
            crai.append(f"{tid}\t{st}\t{sp}\t{coff}\t{lm}\t{sz}\n")
    if repeat > 1:                      # (benchmarks only: the data containers again and again; such a file is not sorted and gets no index)
        body = bytes(out[data_at:])
        index = False
        with open(path, "wb") as f:
            f.write(out[:data_at])
            for _ in range(repeat):
                f.write(body)
            f.write(EOF_V3)
        return text.decode()
    out += EOF_V3
    with open(path, "wb") as f:
        f.write(out)
    if index:
        with open(path + ".crai", "wb") as f:
            f.write(gzip.compress("".join(crai).encode(), 6, mtime=0))
    return text.decode()
