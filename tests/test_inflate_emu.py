"""The device DEFLATE decoder's core (strling_amd/csrc/inflate_wave.h: one wave per stream) compiled for the host, its 64
lanes as loops, against zlib: every block type, every input alignment, multi-block streams, corrupt and truncated streams
(the bytes behind the readable range and around the output are poisoned and checked).  CPU only -- the GPU run of the same
vectors is test_bgzf_device.py."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


class _Form:
    """one decoder form of the shim: `wave` = inflate_wave.h (a wave per stream), `group` = inflate_group.h (G lanes per stream; G = 1 here)"""
    def __init__(self, lib, prefix):
        self.emu_inflate = getattr(lib, prefix + "inflate")
        self.emu_inflate_at = getattr(lib, prefix + "inflate_at")
        self.emu_inflate.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
        self.emu_inflate_at.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32]


def _lib():
    so = os.path.join(HERE, "emu", "libinflate_emu.so")
    src = os.path.join(HERE, "emu", "inflate_emu.cpp")
    cores = [os.path.join(HERE, "..", "strling_amd", "csrc", f) for f in ("inflate_wave.h", "inflate_group.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + cores):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return C.CDLL(so)


@pytest.fixture(scope="module", params=["wave", "group"])
def emu(request):
    return _Form(_lib(), "emu_" if request.param == "wave" else "emu_group_")


def test_group_form_lds_fits_twelve_waves_per_cu():
    """inflate_group.h's budget: a wave of eight 8-lane groups in 1/12 of a CU's 160 KB"""
    L = _lib()
    assert L.emu_group_lds_bytes(8) * 8 * 12 <= 160 * 1024, L.emu_group_lds_bytes(8)


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for o in range(0, len(data), flush_every):
        out += c.compress(data[o:o + flush_every]) + c.flush(zlib.Z_FULL_FLUSH if (o // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + c.flush()


def corpus(rng):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    yield b""
    yield b"A"
    yield bytes(rng.choice(acgt, 65280))                                   # a full BGZF block of sequence-like text
    yield bytes(rng.integers(0, 256, 65280, dtype=np.uint8))               # incompressible
    yield b"\xff" * 65280                                                  # one long run (distance-1 matches of length 258)
    yield (b"CAG" * 30000)[:65000]
    yield (b"ACGTTGCAAT" * 7000)[:65000]                                   # period 10: the 4-bytes-at-a-time ring path
    yield bytes(rng.integers(0, 4, 30000, dtype=np.uint8)) + b"\0" * 20000 + bytes(rng.choice(acgt, 15000))
    rec = bytearray()
    for i in range(230):                                                   # BAM-record-like: binary header + name + packed seq + 0xff quals
        rec += bytes(rng.integers(0, 256, 36, dtype=np.uint8)) + b"q%d\0" % (i * 7919) + bytes(rng.integers(0, 256, 75, dtype=np.uint8)) + b"\xff" * 150
    yield bytes(rec)
    far = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))                # matches further back than the 1 KiB ring
    yield far + bytes(rng.integers(0, 256, 2000, dtype=np.uint8)) + far + far[:1500] + bytes(rng.integers(0, 256, 700, dtype=np.uint8)) + far


VARIANTS = (dict(level=1), dict(level=6), dict(level=9), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED),
            dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=6, strategy=zlib.Z_RLE), dict(level=4, flush_every=5000))


def test_inflate_core_matches_zlib(emu):
    rng = np.random.default_rng(3)
    n = 0
    for data in corpus(rng):
        for kw in VARIANTS:
            s = deflate(data, **kw)
            out = C.create_string_buffer(len(data) + 1)
            rc = emu.emu_inflate(s, len(s), out, len(data))
            assert rc == 0 and out.raw[:len(data)] == data, (n, kw, rc)
            n += 1
    assert n >= 80


def test_inflate_core_every_alignment_and_length(emu):
    """streams of every length mod 64 at every input alignment: the 256-byte input window, the partial first dword and the
    64-byte literal flush all have their edges here"""
    rng = np.random.default_rng(8)
    for n in list(range(1, 140)) + list(range(1000, 1070)) + list(range(37850, 37900)):
        data = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), n))
        s = deflate(data, level=int(rng.integers(1, 10)))
        out = C.create_string_buffer(n + 1)
        assert emu.emu_inflate_at(s, len(s), n % 7, out, n) == 0 and out.raw[:n] == data, n
    data = bytes(rng.integers(0, 6, 3000, dtype=np.uint8)) * 3
    for lvl in (0, 1, 6):
        s = deflate(data, level=lvl, flush_every=700 if lvl else 0)
        for lead in range(0, 9):
            out = C.create_string_buffer(len(data) + 1)
            assert emu.emu_inflate_at(s, len(s), lead, out, len(data)) == 0 and out.raw[:len(data)] == data, (lvl, lead)


def test_inflate_core_long_codes_and_far_matches(emu):
    """skewed symbol statistics give 11..15-bit literal/length codes and 9..15-bit distance codes (the canonical search behind
    the first-level tables); 32 KiB distances; length-258 matches at every distance below 70 (the periodic copy)"""
    rng = np.random.default_rng(11)
    p = 0.5 ** np.arange(1, 257)
    skew = bytes(rng.choice(256, 60000, p=p / p.sum()).astype(np.uint8))
    far = bytes(rng.integers(0, 256, 400, dtype=np.uint8))
    blocks = [skew, far + bytes(32768 - 400) + far + skew[:3000] + far]
    for d in range(1, 70):
        unit = bytes(rng.integers(0, 256, d, dtype=np.uint8))
        blocks.append((unit * (900 // d + 2))[:900] + bytes(rng.integers(0, 256, 30, dtype=np.uint8)))
    for data in blocks:
        for lvl in (1, 6, 9):
            s = deflate(data, level=lvl)
            out = C.create_string_buffer(len(data) + 1)
            assert emu.emu_inflate(s, len(s), out, len(data)) == 0 and out.raw[:len(data)] == data, (len(data), lvl)


def long_then_short_literals(rng, n=65000):
    """bytes whose Huffman code has a chain of 10..15-bit literals (eighteen values occurring once, twice, four times, ...) beside 9-bit ones
    (216 values, ~150 times each) and 4-bit ones; every rare value stands in front of 9-bit ones.  Compressed with
    Z_HUFFMAN_ONLY every symbol is a literal: the bit budget of a decoder that looks several literals up per refill check
    (15 + 9 bits gone, the third lookup's nine no longer there)"""
    vals = rng.permutation(256).astype(np.uint8)
    tail = [1] * 8 + [2] * 4 + [4] * 2 + [8, 16, 32, 64]           # (the eight singletons end up seven levels below the 9-bit values: 15 bits)
    counts = [(n - 216 * 150 - sum(tail)) // 8] * 8 + [150] * 216 + tail
    counts[0] += n - sum(counts)
    data = np.repeat(vals[:len(counts)], counts)
    rng.shuffle(data)
    mid = vals[8:224]
    for t in np.nonzero(np.isin(data, vals[224:224 + len(tail)]))[0]:
        if t + 3 < n:
            data[t + 1:t + 3] = rng.choice(mid, 2)
    return bytes(data)


def test_inflate_core_long_literal_then_short_ones(emu):
    rng = np.random.default_rng(15)
    for k in range(6):
        data = long_then_short_literals(rng)
        for kw in (dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=6), dict(level=1)):
            s = deflate(data, **kw)
            out = C.create_string_buffer(len(data) + 1)
            assert emu.emu_inflate(s, len(s), out, len(data)) == 0 and out.raw[:len(data)] == data, (k, kw)


def _zlib_ok(stream, n):
    try:
        d = zlib.decompressobj(-15)
        o = d.decompress(stream)
        return d.eof and len(o) == n and not d.unused_data
    except zlib.error:
        return False


def test_inflate_core_rejects_bad_streams(emu):
    rng = np.random.default_rng(5)
    data = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 20000))
    s = deflate(data)
    out = C.create_string_buffer(len(data) + 64)
    assert emu.emu_inflate(s, len(s), out, len(data) + 1) != 0                # ISIZE too large
    assert emu.emu_inflate(s, len(s), out, len(data) - 1) != 0                # ISIZE too small
    assert emu.emu_inflate(s[:len(s) // 2], len(s) // 2, out, len(data)) != 0  # truncated stream
    bad = 0
    for k in range(40):
        g = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
        bad += emu.emu_inflate(g, len(g), out, 20000) != 0
    assert bad == 40                                                          # garbage never decodes to exactly 20000 bytes


@pytest.mark.timeout(300)
def test_inflate_core_corrupt_streams_get_zlibs_verdict(emu):
    """bit flips and truncations of valid streams (dynamic, fixed and stored blocks): the decoder stays inside its buffers
    (the shim checks the poison on both sides) and accepts a stream only if zlib inflates it to exactly ISIZE bytes with the
    same content"""
    rng = np.random.default_rng(21)
    base = bytes(rng.choice(np.frombuffer(b"ACGTN\xff", np.uint8), 9000))
    n_acc = 0
    for kw in (dict(level=6), dict(level=1), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED), dict(level=4, flush_every=1500)):
        s = deflate(base, **kw)
        for t in range(160):
            b = bytearray(s)
            if t % 4 == 3:
                b = b[:int(rng.integers(1, len(b)))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    i = int(rng.integers(0, min(len(b), 200) if t % 2 else len(b)))
                    b[i] ^= 1 << int(rng.integers(0, 8))
            b = bytes(b)
            out = C.create_string_buffer(len(base) + 1)
            rc = emu.emu_inflate_at(b, len(b), t % 5, out, len(base))
            assert rc in (0, 1, 2, 3), rc
            if rc == 0:
                n_acc += 1
                assert _zlib_ok(b, len(base)) and out.raw[:len(base)] == zlib.decompress(b, -15), (kw, t)
    assert n_acc < 400


@pytest.mark.timeout(600)
def test_inflate_core_random_shapes_valid_and_damaged(emu):
    """the device fuzzer's block shapes (tests/fuzz/fuzz_inflate.py: geometric alphabets, long-then-short literals, runs, periodic
    text, BAM-like records, far matches) x level / strategy / flush points through the host build -- which takes the device loop's
    steps --, every third stream also bit-flipped or truncated: the decoder returns (a decoder that does not is a hung GPU), keeps
    inside its buffers, and accepts only what zlib inflates to the same bytes"""
    import importlib.util, types
    src = open(os.path.join(HERE, "fuzz", "fuzz_inflate.py")).read().replace("from strling_amd import api\n", "")
    fz = types.ModuleType("fuzz_inflate_shapes")
    fz.__dict__["__file__"] = os.path.join(HERE, "fuzz", "fuzz_inflate.py")
    exec(compile(src, "fuzz_inflate_shapes", "exec"), fz.__dict__)
    rng = np.random.default_rng(77)
    for n in range(700):
        b = fz.block(rng)
        kw = dict(level=int(rng.integers(0, 10)))
        k = int(rng.integers(0, 6))
        if k == 1: kw["strategy"] = zlib.Z_HUFFMAN_ONLY
        elif k == 2: kw["strategy"] = zlib.Z_RLE
        elif k == 3: kw["strategy"] = zlib.Z_FIXED
        elif k == 4: kw["flush_every"] = int(rng.integers(200, 20000))
        st = deflate(b, **kw)
        out = C.create_string_buffer(len(b) + 1)
        assert emu.emu_inflate_at(st, len(st), n % 8, out, len(b)) == 0 and out.raw[:len(b)] == b, (n, kw)
        if n % 3 == 0 and len(st) > 8:
            c = bytearray(st)
            for _ in range(int(rng.integers(1, 4))):
                c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
            if rng.random() < 0.3:
                c = c[:int(rng.integers(1, len(c)))]
            c = bytes(c)
            rc = emu.emu_inflate_at(c, len(c), 0, out, len(b))
            assert rc in (0, 1, 2, 3), rc
            if rc == 0:
                assert _zlib_ok(c, len(b)) and out.raw[:len(b)] == zlib.decompress(c, -15), n
