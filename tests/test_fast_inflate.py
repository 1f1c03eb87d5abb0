"""The host front end's DEFLATE decoder (strling_amd/csrc/cli/fast_inflate.cpp) against zlib: every block type, codes that
need subtables, two-literal table entries, matches at every distance class, sizes around the fast loop's margins, and
corrupted / truncated streams -- which it must either refuse (the reader then asks zlib) or decode exactly as zlib does,
without ever writing outside the output buffer.  CPU only."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "strling_amd", "csrc", "cli", "fast_inflate.cpp")


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(HERE, "emu", "libfast_inflate_test.so")
    shim = os.path.join(HERE, "emu", "fast_inflate_shim.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(SRC), os.path.getmtime(shim)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.dirname(SRC), "-o", so, shim, SRC])
    L = C.CDLL(so)
    L.fi_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.fi_inflate.restype = C.c_int
    return L


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for o in range(0, len(data), flush_every):
        out += c.compress(data[o:o + flush_every]) + c.flush(zlib.Z_FULL_FLUSH if (o // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + c.flush()


def run(lib, comp, n_out, guard=64):
    """-> (rc, bytes) with guard bytes either side of the output checked untouched; 8 readable bytes follow the stream"""
    buf = np.full(n_out + 2 * guard, 0xCD, np.uint8)
    rc = lib.fi_inflate(comp + b"\xAB" * 8, len(comp), buf.ctypes.data + guard, n_out)
    assert (buf[:guard] == 0xCD).all() and (buf[guard + n_out:] == 0xCD).all(), "wrote outside the output buffer"
    return rc, buf[guard:guard + n_out].tobytes()


def corpus(rng):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    yield b""
    yield b"A"
    yield bytes(rng.choice(acgt, 65280))
    yield bytes(rng.integers(0, 256, 65280, dtype=np.uint8))                       # incompressible: stored blocks
    yield bytes(rng.integers(0, 16, 40000, dtype=np.uint8) * 17)                   # SEQ-like nibble pairs: short codes, two-literal entries
    yield b"\xff" * 70000                                                           # distance-1 runs of the longest matches
    yield bytes(np.arange(50000, dtype=np.uint32).view(np.uint8)[:60000])
    yield bytes(np.where(rng.random(65000) < 0.9, 65, rng.integers(0, 256, 65000)).astype(np.uint8))
    skew = np.minimum(rng.geometric(0.03, 65000), 255).astype(np.uint8)             # ~200 distinct literals, long codes: subtables
    yield bytes(skew)
    for n in (1, 2, 7, 255, 256, 257, 280, 289, 290, 291, 300, 600, 4096):          # around the fast loop's output margin
        yield bytes(rng.integers(0, 4, n, dtype=np.uint8) + 65)
    rec = b"".join(b"q%d\x00" % i + bytes(rng.integers(0, 16, 75, dtype=np.uint8) * 17) + b"\xff" * 150 for i in range(230))
    yield rec                                                                       # BAM-record-like


def test_identical_to_zlib_for_every_block_type(lib):
    rng = np.random.default_rng(5)
    n = 0
    for data in corpus(rng):
        for level in (0, 1, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for fe in (0, 5000):
                    comp = deflate(data, level, strategy, fe)
                    rc, out = run(lib, comp, len(data))
                    assert rc == 0 and out == data, (len(data), level, strategy, fe)
                    n += 1
    assert n > 800


def test_wrong_sizes_and_corrupt_streams_never_pass_for_something_else(lib):
    rng = np.random.default_rng(6)
    refused = agreed = 0
    for data in corpus(rng):
        if len(data) < 8:
            continue
        comp = deflate(data, 6)
        assert run(lib, comp, len(data) + 1)[0] != 0 and run(lib, comp, len(data) - 1)[0] != 0      # ISIZE must match exactly
        for k in range(24):
            bad = bytearray(comp)
            if k % 3 == 2:
                bad = bad[: int(rng.integers(1, len(bad)))]                                          # truncated
            else:
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))                  # one flipped bit
            rc, out = run(lib, bytes(bad), len(data))
            if rc != 0:
                refused += 1
                continue
            d = zlib.decompressobj(-15)                 # accepted: zlib must produce the same bytes from the same stream
            try:
                z = d.decompress(bytes(bad)) + d.flush()
            except zlib.error:
                z = None
            assert z == out and d.eof, k
            agreed += 1
    assert refused > 200
