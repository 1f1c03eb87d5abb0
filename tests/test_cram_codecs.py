"""CRAM 3.1 block codecs (cli/cram_codecs.cpp: rANS Nx16, the name tokeniser) against the test encoders of strling_amd/cramio.py,
both written from the CRAM codecs specification.  No htscodecs in this image: parity against htslib-written streams stays
unpinned (verify/run_reference.sh has the samtools round trip for whoever has it)."""
import os
import subprocess

import numpy as np
import pytest

from strling_amd import build, cramio

CLI = build.CLI


def _decode(kind, payload, expect, tmp_path):
    a, b = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    open(a, "wb").write(payload)
    r = subprocess.run([CLI, "_codec", kind, a, b, str(expect)], capture_output=True, text=True)
    return r, (open(b, "rb").read() if r.returncode == 0 else None)


def _samples():
    rng = np.random.default_rng(9)
    return [b"", b"a", b"abc", b"abcd" * 3, bytes(rng.integers(0, 4, 1001, dtype=np.uint8)), bytes(rng.integers(0, 256, 4099, dtype=np.uint8)),
            b"\0" * 77 + b"\1\2\3" * 50, bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), 3000, p=[.3, .2, .2, .29, .01])),
            bytes(np.repeat(rng.integers(30, 42, 400, dtype=np.uint8), rng.integers(1, 40, 400))), bytes(range(256)) * 3,
            bytes(rng.integers(0, 2, 777, dtype=np.uint8) * 7), b"Z" * 500]


@pytest.mark.parametrize("kw", [dict(order=0), dict(order=1), dict(order=0, x32=True), dict(order=1, x32=True), dict(order=1, shift=10), dict(order=1, compress_table=True),
                                dict(cat=True), dict(order=0, pack=True), dict(order=1, pack=True), dict(order=0, rle=True), dict(order=1, rle=True, pack=True),
                                dict(order=0, stripe=4), dict(order=1, stripe=3, pack=True), dict(order=1, x32=True, rle=True)])
def test_rans_nx16_roundtrip(tmp_path, kw):
    for k, data in enumerate(_samples()):
        enc = cramio.rans_nx16_encode(data, **kw)
        r, got = _decode("nx16", enc, len(data), tmp_path)
        assert r.returncode == 0 and got == data, (kw, k, len(data), r.stderr[-200:])


def test_rans_nx16_refuses_damage(tmp_path):
    rng = np.random.default_rng(4)
    data = bytes(np.repeat(rng.integers(30, 42, 300, dtype=np.uint8), rng.integers(1, 30, 300)))
    n_err = 0
    for kw in (dict(order=0), dict(order=1), dict(order=1, rle=True, pack=True), dict(order=0, stripe=4)):
        enc = bytearray(cramio.rans_nx16_encode(data, **kw))
        for _ in range(60):
            b = bytearray(enc)
            at = int(rng.integers(0, len(b)))
            b[at] ^= int(rng.integers(1, 256))
            if rng.random() < 0.3:
                del b[at:at + int(rng.integers(1, 6))]
            r, got = _decode("nx16", bytes(b), len(data), tmp_path)
            assert r.returncode in (0, 1), (kw, at, r.returncode, r.stderr[-200:])      # never a signal
            assert r.returncode == 1 or len(got) == len(data)
            n_err += r.returncode
        r, _ = _decode("nx16", bytes(enc), len(data) + 1, tmp_path)                       # the block's size field disagrees
        assert r.returncode == 1
    assert n_err > 20


def _names(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    lane, tile, x = 1, 1101, 1000
    for i in range(n):
        if rng.random() < 0.02:
            tile += 1
            x = 1000
        x += int(rng.integers(0, 300))
        y = int(rng.integers(1000, 100000))
        name = f"A00123:45:HXXXXDSXX:{lane}:{tile}:{x}:{y:06d}".encode() if i % 7 else f"A00123:45:HXXXXDSXX:{lane}:{tile}:{x}:{y}".encode()
        out.append(name)
        if rng.random() < 0.4:
            out.append(name)                       # the mate: a whole-name duplicate
    return out


@pytest.mark.parametrize("stream_kw", [dict(), dict(order=1), dict(cat=True), dict(order=0, pack=True, rle=True)])
def test_tok3_roundtrip(tmp_path, stream_kw):
    sets = [_names(500, 1), [b"q%d" % i for i in range(300)], [b"read.007", b"read.008", b"read.009", b"read.010", b"x", b"x", b"", b"0", b"00", b"12a34"],
            [b"SRR1.%d" % (i * 1000) for i in range(50)], [b"same"] * 40, [b"only"]]
    for k, names in enumerate(sets):
        enc = cramio.tok3_encode(names, stream_kw=stream_kw)
        want = b"".join(x + b"\0" for x in names)
        r, got = _decode("tok3", enc, len(want), tmp_path)
        assert r.returncode == 0 and got == want, (k, r.stderr[-200:], (got or b"")[:80], want[:80])


def test_tok3_refuses_damage_and_arith(tmp_path):
    names = _names(300, 3)
    enc = cramio.tok3_encode(names)
    want = b"".join(x + b"\0" for x in names)
    rng = np.random.default_rng(8)
    n_err = 0
    for _ in range(150):
        b = bytearray(enc)
        at = int(rng.integers(0, len(b)))
        b[at] ^= int(rng.integers(1, 256))
        r, got = _decode("tok3", bytes(b), len(want), tmp_path)
        assert r.returncode in (0, 1), (at, r.returncode, r.stderr[-200:])
        assert r.returncode == 1 or len(got) == len(want)
        n_err += r.returncode
    assert n_err > 30
    b = bytearray(enc)
    b[8] = 1                                         # "streams use the arithmetic coder"
    r, _ = _decode("tok3", bytes(b), len(want), tmp_path)
    assert r.returncode == 1 and "arithmetic coder" in r.stderr


def test_tok3_hostile_header_is_an_error_not_an_abort(tmp_path):
    """round-5 advisor: a 9-byte name tokeniser payload that claims 2^32 - 1 names sized three vectors by that count
    (std::bad_alloc, process aborted); a token stream that claims a gigabyte from a few bytes likewise.  Both are refused."""
    import struct
    r, _ = _decode("tok3", struct.pack("<IIB", 10, 0xFFFFFFFF, 0), 10, tmp_path)
    assert r.returncode == 1 and "more names than bytes" in r.stderr, (r.returncode, r.stderr[-200:])
    # one token position whose TYPE stream is a stored (CAT) rANS Nx16 stream claiming 2^29 bytes
    def u7(v):
        out = [v & 0x7f]
        v >>= 7
        while v:
            out.insert(0, (v & 0x7f) | 0x80)
            v >>= 7
        return bytes(out)
    stream = bytes([0x20]) + u7(1 << 29) + b"\0" * 8
    payload = struct.pack("<IIB", 40, 2, 0) + bytes([0x80 | 0]) + u7(len(stream)) + stream
    r, _ = _decode("tok3", payload, 40, tmp_path)
    assert r.returncode == 1 and "malformed" in r.stderr, (r.returncode, r.stderr[-200:])
