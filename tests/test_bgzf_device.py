"""The device DEFLATE decoder (strling_amd/csrc/bgzf.hip) against zlib: every block type, many streams per launch."""
import zlib

import numpy as np
import pytest

from strling_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for o in range(0, len(data), flush_every):
        out += c.compress(data[o:o + flush_every]) + c.flush(zlib.Z_FULL_FLUSH if (o // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + c.flush()


def _corpus(rng):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    yield b""
    yield b"A"
    yield bytes(rng.choice(acgt, 65280))                                   # a full BGZF block of sequence-like text
    yield bytes(rng.integers(0, 256, 65280, dtype=np.uint8))               # incompressible
    yield b"\xff" * 65280                                                  # one long run (distance-1 matches of length 258)
    yield (b"CAG" * 30000)[:65000]
    yield bytes(rng.integers(0, 4, 30000, dtype=np.uint8)) + b"\0" * 20000 + bytes(rng.choice(acgt, 15000))
    rec = bytearray()
    for i in range(230):                                                   # BAM-record-like: binary header + name + packed seq + 0xff quals
        rec += bytes(rng.integers(0, 256, 36, dtype=np.uint8)) + b"q%d\0" % (i * 7919) + bytes(rng.integers(0, 256, 75, dtype=np.uint8)) + b"\xff" * 150
    yield bytes(rec)


def test_inflate_matches_zlib_for_every_block_type(ctx):
    rng = np.random.default_rng(3)
    streams, plain = [], []
    for data in _corpus(rng):
        for kw in (dict(level=1), dict(level=6), dict(level=9), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED),
                   dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=6, strategy=zlib.Z_RLE), dict(level=4, flush_every=5000)):
            streams.append(_deflate(data, **kw))
            plain.append(data)
    assert len(streams) > 60                      # more than one wave of lanes
    got = ctx.inflate_blocks(streams, [len(p) for p in plain])
    for i, (g, p) in enumerate(zip(got, plain)):
        assert g == p, (i, len(g), len(p))


def test_inflate_many_blocks_and_bad_data(ctx):
    rng = np.random.default_rng(4)
    plain = [bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(rng.integers(1, 65280)))) for _ in range(700)]
    streams = [_deflate(p, level=int(rng.integers(1, 10))) for p in plain]
    got = ctx.inflate_blocks(streams, [len(p) for p in plain])
    assert got == plain
    bad = list(streams)
    bad[5] = bytes(rng.integers(0, 256, len(bad[5]), dtype=np.uint8))       # garbage instead of a stream
    with pytest.raises(api.StrlingError):
        ctx.inflate_blocks(bad, [len(p) for p in plain])
    with pytest.raises(api.StrlingError):                                    # wrong ISIZE
        ctx.inflate_blocks(streams, [len(p) + (1 if i == 9 else 0) for i, p in enumerate(plain)])
