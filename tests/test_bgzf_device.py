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


from test_inflate_emu import corpus as _corpus, deflate as _deflate, VARIANTS


def test_inflate_matches_zlib_for_every_block_type(ctx):
    rng = np.random.default_rng(3)
    streams, plain = [], []
    for data in _corpus(rng):
        for kw in VARIANTS:
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
