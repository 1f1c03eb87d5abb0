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


@pytest.mark.parametrize("form,lanes", [("wave", 64), ("group", 8), ("group", 4), ("group", 16)])
def test_both_forms_long_codes_and_periodic_matches(form, lanes):
    """both forms of the decoder -- the wave form (inflate_wave.h) and the grouped one (inflate_group.h: `lanes` lanes per stream,
    64 / lanes streams per wave; chosen by STRL_INFLATE_FORM when the library is first used, hence a process of its own): the
    vectors above, the 10..15-bit literal and distance codes and the periodic matches of the CPU suite (the symbol loops decode
    long codes in line), corrupt streams refused"""
    import os, subprocess, sys
    from strling_amd import build as _b
    if form == "group" and not _b.WITH_INFLATE_GROUP:
        pytest.skip("the grouped form is not in the shipped library (STRL_WITH_INFLATE_GROUP=1 python -m strling_amd.build compiles it in)")
    code = (
        "import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, zlib, test_bgzf_device as T, test_inflate_emu as E\n"
        "from strling_amd import api\n"
        "c = api.Context(0)\n"
        "T.test_inflate_matches_zlib_for_every_block_type(c)\n"
        "T.test_inflate_many_blocks_and_bad_data(c)\n"
        "rng = np.random.default_rng(11)\n"
        "p = 0.5 ** np.arange(1, 257)\n"
        "blocks = [bytes(rng.choice(256, 60000, p=p / p.sum()).astype(np.uint8))]\n"
        "far = bytes(rng.integers(0, 256, 400, dtype=np.uint8))\n"
        "blocks.append(far + bytes(32768 - 400) + far + blocks[0][:3000] + far)\n"
        "for d in range(1, 70):\n"
        "    u = bytes(rng.integers(0, 256, d, dtype=np.uint8))\n"
        "    blocks.append((u * (900 // d + 2))[:900] + bytes(rng.integers(0, 256, 30, dtype=np.uint8)))\n"
        "S = [E.deflate(b, level=l) for b in blocks for l in (1, 6, 9)]\n"
        "P = [b for b in blocks for l in (1, 6, 9)]\n"
        "for k in range(40):\n"
        "    b = E.long_then_short_literals(rng)\n"
        "    S.append(E.deflate(b, level=6, strategy=zlib.Z_HUFFMAN_ONLY)); P.append(b)\n"
        "assert c.inflate_blocks(S, [len(x) for x in P]) == P\n"
        "print('form ok', len(S))\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = dict(os.environ, STRL_INFLATE_FORM=form, STRL_INFLATE_G=str(lanes))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "form ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
