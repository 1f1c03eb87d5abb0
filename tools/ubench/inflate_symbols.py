"""cost of a literal / of a match in the device inflate: a Huffman-only stream (every symbol a literal), streams of short near
matches, and the BAM mixes"""
import os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from strling_amd import api
rng = np.random.default_rng(3)
ctx = api.Context(0)
def deflate(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(raw) + c.flush()
def run(name, raws, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, symbols=None):
    S = [deflate(r, level, strategy) for r in raws]
    Z = [len(r) for r in raws]
    rep = max(1, 32768 // len(S))
    S, Z = S * rep, Z * rep
    best = None
    for _ in range(3):
        out = ctx.inflate_blocks(S, Z)
        ms = ctx.inflate_ms()
        best = ms if best is None else min(best, ms)
    assert out[0] == raws[0] and out[-1] == raws[(len(S) - 1) % len(raws)]
    tot = sum(Z)
    waves = 256 * 24
    line = f"{name}: {len(S)} blocks {tot/1e6:.0f} MB ratio {sum(map(len,S))/tot:.2f} kernel {best:.2f} ms = {tot/best/1e6:.1f} GB/s"
    if symbols:
        n_sym = symbols * rep
        line += f"; {n_sym/1e6:.0f} M symbols -> {best*1e-3*2.4e9*waves/n_sym:.0f} wave-cycles per symbol at 24 waves/CU"
    print(line, flush=True)
n_blocks, B = 512, 65280
lit = [rng.choice(np.array([2, 12, 23, 37], np.uint8), B, p=[.03, .07, .15, .75]).tobytes() for _ in range(n_blocks)]
run("huffman-only, 4 quality bins (1-3 bit codes)", lit, 6, zlib.Z_HUFFMAN_ONLY, symbols=n_blocks * B)
rnd = [rng.integers(0, 256, B, dtype=np.uint8).tobytes() for _ in range(n_blocks)]
run("huffman-only, uniform bytes (8 bit codes)", rnd, 6, zlib.Z_HUFFMAN_ONLY, symbols=n_blocks * B)
# matches of length 8 at distance 64: a random 64-byte seed repeated with one fresh byte every 8 -> (1 literal + 1 match of ~7) per 8 bytes
def matchy(step):
    a = rng.integers(0, 256, B, dtype=np.uint8)
    for i in range(64, B):
        if i % step:
            a[i] = a[i - 64]
    return a.tobytes()
for step in (8, 16, 32):
    run(f"1 literal + 1 match of {step - 1} per {step} bytes (distance 64)", [matchy(step) for _ in range(n_blocks)], 6, symbols=n_blocks * (B // step) * 2)
