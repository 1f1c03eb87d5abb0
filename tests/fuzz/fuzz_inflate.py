"""Fuzzer of the device DEFLATE decoder (bgzf.hip: the wave form, or the grouped form with STRL_INFLATE_FORM=group) against zlib's
own inverse: random blocks of many shapes -- skewed literal alphabets with 10..15-bit codes in front of short ones, runs, periodic
text, BAM-like records, incompressible bytes -- deflated with random level / strategy / flush points, inflated on the device in
batches, compared byte for byte.  usage: python tests/fuzz/fuzz_inflate.py [seconds] [master seed]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from strling_amd import api
from test_inflate_emu import deflate, long_then_short_literals


def block(rng):
    kind = int(rng.integers(0, 8))
    n = int(rng.integers(1, 65281))
    if kind == 0:                                           # geometric alphabet: code lengths 1..15
        p = rng.uniform(0.4, 0.9) ** np.arange(256)
        return bytes(rng.permutation(256).astype(np.uint8)[rng.choice(256, n, p=p / p.sum())])
    if kind == 1:
        return long_then_short_literals(rng, max(n, 40000))
    if kind == 2:                                           # runs of few values (distance-1 matches of every length)
        v = rng.integers(0, 256, 4, dtype=np.uint8)
        return bytes(np.repeat(v[rng.integers(0, 4, n // 3 + 1)], rng.integers(1, 40, n // 3 + 1))[:n])
    if kind == 3:                                           # periodic text, period 1..300
        d = int(rng.integers(1, 301))
        unit = bytes(rng.integers(0, 256, d, dtype=np.uint8))
        return (unit * (n // d + 2))[:n]
    if kind == 4:                                           # BAM-like: binary header, name, packed sequence, binned qualities
        out = bytearray()
        bins = np.array([2, 12, 23, 37], np.uint8)
        while len(out) < n:
            out += bytes(rng.integers(0, 256, 36, dtype=np.uint8)) + b"r%d\0" % int(rng.integers(0, 10 ** 9)) + bytes(rng.integers(0, 256, 75, dtype=np.uint8))
            out += bytes(rng.choice(bins, 150, p=[0.03, 0.07, 0.15, 0.75]))
        return bytes(out[:n])
    if kind == 5:
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == 6:                                           # far matches: a piece repeated 1..32 KiB back
        piece = bytes(rng.integers(0, 256, int(rng.integers(3, 400)), dtype=np.uint8))
        gap = bytes(rng.integers(0, 4, int(rng.integers(0, 32768)), dtype=np.uint8))
        return (piece + gap + piece + bytes(rng.integers(0, 256, 50, dtype=np.uint8)) + piece)[:65280]
    return bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), n))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    master = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(master)
    ctx = api.Context(0)
    print(f"fuzz_inflate: master seed {master}, {seconds:.0f} s, form {os.environ.get('STRL_INFLATE_FORM', 'default')}", flush=True)
    t0, n_blocks, n_bytes, batch = time.time(), 0, 0, 0
    while time.time() - t0 < seconds:
        seed = int(rng.integers(0, 2 ** 31))
        r = np.random.default_rng(seed)
        plain, streams = [], []
        for _ in range(int(r.integers(1, 200))):
            b = block(r)
            kw = dict(level=int(r.integers(0, 10)))
            s = int(r.integers(0, 6))
            if s == 1: kw["strategy"] = zlib.Z_HUFFMAN_ONLY
            elif s == 2: kw["strategy"] = zlib.Z_RLE
            elif s == 3: kw["strategy"] = zlib.Z_FIXED
            elif s == 4: kw["flush_every"] = int(r.integers(200, 20000))
            plain.append(b); streams.append(deflate(b, **kw))
        got = ctx.inflate_blocks(streams, [len(p) for p in plain])
        for i, (g, p) in enumerate(zip(got, plain)):
            if g != p:
                print(f"MISMATCH batch seed {seed} block {i} ({len(p)} bytes)", flush=True)
                sys.exit(1)
        n_blocks += len(plain); n_bytes += sum(map(len, plain)); batch += 1
        if batch % 25 == 0:
            print(f"  {batch} batches, {n_blocks} blocks, {n_bytes / 1e6:.0f} MB, {time.time() - t0:.0f} s, last batch seed {seed}", flush=True)
    print(f"fuzz_inflate ok (master seed {master}): {n_blocks} blocks, {n_bytes / 1e6:.0f} MB identical to the input in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
