import sys, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from strling_amd import api
from test_inflate_emu import deflate
ctx = api.Context(0)
rng = np.random.default_rng(4)
plain = [bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(rng.integers(1, 65280)))) for _ in range(700)]
streams = [deflate(p, level=int(rng.integers(1, 10))) for p in plain]
got = ctx.inflate_blocks(streams, [len(p) for p in plain])
bad = [i for i in range(700) if got[i] != plain[i]]
print("bad blocks", len(bad), bad[:20])
for i in bad[:5]:
    g, p = got[i], plain[i]
    first = next(j for j in range(len(p)) if g[j] != p[j])
    nb = sum(1 for j in range(len(p)) if g[j] != p[j])
    off = sum(len(x) for x in plain[:i])
    print(i, "len", len(p), "first mismatch", first, "n mismatching", nb, "uoff%16", off % 16, "lane", i % 64, g[first:first+8], p[first:first+8])
