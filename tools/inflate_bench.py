"""Throughput of the device BGZF inflate on the blocks of a synthetic BAM (run under rocprofv3 --kernel-trace --stats).
usage: python tools/inflate_bench.py [n_pairs]"""
import os, struct, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from strling_amd import api, bamio, synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
bam = "/tmp/inflate_bench.bam"
rec, g = synth.synth_wgs_chunks(max(1, min(32, n_pairs // 65536)), n_pairs // max(1, min(32, n_pairs // 65536)), seed=5)
bamio.write_bam_parallel(bam, rec, level=int(os.environ.get("LEVEL", "1")))
data = open(bam, "rb").read()
streams, sizes, o = [], [], 0
while o < len(data):
    xlen = struct.unpack_from("<H", data, o + 10)[0]
    bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
    isz = struct.unpack_from("<I", data, o + bsize - 4)[0]
    if isz:
        streams.append(data[o + 12 + xlen:o + bsize - 8]); sizes.append(isz)
    o += bsize
print(len(streams), "blocks", sum(sizes) / 1e6, "MB inflated", len(data) / 1e6, "MB compressed")
ctx = api.Context(0)
for _ in range(3):
    t = time.time()
    out = ctx.inflate_blocks(streams, sizes)
    print("inflate_blocks wall %.3f s (incl. copies)" % (time.time() - t))
import zlib
t = time.time()
ref = [zlib.decompress(s, -15) for s in streams[:2000]]
print("zlib 1 thread: %.1f MB/s" % (sum(map(len, ref)) / 1e6 / (time.time() - t)))
assert out[:2000] == ref
