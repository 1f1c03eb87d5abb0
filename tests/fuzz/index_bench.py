"""`strling index` throughput: device window scoring + host merge/trim vs the oracle, on a synthetic chromosome.
usage: python tests/fuzz/index_bench.py [n_bases]   (test infrastructure: uses the oracle; GPU box; the oracle leg runs on a 4 Mbp prefix)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from strling_amd import api, synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
unit = synth.synth_chrom(4_000_000, 5)
seq = (unit * (n // len(unit) + 1))[:n]
ctx = api.Context(0)
ctx.set_opts(0.8, 40, 350)
ctx.index_chrom(seq[:1_000_000])
t0 = time.perf_counter(); words = ctx.index_chrom(seq); t1 = time.perf_counter()
reg = api.index_regions(seq, words); t2 = time.perf_counter()
t3 = time.perf_counter(); exp = O.index_chrom(unit.upper(), 0.8); t4 = time.perf_counter()
head = [r for r in reg if r[1] <= len(unit) - 200]
assert head == [r for r in exp if r[1] <= len(unit) - 200], "parity"
print(json.dumps({"n_bases": n, "windows": int(len(words)), "regions": len(reg), "device_score_s": round(t1 - t0, 4),
                  "host_merge_trim_s": round(t2 - t1, 4), "Mbp_per_s": round(n / (t2 - t0) / 1e6, 1),
                  "oracle_Mbp_per_s": round(len(unit) / (t4 - t3) / 1e6, 2)}))
