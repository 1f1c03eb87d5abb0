"""diagnostics: which treads differ between the file-sized run and the oracle's slab run"""
import os, subprocess, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import e2e_bench
from strling_amd import api, build, bamio
from oracle import oracle as O

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
inp = e2e_bench.make_input(n_pairs, level=1, progress=False)
modes = {"device": {}, "host_front": {"STRL_FRONT": "host"}, "host_pair": {"STRL_PAIR": "host"}}
slabs = e2e_bench.pick_slabs(inp["n_slabs"], 3)
for name, envx in modes.items():
    env = dict(os.environ, **envx)
    r = subprocess.run([build.CLI, "extract", "-v", "-g", inp["bed"], inp["bam"], inp["out"]], capture_output=True, text=True, env=env)
    print(name, "rc", r.returncode, r.stderr.splitlines()[-3:] if r.returncode else "")
    b = api.bin_read(inp["out"])
    t, qo, qn = b["treads"], b["qname_off"].astype(np.int64), b["qnames"]
    names = [qn[qo[i]:qo[i + 1]] for i in range(len(t))]
    med = O.median(b["frag"])
    print(name, "treads", len(t), "median", med)
    for c in slabs:
        rec, g = bamio.slab_records(c, inp["n_slabs"], inp["pairs_per_slab"], inp["seed"])
        et = O.extract(rec, g, O.make_opts(med, 0.8, 40))
        lo, hi = c * inp["pairs_per_slab"], (c + 1) * inp["pairs_per_slab"]
        sel = [i for i, nm in enumerate(names) if lo <= int(nm[1:]) < hi]
        key = lambda tt, nm: (nm, int(tt["tid"]), int(tt["position"]), bytes(tt["repeat"]), int(tt["flag"]), int(tt["split"]), int(tt["mapping_quality"]), int(tt["repeat_count"]), int(tt["align_length"]))
        got = [key(t[i], names[i]) for i in sel]
        exp = [key(et[j], rec.qname(int(et[j]["qname_id"]))) for j in range(len(et))]
        from collections import Counter
        cg, ce = Counter(got), Counter(exp)
        miss = list((ce - cg).elements())
        extra = list((cg - ce).elements())
        print(name, "slab", c, "got", len(got), "exp", len(exp), "missing", len(miss), "extra", len(extra), "same order", got == exp)
        for m in miss[:6]:
            # the records of that qname
            idx = [i for i in range(rec.n) if rec.qname(i) == m[0]]
            print("   missing", m, "records:", [(int(rec.tid[i]), int(rec.pos[i]), hex(int(rec.flag[i])), int(rec.mapq[i]), i, rec.n) for i in idx])
        for m in extra[:6]:
            print("   extra", m)
e2e_bench.cleanup(inp)
