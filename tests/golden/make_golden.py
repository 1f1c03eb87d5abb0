"""Generates tests/golden/s1_small.npz: a small synthetic S1 batch's expected outputs, produced by the ORACLE
(self-generated -- the real Nim binary cannot be built in this image).  The fixture pins the
oracle itself against accidental drift and gives the GPU tests a committed vector to match.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O          # noqa: E402
from strling_amd import synth           # noqa: E402
from helpers import oracle_words        # noqa: E402

PARAMS = dict(n_pairs=3000, seed=2024, contig_len=400_000, p=0.8, min_mapq=40, min_support=3)


def main():
    rec, g = synth.synth_wgs(PARAMS["n_pairs"], seed=PARAMS["seed"], contig_len=PARAMS["contig_len"])
    frag = synth.frag_hist(rec)
    med = O.median(frag)
    opts = O.make_opts(med, PARAMS["p"], PARAMS["min_mapq"])
    whole, _ = oracle_words(O, rec, g, opts)
    treads = O.extract(rec, g, opts)
    window, mcd = O.median(frag, 0.99), int(0.5 * med)
    bounds, unplaced = O.call_bounds(treads, 1, window, min_support=PARAMS["min_support"], max_clip_dist=mcd)
    rows = [O.bounds_row(b, rec.targets[int(b["tid"])][0]) for b in bounds]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s1_small.npz")
    np.savez_compressed(out, whole=whole, tread_tid=treads["tid"], tread_pos=treads["position"], tread_repeat=treads["repeat"],
                        tread_flag=treads["flag"], tread_split=treads["split"], tread_mapq=treads["mapping_quality"],
                        tread_count=treads["repeat_count"], tread_alen=treads["align_length"], tread_qid=treads["qname_id"],
                        bounds_rows=np.array(rows), unplaced=np.array([f"{u}\t{c}" for u, c in unplaced]),
                        frag_median=med, window=window, max_clip_dist=mcd)
    print(out, len(treads), "treads", len(rows), "bounds rows", len(unplaced), "unplaced units")


if __name__ == "__main__":
    main()
