"""verify/: the kit a maintainer with a real `strling` binary pins the unpinned third-party assumptions with.  Here: the committed
expectations are what the oracle produces today (CPU), and the product's CLI reproduces them (GPU)."""
import filecmp
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "verify")


def test_committed_expectations_are_the_oracles(tmp_path, monkeypatch):
    sys.path.insert(0, KIT)
    import make_kit
    monkeypatch.setattr(make_kit, "CASES", str(tmp_path))
    make_kit.main()
    names = sorted(os.listdir(os.path.join(KIT, "cases")))
    assert names == sorted(os.listdir(tmp_path)) and len(names) >= 15
    for n in names:
        assert filecmp.cmp(os.path.join(KIT, "cases", n), os.path.join(tmp_path, n), shallow=False), n
    # the cases do exercise what they are for
    tsv = open(os.path.join(KIT, "cases", "iupac.expected.treads.tsv")).read().splitlines()[1:]
    clip = [l.split("\t")[-1] for l in tsv if l.split("\t")[4] == "0"]
    assert len(clip) >= 12 and all(q.endswith("_A") for q in clip if "_" in q)       # soft-clip treads only where the code counts as A
    blob = open(os.path.join(KIT, "cases", "widths.expected.bin"), "rb").read()
    assert all(bytes([m]) in blob for m in (0xcc, 0xcd, 0xce, 0xd9, 0xff))
    rows = open(os.path.join(KIT, "cases", "manygroups.expected-bounds.txt")).read().splitlines()
    assert len(rows) == 9001


@pytest.mark.gpu
def test_the_product_reproduces_the_kit():
    from strling_amd import build
    r = subprocess.run(["bash", os.path.join(KIT, "run_reference.sh"), build.CLI], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("PASS") == 4, r.stdout + r.stderr
