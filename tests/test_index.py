"""`strling index` (genome_strs.nim:22-92): window scoring on the device, merge + trim on the host."""
import numpy as np
import pytest

from strling_amd import api, synth
from helpers import pack_word


def oracle_window_words(O, seq, p, window=100, step=60):
    up = seq.upper()
    return np.array([pack_word(*O.get_repeat(up[s:s + window], p)) for s in range(0, len(up), step)], np.uint32)


def test_reference_regression_window_fires_trims_doassert(oracle):
    """genome_strs.nim:203-206 (the commented-out `bug` case): no unit-sized step counted from the right end of that
    window is a rotation of CACGAT, so trim's second doAssert (:57) fires.  Both the oracle and the host logic must
    report it instead of returning a region."""
    dna = "ATAACACTTGGGGGTAGCTAAAGTGAACTGTATCCGACATCTGGTTCCTACTTCAGGGTCATAAAGCCTAAATAGCCCACACGTTCCCCTTAAATAAGACATCACGATG"
    assert len(dna) == 16569 - 16460
    words = np.array([pack_word("CACGAT", 15)], np.uint32)
    with pytest.raises(api.StrlingError, match="genome_strs.nim"):
        api.index_regions(dna.encode(), words, window=len(dna), step=len(dna) - 1)
    # the same window with the unit present at both ends trims cleanly to the outermost full units
    ok = "GG" + "CACGAT" * 17 + "TTTTT"
    (start, stop, unit), = api.index_regions(ok.encode(), np.array([pack_word("CACGAT", 17)], np.uint32), window=len(ok), step=len(ok) - 1)
    assert (unit, start % 6, stop) == ("CACGAT", 0, len(ok) - 6) and start <= 6


@pytest.mark.parametrize("n_bases,seed,p", [(60_000, 1, 0.8), (25_000, 2, 0.6), (9_999, 3, 0.9), (130, 4, 0.8), (59, 5, 0.8), (0, 6, 0.8)])
def test_host_merge_trim_matches_oracle(oracle, n_bases, seed, p):
    seq = synth.synth_chrom(n_bases, seed) if n_bases else b""
    exp = oracle.index_chrom(seq.upper(), p)
    words = oracle_window_words(oracle, seq, p)
    got = api.index_regions(seq, words)
    assert got == exp
    if n_bases >= 9_999:
        assert len(exp) >= 2


def test_one_skipped_window_still_merges(oracle):
    rng = np.random.default_rng(0)
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    seq = (rnd(500) + "CAG" * 40 + rnd(70) + "CAG" * 40 + rnd(500) + "AT" * 100 + "GGC" * 60 + rnd(300)).encode()
    exp = oracle.index_chrom(seq, 0.8)
    assert api.index_regions(seq, oracle_window_words(oracle, seq, 0.8)) == exp
    assert [u for _, _, u in exp] == ["CAG", "AT", "CGG"] or len(exp) >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("n_bases,seed,p", [(400_000, 11, 0.8), (50_001, 12, 0.6), (61, 13, 0.8)])
def test_index_windows_and_regions_match_oracle(ctx, oracle, n_bases, seed, p):
    seq = synth.synth_chrom(n_bases, seed)
    ctx.set_opts(p, 40, 350)
    words = ctx.index_chrom(seq)
    exp_words = oracle_window_words(oracle, seq.decode(), p)
    assert np.array_equal(words, exp_words), np.nonzero(words != exp_words)[0][:10]
    assert ctx.index_regions(seq) == oracle.index_chrom(seq.upper(), p)


def test_indexed_records_needs_the_metadata_bins(tmp_path):
    """an index written without the pseudo-bin 37450 (older tools) or without the trailing n_no_coor gives no count: the CLI then
    sizes by the file's bytes as before"""
    import struct
    import subprocess
    from strling_amd import build
    bam = str(tmp_path / "x.bam")
    open(bam, "wb").write(b"")
    plain = b"BAI\1" + struct.pack("<i", 1) + struct.pack("<i", 1) + struct.pack("<IiQQ", 4681, 1, 100 << 16, 200 << 16) + struct.pack("<iQ", 1, 100 << 16)
    meta = b"BAI\1" + struct.pack("<i", 2) + struct.pack("<i", 2) + struct.pack("<IiQQ", 4681, 1, 100 << 16, 200 << 16) + struct.pack("<IiQQQQ", 37450, 2, 100 << 16, 200 << 16, 7, 2) + \
        struct.pack("<iQ", 1, 100 << 16) + struct.pack("<ii", 0, 0)                     # a second reference without records
    for data, want in ((plain, "unknown"), (plain + struct.pack("<Q", 5), "unknown"), (meta, "unknown"), (meta + struct.pack("<Q", 5), "14"), (meta[:40], "unknown")):
        open(bam + ".bai", "wb").write(data)
        r = subprocess.run([build.CLI, "_indexed_records", bam], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == want, (want, r.stdout, r.stderr)


def test_an_index_with_counts_the_file_cannot_hold_is_corrupt_not_an_allocation(tmp_path):
    """n_ref / n_chunk / n_intv of a .bai are trusted only as far as the file's size goes: the three parsers (record counts, share
    cuts, region reads) say unknown / corrupt at once instead of reserving gigabytes"""
    import struct
    import subprocess
    import time
    from strling_amd import bamio, build, synth
    rec, _ = synth.synth_wgs_30x(1, 400, seed=3, procs=1)
    bam = str(tmp_path / "x.bam")
    bamio.write_bam(bam, rec, level=1)
    good = open(bam + ".bai", "rb").read()
    hostile = [b"BAI\1" + struct.pack("<i", 0x7fffffff),                                             # references
               b"BAI\1" + struct.pack("<ii", 1, 1) + struct.pack("<Ii", 4681, 0x0fffffff),             # chunks of a bin
               b"BAI\1" + struct.pack("<ii", 1, 0) + struct.pack("<i", 0x7fffffff)]                    # intervals
    for data in hostile:
        open(bam + ".bai", "wb").write(data)
        t0 = time.time()
        r = subprocess.run([build.CLI, "_indexed_records", bam], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == "unknown"
        r = subprocess.run([build.CLI, "_shares", bam, "4"], capture_output=True, text=True)
        assert r.returncode in (0, 1) and "bad_alloc" not in r.stderr, r.stderr
        r = subprocess.run([build.CLI, "_region", bam, "0", "0", "1000"], capture_output=True, text=True)
        assert r.returncode != 0 and "bad_alloc" not in r.stderr and "corrupt .bai" in r.stderr, r.stderr
        assert time.time() - t0 < 5
    open(bam + ".bai", "wb").write(good)
    assert subprocess.run([build.CLI, "_indexed_records", bam], capture_output=True, text=True).stdout.strip() == str(rec.n)
