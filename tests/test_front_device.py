"""The BAM front end on the device (front.hip: inflate -> record scan -> parse -> scorer -> pair logic) against the path that
parses the same records on the host: treads field by field in .bin order, their qnames, the fragment-length words of every
record, chunk summaries.  Small chunks and odd block sizes put records across BGZF blocks, 16 KiB scan segments and chunks."""
import os

import numpy as np
import pytest

from strling_amd import api, bamio, synth

pytestmark = pytest.mark.gpu

FIELDS = ("tid", "position", "repeat", "flag", "split", "mapping_quality", "repeat_count", "align_length", "qname_id")


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _host_reference(ctx, rec, g, med):
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    exp, _ = ctx.extract(rec)
    return exp


def _check(ctx, rec, g, path, chunk_blocks, exp):
    from oracle import oracle as O
    med = O.median(synth.frag_hist(rec))
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    got = ctx.extract_bam_device(path, chunk_blocks=chunk_blocks)
    assert got["n_records"] == rec.n
    assert [t[0] for t in got["targets"]] == [t[0] for t in rec.targets]
    for f in FIELDS:
        assert np.array_equal(got["treads"][f], exp[f]), (f, chunk_blocks)
    assert got["qnames"] == [rec.qname(int(i)) for i in exp["qname_id"]]
    isz = rec.isize if rec.isize is not None else np.zeros(rec.n, np.int32)
    fw = rec.flag.astype(np.uint32) | (np.where((isz >= 0) & (isz <= 4095), isz, 0xffff).astype(np.uint32) << 16)
    assert np.array_equal(got["fragwords"], fw)
    prim = (rec.flag & 0x900) == 0
    assert sum(c["n_primary"] for c in got["chunks"]) == int(prim.sum())
    n_tail = 0
    while n_tail < rec.n and rec.tid[rec.n - 1 - n_tail] < 0:
        n_tail += 1
    assert got["n_tail"] == n_tail
    if n_tail < rec.n:      # the last chunk with a placed record counts the primary records behind it
        last = [c for c in got["chunks"] if c["last_placed"] >= 0][-1]
        assert last["tail_primary"] <= int(prim[rec.n - n_tail:].sum())
    seen = np.zeros(len(rec.targets), np.uint8)
    seen[np.unique(rec.tid[prim & (rec.tid >= 0)])] = 1
    assert np.array_equal(got["tids_seen"], seen)
    return got


@pytest.mark.parametrize("n_pairs,block,chunk_blocks", [(3000, 0xFF00, 16384), (6000, 4099, 7), (6000, 0xFF00, 3), (20000, 30011, 5)])
def test_front_end_matches_the_host_parsed_path(ctx, tmp_path, n_pairs, block, chunk_blocks):
    from oracle import oracle as O
    rec, g = synth.synth_wgs(n_pairs, seed=11 + n_pairs, contig_len=600_000)
    med = O.median(synth.frag_hist(rec))
    exp = _host_reference(ctx, rec, g, med)
    assert len(exp) > 50
    path = str(tmp_path / "a.bam")
    bamio.write_bam(path, rec, level=6 if block < 0xFF00 else 1, block=block, index=False)
    got = _check(ctx, rec, g, path, chunk_blocks, exp)
    assert len(got["chunks"]) >= 1
    if chunk_blocks < 100:
        assert len(got["chunks"]) > 3


def test_front_end_against_the_oracle(ctx, tmp_path):
    """... and the whole thing against the CPU restatement of the reference (not only the product's other path)"""
    from oracle import oracle as O
    rec, g = synth.synth_wgs(8000, seed=5, contig_len=800_000)
    med = O.median(synth.frag_hist(rec))
    exp = O.extract(rec, g, O.make_opts(med, 0.8, 40))
    path = str(tmp_path / "o.bam")
    bamio.write_bam(path, rec, level=1, block=20011, index=False)
    _check(ctx, rec, g, path, 9, exp)


def test_front_end_many_soft_clips_per_chunk(ctx, tmp_path):
    """more than a quarter of the reads of a chunk carry a scored soft clip (adapter-like libraries): every soft-clip record
    must survive the chunked extract (the per-chunk queue holds two per read)"""
    from oracle import oracle as O
    rec, g = synth.synth_wgs(5000, seed=23, contig_len=500_000, soft_frac=0.6)
    med = O.median(synth.frag_hist(rec))
    exp = O.extract(rec, g, O.make_opts(med, 0.8, 40))
    path = str(tmp_path / "s.bam")
    bamio.write_bam(path, rec, level=1, index=False)
    _check(ctx, rec, g, path, 16384, exp)


def _corrupt(path, out, what):
    """copy of a BAM with one byte changed in the second-to-last data block: its CRC-32 field, or its DEFLATE payload"""
    import struct
    raw = bytearray(open(path, "rb").read())
    offs, o = [], 0
    while o < len(raw):
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        if struct.unpack_from("<I", raw, o + bsize - 4)[0]:
            offs.append((o, bsize))
        o += bsize
    o, bsize = offs[-2]
    if what == "crc":
        raw[o + bsize - 8] ^= 0x40
    else:
        raw[o + 18 + (bsize - 26) // 2] ^= 0x10
    open(out, "wb").write(bytes(raw))


def test_front_end_checks_the_blocks_crc32(ctx, tmp_path):
    """htslib refuses a BGZF block whose inflated bytes do not have the CRC-32 of its trailer; so does the device front end --
    and a damaged DEFLATE payload is reported as a format error (the CLI then lets the host reader, i.e. zlib, judge the file)"""
    from oracle import oracle as O
    rec, g = synth.synth_wgs(4000, seed=3, contig_len=400_000)
    med = O.median(synth.frag_hist(rec))
    ctx.set_opts(0.8, 40, med)
    ctx.set_genome(g)
    good = str(tmp_path / "g.bam")
    bamio.write_bam(good, rec, level=6, block=20011, index=False)
    exp = ctx.extract_bam_device(good, chunk_blocks=5)
    assert exp["n_records"] == rec.n
    bad = str(tmp_path / "c.bam")
    _corrupt(good, bad, "crc")
    with pytest.raises(api.StrlingError) as e:
        ctx.extract_bam_device(bad, chunk_blocks=5)
    assert "CRC32" in str(e.value)
    got = ctx.extract_bam_device(bad, chunk_blocks=5, check_crc=False)        # the bytes themselves are intact
    assert np.array_equal(got["treads"], exp["treads"])
    _corrupt(good, bad, "payload")
    with pytest.raises(api.StrlingError):
        ctx.extract_bam_device(bad, chunk_blocks=5)


@pytest.mark.parametrize("n_ctx,block,chunk_blocks", [(2, 0xFF00, 16384), (3, 4099, 7), (5, 1500, 33)])
def test_shares_through_the_abi_against_the_oracle(tmp_path, n_ctx, block, chunk_blocks):
    """`extract --gpus N` as a host of the C ABI drives it: a contiguous share of the file per context, every share starting at a
    record start and ending where the next begins (strl_front_trim_next / strl_front_tail_bytes), the per-read state appended on
    the first context (strl_ctxs_extract_gather) -- the treads of the oracle over the whole file, field by field and in order;
    cuts moved off the record starts are noticed (a share's tail is not empty, or its records are refused as malformed)"""
    from oracle import oracle as O
    rec, g = synth.synth_wgs(7000, seed=77 + n_ctx, contig_len=700_000)
    med = O.median(synth.frag_hist(rec))
    exp = O.extract(rec, g, O.make_opts(med, 0.8, 40))
    path = str(tmp_path / "sh.bam")
    bamio.write_bam(path, rec, level=1, block=block, index=False)
    ctxs = [api.Context(0) for _ in range(n_ctx)]
    try:
        for c in ctxs:
            c.set_opts(0.8, 40, med)
            c.set_genome(g)
        got, tails = api.extract_bam_shares(ctxs, path, chunk_blocks=chunk_blocks)
        assert tails[:-1] == [0] * (n_ctx - 1) and got is not None
        assert got["n_records"] == rec.n
        for f in FIELDS:
            assert np.array_equal(got["treads"][f], exp[f]), f
        assert got["qnames"] == [rec.qname(int(i)) for i in exp["qname_id"]]
        isz = rec.isize if rec.isize is not None else np.zeros(rec.n, np.int32)
        fw = rec.flag.astype(np.uint32) | (np.where((isz >= 0) & (isz <= 4095), isz, 0xffff).astype(np.uint32) << 16)
        assert np.array_equal(got["fragwords"], fw)
    finally:
        for c in ctxs:
            c.close()
    # a cut that is NOT a record start
    ctxs = [api.Context(0) for _ in range(2)]
    try:
        for c in ctxs:
            c.set_opts(0.8, 40, med)
            c.set_genome(g)
        try:
            got, tails = api.extract_bam_shares(ctxs, path, chunk_blocks=chunk_blocks, cut_shift=5)
            assert got is None and tails[0] != 0
        except api.StrlingError as e:
            assert "malformed BAM record" in str(e) or "more than" in str(e)
    finally:
        for c in ctxs:
            c.close()
