"""`strling call`'s evidence reads through the device (strl_regions_fetch; call.nim:196-218 / collect.nim:132-141: one
bam.query per bound): the ABI entry against a plain-Python walk of the zlib-inflated blocks, and the CLI with the device
path against the CLI with the host reader (whose outputs tests/test_call.py holds against the oracle)."""
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

from strling_amd import api, bamio, build, synth

CLI = build.CLI


def _blocks(path):
    """[(file offset, raw DEFLATE payload, ISIZE, CRC32)] of every BGZF block"""
    raw = open(path, "rb").read()
    out, o = [], 0
    while o < len(raw):
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        assert raw[o + 12:o + 14] == b"BC"
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        crc, isz = struct.unpack_from("<II", raw, o + bsize - 8)
        out.append((o, raw[o + 12 + xlen:o + bsize - 8], isz, crc))
        o += bsize
    return out


def _records(u, at):
    """(offset, block_size, refID, pos, upper bound of the end) of the records of the inflated stream from `at` on"""
    recs = []
    while at + 36 <= len(u):
        bs, ref, pos = struct.unpack_from("<iii", u, at)
        if at + 4 + bs > len(u):
            break
        l_name = u[at + 12]
        n_cig = struct.unpack_from("<H", u, at + 16)[0]
        span = 1 + sum(c >> 4 for c in struct.unpack_from(f"<{n_cig}I", u, at + 36 + l_name))
        recs.append((at, bs, ref, pos, pos + span))
        at += 4 + bs
    return recs, at


@pytest.mark.gpu
@pytest.mark.parametrize("block", [0xFF00, 1500])
def test_regions_fetch_matches_a_python_walk(block, tmp_path):
    rec, g = synth.synth_wgs(3000, seed=12, n_contigs=2, contig_len=20_000)
    bam = str(tmp_path / "r.bam")
    bamio.write_bam(bam, rec, block=block)
    blks = _blocks(bam)
    infl = [zlib.decompress(b[1], -15) for b in blks]
    assert [len(x) for x in infl] == [b[2] for b in blks]
    # where the first record sits: behind the header (magic, text, references)
    u = b"".join(infl)
    l_text = struct.unpack_from("<i", u, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", u, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", u, at)[0] + 4
    recs, _ = _records(u, at)
    assert len(recs) == rec.n
    ustart = np.concatenate([[0], np.cumsum([len(x) for x in infl])])
    rng = np.random.default_rng(3)
    regions, expect = [], []
    for _ in range(60):
        k = int(rng.integers(0, len(recs)))                       # the walk starts at some record (what the linear index would name)
        o, _, ref, pos, _ = recs[k]
        fb = int(np.searchsorted(ustart, o, side="right") - 1)
        nb = int(rng.integers(1, min(len(blks) - fb, 24) + 1))
        beg = pos + int(rng.integers(-50, 400))
        end = beg + int(rng.integers(1, 1500))
        regions.append((fb, nb, o - int(ustart[fb]), ref, beg, end))
        lim = int(ustart[fb + nb])
        keep, stop, status = None, None, 1
        for (ro, bs, rr, rp, ub) in recs[k:]:
            if ro + 36 > lim:
                break
            if rr != ref or rp >= end:
                stop, status = ro, 0
                break
            if ro + 4 + bs > lim:
                break
            if keep is None and ub > beg:
                keep = ro
        if status == 0 and keep is None:
            keep = stop
        expect.append((u[keep:stop] if status == 0 else b"", status))
    ctx = api.Context(0)
    got = ctx.regions_fetch([b[1] for b in blks], [b[2] for b in blks], regions, crcs=[b[3] for b in blks])
    assert sum(1 for e in expect if e[1] == 0 and e[0]) >= 10 and sum(1 for e in expect if e[1] == 1) >= 3
    for r, e, g_ in zip(regions, expect, got):
        assert g_[1] == e[1], r
        assert g_[0] == e[0], r
    # a damaged CRC is refused
    bad = [b[3] for b in blks]
    bad[regions[0][0]] ^= 1
    with pytest.raises(api.StrlingError):
        ctx.regions_fetch([b[1] for b in blks], [b[2] for b in blks], regions[:1], crcs=bad)
    ctx.close()


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([CLI] + args, capture_output=True, text=True, env=e)


@pytest.mark.gpu
@pytest.mark.parametrize("block,batch_mb", [(0xFF00, None), (4000, "0"), (0xFF00, "1")])
def test_call_device_regions_equal_host_regions(tmp_path, block, batch_mb):
    """the three outputs of `strling call` with the evidence reads on the device = with the host reader; contigs of several
    16 KiB index windows so that most regions can be bounded by the index (the last window's cannot: host reader)"""
    rec, g = synth.synth_wgs(12000, seed=31, n_contigs=2, contig_len=100_000)
    bam, bed, binp = str(tmp_path / "s.bam"), str(tmp_path / "ref.str"), str(tmp_path / "s.bin")
    bamio.write_bam(bam, rec, block=block, level=6)
    bamio.write_genome_bed(bed, g, rec.targets)
    r = _run(["extract", "-g", bed, bam, binp])
    assert r.returncode == 0, r.stderr
    outs = {}
    for mode in ("device", "host"):
        prefix = str(tmp_path / mode)
        env = {"STRL_CALL_REGIONS": "host"} if mode == "host" else ({"STRL_CALL_BATCH_MB": batch_mb} if batch_mb is not None else {})
        r = _run(["call", "-v", "-m", "3", "-o", prefix, bam, binp], env)
        assert r.returncode == 0, r.stderr
        m = re.search(r"regions through the device (\d+), on the host (\d+)", r.stderr)
        assert m, r.stderr
        outs[mode] = ([open(prefix + s).read() for s in ("-bounds.txt", "-genotype.txt", "-unplaced.txt")], int(m.group(1)), int(m.group(2)))
    assert outs["device"][0] == outs["host"][0]
    assert outs["device"][0][0].count("\n") >= 4
    assert outs["device"][1] >= 3 and outs["host"][1] == 0
