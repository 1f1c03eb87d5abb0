"""The whole-genome sized BAM writer (bamio.write_bam_slabs + tools/bamgen/bamgen.c: test / bench infrastructure) against the
plain Python writer: same record bytes, an index that says the same thing, slabs that can be regenerated alone."""
import os
import struct
import zlib

import numpy as np

from strling_amd import bamio, synth


def _inflate(path):
    """-> (uncompressed bytes, {file offset of block: offset of its first byte in the uncompressed stream})"""
    data = open(path, "rb").read()
    out, at, o, total = [], {}, 0, 0
    while o < len(data):
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        raw = zlib.decompress(data[o + 18:o + bsize - 8], -15)
        assert zlib.crc32(raw) == struct.unpack_from("<I", data, o + bsize - 8)[0] and len(raw) == struct.unpack_from("<I", data, o + bsize - 4)[0]
        at[o] = total
        out.append(raw)
        total += len(raw)
        o += bsize
    at[len(data)] = total
    return b"".join(out), at


def _bai_abs(path, at):
    """the .bai with every virtual offset turned into an offset of the uncompressed stream"""
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", d, 4)[0]
    o = 8
    refs = []
    ab = lambda v: at[v >> 16] + (v & 0xffff)
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, o)[0]; o += 4
        bins = {}
        for _ in range(n_bin):
            b, nc = struct.unpack_from("<Ii", d, o); o += 8
            if b == 37450:                                   # the metadata pseudo-bin: (file span), (n_mapped, n_unmapped)
                v0, v1, n_map, n_unm = struct.unpack_from("<QQQQ", d, o); o += 32
                assert nc == 2
                bins[b] = [[ab(v0), ab(v1)], [n_map, n_unm]]
                continue
            ch = []
            for _ in range(nc):
                v0, v1 = struct.unpack_from("<QQ", d, o); o += 16
                if ch and ch[-1][1] == ab(v0):
                    ch[-1][1] = ab(v1)                       # adjacent chunks say the same as one
                else:
                    ch.append([ab(v0), ab(v1)])
            bins[b] = ch
        n_intv = struct.unpack_from("<i", d, o)[0]; o += 4
        lin = [ab(v) if v else None for v in struct.unpack_from(f"<{n_intv}Q", d, o)]; o += 8 * n_intv
        refs.append((bins, lin))
    no_coor = struct.unpack_from("<Q", d, o)[0]; o += 8
    assert o == len(d)
    return refs, no_coor


def test_slab_file_equals_the_python_writer(tmp_path):
    n_slabs, pairs, seed = 3, 1500, 77
    a, b = str(tmp_path / "slab.bam"), str(tmp_path / "plain.bam")
    r = bamio.write_bam_slabs(a, n_slabs, pairs, seed=seed, level=1, quals=False, aux=False, bed=str(tmp_path / "slab.str"), procs=2)
    rec, g = synth.synth_wgs_30x(n_slabs, pairs, seed=seed, procs=1)
    assert r["reads"] == rec.n == n_slabs * 2 * pairs and r["targets"] == rec.targets
    bamio.write_bam(b, rec, level=1)
    bamio.write_genome_bed(str(tmp_path / "plain.str"), g, rec.targets)
    ra, at_a = _inflate(a)
    rb, at_b = _inflate(b)
    assert ra == rb                                                    # header + every record, byte for byte
    assert open(tmp_path / "slab.str").read() == open(tmp_path / "plain.str").read()
    (ia, no_coor_a), (ib, no_coor_b) = _bai_abs(a + ".bai", at_a), _bai_abs(b + ".bai", at_b)
    assert len(ia) == len(ib) == 2 * n_slabs
    for (bins_a, lin_a), (bins_b, lin_b) in zip(ia, ib):
        assert bins_a == bins_b                                        # (the metadata pseudo-bin 37450 with them: span and counts)
        assert [x for x in lin_a] == [x for x in lin_b]
    # the counts `samtools idxstats` reads: mapped + placed-unmapped per reference, the unplaced tail -- and the CLI's sum of them
    assert no_coor_a == no_coor_b == int((rec.tid < 0).sum())
    for t, (bins_a, _) in enumerate(ia):
        sel = rec.tid == t
        assert bins_a[37450][1] == [int((sel & ((rec.flag & 4) == 0)).sum()), int((sel & ((rec.flag & 4) != 0)).sum())]
    import subprocess
    from strling_amd import build
    for path in (a, b):
        r = subprocess.run([build.CLI, "_indexed_records", path], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == str(rec.n), (r.stdout, r.stderr)
    os.remove(b + ".bai")
    assert subprocess.run([build.CLI, "_indexed_records", b], capture_output=True, text=True).stdout.strip() == "unknown"


def test_a_slab_regenerates_alone_and_quals_aux_keep_the_fields(tmp_path):
    n_slabs, pairs, seed = 4, 800, 5
    p = str(tmp_path / "q.bam")
    bamio.write_bam_slabs(p, n_slabs, pairs, seed=seed, level=6, quals=True, aux=True, index=False, procs=1)
    raw, _ = _inflate(p)
    full, _ = synth.synth_wgs_30x(n_slabs, pairs, seed=seed, procs=1)
    # walk the records: core fields, names, SEQ as in the merged batch; qualities from the four bins; aux tags parse
    l_text = struct.unpack_from("<i", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]; o += 4 + ln + 4
    i = 0
    while o < len(raw):
        bs = struct.unpack_from("<i", raw, o)[0]
        tid, pos, lrn, mapq, _bin, nc, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", raw, o + 4)
        assert (tid, pos, mapq, flag, l_seq, mtid, mpos) == (int(full.tid[i]), int(full.pos[i]), int(full.mapq[i]), int(full.flag[i]), int(full.l_seq[i]),
                                                             int(full.mtid[i]), int(full.mpos[i]))
        q = o + 36
        assert raw[q:q + lrn - 1] == full.qname(i)
        q += lrn + 4 * nc
        so = int(full.seq_off[i])
        assert raw[q:q + (l_seq + 1) // 2] == bytes(full.seq4[so:so + (l_seq + 1) // 2])
        q += (l_seq + 1) // 2
        assert set(raw[q:q + l_seq]) <= {2, 12, 23, 37}
        q += l_seq
        aux = raw[q:o + 4 + bs]
        assert aux[:3] == b"NMC" and aux[4:7] == b"MDZ" and aux.endswith(b"RGZgrp1\0") and b"ASC" in aux and b"XSC" in aux
        o += 4 + bs
        i += 1
    assert i == full.n
    # slab 2 alone = its records of the merged batch (mapped part on its own contigs, tail names from its own pair ids)
    rec2, g2 = bamio.slab_records(2, n_slabs, pairs, seed)
    sel = np.nonzero((full.tid == 4) | (full.tid == 5))[0]
    m = int((rec2.tid >= 0).sum())
    assert m == sel.size and np.array_equal(rec2.pos[:m], full.pos[sel]) and np.array_equal(rec2.tid[:m], full.tid[sel])
    assert [rec2.qname(k) for k in range(0, m, 97)] == [full.qname(int(sel[k])) for k in range(0, m, 97)]
