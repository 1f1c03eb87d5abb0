"""Synthetic BAM / BED writers for tests and examples (pure Python + zlib; htslib/samtools are not available).

Writes spec-conformant BGZF-compressed BAM (SAM spec sections 4.1-4.2) from a records.RecordBatch so that the
`strling` CLI's own BGZF/BAM reader can be exercised end to end, plus the matching .bai (SAM spec 5.2: binning
index + 16 KiB linear index) that `strling call` needs for its region reads.
"""
import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data, level=1):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def sam_header(targets, sort_order="coordinate"):
    return f"@HD\tVN:1.6\tSO:{sort_order}\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in targets)


def write_bam(path, rec, header_text=None, level=1, block=0xFF00, index=True, repeat=1):
    """repeat > 1 (benchmarks only): the header gets its own BGZF blocks and the record blocks are written `repeat` times;
    such a file is not coordinate sorted and gets no index."""
    targets = rec.targets
    text = (header_text if header_text is not None else sam_header(targets)).encode()
    out = bytearray()
    out += b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(targets))
    for name, length in targets:
        nb = name.encode() + b"\0"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", length)
    isize = rec.isize if rec.isize is not None else np.zeros(rec.n, np.int32)
    rec_off = []
    for i in range(rec.n):
        rec_off.append(len(out))
        qn = rec.qname(i) + b"\0"
        c0, c1 = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
        l_seq = int(rec.l_seq[i])
        so = int(rec.seq_off[i])
        seq = bytes(rec.seq4[so:so + (l_seq + 1) // 2])
        if l_seq & 1 and seq:
            seq = seq[:-1] + bytes([seq[-1] & 0xF0])
        cig = rec.cigar[c0:c1].astype("<u4").tobytes()
        qual = b"\xff" * l_seq
        body = struct.pack("<iiBBHHHiiii", int(rec.tid[i]), int(rec.pos[i]), len(qn), int(rec.mapq[i]), 4680, c1 - c0,
                           int(rec.flag[i]), l_seq, int(rec.mtid[i]), int(rec.mpos[i]), int(isize[i])) + qn + cig + seq + qual
        out += struct.pack("<i", len(body)) + body
    rec_off.append(len(out))
    if repeat > 1:
        n_hdr = rec_off[0]
        with open(path, "wb") as f:
            for o in range(0, n_hdr, block):
                f.write(_bgzf_block(bytes(out[o:min(n_hdr, o + block)]), level))
            body = b"".join(_bgzf_block(bytes(out[o:o + block]), level) for o in range(n_hdr, len(out), block))
            for _ in range(repeat):
                f.write(body)
            f.write(_EOF)
        return text.decode()
    block_off = []
    with open(path, "wb") as f:
        for o in range(0, len(out), block):
            block_off.append(f.tell())
            f.write(_bgzf_block(bytes(out[o:o + block]), level))
        block_off.append(f.tell())
        f.write(_EOF)
    if index:
        write_bai(path + ".bai", rec, rec_off, block_off, block)
    return text.decode()


def _raw_records(rec, a, b):
    isize = rec.isize if rec.isize is not None else np.zeros(rec.n, np.int32)
    out = bytearray()
    qn_all = bytes(rec.qnames)
    for i in range(a, b):
        qn = qn_all[int(rec.qname_off[i]):int(rec.qname_off[i + 1])] + b"\0"
        c0, c1 = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
        l_seq = int(rec.l_seq[i])
        so = int(rec.seq_off[i])
        seq = bytes(rec.seq4[so:so + (l_seq + 1) // 2])
        if l_seq & 1 and seq:
            seq = seq[:-1] + bytes([seq[-1] & 0xF0])
        body = struct.pack("<iiBBHHHiiii", int(rec.tid[i]), int(rec.pos[i]), len(qn), int(rec.mapq[i]), 4680, c1 - c0,
                           int(rec.flag[i]), l_seq, int(rec.mtid[i]), int(rec.mpos[i]), int(isize[i])) + qn + \
            rec.cigar[c0:c1].astype("<u4").tobytes() + seq + b"\xff" * l_seq
        out += struct.pack("<i", len(body)) + body
    return out


_PAR = {}


def _par_worker(rng):
    a, b = rng
    raw = _raw_records(_PAR["rec"], a, b)
    blk, lvl = _PAR["block"], _PAR["level"]
    return b"".join(_bgzf_block(bytes(raw[o:o + blk]), lvl) for o in range(0, len(raw), blk))


def write_bam_parallel(path, rec, level=1, block=0xFF00, procs=None, records_per_task=1 << 16):
    """Benchmark inputs: the same BAM bytes as write_bam would produce record for record, but built and compressed by a pool
    of forked workers (record ranges are compressed independently, so BGZF block boundaries differ) and without an index.
    Call before the GPU runtime is initialised in this process."""
    import multiprocessing as mp
    import os
    text = sam_header(rec.targets).encode()
    hdr = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(rec.targets)))
    for name, length in rec.targets:
        nb = name.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", length)
    tasks = [(a, min(rec.n, a + records_per_task)) for a in range(0, rec.n, records_per_task)]
    _PAR.update(rec=rec, block=block, level=level)
    procs = procs or max(1, (os.cpu_count() or 2) // 2)
    with open(path, "wb") as f:
        for o in range(0, len(hdr), block):
            f.write(_bgzf_block(bytes(hdr[o:o + block]), level))
        if procs == 1 or len(tasks) == 1:
            for t in tasks:
                f.write(_par_worker(t))
        else:
            with mp.get_context("fork").Pool(min(procs, len(tasks))) as pool:
                for part in pool.imap(_par_worker, tasks, chunksize=1):
                    f.write(part)
                pool.close()
                pool.join()
        f.write(_EOF)
    _PAR.clear()
    return text.decode()


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _ref_len(rec, i):
    if int(rec.flag[i]) & 4:
        return 1
    c0, c1 = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
    rl = sum(int(c) >> 4 for c in rec.cigar[c0:c1] if (int(c) & 15) in (0, 2, 3, 7, 8))
    return rl or 1


def write_bai(path, rec, rec_off, block_off, block):
    """rec_off[i] = offset of record i in the uncompressed stream (rec_off[n] = its end); block_off[k] = file offset of
    BGZF block k (uncompressed stream cut every `block` bytes)"""
    def voff(u):
        k = u // block                       # block_off has one entry more than there are data blocks (the EOF block)
        return (block_off[k] << 16) | (u - k * block)
    n_ref = len(rec.targets)
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    for i in range(rec.n):
        t = int(rec.tid[i])
        if t < 0:
            continue
        beg = int(rec.pos[i])
        end = beg + _ref_len(rec, i)
        v0, v1 = voff(rec_off[i]), voff(rec_off[i + 1])
        ch = bins[t].setdefault(_reg2bin(beg, end), [])
        if ch and ch[-1][1] == v0:
            ch[-1][1] = v1
        else:
            ch.append([v0, v1])
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            lin[t].setdefault(w, v0)
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        out += struct.pack("<i", len(bins[t]))
        for b, chunks in sorted(bins[t].items()):
            out += struct.pack("<Ii", b, len(chunks))
            for v0, v1 in chunks:
                out += struct.pack("<QQ", v0, v1)
        n_intv = (max(lin[t]) + 1) if lin[t] else 0
        out += struct.pack("<i", n_intv)
        for w in range(n_intv):
            out += struct.pack("<Q", lin[t].get(w, 0))       # empty windows stay 0, as in files written by older tools
    with open(path, "wb") as f:
        f.write(out)


def write_genome_bed(path, genome, targets, unit="AC"):
    with open(path, "w") as f:
        for t in range(genome.n_tid):
            for j in range(int(genome.iv_off[t]), int(genome.iv_off[t + 1])):
                f.write(f"{targets[t][0]}\t{int(genome.iv_start[j])}\t{int(genome.iv_stop[j])}\t{unit}\n")
