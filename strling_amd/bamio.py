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
    meta = [None] * n_ref                 # [first offset, end offset, n_mapped, n_unmapped] (the pseudo-bin 37450 of samtools index)
    no_coor = 0
    for i in range(rec.n):
        t = int(rec.tid[i])
        if t < 0:
            no_coor += 1
            continue
        if meta[t] is None:
            meta[t] = [voff(rec_off[i]), 0, 0, 0]
        meta[t][1] = voff(rec_off[i + 1])
        meta[t][3 if int(rec.flag[i]) & 4 else 2] += 1
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
        out += struct.pack("<i", len(bins[t]) + (meta[t] is not None))
        for b, chunks in sorted(bins[t].items()):
            out += struct.pack("<Ii", b, len(chunks))
            for v0, v1 in chunks:
                out += struct.pack("<QQ", v0, v1)
        if meta[t] is not None:
            out += struct.pack("<IiQQQQ", 37450, 2, *meta[t])
        n_intv = (max(lin[t]) + 1) if lin[t] else 0
        out += struct.pack("<i", n_intv)
        for w in range(n_intv):
            out += struct.pack("<Q", lin[t].get(w, 0))       # empty windows stay 0, as in files written by older tools
    out += struct.pack("<Q", no_coor)                        # n_no_coor, as samtools writes it
    with open(path, "wb") as f:
        f.write(out)


def write_genome_bed(path, genome, targets, unit="AC"):
    with open(path, "w") as f:
        for t in range(genome.n_tid):
            for j in range(int(genome.iv_off[t]), int(genome.iv_off[t + 1])):
                f.write(f"{targets[t][0]}\t{int(genome.iv_start[j])}\t{int(genome.iv_stop[j])}\t{unit}\n")


# ---- whole-genome sized synthetic BAMs (benchmarks, tests of the product at the size BASELINE.json names) ---------------------
# The per-record Python loop of write_bam needs hours for 5 x 10^8 records; here the records of one synthetic SLAB at a time
# (synth.synth_wgs: an independent sub-sample on its own two contigs, `synth.synth_wgs_30x` geometry) are serialised and
# deflated by a small C helper (tools/bamgen/bamgen.c, zlib) in a pool of worker processes, and the parent only writes bytes.
# File layout = what synth.synth_wgs_chunks(n_slabs, pairs_per_slab, seed) would give as ONE batch: the mapped records of
# slab 0, 1, ... on contigs s0chr1, s0chr2, s1chr1, ... and then every slab's unmapped tail -- so any slab can be
# regenerated alone from (seed + slab) and its part of the output checked against the oracle.
_NATIVE = {}


def _bamgen():
    import ctypes
    import os
    import subprocess
    if "lib" in _NATIVE:
        return _NATIVE["lib"]
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bamgen")
    so, src = os.path.join(d, "libbamgen.so"), os.path.join(d, "bamgen.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", tmp, src, "-lz"])
        os.replace(tmp, so)
    lib = ctypes.CDLL(so)
    P = ctypes.c_void_p
    lib.bamgen_serialize.restype = ctypes.c_int64
    lib.bamgen_serialize.argtypes = [ctypes.c_int64] + [P] * 14 + [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, P, ctypes.c_int64, P, P, P]
    lib.bamgen_bgzf.restype = ctypes.c_int64
    lib.bamgen_bgzf.argtypes = [P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, P, ctypes.c_int64, P, ctypes.c_int64]
    _NATIVE["lib"] = lib
    return lib


def serialize_records(rec, quals=False, aux=False, seed=0):
    """-> (raw BAM record bytes as uint8 array, rec_off uint64[n + 1], ref_end int32[n], bin uint32[n])"""
    lib = _bamgen()
    n = rec.n
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    tid, pos, mtid, mpos = c(rec.tid, np.int32), c(rec.pos, np.int32), c(rec.mtid, np.int32), c(rec.mpos, np.int32)
    flag, mapq = c(rec.flag, np.uint16), c(rec.mapq, np.uint8)
    isize = c(rec.isize if rec.isize is not None else np.zeros(n, np.int32), np.int32)
    coff, cig = c(rec.cigar_off, np.uint32), c(rec.cigar, np.uint32)
    soff, lseq, seq4 = c(rec.seq_off, np.uint64), c(rec.l_seq, np.int32), c(rec.seq4, np.uint8)
    qoff = c(rec.qname_off, np.uint64)
    qn = np.frombuffer(bytes(rec.qnames) + b"\0", np.uint8)
    cap = int(n * (36 + 40 + 4) + int(qoff[-1]) + n + 4 * cig.size + int(((lseq.astype(np.int64) + 1) // 2 + lseq).sum()) + 64)
    out = np.empty(cap, np.uint8)
    rec_off = np.empty(n + 1, np.uint64)
    ref_end = np.empty(n, np.int32)
    bins = np.empty(n, np.uint32)
    p = lambda a: a.ctypes.data
    got = lib.bamgen_serialize(n, p(tid), p(pos), p(mtid), p(mpos), p(flag), p(mapq), p(isize), p(coff), p(cig), p(soff), p(lseq), p(seq4),
                               p(qoff), p(qn), int(bool(quals)), int(bool(aux)), int(seed), p(out), cap, p(rec_off), p(ref_end), p(bins))
    if got < 0:
        raise RuntimeError("bamgen_serialize: buffer too small")
    return out[:got], rec_off, ref_end, bins


def bgzf_compress(raw, level=6, block=0xFF00):
    """raw uint8 array -> (BGZF bytes as uint8 array, csize uint32[n_blocks])"""
    lib = _bamgen()
    raw = np.ascontiguousarray(raw, np.uint8)
    nb = (raw.size + block - 1) // block
    cap = int(raw.size + raw.size // 500 + nb * 64 + 1024)
    out = np.empty(cap, np.uint8)
    csize = np.empty(max(nb, 1), np.uint32)
    got = lib.bamgen_bgzf(raw.ctypes.data, raw.size, int(level), int(block), out.ctypes.data, cap, csize.ctypes.data, csize.size)
    if got < 0:
        raise RuntimeError(f"bamgen_bgzf failed ({got})")
    return out[:got], csize[:nb]


def slab_geometry(n_slabs, pairs_per_slab, read_len=150, coverage=30.0, loci=3200):
    """contig length / hot loci of a slab, exactly synth.synth_wgs_30x's"""
    contig_len = max(20_000, int(2 * pairs_per_slab * read_len / coverage / 2))
    return contig_len, max(1, loci // (2 * n_slabs))


def slab_records(c, n_slabs, pairs_per_slab, seed, **kw):
    """(RecordBatch, GenomeStr) of slab c as it sits in the file: tids shifted to the slab's own two contigs, targets = the
    whole file's; mapped records first, its unmapped tail behind them"""
    from . import synth
    contig_len, hot = slab_geometry(n_slabs, pairs_per_slab)
    rec, g = synth.synth_wgs(pairs_per_slab, seed=seed + c, pair_id_base=c * pairs_per_slab, n_contigs=2, contig_len=contig_len, hot_loci=hot, **kw)
    sh = 2 * c
    rec.tid = np.where(rec.tid >= 0, rec.tid + sh, rec.tid).astype(np.int32)
    rec.mtid = np.where(rec.mtid >= 0, rec.mtid + sh, rec.mtid).astype(np.int32)
    rec.targets = [(f"s{k // 2}chr{k % 2 + 1}", contig_len) for k in range(2 * n_slabs)]
    # ... and the genome table of the whole file's contigs, holding this slab's intervals at its own two
    from .records import GenomeStr
    off = np.zeros(2 * n_slabs + 1, np.int64)
    off[sh + 1:] = g.iv_off[1]
    off[sh + 2:] = g.iv_off[2]
    has = np.zeros(2 * n_slabs, np.uint8)
    has[sh:sh + 2] = g.has_chrom
    return rec, GenomeStr(2 * n_slabs, has, off, g.iv_start, g.iv_stop)


def _slab_worker(a):
    c, n_slabs, pairs, seed, level, block, quals, aux, want_index = a
    rec, g = slab_records(c, n_slabs, pairs, seed)
    m = int((rec.tid >= 0).sum())
    raw, rec_off, ref_end, bins = serialize_records(rec, quals, aux, seed=seed * 1000003 + c)
    cut = int(rec_off[m])
    comp, csize = bgzf_compress(raw[:cut], level, block)
    idx = None
    if want_index:
        # BAI pieces with offsets into the slab's own uncompressed stream (the parent turns them into virtual offsets)
        idx = []
        tid, pos = rec.tid[:m], np.maximum(rec.pos[:m], 0).astype(np.int64)
        for t in (2 * c, 2 * c + 1):
            sel = np.nonzero(tid == t)[0]
            if not sel.size:
                idx.append(None)
                continue
            a0, a1 = int(sel[0]), int(sel[-1]) + 1                     # coordinate sorted: one contiguous run
            b = bins[a0:a1].astype(np.int64)
            order = np.argsort(b, kind="stable")
            sb, si = b[order], order + a0
            new = np.ones(sb.size, bool)
            new[1:] = (sb[1:] != sb[:-1]) | (si[1:] != si[:-1] + 1)
            starts = np.nonzero(new)[0]
            ends = np.append(starts[1:], sb.size) - 1
            chunk_bin, chunk_beg, chunk_end = sb[starts], rec_off[si[starts]], rec_off[si[ends] + 1]
            # linear index: smallest offset of a record overlapping each 16 KiB window
            w0, w1 = pos[a0:a1] >> 14, (np.maximum(ref_end[a0:a1].astype(np.int64), pos[a0:a1] + 1) - 1) >> 14
            n_w = int(w1.max()) + 1
            lin = np.full(n_w, np.iinfo(np.uint64).max, np.uint64)
            ro = rec_off[a0:a1]
            for d in range(int((w1 - w0).max()) + 1):                  # a read spans one or two windows
                ok = w0 + d <= w1
                np.minimum.at(lin, (w0 + d)[ok], ro[ok])
            n_unm = int(((rec.flag[a0:a1].astype(np.int64) & 4) != 0).sum())
            idx.append((chunk_bin, chunk_beg.astype(np.uint64), chunk_end.astype(np.uint64), lin, (int(rec_off[a0]), int(rec_off[a1]), a1 - a0 - n_unm, n_unm)))
    tail = raw[cut:].tobytes()
    genome = [(int(s), int(e)) for s, e in zip(g.iv_start, g.iv_stop)], [int(x) - int(g.iv_off[2 * c]) for x in g.iv_off[2 * c:2 * c + 3]]
    return c, comp.tobytes(), csize, idx, tail, rec.n, m, genome


def _tail_worker(a):
    raw, level, block = a
    comp, csize = bgzf_compress(np.frombuffer(raw, np.uint8), level, block)
    return comp.tobytes()


def write_bam_slabs(path, n_slabs, pairs_per_slab, seed=1234, level=6, quals=True, aux=True, index=True, bed=None, procs=None, block=0xFF00,
                    progress=None):
    """A coordinate-sorted BAM of n_slabs * 2 * pairs_per_slab DISTINCT synthetic reads (30x geometry), its .bai and the
    matching ref.fasta.str BED, written slab by slab by a pool of processes.  Returns dict(reads, bytes, targets, seconds)."""
    import multiprocessing as mp
    import os
    import time
    _bamgen()                       # compile once, before the workers fork
    t0 = time.time()
    contig_len, _ = slab_geometry(n_slabs, pairs_per_slab)
    targets = [(f"s{k // 2}chr{k % 2 + 1}", contig_len) for k in range(2 * n_slabs)]
    text = sam_header(targets).encode()
    hdr = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(targets)))
    for name, length in targets:
        nb = name.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", length)
    procs = procs or max(1, min(n_slabs, int(_cpu_budget())))
    jobs = [(c, n_slabs, pairs_per_slab, seed, level, block, quals, aux, index) for c in range(n_slabs)]
    tails, n_reads, n_placed = [], 0, 0
    bai = [None] * (2 * n_slabs)
    bedf = open(bed, "w") if bed else None
    with open(path, "wb") as f:
        hc, _ = bgzf_compress(np.frombuffer(bytes(hdr), np.uint8), level, block)
        f.write(hc.tobytes())
        pool = mp.get_context("fork").Pool(procs) if procs > 1 and n_slabs > 1 else None
        it = pool.imap(_slab_worker, jobs, chunksize=1) if pool else map(_slab_worker, jobs)
        for c, comp, csize, idx, tail, n, m, genome in it:
            base = f.tell()
            f.write(comp)
            n_reads += n
            tails.append(tail)
            if index:
                boff = np.zeros(csize.size + 1, np.uint64)
                boff[1:] = np.cumsum(csize.astype(np.uint64))
                boff += np.uint64(base)

                def voff(u):
                    k = (u // np.uint64(block)).astype(np.int64)
                    return (boff[k] << np.uint64(16)) | (u - k.astype(np.uint64) * np.uint64(block))
                for j, part in enumerate(idx):
                    if part is None:
                        continue
                    cb, beg, end, lin, (m_beg, m_end, n_map, n_unm) = part
                    # an end offset on a block border belongs to the NEXT block (offset 0), like htslib's bgzf_tell
                    have = lin != np.iinfo(np.uint64).max
                    lv = np.zeros(lin.size, np.uint64)
                    lv[have] = voff(lin[have])
                    mv = voff(np.array([m_beg, m_end], np.uint64))
                    bai[2 * c + j] = (cb, voff(beg), voff(end), lv, (int(mv[0]), int(mv[1]), n_map, n_unm))
                    n_placed += n_map + n_unm
            if bedf:
                ivs, off = genome
                for t in range(2):
                    for k in range(off[t], off[t + 1]):
                        bedf.write(f"{targets[2 * c + t][0]}\t{ivs[k][0]}\t{ivs[k][1]}\tAC\n")
            if progress:
                progress(c + 1, n_slabs, time.time() - t0)
        # the unmapped tails of all slabs, in slab order, behind the last mapped record
        tail_all = b"".join(tails)
        del tails
        piece = 64 * block
        tjobs = [(tail_all[o:o + piece], level, block) for o in range(0, len(tail_all), piece)]
        for comp in (pool.imap(_tail_worker, tjobs, chunksize=1) if pool else map(_tail_worker, tjobs)):
            f.write(comp)
        if pool:
            pool.close()
            pool.join()
        f.write(_EOF)
        total = f.tell()
    if bedf:
        bedf.close()
    if index:
        out = bytearray(b"BAI\1" + struct.pack("<i", len(targets)))
        for t in range(len(targets)):
            if bai[t] is None:
                out += struct.pack("<ii", 0, 0)
                continue
            cb, beg, end, lv, meta = bai[t]
            ub, first = np.unique(cb, return_index=True)          # chunks of a bin are contiguous in cb (sorted by bin)
            counts = np.diff(np.append(first, cb.size))
            out += struct.pack("<i", ub.size + 1)
            for b, o, k in zip(ub.tolist(), first.tolist(), counts.tolist()):
                out += struct.pack("<Ii", b, k)
                out += np.stack([beg[o:o + k], end[o:o + k]], axis=1).astype("<u8").tobytes()
            out += struct.pack("<IiQQQQ", 37450, 2, *meta)        # the metadata pseudo-bin of `samtools index`: file span, mapped / unmapped counts
            out += struct.pack("<i", lv.size) + lv.astype("<u8").tobytes()
        out += struct.pack("<Q", n_reads - n_placed)              # n_no_coor
        with open(path + ".bai", "wb") as f:
            f.write(out)
    return {"reads": n_reads, "bytes": total, "targets": targets, "seconds": time.time() - t0, "procs": procs}


def _cpu_budget():
    import os
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1.0, int(q) / int(per))
    except Exception:
        pass
    return float(os.cpu_count() or 1)
