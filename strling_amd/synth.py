"""Synthetic inputs: S1 = 30x 150 bp paired-end WGS-like record batches with the
read-class mix the extract hot path sees, S2 = multi-sample tread sets for the clustering path.
All randomness comes from numpy's counter-based Philox generator, so the CPU oracle and the GPU
path regenerate identical inputs from (seed, sizes).  No reference code or data is involved.
"""
import numpy as np

from .records import RecordBatch, GenomeStr, pack_codes4, CIGAR_OPS

NIB = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
_ACGT_NIB = np.array([1, 2, 4, 8], np.uint8)
OP = {c: i for i, c in enumerate(CIGAR_OPS)}


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def synth_genome_str(rng, n_contigs, contig_len, overlap_frac=0.03, read_len=150):
    """Reference STR intervals (what `strling index` writes to ref.fasta.str): lengths 30-300 bp, density
    chosen so that ~overlap_frac of 150 bp reads overlap one."""
    mean_len = 165.0
    dens = overlap_frac / (read_len + mean_len)
    per = int(contig_len * dens)
    ivs = {}
    for t in range(n_contigs):
        st = np.sort(rng.integers(0, contig_len - 400, size=per))
        ln = rng.integers(30, 301, size=per)
        ivs[t] = list(zip(st.tolist(), (st + ln).tolist()))
    return GenomeStr.from_lists(n_contigs, ivs)


def synth_wgs(n_pairs, seed=1234, read_len=150, n_contigs=25, contig_len=10_000_000, str_frac=0.01, soft_frac=0.03,
              indel_frac=0.01, unmapped_frac=0.005, interchrom_frac=0.01, genome_overlap=0.03, with_qnames=True, hot_loci=2,
              pair_id_base=0):
    """-> (RecordBatch sorted like a coordinate-sorted BAM with the unmapped tail last, GenomeStr)."""
    rng = _rng(seed)
    L = read_len
    n = 2 * n_pairs
    g = synth_genome_str(rng, n_contigs, contig_len, genome_overlap, L)
    # ---- pair placement ----
    tid1 = rng.integers(0, n_contigs, size=n_pairs).astype(np.int32)
    frag = np.clip(np.rint(rng.normal(350, 120, size=n_pairs)), L, 4095).astype(np.int32)
    pos1 = rng.integers(0, contig_len - 5000, size=n_pairs).astype(np.int32)
    pos2 = pos1 + frag - L
    tid2 = tid1.copy()
    u = rng.random(n_pairs)
    inter = u < interchrom_frac
    tid2[inter] = (tid1[inter] + 1 + rng.integers(0, n_contigs - 1, size=int(inter.sum()))) % n_contigs
    pos2[inter] = rng.integers(0, contig_len - 5000, size=int(inter.sum()))
    unm = (u >= interchrom_frac) & (u < interchrom_frac + unmapped_frac)
    proper = ~inter & ~unm & (rng.random(n_pairs) < 0.99)
    # per read arrays, read 2i = first in pair, 2i+1 = second
    tid = np.empty(n, np.int32); pos = np.empty(n, np.int32)
    tid[0::2], tid[1::2] = tid1, tid2
    pos[0::2], pos[1::2] = pos1, pos2
    isize = np.zeros(n, np.int32)
    isize[0::2] = np.where(proper, frag, 0)
    isize[1::2] = np.where(proper, -frag, 0)
    flag = np.empty(n, np.uint16)
    flag[0::2] = 0x1 | 0x40 | 0x20 | np.where(proper, 0x2, 0)
    flag[1::2] = 0x1 | 0x80 | 0x10 | np.where(proper, 0x2, 0)
    um = np.repeat(unm, 2)
    mq_u = rng.random(n)
    mapq = np.where(mq_u < 0.90, 60, np.where(mq_u < 0.95, 0, rng.integers(1, 60, size=n))).astype(np.uint8)
    # ---- read classes ----
    cls_u = rng.random(n)
    is_str = cls_u < str_frac
    is_str |= np.repeat(unm & (rng.random(n_pairs) < 0.3), 2)   # both-unmapped STR pairs (the tail extract visits twice)
    is_soft = (cls_u >= str_frac) & (cls_u < str_frac + soft_frac) & ~um
    is_indel = (cls_u >= str_frac + soft_frac) & (cls_u < str_frac + soft_frac + indel_frac) & ~um
    # STR-rich reads as an aligner leaves them: 40% sit on a reference STR (so the skip predicate keeps them),
    # 30% are partly soft-clipped, 30% are unmapped but placed at their mate (flag 0x4, no cigar).
    sub = rng.random(n)
    str_on_ref = is_str & ~um & (sub < 0.4)
    str_clip = is_str & ~um & (sub >= 0.4) & (sub < 0.7)
    str_unmapped = is_str & ~um & (sub >= 0.7)
    partner = np.arange(n) ^ 1
    str_unmapped &= ~str_unmapped[partner] | (np.arange(n) % 2 == 0)   # at most one mate of a pair is placed-unmapped
    str_unmapped &= ~(str_unmapped[partner] & (np.arange(n) % 2 == 1))
    # Expanded loci: a few reference STR intervals per contig attract the STR-rich reads, so that they pile up into
    # clusters the way reads around a real expansion do.  Every locus has its own repeat unit.  Reads of the
    # "on_ref" kind lie on the locus (150M, kept by the skip predicate); reads of the "clip" kind straddle one of its
    # boundaries and are soft-clipped exactly there (xMyS ending at iv.start, or ySxM starting at iv.stop).
    H = hot_loci
    n_iv = (g.iv_off[1:] - g.iv_off[:-1]).astype(np.int64)
    hot_iv = g.iv_off[:-1, None] + (rng.random((n_contigs, H)) * n_iv[:, None]).astype(np.int64)
    hot_k = rng.integers(2, 7, size=(n_contigs, H))
    hot_unit = _ACGT_NIB[rng.integers(0, 4, size=(n_contigs, H, 6))]
    loc = rng.integers(0, H, size=n)
    tsafe = np.clip(tid, 0, n_contigs - 1)
    liv = hot_iv[tsafe, loc]
    r_idx = np.nonzero(str_on_ref)[0]
    pos[r_idx] = np.maximum(0, g.iv_start[liv[r_idx]] - rng.integers(0, 100, size=r_idx.size)).astype(np.int32)
    c_idx = np.nonzero(str_clip)[0]
    loc_clip = rng.integers(30, 101, size=n)
    loc_right = rng.random(n) < 0.5           # True: xMyS (clip on the right, read ends inside the repeat)
    pos[c_idx] = np.where(loc_right[c_idx], g.iv_start[liv[c_idx]] - (L - loc_clip[c_idx]), g.iv_stop[liv[c_idx]]).astype(np.int32)
    pos[c_idx] = np.maximum(pos[c_idx], 0)
    mapq[c_idx] = 60
    mapq[r_idx] = np.where(rng.random(r_idx.size) < 0.7, 60, mapq[r_idx])
    # the mate follows its read to the locus (same fragment length, same orientation)
    mv = np.concatenate([r_idx, c_idx])
    mv = mv[(tid[mv] == tid[partner[mv]]) & ~um[mv]]
    fr = np.repeat(frag, 2)[mv] - L
    pos[partner[mv]] = np.maximum(0, np.where(mv % 2 == 0, pos[mv] + fr, pos[mv] - fr)).astype(np.int32)
    u_idx = np.nonzero(str_unmapped)[0]
    tid[u_idx] = tid[partner[u_idx]]
    pos[u_idx] = pos[partner[u_idx]]
    flag[u_idx] = (flag[u_idx] & ~np.uint16(0x2)) | 0x4
    flag[partner[u_idx]] = (flag[partner[u_idx]] & ~np.uint16(0x2)) | 0x8
    mapq[u_idx] = 0
    nocig = um | str_unmapped
    tid[um] = -1; pos[um] = -1
    flag[um] = (flag[um] & ~np.uint16(0x32)) | 0x4 | 0x8
    mapq[um] = 0
    mtid = tid[partner].copy()
    mpos = pos[partner].copy()
    # ---- sequences (nibble codes) ----
    codes = _ACGT_NIB[rng.integers(0, 4, size=(n, L))]
    idx = np.nonzero(is_str)[0]
    if idx.size:
        k = rng.integers(1, 7, size=idx.size)
        unit = _ACGT_NIB[rng.integers(0, 4, size=(idx.size, 6))]
        phase = rng.integers(0, 6, size=idx.size)
        j = (np.arange(L)[None, :] + phase[:, None]) % k[:, None]
        rep = np.take_along_axis(unit, j, axis=1)
        purity = rng.choice(np.array([1.0, 0.97, 0.93, 0.9, 0.85]), size=idx.size)
        keep = rng.random((idx.size, L)) < purity[:, None]
        codes[idx] = np.where(keep, rep, codes[idx])
    # reads at expanded loci carry the locus' unit: all of the read (on_ref) or its clipped part (clip)
    lidx = np.nonzero(str_on_ref | str_clip)[0]
    if lidx.size:
        k = hot_k[tsafe[lidx], loc[lidx]]
        unit = hot_unit[tsafe[lidx], loc[lidx]]
        phase = rng.integers(0, 6, size=lidx.size)
        j = (np.arange(L)[None, :] + phase[:, None]) % k[:, None]
        rep = np.take_along_axis(unit, j, axis=1)
        purity = rng.choice(np.array([1.0, 0.98, 0.95, 0.92]), size=lidx.size)
        keep = rng.random((lidx.size, L)) < purity[:, None]
        col = np.arange(L)[None, :]
        inrep = np.where(str_clip[lidx][:, None],
                         np.where(loc_right[lidx][:, None], col >= (L - loc_clip[lidx])[:, None], col < loc_clip[lidx][:, None]), True)
        flank = _ACGT_NIB[rng.integers(0, 4, size=(lidx.size, L))]
        codes[lidx] = np.where(inrep, np.where(keep, rep, flank), flank)
    # soft clips: clip length 1-100 on one end (10% both); 40% of clipped tails are repeats
    clip_l = np.zeros(n, np.int32); clip_r = np.zeros(n, np.int32)
    clip_r[c_idx] = np.where(loc_right[c_idx], loc_clip[c_idx], 0)
    clip_l[c_idx] = np.where(loc_right[c_idx], 0, loc_clip[c_idx])
    sidx = np.nonzero(is_soft)[0]
    if sidx.size:
        both = rng.random(sidx.size) < 0.10
        left = rng.random(sidx.size) < 0.5
        cl = rng.integers(1, 101, size=sidx.size)
        cr = rng.integers(1, min(101, L - 100), size=sidx.size)
        clip_l[sidx] = np.where(both | left, cl, 0)
        clip_r[sidx] = np.where(both | ~left, np.where(both, cr, cl), 0)
        rep_tail = rng.random(sidx.size) < 0.4
        k = rng.integers(2, 7, size=sidx.size)
        unit = _ACGT_NIB[rng.integers(0, 4, size=(sidx.size, 6))]
        j = np.arange(L)[None, :] % k[:, None]
        rep = np.take_along_axis(unit, j, axis=1)
        col = np.arange(L)[None, :]
        in_clip = (col < clip_l[sidx][:, None]) | (col >= (L - clip_r[sidx])[:, None])
        codes[sidx] = np.where(in_clip & rep_tail[:, None], rep, codes[sidx])
        mapq[sidx] = np.where(rng.random(sidx.size) < 0.8, 60, mapq[sidx])
    is_soft = is_soft | str_clip
    # N bases: 0.1% of bases, and 0.2% of reads with > 20 N
    nmask = rng.random((n, L)) < 0.001
    many = np.nonzero(rng.random(n) < 0.002)[0]
    if many.size:
        nmask[many[:, None], rng.integers(0, L, size=(many.size, 40))] = True
    codes[nmask] = 15
    l_seq = np.full(n, L, np.int32)
    # ---- cigars ----
    ncig = np.ones(n, np.int64)
    ncig[nocig] = 0
    ncig[is_soft] = 1 + (clip_l[is_soft] > 0) + (clip_r[is_soft] > 0)
    ncig[is_indel] = 3
    cig_off = np.zeros(n + 1, np.uint32)
    cig_off[1:] = np.cumsum(ncig)
    cigar = np.zeros(int(cig_off[-1]), np.uint32)
    is_soft &= ~nocig
    is_indel &= ~nocig & ~is_soft
    plain = ~nocig & ~is_soft & ~is_indel
    sidx = np.nonzero(is_soft)[0]          # includes the reads clipped at an expanded locus
    cigar[cig_off[:-1][plain]] = (L << 4) | OP["M"]
    ii = np.nonzero(is_indel)[0]
    if ii.size:
        ins = rng.random(ii.size) < 0.5
        a = rng.integers(20, L - 40, size=ii.size)
        d = rng.integers(1, 6, size=ii.size)
        o = cig_off[:-1][ii]
        cigar[o] = (a << 4) | OP["M"]
        cigar[o + 1] = (d << 4) | np.where(ins, OP["I"], OP["D"])
        cigar[o + 2] = ((L - a - np.where(ins, d, 0)) << 4) | OP["M"]
    if sidx.size:
        o = cig_off[:-1][sidx].astype(np.int64)
        hasl = clip_l[sidx] > 0
        cigar[o[hasl]] = (clip_l[sidx][hasl] << 4) | OP["S"]
        o2 = o + hasl
        cigar[o2] = ((L - clip_l[sidx] - clip_r[sidx]) << 4) | OP["M"]
        hasr = clip_r[sidx] > 0
        cigar[(o2 + 1)[hasr]] = (clip_r[sidx][hasr] << 4) | OP["S"]
    # ---- coordinate sort (unmapped last), keep mate order for equal keys ----
    key = np.where(tid < 0, np.int64(1) << 40, tid.astype(np.int64) << 32) + np.maximum(pos, 0).astype(np.int64)
    order = np.argsort(key, kind="stable")
    pair_id = (np.arange(n) // 2)[order]
    new_off = np.zeros(n + 1, np.uint32)
    new_off[1:] = np.cumsum(ncig[order])
    src = np.repeat(cig_off[:-1][order].astype(np.int64), ncig[order]) + (np.arange(int(new_off[-1])) - np.repeat(new_off[:-1].astype(np.int64), ncig[order]))
    cigar = cigar[src]
    seq4, seq_off = pack_codes4(codes[order], l_seq[order])
    if with_qnames:
        names = np.char.add("q", (pair_id + pair_id_base).astype(str)).astype("S")
        qlen = np.char.str_len(names)
        qoff = np.zeros(n + 1, np.uint64)
        qoff[1:] = np.cumsum(qlen)
        qnames = b"".join(names.tolist())
    else:
        qoff = np.zeros(n + 1, np.uint64)
        qnames = b""
    rec = RecordBatch(tid[order], pos[order], mtid[order], mpos[order], flag[order], mapq[order], new_off, cigar, seq_off,
                      l_seq[order], seq4, qoff, qnames, isize[order],
                      [(f"chr{i + 1}", contig_len) for i in range(n_contigs)])
    return rec, g


def _chunk_worker(a):
    pairs, seed, base, kw = a
    rec, g = synth_wgs(pairs, seed=seed, pair_id_base=base, **kw)
    m = int((rec.tid >= 0).sum())                    # mapped records come first (coordinate order), the unmapped tail last
    stride = int(rec.seq_off[1] - rec.seq_off[0]) if rec.n > 1 else 16
    rows = rec.seq4[: rec.n * stride].reshape(rec.n, stride)
    ncig = np.diff(rec.cigar_off.astype(np.int64))
    qlen = np.diff(rec.qname_off.astype(np.int64))
    c0, q0 = int(rec.cigar_off[m]), int(rec.qname_off[m])
    qn = bytes(rec.qnames)
    parts = []
    for sl, cg, qb in ((slice(0, m), rec.cigar[:c0], qn[:q0]), (slice(m, rec.n), rec.cigar[c0:], qn[q0:])):
        parts.append(dict(tid=rec.tid[sl], pos=rec.pos[sl], mtid=rec.mtid[sl], mpos=rec.mpos[sl], flag=rec.flag[sl], mapq=rec.mapq[sl],
                          ncig=ncig[sl], cigar=cg, rows=rows[sl], l_seq=rec.l_seq[sl], qlen=qlen[sl], qnames=qb, isize=rec.isize[sl]))
    return parts, g, rec.targets


def synth_wgs_chunks(n_chunks, pairs_per_chunk, seed=1234, procs=None, **kw):
    """A large S1 batch of DISTINCT reads: n_chunks independent synth_wgs samples (seed + chunk), each on its own set of
    contigs, generated by a pool of worker processes and merged into one coordinate-sorted batch with one unmapped tail.
    Deterministic in (n_chunks, pairs_per_chunk, seed).  The workers are forked and only run numpy (they never touch the GPU
    runtime the parent may have initialised)."""
    import multiprocessing as mp
    import os
    jobs = [(pairs_per_chunk, seed + c, c * pairs_per_chunk, kw) for c in range(n_chunks)]
    procs = procs or min(n_chunks, max(1, (os.cpu_count() or 2) // 2))
    if n_chunks == 1 or procs == 1:
        res = [_chunk_worker(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_chunk_worker, jobs, chunksize=1)
            pool.close()
            pool.join()        # the workers are gone before the caller goes on (nothing of theirs runs beside a timed region)
    n_contigs = res[0][1].n_tid
    pieces = [r[0][0] for r in res] + [r[0][1] for r in res]          # every chunk's mapped part, then every tail
    shift = [c * n_contigs for c in range(n_chunks)] * 2

    def cat(key, dtype=None):
        a = np.concatenate([p[key] for p in pieces])
        return a.astype(dtype) if dtype is not None else a

    def cat_tid(key):
        return np.concatenate([np.where(p[key] >= 0, p[key] + sh, p[key]).astype(np.int32) for p, sh in zip(pieces, shift)])

    n = sum(p["tid"].size for p in pieces)
    cig_off = np.zeros(n + 1, np.uint32)
    cig_off[1:] = np.cumsum(cat("ncig"))
    qoff = np.zeros(n + 1, np.uint64)
    qoff[1:] = np.cumsum(cat("qlen"))
    rows = np.concatenate([p["rows"] for p in pieces])
    stride = rows.shape[1]
    seq4 = np.zeros(n * stride + 32, np.uint8)
    seq4[: n * stride] = rows.reshape(-1)
    targets = []
    for c, r in enumerate(res):
        targets += [(f"s{c}{name}", ln) for name, ln in r[2]]
    rec = RecordBatch(cat_tid("tid"), cat("pos"), cat_tid("mtid"), cat("mpos"), cat("flag"), cat("mapq"), cig_off, cat("cigar"),
                      np.arange(n, dtype=np.uint64) * np.uint64(stride), cat("l_seq"), seq4, qoff, b"".join(p["qnames"] for p in pieces),
                      cat("isize"), targets)
    gs = [r[1] for r in res]
    off = np.zeros(n_chunks * n_contigs + 1, np.int64)
    base = 0
    for c, g in enumerate(gs):
        off[c * n_contigs + 1:(c + 1) * n_contigs + 1] = g.iv_off[1:] + base
        base += int(g.iv_off[-1])
    g = GenomeStr(n_chunks * n_contigs, np.concatenate([g.has_chrom for g in gs]), off, np.concatenate([g.iv_start for g in gs]),
                  np.concatenate([g.iv_stop for g in gs]))
    return rec, g


def synth_wgs_30x(n_chunks, pairs_per_chunk, seed=1234, procs=None, coverage=30.0, read_len=150, loci=3200, **kw):
    """A slab of a 30x WGS (BASELINE.json: "30x 150 bp PE synthetic WGS"): synth_wgs_chunks with the genome sized so that the
    batch covers it `coverage` times -- consecutive records of the coordinate-sorted batch start ~5 bp apart, as in a real
    30x BAM, instead of hundreds of bases apart.  Every chunk owns two contigs of reads * read_len / coverage / 2 bases;
    the expanded loci the STR-rich reads pile up on stay `loci` in all (so clusters keep the size real expansions have)."""
    contig_len = max(20_000, int(2 * pairs_per_chunk * read_len / coverage / 2))
    hot = max(1, loci // (2 * n_chunks))
    return synth_wgs_chunks(n_chunks, pairs_per_chunk, seed=seed, procs=procs, read_len=read_len, n_contigs=2, contig_len=contig_len,
                            hot_loci=hot, **kw)


def frag_hist(rec):
    """utils.fragment_length_distribution (utils.nim:86-111) on an in-memory batch, without the 100k-record
    skip (the batch IS the sample): histogram of isize in [0, 4095] over proper-pair primary records."""
    f = rec.flag
    ok = ((f & 0x2) != 0) & ((f & 0x900) == 0) & (rec.isize >= 0) & (rec.isize <= 4095)
    return np.bincount(rec.isize[ok], minlength=4096).astype(np.uint32)


def synth_treads(n_samples=4, n_loci=400, seed=1000, n_contigs=25, contig_len=10_000_000, background=0.3, dtype=None):
    """S2: per sample every locus present w.p. 0.3 with 5-60 reads: anchors (Soft.none) within +-400 bp, left/right
    clips at the locus boundary +-{0,1}; plus background singletons.  Returns one structured array in
    "sample order then .bin order" with qname_id = sample index (merge.nim:118-125)."""
    from .api import TREAD_DTYPE
    dtype = dtype or TREAD_DTYPE
    rng = _rng(seed)
    units = []
    bases = "ACGT"
    while len(units) < 60:
        k = int(rng.integers(1, 7))
        u = "".join(bases[int(b)] for b in rng.integers(0, 4, size=k))
        units.append(u)
    zipf = 1.0 / np.arange(1, len(units) + 1)
    zipf /= zipf.sum()
    loc_tid = rng.integers(0, n_contigs, size=n_loci)
    loc_pos = rng.integers(1000, contig_len - 1000, size=n_loci)
    loc_len = rng.integers(0, 120, size=n_loci)
    loc_unit = rng.choice(len(units), size=n_loci, p=zipf)
    out = []
    for s in range(n_samples):
        present = rng.random(n_loci) < 0.3
        rows = []
        for li in np.nonzero(present)[0]:
            nr = int(rng.integers(5, 61))
            kind = rng.random(nr)
            left_b, right_b = int(loc_pos[li]), int(loc_pos[li] + loc_len[li])
            for kd in kind:
                if kd < 0.5:
                    rows.append((loc_tid[li], max(0, left_b + int(rng.integers(-400, 401))), loc_unit[li], 3))
                elif kd < 0.75:   # right-clipped reads end at the left boundary of the repeat
                    rows.append((loc_tid[li], left_b + int(rng.integers(0, 2)), loc_unit[li], 1))
                else:             # left-clipped reads start at its right boundary
                    rows.append((loc_tid[li], right_b + int(rng.integers(0, 2)), loc_unit[li], 0))
        nb = int(len(rows) * background)
        for _ in range(nb):
            rows.append((int(rng.integers(0, n_contigs)), int(rng.integers(0, contig_len)), int(rng.choice(len(units), p=zipf)),
                         int(rng.choice([0, 1, 3, 3]))))
        perm = rng.permutation(len(rows))
        t = np.zeros(len(rows), dtype)
        for j, pi in enumerate(perm):
            r = rows[pi]
            t[j]["tid"] = r[0]; t[j]["position"] = r[1]; t[j]["repeat"] = units[r[2]].encode(); t[j]["split"] = r[3]
        t["qname_id"] = s
        t["mapping_quality"] = 60
        t["repeat_count"] = 40
        t["align_length"] = 150
        out.append(t)
    return np.concatenate(out)


def synth_chrom(n_bases, seed=1, n_str=None, soft_mask=True):
    """One reference chromosome for `strling index`: random ACGT with embedded STR arrays (pure and interrupted,
    units of 1-6 bases, some spanning many 100-base windows, some touching the chromosome ends, adjacent arrays
    with different units), N gaps, IUPAC letters and lower-case (soft-masked) stretches."""
    rng = _rng(seed)
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), n_bases).copy()
    if n_str is None:
        n_str = max(4, n_bases // 4000)
    spots = np.sort(rng.integers(0, max(1, n_bases - 50), n_str))
    for j, at in enumerate(spots):
        k = int(rng.integers(1, 7))
        unit = rng.choice(np.frombuffer(b"ACGT", np.uint8), k)
        ln = int(rng.choice([40, 70, 100, 130, 200, 400, 1500]))
        arr = np.resize(unit, ln).copy()
        nerr = int(ln * rng.choice([0.0, 0.0, 0.02, 0.08, 0.2]))
        if nerr:
            arr[rng.integers(0, ln, nerr)] = rng.choice(np.frombuffer(b"ACGT", np.uint8), nerr)
        if j == 0:
            at = 0
        if j == n_str - 1:
            at = max(0, n_bases - ln)
        ln = min(ln, n_bases - at)
        seq[at:at + ln] = arr[:ln]
        if j % 5 == 4 and at + ln + 120 < n_bases:          # a second array right behind, different unit
            u2 = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(2, 7)))
            seq[at + ln:at + ln + 120] = np.resize(u2, 120)
    for _ in range(max(1, n_bases // 20000)):               # N gaps and stray IUPAC letters
        at = int(rng.integers(0, n_bases))
        seq[at:at + int(rng.choice([5, 25, 300]))] = ord("N")
        seq[int(rng.integers(0, n_bases))] = int(rng.choice(np.frombuffer(b"RYKMSWn", np.uint8)))
    if soft_mask:
        for _ in range(max(1, n_bases // 10000)):
            at = int(rng.integers(0, n_bases))
            seq[at:at + 500] |= 0x20
    return seq.tobytes()


def synth_htt(n_pairs=5000, seed=7, contig_len=200_000, tract_at=100_000, ref_units=19, extra_units=100, read_len=150):
    """SURVEY section 8(d) input S0 (BASELINE.json configs[0], cf. sim/htt_locus.bed): one contig "4" of iid ACGT (seed 42)
    with a (CAG)x19 tract, two haplotypes (reference and +100 CAG units), paired reads with fragment length
    ~ N(350, 120), records synthesised without an aligner:
      * flank reads: 150M, mapq 60, proper pair;
      * reads that run from a flank into the expansion: the part beyond the reference tract is soft-clipped (xMyS / xSyM);
      * reads wholly inside the expansion: unmapped (flag 0x4), mapq 0, placed at the mate's position; when both mates
        are inside, both unmapped with tid -1 at the end of the file.
    -> (RecordBatch, reference sequence bytes)"""
    from .records import RecordBatch
    g = np.random.Generator(np.random.Philox(42))
    ref = g.choice(np.frombuffer(b"ACGT", np.uint8), contig_len)
    tract = np.resize(np.frombuffer(b"CAG", np.uint8), 3 * ref_units)
    ref[tract_at:tract_at + tract.size] = tract
    T0, T1 = tract_at, tract_at + tract.size                          # reference tract [T0, T1)
    ins = 3 * extra_units
    alt = np.concatenate([ref[:T1], np.resize(np.frombuffer(b"CAG", np.uint8), ins), ref[T1:]])   # expansion appended behind the tract
    rng = _rng(seed)
    L = read_len
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    tids, poss, mtids, mposs, flags, mapqs, cigs, seqs, qn, isz = [], [], [], [], [], [], [], [], [], []

    def place(h, a):
        """read covering haplotype coordinates [a, a+L) -> (tid, pos, cigar string, mapq, unmapped)"""
        b = a + L
        if h == 0 or b <= T1:                                         # reference haplotype, or left of the expansion
            return 0, a, f"{L}M", 60, False
        e0, e1 = T1, T1 + ins                                         # expansion in haplotype coordinates
        if a >= e1:                                                   # right flank: shift back
            return 0, a - ins, f"{L}M", 60, False
        if a >= e0 - 0 and b <= e1 + 0 and a >= e0 and b <= e1:       # wholly inside the expansion
            return -1, 0, "*", 0, True
        if a < e0:                                                    # enters the expansion from the left
            m = e0 - a
            if b <= e1:
                return 0, a, f"{m}M{L - m}S", 60, False
            return 0, a, f"{m}M{ins}I{b - e1}M", 60, False             # spans the whole expansion (only if ins < L)
        m = b - e1                                                    # leaves the expansion to the right
        return 0, T1, f"{L - m}S{m}M", 60, False

    for i in range(n_pairs):
        h = int(rng.integers(0, 2))
        hap = alt if h else ref
        frag = int(np.clip(np.rint(rng.normal(350, 120)), L, 4095))
        a = int(rng.integers(0, hap.size - frag))
        r1 = place(h, a)
        r2 = place(h, a + frag - L)
        s1 = bytes(hap[a:a + L])
        s2 = bytes(hap[a + frag - L:a + frag])                         # SEQ is stored on the forward strand for the reverse mate
        both_un = r1[4] and r2[4]
        for k, (r, s, other) in enumerate(((r1, s1, r2), (r2, s2, r1))):
            tid, pos, cig, mq, un = r
            f = 0x1 | (0x40 if k == 0 else 0x80) | (0x10 if k == 1 else 0x20)
            if un:
                f |= 0x4
                if not both_un:
                    tid, pos = other[0], other[1]                      # unmapped mate sits at its mate's position
            if other[4]:
                f |= 0x8
            mt, mp = (other[0], other[1]) if not other[4] else ((tid, pos) if not both_un else (-1, -1))
            if both_un:
                tid, pos, mt, mp = -1, -1, -1, -1
            proper = not un and not other[4]
            if proper:
                f |= 0x2
            tids.append(tid); poss.append(pos); mtids.append(mt); mposs.append(mp); flags.append(f); mapqs.append(mq); cigs.append(cig)
            seqs.append(s.decode()); qn.append(f"htt{i}")
            span = (r2[1] + L) - r1[1] if proper else 0
            isz.append((span if k == 0 else -span) if proper else 0)
    order = sorted(range(len(tids)), key=lambda j: ((1 << 40) if tids[j] < 0 else (tids[j] << 32) + max(poss[j], 0), j))
    pick = lambda x: [x[j] for j in order]
    rec = RecordBatch.from_fields(pick(tids), pick(poss), pick(mtids), pick(mposs), pick(flags), pick(mapqs), pick(cigs), pick(seqs), pick(qn),
                                  isize=pick(isz), targets=[("4", contig_len)])
    return rec, ref.tobytes()
