// front.hip -- the BAM front end of `strling extract` on the GPU (gfx950), behind the inflate kernel of bgzf.hip:
// extract.nim:275-329 iterates the BAM through htslib (one thread: inflate, record by record); here a chunk of BGZF blocks
// (~16 Ki blocks, ~1 GB inflated) goes through
//   inflate_kernel   (bgzf.hip)      one wave per BGZF block
//   carry_kernel                     the partial record the previous chunk ended in is copied in front of this chunk's bytes
//   rec_guess_kernel                 one wave per 16 KiB segment: first offset from which a chain of plausible BAM records runs
//   rec_walk_kernel                  one lane per segment: follow block_size from the segment's (guessed) start to its end:
//                                    record count, SEQ / qname byte sums, where the chain leaves the segment
//   rec_link_kernel                  one block: every walk must arrive EXACTLY at the next segment's guessed start (checked in
//                                    parallel); if one does not, a single lane follows the true chain and re-walks the
//                                    segments that were guessed wrong -- a wrong guess costs time, never a result;
//                                    then exclusive scans give every segment its first record index / SEQ / qname offset
//   rec_emit_kernel                  one lane per segment: the walk again, writing each record's offset, SEQ slot, qname slot
//   rec_parse_kernel                 one lane per record -> strl_read_soa columns, strl_pair_rec row, qname hash, SEQ copy
//                                    (16-byte aligned), qname bytes into the arena, flag/isize word (fragment lengths)
//   rec_tail_kernel                  primary records behind the last placed one (extract.nim:326 visits "*" again)
// The host only walks BGZF headers and copies compressed bytes; it never sees a record.
#include <string.h>
#include <algorithm>
#include "front.h"
#include "device_util.h"

int strl_crc_device(strl_ctx *c, const uint8_t *d_out, const uint64_t *d_uoff, const uint32_t *d_isize, const uint32_t *d_crc, uint32_t n_blocks, uint8_t *d_status,
                    uint32_t *d_err, hipStream_t st);
int strl_inflate_device(strl_ctx *c, const uint8_t *d_comp, uint64_t readable, const uint64_t *d_coff, const uint32_t *d_clen, const uint64_t *d_uoff,
                        const uint32_t *d_isize, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_err, uint8_t *d_status, hipStream_t st, uint64_t out_bytes,
                        uint8_t *d_work, size_t work_bytes);
size_t strl_inflate_work_bytes(uint32_t n_blocks);

namespace strl {

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16u(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// BAM record at q: block_size u32 | refID i32 | pos i32 | l_read_name u8 | mapq u8 | bin u16 | n_cigar_op u16 | flag u16 |
// l_seq u32 | next_refID i32 | next_pos i32 | tlen i32 | read_name | cigar | seq | qual | tags   (SAM spec 4.2)
struct RecHdr { uint32_t bs; int32_t l_seq; uint32_t l_qname, n_cigar; };
// the same validity rule as the host reader's (cli/bam_reader.cpp `header`): what htslib's bam_read1 insists on
__device__ __forceinline__ bool rec_header(const uint8_t *U, uint32_t q, RecHdr &h) {
  h.bs = ld32u(U + q);
  h.l_qname = U[q + 12];
  h.n_cigar = ld16u(U + q + 16);
  h.l_seq = (int32_t)ld32u(U + q + 20);
  return !(h.bs < 32u || h.l_seq < 0 || 32ull + h.l_qname + 4ull * h.n_cigar + ((uint64_t)(uint32_t)h.l_seq + 1) / 2 > h.bs);
}

// 1 plausible record at q, 0 not a record, -1 too few bytes left to tell
__device__ int rec_plausible(const uint8_t *U, uint64_t q, uint32_t end, int32_t n_ref, uint64_t &next) {
  if (q + 36 > end) return -1;
  RecHdr h;
  if (!rec_header(U, (uint32_t)q, h) || h.bs > (1u << 26)) return 0;
  const int32_t ref = (int32_t)ld32u(U + q + 4), pos = (int32_t)ld32u(U + q + 8), nref = (int32_t)ld32u(U + q + 24), npos = (int32_t)ld32u(U + q + 28);
  if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos < -1 || npos < -1 || h.l_qname < 1u) return 0;
  if ((uint64_t)(uint32_t)h.l_seq + ((uint64_t)(uint32_t)h.l_seq + 1) / 2 + 32 + h.l_qname + 4ull * h.n_cigar > h.bs) return 0;   // the qualities fit too
  const uint64_t name_end = q + 36 + h.l_qname - 1;
  if (name_end < end && U[name_end] != 0) return 0;
  for (uint64_t c = q + 36; c < name_end && c < end; ++c) if (U[c] < 33 || U[c] > 126) return 0;
  next = q + 4 + (uint64_t)h.bs;
  return 1;
}

// follow the block_size chain from e through the records that START before seg_end
__device__ void walk_segment(const uint8_t *U, uint32_t e, uint64_t seg_end, uint32_t end, FrontSeg &r) {
  uint64_t q = e;
  uint32_t cnt = 0, seq16 = 0, qn = 0, maxl = 0, flags = 0;
  while (q < seg_end) {
    if (q + 36 > end) { flags |= 1u; break; }
    RecHdr h;
    if (!rec_header(U, (uint32_t)q, h)) { flags |= 3u; break; }
    if (q + 4 + h.bs > end) { flags |= 1u; break; }
    ++cnt;
    seq16 += (((uint32_t)h.l_seq + 1u) / 2u + 15u) / 16u;
    qn += h.l_qname ? h.l_qname - 1u : 0u;
    maxl = max(maxl, (uint32_t)h.l_seq);
    q += 4ull + h.bs;
  }
  r.entry = e; r.exit = (uint32_t)q; r.cnt = cnt; r.seq16 = seq16; r.qn = qn; r.max_l_seq = maxl; r.flags = flags;
}

// the partial record behind the last complete one of the previous chunk goes in front of this chunk's bytes
// behind the record scan of a chunk: its trailing partial record + length into the slot's carry buffer.  The next chunk takes it
// from there (carry_kernel), not from this chunk's inflated bytes: those may then be overwritten as soon as the parse is done
__global__ void carry_out_kernel(const uint8_t *infl, const FrontInfo *info, uint8_t *buf) {
  const uint32_t len = info->carry_len <= FRONT_CARRY_MAX ? info->carry_len : 0u, off = info->carry_off;
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) buf[64 + i] = infl[off + i];
  if (threadIdx.x == 0) *reinterpret_cast<uint32_t *>(buf) = len;
}
__global__ void carry_kernel(const uint8_t *prev_buf, uint8_t *infl, FrontInfo *info) {
  const uint32_t len = *reinterpret_cast<const uint32_t *>(prev_buf);
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) infl[FRONT_CARRY_MAX - len + i] = prev_buf[64 + i];
  if (threadIdx.x == 0) info->start0 = FRONT_CARRY_MAX - len;
}
__global__ void carry_stage_kernel(const uint8_t *stage, uint32_t stage_len, uint32_t prev_end, uint8_t *infl, FrontInfo *info) {
  const FrontInfo *prev = reinterpret_cast<const FrontInfo *>(stage + FRONT_CARRY_MAX + 64);
  const uint32_t len = prev->carry_len <= stage_len ? prev->carry_len : 0u;
  const uint32_t off = prev->carry_off - (prev_end - stage_len);      // carry_off >= prev_end - len >= prev_end - stage_len
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) infl[FRONT_CARRY_MAX - len + i] = stage[off + i];
  if (threadIdx.x == 0) {
    info->start0 = FRONT_CARRY_MAX - len;
    if (prev->carry_len > stage_len) atomicOr(&info->err, FRONT_ERR_CARRY);
  }
}

// One WAVE per segment: the 64 lanes test 64 consecutive candidate offsets at a time (a cheap range test of block_size, which
// 98 % of the candidates fail, then the chain of four plausible records) and the lowest one that passes is the guess -- the
// same answer as a lane scanning the segment byte by byte, in two or three rounds instead of ~150 dependent steps.
__global__ __launch_bounds__(256) void rec_guess_kernel(const uint8_t *U, const FrontInfo *info, FrontSeg *seg, uint32_t n_seg, int32_t n_ref) {
  const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (s >= n_seg) return;
  const uint32_t start0 = info->start0, end = info->end, s0 = start0 / FRONT_SEG;
  uint32_t g = FRONT_NONE;
  if (s > s0) {
    const uint64_t from = (uint64_t)s * FRONT_SEG, lim = min((uint64_t)end, from + FRONT_SEG);
    for (uint64_t o0 = from; o0 < lim; o0 += 64) {      // wave-uniform
      const uint64_t o = o0 + lane;
      bool good = false;
      if (o < lim && o + 36 <= end) {
        const uint32_t bs = ld32u(U + o);
        if (!(bs < 32u || bs > (1u << 26))) {
          uint64_t c = o, nx = 0;
          int n_ok = 0;
          good = true;
          while (n_ok < 4) {
            const int r = rec_plausible(U, c, end, n_ref, nx);
            if (r == 0) { good = false; break; }
            if (r < 0) { good = n_ok >= 1; break; }     // ran into the end of the data: one whole plausible record is all there is
            ++n_ok;
            c = nx;
          }
        }
      }
      const unsigned long long m = __ballot(good);
      if (m) { g = (uint32_t)(o0 + (uint64_t)(__ffsll((long long)m) - 1)); break; }
      if (o0 + 64 + 36 > end) break;                    // (the sequential scan stops at the first offset too close to the end)
    }
  }
  if (lane == 0) seg[s].guess = g;
}

__global__ __launch_bounds__(64) void rec_walk_kernel(const uint8_t *U, const FrontInfo *info, FrontSeg *seg, uint32_t n_seg) {
  const uint32_t s = blockIdx.x * 64u + threadIdx.x;
  if (s >= n_seg) return;
  const uint32_t start0 = info->start0, end = info->end, s0 = start0 / FRONT_SEG;
  FrontSeg r = seg[s];
  const uint32_t e = s == s0 ? start0 : (s > s0 ? r.guess : FRONT_NONE);
  if (e == FRONT_NONE) { r.entry = FRONT_NONE; r.exit = FRONT_NONE; r.cnt = 0; r.seq16 = 0; r.qn = 0; r.max_l_seq = 0; r.flags = 0; }
  else walk_segment(U, e, ((uint64_t)s + 1) * FRONT_SEG, end, r);
  seg[s] = r;
}

// One block.  reached[] = scratch of n_seg words; base3 = [3][n_seg] exclusive sums of cnt / seq16 / qn.
__global__ __launch_bounds__(1024) void rec_link_kernel(const uint8_t *U, FrontSeg *seg, uint32_t n_seg, FrontInfo *info, uint32_t *reached, uint32_t *base3) {
  __shared__ uint32_t sh_bad, sh_stop, sh_err, sh_maxl, sh_carry, sh_slow;
  __shared__ uint64_t sh_part[3][1024];
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  const uint32_t start0 = info->start0, end = info->end, s0 = start0 / FRONT_SEG;
  if (t == 0) { sh_bad = 0; sh_stop = 0; sh_err = 0; sh_maxl = 0; sh_carry = end; sh_slow = 0; }
  for (uint32_t s = t; s < n_seg; s += nt) reached[s] = FRONT_NONE;
  __syncthreads();
  // every walked segment says where its chain arrives
  for (uint32_t s = t; s < n_seg; s += nt) {
    const FrontSeg r = seg[s];
    if (r.entry == FRONT_NONE) continue;
    // (a chain that arrives where fewer than 36 bytes are left -- a partial header, possibly in a fresh segment nobody could guess
    // a start in -- ends there: the chunk's carry begins at that offset.  Round-3 advisor: this used to count as a mismatch and sent
    // one lane down the whole chain.)
    if ((r.flags & 1u) || r.exit >= end || r.exit + 36u > end) { atomicAdd(&sh_stop, 1u); continue; }
    if (atomicExch(&reached[r.exit / FRONT_SEG], r.exit) != FRONT_NONE) atomicOr(&sh_bad, 1u);
  }
  __syncthreads();
  // ... and every segment checks that it is reached exactly where it started (the first one: by nobody)
  for (uint32_t s = t; s < n_seg; s += nt) {
    const uint32_t e = seg[s].entry, r = reached[s];
    const uint32_t want = (s == s0 || e == FRONT_NONE) ? FRONT_NONE : e;
    if (r != want) atomicOr(&sh_bad, 1u);
  }
  __syncthreads();
  const bool ok = !sh_bad && sh_stop == 1u;
  if (!ok) {
    // the exact chain, one lane: segments it visits keep (or get) their walk, everything else holds no records
    for (uint32_t s = t; s < n_seg; s += nt) reached[s] = 0;      // now: visited flags
    __syncthreads();
    if (t == 0) {
      uint32_t e = start0, slow = 0;
      for (;;) {
        const uint32_t s = e / FRONT_SEG;
        if (s >= n_seg) break;
        if (seg[s].entry != e) {
          FrontSeg r = seg[s];
          walk_segment(U, e, ((uint64_t)s + 1) * FRONT_SEG, end, r);
          seg[s] = r;
          ++slow;
        }
        reached[s] = 1;
        if ((seg[s].flags & 1u) || seg[s].exit >= end) break;
        e = seg[s].exit;
      }
      sh_slow = slow;
    }
    __syncthreads();
    for (uint32_t s = t; s < n_seg; s += nt)
      if (!reached[s]) { seg[s].entry = FRONT_NONE; seg[s].cnt = 0; seg[s].seq16 = 0; seg[s].qn = 0; seg[s].flags = 0; seg[s].max_l_seq = 0; }
    __syncthreads();
  }
  // exclusive scans over the segments (each thread a contiguous run), totals, the carry
  const uint32_t per = (n_seg + nt - 1) / nt, a = min(n_seg, t * per), b = min(n_seg, a + per);
  uint64_t s_cnt = 0, s_seq = 0, s_qn = 0;
  uint32_t maxl = 0, err = 0, carry = FRONT_NONE;
  for (uint32_t s = a; s < b; ++s) {
    const FrontSeg r = seg[s];
    s_cnt += r.cnt; s_seq += r.seq16; s_qn += r.qn;
    maxl = max(maxl, r.max_l_seq);
    if (r.entry != FRONT_NONE) {
      if (r.flags & 2u) err |= FRONT_ERR_RECORD;
      if ((r.flags & 1u) || (r.exit < end && r.exit + 36u > end)) carry = r.exit;
    }
  }
  sh_part[0][t] = s_cnt; sh_part[1][t] = s_seq; sh_part[2][t] = s_qn;
  if (maxl) atomicMax(&sh_maxl, maxl);
  if (err) atomicOr(&sh_err, err);
  if (carry != FRONT_NONE) atomicMin(&sh_carry, carry);
  __syncthreads();
  if (t < 3) {      // three lanes, one quantity each: 1024 partial sums
    uint64_t run = 0;
    for (uint32_t i = 0; i < nt; ++i) { const uint64_t v = sh_part[t][i]; sh_part[t][i] = run; run += v; }
    if (t == 0) info->n_records = (uint32_t)run;
    if (t == 1) info->seq_bytes = run * 16ull;
    if (t == 2) info->qname_bytes = run;
  }
  __syncthreads();
  uint64_t r0 = sh_part[0][t], r1 = sh_part[1][t], r2 = sh_part[2][t];
  for (uint32_t s = a; s < b; ++s) {
    const FrontSeg r = seg[s];
    base3[s] = (uint32_t)r0; base3[n_seg + s] = (uint32_t)r1; base3[2 * n_seg + s] = (uint32_t)r2;
    r0 += r.cnt; r1 += r.seq16; r2 += r.qn;
  }
  if (t == 0) {
    const uint32_t co = sh_carry;
    uint32_t e = sh_err;
    info->carry_off = co;
    info->carry_len = end - co;
    if (end - co > FRONT_CARRY_MAX) e |= FRONT_ERR_CARRY;
    info->max_l_seq = sh_maxl;
    info->all_ok = ok ? 1u : 0u;
    info->slow_segments = sh_slow;
    info->err |= e | ((info->inflate_err & 3u) ? FRONT_ERR_INFLATE : 0u) | ((info->inflate_err & 4u) ? FRONT_ERR_CRC : 0u);
    info->n_primary = 0; info->last_placed = -1; info->tail_primary = 0;
  }
}

__global__ __launch_bounds__(64) void rec_emit_kernel(const uint8_t *U, const FrontInfo *info, const FrontSeg *seg, uint32_t n_seg, const uint32_t *base3,
                                                      uint32_t *recoff, uint32_t *seqoff, uint32_t *qoff) {
  const uint32_t s = blockIdx.x * 64u + threadIdx.x;
  if (s >= n_seg) return;
  const FrontSeg r = seg[s];
  if (r.entry == FRONT_NONE || !r.cnt) return;
  uint32_t i = base3[s], so = base3[n_seg + s], qo = base3[2 * n_seg + s];
  uint64_t q = r.entry;
  for (uint32_t k = 0; k < r.cnt; ++k) {
    RecHdr h;
    (void)rec_header(U, (uint32_t)q, h);
    recoff[i] = (uint32_t)q; seqoff[i] = so; qoff[i] = qo;
    ++i;
    so += (((uint32_t)h.l_seq + 1u) / 2u + 15u) / 16u;
    qo += h.l_qname ? h.l_qname - 1u : 0u;
    q += 4ull + h.bs;
  }
}

struct ParseParams {
  const uint8_t *U;
  const uint32_t *recoff, *seqoff, *qoff;
  uint32_t n;
  FrontInfo *info;
  FrontParseOut o;
  uint8_t *tid_seen;      // [n_ref] contigs that have a primary record (the CLI's "extracting chromosome" lines)
  int32_t n_ref;
};

__global__ __launch_bounds__(256) void rec_parse_kernel(ParseParams P) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool act = i < P.n;
  bool primary = false;
  int32_t placed_idx = -1;
  const uint8_t *seq_src = nullptr;
  uint32_t seq_so = 0, seq_nb = 0;
  if (act) {
    const uint32_t q = P.recoff[i];
    const uint8_t *R = P.U + q;
    const int32_t tid = (int32_t)ld32u(R + 4), pos = (int32_t)ld32u(R + 8);
    const uint32_t l_qname = R[12], mapq = R[13], n_cigar = ld16u(R + 16), flag = ld16u(R + 18);
    const uint32_t l_seq = ld32u(R + 20);
    const int32_t mtid = (int32_t)ld32u(R + 24), mpos = (int32_t)ld32u(R + 28), isize = (int32_t)ld32u(R + 32);
    const uint8_t *name = R + 36, *cg = name + l_qname, *sq = cg + 4u * n_cigar;
    // cigar: what extract.nim:30-38,83-87,98-119 asks of it (strl_soa_from_records is the host's version), and bam_endpos
    uint32_t cbits = 0, cl = 0, cr = 0;
    int64_t rl = 0;
    if (n_cigar == 0) cbits = STRL_CIG_NONE;
    else {
      const uint32_t c0 = ld32u(cg), cL = ld32u(cg + 4u * (n_cigar - 1u));
      if (n_cigar == 1) cbits |= STRL_CIG_ONE_OP;
      if (n_cigar == 1 && (c0 & 15u) == 0u) { cbits |= STRL_CIG_SINGLE_M; cl = c0 >> 4; }
      if ((c0 & 15u) == 4u) { cbits |= STRL_CIG_FIRST_S; cl = c0 >> 4; }
      if ((cL & 15u) == 4u) { cbits |= STRL_CIG_LAST_S; cr = cL >> 4; }
      if (!(flag & 4u))
        for (uint32_t j = 0; j < n_cigar; ++j) {
          const uint32_t c = ld32u(cg + 4u * j), op = c & 15u;
          if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) rl += c >> 4;
        }
    }
    const int32_t end = (int32_t)((int64_t)pos + (rl ? rl : 1));
    if (l_seq > (uint32_t)STRL_MAX_READ_LEN) atomicOr(&P.info->err, FRONT_ERR_LSEQ);   // (what the 16-bit columns hold; beyond STRL_DEVICE_READ_LEN: the host twin)
    const uint16_t ls16 = (uint16_t)min(l_seq, 65535u), cl16 = (uint16_t)min(cl, 65535u), cr16 = (uint16_t)min(cr, 65535u);
    P.o.tid[i] = tid; P.o.pos[i] = pos; P.o.end[i] = end;
    const uint32_t so = P.seqoff[i];
    P.o.seq_off[i] = so; P.o.l_seq[i] = ls16; P.o.clip_l[i] = cl16; P.o.clip_r[i] = cr16;
    P.o.mapq[i] = (uint8_t)mapq; P.o.cig[i] = (uint8_t)cbits;
    P.o.meta[i] = make_uint4(so, (uint32_t)ls16 | ((uint32_t)cl16 << 16), (uint32_t)cr16 | (cbits << 16) | (mapq << 24), 0u);
    strl_pair_rec row;
    row.tid = tid; row.pos = pos; row.mtid = mtid; row.mpos = mpos; row.end = end;
    row.flag = (uint16_t)flag; row.l_seq = ls16; row.clip_l = cl16; row.clip_r = cr16; row.mapq = (uint8_t)mapq; row.cig = (uint8_t)cbits; row.pad = 0;
    P.o.rows[i] = row;
    // (SEQ: copied by the whole wave behind this block, one record at a time)
    seq_src = sq; seq_so = so; seq_nb = (l_seq + 1u) / 2u;
    // qname: bytes into the arena, FNV-1a hash like strl_qname_hash (the Cache of extract.nim:198,245 is keyed by it)
    const uint32_t ql = l_qname ? l_qname - 1u : 0u;
    const uint64_t qat = P.o.qarena_at + P.qoff[i];
    uint8_t *qd = P.o.qarena + qat;
    uint64_t hsh = 0xcbf29ce484222325ull;
    for (uint32_t j = 0; j < ql; ++j) {
      const uint8_t ch = name[j];
      qd[j] = ch;
      hsh ^= ch;
      hsh *= 0x100000001b3ull;
    }
    P.o.qhash[i] = hsh ^ (hsh >> 29);
    P.o.qref[i] = (qat << 8) | ql;
    P.o.fragw[i] = flag | ((isize >= 0 && isize <= 4095 ? (uint32_t)isize : 0xffffu) << 16);
    primary = !(flag & 0x900u);
    P.o.tidflag[i] = (uint8_t)((primary ? 1u : 0u) | (tid >= 0 ? 2u : 0u));
    if (tid >= 0) placed_idx = (int32_t)i;
    if (primary && tid >= 0 && tid < P.n_ref) P.tid_seen[tid] = 1;
  }
  // SEQ as it sits in the record (4-bit codes), to a 16-byte aligned slot; the bytes behind it in the last dword zeroed.
  // Copied by the WAVE, record by record: the lanes take consecutive dwords of one record (two cache lines read, two written
  // per record).  A lane copying its own record dword by dword made every load and every store instruction of the wave touch
  // 64 different lines, four bytes of each: the parse was a quarter of the device time of `strling extract`
  // (profiles/r04/e2e_kernel_stats.csv: 7.2 ms per 4.2e6 records, 0.26 TB/s).
  {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t src64 = (uint64_t)reinterpret_cast<uintptr_t>(seq_src);
    const uint32_t src_lo = (uint32_t)src64, src_hi = (uint32_t)(src64 >> 32);
    const unsigned long long have = __ballot(seq_nb != 0u);
    for (unsigned long long m = have; m; m &= m - 1ull) {       // wave-uniform
      const int r = __ffsll((long long)m) - 1;
      const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)seq_nb, r), so = (uint32_t)__builtin_amdgcn_readlane((int)seq_so, r);
      const uint8_t *sq = reinterpret_cast<const uint8_t *>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)src_hi, r) << 32) |
                                                                          (uint32_t)__builtin_amdgcn_readlane((int)src_lo, r)));
      uint32_t *dst = reinterpret_cast<uint32_t *>(P.o.seq4 + (uint64_t)so * 16u);
      for (uint32_t j = lane; j * 4u < nb; j += 64u) {
        uint32_t w = ld32u(sq + 4u * j);
        const uint32_t left = nb - 4u * j;
        if (left < 4u) w &= (1u << (8u * left)) - 1u;
        dst[j] = w;
      }
    }
  }
  // chunk summary: one atomic per wave
  const uint64_t pm = __ballot(primary);
  int32_t mx = placed_idx;
  for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
  if ((threadIdx.x & 63u) == 0u) {
    if (pm) atomicAdd(&P.info->n_primary, (uint32_t)__popcll(pm));
    if (mx >= 0) atomicMax(&P.info->last_placed, mx);
  }
}

__global__ void rec_tail_kernel(const uint8_t *tidflag, uint32_t n, FrontInfo *info) {
  const int32_t lp = info->last_placed;
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if ((int32_t)i > lp && (tidflag[i] & 1u)) ++c;
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
  if ((threadIdx.x & 63u) == 0u && c) atomicAdd(&info->tail_primary, c);
}

__global__ void name_refs_kernel(const uint64_t *qref, const uint32_t *ids, uint32_t n, uint64_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = qref[ids[i]];
}
__global__ void name_copy_kernel(const uint8_t *arena, const uint64_t *ref, const uint64_t *dst_off, uint32_t n, uint8_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = ref[i];
  const uint8_t *s = arena + (r >> 8);
  uint8_t *d = out + dst_off[i];
  for (uint32_t j = 0; j < (uint32_t)(r & 255u); ++j) d[j] = s[j];
}

// names of the resident treads (in .bin order): reference and length per tread, exclusive scan of the lengths (one block),
// byte copies
__global__ void tread_name_refs_kernel(const uint64_t *qref, const strl_tread *treads, const uint32_t *n_dev, uint32_t cap, uint64_t *ref, uint32_t *len) {
  const uint32_t n = min(*n_dev, cap);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = qref[(uint64_t)treads[i].qname_id];
  ref[i] = r;
  len[i] = (uint32_t)(r & 255u);
}
// Exclusive scan of the name lengths: tiles of 2048 lengths, a workgroup each (sums, then the tile sums scanned by one
// workgroup, then every tile placed).  (Until round 6 ONE workgroup did it, each thread walking a contiguous 1/1024th of the
// lengths -- 64 cache lines per wave and load, 7800 dependent turns twice over for a genome's 8e6 names: most of the 0.043 s
// `strling extract` spent between the pair pass and the .bin.)
constexpr uint32_t NAME_TILE = 2048;
__global__ __launch_bounds__(256) void name_tile_sums_kernel(const uint32_t *len, const uint32_t *n_dev, uint32_t cap, uint64_t *tile_sum) {
  __shared__ uint32_t wsum[4];
  const uint32_t n = min(*n_dev, cap), t = threadIdx.x, base = blockIdx.x * NAME_TILE;
  uint32_t s = 0;                                               // (a tile's lengths sum to < 2048 * 256)
  for (uint32_t j = 0; j < NAME_TILE / 256; ++j) { const uint32_t i = base + j * 256 + t; if (i < n) s += len[i]; }
  for (int d = 32; d; d >>= 1) s += __shfl_down(s, d, 64);
  if ((t & 63) == 0) wsum[t >> 6] = s;
  __syncthreads();
  if (t == 0) tile_sum[blockIdx.x] = (uint64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(1024) void name_tile_scan_kernel(uint64_t *tile_sum, uint32_t n_tiles_cap, const uint32_t *n_dev, uint32_t cap, uint64_t *off) {
  __shared__ uint64_t part[1024];
  const uint32_t n = min(*n_dev, cap), t = threadIdx.x;
  const uint32_t nt = min((n + NAME_TILE - 1) / NAME_TILE, n_tiles_cap);
  const uint32_t per = (nt + 1023u) / 1024u, a = min(nt, t * per), b = min(nt, a + per);
  uint64_t s = 0;
  for (uint32_t i = a; i < b; ++i) s += tile_sum[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) { uint64_t run = 0; for (int k = 0; k < 1024; ++k) { const uint64_t v = part[k]; part[k] = run; run += v; } off[n] = run; }
  __syncthreads();
  uint64_t run = part[t];
  for (uint32_t i = a; i < b; ++i) { const uint64_t v = tile_sum[i]; tile_sum[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void name_tile_place_kernel(const uint32_t *len, const uint32_t *n_dev, uint32_t cap, const uint64_t *tile_sum, uint64_t *off) {
  __shared__ uint32_t wsum[4];
  const uint32_t n = min(*n_dev, cap), t = threadIdx.x, base = blockIdx.x * NAME_TILE + t * (NAME_TILE / 256);
  uint32_t l[NAME_TILE / 256], s = 0;                            // a thread's eight lengths are neighbours: 32 bytes in, 64 out
  for (uint32_t j = 0; j < NAME_TILE / 256; ++j) { l[j] = base + j < n ? len[base + j] : 0u; s += l[j]; }
  uint32_t incl = s;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if ((int)(t & 63) >= d) incl += v; }
  if ((t & 63) == 63) wsum[t >> 6] = incl;
  __syncthreads();
  uint64_t run = tile_sum[blockIdx.x] + (incl - s);
  for (uint32_t w = 0; w < (t >> 6); ++w) run += wsum[w];
  for (uint32_t j = 0; j < NAME_TILE / 256; ++j) { if (base + j < n) off[base + j] = run; run += l[j]; }
}
__global__ void tread_name_copy_kernel(const uint8_t *arena, const uint64_t *ref, const uint64_t *off, const uint32_t *n_dev, uint32_t cap, uint8_t *out, uint64_t out_cap) {
  const uint32_t n = min(*n_dev, cap);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r = ref[i], o = off[i];
  const uint32_t l = (uint32_t)(r & 255u);
  if (o + l > out_cap) return;
  const uint8_t *s = arena + (r >> 8);
  for (uint32_t j = 0; j < l; ++j) out[o + j] = s[j];
}
size_t front_name_tiles(uint32_t cap) { return ((size_t)cap + NAME_TILE - 1) / NAME_TILE + 1; }
int front_tread_names(strl_ctx *c, strl_front *F, const strl_tread *d_treads, const uint32_t *d_n, uint32_t cap, uint64_t *d_ref, uint32_t *d_len, uint64_t *d_off,
                      uint8_t *d_out, uint64_t out_cap, uint64_t *d_tile_sums, hipStream_t st) {
  const unsigned g = (cap + 255) / 256, tiles = (cap + NAME_TILE - 1) / NAME_TILE;       // d_tile_sums: front_name_tiles(cap) words
  hipLaunchKernelGGL(tread_name_refs_kernel, dim3(g), dim3(256), 0, st, F->qref.as<uint64_t>(), d_treads, d_n, cap, d_ref, d_len);
  hipLaunchKernelGGL(name_tile_sums_kernel, dim3(tiles), dim3(256), 0, st, d_len, d_n, cap, d_tile_sums);
  hipLaunchKernelGGL(name_tile_scan_kernel, dim3(1), dim3(1024), 0, st, d_tile_sums, tiles, d_n, cap, d_off);
  hipLaunchKernelGGL(name_tile_place_kernel, dim3(tiles), dim3(256), 0, st, d_len, d_n, cap, d_tile_sums, d_off);
  hipLaunchKernelGGL(tread_name_copy_kernel, dim3(g), dim3(256), 0, st, F->qarena.as<uint8_t>(), d_ref, d_off, d_n, cap, d_out, out_cap);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}

static int tick(strl_front *F, hipStream_t st) {   // STRL_FRONT_TIMING: an event behind every stage
  static const bool on = getenv("STRL_FRONT_TIMING") != nullptr;
  if (!on) return STRL_OK;
  hipEvent_t e;
  STRL_HIP(hipEventCreate(&e));
  STRL_HIP(hipEventRecord(e, st));
  F->tev.push_back(e);
  return STRL_OK;
}

// H2D of the chunk's compressed bytes + block table (st_c), inflate + CRC (st_i), record scan (st_a); asynchronous.  The slot's
// previous occupant must have been parsed (ev_b) before its buffers are overwritten: waited for on the device.  Nothing else
// orders the inflates of consecutive chunks (each slot has a stream): the scan of chunk k, and the last waves of its inflate,
// run beside the inflate of chunk k+1.
// the st_c part: block tables + compressed bytes to the device.  Needs only that the slot's previous inflate + CRC are done
// (ev_i, waited for on the device), so it can be queued a whole chunk early (strl_front_stage).
int front_copy(strl_ctx *c, strl_front *F, int si, const FrontChunkDesc &d) {
  FrontSlot &S = F->slot[si];
  int rc;
  const uint32_t nb = d.n_blocks;
  if (S.h2d_pending) { STRL_HIP(hipEventSynchronize(S.ev_h2d)); S.h2d_pending = false; }   // h_uoff is the source of the slot's previous copy (long done)
  if (S.h_uoff_cap < nb) {     // (page-locked: the copy below must not block this thread)
    if (S.h_uoff) { (void)hipHostFree(S.h_uoff); S.h_uoff = nullptr; }
    S.h_uoff_cap = std::max(nb + nb / 4 + 1024, F->hint_blocks);
    STRL_HIP(hipHostMalloc(reinterpret_cast<void **>(&S.h_uoff), (size_t)S.h_uoff_cap * 8, hipHostMallocDefault));
  }
  uint64_t *uoff = S.h_uoff;
  uint64_t tot = 0;
  for (uint32_t i = 0; i < nb; ++i) {
    if (d.coff[i] + d.clen[i] > d.comp_bytes) { set_error("block %u reaches past the chunk's compressed bytes", i); return STRL_ERR_ARG; }
    uoff[i] = FRONT_CARRY_MAX + tot;
    tot += d.isize[i];
  }
  if (tot + FRONT_CARRY_MAX > 0xf0000000ull) { set_error("chunk inflates to %llu bytes (limit 3.7 GB)", (unsigned long long)tot); return STRL_ERR_ARG; }
  const uint64_t readable = (d.comp_bytes + 3) & ~(uint64_t)3;
  auto want = [](uint64_t need) { return (size_t)(need + need / 4 + 4096); };   // head-room: later chunks rarely reallocate
  if (S.comp.cap < readable + 16 && (rc = S.comp.reserve(want(readable + 16)))) return rc;
  if (S.coff.cap < (size_t)nb * 8 && ((rc = S.coff.reserve(want((size_t)nb * 8))) || (rc = S.uoff.reserve(want((size_t)nb * 8))) ||
                                       (rc = S.clen.reserve(want((size_t)nb * 4))) || (rc = S.isize.reserve(want((size_t)nb * 4))) || (rc = S.crc.reserve(want((size_t)nb * 4))) ||
                                       (rc = S.status.reserve(want(nb)))))
    return rc;
  // the copies go on a stream of their own: they only have to wait until the slot's previous inflate + CRC (which read
  // these buffers) are done, and this chunk's inflate waits for them
  hipStream_t sc = F->st_c;
  if (S.i_pending) STRL_HIP(hipStreamWaitEvent(sc, S.ev_i, 0));
  STRL_HIP(hipMemcpyAsync(S.comp.p, d.comp, d.comp_bytes, hipMemcpyHostToDevice, sc));
  STRL_HIP(hipMemsetAsync(static_cast<uint8_t *>(S.comp.p) + d.comp_bytes, 0, 16, sc));
  STRL_HIP(hipMemcpyAsync(S.coff.p, d.coff, (size_t)nb * 8, hipMemcpyHostToDevice, sc));
  STRL_HIP(hipMemcpyAsync(S.uoff.p, uoff, (size_t)nb * 8, hipMemcpyHostToDevice, sc));
  STRL_HIP(hipMemcpyAsync(S.clen.p, d.clen, (size_t)nb * 4, hipMemcpyHostToDevice, sc));
  STRL_HIP(hipMemcpyAsync(S.isize.p, d.isize, (size_t)nb * 4, hipMemcpyHostToDevice, sc));
  if (d.crc) STRL_HIP(hipMemcpyAsync(S.crc.p, d.crc, (size_t)nb * 4, hipMemcpyHostToDevice, sc));
  STRL_HIP(hipEventRecord(S.ev_h2d, sc));
  S.h2d_pending = true;
  if (F->next_trim > tot || (F->next_trim && F->next_trim > d.isize[nb - 1])) { set_error("strl_front_trim_next: more bytes than the chunk's last block holds"); return STRL_ERR_ARG; }
  S.staged = true; S.staged_comp = d.comp; S.staged_bytes = d.comp_bytes; S.staged_blocks = nb; S.staged_tot = tot;
  S.staged_trim = F->next_trim;
  F->next_trim = 0;
  return STRL_OK;
}

// buffers of both slots for chunks of up to max_blocks blocks / max_comp_bytes compressed bytes, before the first chunk: a slot
// that grows in the middle of the file (the short first chunk's, two chunks later) frees its buffers -- a device-wide wait
int front_reserve(strl_ctx *c, strl_front *F, uint32_t max_blocks, uint64_t max_comp_bytes) {
  int rc;
  F->hint_blocks = max_blocks;
  const uint64_t tot = (uint64_t)max_blocks * 65280u, end = FRONT_CARRY_MAX + tot, n_seg = (end + FRONT_SEG - 1) / FRONT_SEG, rec_cap = tot / 36 + 16;
  if (end > 0xf0000000ull) { set_error("chunks of %u blocks inflate to more than 3.7 GB", max_blocks); return STRL_ERR_ARG; }
  for (FrontSlot &S : F->slot) {
    if ((rc = S.comp.reserve((size_t)max_comp_bytes + 64)) || (rc = S.infl.reserve((size_t)end + 256)) || (rc = S.coff.reserve((size_t)max_blocks * 8)) ||
        (rc = S.uoff.reserve((size_t)max_blocks * 8)) || (rc = S.clen.reserve((size_t)max_blocks * 4)) || (rc = S.isize.reserve((size_t)max_blocks * 4)) ||
        (rc = S.crc.reserve((size_t)max_blocks * 4)) || (rc = S.status.reserve(max_blocks)) || (rc = S.seg.reserve((size_t)n_seg * sizeof(FrontSeg))) ||
        (rc = S.base3.reserve((size_t)n_seg * 16)) || (rc = S.recoff.reserve((size_t)rec_cap * 4)) || (rc = S.seqoff.reserve((size_t)rec_cap * 4)) ||
        (rc = S.qoff.reserve((size_t)rec_cap * 4)))
      return rc;
  }
  return STRL_OK;
}

int front_stage_a(strl_ctx *c, strl_front *F, int si, const FrontChunkDesc &d, bool first, const FrontCarrySrc *carry) {
  FrontSlot &S = F->slot[si];
  // STRL_FRONT_SERIAL (profiling): inflate, CRC and record scan of every chunk on the context's ONE stream, behind the previous
  // chunk's parse and scorer -- no launch runs beside another, so a kernel trace attributes each kernel its own time
  static const bool serial = getenv("STRL_FRONT_SERIAL") != nullptr;
  hipStream_t st = serial ? c->stream : F->st_a, sti = serial ? c->stream : F->st_i[si];
  int rc;
  const uint32_t nb = d.n_blocks;
  if (S.staged && (S.staged_comp != d.comp || S.staged_bytes != d.comp_bytes || S.staged_blocks != nb)) {
    set_error("strl_front_push: not the chunk handed to strl_front_stage");
    return STRL_ERR_ARG;
  }
  if (!S.staged && (rc = front_copy(c, F, si, d))) return rc;
  S.staged = false;
  const uint64_t tot = S.staged_tot;
  // (a share's last chunk: the record scan ends in front of the bytes that are the next share's; the buffers hold the whole blocks)
  const uint32_t end_full = (uint32_t)(FRONT_CARRY_MAX + tot), end = end_full - S.staged_trim, n_seg = (end + FRONT_SEG - 1) / FRONT_SEG;
  const uint64_t readable = (d.comp_bytes + 3) & ~(uint64_t)3;
  const uint64_t rec_cap = tot / 36 + 16;          // no record is shorter than 36 bytes
  if (S.b_pending) STRL_HIP(hipStreamWaitEvent(sti, S.ev_b, 0));
  if (S.read_pending) { STRL_HIP(hipStreamWaitEvent(sti, S.wait_read, 0)); S.read_pending = false; }   // another context took its carry from this slot
  auto want = [](uint64_t need) { return (size_t)(need + need / 4 + 4096); };   // head-room: later chunks rarely reallocate
  if (S.infl.cap < (uint64_t)end_full + 256 && (rc = S.infl.reserve(want((uint64_t)end_full + 256)))) return rc;
  if (S.seg.cap < (size_t)n_seg * sizeof(FrontSeg) && ((rc = S.seg.reserve(want((size_t)n_seg * sizeof(FrontSeg)))) || (rc = S.base3.reserve(want((size_t)n_seg * 16)))))
    return rc;
  if (S.recoff.cap < rec_cap * 4 && ((rc = S.recoff.reserve(want(rec_cap * 4))) || (rc = S.seqoff.reserve(want(rec_cap * 4))) || (rc = S.qoff.reserve(want(rec_cap * 4)))))
    return rc;
  if ((rc = S.info.reserve(sizeof(FrontInfo)))) return rc;
  const size_t iwork = strl_inflate_work_bytes(nb);
  if (S.iwork.cap < iwork && (rc = S.iwork.reserve(want(iwork)))) return rc;
  S.n_blocks = nb; S.n_seg = n_seg; S.infl_bytes = tot; S.comp_bytes = d.comp_bytes;
  if ((rc = tick(F, sti))) return rc;
  STRL_HIP(hipStreamWaitEvent(sti, S.ev_h2d, 0));
  FrontInfo &hi = S.h_info[2];
  memset(&hi, 0, sizeof hi);
  hi.start0 = (uint32_t)(FRONT_CARRY_MAX + (first ? F->first_off : 0));
  hi.end = end;
  hi.last_placed = -1;
  STRL_HIP(hipMemcpyAsync(S.info.p, &hi, sizeof hi, hipMemcpyHostToDevice, sti));
  STRL_HIP(hipMemsetAsync(static_cast<uint8_t *>(S.infl.p) + end_full, 0, 256, sti));      // the parse may load a dword across the end
  if ((rc = tick(F, sti))) return rc;
  FrontInfo *info = S.info.as<FrontInfo>();
  if ((rc = strl_inflate_device(c, S.comp.as<uint8_t>(), readable, S.coff.as<uint64_t>(), S.clen.as<uint32_t>(), S.uoff.as<uint64_t>(), S.isize.as<uint32_t>(), nb,
                                S.infl.as<uint8_t>(), &info->inflate_err, S.status.as<uint8_t>(), sti, end_full, S.iwork.as<uint8_t>(), iwork)))
    return rc;
  if (d.crc && (rc = strl_crc_device(c, S.infl.as<uint8_t>(), S.uoff.as<uint64_t>(), S.isize.as<uint32_t>(), S.crc.as<uint32_t>(), nb, S.status.as<uint8_t>(),
                                     &info->inflate_err, sti)))
    return rc;
  if ((rc = tick(F, sti))) return rc;
  STRL_HIP(hipEventRecord(S.ev_i, sti));
  S.i_pending = true;
  STRL_HIP(hipStreamWaitEvent(st, S.ev_i, 0));
  if ((rc = F->carry_buf[si].reserve((size_t)FRONT_CARRY_MAX + 64))) return rc;
  if (!first && !carry) {
    hipLaunchKernelGGL(carry_kernel, dim3(1), dim3(1024), 0, st, F->carry_buf[si ^ 1].as<uint8_t>(), S.infl.as<uint8_t>(), info);
    STRL_HIP(hipGetLastError());
  } else if (!first) {
    // the previous chunk lives in another context (possibly on another device): its tail and its summary come over by a copy
    // behind its record scan, then the carry is cut out locally
    const uint32_t stage_len = std::min<uint32_t>(carry->end, FRONT_CARRY_MAX);
    if ((rc = S.carry_stage.reserve((size_t)FRONT_CARRY_MAX + 64 + sizeof(FrontInfo) + 64))) return rc;
    STRL_HIP(hipStreamWaitEvent(st, carry->ev_a, 0));
    uint8_t *stg = S.carry_stage.as<uint8_t>();
    if (carry->device == c->device) {
      STRL_HIP(hipMemcpyAsync(stg, carry->infl + (carry->end - stage_len), stage_len, hipMemcpyDeviceToDevice, st));
      STRL_HIP(hipMemcpyAsync(stg + FRONT_CARRY_MAX + 64, carry->info, sizeof(FrontInfo), hipMemcpyDeviceToDevice, st));
    } else {
      STRL_HIP(hipMemcpyPeerAsync(stg, c->device, carry->infl + (carry->end - stage_len), carry->device, stage_len, st));
      STRL_HIP(hipMemcpyPeerAsync(stg + FRONT_CARRY_MAX + 64, c->device, carry->info, carry->device, sizeof(FrontInfo), st));
    }
    STRL_HIP(hipEventRecord(S.ev_carry, st));          // this context's own event, on its own stream (round-3 advisor finding)
    *carry->wait_read = S.ev_carry;
    *carry->read_pending = true;
    hipLaunchKernelGGL(carry_stage_kernel, dim3(1), dim3(1024), 0, st, stg, stage_len, carry->end, S.infl.as<uint8_t>(), info);
    STRL_HIP(hipGetLastError());
  }
  const unsigned gb = (n_seg + 63) / 64;
  hipLaunchKernelGGL(rec_guess_kernel, dim3((n_seg + 3) / 4), dim3(256), 0, st, S.infl.as<uint8_t>(), info, S.seg.as<FrontSeg>(), n_seg, (int32_t)F->n_ref);
  STRL_HIP(hipGetLastError());
  hipLaunchKernelGGL(rec_walk_kernel, dim3(gb), dim3(64), 0, st, S.infl.as<uint8_t>(), info, S.seg.as<FrontSeg>(), n_seg);
  STRL_HIP(hipGetLastError());
  uint32_t *base3 = S.base3.as<uint32_t>();
  hipLaunchKernelGGL(rec_link_kernel, dim3(1), dim3(1024), 0, st, S.infl.as<uint8_t>(), S.seg.as<FrontSeg>(), n_seg, info, base3 + 3 * (size_t)n_seg, base3);
  STRL_HIP(hipGetLastError());
  hipLaunchKernelGGL(rec_emit_kernel, dim3(gb), dim3(64), 0, st, S.infl.as<uint8_t>(), info, S.seg.as<FrontSeg>(), n_seg, base3, S.recoff.as<uint32_t>(),
                     S.seqoff.as<uint32_t>(), S.qoff.as<uint32_t>());
  STRL_HIP(hipGetLastError());
  hipLaunchKernelGGL(carry_out_kernel, dim3(1), dim3(1024), 0, st, S.infl.as<uint8_t>(), info, F->carry_buf[si].as<uint8_t>());
  STRL_HIP(hipGetLastError());
  if ((rc = tick(F, st))) return rc;
  STRL_HIP(hipMemcpyAsync(S.h_info, S.info.p, sizeof(FrontInfo), hipMemcpyDeviceToHost, st));
  STRL_HIP(hipEventRecord(S.ev_a, st));
  S.a_pending = true;
  F->last_slot = si;
  F->last_end = end;
  return STRL_OK;
}

int front_parse(strl_ctx *c, strl_front *F, int si, uint32_t n, const FrontParseOut &o, hipStream_t st) {
  FrontSlot &S = F->slot[si];
  if (n) {
    ParseParams P{S.infl.as<uint8_t>(), S.recoff.as<uint32_t>(), S.seqoff.as<uint32_t>(), S.qoff.as<uint32_t>(), n, S.info.as<FrontInfo>(), o,
                  F->tid_seen.as<uint8_t>(), (int32_t)F->n_ref};
    hipLaunchKernelGGL(rec_parse_kernel, dim3((n + 255) / 256), dim3(256), 0, st, P);
    STRL_HIP(hipGetLastError());
    hipLaunchKernelGGL(rec_tail_kernel, dim3(std::min<unsigned>((n + 255) / 256, 1024)), dim3(256), 0, st, o.tidflag, n, S.info.as<FrontInfo>());
    STRL_HIP(hipGetLastError());
  }
  STRL_HIP(hipMemcpyAsync(S.h_info + 1, S.info.p, sizeof(FrontInfo), hipMemcpyDeviceToHost, st));
  return STRL_OK;
}

int front_gather_names(strl_ctx *c, strl_front *F, const uint32_t *d_ids, uint32_t n, uint64_t *d_ref_out, hipStream_t st) {
  if (!n) return STRL_OK;
  hipLaunchKernelGGL(name_refs_kernel, dim3((n + 255) / 256), dim3(256), 0, st, F->qref.as<uint64_t>(), d_ids, n, d_ref_out);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}
int front_copy_names(strl_ctx *c, strl_front *F, const uint64_t *d_ref, const uint64_t *d_dst_off, uint32_t n, uint8_t *d_out, hipStream_t st) {
  if (!n) return STRL_OK;
  hipLaunchKernelGGL(name_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, F->qarena.as<uint8_t>(), d_ref, d_dst_off, n, d_out);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}

void front_destroy(strl_front *F) {
  if (!F) return;
  if (F->big) {
    if (F->big->th.joinable()) F->big->th.join();
    for (DevBuf *b : {&F->big->rows, &F->big->qhash, &F->big->whole, &F->big->qref, &F->big->fragw, &F->big->qarena}) b->release();
    delete F->big;
    F->big = nullptr;
  }
  for (DevBuf &b : F->trash) b.release();
  F->trash.clear();
  for (FrontSlot &S : F->slot) {
    for (DevBuf *b : {&S.comp, &S.infl, &S.coff, &S.clen, &S.uoff, &S.isize, &S.crc, &S.status, &S.seg, &S.recoff, &S.seqoff, &S.qoff, &S.info, &S.base3, &S.carry_stage, &S.iwork}) b->release();
    if (S.ev_a) (void)hipEventDestroy(S.ev_a);
    if (S.ev_b) (void)hipEventDestroy(S.ev_b);
    if (S.ev_h2d) (void)hipEventDestroy(S.ev_h2d);
    if (S.ev_i) (void)hipEventDestroy(S.ev_i);
    if (S.ev_carry) (void)hipEventDestroy(S.ev_carry);
    if (S.h_info) (void)hipHostFree(S.h_info);
    if (S.h_uoff) (void)hipHostFree(S.h_uoff);
  }
  for (DevBuf *b : {&F->qref, &F->qarena, &F->fragw, &F->tidflag, &F->tid_seen, &F->s_tid, &F->s_pos, &F->s_end, &F->s_seqoff, &F->s_lseq, &F->s_clipl, &F->s_clipr, &F->s_mapq,
                    &F->s_cig, &F->s_seq4, &F->s_meta})
    b->release();
  for (hipEvent_t e : F->tev) (void)hipEventDestroy(e);
  if (F->st_a) (void)hipStreamDestroy(F->st_a);
  for (hipStream_t q : F->st_i) if (q) (void)hipStreamDestroy(q);
  for (DevBuf &b : F->carry_buf) b.release();
  if (F->st_c) (void)hipStreamDestroy(F->st_c);
  delete F;
}

}  // namespace strl
