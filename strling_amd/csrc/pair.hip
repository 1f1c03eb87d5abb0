// pair.hip -- the mate-pairing logic of `strling extract` (Cache.add, extract.nim:192-248, with to_tread :63-87,
// add_soft :93-132, adjust_by :141-179, unplaced_pair :182-190, canonical_repeat utils.nim:304-316) on the GPU, gfx950.
//
// The reference keeps a qname-keyed hash table of first-seen reads and walks the BAM sequentially.  Two facts make the
// walk data-parallel:
//   1. qname groups never interact (the table is keyed by qname; every emission happens while a record of the group is
//      being processed), and
//   2. a group none of whose records carries a repeat (whole-read count 0 and no soft-clip result) emits nothing.
// So the device joins only the HOT groups -- a few per cent of the reads:
//   mark   : every scored read / soft-clip record with a non-zero count sets two bits of a Bloom bitmap keyed by the
//            64-bit qname hash (whole reads: inside score_kernel; soft-clip records: pair_soft_items_kernel, which also
//            turns each such record into a join item),
//   probe  : one streaming pass over the qname hashes of ALL reads (8 B per read -- the only full pass the pair logic
//            adds) tests the bitmap (2 MB, L2 resident) and turns the hits into join items,
//   join   : the items are radix-sorted by hash (sort.hip); a run of equal hashes = one qname group (reads + its hot
//            soft-clip records),
//   replay : one lane per run replays Cache.add over the group's records in file order -- first pass over all records,
//            second pass over the unmapped tail (extract.nim:326-329) -- with the reference's uint8 / uint32 / float64
//            arithmetic, and buffers what the group emits,
//   order  : every emitted tread carries (pass, record index, sequence number) = the position of its emission in the
//            reference's sequential walk; one more radix sort puts the treads into exactly the order of the .bin file.
// Bloom false positives only add groups that emit nothing.
#include <string.h>
#include <algorithm>
#include "common.h"
#include "device_util.h"
#include "sort.h"
#include "front.h"

namespace strl {

constexpr uint16_t F_PROPER = 0x2, F_REVERSE = 0x10, F_MREVERSE = 0x20, F_SECONDARY = 0x100, F_SUPPL = 0x800;
constexpr int PAIR_MAXM = 15;   // items (reads + hot soft-clip records) of one hash run a lane replays out of its block's LDS
constexpr int PAIR_LONG_MAX = PAIR_LONG_MAX_ITEMS;     // ... longer runs go to pair_long_kernel (a block per run), up to this many items
constexpr uint32_t PAIR_SPILL_CAP = 1u << 16;

struct PairParams {
  uint32_t n;               // reads of the batch
  uint32_t tail_start;      // first record of the unmapped tail visited a second time (n if none)
  const strl_pair_rec *rec;    // one 32-byte row per read
  const uint64_t *qhash;
  const uint32_t *whole;
  const strl_soft_rec *soft;
  const uint32_t *d_n_soft;    // number of soft-clip records (device)
  uint32_t scap;
  uint32_t *bloom;
  uint32_t bloom_mask;         // bits - 1
  uint64_t *item_key;
  uint32_t *item_val;          // read index, or 0x80000000 | soft record index
  uint32_t item_cap;
  uint32_t *pc;                // PC_* counters
  strl_tread *emit;
  uint64_t *emit_key;
  uint32_t *emit_val;
  uint32_t emit_cap;
  double p;
  uint32_t min_mapq;
  int32_t frag_median;
  uint32_t *spill;             // [PAIR_SPILL_CAP] sorted-item index of the first item of every run the block replay passed on
  const uint64_t *qref;        // device front end only: (qname arena offset << 8) | length per record, else null
  const uint8_t *qarena;
};

// The Cache of extract.nim:198,245 is keyed by the qname STRING; the join keys on its 64-bit hash.  Where the names are on
// the device (front end), the records of one hash group are checked to carry one name: a collision is reported
// (PAIR_ERR_COLLISION -> the caller repeats the extraction with the string-keyed host Cache), never silently merged.
__device__ inline bool qname_differs(const PairParams &P, uint32_t r0, uint32_t r1) {
  if (!P.qref) return false;
  const uint64_t a = P.qref[r0], b = P.qref[r1];
  if ((a & 255u) != (b & 255u)) return true;
  const uint8_t *x = P.qarena + (a >> 8), *y = P.qarena + (b >> 8);
  for (uint32_t j = 0; j < (uint32_t)(a & 255u); ++j) if (x[j] != y[j]) return true;
  return false;
}

// Soft-clip records with a result under either threshold: mark their read's qname group and make them join items.
// One queue-space atomic per 2048 records (the same-address atomic rate of the L2 is ~88 per microsecond).
__global__ __launch_bounds__(1024) void pair_soft_items_kernel(PairParams P) {
  __shared__ uint32_t wcnt[128];
  __shared__ uint32_t base_sh;
  constexpr int U = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t n_src = *P.d_n_soft;
  if (n_src > P.scap) n_src = P.scap;
  for (uint32_t b0 = blockIdx.x * (1024u * U); b0 < n_src; b0 += gridDim.x * (1024u * U)) {
    uint64_t m[U];
    bool hot[U];
    unsigned long long bal[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t j = b0 + 1024u * u + threadIdx.x;
      hot[u] = false;
      m[u] = 0;
      if (j < n_src) {
        const strl_soft_rec s = P.soft[j];
        hot[u] = (STRL_RES_COUNT(s.res_first) | STRL_RES_COUNT(s.res_after)) != 0;
        // (Secondary / supplementary records never reach Cache.add (extract.nim:309,327).  Their clips are NOT filtered out here
        // any more: the replay never looks a clip up for a record it skips, a group that joins for nothing emits nothing, and the
        // test cost every hot record a gather into the 32-byte rows -- a 128-byte line for two flag bits, 40 % of this kernel's
        // 140 MB (profiles/r06/traffic.json: 141 B moved per soft-clip record of 16).)
        if (hot[u]) {
          m[u] = fmix64(P.qhash[s.read_side >> 1]);
          bloom_set(P.bloom, P.bloom_mask, m[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bal[u] = __ballot(hot[u]);
      if (lane == 0) wcnt[u * 16 + wave] = (uint32_t)__popcll(bal[u]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (int w = 0; w < 16 * U; ++w) { const uint32_t c = wcnt[w]; wcnt[w] = tot; tot += c; }
      base_sh = tot ? atomicAdd(&P.pc[PC_ITEMS], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (hot[u]) {
        const uint32_t d = base_sh + wcnt[u * 16 + wave] + (uint32_t)__popcll(bal[u] & below);
        if (d < P.item_cap) { P.item_key[d] = m[u]; P.item_val[d] = 0x80000000u | (b0 + 1024u * u + threadIdx.x); }
        else atomicOr(&P.pc[PC_ERR], PAIR_ERR_ITEMS);
      }
    }
    __syncthreads();
  }
}

// The one full pass of the pair logic: qname hash of every read against the bitmap.  Each wave owns a contiguous range
// of reads and stages its hits in LDS; a flush reserves item space with ONE atomic (same-address atomics run at ~88 per
// microsecond on the L2, so a wave flushes once, at the end, unless its stage fills) and gathers the hashes of the
// staged reads again.
constexpr int PR_STAGE = 1024, PR_ILP = 8;
__global__ __launch_bounds__(256) void pair_probe_kernel(PairParams P) {
  __shared__ uint32_t stage[4][PR_STAGE + 64 * PR_ILP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *buf = stage[wave];
  const uint64_t n_waves = (uint64_t)gridDim.x * 4u;
  const uint64_t gw = (uint64_t)blockIdx.x * 4u + wave;
  const uint64_t per = (((P.n + n_waves - 1) / n_waves) + 63ull) & ~63ull;
  const uint64_t r0 = gw * per;
  const uint64_t r1 = r0 + per < P.n ? r0 + per : P.n;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t cnt = 0;
  auto flush = [&]() {
    // (Secondary / supplementary records are NOT filtered here: Cache.add's replay skips them (extract.nim:309,327), and a
    // qname whose copies make its run longer than 15 items goes to pair_long_kernel like any long run.  Testing the flag here
    // cost one gather into the 32-byte rows per true hit, all of them in the waves' final flush: 0.03 ms of 0.21.)
    if (cnt) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(&P.pc[PC_ITEMS], cnt);
      b = __shfl(b, 0);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < cnt; i += 64) {
        const uint32_t r = buf[i];
        const uint32_t d = b + i;
        if (d < P.item_cap) { P.item_key[d] = fmix64(P.qhash[r]); P.item_val[d] = r; }
        else atomicOr(&P.pc[PC_ERR], PAIR_ERR_ITEMS);
      }
      __builtin_amdgcn_wave_barrier();
      cnt = 0;
    }
  };
  uint64_t cur[PR_ILP], nxt[PR_ILP];
  auto load = [&](uint64_t base, uint64_t (&x)[PR_ILP]) {
#pragma unroll
    for (int j = 0; j < PR_ILP; ++j) {
      const uint64_t r = base + 64ull * j + lane;
      x[j] = r < r1 ? P.qhash[r] : 0ull;
    }
  };
  load(r0, cur);
  for (uint64_t base = r0; base < r1; base += 64 * PR_ILP) {
    load(base + 64 * PR_ILP, nxt);
    // first bit of all PR_ILP reads (independent loads), then the second bit of the few that passed
    uint64_t m[PR_ILP];
    uint32_t w0[PR_ILP];
#pragma unroll
    for (int j = 0; j < PR_ILP; ++j) {
      m[j] = fmix64(cur[j]);
      w0[j] = P.bloom[((uint32_t)m[j] & P.bloom_mask) >> 5];
    }
    bool hit[PR_ILP];
#pragma unroll
    for (int j = 0; j < PR_ILP; ++j) {
      const uint64_t r = base + 64ull * j + lane;
      hit[j] = r < r1 && ((w0[j] >> ((uint32_t)m[j] & 31u)) & 1u);
    }
#pragma unroll
    for (int j = 0; j < PR_ILP; ++j) {
      if (hit[j]) {
        const uint32_t b1 = (uint32_t)(m[j] >> 32) & P.bloom_mask;
        hit[j] = (P.bloom[b1 >> 5] >> (b1 & 31u)) & 1u;
      }
    }
#pragma unroll
    for (int j = 0; j < PR_ILP; ++j) {
      const unsigned long long mk = __ballot(hit[j]);
      if (hit[j]) buf[cnt + __popcll(mk & below)] = (uint32_t)(base + 64ull * j + lane);
      cnt += (uint32_t)__popcll(mk);
    }
    if (cnt >= PR_STAGE) flush();
#pragma unroll
    for (int j = 0; j < PR_ILP; ++j) cur[j] = nxt[j];
  }
  flush();
}

// ---- the replay of Cache.add for one qname group ------------------------------------------------------------------
struct DTread {   // tread (cluster.nim:23-32) with the unit as (length, kmer code)
  int32_t tid;
  uint32_t position;
  uint32_t k, code;
  uint16_t flag;
  uint8_t split, mapq, count, align_length;
  uint32_t qid;
};

__device__ __forceinline__ double p_repeat(const DTread &t) {   // extract.nim:56-58 (uint8 product)
  const uint8_t prod = (uint8_t)(t.count * (uint8_t)t.k);
  const uint8_t al = t.align_length ? t.align_length : 1;
  return (double)prod / (double)al;
}
// minimum rotation of the reverse complement (utils.nim:61-80) on 2-bit "CATG" codes: complement of c is 3 - c
__device__ inline uint32_t min_rev_complement(uint32_t code, uint32_t k) {
  if (k == 0) return code;
  uint32_t c = 0;
  for (int i = (int)k - 1; i >= 0; --i) c = (c << 2) | (3u - ((code >> (2 * (k - 1 - i))) & 3u));
  const uint32_t mask = (1u << (2 * k)) - 1u;
  uint32_t best = c, f = c;
  for (uint32_t j = 0; j < k; ++j) {
    f = ((f << 2) | (f >> (2 * (k - 1)))) & mask;
    best = f < best ? f : best;
  }
  return best;
}
// canonical_repeat, utils.nim:291-316: the rev-comp rotation minimum if it is ASCII-smaller (A < C < G < T)
__device__ inline uint32_t canonical_repeat(uint32_t code, uint32_t k) {
  const uint32_t r = min_rev_complement(code, k);
  for (uint32_t j = 0; j < k; ++j) {
    const uint32_t a = (r >> (2 * (k - 1 - j))) & 3u, b = (code >> (2 * (k - 1 - j))) & 3u;
    // code -> ASCII rank: C(0) -> 1, A(1) -> 0, T(2) -> 3, G(3) -> 2
    const uint32_t ra = a ^ 1u, rb = b ^ 1u;
    if (ra != rb) return ra < rb ? r : code;
  }
  return code;
}
__device__ __forceinline__ bool should_reverse(uint16_t f) {   // extract.nim:134-139
  const bool r = !(f & F_MREVERSE);
  return (f & F_REVERSE) ? !r : r;
}
// adjust_by, extract.nim:141-179
__device__ inline bool adjust_by(DTread &A, const DTread &B, const PairParams &P, uint32_t B_position) {
  if (A.count == 0) return false;
  const uint32_t half = (uint32_t)((double)A.align_length / 2.0 + 0.5);
  if (B.mapq > P.min_mapq && ((p_repeat(A) > P.p && p_repeat(B) < 0.2) || (!(A.flag & F_PROPER) && A.mapq < P.min_mapq))) {
    if (B.flag & F_REVERSE) {
      A.position = B_position - (uint32_t)P.frag_median + B.align_length + half;
      if (B.split == STRL_SOFT_NONE_LEFT) A.position = B_position;
    } else {
      A.position = B_position + (uint32_t)P.frag_median - half;
      if (B.split == STRL_SOFT_NONE_RIGHT) A.position = B_position + (uint32_t)B.align_length;
    }
    A.split = STRL_SOFT_NONE;
    A.tid = B.tid;
    A.mapq = A.mapq > B.mapq ? A.mapq : B.mapq;
    if (should_reverse(A.flag)) A.code = min_rev_complement(A.code, A.k);
  } else if (A.mapq >= P.min_mapq || (A.flag & F_PROPER)) {
    A.position += half;
    A.mapq = A.mapq > B.mapq ? A.mapq : B.mapq;
  }
  return true;
}
__device__ inline bool unplaced_pair(const DTread &A, const DTread &B, const PairParams &P) {   // extract.nim:182-190
  if (p_repeat(A) > P.p && p_repeat(B) > P.p) return true;
  if (p_repeat(A) > P.p && B.mapq < P.min_mapq) return true;
  if (p_repeat(B) > P.p && A.mapq < P.min_mapq) return true;
  return false;
}

// One sorted join item with everything the replay needs, gathered by the item's own lane (coalesced over the block) and
// kept in LDS: the head lane of a run then works out of LDS instead of chasing a dozen arrays per record.
struct PItem {   // 48 bytes
  uint64_t key;      // mixed qname hash
  uint32_t val;      // read index, or 0x80000000 | soft record index
  int32_t tid, pos, mtid, mpos, end;      // soft record: tid = read_side, pos = res_first, mtid = res_after
  uint32_t whole;
  uint16_t flag, clip_l, clip_r, l_seq;
  uint8_t mapq, cig;
  uint16_t pad;
};
static_assert(sizeof(PItem) == 48, "PItem layout");

constexpr int PG_BLOCK = 512;                // items per block
constexpr int PG_HALO = 16;                  // a run that starts in the block may reach this far into the next one
constexpr int PG_EMIT = 256;                 // treads one block may emit (LDS staging; 35 KB per block in all: four blocks per CU)

struct EmitStage {   // block-wide staging of the emitted treads in LDS
  strl_tread *t;
  uint64_t *key;
  uint32_t *count;
  const PairParams *P;
};
__device__ inline void emit(const EmitStage &E, const DTread &d, uint64_t vidx, uint32_t &seq, uint32_t &err) {
  const uint32_t slot = atomicAdd(E.count, 1u);
  const uint32_t k = seq++;
  strl_tread t;
  t.tid = d.tid;
  t.position = d.position;
#pragma unroll
  for (int j = 0; j < 6; ++j) t.repeat[j] = (uint32_t)j < d.k ? "CATG"[(d.code >> (2 * (d.k - 1 - j))) & 3u] : (char)0;
  t.flag = d.flag;
  t.split = d.split;
  t.mapping_quality = d.mapq;
  t.repeat_count = d.count;
  t.align_length = d.align_length;
  t.qname_id = (int64_t)d.qid;
  const uint64_t key = (vidx << 2) | (uint64_t)(k & 3u);
  if (slot < (uint32_t)PG_EMIT) { E.t[slot] = t; E.key[slot] = key; return; }
  // the block's staging area is full (nearly every group of this block emits): straight to the output, one atomic each
  const uint32_t g = atomicAdd(&E.P->pc[PC_EMIT], 1u);
  if (g < E.P->emit_cap) { E.P->emit[g] = t; E.P->emit_key[g] = key; E.P->emit_val[g] = g; }
  else err |= PAIR_ERR_EMIT;
}

struct GroupCtx {
  const PairParams &P;
  const PItem *it;        // the block's items in LDS
  uint64_t perm;          // 4-bit indices (relative to `base`) of the run's items sorted by (hash, value)
  int base, i0, i1;       // perm entries [i0, i1) = this qname group: reads first, then its soft records
  uint32_t *err;
  bool ident;             // pair_long_kernel: `it` holds the run sorted by (hash, value), no permutation
  __device__ const PItem &item(int j) const { return ident ? it[base + j] : it[base + (int)((perm >> (4 * j)) & 15u)]; }
};

// to_tread, extract.nim:63-87, from the packed scorer word and the SoA metadata
__device__ inline DTread to_tread(const PItem &x, uint32_t &err) {
  const uint32_t w = x.whole;
  DTread t;
  t.k = STRL_RES_K(w);
  t.code = STRL_RES_CODE(w);
  const uint32_t cnt = STRL_RES_COUNT(w);
  if (cnt >= 256) err |= PAIR_ERR_ASSERT;   // doAssert extract.nim:72
  const uint32_t cg = x.cig;
  // extract.nim:33 / :38: the M length of a skipped read (clip_l carries it for single-M cigars), else len(read)
  const uint32_t al = (w & STRL_RES_SKIPPED) ? (uint32_t)x.clip_l : (uint32_t)x.l_seq;
  t.tid = x.tid;
  t.position = (uint32_t)(x.pos > 0 ? x.pos : 0);
  t.flag = x.flag;
  t.count = (uint8_t)cnt;
  t.align_length = (uint8_t)al;
  t.split = STRL_SOFT_NONE;
  t.mapq = x.mapq;
  t.qid = x.val;
  const bool multi = !(cg & (STRL_CIG_ONE_OP | STRL_CIG_NONE));
  if (multi && (cg & STRL_CIG_FIRST_S) && x.clip_l > 16) t.split = STRL_SOFT_NONE_LEFT;
  if (multi && (cg & STRL_CIG_LAST_S) && x.clip_r > 16) t.split = STRL_SOFT_NONE_RIGHT;
  return t;
}

// add_soft, extract.nim:93-132, consuming the soft-clip records that joined the group (records without a result
// under either threshold never joined: they could only `continue` at :117)
__device__ inline void add_soft(const GroupCtx &G, const PItem &x, bool first_seen, uint32_t read_k, const EmitStage &E, uint64_t vidx, uint32_t &seq) {
  const PairParams &P = G.P;
  if (x.mapq < P.min_mapq) return;
  const uint32_t cg = x.cig;
  if ((cg & STRL_CIG_NONE) || !(cg & (STRL_CIG_FIRST_S | STRL_CIG_LAST_S))) return;
  for (int q = 0; q < 2; ++q) {
    // cig_index in [0, L-1]: with a single op both iterations look at op 0
    const int side = (q == 0 || (cg & STRL_CIG_ONE_OP)) ? 0 : 1;
    if (!(cg & (side == 0 ? STRL_CIG_FIRST_S : STRL_CIG_LAST_S))) continue;
    const uint32_t clen = side == 0 ? x.clip_l : x.clip_r;
    if (read_k == 0 && clen <= 16) continue;
    uint32_t w = 0;
    const uint32_t want = (x.val << 1) | (uint32_t)side;
    for (int j = G.i0; j < G.i1; ++j) {
      const PItem &s = G.item(j);
      if (!(s.val & 0x80000000u)) continue;
      if ((uint32_t)s.tid == want) { w = first_seen ? (uint32_t)s.pos : (uint32_t)s.mtid; break; }
    }
    const uint32_t cnt = STRL_RES_COUNT(w);
    if (cnt == 0) continue;
    if (cnt >= 256) *G.err |= PAIR_ERR_ASSERT;
    DTread t;
    t.k = STRL_RES_K(w);
    t.code = STRL_RES_CODE(w);
    t.tid = x.tid;
    const int32_t p = side == 0 ? x.pos : x.end;
    t.position = (uint32_t)(p > 0 ? p : 0);
    t.flag = x.flag;
    t.count = (uint8_t)cnt;
    t.align_length = (uint8_t)clen;
    t.split = side == 0 ? STRL_SOFT_LEFT : STRL_SOFT_RIGHT;
    t.mapq = x.mapq;
    t.qid = x.val;
    if (p_repeat(t) < 0.9) continue;
    emit(E, t, vidx, seq, *G.err);
  }
}

// Cache.add, extract.nim:192-248, for one record of the group
__device__ inline void cache_add(const GroupCtx &G, const PItem &x, uint64_t vidx, bool &stored, DTread &S, const EmitStage &E) {
  const PairParams &P = G.P;
  if (x.flag & (F_SECONDARY | F_SUPPL)) return;   // extract.nim:309,327
  const bool after_mate = x.tid > x.mtid || (x.tid == x.mtid && (x.pos > x.mpos || (x.pos == x.mpos && stored)));
  uint32_t seq = 0;
  if (after_mate) {
    if (!stored) return;
    DTread mate = S;
    stored = false;
    DTread self = to_tread(x, *G.err);
    add_soft(G, x, false, self.k, E, vidx, seq);
    if (mate.count == 0 && self.count == 0) return;
    if (unplaced_pair(self, mate, P)) {
      if (self.k == 0 || mate.k == 0) return;
      self.code = canonical_repeat(self.code, self.k);
      self.position = 0;
      self.tid = -1;
      mate.code = canonical_repeat(mate.code, mate.k);
      mate.position = 0;
      mate.tid = -1;
      emit(E, self, vidx, seq, *G.err);
      emit(E, mate, vidx, seq, *G.err);
      return;
    }
    const uint32_t mp = mate.position;
    if (adjust_by(mate, self, P, self.position)) emit(E, mate, vidx, seq, *G.err);
    if (adjust_by(self, mate, P, mp)) emit(E, self, vidx, seq, *G.err);
  } else {
    const DTread tr = to_tread(x, *G.err);
    add_soft(G, x, true, tr.k, E, vidx, seq);
    if (stored) stored = false;   // hasKeyOrPut hit: warn + take, the new tread is not stored (:245-248)
    else { S = tr; stored = true; }
  }
}

// Replay of one hash run of m items (G.item(0..m)): per qname group -- equal 64-bit keys, reads in file order before their
// soft-clip records -- the first pass over all records (extract.nim:308-322), then the unmapped tail once more (:326-329)
template <class Ctx> __device__ inline void replay_run(Ctx &G, int m, const EmitStage &E) {
  const PairParams &P = G.P;
  int a = 0;
  while (a < m) {
    int b = a + 1;
    while (b < m && G.item(b).key == G.item(a).key) ++b;
    G.i0 = a; G.i1 = b;
    if (P.qref)
      for (int j = a + 1; j < b; ++j) {
        const PItem &x = G.item(j);
        if (x.val & 0x80000000u) break;
        if (qname_differs(P, G.item(a).val, x.val)) *G.err |= PAIR_ERR_COLLISION;
      }
    bool stored = false;
    DTread S{};
    for (int j = a; j < b; ++j) {
      const PItem &x = G.item(j);
      if (x.val & 0x80000000u) break;
      cache_add(G, x, (uint64_t)x.val, stored, S, E);
    }
    for (int j = a; j < b; ++j) {
      const PItem &x = G.item(j);
      if (x.val & 0x80000000u) break;
      if (x.val >= P.tail_start) cache_add(G, x, (uint64_t)P.n + (uint64_t)(x.val - P.tail_start), stored, S, E);
    }
    a = b;
  }
}


// A block takes 1024 consecutive sorted items.  Phase 1: every lane gathers the metadata of ITS item into LDS (all
// gathers of the block in flight together).  Phase 2: the lane of the first item of a run (equal low 32 hash bits)
// replays Cache.add for the run out of LDS; emitted treads are staged in LDS.  Phase 3: one atomic reserves the block's
// output space, all lanes copy the staging area out.
__global__ __launch_bounds__(PG_BLOCK) void pair_groups_kernel(PairParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pg_lds[];
  PItem *items = reinterpret_cast<PItem *>(pg_lds);
  strl_tread *st_t = reinterpret_cast<strl_tread *>(pg_lds + sizeof(PItem) * (PG_BLOCK + PG_HALO));
  uint64_t *st_k = reinterpret_cast<uint64_t *>(st_t + PG_EMIT);
  uint32_t *sh = reinterpret_cast<uint32_t *>(st_k + PG_EMIT);   // [0] emitted count, [1] output base
  uint32_t n_items = P.pc[PC_ITEMS];
  if (n_items > P.item_cap) n_items = P.item_cap;
  const EmitStage E{st_t, st_k, sh, &P};
  for (uint32_t b0 = blockIdx.x * (uint32_t)PG_BLOCK; b0 < n_items; b0 += gridDim.x * (uint32_t)PG_BLOCK) {
    if (threadIdx.x == 0) sh[0] = 0;
    // ---- phase 1: gather ----
    for (uint32_t l = threadIdx.x; l < (uint32_t)(PG_BLOCK + PG_HALO); l += PG_BLOCK) {
      const uint32_t i = b0 + l;
      PItem x{};
      x.key = ~0ull;
      if (i < n_items) {
        x.key = P.item_key[i];
        x.val = P.item_val[i];
        if (x.val & 0x80000000u) {
          const strl_soft_rec s = P.soft[x.val & 0x7fffffffu];
          x.tid = (int32_t)s.read_side; x.pos = (int32_t)s.res_first; x.mtid = (int32_t)s.res_after;
        } else {
          const uint32_t r = x.val;
          const uint4 *rp = reinterpret_cast<const uint4 *>(P.rec + r);
          union { uint4 q[2]; strl_pair_rec o; } u;
          u.q[0] = rp[0];
          u.q[1] = rp[1];
          x.tid = u.o.tid; x.pos = u.o.pos; x.mtid = u.o.mtid; x.mpos = u.o.mpos; x.end = u.o.end;
          x.whole = P.whole[r]; x.flag = u.o.flag; x.clip_l = u.o.clip_l; x.clip_r = u.o.clip_r; x.l_seq = u.o.l_seq;
          x.mapq = u.o.mapq; x.cig = u.o.cig;
        }
      }
      items[l] = x;
    }
    __syncthreads();
    // ---- phase 2: replay ----
    const uint32_t i = b0 + threadIdx.x;
    uint32_t err = 0;
    if (i < n_items) {
      const int me = (int)threadIdx.x;
      const uint32_t lo = (uint32_t)items[me].key;
      const bool head = i == 0 || (me > 0 ? (uint32_t)items[me - 1].key : (uint32_t)P.item_key[i - 1]) != lo;
      if (head) {
        // the run, as a permutation sorted by (hash, value): reads (ascending record index) before the soft-clip
        // records of the same hash
        uint64_t perm = 0;
        int m = 0;
        bool too_long = false;
        for (int j = me; j < PG_BLOCK + PG_HALO && b0 + (uint32_t)j < n_items; ++j) {
          if ((uint32_t)items[j].key != lo) break;
          if (m == PAIR_MAXM) { too_long = true; break; }
          const uint64_t k = items[j].key;
          const uint32_t v = items[j].val;
          int q = m;
          while (q > 0) {
            const PItem &o = items[me + (int)((perm >> (4 * (q - 1))) & 15u)];
            if (o.key > k || (o.key == k && o.val > v)) --q; else break;
          }
          const uint64_t lowmask = (1ull << (4 * q)) - 1ull;
          perm = (perm & lowmask) | ((uint64_t)(j - me) << (4 * q)) | ((perm & ~lowmask) << 4);
          ++m;
        }
        if (too_long) {            // a block of its own replays this run (pair_long_kernel)
          const uint32_t sl = atomicAdd(&P.pc[PC_SPILL], 1u);
          if (sl < PAIR_SPILL_CAP) P.spill[sl] = i;
          else err |= PAIR_ERR_RUN;
        } else {
          GroupCtx G{P, items, perm, me, 0, 0, &err, false};
          replay_run(G, m, E);
        }
      }
    }
    if (err) atomicOr(&P.pc[PC_ERR], err);
    __syncthreads();
    // ---- phase 3: write the block's treads ----
    uint32_t ne = sh[0];
    if (ne > (uint32_t)PG_EMIT) ne = PG_EMIT;
    if (threadIdx.x == 0) sh[1] = ne ? atomicAdd(&P.pc[PC_EMIT], ne) : 0u;
    __syncthreads();
    const uint32_t base = sh[1];
    for (uint32_t e = threadIdx.x; e < ne; e += PG_BLOCK) {
      const uint32_t d = base + e;
      if (d < P.emit_cap) {
        const uint4 *src = reinterpret_cast<const uint4 *>(st_t + e);
        uint4 *dst = reinterpret_cast<uint4 *>(P.emit + d);
        dst[0] = src[0]; dst[1] = src[1];
        P.emit_key[d] = st_k[e];
        P.emit_val[d] = d;
      } else atomicOr(&P.pc[PC_ERR], PAIR_ERR_EMIT);
    }
    __syncthreads();
  }
}

// Hash runs of more than PAIR_MAXM items -- several qname groups whose mixed hashes share their low 32 bits (a whole
// genome's ~10^7 hot groups hold a few triples), or one qname carried by many records -- one block per run: the items are
// gathered into LDS, ordered by (hash, value) with a rank sort, and one lane replays them like the block replay does.
// Runs beyond PAIR_LONG_MAX items set PAIR_ERR_RUN (the caller falls back to the host's string-keyed Cache).
__global__ __launch_bounds__(PAIR_LONG_MAX) void pair_long_kernel(PairParams P) {
  __shared__ PItem raw[PAIR_LONG_MAX], srt[PAIR_LONG_MAX];
  __shared__ strl_tread st_t[PG_EMIT];
  __shared__ uint64_t st_k[PG_EMIT];
  __shared__ uint32_t sh[4];
  uint32_t n_items = P.pc[PC_ITEMS];
  if (n_items > P.item_cap) n_items = P.item_cap;
  uint32_t n_spill = P.pc[PC_SPILL];
  if (n_spill > PAIR_SPILL_CAP) n_spill = PAIR_SPILL_CAP;
  const EmitStage E{st_t, st_k, sh, &P};
  for (uint32_t e = blockIdx.x; e < n_spill; e += gridDim.x) {
    const uint32_t i0 = P.spill[e];
    const uint32_t lo = (uint32_t)P.item_key[i0];
    if (threadIdx.x == 0) { sh[0] = 0; sh[3] = PAIR_LONG_MAX; }
    __syncthreads();
    // the run is contiguous from i0: its length = the first index that is not part of it
    const uint32_t i = i0 + threadIdx.x;
    const bool mine = i < n_items && (uint32_t)P.item_key[i] == lo;
    if (!mine) atomicMin(&sh[3], threadIdx.x);
    __syncthreads();
    const uint32_t m = sh[3];
    const bool beyond = m == (uint32_t)PAIR_LONG_MAX && i0 + PAIR_LONG_MAX < n_items && (uint32_t)P.item_key[i0 + PAIR_LONG_MAX] == lo;
    uint32_t err = 0;
    if (beyond) { if (threadIdx.x == 0) atomicOr(&P.pc[PC_ERR], PAIR_ERR_RUN); __syncthreads(); continue; }
    if (threadIdx.x < m) {
      PItem x{};
      x.key = P.item_key[i];
      x.val = P.item_val[i];
      if (x.val & 0x80000000u) {
        const strl_soft_rec s = P.soft[x.val & 0x7fffffffu];
        x.tid = (int32_t)s.read_side; x.pos = (int32_t)s.res_first; x.mtid = (int32_t)s.res_after;
      } else {
        const strl_pair_rec o = P.rec[x.val];
        x.tid = o.tid; x.pos = o.pos; x.mtid = o.mtid; x.mpos = o.mpos; x.end = o.end;
        x.whole = P.whole[x.val]; x.flag = o.flag; x.clip_l = o.clip_l; x.clip_r = o.clip_r; x.l_seq = o.l_seq;
        x.mapq = o.mapq; x.cig = o.cig;
      }
      raw[threadIdx.x] = x;
    }
    __syncthreads();
    if (threadIdx.x < m) {      // rank sort by (hash, value): (key, val) pairs are distinct
      const PItem x = raw[threadIdx.x];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < m; ++j) {
        const PItem &o = raw[j];
        if (o.key < x.key || (o.key == x.key && o.val < x.val)) ++rank;
      }
      srt[rank] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      GroupCtx G{P, srt, 0, 0, 0, 0, &err, true};
      replay_run(G, (int)m, E);
      if (err) atomicOr(&P.pc[PC_ERR], err);
    }
    __syncthreads();
    uint32_t ne = sh[0];
    if (ne > (uint32_t)PG_EMIT) ne = PG_EMIT;
    if (threadIdx.x == 0) sh[1] = ne ? atomicAdd(&P.pc[PC_EMIT], ne) : 0u;
    __syncthreads();
    const uint32_t base = sh[1];
    for (uint32_t k = threadIdx.x; k < ne; k += blockDim.x) {
      const uint32_t d = base + k;
      if (d < P.emit_cap) { P.emit[d] = st_t[k]; P.emit_key[d] = st_k[k]; P.emit_val[d] = d; }
      else atomicOr(&P.pc[PC_ERR], PAIR_ERR_EMIT);
    }
    __syncthreads();
  }
}

// treads in the order of the reference's .bin file
__global__ __launch_bounds__(256) void pair_order_kernel(const uint32_t *pc, uint32_t emit_cap, const strl_tread *emit, const uint32_t *perm,
                                                         strl_tread *out, uint32_t *n_out) {
  uint32_t n = pc[PC_EMIT];
  if (n > emit_cap) n = emit_cap;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) *n_out = n;
  if (i >= n) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(emit + perm[i]);
  uint4 *dst = reinterpret_cast<uint4 *>(out + i);
  dst[0] = src[0];
  dst[1] = src[1];
}

// The pair rules on single treads, for known-answer tests of exactly the device functions the replay calls
__device__ inline DTread dtread_from(const strl_tread &t) {
  DTread d{};
  d.tid = t.tid; d.position = t.position; d.flag = t.flag; d.split = t.split; d.mapq = t.mapping_quality; d.count = t.repeat_count;
  d.align_length = t.align_length; d.qid = (uint32_t)t.qname_id;
  for (int j = 0; j < 6 && t.repeat[j]; ++j) {
    uint32_t c = 1;
    switch (t.repeat[j]) { case 'C': c = 0; break; case 'A': c = 1; break; case 'T': c = 2; break; case 'G': c = 3; break; }
    d.code = (d.code << 2) | c;
    ++d.k;
  }
  return d;
}
__device__ inline void dtread_to(const DTread &d, strl_tread &t) {
  t.tid = d.tid; t.position = d.position; t.flag = d.flag; t.split = d.split; t.mapping_quality = d.mapq; t.repeat_count = d.count;
  t.align_length = d.align_length;
  for (int j = 0; j < 6; ++j) t.repeat[j] = (uint32_t)j < d.k ? "CATG"[(d.code >> (2 * (d.k - 1 - j))) & 3u] : (char)0;
}
__global__ void pair_rules_kernel(int op, strl_tread *A, const strl_tread *B, PairParams P, uint32_t B_position, int *res) {
  DTread a = dtread_from(*A);
  const DTread b = dtread_from(*B);
  if (op == 0) *res = adjust_by(a, b, P, B_position) ? 1 : 0;
  else if (op == 1) *res = unplaced_pair(a, b, P) ? 1 : 0;
  else { a.code = canonical_repeat(a.code, a.k); *res = 0; }
  dtread_to(a, *A);
}

}  // namespace strl

using namespace strl;

extern "C" int strl_pair_rule_device(strl_ctx *c, int op, strl_tread *A, const strl_tread *B, const strl_opts *o, uint32_t B_position, int *result) {
  if (!c || !A || !B || !o || !result || op < 0 || op > 2) { set_error("bad argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  DevBuf buf;
  int rc;
  if ((rc = buf.reserve(2 * sizeof(strl_tread) + 16))) return rc;
  strl_tread *dA = buf.as<strl_tread>(), *dB = dA + 1;
  int *dres = reinterpret_cast<int *>(dB + 1);
  STRL_HIP(hipMemcpyAsync(dA, A, sizeof *A, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipMemcpyAsync(dB, B, sizeof *B, hipMemcpyHostToDevice, c->stream));
  PairParams P{};
  P.p = o->proportion_repeat; P.min_mapq = o->min_mapq; P.frag_median = o->median_fragment_length;
  hipLaunchKernelGGL(pair_rules_kernel, dim3(1), dim3(1), 0, c->stream, op, dA, dB, P, B_position, dres);
  STRL_HIP(hipGetLastError());
  STRL_HIP(hipMemcpyAsync(A, dA, sizeof *A, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipMemcpyAsync(result, dres, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  buf.release();
  return STRL_OK;
}

// Treads of the last strl_pair_device call into the order of the reference's .bin file (c->treads); idempotent.
int strl_pair_order(strl_ctx *c, hipStream_t on_stream) {
  const bool on_main = !on_stream || on_stream == c->stream;
  // on the main stream: an overlapped clustering may still read the unordered treads; on the tail's own side stream the
  // order of the launches is the order of the work
  if (on_main) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (!c->n_treads_dev) { set_error("no strl_extract_device call on this context"); return STRL_ERR_ARG; }
  if (c->pair_ordered) return STRL_OK;
  hipStream_t st = on_main ? c->stream : on_stream;
  const uint32_t ecap = c->tread_cap;
  uint64_t *ok = nullptr;
  uint32_t *ov = nullptr;
  const int e = radix_sort_pairs(st, c->pair_cnt.as<uint32_t>() + PC_EMIT, ecap, c->po_key, c->po_val, c->po_key_alt, c->po_val_alt, c->sort_scratch.p,
                                 c->sort_scratch.cap, 0, c->po_bits, &ok, &ov);
  if (e) { set_error("radix_sort_pairs failed: %s", hipGetErrorString((hipError_t)e)); return STRL_ERR_HIP; }
  uint32_t *n_out = reinterpret_cast<uint32_t *>(c->treads.as<uint8_t>() + (size_t)ecap * sizeof(strl_tread));
  hipLaunchKernelGGL(pair_order_kernel, dim3((ecap + 255) / 256), dim3(256), 0, st, c->pair_cnt.as<uint32_t>(), ecap, c->p_emit.as<strl_tread>(), ov,
                     c->treads.as<strl_tread>(), n_out);
  STRL_HIP(hipGetLastError());
  c->n_treads_dev = n_out;
  c->pair_ordered = true;
  return STRL_OK;
}

// Enqueue the pair logic behind a scoring pass of the same batch (score_device has run on c->stream with the pairing
// arrays given, so the whole-read marks are in the bitmap).  Everything is asynchronous; results stay on the device:
// c->treads[0, *c->n_treads).
int strl_pair_device(strl_ctx *c, uint64_t n, const strl_pair_soa *pp, const uint32_t *whole, const strl_soft_rec *soft,
                     const uint32_t *d_n_soft, uint64_t soft_cap, int64_t n_tail, uint64_t item_cap, uint64_t tread_cap, hipStream_t on_stream) {
  if (n > strl_record_limit()) { set_error("pair logic: more than %llu records in one device pass", (unsigned long long)strl_record_limit()); return STRL_ERR_LIMIT; }
  if (n_tail < 0 || (uint64_t)n_tail > n) { set_error("strl_pair_device: n_tail must be in [0, n]"); return STRL_ERR_ARG; }
  if (item_cap > 0x7ffffff0ull || tread_cap > 0x7ffffff0ull) { set_error("pair capacities too large"); return STRL_ERR_ARG; }
  hipStream_t st = on_stream ? on_stream : c->stream;
  int rc;
  // the previous batch's clustering (side stream) reads the buffers written here; on the side stream itself the order is given
  if (st == c->stream && (rc = side_join(c))) return rc;
  c->pair_on_side = false;
  const uint32_t icap = (uint32_t)std::max<uint64_t>(item_cap, 1024), ecap = (uint32_t)std::max<uint64_t>(tread_cap, 1024);
  int ebits = 3;   // emission key: (virtual record index < 2n) << 2 | sequence number
  while (ebits < 40 && ((2 * n) >> (ebits - 2))) ++ebits;
  const size_t sb = std::max(radix_sort_scratch_bytes(icap, 32), radix_sort_scratch_bytes(ecap, ebits));
  if ((rc = c->p_key0.reserve((size_t)std::max(icap, ecap) * 8)) || (rc = c->p_key1.reserve((size_t)std::max(icap, ecap) * 8)) ||
      (rc = c->p_val0.reserve((size_t)std::max(icap, ecap) * 4)) || (rc = c->p_val1.reserve((size_t)std::max(icap, ecap) * 4)) ||
      (rc = c->p_emit.reserve((size_t)ecap * sizeof(strl_tread))) || (rc = c->treads.reserve((size_t)ecap * sizeof(strl_tread) + 64)) ||
      (rc = c->sort_scratch.reserve(sb)) || (rc = c->pair_cnt.reserve(PC_WORDS * 4 + 64)) || (rc = c->p_spill.reserve((size_t)PAIR_SPILL_CAP * 4)))
    return rc;
  void *jt = nullptr;
  size_t jt_bytes = 0;
  radix_sort_tables(c->sort_scratch.p, icap, 32, &jt, &jt_bytes);           // the join sort's chunk tables: zeroed with the counters
  STRL_HIP(zero_words2(c->pair_cnt.p, PC_WORDS * 4 + 64, jt, jt_bytes, st));
  PairParams P{};
  P.n = (uint32_t)n;
  P.tail_start = (uint32_t)(n - (uint64_t)n_tail);
  P.rec = pp->rec; P.qhash = pp->qhash;
  P.whole = whole; P.soft = soft; P.d_n_soft = d_n_soft;
  P.scap = (uint32_t)std::max<uint64_t>(std::min<uint64_t>(soft_cap, 2 * n), 1);
  P.bloom = c->bloom.as<uint32_t>(); P.bloom_mask = c->bloom_mask;
  P.item_key = c->p_key0.as<uint64_t>(); P.item_val = c->p_val0.as<uint32_t>(); P.item_cap = icap;
  P.pc = c->pair_cnt.as<uint32_t>();
  P.emit = c->p_emit.as<strl_tread>(); P.emit_key = c->p_key0.as<uint64_t>(); P.emit_val = c->p_val0.as<uint32_t>(); P.emit_cap = ecap;
  P.p = c->opts.proportion_repeat; P.min_mapq = c->opts.min_mapq; P.frag_median = c->opts.median_fragment_length;
  P.spill = c->p_spill.as<uint32_t>();
  // the qnames of every record, when the device front end parsed them and this is its chunked extract (x_rows are the rows)
  const bool names = c->x_front && c->front && pp->rec == c->x_rows.as<strl_pair_rec>();
  P.qref = names ? c->front->qref.as<uint64_t>() : nullptr;
  P.qarena = names ? c->front->qarena.as<uint8_t>() : nullptr;
  hipEvent_t *ev = c->timing ? c->pev : nullptr;
  if (ev) STRL_HIP(hipEventRecord(ev[0], st));
  if (n) {
    hipLaunchKernelGGL(pair_soft_items_kernel, dim3(512), dim3(1024), 0, st, P);
    if (ev) STRL_HIP(hipEventRecord(ev[1], st));
    static const int env_p = getenv("STRL_GRID_P") ? atoi(getenv("STRL_GRID_P")) : 0;
    const int pblocks = (int)std::min<uint64_t>((n + 2047) / 2048, env_p > 0 ? (uint64_t)env_p : 1024);   // 4 resident blocks (40 KB of LDS each) per CU; measured 256..4096
    hipLaunchKernelGGL(pair_probe_kernel, dim3(pblocks), dim3(256), 0, st, P);
    STRL_HIP(hipGetLastError());
  }
  else if (ev) STRL_HIP(hipEventRecord(ev[1], st));
  if (ev) STRL_HIP(hipEventRecord(ev[2], st));
  // join: sort the items by the low 32 bits of the (mixed) hash; runs are disambiguated by the full hash in the replay
  uint64_t *ik = nullptr;
  uint32_t *iv = nullptr;
  int e = radix_sort_pairs(st, P.pc + PC_ITEMS, icap, c->p_key0.as<uint64_t>(), c->p_val0.as<uint32_t>(), c->p_key1.as<uint64_t>(),
                           c->p_val1.as<uint32_t>(), c->sort_scratch.p, c->sort_scratch.cap, 0, 32, &ik, &iv, true);
  if (e) { set_error("radix_sort_pairs failed: %s", hipGetErrorString((hipError_t)e)); return STRL_ERR_HIP; }
  if (ev) STRL_HIP(hipEventRecord(ev[3], st));
  // the sorted items sit in (ik, iv); the emission keys go to the other pair of buffers
  P.item_key = ik; P.item_val = iv;
  const bool in0 = ik == c->p_key0.as<uint64_t>();
  uint64_t *ek = in0 ? c->p_key1.as<uint64_t>() : c->p_key0.as<uint64_t>();
  uint32_t *evl = in0 ? c->p_val1.as<uint32_t>() : c->p_val0.as<uint32_t>();
  P.emit_key = ek; P.emit_val = evl;
  if (n) {
    const size_t pg_shmem = sizeof(PItem) * (PG_BLOCK + PG_HALO) + (sizeof(strl_tread) + 8) * PG_EMIT + 16;
    if (!c->pg_attr_done) {      // (per context: contexts may sit on different devices)
      STRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_groups_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pg_shmem));
      c->pg_attr_done = true;
    }
    hipLaunchKernelGGL(pair_groups_kernel, dim3((unsigned)std::min<uint32_t>((icap + PG_BLOCK - 1) / PG_BLOCK, 8192)), dim3(PG_BLOCK), pg_shmem, st, P);
    STRL_HIP(hipGetLastError());
    hipLaunchKernelGGL(pair_long_kernel, dim3(64), dim3(PAIR_LONG_MAX), 0, st, P);      // (no spilled runs: 64 blocks read one counter)
    STRL_HIP(hipGetLastError());
  }
  if (ev) STRL_HIP(hipEventRecord(ev[4], st));
  // The treads now sit in p_emit in arbitrary order, each with its emission key.  Putting them into the order of the .bin
  // file costs another sort; it is done when somebody asks for the ordered array (strl_treads_fetch, the multi-GPU
  // gather), not here: clustering only needs the keys (first appearance of a group), see strl_cluster_resident.
  c->po_key = ek; c->po_val = evl; c->po_key_alt = ik; c->po_val_alt = iv; c->po_bits = ebits;
  c->pair_ordered = false;
  c->n_treads_dev = c->pair_cnt.as<uint32_t>() + PC_EMIT;          // (clamped to tread_cap by every reader)
  if (ev) STRL_HIP(hipEventRecord(ev[5], st));
  c->tread_cap = ecap;
  c->pair_item_cap = icap;
  return STRL_OK;
}
