// cluster.hip -- STR-read clustering on the GPU (call.nim:118-130,223-235; merge.nim:121-187;
// cluster.nim:175-374; callclusters.nim:52-66), gfx950.
//
// The reference groups treads in a hash table keyed by (tid, repeat), merge-sorts every group by
// position and sweeps each group sequentially.  Here the whole tread set -- a host array, or the treads the device
// pair logic left resident in the context -- goes through
//   tread_keys_kernel : tread -> composite key (tid, unit) << pos_bits | position,
//   radix sort        : ONE stable LSD sort of the composite key (sort.hip; two sorts when the key needs > 64 bits),
//   heads / gather    : group heads counted per tile, then group ids, the sorted payload and the per-group tables in one
//                       launch (a tile sums the head counts of the tiles before it: no scan kernel),
//   ends_kernel       : one tread per lane computes where the cluster that STARTS at it would end -- the reference's
//                       growth rule depends only on the start: <= 9 warm-up steps of the running median-of-first-9,
//                       then a binary search for pos > posmed + max_dist + 100,
//   walk_kernel       : one lane per group follows start -> end links and flags the cluster heads,
//   bounds_filter_kernel / bounds_rows_kernel : trim + support gate per cluster head, then anchor gate, split_cluster,
//                       bounds() and the callclusters gate, with Nim CountTable.largest slot-order tie-breaks
//                       reproduced on the device (nim_tables.h).
// Every count (treads, groups, candidates) lives on the device; launches are sized by host-known upper bounds, so the
// whole pass is enqueued without a host round trip (12 + number-of-sort-passes launches).
// HBM-bound integer work in principle; at ~1e6 treads per 30x sample it is bound by launch boundaries and dependent-load
// latency (5e7 treads for a 50-sample merge).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>
#include "common.h"
#include "nim_tables.h"
#include "sort.h"
#include "device_util.h"

namespace nim { std::vector<int64_t> table_slot_order(const std::vector<uint64_t> &hcodes, uint64_t initial_size); }

namespace strl {

struct RawBounds {  // one candidate row per (candidate cluster, half)
  uint32_t valid;   // 1 = passes every gate
  uint32_t first;   // sorted index of the first read
  uint32_t gid;     // group of the cluster
  uint32_t left, left_most, right, right_most, center_mass;
  uint16_t n_left, n_right, n_total, pad;
};

// device-side counters of one clustering pass
constexpr int CC_N = 0, CC_NGROUPS = 1, CC_NCAND = 2, CC_NCLUSTERS = 3, CC_BIG = 4, CC_ERR = 5, CC_NSEEN = 6, CC_WORDS = 16;   // CC_NSEEN: *d_n as the pass saw it
constexpr uint32_t CERR_UNIT = 1u, CERR_TID = 2u, CERR_POS = 4u, CERR_CAND = 8u, CERR_BIG = 16u;

struct ClusterParams {
  const uint32_t *d_n;      // number of treads (device)
  uint32_t n_max;
  const uint32_t *pos;      // sorted
  const uint8_t *split;     // sorted
  const uint32_t *sample;   // sorted (qname_id)
  const uint32_t *gid;      // group index of each sorted tread
  const uint32_t *gstart;   // [n_groups+1]
  const uint8_t *gplaced;   // [n_groups] 1 if tid >= 0
  uint32_t *ends;           // e(s)
  uint32_t *is_start;       // cluster-head flags
  uint32_t *cnt;            // CC_* counters
  uint32_t *big;            // scratch of clusters too large for LDS, handed out with a bump cursor
  uint32_t big_words;
  uint4 *cand;              // clusters that pass the support gate: {cluster start, first read after trim, end, group}
  uint32_t cand_cap;
  RawBounds *out;           // [2 * cand_cap]
  uint32_t max_dist;
  int32_t min_support;
  uint32_t min_clip, min_clip_total, max_clip_dist;
  int32_t mode;
};

__device__ __forceinline__ uint32_t n_of(const ClusterParams &P) {
  const uint32_t n = *P.d_n;
  return n < P.n_max ? n : P.n_max;
}

__device__ __forceinline__ uint32_t posmed_at(const uint32_t *pos, uint32_t s, uint32_t n) {
  const uint32_t m = n < 9u ? n : 9u;         // cluster.nim:59-62: reads[int(min(9, n) / 2 - 0.5)]
  return pos[s + ((m - 1u) >> 1)];
}

__global__ __launch_bounds__(256) void ends_kernel(ClusterParams P) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_of(P)) return;
  P.is_start[s] = 0u;
  const uint32_t g = P.gid[s];
  const uint32_t ge = P.gstart[g + 1];
  const uint32_t add = P.max_dist + 100u;     // uint32 arithmetic as in cluster.nim:336
  uint32_t n = 1, j = s + 1;
  bool open = true;
  while (j < ge && n < 9u) {                  // warm-up: the median still moves
    if (P.pos[j] <= posmed_at(P.pos, s, n) + add) { ++n; ++j; }
    else { open = false; break; }
  }
  if (open && j < ge) {                       // n == 9: median fixed at pos[s+4]
    const uint32_t limit = P.pos[s + 4] + add;
    uint32_t lo = j, hi = ge;
    while (lo < hi) {                         // first index with pos > limit
      const uint32_t mid = (lo + hi) >> 1;
      if (P.pos[mid] <= limit) lo = mid + 1; else hi = mid;
    }
    j = lo;
  }
  P.ends[s] = j;
}

__global__ __launch_bounds__(64) void walk_kernel(ClusterParams P) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nc = 0;
  if (g < P.cnt[CC_NGROUPS] && P.gplaced[g]) {
    uint32_t s = P.gstart[g];
    const uint32_t ge = P.gstart[g + 1];
    while (s < ge) {                          // trcluster: the next cluster starts at the first rejected read
      P.is_start[s] = 1u;
      s = P.ends[s];
      ++nc;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) nc += __shfl_down(nc, d);
  if (threadIdx.x == 0 && nc) atomicAdd(&P.cnt[CC_NCLUSTERS], nc);
}

// Where the sorted reads of the cluster being worked on live: the global arrays, or a copy the wave staged in LDS
// (indices stay the global sorted indices).
struct GlobalView {
  const uint32_t *p; const uint8_t *sp; const uint32_t *sm;
  __device__ uint32_t pos(uint32_t i) const { return p[i]; }
  __device__ uint8_t split(uint32_t i) const { return sp[i]; }
  __device__ uint32_t sample(uint32_t i) const { return sm[i]; }
};
struct LdsView {
  const uint32_t *p; const uint8_t *sp; const uint32_t *sm; uint32_t base;
  __device__ uint32_t pos(uint32_t i) const { return p[i - base]; }
  __device__ uint8_t split(uint32_t i) const { return sp[i - base]; }
  __device__ uint32_t sample(uint32_t i) const { return sm[i - base]; }
};

// CountTable over the clip positions of one kind in [a, b) (sorted by position => equal keys adjacent).
template <bool FILTER, class V>
__device__ void clip_table(const ClusterParams &P, const V &R, uint32_t a, uint32_t b, uint8_t kind, int32_t cm, nim::CountTable &ct,
                           uint32_t *scratch, uint32_t cap, uint32_t &n_reads, uint32_t &n_distinct) {
  ct.init(scratch, cap);
  n_reads = 0;
  n_distinct = 0;
  uint32_t run_key = 0, run = 0;
  for (uint32_t i = a; i < b; ++i) {
    if (R.split(i) != kind) continue;
    const uint32_t p = R.pos(i);
    if (FILTER) {                              // cluster.nim:193,197
      if (kind == STRL_SOFT_LEFT && !((int32_t)p < cm + (int32_t)P.max_clip_dist)) continue;
      if (kind == STRL_SOFT_RIGHT && !((int32_t)p > cm - (int32_t)P.max_clip_dist)) continue;
    }
    ++n_reads;
    if (run && p == run_key) { ++run; continue; }
    if (run) { ct.insert_new(run_key, run); ++n_distinct; }
    run_key = p;
    run = 1;
  }
  if (run) { ct.insert_new(run_key, run); ++n_distinct; }
}

// bounds() (cluster.nim:175-250) + gate (callclusters.nim:52-66) for reads [a, b)
template <class V>
__device__ void emit_bounds(const ClusterParams &P, const V &R, uint32_t a, uint32_t b, uint32_t cl_left_most, uint32_t cl_right_most,
                            uint32_t *scratch, uint32_t cap, RawBounds &o) {
  o.valid = 0;
  o.first = a;
  const uint32_t n = b - a;
  if (P.mode == STRL_MODE_MERGE) {            // has_per_sample_reads, merge.nim:18-25: small open-addressing count table
    uint32_t tl = 16;
    while (tl < 2 * n) tl <<= 1;              // <= 4n <= scratch region (4 * (16 + 3n) dwords)
    uint32_t *k = scratch, *v = scratch + tl;
    for (uint32_t i = 0; i < tl; ++i) v[i] = 0;
    uint32_t best = 0;
    for (uint32_t i = a; i < b; ++i) {
      const uint32_t sm = R.sample(i);
      uint32_t h = (sm * 0x9E3779B1u) & (tl - 1);
      while (v[h] && k[h] != sm) h = (h + 1) & (tl - 1);
      k[h] = sm;
      const uint32_t c = ++v[h];
      best = c > best ? c : best;
    }
    if ((int32_t)best < P.min_support) return;
  }
  if (n >= 65535u) return;                    // callclusters.nim:53-55
  const uint32_t center = R.pos(a + (n >> 1));
  const int32_t cm = (int32_t)center;
  nim::CountTable ct;
  uint32_t nl, nr, dl, dr, left = 0, right = 0, key, val;
  clip_table<true>(P, R, a, b, STRL_SOFT_LEFT, cm, ct, scratch, cap, nl, dl);
  if (dl) { ct.largest(key, val); if (val > 1) left = key; }          // cluster.nim:204-207
  clip_table<true>(P, R, a, b, STRL_SOFT_RIGHT, cm, ct, scratch, cap, nr, dr);
  if (dr) { ct.largest(key, val); if (val > 1) right = key; }         // :208-211
  if (left == 0) left = center;                                       // :214-217
  if (right == 0) right = left + 1;
  if (left >= right) {                                                // :227-231
    if (nl > 0 && nr > 0) { const uint32_t t = left; left = right; right = t; }
    else left = right - 1;
  }
  uint32_t lm = cl_left_most > 0 ? cl_left_most : R.pos(a);           // :234-241 (posns sorted: min/max are the ends)
  uint32_t rm = cl_right_most > 0 ? cl_right_most : R.pos(b - 1);
  if (lm > left) lm = left;                                           // :244-247
  if (rm < right) rm = right;
  if (right - left > 1000u) return;                                   // callclusters.nim:57-59
  if (nl < P.min_clip) return;                                        // :62-65
  if (nr < P.min_clip) return;
  if (((nl + nr) & 0xffffu) < P.min_clip_total) return;
  o.left = left; o.right = right; o.left_most = lm; o.right_most = rm; o.center_mass = center;
  o.n_left = (uint16_t)nl; o.n_right = (uint16_t)nr; o.n_total = (uint16_t)n; o.pad = 0;
  o.valid = 1;
}

// has_anchor, split_cluster and bounds() of one cluster that passed the support gate (reads [s, e) after trim)
template <class V>
__device__ void cluster_rows(const ClusterParams &P, const V &R, uint32_t s, uint32_t e, uint32_t left_most, uint32_t right_most,
                             uint32_t *scratch, uint32_t cap, RawBounds *o) {
  bool anchor = false;
  for (uint32_t i = s; i < e && !anchor; ++i) anchor = R.split(i) == STRL_SOFT_NONE; // has_anchor :275-281
  if (!anchor) return;
  // split_cluster, cluster.nim:283-320
  nim::CountTable ct;
  uint32_t nl, nr, dl, dr, llk = 0, llv = 0, rlk = 0, rlv = 0;
  clip_table<false>(P, R, s, e, STRL_SOFT_RIGHT, 0, ct, scratch, cap, nr, dr);
  if (dr) ct.largest(rlk, rlv);
  clip_table<false>(P, R, s, e, STRL_SOFT_LEFT, 0, ct, scratch, cap, nl, dl);
  if (dl) ct.largest(llk, llv);
  if (dr && dl && rlk < llk && (int64_t)rlv >= P.min_support && (int64_t)llv >= P.min_support &&
      (double)llv / (double)dl > 0.5 && (double)rlv / (double)dr > 0.5) {
    const uint32_t mid = (uint32_t)(0.5 + ((double)rlk + (double)llk) / 2.0);
    uint32_t m = s;
    while (m < e && R.pos(m) < mid) ++m;
    emit_bounds(P, R, s, m, 0u, mid - 1u, scratch, cap, o[0]);                      // :313
    emit_bounds(P, R, m, e, mid, 0u, scratch, cap, o[1]);                           // :314
  } else {
    emit_bounds(P, R, s, e, left_most, right_most, scratch, cap, o[0]);
  }
}

// bounds in two launches.  bounds_filter_kernel: one lane per sorted tread; a lane whose tread starts a cluster does the
// cheap part (trim, support gate: 98 % of the clusters of a WGS sample end there); survivors are appended to a candidate
// list with one atomic per 1024 treads.  bounds_rows_kernel: one WAVE per candidate -- all lanes copy the cluster's
// reads into LDS in one coalesced round trip, then one lane runs the sequential CountTable logic out of LDS (its
// tables live there too).
constexpr uint32_t BK_LIM = 256;   // reads of a cluster staged in LDS; larger clusters use the global arrays
__global__ __launch_bounds__(1024) void bounds_filter_kernel(ClusterParams P) {
  __shared__ uint32_t wcnt[16];
  __shared__ uint32_t base_sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t s0 = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nn = n_of(P);
  uint32_t s = s0, e = 0;
  bool cand = false;
  if (s0 < nn && P.is_start[s0]) {
    e = P.ends[s0];
    uint32_t n = e - s;
    // trim(max_dist + 100), cluster.nim:252-257 -- `lo` is computed once
    const uint32_t md = P.max_dist + 100u;
    const int64_t lo64 = (int64_t)posmed_at(P.pos, s, n) - (int64_t)md;
    const uint32_t lo = lo64 < 0 ? 0u : (uint32_t)lo64;
    while (n > 1 && P.pos[s] < lo) { ++s; --n; }
    cand = (int64_t)n >= (int64_t)P.min_support;                                      // :346
  }
  const unsigned long long m = __ballot(cand);
  if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < 16; ++w) { const uint32_t x = wcnt[w]; wcnt[w] = tot; tot += x; }
    base_sh = tot ? atomicAdd(&P.cnt[CC_NCAND], tot) : 0u;
  }
  __syncthreads();
  if (cand) {
    const uint32_t d = base_sh + wcnt[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (d < P.cand_cap) P.cand[d] = make_uint4(s0, s, e, P.gid[s0]);
    else atomicOr(&P.cnt[CC_ERR], CERR_CAND);
  }
}

__global__ __launch_bounds__(64) void bounds_rows_kernel(ClusterParams P) {
  __shared__ uint32_t l_pos[BK_LIM], l_sample[BK_LIM];
  __shared__ uint8_t l_split[BK_LIM];
  __shared__ uint32_t l_scratch[4 * (16 + 3 * BK_LIM)];
  const int lane = threadIdx.x;
  uint32_t n_cand = P.cnt[CC_NCAND];
  if (n_cand > P.cand_cap) n_cand = P.cand_cap;
  for (uint32_t k = blockIdx.x; k < n_cand; k += gridDim.x) {
    const uint4 cd = P.cand[k];
    const uint32_t cs = cd.y, ce = cd.z, cn = ce - cs;
    RawBounds *o = P.out + 2 * (uint64_t)k;
    const bool in_lds = cn <= BK_LIM;
    if (in_lds) {
      for (uint32_t i = lane; i < cn; i += 64) {
        l_pos[i] = P.pos[cs + i];
        l_split[i] = P.split[cs + i];
        l_sample[i] = P.sample[cs + i];
      }
      __syncthreads();
    }
    if (lane == 0) {
      o[0].valid = 0; o[1].valid = 0;
      o[0].gid = cd.w; o[1].gid = cd.w;
      o[0].first = cs; o[1].first = cs;
      const uint32_t n = cn;
      const uint32_t pm = in_lds ? l_pos[((n < 9u ? n : 9u) - 1u) >> 1] : posmed_at(P.pos, cs, n);
      const uint32_t last = in_lds ? l_pos[n - 1] : P.pos[ce - 1], firstp = in_lds ? l_pos[0] : P.pos[cs];
      const uint32_t right_most = last > pm + P.max_dist ? last : pm + P.max_dist;    // :343
      const uint32_t left_most = firstp < pm - P.max_dist ? firstp : pm - P.max_dist; // :344 (uint32 wrap kept)
      if (in_lds) cluster_rows(P, LdsView{l_pos, l_split, l_sample, cs}, cs, ce, left_most, right_most, l_scratch, 16u + 3u * cn, o);
      else {
        // large cluster: its tables come from the global scratch (bump cursor; at most n / 256 such clusters)
        const uint32_t need = 4u * (16u + 3u * cn);
        const uint32_t at = atomicAdd(&P.cnt[CC_BIG], need);
        if (at + need <= P.big_words) cluster_rows(P, GlobalView{P.pos, P.split, P.sample}, cs, ce, left_most, right_most, P.big + at, 16u + 3u * cn, o);
        else atomicOr(&P.cnt[CC_ERR], CERR_BIG);
      }
    }
    __syncthreads();
  }
}

// tread -> sort key + the payload the sweep needs.  composite: key = (tid + 1, unit) << pos_bits | position, one sort;
// else key = position and the group key goes to gkey_in for the second sort.
struct KeyParams {
  const uint32_t *d_n;
  uint32_t n_max;
  const strl_tread *treads;
  uint64_t *key;
  uint32_t *val;
  uint32_t *pos_in, *sample_in;
  uint8_t *split_in;
  uint64_t *gkey_in;
  uint64_t *first_in;           // what orders the groups' first appearances: the input index, or ...
  const uint64_t *first_key;    // ... the emission key of an unordered tread (treads straight from the pair logic), or nullptr
  uint32_t *cnt;
  int composite, pos_bits, fold;
  int32_t n_tid;    // tids must be < n_tid
};
__global__ __launch_bounds__(256) void tread_keys_kernel(KeyParams K) {
  uint32_t n = *K.d_n;
  if (blockIdx.x == 0 && threadIdx.x == 0) K.cnt[CC_NSEEN] = n;     // (the counter itself may belong to the next batch by the time the host collects)
  if (n > K.n_max) n = K.n_max;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(K.treads + i);
  union { uint4 q[2]; strl_tread t; } u;
  u.q[0] = src[0];
  u.q[1] = src[1];
  const strl_tread &t = u.t;
  uint32_t len = 0, code = 0, err = 0;
  bool ended = false;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const char ch = t.repeat[j];
    if (ch == 0) { ended = true; continue; }
    if (ended) { err |= CERR_UNIT; continue; }
    uint32_t cd = 0;
    switch (ch) { case 'C': cd = 0; break; case 'A': cd = 1; break; case 'T': cd = 2; break; case 'G': cd = 3; break; default: err |= CERR_UNIT; }
    code = (code << 2) | cd;
    ++len;
  }
  if (t.tid < -1 || t.tid >= K.n_tid) err |= CERR_TID;
  const uint64_t gkey = ((uint64_t)(uint32_t)(t.tid + 1) << 15) | ((uint64_t)len << 12) | code;
  K.pos_in[i] = t.position;
  K.split_in[i] = t.split;
  K.sample_in[i] = (uint32_t)t.qname_id;
  K.first_in[i] = K.first_key ? K.first_key[i] : (uint64_t)i;
  K.val[i] = i;
  if (K.composite) {
    // A caller-given pos_bits < 32 covers [0, 2^(pos_bits-1)) and, folded into the upper half of the field in the same
    // uint32 order, the positions adjust_by wrapped below zero (utils.nim:304-310: uint32 arithmetic near a contig start).
    uint32_t pe = t.position;
    if (!K.fold) {
      if (K.pos_bits < 32 && (pe >> K.pos_bits)) err |= CERR_POS;
    } else if (K.pos_bits < 32) {
      const uint32_t half = 1u << (K.pos_bits - 1);
      if (pe >= half) {
        if (pe < 0u - half) err |= CERR_POS;
        pe += 1u << K.pos_bits;            // mod 2^32: [2^32 - half, 2^32) -> [half, 2^pos_bits)
        pe &= (1u << K.pos_bits) - 1u;
      }
    }
    K.key[i] = (gkey << K.pos_bits) | pe;
  } else {
    K.key[i] = t.position;
    K.gkey_in[i] = gkey;
  }
  if (err) atomicOr(&K.cnt[CC_ERR], err);
}
__global__ void regather_key_kernel(const uint32_t *d_n, uint32_t n_max, const uint32_t *perm, const uint64_t *in, uint64_t *out) {
  uint32_t n = *d_n;
  if (n > n_max) n = n_max;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

// ---- group heads, ids and tables from the sorted keys ----------------------------------------------------------------
constexpr uint32_t GT_TILE = 2048;   // 4 waves x 8 rounds of 64 consecutive keys
struct GatherParams {
  const uint32_t *d_n;
  uint32_t n_max;
  const uint64_t *key;      // sorted
  const uint32_t *perm;     // sorted index -> input index
  int shift;                // group key = key >> shift
  const uint32_t *pos_in, *sample_in;
  const uint8_t *split_in;
  const uint64_t *first_in;
  uint32_t *tile_heads;     // [tiles] heads per tile (exclusive prefix when `scanned`)
  int scanned;
  uint32_t *pos, *sample, *gid;
  uint8_t *split;
  uint32_t *gstart;
  uint64_t *gfirst;
  uint64_t *gkeys;
  uint8_t *gplaced;
  uint32_t *cnt;
};
__device__ __forceinline__ bool is_head(const GatherParams &G, uint32_t j, uint64_t k) {
  return j == 0 || (G.key[j - 1] >> G.shift) != (k >> G.shift);
}
__global__ __launch_bounds__(256) void heads_kernel(GatherParams G) {
  __shared__ uint32_t wsum[4];
  uint32_t n = *G.d_n;
  if (n > G.n_max) n = G.n_max;
  const uint32_t t = blockIdx.x;
  if ((uint64_t)t * GT_TILE >= n) return;
  uint32_t c = 0;
#pragma unroll
  for (uint32_t r = 0; r < 8; ++r) {
    const uint32_t j = t * GT_TILE + r * 256u + threadIdx.x;
    if (j < n) {
      G.gfirst[j] = ~0ull;                // (n_groups <= n) first appearance, filled by atomicMin in gather_kernel
      c += is_head(G, j, G.key[j]) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) G.tile_heads[t] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// more than 1024 tiles: exclusive scan of the tile counts by one block (in place)
__global__ __launch_bounds__(1024) void tile_scan_kernel(GatherParams G) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_sh;
  uint32_t n = *G.d_n;
  if (n > G.n_max) n = G.n_max;
  const uint32_t ntiles = (n + GT_TILE - 1) / GT_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  for (uint32_t b = 0; b < ntiles; b += 1024) {
    const uint32_t i = b + threadIdx.x;
    const uint32_t v = i < ntiles ? G.tile_heads[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = carry_sh;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (i < ntiles) G.tile_heads[i] = base + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_sh = base + inc;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void gather_kernel(GatherParams G) {
  __shared__ uint32_t wsum[4], wheads[4];
  uint32_t n = *G.d_n;
  if (n > G.n_max) n = G.n_max;
  const uint32_t t = blockIdx.x;
  if ((uint64_t)t * GT_TILE >= n) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // groups that start in earlier tiles
  uint32_t prefix = 0;
  if (G.scanned) prefix = G.tile_heads[t];
  else {
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < t; i += 256) c += G.tile_heads[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    prefix = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  // wave w owns the 512 consecutive keys [w * 512, (w + 1) * 512) of the tile, 64 per round
  const uint32_t j0 = t * GT_TILE + (uint32_t)wave * 512u + (uint32_t)lane;
  uint64_t key[8];
  unsigned long long hm[8];
  uint32_t wtot = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t j = j0 + 64u * r;
    key[r] = j < n ? G.key[j] : 0ull;
    hm[r] = __ballot(j < n && is_head(G, j, key[r]));
    wtot += (uint32_t)__popcll(hm[r]);
  }
  if (lane == 0) wheads[wave] = wtot;
  __syncthreads();
  uint32_t run = prefix;
  for (int w = 0; w < wave; ++w) run += wheads[w];
  const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t j = j0 + 64u * r;
    const bool in = j < n;
    uint32_t src = 0xffffffffu, g = 0xffffffffu;
    uint64_t fk = ~0ull;
    if (in) {
      g = run + (uint32_t)__popcll(hm[r] & le) - 1u;
      src = G.perm[j];
      fk = G.first_in[src];
      G.pos[j] = G.pos_in[src];
      G.split[j] = G.split_in[src];
      G.sample[j] = G.sample_in[src];
      G.gid[j] = g;
      if ((hm[r] >> lane) & 1ull) {
        const uint64_t k = key[r] >> G.shift;
        G.gstart[g] = j; G.gkeys[g] = k; G.gplaced[g] = (k >> 15) != 0;
      }
      if (j == n - 1) { G.gstart[g + 1] = n; G.cnt[CC_NGROUPS] = g + 1; }
    }
    run += (uint32_t)__popcll(hm[r]);
    // First appearance in input order (=> Nim Table insertion order) = min of the first-appearance key per group.  Group ids
    // are non-decreasing along the wave, so a segmented suffix-min leaves each run's minimum in its first lane and only
    // that lane touches memory (one atomic per (wave-round, group) instead of one per tread on a handful of hot addresses).
    uint64_t v = fk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t ov = ((uint64_t)(uint32_t)__shfl_down((int)(v >> 32), d) << 32) | (uint32_t)__shfl_down((int)(v & 0xffffffffu), d);
      const uint32_t og = __shfl_down(g, d);
      if (lane + d < 64 && og == g) v = v < ov ? v : ov;
    }
    const uint32_t pg = __shfl_up(g, 1);
    if (in && (lane == 0 || pg != g)) atomicMin(reinterpret_cast<unsigned long long *>(&G.gfirst[g]), (unsigned long long)v);
  }
}

// bounds() + the callclusters gate on ONE bare cluster (Cluster.left_most = right_most = 0), reads [0, n) sorted by
// position: what the reference's tests/test_cluster.nim bounds() vectors exercise
__global__ void bounds_bare_kernel(ClusterParams P, uint32_t n, uint32_t *scratch, uint32_t cap) {
  if (threadIdx.x == 0 && blockIdx.x == 0) emit_bounds(P, GlobalView{P.pos, P.split, P.sample}, 0u, n, 0u, 0u, scratch, cap, P.out[0]);
}

// ---- multi-GPU: the share of an all-gathered tread set this rank clusters ------------------------------------------
// owner of a (tid, unit) group: any function of the key works, groups never interact (strling_amd/dist.py group_owner)
__device__ __forceinline__ uint32_t group_owner(const strl_tread &t, uint32_t world) {
  uint64_t h = (uint64_t)(int64_t)t.tid * 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int j = 0; j < 6; ++j) h = (h ^ (uint64_t)(uint8_t)t.repeat[j]) * 0x100000001B3ull;
  return (uint32_t)((h >> 17) % (uint64_t)world);
}
// gathered = world x pad treads, rank-major, rank r's treads in [r * pad, r * pad + counts[r]).  key 0 = mine, 1 = not
// mine / padding; a stable one-bit sort then brings my treads to the front in global (rank, .bin) order.
__global__ __launch_bounds__(1024) void owned_keys_kernel(const strl_tread *gathered, const uint32_t *counts, uint32_t world, uint32_t pad, uint32_t rank,
                                                          uint64_t *key, uint32_t *val, uint32_t *n_total, uint32_t *n_owned, uint32_t *err) {
  __shared__ uint32_t wcnt[16];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t tot = world * pad;
  bool mine = false;
  if (i < tot) {
    const uint32_t r = i / pad, j = i - r * pad;
    const uint32_t c = counts[r];
    if (j == 0 && c > pad) atomicOr(err, CERR_CAND);
    if (j < (c < pad ? c : pad)) mine = group_owner(gathered[i], world) == rank;
    key[i] = mine ? 0ull : 1ull;
    val[i] = i;
  }
  if (i == 0) *n_total = tot;
  const unsigned long long m = __ballot(mine);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < 16; ++w) t += wcnt[w];
    if (t) atomicAdd(n_owned, t);
  }
}
__global__ void owned_gather_kernel(const strl_tread *gathered, const uint32_t *perm, const uint32_t *n_owned, uint32_t n_max, strl_tread *out) {
  uint32_t n = *n_owned;
  if (n > n_max) n = n_max;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 *src = reinterpret_cast<const uint4 *>(gathered + perm[i]);
  uint4 *dst = reinterpret_cast<uint4 *>(out + i);
  dst[0] = src[0];
  dst[1] = src[1];
}

static inline uint32_t base_code(char b, bool &ok) {
  switch (b) { case 'C': return 0; case 'A': return 1; case 'T': return 2; case 'G': return 3; default: ok = false; return 0; }
}

}  // namespace strl

using namespace strl;

enum { B_TREADS, B_KEY0, B_KEY1, B_VAL0, B_VAL1, B_SORT, B_IN, B_SORTED, B_TILES, B_GROUPS, B_CAND, B_OUT, B_BIG, B_CNT, B_GKEY, B_FIRST };

// The whole device side of a clustering pass over `treads` (device memory; count at d_n, at most n_max): keys, sort,
// group tables, ends/walk sweep, bounds.  Asynchronous: no host synchronisation, every launch is sized by n_max.
static int cluster_device_pass(strl_ctx *c, const strl_tread *treads, const uint32_t *d_n, hipStream_t st = nullptr) {
  ClusterRun &R = c->cl_run;
  const uint32_t n_max = R.n_max;
  strl::DevBuf *B = c->c_buf;
  if (!st) st = c->stream;
  int rc;
  const size_t n1 = std::max<size_t>(n_max, 1);
  const uint32_t cand_cap = (uint32_t)(n1 / (size_t)std::max(1, R.min_support) + 1);
  const uint32_t big_words = (uint32_t)std::min<size_t>(13 * n1 + 1024, 0xfffffff0u);
  const int sort_bits = R.composite ? R.pos_bits + R.kbits : std::max(32, R.kbits);
  const size_t sb = radix_sort_scratch_bytes(n_max, sort_bits);
  const uint32_t ntiles = (uint32_t)((n1 + GT_TILE - 1) / GT_TILE);
  auto need = [&](int i, size_t bytes) { return B[i].reserve(std::max<size_t>(bytes, 256)); };
  if ((rc = need(B_KEY0, n1 * 8)) || (rc = need(B_KEY1, n1 * 8)) || (rc = need(B_VAL0, n1 * 4)) || (rc = need(B_VAL1, n1 * 4)) ||
      (rc = need(B_SORT, sb)) || (rc = need(B_IN, n1 * 9 + 64)) || (rc = need(B_SORTED, n1 * 21 + 64)) || (rc = need(B_TILES, (size_t)ntiles * 4)) ||
      (rc = need(B_GROUPS, n1 * 21 + 64)) || (rc = need(B_FIRST, n1 * 8)) || (rc = need(B_CAND, (size_t)cand_cap * 16)) || (rc = need(B_OUT, (size_t)cand_cap * 2 * sizeof(RawBounds))) ||
      (rc = need(B_BIG, (size_t)big_words * 4)) || (rc = need(B_CNT, CC_WORDS * 4)) || (!R.composite && (rc = need(B_GKEY, n1 * 8))))
    return rc;
  uint32_t *cnt = B[B_CNT].as<uint32_t>();
  const int sort_bits0 = R.composite ? R.pos_bits + R.kbits : 32;
  void *st_tab = nullptr;
  size_t st_bytes = 0;
  radix_sort_tables(B[B_SORT].p, n_max, sort_bits0, &st_tab, &st_bytes);   // the (first) sort's chunk tables: zeroed with the counters
  STRL_HIP(zero_words2(cnt + 1, (CC_WORDS - 1) * 4, st_tab, st_bytes, st));   // CC_N stays (host path stores n there)
  const int TB = 256;
  const uint32_t nb = (uint32_t)((n1 + TB - 1) / TB);
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[4], st));
  // ---- keys + sort ------------------------------------------------------------------------------------
  uint32_t *d_posin = B[B_IN].as<uint32_t>(), *d_samplein = d_posin + n1;
  uint8_t *d_splitin = reinterpret_cast<uint8_t *>(d_samplein + n1);
  KeyParams K{};
  K.d_n = d_n; K.n_max = n_max; K.treads = treads; K.key = B[B_KEY0].as<uint64_t>(); K.val = B[B_VAL0].as<uint32_t>();
  K.pos_in = d_posin; K.sample_in = d_samplein; K.split_in = d_splitin; K.gkey_in = R.composite ? nullptr : B[B_GKEY].as<uint64_t>();
  K.first_in = B[B_FIRST].as<uint64_t>(); K.first_key = R.first_key;
  K.cnt = cnt; K.composite = R.composite ? 1 : 0; K.pos_bits = R.pos_bits; K.fold = R.fold ? 1 : 0; K.n_tid = R.n_tid;
  hipLaunchKernelGGL(tread_keys_kernel, dim3(nb), dim3(TB), 0, st, K);
  uint64_t *sk = nullptr;
  uint32_t *sv = nullptr;
  int e;
  if (R.composite) {
    e = radix_sort_pairs(st, d_n, n_max, B[B_KEY0].as<uint64_t>(), B[B_VAL0].as<uint32_t>(), B[B_KEY1].as<uint64_t>(), B[B_VAL1].as<uint32_t>(),
                         B[B_SORT].p, B[B_SORT].cap, 0, R.pos_bits + R.kbits, &sk, &sv, true);
  } else {
    // stable sort by position, then stable sort by (tid, unit)  ==  group + algorithm.sort by position
    e = radix_sort_pairs(st, d_n, n_max, B[B_KEY0].as<uint64_t>(), B[B_VAL0].as<uint32_t>(), B[B_KEY1].as<uint64_t>(), B[B_VAL1].as<uint32_t>(),
                         B[B_SORT].p, B[B_SORT].cap, 0, 32, &sk, &sv, true);
    if (!e) {
      uint64_t *ok = sk == B[B_KEY0].as<uint64_t>() ? B[B_KEY1].as<uint64_t>() : B[B_KEY0].as<uint64_t>();
      uint32_t *ov = sv == B[B_VAL0].as<uint32_t>() ? B[B_VAL1].as<uint32_t>() : B[B_VAL0].as<uint32_t>();
      hipLaunchKernelGGL(regather_key_kernel, dim3(nb), dim3(TB), 0, st, d_n, n_max, sv, B[B_GKEY].as<uint64_t>(), sk);
      e = radix_sort_pairs(st, d_n, n_max, sk, sv, ok, ov, B[B_SORT].p, B[B_SORT].cap, 0, R.kbits, &sk, &sv);
    }
  }
  if (e) { set_error("radix_sort_pairs failed: %s", hipGetErrorString((hipError_t)e)); return STRL_ERR_HIP; }
  R.perm = sv;
  // ---- group heads -> ids, sorted payload, per-group tables ---------------------------------------------------
  uint32_t *d_pos = B[B_SORTED].as<uint32_t>(), *d_sample = d_pos + n1, *d_gid = d_sample + n1, *d_ends = d_gid + n1, *d_isstart = d_ends + n1;
  uint8_t *d_split = reinterpret_cast<uint8_t *>(d_isstart + n1);
  uint64_t *d_gkeys = B[B_GROUPS].as<uint64_t>();
  uint64_t *d_gfirst = d_gkeys + n1;
  uint32_t *d_gstart = reinterpret_cast<uint32_t *>(d_gfirst + n1);
  uint8_t *d_gplaced = reinterpret_cast<uint8_t *>(d_gstart + (n1 + 1));
  GatherParams G{};
  G.d_n = d_n; G.n_max = n_max; G.key = sk; G.perm = sv; G.shift = R.composite ? R.pos_bits : 0;
  G.pos_in = d_posin; G.sample_in = d_samplein; G.split_in = d_splitin; G.first_in = B[B_FIRST].as<uint64_t>(); G.tile_heads = B[B_TILES].as<uint32_t>();
  G.scanned = ntiles > 4096 ? 1 : 0;   // below that a tile sums the head counts before it itself (<= 16 KB, one launch less)
  G.pos = d_pos; G.sample = d_sample; G.gid = d_gid; G.split = d_split; G.gstart = d_gstart; G.gfirst = d_gfirst; G.gkeys = d_gkeys;
  G.gplaced = d_gplaced; G.cnt = cnt;
  hipLaunchKernelGGL(heads_kernel, dim3(ntiles), dim3(256), 0, st, G);
  if (G.scanned) hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, G);
  hipLaunchKernelGGL(gather_kernel, dim3(ntiles), dim3(256), 0, st, G);
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[5], st));
  // ---- sweep ---------------------------------------------------------------------------------------
  ClusterParams P{};
  P.d_n = d_n; P.n_max = n_max; P.pos = d_pos; P.split = d_split; P.sample = d_sample; P.gid = d_gid; P.gstart = d_gstart; P.gplaced = d_gplaced;
  P.ends = d_ends; P.is_start = d_isstart; P.cnt = cnt; P.big = B[B_BIG].as<uint32_t>(); P.big_words = big_words;
  P.cand = B[B_CAND].as<uint4>(); P.cand_cap = cand_cap; P.out = B[B_OUT].as<RawBounds>();
  P.max_dist = R.window; P.min_support = R.min_support;
  P.min_clip = R.min_clip; P.min_clip_total = R.min_clip_total; P.max_clip_dist = R.max_clip_dist; P.mode = R.mode;
  hipLaunchKernelGGL(ends_kernel, dim3(nb), dim3(TB), 0, st, P);
  hipLaunchKernelGGL(walk_kernel, dim3((uint32_t)((n1 + 63) / 64)), dim3(64), 0, st, P);
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[6], st));
  // ---- bounds ---------------------------------------------------------------------------------------
  hipLaunchKernelGGL(bounds_filter_kernel, dim3((uint32_t)((n1 + 1023) / 1024)), dim3(1024), 0, st, P);
  hipLaunchKernelGGL(bounds_rows_kernel, dim3(std::min<uint32_t>(cand_cap, 32768u)), dim3(64), 0, st, P);
  STRL_HIP(hipGetLastError());
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[7], st));
  return STRL_OK;
}

// c.reads of every returned bound: the sorted permutation is still resident
// The last clustering pass may live in one of the context's other tail sets (overlapped strl_extract_device calls have rotated the
// sets since): bring it in for the duration of a call that reads it.
namespace {
struct LastClusterScope {
  strl_ctx *c;
  int k;       // position of the last clustering pass' set; N_SETS - k rotations bring it to the front, k more restore the order
  explicit LastClusterScope(strl_ctx *ctx) : c(ctx), k(ctx ? ctx->cl_where : 0) { if (k) for (int i = 0; i < N_SETS - k; ++i) rotate_tail(c); }
  ~LastClusterScope() { for (int i = 0; i < k; ++i) rotate_tail(c); }
};
}  // namespace

extern "C" int strl_cluster_members(strl_ctx *c, uint64_t *member_off, uint32_t *members, uint64_t cap, uint64_t *n_members) {
  if (c) { const int rcj = side_join(c); if (rcj) return rcj; }
  LastClusterScope last_cluster(c);
  if (!c || !member_off || !n_members) { set_error("null argument"); return STRL_ERR_ARG; }
  const ClusterRun &R = c->cl_run;
  uint64_t tot = 0;
  for (size_t j = 0; j < R.b_first.size(); ++j) { member_off[j] = tot; tot += R.b_count[j]; }
  member_off[R.b_first.size()] = tot;
  *n_members = tot;
  if (!members) return STRL_OK;
  if (tot > cap) { set_error("member capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)tot); return STRL_ERR_CAPACITY; }
  if (!tot) return STRL_OK;
  STRL_HIP(hipSetDevice(c->device));
  std::vector<uint32_t> perm(R.n);
  if (!R.perm) { set_error("strl_cluster_members: no clustering pass on this context"); return STRL_ERR_ARG; }
  STRL_HIP(hipMemcpyAsync(perm.data(), R.perm, (size_t)R.n * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  uint64_t k = 0;
  for (size_t j = 0; j < R.b_first.size(); ++j)
    for (uint32_t q = 0; q < R.b_count[j]; ++q) {
      const uint32_t src = perm[R.b_first[j] + q];
      members[k++] = R.kept.empty() ? src : R.kept[src];
    }
  return STRL_OK;
}


// ---- results of the last device pass -> the reference's row order (host) ------------------------------------------------
// Row order of -bounds.txt = Nim Table slot order of the (tid, repeat) groups inserted in first-appearance order
// (call.nim:223, merge.nim:172), clusters in position order within a group.
static int cluster_collect(strl_ctx *c, const std::vector<std::pair<uint64_t, uint32_t>> &ghosts_in, strl_bounds *out, uint64_t cap,
                           uint64_t *n_out, strl_unplaced *unplaced, uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats) {
  ClusterRun &R = c->cl_run;
  strl::DevBuf *B = c->c_buf;
  hipStream_t st = c->stream;
  const bool tm = getenv("STRL_CLUSTER_TIMING") != nullptr;
  const auto tm0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (tm) fprintf(stderr, "[cluster_collect] %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count()); };
  uint32_t cnt[CC_WORDS];
  uint32_t n_dev = 0;
  uint32_t part_err = 0;
  STRL_HIP(hipMemcpyAsync(cnt, B[B_CNT].p, CC_WORDS * 4, hipMemcpyDeviceToHost, st));
  if (R.part_err) STRL_HIP(hipMemcpyAsync(&part_err, R.part_err, 4, hipMemcpyDeviceToHost, st));
  STRL_HIP(hipStreamSynchronize(st));
  // (also the asynchronous exchange paths, which never look at the flag themselves: round-3 advisor finding)
  if (part_err) { set_error("strl_cluster_gathered: a rank sent more treads than `pad`"); return STRL_ERR_CAPACITY; }
  n_dev = cnt[CC_NSEEN];
  if (n_dev > R.n_max) { set_error("clustering: %u treads, capacity %u", n_dev, R.n_max); return STRL_ERR_CAPACITY; }
  const uint32_t err = cnt[CC_ERR];
  if (err & CERR_UNIT) { set_error("a tread's repeat unit is not a NUL-padded ACGT string"); return STRL_ERR_ARG; }
  if (err & CERR_TID) { set_error("a tread's tid is outside [-1, %d)", R.n_tid); return STRL_ERR_ARG; }
  if (err & CERR_POS) { set_error("a tread's position needs more than %d bits", R.pos_bits); return STRL_ERR_ARG; }
  if (err & (CERR_CAND | CERR_BIG)) { set_error("clustering scratch exhausted (internal capacity)"); return STRL_ERR_CAPACITY; }
  R.n = n_dev;
  const uint32_t n1 = std::max<uint32_t>(R.n_max, 1);
  const uint32_t n_groups = n_dev ? cnt[CC_NGROUPS] : 0, n_cand = n_dev ? cnt[CC_NCAND] : 0;
  R.n_groups = n_groups; R.n_clusters = cnt[CC_NCLUSTERS];
  R.b_first.clear(); R.b_count.clear();
  if (stats) {
    stats->n_treads = n_dev; stats->n_groups = n_groups; stats->n_clusters = cnt[CC_NCLUSTERS];
    if (c->timing) {
      (void)hipEventElapsedTime(&stats->ms_sort, c->ev[4], c->ev[5]);
      (void)hipEventElapsedTime(&stats->ms_sweep, c->ev[5], c->ev[6]);
      (void)hipEventElapsedTime(&stats->ms_bounds, c->ev[6], c->ev[7]);
    }
  }
  if (n_dev == 0 && ghosts_in.empty()) return STRL_OK;
  std::vector<RawBounds> raw((size_t)2 * n_cand);
  std::vector<uint64_t> g_keys(n_groups);
  std::vector<uint32_t> g_start(n_groups + 1);
  std::vector<uint64_t> g_first(n_groups);
  {
    uint64_t *d_gkeys = B[B_GROUPS].as<uint64_t>();
    uint64_t *d_gfirst = d_gkeys + n1;
    uint32_t *d_gstart = reinterpret_cast<uint32_t *>(d_gfirst + n1);
    if (n_cand) STRL_HIP(hipMemcpyAsync(raw.data(), B[B_OUT].p, raw.size() * sizeof(RawBounds), hipMemcpyDeviceToHost, st));
    if (n_groups) {
      STRL_HIP(hipMemcpyAsync(g_keys.data(), d_gkeys, (size_t)n_groups * 8, hipMemcpyDeviceToHost, st));
      STRL_HIP(hipMemcpyAsync(g_start.data(), d_gstart, (size_t)(n_groups + 1) * 4, hipMemcpyDeviceToHost, st));
      STRL_HIP(hipMemcpyAsync(g_first.data(), d_gfirst, (size_t)n_groups * 8, hipMemcpyDeviceToHost, st));
    }
    STRL_HIP(hipStreamSynchronize(st));
  }
  lap("counters + candidates + group tables on the host");
  if (tm) fprintf(stderr, "[cluster_collect] %u treads, %u groups, %u candidates\n", n_dev, n_groups, n_cand);
  // candidates arrive in arbitrary order: by (group, first read) they are the clusters of a group in position order
  std::vector<uint32_t> cidx(n_cand);
  for (uint32_t k = 0; k < n_cand; ++k) cidx[k] = k;
  std::sort(cidx.begin(), cidx.end(), [&](uint32_t a, uint32_t b) {
    const RawBounds &x = raw[(size_t)2 * a], &y = raw[(size_t)2 * b];
    return x.gid != y.gid ? x.gid < y.gid : x.first < y.first;
  });
  std::vector<uint32_t> cl_lo(n_groups + 1, 0);
  {
    uint32_t ci = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
      cl_lo[g] = ci;
      while (ci < n_cand && raw[(size_t)2 * cidx[ci]].gid == g) ++ci;
    }
    cl_lo[n_groups] = ci;
  }
  lap("candidates sorted");
  // Table keys in insertion order = first appearance in the caller's array, counting the place-holding treads too
  struct KeyEnt { uint64_t key; uint64_t first; int32_t g; };
  std::vector<KeyEnt> ents;
  std::vector<std::pair<uint64_t, uint32_t>> ghosts = ghosts_in;
  ents.reserve(n_groups + ghosts.size());
  for (uint32_t g = 0; g < n_groups; ++g) ents.push_back(KeyEnt{g_keys[g], R.kept.empty() ? g_first[g] : (uint64_t)R.kept[(size_t)g_first[g]], (int32_t)g});
  if (!ghosts.empty()) {
    std::sort(ghosts.begin(), ghosts.end());
    std::vector<KeyEnt> real = ents;
    std::sort(real.begin(), real.end(), [](const KeyEnt &a, const KeyEnt &b) { return a.key < b.key; });
    for (size_t q = 0; q < ghosts.size(); ++q) {
      if (q && ghosts[q].first == ghosts[q - 1].first) continue;       // first (smallest index) ghost of each key
      auto it = std::lower_bound(real.begin(), real.end(), ghosts[q].first, [](const KeyEnt &a, uint64_t k) { return a.key < k; });
      if (it != real.end() && it->key == ghosts[q].first) { KeyEnt &e = ents[(size_t)it->g]; e.first = std::min<uint64_t>(e.first, ghosts[q].second); }
      else ents.push_back(KeyEnt{ghosts[q].first, (uint64_t)ghosts[q].second, -1});
    }
  }
  // by first appearance (distinct indices into the caller's array): an LSD radix sort over the bits they have -- std::sort of a
  // whole genome's 1.1e6 entries was a third of this function's host time
  if (ents.size() < 4096) std::sort(ents.begin(), ents.end(), [](const KeyEnt &a, const KeyEnt &b) { return a.first < b.first; });
  else {
    uint64_t mx = 0;
    for (const KeyEnt &e : ents) mx = std::max(mx, e.first);
    std::vector<KeyEnt> tmp(ents.size());
    for (int shift = 0; shift < 64 && (mx >> shift); shift += 11) {
      uint32_t cnt[2049] = {0};
      for (const KeyEnt &e : ents) ++cnt[((e.first >> shift) & 2047u) + 1];
      for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
      for (const KeyEnt &e : ents) tmp[cnt[(e.first >> shift) & 2047u]++] = e;
      ents.swap(tmp);
    }
  }
  auto key_unit = [](uint64_t key, char rep[7]) {
    const uint32_t len = (uint32_t)(key >> 12) & 7u, code = (uint32_t)key & 0xfffu;
    memset(rep, 0, 7);
    for (uint32_t j = 0; j < len; ++j) rep[j] = "CATG"[(code >> (2 * (len - 1 - j))) & 3u];
  };
  std::vector<uint64_t> hcodes(ents.size());
  {   // (a whole genome has ~1e6 groups: the Nim hash of every key, on a few threads)
    auto part = [&](size_t a, size_t b) {
      for (size_t q = a; q < b; ++q) {
        char rep[7];
        key_unit(ents[q].key, rep);
        hcodes[q] = nim::hash_tid_rep((int32_t)(ents[q].key >> 15) - 1, rep);
      }
    };
    const size_t nt = ents.size() > 200000 ? 8 : 1, per = (ents.size() + nt - 1) / nt;
    std::vector<std::thread> th;
    for (size_t k = 1; k < nt; ++k) th.emplace_back(part, std::min(ents.size(), k * per), std::min(ents.size(), (k + 1) * per));
    part(0, std::min(ents.size(), per));
    for (auto &t : th) t.join();
  }
  lap("group keys hashed");
  const std::vector<int64_t> order = nim::table_slot_order(hcodes, 8192);           // newTable(8192): call.nim:118, merge.nim:92
  lap("table order");
  uint64_t no = 0, nu = 0;
  for (int64_t q : order) {
    if (ents[(size_t)q].g < 0) continue;                                              // every read of the group went to a locus
    const uint32_t g = (uint32_t)ents[(size_t)q].g;
    const uint64_t key = g_keys[g];
    const int32_t tid = (int32_t)(key >> 15) - 1;
    if (tid >= 0 && cl_lo[g] == cl_lo[g + 1]) continue;                               // (a placed group without a cluster writes nothing: 99.8 % of a genome's)
    char rep[7];
    key_unit(key, rep);
    if (tid < 0) {                                                                    // call.nim:226-228
      if (R.mode == STRL_MODE_CALL) {
        if (unplaced && nu < unplaced_cap) { memcpy(unplaced[nu].repeat, rep, 7); unplaced[nu].count = (int64_t)(g_start[g + 1] - g_start[g]); }
        ++nu;
      }
      continue;
    }
    for (uint32_t ci = cl_lo[g]; ci < cl_lo[g + 1]; ++ci)
      for (int half = 0; half < 2; ++half) {
        const RawBounds &r = raw[(size_t)2 * cidx[ci] + half];
        if (!r.valid) continue;
        if (no < cap && out) {
          strl_bounds &b = out[no];
          b.tid = tid; b.left = r.left; b.left_most = r.left_most; b.right = r.right; b.right_most = r.right_most;
          b.center_mass = r.center_mass; b.n_left = r.n_left; b.n_right = r.n_right; b.n_total = r.n_total;
          memcpy(b.repeat, rep, 7);
          R.b_first.push_back(r.first); R.b_count.push_back(r.n_total);
        }
        ++no;
      }
  }
  lap("rows written");
  if (n_out) *n_out = no;
  if (n_unplaced) *n_unplaced = nu;
  if (stats) stats->n_bounds = no;
  if (out && no > cap) { set_error("bounds capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)no); return STRL_ERR_CAPACITY; }
  return STRL_OK;
}

extern "C" int strl_bounds_bare(strl_ctx *c, const uint32_t *positions, const uint8_t *splits, uint32_t n, uint16_t min_clip, uint16_t min_clip_total,
                                uint16_t max_clip_dist, strl_bounds *out, int *good) {
  if (c) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (!c || !positions || !splits || !out || !good || n == 0 || n > (1u << 20)) { set_error("bad argument"); return STRL_ERR_ARG; }
  for (uint32_t i = 1; i < n; ++i) if (positions[i] < positions[i - 1]) { set_error("strl_bounds_bare: reads must be sorted by position"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  DevBuf buf;
  const size_t sw = 4ull * (16ull + 3ull * n);
  int rc;
  if ((rc = buf.reserve((size_t)n * 9 + sw * 4 + sizeof(RawBounds) + 64))) return rc;
  uint32_t *d_pos = buf.as<uint32_t>(), *d_sample = d_pos + n, *d_scratch = d_sample + n;
  RawBounds *d_out = reinterpret_cast<RawBounds *>(d_scratch + sw);
  uint8_t *d_split = reinterpret_cast<uint8_t *>(d_out + 1);
  STRL_HIP(hipMemcpyAsync(d_pos, positions, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipMemsetAsync(d_sample, 0, (size_t)n * 4, c->stream));
  STRL_HIP(hipMemcpyAsync(d_split, splits, n, hipMemcpyHostToDevice, c->stream));
  ClusterParams P{};
  P.pos = d_pos; P.split = d_split; P.sample = d_sample; P.out = d_out; P.mode = STRL_MODE_CALL; P.min_support = 0;
  P.min_clip = min_clip; P.min_clip_total = min_clip_total; P.max_clip_dist = max_clip_dist;
  hipLaunchKernelGGL(bounds_bare_kernel, dim3(1), dim3(64), 0, c->stream, P, n, d_scratch, 16u + 3u * n);
  STRL_HIP(hipGetLastError());
  RawBounds r{};
  STRL_HIP(hipMemcpyAsync(&r, d_out, sizeof r, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  buf.release();
  memset(out, 0, sizeof *out);
  *good = (int)r.valid;
  out->left = r.left; out->left_most = r.left_most; out->right = r.right; out->right_most = r.right_most; out->center_mass = r.center_mass;
  out->n_left = r.n_left; out->n_right = r.n_right; out->n_total = r.n_total;
  return STRL_OK;
}

extern "C" int strl_ctx_cluster_times(strl_ctx *c, double ms[3]) {
  if (c) { const int rcj = side_join(c); if (rcj) return rcj; }
  LastClusterScope last_cluster(c);
  if (!c || !ms) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 3; ++k) {
    float f = 0.f;
    if (c->timing) (void)hipEventElapsedTime(&f, c->ev[4 + k], c->ev[5 + k]);
    ms[k] = f;
  }
  return STRL_OK;
}

static inline int bits_for(uint64_t v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; }

// Re-run the device side of the last clustering pass on the same resident treads, asynchronously.
extern "C" int strl_cluster_replay(strl_ctx *c) {
  if (c) { const int rcj = side_join(c); if (rcj) return rcj; }
  LastClusterScope last_cluster(c);
  if (!c) return STRL_ERR_ARG;
  if (!c->cl_run.treads) { set_error("strl_cluster_replay: no previous clustering pass on this context"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  return cluster_device_pass(c, c->cl_run.treads, c->cl_run.d_n);
}

extern "C" int strl_cluster_resident(strl_ctx *c, int mode, int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support, uint16_t min_clip,
                                     uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap, uint64_t *n_out,
                                     strl_unplaced *unplaced, uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats) {
  static const bool no_overlap = getenv("STRL_NO_OVERLAP") != nullptr;
  const bool async = !out && !n_out && !stats && !n_unplaced;     // results stay on the device
  const bool on_side = c && async && !c->timing && !no_overlap;   // the side stream runs its work in order: no join needed
  if (c && !on_side) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (on_side) { const int rcs = side_streams(c); if (rcs) return rcs; }
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->n_treads_dev) { set_error("strl_cluster_resident: no strl_extract_device call on this context"); return STRL_ERR_ARG; }
  if (mode != STRL_MODE_CALL) { set_error("strl_cluster_resident clusters the treads of one sample (STRL_MODE_CALL)"); return STRL_ERR_ARG; }
  if (n_tid < 0 || pos_bits < 0 || pos_bits > 32 || pos_bits == 1) { set_error("bad argument"); return STRL_ERR_ARG; }
  if (n_out) *n_out = 0;
  if (n_unplaced) *n_unplaced = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  STRL_HIP(hipSetDevice(c->device));
  c->cl_where = 0;
  ClusterRun &R = c->cl_run;
  R = ClusterRun{};
  R.n_max = c->tread_cap; R.n_tid = n_tid; R.mode = mode; R.window = window; R.min_support = min_support; R.min_clip = min_clip;
  R.min_clip_total = min_clip_total; R.max_clip_dist = max_clip_dist;
  R.pos_bits = pos_bits ? pos_bits : 32;
  R.fold = true;
  R.kbits = bits_for((uint64_t)n_tid) + 15;
  R.composite = R.pos_bits + R.kbits <= 64;
  // The treads as the pair logic emitted them (unordered) with their emission keys: clustering does not need the .bin order,
  // only which group appears first in it (ties between equal positions never change a row: bounds() works on counts)
  if (c->pair_ordered) { R.treads = c->treads.as<strl_tread>(); R.d_n = c->n_treads_dev; R.first_key = nullptr; }   // (already ordered for a fetch)
  else { R.treads = c->p_emit.as<strl_tread>(); R.d_n = c->pair_cnt.as<uint32_t>() + PC_EMIT; R.first_key = c->po_key; }
  int rc;
  if (on_side) {
    // overlapped with whatever the main stream is given next (the scorer of the next batch): side stream, fenced by events
    if (!c->pair_on_side) {       // treads made on the main stream: fence; made on the side stream: already in order there
      STRL_HIP(hipEventRecord(c->ev_main_done, c->stream));
      STRL_HIP(hipStreamWaitEvent(c->stream2, c->ev_main_done, 0));
    }
    if ((rc = cluster_device_pass(c, R.treads, R.d_n, c->stream2))) return rc;
    STRL_HIP(hipEventRecord(c->ev_side_done, c->stream2));
    c->side_pending = true;
    return STRL_OK;
  }
  if ((rc = cluster_device_pass(c, R.treads, R.d_n))) return rc;
  if (async) return STRL_OK;
  return cluster_collect(c, {}, out, cap, n_out, unplaced, unplaced_cap, n_unplaced, stats);
}

extern "C" int strl_cluster_collect(strl_ctx *c, strl_bounds *out, uint64_t cap, uint64_t *n_out, strl_unplaced *unplaced, uint64_t unplaced_cap,
                                    uint64_t *n_unplaced, strl_cluster_stats *stats) {
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  { const ClusterRun &lr = c->cl_where ? c->alt[c->cl_where - 1].cl_run : c->cl_run;
    if (!lr.n_max && !lr.d_n) { set_error("strl_cluster_collect: no clustering pass on this context"); return STRL_ERR_ARG; } }
  if (n_out) *n_out = 0;
  if (n_unplaced) *n_unplaced = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  STRL_HIP(hipSetDevice(c->device));
  { const int rcj = side_join(c); if (rcj) return rcj; }
  LastClusterScope last_cluster(c);
  return cluster_collect(c, {}, out, cap, n_out, unplaced, unplaced_cap, n_unplaced, stats);
}

// The stream the tail of the last strl_extract_device call runs on: work enqueued there (a collective over the buffers
// strl_ctx_treads_device returns, strl_cluster_gathered) is ordered behind that batch's pair logic and overlaps the next
// batch's scorer on the main stream.
static bool tail_on_side(strl_ctx *c) {
  static const bool no_overlap = getenv("STRL_NO_OVERLAP") != nullptr;
  return c->pair_on_side && !c->timing && !no_overlap;
}
extern "C" void *strl_ctx_tail_stream(strl_ctx *c) {
  if (!c) return nullptr;
  return (void *)(tail_on_side(c) ? c->stream2 : c->stream);
}

// Make a host array of treads the context's resident treads (what strl_extract_device would have left): the multi-GPU
// `strling merge` shards the treads of its .bin files over the contexts this way before the exchange step.
extern "C" int strl_ctx_set_treads(strl_ctx *c, const strl_tread *treads, uint64_t n) {
  if (!c || (n && !treads) || n > 0x7ffffff0ull) { set_error("strl_ctx_set_treads: bad argument"); return STRL_ERR_ARG; }
  { const int rcj = side_join(c); if (rcj) return rcj; }
  STRL_HIP(hipSetDevice(c->device));
  const uint32_t cap = (uint32_t)std::max<uint64_t>(n, 1);
  int rc;
  if ((rc = c->treads.reserve((size_t)cap * sizeof(strl_tread) + 64))) return rc;
  if (n) STRL_HIP(hipMemcpyAsync(c->treads.p, treads, (size_t)n * sizeof(strl_tread), hipMemcpyHostToDevice, c->stream));
  uint32_t *cnt = reinterpret_cast<uint32_t *>(c->treads.as<uint8_t>() + (size_t)cap * sizeof(strl_tread));
  const uint32_t n32 = (uint32_t)n;
  STRL_HIP(hipMemcpyAsync(cnt, &n32, 4, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  c->n_treads_dev = cnt;
  c->tread_cap = cap;
  c->pair_ordered = true;
  c->pair_on_side = false;
  return STRL_OK;
}

extern "C" int strl_ctx_treads_device(strl_ctx *c, void **treads, uint64_t *cap, void **count) {
  const bool side = c && c->n_treads_dev && tail_on_side(c);
  if (c && !side) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (!c || !c->n_treads_dev) { set_error("strl_ctx_treads_device: no strl_extract_device call on this context"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  { const int rc0 = strl_pair_order(c, side ? c->stream2 : nullptr); if (rc0) return rc0; }           // the gather wants the .bin order
  if (side) { STRL_HIP(hipEventRecord(c->ev_side_done, c->stream2)); c->side_pending = true; }
  if (treads) *treads = c->treads.p;
  if (cap) *cap = c->tread_cap;
  if (count) *count = c->n_treads_dev;
  return STRL_OK;
}

extern "C" int strl_cluster_gathered(strl_ctx *c, const strl_tread *gathered, const uint32_t *counts, int world, uint32_t pad, int rank, int mode,
                                     int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support, uint16_t min_clip, uint16_t min_clip_total,
                                     uint16_t max_clip_dist, strl_bounds *out, uint64_t cap, uint64_t *n_out, strl_unplaced *unplaced,
                                     uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats) {
  // Asynchronous call behind an extract whose pair logic ran on a side stream: the whole exchange step (the caller's
  // collectives on strl_ctx_tail_stream, then this) stays on that stream and overlaps the next batch's scorer.
  static const bool no_overlap = getenv("STRL_NO_OVERLAP") != nullptr;
  const bool on_side = c && !out && !n_out && !stats && !n_unplaced && c->pair_on_side && !c->timing && !no_overlap;
  if (c && !on_side) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (!c || !gathered || !counts || world < 1 || rank < 0 || rank >= world || pad == 0) { set_error("bad argument"); return STRL_ERR_ARG; }
  if ((uint64_t)world * pad > 0x7ffffff0ull || n_tid < 0 || pos_bits < 0 || pos_bits > 32) { set_error("bad argument"); return STRL_ERR_ARG; }
  if (n_out) *n_out = 0;
  if (n_unplaced) *n_unplaced = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  STRL_HIP(hipSetDevice(c->device));
  const uint32_t tot = (uint32_t)world * pad;
  strl::DevBuf *B = c->c_buf;
  hipStream_t st = on_side ? c->stream2 : c->stream;
  int rc;
  if ((rc = B[B_TREADS].reserve((size_t)tot * sizeof(strl_tread))) || (rc = B[B_CNT].reserve(CC_WORDS * 4)) || (rc = B[B_KEY0].reserve((size_t)tot * 8)) ||
      (rc = B[B_KEY1].reserve((size_t)tot * 8)) || (rc = B[B_VAL0].reserve((size_t)tot * 4)) || (rc = B[B_VAL1].reserve((size_t)tot * 4)) ||
      (rc = B[B_SORT].reserve(radix_sort_scratch_bytes(tot, 64))) || (rc = c->g_aux.reserve(256)))
    return rc;
  // counters of the partition live next to the clustering counters: [CC_N] = owned count, scratch words for total / error
  uint32_t *cnt = B[B_CNT].as<uint32_t>();
  uint32_t *aux = c->g_aux.as<uint32_t>();          // [0] total, [1] error flag of the partition
  STRL_HIP(hipMemsetAsync(cnt, 0, 4, st));
  STRL_HIP(hipMemsetAsync(aux, 0, 8, st));
  hipLaunchKernelGGL(owned_keys_kernel, dim3((tot + 1023) / 1024), dim3(1024), 0, st, gathered, counts, (uint32_t)world, pad, (uint32_t)rank,
                     B[B_KEY0].as<uint64_t>(), B[B_VAL0].as<uint32_t>(), aux, cnt + CC_N, aux + 1);
  uint64_t *sk = nullptr;
  uint32_t *sv = nullptr;
  const int e = radix_sort_pairs(st, aux, tot, B[B_KEY0].as<uint64_t>(), B[B_VAL0].as<uint32_t>(), B[B_KEY1].as<uint64_t>(), B[B_VAL1].as<uint32_t>(),
                                 B[B_SORT].p, B[B_SORT].cap, 0, 1, &sk, &sv);
  if (e) { set_error("radix_sort_pairs failed: %s", hipGetErrorString((hipError_t)e)); return STRL_ERR_HIP; }
  hipLaunchKernelGGL(owned_gather_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, gathered, sv, cnt + CC_N, tot, B[B_TREADS].as<strl_tread>());
  STRL_HIP(hipGetLastError());
  c->cl_where = 0;
  ClusterRun &R = c->cl_run;
  R = ClusterRun{};
  R.n_max = tot; R.n_tid = n_tid; R.mode = mode; R.window = window; R.min_support = min_support; R.min_clip = min_clip;
  R.min_clip_total = min_clip_total; R.max_clip_dist = max_clip_dist;
  R.pos_bits = pos_bits ? pos_bits : 32;
  R.fold = true;
  R.kbits = bits_for((uint64_t)n_tid) + 15;
  R.composite = R.pos_bits + R.kbits <= 64;
  R.treads = B[B_TREADS].as<strl_tread>(); R.d_n = cnt + CC_N;
  R.part_err = aux + 1;
  if ((rc = cluster_device_pass(c, R.treads, R.d_n, st))) return rc;
  if (on_side) {
    STRL_HIP(hipEventRecord(c->ev_side_done, c->stream2));
    c->side_pending = true;
    return STRL_OK;
  }
  if (!out && !n_out && !stats && !n_unplaced) return STRL_OK;
  return cluster_collect(c, {}, out, cap, n_out, unplaced, unplaced_cap, n_unplaced, stats);
}

extern "C" int strl_cluster(strl_ctx *c, const strl_tread *treads, uint64_t n_in, int mode, uint32_t window, int32_t min_support,
                            uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap,
                            uint64_t *n_out, strl_unplaced *unplaced, uint64_t unplaced_cap, uint64_t *n_unplaced,
                            strl_cluster_stats *stats) {
  if (c) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (!c || (!treads && n_in) || (!out && cap)) { set_error("null argument"); return STRL_ERR_ARG; }
  const bool tm = getenv("STRL_CLUSTER_TIMING") != nullptr;      // host-side phases of this call on stderr
  const auto tm0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (tm) fprintf(stderr, "[strl_cluster] %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count()); };
  if (n_out) *n_out = 0;
  if (n_unplaced) *n_unplaced = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  STRL_HIP(hipSetDevice(c->device));
  // ---- host: drop what the reference drops on load, note the place holders -------------------------
  std::vector<strl_tread> ft;
  std::vector<uint32_t> kept;
  std::vector<std::pair<uint64_t, uint32_t>> ghosts;   // (group key, index) of treads that only hold their group's place
  bool any_skipped = false;
  int32_t max_tid = -1;
  uint32_t max_pos = 0;
  // (the usual input -- `call` without -l / -b, `merge`'s reads the CLI has filtered already -- drops nothing: one pass to see
  // that, and the caller's array is uploaded as it is instead of a copy of a whole genome's quarter gigabyte of treads)
  bool drops = false;
  for (uint64_t i = 0; i < n_in && !drops; ++i) drops = treads[i].tid < 0 ? (mode == STRL_MODE_MERGE || treads[i].tid < -1 || treads[i].split == STRL_SOFT_TAKEN) : treads[i].split == STRL_SOFT_TAKEN;
  if (!drops) {
    for (uint64_t i = 0; i < n_in; ++i) { max_tid = std::max(max_tid, treads[i].tid); max_pos = std::max(max_pos, treads[i].position); }
  } else ft.reserve(n_in);
  for (uint64_t i = 0; drops && i < n_in; ++i) {
    const strl_tread &t = treads[i];
    if (mode == STRL_MODE_MERGE && t.tid < 0) { any_skipped = true; continue; }   // unpack_file(drop_unplaced=true), merge.nim:101
    if (t.tid < -1) { set_error("tread %llu: tid %d", (unsigned long long)i, t.tid); return STRL_ERR_ARG; }
    if (t.split == STRL_SOFT_TAKEN) {   // given to a -l/-b locus: still a key of the table, no longer a read
      bool ok = true;
      uint32_t len = 0, code = 0;
      while (len < 6 && t.repeat[len]) { code = (code << 2) | base_code(t.repeat[len], ok); ++len; }
      for (uint32_t j = len; j < 6; ++j) if (t.repeat[j]) ok = false;
      if (!ok) { set_error("tread %llu: repeat unit is not a NUL-padded ACGT string", (unsigned long long)i); return STRL_ERR_ARG; }
      ghosts.push_back({((uint64_t)(uint32_t)(t.tid + 1) << 15) | ((uint64_t)len << 12) | code, (uint32_t)i});
      any_skipped = true;
      continue;
    }
    max_tid = std::max(max_tid, t.tid);
    max_pos = std::max(max_pos, t.position);
    ft.push_back(t);
    kept.push_back((uint32_t)i);
  }
  if (!any_skipped) kept.clear();                     // identity
  const strl_tread *up = drops ? ft.data() : treads;
  const uint64_t n64 = drops ? ft.size() : n_in;
  if (n64 > 0x7ffffff0ull) { set_error("too many treads"); return STRL_ERR_ARG; }
  const uint32_t n = (uint32_t)n64;
  if (stats) stats->n_treads = n;
  if (n == 0) return STRL_OK;                        // (only ghosts left: nothing to cluster, nothing to report)
  lap("host scan done");
  // ---- upload, one device pass, results ---------------------------------------------------------------
  strl::DevBuf *B = c->c_buf;
  int rc;
  if ((rc = B[B_TREADS].reserve((size_t)n * sizeof(strl_tread))) || (rc = B[B_CNT].reserve(CC_WORDS * 4))) return rc;
  hipStream_t st = c->stream;
  STRL_HIP(hipMemcpyAsync(B[B_TREADS].p, up, (size_t)n * sizeof(strl_tread), hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(B[B_CNT].p, &n, 4, hipMemcpyHostToDevice, st));
  c->cl_where = 0;
  ClusterRun &R = c->cl_run;
  R = ClusterRun{};
  R.n_max = n; R.n = n; R.n_tid = max_tid + 1; R.mode = mode; R.window = window; R.min_support = min_support; R.min_clip = min_clip;
  R.min_clip_total = min_clip_total; R.max_clip_dist = max_clip_dist;
  R.kbits = bits_for((uint64_t)(uint32_t)(max_tid + 1)) + 15;
  R.pos_bits = bits_for(max_pos);
  R.fold = false;
  if (getenv("STRL_CLUSTER_TWO_SORTS")) R.pos_bits = 64;   // tests: force the two-sort path
  R.composite = R.pos_bits + R.kbits <= 64;
  if (!R.composite) R.pos_bits = 32;
  R.kept.swap(kept);
  R.treads = B[B_TREADS].as<strl_tread>(); R.d_n = B[B_CNT].as<uint32_t>() + CC_N;
  lap("upload enqueued");
  if (tm) { STRL_HIP(hipStreamSynchronize(st)); lap("upload done"); }
  if ((rc = cluster_device_pass(c, R.treads, R.d_n))) return rc;
  lap("device pass enqueued");
  if (tm) { STRL_HIP(hipStreamSynchronize(st)); lap("device pass done"); }
  rc = cluster_collect(c, ghosts, out, cap, n_out, unplaced, unplaced_cap, n_unplaced, stats);
  lap("rows collected");
  return rc;
}
