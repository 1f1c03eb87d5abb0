// cluster.hip -- STR-read clustering on the GPU (call.nim:118-130,223-235; merge.nim:121-187;
// cluster.nim:175-374; callclusters.nim:52-66), gfx950.
//
// The reference groups treads in a hash table keyed by (tid, repeat), merge-sorts every group by
// position and sweeps each group sequentially.  Here the whole tread set is ONE keyed stable
// radix sort (position pass, then (tid, unit) pass), after which
//   ends_kernel   : one tread per lane computes where the cluster that STARTS at it would end --
//                   the reference's growth rule depends only on the start: <= 9 warm-up steps of
//                   the running median-of-first-9, then a binary search for pos > posmed+max_dist+100,
//   walk_kernel   : one lane per group follows start -> end links and flags the cluster heads,
//   bounds_filter_kernel / bounds_rows_kernel : trim + support gate per cluster, then anchor gate, split_cluster, bounds() and
//                   the callclusters gate, with Nim CountTable.largest slot-order tie-breaks
//                   reproduced on the device (nim_tables.h).
// HBM-bound integer work; sizes are ~1e6 treads per 30x sample (5e7 for a 50-sample merge).
#include <hipcub/hipcub.hpp>
#include <string.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "nim_tables.h"

namespace nim { std::vector<int64_t> table_slot_order(const std::vector<uint64_t> &hcodes, uint64_t initial_size); }

namespace strl {

struct RawBounds {  // one candidate row per (cluster, half)
  uint32_t valid;   // 1 = passes every gate
  uint32_t first;   // sorted index of the first read (diagnostics / ordering)
  uint32_t left, left_most, right, right_most, center_mass;
  uint16_t n_left, n_right, n_total, pad;
};

struct ClusterParams {
  uint32_t n;
  const uint32_t *pos;      // sorted
  const uint8_t *split;     // sorted
  const uint32_t *sample;   // sorted (qname_id)
  const uint32_t *gid;      // group index of each sorted tread
  const uint32_t *gstart;   // [n_groups+1]
  const uint8_t *gplaced;   // [n_groups] 1 if tid >= 0
  uint32_t n_groups;
  uint32_t *ends;           // e(s)
  uint32_t *is_start;       // cluster-head flags
  const uint32_t *cl_start; // [n_clusters] compacted heads
  const uint32_t *n_clusters;
  uint32_t *scratch;        // 4 * (16 * n_clusters + 3 * n) dwords
  uint4 *cand;              // [n_clusters] clusters that pass the support gate: {cluster, first read after trim, end}
  uint32_t *n_cand;
  RawBounds *out;           // [2 * n_clusters]
  uint32_t max_dist;
  int32_t min_support;
  uint32_t min_clip, min_clip_total, max_clip_dist;
  int32_t mode;
};

__device__ __forceinline__ uint32_t posmed_at(const uint32_t *pos, uint32_t s, uint32_t n) {
  const uint32_t m = n < 9u ? n : 9u;         // cluster.nim:59-62: reads[int(min(9, n) / 2 - 0.5)]
  return pos[s + ((m - 1u) >> 1)];
}

__global__ __launch_bounds__(256) void ends_kernel(ClusterParams P) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= P.n) return;
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
  if (g >= P.n_groups || !P.gplaced[g]) return;
  uint32_t s = P.gstart[g];
  const uint32_t ge = P.gstart[g + 1];
  while (s < ge) {                            // trcluster: the next cluster starts at the first rejected read
    P.is_start[s] = 1u;
    s = P.ends[s];
  }
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

// bounds in two launches.  bounds_filter_kernel: one lane per cluster for the cheap part (trim, support gate: 98 % of the
// clusters of a WGS sample end there); survivors are appended to a candidate list with one atomic per 1024 clusters.
// bounds_rows_kernel: one WAVE per candidate -- all lanes copy the cluster's reads into LDS in one coalesced round trip,
// then one lane runs the sequential CountTable logic out of LDS (its tables live there too).  With one lane per cluster
// and every read and table access a dependent global load, the single kernel took 80 us for 5x10^5 treads, all latency.
constexpr uint32_t BK_LIM = 256;   // reads of a cluster staged in LDS; larger clusters use the global arrays
__global__ __launch_bounds__(1024) void bounds_filter_kernel(ClusterParams P) {
  __shared__ uint32_t wcnt[16];
  __shared__ uint32_t base_sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nc = *P.n_clusters;
  uint32_t s = 0, e = 0;
  bool cand = false;
  if (c < nc) {
    RawBounds *o = P.out + 2 * (uint64_t)c;
    o[0].valid = 0; o[1].valid = 0;
    s = P.cl_start[c];
    e = P.ends[s];
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
    base_sh = tot ? atomicAdd(P.n_cand, tot) : 0u;
  }
  __syncthreads();
  if (cand) P.cand[base_sh + wcnt[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint4(c, s, e, 0u);
}

__global__ __launch_bounds__(64) void bounds_rows_kernel(ClusterParams P) {
  __shared__ uint32_t l_pos[BK_LIM], l_sample[BK_LIM];
  __shared__ uint8_t l_split[BK_LIM];
  __shared__ uint32_t l_scratch[4 * (16 + 3 * BK_LIM)];
  const int lane = threadIdx.x;
  const uint32_t n_cand = *P.n_cand;
  for (uint32_t k = blockIdx.x; k < n_cand; k += gridDim.x) {
    const uint4 cd = P.cand[k];
    const uint32_t cc = cd.x, cs = cd.y, ce = cd.z, cn = ce - cs;
    RawBounds *o = P.out + 2 * (uint64_t)cc;
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
      const uint32_t n = cn;
      const uint32_t pm = in_lds ? l_pos[((n < 9u ? n : 9u) - 1u) >> 1] : posmed_at(P.pos, cs, n);
      const uint32_t last = in_lds ? l_pos[n - 1] : P.pos[ce - 1], firstp = in_lds ? l_pos[0] : P.pos[cs];
      const uint32_t right_most = last > pm + P.max_dist ? last : pm + P.max_dist;    // :343
      const uint32_t left_most = firstp < pm - P.max_dist ? firstp : pm - P.max_dist; // :344 (uint32 wrap kept)
      if (in_lds) cluster_rows(P, LdsView{l_pos, l_split, l_sample, cs}, cs, ce, left_most, right_most, l_scratch, 16u + 3u * cn, o);
      else {
        // scratch of the global path: the region of the ORIGINAL cluster start is as large as this cluster needs
        const uint32_t s0 = P.cl_start[cc];
        cluster_rows(P, GlobalView{P.pos, P.split, P.sample}, cs, ce, left_most, right_most, P.scratch + 4ull * (16ull * cc + 3ull * s0),
                     16u + 3u * (ce - s0), o);
      }
    }
    __syncthreads();
  }
}

__global__ void heads_kernel(const uint64_t *gkey, uint32_t n, uint32_t *head, uint32_t shift) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || (gkey[i] >> shift) != (gkey[i - 1] >> shift)) ? 1u : 0u;
}
__global__ void gather_kernel(uint32_t n, const uint32_t *perm, const uint32_t *pos_in, const uint8_t *split_in, const uint32_t *sample_in,
                              const uint32_t *head, const uint32_t *gid_incl, const uint64_t *gkey_sorted, uint32_t *pos, uint8_t *split,
                              uint32_t *sample, uint32_t *gid, uint32_t *gstart, uint32_t *gfirst, uint64_t *gkeys, uint8_t *gplaced,
                              uint32_t shift) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = i < n;
  uint32_t src = 0xffffffffu, g = 0xffffffffu;
  if (in) {
    src = perm[i];
    pos[i] = pos_in[src];
    split[i] = split_in[src];
    sample[i] = sample_in[src];
    g = gid_incl[i] - 1u;
    gid[i] = g;
    if (head[i]) { const uint64_t k = gkey_sorted[i] >> shift; gstart[g] = i; gkeys[g] = k; gplaced[g] = (k >> 15) != 0; }
    if (i == n - 1) gstart[g + 1] = n;
  }
  // First appearance in input order (=> Nim Table insertion order) = min of `src` per group.  Group ids are
  // non-decreasing along the wave, so a segmented suffix-min leaves each run's minimum in its first lane and only
  // that lane touches memory (one atomic per (wave, group) instead of one per tread on a handful of hot addresses).
  uint32_t v = src;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t ov = __shfl_down(v, d), og = __shfl_down(g, d);
    if (lane + d < 64 && og == g) v = v < ov ? v : ov;
  }
  const uint32_t pg = __shfl_up(g, 1);
  if (in && (lane == 0 || pg != g)) atomicMin(&gfirst[g], v);
}
__global__ void scatter_starts_kernel(uint32_t n, const uint32_t *is_start, const uint32_t *excl, uint32_t *cl_start, uint32_t *n_clusters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (is_start[i]) cl_start[excl[i]] = i;
  if (i == n - 1) *n_clusters = excl[i] + is_start[i];
}
__global__ void iota_kernel(uint32_t n, uint32_t *v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ void gather_key_kernel(uint32_t n, const uint32_t *perm, const uint64_t *in, uint64_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

static inline uint32_t base_code(char b, bool &ok) {
  switch (b) { case 'C': return 0; case 'A': return 1; case 'T': return 2; case 'G': return 3; default: ok = false; return 0; }
}

}  // namespace strl

using namespace strl;

enum { B_POSIN, B_SPLITIN, B_SAMPLEIN, B_KEYIN, B_PERM0, B_PERM1, B_POSK, B_KEYG, B_KEYS, B_TMP, B_A, B_B, B_C, B_D, B_E, B_F };

// The whole device side of strl_cluster over the tread arrays resident in the context (B_POSIN, B_SPLITIN,
// B_SAMPLEIN, B_KEYIN): two stable radix sorts, group tables, ends/walk sweep, cluster compaction, bounds.
// First pass (replay = false): sizes buffers as it goes and learns n_groups / n_clusters (two host syncs).
// Replay (strl_cluster_replay): same data => same sizes, so it runs without any host synchronisation; this is what
// bench.py times as the clustering part of a step.
static int cluster_device_pass(strl_ctx *c, bool replay) {
  ClusterRun &R = c->cl_run;
  const uint32_t n = R.n;
  strl::DevBuf *B = c->c_buf;
  hipStream_t st = c->stream;
  int rc;
  auto need = [&](int i, size_t bytes) { return replay ? STRL_OK : B[i].reserve(std::max<size_t>(bytes, 256)); };
  if ((rc = need(B_PERM0, (size_t)n * 4)) || (rc = need(B_PERM1, (size_t)n * 4)) || (rc = need(B_POSK, (size_t)n * 4)) ||
      (rc = need(B_KEYG, (size_t)n * 8)) || (rc = need(B_KEYS, (size_t)n * 8)))
    return rc;
  const int TB = 256;
  const uint32_t nb = (n + TB - 1) / TB;
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[4], st));
  // stable sort by position, then stable sort by (tid, unit)  ==  group + algorithm.sort by position.
  // When the group key fits 32 bits both collapse into ONE stable sort of the composite key (group << 32 | position),
  // which the host uploads in B_KEYIN (the sorts are launch-bound at ~10^6 treads: 36 small kernels each).
  hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(TB), 0, st, n, B[B_PERM1].as<uint32_t>());
  const uint32_t shift = R.composite ? 32u : 0u;
  if (!replay) {
    size_t tmp1 = 0, tmp2 = 0, tmp3 = 0;
    STRL_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp1, B[B_POSIN].as<uint32_t>(), B[B_POSK].as<uint32_t>(), B[B_PERM1].as<uint32_t>(),
                                               B[B_PERM0].as<uint32_t>(), (int)n, 0, 32, st));
    STRL_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp2, B[B_KEYG].as<uint64_t>(), B[B_KEYS].as<uint64_t>(), B[B_PERM1].as<uint32_t>(),
                                               B[B_PERM0].as<uint32_t>(), (int)n, 0, 64, st));
    STRL_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tmp3, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, st));
    R.tmpb = std::max(tmp1, std::max(tmp2, tmp3)) + 256;
    if ((rc = need(B_TMP, R.tmpb))) return rc;
  }
  const size_t tmpb = R.tmpb;
  size_t t = tmpb;
  if (R.composite) {
    STRL_HIP(hipcub::DeviceRadixSort::SortPairs(B[B_TMP].p, t, B[B_KEYIN].as<uint64_t>(), B[B_KEYS].as<uint64_t>(), B[B_PERM1].as<uint32_t>(),
                                               B[B_PERM0].as<uint32_t>(), (int)n, 0, 32 + R.kbits, st));
  } else {
    STRL_HIP(hipcub::DeviceRadixSort::SortPairs(B[B_TMP].p, t, B[B_POSIN].as<uint32_t>(), B[B_POSK].as<uint32_t>(), B[B_PERM1].as<uint32_t>(),
                                               B[B_PERM0].as<uint32_t>(), (int)n, 0, 32, st));
    hipLaunchKernelGGL(gather_key_kernel, dim3(nb), dim3(TB), 0, st, n, B[B_PERM0].as<uint32_t>(), B[B_KEYIN].as<uint64_t>(), B[B_KEYG].as<uint64_t>());
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(TB), 0, st, n, B[B_PERM1].as<uint32_t>());   // reused below as scratch values
    t = tmpb;
    // second pass carries the first pass' permutation as its values
    STRL_HIP(hipcub::DeviceRadixSort::SortPairs(B[B_TMP].p, t, B[B_KEYG].as<uint64_t>(), B[B_KEYS].as<uint64_t>(), B[B_PERM0].as<uint32_t>(),
                                               B[B_PERM1].as<uint32_t>(), (int)n, 0, R.kbits, st));
    STRL_HIP(hipMemcpyAsync(B[B_PERM0].p, B[B_PERM1].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  }
  // B_PERM0 = final permutation (sorted index -> input index), B_KEYS = sorted group keys; group heads -> group ids
  if ((rc = need(B_A, (size_t)n * 4)) || (rc = need(B_B, (size_t)n * 4))) return rc;   // A: head flags / is_start, B: scans
  uint32_t *d_head = B[B_A].as<uint32_t>(), *d_scan = B[B_B].as<uint32_t>();
  hipLaunchKernelGGL(heads_kernel, dim3(nb), dim3(TB), 0, st, B[B_KEYS].as<uint64_t>(), n, d_head, shift);
  t = tmpb;
  STRL_HIP(hipcub::DeviceScan::InclusiveSum(B[B_TMP].p, t, d_head, d_scan, (int)n, st));
  if (!replay) {
    STRL_HIP(hipMemcpyAsync(&R.n_groups, d_scan + (n - 1), 4, hipMemcpyDeviceToHost, st));
    STRL_HIP(hipStreamSynchronize(st));
  }
  const uint32_t n_groups = R.n_groups;
  // sorted payload + per-group tables.  Layout of B_C: pos | sample | gid | ends ; B_D: split ; B_E: group tables
  if ((rc = need(B_C, (size_t)n * 16)) || (rc = need(B_D, n))) return rc;
  const size_t gt_bytes = (size_t)(n_groups + 1) * 4 + (size_t)n_groups * 4 + (size_t)n_groups * 8 + (size_t)n_groups + 64;
  if ((rc = need(B_E, gt_bytes))) return rc;
  uint32_t *d_pos = B[B_C].as<uint32_t>(), *d_sample = d_pos + n, *d_gid = d_sample + n, *d_ends = d_gid + n;
  uint8_t *d_split = B[B_D].as<uint8_t>();
  uint64_t *d_gkeys = B[B_E].as<uint64_t>();
  uint32_t *d_gstart = reinterpret_cast<uint32_t *>(d_gkeys + n_groups);
  uint32_t *d_gfirst = d_gstart + (n_groups + 1);
  uint8_t *d_gplaced = reinterpret_cast<uint8_t *>(d_gfirst + n_groups);
  STRL_HIP(hipMemsetAsync(d_gfirst, 0xff, (size_t)n_groups * 4, st));
  hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(TB), 0, st, n, B[B_PERM0].as<uint32_t>(), B[B_POSIN].as<uint32_t>(), B[B_SPLITIN].as<uint8_t>(),
                     B[B_SAMPLEIN].as<uint32_t>(), d_head, d_scan, B[B_KEYS].as<uint64_t>(), d_pos, d_split, d_sample, d_gid, d_gstart, d_gfirst,
                     d_gkeys, d_gplaced, shift);
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[5], st));
  // ---- sweep ---------------------------------------------------------------------------------------
  ClusterParams P{};
  P.n = n; P.pos = d_pos; P.split = d_split; P.sample = d_sample; P.gid = d_gid; P.gstart = d_gstart; P.gplaced = d_gplaced;
  P.n_groups = n_groups; P.ends = d_ends; P.is_start = d_head; P.max_dist = R.window; P.min_support = R.min_support;
  P.min_clip = R.min_clip; P.min_clip_total = R.min_clip_total; P.max_clip_dist = R.max_clip_dist; P.mode = R.mode;
  STRL_HIP(hipMemsetAsync(d_head, 0, (size_t)n * 4, st));
  hipLaunchKernelGGL(ends_kernel, dim3(nb), dim3(TB), 0, st, P);
  hipLaunchKernelGGL(walk_kernel, dim3((n_groups + 63) / 64), dim3(64), 0, st, P);
  t = tmpb;
  STRL_HIP(hipcub::DeviceScan::ExclusiveSum(B[B_TMP].p, t, d_head, d_scan, (int)n, st));
  if ((rc = need(B_F, (size_t)n * 4 + 96 + (size_t)n * 16))) return rc;   // cluster starts, two counters, candidate list
  uint32_t *d_cl_start = B[B_F].as<uint32_t>(), *d_ncl = d_cl_start + n;
  hipLaunchKernelGGL(scatter_starts_kernel, dim3(nb), dim3(TB), 0, st, n, d_head, d_scan, d_cl_start, d_ncl);
  if (!replay) {
    STRL_HIP(hipMemcpyAsync(&R.n_clusters, d_ncl, 4, hipMemcpyDeviceToHost, st));
    STRL_HIP(hipStreamSynchronize(st));
  }
  const uint32_t n_clusters = R.n_clusters;
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[6], st));
  // ---- bounds ---------------------------------------------------------------------------------------
  if (n_clusters) {
    const size_t scratch_dw = 4ull * (16ull * n_clusters + 3ull * n) + 64;
    if (!replay) {
      if ((rc = c->soft_tmp.reserve(scratch_dw * 4))) return rc;
      if ((rc = need(B_KEYG, std::max((size_t)n * 8, (size_t)2 * n_clusters * sizeof(RawBounds))))) return rc;   // reused as output
    }
    P.cl_start = d_cl_start; P.n_clusters = d_ncl; P.scratch = c->soft_tmp.as<uint32_t>(); P.out = B[B_KEYG].as<RawBounds>();
    P.n_cand = d_ncl + 1;
    P.cand = reinterpret_cast<uint4 *>(B[B_F].as<uint8_t>() + (((size_t)n * 4 + 64 + 15) & ~(size_t)15));
    STRL_HIP(hipMemsetAsync(P.n_cand, 0, 4, st));
    hipLaunchKernelGGL(bounds_filter_kernel, dim3((n_clusters + 1023) / 1024), dim3(1024), 0, st, P);
    hipLaunchKernelGGL(bounds_rows_kernel, dim3(std::min<uint32_t>(n_clusters, 32768u)), dim3(64), 0, st, P);
    STRL_HIP(hipGetLastError());
  }
  if (c->timing) STRL_HIP(hipEventRecord(c->ev[7], st));
  return STRL_OK;
}

// c.reads of every returned bound: the sorted permutation is still resident (B_PERM0)
extern "C" int strl_cluster_members(strl_ctx *c, uint64_t *member_off, uint32_t *members, uint64_t cap, uint64_t *n_members) {
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
  STRL_HIP(hipMemcpyAsync(perm.data(), c->c_buf[B_PERM0].p, (size_t)R.n * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  uint64_t k = 0;
  for (size_t j = 0; j < R.b_first.size(); ++j)
    for (uint32_t q = 0; q < R.b_count[j]; ++q) {
      const uint32_t src = perm[R.b_first[j] + q];
      members[k++] = R.kept.empty() ? src : R.kept[src];
    }
  return STRL_OK;
}

// Re-run the device side of the last strl_cluster call on the same resident treads, asynchronously.
extern "C" int strl_cluster_replay(strl_ctx *c) {
  if (!c) return STRL_ERR_ARG;
  if (c->cl_run.n == 0) { set_error("strl_cluster_replay: no previous strl_cluster call on this context"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  return cluster_device_pass(c, true);
}

extern "C" int strl_cluster(strl_ctx *c, const strl_tread *treads, uint64_t n_in, int mode, uint32_t window, int32_t min_support,
                            uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap,
                            uint64_t *n_out, strl_unplaced *unplaced, uint64_t unplaced_cap, uint64_t *n_unplaced,
                            strl_cluster_stats *stats) {
  if (!c || (!treads && n_in) || (!out && cap)) { set_error("null argument"); return STRL_ERR_ARG; }
  if (n_out) *n_out = 0;
  if (n_unplaced) *n_unplaced = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  STRL_HIP(hipSetDevice(c->device));
  // ---- host: keys -------------------------------------------------------------------------------
  std::vector<uint32_t> h_pos, h_sample;
  std::vector<uint8_t> h_split;
  std::vector<uint64_t> h_key;
  std::vector<uint32_t> kept;
  std::vector<std::pair<uint64_t, uint32_t>> ghosts;   // (group key, index) of treads that only hold their group's place
  bool any_skipped = mode == STRL_MODE_MERGE;
  h_pos.reserve(n_in); h_sample.reserve(n_in); h_split.reserve(n_in); h_key.reserve(n_in);
  for (uint64_t i = 0; i < n_in; ++i) {
    const strl_tread &t = treads[i];
    if (mode == STRL_MODE_MERGE && t.tid < 0) continue;           // unpack_file(drop_unplaced=true), merge.nim:101
    if (t.tid < -1) { set_error("tread %llu: tid %d", (unsigned long long)i, t.tid); return STRL_ERR_ARG; }
    const bool ghost = t.split == STRL_SOFT_TAKEN;   // given to a -l/-b locus: still a key of the table, no longer a read
    any_skipped |= ghost;
    bool ok = true;
    uint32_t len = 0, code = 0;
    while (len < 6 && t.repeat[len]) { code = (code << 2) | base_code(t.repeat[len], ok); ++len; }
    for (uint32_t j = len; j < 6; ++j) if (t.repeat[j]) ok = false;
    if (!ok) { set_error("tread %llu: repeat unit is not a NUL-padded ACGT string", (unsigned long long)i); return STRL_ERR_ARG; }
    const uint64_t gkey = ((uint64_t)(uint32_t)(t.tid + 1) << 15) | ((uint64_t)len << 12) | code;
    if (ghost) { ghosts.push_back({gkey, (uint32_t)i}); continue; }
    h_key.push_back(gkey);
    h_pos.push_back(t.position);
    h_split.push_back(t.split);
    h_sample.push_back((uint32_t)t.qname_id);
    kept.push_back((uint32_t)i);
  }
  if (!any_skipped) kept.clear();                     // identity
  const uint64_t n64 = h_pos.size();
  if (n64 > 0x7ffffff0ull) { set_error("too many treads"); return STRL_ERR_ARG; }
  const uint32_t n = (uint32_t)n64;
  if (stats) stats->n_treads = n;
  if (n == 0) return STRL_OK;                        // (only ghosts left: nothing to cluster, nothing to report)

  // ---- upload, then one device pass ------------------------------------------------------------------
  strl::DevBuf *B = c->c_buf;
  int rc;
  auto need = [&](int i, size_t bytes) { return B[i].reserve(std::max<size_t>(bytes, 256)); };
  if ((rc = need(B_POSIN, (size_t)n * 4)) || (rc = need(B_SPLITIN, n)) || (rc = need(B_SAMPLEIN, (size_t)n * 4)) ||
      (rc = need(B_KEYIN, (size_t)n * 8)))
    return rc;
  hipStream_t st = c->stream;
  STRL_HIP(hipMemcpyAsync(B[B_POSIN].p, h_pos.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(B[B_SPLITIN].p, h_split.data(), n, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(B[B_SAMPLEIN].p, h_sample.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  uint64_t maxkey = 0;
  for (uint64_t k : h_key) maxkey = std::max(maxkey, k);
  int kbits = 1;
  while (kbits < 64 && (maxkey >> kbits)) ++kbits;
  const bool composite = kbits <= 32;
  if (composite) for (uint32_t i = 0; i < n; ++i) h_key[i] = (h_key[i] << 32) | h_pos[i];
  STRL_HIP(hipMemcpyAsync(B[B_KEYIN].p, h_key.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
  ClusterRun &R = c->cl_run;
  R = ClusterRun{};
  R.n = n; R.kbits = kbits; R.composite = composite; R.mode = mode; R.window = window; R.min_support = min_support; R.min_clip = min_clip;
  R.min_clip_total = min_clip_total; R.max_clip_dist = max_clip_dist;
  R.kept.swap(kept);
  if ((rc = cluster_device_pass(c, false))) return rc;
  const uint32_t n_groups = R.n_groups, n_clusters = R.n_clusters;
  // ---- results back ------------------------------------------------------------------------------------
  std::vector<RawBounds> raw((size_t)2 * n_clusters);
  std::vector<uint64_t> g_keys(n_groups);
  std::vector<uint32_t> g_start(n_groups + 1), g_first(n_groups), cl_start(n_clusters);
  {
    uint64_t *d_gkeys = B[B_E].as<uint64_t>();
    uint32_t *d_gstart = reinterpret_cast<uint32_t *>(d_gkeys + n_groups);
    uint32_t *d_gfirst = d_gstart + (n_groups + 1);
    if (n_clusters) STRL_HIP(hipMemcpyAsync(raw.data(), B[B_KEYG].p, raw.size() * sizeof(RawBounds), hipMemcpyDeviceToHost, st));
    STRL_HIP(hipMemcpyAsync(g_keys.data(), d_gkeys, (size_t)n_groups * 8, hipMemcpyDeviceToHost, st));
    STRL_HIP(hipMemcpyAsync(g_start.data(), d_gstart, (size_t)(n_groups + 1) * 4, hipMemcpyDeviceToHost, st));
    STRL_HIP(hipMemcpyAsync(g_first.data(), d_gfirst, (size_t)n_groups * 4, hipMemcpyDeviceToHost, st));
    if (n_clusters) STRL_HIP(hipMemcpyAsync(cl_start.data(), B[B_F].p, (size_t)n_clusters * 4, hipMemcpyDeviceToHost, st));
    STRL_HIP(hipStreamSynchronize(st));
  }

  // ---- host: reference row order = Nim Table slot order of the groups (insertion = first appearance) ----
  // Table keys in insertion order = first appearance in the caller's array, counting the place-holding treads too
  struct KeyEnt { uint64_t key; uint32_t first; int32_t g; };
  std::vector<KeyEnt> ents;
  ents.reserve(n_groups + ghosts.size());
  for (uint32_t g = 0; g < n_groups; ++g) ents.push_back(KeyEnt{g_keys[g], R.kept.empty() ? g_first[g] : R.kept[g_first[g]], (int32_t)g});
  if (!ghosts.empty()) {
    std::sort(ghosts.begin(), ghosts.end());
    std::vector<KeyEnt> real = ents;
    std::sort(real.begin(), real.end(), [](const KeyEnt &a, const KeyEnt &b) { return a.key < b.key; });
    for (size_t q = 0; q < ghosts.size(); ++q) {
      if (q && ghosts[q].first == ghosts[q - 1].first) continue;       // first (smallest index) ghost of each key
      auto it = std::lower_bound(real.begin(), real.end(), ghosts[q].first, [](const KeyEnt &a, uint64_t k) { return a.key < k; });
      if (it != real.end() && it->key == ghosts[q].first) { KeyEnt &e = ents[(size_t)it->g]; e.first = std::min(e.first, ghosts[q].second); }
      else ents.push_back(KeyEnt{ghosts[q].first, ghosts[q].second, -1});
    }
  }
  std::sort(ents.begin(), ents.end(), [](const KeyEnt &a, const KeyEnt &b) { return a.first < b.first; });
  auto key_unit = [](uint64_t key, char rep[7]) {
    const uint32_t len = (uint32_t)(key >> 12) & 7u, code = (uint32_t)key & 0xfffu;
    memset(rep, 0, 7);
    for (uint32_t j = 0; j < len; ++j) rep[j] = "CATG"[(code >> (2 * (len - 1 - j))) & 3u];
  };
  std::vector<uint64_t> hcodes(ents.size());
  for (size_t q = 0; q < ents.size(); ++q) {
    char rep[7];
    key_unit(ents[q].key, rep);
    hcodes[q] = nim::hash_tid_rep((int32_t)(ents[q].key >> 15) - 1, rep);
  }
  const std::vector<int64_t> order = nim::table_slot_order(hcodes, 8192);           // newTable(8192): call.nim:118, merge.nim:92
  // clusters are in sorted order => grouped; index them per group
  std::vector<uint32_t> cl_lo(n_groups + 1, 0);
  {
    uint32_t ci = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
      cl_lo[g] = ci;
      while (ci < n_clusters && cl_start[ci] < g_start[g + 1]) ++ci;
    }
    cl_lo[n_groups] = ci;
  }
  uint64_t no = 0, nu = 0;
  for (int64_t q : order) {
    if (ents[(size_t)q].g < 0) continue;                                              // every read of the group went to a locus
    const uint32_t g = (uint32_t)ents[(size_t)q].g;
    const uint64_t key = g_keys[g];
    const int32_t tid = (int32_t)(key >> 15) - 1;
    char rep[7];
    key_unit(key, rep);
    if (tid < 0) {                                                                    // call.nim:226-228
      if (mode == STRL_MODE_CALL) {
        if (unplaced && nu < unplaced_cap) { memcpy(unplaced[nu].repeat, rep, 7); unplaced[nu].count = (int64_t)(g_start[g + 1] - g_start[g]); }
        ++nu;
      }
      continue;
    }
    for (uint32_t ci = cl_lo[g]; ci < cl_lo[g + 1]; ++ci)
      for (int half = 0; half < 2; ++half) {
        const RawBounds &r = raw[(size_t)2 * ci + half];
        if (!r.valid) continue;
        if (no < cap) {
          strl_bounds &b = out[no];
          b.tid = tid; b.left = r.left; b.left_most = r.left_most; b.right = r.right; b.right_most = r.right_most;
          b.center_mass = r.center_mass; b.n_left = r.n_left; b.n_right = r.n_right; b.n_total = r.n_total;
          memcpy(b.repeat, rep, 7);
          R.b_first.push_back(r.first); R.b_count.push_back(r.n_total);
        }
        ++no;
      }
  }
  if (n_out) *n_out = no;
  if (n_unplaced) *n_unplaced = nu;
  if (stats) {
    stats->n_groups = n_groups; stats->n_clusters = n_clusters; stats->n_bounds = no;
    if (c->timing) {
      (void)hipEventElapsedTime(&stats->ms_sort, c->ev[4], c->ev[5]);
      (void)hipEventElapsedTime(&stats->ms_sweep, c->ev[5], c->ev[6]);
      (void)hipEventElapsedTime(&stats->ms_bounds, c->ev[6], c->ev[7]);
    }
  }
  if (no > cap) { set_error("bounds capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)no); return STRL_ERR_CAPACITY; }
  return STRL_OK;
}
