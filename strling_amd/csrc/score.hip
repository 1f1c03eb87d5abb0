// score.hip -- HIP kernels + C-ABI entry points for the extract-side hot path (gfx950).
//
//  classify_kernel : eight reads per lane and iteration, a wave owns a contiguous range of reads; streams the
//                    13 B/read of coordinates + cigar class, evaluates the skip predicate of
//                    extract.nim:30-34 against the genome STR intervals (a wave-level merge join against a
//                    window of {start, running max stop} entries; bin directory + short scan for reads
//                    outside it), writes the "skipped" result word or stages the read for the scoring
//                    queue (one queue atomic per 512 staged reads).  HBM-bound.
//  score_kernel<0> : one queued read per lane -> utils.get_repeat on the whole read
//                    (score_core.h), writes the packed result, queues the soft-clipped ends
//                    add_soft (extract.nim:93-106) would look at.  Bound by integer VALU issue (0.75-0.87 of
//                    all issue slots, profiles/r02).
//  score_kernel<1> : one queued soft-clipped end per lane, scored once, evaluated against both
//                    lowered thresholds (extract.nim:207-211 and :241-244).
#include <stdarg.h>
#include <string.h>
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
#include "common.h"
#include "host_score.h"
#include "front.h"
#include "device_util.h"
#include "score_core.h"
#include "score_tables.h"
#include "sort.h"

namespace strl {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// Device memory: every buffer of the library comes from here.  A failed allocation is STRL_ERR_NOMEM with the sizes in the
// message, not a bare HIP error; STRL_DEVICE_MEM_LIMIT_MB (tests) makes the library refuse to go past that much.
static std::atomic<uint64_t> g_dev_bytes{0};
static int dev_alloc(void **p, size_t want) {
  static const uint64_t cap = getenv("STRL_DEVICE_MEM_LIMIT_MB") ? strtoull(getenv("STRL_DEVICE_MEM_LIMIT_MB"), nullptr, 10) << 20 : 0;
  hipError_t e = hipSuccess;
  if (cap && g_dev_bytes.load() + want > cap) e = hipErrorOutOfMemory;
  else {
    static const bool timing = getenv("STRL_ALLOC_TIMING") != nullptr;      // (diagnosis: where the start of a whole-genome run goes)
    const auto t0 = std::chrono::steady_clock::now();
    e = hipMalloc(p, want);
    if (timing && want >= ((size_t)64 << 20))
      fprintf(stderr, "[strling] hipMalloc %.2f GB: %.3f s\n", (double)want / 1e9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  if (e == hipSuccess) { g_dev_bytes += want; return STRL_OK; }
  (void)hipGetLastError();
  size_t fr = 0, tot = 0;
  (void)hipMemGetInfo(&fr, &tot);
  set_error("out of device memory: %.2f GB more wanted, %.2f GB held by this process, %.2f of %.1f GB free on the device (%s)", (double)want / 1e9, (double)g_dev_bytes.load() / 1e9,
            (double)fr / 1e9, (double)tot / 1e9, e == hipErrorOutOfMemory ? "the input's per-read state does not fit" : hipGetErrorString(e));
  return e == hipErrorOutOfMemory ? STRL_ERR_NOMEM : STRL_ERR_HIP;
}
static void dev_free(void *p, size_t cap) {
  if (!p) return;
  (void)hipFree(p);
  g_dev_bytes -= cap;
}

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap && p) return STRL_OK;
  dev_free(p, cap);
  p = nullptr;
  cap = 0;
  size_t want = bytes + bytes / 8 + 256;
  const int rc = dev_alloc(&p, want);
  if (rc) { p = nullptr; return rc; }
  cap = want;
  return STRL_OK;
}
int DevBuf::grow(size_t bytes, size_t keep_bytes, hipStream_t st) {
  if (bytes <= cap && p) return STRL_OK;
  void *np = nullptr;
  size_t want = std::max(bytes + bytes / 8 + 256, cap * 2);
  int rc = dev_alloc(&np, want);
  if (rc == STRL_ERR_NOMEM && want > bytes + 256) { want = bytes + 256; rc = dev_alloc(&np, want); }     // (no room to double: exactly what is asked for)
  if (rc) return rc;
  if (p && keep_bytes) {
    STRL_HIP(hipMemcpyAsync(np, p, std::min(keep_bytes, cap), hipMemcpyDeviceToDevice, st));
    STRL_HIP(hipStreamSynchronize(st));
  } else if (p) {
    STRL_HIP(hipStreamSynchronize(st));
  }
  dev_free(p, cap);
  p = np;
  cap = want;
  return STRL_OK;
}
void DevBuf::release() {
  dev_free(p, cap);
  p = nullptr;
  cap = 0;
}

// per-tid view of the genome STR table (32 B, one or two cache lines for a whole genome's contigs)
struct TidInfo {
  int64_t iv_off;    // first interval of the tid in g_start / g_pmax
  int64_t bin_off;   // first bin of the tid in g_bins
  int32_t n_iv;      // intervals of the tid
  int32_t n_bins;    // bins of the tid (bin b covers starts in [b << BIN_SHIFT, (b+1) << BIN_SHIFT))
  int32_t has;       // chromosome is a key of the table (extract.nim:30)
  int32_t pad;
};
constexpr int BIN_SHIFT = 12;

// Work items are self-contained 16-byte queue entries, so the scorer never chases metadata pointers:
//   whole read : id = read index            | seq_off | l_seq | clip_l << 16 | clip_r | cig << 16 | mapq << 24
//   segment    : id (read index << 1 | side, or a window index) | seq_off | first base | length
// Stage-B items are 32 bytes: the entry + {slot, best, res0, res1}.
constexpr uint32_t EMPTY = 0xffffffffu;

struct ScoreParams {
  const uint4 *meta;      // optional: strl_read_meta rows (seq_off | l_seq, clip_l | clip_r, cig, mapq | pad): what a queue entry carries of a read
  uint64_t n;
  const int32_t *tid, *pos, *end;
  const uint32_t *seq_off;
  const uint16_t *l_seq, *clip_l, *clip_r;
  const uint8_t *mapq, *cig;
  const uint8_t *seq4;
  const TidInfo *g_tid;
  const int2 *g_iv;         // per tid, sorted by start: {start_i, max stop of the earlier intervals} + sentinel
  const uint2 *g_bins;      // per tid and 4 KiB bin: {#starts below the bin, #starts below the next bin}
  int32_t n_tid;
  const uint16_t *lut;
  const uint32_t *ta;    // stage A's tables (score_core.h TA_*)
  uint32_t *inv_spill;   // [NW][threads of the launch]: Seg::inv of the waves that need it
  const uint64_t *thr;
  uint32_t *whole;
  uint32_t *queue_id;  // [n]      ids of the reads to score (classify -> stage A)
  uint4 *queue;        // [n]      their 16-byte entries (stage A, which gathers them -> compaction kernels)
  uint8_t *soft_flag;  // [n]      per scored read: bit 0 / 1 = its left / right clip has to be scored (add_soft gates)
  uint4 *soft_queue;   // [scap]   compacted soft-clip items
  uint4 *sb_state[2];  // dense hand-over of stage A: {best | EMPTY, res0, res1, -} per item
  uint4 *sb_queue[2];  // compacted stage-B items (2 x uint4 each)
  uint32_t scap;
  uint32_t *counters;  // CNT_* layout
  strl_soft_rec *soft_out;
  uint32_t soft_cap;
  uint32_t min_mapq;
  int32_t seg_row0, seg_row1;   // threshold rows of the segment scorer: (2,3) for soft clips, (1,1) for genome windows
  // pair logic (pair.hip): a whole read with a repeat marks its qname group in the Bloom bitmap; nullptr = no pairing
  const uint64_t *qhash;
  uint32_t *bloom;
  uint32_t bloom_mask;
};

constexpr int LUT_DWORDS = LUT_ENTRIES / 2;
constexpr int LUT_A_DWORDS = TA_WORDS;       // stage A reads its own tables (k <= 4)
constexpr int CL_STAGE = 512;    // queue entries a wave stages in LDS before one bulk append (16 KB per block: 8 blocks per CU)
constexpr int CL_ILP = 8;        // reads per lane per iteration (independent lookup chains in flight)

// extract.nim:30-34: single-M cigar, chromosome in the table, no interval overlapping [start, stop)
// The predicate for N independent reads of one lane, written as straight-line PHASES (all directory loads, then all
// bin loads, then all interval loads) so that the N lookup chains are in flight together: N separately inlined
// while-loops serialise their chains, and the kernel is bound by exactly that latency.
// One read against the table, the general way: bin directory, then a short scan inside the bin.  Out of line on purpose:
// it serves the rare wave iteration the merge join below cannot (a contig boundary, unsorted input, a very dense
// stretch), and inlining N copies of its three dependent loads costs the common path ~90 VGPRs (a wave less per SIMD).
__device__ __noinline__ bool skip_one(const TidInfo *g_tid, const uint2 *g_bins, const int2 *g_iv, int32_t t, int32_t start, int32_t stop) {
  const TidInfo ti = g_tid[t];
  if (!ti.has) return false;
  const int32_t b = stop > 0 ? (stop >> BIN_SHIFT) : 0;
  int32_t idx = ti.n_iv;                 // idx = number of intervals with iv.start < stop
  if (b < ti.n_bins) idx = (int32_t)g_bins[ti.bin_off + b].x;
  int2 c = g_iv[ti.iv_off + idx];
  while (c.x < stop) {                   // a start inside the bin below `stop`: rare, and the sentinel ends it
    ++idx;
    c = g_iv[ti.iv_off + idx];
  }
  return !(c.y > start);                 // c.y = longest reach of the intervals starting before `stop`
}

// The merge join's wave-uniform state, carried ACROSS the iterations of a wave: the contig entry and a window of WIN
// consecutive {start, running max stop} entries.  Consecutive iterations of a coordinate-sorted wave advance a few
// kilobases, a window spans tens of kilobases: most iterations reuse it and issue no table load at all.
constexpr int JOIN_WIN = 8;
struct JoinState {
  int32_t t;              // contig the entry / window belong to (-2: none yet)
  int32_t lo;             // every table entry before the window starts below `lo`
  int32_t valid;
  TidInfo tu;
  int2 e[JOIN_WIN];
};

template <int N>
__device__ __forceinline__ void skip_predicate_n(const ScoreParams &P, JoinState &J, const uint32_t (&cg)[N], const int32_t (&t)[N],
                                                 const int32_t (&start)[N], const int32_t (&stop)[N], const bool (&in)[N], bool (&skip)[N]) {
  bool cand[N];
  bool any = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    cand[j] = in[j] && (cg[j] & STRL_CIG_SINGLE_M) && t[j] >= 0 && t[j] < P.n_tid;
    any |= cand[j];
  }
  // lapper.find(start, stop) <=> any interval with iv.start < stop and iv.stop > start.
  // With idx = number of intervals with iv.start < stop, that is: (max stop of intervals 0 .. idx-1) > start = g_iv[idx].y.
  //
  // Merge join.  The reads of a wave iteration are consecutive records of a coordinate-sorted file: they sit on ONE
  // contig within a few kilobases, and only a handful of intervals START inside that span.  So the wave looks the span
  // up ONCE -- contig entry, bin directory, then a window of JOIN_WIN consecutive {start, running max stop} entries, all
  // at wave-uniform addresses -- and every read finds its idx by comparing its stop with the window's starts in
  // registers: no per-lane dependent global loads at all.
  int32_t t0 = 0;
#pragma unroll
  for (int j = N - 1; j >= 0; --j) if (cand[j]) t0 = t[j];
  t0 = __builtin_amdgcn_readfirstlane(any ? t0 : __builtin_amdgcn_readfirstlane(0));
  bool same = true;
#pragma unroll
  for (int j = 0; j < N; ++j) same = same && (!cand[j] || t[j] == t0);
  if (!__any(any)) {
#pragma unroll
    for (int j = 0; j < N; ++j) skip[j] = false;
    return;
  }
  constexpr int WIN = JOIN_WIN;
  if (__all(same)) {
    if (t0 != J.t) {
      J.tu = P.g_tid[__builtin_amdgcn_readfirstlane(t0 < 0 ? 0 : (t0 < P.n_tid ? t0 : 0))];
      J.t = t0;
      J.valid = 0;
    }
    const TidInfo &tu = J.tu;
    int32_t smin = INT32_MAX, smax = INT32_MIN;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      cand[j] = cand[j] && tu.has;
      if (cand[j]) { smin = stop[j] < smin ? stop[j] : smin; smax = stop[j] > smax ? stop[j] : smax; }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t a = __shfl_xor(smin, d), b = __shfl_xor(smax, d);
      smin = a < smin ? a : smin;
      smax = b > smax ? b : smax;
    }
    smin = __builtin_amdgcn_readfirstlane(smin);
    smax = __builtin_amdgcn_readfirstlane(smax);
    if (smin > smax) {          // no candidate in the wave
#pragma unroll
      for (int j = 0; j < N; ++j) skip[j] = false;
      return;
    }
    if (!(J.valid && smin >= J.lo && J.e[WIN - 1].x >= smax)) {          // (re)load the window at the bin of the smallest stop
      const int32_t b0 = smin > 0 ? (smin >> BIN_SHIFT) : 0;
      int32_t i0 = tu.n_iv;
      if (b0 < tu.n_bins) i0 = (int32_t)P.g_bins[tu.bin_off + b0].x;    // # starts below the bin: <= idx(smin)
      i0 = __builtin_amdgcn_readfirstlane(i0);
      const int2 *w = P.g_iv + tu.iv_off + i0;
      const int32_t rem = tu.n_iv - i0;                                  // entries i0 .. n_iv exist (n_iv = the sentinel)
#pragma unroll
      for (int q = 0; q < WIN; ++q) {
        const int2 v = w[q < rem ? q : rem];
        J.e[q].x = __builtin_amdgcn_readfirstlane(v.x);
        J.e[q].y = __builtin_amdgcn_readfirstlane(v.y);
      }
      J.lo = b0 << BIN_SHIFT;                                            // (past the last bin every start lies below it, too)
      J.valid = 1;
    }
    if (J.e[WIN - 1].x >= smax) {                                        // the window holds every start below the largest stop
#pragma unroll
      for (int j = 0; j < N; ++j) {
        int32_t pm = J.e[0].y;
#pragma unroll
        for (int q = 0; q + 1 < WIN; ++q) pm = J.e[q].x < stop[j] ? J.e[q + 1].y : pm;   // starts ascend: the last true one decides
        skip[j] = cand[j] && !(pm > start[j]);
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) skip[j] = cand[j] && skip_one(P.g_tid, P.g_bins, P.g_iv, t[j], start[j], stop[j]);
}

// VEC: a lane owns 4 consecutive reads per group and fetches their coordinates with 16-byte loads (needs 16-byte aligned
// arrays; the host picks the scalar variant otherwise).
template <bool VEC>
__global__ __launch_bounds__(256, 4) void classify_kernel(ScoreParams P) {
  // Each wave owns one contiguous range of reads.  Kept read indices are staged in LDS; a flush reserves queue
  // space with ONE global atomic (one same-address atomic per wave-iteration ran into the ~88 ops/us limit of
  // the L2 atomic unit: 12 ms per 2^25 reads) and gathers the reads' metadata into self-contained 16-byte items.
  __shared__ uint32_t stage[4][CL_STAGE + 64 * CL_ILP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *buf = stage[wave];
  const uint64_t n_waves = (uint64_t)gridDim.x * 4u;
  const uint64_t gw = (uint64_t)blockIdx.x * 4u + wave;
  const uint64_t per = (((P.n + n_waves - 1) / n_waves) + 255ull) & ~255ull;
  const uint64_t r0 = gw * per;
  const uint64_t r1 = r0 + per < P.n ? r0 + per : P.n;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t cnt = 0, nskip = 0;
  // read handled by this lane as sub-item j of the iteration starting at `base`
  auto ridx = [&](uint64_t base, int j) -> uint64_t {
    return VEC ? base + 256ull * (uint64_t)(j >> 2) + 4ull * (uint64_t)lane + (uint64_t)(j & 3) : base + 64ull * (uint64_t)j + (uint64_t)lane;
  };
  auto flush = [&]() {
    if (cnt) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(&P.counters[CNT_QUEUE], cnt);
      b = __shfl(b, 0);
      __builtin_amdgcn_wave_barrier();
      // Only the ids leave this kernel: it is the one launch of the step that HBM bounds, and gathering the kept reads' rows
      // here (8.5 % of the reads: three quarters of every 64-byte line fetched for nothing) was 160 MB of its 800 MB.
      // Stage A of the scorer, which integer issue bounds, gathers them beside its arithmetic and writes the entries.
      for (uint32_t i = lane; i < cnt; i += 64) P.queue_id[b + i] = buf[i];
      __builtin_amdgcn_wave_barrier();
      cnt = 0;
    }
  };
  // What a load leaves in registers is the RAW vector (for VEC: a group of 4 consecutive reads), unpacked only when the
  // iteration that consumes it starts: unpacking at load time (byte extraction of the cigar classes) put an s_waitcnt
  // right behind the loads and serialised every prefetch with its own latency.
  struct In { int4 t, s, e; uint32_t c; };                    // VEC: 4 reads; scalar variant: .x / low byte only
  constexpr int NIN = VEC ? CL_ILP / 4 : CL_ILP;
  auto load = [&](uint64_t base, In (&x)[NIN]) {
    if (VEC) {
#pragma unroll
      for (int gq = 0; gq < NIN; ++gq) {
        const uint64_t r = base + 256ull * gq + 4ull * lane;
        if (r + 3 < r1) {
          x[gq].t = reinterpret_cast<const int4 *>(P.tid)[r >> 2];
          x[gq].s = reinterpret_cast<const int4 *>(P.pos)[r >> 2];
          x[gq].e = reinterpret_cast<const int4 *>(P.end)[r >> 2];
          x[gq].c = reinterpret_cast<const uint32_t *>(P.cig)[r >> 2];
        } else {                                               // the ragged end of the wave's range
          int32_t t[4], st[4], en[4];
          uint32_t cg = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool in = r + q < r1;
            cg |= (in ? (uint32_t)P.cig[r + q] : 0u) << (8 * q);
            t[q] = in ? P.tid[r + q] : -1; st[q] = in ? P.pos[r + q] : 0; en[q] = in ? P.end[r + q] : 0;
          }
          x[gq].t = make_int4(t[0], t[1], t[2], t[3]); x[gq].s = make_int4(st[0], st[1], st[2], st[3]);
          x[gq].e = make_int4(en[0], en[1], en[2], en[3]); x[gq].c = cg;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NIN; ++j) {
        const uint64_t r = base + 64 * j + lane;
        const bool in = r < r1;
        x[j].c = in ? P.cig[r] : 0u;
        x[j].t.x = in ? P.tid[r] : -1;
        x[j].s.x = in ? P.pos[r] : 0;
        x[j].e.x = in ? P.end[r] : 0;
      }
    }
  };
  auto comp = [](const int4 &v, int q) -> int32_t { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };
  In cur[NIN], nxt[NIN];
  JoinState join;
  join.t = -2; join.valid = 0; join.lo = 0;
  load(r0, cur);
  for (uint64_t base = r0; base < r1; base += 64 * CL_ILP) {
    load(base + 64 * CL_ILP, nxt);   // next iteration's streaming loads fly while this one chases the interval table
    bool need[CL_ILP], skipped[CL_ILP], inr[CL_ILP];
    uint32_t cgs[CL_ILP];
    int32_t ts[CL_ILP], sts[CL_ILP], ens[CL_ILP];
#pragma unroll
    for (int j = 0; j < CL_ILP; ++j) {
      inr[j] = ridx(base, j) < r1;
      if (VEC) {
        const In &g4 = cur[j >> 2];
        cgs[j] = (g4.c >> (8 * (j & 3))) & 0xffu; ts[j] = comp(g4.t, j & 3); sts[j] = comp(g4.s, j & 3); ens[j] = comp(g4.e, j & 3);
      } else {
        cgs[j] = cur[j].c; ts[j] = cur[j].t.x; sts[j] = cur[j].s.x; ens[j] = cur[j].e.x;
      }
    }
    skip_predicate_n<CL_ILP>(P, join, cgs, ts, sts, ens, inr, skipped);
#pragma unroll
    for (int j = 0; j < CL_ILP; ++j) {
      need[j] = inr[j] && !skipped[j];
      if (!VEC && skipped[j]) P.whole[ridx(base, j)] = STRL_RES_SKIPPED;
    }
    if (VEC) {   // one 16-byte store per group; kept reads get 0 here and their real word from the scorer later
#pragma unroll
      for (int gq = 0; gq < CL_ILP / 4; ++gq) {
        const uint64_t r = base + 256ull * gq + 4ull * lane;
        if (r + 3 < r1) {
          reinterpret_cast<uint4 *>(P.whole)[r >> 2] = make_uint4(skipped[4 * gq] ? STRL_RES_SKIPPED : 0u, skipped[4 * gq + 1] ? STRL_RES_SKIPPED : 0u,
                                                                 skipped[4 * gq + 2] ? STRL_RES_SKIPPED : 0u, skipped[4 * gq + 3] ? STRL_RES_SKIPPED : 0u);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (skipped[4 * gq + q]) P.whole[r + q] = STRL_RES_SKIPPED;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < CL_ILP; ++j) {
      const unsigned long long m = __ballot(need[j]);
      nskip += (uint32_t)__popcll(__ballot(skipped[j]));
      if (need[j]) buf[cnt + __popcll(m & below)] = (uint32_t)ridx(base, j);
      cnt += (uint32_t)__popcll(m);
    }
    if (cnt >= CL_STAGE) flush();
#pragma unroll
    for (int j = 0; j < NIN; ++j) cur[j] = nxt[j];
  }
  flush();
  if (lane == 0 && nskip) atomicAdd(&P.counters[CNT_SKIP], nskip);
}

// Order-preserving-within-block compaction of a dense array with EMPTY holes into a queue: one global atomic
// per 8192 slots, issued by a kernel that has nothing else to wait for (the scorers themselves never wait on
// an atomic).  KIND 0: soft-clip slots -> soft queue.  KIND 1: stage-A survivors -> 32-byte stage-B items.
template <int KIND, int MODE>
__global__ __launch_bounds__(1024) void compact_kernel(ScoreParams P) {
  __shared__ uint32_t wcnt[128];
  __shared__ uint32_t base_sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t n_src, cap, cnt_idx;
  const uint4 *src;
  const uint4 *ent = nullptr;
  uint4 *dst;
  static_assert(KIND == 1, "soft items have their own compaction kernel");
  {
    n_src = MODE == 0 ? P.counters[CNT_QUEUE] : min(P.counters[CNT_SOFT], P.scap);
    src = P.sb_state[MODE]; ent = MODE == 0 ? P.queue : P.soft_queue; dst = P.sb_queue[MODE];
    cap = 0xffffffffu; cnt_idx = MODE == 0 ? CNT_SBW : CNT_SBS;
  }
  // U slots per thread and round: the same-address atomic that reserves queue space is what bounds this kernel
  // (~88 per microsecond on the L2 atomic unit), so a round covers 8192 slots, not 1024.
  constexpr int U = 8;
  for (uint32_t b0 = blockIdx.x * (1024u * U); b0 < n_src; b0 += gridDim.x * (1024u * U)) {
    uint4 v[U];
    unsigned long long m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = b0 + 1024u * u + threadIdx.x;
      v[u] = make_uint4(EMPTY, 0, 0, 0);
      if (i < n_src) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = __ballot(v[u].x != EMPTY);
      if (lane == 0) wcnt[u * 16 + wave] = (uint32_t)__popcll(m[u]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (int w = 0; w < 16 * U; ++w) { const uint32_t c = wcnt[w]; wcnt[w] = tot; tot += c; }
      base_sh = tot ? atomicAdd(&P.counters[cnt_idx], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v[u].x != EMPTY) {
        const uint32_t i = b0 + 1024u * u + threadIdx.x;
        const uint32_t d = base_sh + wcnt[u * 16 + wave] + (uint32_t)__popcll(m[u] & below);
        if (KIND == 0) { if (d < cap) dst[d] = v[u]; }
        else {
          dst[2 * (uint64_t)d] = ent[i];
          dst[2 * (uint64_t)d + 1] = make_uint4(i, v[u].x, v[u].y, v[u].z);   // slot, best, res0, res1
        }
      }
    }
    __syncthreads();
  }
}

// Soft-clip items of the scored reads: flag byte + the read's queue entry -> segment items {id << 1 | side, seq_off, first
// base, length} in the soft queue (left clip before right clip, reads in queue order within a block round).  One
// queue-space atomic per 8192 reads.
__global__ __launch_bounds__(1024) void soft_compact_kernel(ScoreParams P) {
  __shared__ uint32_t wcnt[128];
  __shared__ uint32_t base_sh;
  constexpr int U = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t n_src = P.counters[CNT_QUEUE];
  for (uint32_t b0 = blockIdx.x * (1024u * U); b0 < n_src; b0 += gridDim.x * (1024u * U)) {
    uint32_t f[U];
    unsigned long long m0[U], m1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = b0 + 1024u * u + threadIdx.x;
      f[u] = i < n_src ? (uint32_t)P.soft_flag[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m0[u] = __ballot((f[u] & 1u) != 0);
      m1[u] = __ballot((f[u] & 2u) != 0);
      if (lane == 0) wcnt[u * 16 + wave] = (uint32_t)(__popcll(m0[u]) + __popcll(m1[u]));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (int w = 0; w < 16 * U; ++w) { const uint32_t c = wcnt[w]; wcnt[w] = tot; tot += c; }
      base_sh = tot ? atomicAdd(&P.counters[CNT_SOFT], tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (f[u]) {
        const uint32_t i = b0 + 1024u * u + threadIdx.x;
        const uint4 e = P.queue[i];
        const uint32_t L = e.z & 0xffffu, cl = e.z >> 16, cr = e.w & 0xffffu;
        uint32_t d = base_sh + wcnt[u * 16 + wave] + (uint32_t)(__popcll(m0[u] & below) + __popcll(m1[u] & below));
        if (f[u] & 1u) { if (d < P.scap) P.soft_queue[d] = make_uint4(e.x << 1, e.y, 0u, cl < L ? cl : L); ++d; }
        if (f[u] & 2u) { const uint32_t c2 = cr < L ? cr : L; if (d < P.scap) P.soft_queue[d] = make_uint4((e.x << 1) | 1u, e.y, L - c2, c2); }
      }
    }
    __syncthreads();
  }
}

// rows ([row][lane] dwords) of a wave's LDS region: raw SEQ staging, the class bins of k <= 4 (24 for k = 3), long-read hash slots
template <int NW, int SLOTS, int STAGE> constexpr int table_rows() {
  constexpr int raw = 4 * ((16 * NW + 62) / 32);
  constexpr int need = STAGE == 0 ? 24 : (NW <= 10 ? 0 : SLOTS);
  return raw > need ? raw : need;
}

// `strling index`: the scorer over fixed windows of a chromosome (genome_strs.nim:61-92: window 100, step 60)
__global__ void window_items_kernel(ScoreParams P, uint32_t n_win, uint32_t window, uint32_t step, uint64_t n_bases) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) P.counters[CNT_SOFT] = n_win;
  if (i >= n_win) return;
  const uint64_t s0 = (uint64_t)i * step;
  const uint64_t len = s0 + window <= n_bases ? window : n_bases - s0;
  P.soft_queue[i] = make_uint4(i, 0u, (uint32_t)s0, (uint32_t)len);
}

// chromosome text -> BAM nibble packing (what the scorer reads), 8 bases per lane per round.  Letters are folded to
// upper case (genome_strs.nim:71 toUpperAscii); anything that is not an IUPAC letter becomes '=' (code 0): not 'N',
// never a match -- the same thing the kmer table makes of it.
__device__ inline uint32_t nt16_code(uint32_t ch) {
  switch (ch & ~0x20u) {
    case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6;
    case 'V': return 7; case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12;
    case 'D': return 13; case 'B': return 14; case 'N': return 15; default: return 0;
  }
}
__global__ __launch_bounds__(256) void pack_text_kernel(const uint8_t *__restrict__ text, uint32_t *__restrict__ seq4, uint64_t n_bases, uint64_t n_dwords) {
  __shared__ uint8_t tbl[256];
  tbl[threadIdx.x] = (uint8_t)nt16_code(threadIdx.x);
  __syncthreads();
  for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d < n_dwords; d += (uint64_t)gridDim.x * 256) {
    uint32_t out = 0;
    if (8 * d + 8 <= n_bases) {
      const uint2 t = *reinterpret_cast<const uint2 *>(text + 8 * d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w = j < 2 ? t.x >> (16 * j) : t.y >> (16 * (j - 2));
        out |= (((uint32_t)tbl[w & 0xffu] << 4) | tbl[(w >> 8) & 0xffu]) << (8 * j);
      }
    } else {
      for (int b = 0; b < 8; ++b) {
        const uint64_t i = 8 * d + b;
        const uint32_t cde = i < n_bases ? tbl[text[i]] : 0u;
        out |= cde << (8 * (b >> 1) + ((b & 1) ? 0 : 4));
      }
    }
    seq4[d] = out;
  }
}

template <int MODE, int STAGE> struct Item {
  uint32_t id, seq_off, slot;
  int L, len, s0;
  uint32_t cl, cr, cg, mq;
  int best;
  uint32_t res0, res1;
  bool act;
};

// STAGE 2 (segments of the short-read class): both stages in one launch -- 98 % of the clipped ends reach k = 5 anyway, so the
//          hand-over, the compaction and the second fetch + conversion of the segment buy nothing there.
// STAGE 0: k = 2..4 on queued items; a lane whose ladder goes on leaves (best, res0, res1) in the dense
//          hand-over array.  STAGE 1: k = 5, 6 on the compacted survivors.  Whoever finishes an item writes its
//          result and the item's two soft-clip slots.  The loop is software-pipelined: while item i runs
//          the ladder, the SEQ chunks and thresholds of item i+1 and the queue entry of item i+2 are in flight.
#ifndef STRL_SCORE_OCC
#define STRL_SCORE_OCC 5     // resident 256-thread blocks per CU stage A of the short-read class is compiled for (five waves per SIMD: 96 registers;
                             // stage B carries the k = 5, 6 tables in LDS: four blocks fit)
#endif
template <int NW, int SLOTS, int MODE, int STAGE, int BLOCK>
__global__ __launch_bounds__(BLOCK, (BLOCK == 256 && NW <= 10) ? (STAGE == 0 ? STRL_SCORE_OCC : 4) : 1) void score_kernel(ScoreParams P) {
  constexpr int L56 = (LUT_ENTRIES - LUT_OFF5) / 2;                                                // dwords of the k = 5, 6 code tables
  constexpr int LUTK = STAGE == 0 ? LUT_A_DWORDS : STAGE == 1 ? LUT_DWORDS : LUT_A_DWORDS + L56;   // k-mer tables this stage looks up
  constexpr int LUTW = LUTK + 256;                                  // + the byte -> 2-bit conversion table
  constexpr int NSLOT = STAGE == 2 ? INV_SLOTS / 2 : INV_SLOTS;     // (the fused kernel's four blocks must fit a CU's LDS)
  // Statically sized LDS where it fits the 64 KB a static allocation may have: the tables then sit at compile-time
  // addresses that fold into the ds_* offset fields (with a dynamic allocation every table access paid a `v_add 0` for
  // the unknown base: ~100 VALU instructions per read in a kernel that is bound by exactly those).
  constexpr int INV_AT = LUTW + (BLOCK / 64) * table_rows<NW, SLOTS, STAGE>() * 64;   // Seg::inv_lds of the block's waves
  constexpr int LDS_WORDS = INV_AT + (BLOCK / 64) * NSLOT * NW;
  constexpr bool STATIC_LDS = (size_t)LDS_WORDS * 4 <= 65536;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  __shared__ __attribute__((aligned(16))) uint32_t lds_st[STATIC_LDS ? LDS_WORDS : 4];
  uint32_t *const lds = STATIC_LDS ? lds_st : lds_dyn;
  if (STAGE == 2) {
    for (int i = threadIdx.x; i < LUT_A_DWORDS; i += BLOCK) lds[i] = P.ta[i];
    for (int i = threadIdx.x; i < L56; i += BLOCK) lds[LUT_A_DWORDS + i] = reinterpret_cast<const uint32_t *>(P.lut)[LUT_OFF5 / 2 + i];
  } else {
    for (int i = threadIdx.x; i < LUTK; i += BLOCK) lds[i] = STAGE == 0 ? P.ta[i] : reinterpret_cast<const uint32_t *>(P.lut)[i];
  }
  for (int i = threadIdx.x; i < 256; i += BLOCK) lds[LUTK + i] = reinterpret_cast<const uint32_t *>(P.lut)[LUT_DWORDS + i];
  __syncthreads();
  // (stage B indexes `lut + LutOff<5 | 6>`: in the fused kernel only those two tables are resident, behind stage A's)
  const uint16_t *lut = STAGE == 2 ? reinterpret_cast<const uint16_t *>(lds + LUT_A_DWORDS) - LUT_OFF5 : reinterpret_cast<const uint16_t *>(lds);
  const uint32_t *clut = lds + LUTK;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *wave_tab = lds + LUTW + wave * (table_rows<NW, SLOTS, STAGE>() * 64);
  uint32_t *col = wave_tab + lane;
  constexpr int MAXCH = (16 * NW + 62) / 32;
  const int ROW0 = MODE == 0 ? 1 : P.seg_row0, ROW1 = MODE == 0 ? 1 : P.seg_row1;
  uint32_t n_items;
  const uint4 *q;
  if (STAGE == 1) { n_items = P.counters[MODE == 0 ? CNT_SBW : CNT_SBS]; q = P.sb_queue[MODE]; }
  else if (MODE == 0) { n_items = P.counters[CNT_QUEUE]; q = nullptr; }
  else { n_items = min(P.counters[CNT_SOFT], P.scap); q = P.soft_queue; }
  const uint32_t stride = gridDim.x * BLOCK;

  // Whole reads, stage A: the queue holds read ids; the lane gathers its read's row (or the six columns) and leaves the
  // 16-byte entry in P.queue for the two compaction kernels behind this launch.  The id is loaded one item further ahead.
  constexpr bool BY_ID = MODE == 0 && STAGE == 0;
  auto load_id = [&](uint32_t item) -> uint32_t { return (BY_ID && item < n_items) ? P.queue_id[item] : 0u; };
  // What a fetch leaves in registers is the RAW loaded words; they are unpacked when the item's own iteration starts.
  // (Unpacking at load time puts an s_waitcnt right behind the load: a gather's full latency at the top of every iteration.)
  struct Raw { uint4 e, x; };
  auto fetch = [&](uint32_t item, uint32_t rid) -> Raw {
    Raw r;
    r.e = make_uint4(0, 0, 0, 0);
    r.x = make_uint4(0, 0, 0, 0);
    if (BY_ID) {       // unconditional (an idle lane reads row 0): a conditional load's result is copied at the join -- a use, a wait
      const uint4 m = P.meta[rid];
      r.e = make_uint4(rid, m.x, m.y, m.z);
      return r;
    }
    if (item < n_items) {
      if (BY_ID) { }
      else if (STAGE != 1) r.e = q[item];
      else { r.e = q[2 * (uint64_t)item]; r.x = q[2 * (uint64_t)item + 1]; }
    }
    return r;
  };
  auto unpack = [&](const Raw &r, uint32_t item) {
    Item<MODE, STAGE> it;
    const uint4 e = r.e, x = r.x;
    it.act = item < n_items;
    it.id = e.x; it.seq_off = e.y;
    it.L = (int)(e.z & 0xffffu);
    it.slot = STAGE != 1 ? item : x.x;
    it.best = STAGE != 1 ? -1 : (int)x.y;
    it.res0 = x.z; it.res1 = x.w;
    if (MODE == 0) {
      it.cl = e.z >> 16; it.cr = e.w & 0xffffu; it.cg = (e.w >> 16) & 0xffu; it.mq = e.w >> 24;
      // a read of more bases than a lane's byte counters are exact for is the host twin's (long_reads_pass below): here it
      // is an empty read without clipped ends -- no word that counts, no soft-clip items -- whose entry keeps the device busy
      // for one turn of the ladder
      if (it.L > STRL_DEVICE_READ_LEN) { it.L = 0; it.cl = 0; it.cr = 0; }
      it.len = it.L; it.s0 = 0;
    } else {
      it.cl = it.cr = it.cg = it.mq = 0;
      it.s0 = (int)e.z;
      it.len = (int)e.w;
      it.L = it.len;
    }
    if (it.len > 16 * NW) it.len = 16 * NW;  // host picks NW from max_l_seq; never taken
    if (!it.act) { it.len = 0; it.s0 = 0; }
    return it;
  };
  struct Pre { uint4 s[MAXCH]; LaneThr t; };
  auto prefetch = [&](const Item<MODE, STAGE> &it, Pre &p) {
    const int nch = it.act ? ((it.s0 & 31) + it.len + 31) >> 5 : 0;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.seq4 + (uint64_t)it.seq_off * 16u) + (it.s0 >> 5);
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) p.s[c] = (c < nch) ? src[c] : make_uint4(0, 0, 0, 0);
    load_thr(P.thr, ROW0, ROW1, it.len, p.t);
  };

  uint32_t base = blockIdx.x * BLOCK + wave * 64;
  Raw cur_r = fetch(base + lane, load_id(base + lane)), nxt_r = fetch(base + stride + lane, load_id(base + stride + lane));
  uint32_t id2 = load_id(base + 2 * stride + lane);
  Pre pc;
  for (; base < n_items; base += stride) {  // wave-uniform
    const Raw nn_r = fetch(base + 2 * stride + lane, id2);
    id2 = load_id(base + 3 * stride + lane);
    const Item<MODE, STAGE> cur = unpack(cur_r, base + lane);
    // The SEQ chunks and thresholds of the item are loaded here, not one item ahead: the 30 registers a prefetched item
    // occupies through the whole ladder cost a wave per SIMD (96 registers: five waves), and five waves hide this
    // latency better than a prefetch under four did (measured: 0.213 -> 0.191 ms for stage A of the whole reads).
    prefetch(cur, pc);
    // ---- run item `cur` ----
    ScoreState st;
    st.best = cur.best; st.alive = STAGE == 1 && cur.act; st.res0 = cur.res0; st.res1 = cur.res1;
    st.ph_t = __builtin_readcyclecounter();
    Seg<NW> sg;
    sg.inv_lds = lds + INV_AT + wave * (NSLOT * NW);
    sg.inv_nslots = NSLOT;
    sg.inv = P.inv_spill;
    sg.inv_stride = gridDim.x * BLOCK;
    const LenBounds lb = len_bounds(cur.act, cur.len);
    if (MODE == 0 && STAGE == 0) {
      // whole reads start on a 16-byte boundary of the SEQ array: converted straight from the prefetched registers
      uint32_t raw[4 * MAXCH];
#pragma unroll
      for (int c = 0; c < MAXCH; ++c) { raw[4 * c] = pc.s[c].x; raw[4 * c + 1] = pc.s[c].y; raw[4 * c + 2] = pc.s[c].z; raw[4 * c + 3] = pc.s[c].w; }
      STRL_PH(st, 0);
      seg_from_words<NW>(raw, clut, cur.len, lb, sg);
    } else {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < MAXCH; ++c) {   // rows a lane does not need just receive zeros
        col[(4 * c + 0) * 64] = pc.s[c].x;
        col[(4 * c + 1) * 64] = pc.s[c].y;
        col[(4 * c + 2) * 64] = pc.s[c].z;
        col[(4 * c + 3) * 64] = pc.s[c].w;
      }
      __builtin_amdgcn_wave_barrier();
      STRL_PH(st, 0);
      seg_from_raw<NW>(col, clut, cur.s0 & 31, cur.len, sg);
    }
    STRL_PH(st, 1);
    if (STAGE != 1) score_stage_a<NW, SLOTS>(sg, cur.act, wave_tab, lds + LUTW, lane, lds, pc.t, lb, st);
    if (STAGE != 0) score_stage_b<NW, SLOTS>(sg, wave_tab, lane, lut, pc.t, lb, st);

    if (BY_ID && cur.act)    // the entry the compaction kernels read (stored here with the item's other results: a store beside the
                             // gather would wait out its latency, one in front of the SEQ loads would make them wait for it)
      P.queue[base + lane] = make_uint4(cur.id, cur.seq_off, (uint32_t)cur.L | (cur.cl << 16), cur.cr | (cur.cg << 16) | (cur.mq << 24));
    const bool fwd = STAGE == 0 && cur.act && st.alive;
    const bool fin = cur.act && !fwd;
    const uint32_t o0 = reduce_packed(st.res0), o1 = reduce_packed(st.res1);
    if (STAGE == 0 && cur.act)
      P.sb_state[MODE][cur.slot] = fwd ? make_uint4((uint32_t)st.best, st.res0, st.res1, 0u) : make_uint4(EMPTY, 0u, 0u, 0u);
    if (MODE == 0) {
      if (cur.act) {
        uint32_t flag = 0;
        if (fin) {
          P.whole[cur.id] = o0;
          if (P.bloom && STRL_RES_COUNT(o0)) bloom_set(P.bloom, P.bloom_mask, fmix64(P.qhash[cur.id]));
          // add_soft gates, extract.nim:97-106
          if (cur.mq >= P.min_mapq && (cur.cg & (STRL_CIG_FIRST_S | STRL_CIG_LAST_S))) {
            const bool has_unit = STRL_RES_K(o0) != 0;
            if ((cur.cg & STRL_CIG_FIRST_S) && (has_unit || cur.cl > 16)) flag |= 1u;
            // with a single cigar op both loop iterations are cig_index == 0 (the host replays the duplicate)
            if ((cur.cg & STRL_CIG_LAST_S) && !(cur.cg & STRL_CIG_ONE_OP) && (has_unit || cur.cr > 16)) flag |= 2u;
          }
        }
        P.soft_flag[cur.slot] = (uint8_t)flag;           // forwarded items leave 0; stage B overwrites
      }
    } else {
      if (fin && cur.slot < P.soft_cap) {
        strl_soft_rec o;
        o.read_side = cur.id;
        o.res_first = o0;
        o.res_after = o1;
        o.seg_len = (uint32_t)cur.len;
        P.soft_out[cur.slot] = o;
      }
    }
    STRL_PH(st, 12);
    cur_r = nxt_r; nxt_r = nn_r;
  }
}

// ---- host side of this translation unit -------------------------------------------------------
template <int NW, int SLOTS, int MODE, int STAGE, int BLOCK> static int launch_score(strl_ctx *ctx, const ScoreParams &P, int blocks) {
  auto kfn = score_kernel<NW, SLOTS, MODE, STAGE, BLOCK>;
  size_t shmem = (size_t)((STAGE == 0 ? LUT_A_DWORDS : STAGE == 1 ? LUT_DWORDS : LUT_A_DWORDS + (LUT_ENTRIES - LUT_OFF5) / 2) + 256) * 4 +
                 (size_t)(BLOCK / 64) * (table_rows<NW, SLOTS, STAGE>() * 64 + (STAGE == 2 ? INV_SLOTS / 2 : INV_SLOTS) * NW) * 4;
  if (shmem <= 65536) shmem = 0;      // the kernel allocates it statically (see STATIC_LDS there)
  // (only the long-read classes allocate dynamically; set on every such launch: the attribute belongs to the current DEVICE,
  // and contexts of one process may sit on different ones)
  if (shmem) STRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  ScoreParams Q = P;
  // sized for the largest grid this class is ever launched with, so that it is allocated once per context and class (an
  // allocation synchronises the device: in the chunked extract every growing chunk would have paid for it)
  const size_t spill = (size_t)std::max(blocks, 8192) * BLOCK * NW * 4;
  if (ctx->inv_spill.cap < spill) {
    STRL_HIP(hipStreamSynchronize(ctx->stream));
    int rc = ctx->inv_spill.reserve(spill);
    if (rc) return rc;
  }
  Q.inv_spill = ctx->inv_spill.as<uint32_t>();
  hipLaunchKernelGGL(kfn, dim3(blocks), dim3(BLOCK), shmem, ctx->stream, Q);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}

// stage A -> compaction of the survivors -> stage B of one MODE, kernel class picked by the longest read
// ev (optional): two events, recorded behind stage A and behind the compaction
template <int MODE> static int launch_score_class(strl_ctx *ctx, const ScoreParams &P, uint32_t max_l, hipEvent_t *ev = nullptr) {
  int rc;
  // Grid: the kernels stride over the queue, so any grid works; measured on 2^25-read batches the time keeps falling
  // until ~8192 blocks for stage A and ~4096 for stage B (1.51 -> 1.29 ms per step against 512 blocks = two resident
  // blocks per CU): surplus blocks are what lets the hardware even out the very uneven cost of the items.  The queue
  // length lives on the device; the host knows an upper bound (reads of the batch / capacity of the segment queue).
  const uint64_t upper = MODE == 0 ? P.n : (uint64_t)P.scap;
  static const int env_a = getenv("STRL_GRID_A") ? atoi(getenv("STRL_GRID_A")) : 0, env_b = getenv("STRL_GRID_B") ? atoi(getenv("STRL_GRID_B")) : 0;
  const int ga = env_a > 0 ? env_a : (int)std::min<uint64_t>(8192, std::max<uint64_t>(256, (upper + 255) / 256));
  const int gb = env_b > 0 ? env_b : std::max(256, ga / 2);
  static const bool split_segments = getenv("STRL_SPLIT_SEGMENTS") != nullptr;
  if constexpr (MODE == 1) {
    if (max_l <= 160 && !split_segments) {        // segments of the short-read class: one fused launch
      if ((rc = launch_score<10, 64, MODE, 2, 256>(ctx, P, ga))) return rc;
      if (ev) { STRL_HIP(hipEventRecord(ev[0], ctx->stream)); STRL_HIP(hipEventRecord(ev[1], ctx->stream)); }
      return STRL_OK;
    }
  }
  if (max_l <= 160) rc = launch_score<10, 64, MODE, 0, 256>(ctx, P, ga);
  else if (max_l <= 256) rc = launch_score<16, 128, MODE, 0, 256>(ctx, P, std::max(256, ga / 4));
  else rc = launch_score<32, 256, MODE, 0, 64>(ctx, P, std::max(512, ga / 2));
  if (rc) return rc;
  if (ev) STRL_HIP(hipEventRecord(ev[0], ctx->stream));
  hipLaunchKernelGGL((compact_kernel<1, MODE>), dim3(512), dim3(1024), 0, ctx->stream, P);
  STRL_HIP(hipGetLastError());
  if (ev) STRL_HIP(hipEventRecord(ev[1], ctx->stream));
  if (max_l <= 160) return launch_score<10, 64, MODE, 1, 256>(ctx, P, gb);
  if (max_l <= 256) return launch_score<16, 128, MODE, 1, 256>(ctx, P, std::max(256, gb / 4));
  return launch_score<32, 256, MODE, 1, 64>(ctx, P, std::max(512, gb / 2));
}

}  // namespace strl

using namespace strl;

static constexpr uint64_t RING = 256;
// events per recorded strl_score_reads call: start | classify | stage A, compaction, stage B (whole reads) | soft-item
// compaction | stage A, compaction, stage B (segments)
static constexpr int EV_PER = 9;

int side_join(strl_ctx *c) {
  if (c->side_pending) {
    STRL_HIP(hipStreamWaitEvent(c->stream, c->ev_side_done, 0));
    c->side_pending = false;
  }
  for (auto &a : c->alt)
    if (a.side_pending) {
      STRL_HIP(hipStreamWaitEvent(c->stream, a.ev_side_done, 0));
      a.side_pending = false;
    }
  return STRL_OK;
}

int side_streams(strl_ctx *c) {
  if (c->stream2) return STRL_OK;
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  for (auto &a : c->alt) if (!a.stream2) STRL_HIP(hipStreamCreateWithFlags(&a.stream2, hipStreamNonBlocking));
  return STRL_OK;
}

static void swap_tail_with(strl_ctx *c, TailSet &a) {
  for (int i = 0; i < 16; ++i) std::swap(c->c_buf[i], a.c_buf[i]);
  std::swap(c->cl_run, a.cl_run);
  std::swap(c->p_key0, a.p_key0); std::swap(c->p_key1, a.p_key1); std::swap(c->p_val0, a.p_val0); std::swap(c->p_val1, a.p_val1);
  std::swap(c->p_emit, a.p_emit); std::swap(c->sort_scratch, a.sort_scratch); std::swap(c->pair_cnt, a.pair_cnt); std::swap(c->treads, a.treads);
  std::swap(c->n_treads_dev, a.n_treads_dev); std::swap(c->tread_cap, a.tread_cap); std::swap(c->pair_item_cap, a.pair_item_cap);
  std::swap(c->po_key, a.po_key); std::swap(c->po_key_alt, a.po_key_alt); std::swap(c->po_val, a.po_val); std::swap(c->po_val_alt, a.po_val_alt);
  std::swap(c->po_bits, a.po_bits); std::swap(c->pair_ordered, a.pair_ordered);
  std::swap(c->stream2, a.stream2); std::swap(c->ev_side_done, a.ev_side_done);
  std::swap(c->side_pending, a.side_pending); std::swap(c->pair_on_side, a.pair_on_side);
}
// current -> alt[0] -> alt[1] -> ... -> current: the least recently used set (the last alternative) becomes current, the set
// that was current becomes alt[0].  N_SETS rotations restore the arrangement.
void rotate_tail(strl_ctx *c) {
  for (int k = 0; k < N_SETS - 1; ++k) swap_tail_with(c, c->alt[k]);
  c->cl_where = (c->cl_where + 1) % N_SETS;
}
static void rotate_head(strl_ctx *c) {
  for (int k = 0; k < N_SETS - 1; ++k) {
    strl_ctx::HeadSet &h = c->head_alt[k];
    std::swap(c->st_whole, h.st_whole); std::swap(c->st_soft, h.st_soft); std::swap(c->counters, h.counters);
    std::swap(c->bloom, h.bloom); std::swap(c->bloom_mask, h.bloom_mask);
  }
  c->set = (c->set + 1) % N_SETS;
}

extern "C" {

#ifdef STRL_PHASE_TIMING
int strl_debug_phase(unsigned long long *out, int reset) {   // debug builds only (not part of the ABI)
  if (out) { if (hipMemcpyFromSymbol(out, HIP_SYMBOL(strl::g_phase), 32 * 8) != hipSuccess) return -1; }
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(strl::g_phase), z, 32 * 8) != hipSuccess) return -1; }
  return 0;
}
#endif

int strl_version(void) { return 100; }
const char *strl_last_error(void) { return strl::g_err; }

int strl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int strl_ctx_mem_info(strl_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (!c) { set_error("null context"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  size_t f = 0, t = 0;
  STRL_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return STRL_OK;
}

int strl_ctx_create(int device_ordinal, strl_ctx **out) {
  if (!out) { set_error("ctx out pointer is NULL"); return STRL_ERR_ARG; }
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    set_error("no HIP device available: strling_amd has no CPU fallback");
    return STRL_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= n) { set_error("device ordinal %d out of range (%d devices)", device_ordinal, n); return STRL_ERR_ARG; }
  static const bool lap_on = getenv("STRL_CTX_TIMING") != nullptr;
  const auto lap0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (lap_on) fprintf(stderr, "[strl_ctx_create] %s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - lap0).count()); };
  lap("device count known (the runtime is up)");
  STRL_HIP(hipSetDevice(device_ordinal));
  strl_ctx *c = new strl_ctx();
  c->device = device_ordinal;
  lap("hipSetDevice");
  STRL_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  lap("first stream");
  // (the side streams -- one per tail set, for the overlapped batches of strl_extract_device / an asynchronous clustering -- are
  // made when that mode is first asked for, side_streams(): a stream is ~9.5 ms here, and `strling extract / call / merge`,
  // which never overlap batches that way, waited for three of them at every start)
  for (auto &a : c->alt) STRL_HIP(hipEventCreateWithFlags(&a.ev_side_done, hipEventDisableTiming));
  STRL_HIP(hipEventCreateWithFlags(&c->ev_main_done, hipEventDisableTiming));
  STRL_HIP(hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming));
  STRL_HIP(hipEventCreateWithFlags(&c->ev_head_done, hipEventDisableTiming));
  for (auto &e : c->ev_set_free) STRL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : c->ev) STRL_HIP(hipEventCreate(&e));
  for (auto &e : c->pev) STRL_HIP(hipEventCreate(&e));
  lap("streams and events");
  std::vector<uint16_t> lut;
  build_lut(lut);
  std::vector<uint32_t> clut, ta;
  build_conv_lut(clut);
  build_stage_a_tables(lut, ta);
  lap("scorer tables built on the host");
  int rc = c->lut.reserve(lut.size() * 2 + clut.size() * 4 + ta.size() * 4);
  if (rc) return rc;
  lap("first hipMalloc");
  {   // (one copy for the three tables: each synchronous copy out of pageable memory is ~3 ms at a process' start)
    std::vector<uint8_t> all(lut.size() * 2 + clut.size() * 4 + ta.size() * 4);
    memcpy(all.data(), lut.data(), lut.size() * 2);
    memcpy(all.data() + lut.size() * 2, clut.data(), clut.size() * 4);
    memcpy(all.data() + lut.size() * 2 + clut.size() * 4, ta.data(), ta.size() * 4);
    STRL_HIP(hipMemcpy(c->lut.p, all.data(), all.size(), hipMemcpyHostToDevice));
  }
  rc = c->counters.reserve(CNT_WORDS * 4);
  if (rc) return rc;
  lap("tables on the device");
  *out = c;
  return STRL_OK;
}

void strl_ctx_destroy(strl_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  for (auto &a : c->alt) if (a.stream2) (void)hipStreamSynchronize(a.stream2);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) { strl::comm_destroy(c->comm); c->comm = nullptr; }
  if (c->x_soft_seen_ev) (void)hipEventDestroy(c->x_soft_seen_ev);
  if (c->x_soft_seen) (void)hipHostFree(c->x_soft_seen);
  if (c->front) { if (c->front->st_c) (void)hipStreamSynchronize(c->front->st_c); for (hipStream_t q : c->front->st_i) if (q) (void)hipStreamSynchronize(q); if (c->front->st_a) (void)hipStreamSynchronize(c->front->st_a); strl::front_destroy(c->front); c->front = nullptr; }
  strl::DevBuf *bufs[] = {&c->lut, &c->thr, &c->g_tid, &c->g_bins, &c->g_start, &c->g_pmax, &c->queue, &c->soft_queue, &c->counters,
                          &c->soft_tmp, &c->sb_whole, &c->sb_soft, &c->queue_r, &c->soft_dense, &c->sb_state_w, &c->sb_state_s, &c->st_tid, &c->st_pos, &c->st_end, &c->st_seqoff, &c->st_lseq, &c->st_clipl, &c->st_clipr,
                          &c->st_mapq, &c->st_cig, &c->st_seq4, &c->st_whole, &c->st_soft, &c->st_text, &c->st_meta, &c->long_list, &c->long_seq,
                          &c->p_key0, &c->p_key1, &c->p_val0, &c->p_val1, &c->p_emit, &c->sort_scratch, &c->pair_cnt, &c->bloom, &c->treads,
                          &c->st_mtid, &c->st_mpos, &c->st_flag, &c->st_qhash, &c->x_rows, &c->x_qhash, &c->x_whole, &c->x_soft, &c->x_cnt, &c->g_aux, &c->crc_tab, &c->p_spill};
  for (auto *b : bufs) b->release();
  for (auto &r : c->rg) {
    if (r.st) { (void)hipStreamSynchronize(r.st); (void)hipStreamDestroy(r.st); }
    r.comp.release(); r.meta.release(); r.u.release(); r.out.release(); r.rq.release(); r.work.release();
  }
  for (auto &b : c->c_buf) b.release();
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->pev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->ring) if (e) (void)hipEventDestroy(e);
  if (c->ev_main_done) (void)hipEventDestroy(c->ev_main_done);
  if (c->ev_side_done) (void)hipEventDestroy(c->ev_side_done);
  if (c->ev_head_done) (void)hipEventDestroy(c->ev_head_done);
  for (auto &e : c->ev_set_free) if (e) (void)hipEventDestroy(e);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  for (auto &a : c->alt) {
    if (a.stream2) (void)hipStreamDestroy(a.stream2);
    if (a.ev_side_done) (void)hipEventDestroy(a.ev_side_done);
    strl::DevBuf *ab[] = {&a.p_key0, &a.p_key1, &a.p_val0, &a.p_val1, &a.p_emit, &a.sort_scratch, &a.pair_cnt, &a.treads};
    for (auto *b : ab) b->release();
    for (auto &b : a.c_buf) b.release();
  }
  for (auto &h : c->head_alt) { h.st_whole.release(); h.st_soft.release(); h.counters.release(); h.bloom.release(); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void *strl_ctx_stream(strl_ctx *c) { return c ? (void *)c->stream : nullptr; }
int strl_ctx_sync(strl_ctx *c) {
  if (!c) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  { const int rc = side_join(c); if (rc) return rc; }
  STRL_HIP(hipStreamSynchronize(c->stream));
  return STRL_OK;
}
int strl_ctx_enable_timing(strl_ctx *c, int on) {
  if (!c) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  c->timing = on != 0;
  c->ring_pos = 0;
  if (c->timing && c->ring.empty()) {
    c->ring.resize(RING * EV_PER);
    for (auto &e : c->ring) STRL_HIP(hipEventCreate(&e));
  }
  return STRL_OK;
}
int strl_ctx_kernel_times_detail(strl_ctx *c, double ms_sum[8], uint64_t *n_launches) {
  if (!c || !ms_sum) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamSynchronize(c->stream));
  for (int k = 0; k < EV_PER - 1; ++k) ms_sum[k] = 0.0;
  const uint64_t n = std::min<uint64_t>(c->ring_pos, RING);
  for (uint64_t q = 0; q < n; ++q) {
    hipEvent_t *e = &c->ring[q * EV_PER];
    for (int k = 0; k < EV_PER - 1; ++k) {
      float ms = 0.f;
      STRL_HIP(hipEventElapsedTime(&ms, e[k], e[k + 1]));
      ms_sum[k] += ms;
    }
  }
  if (n_launches) *n_launches = n;
  return STRL_OK;
}

int strl_ctx_kernel_times(strl_ctx *c, double ms_sum[3], uint64_t *n_launches) {
  if (!c || !ms_sum) return STRL_ERR_ARG;
  double d[EV_PER - 1];
  const int rc = strl_ctx_kernel_times_detail(c, d, n_launches);
  if (rc) return rc;
  ms_sum[0] = d[0];
  ms_sum[1] = d[1] + d[2] + d[3];
  ms_sum[2] = d[4] + d[5] + d[6] + d[7];
  return STRL_OK;
}

int strl_ctx_set_opts(strl_ctx *c, const strl_opts *o) {
  if (!c || !o) { set_error("null argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  c->opts = *o;
  std::vector<uint64_t> thr;
  build_thr(*o, thr);
  int rc = c->thr.reserve(thr.size() * 8);
  if (rc) return rc;
  STRL_HIP(hipMemcpyAsync(c->thr.p, thr.data(), thr.size() * 8, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  c->have_opts = true;
  return STRL_OK;
}

int strl_ctx_set_genome(strl_ctx *c, const strl_genome_str *g) {
  if (!c) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamSynchronize(c->stream));     // a skip-predicate pass still in flight reads the tables replaced below
  if (!g || g->n_tid <= 0) {
    // empty table: no chromosome is a key, nothing is skipped.  The kernels still dereference entry 0 of each array for
    // lanes without a candidate read, so the arrays must exist.
    const TidInfo t0{};
    const int2 iv0 = make_int2(INT32_MAX, INT32_MIN);
    const uint2 b0 = make_uint2(0, 0);
    int rc0;
    if ((rc0 = c->g_tid.reserve(sizeof t0)) || (rc0 = c->g_bins.reserve(sizeof b0)) || (rc0 = c->g_start.reserve(sizeof iv0))) return rc0;
    STRL_HIP(hipMemcpy(c->g_tid.p, &t0, sizeof t0, hipMemcpyHostToDevice));
    STRL_HIP(hipMemcpy(c->g_bins.p, &b0, sizeof b0, hipMemcpyHostToDevice));
    STRL_HIP(hipMemcpy(c->g_start.p, &iv0, sizeof iv0, hipMemcpyHostToDevice));
    c->n_tid = 0; c->n_iv = 0;
    return STRL_OK;
  }
  const int32_t nt = g->n_tid;
  const int64_t niv = g->iv_off[nt];
  std::vector<int32_t> st((size_t)std::max<int64_t>(niv, 1));
  // per tid, n_iv + 1 elements sorted by start: element i = {start_i, max(stop_0..stop_{i-1})}; the last one is the
  // sentinel {INT32_MAX, max of all stops}.  One 8-byte load answers "does a start lie here" AND "does an earlier
  // interval reach past my start".
  std::vector<int2> ivs;
  ivs.reserve((size_t)niv + (size_t)nt);
  std::vector<TidInfo> ti((size_t)nt);
  std::vector<uint2> bins;   // bins[k] = {#starts < k << BIN_SHIFT, #starts < (k+1) << BIN_SHIFT}
  std::vector<int64_t> idx;
  for (int32_t t = 0; t < nt; ++t) {
    const int64_t a = g->iv_off[t], b = g->iv_off[t + 1];
    if (b - a > 0x7ffffff0ll) { set_error("too many intervals on tid %d", t); return STRL_ERR_ARG; }
    idx.resize((size_t)(b - a));
    for (int64_t i = a; i < b; ++i) idx[(size_t)(i - a)] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) { return g->iv_start[x] < g->iv_start[y]; });
    TidInfo &x = ti[(size_t)t];
    x.iv_off = (int64_t)ivs.size();
    int32_t run = INT32_MIN;
    for (int64_t i = a; i < b; ++i) {
      const int64_t s = idx[(size_t)(i - a)];
      st[(size_t)i] = g->iv_start[s];
      ivs.push_back(make_int2(g->iv_start[s], run));
      run = std::max(run, g->iv_stop[s]);
    }
    ivs.push_back(make_int2(INT32_MAX, run));
    x.n_iv = (int32_t)(b - a);
    x.has = g->has_chrom[t] ? 1 : 0;
    x.pad = 0;
    x.bin_off = (int64_t)bins.size();
    const int32_t max_start = b > a ? std::max(0, st[(size_t)(b - 1)]) : 0;
    x.n_bins = b > a ? (max_start >> BIN_SHIFT) + 1 : 0;
    int64_t j = a;
    uint32_t prev = 0;
    for (int32_t k = 0; k <= x.n_bins; ++k) {
      const int64_t lim = (int64_t)k << BIN_SHIFT;
      while (j < b && (int64_t)st[(size_t)j] < lim) ++j;
      const uint32_t cntk = (uint32_t)(j - a);
      if (k > 0) bins.push_back(make_uint2(prev, cntk));
      prev = cntk;
    }
  }
  if (bins.empty()) bins.push_back(make_uint2(0, 0));
  int rc;
  if ((rc = c->g_tid.reserve(ti.size() * sizeof(TidInfo)))) return rc;
  if ((rc = c->g_bins.reserve(bins.size() * 8))) return rc;
  if ((rc = c->g_start.reserve(ivs.size() * 8))) return rc;
  STRL_HIP(hipMemcpy(c->g_tid.p, ti.data(), ti.size() * sizeof(TidInfo), hipMemcpyHostToDevice));
  STRL_HIP(hipMemcpy(c->g_bins.p, bins.data(), bins.size() * 8, hipMemcpyHostToDevice));
  STRL_HIP(hipMemcpy(c->g_start.p, ivs.data(), ivs.size() * 8, hipMemcpyHostToDevice));
  c->n_tid = nt;
  c->n_iv = (uint64_t)niv;
  return STRL_OK;
}

// Bloom bitmap of the hot qname groups: ~n/2 bits (2 MB for 2^25 reads: L2 resident), two bits per key
static int bloom_reset(strl_ctx *c, uint64_t n) {
  uint64_t bits = 1ull << 16;
  while (bits < n / 2 && bits < (1ull << 27)) bits <<= 1;
  int rc;
  if ((rc = c->bloom.reserve((size_t)(bits / 8)))) return rc;
  STRL_HIP(zero_words(c->bloom.p, (size_t)(bits / 8), c->stream));
  c->bloom_mask = (uint32_t)(bits - 1);
  return STRL_OK;
}

namespace strl {
__global__ void meta_rows_kernel(const uint32_t *seq_off, const uint16_t *l_seq, const uint16_t *clip_l, const uint16_t *clip_r, const uint8_t *cig, const uint8_t *mapq,
                                 uint32_t n, uint4 *out);
}
// Reads of more than STRL_DEVICE_READ_LEN bases (extract.nim:36-40 scores any length; the reference's uint8 histograms wrap,
// utils.nim:192-195): behind the batch's launches the device lists them (index, row, skipped or not), packs their SEQ bytes,
// the host twin of the scorer (host_score.cpp) scores them on a few threads, and two small launches put the words where the
// kernels would have put them -- whole[], the Bloom mark, soft-clip records behind the device's own.  Synchronises the stream:
// only batches that hold such a read pay for it.
namespace strl {
__global__ void long_scan_kernel(const uint4 *meta, const uint32_t *whole, uint32_t n, uint4 *out, uint32_t *cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 m = meta[i];
  if ((m.y & 0xffffu) <= (uint32_t)STRL_DEVICE_READ_LEN) return;
  const uint32_t k = atomicAdd(cnt, 1u);
  out[k] = make_uint4(i | ((whole[i] & STRL_RES_SKIPPED) ? 0x80000000u : 0u), m.x, m.y, m.z);
}
// one wave per listed read: its SEQ bytes to off[k] of a dense buffer, as dwords (slots and offsets are 16-byte aligned)
__global__ __launch_bounds__(64) void long_gather_kernel(const uint8_t *seq4, const uint4 *list, const uint64_t *off, uint8_t *out) {
  const uint4 e = list[blockIdx.x];
  const uint32_t nb = ((e.z & 0xffffu) + 1u) / 2u, nd = (nb + 3u) / 4u;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(seq4 + (uint64_t)e.y * 16u);
  uint32_t *dst = reinterpret_cast<uint32_t *>(out + off[blockIdx.x]);
  for (uint32_t j = threadIdx.x; j < nd; j += 64u) dst[j] = src[j];
}
__global__ void long_patch_kernel(const uint32_t *ids, const uint32_t *words, uint32_t n, uint32_t *whole, const uint64_t *qhash, uint32_t *bloom, uint32_t bloom_mask) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t id = ids[k], w = words[k];
  whole[id] = w;
  if (bloom && STRL_RES_COUNT(w)) bloom_set(bloom, bloom_mask, fmix64(qhash[id]));
}
__global__ __launch_bounds__(1024) void long_soft_kernel(const strl_soft_rec *src, uint32_t n, strl_soft_rec *dst, uint32_t cap, uint32_t *counters) {
  __shared__ uint32_t base_sh;
  if (threadIdx.x == 0) base_sh = atomicAdd(&counters[CNT_SOFT], n);     // (one block, behind the segment scorer: the only writer now)
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
    if (base_sh + i < cap) dst[base_sh + i] = src[i];
}
}  // namespace strl

static int long_reads_pass(strl_ctx *c, const ScoreParams &P, uint64_t scap) {
  const uint32_t n = (uint32_t)P.n;
  int rc;
  if ((rc = c->long_list.reserve((size_t)n * 16 + 64))) return rc;
  uint32_t *cnt = c->long_list.as<uint32_t>() + (size_t)n * 4;            // the counter behind the list
  STRL_HIP(hipMemsetAsync(cnt, 0, 4, c->stream));
  hipLaunchKernelGGL(strl::long_scan_kernel, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, P.meta, P.whole, n, c->long_list.as<uint4>(), cnt);
  STRL_HIP(hipGetLastError());
  uint32_t n_long = 0;
  STRL_HIP(hipMemcpyAsync(&n_long, cnt, 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  if (!n_long) return STRL_OK;
  std::vector<uint4> list(n_long);
  STRL_HIP(hipMemcpy(list.data(), c->long_list.p, (size_t)n_long * 16, hipMemcpyDeviceToHost));
  std::vector<uint64_t> off((size_t)n_long + 1);
  uint64_t total = 0;
  for (uint32_t k = 0; k < n_long; ++k) { off[k] = total; total += ((((list[k].z & 0xffffu) + 1u) / 2u) + 15u) & ~15ull; }
  off[n_long] = total;
  if ((rc = c->long_seq.reserve((size_t)total + (size_t)(n_long + 1) * 8 + 64))) return rc;
  uint64_t *d_off = reinterpret_cast<uint64_t *>(c->long_seq.as<uint8_t>() + total);      // (total is a multiple of 16)
  STRL_HIP(hipMemcpyAsync(d_off, off.data(), (size_t)(n_long + 1) * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(strl::long_gather_kernel, dim3(n_long), dim3(64), 0, c->stream, P.seq4, c->long_list.as<uint4>(), d_off, c->long_seq.as<uint8_t>());
  STRL_HIP(hipGetLastError());
  std::vector<uint8_t> seq((size_t)total + 16);
  STRL_HIP(hipMemcpyAsync(seq.data(), c->long_seq.p, (size_t)total, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  std::vector<uint32_t> ids(n_long), words(n_long);
  std::vector<strl_soft_rec> soft((size_t)n_long * 2);
  std::vector<uint8_t> n_soft(n_long, 0);
  const strl_opts o = c->opts;
  auto run = [&](uint32_t k0, uint32_t k1) {
    for (uint32_t k = k0; k < k1; ++k) {
      const uint4 e = list[k];
      const uint32_t id = e.x & 0x7fffffffu;
      int ns = 0;
      uint32_t w = 0;
      strl::host_score_long_read(seq.data() + off[k], e.z & 0xffffu, e.z >> 16, e.w & 0xffffu, (e.w >> 16) & 0xffu, e.w >> 24, o, (e.x >> 31) != 0, id, w, &soft[(size_t)k * 2], ns);
      ids[k] = id;
      words[k] = (e.x >> 31) ? (uint32_t)STRL_RES_SKIPPED : w;
      n_soft[k] = (uint8_t)ns;
    }
  };
  const uint32_t workers = (uint32_t)std::min<uint64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n_long / 64 + 1);
  if (workers <= 1) run(0, n_long);
  else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < workers; ++t) th.emplace_back(run, (uint32_t)((uint64_t)n_long * t / workers), (uint32_t)((uint64_t)n_long * (t + 1) / workers));
    for (auto &t : th) t.join();
  }
  std::vector<strl_soft_rec> dense;
  for (uint32_t k = 0; k < n_long; ++k)
    for (int q = 0; q < n_soft[k]; ++q) dense.push_back(soft[(size_t)k * 2 + q]);
  const size_t b_ids = ((size_t)n_long * 4 + 15) & ~(size_t)15, b_soft = dense.size() * sizeof(strl_soft_rec);
  if ((rc = c->long_seq.reserve(2 * b_ids + b_soft + 64))) return rc;
  uint8_t *base = c->long_seq.as<uint8_t>();
  STRL_HIP(hipMemcpyAsync(base, ids.data(), (size_t)n_long * 4, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipMemcpyAsync(base + b_ids, words.data(), (size_t)n_long * 4, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(strl::long_patch_kernel, dim3((n_long + 255u) / 256u), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t *>(base),
                     reinterpret_cast<const uint32_t *>(base + b_ids), n_long, P.whole, P.qhash, P.bloom, P.bloom_mask);
  STRL_HIP(hipGetLastError());
  if (!dense.empty() && scap && P.soft_out) {
    STRL_HIP(hipMemcpyAsync(base + 2 * b_ids, dense.data(), b_soft, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(strl::long_soft_kernel, dim3(1), dim3(1024), 0, c->stream, reinterpret_cast<const strl_soft_rec *>(base + 2 * b_ids), (uint32_t)dense.size(), P.soft_out,
                       (uint32_t)std::min<uint64_t>(scap, P.soft_cap), P.counters);
    STRL_HIP(hipGetLastError());
  }
  STRL_HIP(hipStreamSynchronize(c->stream));       // (the host vectors above are the copies' sources)
  return STRL_OK;
}

static int score_device(strl_ctx *c, const strl_read_soa *s, uint32_t *whole, strl_soft_rec *soft, uint64_t soft_cap,
                        uint64_t *n_soft, strl_score_stats *stats, bool sync_counts, const strl_pair_soa *pp = nullptr,
                        bool fresh_bloom = true, bool side_busy_ok = false) {
  const uint64_t n = s->n;
  // the side stream may still read this context's counters / Bloom bitmap / results (pair logic of the previous batch):
  // every scoring pass waits for it, except the overlapped strl_extract_device, which works on the other set of buffers
  if (!side_busy_ok) { const int rcj = side_join(c); if (rcj) return rcj; }
  if (n > 0x3fffffffull) { set_error("batch too large (%llu reads; limit 2^30-1)", (unsigned long long)n); return STRL_ERR_ARG; }
  if (s->max_l_seq > STRL_MAX_READ_LEN) { set_error("read of %u bases exceeds STRL_MAX_READ_LEN=%d", s->max_l_seq, STRL_MAX_READ_LEN); return STRL_ERR_ARG; }
  // reads of more than STRL_DEVICE_READ_LEN bases: the kernels pass them by, the host twin scores them behind the launches
  const bool long_reads = s->max_l_seq > (uint32_t)STRL_DEVICE_READ_LEN;
  const uint32_t class_l = std::min<uint32_t>(s->max_l_seq, STRL_DEVICE_READ_LEN);
  int rc;
  const uint64_t n1 = std::max<uint64_t>(n, 1);
  const uint64_t scap = std::max<uint64_t>(std::min<uint64_t>(soft_cap, 2 * n), 1);
  if ((rc = c->queue.reserve((size_t)n1 * 16))) return rc;
  if ((rc = c->queue_r.reserve((size_t)n1 * 4))) return rc;
  if ((rc = c->soft_dense.reserve((size_t)n1 + 64))) return rc;
  if ((rc = c->soft_queue.reserve((size_t)scap * 16))) return rc;
  if ((rc = c->sb_state_w.reserve((size_t)n1 * 16))) return rc;
  if ((rc = c->sb_state_s.reserve((size_t)scap * 16))) return rc;
  if ((rc = c->sb_whole.reserve((size_t)n1 * 32))) return rc;
  if ((rc = c->sb_soft.reserve((size_t)scap * 32))) return rc;
  STRL_HIP(zero_words(c->counters.p, CNT_WORDS * 4, c->stream));
  ScoreParams P{};
  P.n = n;
  P.tid = s->tid; P.pos = s->pos; P.end = s->end; P.seq_off = s->seq_off; P.l_seq = s->l_seq;
  P.clip_l = s->clip_l; P.clip_r = s->clip_r; P.mapq = s->mapq; P.cig = s->cig; P.seq4 = s->seq4;
  P.meta = reinterpret_cast<const uint4 *>(s->meta);
  if (!P.meta) {   // device-resident columns without the packed rows: stage A gathers rows, so they are built here (a pass over the batch:
                   // callers that care hand over strl_read_soa.meta)
    if ((rc = c->st_meta.reserve(std::max<size_t>((size_t)n * 16, 64)))) return rc;
    if (n) {
      hipLaunchKernelGGL(strl::meta_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, s->seq_off, s->l_seq, s->clip_l, s->clip_r, s->cig, s->mapq,
                         (uint32_t)n, c->st_meta.as<uint4>());
      STRL_HIP(hipGetLastError());
    }
    P.meta = c->st_meta.as<uint4>();
  }
  if (!c->g_tid.p && (rc = strl_ctx_set_genome(c, nullptr))) return rc;   // never set: the empty table
  P.g_tid = c->g_tid.as<TidInfo>(); P.g_bins = c->g_bins.as<uint2>(); P.g_iv = c->g_start.as<int2>();
  P.n_tid = c->n_tid;
  P.lut = c->lut.as<uint16_t>(); P.thr = c->thr.as<uint64_t>();
  P.ta = c->lut.as<uint32_t>() + LUT_DWORDS + 256;
  P.whole = whole; P.queue = c->queue.as<uint4>(); P.queue_id = c->queue_r.as<uint32_t>(); P.soft_flag = c->soft_dense.as<uint8_t>();
  P.soft_queue = c->soft_queue.as<uint4>();
  P.sb_state[0] = c->sb_state_w.as<uint4>(); P.sb_state[1] = c->sb_state_s.as<uint4>();
  P.sb_queue[0] = c->sb_whole.as<uint4>(); P.sb_queue[1] = c->sb_soft.as<uint4>();
  P.scap = (uint32_t)scap;
  P.counters = c->counters.as<uint32_t>(); P.soft_out = soft;
  P.soft_cap = (uint32_t)std::min<uint64_t>(soft_cap, 0xffffffffull);
  P.min_mapq = c->opts.min_mapq;
  P.seg_row0 = 2; P.seg_row1 = 3;
  if (pp) {
    if (fresh_bloom && (rc = bloom_reset(c, n))) return rc;
    P.qhash = pp->qhash; P.bloom = c->bloom.as<uint32_t>(); P.bloom_mask = c->bloom_mask;
  }
  hipEvent_t *tev = c->timing ? &c->ring[(c->ring_pos % RING) * EV_PER] : nullptr;
  if (c->timing) ++c->ring_pos;
  if (tev) STRL_HIP(hipEventRecord(tev[0], c->stream));
  if (n) {
    static const int env_c = getenv("STRL_GRID_C") ? atoi(getenv("STRL_GRID_C")) : 0;
    const int cblocks = (int)std::min<uint64_t>((n + 255) / 256, env_c > 0 ? (uint64_t)env_c : 1024);   // one round: 256 CUs x 4 resident blocks (124 VGPRs, 16 KB of LDS); measured 768..10240
    const bool vec = (((uintptr_t)P.tid | (uintptr_t)P.pos | (uintptr_t)P.end | (uintptr_t)P.whole) & 15u) == 0 && ((uintptr_t)P.cig & 3u) == 0;
    if (vec) hipLaunchKernelGGL(classify_kernel<true>, dim3(cblocks), dim3(256), 0, c->stream, P);
    else hipLaunchKernelGGL(classify_kernel<false>, dim3(cblocks), dim3(256), 0, c->stream, P);
    STRL_HIP(hipGetLastError());
  }
  if (tev) STRL_HIP(hipEventRecord(tev[1], c->stream));
  if (n) { if ((rc = launch_score_class<0>(c, P, class_l, tev ? tev + 2 : nullptr))) return rc; }
  else if (tev) { STRL_HIP(hipEventRecord(tev[2], c->stream)); STRL_HIP(hipEventRecord(tev[3], c->stream)); }
  if (tev) STRL_HIP(hipEventRecord(tev[4], c->stream));
  if (n && soft_cap) {
    hipLaunchKernelGGL(soft_compact_kernel, dim3(1024), dim3(1024), 0, c->stream, P);
    STRL_HIP(hipGetLastError());
    if (tev) STRL_HIP(hipEventRecord(tev[5], c->stream));
    if ((rc = launch_score_class<1>(c, P, class_l, tev ? tev + 6 : nullptr))) return rc;
  } else if (tev) { for (int k = 5; k <= 7; ++k) STRL_HIP(hipEventRecord(tev[k], c->stream)); }
  if (tev) STRL_HIP(hipEventRecord(tev[8], c->stream));
  if (n && long_reads && (rc = long_reads_pass(c, P, soft_cap ? scap : 0))) return rc;
  if (sync_counts) {
    uint32_t raw[CNT_WORDS];
    STRL_HIP(hipMemcpyAsync(raw, c->counters.p, CNT_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
    STRL_HIP(hipStreamSynchronize(c->stream));
    const uint64_t softs = soft_cap ? raw[CNT_SOFT] : 0;
    if (softs > scap || softs > soft_cap) { set_error("soft-clip queue overflow: %llu items, capacity %llu", (unsigned long long)softs, (unsigned long long)std::min(scap, soft_cap)); return STRL_ERR_CAPACITY; }
    if (n_soft) *n_soft = softs;
    if (stats) {
      memset(stats, 0, sizeof *stats);
      stats->n_reads = n; stats->n_skipped = raw[CNT_SKIP]; stats->n_scored = raw[CNT_QUEUE]; stats->n_soft_items = softs;
      stats->n_stage_b_whole = raw[CNT_SBW]; stats->n_stage_b_soft = soft_cap ? raw[CNT_SBS] : 0;
      if (tev) {
        (void)hipEventElapsedTime(&stats->ms_classify, tev[0], tev[1]);
        (void)hipEventElapsedTime(&stats->ms_score, tev[1], tev[4]);
        (void)hipEventElapsedTime(&stats->ms_soft, tev[4], tev[8]);
      }
    }
  }
  return STRL_OK;
}

// host-memory batch -> staging buffers in HBM (asynchronous copies on the context stream)
namespace strl {
__global__ void meta_rows_kernel(const uint32_t *seq_off, const uint16_t *l_seq, const uint16_t *clip_l, const uint16_t *clip_r, const uint8_t *cig, const uint8_t *mapq,
                                 uint32_t n, uint4 *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_uint4(seq_off[i], (uint32_t)l_seq[i] | ((uint32_t)clip_l[i] << 16), (uint32_t)clip_r[i] | ((uint32_t)cig[i] << 16) | ((uint32_t)mapq[i] << 24), 0u);
}
}  // namespace strl

static int stage_batch(strl_ctx *c, const strl_read_soa *s, const strl_pair_soa *pp, strl_read_soa *d, strl_pair_soa *dpp) {
  const uint64_t n = s->n;
  *d = *s;
  int rc;
  struct { strl::DevBuf *b; const void *src; size_t bytes; const void **dst; } cp[] = {
      {&c->st_tid, s->tid, (size_t)n * 4, (const void **)&d->tid},         {&c->st_pos, s->pos, (size_t)n * 4, (const void **)&d->pos},
      {&c->st_end, s->end, (size_t)n * 4, (const void **)&d->end},         {&c->st_seqoff, s->seq_off, (size_t)n * 4, (const void **)&d->seq_off},
      {&c->st_lseq, s->l_seq, (size_t)n * 2, (const void **)&d->l_seq},    {&c->st_clipl, s->clip_l, (size_t)n * 2, (const void **)&d->clip_l},
      {&c->st_clipr, s->clip_r, (size_t)n * 2, (const void **)&d->clip_r}, {&c->st_mapq, s->mapq, (size_t)n, (const void **)&d->mapq},
      {&c->st_cig, s->cig, (size_t)n, (const void **)&d->cig},             {&c->st_seq4, s->seq4, (size_t)s->seq4_bytes, (const void **)&d->seq4}};
  for (auto &x : cp) {
    if ((rc = x.b->reserve(std::max<size_t>(x.bytes, 64)))) return rc;
    if (x.bytes) STRL_HIP(hipMemcpyAsync(x.b->p, x.src, x.bytes, hipMemcpyHostToDevice, c->stream));
    *x.dst = x.b->p;
  }
  d->mem = STRL_MEM_DEVICE;
  if ((rc = c->st_meta.reserve(std::max<size_t>((size_t)n * 16, 64)))) return rc;
  if (n) {
    hipLaunchKernelGGL(strl::meta_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d->seq_off, d->l_seq, d->clip_l, d->clip_r, d->cig, d->mapq,
                       (uint32_t)n, c->st_meta.as<uint4>());
    STRL_HIP(hipGetLastError());
  }
  d->meta = c->st_meta.as<strl_read_meta>();
  if (pp) {
    struct { strl::DevBuf *b; const void *src; size_t bytes; const void **dst; } cq[] = {
        {&c->st_mtid, pp->rec, (size_t)n * sizeof(strl_pair_rec), (const void **)&dpp->rec},
        {&c->st_qhash, pp->qhash, (size_t)n * 8, (const void **)&dpp->qhash}};
    for (auto &x : cq) {
      if ((rc = x.b->reserve(std::max<size_t>(x.bytes, 64)))) return rc;
      if (x.bytes) STRL_HIP(hipMemcpyAsync(x.b->p, x.src, x.bytes, hipMemcpyHostToDevice, c->stream));
      *x.dst = x.b->p;
    }
  }
  return STRL_OK;
}

int strl_index_chrom(strl_ctx *c, const char *seq, uint64_t n_bases, uint32_t window, uint32_t step, uint32_t *words, uint64_t *n_windows) {
  if (!c || (!seq && n_bases) || !window || !step) { set_error("bad argument"); return STRL_ERR_ARG; }
  if (!c->have_opts) { set_error("strl_ctx_set_opts must be called before scoring"); return STRL_ERR_ARG; }
  if (window > 160) { set_error("window %u too long (<= 160)", window); return STRL_ERR_ARG; }
  if (n_bases >= (1ull << 31)) { set_error("sequence too long (< 2^31 bases)"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  const uint64_t nw = n_bases ? (n_bases + step - 1) / step : 0;
  if (n_windows) *n_windows = nw;
  if (!nw || !words) return STRL_OK;
  const uint64_t n_dwords = (n_bases + 7) / 8 + 16;       // 64 bytes of zero slack behind the last base
  int rc;
  if ((rc = c->st_text.reserve((size_t)n_bases + 8))) return rc;
  if ((rc = c->st_seq4.reserve((size_t)n_dwords * 4))) return rc;
  if ((rc = c->soft_queue.reserve((size_t)nw * 16))) return rc;
  if ((rc = c->sb_state_s.reserve((size_t)nw * 16))) return rc;
  if ((rc = c->sb_soft.reserve((size_t)nw * 32))) return rc;
  if ((rc = c->st_soft.reserve((size_t)nw * sizeof(strl_soft_rec)))) return rc;
  STRL_HIP(hipMemcpyAsync(c->st_text.p, seq, (size_t)n_bases, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(pack_text_kernel, dim3((unsigned)std::min<uint64_t>((n_dwords + 255) / 256, 4096)), dim3(256), 0, c->stream,
                     c->st_text.as<uint8_t>(), c->st_seq4.as<uint32_t>(), n_bases, n_dwords);
  STRL_HIP(hipGetLastError());
  STRL_HIP(hipMemsetAsync(c->counters.p, 0, CNT_WORDS * 4, c->stream));
  ScoreParams P{};
  P.n = nw; P.seq4 = c->st_seq4.as<uint8_t>();
  P.lut = c->lut.as<uint16_t>(); P.thr = c->thr.as<uint64_t>();
  P.ta = c->lut.as<uint32_t>() + LUT_DWORDS + 256;
  P.soft_queue = c->soft_queue.as<uint4>(); P.scap = (uint32_t)nw;
  P.sb_state[1] = c->sb_state_s.as<uint4>(); P.sb_queue[1] = c->sb_soft.as<uint4>();
  P.counters = c->counters.as<uint32_t>(); P.soft_out = c->st_soft.as<strl_soft_rec>(); P.soft_cap = (uint32_t)nw;
  P.seg_row0 = 1; P.seg_row1 = 1;                       // genome windows are scored with the plain -p threshold (genome_strs.nim:74)
  hipLaunchKernelGGL(window_items_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c->stream, P, (uint32_t)nw, window, step, n_bases);
  STRL_HIP(hipGetLastError());
  if ((rc = launch_score_class<1>(c, P, window))) return rc;
  std::vector<strl_soft_rec> out((size_t)nw);
  STRL_HIP(hipMemcpyAsync(out.data(), c->st_soft.p, (size_t)nw * sizeof(strl_soft_rec), hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  for (uint64_t i = 0; i < nw; ++i) words[out[(size_t)i].read_side] = out[(size_t)i].res_first;
  return STRL_OK;
}

int strl_score_reads(strl_ctx *c, const strl_read_soa *s, uint32_t *whole, strl_soft_rec *soft, uint64_t soft_cap,
                     uint64_t *n_soft, strl_score_stats *stats) {
  if (!c || !s || (!whole && s->n)) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->have_opts) { set_error("strl_ctx_set_opts must be called before scoring"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  if (s->mem == STRL_MEM_DEVICE) {
    // n_soft == NULL and stats == NULL: fully asynchronous (bench / pipelines read counters later)
    return score_device(c, s, whole, soft, soft_cap, n_soft, stats, n_soft != nullptr || stats != nullptr);
  }
  // host batch: stage to HBM, run, copy results back
  const uint64_t n = s->n;
  strl_read_soa d;
  int rc;
  if ((rc = stage_batch(c, s, nullptr, &d, nullptr))) return rc;
  if ((rc = c->st_whole.reserve((size_t)std::max<uint64_t>(n, 1) * 4))) return rc;
  const uint64_t dcap = 2 * n + 2;   // at most two clipped ends per read
  if ((rc = c->st_soft.reserve((size_t)std::max<uint64_t>(dcap, 1) * sizeof(strl_soft_rec)))) return rc;
  uint64_t ns = 0;
  strl_score_stats st{};
  rc = score_device(c, &d, c->st_whole.as<uint32_t>(), c->st_soft.as<strl_soft_rec>(), dcap, &ns, &st, true);
  if (rc) return rc;
  if (ns > soft_cap) { set_error("soft capacity %llu too small, need %llu", (unsigned long long)soft_cap, (unsigned long long)ns); return STRL_ERR_CAPACITY; }
  if (n) STRL_HIP(hipMemcpyAsync(whole, c->st_whole.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  if (ns && soft) STRL_HIP(hipMemcpyAsync(soft, c->st_soft.p, (size_t)ns * sizeof(strl_soft_rec), hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  if (soft && ns) std::sort(soft, soft + ns, [](const strl_soft_rec &a, const strl_soft_rec &b) { return a.read_side < b.read_side; });
  if (n_soft) *n_soft = ns;
  if (stats) *stats = st;
  return STRL_OK;
}

int strl_extract_device(strl_ctx *c, const strl_read_soa *s, const strl_pair_soa *pp, int64_t n_tail, uint64_t item_cap, uint64_t tread_cap) {
  if (!c || !s || !pp) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->have_opts) { set_error("strl_ctx_set_opts must be called before scoring"); return STRL_ERR_ARG; }
  if (s->n && (!pp->rec || !pp->qhash)) { set_error("strl_extract_device: incomplete strl_pair_soa"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  const uint64_t n = s->n;
  if (n_tail < 0 || (uint64_t)n_tail > n) { set_error("strl_extract_device: n_tail must be in [0, n]"); return STRL_ERR_ARG; }
  if (!item_cap) item_cap = n / 8 + 65536;
  if (!tread_cap) tread_cap = n / 16 + 65536;
  item_cap = std::min<uint64_t>(item_cap, 3 * n + 16);    // every read and both of its clipped ends
  tread_cap = std::min<uint64_t>(tread_cap, 8 * n + 16);
  strl_read_soa d = *s;
  strl_pair_soa dp = *pp;
  int rc;
  if (s->mem != STRL_MEM_DEVICE && (rc = stage_batch(c, s, pp, &d, &dp))) return rc;
  const uint64_t soft_cap = std::min<uint64_t>(item_cap, 2 * n + 2);
  // Device-resident input: classify + scorer of this batch on the main stream, its pair logic on the side stream -- where
  // the previous batch's pair logic and clustering may still be running while this call's scorer already executes.
  static const bool no_overlap = getenv("STRL_NO_OVERLAP") != nullptr;
  const bool overlap = s->mem == STRL_MEM_DEVICE && !c->timing && !no_overlap;
  if (overlap) {
    if ((rc = side_streams(c))) return rc;
    rotate_head(c);                 // the scorer's output of this batch: the least recently used set
    rotate_tail(c);                 // this batch's pair logic and clustering: likewise, on that set's own side stream
    if ((rc = c->counters.reserve(CNT_WORDS * 4))) return rc;
    if (c->set_used[c->set]) STRL_HIP(hipStreamWaitEvent(c->stream, c->ev_set_free[c->set], 0));   // the side stream is done with this set
  }
  if ((rc = c->st_whole.reserve((size_t)std::max<uint64_t>(n, 1) * 4))) return rc;
  if ((rc = c->st_soft.reserve((size_t)std::max<uint64_t>(soft_cap, 1) * sizeof(strl_soft_rec)))) return rc;
  c->ex_n = n; c->ex_soft_cap = soft_cap; c->x_mode = false;
  if ((rc = score_device(c, &d, c->st_whole.as<uint32_t>(), c->st_soft.as<strl_soft_rec>(), soft_cap, nullptr, nullptr, false, &dp, true, overlap))) return rc;
  if (!overlap)
    return strl_pair_device(c, n, &dp, c->st_whole.as<uint32_t>(), c->st_soft.as<strl_soft_rec>(), c->counters.as<uint32_t>() + CNT_SOFT, soft_cap,
                            n_tail, item_cap, tread_cap);
  STRL_HIP(hipEventRecord(c->ev_head_done, c->stream));
  STRL_HIP(hipStreamWaitEvent(c->stream2, c->ev_head_done, 0));
  if ((rc = strl_pair_device(c, n, &dp, c->st_whole.as<uint32_t>(), c->st_soft.as<strl_soft_rec>(), c->counters.as<uint32_t>() + CNT_SOFT, soft_cap,
                             n_tail, item_cap, tread_cap, c->stream2)))
    return rc;
  STRL_HIP(hipEventRecord(c->ev_set_free[c->set], c->stream2));
  c->set_used[c->set] = true;
  STRL_HIP(hipEventRecord(c->ev_side_done, c->stream2));
  c->side_pending = true;
  c->pair_on_side = true;
  return STRL_OK;
}

// ---- the same in chunks: a BAM being decoded hands over batches in file order, the pair logic runs once at the end ----
namespace strl {
__global__ void soft_append_kernel(const strl_soft_rec *src, const uint32_t *cnt, uint32_t src_cap, uint32_t read_base, strl_soft_rec *dst,
                                   uint32_t dst_cap, uint32_t *xc) {
  __shared__ uint32_t base_sh;
  uint32_t n = cnt[CNT_SOFT];
  if (n > src_cap) n = src_cap;                 // (cannot happen: the per-chunk queue holds two records per read)
  if (threadIdx.x == 0) {
    base_sh = atomicAdd(&xc[XC_SOFT], n);   // one block: this is the only writer of the counter
    xc[XC_SKIP] += cnt[CNT_SKIP]; xc[XC_QUEUE] += cnt[CNT_QUEUE]; xc[XC_SBW] += cnt[CNT_SBW]; xc[XC_SBS] += cnt[CNT_SBS];
    xc[XC_SOFT_ITEMS] += cnt[CNT_SOFT];
  }
  __syncthreads();
  const uint32_t base = base_sh;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    strl_soft_rec s = src[i];
    s.read_side += read_base << 1;
    if (base + i < dst_cap) dst[base + i] = s;
  }
  if (threadIdx.x == 0 && (cnt[CNT_SOFT] > src_cap || (uint64_t)base + n > dst_cap)) xc[XC_OVERFLOW] = 1;   // reported by strl_treads_fetch
}
}  // namespace strl

namespace strl {
// multi-GPU extract, gather of the contexts' per-read state on one of them: soft-clip records carry the index of their read
// in the context that scored them -> its index in the file.  lbase / gbase: first local / global record of that context's chunks.
__global__ void soft_rebase_kernel(strl_soft_rec *soft, uint32_t n, const uint32_t *lbase, const uint32_t *gbase, uint32_t n_chunks) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t rs = soft[i].read_side, r = rs >> 1;
  uint32_t lo = 0, hi = n_chunks;              // last chunk with lbase <= r
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (lbase[mid] <= r) lo = mid; else hi = mid; }
  soft[i].read_side = ((gbase[lo] + (r - lbase[lo])) << 1) | (rs & 1u);
}
__global__ void qref_rebase_kernel(uint64_t *qref, uint32_t n, uint64_t arena_base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) qref[i] += arena_base << 8;
}
__global__ void words_or_kernel(uint32_t *dst, const uint32_t *src, size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) dst[i] |= src[i];
}
}  // namespace strl

static int copy_between(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st) {
  if (!bytes) return STRL_OK;
  if (dst_dev == src_dev) STRL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
  else STRL_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st));
  return STRL_OK;
}

// `strling extract --gpus N`: the chunks of one file went round-robin over n contexts (strl_front_push_after), each scored
// its chunks.  The pair logic needs every record of a qname group in one place, and what it needs of a record is small
// (32-byte row, hash, scorer word, name reference: 52 B against the ~290 B of the record and the work of inflating and
// scoring it): everything is gathered on ctxs[0] in FILE order -- per chunk copies over xGMI (peer DMA) --, soft-clip
// records and name references are re-based, the Bloom bitmaps OR-ed; ctxs[0] then is in the state of a one-GPU chunked
// extract of the whole file (strl_extract_finish, strl_front_fragwords, strl_front_qnames work as usual).
// chunk_owner[k] / chunk_records[k]: context and record count (strl_front_chunk.n_records) of the file's k-th chunk.
int strl_ctxs_extract_gather(strl_ctx **ctxs, int n, const uint32_t *chunk_owner, const uint64_t *chunk_records, uint64_t n_chunks) {
  using namespace strl;
  if (!ctxs || n < 1 || (n_chunks && (!chunk_owner || !chunk_records))) { set_error("strl_ctxs_extract_gather: bad argument"); return STRL_ERR_ARG; }
  for (int g = 0; g < n; ++g) {
    if (!ctxs[g] || !ctxs[g]->front || !ctxs[g]->x_open) { set_error("strl_ctxs_extract_gather: context %d has no open front end", g); return STRL_ERR_ARG; }
    if (ctxs[g]->bloom_mask != ctxs[0]->bloom_mask) { set_error("strl_ctxs_extract_gather: Bloom bitmaps differ in size"); return STRL_ERR_ARG; }
    STRL_HIP(hipSetDevice(ctxs[g]->device));
    STRL_HIP(hipStreamSynchronize(ctxs[g]->stream));
  }
  if (n == 1) return STRL_OK;
  strl_ctx *c0 = ctxs[0];
  strl_front *F0 = c0->front;
  std::vector<uint64_t> local_n((size_t)n, 0), gbase((size_t)n_chunks, 0), lbase((size_t)n_chunks, 0);
  uint64_t tot = 0;
  for (uint64_t k = 0; k < n_chunks; ++k) {
    if (chunk_owner[k] >= (uint32_t)n) { set_error("strl_ctxs_extract_gather: chunk owner out of range"); return STRL_ERR_ARG; }
    gbase[(size_t)k] = tot; lbase[(size_t)k] = local_n[chunk_owner[k]];
    tot += chunk_records[k]; local_n[chunk_owner[k]] += chunk_records[k];
  }
  for (int g = 0; g < n; ++g)
    if (local_n[(size_t)g] != ctxs[g]->x_n) { set_error("strl_ctxs_extract_gather: context %d holds %llu records, its chunks say %llu", g, (unsigned long long)ctxs[g]->x_n, (unsigned long long)local_n[(size_t)g]); return STRL_ERR_ARG; }
  if (tot > strl_record_limit()) { set_error("chunked extract: more than %llu records in one device pass", (unsigned long long)strl_record_limit()); return STRL_ERR_LIMIT; }
  // totals of the soft-clip records, the name arenas, the counters
  std::vector<uint32_t> xc((size_t)n * XC_WORDS);
  std::vector<uint64_t> soft_at((size_t)n + 1, 0), arena_at((size_t)n + 1, 0);
  for (int g = 0; g < n; ++g) {
    STRL_HIP(hipSetDevice(ctxs[g]->device));
    STRL_HIP(hipMemcpy(&xc[(size_t)g * XC_WORDS], ctxs[g]->x_cnt.p, XC_WORDS * 4, hipMemcpyDeviceToHost));
    if (xc[(size_t)g * XC_WORDS + XC_OVERFLOW]) { set_error("chunked extract: soft-clip records of a chunk were dropped"); return STRL_ERR_CAPACITY; }
    soft_at[(size_t)g + 1] = soft_at[(size_t)g] + xc[(size_t)g * XC_WORDS + XC_SOFT];
    arena_at[(size_t)g + 1] = arena_at[(size_t)g] + ctxs[g]->front->qarena_used;
  }
  STRL_HIP(hipSetDevice(c0->device));
  hipStream_t st = c0->stream;
  const uint64_t t1 = std::max<uint64_t>(tot, 1), s1 = std::max<uint64_t>(soft_at[(size_t)n], 1);
  // Shares (the first context holds the FIRST records of the file, all of them, and nothing else): its columns stay where
  // they are and the other contexts' shares are appended behind them -- no second copy of the per-read state, no copy of the
  // first share, and no allocation when strl_front_begin sized the first context for the whole file (the CLI does).
  bool in_place = true;
  {
    uint64_t own = 0;
    for (uint64_t k = 0; k < n_chunks; ++k) {
      if (chunk_owner[k] == 0) { if (gbase[(size_t)k] != lbase[(size_t)k]) in_place = false; own += chunk_records[k]; }
    }
    if (own != local_n[0]) in_place = false;
  }
  DevBuf rows, qhash, whole, qref, fragw, soft, arena, tmp, tab;
  int rc;
  if (in_place) {
    const uint64_t n0 = local_n[0];
    if ((rc = c0->x_rows.grow((size_t)t1 * sizeof(strl_pair_rec), (size_t)n0 * sizeof(strl_pair_rec), st)) || (rc = c0->x_qhash.grow((size_t)t1 * 8, (size_t)n0 * 8, st)) ||
        (rc = c0->x_whole.grow((size_t)t1 * 4, (size_t)n0 * 4, st)) || (rc = F0->qref.grow((size_t)t1 * 8, (size_t)n0 * 8, st)) || (rc = F0->fragw.grow((size_t)t1 * 4, (size_t)n0 * 4, st)) ||
        (rc = c0->x_soft.grow((size_t)s1 * sizeof(strl_soft_rec), (size_t)soft_at[1] * sizeof(strl_soft_rec), st)) ||
        (rc = F0->qarena.grow((size_t)arena_at[(size_t)n] + 64, (size_t)arena_at[1], st)))
      return rc;
    rows = c0->x_rows; qhash = c0->x_qhash; whole = c0->x_whole; qref = F0->qref; fragw = F0->fragw; soft = c0->x_soft; arena = F0->qarena;     // (views: ownership stays with the context)
  } else if ((rc = rows.reserve((size_t)t1 * sizeof(strl_pair_rec))) || (rc = qhash.reserve((size_t)t1 * 8)) || (rc = whole.reserve((size_t)t1 * 4)) ||
             (rc = qref.reserve((size_t)t1 * 8)) || (rc = fragw.reserve((size_t)t1 * 4)) || (rc = soft.reserve((size_t)s1 * sizeof(strl_soft_rec))) ||
             (rc = arena.reserve((size_t)arena_at[(size_t)n] + 64)))
    return rc;
  if ((rc = tmp.reserve(std::max<size_t>((size_t)c0->bloom_mask / 8 + 64, (size_t)F0->n_ref + 64))) || (rc = tab.reserve((size_t)std::max<uint64_t>(n_chunks, 1) * 8 + 64))) return rc;
  // Runs of consecutive chunks of one owner (a share = one run) are contiguous on both sides: one copy per column.  A
  // context's columns, names and soft-clip records travel on ITS stream -- each source device drives its own link to the first,
  // the links work side by side -- and the first context's stream waits for one event per source before it re-bases.
  struct Run { uint32_t owner; uint64_t lo, go, m; };
  std::vector<Run> runs;
  for (uint64_t k = 0; k < n_chunks; ++k) {
    const uint64_t m = chunk_records[k];
    if (!m) continue;
    if (!runs.empty() && runs.back().owner == chunk_owner[k] && runs.back().lo + runs.back().m == lbase[(size_t)k] && runs.back().go + runs.back().m == gbase[(size_t)k]) runs.back().m += m;
    else runs.push_back(Run{chunk_owner[k], lbase[(size_t)k], gbase[(size_t)k], m});
  }
  std::vector<hipEvent_t> src_done((size_t)n, nullptr);
  auto drop_events = [&] { for (hipEvent_t e : src_done) if (e) (void)hipEventDestroy(e); };
  std::vector<std::vector<uint32_t>> tls((size_t)n), tgs((size_t)n);
  for (int g = 0; g < n; ++g) {
    strl_ctx *cg = ctxs[g];
    STRL_HIP(hipSetDevice(cg->device));
    hipStream_t sg = cg->stream;
    if (g == 0 && in_place) continue;            // its records, names and soft-clip records are where they belong already
    for (const Run &r : runs) {
      if (r.owner != (uint32_t)g) continue;
      if ((rc = copy_between(rows.as<strl_pair_rec>() + r.go, c0->device, cg->x_rows.as<strl_pair_rec>() + r.lo, cg->device, (size_t)r.m * sizeof(strl_pair_rec), sg)) ||
          (rc = copy_between(qhash.as<uint64_t>() + r.go, c0->device, cg->x_qhash.as<uint64_t>() + r.lo, cg->device, (size_t)r.m * 8, sg)) ||
          (rc = copy_between(whole.as<uint32_t>() + r.go, c0->device, cg->x_whole.as<uint32_t>() + r.lo, cg->device, (size_t)r.m * 4, sg)) ||
          (rc = copy_between(qref.as<uint64_t>() + r.go, c0->device, cg->front->qref.as<uint64_t>() + r.lo, cg->device, (size_t)r.m * 8, sg)) ||
          (rc = copy_between(fragw.as<uint32_t>() + r.go, c0->device, cg->front->fragw.as<uint32_t>() + r.lo, cg->device, (size_t)r.m * 4, sg))) { drop_events(); return rc; }
    }
    if ((rc = copy_between(arena.as<uint8_t>() + arena_at[(size_t)g], c0->device, cg->front->qarena.p, cg->device, (size_t)cg->front->qarena_used, sg))) { drop_events(); return rc; }
    const uint64_t ns = soft_at[(size_t)g + 1] - soft_at[(size_t)g];
    if (ns && (rc = copy_between(soft.as<strl_soft_rec>() + soft_at[(size_t)g], c0->device, cg->x_soft.p, cg->device, (size_t)ns * sizeof(strl_soft_rec), sg))) { drop_events(); return rc; }
    if (g) {      // (an event of the SOURCE's device on the source's stream; the wait below is the cross-device half, which is legal)
      STRL_HIP(hipEventCreateWithFlags(&src_done[(size_t)g], hipEventDisableTiming));
      STRL_HIP(hipEventRecord(src_done[(size_t)g], sg));
    }
  }
  STRL_HIP(hipSetDevice(c0->device));
  // re-basing on the first context, behind each source's copies
  size_t tab_at = 0;
  for (int g = 0; g < n; ++g) {
    strl_ctx *cg = ctxs[g];
    if (g) STRL_HIP(hipStreamWaitEvent(st, src_done[(size_t)g], 0));
    const uint64_t ab = arena_at[(size_t)g];
    if (ab)
      for (const Run &r : runs) {
        if (r.owner != (uint32_t)g) continue;
        hipLaunchKernelGGL(qref_rebase_kernel, dim3((unsigned)((r.m + 255) / 256)), dim3(256), 0, st, qref.as<uint64_t>() + r.go, (uint32_t)r.m, ab);
        STRL_HIP(hipGetLastError());
      }
    const uint64_t ns = soft_at[(size_t)g + 1] - soft_at[(size_t)g];
    if (ns && !(g == 0 && in_place)) {
      std::vector<uint32_t> &tl = tls[(size_t)g], &tg = tgs[(size_t)g];      // this context's chunks: first local / first global record
      for (uint64_t k = 0; k < n_chunks; ++k) if (chunk_owner[k] == (uint32_t)g) { tl.push_back((uint32_t)lbase[(size_t)k]); tg.push_back((uint32_t)gbase[(size_t)k]); }
      STRL_HIP(hipMemcpyAsync(tab.as<uint32_t>() + tab_at, tl.data(), tl.size() * 4, hipMemcpyHostToDevice, st));
      STRL_HIP(hipMemcpyAsync(tab.as<uint32_t>() + n_chunks + 8 + tab_at, tg.data(), tg.size() * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(soft_rebase_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, soft.as<strl_soft_rec>() + soft_at[(size_t)g], (uint32_t)ns,
                         tab.as<uint32_t>() + tab_at, tab.as<uint32_t>() + n_chunks + 8 + tab_at, (uint32_t)tl.size());
      STRL_HIP(hipGetLastError());
      tab_at += tl.size();
    }
    if (g) {       // contigs that had a primary record (the CLI's "extracting chromosome" lines)
      const size_t tw = ((size_t)std::min(F0->n_ref, cg->front->n_ref) + 3) / 4;
      if (tw) {
        if ((rc = copy_between(tmp.p, c0->device, cg->front->tid_seen.p, cg->device, tw * 4, st))) { drop_events(); return rc; }
        hipLaunchKernelGGL(words_or_kernel, dim3(16), dim3(256), 0, st, F0->tid_seen.as<uint32_t>(), tmp.as<uint32_t>(), tw);
        STRL_HIP(hipGetLastError());
      }
    }
    if (g) {
      const size_t bw = ((size_t)c0->bloom_mask + 1) / 32;
      if ((rc = copy_between(tmp.p, c0->device, cg->bloom.p, cg->device, bw * 4, st))) { drop_events(); return rc; }
      hipLaunchKernelGGL(words_or_kernel, dim3(1024), dim3(256), 0, st, c0->bloom.as<uint32_t>(), tmp.as<uint32_t>(), bw);
      STRL_HIP(hipGetLastError());
    }
  }
  uint32_t sum[XC_WORDS] = {0};
  for (int g = 0; g < n; ++g) for (int w = 0; w < XC_WORDS; ++w) sum[w] += xc[(size_t)g * XC_WORDS + w];
  STRL_HIP(hipStreamSynchronize(st));
  drop_events();
  STRL_HIP(hipMemcpy(c0->x_cnt.p, sum, XC_WORDS * 4, hipMemcpyHostToDevice));
  // ctxs[0] takes the gathered state over
  if (!in_place) {
    c0->x_rows.release(); c0->x_qhash.release(); c0->x_whole.release(); c0->x_soft.release();
    F0->qref.release(); F0->fragw.release(); F0->qarena.release();
    c0->x_rows = rows; c0->x_qhash = qhash; c0->x_whole = whole; c0->x_soft = soft;
    F0->qref = qref; F0->fragw = fragw; F0->qarena = arena;
  }
  F0->qarena_used = arena_at[(size_t)n];
  c0->x_n = tot;
  c0->x_soft_cap = in_place ? c0->x_soft.cap / sizeof(strl_soft_rec) : s1;
  c0->x_soft_known = soft_at[(size_t)n]; c0->x_soft_known_at = tot;
  tmp.release(); tab.release();
  return STRL_OK;
}

// n_now: reads the big per-read columns are sized for right away (0: the hint); the front end passes a fraction and has the
// rest allocated beside its first chunks (FrontBigAlloc)
static int extract_begin_sized(strl_ctx *c, uint64_t n_reads_hint, uint64_t n_now);
int strl_extract_begin(strl_ctx *c, uint64_t n_reads_hint) { return extract_begin_sized(c, n_reads_hint, 0); }
static int extract_begin_sized(strl_ctx *c, uint64_t n_reads_hint, uint64_t n_now) {
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->have_opts) { set_error("strl_ctx_set_opts must be called before scoring"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  int rc;
  const uint64_t hint = std::max<uint64_t>(n_reads_hint, 1 << 20), first = n_now ? std::min(n_now, hint) : hint;
  if ((rc = c->x_rows.grow((size_t)first * sizeof(strl_pair_rec), 0, c->stream)) || (rc = c->x_qhash.grow((size_t)first * 8, 0, c->stream)) ||
      (rc = c->x_whole.grow((size_t)first * 4, 0, c->stream)) || (rc = c->x_soft.grow((size_t)(hint / 8 + 65536) * sizeof(strl_soft_rec), 0, c->stream)) ||
      (rc = c->x_cnt.reserve(XC_WORDS * 4)))
    return rc;
  STRL_HIP(hipMemsetAsync(c->x_cnt.p, 0, XC_WORDS * 4, c->stream));
  if ((rc = bloom_reset(c, std::max<uint64_t>(hint, 1ull << 28)))) return rc;   // 16 MB: sized for a whole genome of reads
  c->x_n = 0; c->x_soft_cap = c->x_soft.cap / sizeof(strl_soft_rec); c->x_open = true; c->x_front = false;
  c->x_soft_known = 0; c->x_soft_known_at = 0;
  if (c->x_soft_pending) { STRL_HIP(hipEventSynchronize(c->x_soft_seen_ev)); c->x_soft_pending = false; }
  return STRL_OK;
}

// the per-chunk part shared by strl_extract_add and the device front end: score the device-resident chunk `d` whose rows and
// qname hashes already sit at x_rows / x_qhash [at, at + n), append its soft-clip records
static int extract_add_scored(strl_ctx *c, const strl_read_soa *d, uint64_t at) {
  const uint64_t n = d->n;
  int rc;
  // Soft-clip records: a chunk can add two per read (its hard bound, which the per-chunk queue is sized for), the typical
  // rate is a few per cent.  The running total lives on the device; the host keeps an upper bound of it -- the last total it
  // has seen (read back asynchronously behind every chunk) plus two per read added since -- and grows x_soft ahead of that.
  if (c->x_soft_seen_ev) {
    while (c->x_soft_pending && hipEventQuery(c->x_soft_seen_ev) == hipSuccess) {
      c->x_soft_known = *c->x_soft_seen;
      c->x_soft_known_at = c->x_soft_seen_at;
      c->x_soft_pending = false;
    }
  }
  const uint64_t bound = c->x_soft_known + 2 * ((at + n) - c->x_soft_known_at) + 2;
  if (bound > c->x_soft_cap) {
    c->x_soft_cap = std::max<uint64_t>(bound, (at + n) / 4 + 65536);
    if ((rc = c->x_soft.grow((size_t)c->x_soft_cap * sizeof(strl_soft_rec), c->x_soft.cap, c->stream))) return rc;
  }
  const strl_pair_soa dp{c->x_rows.as<strl_pair_rec>() + at, c->x_qhash.as<uint64_t>() + at};
  const uint64_t chunk_soft = 2 * n + 2;
  if ((rc = c->st_soft.reserve((size_t)chunk_soft * sizeof(strl_soft_rec)))) return rc;
  // the skip-predicate pass stores its words 16 bytes at a time when the destination is aligned (its other variant is ~80x
  // slower): a chunk that starts at an index that is not a multiple of 4 is scored into a scratch array and copied over
  uint32_t *whole = c->x_whole.as<uint32_t>() + at;
  const bool bounce = (at & 3u) != 0;
  if (bounce) {
    if ((rc = c->st_whole.reserve((size_t)std::max<uint64_t>(n, 1) * 4))) return rc;
    whole = c->st_whole.as<uint32_t>();
  }
  if ((rc = score_device(c, d, whole, c->st_soft.as<strl_soft_rec>(), chunk_soft, nullptr, nullptr, false, &dp, false))) return rc;
  if (bounce && n) STRL_HIP(hipMemcpyAsync(c->x_whole.as<uint32_t>() + at, whole, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(strl::soft_append_kernel, dim3(1), dim3(1024), 0, c->stream, c->st_soft.as<strl_soft_rec>(), c->counters.as<uint32_t>(), (uint32_t)chunk_soft,
                     (uint32_t)at, c->x_soft.as<strl_soft_rec>(), (uint32_t)std::min<uint64_t>(c->x_soft_cap, 0xffffffffull), c->x_cnt.as<uint32_t>());
  STRL_HIP(hipGetLastError());
  if (!c->x_soft_seen_ev) {
    STRL_HIP(hipEventCreateWithFlags(&c->x_soft_seen_ev, hipEventDisableTiming));
    STRL_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->x_soft_seen), 64, hipHostMallocDefault));
  }
  if (!c->x_soft_pending) {
    STRL_HIP(hipMemcpyAsync(c->x_soft_seen, c->x_cnt.as<uint32_t>() + XC_SOFT, 4, hipMemcpyDeviceToHost, c->stream));
    STRL_HIP(hipEventRecord(c->x_soft_seen_ev, c->stream));
    c->x_soft_seen_at = at + n;
    c->x_soft_pending = true;
  }
  c->x_n = at + n;
  return STRL_OK;
}

int strl_extract_add(strl_ctx *c, const strl_read_soa *s, const strl_pair_soa *pp) {
  if (!c || !s || !pp) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->x_open) { set_error("strl_extract_add without strl_extract_begin"); return STRL_ERR_ARG; }
  if (s->n && (!pp->rec || !pp->qhash)) { set_error("strl_extract_add: incomplete strl_pair_soa"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  const uint64_t n = s->n, at = c->x_n;
  if (!n) return STRL_OK;
  if (at + n > strl_record_limit()) { set_error("chunked extract: more than %llu records in one device pass", (unsigned long long)strl_record_limit()); return STRL_ERR_LIMIT; }
  int rc;
  if ((rc = c->x_rows.grow((size_t)(at + n) * sizeof(strl_pair_rec), (size_t)at * sizeof(strl_pair_rec), c->stream)) ||
      (rc = c->x_qhash.grow((size_t)(at + n) * 8, (size_t)at * 8, c->stream)) || (rc = c->x_whole.grow((size_t)(at + n) * 4, (size_t)at * 4, c->stream)))
    return rc;
  strl_read_soa d = *s;
  const hipMemcpyKind kind = s->mem == STRL_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (s->mem != STRL_MEM_DEVICE && (rc = stage_batch(c, s, nullptr, &d, nullptr))) return rc;
  STRL_HIP(hipMemcpyAsync(c->x_rows.as<strl_pair_rec>() + at, pp->rec, (size_t)n * sizeof(strl_pair_rec), kind, c->stream));
  STRL_HIP(hipMemcpyAsync(c->x_qhash.as<uint64_t>() + at, pp->qhash, (size_t)n * 8, kind, c->stream));
  return extract_add_scored(c, &d, at);
}

int strl_extract_finish(strl_ctx *c, int64_t n_tail, uint64_t item_cap, uint64_t tread_cap) {
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->x_open && !c->x_mode) { set_error("strl_extract_finish without strl_extract_begin"); return STRL_ERR_ARG; }   // (again after a capacity error: fine)
  STRL_HIP(hipSetDevice(c->device));
  const uint64_t n = c->x_n;
  if (n_tail < 0 || (uint64_t)n_tail > n) { set_error("strl_extract_finish: n_tail must be in [0, n]"); return STRL_ERR_ARG; }
  if (!item_cap) item_cap = n / 8 + 65536;
  if (!tread_cap) tread_cap = n / 16 + 65536;
  item_cap = std::min<uint64_t>(item_cap, 3 * n + 16);
  tread_cap = std::min<uint64_t>(tread_cap, 8 * n + 16);
  c->x_open = false; c->x_mode = true;
  c->ex_n = n; c->ex_soft_cap = c->x_soft_cap;
  const strl_pair_soa dp{c->x_rows.as<strl_pair_rec>(), c->x_qhash.as<uint64_t>()};
  return strl_pair_device(c, n, &dp, c->x_whole.as<uint32_t>(), c->x_soft.as<strl_soft_rec>(), c->x_cnt.as<uint32_t>() + XC_SOFT,
                          std::max<uint64_t>(c->x_soft_cap, 1), n_tail, item_cap, tread_cap);
}

int strl_treads_fetch(strl_ctx *c, strl_tread *out, uint64_t cap, uint64_t *n_out, strl_score_stats *stats) {
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!c->n_treads_dev) { set_error("strl_treads_fetch: no strl_extract_device call on this context"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  { const int rc0 = strl_pair_order(c); if (rc0) return rc0; }
  uint32_t raw[CNT_WORDS], pc[PC_WORDS], xc[XC_WORDS];
  STRL_HIP(hipMemcpyAsync(raw, c->counters.p, CNT_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipMemcpyAsync(pc, c->pair_cnt.p, PC_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
  if (c->x_mode) STRL_HIP(hipMemcpyAsync(xc, c->x_cnt.p, XC_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  if (c->x_mode) {   // chunked extract: the sums over the chunks
    raw[CNT_SKIP] = xc[XC_SKIP]; raw[CNT_QUEUE] = xc[XC_QUEUE]; raw[CNT_SBW] = xc[XC_SBW]; raw[CNT_SBS] = xc[XC_SBS]; raw[CNT_SOFT] = xc[XC_SOFT];
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->n_reads = c->ex_n; stats->n_skipped = raw[CNT_SKIP]; stats->n_scored = raw[CNT_QUEUE]; stats->n_soft_items = raw[CNT_SOFT];
    stats->n_stage_b_whole = raw[CNT_SBW]; stats->n_stage_b_soft = raw[CNT_SBS];
  }
  if (n_out) *n_out = pc[PC_EMIT];
  if (c->x_mode && xc[XC_OVERFLOW]) { set_error("chunked extract: soft-clip records of a chunk were dropped (%u kept of %u)", xc[XC_SOFT], xc[XC_SOFT_ITEMS]); return STRL_ERR_CAPACITY; }
  if (raw[CNT_SOFT] > c->ex_soft_cap) { set_error("soft-clip queue overflow: %u items, capacity %llu (raise item_cap)", raw[CNT_SOFT], (unsigned long long)c->ex_soft_cap); return STRL_ERR_CAPACITY; }
  const uint32_t err = pc[PC_ERR];
  if (err & PAIR_ERR_ITEMS) { set_error("pair logic: %u join items, capacity %u (raise item_cap)", pc[PC_ITEMS], c->pair_item_cap); return STRL_ERR_CAPACITY; }
  if (err & PAIR_ERR_EMIT) { set_error("pair logic: %u treads, capacity %u (raise tread_cap)", pc[PC_EMIT], c->tread_cap); return STRL_ERR_CAPACITY; }
  if (err & PAIR_ERR_COLLISION) { set_error("pair logic: two different qnames share one 64-bit hash (use the host pair logic, strl_pair_reads: it keys on the string)"); return STRL_ERR_FORMAT; }
  if (err & (PAIR_ERR_RUN | PAIR_ERR_LOCAL)) { set_error("pair logic: more than %d join items share the low 32 bits of their qname hash (use the host pair logic, strl_pair_reads)", strl::PAIR_LONG_MAX_ITEMS); return STRL_ERR_FORMAT; }
  if (err & PAIR_ERR_ASSERT) { set_error("repeat_count >= 256 (doAssert extract.nim:72)"); return STRL_ERR_ASSERT; }
  const uint64_t n = pc[PC_EMIT];
  if (out) {
    if (n > cap) { set_error("tread capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)n); return STRL_ERR_CAPACITY; }
    if (n) STRL_HIP(hipMemcpy(out, c->treads.p, (size_t)n * sizeof(strl_tread), hipMemcpyDeviceToHost));
  }
  return STRL_OK;
}

int strl_ctx_pair_times(strl_ctx *c, double ms[5]) {
  if (!c || !ms) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 5; ++k) {
    float f = 0.f;
    if (c->timing) (void)hipEventElapsedTime(&f, c->pev[k], c->pev[k + 1]);
    ms[k] = f;
  }
  return STRL_OK;
}


// ---- `strling extract` with the BAM front end on the device (front.hip): the host hands over BGZF blocks, never a record ----
static int front_fill_done(strl_ctx *c, strl::FrontSlot &S, strl_front_chunk *done) {
  STRL_HIP(hipEventSynchronize(S.ev_b));
  S.b_pending = false;
  const strl::FrontInfo &I = S.h_info[1];
  if (I.err & strl::FRONT_ERR_LSEQ) { set_error("a record's l_seq is outside [0, %d]", STRL_MAX_READ_LEN); return STRL_ERR_ARG; }   // (kept: stage B refuses the chunk before the parse)
  if (done) {
    done->n_records = I.n_records; done->n_primary = I.n_primary; done->last_placed = I.last_placed; done->tail_primary = I.tail_primary;
    done->max_l_seq = I.max_l_seq; done->scan_slow_segments = I.slow_segments;
  }
  return STRL_OK;
}

// the full-size per-read buffers are there (or the small ones are full: then this waits for them): what the chunks so far have
// filled is copied over on the context's stream -- behind every kernel that wrote it -- and the small buffers are kept until the
// front end goes (nothing waits for them to be free)
static int front_adopt_big(strl_ctx *c, strl::strl_front *F, uint64_t at) {
  strl::FrontBigAlloc *B = F->big;
  if (!B) return STRL_OK;
  if (B->th.joinable()) B->th.join();
  F->big = nullptr;
  if (B->rc) {
    set_error("%s", B->err.c_str());
    const int rc = B->rc;
    for (strl::DevBuf *b : {&B->rows, &B->qhash, &B->whole, &B->qref, &B->fragw, &B->qarena}) b->release();
    delete B;
    return rc;
  }
  struct Mv { strl::DevBuf *cur, *big; size_t used; };
  const Mv mv[6] = {{&c->x_rows, &B->rows, (size_t)at * sizeof(strl_pair_rec)}, {&c->x_qhash, &B->qhash, (size_t)at * 8}, {&c->x_whole, &B->whole, (size_t)at * 4},
                    {&F->qref, &B->qref, (size_t)at * 8}, {&F->fragw, &B->fragw, (size_t)at * 4}, {&F->qarena, &B->qarena, (size_t)F->qarena_used}};
  for (const Mv &m : mv) {
    if (m.big->cap <= m.cur->cap) { m.big->release(); continue; }        // (the small one grew past it meanwhile)
    if (m.used) STRL_HIP(hipMemcpyAsync(m.big->p, m.cur->p, std::min(m.used, m.cur->cap), hipMemcpyDeviceToDevice, c->stream));
    F->trash.push_back(*m.cur);
    *m.cur = *m.big;
    m.big->p = nullptr; m.big->cap = 0;
  }
  delete B;
  return STRL_OK;
}

// parse + score the chunk in slot si (its record scan was enqueued earlier): waits on the HOST for the scan's counts -- the
// next chunk's inflate is already queued behind it, so the device does not idle
static int front_stage_b(strl_ctx *c, strl::strl_front *F, int si) {
  using namespace strl;
  FrontSlot &S = F->slot[si];
  STRL_HIP(hipEventSynchronize(S.ev_a));
  const FrontInfo I = S.h_info[0];
  if (I.err & FRONT_ERR_INFLATE) { set_error("invalid BGZF block (DEFLATE data or ISIZE)"); return STRL_ERR_FORMAT; }
  if (I.err & FRONT_ERR_CRC) { set_error("CRC32 checksum mismatch in a BGZF block"); return STRL_ERR_CRC; }
  if (I.err & FRONT_ERR_RECORD) { set_error("malformed BAM record"); return STRL_ERR_FORMAT; }
  if (I.err & FRONT_ERR_CARRY) { set_error("BAM record of more than %u bytes", FRONT_CARRY_MAX); return STRL_ERR_FORMAT; }
  const uint64_t n = I.n_records, at = c->x_n;
  if (at + n > strl_record_limit()) { set_error("chunked extract: more than %llu records in one device pass", (unsigned long long)strl_record_limit()); return STRL_ERR_LIMIT; }
  if (I.max_l_seq > (uint32_t)STRL_MAX_READ_LEN) { set_error("a record's l_seq %u is outside [0, %d]", I.max_l_seq, STRL_MAX_READ_LEN); return STRL_ERR_ARG; }
  int rc;       // (records of STRL_DEVICE_READ_LEN < l_seq <= STRL_MAX_READ_LEN bases: scored by the host twin inside score_device)
  const uint64_t n1 = std::max<uint64_t>(n, 1);
  if (F->big && (F->big->done.load(std::memory_order_acquire) || at + n1 > F->small_reads || F->qarena_used + I.qname_bytes + 16 > F->qarena.cap) && (rc = front_adopt_big(c, F, at))) return rc;
  if ((rc = c->x_rows.grow((size_t)(at + n1) * sizeof(strl_pair_rec), (size_t)at * sizeof(strl_pair_rec), c->stream)) ||
      (rc = c->x_qhash.grow((size_t)(at + n1) * 8, (size_t)at * 8, c->stream)) || (rc = c->x_whole.grow((size_t)(at + n1) * 4, (size_t)at * 4, c->stream)) ||
      (rc = F->qref.grow((size_t)(at + n1) * 8, (size_t)at * 8, c->stream)) || (rc = F->fragw.grow((size_t)(at + n1) * 4, (size_t)at * 4, c->stream)) ||
      (rc = F->qarena.grow((size_t)(F->qarena_used + I.qname_bytes + 16), (size_t)F->qarena_used, c->stream)))
    return rc;
  const uint64_t seq_bytes = I.seq_bytes + 64;
  auto room = [](uint64_t need) { return (size_t)(need + need / 4 + 4096); };
  if (F->s_tid.cap < n1 * 4 && ((rc = F->s_tid.reserve(room(n1 * 4))) || (rc = F->s_pos.reserve(room(n1 * 4))) || (rc = F->s_end.reserve(room(n1 * 4))) ||
                                (rc = F->s_seqoff.reserve(room(n1 * 4))) || (rc = F->s_lseq.reserve(room(n1 * 2))) || (rc = F->s_clipl.reserve(room(n1 * 2))) ||
                                (rc = F->s_clipr.reserve(room(n1 * 2))) || (rc = F->s_mapq.reserve(room(n1))) || (rc = F->s_cig.reserve(room(n1))) ||
                                (rc = F->tidflag.reserve(room(n1))) || (rc = F->s_meta.reserve(room(n1 * 16)))))
    return rc;
  if (F->s_seq4.cap < seq_bytes && (rc = F->s_seq4.reserve(room(seq_bytes)))) return rc;
  FrontParseOut o;
  o.tid = F->s_tid.as<int32_t>(); o.pos = F->s_pos.as<int32_t>(); o.end = F->s_end.as<int32_t>(); o.seq_off = F->s_seqoff.as<uint32_t>();
  o.l_seq = F->s_lseq.as<uint16_t>(); o.clip_l = F->s_clipl.as<uint16_t>(); o.clip_r = F->s_clipr.as<uint16_t>();
  o.mapq = F->s_mapq.as<uint8_t>(); o.cig = F->s_cig.as<uint8_t>(); o.seq4 = F->s_seq4.as<uint8_t>(); o.meta = F->s_meta.as<uint4>();
  o.rows = c->x_rows.as<strl_pair_rec>() + at; o.qhash = c->x_qhash.as<uint64_t>() + at; o.qref = F->qref.as<uint64_t>() + at;
  o.qarena = F->qarena.as<uint8_t>(); o.qarena_at = F->qarena_used; o.fragw = F->fragw.as<uint32_t>() + at; o.tidflag = F->tidflag.as<uint8_t>();
  // (the scan finished: the host waited for it.  The previous chunk's scorer may still read the chunk-temporary columns:
  // same stream, so the parse queues behind it.)
  if ((rc = front_parse(c, F, si, (uint32_t)n, o, c->stream))) return rc;
  F->qarena_used += I.qname_bytes;
  if (n) {
    strl_read_soa d{};
    d.n = n; d.tid = o.tid; d.pos = o.pos; d.end = o.end; d.seq_off = o.seq_off; d.l_seq = o.l_seq; d.clip_l = o.clip_l; d.clip_r = o.clip_r;
    d.mapq = o.mapq; d.cig = o.cig; d.seq4 = o.seq4; d.seq4_bytes = seq_bytes; d.max_l_seq = I.max_l_seq; d.mem = STRL_MEM_DEVICE;
    d.meta = reinterpret_cast<const strl_read_meta *>(o.meta);
    if ((rc = extract_add_scored(c, &d, at))) return rc;
  }
  STRL_HIP(hipEventRecord(S.ev_b, c->stream));
  S.b_pending = true;
  return STRL_OK;
}

int strl_front_begin(strl_ctx *c, int32_t n_ref, uint64_t first_record_offset, uint64_t n_reads_hint) {
  if (!c || n_ref < 0) { set_error("bad argument"); return STRL_ERR_ARG; }
  const uint64_t hint = std::max<uint64_t>(n_reads_hint, 1 << 20);
  // (STRL_ASYNC_ALLOC=1: the full-size buffers on a thread beside the first chunks, front.h.  Measured: 0.15 -> 0.03 s in front of
  // the loop at 1.3e8 reads, but the loop pays for it -- at 5.4e8 reads 2.60 s against 2.41 s with everything allocated up
  // front, wall 2.99 against 2.80 s (profiles/r05/full_size_shares.log): allocating tens of gigabytes beside running kernels
  // slows the launches down by more than it hides.  Off by default.)
  static const bool sync_alloc = getenv("STRL_ASYNC_ALLOC") == nullptr;
  const uint64_t small = (sync_alloc || hint <= (1ull << 25)) ? hint : std::max<uint64_t>(1ull << 24, hint / 8);
  // The per-read state of a whole file is tens of gigabytes in seven buffers.  hipMalloc returns at once for them on a settled
  // device (0.03 s for all of a whole genome's), but a large allocation made while the driver still reclaims what an earlier
  // process held stalls for ~0.48 s -- each one (profiles/r06/state_alloc_diag.log: 23 GB 0.483 s, 17.5 GB 0.483 s, 5.9 GB 0.121 s
  // in the first process on a box, one 0.483 s in the second, none from the third on).  Made side by side, the stalls overlap.
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamSynchronize(c->stream));
  strl::DevBuf pre_qref, pre_fragw, pre_qarena;
  // ... and the front end's four streams beside them: a stream is ~9.5 ms of the runtime's time here (profiles/r06/ctx_laps.log),
  // made one after the other behind the allocations they were most of this call
  hipStream_t pre_st[4] = {nullptr, nullptr, nullptr, nullptr};       // inflate 0, inflate 1, record scan, copies
  int pre_st_rc[4] = {0, 0, 0, 0};
  {
    struct Want { strl::DevBuf *b; size_t bytes; int rc; std::string err; };
    // ... and with them what the pair pass over the whole file takes at the END (strl_extract_finish -> strl_pair_device: join items,
    // emission keys, treads, the sort's scratch -- 4 GB for a genome).  Allocated there, behind the loop, they made that pass 0.05 -
    // 0.09 s in this round's earlier lines (0.016 s in round 5's); in place beforehand it is 0.016 - 0.017 s in six runs of six
    // (profiles/r06/pair_prealloc_full_size.log).  Sized as that call sizes them for `hint` reads, so that it finds them in place
    const uint64_t p_icap = std::max<uint64_t>(std::min<uint64_t>(hint / 8 + 65536, 3 * hint + 16), 1024), p_ecap = std::max<uint64_t>(std::min<uint64_t>(hint / 16 + 65536, 8 * hint + 16), 1024);
    int p_ebits = 3;
    while (p_ebits < 40 && ((2 * hint) >> (p_ebits - 2))) ++p_ebits;
    const size_t p_sb = std::max(radix_sort_scratch_bytes((uint32_t)p_icap, 32), radix_sort_scratch_bytes((uint32_t)p_ecap, p_ebits));
    const size_t p_max = (size_t)std::max(p_icap, p_ecap);
    Want want[] = {{&c->x_rows, (size_t)small * sizeof(strl_pair_rec), 0, {}}, {&pre_qarena, (size_t)small * 24, 0, {}}, {&c->x_qhash, (size_t)small * 8, 0, {}},
                   {&pre_qref, (size_t)small * 8, 0, {}}, {&c->x_whole, (size_t)small * 4, 0, {}}, {&pre_fragw, (size_t)small * 4, 0, {}},
                   {&c->x_soft, (size_t)(hint / 8 + 65536) * sizeof(strl_soft_rec), 0, {}},
                   {&c->p_key0, p_max * 8, 0, {}}, {&c->p_key1, p_max * 8, 0, {}}, {&c->p_val0, p_max * 4, 0, {}}, {&c->p_val1, p_max * 4, 0, {}},
                   {&c->p_emit, (size_t)p_ecap * sizeof(strl_tread), 0, {}}, {&c->treads, (size_t)p_ecap * sizeof(strl_tread) + 64, 0, {}}, {&c->sort_scratch, p_sb, 0, {}}};
    const int dev = c->device;
    int least = 0, greatest = 0;          // (numerically: least >= greatest; equal where the device has one level)
    STRL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));      // (in front of the threads: nothing may return between their start and their join)
    std::vector<std::thread> th;
    for (Want &w : want)
      th.emplace_back([&w, dev] {
        if (hipSetDevice(dev) != hipSuccess) { w.rc = STRL_ERR_HIP; w.err = "hipSetDevice"; return; }
        if (w.b->cap >= w.bytes && w.b->p) return;
        w.rc = w.b->reserve(w.bytes);
        if (w.rc) w.err = strl_last_error();
      });
    for (int k = 0; k < 4; ++k)
      th.emplace_back([&pre_st, &pre_st_rc, k, dev, least, greatest] {
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess) e = k == 3 ? hipStreamCreateWithFlags(&pre_st[k], hipStreamNonBlocking) : hipStreamCreateWithPriority(&pre_st[k], hipStreamNonBlocking, k == 2 ? greatest : least);
        pre_st_rc[k] = (int)e;
      });
    for (auto &t : th) t.join();
    auto drop_streams = [&] { for (hipStream_t &q : pre_st) if (q) { (void)hipStreamDestroy(q); q = nullptr; } };
    for (int k = 0; k < 4; ++k)
      if (pre_st_rc[k]) {
        set_error("hipStreamCreate: %s", hipGetErrorString((hipError_t)pre_st_rc[k]));
        drop_streams();
        pre_qref.release(); pre_fragw.release(); pre_qarena.release();
        return STRL_ERR_HIP;
      }
    for (Want &w : want)
      if (w.rc) {
        set_error("%s", w.err.c_str());
        drop_streams();
        pre_qref.release(); pre_fragw.release(); pre_qarena.release();
        return w.rc;
      }
  }
  int rc = extract_begin_sized(c, n_reads_hint, small);
  if (rc) { for (hipStream_t q : pre_st) if (q) (void)hipStreamDestroy(q); pre_qref.release(); pre_fragw.release(); pre_qarena.release(); return rc; }
  if (c->front) {
    for (hipStream_t q : c->front->st_i) if (q) (void)hipStreamSynchronize(q);
    if (c->front->st_a) (void)hipStreamSynchronize(c->front->st_a);
    strl::front_destroy(c->front); c->front = nullptr;
  }
  strl::strl_front *F = new strl::strl_front();
  c->front = F;
  c->x_front = true;
  F->qref = pre_qref; F->fragw = pre_fragw; F->qarena = pre_qarena;        // (allocated above, beside the others; the context owns them from here)
  F->n_ref = n_ref; F->first_off = first_record_offset;
  F->st_i[0] = pre_st[0]; F->st_i[1] = pre_st[1]; F->st_a = pre_st[2]; F->st_c = pre_st[3];   // (made above, beside the allocations)
  const unsigned host_waited = hipEventDisableTiming | (c->blocking_waits ? hipEventBlockingSync : 0u);   // ev_a, ev_b: what the feeding thread waits for
  for (strl::FrontSlot &S : F->slot) {
    STRL_HIP(hipEventCreateWithFlags(&S.ev_a, host_waited));
    STRL_HIP(hipEventCreateWithFlags(&S.ev_b, host_waited));
    STRL_HIP(hipEventCreateWithFlags(&S.ev_h2d, hipEventDisableTiming));
    STRL_HIP(hipEventCreateWithFlags(&S.ev_i, hipEventDisableTiming));
    STRL_HIP(hipEventCreateWithFlags(&S.ev_carry, hipEventDisableTiming));
    STRL_HIP(hipHostMalloc(reinterpret_cast<void **>(&S.h_info), 3 * sizeof(strl::FrontInfo), hipHostMallocDefault));
  }
  if ((rc = F->tid_seen.reserve((size_t)n_ref + 16))) return rc;
  STRL_HIP(hipMemsetAsync(F->tid_seen.p, 0, (size_t)n_ref + 16, c->stream));
  if ((rc = F->qref.grow((size_t)small * 8, 0, c->stream)) || (rc = F->fragw.grow((size_t)small * 4, 0, c->stream)) || (rc = F->qarena.grow((size_t)small * 24, 0, c->stream)))
    return rc;
  F->small_reads = small;
  if (small < hint) {        // the full-size buffers: allocated beside the first chunks (front.h, FrontBigAlloc)
    strl::FrontBigAlloc *B = new strl::FrontBigAlloc();
    F->big = B;
    const int dev = c->device;
    B->th = std::thread([B, dev, hint] {
      auto one = [&](strl::DevBuf &b, size_t bytes) {
        if (hipSetDevice(dev) != hipSuccess) return (int)STRL_ERR_HIP;
        return b.reserve(bytes);
      };
      // (side by side: the driver takes several allocations at once)
      int r[6] = {0, 0, 0, 0, 0, 0};
      std::thread t1([&] { r[0] = one(B->rows, (size_t)hint * sizeof(strl_pair_rec)); });
      std::thread t2([&] { r[1] = one(B->qarena, (size_t)hint * 24); });
      std::thread t3([&] { r[2] = one(B->qhash, (size_t)hint * 8); r[3] = one(B->qref, (size_t)hint * 8); });
      r[4] = one(B->whole, (size_t)hint * 4); r[5] = one(B->fragw, (size_t)hint * 4);
      t1.join(); t2.join(); t3.join();
      for (int x : r) if (x && !B->rc) { B->rc = x; B->err = strl_last_error(); }
      B->done.store(1, std::memory_order_release);
    });
  }
  STRL_HIP(hipStreamSynchronize(c->stream));
  return STRL_OK;
}

int strl_front_push(strl_ctx *c, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize, const uint32_t *crc32,
                    uint32_t n_blocks, strl_front_chunk *done, int *n_done) {
  return strl_front_push_after(c, nullptr, comp, comp_bytes, coff, clen, isize, crc32, n_blocks, done, n_done);
}

// the same when the chunks of ONE file go round-robin over several contexts (`strling extract --gpus N`): `prev` = the
// context the previous chunk of the file was pushed to (null / c itself: this context) -- the partial record in front of
// this chunk is taken from there
int strl_front_push_after(strl_ctx *c, strl_ctx *prev, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize,
                          const uint32_t *crc32, uint32_t n_blocks, strl_front_chunk *done, int *n_done) {
  const int rc = strl_front_enqueue_after(c, prev, comp, comp_bytes, coff, clen, isize, crc32, n_blocks, done, n_done);
  return rc ? rc : strl_front_collect(c);
}

// the two halves of a push, for a caller that has something to do between them (strl_front_stage of the chunk after this
// one): enqueue = this chunk's copy (unless staged) + inflate + record scan, and the summary of the chunk two back;
// collect = wait for the PREVIOUS chunk's record scan, enqueue its parse + scoring
int strl_front_collect(strl_ctx *c) {
  if (!c || !c->front) { set_error("strl_front_collect without strl_front_begin"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  strl::strl_front *F = c->front;
  int rc;
  while (F->b_issued + 1 < F->chunks) {
    if ((rc = front_stage_b(c, F, (int)(F->b_issued & 1)))) return rc;
    ++F->b_issued;
  }
  return STRL_OK;
}

int strl_ctx_blocking_waits(strl_ctx *c, int on) {
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  c->blocking_waits = on != 0;
  return STRL_OK;
}

int strl_front_trim_next(strl_ctx *c, uint32_t tail_bytes) {
  if (!c || !c->front || !c->x_open) { set_error("strl_front_trim_next without strl_front_begin"); return STRL_ERR_ARG; }
  if (tail_bytes > 65536u) { set_error("strl_front_trim_next: more than a BGZF block"); return STRL_ERR_ARG; }
  c->front->next_trim = tail_bytes;
  return STRL_OK;
}

// bytes behind the last complete record of the last chunk handed over (after strl_front_finish: every scan has been waited for)
int strl_front_tail_bytes(strl_ctx *c, uint32_t *tail_bytes) {
  if (!c || !c->front || !tail_bytes) { set_error("strl_front_tail_bytes: bad argument"); return STRL_ERR_ARG; }
  strl::strl_front *F = c->front;
  if (F->b_issued < F->chunks) { set_error("strl_front_tail_bytes before strl_front_finish"); return STRL_ERR_ARG; }
  *tail_bytes = F->last_slot < 0 ? 0u : F->slot[F->last_slot].h_info[0].carry_len;
  return STRL_OK;
}

int strl_front_reserve(strl_ctx *c, uint32_t max_blocks, uint64_t max_comp_bytes) {
  if (!c || !c->front || !max_blocks) { set_error("strl_front_reserve: bad argument / no strl_front_begin"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  // (the parse columns and the scorer's per-chunk queues are left to their first full chunk: reserving them here too cost
  // 40 - 60 ms of hipMalloc before the loop against 8 ms of one late inflate inside it)
  return strl::front_reserve(c, c->front, max_blocks, max_comp_bytes);
}

// starts the copy to the device of the chunk the NEXT strl_front_push / _enqueue_after of this context will hand over -- or, if that
// one is staged already, of the chunk after it
int strl_front_stage(strl_ctx *c, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize, const uint32_t *crc32,
                     uint32_t n_blocks) {
  if (!c || !c->front || !c->x_open || !n_blocks || !comp || !coff || !clen || !isize) { set_error("strl_front_stage: bad argument / no strl_front_begin"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  strl::strl_front *F = c->front;
  int si = (int)(F->chunks & 1);
  if (F->slot[si].staged) si ^= 1;          // the next push's chunk is staged: this is the one behind it (the caller stages in file order)
  if (F->slot[si].staged) { set_error("strl_front_stage: two chunks are staged already"); return STRL_ERR_ARG; }
  const strl::FrontChunkDesc d{comp, comp_bytes, coff, clen, isize, crc32, n_blocks};
  return strl::front_copy(c, F, si, d);
}

int strl_front_enqueue_after(strl_ctx *c, strl_ctx *prev, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize,
                             const uint32_t *crc32, uint32_t n_blocks, strl_front_chunk *done, int *n_done) {
  if (!c || !c->front || !c->x_open || (n_blocks && (!comp || !coff || !clen || !isize))) { set_error("strl_front_push: bad argument / no strl_front_begin"); return STRL_ERR_ARG; }
  if (prev == c) prev = nullptr;
  if (prev && (!prev->front || prev->front->last_slot < 0)) { set_error("strl_front_push_after: the previous context has no chunk"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  strl::strl_front *F = c->front;
  if (n_done) *n_done = 0;
  if (!n_blocks) return STRL_OK;
  const int si = (int)(F->chunks & 1);
  int rc;
  if ((rc = strl_front_collect(c))) return rc;        // (a caller that left it out: the slot's previous occupant must have been handed to the scorer)
  const bool reuse = F->slot[si].b_pending;          // the chunk before the previous one: its slot is reused now
  const strl::FrontChunkDesc d{comp, comp_bytes, coff, clen, isize, crc32, n_blocks};
  if (prev) {
    strl::FrontSlot &PS = prev->front->slot[prev->front->last_slot];
    const strl::FrontCarrySrc cs{PS.infl.as<uint8_t>(), PS.info.as<strl::FrontInfo>(), prev->front->last_end, prev->device, PS.ev_a, &PS.wait_read, &PS.read_pending};
    rc = strl::front_stage_a(c, F, si, d, false, &cs);
  } else {
    rc = strl::front_stage_a(c, F, si, d, F->chunks == 0 && !F->not_first);
  }
  if (rc) return rc;
  // The summary of the slot's previous occupant is waited for AFTER this chunk's work has been queued (the device waits for
  // that parse itself, ev_b).  The other order kept this chunk's copy to the device from being queued until the parse two
  // chunks back had finished -- it runs beside an inflate that leaves it few wave slots, up to 16 ms -- and every second
  // inflate started 7 ms late (profiles/r04/extract_timeline_before.txt).
  if (reuse) {
    if ((rc = front_fill_done(c, F->slot[si], done))) return rc;
    if (n_done) *n_done = 1;
  }
  ++F->chunks;
  F->comp_total += comp_bytes;
  F->infl_total += F->slot[si].infl_bytes;
  return STRL_OK;
}

int strl_front_finish(strl_ctx *c, strl_front_chunk done[2], int *n_done) {
  if (!c || !c->front) { set_error("strl_front_finish without strl_front_begin"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  strl::strl_front *F = c->front;
  if (n_done) *n_done = 0;
  int rc, k = 0;
  if (!F->chunks) return STRL_OK;
  const int last = (int)((F->chunks - 1) & 1);
  for (; F->b_issued < F->chunks; ++F->b_issued)
    if ((rc = front_stage_b(c, F, (int)(F->b_issued & 1)))) return rc;
  if (F->big && (rc = front_adopt_big(c, F, c->x_n))) return rc;        // (a file shorter than its hint: the thread is joined here at the latest)
  for (int si : {last ^ 1, last}) {
    if (!F->slot[si].b_pending) continue;
    if ((rc = front_fill_done(c, F->slot[si], done ? &done[k] : nullptr))) return rc;
    ++k;
  }
  if (n_done) *n_done = k;
  static const bool timing = getenv("STRL_FRONT_TIMING") != nullptr;
  if (timing && F->tev.size() >= 5) {       // per chunk: [0] start [1] copies queued [2] inflate done [3] ... scan done
    STRL_HIP(hipStreamSynchronize(F->st_a));
    for (hipStream_t q : F->st_i) STRL_HIP(hipStreamSynchronize(q));
    // inflate: the time at least one chunk's inflate was running (consecutive chunks' launches overlap); the other two: sums
    double h2d = 0, inf = 0, scan = 0, open_until = 0;
    for (size_t i = 0; i + 3 < F->tev.size(); i += 4) {
      float a = 0, b0 = 0, b1 = 0, d = 0;
      (void)hipEventElapsedTime(&a, F->tev[i], F->tev[i + 1]);
      (void)hipEventElapsedTime(&b0, F->tev[1], F->tev[i + 1]);
      (void)hipEventElapsedTime(&b1, F->tev[1], F->tev[i + 2]);
      (void)hipEventElapsedTime(&d, F->tev[i + 2], F->tev[i + 3]);
      h2d += a; scan += d;
      const double lo = std::max<double>(b0, open_until);
      if (b1 > lo) { inf += b1 - lo; open_until = b1; }
    }
    fprintf(stderr, "[strling] device front end, ms over %llu chunks: copies to the device %.1f  inflate %.1f  record scan %.1f  (%.1f MB compressed -> %.1f MB inflated)\n",
            (unsigned long long)F->chunks, h2d, inf, scan, (double)F->comp_total / 1e6, (double)F->infl_total / 1e6);
  }
  return STRL_OK;
}

int strl_front_fragwords(strl_ctx *c, uint64_t first, uint64_t n, uint32_t *out) {
  if (!c || !c->front || (n && !out) || first + n > c->x_n) { set_error("strl_front_fragwords: bad range"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  if (n) STRL_HIP(hipMemcpyAsync(out, c->front->fragw.as<uint32_t>() + first, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  return STRL_OK;
}

// the same without waiting: the copy is enqueued behind the parse of every chunk handed over so far (`out` page-locked);
// *done receives an event for strl_event_wait -- from any thread, so the fragment-length histogram of the first two million
// records can be made beside the rest of the file
int strl_front_fragwords_async(strl_ctx *c, uint64_t first, uint64_t n, uint32_t *out, void **done) {
  if (!c || !c->front || !done || (n && !out) || first + n > c->x_n) { set_error("strl_front_fragwords_async: bad range"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  hipEvent_t ev;
  STRL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (n) STRL_HIP(hipMemcpyAsync(out, c->front->fragw.as<uint32_t>() + first, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipEventRecord(ev, c->stream));
  *done = ev;
  return STRL_OK;
}
int strl_event_wait(void *event) {
  if (!event) return STRL_OK;
  hipEvent_t ev = static_cast<hipEvent_t>(event);
  STRL_HIP(hipEventSynchronize(ev));
  (void)hipEventDestroy(ev);
  return STRL_OK;
}
// the extraction the front end fed is given up (the context stays; the buffers stay allocated until the context goes or the next
// strl_front_begin): a caller that used the front end for a PREFIX of a file -- `strling call`'s fragment-length sample,
// call.nim:92 -- and goes on to other work on the context.  Everything the front end had in flight has completed on return.
int strl_front_end(strl_ctx *c) {
  if (!c) { set_error("null argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  if (c->front) {
    for (hipStream_t q : c->front->st_i) if (q) (void)hipStreamSynchronize(q);
    if (c->front->st_a) (void)hipStreamSynchronize(c->front->st_a);
    if (c->front->st_c) (void)hipStreamSynchronize(c->front->st_c);
  }
  STRL_HIP(hipStreamSynchronize(c->stream));
  { const int rcj = side_join(c); if (rcj) return rcj; }
  // (nothing is freed here: every hipFree synchronises the device -- 24 ms for the front end's twenty-odd buffers, in front of the
  // caller's next phase.  The buffers go with the context, or with the next strl_front_begin.)
  c->x_open = false; c->x_mode = false; c->x_n = 0;
  return STRL_OK;
}

int strl_front_records(strl_ctx *c, uint64_t *n) {
  if (!c || !n) { set_error("null argument"); return STRL_ERR_ARG; }
  *n = c->x_n;
  return STRL_OK;
}

int strl_front_tids(strl_ctx *c, uint8_t *seen, int32_t n_ref) {
  if (!c || !c->front || n_ref > c->front->n_ref || (n_ref && !seen)) { set_error("strl_front_tids: bad argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  if (n_ref) STRL_HIP(hipMemcpyAsync(seen, c->front->tid_seen.p, (size_t)n_ref, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  return STRL_OK;
}

int strl_front_qnames(strl_ctx *c, const int64_t *record_ids, uint64_t n, uint64_t *qname_off, char *names, uint64_t cap, uint64_t *need) {
  if (!c || !c->front || (n && (!record_ids || !qname_off))) { set_error("strl_front_qnames: bad argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  strl::strl_front *F = c->front;
  if (qname_off) qname_off[0] = 0;
  if (need) *need = 0;
  if (!n) return STRL_OK;
  if (n > 0x7fffffffull) { set_error("strl_front_qnames: too many names"); return STRL_ERR_ARG; }
  std::vector<uint32_t> ids((size_t)n);
  for (uint64_t i = 0; i < n; ++i) {
    if (record_ids[i] < 0 || (uint64_t)record_ids[i] >= c->x_n) { set_error("strl_front_qnames: record %lld out of range", (long long)record_ids[i]); return STRL_ERR_ARG; }
    ids[(size_t)i] = (uint32_t)record_ids[i];
  }
  strl::DevBuf d_ids, d_ref, d_off, d_out;
  int rc;
  if ((rc = d_ids.reserve((size_t)n * 4)) || (rc = d_ref.reserve((size_t)n * 8)) || (rc = d_off.reserve((size_t)n * 8))) return rc;
  STRL_HIP(hipMemcpyAsync(d_ids.p, ids.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  if ((rc = strl::front_gather_names(c, F, d_ids.as<uint32_t>(), (uint32_t)n, d_ref.as<uint64_t>(), c->stream))) return rc;
  std::vector<uint64_t> ref((size_t)n);
  STRL_HIP(hipMemcpyAsync(ref.data(), d_ref.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  uint64_t tot = 0;
  for (uint64_t i = 0; i < n; ++i) { qname_off[i] = tot; tot += ref[(size_t)i] & 255u; }
  qname_off[n] = tot;
  if (need) *need = tot;
  if (tot > cap || (tot && !names)) { d_ids.release(); d_ref.release(); d_off.release(); set_error("strl_front_qnames: %llu bytes of names, capacity %llu", (unsigned long long)tot, (unsigned long long)cap); return STRL_ERR_CAPACITY; }
  if (tot) {
    if ((rc = d_out.reserve((size_t)tot))) return rc;
    STRL_HIP(hipMemcpyAsync(d_off.p, qname_off, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = strl::front_copy_names(c, F, d_ref.as<uint64_t>(), d_off.as<uint64_t>(), (uint32_t)n, d_out.as<uint8_t>(), c->stream))) return rc;
    STRL_HIP(hipMemcpyAsync(names, d_out.p, (size_t)tot, hipMemcpyDeviceToHost, c->stream));
    STRL_HIP(hipStreamSynchronize(c->stream));
  }
  d_ids.release(); d_ref.release(); d_off.release(); d_out.release();
  return STRL_OK;
}

// Treads of the last extract in .bin order WITH their qnames, in one go: strl_treads_fetch + strl_front_qnames without the
// host round trips in between (references, exclusive scan of the lengths and byte copies are kernels behind the order sort).
// treads[cap] / qname_off[cap + 1] / names[names_cap] may be page-locked memory (then the copies need no staging).
// tread.qname_id stays the record index.  STRL_ERR_CAPACITY with *n_out / *names_need set when something does not fit.
int strl_front_treads_named(strl_ctx *c, strl_tread *treads, uint64_t cap, uint64_t *n_out, uint64_t *qname_off, char *names, uint64_t names_cap, uint64_t *names_need) {
  if (!c || !c->front || !n_out) { set_error("strl_front_treads_named: bad argument"); return STRL_ERR_ARG; }
  uint64_t nt = 0;
  int rc = strl_treads_fetch(c, nullptr, 0, &nt, nullptr);       // orders the treads, checks the error flags
  *n_out = nt;
  if (rc) return rc;
  if (names_need) *names_need = 0;
  if (!treads) return STRL_OK;
  if (nt > cap) { set_error("tread capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)nt); return STRL_ERR_CAPACITY; }
  if (qname_off) qname_off[0] = 0;
  if (!nt) return STRL_OK;
  STRL_HIP(hipSetDevice(c->device));
  // Work space: the pair pass's own scratch.  Once the treads are ordered (strl_treads_fetch above, on this stream) the join /
  // emission keys and values, and the unordered treads, are dead until the next pair pass rewrites them -- and an allocation of
  // its own is four hipMalloc + four hipFree, each of which waits for the device.  A buffer of its own only for what does not fit.
  strl::DevBuf own[5];
  const uint64_t ocap = std::max<uint64_t>(std::min<uint64_t>(names_cap, nt * 255), 16);
  auto room = [&](strl::DevBuf &scratch, strl::DevBuf &mine, size_t bytes, void **p) -> int {
    if (scratch.p && scratch.cap >= bytes) { *p = scratch.p; return STRL_OK; }
    const int r = mine.reserve(bytes);
    *p = mine.p;
    return r;
  };
  void *p_ref = nullptr, *p_len = nullptr, *p_off = nullptr, *p_out = nullptr, *p_tiles = nullptr;
  if ((rc = room(c->p_key0, own[0], (size_t)nt * 8, &p_ref)) || (rc = room(c->p_val0, own[1], (size_t)nt * 4, &p_len)) ||
      (rc = room(c->p_key1, own[2], (size_t)(nt + 1) * 8, &p_off)) || (rc = room(c->p_emit, own[3], (size_t)ocap, &p_out)) ||
      (rc = room(c->p_val1, own[4], strl::front_name_tiles((uint32_t)nt) * 8, &p_tiles)))
    return rc;
  hipStream_t st = c->stream;
  static const bool lap_on = getenv("STRL_FRONT_TIMING") != nullptr;
  const auto lap0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!lap_on) return;
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[strl_front_treads_named] %s at %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - lap0).count());
  };
  if ((rc = strl::front_tread_names(c, c->front, c->treads.as<strl_tread>(), c->n_treads_dev, (uint32_t)nt, static_cast<uint64_t *>(p_ref), static_cast<uint32_t *>(p_len),
                                    static_cast<uint64_t *>(p_off), static_cast<uint8_t *>(p_out), ocap, static_cast<uint64_t *>(p_tiles), st)))
    return rc;
  lap("references, offsets and name bytes on the device");
  STRL_HIP(hipMemcpyAsync(treads, c->treads.p, (size_t)nt * sizeof(strl_tread), hipMemcpyDeviceToHost, st));
  uint64_t total = 0;
  STRL_HIP(hipMemcpyAsync(&total, static_cast<uint64_t *>(p_off) + nt, 8, hipMemcpyDeviceToHost, st));
  if (qname_off) STRL_HIP(hipMemcpyAsync(qname_off, p_off, (size_t)(nt + 1) * 8, hipMemcpyDeviceToHost, st));
  STRL_HIP(hipStreamSynchronize(st));
  if (names_need) *names_need = total;
  int ret = STRL_OK;
  if (qname_off && names) {
    if (total > names_cap) { set_error("strl_front_treads_named: %llu bytes of names, capacity %llu", (unsigned long long)total, (unsigned long long)names_cap); ret = STRL_ERR_CAPACITY; }
    else if (total) STRL_HIP(hipMemcpy(names, p_out, (size_t)total, hipMemcpyDeviceToHost));
  }
  lap("treads, offsets and names on the host");
  for (strl::DevBuf &b : own) b.release();
  return ret;
}

// Page-locked host memory.  hipHostMalloc takes 0.25 s per GB here (4 KB pages faulted and pinned one by one: 0.36 s for the four
// chunk buffers of `strling extract`, longer than creating the device context beside it).  An anonymous mapping advised to use
// 2 MB pages, touched by a few threads and then registered takes 0.012 s for the same 1.3 GB, and copies from it run at 57 GB/s
// instead of 36 - 50 (tools/ubench/pin_probe.hip, profiles/r04/pin_probe.txt).  hipHostMalloc is the fallback.
namespace {
struct PinnedMap { void *base; size_t len; };
std::mutex g_pinned_mu;
std::vector<std::pair<void *, PinnedMap>> g_pinned;      // registered mappings by the pointer handed out
}  // namespace
void *strl_pinned_alloc(uint64_t bytes) {
  const size_t huge = (size_t)2 << 20;
  if (bytes >= huge && !getenv("STRL_PINNED_PLAIN")) {
    const size_t len = (((size_t)bytes + huge - 1) & ~(huge - 1)) + huge;
    void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base != MAP_FAILED) {
      char *a = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(base) + huge - 1) & ~(uintptr_t)(huge - 1));
      const size_t span = len - huge;
      (void)madvise(a, span, MADV_HUGEPAGE);
      const size_t T = std::min<size_t>(8, std::max<size_t>(1, span >> 26));      // a thread per 64 MB, up to 8
      std::vector<std::thread> th;
      for (size_t k = 0; k < T; ++k)
        th.emplace_back([=] { for (size_t o = span / T * k, e = k + 1 == T ? span : span / T * (k + 1); o < e; o += 4096) a[o] = 0; });
      for (auto &x : th) x.join();
      if (hipHostRegister(a, span, hipHostRegisterPortable) == hipSuccess) {      // (portable: contexts on every device of the process copy from it)
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned.push_back({a, PinnedMap{base, len}});
        return a;
      }
      (void)hipGetLastError();
      (void)munmap(base, len);
    }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void strl_pinned_free(void *p) {
  if (!p) return;
  PinnedMap m{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    for (size_t i = 0; i < g_pinned.size(); ++i)
      if (g_pinned[i].first == p) { m = g_pinned[i].second; g_pinned.erase(g_pinned.begin() + (long)i); break; }
  }
  if (m.base) { (void)hipHostUnregister(p); (void)munmap(m.base, m.len); }
  else (void)hipHostFree(p);
}

}  // extern "C"
