// score.hip -- HIP kernels + C-ABI entry points for the extract-side hot path (gfx950).
//
//  classify_kernel : one read per lane, streams the 13 B/read of coordinates + cigar class,
//                    evaluates the skip predicate of extract.nim:30-34 against the genome STR
//                    intervals (binary search + prefix-max of stops), writes the "skipped"
//                    result word or appends the read to the scoring queue (one atomic per wave).
//                    HBM-bound.
//  score_kernel<0> : one queued read per lane -> utils.get_repeat on the whole read
//                    (score_core.h), writes the packed result, queues the soft-clipped ends
//                    add_soft (extract.nim:93-106) would look at.  Integer-ALU/LDS bound.
//  score_kernel<1> : one queued soft-clipped end per lane, scored once, evaluated against both
//                    lowered thresholds (extract.nim:207-211 and :241-244).
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include "common.h"
#include "score_core.h"
#include "score_tables.h"

namespace strl {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap && p) return STRL_OK;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  size_t want = bytes + bytes / 8 + 256;
  STRL_HIP(hipMalloc(&p, want));
  cap = want;
  return STRL_OK;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
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

// Number of sub-queues.  Measured on MI355X (2^25 reads): 64 statically partitioned sub-queues made the scorer 4x
// SLOWER (5.07 vs 1.29 ms) because scored-read density is very uneven along the coordinate-sorted input (the
// unmapped tail is 100 % scored), while the LDS-staged appends already keep the single counter far below the
// L2 atomic limit (8192 atomics per launch).  One queue, blocks stride over it => balanced by construction.
#ifndef STRL_NQ
#define STRL_NQ 1
#endif
constexpr int NQ = STRL_NQ;
constexpr int CNT_STRIDE = 16;   // counters 64 B apart
constexpr int CNT_QUEUE = 0, CNT_SOFT = NQ * CNT_STRIDE, CNT_SKIP = 2 * NQ * CNT_STRIDE, CNT_WORDS = 2 * NQ * CNT_STRIDE + 16;

struct ScoreParams {
  uint64_t n;
  const int32_t *tid, *pos, *end;
  const uint32_t *seq_off;
  const uint16_t *l_seq, *clip_l, *clip_r;
  const uint8_t *mapq, *cig;
  const uint8_t *seq4;
  const TidInfo *g_tid;
  const int32_t *g_start, *g_pmax;
  const uint32_t *g_bins;
  int32_t n_tid;
  const uint16_t *lut, *thr;
  uint32_t *whole;
  uint32_t *queue, *soft_queue;  // NQ regions of qcap / scap entries
  uint32_t qcap, scap;
  uint32_t *counters;            // CNT_* layout
  strl_soft_rec *soft_out;
  uint32_t soft_cap;
  uint32_t min_mapq;
};

constexpr int LUT_DWORDS = LUT_ENTRIES / 2;
constexpr int CL_STAGE = 1024;   // queue entries a wave stages in LDS before one bulk append
constexpr int CL_ILP = 4;        // reads per lane per iteration (independent lookup chains in flight)

// extract.nim:30-34: single-M cigar, chromosome in the table, no interval overlapping [start, stop)
__device__ __forceinline__ bool skip_predicate(const ScoreParams &P, uint32_t cg, int32_t t, int32_t start, int32_t stop) {
  if (!(cg & STRL_CIG_SINGLE_M) || t < 0 || t >= P.n_tid) return false;
  const TidInfo ti = P.g_tid[t];
  if (!ti.has) return false;
  // lapper.find(start, stop) <=> any interval with iv.start < stop and iv.stop > start.
  // idx = number of intervals with iv.start < stop: bin directory, then a short scan inside the bin
  int32_t idx;
  const int32_t b = stop > 0 ? (stop >> BIN_SHIFT) : 0;
  if (b >= ti.n_bins) idx = ti.n_iv;
  else {
    const uint32_t *bins = P.g_bins + ti.bin_off + b;
    idx = (int32_t)bins[0];
    const int32_t hi = (int32_t)bins[1];
    while (idx < hi && P.g_start[ti.iv_off + idx] < stop) ++idx;
  }
  const bool overlap = idx > 0 && P.g_pmax[ti.iv_off + idx - 1] > start;
  return !overlap;
}

__global__ __launch_bounds__(256) void classify_kernel(ScoreParams P) {
  // Each wave owns one contiguous range of reads; queue entries are staged in LDS and appended to the
  // wave's sub-queue with one atomic per ~1000 entries (one same-address atomic per wave-iteration ran into
  // the ~88 ops/us limit of the L2 atomic unit: 12 ms per 2^25 reads).
  __shared__ uint32_t stage[4][CL_STAGE + 64 * CL_ILP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *buf = stage[wave];
  const uint64_t n_waves = (uint64_t)gridDim.x * 4u;
  const uint64_t gw = (uint64_t)blockIdx.x * 4u + wave;
  const uint64_t per = (((P.n + n_waves - 1) / n_waves) + 63ull) & ~63ull;
  const uint64_t r0 = gw * per;
  const uint64_t r1 = r0 + per < P.n ? r0 + per : P.n;
  const uint32_t q = (uint32_t)(gw % NQ);
  uint32_t *qbase = P.queue + (uint64_t)q * P.qcap;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t cnt = 0, nskip = 0;
  auto flush = [&]() {
    if (cnt) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(&P.counters[CNT_QUEUE + q * CNT_STRIDE], cnt);
      b = __shfl(b, 0);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < cnt; i += 64) qbase[b + i] = buf[i];
      __builtin_amdgcn_wave_barrier();
      cnt = 0;
    }
  };
  for (uint64_t base = r0; base < r1; base += 64 * CL_ILP) {
    bool need[CL_ILP], skipped[CL_ILP];
    uint32_t cg[CL_ILP];
    int32_t t[CL_ILP], st[CL_ILP], en[CL_ILP];
#pragma unroll
    for (int j = 0; j < CL_ILP; ++j) {
      const uint64_t r = base + 64 * j + lane;
      const bool in = r < r1;
      cg[j] = in ? P.cig[r] : 0u;
      t[j] = in ? P.tid[r] : -1;
      st[j] = in ? P.pos[r] : 0;
      en[j] = in ? P.end[r] : 0;
      need[j] = in;
    }
#pragma unroll
    for (int j = 0; j < CL_ILP; ++j) {
      skipped[j] = need[j] && skip_predicate(P, cg[j], t[j], st[j], en[j]);
      if (skipped[j]) { P.whole[base + 64 * j + lane] = STRL_RES_SKIPPED; need[j] = false; }
    }
#pragma unroll
    for (int j = 0; j < CL_ILP; ++j) {
      const unsigned long long m = __ballot(need[j]);
      nskip += (uint32_t)__popcll(__ballot(skipped[j]));
      if (need[j]) buf[cnt + __popcll(m & below)] = (uint32_t)(base + 64 * j + lane);
      cnt += (uint32_t)__popcll(m);
    }
    if (cnt >= CL_STAGE) flush();
  }
  flush();
  if (lane == 0 && nskip) atomicAdd(&P.counters[CNT_SKIP], nskip);
}

template <int NW, int SLOTS, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void score_kernel(ScoreParams P) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < LUT_DWORDS; i += BLOCK) lds[i] = reinterpret_cast<const uint32_t *>(P.lut)[i];
  __syncthreads();
  const uint16_t *lut = reinterpret_cast<const uint16_t *>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *wave_tab = lds + LUT_DWORDS + wave * (SLOTS * 64);
  uint32_t *col = wave_tab + lane;
  constexpr int MAXCH = (16 * NW + 62) / 32;
  // block b works on sub-queue b % NQ (gridDim.x is a multiple of NQ)
  const uint32_t sq = blockIdx.x % NQ, bq = blockIdx.x / NQ, nbq = gridDim.x / NQ;
  uint32_t n_items, out_base = 0;
  const uint32_t *q;
  if (MODE == 0) {
    n_items = P.counters[CNT_QUEUE + sq * CNT_STRIDE];
    q = P.queue + (uint64_t)sq * P.qcap;
  } else {
    n_items = min(P.counters[CNT_SOFT + sq * CNT_STRIDE], P.scap);
    q = P.soft_queue + (uint64_t)sq * P.scap;
    for (uint32_t i = 0; i < sq; ++i) out_base += min(P.counters[CNT_SOFT + i * CNT_STRIDE], P.scap);  // compact output
  }
  uint32_t *softq = P.soft_queue + (uint64_t)sq * P.scap;

  for (uint32_t base = bq * BLOCK + wave * 64; base < n_items; base += nbq * BLOCK) {  // wave-uniform
    const uint32_t item = base + lane;
    const bool act = item < n_items;
    uint32_t qv = 0, r = 0;
    int s0 = 0, len = 0;
    if (act) {
      qv = q[item];
      if (MODE == 0) {
        r = qv;
        len = P.l_seq[r];
      } else {
        r = qv >> 1;
        const int L = P.l_seq[r];
        len = (qv & 1u) ? P.clip_r[r] : P.clip_l[r];
        if (len > L) len = L;
        s0 = (qv & 1u) ? L - len : 0;
      }
      if (len > 16 * NW) len = 16 * NW;  // host picks NW from max_l_seq; never taken
    }
    // stage the BAM-packed bases that cover [s0, s0+len) into this lane's LDS column
    const int s0l = s0 & 31;
    const int nch = act ? (s0l + len + 31) >> 5 : 0;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.seq4 + (uint64_t)P.seq_off[r] * 16u) + (s0 >> 5);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
      if (c < nch) {
        const uint4 v = src[c];
        col[(4 * c + 0) * 64] = v.x;
        col[(4 * c + 1) * 64] = v.y;
        col[(4 * c + 2) * 64] = v.z;
        col[(4 * c + 3) * 64] = v.w;
      }
    }
    __builtin_amdgcn_wave_barrier();
    Seg<NW> sg;
    seg_from_raw<NW>(col, s0l, len, sg);
    uint32_t o0, o1;
    score_segment<NW, SLOTS>(sg, act, wave_tab, lane, lut, P.thr, MODE == 0 ? 1 : 2, MODE == 0 ? 1 : 3, o0, o1);

    if (MODE == 0) {
      bool pl = false, pr = false;
      if (act) {
        P.whole[r] = o0;
        // add_soft gates, extract.nim:97-106
        const uint32_t cg = P.cig[r];
        if (P.mapq[r] >= P.min_mapq && (cg & (STRL_CIG_FIRST_S | STRL_CIG_LAST_S))) {
          const bool has_unit = STRL_RES_K(o0) != 0;
          pl = (cg & STRL_CIG_FIRST_S) && (has_unit || P.clip_l[r] > 16);
          // with a single cigar op both loop iterations are cig_index == 0 (the host replays the duplicate)
          pr = (cg & STRL_CIG_LAST_S) && !(cg & STRL_CIG_ONE_OP) && (has_unit || P.clip_r[r] > 16);
        }
      }
      const unsigned long long ml = __ballot(pl), mr = __ballot(pr);
      if (ml | mr) {
        const int leader = __ffsll((unsigned long long)(ml | mr)) - 1;
        uint32_t b = 0;
        if (lane == leader) b = atomicAdd(&P.counters[CNT_SOFT + sq * CNT_STRIDE], (uint32_t)(__popcll(ml) + __popcll(mr)));
        b = __shfl(b, leader);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (pl) { const uint32_t s = b + __popcll(ml & below); if (s < P.scap) softq[s] = r << 1; }
        if (pr) { const uint32_t s = b + __popcll(ml) + __popcll(mr & below); if (s < P.scap) softq[s] = (r << 1) | 1u; }
      }
    } else {
      if (act && out_base + item < P.soft_cap) {
        strl_soft_rec o;
        o.read_side = qv;
        o.res_first = o0;
        o.res_after = o1;
        o.seg_len = (uint32_t)len;
        P.soft_out[out_base + item] = o;
      }
    }
  }
}

// ---- host side of this translation unit -------------------------------------------------------
template <int NW, int SLOTS, int MODE, int BLOCK> static int launch_score(strl_ctx *ctx, const ScoreParams &P, int blocks) {
  auto kfn = score_kernel<NW, SLOTS, MODE, BLOCK>;
  const size_t shmem = (size_t)LUT_DWORDS * 4 + (size_t)(BLOCK / 64) * SLOTS * 64 * 4;
  static bool attr_done = false;
  if (!attr_done) {
    STRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(blocks), dim3(BLOCK), shmem, ctx->stream, P);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}

template <int MODE> static int launch_score_class(strl_ctx *ctx, const ScoreParams &P, uint32_t max_l) {
  if (max_l <= 160) return launch_score<10, 64, MODE, 256>(ctx, P, 512);
  if (max_l <= 256) return launch_score<16, 128, MODE, 256>(ctx, P, 256);
  return launch_score<32, 256, MODE, 64>(ctx, P, 512);
}

}  // namespace strl

using namespace strl;

static constexpr uint64_t RING = 256;

extern "C" {

int strl_version(void) { return 100; }
const char *strl_last_error(void) { return strl::g_err; }

int strl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
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
  STRL_HIP(hipSetDevice(device_ordinal));
  strl_ctx *c = new strl_ctx();
  c->device = device_ordinal;
  STRL_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto &e : c->ev) STRL_HIP(hipEventCreate(&e));
  std::vector<uint16_t> lut;
  build_lut(lut);
  int rc = c->lut.reserve(lut.size() * 2);
  if (rc) return rc;
  STRL_HIP(hipMemcpy(c->lut.p, lut.data(), lut.size() * 2, hipMemcpyHostToDevice));
  rc = c->counters.reserve(CNT_WORDS * 4);
  if (rc) return rc;
  *out = c;
  return STRL_OK;
}

void strl_ctx_destroy(strl_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  strl::DevBuf *bufs[] = {&c->lut, &c->thr, &c->g_tid, &c->g_bins, &c->g_start, &c->g_pmax, &c->queue, &c->soft_queue, &c->counters,
                          &c->soft_tmp, &c->st_tid, &c->st_pos, &c->st_end, &c->st_seqoff, &c->st_lseq, &c->st_clipl, &c->st_clipr,
                          &c->st_mapq, &c->st_cig, &c->st_seq4, &c->st_whole, &c->st_soft};
  for (auto *b : bufs) b->release();
  for (auto &b : c->c_buf) b.release();
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->ring) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void *strl_ctx_stream(strl_ctx *c) { return c ? (void *)c->stream : nullptr; }
int strl_ctx_sync(strl_ctx *c) {
  if (!c) return STRL_ERR_ARG;
  STRL_HIP(hipStreamSynchronize(c->stream));
  return STRL_OK;
}
int strl_ctx_enable_timing(strl_ctx *c, int on) {
  if (!c) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  c->timing = on != 0;
  c->ring_pos = 0;
  if (c->timing && c->ring.empty()) {
    c->ring.resize(RING * 4);
    for (auto &e : c->ring) STRL_HIP(hipEventCreate(&e));
  }
  return STRL_OK;
}
int strl_ctx_kernel_times(strl_ctx *c, double ms_sum[3], uint64_t *n_launches) {
  if (!c || !ms_sum) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  STRL_HIP(hipStreamSynchronize(c->stream));
  ms_sum[0] = ms_sum[1] = ms_sum[2] = 0.0;
  const uint64_t n = std::min<uint64_t>(c->ring_pos, RING);
  for (uint64_t q = 0; q < n; ++q) {
    hipEvent_t *e = &c->ring[q * 4];
    for (int k = 0; k < 3; ++k) {
      float ms = 0.f;
      STRL_HIP(hipEventElapsedTime(&ms, e[k], e[k + 1]));
      ms_sum[k] += ms;
    }
  }
  if (n_launches) *n_launches = n;
  return STRL_OK;
}

int strl_ctx_set_opts(strl_ctx *c, const strl_opts *o) {
  if (!c || !o) { set_error("null argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  c->opts = *o;
  std::vector<uint16_t> thr;
  build_thr(*o, thr);
  int rc = c->thr.reserve(thr.size() * 2);
  if (rc) return rc;
  STRL_HIP(hipMemcpyAsync(c->thr.p, thr.data(), thr.size() * 2, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  c->have_opts = true;
  return STRL_OK;
}

int strl_ctx_set_genome(strl_ctx *c, const strl_genome_str *g) {
  if (!c) return STRL_ERR_ARG;
  STRL_HIP(hipSetDevice(c->device));
  if (!g || g->n_tid <= 0) { c->n_tid = 0; c->n_iv = 0; return STRL_OK; }
  const int32_t nt = g->n_tid;
  const int64_t niv = g->iv_off[nt];
  std::vector<int32_t> st((size_t)std::max<int64_t>(niv, 1)), pm((size_t)std::max<int64_t>(niv, 1));
  std::vector<TidInfo> ti((size_t)nt);
  std::vector<uint32_t> bins;
  std::vector<int64_t> idx;
  for (int32_t t = 0; t < nt; ++t) {
    const int64_t a = g->iv_off[t], b = g->iv_off[t + 1];
    if (b - a > 0x7ffffff0ll) { set_error("too many intervals on tid %d", t); return STRL_ERR_ARG; }
    idx.resize((size_t)(b - a));
    for (int64_t i = a; i < b; ++i) idx[(size_t)(i - a)] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) { return g->iv_start[x] < g->iv_start[y]; });
    int32_t run = INT32_MIN;
    for (int64_t i = a; i < b; ++i) {          // sorted starts + running maximum of stops
      const int64_t s = idx[(size_t)(i - a)];
      st[(size_t)i] = g->iv_start[s];
      run = std::max(run, g->iv_stop[s]);
      pm[(size_t)i] = run;
    }
    TidInfo &x = ti[(size_t)t];
    x.iv_off = a;
    x.n_iv = (int32_t)(b - a);
    x.has = g->has_chrom[t] ? 1 : 0;
    x.pad = 0;
    x.bin_off = (int64_t)bins.size();
    // bins[k] = number of intervals with start < (k << BIN_SHIFT); one extra entry closes the last bin
    const int32_t max_start = b > a ? std::max(0, st[(size_t)(b - 1)]) : 0;
    x.n_bins = b > a ? (max_start >> BIN_SHIFT) + 1 : 0;
    int64_t j = a;
    for (int32_t k = 0; k <= x.n_bins; ++k) {
      const int64_t lim = (int64_t)k << BIN_SHIFT;
      while (j < b && (int64_t)st[(size_t)j] < lim) ++j;
      bins.push_back((uint32_t)(j - a));
    }
  }
  if (bins.empty()) bins.push_back(0);
  int rc;
  if ((rc = c->g_tid.reserve(ti.size() * sizeof(TidInfo)))) return rc;
  if ((rc = c->g_bins.reserve(bins.size() * 4))) return rc;
  if ((rc = c->g_start.reserve(st.size() * 4))) return rc;
  if ((rc = c->g_pmax.reserve(pm.size() * 4))) return rc;
  STRL_HIP(hipMemcpy(c->g_tid.p, ti.data(), ti.size() * sizeof(TidInfo), hipMemcpyHostToDevice));
  STRL_HIP(hipMemcpy(c->g_bins.p, bins.data(), bins.size() * 4, hipMemcpyHostToDevice));
  STRL_HIP(hipMemcpy(c->g_start.p, st.data(), st.size() * 4, hipMemcpyHostToDevice));
  STRL_HIP(hipMemcpy(c->g_pmax.p, pm.data(), pm.size() * 4, hipMemcpyHostToDevice));
  c->n_tid = nt;
  c->n_iv = (uint64_t)niv;
  return STRL_OK;
}

static int score_device(strl_ctx *c, const strl_read_soa *s, uint32_t *whole, strl_soft_rec *soft, uint64_t soft_cap,
                        uint64_t *n_soft, strl_score_stats *stats, bool sync_counts) {
  const uint64_t n = s->n;
  if (n > 0x7fffffffull) { set_error("batch too large (%llu reads; limit 2^31-1)", (unsigned long long)n); return STRL_ERR_ARG; }
  if (s->max_l_seq > STRL_MAX_READ_LEN) { set_error("read of %u bases exceeds STRL_MAX_READ_LEN=%d", s->max_l_seq, STRL_MAX_READ_LEN); return STRL_ERR_ARG; }
  int rc;
  // classify grid: a multiple of 16 blocks (64 waves) so every sub-queue gets the same number of wave ranges
  int cblocks = (int)std::min<uint64_t>((n + 255) / 256, 2048);
  cblocks = std::max(16, (cblocks + 15) & ~15);
  const uint64_t n_waves = (uint64_t)cblocks * 4;
  const uint64_t per = (((n + n_waves - 1) / n_waves) + 63ull) & ~63ull;
  const uint64_t qcap = (n_waves / NQ) * per;                                  // reads routed to one sub-queue
  const uint64_t scap = std::min<uint64_t>((soft_cap + NQ - 1) / NQ, 2 * qcap); // soft items of one sub-queue
  if ((rc = c->queue.reserve((size_t)std::max<uint64_t>(NQ * qcap, 1) * 4))) return rc;
  if ((rc = c->soft_queue.reserve((size_t)std::max<uint64_t>(NQ * scap, 1) * 4))) return rc;
  STRL_HIP(hipMemsetAsync(c->counters.p, 0, CNT_WORDS * 4, c->stream));
  ScoreParams P{};
  P.n = n;
  P.tid = s->tid; P.pos = s->pos; P.end = s->end; P.seq_off = s->seq_off; P.l_seq = s->l_seq;
  P.clip_l = s->clip_l; P.clip_r = s->clip_r; P.mapq = s->mapq; P.cig = s->cig; P.seq4 = s->seq4;
  P.g_tid = c->g_tid.as<TidInfo>(); P.g_bins = c->g_bins.as<uint32_t>(); P.g_start = c->g_start.as<int32_t>();
  P.g_pmax = c->g_pmax.as<int32_t>(); P.n_tid = c->n_tid;
  P.lut = c->lut.as<uint16_t>(); P.thr = c->thr.as<uint16_t>();
  P.whole = whole; P.queue = c->queue.as<uint32_t>(); P.soft_queue = c->soft_queue.as<uint32_t>();
  P.qcap = (uint32_t)qcap; P.scap = (uint32_t)scap;
  P.counters = c->counters.as<uint32_t>(); P.soft_out = soft;
  P.soft_cap = (uint32_t)std::min<uint64_t>(soft_cap, 0xffffffffull);
  P.min_mapq = c->opts.min_mapq;
  hipEvent_t *tev = c->timing ? &c->ring[(c->ring_pos % RING) * 4] : nullptr;
  if (c->timing) ++c->ring_pos;
  if (tev) STRL_HIP(hipEventRecord(tev[0], c->stream));
  if (n) {
    hipLaunchKernelGGL(classify_kernel, dim3(cblocks), dim3(256), 0, c->stream, P);
    STRL_HIP(hipGetLastError());
  }
  if (tev) STRL_HIP(hipEventRecord(tev[1], c->stream));
  if (n) { if ((rc = launch_score_class<0>(c, P, s->max_l_seq))) return rc; }
  if (tev) STRL_HIP(hipEventRecord(tev[2], c->stream));
  if (n && scap) { if ((rc = launch_score_class<1>(c, P, s->max_l_seq))) return rc; }
  if (tev) STRL_HIP(hipEventRecord(tev[3], c->stream));
  if (sync_counts) {
    std::vector<uint32_t> raw(CNT_WORDS);
    STRL_HIP(hipMemcpyAsync(raw.data(), c->counters.p, CNT_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
    STRL_HIP(hipStreamSynchronize(c->stream));
    uint64_t scored = 0, softs = 0;
    for (int q = 0; q < NQ; ++q) {
      scored += raw[CNT_QUEUE + q * CNT_STRIDE];
      const uint32_t sc = raw[CNT_SOFT + q * CNT_STRIDE];
      if (sc > scap) { set_error("soft-clip queue overflow: sub-queue %d has %u items, capacity %llu (soft_cap %llu)", q, sc, (unsigned long long)scap, (unsigned long long)soft_cap); return STRL_ERR_CAPACITY; }
      softs += sc;
    }
    if (softs > soft_cap) { set_error("soft-clip output overflow: %llu items, capacity %llu", (unsigned long long)softs, (unsigned long long)soft_cap); return STRL_ERR_CAPACITY; }
    if (n_soft) *n_soft = softs;
    if (stats) {
      memset(stats, 0, sizeof *stats);
      stats->n_reads = n; stats->n_skipped = raw[CNT_SKIP]; stats->n_scored = scored; stats->n_soft_items = softs;
      if (tev) {
        (void)hipEventElapsedTime(&stats->ms_classify, tev[0], tev[1]);
        (void)hipEventElapsedTime(&stats->ms_score, tev[1], tev[2]);
        (void)hipEventElapsedTime(&stats->ms_soft, tev[2], tev[3]);
      }
    }
  }
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
  strl_read_soa d = *s;
  int rc;
  struct { strl::DevBuf *b; const void *src; size_t bytes; const void **dst; } cp[] = {
      {&c->st_tid, s->tid, (size_t)n * 4, (const void **)&d.tid},         {&c->st_pos, s->pos, (size_t)n * 4, (const void **)&d.pos},
      {&c->st_end, s->end, (size_t)n * 4, (const void **)&d.end},         {&c->st_seqoff, s->seq_off, (size_t)n * 4, (const void **)&d.seq_off},
      {&c->st_lseq, s->l_seq, (size_t)n * 2, (const void **)&d.l_seq},    {&c->st_clipl, s->clip_l, (size_t)n * 2, (const void **)&d.clip_l},
      {&c->st_clipr, s->clip_r, (size_t)n * 2, (const void **)&d.clip_r}, {&c->st_mapq, s->mapq, (size_t)n, (const void **)&d.mapq},
      {&c->st_cig, s->cig, (size_t)n, (const void **)&d.cig},             {&c->st_seq4, s->seq4, (size_t)s->seq4_bytes, (const void **)&d.seq4}};
  for (auto &x : cp) {
    if ((rc = x.b->reserve(std::max<size_t>(x.bytes, 64)))) return rc;
    if (x.bytes) STRL_HIP(hipMemcpyAsync(x.b->p, x.src, x.bytes, hipMemcpyHostToDevice, c->stream));
    *x.dst = x.b->p;
  }
  d.mem = STRL_MEM_DEVICE;
  if ((rc = c->st_whole.reserve((size_t)std::max<uint64_t>(n, 1) * 4))) return rc;
  const uint64_t dcap = 4 * n + 65536;   // >= NQ * (worst case of one sub-queue): no overflow possible in host mode
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

}  // extern "C"
