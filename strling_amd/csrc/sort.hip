// sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs, hand-written for gfx950 (wave64).
//
// Replaces the reference's per-group `algorithm.sort` (call.nim:127-130, merge.nim:132-135: stable merge sort by
// position inside a Table[(tid, repeat)]) with ONE keyed sort of the composite key, and serves the qname-hash join of
// the pair logic.  The sizes on this path are 10^5..10^6 keys per sample (5x10^7 for a 50-sample merge), where a
// sort is bound by launch boundaries and dependent-load latency, not by HBM bandwidth, so the design minimises
// launches:
//
//   * 8-bit digits, one `scatter_kernel` launch per digit.  A tile is 2048 consecutive keys (4 waves x 8 rounds of 64).
//   * In-tile stable ranking without sorting: per round every lane finds the lanes of its wave that hold the same digit
//     with 8 ballots ("match"), the lowest such lane bumps the wave's private LDS counter once for the whole peer
//     group, ranks follow from popcounts.  Waves own disjoint LDS rows, so no cross-wave atomics.
//   * No scan kernel: the digit histograms are kept per tile, H[tile][256], and per chunk of 32 tiles, C[chunk][256]
//     (1 KiB rows).  A tile obtains "keys with my digit in earlier tiles" by summing the chunk rows before its chunk
//     and the tile rows of its own chunk before it, and the digit bases from the totals of all chunk rows -- each wave
//     reads whole rows with one 16-byte load per lane, 8 loads in flight: (tiles / 32 + 31) rows at most.
//     The histogram kernel of a pass writes its tile row and adds it to its chunk row (256 uncontended atomics per
//     tile).  Building the next pass' tables with per-key atomics while scattering was tried: skewed digits put
//     thousands of atomics on one address (~88 per microsecond on the L2): 0.23 ms per pass instead of 0.02.
//   * The element count is read from device memory, so sorts can be enqueued behind the kernels that produce their
//     input without a host round trip.
//
// HBM traffic per pass: 12 B read + 12 B written per pair (+ the 1 KiB histogram rows); at 5x10^5 pairs that is 12 MB
// per pass -- microseconds at HBM speed; what is left is the launch boundary and the L2 latency of the row sums.
#include "sort.h"
#include "device_util.h"

namespace strl {

namespace {

struct SortPass {
  const uint32_t *d_n;
  uint32_t n_max;
  const uint64_t *kin;
  const uint32_t *vin;
  uint64_t *kout;
  uint32_t *vout;
  uint32_t *H;    // [tiles][256] digit histogram of every tile of the INPUT order of this pass
  uint32_t *C;    // [chunks][256] the same per chunk of SORT_CHUNK tiles
  int shift;
  uint32_t mask;
};

__device__ __forceinline__ uint32_t active_n(const SortPass &a) {
  const uint32_t n = *a.d_n;
  return n < a.n_max ? n : a.n_max;
}

// tile + chunk histograms of the input order of a pass
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(SortPass a) {
  __shared__ uint32_t bins[256];
  const uint32_t n = active_n(a);
  const uint32_t t = blockIdx.x;
  if ((uint64_t)t * SORT_TILE >= n) return;
  bins[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t r = 0; r < SORT_KPT; ++r) {
    const uint32_t e = t * SORT_TILE + r * SORT_THREADS + threadIdx.x;
    if (e < n) atomicAdd(&bins[(uint32_t)(a.kin[e] >> a.shift) & a.mask], 1u);
  }
  __syncthreads();
  const uint32_t c = bins[threadIdx.x];
  a.H[(uint64_t)t * 256 + threadIdx.x] = c;
  if (c) __hip_atomic_fetch_add(&a.C[(uint64_t)(t / SORT_CHUNK) * 256 + threadIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of one value per thread over the 256 threads of a block (digit bases)
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t *wsum /* [4] shared */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  __syncthreads();
  return base + inc - v;
}

// column sums of the 1 KiB rows [lo, hi) of a table: wave q takes rows lo + q, lo + q + 4, ...; lane l the digits
// 4l .. 4l+3.  `below` only accumulates rows < cut.
__device__ __forceinline__ void row_sums(const uint32_t *T, uint32_t lo, uint32_t hi, uint32_t cut, uint4 &tot, uint4 &below) {
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const uint4 *R = reinterpret_cast<const uint4 *>(T);
  auto acc = [&](const uint4 &h, uint32_t row) {
    tot.x += h.x; tot.y += h.y; tot.z += h.z; tot.w += h.w;
    if (row < cut) { below.x += h.x; below.y += h.y; below.z += h.z; below.w += h.w; }
  };
  uint32_t r = lo + (uint32_t)q;
  for (; r + 28 < hi; r += 32) {   // 8 independent 16-byte loads in flight per lane
    uint4 h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = R[(uint64_t)(r + 4 * u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc(h[u], r + 4 * u);
  }
  for (; r < hi; r += 4) acc(R[(uint64_t)r * 64 + lane], r);
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(SortPass a) {
  __shared__ uint32_t cnt[4][256];      // per-wave digit counters, later the first output slot of (wave, digit)
  __shared__ uint4 pb[4][64], pt[4][64];
  __shared__ uint32_t wsum[4];
  const uint32_t n = active_n(a);
  const uint32_t t = blockIdx.x;
  if ((uint64_t)t * SORT_TILE >= n) return;
  const uint32_t ntiles = (n + SORT_TILE - 1) / SORT_TILE, nchunks = (ntiles + SORT_CHUNK - 1) / SORT_CHUNK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;

  // ---- keys of this tile: wave w owns elements [w * 512, (w + 1) * 512) of the tile, 64 per round ----
  uint64_t key[SORT_KPT];
  uint32_t val[SORT_KPT];
  const uint32_t e0 = t * SORT_TILE + (uint32_t)wave * (SORT_KPT * 64u) + (uint32_t)lane;
#pragma unroll
  for (uint32_t r = 0; r < SORT_KPT; ++r) {
    const uint32_t e = e0 + r * 64u;
    key[r] = e < n ? a.kin[e] : ~0ull;
    val[r] = e < n ? a.vin[e] : 0u;
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;

  // ---- where digit d of this tile starts in the output: digit base + same digit in earlier tiles.
  // Two levels, both summed here: the chunk table (every chunk: totals; chunks before mine: below) and the tile rows
  // of my own chunk before me.
  {
    const uint32_t c = t / SORT_CHUNK;
    uint4 tot = make_uint4(0, 0, 0, 0), below = make_uint4(0, 0, 0, 0), dummy = make_uint4(0, 0, 0, 0);
    row_sums(a.C, 0u, nchunks, c, tot, below);
    row_sums(a.H, c * SORT_CHUNK, t, t, dummy, below);
    pb[wave][lane] = below;
    pt[wave][lane] = tot;
  }
  __syncthreads();   // also orders the zeroing of cnt before the ranking below
  uint32_t off;
  {
    const uint32_t d = threadIdx.x;
    const uint32_t *b = reinterpret_cast<const uint32_t *>(pb), *tt = reinterpret_cast<const uint32_t *>(pt);
    uint32_t below = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { below += b[w * 256 + d]; tot += tt[w * 256 + d]; }
    off = block_excl_scan256(tot, wsum) + below;
  }

  // ---- stable rank inside the wave: match lanes with the same digit, one LDS add per (round, digit) ----
  uint32_t rank[SORT_KPT];
#pragma unroll
  for (uint32_t r = 0; r < SORT_KPT; ++r) {
    const bool valid = e0 + r * 64u < n;
    const uint32_t d = (uint32_t)(key[r] >> a.shift) & a.mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    uint32_t old = 0;
    const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
    if (valid && lane == leader) old = atomicAdd(&cnt[wave][d], (uint32_t)__popcll(peers));
    old = __shfl(old, leader);
    rank[r] = old + (uint32_t)__popcll(peers & lt);
  }
  __syncthreads();
  {
    const uint32_t d = threadIdx.x;
    uint32_t run = off;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t c = cnt[w][d]; cnt[w][d] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t r = 0; r < SORT_KPT; ++r) {
    if (e0 + r * 64u < n) {
      const uint32_t d = (uint32_t)(key[r] >> a.shift) & a.mask;
      const uint32_t p = cnt[wave][d] + rank[r];
      a.kout[p] = key[r];
      a.vout[p] = val[r];
    }
  }
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace

size_t radix_sort_scratch_bytes(uint32_t n_max, int bits) {
  const uint32_t P = (uint32_t)(bits + 7) / 8;
  const uint32_t ntiles = div_up(n_max ? n_max : 1, SORT_TILE), nchunks = div_up(ntiles, SORT_CHUNK);
  return (size_t)(P ? P : 1) * ((size_t)ntiles + nchunks) * 1024 + 256;
}

void radix_sort_tables(void *scratch, uint32_t n_max, int bits, void **p, size_t *bytes) {
  *p = nullptr; *bytes = 0;
  if (bits <= 0 || n_max == 0 || !scratch) return;
  const uint32_t P = (uint32_t)(bits + 7) / 8;
  const uint32_t ntiles = div_up(n_max, SORT_TILE), nchunks = div_up(ntiles, SORT_CHUNK);
  *p = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~(uintptr_t)255);
  *bytes = (size_t)P * nchunks * 256 * 4;
}

int radix_sort_pairs(hipStream_t st, const uint32_t *d_n, uint32_t n_max, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt,
                     uint32_t *vals_alt, void *scratch, size_t scratch_bytes, int bit_lo, int bits, uint64_t **out_keys,
                     uint32_t **out_vals, bool tables_zeroed) {
  if (out_keys) *out_keys = keys;
  if (out_vals) *out_vals = vals;
  if (bits <= 0 || n_max == 0) return 0;
  if (scratch_bytes < radix_sort_scratch_bytes(n_max, bits)) return (int)hipErrorInvalidValue;
  const uint32_t P = (uint32_t)(bits + 7) / 8;
  const uint32_t ntiles = div_up(n_max, SORT_TILE), nchunks = div_up(ntiles, SORT_CHUNK);
  // scratch: chunk tables of all passes, then tile tables of all passes
  uint32_t *C = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~(uintptr_t)255);
  const size_t cwords = (size_t)nchunks * 256, hwords = (size_t)ntiles * 256;
  uint32_t *H = C + (size_t)P * cwords;
  hipError_t e;
  // the chunk tables are accumulated with (256 per tile, uncontended) atomics by the histogram kernel
  if (!tables_zeroed && (e = zero_words(C, (size_t)P * cwords * 4, st)) != hipSuccess) return (int)e;
  uint64_t *kin = keys, *kout = keys_alt;
  uint32_t *vin = vals, *vout = vals_alt;
  for (uint32_t p = 0; p < P; ++p) {
    SortPass a{};
    a.d_n = d_n; a.n_max = n_max; a.kin = kin; a.vin = vin; a.kout = kout; a.vout = vout;
    a.H = H + (size_t)p * hwords;
    a.C = C + (size_t)p * cwords;
    a.shift = bit_lo + 8 * (int)p;
    const int rem = bits - 8 * (int)p;
    a.mask = rem >= 8 ? 0xffu : ((1u << rem) - 1u);
    hipLaunchKernelGGL(sort_hist_kernel, dim3(ntiles), dim3(SORT_THREADS), 0, st, a);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(ntiles), dim3(SORT_THREADS), 0, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    uint64_t *tk = kin; kin = kout; kout = tk;
    uint32_t *tv = vin; vin = vout; vout = tv;
  }
  if (out_keys) *out_keys = kin;
  if (out_vals) *out_vals = vin;
  return 0;
}

}  // namespace strl

// ---- C ABI: the sort on host arrays (tests compare it with a stable host sort; see include/strling_amd.h) ----
#include "common.h"

extern "C" int strl_sort_pairs(strl_ctx *c, uint64_t *keys, uint32_t *vals, uint64_t n, uint64_t n_max, int bit_lo, int bits) {
  using namespace strl;
  if (!c || (!keys && n) || (!vals && n)) { set_error("null argument"); return STRL_ERR_ARG; }
  if (n_max < n) n_max = n;
  if (n_max > 0x7fffffffull || bit_lo < 0 || bits < 0 || bit_lo + bits > 64) { set_error("bad sort arguments"); return STRL_ERR_ARG; }
  if (n == 0) return STRL_OK;
  STRL_HIP(hipSetDevice(c->device));
  DevBuf k0, k1, v0, v1, sc, dn;
  const size_t sb = radix_sort_scratch_bytes((uint32_t)n_max, bits);
  int rc;
  if ((rc = k0.reserve(n_max * 8)) || (rc = k1.reserve(n_max * 8)) || (rc = v0.reserve(n_max * 4)) || (rc = v1.reserve(n_max * 4)) ||
      (rc = sc.reserve(sb)) || (rc = dn.reserve(256)))
    return rc;
  const uint32_t n32 = (uint32_t)n;
  STRL_HIP(hipMemcpyAsync(k0.p, keys, n * 8, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipMemcpyAsync(v0.p, vals, n * 4, hipMemcpyHostToDevice, c->stream));
  STRL_HIP(hipMemcpyAsync(dn.p, &n32, 4, hipMemcpyHostToDevice, c->stream));
  uint64_t *ok = nullptr;
  uint32_t *ov = nullptr;
  const int e = radix_sort_pairs(c->stream, dn.as<uint32_t>(), (uint32_t)n_max, k0.as<uint64_t>(), v0.as<uint32_t>(), k1.as<uint64_t>(),
                                 v1.as<uint32_t>(), sc.p, sb, bit_lo, bits, &ok, &ov);
  if (e) { set_error("radix_sort_pairs failed: %s", hipGetErrorString((hipError_t)e)); return STRL_ERR_HIP; }
  STRL_HIP(hipMemcpyAsync(keys, ok, n * 8, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipMemcpyAsync(vals, ov, n * 4, hipMemcpyDeviceToHost, c->stream));
  STRL_HIP(hipStreamSynchronize(c->stream));
  k0.release(); k1.release(); v0.release(); v1.release(); sc.release(); dn.release();
  return STRL_OK;
}
