// score_core.h -- per-lane repeat-unit scorer for one read segment (CDNA4 / gfx950).
//
// What it computes: utils.get_repeat (src/strpkg/utils.nim:236-271) of the reference, i.e. for
// k = 2..6 the most frequent minimum-rotation k-mer over NON-overlapping windows
// (slide_by utils.nim:10-34, count :205-211 with the running-argmax tie rule of inc :192-195),
// the greedy literal recount (strutils.count, utils.nim:254), the score/threshold ladder
// (:250-263) and reduce_repeat (:220-233, :271).  It is written from the algorithm for a
// 64-wide wavefront with ONE READ PER LANE:
//   * the read lives in registers as a 2-bit stream (kmer "CATG" code, base i at bits 2i..2i+1
//     of word i/16) plus a same-layout invalid-base stream,
//   * window -> canonical (min-rotation) code is one LDS table lookup,
//   * per-lane uint8 histograms (k <= 4) / open-addressing tables (k = 5,6) live in LDS in a
//     [row][lane] layout, so every lane hits its own bank: no conflicts, no atomics between lanes,
//   * the literal recount is bit-parallel (XOR against the replicated unit, 16 bases per op),
//   * the k loop is wave-uniform (lanes that `break` just go idle), so table clears are
//     cooperative 16-byte stores.
// The same source compiles for the host (STRL_EMU, one "lane", LANES = 1) purely so that the
// CPU-only test-suite can exercise the device logic; the product never runs that build.
#pragma once
#include <stdint.h>

#ifdef STRL_EMU
#define STRL_DEV inline
#define STRL_HD inline
#define STRL_LANES 1
static inline uint32_t strl_funnel_r(uint32_t lo, uint32_t hi, uint32_t s) {
  return s == 0 ? lo : (uint32_t)(((((uint64_t)hi) << 32) | lo) >> s);
}
static inline int strl_ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int strl_popc(uint32_t x) { return __builtin_popcount(x); }
static inline bool strl_any(bool p) { return p; }
static inline int strl_wave_min(int v) { return v; }
static inline uint32_t strl_max3(uint32_t a, uint32_t b, uint32_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
static inline uint32_t strl_bfe(uint32_t x, uint32_t off, uint32_t w) { return (x >> off) & ((1u << w) - 1u); }
static inline uint32_t strl_lds_add(uint32_t *a, uint32_t v) { uint32_t o = *a; *a = o + v; return o; }
#else
#include <hip/hip_runtime.h>
#define STRL_DEV __device__ __forceinline__
#define STRL_HD __host__ __device__ __forceinline__
#define STRL_LANES 64
STRL_DEV uint32_t strl_funnel_r(uint32_t lo, uint32_t hi, uint32_t s) { return __funnelshift_r(lo, hi, s); }
STRL_DEV int strl_ffs(uint32_t x) { return __ffs((int)x); }
STRL_DEV int strl_popc(uint32_t x) { return __popc(x); }
STRL_DEV bool strl_any(bool p) { return __any(p) != 0; }
STRL_DEV int strl_wave_min(int v) {   // wave-uniform minimum (butterfly; once per k pass)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d); v = o < v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
STRL_DEV uint32_t strl_max3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }   // v_max3_u32
STRL_DEV uint32_t strl_bfe(uint32_t x, uint32_t off, uint32_t w) { return __builtin_amdgcn_ubfe(x, off, w); }
STRL_DEV uint32_t strl_lds_add(uint32_t *a, uint32_t v) { return atomicAdd(a, v); }  // lane-private: ds_add_rtn_u32
#endif

namespace strl {

// Optional per-phase cycle accounting (debug builds with -DSTRL_PHASE_TIMING only; see tools/phase_timing.py)
#if defined(STRL_PHASE_TIMING) && !defined(STRL_EMU)
__device__ unsigned long long g_phase[32];
#define STRL_PH(st, i)                                                                   \
  do {                                                                                   \
    const unsigned long long now__ = __builtin_readcyclecounter();                       \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_phase[i], now__ - (st).ph_t);              \
    (st).ph_t = now__;                                                                   \
  } while (0)
#else
#define STRL_PH(st, i) do { } while (0)
#endif

// canonical-code lookup tables, indexed by the window value as it comes out of the LSB-first
// 2-bit stream (first base in the LOW bits).  For k = 5, 6 an entry is the reference's code (first base HIGH).  For
// k <= 4 it is the dense CLASS ID of that code -- its rank among the 10 / 24 / 70 minimum-rotation codes -- so that a
// k pass needs 10 / 24 / 18 rows of LDS bins instead of 16 / 64 / 64 (the histogram rows are what limits how many
// waves fit a CU); LUT_C<k>[id] turns the winning id back into the code once per pass.
constexpr int LUT_OFF2 = 0, LUT_OFF3 = 16, LUT_OFF4 = 80, LUT_C2 = 336, LUT_C3 = 346, LUT_C4 = 370, LUT_OFF5 = 440, LUT_OFF6 = 1464,
              LUT_ENTRIES = 5560;
template <int K> struct LutCls { static constexpr int off = K == 2 ? LUT_C2 : K == 3 ? LUT_C3 : LUT_C4, n = K == 2 ? 10 : K == 3 ? 24 : 70; };
template <int K> struct LutOff;
template <> struct LutOff<2> { static constexpr int v = LUT_OFF2; };
template <> struct LutOff<3> { static constexpr int v = LUT_OFF3; };
template <> struct LutOff<4> { static constexpr int v = LUT_OFF4; };
template <> struct LutOff<5> { static constexpr int v = LUT_OFF5; };
template <> struct LutOff<6> { static constexpr int v = LUT_OFF6; };

// threshold tables (host-computed with the reference's float64 expressions, so the device does
// no floating point at all): thr[row][L] = five bytes, byte k-2 = the value for k; rows: 0 = int(L*0.12/k) (utils.nim:251),
// 1 = int(L*p/k), 2 = int(L*(p-0.07)/k), 3 = int(L*min(p,0.6)/k)  (utils.nim:259, extract.nim:208,242)
constexpr int THR_LMAX = 512;

// ---- BAM 4-bit -> 2-bit -----------------------------------------------------------------------
// squeeze the two low bits of each nibble of z into 16 contiguous bits, in base order
// (BAM stores base 2m in the HIGH nibble of byte m).
STRL_HD uint32_t squeeze8(uint32_t z) {
  z = ((z >> 4) & 0x03030303u) | ((z & 0x03030303u) << 2);
  z = (z | (z >> 4)) & 0x00FF00FFu;
  z = (z | (z >> 8)) & 0x0000FFFFu;
  return z;
}
// 8 BAM-packed bases -> 8 code pairs (A=1 C=0 G=3 T=2, anything else 1 like the kmer module's
// lookup) and 8 flag pairs (bit 2j: base j is not ACGT, bit 2j+1: base j is 'N').
STRL_HD void conv8(uint32_t x, uint32_t &pairs, uint32_t &flags) {
  uint32_t b0 = x & 0x11111111u, b1 = (x >> 1) & 0x11111111u, b2 = (x >> 2) & 0x11111111u, b3 = (x >> 3) & 0x11111111u;
  uint32_t sum = b0 + b1 + b2 + b3;  // per-nibble popcount
  uint32_t t = sum ^ 0x11111111u;
  uint32_t inv = (t | (t >> 1) | (t >> 2)) & 0x11111111u;  // popcount != 1
  uint32_t l = (b0 | b2) | inv;
  uint32_t h = (b3 | b2) & ~inv;
  pairs = squeeze8(l | (h << 1));
  flags = 0;
  if (inv) {
    uint32_t isn = (sum >> 2) & 0x11111111u;  // popcount == 4 <=> 'N'
    flags = squeeze8(inv | (isn << 1));
  }
}

// Same conversion through a 256-entry table indexed by one BAM byte (two bases): entry bits 0-3 = the two 2-bit
// codes, bits 16-19 = their flag pairs.  Four lookups + three shift-ors replace ~35 bit-trick ops per dword, and
// the integer VALU is the binding resource of the scorer.  (conv8 above stays as the table's specification.)
STRL_DEV void conv8_lut(uint32_t x, const uint32_t *clut, uint32_t &pairs, uint32_t &flags) {
  const uint32_t r = clut[x & 0xffu] | (clut[(x >> 8) & 0xffu] << 4) | (clut[(x >> 16) & 0xffu] << 8) | (clut[x >> 24] << 12);
  pairs = r & 0xffffu;
  flags = r >> 16;
}

// One segment in registers.
template <int NW> struct Seg {
  uint32_t seq[NW];  // 2-bit codes, 16 bases per word, zero beyond len
  uint32_t inv[NW];  // bit 2i set: base i is not ACGT (never matches a literal unit)
  int len;
  int n_N;
  bool has_inv;
};

// Build a Seg from raw BAM-packed dwords staged in this lane's LDS column (`raw[i * STRL_LANES]`
// is dword i of the staged chunk), starting at base `s0` (< 32) of the chunk.
template <int NW> STRL_DEV void seg_from_raw(const uint32_t *raw, const uint32_t *clut, int s0, int len, Seg<NW> &sg) {
  const int dw0 = s0 >> 3, sh = s0 & 7;
  uint32_t p, f;
  conv8_lut(raw[dw0 * STRL_LANES], clut, p, f);
  uint32_t cur = p >> (2 * sh), curf = f >> (2 * sh);
  const int fill = 16 - 2 * sh;
  uint32_t any_f = 0;
  int nn = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t sw = 0, fw = 0;
    if (16 * w < len) {
      uint32_t pa, fa, pb, fb;
      conv8_lut(raw[(dw0 + 1 + 2 * w) * STRL_LANES], clut, pa, fa);
      conv8_lut(raw[(dw0 + 2 + 2 * w) * STRL_LANES], clut, pb, fb);
      uint64_t buf = (uint64_t)cur | ((uint64_t)pa << fill) | ((uint64_t)pb << (fill + 16));
      uint64_t bf = (uint64_t)curf | ((uint64_t)fa << fill) | ((uint64_t)fb << (fill + 16));
      sw = (uint32_t)buf;
      fw = (uint32_t)bf;
      cur = (uint32_t)(buf >> 32);
      curf = (uint32_t)(bf >> 32);
      const int nv = len - 16 * w;
      if (nv < 16) {
        const uint32_t m = (1u << (2 * nv)) - 1u;
        sw &= m;
        fw &= m;
      }
    }
    sg.seq[w] = sw;
    sg.inv[w] = fw & 0x55555555u;
    any_f |= fw;
    nn += strl_popc(fw & 0xAAAAAAAAu);
  }
  sg.len = len;
  sg.n_N = nn;
  sg.has_inv = (any_f & 0x55555555u) != 0;
}

// ---- per-k histogram pass (utils.nim:205-211) ---------------------------------------------------
// tab: this lane's LDS column (row stride STRL_LANES dwords), already zeroed for this k.
// Returns count (A[imax]) and the winning code (4^K-1 when there is no window: decode of -1).
// Window i of a k pass sits at the fixed bit offset 2*k*i of the stream, so everything indexes registers statically.
template <int K, int NW> STRL_DEV uint32_t window_code(const Seg<NW> &sg, const uint16_t *lk, int i) {
  constexpr uint32_t MASK = (1u << (2 * K)) - 1u;
  const int bit = 2 * K * i, w = bit >> 5, sh = bit & 31;
  const uint32_t lo = sg.seq[w], hi = (w + 1 < NW) ? sg.seq[w + 1] : 0u;
  const uint32_t v = ((sh + 2 * K <= 32) ? (lo >> sh) : strl_funnel_r(lo, hi, sh)) & MASK;
  return lk[v];
}

// The count of a k pass is the running count inc() (utils.nim:192-195) would have seen at each window:
//   newc_i = 1 + #{j < i : code_j == code_i};  count = max_i newc_i;  winner = code at the first i reaching it.
// Four interchangeable ways to get newc_i, picked per k by what is cheapest on CDNA4:
//   k = 2,3 : 32-bit bins in LDS ([bin][lane], ds_add_rtn returns the old count), 16 / 64 rows
//   k = 4   : 256 uint8 bins packed 4 per dword in LDS, 64 rows.  (Counting k = 4 in registers with byte-parallel
//             equality was tried: 2.6x the VALU instructions of the LDS bins, and integer VALU is the binding
//             resource of this kernel -- 4 cycles per wave64 op, ~90 % busy -- so it lost.)
//   k = 5,6 : in registers, one code per VGPR (<= 32 windows for reads <= 160 bases); longer reads fall back to a
//             per-lane open-addressing table in LDS.
template <int K, int NW, int SLOTS>
STRL_DEV void hist_pass(const Seg<NW> &sg, bool active, uint32_t *tab, const uint16_t *lut, int &cmax, uint32_t &imax) {
  constexpr uint32_t MASK = (1u << (2 * K)) - 1u;
  constexpr int NWIN = NW * 16 / K;
  constexpr int B = 8;  // windows per batch: 8 LUT reads, then 8 table updates in flight at once (LDS latency)
  const uint16_t *lk = lut + LutOff<K>::v;
  const int nwin = active ? sg.len / K : 0;
  cmax = 0;
  imax = MASK;
  if (K >= 5 && NW <= 10) {
    uint32_t code[NWIN];
#pragma unroll
    for (int i = 0; i < NWIN; ++i) code[i] = window_code<K, NW>(sg, lk, i);
#pragma unroll
    for (int i = 0; i < NWIN; ++i) {
      uint32_t ne = 0;  // earlier windows that differ (pure VALU: xor, min, add -- no compare-to-SGPR round trips)
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const uint32_t x = code[j] ^ code[i];
        ne += x < 1u ? x : 1u;
      }
      const int newc = 1 + i - (int)ne;
      if (i < nwin && newc > cmax) { cmax = newc; imax = code[i]; }
    }
    return;
  }
  if (K <= 4) {
    // Running argmax without compare/select chains: inside a batch of 8 windows key = newc << 15 | (7 - j) << 12 | code and
    // one unsigned max picks the largest running count and, among equals, the earliest window; across batches an earlier
    // batch keeps the lead unless a later one has a strictly larger count -- together the code inc() (utils.nim:192-195)
    // would have kept.  (A single key with the global window index needs ~170 distinct constants, which the compiler
    // parks in VGPRs for the whole kernel.)  Batches that lie below the shortest live segment of the wave (`umin`,
    // wave-uniform) need no per-window bounds masks at all: idle lanes count into their own bins, their result is ignored.
    static_assert(B == 8, "three tie-break bits");
    const int umin = strl_wave_min(active ? nwin : 0x7fffffff);
    uint32_t best = 0;
#pragma unroll
    for (int b0 = 0; b0 < NWIN; b0 += B) {
      if (!strl_any(b0 < nwin)) break;  // wave-uniform early exit for short segments
      uint32_t code[B], key[B], kc[B];
#pragma unroll
      for (int j = 0; j < B; ++j) code[j] = (b0 + j < NWIN) ? window_code<K, NW>(sg, lk, b0 + j) : 0u;
#pragma unroll
      for (int j = 0; j < B; ++j) kc[j] = (1u << 15) | ((7u - (uint32_t)j) << 12);
      const bool full = b0 + B <= umin && b0 + B <= NWIN;      // wave-uniform
      if (K <= 3) {  // 10 / 24 x 32-bit bins (one per class); windows past a lane's end hit a dummy row and get key 0
        constexpr uint32_t DUMMY = (uint32_t)LutCls<K>::n;
        uint32_t old[B];
        // The bins count in units of 1 << 15, so the value the LDS atomic returns already sits in the key's count field
        // and the key is ONE three-operand add (count field + tie-break constant + class: no overlapping bits).
        if (full) {
#pragma unroll
          for (int j = 0; j < B; ++j) old[j] = strl_lds_add(tab + code[j] * STRL_LANES, 1u << 15);
#pragma unroll
          for (int j = 0; j < B; ++j) key[j] = old[j] + kc[j] + code[j];
        } else {
#pragma unroll
          for (int j = 0; j < B; ++j) {
            old[j] = 0;
            if (b0 + j < NWIN) old[j] = strl_lds_add(tab + ((b0 + j < nwin) ? code[j] : DUMMY) * STRL_LANES, 1u << 15);
          }
#pragma unroll
          for (int j = 0; j < B; ++j) {
            const uint32_t k = old[j] + kc[j] + code[j];
            key[j] = (b0 + j < NWIN && b0 + j < nwin) ? k : 0u;
          }
        }
      } else {  // K == 4: 70 uint8 bins packed 4 per dword (18 rows + dummy); a lane's updates to one bin stay ordered
        uint32_t old[B], sh[B];
#pragma unroll
        for (int j = 0; j < B; ++j) sh[j] = (code[j] & 3u) * 8u;
        if (full) {
#pragma unroll
          for (int j = 0; j < B; ++j) old[j] = strl_lds_add(tab + (code[j] >> 2) * STRL_LANES, 1u << sh[j]);
#pragma unroll
          for (int j = 0; j < B; ++j) key[j] = ((strl_bfe(old[j], sh[j], 8) << 15) + kc[j]) | code[j];
        } else {
#pragma unroll
          for (int j = 0; j < B; ++j) {
            old[j] = 0;
            if (b0 + j < NWIN) old[j] = strl_lds_add(tab + ((b0 + j < nwin) ? (code[j] >> 2) : 18u) * STRL_LANES, 1u << sh[j]);
          }
#pragma unroll
          for (int j = 0; j < B; ++j) {
            const uint32_t k = ((strl_bfe(old[j], sh[j], 8) << 15) + kc[j]) | code[j];
            key[j] = (b0 + j < NWIN && b0 + j < nwin) ? k : 0u;
          }
        }
      }
      uint32_t bb = key[0] > key[1] ? key[0] : key[1];
#pragma unroll
      for (int j = 2; j + 1 < B; j += 2) bb = strl_max3(bb, key[j], key[j + 1]);
      if (bb > (best | 0x7fffu)) best = bb;      // strictly larger count only
    }
    if (best) { cmax = (int)(best >> 15); imax = lut[LutCls<K>::off + (best & 0xfffu)]; }   // no window at all: count 0, code "all ones" (utils.nim:197-198)
    return;
  }
#pragma unroll
  for (int b0 = 0; b0 < NWIN; b0 += B) {
    if (!strl_any(b0 < nwin)) break;  // wave-uniform early exit for short segments
    uint32_t code[B];
#pragma unroll
    for (int j = 0; j < B; ++j) code[j] = (b0 + j < NWIN) ? window_code<K, NW>(sg, lk, b0 + j) : 0u;
    {  // k = 5, 6 on long reads: open addressing, entry = (code+1) << 8 | count
#pragma unroll
      for (int j = 0; j < B; ++j) {
        if (b0 + j < NWIN && b0 + j < nwin) {
          const uint32_t c = code[j];
          uint32_t hsh = ((c * 0x9E3779B1u) >> 16) & (uint32_t)(SLOTS - 1);
          int newc;
          for (;;) {
            const uint32_t e = tab[hsh * STRL_LANES];
            if (e == 0) { tab[hsh * STRL_LANES] = ((c + 1u) << 8) | 1u; newc = 1; break; }
            if ((e >> 8) == c + 1u) { tab[hsh * STRL_LANES] = e + 1u; newc = (int)(e & 0xffu) + 1; break; }
            hsh = (hsh + 1u) & (uint32_t)(SLOTS - 1);
          }
          if (newc > cmax) { cmax = newc; imax = c; }
        }
      }
    }
  }
}

// ---- greedy non-overlapping literal count of the decoded unit (strutils.count, utils.nim:254) ----
template <int K, int NW> STRL_DEV int recount(const Seg<NW> &sg, uint32_t code) {
  uint32_t acc[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) acc[w] = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t pat = ((code >> (2 * (K - 1 - j))) & 3u) * 0x55555555u;  // unit base j replicated
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t lo = sg.seq[w], hi = (w + 1 < NW) ? sg.seq[w + 1] : 0u;
      acc[w] |= strl_funnel_r(lo, hi, 2 * j) ^ pat;
    }
  }
  if (sg.has_inv) {  // rare: a non-ACGT base inside the window kills the match
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t lo = sg.inv[w], hi = (w + 1 < NW) ? sg.inv[w + 1] : 0u;
        acc[w] |= strl_funnel_r(lo, hi, 2 * j);
      }
    }
  }
  const int limit = sg.len - K + 1;  // number of start positions
  uint32_t m[NW];                    // bit 2i of m[w] set <=> the unit matches at base 16w + i
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t x = ~(acc[w] | (acc[w] >> 1)) & 0x55555555u;
    const int nv = limit - 16 * w;
    if (nv < 16) x &= (nv <= 0) ? 0u : ((1u << (2 * nv)) - 1u);
    m[w] = x;
  }
  int cnt = 0;
  if (K == 2) {
    // Two matches collide only when adjacent (homodimer unit), and greedy left-to-right then keeps every other
    // match of a run.  Runs are resolved with one multi-word addition: adding a run's start bit to the run
    // (both bits of every pair filled) carries through exactly that run, which tells every pair the parity of
    // its run's start.  Exact for hetero-dimers too (their runs have length 1).
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t prev = w ? m[w - 1] : 0u;
      const uint32_t start = m[w] & ~((m[w] << 2) | (prev >> 30));   // first match of a run
      const uint32_t fill = m[w] | (m[w] << 1);                        // runs as solid bit strings
      const uint64_t t = (uint64_t)fill + (start & 0x11111111u) + carry;  // only runs starting on an even pair
      carry = (uint32_t)(t >> 32);
      const uint32_t even_runs = fill & ~(uint32_t)t;                  // bits those runs lost to the carry
      const uint32_t sel = (m[w] & even_runs & 0x11111111u) | (m[w] & ~even_runs & 0x44444444u);
      cnt += strl_popc(sel);
    }
  } else {
    // does the unit overlap a shifted copy of itself (s[0..K-d) == s[d..K))?  Only then can matches collide.
    bool border = false;
#pragma unroll
    for (int d = 1; d < K; ++d) border |= (code >> (2 * d)) == (code & ((1u << (2 * (K - d))) - 1u));
    if (!strl_any(border)) {
#pragma unroll
      for (int w = 0; w < NW; ++w) cnt += strl_popc(m[w]);
    } else {  // rare (periodic units whose shorter period did not already win): literal greedy walk
      int skip = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        uint32_t x = m[w] & ~((1u << (2 * skip)) - 1u);
        skip = 0;
        while (x) {
          const int i = (strl_ffs(x) - 1) >> 1;
          ++cnt;
          const int nx = i + K;
          if (nx >= 16) { skip = nx - 16; x = 0; }
          else x &= ~((1u << (2 * nx)) - 1u);
        }
      }
    }
  }
  return cnt;
}

// cooperative zeroing of `rows` rows ([row][lane] dwords) of this wave's table region
STRL_DEV void clear_rows(uint32_t *wave_tab, int lane, int rows) {
#ifdef STRL_EMU
  for (int i = 0; i < rows; ++i) wave_tab[i] = 0;
  (void)lane;
#else
  uint4 z = make_uint4(0, 0, 0, 0);
  uint4 *p = reinterpret_cast<uint4 *>(wave_tab);
  const int n16 = rows * (STRL_LANES / 4);  // 16-byte units
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < n16; i += STRL_LANES) p[i] = z;
  __builtin_amdgcn_wave_barrier();
#endif
}

struct ScoreState {
  int best;
  bool alive;
  uint32_t res0, res1;  // packed results for the two threshold rows
  unsigned long long ph_t;  // phase clock (debug builds)
};

// rows of the wave's table region a k pass touches (direct bins + the dummy row, or the hash slots)
template <int K, int SLOTS> struct KRows { static constexpr int v = (K == 2) ? 11 : (K == 3) ? 25 : (K == 4) ? 19 : SLOTS; };

// The lane's thresholds for its segment length.  The host packs the five k-values of one (row, L) into one
// 64-bit word (one byte each; L <= 510 and p <= 1 keep them <= 255), so an item needs THREE loads, issued with
// the rest of its global traffic.  (Loading thresholds at the decision points -- which is also where the
// compiler sinks scalarised loads to -- put a global round trip on the critical path of every k: the
// "decide" phases were 38 % of all wave cycles.)
struct LaneThr {
  uint64_t w12, w0, w1;
};
STRL_DEV void load_thr(const uint64_t *thr, int row0, int row1, int L, LaneThr &t) {
  t.w12 = thr[0 * THR_LMAX + L];
  t.w0 = thr[row0 * THR_LMAX + L];
  t.w1 = (row1 == row0) ? t.w0 : thr[row1 * THR_LMAX + L];
}
template <int K> STRL_DEV int thr_get(uint64_t w) { return (int)((w >> (8 * (K - 2))) & 0xffu); }

template <int K, int NW, int SLOTS>
STRL_DEV void score_k(const Seg<NW> &sg, ScoreState &st, uint32_t *wave_tab, int lane, const uint16_t *lut, const LaneThr &t) {
  if (!strl_any(st.alive)) return;  // wave-uniform
  if (K <= 4 || NW > 10) clear_rows(wave_tab, lane, KRows<K, SLOTS>::v);   // k = 5,6 of short reads count in registers
  int c;
  uint32_t code;
  hist_pass<K, NW, SLOTS>(sg, st.alive, wave_tab + lane, lut, c, code);
  STRL_PH(st, 2 + 2 * (K - 2));
  if (st.alive) {
    int score = c * K;
    if (score <= st.best) {  // utils.nim:250-253
      if (c < thr_get<K>(t.w12)) st.alive = false;  // break
    } else {
      c = recount<K, NW>(sg, code);  // utils.nim:254
      score = c * K;
      if (score >= st.best) {  // :256
        st.best = score;
        const uint32_t packed = code | ((uint32_t)K << 12) | ((uint32_t)c << 16);
        if (c > thr_get<K>(t.w0)) st.res0 = packed;  // :259-263
        if (c > thr_get<K>(t.w1)) st.res1 = packed;
      }
    }
  }
  STRL_PH(st, 3 + 2 * (K - 2));
}

// reduce_repeat (utils.nim:220-233) + the final multiply (:271) on a packed result
STRL_DEV uint32_t reduce_packed(uint32_t r) {
  const uint32_t k = (r >> 12) & 7u;
  if (k == 0) return 0;
  const uint32_t code = r & 0xfffu, b = code & 3u;
  const uint32_t rep = b * (((1u << (2 * k)) - 1u) / 3u);
  if (code != rep) return r;
  return b | (1u << 12) | (((r >> 16) * k) << 16);
}

// The ladder is cut in two so that the expensive, rarely reached k = 4..6 passes run on a dense
// set of survivors instead of dragging every wave through them for one or two live lanes:
//   stage A: N filter (utils.nim:238), k = 2, 3, 4.   st.alive afterwards <=> the loop reaches k = 5
//            (~12 % of scored 150 bp reads in the S1 mix; 60 % would survive a cut after k = 3).
//   stage B: k = 5, 6 from a carried (best, res0, res1).
template <int NW, int SLOTS>
STRL_DEV void score_stage_a(const Seg<NW> &sg, bool active, uint32_t *wave_tab, int lane, const uint16_t *lut, const LaneThr &t,
                            ScoreState &st) {
  st.best = -1;
  st.alive = active && sg.n_N <= 20;
  st.res0 = st.res1 = 0;
  score_k<2, NW, SLOTS>(sg, st, wave_tab, lane, lut, t);
  score_k<3, NW, SLOTS>(sg, st, wave_tab, lane, lut, t);
  score_k<4, NW, SLOTS>(sg, st, wave_tab, lane, lut, t);
}
template <int NW, int SLOTS>
STRL_DEV void score_stage_b(const Seg<NW> &sg, uint32_t *wave_tab, int lane, const uint16_t *lut, const LaneThr &t, ScoreState &st) {
  score_k<5, NW, SLOTS>(sg, st, wave_tab, lane, lut, t);
  score_k<6, NW, SLOTS>(sg, st, wave_tab, lane, lut, t);
}

}  // namespace strl
