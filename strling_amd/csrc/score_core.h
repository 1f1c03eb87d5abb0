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
static inline int strl_wave_max(int v) { return v; }
static inline int strl_rank(bool) { return 0; }
static inline uint32_t strl_thread_at() { return 0; }
#else
#include <hip/hip_runtime.h>
#define STRL_DEV __device__ __forceinline__
#define STRL_HD __host__ __device__ __forceinline__
#define STRL_LANES 64
STRL_DEV uint32_t strl_funnel_r(uint32_t lo, uint32_t hi, uint32_t s) { return __funnelshift_r(lo, hi, s); }
STRL_DEV int strl_ffs(uint32_t x) { return __ffs((int)x); }
STRL_DEV int strl_popc(uint32_t x) { return __popc(x); }
STRL_DEV bool strl_any(bool p) { return __any(p) != 0; }
// Wave-uniform minimum / maximum, in converged code (every lane of the wave active).  Six DPP-modified min / max instructions
// (pairs, quads, mirrored halves of 8 and 16 lanes, then lane 15 / 31 broadcast into the next rows) leave the result in lane 63;
// the __shfl_xor butterfly is six ds_bpermute round trips through the LDS, each waited for.
template <bool MIN> STRL_DEV int strl_wave_red(int v) {
  auto op = [](int a, int b) { return MIN ? (a < b ? a : b) : (a > b ? a : b); };
  const int idn = MIN ? 0x7fffffff : (int)0x80000000;
  v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
  v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x140, 0xf, 0xf, false));   // row_mirror
  v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
STRL_DEV int strl_wave_min(int v) { return strl_wave_red<true>(v); }
STRL_DEV int strl_rank(bool p) {   // number of lower lanes with p
  return __popcll(__ballot(p) & ((1ull << (threadIdx.x & 63)) - 1ull));
}
STRL_DEV uint32_t strl_thread_at() { return blockIdx.x * blockDim.x + threadIdx.x; }
STRL_DEV int strl_wave_max(int v) { return strl_wave_red<false>(v); }
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

// Stage A (k = 2, 3, 4) reads its own tables, shaped so that a window costs three vector instructions besides the two LDS
// operations (u32 words; ROWB = bytes between two rows of a lane's bin column):
//   TA_K2 [256]  one byte of the 2-bit stream = TWO k = 2 windows -> ROWB * class of the first | ROWB * class of the second << 16
//   TA_K4 [256]  one byte = one k = 4 window -> ROWB * (class >> 2) | 8 * (class & 3) << 16   (uint8 bins, four per dword)
//   TA_K3 [16]   64 x uint8: 6-bit window -> class
//   TA_C  [52]   104 x uint16: class -> the reference's code, k = 2 | 3 | 4 at 0 | 10 | 34
constexpr int TA_K2 = 0, TA_K4 = 256, TA_K3 = 512, TA_C = 528, TA_WORDS = 580;
constexpr uint32_t ROWB = 4u * STRL_LANES;
template <int K> struct TaCls { static constexpr int off = K == 2 ? 0 : K == 3 ? 10 : 34; };

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

// The table below is indexed by a SWIZZLED byte, b ^ ((b >> 3) & 0x1f) (a bijection on bytes; four bytes of a dword at once:
// x ^ ((x >> 3) & 0x1f1f1f1f)): the sixteen byte values two ACGT bases can take -- 0x11 ... 0x88, nearly every lookup -- fall
// into banks 1, 2, 4, 8 three times over when the byte itself is the index; swizzled they hit sixteen different banks.
STRL_HD uint32_t conv_swizzle(uint32_t x) { return x ^ ((x >> 3) & 0x1f1f1f1fu); }

// Same conversion through a 256-entry table indexed by one BAM byte (two bases): entry bits 0-3 = the two 2-bit
// codes, bits 16-19 = their flag pairs.  Four lookups + three shift-ors replace ~35 bit-trick ops per dword, and
// the integer VALU is the binding resource of the scorer.  (conv8 above stays as the table's specification.)
STRL_DEV void conv8_lut(uint32_t x, const uint32_t *clut, uint32_t &pairs, uint32_t &flags) {
  x = conv_swizzle(x);
  const uint32_t r = clut[x & 0xffu] | (clut[(x >> 8) & 0xffu] << 4) | (clut[(x >> 16) & 0xffu] << 8) | (clut[x >> 24] << 12);
  pairs = r & 0xffffu;
  flags = r >> 16;
}

// One segment in registers.
template <int NW> struct Seg {
  uint32_t seq[NW];  // 2-bit codes, 16 bases per word, zero beyond len
  // inv word w, bit 2i set: base 16 w + i is not ACGT (never matches a literal unit).  Written and read only by the LANES
  // that hold such a base (has_inv; a per cent of the reads, but every other WAVE has one), so it does not live in NW
  // registers -- dead weight for every other lane, and registers are what limits the waves per SIMD here (96 instead of
  // 106: five waves instead of four).  The first INV_SLOTS such lanes of a wave keep theirs in a small LDS area of the wave,
  // any further ones in a per-thread column of a global spill array.
  uint32_t *inv_lds;      // the wave's slots, [INV_SLOTS][NW]                                   (wave-uniform)
  uint32_t *inv;          // global spill, column of thread t at inv[t + w * inv_stride]           (wave-uniform)
  uint32_t inv_stride;
  int inv_nslots;         // LDS slots of the wave (<= INV_SLOTS)                                  (wave-uniform)
  int inv_slot;           // -1: every base of the segment is one of ACGT
  int len;
  int n_N;
  STRL_DEV bool has_inv() const { return inv_slot >= 0; }
};

constexpr int INV_SLOTS = 16;   // (the benchmark's reads carry a sequencer "N" in one of seven: nine lanes of a wave on average)
template <int NW> STRL_DEV int inv_store(Seg<NW> &sg, bool flagged, const uint32_t (&fl)[NW]) {   // returns the number of 'N' bases
  const int rank = strl_rank(flagged);
  sg.inv_slot = flagged ? rank : -1;
#if defined(STRL_PHASE_TIMING) && !defined(STRL_EMU)
  {
    const unsigned long long bm = __ballot(flagged);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&g_phase[20], (unsigned long long)__popcll(bm)); atomicAdd(&g_phase[21], bm ? 1ull : 0ull); atomicAdd(&g_phase[22], 1ull); }
  }
#endif
  int nn = 0;
  if (flagged) {
    if (sg.inv_slot < sg.inv_nslots) {
#pragma unroll
      for (int w = 0; w < NW; ++w) sg.inv_lds[sg.inv_slot * NW + w] = fl[w] & 0x55555555u;
    } else {
#pragma unroll
      for (int w = 0; w < NW; ++w) sg.inv[strl_thread_at() + (uint32_t)w * sg.inv_stride] = fl[w] & 0x55555555u;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) nn += strl_popc(fl[w] & 0xAAAAAAAAu);
  }
  return nn;
}
template <int NW> STRL_DEV void inv_load(const Seg<NW> &sg, uint32_t (&v)[NW]) {
  if (sg.inv_slot < sg.inv_nslots) {
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = sg.inv_lds[sg.inv_slot * NW + w];
  } else {
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = sg.inv[strl_thread_at() + (uint32_t)w * sg.inv_stride];
  }
}

// wave-uniform bounds of the segment lengths of the lanes that carry an item (idle lanes: nothing to bound)
struct LenBounds { int lo, hi; };
STRL_DEV LenBounds len_bounds(bool active, int len) {
  LenBounds b;
  b.lo = strl_wave_min(active ? len : 0x7fffffff);
  b.hi = strl_wave_max(active ? len : 0);
  return b;
}

// Build a Seg whose first base is base 0 of raw[0] (whole reads: SEQ starts on a 16-byte boundary of the batch's SEQ array),
// straight from the registers the prefetch loaded: no staging through LDS, no funnel shifts.  Words every lane of the wave
// fills completely (16 (w + 1) <= lb.lo) skip the end-of-read mask.
template <int NW, int NRAW> STRL_DEV void seg_from_words(const uint32_t (&raw)[NRAW], const uint32_t *clut, int len, const LenBounds &lb, Seg<NW> &sg) {
  static_assert(NRAW >= 2 * NW, "two raw dwords per word of the 2-bit stream");
  uint32_t any_f = 0, fl[NW];
  // all table lookups of a half are in flight together (a branch per word would serialise ten LDS round trips); the second
  // half is skipped when no lane of the wave reaches it
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int w0 = h ? NW / 2 : 0, w1 = h ? NW : NW / 2;
#pragma unroll
    for (int w = w0; w < w1; ++w) { sg.seq[w] = 0; fl[w] = 0; }
    if (h && 16 * w0 >= lb.hi) continue;   // wave-uniform
#pragma unroll
    for (int w = w0; w < w1; ++w) {
      const uint32_t a = conv_swizzle(raw[2 * w]), b = conv_swizzle(raw[2 * w + 1]);
      const uint32_t ra = clut[a & 0xffu] | (clut[(a >> 8) & 0xffu] << 4) | (clut[(a >> 16) & 0xffu] << 8) | (clut[a >> 24] << 12);
      const uint32_t rb = clut[b & 0xffu] | (clut[(b >> 8) & 0xffu] << 4) | (clut[(b >> 16) & 0xffu] << 8) | (clut[b >> 24] << 12);
      sg.seq[w] = (ra & 0xffffu) | (rb << 16);
      fl[w] = (ra >> 16) | (rb & 0xffff0000u);
    }
  }
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    if (16 * (w + 1) > lb.lo) {   // wave-uniform: some lane ends inside this word (or before it)
      const int nv = len - 16 * w;
      const uint32_t m = nv >= 16 ? 0xffffffffu : nv <= 0 ? 0u : ((1u << (2 * nv)) - 1u);
      sg.seq[w] &= m;
      fl[w] &= m;
    }
    any_f |= fl[w];
  }
  const int nn = inv_store<NW>(sg, any_f != 0, fl);
  sg.len = len;
  sg.n_N = nn;
}

// Build a Seg from raw BAM-packed dwords staged in this lane's LDS column (`raw[i * STRL_LANES]`
// is dword i of the staged chunk), starting at base `s0` (< 32) of the chunk.
template <int NW> STRL_DEV void seg_from_raw(const uint32_t *raw, const uint32_t *clut, int s0, int len, Seg<NW> &sg) {
  const int dw0 = s0 >> 3, sh = s0 & 7;
  uint32_t p, f;
  conv8_lut(raw[dw0 * STRL_LANES], clut, p, f);
  uint32_t cur = p >> (2 * sh), curf = f >> (2 * sh);
  const int fill = 16 - 2 * sh;
  uint32_t any_f = 0, fl[NW];
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
    fl[w] = fw;
    any_f |= fw;
  }
  const int nn = inv_store<NW>(sg, any_f != 0, fl);
  sg.len = len;
  sg.n_N = nn;
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
// Ways to get newc_i, picked per k by what is cheapest on CDNA4 (integer VALU issue is what binds the kernel):
//   k = 2,3 : one 32-bit bin per CLASS (10 / 24 minimum-rotation classes) in LDS, [bin][lane]; ds_add_rtn returns the old count
//   k = 4   : 70 uint8 bins packed 4 per dword, 18 rows.  (Counting k = 4 in registers with byte-parallel equality was
//             tried: 2.6x the VALU instructions of the LDS bins.)
//   k = 5,6 : in registers, one code per VGPR (<= 32 windows for reads <= 160 bases); longer reads fall back to a
//             per-lane open-addressing table in LDS.
//
// k <= 4 (stage A).  Running arg-max without compare / select chains: inside a batch of 8 windows
//   key = newc << CS | (7 - j) << (CS - 3) | tag      (tag: which bin -- see below)
// and one unsigned max picks the largest running count and, among equals, the earliest window; across batches an earlier
// batch keeps the lead unless a later one has a strictly larger count -- together the code inc() would have kept.
// What a window costs besides its two LDS operations (table read, ds_add_rtn):
//   k = 2: the table is indexed by a BYTE of the stream (two windows; sub-dword operand select, one shift for the pair) and
//          holds the two bins' byte offsets; bin address = offset + column (one add, 16-bit operand select); the bins count in
//          units of 1 << CS and the tag IS that address -- the byte offset from `bins0`, the first wave's bin region, a
//          compile-time LDS address that folds into the ds instruction's offset field (a static allocation is at most
//          64 KB: 16 bits) -- so key = old + constant + address is one three-operand add.
//   k = 3: 6-bit windows do not align: field extract, uint8 class table, address = class << 8 + column, same key.
//   k = 4: byte-indexed table -> row offset | shift << 16; increment 1 << shift, address, old >> shift into the top byte
//          (all with sub-dword operand selects), key = that + (entry + constant): five instructions.
// Windows past the longest segment of the wave are not executed (lb.hi); batches below the shortest (lb.lo) need no
// per-lane masks; in between a lane past its own end still counts into its bins -- nothing reads them again -- and only its
// key is zeroed.
template <int K, int NW>
STRL_DEV void hist_pass_a(const Seg<NW> &sg, bool active, uint32_t *bins0, uint32_t col_off, const uint32_t *ta, const LenBounds &lb, int &cmax, uint32_t &imax) {
  static_assert(K >= 2 && K <= 4, "stage A");
  constexpr uint32_t MASK = (1u << (2 * K)) - 1u;
  constexpr int NWIN = NW * 16 / K;
  constexpr int B = 8;                       // three tie-break bits
  constexpr int CS = K == 4 ? 24 : 19;       // the count field; below it 3 tie-break bits and the tag (16 bits; k = 4: 21)
  constexpr uint32_t LOW = (1u << CS) - 1u;
  const int nwin = active ? sg.len / K : 0;
  const int umin = lb.lo == 0x7fffffff ? 0 : lb.lo / K, umax = lb.hi / K;   // wave-uniform
  const uint8_t *t3 = reinterpret_cast<const uint8_t *>(ta + TA_K3);
  cmax = 0;
  imax = MASK;
  uint32_t best = 0;
  auto kc = [](int j) { return (1u << CS) | ((7u - (uint32_t)j) << (CS - 3)); };
  // table entry of window i (shared by two windows at k = 2; a window past the segment reads zero bases: some valid entry)
  auto entry = [&](int i) -> uint32_t {
    if (K == 2) return ta[TA_K2 + ((sg.seq[i >> 3] >> (8 * ((i >> 1) & 3))) & 0xffu)];
    if (K == 4) return ta[TA_K4 + ((sg.seq[i >> 2] >> (8 * (i & 3))) & 0xffu)];
    const int bit = 6 * i, w = bit >> 5, sh = bit & 31;
    const uint32_t lo = sg.seq[w], hi = (w + 1 < NW) ? sg.seq[w + 1] : 0u;
    return t3[((sh + 6 <= 32) ? (lo >> sh) : strl_funnel_r(lo, hi, sh)) & 63u];
  };
  auto entries = [&](int b0, uint32_t (&e)[B]) {
#pragma unroll
    for (int j = 0; j < B; ++j) e[j] = (b0 + j >= NWIN) ? 0u : (K == 2 && (j & 1)) ? e[j - 1] : entry(b0 + j);
  };
  // count + key of window j of a batch
  auto count = [&](int j, uint32_t e) -> uint32_t {
    if (K == 4) {
      const uint32_t sh = (e >> 16) & 0xffu;
      uint32_t *bin = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(bins0) + (col_off + (e & 0xffffu)));
      const uint32_t old = strl_lds_add(bin, 1u << sh);
      return ((old >> sh) << 24) + (e + kc(j));
    }
    const uint32_t at = col_off + (K == 2 ? ((j & 1) ? (e >> 16) : (e & 0xffffu)) : e * ROWB);
    const uint32_t old = strl_lds_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(bins0) + at), 1u << CS);
    return old + kc(j) + at;
  };
  // The table reads run one batch ahead of the counting (issued before the branches of the batch in hand, so that they
  // overlap its ds_add round trip): one LDS latency per batch instead of two.
  uint32_t e[B], en[B];
  entries(0, en);
#pragma unroll
  for (int b0 = 0; b0 < NWIN; b0 += B) {
#pragma unroll
    for (int j = 0; j < B; ++j) e[j] = en[j];
    if (b0 + B < NWIN) entries(b0 + B, en);
    if (b0 >= umax) continue;                // wave-uniform (not a `break`: the trip count stays a constant and the loop unrolls)
    uint32_t key[B];
    if (b0 + B <= umin && b0 + B <= NWIN) {    // wave-uniform: every lane has all eight windows
#pragma unroll
      for (int j = 0; j < B; ++j) key[j] = count(j, e[j]);
    } else {
      // two half batches, the second only if some lane reaches it; a lane past its own end counts into its bins all the
      // same -- nothing reads them again -- and zeroes its key
#pragma unroll
      for (int h = 0; h < B; h += B / 2) {
#pragma unroll
        for (int j = 0; j < B / 2; ++j) key[h + j] = 0;
        if (b0 + h < NWIN && b0 + h < umax) {   // wave-uniform
#pragma unroll
          for (int j = 0; j < B / 2; ++j)
            if (b0 + h + j < NWIN) {
              const uint32_t k = count(h + j, e[h + j]);
              key[h + j] = (b0 + h + j < nwin) ? k : 0u;
            }
        }
      }
    }
    uint32_t bb = key[0] > key[1] ? key[0] : key[1];
#pragma unroll
    for (int j = 2; j + 1 < B; j += 2) bb = strl_max3(bb, key[j], key[j + 1]);
    if (bb > (best | LOW)) best = bb;          // strictly larger count only
  }
  if (best) {   // no window at all: count 0, code "all ones" (utils.nim:197-198)
    const uint16_t *tc = reinterpret_cast<const uint16_t *>(ta + TA_C) + TaCls<K>::off;
    cmax = (int)(best >> CS);
    if (K == 4) imax = tc[4u * ((best & 0xffffu) / ROWB) + ((best >> 19) & 3u)];
    else imax = tc[((best & 0xffffu) - col_off) / ROWB];
  }
}

// k = 5, 6 (stage B)
template <int K, int NW, int SLOTS>
STRL_DEV void hist_pass(const Seg<NW> &sg, bool active, uint32_t *tab, const uint16_t *lut, int &cmax, uint32_t &imax) {
  static_assert(K >= 5, "stage B");
  constexpr uint32_t MASK = (1u << (2 * K)) - 1u;
  constexpr int NWIN = NW * 16 / K;
  constexpr int B = 8;
  const uint16_t *lk = lut + LutOff<K>::v;
  const int nwin = active ? sg.len / K : 0;
  cmax = 0;
  imax = MASK;
  if (NW <= 10) {
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
#pragma unroll
  for (int b0 = 0; b0 < NWIN; b0 += B) {
    if (!strl_any(b0 < nwin)) break;  // wave-uniform early exit for short segments
    uint32_t code[B];
#pragma unroll
    for (int j = 0; j < B; ++j) code[j] = (b0 + j < NWIN) ? window_code<K, NW>(sg, lk, b0 + j) : 0u;
    {  // long reads: open addressing, entry = (code+1) << 8 | count
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
// lo_len: a wave-uniform lower bound of the lengths of the lanes that call (0: unknown) -- words that lie wholly below it
// need no end-of-read mask.
template <int K, int NW> STRL_DEV int recount(const Seg<NW> &sg, uint32_t code, int lo_len) {
  uint32_t acc[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) acc[w] = 0;
  // rare per lane (every other wave though): a non-ACGT base inside the window kills the match.  The words are fetched from
  // the lane's slot first, so that the round trip runs beside the comparison below.
  uint32_t iv[NW];
  if (sg.has_inv()) inv_load<NW>(sg, iv);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const uint32_t pat = ((code >> (2 * (K - 1 - j))) & 3u) * 0x55555555u;  // unit base j replicated
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t lo = sg.seq[w], hi = (w + 1 < NW) ? sg.seq[w + 1] : 0u;
      acc[w] |= strl_funnel_r(lo, hi, 2 * j) ^ pat;
    }
  }
  if (sg.has_inv()) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t lo = iv[w], hi = (w + 1 < NW) ? iv[w + 1] : 0u;
        acc[w] |= strl_funnel_r(lo, hi, 2 * j);
      }
    }
  }
  const int limit = sg.len - K + 1;  // number of start positions
  uint32_t m[NW];                    // bit 2i of m[w] set <=> the unit matches at base 16w + i
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    uint32_t x = ~(acc[w] | (acc[w] >> 1)) & 0x55555555u;
    if (16 * (w + 1) > lo_len - K + 1) {   // wave-uniform
      const int nv = limit - 16 * w;
      if (nv < 16) x &= (nv <= 0) ? 0u : ((1u << (2 * nv)) - 1u);
    }
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

// rows of the wave's table region a k pass touches (one bin per class, or the hash slots)
template <int K, int SLOTS> struct KRows { static constexpr int v = (K == 2) ? 10 : (K == 3) ? 24 : (K == 4) ? 18 : SLOTS; };

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

// tables: stage A's u32 tables (k <= 4) or the u16 code tables (k = 5, 6)
template <int K, int NW, int SLOTS>
STRL_DEV void score_k(const Seg<NW> &sg, ScoreState &st, uint32_t *wave_tab, int lane, const void *tables, const LaneThr &t, const LenBounds &lb, uint32_t *bins0) {
  if (!strl_any(st.alive)) return;  // wave-uniform
  if (K <= 4 || NW > 10) clear_rows(wave_tab, lane, KRows<K, SLOTS>::v);   // k = 5,6 of short reads count in registers
  int c;
  uint32_t code;
  if constexpr (K <= 4) hist_pass_a<K, NW>(sg, st.alive, bins0, (uint32_t)((wave_tab - bins0) + lane) * 4u, static_cast<const uint32_t *>(tables), lb, c, code);
  else hist_pass<K, NW, SLOTS>(sg, st.alive, wave_tab + lane, static_cast<const uint16_t *>(tables), c, code);
  STRL_PH(st, 2 + 2 * (K - 2));
  if (st.alive) {
    int score = c * K;
    if (score <= st.best) {  // utils.nim:250-253
      if (c < thr_get<K>(t.w12)) st.alive = false;  // break
    } else {
#if defined(STRL_PHASE_TIMING) && !defined(STRL_EMU)
      {
        const unsigned long long bm = __ballot(true);
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)bm) - 1)) { atomicAdd(&g_phase[23 + 2 * (K - 2) - (K > 4 ? 4 : 0)], 1ull); atomicAdd(&g_phase[24 + 2 * (K - 2) - (K > 4 ? 4 : 0)], (unsigned long long)__popcll(bm)); }
      }
#endif
#ifdef STRL_EXP_SKIP_RECOUNT34      // timing experiment only (wrong results): what the k = 3, 4 literal recounts cost stage A -- the
                                    // ceiling of anything that would pool them over a block's lanes (profiles/r06/stage_a_recount_bound.txt)
      if (K != 3 && K != 4)
#endif
      c = recount<K, NW>(sg, code, lb.lo == 0x7fffffff ? 0 : lb.lo);  // utils.nim:254
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
STRL_DEV void score_stage_a(const Seg<NW> &sg, bool active, uint32_t *wave_tab, uint32_t *bins0, int lane, const uint32_t *ta, const LaneThr &t,
                            const LenBounds &lb, ScoreState &st) {
  st.best = -1;
  st.alive = active && sg.n_N <= 20;
  st.res0 = st.res1 = 0;
  score_k<2, NW, SLOTS>(sg, st, wave_tab, lane, ta, t, lb, bins0);
  score_k<3, NW, SLOTS>(sg, st, wave_tab, lane, ta, t, lb, bins0);
  score_k<4, NW, SLOTS>(sg, st, wave_tab, lane, ta, t, lb, bins0);
}
template <int NW, int SLOTS>
STRL_DEV void score_stage_b(const Seg<NW> &sg, uint32_t *wave_tab, int lane, const uint16_t *lut, const LaneThr &t, const LenBounds &lb, ScoreState &st) {
  score_k<5, NW, SLOTS>(sg, st, wave_tab, lane, lut, t, lb, nullptr);
  score_k<6, NW, SLOTS>(sg, st, wave_tab, lane, lut, t, lb, nullptr);
}

}  // namespace strl
