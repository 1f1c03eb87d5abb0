// inflate_wave.h -- DEFLATE (RFC 1951) decoder of bgzf.hip: ONE WAVEFRONT per BGZF block (extract.nim:275,289 read the BAM
// through htslib's single-threaded inflate).
//
// DEFLATE is serial inside a stream, a BAM is tens of thousands to millions of independent <= 64 KiB streams.  Round 2 gave
// every LANE a stream: 64 divergent decoders per wave, tables too big for more than one wave per CU -> 6.6 GB/s.  Here a
// whole wave owns one stream and nothing diverges:
//   * the symbol loop is WAVE-UNIFORM: bit buffer, positions and table entries live in scalar registers (the compiler keeps
//     uniform values in SGPRs and runs the shifts/masks on the scalar ALU); the only vector work per symbol is the LDS table
//     lookup (one broadcast ds_read + v_readfirstlane);
//   * input: each lane holds one dword of a 256-byte window of the compressed stream (one coalesced load per 256 bytes, the
//     next window prefetched); the bit buffer is refilled with v_readlane -- no LDS, no scalar-cache traffic;
//   * literals are collected with v_writelane into one VGPR (lane = output position mod 64) and leave as one coalesced
//     64-byte store; LZ77 matches are copied by all 64 lanes at once, periodic (distance < length) matches included:
//     byte k of a match is out[pos - D + k mod D], which only reads bytes written BEFORE the match;
//   * Huffman tables are built by the 64 lanes together: per-length ranks by ballots, then every lane fills its share of the
//     direct-lookup tables entry by entry through a canonical decode of the entry's own index (no scattered replication);
//     9-bit first level for literal/length codes, 8-bit for distances; longer codes go through a canonical search -- inside the
//     symbol loop for literals and distances (round 5), in the C++ around it for the rest.
//   * there is NOT ONE lane-dependent branch in the kernel: lanes that have nothing to load or store in a step get an
//     out-of-range offset into a bounds-checked buffer descriptor (the hardware drops the access) or a dummy LDS slot.
//     This is what keeps the symbol loop scalar: with a divergent `if` anywhere near it LLVM sinks uniform code into the
//     branch's arms and the uniformity analysis then marks everything behind the join as divergent (VGPRs, exec masks).
//     The descriptors also bound every global access to the stream's input and the block's output: corrupt data cannot make
//     the decoder read or write anywhere else.
// 4.4 KB of LDS per wave and 80 VGPRs: six waves per SIMD; no divergence, every global access coalesced.
//
// The same source compiles for the host (STRL_EMU: the 64 lanes become loops) purely so that the CPU-only test-suite can
// run the decoder logic against zlib; the product never runs that build.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef STRL_EMU
#include <string.h>
#define IW_DEV inline
#define IW_FOR_LANES for (int lane = 0; lane < 64; ++lane)
#define IW_U(...) (__VA_ARGS__)
#define IW_RCP(x) (1.0f / (x))
#define IW_SYNC() ((void)0)
#define IW_BALLOT(out, expr)                                          \
  do {                                                                \
    uint64_t iw_m_ = 0;                                               \
    for (int lane = 0; lane < 64; ++lane)                             \
      if (expr) iw_m_ |= 1ull << lane;                                \
    (out) = iw_m_;                                                    \
  } while (0)
namespace strl {
template <class T> struct IwLane {
  T v[64];
  IW_DEV T &operator[](int l) { return v[l]; }
  IW_DEV const T &operator[](int l) const { return v[l]; }
};
IW_DEV uint32_t iw_readlane(const IwLane<uint32_t> &r, uint32_t k) { return r.v[k]; }
IW_DEV void iw_writelane(IwLane<uint32_t> &r, uint32_t k, uint32_t val) { r.v[k] = val; }
IW_DEV uint32_t iw_brev(uint32_t x) {
  x = (x >> 16) | (x << 16);
  x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
  x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
  x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
  x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
  return x;
}
IW_DEV uint32_t iw_popc_below(uint64_t m, int lane) { return (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)); }
IW_DEV uint32_t iw_popc(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }
// bounds-checked view of global memory: loads outside [0, n) give 0, stores outside are dropped
struct IwBuf { uint8_t *p; uint32_t n; };
IW_DEV IwBuf iw_make_buf(const void *p, uint64_t n) { return IwBuf{(uint8_t *)p, n > 0x7ffffffcull ? 0x7ffffffcu : (uint32_t)n}; }
IW_DEV uint32_t iw_ld32(const IwBuf &b, uint32_t off) { uint32_t v = 0; if ((uint64_t)off + 4 <= b.n) memcpy(&v, b.p + off, 4); return v; }
IW_DEV uint32_t iw_ld8(const IwBuf &b, uint32_t off) { return off < b.n ? b.p[off] : 0u; }
IW_DEV void iw_st8(const IwBuf &b, uint32_t off, uint32_t v) { if (off < b.n) b.p[off] = (uint8_t)v; }
}  // namespace strl
#else
#include <hip/hip_runtime.h>
#define IW_DEV __device__ __forceinline__
#define IW_FOR_LANES for (int lane = (int)threadIdx.x, iw_once_ = 1; iw_once_; iw_once_ = 0)
#define IW_U(...) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(__VA_ARGS__)))
#define IW_RCP(x) __builtin_amdgcn_rcpf(x)
#define IW_SYNC()                                              \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
  } while (0)
#define IW_BALLOT(out, expr)                 \
  do {                                       \
    const int lane = (int)threadIdx.x;       \
    (out) = __ballot(expr);                  \
  } while (0)
namespace strl {
template <class T> struct IwLane {
  T x;
  IW_DEV T &operator[](int) { return x; }
  IW_DEV const T &operator[](int) const { return x; }
};
IW_DEV uint32_t iw_readlane(const IwLane<uint32_t> &r, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)r.x, (int)k); }
__device__ int iw_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane");
IW_DEV void iw_writelane(IwLane<uint32_t> &r, uint32_t k, uint32_t val) { r.x = (uint32_t)iw_llvm_writelane((int)val, (int)k, (int)r.x); }   // k, val wave-uniform
IW_DEV uint32_t iw_brev(uint32_t x) { return __builtin_bitreverse32(x); }
IW_DEV uint32_t iw_popc_below(uint64_t m, int) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
IW_DEV uint32_t iw_popc(uint64_t m) { return (uint32_t)__popcll(m); }
// bounds-checked view of global memory (raw buffer descriptor, stride 0): the hardware returns 0 for loads outside
// [0, num_records) and drops stores there
struct IwBuf { __amdgpu_buffer_rsrc_t r; };
IW_DEV IwBuf iw_make_buf(const void *p, uint64_t n) {
  return IwBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, n > 0x7ffffffcull ? 0x7ffffffc : (int)n, 0x00020000)};
}
IW_DEV uint32_t iw_ld32(const IwBuf &b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)off, 0, 0); }
IW_DEV uint32_t iw_ld8(const IwBuf &b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b8(b.r, (int)off, 0, 0); }
IW_DEV void iw_st8(const IwBuf &b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, b.r, (int)off, 0, 0); }
}  // namespace strl
#endif

namespace strl {

constexpr int IW_ERR_DATA = 1, IW_ERR_SIZE = 2, IW_ERR_CRC = 4;
constexpr uint32_t IW_OOB = 0x80000000u;   // an offset no descriptor of this file covers (num_records < 2^31)
#ifndef IW_LIT_ROOT_BITS
#define IW_LIT_ROOT_BITS 9     // measured: 9 bits (2 KB, 4.4 KB of LDS per wave: 28 waves per CU) 95.6 / 70.7 GB/s, 10 bits 87.9 / 64.2 (profiles/r04/inflate_hybrid.txt)
#endif
constexpr int IW_LIT_ROOT = IW_LIT_ROOT_BITS, IW_DIST_ROOT = 8, IW_CL_ROOT = 7;

// Table entry (u32): [3:0] code length (0 = "not in the first-level table": canonical search / invalid), [7:4] extra bits,
// [9:8] kind (0 literal / plain value, 1 base value of a length or distance code, 2 end of block), [14:10] code length + extra
// bits, [15] a length with its extra bits resolved (IW_LEN_DONE), [30:16] value (<= 24577), [31] literal: the sign bit is what
// the symbol loop tests.
constexpr uint32_t IW_KIND_BASE = 1u << 8, IW_KIND_EOB = 2u << 8, IW_KIND_BAD = 3u << 8;
// [15] (literal/length table, first level) a length whose extra bits are already in the value: the index held them (code + extra
// bits <= the index bits); [7:4] is 0 and [3:0] = [14:10] = code + extra bits, so every reader of the plain format decodes it too.
constexpr uint32_t IW_LEN_DONE = 1u << 15;
IW_DEV uint32_t iw_with_total(uint32_t e, uint32_t l) { return e | l | ((l + ((e >> 4) & 15u)) << 10); }   // code length + the bits the symbol consumes in all
constexpr uint32_t IW_FAST_LIT = 1u << 31;   // set in literal entries
IW_DEV uint32_t iw_val(uint32_t e) { return (e >> 16) & 0x7fffu; }

// LDS of one wave.  The code-length code's tables are only alive while the literal/length and distance code lengths are
// being read, before the distance table is built: they share its storage.
struct IwLds {
  uint32_t lit_tab[1 << IW_LIT_ROOT];
  union {
    uint32_t dist_tab[1 << IW_DIST_ROOT];
    struct {
      uint32_t cl_tab[1 << IW_CL_ROOT];
      uint16_t cl_sorted[32];                          // 19 symbols; [31] = the dummy slot of idle lanes
      uint32_t cl_limit[16], cl_delta[16], cl_offs[16];
    } cl;
  };
  uint16_t ll_sorted[288 + 2];                         // symbols ordered by (code length, symbol); [288] / [32]: dummy slots of idle lanes
  uint16_t d_sorted[32 + 2];
  uint32_t ll_limit[16], ll_delta[16], ll_offs[16];   // per code length 1..15: see iw_build
  uint32_t d_limit[16], d_delta[16], d_offs[16];
  uint8_t lens[320 + 4];                               // code lengths: literal/length symbols, then distance symbols; [320] dummy
};

// RFC 1951 3.2.5 in closed form.
//   length symbol 257 + c: c < 8: 3 + c; c == 28: 258; else e = (c - 4) / 4 extra bits, base ((4 + c % 4) << e) + 3
//   distance symbol d: d < 4: 1 + d; else e = (d - 2) / 2 extra bits, base ((2 + d % 2) << e) + 1
enum { IW_CODES = 0, IW_LENS = 1, IW_DISTS = 2 };
template <int KIND> IW_DEV uint32_t iw_entry_of(uint32_t s) {
  if (KIND == IW_CODES) return s << 16;
  if (KIND == IW_LENS) {
    if (s < 256u) return (s << 16) | IW_FAST_LIT;
    if (s == 256u) return IW_KIND_EOB;
    const uint32_t c = s - 257u;
    if (c > 28u) return IW_KIND_BAD;                                   // 286, 287: take part in the code, never valid
    if (c < 8u) return ((3u + c) << 16) | IW_KIND_BASE;
    if (c == 28u) return (258u << 16) | IW_KIND_BASE;
    const uint32_t e = (c - 4u) >> 2;
    return ((((4u + (c & 3u)) << e) + 3u) << 16) | IW_KIND_BASE | (e << 4);
  }
  if (s > 29u) return IW_KIND_BASE;                                    // 30, 31: never valid (distance 0)
  if (s < 4u) return ((1u + s) << 16) | IW_KIND_BASE;
  const uint32_t e = (s - 2u) >> 1;
  return ((((2u + (s & 1u)) << e) + 1u) << 16) | IW_KIND_BASE | (e << 4);
}

// order of the code-length code lengths (RFC 1951 3.2.7), 5 bits each: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
IW_DEV uint32_t iw_cl_order(uint32_t i) {
  const uint64_t lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) |
                      (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
  const uint64_t hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
  return (uint32_t)((i < 12u ? lo >> (5u * i) : hi >> (5u * (i - 12u))) & 31u);
}

// Build the decoding tables of one Huffman code from its code lengths (lane s + 64 j holds the length of symbol s + 64 j in
// len[j]; 0 = unused).  Canonical codes (RFC 1951 3.2.2): codes of one length are consecutive, in symbol order, shorter
// codes first.  With v16 = the next 16 stream bits read most-significant-bit first,
//   limit[l] = (first code of length l + number of codes of length l) << (16 - l): v16 < limit[l] <=> the code has <= l bits
//   delta[l] = (index of the first length-l symbol in `sorted`) - (first code of length l)
//   sorted[] = the symbols ordered by (length, symbol)
// so a code of length l decodes to the symbol sorted[(v16 >> (16 - l)) + delta[l]] (iw_entry_of makes its table entry).  tab[] is the direct table for the first ROOT bits
// (stream order = least significant bit first): every lane fills entries lane, lane + 64, ... by decoding the entry's own
// index; longer codes leave 0 there and go through iw_slow.  `dummy` = a slot of sorted[] behind the symbols.
// Returns 0 for an over-subscribed code, or an incomplete one where zlib refuses it (inftrees.c: incomplete codes are
// only allowed for a literal/length or distance code consisting of ONE code of length 1; no codes at all is allowed too).
template <int ROUNDS, int ROOT, int KIND>
IW_DEV uint32_t iw_build(const IwLane<uint32_t> (&len)[ROUNDS], uint32_t *tab, uint16_t *sorted, uint32_t dummy, uint32_t *limit, uint32_t *delta,
                         uint32_t *offs) {
  IwLane<uint32_t> rank[ROUNDS];
  IW_FOR_LANES {
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) rank[j][lane] = 0;
  }
  uint32_t cnt[16];
#pragma unroll
  for (int l = 1; l <= 15; ++l) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
      uint64_t m;
      IW_BALLOT(m, len[j][lane] == (uint32_t)l);
      IW_FOR_LANES {
        const uint32_t r = c + iw_popc_below(m, lane);
        rank[j][lane] = len[j][lane] == (uint32_t)l ? r : rank[j][lane];
      }
      c += iw_popc(m);
    }
    cnt[l] = c;
  }
  uint32_t code = 0, off = 0, max_len = 0, over = 0;
#pragma unroll
  for (int l = 1; l <= 15; ++l) {
    const uint32_t first = code, end = first + cnt[l];
    over |= (uint32_t)(end > (1u << l));
    max_len = cnt[l] ? (uint32_t)l : max_len;
    const uint32_t lim = end << (16 - l), del = off - first, o = off;
    limit[l] = lim; delta[l] = del; offs[l] = o;                     // every lane stores the same value: no branch
    code = end << 1;
    off += cnt[l];
  }
  const uint32_t complete = (uint32_t)(code == (1u << 16));           // after l = 15: code = (first + cnt) << 1
  const uint32_t bad = over | (uint32_t)(!complete && off != 0u && (KIND == IW_CODES || max_len != 1u));
  IW_SYNC();
  IW_FOR_LANES {
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
      const uint32_t l = len[j][lane];
      const uint32_t at = l ? offs[l] + rank[j][lane] : dummy;        // (offs[0] is never written: the select keeps it out)
      sorted[at < dummy ? at : dummy] = (uint16_t)(j * 64 + lane);
    }
  }
  IW_SYNC();
  uint32_t lim[ROOT + 1];
#pragma unroll
  for (int l = 1; l <= ROOT; ++l) lim[l] = IW_U(limit[l]);
  for (uint32_t i0 = 0; i0 < (1u << ROOT); i0 += 64u) {
    IW_FOR_LANES {
      const uint32_t i = i0 + (uint32_t)lane;
      const uint32_t v16 = iw_brev(i) >> 16;
      uint32_t l = 0;
#pragma unroll
      for (int k = ROOT; k >= 1; --k) l = v16 < lim[k] ? (uint32_t)k : l;
      const uint32_t at = l ? (v16 >> (16u - l)) + delta[l] : dummy;
      uint32_t e = iw_with_total(iw_entry_of<KIND>(sorted[at < dummy ? at : dummy]), l);
      if (KIND == IW_LENS) {
        // a length code whose extra bits lie inside the index: the entry names the length itself (nearly every match of a BAM
        // block: lengths up to 34 have <= 2 extra bits behind 4..7-bit codes) -- the symbol loop then needs no field of it but two
        const uint32_t x = (e >> 4) & 15u, done = (uint32_t)((e & (IW_FAST_LIT | (3u << 8))) == IW_KIND_BASE && l + x <= (uint32_t)ROOT);
        const uint32_t L = iw_val(e) + ((i >> l) & ((1u << x) - 1u)), t = l + x;
        e = done ? (L << 16) | IW_KIND_BASE | IW_LEN_DONE | (t << 10) | t : e;
      }
      tab[i] = l ? e : 0u;
    }
  }
  IW_SYNC();
  return IW_U(bad) ^ 1u;
}

// A code longer than the first-level table (or an unused prefix): canonical search over the remaining lengths.  0 = invalid.
template <int KIND> IW_DEV uint32_t iw_slow(uint64_t bb, const uint32_t *limit, const uint32_t *delta, const uint16_t *sorted, uint32_t dummy, int root) {
  const uint32_t v16 = iw_brev((uint32_t)bb) >> 16;
#ifdef STRL_EMU
  for (int l = root + 1; l <= 15; ++l) {
    const uint32_t lim = IW_U(limit[l]);
    if (v16 < lim) {
      const uint32_t at = (v16 >> (16 - l)) + IW_U(delta[l]);
      return iw_with_total(iw_entry_of<KIND>(IW_U(sorted[at < dummy ? at : dummy])), (uint32_t)l);
    }
  }
  return 0u;
#else
  // One in twenty symbols of a level-6 BAM block comes here (tools: IG_STATS count, profiles/r05/inflate_group/README.md: 719 long
  // literals and 159 long distance codes per block), and the walk over the lengths was an LDS round trip each, up to six in a row.
  // Lane k takes length root + 1 + k: every limit and delta in ONE round trip, the first length whose limit holds by a ballot.
  const int lane = (int)threadIdx.x, n = 15 - root;
  const int idx = root + 1 + (lane < n ? lane : 0);
  const uint32_t lim = limit[idx], del = delta[idx];
  const uint64_t m = __ballot(lane < n && v16 < lim);
  if (!m) return 0u;
  const int k = __ffsll((long long)m) - 1, l = root + 1 + k;
  const uint32_t at = (v16 >> (16 - l)) + (uint32_t)__builtin_amdgcn_readlane((int)del, k);
  return iw_with_total(iw_entry_of<KIND>(IW_U(sorted[at < dummy ? at : dummy])), (uint32_t)l);
#endif
}

// The compressed stream as the wave sees it: 64 dwords per lane-register, the next 64 prefetched.
struct IwBits {
  IwBuf in;                    // the stream piece from its dword-aligned origin on, bounds-checked
  uint32_t w0, widx;           // dword index of cur[lane 0]; next dword of cur to enter the bit buffer
  uint32_t skip;               // bytes between the origin and the first byte of the stream piece this reader was started on
  IwLane<uint32_t> cur, nxt;
  uint64_t bb;
  uint32_t nbits;

  IW_DEV void load(IwLane<uint32_t> &r, uint32_t first) {
    IW_FOR_LANES { r[lane] = iw_ld32(in, 4u * (first + (uint32_t)lane)); }
  }
  // start reading at byte `off` of `comp`; `readable` = bytes of comp that may be loaded (a multiple of 4 behind every stream)
  IW_DEV void init(const uint8_t *comp, uint64_t off, uint64_t readable) {
    const uint64_t a = off & ~(uint64_t)3;
    skip = (uint32_t)(off & 3u);
    in = iw_make_buf(comp + a, readable > a ? readable - a : 0);
    w0 = 0; widx = 0; bb = 0; nbits = 0;
    load(cur, 0);
    load(nxt, 64);
    refill();
    bb >>= 8u * skip;
    nbits -= 8u * skip;
  }
  IW_DEV void refill() {            // afterwards 30 <= nbits <= 61 (iw_run keeps the buffer two bits up and a sentinel bit above the valid ones)
    if (nbits < 30u) {
      const uint32_t w = iw_readlane(cur, widx);
      bb |= (uint64_t)w << nbits;
      nbits += 32u;
      if (++widx == 64u) rotate();
    }
  }
  IW_DEV void rotate() {            // the window is used up: the prefetched one takes over, the one behind it is requested
    cur = nxt;
    w0 += 64u;
    widx = 0;
    load(nxt, w0 + 64u);
  }
  IW_DEV uint32_t bits(uint32_t n) {   // n <= 16 (0 allowed); the caller keeps nbits >= n
    const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
    bb >>= n;
    nbits -= n;
    return v;
  }
  // bytes from the origin to the next unread bit (exact once nbits is a multiple of 8)
  IW_DEV uint32_t byte_pos() const { return 4u * (w0 + widx) - (nbits >> 3); }
  // bits consumed since init
  IW_DEV uint64_t consumed() const { return 32ull * (w0 + widx) - nbits - 8ull * skip; }
};

// Output of one stream: literals wait in a lane register (lane = position mod 64) for one coalesced store.
struct IwOut {
  IwBuf out;                      // the block's output [0, isize), bounds-checked
  uint32_t isize, pos;            // next output position
  // Literals wait in `pend` as their table entries (bit 31 set, the byte in bits 23:16), lane = position mod 64, until the
  // position leaves their 64-byte window (or a match reads from it): every staged literal lies in [(pos - 1) & ~63, pos).
  IwLane<uint32_t> pend;
  IwLane<uint32_t> pdata, paddr;  // (device, iw_run) bytes a match has loaded and where they go; IW_OOB: nothing pending
  IW_DEV void flush() {
    const uint32_t w = (pos - 1u) & ~63u;
    IW_FOR_LANES {
      const uint32_t v = pend[lane];
      iw_st8(out, (v >> 31) ? w + (uint32_t)lane : IW_OOB, v >> 16);
      pend[lane] = 0;
    }
  }
  IW_DEV void literal(uint32_t b) {   // caller checked pos < isize
    iw_writelane(pend, pos & 63u, (b << 16) | IW_FAST_LIT);
    ++pos;
    if ((pos & 63u) == 0u) flush();
  }
  // LZ77 match: length L (3..258), distance D (1..pos).  Byte k is out[pos - D + k mod D]: every source byte was written
  // before the match started, so all lanes copy at once, also when the match overlaps itself (D < L).
  IW_DEV void match(uint32_t L, uint32_t D) {
    flush();
    const uint32_t src = pos - D;
    if (D >= L) {
      for (uint32_t k0 = 0; k0 < L; k0 += 64u) {
        IW_FOR_LANES {
          const uint32_t k = k0 + (uint32_t)lane;
          const uint32_t v = iw_ld8(out, k < L ? src + k : IW_OOB);
          iw_st8(out, k < L ? pos + k : IW_OOB, v);
        }
      }
    } else {
      const float rcp = IW_RCP((float)D);
      for (uint32_t k0 = 0; k0 < L; k0 += 64u) {
        IW_FOR_LANES {
          const uint32_t k = k0 + (uint32_t)lane;
          const uint32_t q = (uint32_t)((float)k * rcp);
          int32_t r = (int32_t)k - (int32_t)(q * D);         // q is within one of k / D: one correction either way
          r = r < 0 ? r + (int32_t)D : r;
          r = r >= (int32_t)D ? r - (int32_t)D : r;
          const uint32_t v = iw_ld8(out, k < L ? src + (uint32_t)r : IW_OOB);
          iw_st8(out, k < L ? pos + k : IW_OOB, v);
        }
      }
    }
    pos += L;
  }
  // `n` bytes from the compressed stream itself (a stored block), starting `p` bytes behind the reader's origin
  IW_DEV void raw(const IwBuf &in, uint32_t p, uint32_t n) {
    flush();
    for (uint32_t k0 = 0; k0 < n; k0 += 64u) {
      IW_FOR_LANES {
        const uint32_t k = k0 + (uint32_t)lane;
        const uint32_t v = iw_ld8(in, k < n ? p + k : IW_OOB);
        iw_st8(out, k < n ? pos + k : IW_OOB, v);
      }
    }
    pos += n;
  }
};

// The symbol loop proper: literals and plain matches (distance >= length) -- codes longer than the first-level tables included
// since round 5 --, everything rarer handed back.
//   code 0: `e` is the entry of a symbol that is neither (nothing consumed; >= 20 bits in the buffer): end of block, a long
//           LENGTH code (or no code at all), an invalid length symbol
//   code 1: the 256-byte input window is used up (rotate it and come back)
//   code 4: a match of length L whose distance code is invalid or no code at all (L consumed; >= 30 bits in the buffer)
//   code 5: a match (L, D) the fast copy does not take: it overlaps itself (D < L) or fails a check (caller decides)
//   code 6: as 4, and the input window is used up
//   code 7: the output position passed ISIZE
// On the device this is hand-written ISA.  hipcc keeps the wave-uniform decoder state in scalar registers and turns the loop
// into a state machine of 64-bit flag registers: 55 scalar instructions per literal, ~150 per match (36 GB/s).
// What bounds the hand-written forms (level-1 blocks with constant qualities / level-6 blocks with binned random qualities):
//   * The data is match-dominated (tools/ubench/deflate_stats.py: per output byte 0.16 literals and 0.11 matches of mean
//     length 7.4 in the level-6 blocks; 0.10 and 0.085 of mean length 10.6 in the level-1 blocks), and a CU issues one
//     scalar-port instruction (scalar ALU, branch, s_waitcnt) and one wave64 vector instruction a cycle, whatever the waves.
//     The counters of each form (profiles/r04/inflate_pmc_*.txt) give the port's load = instructions / (CUs x kernel cycles):
//       round 3, all scalar              16 scalar + 3 vector per literal, 45 + 9 per match               81 / 59 GB/s
//       all vector                        3 + 10, 6 + 38                                                   88 / 64
//       split (bit buffer, length fields, output position scalar; the rest vector)   scalar 70 %, vector 66 %   88 / 64 -> 107 / 80 with
//                                         9-bit tables, the copy ahead of the literal store, the distance lookup issued early
//       literals by v_writelane, sentinel bit buffer, three literals per refill check: vector -30 %, scalar +5 %: 75 % / 47 %   108 / 81
//       literals staged across matches (no store per match): the same counts on the scalar side                 109 / 81
//       round 4's form: length and distance put together on the vector unit, one branch for a match's checks: 63 % / 68 %   117 / 87
//       round 5 (profiles/r05/inflate_long_codes.txt has the ledger): the codes longer than the first-level tables decoded IN the loop
//       (a level-6 block has ~880 among 17 600 symbols, and each had left the loop for the C++ around it: ~3 symbols' time), length
//       entries with the extra bits resolved by the table's index, the match path in front of the loop's head, a refill of 8 scalar
//       instructions, table addresses by one v_bfi from a buffer kept two bits up, no second lookup for a length behind one literal,
//       one field on the scalar unit: scalar instructions -27 %, vector -20 %, both ports ~70 % of a shorter launch   144 / 107
//     Neither port gets past ~70 %: a wave is parked on s_waitcnt for half its cycles (LDS lookups, the copy's loads) and each
//     SIMD holds seven of them.  Every step that took instructions off the busier port paid; taking them off the other did not
//     (-30 % vector instructions: +1 %), and more waves do not either (6 / 7 / 8 per SIMD: the same; profiles/r04/inflate_vmatch_waves.txt).
//   * Earlier readings, kept for the record: throughput follows the waves per CU up to 24 (inflate_occupancy.txt); a 9-bit first
//     level (4.4 KB of LDS per wave) beat 10 bits by 9 %; removing the copy's waits altogether gives +13 %, header parse + table
//     build are 3.6 % (inflate_exp_waits.txt, inflate_exp_header_only.txt).
//   scalar: the 64-bit bit buffer (a sentinel bit above its valid bits, the whole two bits up: "fewer than 30" is "high word zero";
//           no bit count), its shifts by the entry's own fields, the output position (in m0: v_writelane's lane select);
//   vector: the table address (a bit-field insert of the buffer's low word into the table's own address) + lookup, the literal's placement (v_writelane of the entry), length and distance from their
//           entries' fields (every lane the same value), the checks of a match folded into one sign test, the copy;
// values cross where an operand may sit in either file (a vector instruction reads one scalar register for free) and through
// v_readfirstlane otherwise (the table entry, the bits the distance code used, the position behind the match).
// The copy of a match is software-pipelined: its bytes are LOADED when the match is decoded and STORED when the next match (or an
// exit) comes around -- the load latency passes while the next symbols are decoded; every store is issued before any later
// load, so a later match that reads these bytes sees them.  bb lives in s[90:91]; s92..s95, vcc are scratch; m0 is saved and
// restored around the loop.
#if defined(STRL_EMU) && !defined(IW_EMU_PLAIN)
// The host build's twin of the hand-written loop below, label for label: the same sentinel bit buffer two bits up, the same three
// unrolled lookups and their bit budget, the same shortcuts, the pending copy, the exits.  (The plain C loop further down -- a
// refill check per symbol, no unrolling -- accepted two budget mistakes of round 5 that only the GPU then showed: a decoder that
// is tested on the CPU has to take the device's steps.  -DIW_EMU_PLAIN builds the plain loop instead.)
IW_DEV void iw_run(IwBits &br, IwOut &o, const IwLds &S, const IwLane<uint32_t> &llim, const IwLane<uint32_t> &ldel, uint32_t &e, uint32_t &code, uint32_t &L, uint32_t &D) {
  constexpr uint32_t LM = (1u << IW_LIT_ROOT) - 1u, DM = (1u << IW_DIST_ROOT) - 1u;
  uint64_t bb = (br.bb | (1ull << br.nbits)) << 2;
  uint32_t m0 = o.pos, wi = br.widx - 64u, ve = 0, vL = L, vD = D, vn = 0, vt1 = 0, vt2 = 0, vsrc = 0, s92 = 0, s93 = 0, s94 = 0, s95 = 0;
  e = 0;
  auto lo = [&]() { return (uint32_t)bb; };
  auto hi0 = [&]() { return (uint32_t)(bb >> 32) == 0u; };
  auto flush_pend = [&](uint32_t base) {                        // the staged literals (entries with bit 31) of the window at `base`
    for (int lane = 0; lane < 64; ++lane) {
      const uint32_t v = o.pend[lane];
      iw_st8(o.out, (v >> 31) ? (base | (uint32_t)lane) : IW_OOB, v >> 16);
      o.pend[lane] = 0;
    }
  };
  auto copy_round = [&](uint32_t k0, uint32_t n) {              // store what the previous round loaded, load this round's bytes
    for (int lane = 0; lane < 64; ++lane) iw_st8(o.out, o.paddr[lane], o.pdata[lane]);
    for (int lane = 0; lane < 64; ++lane) {
      const uint32_t k = k0 + (uint32_t)lane;
      o.pdata[lane] = iw_ld8(o.out, vsrc + k);
      o.paddr[lane] = n > k ? m0 + k : IW_OOB;
    }
  };
  auto refill = [&]() -> bool {                                 // true: the window's last dword has gone in
    const uint32_t p = 31u - (uint32_t)__builtin_clz(lo());
    const uint64_t w = (1ull << 32) | iw_readlane(br.cur, wi & 63u);
    bb = (bb & ~(1ull << p)) | (w << p);
    return ++wi == 0u;
  };
  auto v16 = [&]() { return iw_brev(lo() >> 2) >> 16; };
  goto loop;
notlit:
  if (!(e & IW_LEN_DONE)) goto lenx;
lendone:
  vL = ve >> 16;
  bb >>= (e & 15u);
  ve = S.dist_tab[(lo() >> 2) & DM];
  if (hi0()) goto refill2;
have2:
  vt2 = m0 + vL;
have2b:
  vn = ve & 15u;
  vt1 = (ve >> 4) & 15u;
  vD = (((lo() >> vn) >> 2) & ((1u << vt1) - 1u)) + (ve >> 16);
  s92 = (ve >> 10) & 31u;
  s93 = vt2;
  vsrc = m0 - vD;
  s94 = m0 & ~63u;
  vt1 = vL + vsrc;
  if ((int32_t)((vD - vL) | vsrc) < 0) goto hard;
  bb >>= s92;
  if (s94 < vt1) { flush_pend(s94); }                          // (L_iw_flushfirst)
  copy_round(0, vL);
  if ((s93 ^ m0) > 63u) goto cross;
  m0 = s93;
loop:
  if (hi0()) goto refill1;
have:
  ve = S.lit_tab[(lo() >> 2) & LM]; e = ve;
  if (!(e & IW_FAST_LIT)) goto notlit;
  bb >>= (e & 63u); iw_writelane(o.pend, m0 & 63u, e); ++m0;
  if (!(m0 & 63u)) goto full;
  ve = S.lit_tab[(lo() >> 2) & LM]; e = ve;
  if (!(e & IW_FAST_LIT)) goto second;
  bb >>= (e & 63u); iw_writelane(o.pend, m0 & 63u, e); ++m0;
  if (!(m0 & 63u)) goto full;
  ve = S.lit_tab[(lo() >> 2) & LM]; e = ve;
  if (!(e & IW_FAST_LIT)) goto loop;
lit3:
  bb >>= (e & 63u); iw_writelane(o.pend, m0 & 63u, e); ++m0;
  if (m0 & 63u) goto loop;
full:
  flush_pend(m0 - 64u);
  if (!(m0 > o.isize)) goto loop;
  code = 7; goto end;
second:
  if (e & IW_LEN_DONE) goto lendone;
  goto loop;
refill1:
  if (!refill()) goto have;
  code = 1; goto end;
cross:
  s95 = s93 - m0;
  for (s94 = 64; s95 > 64u && s94 < s95; s94 += 64u) copy_round(s94, s95);
  flush_pend(m0 & ~63u);
  m0 = s93;
  goto loop;
refill2:
  if (!refill()) goto have2;
  code = 6; goto end;
hard:
  if (vn == 0u) goto longd;
  bb >>= s92;
  code = 5; goto end;
longd: {
    const uint32_t v = v16();
    int k = 0;
    while (k < 15 - IW_DIST_ROOT && !(v < llim[8 + k])) ++k;
    if (k == 15 - IW_DIST_ROOT) goto exit4;
    const uint32_t l = (uint32_t)(IW_DIST_ROOT + 1 + k);
    uint32_t at = (v >> (uint32_t)(15 - IW_DIST_ROOT - k)) + ldel[8 + k];
    at = at < 32u ? at : 32u;
    const uint32_t sym = S.d_sorted[at];
    if (sym > 29u) goto exit4;
    ve = iw_with_total(iw_entry_of<IW_DISTS>(sym), l);
    vt2 = m0 + vL;
    goto have2b;
  }
exit4:
  code = 4; goto end;
lenx:
  if ((e & (3u << 8)) != IW_KIND_BASE) goto other;
  vL = (((lo() >> (ve & 15u)) >> 2) & ((1u << ((ve >> 4) & 15u)) - 1u)) + (ve >> 16);
  bb >>= ((e >> 10) & 31u);
  ve = S.dist_tab[(lo() >> 2) & DM];
  if (hi0()) goto refill2;
  goto have2;
other: {
    if (e & 15u) goto exit0;
    const uint32_t v = v16();
    int k = 0;
    while (k < 15 - IW_LIT_ROOT && !(v < llim[k])) ++k;
    if (k == 15 - IW_LIT_ROOT) goto exit0;
    uint32_t at = (v >> (uint32_t)(15 - IW_LIT_ROOT - k)) + ldel[k];
    at = at < 288u ? at : 288u;
    const uint32_t sym = S.ll_sorted[at];
    if (!(sym < 256u)) goto exit0;
    e = (sym << 16) | (uint32_t)(IW_LIT_ROOT + 1 + k) | IW_FAST_LIT;
    goto lit3;
  }
exit0:
  code = 0;
end:
  br.widx = wi + 64u;
  for (int lane = 0; lane < 64; ++lane) { iw_st8(o.out, o.paddr[lane], o.pdata[lane]); o.paddr[lane] = IW_OOB; }
  bb >>= 2;
  br.nbits = 63u - (uint32_t)__builtin_clzll(bb);
  br.bb = bb & ~(1ull << br.nbits);
  o.pos = m0;
  L = vL; D = vD;
  (void)vt1; (void)s95;
}
#elif defined(STRL_EMU)
IW_DEV void iw_run(IwBits &br, IwOut &o, const IwLds &S, const IwLane<uint32_t> &llim, const IwLane<uint32_t> &ldel, uint32_t &e, uint32_t &code, uint32_t &L, uint32_t &D) {
  const uint32_t *lit_tab = S.lit_tab, *dist_tab = S.dist_tab;
  for (;;) {
    if (br.nbits < 32u) {
      const uint32_t w = iw_readlane(br.cur, br.widx);
      br.bb |= (uint64_t)w << br.nbits;
      br.nbits += 32u;
      if (++br.widx == 64u) { code = 1; return; }
    }
    e = lit_tab[(uint32_t)br.bb & ((1u << IW_LIT_ROOT) - 1u)];
    if (e & IW_FAST_LIT) {
      br.bb >>= e & 15u;
      br.nbits -= e & 15u;
      iw_writelane(o.pend, o.pos & 63u, e);
      ++o.pos;
      if (!(o.pos & 63u)) {
        o.flush();
        if (o.pos > o.isize) { code = 7; return; }
      }
      continue;
    }
    if (!(e & 15u)) {                                                // a code longer than the first-level table: a literal stays in the loop
      const uint32_t v16 = iw_brev((uint32_t)br.bb) >> 16;
      int k = 0;
      while (k < 15 - IW_LIT_ROOT && !(v16 < llim[k])) ++k;
      if (k < 15 - IW_LIT_ROOT) {
        const uint32_t l = (uint32_t)(IW_LIT_ROOT + 1 + k), at = (v16 >> (16u - l)) + ldel[k], s = S.ll_sorted[at < 288u ? at : 288u];
        if (s < 256u) {
          br.bb >>= l;
          br.nbits -= l;
          iw_writelane(o.pend, o.pos & 63u, (s << 16) | IW_FAST_LIT | l);
          ++o.pos;
          if (!(o.pos & 63u)) {
            o.flush();
            if (o.pos > o.isize) { code = 7; return; }
          }
          continue;
        }
      }
      code = 0;
      return;
    }
    if ((e & (3u << 8)) != IW_KIND_BASE) { code = 0; return; }
    br.bits(e & 15u);
    L = iw_val(e) + br.bits((e >> 4) & 15u);
    if (br.nbits < 32u) {
      const uint32_t w = iw_readlane(br.cur, br.widx);
      br.bb |= (uint64_t)w << br.nbits;
      br.nbits += 32u;
      if (++br.widx == 64u) { code = 6; return; }
    }
    uint32_t d = dist_tab[(uint32_t)br.bb & ((1u << IW_DIST_ROOT) - 1u)];
    if (!(d & 15u)) {                                                // a distance code longer than the first level: lanes 8.. of llim / ldel
      const uint32_t v16 = iw_brev((uint32_t)br.bb) >> 16;
      int k = 0;
      while (k < 15 - IW_DIST_ROOT && !(v16 < llim[8 + k])) ++k;
      if (k == 15 - IW_DIST_ROOT) { code = 4; return; }
      const uint32_t l = (uint32_t)(IW_DIST_ROOT + 1 + k), at = (v16 >> (16u - l)) + ldel[8 + k], sym = S.d_sorted[at < 32u ? at : 32u];
      if (sym > 29u) { code = 4; return; }
      d = iw_with_total(iw_entry_of<IW_DISTS>(sym), l);
    }
    br.bits(d & 15u);
    D = iw_val(d) + br.bits((d >> 4) & 15u);
    if (D < L || D > o.pos || o.pos + L > o.isize) { code = 5; return; }
    o.match(L, D);
  }
}
#else
// (timing experiments only, never set in the product build: IW_EXP 1 = the copy rounds do not wait for their loads, 2 = no copy at all)
#if defined(IW_EXP) && IW_EXP == 1
#define IW_EXP_WAIT
#else
#define IW_EXP_WAIT "s_waitcnt vmcnt(0)\n\t"
#endif
#if defined(IW_EXP) && IW_EXP == 2
#define IW_EXP_COPY
#else
#define IW_EXP_COPY "buffer_store_byte %[pdata], %[paddr], %[rsrc], 0 offen\n\t" "buffer_load_ubyte %[pdata], %[vt1], %[rsrc], 0 offen\n\t"
#endif
IW_DEV void iw_run(IwBits &br, IwOut &o, const IwLds &S, const IwLane<uint32_t> &llim, const IwLane<uint32_t> &ldel, uint32_t &e, uint32_t &code, uint32_t &L, uint32_t &D) {
  const uint32_t *lit_tab = S.lit_tab, *dist_tab = S.dist_tab;
  uint32_t ve, vn, vt0, vt1, vt2, vD, vL, vsrc, m0save;
  const uint32_t vlit = (uint32_t)reinterpret_cast<uintptr_t>(lit_tab);     // LDS byte addresses (low half of the flat address)
  const uint32_t vdist = (uint32_t)reinterpret_cast<uintptr_t>(dist_tab);
  const uint32_t vlane = threadIdx.x;
  asm volatile(
      // the bit buffer carries a sentinel bit above its valid bits and sits TWO BITS UP inside the loop (nbits <= 61 on entry): "high
      // word zero" is "fewer than 30 valid bits", and the low word's bits [10:2] / [9:2] ARE a table entry's byte offset -- one v_bfi
      // into the table's address (the tables sit on multiples of their size) where a scalar AND and a vector shift-add stood: one
      // scalar instruction less per lookup, on the port that binds
      "s_mov_b32 s94, 1\n\t"
      "s_mov_b32 s95, 0\n\t"
      "s_lshl_b64 s[94:95], s[94:95], %[nb]\n\t"
      "s_mov_b32 %[m0save], m0\n\t"
      "s_mov_b32 m0, %[pos]\n\t"
      "s_or_b64 s[90:91], s[90:91], s[94:95]\n\t"
      "s_lshl_b64 s[90:91], s[90:91], 2\n\t"                      // (inside the loop the buffer sits two bits up: see the lookups)
      "v_mov_b32_e32 %[vD], %[D]\n\t"
      "v_mov_b32_e32 %[vL], %[L]\n\t"
      "v_mov_b32_e32 %[ve], 0\n\t"
      "s_mov_b32 %[e], 0\n\t"
      "s_mov_b32 s97, 1\n\t"
      "s_sub_u32 %[wi], %[wi], 64\n\t"
      "s_branch L_iw_loop_%=\n"
      // (the match path sits IN FRONT of the loop's head: the position behind a copied match falls through into the next lookup, where it
      // took a branch -- one taken branch less per match on a scalar port that is 71 % busy)
      // ---- not a literal: a length code of the first-level table, or something for the caller.  What bounds this kernel is the
      // scalar issue port (scalar ALU + branches + waits: one a cycle per CU, 75 % busy -- profiles/r04/inflate_pmc_window.txt): the
      // length and the distance are put together on the vector unit (every lane the same value), the scalar unit only shifts the bit
      // buffer by the entry's "all bits" field and forms the distance table's index; the checks of a match share one branch.
      "L_iw_notlit_%=:\n\t"
      // a length whose extra bits the table's index held (IW_LEN_DONE: nearly every match): the entry's high half IS the length, its low
      // four bits what the symbol consumes -- one vector instruction where the plain format below takes five
      "s_bitcmp1_b32 %[e], 15\n\t"
      "s_cbranch_scc0 L_iw_lenx_%=\n"
      "L_iw_lendone_%=:\n\t"
      "v_lshrrev_b32_e32 %[vL], 16, %[ve]\n\t"
      "s_and_b32 s95, %[e], 15\n\t"
      "s_lshr_b64 s[90:91], s[90:91], s95\n\t"
      "v_bfi_b32 %[vt0], %[vmdist], s90, %[vdist]\n\t"
      "ds_read_b32 %[ve], %[vt0]\n\t"
      "s_cmp_eq_u32 s91, 0\n\t"
      "s_cbranch_scc1 L_iw_refill2_%=\n"
      // (32 more bits, if they were needed, have come in on top: the entry being read stands) the distance code's fields
      "L_iw_have2_%=:\n\t"
      "v_add_u32_e32 %[vt2], m0, %[vL]\n\t"
      "s_waitcnt lgkmcnt(0)\n"
      "L_iw_have2b_%=:\n\t"
      "v_and_b32_e32 %[vn], 15, %[ve]\n\t"
      "v_bfe_u32 %[vt1], %[ve], 4, 4\n\t"
      "v_lshrrev_b32_e64 %[vt0], %[vn], s90\n\t"
      "v_bfe_u32 %[vt0], %[vt0], 2, %[vt1]\n\t"
      "v_add_u32_sdwa %[vD], %[vt0], %[ve] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
      "v_readfirstlane_b32 s92, %[ve]\n\t"
      "v_readfirstlane_b32 s93, %[vt2]\n\t"
      "s_bfe_u32 s92, s92, 0x5000a\n\t"                           // the bits the distance takes in all (on the scalar unit: the vector one is the busier now)
      // the fast copy takes a first-level distance code (the entry of any other is 0: D = 0 < L), D >= L, D <= pos: each
      // difference wraps to a value with the sign bit when its condition fails.  (pos + L <= isize is NOT checked here any more:
      // the block's descriptor drops every store behind isize, the flush of a window and the end of the stream compare the
      // position with isize -- a stream that is too long is refused a little later, and a vector instruction per match is gone)
      "v_sub_u32_e32 %[vsrc], m0, %[vD]\n\t"
      "v_sub_u32_e32 %[vt0], %[vD], %[vL]\n\t"
      "v_or_b32_e32 %[vt0], %[vt0], %[vsrc]\n\t"
      "s_andn2_b32 s94, m0, 63\n\t"
      "v_add_u32_e32 %[vt1], %[vL], %[vsrc]\n\t"
      "v_cmp_gt_i32_e32 vcc, 0, %[vt0]\n\t"
      "s_cbranch_vccnz L_iw_hard_%=\n\t"
      "s_lshr_b64 s[90:91], s[90:91], s92\n\t"
      // Staged literals lie in [W, pos), W = pos & ~63, and stay staged across matches: they are stored when the position leaves
      // the window (flushing before every match was 3 scalar + 5 vector instructions and a store per match, in data with a match
      // every 8.7 bytes).  A match whose source reaches into the window (pos - D + L > W) has them stored first.
      "v_cmp_lt_u32_e32 vcc, s94, %[vt1]\n\t"
      "s_cbranch_vccnz L_iw_flushfirst_%=\n"
      // the first 64 bytes: store what the previous match loaded, load this match's bytes (a lane beyond L loads a byte nobody
      // uses -- the descriptor bounds it -- and its store address is out of range)
      "L_iw_copy_%=:\n\t"
      "v_cmp_gt_u32_e32 vcc, %[vL], %[vlane]\n\t"
      "v_add_u32_e32 %[vt1], %[vsrc], %[vlane]\n\t"
      "v_add_u32_e32 %[vt0], m0, %[vlane]\n\t"
      IW_EXP_WAIT
      IW_EXP_COPY
      "v_cndmask_b32_e32 %[paddr], %[voob], %[vt0], vcc\n\t"
      "s_xor_b32 s92, s93, m0\n\t"
      "s_cmp_gt_u32 s92, 63\n\t"
      "s_cbranch_scc1 L_iw_cross_%=\n\t"
      "s_mov_b32 m0, s93\n"
      // ---- next symbol: first-level literal/length lookup (the refill sits out of line).  >= 30 valid bits here; a first-level
      // code has <= 9: three literals are decoded per check (21, 12 bits left for the second and third lookup)
      "L_iw_loop_%=:\n\t"
      "s_cmp_eq_u32 s91, 0\n\t"
      "s_cbranch_scc1 L_iw_refill_%=\n"
      "L_iw_have_%=:\n\t"
      "v_bfi_b32 %[vt0], %[vmlit], s90, %[vlit]\n\t"
      "ds_read_b32 %[ve], %[vt0]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[e], %[ve]\n\t"
      "s_bitcmp1_b32 %[e], 31\n\t"
      "s_cbranch_scc0 L_iw_notlit_%=\n\t"
      // a literal: the shift takes the code length from the entry's low six bits (a literal has no extra bits); the ENTRY goes
      // into the staging register, lane = position mod 64 (the byte is its bits 23:16: shifted down when the register is stored)
      "s_lshr_b64 s[90:91], s[90:91], %[e]\n\t"
      "v_writelane_b32 %[pend], %[e], m0\n\t"
      "s_add_u32 m0, m0, 1\n\t"
      "s_and_b32 s92, m0, 63\n\t"
      "s_cbranch_scc0 L_iw_full_%=\n\t"
      "v_bfi_b32 %[vt0], %[vmlit], s90, %[vlit]\n\t"
      "ds_read_b32 %[ve], %[vt0]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[e], %[ve]\n\t"
      "s_bitcmp1_b32 %[e], 31\n\t"
      "s_cbranch_scc0 L_iw_second_%=\n\t"
      // a literal: the shift takes the code length from the entry's low six bits (a literal has no extra bits); the ENTRY goes
      // into the staging register, lane = position mod 64 (the byte is its bits 23:16: shifted down when the register is stored)
      "s_lshr_b64 s[90:91], s[90:91], %[e]\n\t"
      "v_writelane_b32 %[pend], %[e], m0\n\t"
      "s_add_u32 m0, m0, 1\n\t"
      "s_and_b32 s92, m0, 63\n\t"
      "s_cbranch_scc0 L_iw_full_%=\n\t"
      "v_bfi_b32 %[vt0], %[vmlit], s90, %[vlit]\n\t"
      "ds_read_b32 %[ve], %[vt0]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 %[e], %[ve]\n\t"
      "s_bitcmp1_b32 %[e], 31\n\t"
      "s_cbranch_scc0 L_iw_loop_%=\n"
      // a literal: the shift takes the code length from the entry's low six bits (a literal has no extra bits); the ENTRY goes
      // into the staging register, lane = position mod 64 (the byte is its bits 23:16: shifted down when the register is stored)
      "L_iw_lit3_%=:\n\t"
      "s_lshr_b64 s[90:91], s[90:91], %[e]\n\t"
      "v_writelane_b32 %[pend], %[e], m0\n\t"
      "s_add_u32 m0, m0, 1\n\t"
      "s_and_b32 s92, m0, 63\n\t"
      "s_cbranch_scc1 L_iw_loop_%=\n"
      // the position has left a 64-byte window: one store of the literals staged in it (the lanes whose entry has bit 31)
      "L_iw_full_%=:\n\t"
      "s_sub_u32 s94, m0, 64\n\t"
      "v_cmp_gt_i32_e32 vcc, 0, %[pend]\n\t"
      "v_or_b32_e32 %[vt1], s94, %[vlane]\n\t"
      "v_lshrrev_b32_e32 %[vn], 16, %[pend]\n\t"
      "v_cndmask_b32_e32 %[vt1], %[voob], %[vt1], vcc\n\t"
      "buffer_store_byte %[vn], %[vt1], %[rsrc], 0 offen\n\t"
      "v_mov_b32_e32 %[pend], 0\n\t"
      "s_cmp_gt_u32 m0, %[isize]\n\t"
      "s_cbranch_scc0 L_iw_loop_%=\n\t"
      "s_mov_b32 %[code], 7\n\t"
      "s_branch L_iw_end_%=\n"
      // a length with resolved extra bits in the SECOND place goes straight on with the entry in hand: >= 21 valid bits were there, <= 9 go,
      // and the distance table's index -- formed in front of the refill check -- needs 8 of the 12 left.  (Not from the third place:
      // 12 - 9 bits are not an index.)  A match behind one literal, the commonest sequence of a BAM block, no longer pays a second lookup.
      "L_iw_second_%=:\n\t"
      "s_bitcmp1_b32 %[e], 15\n\t"
      "s_cbranch_scc1 L_iw_lendone_%=\n\t"
      "s_branch L_iw_loop_%=\n"
      // ---- 32 more bits into the buffer (every fourth symbol or so): the sentinel's position is the bit count
      "L_iw_refill_%=:\n\t"
      "s_flbit_i32_b32 s94, s90\n\t"
      "v_readlane_b32 s96, %[cur], %[wi]\n\t"                   // (the lane select is the register's low six bits: wi counts -64 .. -1)
      "s_sub_u32 s94, 31, s94\n\t"
      "s_bitset0_b32 s90, s94\n\t"
      "s_lshl_b64 s[92:93], s[96:97], s94\n\t"                    // s97 = 1 throughout: the new sentinel
      "s_or_b64 s[90:91], s[90:91], s[92:93]\n\t"
      "s_add_u32 %[wi], %[wi], 1\n\t"                           // carries when the window's 64th dword has gone in
      "s_cbranch_scc0 L_iw_have_%=\n\t"
      "s_mov_b32 %[code], 1\n\t"
      "s_branch L_iw_end_%=\n"
      // the match ends in another window: more rounds of the copy if it is longer than 64 bytes, then the literals staged in
      // this window are stored
      "L_iw_cross_%=:\n\t"
      "s_sub_u32 s95, s93, m0\n\t"
      "s_mov_b32 s94, 64\n\t"
      "s_cmp_gt_u32 s95, 64\n\t"
      "s_cbranch_scc0 L_iw_copied_%=\n"
      "L_iw_round_%=:\n\t"
      "v_add_u32_e32 %[vt0], s94, %[vlane]\n\t"
      "v_cmp_gt_u32_e32 vcc, s95, %[vt0]\n\t"
      "v_add_u32_e32 %[vt1], %[vsrc], %[vt0]\n\t"
      "v_add_u32_e32 %[vt0], m0, %[vt0]\n\t"
      IW_EXP_WAIT
      IW_EXP_COPY
      "v_cndmask_b32_e32 %[paddr], %[voob], %[vt0], vcc\n\t"
      "s_add_u32 s94, s94, 64\n\t"
      "s_cmp_lt_u32 s94, s95\n\t"
      "s_cbranch_scc1 L_iw_round_%=\n"
      "L_iw_copied_%=:\n\t"
      "s_andn2_b32 s94, m0, 63\n\t"
      "v_cmp_gt_i32_e32 vcc, 0, %[pend]\n\t"
      "v_or_b32_e32 %[vt1], s94, %[vlane]\n\t"
      "v_lshrrev_b32_e32 %[vn], 16, %[pend]\n\t"
      "v_cndmask_b32_e32 %[vt1], %[voob], %[vt1], vcc\n\t"
      "buffer_store_byte %[vn], %[vt1], %[rsrc], 0 offen\n\t"
      "v_mov_b32_e32 %[pend], 0\n\t"
      "s_mov_b32 m0, s93\n\t"
      "s_branch L_iw_loop_%=\n"
      "L_iw_flushfirst_%=:\n\t"
      "v_cmp_gt_i32_e32 vcc, 0, %[pend]\n\t"
      "v_or_b32_e32 %[vt1], s94, %[vlane]\n\t"
      "v_lshrrev_b32_e32 %[vn], 16, %[pend]\n\t"
      "v_cndmask_b32_e32 %[vt1], %[voob], %[vt1], vcc\n\t"
      "buffer_store_byte %[vn], %[vt1], %[rsrc], 0 offen\n\t"
      "v_mov_b32_e32 %[pend], 0\n\t"
      "s_branch L_iw_copy_%=\n"
      // ---- rarer paths
      "L_iw_refill2_%=:\n\t"
      "s_flbit_i32_b32 s94, s90\n\t"
      "v_readlane_b32 s96, %[cur], %[wi]\n\t"                   // (the lane select is the register's low six bits: wi counts -64 .. -1)
      "s_sub_u32 s94, 31, s94\n\t"
      "s_bitset0_b32 s90, s94\n\t"
      "s_lshl_b64 s[92:93], s[96:97], s94\n\t"                    // s97 = 1 throughout: the new sentinel
      "s_or_b64 s[90:91], s[90:91], s[92:93]\n\t"
      "s_add_u32 %[wi], %[wi], 1\n\t"                           // carries when the window's 64th dword has gone in
      "s_cbranch_scc0 L_iw_have2_%=\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %[code], 6\n\t"
      "s_branch L_iw_end_%=\n"
      // a check failed: a distance code outside the first-level table (nothing of it consumed), or a match for the caller
      "L_iw_hard_%=:\n\t"
      "v_cmp_eq_u32_e32 vcc, 0, %[vn]\n\t"
      "s_cbranch_vccnz L_iw_longd_%=\n\t"
      "s_lshr_b64 s[90:91], s[90:91], s92\n\t"
      "s_mov_b32 %[code], 5\n\t"
      "s_branch L_iw_end_%=\n"
      // a distance code longer than the first level (nothing of it consumed; >= 32 valid bits): lanes 8 + k of vlim / vdel hold limit and
      // delta of length DROOT + 1 + k.  The symbol's table entry is put together here (RFC 1951 3.2.5 in closed form, like iw_entry_of)
      // and the distance decoded from it by the code above; an invalid code (30, 31, none at all) leaves for the caller as before.
      "L_iw_longd_%=:\n\t"
      "s_lshr_b32 s92, s90, 2\n\t"
      "s_brev_b32 s92, s92\n\t"
      "s_lshr_b32 s92, s92, 16\n\t"
      "v_cmp_lt_u32_e32 vcc, s92, %[vlim]\n\t"
      "s_lshr_b32 s93, vcc_lo, 8\n\t"
      "s_and_b32 s93, s93, 0x7f\n\t"
      "s_cbranch_scc0 L_iw_exit4_%=\n\t"
      "s_ff1_i32_b32 s93, s93\n\t"
      "s_add_u32 s94, s93, 8\n\t"
      "s_sub_u32 s95, %[ndlong], s93\n\t"
      "v_readlane_b32 s94, %[vdel], s94\n\t"
      "s_lshr_b32 s92, s92, s95\n\t"
      "s_add_u32 s92, s92, s94\n\t"
      "s_min_u32 s92, s92, 32\n\t"
      "v_lshl_add_u32 %[vt0], s92, 1, %[vlit]\n\t"
      "ds_read_u16 %[ve], %[vt0] offset:%[dsoff]\n\t"
      "s_add_u32 s93, s93, %[droot1]\n\t"                          // the code's length
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 s92, %[ve]\n\t"                       // the distance symbol
      "s_cmp_gt_u32 s92, 29\n\t"
      "s_cbranch_scc1 L_iw_exit4_%=\n\t"
      "s_add_u32 s94, s92, 1\n\t"                                 // symbols 0..3: distance 1 + s, no extra bits
      "s_mov_b32 s95, 0\n\t"
      "s_cmp_lt_u32 s92, 4\n\t"
      "s_cbranch_scc1 L_iw_longd_entry_%=\n\t"
      "s_sub_u32 s95, s92, 2\n\t"
      "s_lshr_b32 s95, s95, 1\n\t"                                // extra bits e = (s - 2) / 2
      "s_and_b32 s94, s92, 1\n\t"
      "s_add_u32 s94, s94, 2\n\t"
      "s_lshl_b32 s94, s94, s95\n\t"
      "s_add_u32 s94, s94, 1\n"                                    // base ((2 + s % 2) << e) + 1
      "L_iw_longd_entry_%=:\n\t"
      "s_lshl_b32 s94, s94, 16\n\t"                               // [30:16] base | [14:10] length + extra | [9:8] kind 1 | [7:4] extra | [3:0] length
      "s_add_u32 s92, s93, s95\n\t"
      "s_lshl_b32 s92, s92, 10\n\t"
      "s_lshl_b32 s95, s95, 4\n\t"
      "s_or_b32 s94, s94, s92\n\t"
      "s_or_b32 s94, s94, s95\n\t"
      "s_or_b32 s94, s94, s93\n\t"
      "s_bitset1_b32 s94, 8\n\t"
      "v_mov_b32_e32 %[ve], s94\n\t"
      "v_add_u32_e32 %[vt2], m0, %[vL]\n\t"
      "s_branch L_iw_have2b_%=\n"
      "L_iw_exit4_%=:\n\t"
      "s_mov_b32 %[code], 4\n\t"
      "s_branch L_iw_end_%=\n"
      // ---- a length code of the first level in the plain format (its extra bits reach past the index): length = value + extra bits
      "L_iw_lenx_%=:\n\t"
      "s_bfe_u32 s92, %[e], 0x20008\n\t"
      "s_cmp_eq_u32 s92, 1\n\t"
      "s_cbranch_scc0 L_iw_other_%=\n\t"
      "v_and_b32_e32 %[vt1], 15, %[ve]\n\t"
      "s_bfe_u32 s95, %[e], 0x5000a\n\t"
      "v_bfe_u32 %[vt0], %[ve], 4, 4\n\t"
      "v_lshrrev_b32_e64 %[vL], %[vt1], s90\n\t"
      "s_lshr_b64 s[90:91], s[90:91], s95\n\t"
      "v_bfe_u32 %[vL], %[vL], 2, %[vt0]\n\t"
      "v_bfi_b32 %[vt0], %[vmdist], s90, %[vdist]\n\t"
      "v_add_u32_sdwa %[vL], %[vL], %[ve] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"   // + the entry's value (its high half)
      "ds_read_b32 %[ve], %[vt0]\n\t"
      "s_cmp_eq_u32 s91, 0\n\t"
      "s_cbranch_scc1 L_iw_refill2_%=\n\t"
      "s_branch L_iw_have2_%=\n"
      // ---- not a length code of the first level either.  A length field of 0 is a code LONGER than the first level: one symbol in
      // twenty-five of a level-6 BAM block, nearly all of them literals (profiles/r05/inflate_group/README.md), and leaving the loop
      // for each cost ~3 symbols' time.  Lane k of vlim / vdel holds limit and delta of length ROOT + 1 + k: one compare finds the
      // length, one lookup the symbol; a literal joins the LAST of the three literal blocks with an entry made up for it (the one that
      // goes back to the refill check: up to 15 bits are gone, the next lookups' nine are no longer certain), anything else (a long
      // length code, the end of the block, no code at all) leaves for the caller with nothing consumed -- as before.
      "L_iw_other_%=:\n\t"
      "s_and_b32 s92, %[e], 15\n\t"
      "s_cbranch_scc1 L_iw_exit0_%=\n\t"
      "s_lshr_b32 s92, s90, 2\n\t"
      "s_brev_b32 s92, s92\n\t"
      "s_lshr_b32 s92, s92, 16\n\t"
      "v_cmp_lt_u32_e32 vcc, s92, %[vlim]\n\t"
      "s_and_b32 s93, vcc_lo, 0x3f\n\t"                           // (lanes 8.. are the distance code's)
      "s_cbranch_scc0 L_iw_exit0_%=\n\t"
      "s_ff1_i32_b32 s93, s93\n\t"
      "s_sub_u32 s95, %[nlong], s93\n\t"
      "v_readlane_b32 s94, %[vdel], s93\n\t"
      "s_lshr_b32 s92, s92, s95\n\t"
      "s_add_u32 s92, s92, s94\n\t"
      "s_min_u32 s92, s92, 0x120\n\t"
      "v_lshl_add_u32 %[vt0], s92, 1, %[vlit]\n\t"
      "ds_read_u16 %[ve], %[vt0] offset:%[soff]\n\t"
      "s_add_u32 s93, s93, %[root1]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_readfirstlane_b32 s92, %[ve]\n\t"
      "s_cmpk_lt_u32 s92, 0x100\n\t"
      "s_cbranch_scc0 L_iw_exit0_%=\n\t"
      "s_lshl_b32 s92, s92, 16\n\t"
      "s_or_b32 %[e], s92, s93\n\t"
      "s_bitset1_b32 %[e], 31\n\t"
      "s_branch L_iw_lit3_%=\n"
      "L_iw_exit0_%=:\n\t"
      "s_mov_b32 %[code], 0\n"
      // the caller may read or write the output itself: nothing stays pending; position, bit count (the sentinel's place) and the
      // vector-held length and distance go back to their registers, the sentinel is taken out
      "L_iw_end_%=:\n\t"
      "s_add_u32 %[wi], %[wi], 64\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "buffer_store_byte %[pdata], %[paddr], %[rsrc], 0 offen\n\t"
      "v_mov_b32_e32 %[paddr], %[voob]\n\t"
      "s_lshr_b64 s[90:91], s[90:91], 2\n\t"
      "s_flbit_i32_b64 s92, s[90:91]\n\t"
      "s_mov_b32 %[pos], m0\n\t"
      "s_mov_b32 m0, %[m0save]\n\t"
      "s_sub_u32 %[nb], 63, s92\n\t"
      "v_readfirstlane_b32 %[D], %[vD]\n\t"
      "v_readfirstlane_b32 %[L], %[vL]\n\t"
      "s_bitset0_b64 s[90:91], %[nb]\n\t"
      : "+{s[90:91]}"(br.bb), [nb] "+s"(br.nbits), [wi] "+s"(br.widx), [pos] "+s"(o.pos), [pend] "+v"(o.pend.x),
        [pdata] "+v"(o.pdata.x), [paddr] "+v"(o.paddr.x), [e] "=&s"(e), [code] "=&s"(code), [L] "+s"(L), [D] "+s"(D), [m0save] "=&s"(m0save),
        [ve] "=&v"(ve), [vn] "=&v"(vn), [vt0] "=&v"(vt0), [vt1] "=&v"(vt1), [vt2] "=&v"(vt2), [vD] "=&v"(vD), [vL] "=&v"(vL), [vsrc] "=&v"(vsrc)
      : [cur] "v"(br.cur.x), [vlit] "v"(vlit), [vdist] "v"(vdist), [vlane] "v"(vlane), [voob] "v"(IW_OOB), [isize] "s"(o.isize), [rsrc] "s"(o.out.r),
        [vmlit] "v"(((1u << IW_LIT_ROOT) - 1u) << 2), [vmdist] "v"(((1u << IW_DIST_ROOT) - 1u) << 2), [vlim] "v"(llim.x), [vdel] "v"(ldel.x), [nlong] "i"(15 - IW_LIT_ROOT), [root1] "i"(IW_LIT_ROOT + 1),
        [soff] "i"(offsetof(IwLds, ll_sorted)), [dsoff] "i"(offsetof(IwLds, d_sorted)), [ndlong] "i"(15 - IW_DIST_ROOT), [droot1] "i"(IW_DIST_ROOT + 1)
      : "s92", "s93", "s94", "s95", "s96", "s97", "vcc", "scc", "memory");   // (m0: holds the output position inside; a reserved register cannot be listed as clobbered, so it is saved and restored)
}
#endif

// Inflate the raw DEFLATE stream comp[off, off + clen) into out[0, isize).  comp[0, readable) may be loaded (readable is a
// multiple of 4 and >= off + clen).  Returns 0 or IW_ERR_* flags; never touches memory outside comp[0, readable) and
// out[0, isize).
IW_DEV int iw_inflate(const uint8_t *comp, uint64_t off, uint32_t clen, uint64_t readable, uint8_t *out, uint32_t isize, IwLds &S) {
  IwBits br;
  br.init(comp, off, readable);
  uint64_t stream_bits = 8ull * clen;     // bits of the stream still ahead of this reader's origin (stored blocks restart the reader)
  uint64_t origin = off & ~(uint64_t)3;   // where the reader's origin sits in comp
  IwOut o;
  o.out = iw_make_buf(out, isize);
  o.isize = isize; o.pos = 0;
  IW_FOR_LANES { o.pend[lane] = 0; o.pdata[lane] = 0; o.paddr[lane] = IW_OOB; }
  for (;;) {
    br.refill();
    const uint32_t bfinal = br.bits(1), btype = br.bits(2);
    if (btype == 3u) return IW_ERR_DATA;
    if (btype == 0u) {
      br.bits(br.nbits & 7u);                                   // to the next byte boundary
      br.refill();
      const uint32_t len = br.bits(16);
      br.refill();
      const uint32_t nlen = br.bits(16);
      if ((len ^ 0xffffu) != nlen) return IW_ERR_DATA;
      const uint64_t used = br.consumed();
      if (used + 8ull * len > stream_bits) return IW_ERR_DATA;  // the stored bytes reach past the stream
      if (o.pos + len > isize) return IW_ERR_SIZE;
      const uint32_t p = br.byte_pos();
      o.raw(br.in, p, len);
      stream_bits -= used + 8ull * len;
      const uint64_t noff = origin + p + len;
      origin = noff & ~(uint64_t)3;
      br.init(comp, noff, readable);
    } else {
      uint32_t hlit, hdist;
      if (btype == 1u) {
        hlit = 288; hdist = 32;
        for (uint32_t s0 = 0; s0 < 320u; s0 += 64u) {
          IW_FOR_LANES {
            const uint32_t s = s0 + (uint32_t)lane;
            S.lens[s] = (uint8_t)(s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : s < 288u ? 8 : 5);
          }
        }
      } else {
        hlit = br.bits(5) + 257u;
        hdist = br.bits(5) + 1u;
        const uint32_t hclen = br.bits(4) + 4u;
        if (hlit > 286u || hdist > 30u) return IW_ERR_DATA;
        uint64_t clv = 0;                                       // 3 bits per code-length symbol
        for (uint32_t i = 0; i < hclen; ++i) {
          br.refill();
          clv |= (uint64_t)br.bits(3) << (3u * iw_cl_order(i));
        }
        IwLane<uint32_t> cll[1];
        IW_FOR_LANES { cll[0][lane] = lane < 19 ? (uint32_t)(clv >> (3 * (lane < 19 ? lane : 0))) & 7u : 0u; }
        if (!iw_build<1, IW_CL_ROOT, IW_CODES>(cll, S.cl.cl_tab, S.cl.cl_sorted, 31u, S.cl.cl_limit, S.cl.cl_delta, S.cl.cl_offs)) return IW_ERR_DATA;
        const uint32_t n = hlit + hdist;
        uint32_t i = 0, prev = 0;
        while (i < n) {
          br.refill();
          const uint32_t e = IW_U(S.cl.cl_tab[(uint32_t)br.bb & ((1u << IW_CL_ROOT) - 1u)]);
          if (!(e & 15u)) return IW_ERR_DATA;
          br.bits(e & 15u);
          const uint32_t s = iw_val(e);
          uint32_t rep, val;
          if (s < 16u) { rep = 1; val = s; }
          else if (s == 16u) { if (!i) return IW_ERR_DATA; rep = 3u + br.bits(2); val = prev; }
          else if (s == 17u) { rep = 3u + br.bits(3); val = 0; }
          else { rep = 11u + br.bits(7); val = 0; }
          if (i + rep > n) return IW_ERR_DATA;
          for (uint32_t r0 = 0; r0 < rep; r0 += 64u) {
            IW_FOR_LANES {
              const uint32_t r = r0 + (uint32_t)lane;
              S.lens[r < rep ? i + r : 320u] = (uint8_t)val;
            }
          }
          i += rep;
          prev = val;
        }
        IW_SYNC();
        if (IW_U(S.lens[256]) == 0u) return IW_ERR_DATA;        // no end-of-block code (inflate.c: "missing end-of-block")
      }
      IW_SYNC();
      {
        IwLane<uint32_t> ll[5];
        IW_FOR_LANES {
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const uint32_t s = (uint32_t)(j * 64 + lane);
            const uint32_t v = S.lens[s < 320u ? s : 320u];
            ll[j][lane] = s < hlit ? v : 0u;
          }
        }
        IwLane<uint32_t> dl[1];
        IW_FOR_LANES {
          const uint32_t v = S.lens[(uint32_t)lane < 32u ? hlit + (uint32_t)lane : 320u];
          dl[0][lane] = (uint32_t)lane < hdist ? v : 0u;
        }
        IW_SYNC();                                              // (the code-length tables alias the distance table)
        if (!iw_build<5, IW_LIT_ROOT, IW_LENS>(ll, S.lit_tab, S.ll_sorted, 288u, S.ll_limit, S.ll_delta, S.ll_offs)) return IW_ERR_DATA;
        if (!iw_build<1, IW_DIST_ROOT, IW_DISTS>(dl, S.dist_tab, S.d_sorted, 32u, S.d_limit, S.d_delta, S.d_offs)) return IW_ERR_DATA;
      }
#if defined(IW_EXP) && IW_EXP == 3
      return 0;      // (timing experiment: header parse + table build only)
#endif
      // The symbol loop: literals and plain matches in iw_run, what it hands back here.
      uint32_t why = 0;      // IW_ERR_* when the loop ends for another reason than the end-of-block code
      uint32_t L = 0, D = 0;
      // iw_run's long codes: lane k holds limit and delta of literal/length code length ROOT + 1 + k, lane 8 + k those of distance
      // code length DROOT + 1 + k
      static_assert(15 - IW_LIT_ROOT <= 8 && 15 - IW_DIST_ROOT <= 7, "lanes 0..7 and 8..14");
      IwLane<uint32_t> llim, ldel;
      IW_FOR_LANES {
        const bool is_l = lane < 15 - IW_LIT_ROOT, is_d = lane >= 8 && lane < 8 + 15 - IW_DIST_ROOT;
        const int il = is_l ? IW_LIT_ROOT + 1 + lane : 0, id = is_d ? IW_DIST_ROOT + 1 + lane - 8 : 0;
        const uint32_t a = S.ll_limit[il], b = S.ll_delta[il], c = S.d_limit[id], d = S.d_delta[id];
        llim[lane] = is_l ? a : is_d ? c : 0u;
        ldel[lane] = is_l ? b : is_d ? d : 0u;
      }
      for (;;) {
        uint32_t e, code;
        iw_run(br, o, S, llim, ldel, e, code, L, D);
        // (The loop no longer compares a match's end with isize: stores behind isize are dropped by the block's descriptor.  What
        // BOUNDS it is this: it comes back here every 256 bytes of input at the latest -- code 1 / 6 --, and a position past isize ends
        // the stream.  Without that a truncated stream, read as zeros past its end, whose zero bits are a match would never stop.)
        if ((code == 1u || code == 6u) && o.pos > isize) { why = IW_ERR_SIZE; break; }
        if (code == 1u) { br.rotate(); continue; }
        if (code == 7u) { why = IW_ERR_SIZE; break; }
        if (code == 0u) {
          if (!(e & 15u)) {
            e = iw_slow<IW_LENS>(br.bb, S.ll_limit, S.ll_delta, S.ll_sorted, 288u, IW_LIT_ROOT);
            if (!e) { why = IW_ERR_DATA; break; }
          }
          if ((e & (3u << 8)) == IW_KIND_BAD) { why = IW_ERR_DATA; break; }   // a length symbol that is never valid (286, 287)
          br.bits(e & 15u);
          const uint32_t kind = e & (3u << 8);
          if (kind == 0u) {                                     // a literal with a code longer than the first-level table
            iw_writelane(o.pend, o.pos & 63u, e);
            ++o.pos;
            if ((o.pos & 63u) == 0u) o.flush();
            continue;
          }
          if (kind == IW_KIND_EOB) break;
          if (kind != IW_KIND_BASE) { why = IW_ERR_DATA; break; }
          L = iw_val(e) + br.bits((e >> 4) & 15u);
          br.refill();
          code = 4u;
        }
        if (code == 6u) { br.rotate(); code = 4u; }
        if (code == 4u) {                                       // distance code: first-level table or the canonical search
          uint32_t d = IW_U(S.dist_tab[(uint32_t)br.bb & ((1u << IW_DIST_ROOT) - 1u)]);
          if (!(d & 15u)) {
            d = iw_slow<IW_DISTS>(br.bb, S.d_limit, S.d_delta, S.d_sorted, 32u, IW_DIST_ROOT);
            if (!d) { why = IW_ERR_DATA; break; }
          }
          br.bits(d & 15u);
          D = iw_val(d) + br.bits((d >> 4) & 15u);
        }
        if (L < 3u || D == 0u || D > o.pos) { why = IW_ERR_DATA; break; }
        if (o.pos + L > isize) { why = IW_ERR_SIZE; break; }
        o.match(L, D);
      }
      if (why) return (int)why;
      if (o.pos > isize) return IW_ERR_SIZE;
    }
    if (bfinal) break;
  }
  o.flush();
  if (br.consumed() > stream_bits) return IW_ERR_DATA;          // the decoder read past the end of the stream
  if (o.pos != isize) return IW_ERR_SIZE;
  return 0;
}

}  // namespace strl
