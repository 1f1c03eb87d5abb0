// inflate_group.h -- DEFLATE (RFC 1951) decoder of bgzf.hip, second form: G LANES per BGZF block, 64 / G blocks per wavefront
// (extract.nim:275,289 read the BAM through htslib's single-threaded inflate).
//
// inflate_wave.h gives a whole wave to one stream: the symbol loop is wave-uniform and runs on the scalar unit plus ONE useful
// lane of the vector unit, and a CU issues one scalar-port and one wave64 vector instruction a cycle whatever the waves hold:
// both ports end up two thirds busy at 87 GB/s (profiles/r04/inflate_pmc_*.txt) -- every vector instruction spends 4 cycles of
// a SIMD on one lane's worth of work.  Round 2 had the other extreme (a stream per lane: 64 table sets per wave, one wave per
// CU, byte stores that never coalesce: 6.6 GB/s).  This is the middle:
//   * a GROUP of G = 8 lanes owns a stream, a wave decodes 8 streams.  All G lanes of a group run the same decoder on the same
//     values (replicated, so no cross-lane traffic at all: the state is plain per-thread registers, the code plain per-thread
//     C++ without a line of assembly); they differ only in which byte of a match they copy, which symbols of a code-length
//     list they sort and which table entries they fill.  One vector instruction now advances eight streams.
//   * what made "a stream per lane" impossible was LDS: here a stream has 1.7 KB (16-bit first-level literal/length entries with
//     the length's extra bits resolved where code + extra bits fit the 9 index bits; 7-bit distance table; the long lengths'
//     bounds; a 64-byte input ring -- the canonical search's sorted symbols live in a global workspace), a wave 13 KB: twelve
//     waves per CU by LDS = 96 streams per CU (measured: eight waves are faster than twelve, profiles/r05/inflate_group/).
//   * a step of a group is "up to two literals and a match, or three literals", written without branches between the groups'
//     cases (a group whose next symbol is not a literal looks the same entry up again and consumes nothing); the match half is
//     decoded into temporaries and committed only when it is the common case (first-level codes, extra bits resolved, distance
//     within the output) -- anything else leaves the symbol to a general one-symbol routine on the next turn.
//   * input: the group's lanes load one dword each of a 4 G-byte window a window ahead (one coalesced request), park it in
//     the LDS ring, and every lane reads the dword it is about to need one refill ahead: no global latency in the symbol chain.
//   * output: a literal is one byte store of one lane; a match is copied G bytes a round, up to four rounds LOADED at the top
//     of a turn of the loop and STORED at its bottom, behind the lookups and stores of the literals that follow the match (the
//     L2 round trip passes under them; nothing in flight is carried around the loop: a register a load is still writing must
//     not meet the copies a compiler puts on a loop's back edge); periodic matches (distance < length) read byte k from
//     src + k mod D like inflate_wave.h: only bytes written before.  The ring's next window is requested and parked the same way.
//   * all global accesses go through bounds-checked buffer descriptors (the launch's compressed bytes, the launch's output,
//     the workspace) with 32-bit offsets; a lane that has nothing to load or store gets an offset outside them.  Per stream, the
//     decoder checks pos + L <= ISIZE and D <= pos itself: corrupt data never reaches a neighbour's bytes.
// The same source compiles for the host (STRL_EMU: G = 1, one "lane") so that the CPU-only test-suite runs the decoder logic
// against zlib; the product never runs that build.
#pragma once
#include <stdint.h>
#include "inflate_wave.h"   // IwBuf + iw_ld32 / iw_ld8 / iw_st8, IW_ERR_*, IW_OOB, iw_brev, iw_cl_order

#ifdef STRL_EMU
#define IG_SYNC() ((void)0)
namespace strl {
IW_DEV uint32_t ig_add_rtn(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
IW_DEV void ig_add(uint32_t *p, uint32_t v) { *p += v; }
IW_DEV void ig_min(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
IW_DEV uint32_t ig_alignbit(uint32_t hi, uint32_t lo, uint32_t n) { return n ? (lo >> n) | (hi << (32u - n)) : lo; }
typedef uint32_t ig_byte;
IW_DEV ig_byte ig_ld8_raw(const IwBuf &b, uint32_t off) { return iw_ld8(b, off); }
IW_DEV void ig_st8_raw(const IwBuf &b, uint32_t off, ig_byte v) { iw_st8(b, off, v); }
IW_DEV uint32_t ig_ld16(const IwBuf &b, uint32_t off) { uint16_t v = 0; if ((uint64_t)off + 2 <= b.n) memcpy(&v, b.p + off, 2); return v; }
IW_DEV void ig_st16(const IwBuf &b, uint32_t off, uint32_t v) { const uint16_t h = (uint16_t)v; if ((uint64_t)off + 2 <= b.n) memcpy(b.p + off, &h, 2); }
IW_DEV void ig_st32(const IwBuf &b, uint32_t off, uint32_t v) { if ((uint64_t)off + 4 <= b.n) memcpy(b.p + off, &v, 4); }
}  // namespace strl
#else
#define IG_SYNC() IW_SYNC()
namespace strl {
IW_DEV uint32_t ig_add_rtn(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }   // LDS: ds_add_rtn_u32
IW_DEV void ig_add(uint32_t *p, uint32_t v) { (void)atomicAdd(p, v); }            // ds_add_u32: nothing waited for
IW_DEV void ig_min(uint32_t *p, uint32_t v) { (void)atomicMin(p, v); }            // ds_min_u32
IW_DEV uint32_t ig_alignbit(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbit(hi, lo, n); }
// A byte on its way from one place of the output to another.  Nothing may compute on it -- whatever did would wait for the load
// long before the store needs it.  A byte-typed load does not survive that: the compiler zero-extends it, or packs four of them
// into one register, on the spot.  So the load is a DWORD load at the byte's own (unaligned) offset -- gfx950 takes unaligned
// buffer accesses; the descriptor is three bytes longer than the output so that its last bytes can be sources -- and the store
// writes the register's low byte.
typedef uint32_t ig_byte;
// (IG_EXP, timing experiments only, never set in the product build: 1 = the symbol loop's loads of match bytes return a constant,
//  2 = that and none of its stores happen)
#if defined(IG_EXP) && IG_EXP >= 1
IW_DEV ig_byte ig_ld8_raw(const IwBuf &, uint32_t off) { return off; }
#else
IW_DEV ig_byte ig_ld8_raw(const IwBuf &b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)off, 0, 0); }
#endif
#if defined(IG_EXP) && IG_EXP >= 2
IW_DEV void ig_st8_raw(const IwBuf &b, uint32_t off, ig_byte v) { if (off == 0x7ffffffeu) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, b.r, (int)off, 0, 0); }
#else
IW_DEV void ig_st8_raw(const IwBuf &b, uint32_t off, ig_byte v) { __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, b.r, (int)off, 0, 0); }
#endif
IW_DEV uint32_t ig_ld16(const IwBuf &b, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b16(b.r, (int)off, 0, 0); }
IW_DEV void ig_st16(const IwBuf &b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, b.r, (int)off, 0, 0); }
IW_DEV void ig_st32(const IwBuf &b, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, b.r, (int)off, 0, 0); }
}  // namespace strl
#endif

#if defined(STRL_EMU) && defined(IG_STATS)
extern "C" { extern unsigned long long ig_stats[16]; }
#define IG_STAT(i) (++ig_stats[i])
#else
#define IG_STAT(i) ((void)0)
#endif

namespace strl {

constexpr int IG_LIT_ROOT = 9, IG_DIST_ROOT = 7, IG_CL_ROOT = 7;
constexpr uint32_t IG_NONE = 0xffffffffu;

// First-level literal/length entry (u16), [3:0] = the bits the symbol consumes (0: not in the table):
//   bit 15     literal, byte in [11:4]
//   bit 14     length with its extra bits resolved from the index (code + extra bits <= 9): L in [12:4], [3:0] = code + extra bits
//   bit 13     length, extra bits still to read: length symbol - 257 in [8:4], [3:0] = code length
//   bit 12     end of block
//   else       [3:0] != 0: a symbol that takes part in the code and is never valid (286, 287)
// The canonical search hands out the same format (code lengths up to 15 fit the four bits; nothing is resolved there).
constexpr uint32_t IG_LIT = 0x8000u, IG_LEN = 0x4000u, IG_LENX = 0x2000u, IG_EOB = 0x1000u;
// First-level distance entry (u32): [3:0] code length (0: not in the table), [7:4] extra bits, [8] never valid (30, 31),
// [31:16] base distance.
constexpr uint32_t IG_DBAD = 0x100u;
constexpr uint32_t IG_LONG = 0x10000u;       // (symbol loop) the next symbol's code is not in the first-level literal/length table

IW_DEV uint32_t ig_len_extra(uint32_t c) { return (c < 8u || c >= 28u) ? 0u : (c - 4u) >> 2; }           // c = symbol - 257
IW_DEV uint32_t ig_len_base(uint32_t c) { return c < 8u ? 3u + c : c >= 28u ? 258u : ((4u + (c & 3u)) << ((c - 4u) >> 2)) + 3u; }
IW_DEV uint32_t ig_lit_entry(uint32_t s, uint32_t l, uint32_t index, int root) {
  if (s < 256u) return IG_LIT | (s << 4) | l;
  if (s == 256u) return IG_EOB | l;
  const uint32_t c = s - 257u;
  if (c > 28u) return 0x10u | l;
  const uint32_t eb = ig_len_extra(c);
  if (l + eb <= (uint32_t)root) return IG_LEN | ((ig_len_base(c) + ((index >> l) & ((1u << eb) - 1u))) << 4) | (l + eb);
  return IG_LENX | (c << 4) | l;
}
IW_DEV uint32_t ig_dist_entry(uint32_t s, uint32_t l) {
  if (s > 29u) return IG_DBAD | l;
  if (s < 4u) return ((1u + s) << 16) | l;
  const uint32_t e = (s - 2u) >> 1;
  return ((((2u + (s & 1u)) << e) + 1u) << 16) | (e << 4) | l;
}

// Where a sorted code lives -- the symbols ordered by (code length, symbol), and per code length 1..15 limit[] / delta[]
// (inflate_wave.h: iw_build).  The literal/length and the distance code keep theirs in a slice of a global WORKSPACE (1 KiB a
// stream): after the tables are filled only codes longer than the first level come back for them, and their 830 bytes were a
// third of a stream's LDS -- of what decides how many streams a CU holds.  The code-length code's (19 symbols) stay in LDS.
// Byte offsets: sorted (u16) at `so`, limit (u32) at `lo`, delta (i16) at `de`.
// The lengths ABOVE the first level keep limit >> (16 - l) (<= 2^15) and delta in LDS as well (`hi`, 4 bytes a length): a long
// code's search then costs one workspace load -- the symbol -- not one per length tried.
struct IgCodeW {                             // in the workspace
  IwBuf b;
  uint32_t so, lo, de;
  uint16_t *hi;                              // [2 (15 - root)]: end, delta of lengths root + 1 .. 15
  uint32_t root;
  uint32_t *nlit;                            // (literal/length code) [15 - root]: the slots of a length's LITERALS end here -- symbols of
                                             // one length are in symbol order, so a long code is a literal iff its slot is below this
  IW_DEV void sorted(uint32_t i, uint32_t s) const { ig_st16(b, so + 2u * i, s); }
  IW_DEV uint32_t sorted(uint32_t i) const { return ig_ld16(b, so + 2u * i); }
  IW_DEV void lim(uint32_t l, uint32_t limit, int32_t delta) const {
    ig_st32(b, lo + 4u * l, limit);
    ig_st16(b, de + 2u * l, (uint32_t)delta & 0xffffu);
    if (l > root) { hi[2u * (l - root - 1u)] = (uint16_t)(limit >> (16u - l)); hi[2u * (l - root - 1u) + 1u] = (uint16_t)delta; }
  }
  IW_DEV void slots(uint32_t l, uint32_t end) const { if (nlit && l > root) nlit[l - root - 1u] = end; }          // before the symbols are placed
  IW_DEV void placed(uint32_t l, uint32_t slot, uint32_t s) const { if (nlit && l > root && s >= 256u) ig_min(&nlit[l - root - 1u], slot); }
  IW_DEV uint32_t limit(uint32_t l) const { return iw_ld32(b, lo + 4u * l); }
  IW_DEV int32_t delta(uint32_t l) const { return (int32_t)(int16_t)ig_ld16(b, de + 2u * l); }
  IW_DEV bool within(uint32_t v16, uint32_t l) const { return (v16 >> (16u - l)) < hi[2u * (l - root - 1u)]; }   // l > root
  IW_DEV int32_t delta_hi(uint32_t l) const { return (int32_t)(int16_t)hi[2u * (l - root - 1u) + 1u]; }
};
struct IgCodeL {                             // in LDS
  uint16_t *s;
  uint32_t *li;
  int16_t *d;
  IW_DEV void sorted(uint32_t i, uint32_t v) const { s[i] = (uint16_t)v; }
  IW_DEV uint32_t sorted(uint32_t i) const { return s[i]; }
  IW_DEV void lim(uint32_t l, uint32_t limit, int32_t delta) const { li[l] = limit; d[l] = (int16_t)delta; }
  IW_DEV uint32_t limit(uint32_t l) const { return li[l]; }
  IW_DEV int32_t delta(uint32_t l) const { return d[l]; }
  IW_DEV bool within(uint32_t v16, uint32_t l) const { return v16 < li[l]; }
  IW_DEV int32_t delta_hi(uint32_t l) const { return d[l]; }
  IW_DEV void slots(uint32_t, uint32_t) const {}
  IW_DEV void placed(uint32_t, uint32_t, uint32_t) const {}
};
constexpr uint32_t IG_WORK_STRIDE = 1024, IG_W_LL = 0, IG_W_D = 576, IG_W_LL_LIM = 640, IG_W_D_LIM = 704, IG_W_LL_DEL = 768, IG_W_D_DEL = 800;

// windows of the input ring: the ring is topped up once a turn of the symbol loop, a turn refills twice at most -- two windows of
// eight dwords leave eight refills of slack; the host build's windows are one dword
template <int G> struct IgRing { static constexpr int W = G == 1 ? 4 : 2; };
// LDS of one stream: 1680 bytes at G = 8 -- twelve 8-stream waves in a CU's 160 KB.  What is alive when:
//   header parse      cl (the code-length code's table, lengths, sorted symbols: in dist_tab's place), lens (being written),
//                     rows / tots (the code-length code's build)
//   builds            lens (read), rows / tots (scratch)
//   table fill        lit_tab overwrites lens, rows, tots; dist_tab overwrites cl
//   symbol loop       lit_tab, dist_tab, ring
template <int G> struct IgLds {
  union {
    uint16_t lit_tab[1 << IG_LIT_ROOT];
    struct {
      uint8_t lens[320 + 8];                // code lengths: literal/length symbols, then the distance symbols
      uint32_t tots[16];                    // symbols per code length, then the first slot of each length
      uint32_t rows[G * 16];                // [lane][code length]: the lane's symbols of that length, then its next slot
    } b;
  };
  union {
    uint32_t dist_tab[1 << IG_DIST_ROOT];
    struct { uint8_t tab[1 << IG_CL_ROOT]; uint8_t lens[20]; uint16_t sorted[20]; uint32_t limit[16]; int16_t delta[16]; } cl;
  };
  uint16_t ll_hi[2 * (15 - IG_LIT_ROOT)], d_hi[2 * (15 - IG_DIST_ROOT)];   // IgCodeW::hi of the two codes
  uint32_t ll_nlit[15 - IG_LIT_ROOT];       // IgCodeW::nlit
  uint32_t ring[IgRing<G>::W * G];          // windows of G compressed dwords, dword j at ring[j mod (W G)]
};

// ---- counting sort of one code's symbols by code length (lens[0, n), values 0..15; 0 = unused) --------------------------------
// The group's lanes take contiguous shares of the symbols and count them per length in their row; lane j then sums the rows'
// columns j, j + G, ...; every lane runs the scan over the fifteen sums (the codes' first values, limit[] / delta[]); lane j
// turns its columns into each lane's first slot; and the lanes place their symbols -- the slots of one length dealt out in
// lane order = symbol order.  Returns 0 for an over-subscribed code, or an incomplete one where zlib refuses it (inftrees.c:
// only a literal/length or distance code of ONE code of length 1 may be incomplete; no codes at all is allowed too).
template <int G, bool IS_CL, class C>
IW_DEV uint32_t ig_sort(const uint8_t *lens, uint32_t n, uint32_t *rows, uint32_t *tots, const C &code_at, int sub) {
  const uint32_t per = (n + (uint32_t)G - 1u) / (uint32_t)G, s0 = (uint32_t)sub * per, s1 = s0 + per < n ? s0 + per : n;
  uint32_t *mine = rows + 16 * sub;
  for (int l = 0; l < 16; ++l) mine[l] = 0;
  IG_SYNC();
  for (uint32_t s = s0; s < s1; ++s) ig_add(&mine[lens[s]], 1u);
  IG_SYNC();
  for (uint32_t l = (uint32_t)sub; l < 16u; l += (uint32_t)G) {
    uint32_t t = 0;
    for (int g = 0; g < G; ++g) t += rows[16 * g + (int)l];
    tots[l] = t;
  }
  IG_SYNC();
  uint32_t code = 0, off = 0, over = 0, max_len = 0;
  for (uint32_t l = 1; l <= 15u; ++l) {
    const uint32_t tot = tots[l], first = code, end = first + tot;
    over |= (uint32_t)(end > (1u << l));
    max_len = tot ? l : max_len;
    if ((l & (uint32_t)(G - 1)) == (uint32_t)sub) { code_at.lim(l, end << (16u - l), (int32_t)off - (int32_t)first); code_at.slots(l, off + tot); }
    tots[l] = off;                           // (every lane the same value, behind every lane's read)
    code = end << 1;
    off += tot;
  }
  const uint32_t complete = (uint32_t)(code == (1u << 16));
  const uint32_t bad = over | (uint32_t)(!complete && off != 0u && (IS_CL || max_len != 1u));
  IG_SYNC();
  for (uint32_t l = (uint32_t)sub; l < 16u; l += (uint32_t)G) {
    uint32_t run = tots[l];
    for (int g = 0; g < G; ++g) {
      const uint32_t c = rows[16 * g + (int)l];
      rows[16 * g + (int)l] = run;
      run += c;
    }
  }
  IG_SYNC();
  for (uint32_t s = s0; s < s1; ++s) {
    const uint32_t l = lens[s];
    if (l) {
      const uint32_t slot = ig_add_rtn(&mine[l], 1u);
      code_at.sorted(slot, s);
      code_at.placed(l, slot, s);
    }
  }
  IG_SYNC();
  return bad ^ 1u;
}

// ---- first-level table of a sorted code: lane `sub` fills entries sub, sub + G, ... by decoding the entry's own index ----------
enum { IG_K_CL = 0, IG_K_LIT = 1, IG_K_DIST = 2 };
template <int G, int ROOT, int KIND, class T, class C>
IW_DEV void ig_fill(T *tab, const C &code_at, int sub) {
  uint32_t lim[ROOT + 1];
  int32_t del[ROOT + 1];
#pragma unroll
  for (int l = 1; l <= ROOT; ++l) { lim[l] = code_at.limit((uint32_t)l); del[l] = code_at.delta((uint32_t)l); }
  IG_SYNC();
  for (uint32_t i = (uint32_t)sub; i < (1u << ROOT); i += (uint32_t)G) {
    const uint32_t v16 = iw_brev(i) >> 16;
    uint32_t l = 0;
    int32_t d = 0;
#pragma unroll
    for (int k = ROOT; k >= 1; --k) {
      const bool in = v16 < lim[k];
      l = in ? (uint32_t)k : l;
      d = in ? del[k] : d;
    }
    const uint32_t s = code_at.sorted(l ? (uint32_t)((int32_t)(v16 >> (16u - l)) + d) : 0u);
    uint32_t e;
    if (KIND == IG_K_CL) e = s | (l << 5);
    else if (KIND == IG_K_LIT) e = ig_lit_entry(s, l, i, ROOT);
    else e = ig_dist_entry(s, l);
    tab[i] = (T)(l ? e : 0u);
  }
}

// A code longer than the first-level table (or an unused prefix): canonical search over the remaining lengths.  v16 = the next
// 16 stream bits, most significant first.  Returns the symbol and its length, or l = 0.
template <class C> IW_DEV uint32_t ig_slow(uint32_t v16, const C &code_at, int root, uint32_t &len) {
  for (int l = root + 1; l <= 15; ++l) {
    if (code_at.within(v16, (uint32_t)l)) {
      len = (uint32_t)l;
      return code_at.sorted((uint32_t)((int32_t)(v16 >> (16 - l)) + code_at.delta_hi((uint32_t)l)));
    }
  }
  len = 0;
  return 0;
}

// ---- the compressed stream as a group sees it ----------------------------------------------------------------------------------
template <int G> struct IgBits {
  IwBuf in;                    // the launch's compressed bytes, bounds-checked
  uint32_t *ring;
  uint32_t org, skip;          // origin of this reader in `in` (a multiple of 4); bytes between it and the stream piece's first byte
  uint32_t lo, hi, nbits;      // bit buffer: nbits valid bits, zero above them
  uint32_t iw;                 // dwords that have entered the bit buffer
  uint32_t nxt;                // dword iw (from the ring)
  uint32_t rot_at;             // the ring's oldest window is used up once iw >= rot_at: the window RW on takes its place; before iw = rot_at + (RW - 1) G
  int sub;
  static constexpr int RW = IgRing<G>::W;

  IW_DEV void init(uint32_t off) {
    org = off & ~3u;
    skip = off & 3u;
    lo = hi = nbits = iw = 0;
    IG_SYNC();
#pragma unroll
    for (int w = 0; w < RW; ++w) ring[w * G + sub] = iw_ld32(in, org + 4u * (uint32_t)(w * G + sub));
    rot_at = (uint32_t)G;
    IG_SYNC();
    nxt = ring[0];
    refill();
    shr(8u * skip);
  }
  IW_DEV void shr(uint32_t n) {             // n < 32, n <= nbits
    lo = ig_alignbit(hi, lo, n);
    hi >>= n;
    nbits -= n;
  }
  IW_DEV void refill() {                    // afterwards 32 <= nbits <= 63
    if (nbits < 32u) {
      lo |= nxt << nbits;
      hi |= (nxt >> 1) >> (31u - nbits);
      nbits += 32u;
      ++iw;
      nxt = ring[iw & (uint32_t)(RW * G - 1)];
    }
  }
  // The ring's upkeep in two halves (the symbol loop puts a turn's work between them): request this lane's dword of the window
  // that replaces the used-up one; park it.  At least once per (RW - 1) G refills.
  IW_DEV bool rot_due() const { return iw >= rot_at; }
  IW_DEV uint32_t rot_load(bool due) const { return iw_ld32(in, due ? org + 4u * (rot_at + (uint32_t)((RW - 1) * G + sub)) : IW_OOB); }
  IW_DEV void rot_store(bool due, uint32_t w) {
    if (due) {
      ring[((rot_at + (uint32_t)((RW - 1) * G)) & (uint32_t)(RW * G - 1)) + (uint32_t)sub] = w;
      rot_at += (uint32_t)G;
    }
  }
  IW_DEV void rotate() {
    bool due = rot_due();
    rot_store(due, rot_load(due));
    if (G == 1) while ((due = rot_due())) rot_store(due, rot_load(due));   // (the host build's windows are one dword: a step may use up two)
  }
  IW_DEV void refill_r() { refill(); rotate(); }
  IW_DEV uint32_t bits(uint32_t n) {        // n <= 16 (0 allowed); the caller keeps nbits >= n
    const uint32_t v = lo & ((1u << n) - 1u);
    shr(n);
    return v;
  }
  IW_DEV uint32_t byte_pos() const { return 4u * iw - (nbits >> 3); }                          // from the origin to the next unread byte
  IW_DEV uint64_t consumed() const { return 32ull * iw - nbits - 8ull * skip; }              // bits since init
};

// ---- output of one stream ------------------------------------------------------------------------------------------------------
// k mod D for the copy of a periodic match (k <= 257, D <= 257): the quotient through a float reciprocal, corrected by one either way
struct IgMod {
  float rcp;
  uint32_t D;
  IW_DEV uint32_t operator()(uint32_t k) const {
    const uint32_t q = (uint32_t)((float)k * rcp);
    int32_t m = (int32_t)k - (int32_t)(q * D);
    m = m < 0 ? m + (int32_t)D : m;
    m = m >= (int32_t)D ? m - (int32_t)D : m;
    return (uint32_t)m;
  }
};
template <int G> struct IgOut {
  IwBuf out;                   // the launch's output, bounds-checked
  uint32_t base, isize, pos;   // the stream's bytes are out[base, base + isize); next position
  int sub;

  IW_DEV void literal(uint32_t b) {         // caller checked pos < isize
    iw_st8(out, sub == 0 ? base + pos : IW_OOB, b);
    ++pos;
  }
  // LZ77 match, the plain way: 3 <= L <= 258, 1 <= D <= pos, pos + L <= isize.  Byte k is out[pos - D + k mod D]: every source
  // byte was written before the match started, so the rounds may go in any order, also when the match overlaps itself.
  IW_DEV void match(uint32_t L, uint32_t D) {
    const uint32_t src = base + pos - D, dst = base + pos;
    if (D >= L) {
      for (uint32_t k = (uint32_t)sub; k < L; k += (uint32_t)G) iw_st8(out, dst + k, iw_ld8(out, src + k));
    } else {
      const IgMod mod{IW_RCP((float)D), D};
      for (uint32_t k = (uint32_t)sub; k < L; k += (uint32_t)G) iw_st8(out, dst + k, iw_ld8(out, src + mod(k)));
    }
    pos += L;
  }
  // `n` bytes of the compressed stream itself (a stored block), from offset `p` of `in`
  IW_DEV void raw(const IwBuf &in, uint32_t p, uint32_t n) {
    for (uint32_t k = (uint32_t)sub; k < n; k += (uint32_t)G) iw_st8(out, base + pos + k, iw_ld8(in, p + k));
    pos += n;
  }
};

// One symbol the general way (the symbol loop's rare cases: codes longer than the first-level tables, lengths whose extra bits
// the index does not hold, the end of a block, the last bytes of a stream, anything invalid).  0: go on; 1: end of block;
// 0x100 | IW_ERR_*: the stream is refused.
template <int G>
IW_DEV uint32_t ig_symbol(IgBits<G> &br, IgOut<G> &o, const IgLds<G> &S, const IgCodeW &ll, const IgCodeW &dd) {
  br.refill_r();
  IG_STAT(1);
  uint32_t e = S.lit_tab[br.lo & ((1u << IG_LIT_ROOT) - 1u)];
  if (!(e & 15u)) {
    uint32_t l;
    IG_STAT(2);
    const uint32_t s = ig_slow(iw_brev(br.lo) >> 16, ll, IG_LIT_ROOT, l);
    if (!l) return 0x100u | IW_ERR_DATA;
    e = ig_lit_entry(s, l, 0u, 0);
  }
  br.shr(e & 15u);
  if (e & IG_LIT) {
    if (o.pos >= o.isize) return 0x100u | IW_ERR_SIZE;
    o.literal((e >> 4) & 0xffu);
    return 0;
  }
  uint32_t L;
  if (e & IG_LEN) L = (e >> 4) & 0x1ffu;
  else if (e & IG_EOB) return 1;
  else if (e & IG_LENX) {
    const uint32_t c = (e >> 4) & 31u;
    IG_STAT(3);
    L = ig_len_base(c) + br.bits(ig_len_extra(c));
  } else return 0x100u | IW_ERR_DATA;                            // 286, 287
  br.refill_r();
  uint32_t d = S.dist_tab[br.lo & ((1u << IG_DIST_ROOT) - 1u)];
  if (!(d & 15u)) {
    uint32_t l;
    IG_STAT(4);
    const uint32_t s = ig_slow(iw_brev(br.lo) >> 16, dd, IG_DIST_ROOT, l);
    if (!l) return 0x100u | IW_ERR_DATA;
    d = ig_dist_entry(s, l);
  }
  if (d & IG_DBAD) return 0x100u | IW_ERR_DATA;
  br.shr(d & 15u);
  const uint32_t D = (d >> 16) + br.bits((d >> 4) & 15u);
  if (D > o.pos) return 0x100u | IW_ERR_DATA;
  if (o.pos + L > o.isize) return 0x100u | IW_ERR_SIZE;
  o.match(L, D);
  return 0;
}

// Inflate the raw DEFLATE stream in[off, off + clen) into out[obase, obase + isize).  Returns 0 or IW_ERR_* flags; never loads
// outside `in`, never stores outside out[obase, obase + isize).  Every lane of the group calls it with the same arguments and
// its own `sub` (0 .. G - 1), and gets the same result.
// (out_ld: the same bytes as `out` for the match copies' loads -- on the device three bytes longer, see ig_ld8_raw; work[wbase, wbase + IG_WORK_STRIDE): the stream's slice of the workspace)
template <int G>
IW_DEV int ig_inflate(const IwBuf &in, uint32_t off, uint32_t clen, const IwBuf &out, const IwBuf &out_ld, uint32_t obase, uint32_t isize, const IwBuf &work,
                      uint32_t wbase, IgLds<G> &S, int sub) {
  const IgCodeW ll{work, wbase + IG_W_LL, wbase + IG_W_LL_LIM, wbase + IG_W_LL_DEL, S.ll_hi, (uint32_t)IG_LIT_ROOT, S.ll_nlit},
      dd{work, wbase + IG_W_D, wbase + IG_W_D_LIM, wbase + IG_W_D_DEL, S.d_hi, (uint32_t)IG_DIST_ROOT, nullptr};
  const IgCodeL cl{S.cl.sorted, S.cl.limit, S.cl.delta};
  IgBits<G> br;
  br.in = in;
  br.ring = S.ring;
  br.sub = sub;
  br.init(off);
  uint64_t stream_bits = 8ull * clen;      // bits of the stream still ahead of this reader's origin (stored blocks restart the reader)
  IgOut<G> o;
  o.out = out;
  o.base = obase; o.isize = isize; o.pos = 0; o.sub = sub;
  for (;;) {
    br.refill_r();
    const uint32_t bfinal = br.bits(1), btype = br.bits(2);
    if (btype == 3u) return IW_ERR_DATA;
    if (btype == 0u) {
      br.bits(br.nbits & 7u);                                   // to the next byte boundary
      br.refill_r();
      const uint32_t len = br.bits(16);
      br.refill_r();
      const uint32_t nlen = br.bits(16);
      if ((len ^ 0xffffu) != nlen) return IW_ERR_DATA;
      const uint64_t used = br.consumed();
      if (used + 8ull * len > stream_bits) return IW_ERR_DATA;  // the stored bytes reach past the stream
      if (o.pos + len > isize) return IW_ERR_SIZE;
      const uint32_t p = br.org + br.byte_pos();
      o.raw(in, p, len);
      stream_bits -= used + 8ull * len;
      br.init(p + len);
    } else {
      uint32_t hlit, hdist;
      if (btype == 1u) {
        hlit = 288; hdist = 32;
        for (uint32_t s = (uint32_t)sub; s < 320u; s += (uint32_t)G) S.b.lens[s] = (uint8_t)(s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : s < 288u ? 8 : 5);
      } else {
        hlit = br.bits(5) + 257u;
        hdist = br.bits(5) + 1u;
        const uint32_t hclen = br.bits(4) + 4u;
        if (hlit > 286u || hdist > 30u) return IW_ERR_DATA;
        uint64_t clv = 0;                                       // 3 bits per code-length symbol
        for (uint32_t i = 0; i < hclen; ++i) {
          br.refill_r();
          clv |= (uint64_t)br.bits(3) << (3u * iw_cl_order(i));
        }
        for (uint32_t s = (uint32_t)sub; s < 19u; s += (uint32_t)G) S.cl.lens[s] = (uint8_t)((clv >> (3u * s)) & 7u);
        IG_SYNC();
        if (!ig_sort<G, true>(S.cl.lens, 19u, S.b.rows, S.b.tots, cl, sub)) return IW_ERR_DATA;
        ig_fill<G, IG_CL_ROOT, IG_K_CL, uint8_t>(S.cl.tab, cl, sub);
        IG_SYNC();
        const uint32_t n = hlit + hdist;
        uint32_t i = 0, prev = 0;
        while (i < n) {
          br.refill_r();
          const uint32_t e = S.cl.tab[br.lo & ((1u << IG_CL_ROOT) - 1u)];
          if (!e) return IW_ERR_DATA;
          br.shr(e >> 5);
          const uint32_t s = e & 31u;
          uint32_t rep, val;
          if (s < 16u) { rep = 1; val = s; }
          else if (s == 16u) { if (!i) return IW_ERR_DATA; rep = 3u + br.bits(2); val = prev; }
          else if (s == 17u) { rep = 3u + br.bits(3); val = 0; }
          else { rep = 11u + br.bits(7); val = 0; }
          if (i + rep > n) return IW_ERR_DATA;
          for (uint32_t r = (uint32_t)sub; r < rep; r += (uint32_t)G) S.b.lens[i + r] = (uint8_t)val;
          i += rep;
          prev = val;
        }
        IG_SYNC();
        if (S.b.lens[256] == 0u) return IW_ERR_DATA;              // no end-of-block code (inflate.c: "missing end-of-block")
      }
      IG_SYNC();
      // both codes are sorted before either table is filled: the literal/length table takes the place of the code lengths
      const uint32_t ok_ll = ig_sort<G, false>(S.b.lens, hlit, S.b.rows, S.b.tots, ll, sub);
      const uint32_t ok_d = ig_sort<G, false>(S.b.lens + hlit, hdist, S.b.rows, S.b.tots, dd, sub);
      if (!ok_ll || !ok_d) return IW_ERR_DATA;
      ig_fill<G, IG_LIT_ROOT, IG_K_LIT, uint16_t>(S.lit_tab, ll, sub);
      ig_fill<G, IG_DIST_ROOT, IG_K_DIST, uint32_t>(S.dist_tab, dd, sub);
      IG_SYNC();
      // The symbol loop.  A turn: (A) the match whose length code the previous turn's third lookup found -- the buffer is filled
      // (>= 32 bits), length (<= 9 bits, extra bits resolved) and distance (<= 7 + 13) decoded into temporaries and committed
      // if this is the common case, the copy's bytes requested; (B) the buffer filled again, three lookups, each consuming its
      // literal if it is one (<= 27 bits); (C) the copy's bytes stored.  Anything else is left to ig_symbol on the next turn.
      uint32_t why = 0, e = 0;
      bool general = false;
      for (;;) {
        if (general || o.pos + 261u > isize) {                 // (the longest match and three literals fit otherwise)
          general = false;
          e = 0;
          const uint32_t r = ig_symbol<G>(br, o, S, ll, dd);
          if (r == 1u) break;
          if (r) { why = r & 0xffu; break; }
          continue;
        }
        IG_STAT(0);
        const bool due = br.rot_due();
        const uint32_t gw = br.rot_load(due);
        // (A)
        br.refill();
        uint32_t L, D;
        // a literal with a code longer than the first level (7 % of the turns of a level-6 BAM block: one wave in two would
        // be in ig_symbol every turn): the lengths' bounds are in LDS, the slot says "literal", and the byte itself -- the
        // symbol, in the workspace -- is only ever stored: requested here, stored at (C) like the bytes of a match
        uint32_t sl_at = IW_OOB, sl_to = IW_OOB;
        if (e == IG_LONG) {
          const uint32_t v16 = iw_brev(br.lo) >> 16;
          uint32_t l = 0, slot = 0, lim = 0;
#pragma unroll
          for (int k = 15; k > IG_LIT_ROOT; --k) {
            const bool in = (v16 >> (16 - k)) < S.ll_hi[2 * (k - IG_LIT_ROOT - 1)];
            l = in ? (uint32_t)k : l;
            slot = in ? (uint32_t)((int32_t)(v16 >> (16 - k)) + (int32_t)(int16_t)S.ll_hi[2 * (k - IG_LIT_ROOT - 1) + 1]) : slot;
            lim = in ? S.ll_nlit[k - IG_LIT_ROOT - 1] : lim;
          }
          if (slot < lim) {                                     // (l = 0: 0 < 0 fails)
            IG_STAT(9);
            br.shr(l);
            sl_at = ll.so + 2u * slot;
            sl_to = sub == 0 ? o.base + o.pos : IW_OOB;
            ++o.pos;
          } else general = true;
          e = 0;
        }
        const ig_byte sl_b = ig_ld8_raw(work, sl_at);
        bool ok = (e & IG_LEN) != 0u;
        {
          const uint32_t n = e & 15u;
          L = (e >> 4) & 0x1ffu;
          const uint32_t tlo = ig_alignbit(br.hi, br.lo, n), thi = br.hi >> n;
          const uint32_t d = S.dist_tab[tlo & ((1u << IG_DIST_ROOT) - 1u)];
          const uint32_t dl = d & 15u, eb = (d >> 4) & 15u;
          const uint32_t t2 = ig_alignbit(thi, tlo, dl), t2h = thi >> dl;
          D = (d >> 16) + (t2 & ((1u << eb) - 1u));
          const bool good = dl != 0u && !(d & IG_DBAD) && D <= o.pos;
          general = general || (ok && !good);
          ok = ok && good;
          br.lo = ok ? ig_alignbit(t2h, t2, eb) : br.lo;
          br.hi = ok ? t2h >> eb : br.hi;
          br.nbits -= ok ? n + dl + eb : 0u;
          L = ok ? L : 0u;
        }
        const uint32_t src = o.base + o.pos - D, dst = o.base + o.pos;
        // offsets of this lane's four bytes behind the source's start: k, or k mod D where the match overlaps itself (all of
        // it arithmetic, ahead of the first load: a branch between loads makes the compiler wait for the ones in flight)
        uint32_t kk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) kk[r] = (uint32_t)(sub + G * r);
        const bool periodic = D < L;
        if (L) { IG_STAT(5); if (periodic) IG_STAT(6); if (L > 16u) IG_STAT(7); if (L > 32u) IG_STAT(8); }
        if (periodic && D < (uint32_t)(4 * G)) {                // (k < 4 G <= D otherwise)
          // q = k * M >> 16 is k / D exactly for M in (65536 / D, 65536 / D + 2.1], k < 32, D < 32: k (M D - 65536) < 65536;
          // the float reciprocal is within a millionth, the truncation within one
          const uint32_t M = (uint32_t)(65536.0f * IW_RCP((float)D)) + 2u;
#pragma unroll
          for (int r = 0; r < 4; ++r) kk[r] -= ((kk[r] * M) >> 16) * D;
        }
        ig_byte pd[4];
        uint32_t pa[4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint32_t k = (uint32_t)(sub + G * r);
          pd[r] = ig_ld8_raw(out_ld, k < L ? src + kk[r] : IW_OOB);
          pa[r] = k < L ? dst + k : IW_OOB;
        }
        pa[2] = pa[3] = IW_OOB;
        pd[2] = pd[3] = 0;
        if (L > (uint32_t)(2 * G)) {
#pragma unroll
          for (int r = 2; r < 4; ++r) {
            const uint32_t k = (uint32_t)(sub + G * r);
            pd[r] = ig_ld8_raw(out_ld, k < L ? src + kk[r] : IW_OOB);
            pa[r] = k < L ? dst + k : IW_OOB;
          }
          if (L > (uint32_t)(4 * G)) {                          // (rare: the rest of a long match the plain way)
            const IgMod mod{IW_RCP((float)D), D};
            for (uint32_t k = (uint32_t)(4 * G + sub); k < L; k += (uint32_t)G) iw_st8(out, dst + k, iw_ld8(out, src + (periodic ? mod(k) : k)));
          }
        }
        o.pos += L;
        // (B)
        br.refill();
        {
          uint32_t lo = br.lo, hi = br.hi, nb = br.nbits, pos = o.pos;
          e = S.lit_tab[lo & ((1u << IG_LIT_ROOT) - 1u)];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const bool isl = (e & IG_LIT) != 0u;
            ig_st8_raw(out, (isl && sub == 0) ? o.base + pos : IW_OOB, e >> 4);
            const uint32_t n = isl ? (e & 15u) : 0u;
            lo = ig_alignbit(hi, lo, n);
            hi >>= n;
            nb -= n;
            pos += isl ? 1u : 0u;
            if (j < 2) e = S.lit_tab[lo & ((1u << IG_LIT_ROOT) - 1u)];
          }
          br.lo = lo; br.hi = hi; br.nbits = nb; o.pos = pos;
          if (e & IG_LIT) e = 0;                                // (consumed)
          else if (!(e & 15u)) e = IG_LONG;
          else if (!(e & IG_LEN)) { general = true; e = 0; }
        }
        // (C)
#pragma unroll
        for (int r = 0; r < 4; ++r) ig_st8_raw(out, pa[r], pd[r]);
        ig_st8_raw(out, sl_to, sl_b);
        br.rot_store(due, gw);
      }
      if (why) return (int)why;
    }
    if (bfinal) break;
  }
  if (br.consumed() > stream_bits) return IW_ERR_DATA;          // the decoder read past the end of the stream
  if (o.pos != isize) return IW_ERR_SIZE;
  return 0;
}

}  // namespace strl
