// inflate_core.h -- the per-lane DEFLATE (RFC 1951) decoder of bgzf.hip.  One lane decodes one BGZF block; the lane's
// tables and its output ring live in LDS columns with a stride of INF_LANES elements (each lane its own bank).
// The same source compiles for the host (STRL_EMU: one "lane", stride 1) purely so that the CPU-only test-suite can run
// the device logic against zlib; the product never runs that build.
#pragma once
#include <stdint.h>
#ifdef STRL_EMU
#define INF_DEV inline
#define INF_NOINLINE __attribute__((noinline))
#define INF_LANES 1
#define INF_TABLE static const
#else
#include <hip/hip_runtime.h>
#define INF_DEV __device__ __forceinline__
#define INF_NOINLINE __device__ __attribute__((noinline))
#define INF_LANES 64
#define INF_TABLE __device__ const
#endif

namespace strl {

constexpr int INF_ERR_DATA = 1, INF_ERR_SIZE = 2;

// RFC 1951 3.2.5 in closed form (a table in global memory would be a dependent ~600-cycle load per match):
//   length code s (0..28): s < 8: 3 + s, no extra bits; s == 28: 258; else e = (s - 4) / 4 extra bits, base ((4 + s % 4) << e) + 3
//   distance code d (0..29): d < 4: 1 + d; else e = (d - 2) / 2 extra bits, base ((2 + d % 2) << e) + 1
INF_DEV void len_code(int s, uint32_t &base, int &ext) {
  if (s < 8) { base = 3u + (uint32_t)s; ext = 0; }
  else if (s == 28) { base = 258; ext = 0; }
  else { ext = (s - 4) >> 2; base = ((4u + ((uint32_t)s & 3u)) << ext) + 3u; }
}
INF_DEV void dist_code(int d, uint32_t &base, int &ext) {
  if (d < 4) { base = 1u + (uint32_t)d; ext = 0; }
  else { ext = (d - 2) >> 1; base = ((2u + ((uint32_t)d & 1u)) << ext) + 1u; }
}
// order of the code-length code lengths (RFC 1951 3.2.7), 5 bits each: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
INF_DEV int cl_order(int i) {
  const uint64_t lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) |
                      (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
  const uint64_t hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
  return (int)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31u);
}

// LDS of one wave: symbol tables + count scratch (u16 rows), the output ring (dword rows), the code-length column.
constexpr int L_SYMLL = 288, L_SYMD = 32, L_LENS = 320, L_CNT = 16;
constexpr int INF_R = 1024;                                       // bytes of output a lane keeps in LDS (ring, flushed 256 bytes at a time)
constexpr int INF_TAB_BYTES = (L_SYMLL + L_SYMD + 2 * L_CNT) * INF_LANES * 2;
constexpr int INF_WIN_BYTES = INF_R * INF_LANES;
constexpr int INF_LENS_BYTES = L_LENS * INF_LANES;
constexpr int INF_LDS_BYTES = INF_TAB_BYTES + INF_WIN_BYTES + INF_LENS_BYTES;

// The code-length column used while a block's tables are built: its own LDS rows (dword-interleaved like the ring: byte s
// of a lane in dword s / 4 of the lane's column), so that a table build leaves the ring -- the LZ77 history -- intact.
struct LensCol {
  uint32_t *col;
  INF_DEV uint8_t get(int s) const { return reinterpret_cast<const uint8_t *>(col + (s >> 2) * INF_LANES)[s & 3]; }
  INF_DEV void set(int s, uint8_t v) const { reinterpret_cast<uint8_t *>(col + (s >> 2) * INF_LANES)[s & 3] = v; }
};

struct BitReader {
  const uint32_t *wp;
  uint64_t buf;
  uint32_t ahead;  // the next input dword, loaded one refill early (its latency hides behind ~3 symbols of work)
  int cnt;
  int64_t left;   // bits of the stream not yet loaded into buf
  INF_DEV void init(const uint8_t *p, uint32_t nbytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    wp = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const int skip = (int)(a & 3);
    buf = (uint64_t)(*wp++) >> (8 * skip);
    cnt = 32 - 8 * skip;
    left = (int64_t)nbytes * 8 - cnt;
    ahead = *wp++;
  }
  INF_DEV void refill() {
    if (cnt <= 32) { buf |= (uint64_t)ahead << cnt; cnt += 32; left -= 32; ahead = *wp++; }
  }
  INF_DEV uint32_t bits(int n) {   // n <= 16, caller keeps cnt >= n
    const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
    buf >>= n;
    cnt -= n;
    return v;
  }
  INF_DEV uint32_t bit() { const uint32_t v = (uint32_t)buf & 1u; buf >>= 1; --cnt; return v; }
  INF_DEV bool overrun() const { return left + cnt < 0; }   // consumed more bits than the stream holds
};

// code counts per length, 10 bits each, three per register: c[(len - 1) / 3] >> 10 * ((len - 1) % 3)
struct Counts { uint32_t c[5]; };
#define INF_CNT_OF(C, len) (((C).c[((len) - 1) / 3] >> (10 * (((len) - 1) % 3))) & 0x3ffu)

// canonical Huffman decode (one bit at a time, codes are packed most-significant bit first): <= 15 steps of register work,
// then ONE table access.  Returns -1 for an invalid code.
INF_DEV int huff_decode(BitReader &br, const Counts &C, const uint16_t *sym) {
  int code = 0, first = 0, index = 0;
  uint32_t w = (uint32_t)br.buf;            // the next >= 15 bits (the caller refilled): 32-bit work, one 64-bit shift at the end
#pragma unroll
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(w & 1u);
    w >>= 1;
    const int count = (int)INF_CNT_OF(C, len);
    if (code - count < first) {
      br.buf >>= len;
      br.cnt -= len;
      return (int)sym[(index + (code - first)) * INF_LANES];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// build the decoding table of `n` symbols whose code lengths sit in lens (column of bytes): counts per length -> C, symbols
// ordered by (length, value) -> sym.  cnt / offs: scratch columns of 16 u16.  Returns false for an over-subscribed set of
// lengths (incomplete sets are allowed: a single distance code is legal).
INF_DEV bool huff_build(const LensCol &lens, int base, int n, uint16_t *cnt, uint16_t *offs, uint16_t *sym, Counts &C) {
  for (int l = 0; l < 16; ++l) cnt[l * INF_LANES] = 0;
  for (int s = 0; s < n; ++s) { const int l = lens.get(base + s); cnt[l * INF_LANES] = (uint16_t)(cnt[l * INF_LANES] + 1); }
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= (int)cnt[l * INF_LANES];
    if (left < 0) return false;
  }
  offs[1 * INF_LANES] = 0;
  for (int l = 1; l < 15; ++l) offs[(l + 1) * INF_LANES] = (uint16_t)(offs[l * INF_LANES] + cnt[l * INF_LANES]);
  for (int s = 0; s < n; ++s) {
    const int l = lens.get(base + s);
    if (l) { const int o = offs[l * INF_LANES]; sym[o * INF_LANES] = (uint16_t)s; offs[l * INF_LANES] = (uint16_t)(o + 1); }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) C.c[q] = 0;
#pragma unroll
  for (int l = 1; l <= 15; ++l) C.c[(l - 1) / 3] |= (uint32_t)cnt[l * INF_LANES] << (10 * ((l - 1) % 3));
  return true;
}

// The inflated bytes of a lane go through a 1 KiB ring in LDS, dword-interleaved across the lanes (dword j of lane l at
// j * 64 + l: every lane its own bank).  Bytes are assembled into a pending dword in a register and committed to the ring
// a dword at a time; when the ring is full its older half leaves as 16-byte global stores (a byte store per literal would be
// one memory transaction per byte and lane), so the last >= 512 bytes are always in LDS.  LZ77 matches read their source
// from the pending register / the ring, four bytes at a time when the distance allows; only sources further back than the
// ring come from global memory (flushed long before: no load-after-store round trips).
struct OutRing {
  uint32_t *ring;      // this lane's dword 0 (dword j at ring[j * INF_LANES])
  uint32_t *ring0;     // lane 0's dword 0: the cooperative flush reads other lanes' columns
  int lane;
  uint8_t *gbase;      // 16-byte aligned global address of virtual position 0
  uint32_t a0;         // virtual position of the stream's first byte
  uint32_t v;          // virtual position of the next byte
  uint32_t cur;        // pending dword: bytes [v & ~3, v)
  uint32_t lo;         // bytes at virtual positions >= lo (and below v & ~3) are valid in the ring
  uint32_t flushed;    // bytes below `flushed` are in global memory
  INF_DEV void init(uint32_t *ring_lane0, int lane_, uint8_t *out) {
    ring0 = ring_lane0; lane = lane_;
    ring = ring_lane0 + lane_;
    const uintptr_t a = reinterpret_cast<uintptr_t>(out);
    a0 = (uint32_t)(a & 15u);
    gbase = out - a0;
    v = a0; cur = 0; lo = a0 & ~3u; flushed = a0;
  }
  INF_DEV uint32_t ring_dword(uint32_t pos) const { return ring[((pos & (uint32_t)(INF_R - 1)) >> 2) * INF_LANES]; }
  // committed ring bytes [flushed, to) -> global memory
  INF_DEV void flush_to(uint32_t to) {
    if (to <= flushed) return;
    uint32_t p = flushed;
    while (p < to) {
      if ((p & 15u) == 0 && p + 16 <= to) {
        uint32_t *g = reinterpret_cast<uint32_t *>(gbase + p);
#ifdef STRL_EMU
        g[0] = ring_dword(p); g[1] = ring_dword(p + 4); g[2] = ring_dword(p + 8); g[3] = ring_dword(p + 12);
#else
        *reinterpret_cast<uint4 *>(g) = make_uint4(ring_dword(p), ring_dword(p + 4), ring_dword(p + 8), ring_dword(p + 12));
#endif
        p += 16;
      } else {
        gbase[p] = (uint8_t)(ring_dword(p) >> (8 * (p & 3u)));
        ++p;
      }
    }
    flushed = to;
  }
  // Room for the longest symbol (258 bytes) before a symbol is decoded, so that put() never has to flush.  The lanes of a
  // wave that are here together flush for one another: a lane that is short of room has its oldest <= 256 bytes written
  // by ALL of them, one dword each -- one coalesced 256-byte store instead of a 16-iteration loop that the other 63 lanes
  // would sit through (a lane fills its ring every few hundred symbols, so with 64 lanes somebody nearly always would).
  INF_DEV void make_room() {
#ifdef STRL_EMU
    while (v - flushed > (uint32_t)(INF_R - 264)) {
      const uint32_t to = (flushed + 256u) & ~255u;
      flush_to(to);
      if (lo < to) lo = to;
    }
#else
    for (;;) {
      const bool need = v - flushed > (uint32_t)(INF_R - 264);
      unsigned long long m = __ballot(need);
      if (!m) break;
      const unsigned long long act = __ballot(true);
      const uint32_t n_act = (uint32_t)__popcll(act), rank = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
      while (m) {
        const int L = __ffsll((long long)m) - 1;
        m &= m - 1;
        uint32_t p = (uint32_t)__shfl((int)flushed, L);
        const uint32_t q = (p + 256u) & ~255u;
        const uint64_t gb = ((uint64_t)(uint32_t)__shfl((int)(reinterpret_cast<uintptr_t>(gbase) >> 32), L) << 32) |
                            (uint32_t)__shfl((int)(reinterpret_cast<uintptr_t>(gbase) & 0xffffffffu), L);
        uint8_t *g = reinterpret_cast<uint8_t *>(gb);
        if (lane == L) while ((p & 3u) && p < q) { g[p] = (uint8_t)(ring_dword(p) >> (8 * (p & 3u))); ++p; }   // unaligned start of the stream
        p = (p + 3u) & ~3u;
        for (uint32_t d = rank; p + 4 * d < q; d += n_act) {
          const uint32_t pos = p + 4 * d;
          *reinterpret_cast<uint32_t *>(g + pos) = ring0[((pos & (uint32_t)(INF_R - 1)) >> 2) * INF_LANES + L];
        }
        if (lane == L) { flushed = q; if (lo < q) lo = q; }
      }
    }
#endif
  }
  INF_DEV void commit() {   // v is dword aligned: cur holds bytes [v - 4, v); make_room() guaranteed the slot is free
    ring[(((v - 4) & (uint32_t)(INF_R - 1)) >> 2) * INF_LANES] = cur;
    cur = 0;
  }
  INF_DEV void put(uint32_t b) {
    cur |= b << (8 * (v & 3u));
    ++v;
    if (!(v & 3u)) commit();
  }
  INF_DEV void put4(uint32_t x) {   // four bytes, x's low byte first
    const uint32_t k = v & 3u;
    cur |= x << (8 * k);
    v += 4 - k;
    commit();
    if (k) { cur = x >> (8 * (4 - k)); v += k; }
  }
  INF_DEV uint32_t get(uint32_t sv) const {   // one byte at virtual position sv < v
    if (sv >= (v & ~3u)) return (cur >> (8 * (sv & 3u))) & 0xffu;
    if (sv >= lo) return (ring_dword(sv) >> (8 * (sv & 3u))) & 0xffu;
    return gbase[sv];
  }
  INF_DEV uint32_t get4(uint32_t sv) const {  // bytes sv .. sv + 3, all committed and >= lo
    const uint32_t sh = sv & 3u;
    const uint32_t w0 = ring_dword(sv);
    if (!sh) return w0;
    const uint32_t w1 = ring_dword(sv + 4);
    return (uint32_t)((((uint64_t)w1 << 32) | w0) >> (8 * sh));
  }
  // end of a deflate block / of the stream: make everything so far visible in global memory
  INF_DEV void flush_all() {
    flush_to(v & ~3u);                                                        // the committed part first: the slot of the
    if (v & 3u) {                                                             // partial dword may still hold bytes 1 KiB older
      ring[((v & (uint32_t)(INF_R - 1)) >> 2) * INF_LANES] = cur;             // (written without advancing)
      flush_to(v);
    }
  }
};

INF_DEV void lz_copy(OutRing &W, uint32_t dist, uint32_t len) {
  uint32_t sv = W.v - dist;
  if (dist < 8) {                                  // short period: replay it from a register
    uint64_t pat = 0;
    for (uint32_t j = 0; j < dist; ++j) pat |= (uint64_t)W.get(sv + j) << (8 * j);
    if (dist == 1) {
      const uint32_t x = (uint32_t)pat * 0x01010101u;
      while (len && (W.v & 3u)) { W.put(x & 0xffu); --len; }
      for (; len >= 4; len -= 4) W.put4(x);
      while (len) { W.put(x & 0xffu); --len; }
      return;
    }
    uint32_t ph = 0;
    for (uint32_t i = 0; i < len; ++i) {
      W.put((uint32_t)(pat >> (8 * ph)) & 0xffu);
      ph = ph + 1 == dist ? 0 : ph + 1;
    }
    return;
  }
  // dist >= 8: four source bytes at a time are always committed; from the ring when they are recent enough
  while (len >= 8 && sv + 7 < W.lo) {             // older than the ring: flushed long ago, eight independent loads at a time
    const uint8_t *g = W.gbase + sv;
    const uint32_t b0 = g[0], b1 = g[1], b2 = g[2], b3 = g[3], b4 = g[4], b5 = g[5], b6 = g[6], b7 = g[7];
    W.put4(b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
    W.put4(b4 | (b5 << 8) | (b6 << 16) | (b7 << 24));
    sv += 8; len -= 8;
  }
  while (len && sv < W.lo) { W.put(W.get(sv)); ++sv; --len; }
  while (len >= 4) { W.put4(W.get4(sv)); sv += 4; len -= 4; }
  while (len) { W.put(W.get(sv)); ++sv; --len; }
}

// One DEFLATE stream: comp[0, clen) -> out[0, isize).  sym_ll / sym_d / cnt / offs: this lane's u16 table columns, win: this
// lane's ring (dword column) whose bytes double as the code-length column `lens` while tables are built.  Returns 0 or
// INF_ERR_*.
INF_DEV int inflate_lane(const uint8_t *comp, uint32_t clen, uint8_t *out_ptr, uint32_t isize, uint16_t *sym_ll, uint16_t *sym_d, uint16_t *cnt,
                         uint16_t *offs, uint32_t *win0, uint32_t *lens0, int lane) {
  const LensCol lens{lens0 + lane};
  BitReader br;
  br.init(comp, clen);
  OutRing W;
  W.init(win0, lane, out_ptr);
  const uint32_t vend = W.a0 + isize;              // virtual position one past the last byte
  int err = 0;
  bool last = false;
  while (!last && !err) {
    br.refill();
    last = br.bit() != 0;
    const uint32_t type = br.bits(2);
    if (type == 0) {                       // stored
      br.bits(br.cnt & 7);                 // to the byte boundary (cnt and the stream position are congruent mod 8)
      br.refill();
      const uint32_t len = br.bits(16);
      br.refill();
      const uint32_t nlen = br.bits(16);
      if ((len ^ 0xffffu) != nlen || W.v + len > vend) { err = INF_ERR_DATA; break; }
      for (uint32_t i = 0; i < len; ++i) {
        if (!(i & 127u)) W.make_room();
        br.refill();
        W.put(br.bits(8));
      }
      if (br.overrun()) { err = INF_ERR_DATA; break; }
      continue;
    }
    if (type == 3) { err = INF_ERR_DATA; break; }
    Counts CL{}, CD{};
    if (type == 1) {                       // fixed codes
      for (int s = 0; s < 144; ++s) lens.set(s, 8);
      for (int s = 144; s < 256; ++s) lens.set(s, 9);
      for (int s = 256; s < 280; ++s) lens.set(s, 7);
      for (int s = 280; s < 288; ++s) lens.set(s, 8);
      huff_build(lens, 0, 288, cnt, offs, sym_ll, CL);
      for (int s = 0; s < 30; ++s) lens.set(s, 5);
      huff_build(lens, 0, 30, cnt, offs, sym_d, CD);
    } else {                               // dynamic codes
      const int nlen = (int)br.bits(5) + 257;
      const int ndist = (int)br.bits(5) + 1;
      const int ncode = (int)br.bits(4) + 4;
      if (nlen > 286 || ndist > 30) { err = INF_ERR_DATA; break; }
      for (int i = 0; i < 19; ++i) lens.set(i, 0);
      for (int i = 0; i < ncode; ++i) { br.refill(); lens.set(cl_order(i), (uint8_t)br.bits(3)); }
      Counts CC{};
      if (!huff_build(lens, 0, 19, cnt, offs, sym_d, CC)) { err = INF_ERR_DATA; break; }   // the code-length code lives in sym_d for now
      // the decoded lengths must not overwrite the code-length code's own lengths while they are read: they are not read
      // again (huff_build consumed them), so lens can be reused from index 0
      int idx = 0;
      while (idx < nlen + ndist) {
        br.refill();
        const int s = huff_decode(br, CC, sym_d);
        if (s < 0) { err = INF_ERR_DATA; break; }
        if (s < 16) lens.set(idx++, (uint8_t)s);
        else {
          int prev = 0, rep;
          if (s == 16) {
            if (idx == 0) { err = INF_ERR_DATA; break; }
            prev = lens.get(idx - 1);
            rep = 3 + (int)br.bits(2);
          } else if (s == 17) rep = 3 + (int)br.bits(3);
          else rep = 11 + (int)br.bits(7);
          if (idx + rep > nlen + ndist) { err = INF_ERR_DATA; break; }
          while (rep--) lens.set(idx++, (uint8_t)prev);
        }
      }
      if (err) break;
      if (lens.get(256) == 0) { err = INF_ERR_DATA; break; }                     // no end-of-block code
      if (!huff_build(lens, 0, nlen, cnt, offs, sym_ll, CL)) { err = INF_ERR_DATA; break; }
      if (!huff_build(lens, nlen, ndist, cnt, offs, sym_d, CD)) { err = INF_ERR_DATA; break; }
    }
    // ---- the symbols of this block ----
    for (;;) {
      W.make_room();
      br.refill();
      int s = huff_decode(br, CL, sym_ll);
      if (s < 0) { err = INF_ERR_DATA; break; }
      if (s < 256) {
        if (W.v >= vend) { err = INF_ERR_SIZE; break; }
        W.put((uint32_t)s);
      } else if (s == 256) break;
      else {
        s -= 257;
        if (s >= 29) { err = INF_ERR_DATA; break; }
        uint32_t lb, db;
        int le, de;
        len_code(s, lb, le);
        const uint32_t len = lb + br.bits(le);
        br.refill();
        const int ds = huff_decode(br, CD, sym_d);
        if (ds < 0 || ds >= 30) { err = INF_ERR_DATA; break; }
        dist_code(ds, db, de);
        const uint32_t dist = db + br.bits(de);
        if (dist > W.v - W.a0 || W.v + len > vend) { err = INF_ERR_DATA; break; }
        lz_copy(W, dist, len);
      }
      if (br.overrun()) { err = INF_ERR_DATA; break; }
    }
  }
  if (!err) {
    W.flush_all();
    if (W.v != vend) err = INF_ERR_SIZE;
  }
  return err;
}

}  // namespace strl
