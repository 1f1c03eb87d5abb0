// fast_inflate.cpp -- see fast_inflate.h.  RFC 1951 decoder for whole BGZF blocks.
//
// Table entry (32 bits):  payload << 16 | flags << 12 | extra << 8 | len
//   literal        : payload = the byte,            flags = LIT, len = code length
//   two literals   : payload = byte0 | byte1 << 8,  flags = LIT, extra = 1, len = both code lengths (root table only): the
//                    lookup -> shift -> lookup chain is what bounds literal-heavy streams (BAM SEQ bytes), so a root
//                    index that holds two whole literal codes yields both with one lookup
//   end of block   : flags = EOB
//   length / dist  : payload = base value, extra = number of extra bits, len = code length
//   subtable link  : payload = index of the subtable, extra = its index bits, len = root bits, flags = SUB
//   unused code    : flags = BAD
// Codes longer than the root bits resolve through one subtable (entries carry the bits that follow the root bits).
#include "fast_inflate.h"

#include <string.h>

namespace strl {
namespace {

constexpr int LL_BITS = 11, D_BITS = 8, PRE_BITS = 7;
constexpr uint32_t F_LIT = 1u << 12, F_EOB = 2u << 12, F_SUB = 4u << 12, F_BAD = 8u << 12;
constexpr int LL_SIZE = (1 << LL_BITS) + 288 * 16;   // every long code owns at most 2^(15 - 11) subtable slots
constexpr int D_SIZE = (1 << D_BITS) + 32 * 128;
constexpr int K_LL = 0, K_DIST = 1, K_PRE = 2;

inline uint32_t mk(uint32_t payload, uint32_t flags, uint32_t extra, uint32_t len) { return payload << 16 | flags | extra << 8 | len; }

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t symbol_entry(int kind, int s, uint32_t len) {
  if (kind == K_LL) {
    if (s < 256) return mk((uint32_t)s, F_LIT, 0, len);
    if (s == 256) return mk(0, F_EOB, 0, len);
    if (s <= 285) return mk(LEN_BASE[s - 257], 0, LEN_EXTRA[s - 257], len);
    return mk(0, F_BAD, 0, len);
  }
  if (kind == K_DIST) return s < 30 ? mk(DIST_BASE[s], 0, DIST_EXTRA[s], len) : mk(0, F_BAD, 0, len);
  return mk((uint32_t)s, 0, 0, len);
}

// Canonical Huffman decode table from code lengths.  false: over-subscribed, or incomplete in a way zlib rejects.
bool build_table(const uint8_t *lens, int n, int root, uint32_t *tab, int cap, int kind) {
  uint16_t count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  const int used = n - count[0];
  const uint32_t bad = mk(0, F_BAD, 0, 1);
  for (int i = 0; i < (1 << root); ++i) tab[i] = bad;
  if (used == 0) return kind == K_DIST;             // a block of literals only may leave the distance code empty
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) return false;
  }
  if (left > 0 && (kind == K_PRE || !(used == 1 && count[1] == 1))) return false;
  uint16_t next[16];
  uint32_t code = 0;
  count[0] = 0;
  for (int l = 1; l <= 15; ++l) {
    code = (code + count[l - 1]) << 1;
    next[l] = (uint16_t)code;
  }
  uint16_t rev[320];
  const uint32_t rmask = (1u << root) - 1u;
  int pos = 1 << root;
  // pass 1: bit-reversed codes; the widest code under every root prefix decides its subtable's size
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    uint32_t c = next[l]++, r = 0;
    for (int b = 0; b < l; ++b) { r = (r << 1) | (c & 1u); c >>= 1; }
    rev[s] = (uint16_t)r;
    if (l > root) {
      uint32_t &e = tab[r & rmask];
      const uint32_t need = (uint32_t)(l - root);
      if (!(e & F_SUB)) e = mk(0, F_SUB, need, (uint32_t)root);
      else if (((e >> 8) & 15u) < need) e = mk(0, F_SUB, need, (uint32_t)root);
    }
  }
  // place the subtables
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (l <= root) continue;
    uint32_t &e = tab[rev[s] & rmask];
    if (e >> 16) continue;                          // already placed (index 0 is never a subtable)
    const uint32_t b = (e >> 8) & 15u;
    if (pos + (1 << b) > cap) return false;
    e |= (uint32_t)pos << 16;
    for (int i = 0; i < (1 << b); ++i) tab[pos + i] = bad;
    pos += 1 << b;
  }
  // pass 2: fill
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t r = rev[s];
    if (l <= root) {
      const uint32_t e = symbol_entry(kind, s, (uint32_t)l);
      for (uint32_t i = r; i < (1u << root); i += 1u << l) tab[i] = e;
    } else {
      const uint32_t link = tab[r & rmask], base = link >> 16, b = (link >> 8) & 15u, ls = (uint32_t)(l - root);
      const uint32_t e = symbol_entry(kind, s, ls);
      for (uint32_t i = r >> root; i < (1u << b); i += 1u << ls) tab[base + i] = e;
    }
  }
  // two literals per lookup where the root index holds two whole literal codes
  if (kind == K_LL) {
    for (uint32_t i = (1u << root); i-- > 0;) {          // descending: tab[i >> l1] is still a single-literal entry
      const uint32_t e1 = tab[i];
      if (!(e1 & F_LIT)) continue;
      const uint32_t l1 = e1 & 0xffu;
      if (l1 >= (uint32_t)root) continue;
      const uint32_t e2 = tab[i >> l1];               // the bits behind the first code, zero-extended
      if (!(e2 & F_LIT)) continue;
      const uint32_t l2 = e2 & 0xffu;
      if (l1 + l2 > (uint32_t)root) continue;         // the second code is not fully inside the index
      tab[i] = mk((e1 >> 16) | ((e2 >> 16) << 8), F_LIT, 1, l1 + l2);
    }
  }
  return true;
}

struct Decoder {
  uint32_t ll[LL_SIZE];
  uint32_t d[D_SIZE];
  uint32_t pre[1 << PRE_BITS];
  uint32_t fixed_ll[1 << LL_BITS];     // fixed codes are at most 9 / 5 bits long: no subtables
  uint32_t fixed_d[1 << D_BITS];
  bool fixed_ready = false;
  void fixed() {
    if (fixed_ready) return;
    uint8_t lens[288];
    for (int i = 0; i < 144; ++i) lens[i] = 8;
    for (int i = 144; i < 256; ++i) lens[i] = 9;
    for (int i = 256; i < 280; ++i) lens[i] = 7;
    for (int i = 280; i < 288; ++i) lens[i] = 8;
    build_table(lens, 288, LL_BITS, fixed_ll, 1 << LL_BITS, K_LL);
    for (int i = 0; i < 32; ++i) lens[i] = 5;
    build_table(lens, 32, D_BITS, fixed_d, 1 << D_BITS, K_DIST);
    fixed_ready = true;
  }
};

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }   // little-endian host (x86-64)
inline void put2(uint8_t *d, uint32_t e) { const uint16_t v = (uint16_t)(e >> 16); memcpy(d, &v, 2); }   // (a lone literal's second byte is overwritten next)
inline void copy8(uint8_t *d, const uint8_t *s) { uint64_t v; memcpy(&v, s, 8); memcpy(d, &v, 8); }

}  // namespace

int fast_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
  static thread_local Decoder D;
  const uint8_t *in_next = in, *const in_end = in + in_len, *const in_lim = in_end + 8;
  uint8_t *out_next = out, *const out_end = out + out_len;
  uint64_t bitbuf = 0;
  uint32_t bitcnt = 0;
  bool overrun = false;
  // byte-wise refill to >= 56 bits; past the 8 readable bytes behind the stream it feeds zeros and remembers it
  auto refill = [&]() {
    while (bitcnt < 56) {
      if (in_next < in_lim) bitbuf |= (uint64_t)*in_next++ << bitcnt;
      else overrun = true;
      bitcnt += 8;
    }
  };
  auto take = [&](uint32_t n) -> uint32_t {
    const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1ull));
    bitbuf >>= n;
    bitcnt -= n;
    return v;
  };
  uint32_t bfinal;
  do {
    refill();
    bfinal = take(1);
    const uint32_t btype = take(2);
    const uint32_t *ll, *dt;
    if (btype == 0) {
      // stored: drop to the byte boundary, give back the whole bytes still buffered
      if (overrun) return 1;
      bitcnt &= ~7u;
      in_next -= bitcnt >> 3;
      bitbuf = 0; bitcnt = 0;
      if (in_end - in_next < 4) return 1;
      const uint32_t len = in_next[0] | (in_next[1] << 8), nlen = in_next[2] | (in_next[3] << 8);
      in_next += 4;
      if ((len ^ nlen) != 0xffffu) return 1;
      if ((size_t)(in_end - in_next) < len || (size_t)(out_end - out_next) < len) return 1;
      memcpy(out_next, in_next, len);
      in_next += len; out_next += len;
      continue;
    }
    if (btype == 1) {
      D.fixed();
      ll = D.fixed_ll; dt = D.fixed_d;
    } else if (btype == 2) {
      const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
      if (hlit > 286 || hdist > 30) return 1;
      uint8_t lens[320];
      memset(lens, 0, 19);
      refill();
      for (uint32_t i = 0; i < hclen; ++i) {
        if (bitcnt < 3) refill();
        lens[PRE_ORDER[i]] = (uint8_t)take(3);
      }
      if (!build_table(lens, 19, PRE_BITS, D.pre, 1 << PRE_BITS, K_PRE)) return 1;
      uint32_t i = 0;
      const uint32_t total = hlit + hdist;
      while (i < total) {
        refill();
        const uint32_t e = D.pre[bitbuf & ((1u << PRE_BITS) - 1u)];
        if (e & F_BAD) return 1;
        take(e & 0xffu);
        const uint32_t sym = e >> 16;
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        uint32_t rep, val = 0;
        if (sym == 16) {
          if (i == 0) return 1;
          val = lens[i - 1];
          rep = 3 + take(2);
        } else if (sym == 17) rep = 3 + take(3);
        else rep = 11 + take(7);
        if (i + rep > total) return 1;
        memset(lens + i, (int)val, rep);
        i += rep;
      }
      if (lens[256] == 0) return 1;                 // no end-of-block code
      uint8_t dl[32];
      memcpy(dl, lens + hlit, hdist);
      if (!build_table(lens, (int)hlit, LL_BITS, D.ll, LL_SIZE, K_LL)) return 1;
      if (!build_table(dl, (int)hdist, D_BITS, D.d, D_SIZE, K_DIST)) return 1;
      ll = D.ll; dt = D.d;
    } else return 1;
    if (overrun) return 1;

    constexpr uint32_t LLM = (1u << LL_BITS) - 1u, DM = (1u << D_BITS) - 1u;
    bool done = false;
    // ---- fast loop ----
    // Software-pipelined: the table entry of the NEXT symbol is looked up before the current match is copied, so the
    // lookup's latency hides behind the copy.  Bounds: at the top at least 16 stream bytes are left (two 8-byte loads per
    // round stay inside the stream + its 8 readable bytes) and the output has room for three lookups of literals, the
    // longest match and the copies' overshoot.  Short matches are copied with two unconditional word moves: most
    // matches in BAM blocks are 3-16 bytes, a loop per match is a mispredicted branch per match.
#define STRL_REFILL() do { bitbuf |= load64(in_next) << bitcnt; in_next += (63 - bitcnt) >> 3; bitcnt |= 56; } while (0)
    if ((in_end - in_next) >= 16 && (size_t)(out_end - out_next) >= 6 + 1 + 258 + 32) {
      STRL_REFILL();
      uint32_t e = ll[bitbuf & LLM];
      for (;;) {
        if (e & F_LIT) {
          bitbuf >>= (e & 0xffu); bitcnt -= (e & 0xffu);
          put2(out_next, e);
          out_next += 1 + ((e >> 8) & 1u);
          e = ll[bitbuf & LLM];
          if (e & F_LIT) {
            bitbuf >>= (e & 0xffu); bitcnt -= (e & 0xffu);
            put2(out_next, e);
            out_next += 1 + ((e >> 8) & 1u);
            e = ll[bitbuf & LLM];
            if (e & F_LIT) {
              bitbuf >>= (e & 0xffu); bitcnt -= (e & 0xffu);
              put2(out_next, e);
              out_next += 1 + ((e >> 8) & 1u);
              STRL_REFILL();
              e = ll[bitbuf & LLM];
              if ((in_end - in_next) < 16 || (size_t)(out_end - out_next) < 6 + 1 + 258 + 32) break;
              continue;
            }
          }
          STRL_REFILL();                             // (the low bits e was looked up with do not change)
        }
        if (e & F_SUB) {
          bitbuf >>= LL_BITS; bitcnt -= LL_BITS;
          e = ll[(e >> 16) + (uint32_t)(bitbuf & ((1u << ((e >> 8) & 15u)) - 1u))];
          if (e & F_LIT) {
            bitbuf >>= (e & 0xffu); bitcnt -= (e & 0xffu);
            *out_next++ = (uint8_t)(e >> 16);
            STRL_REFILL();
            e = ll[bitbuf & LLM];
            if ((in_end - in_next) < 16 || (size_t)(out_end - out_next) < 6 + 1 + 258 + 32) break;
            continue;
          }
        }
        if (e & (F_EOB | F_BAD)) {
          if (e & F_BAD) return 1;
          bitbuf >>= (e & 0xffu); bitcnt -= (e & 0xffu);
          done = true;
          break;
        }
        bitbuf >>= (e & 0xffu);
        const uint32_t lx = (e >> 8) & 15u;
        const uint32_t len = (e >> 16) + (uint32_t)(bitbuf & ((1u << lx) - 1u));
        bitbuf >>= lx;
        bitcnt -= (e & 0xffu) + lx;
        uint32_t de = dt[bitbuf & DM];
        if (de & F_SUB) {
          bitbuf >>= D_BITS; bitcnt -= D_BITS;
          de = dt[(de >> 16) + (uint32_t)(bitbuf & ((1u << ((de >> 8) & 15u)) - 1u))];
        }
        if (de & F_BAD) return 1;
        bitbuf >>= (de & 0xffu);
        const uint32_t dx = (de >> 8) & 15u;
        const uint32_t dist = (de >> 16) + (uint32_t)(bitbuf & ((1u << dx) - 1u));
        bitbuf >>= dx;
        bitcnt -= (de & 0xffu) + dx;
        STRL_REFILL();
        e = ll[bitbuf & LLM];                        // the next symbol's entry, on its way while the match is copied
        if (dist > (size_t)(out_next - out)) return 1;
        const uint8_t *src = out_next - dist;
        uint8_t *dst = out_next;
        out_next += len;
        if (dist >= 8) {
          copy8(dst, src); copy8(dst + 8, src + 8);
          if (len > 16) {
            dst += 16; src += 16;
            do { copy8(dst, src); dst += 8; src += 8; } while (dst < out_next);
          }
        } else if (dist == 1) {
          const uint64_t v = 0x0101010101010101ull * *src;
          do { memcpy(dst, &v, 8); dst += 8; } while (dst < out_next);
        } else {
          do { *dst++ = *src++; } while (dst < out_next);
        }
        if ((in_end - in_next) < 16 || (size_t)(out_end - out_next) < 6 + 1 + 258 + 32) break;
      }
    }
#undef STRL_REFILL
    // ---- careful loop: the last few hundred bytes of the block's output / the last bytes of its input ----
    while (!done) {
      refill();
      uint32_t e = ll[bitbuf & LLM];
      if (e & F_SUB) {
        take(LL_BITS);
        e = ll[(e >> 16) + (uint32_t)(bitbuf & ((1u << ((e >> 8) & 15u)) - 1u))];
      }
      if (e & F_BAD) return 1;
      take(e & 0xffu);
      if (e & F_LIT) {
        const uint32_t nl = 1 + ((e >> 8) & 1u);
        if ((size_t)(out_end - out_next) < nl) return 1;
        *out_next++ = (uint8_t)(e >> 16);
        if (nl == 2) *out_next++ = (uint8_t)(e >> 24);
        continue;
      }
      if (e & F_EOB) break;
      const uint32_t len = (e >> 16) + take((e >> 8) & 15u);
      refill();
      uint32_t de = dt[bitbuf & DM];
      if (de & F_SUB) {
        take(D_BITS);
        de = dt[(de >> 16) + (uint32_t)(bitbuf & ((1u << ((de >> 8) & 15u)) - 1u))];
      }
      if (de & F_BAD) return 1;
      take(de & 0xffu);
      const uint32_t dist = (de >> 16) + take((de >> 8) & 15u);
      if (dist > (size_t)(out_next - out) || len > (size_t)(out_end - out_next)) return 1;
      const uint8_t *src = out_next - dist;
      for (uint32_t k = 0; k < len; ++k) out_next[k] = src[k];
      out_next += len;
      if (overrun) return 1;
    }
    if (overrun) return 1;
    // bits consumed must come from the stream itself, not from the 8 bytes behind it
    if ((size_t)(in_next - in) > in_len + (bitcnt >> 3)) return 1;
  } while (!bfinal);
  return out_next == out_end ? 0 : 1;
}

}  // namespace strl
