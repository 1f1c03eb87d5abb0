// cram_codecs.cpp -- the CRAM 3.1 block codecs a default `samtools view -C` of the htslib 1.22 line writes (the reference pins
// htslib 1.22.1: .github/workflows/ci.yml:11-12; its "normal" profile = rANS Nx16 for the numeric and base series, the name
// tokeniser for RN): block compression methods 5 (rANS Nx16) and 8 (tok3), restated from the CRAM codecs specification
// (CRAMcodecs, hts-specs): section 3 (rANS Nx16: order 0 / 1, 4 or 32 interleaved states, 16-bit renormalisation, the PACK /
// RLE / STRIPE / CAT / NOSZ transforms) and section 5 (name tokenisation).  Methods 6 (adaptive arithmetic coder) and 7
// (fqzcomp) belong to the "small" / "archive" profiles; fqzcomp only ever holds qualities, which strling never reads (their
// blocks are not decompressed at all: cram_reader.cpp, Ctx::stream), the arithmetic coder is refused with the way out.
// htslib / htscodecs are not in this image: nothing here has seen a file they wrote (verify/run_reference.sh converts the kit's
// BAM with samtools for whoever has it).
#include "cram_codecs.h"
#include <string.h>
#include <algorithm>

namespace strl {
namespace {

struct In {
  const uint8_t *p, *e;
  bool ok = true;
  uint8_t u8() { if (p >= e) { ok = false; return 0; } return *p++; }
  uint32_t u32le() { if (e - p < 4) { ok = false; p = e; return 0; } uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
  uint32_t u7() {                       // variable-length integer, most significant 7-bit group first (spec 1.2 / htscodecs var_get_u32)
    uint32_t v = 0;
    for (int k = 0; k < 5; ++k) {
      const uint8_t c = u8();
      v = (v << 7) | (c & 0x7fu);
      if (!(c & 0x80u)) return v;
    }
    ok = false;
    return v;
  }
};

constexpr uint32_t RANS16_L = 1u << 15;
constexpr size_t MAX_OUT = (size_t)1 << 30;
// what one rANS Nx16 stream may claim to expand to right now: MAX_OUT for a block (whose size the container states), far less for
// the sub-streams of a name tokeniser block, which state their own (cram_tok3_decode)
thread_local size_t g_stream_cap = MAX_OUT;

// the symbols that occur: sym, then either the next symbol or -- when that is sym + 1 -- a count of further consecutive ones
bool read_alphabet(In &r, bool A[256]) {
  memset(A, 0, 256);
  int rle = 0;
  uint32_t j = r.u8();
  do {
    A[j] = true;
    if (!rle && r.p < r.e && j + 1 == *r.p) { j = r.u8(); rle = r.u8(); }
    else if (rle) { --rle; ++j; if (j > 255) return false; }
    else j = r.u8();
  } while (j && r.ok);
  return r.ok;
}

// frequencies that sum to less than 2^bits are scaled up by the power of two that gets them there
bool normalise(uint32_t F[256], uint32_t bits) {
  uint64_t tot = 0;
  for (int j = 0; j < 256; ++j) tot += F[j];
  if (tot == 0 || tot == (1ull << bits)) return true;
  if (tot > (1ull << bits)) return false;
  int shift = 0;
  while (tot < (1ull << bits)) { tot *= 2; ++shift; }
  for (int j = 0; j < 256; ++j) F[j] <<= shift;
  return true;
}

struct Tab0 { uint32_t F[256], C[256]; uint8_t R[4096]; };

bool decode_order0(In &r, uint8_t *out, size_t n, int N, std::string &err) {
  static thread_local Tab0 t;
  bool A[256];
  if (!read_alphabet(r, A)) { err = "malformed rANS Nx16 alphabet"; return false; }
  for (int j = 0; j < 256; ++j) t.F[j] = A[j] ? r.u7() : 0;
  if (!r.ok || !normalise(t.F, 12)) { err = "malformed rANS Nx16 frequency table"; return false; }
  uint32_t x = 0;
  for (int j = 0; j < 256; ++j) {
    t.C[j] = x;
    if (x + t.F[j] > 4096) { err = "malformed rANS Nx16 frequency table"; return false; }
    memset(t.R + x, j, t.F[j]);
    x += t.F[j];
  }
  if (x < 4096) memset(t.R + x, 0, 4096 - x);
  uint32_t R[32];
  for (int k = 0; k < N; ++k) R[k] = r.u32le();
  if (!r.ok) { err = "truncated rANS Nx16 block"; return false; }
  const size_t full = n - n % (size_t)N;
  const uint8_t *p = r.p, *e = r.e;
  for (size_t i = 0; i < full; i += (size_t)N)
    for (int k = 0; k < N; ++k) {
      const uint32_t m = R[k] & 4095u;
      const uint8_t c = t.R[m];
      out[i + (size_t)k] = c;
      uint32_t v = t.F[c] * (R[k] >> 12) + m - t.C[c];
      if (v < RANS16_L && e - p >= 2) { v = (v << 16) | (uint32_t)p[0] | ((uint32_t)p[1] << 8); p += 2; }
      R[k] = v;
    }
  for (size_t k = 0; k < n - full; ++k) out[full + k] = t.R[R[k] & 4095u];
  r.p = p;
  return true;
}

struct Tab1 { uint32_t F[256][256]; uint32_t C[256][256]; std::vector<uint8_t> R; uint8_t used[256]; };

bool rans_nx16(In &r, size_t expect, bool have_expect, std::vector<uint8_t> &out, std::string &err, int depth);

bool decode_order1(In &r, uint8_t *out, size_t n, int N, std::string &err) {
  static thread_local Tab1 *tp = nullptr;
  if (!tp) tp = new Tab1();
  Tab1 &t = *tp;
  const uint8_t comp = r.u8();
  const uint32_t shift = comp >> 4;
  if (!r.ok || shift < 1 || shift > 12) { err = "malformed rANS Nx16 order-1 header"; return false; }
  std::vector<uint8_t> table;
  In tr = r;
  if (comp & 1) {                         // the frequency table itself is an order-0 stream
    const uint32_t ulen = r.u7(), clen = r.u7();
    if (!r.ok || clen > (size_t)(r.e - r.p) || ulen > (1u << 20)) { err = "malformed rANS Nx16 order-1 table"; return false; }
    In sub{r.p, r.p + clen};
    table.resize(ulen);
    if (!decode_order0(sub, table.data(), ulen, 4, err)) return false;
    r.p += clen;
    tr = In{table.data(), table.data() + table.size()};
  }
  bool A[256];
  if (!read_alphabet(tr, A)) { err = "malformed rANS Nx16 alphabet"; return false; }
  const size_t slots = (size_t)1 << shift;
  t.R.resize(256 * slots);
  memset(t.used, 0, sizeof t.used);
  for (int i = 0; i < 256; ++i) {
    if (!A[i]) continue;
    uint32_t *F = t.F[i];
    memset(F, 0, 256 * 4);
    int run = 0;
    for (int j = 0; j < 256; ++j) {
      if (!A[j]) continue;
      if (run) { --run; continue; }
      F[j] = tr.u7();
      if (!F[j]) run = tr.u8();
    }
    if (!tr.ok || !normalise(F, shift)) { err = "malformed rANS Nx16 order-1 frequencies"; return false; }
    uint32_t x = 0;
    uint8_t *R = t.R.data() + (size_t)i * slots;
    for (int j = 0; j < 256; ++j) {
      t.C[i][j] = x;
      if (x + F[j] > slots) { err = "malformed rANS Nx16 order-1 frequencies"; return false; }
      memset(R + x, j, F[j]);
      x += F[j];
    }
    if (x < slots) memset(R + x, 0, slots - x);
    t.used[i] = 1;
  }
  if (!(comp & 1)) r.p = tr.p;
  uint32_t R[32];
  uint8_t last[32];
  for (int k = 0; k < N; ++k) { R[k] = r.u32le(); last[k] = 0; }
  if (!r.ok) { err = "truncated rANS Nx16 block"; return false; }
  const size_t seg = n / (size_t)N;
  const uint32_t mask = (uint32_t)slots - 1u;
  const uint8_t *p = r.p, *e = r.e;
  auto step = [&](int k, size_t at) -> bool {
    const uint32_t ctx = last[k];
    if (!t.used[ctx]) return false;                       // a context the table has no row for
    const uint32_t m = R[k] & mask;
    const uint8_t c = t.R[(size_t)ctx * slots + m];
    out[at] = c;
    uint32_t v = t.F[ctx][c] * (R[k] >> shift) + m - t.C[ctx][c];
    if (v < RANS16_L && e - p >= 2) { v = (v << 16) | (uint32_t)p[0] | ((uint32_t)p[1] << 8); p += 2; }
    R[k] = v;
    last[k] = c;
    return true;
  };
  // state k decodes the k-th of N equal segments; what is left over belongs to the last state
  for (size_t i = 0; i < seg; ++i)
    for (int k = 0; k < N; ++k)
      if (!step(k, (size_t)k * seg + i)) { err = "rANS Nx16 order-1: symbol in a context without frequencies"; return false; }
  for (size_t at = seg * (size_t)N; at < n; ++at)
    if (!step(N - 1, at)) { err = "rANS Nx16 order-1: symbol in a context without frequencies"; return false; }
  r.p = p;
  return true;
}

// one rANS Nx16 stream (spec 3.1): flags, sizes, transform metadata, the entropy-coded (or stored) bytes; transforms undone
// in the order RLE, PACK
bool rans_nx16(In &r, size_t expect, bool have_expect, std::vector<uint8_t> &out, std::string &err, int depth) {
  if (depth > 4) { err = "rANS Nx16 streams nested too deeply"; return false; }
  const uint8_t flags = r.u8();
  size_t len = expect;
  if (!(flags & 0x10)) {
    len = r.u7();
    if (have_expect && len != expect) { err = "rANS Nx16 block: the stream's own size differs from the block's"; return false; }
  } else if (!have_expect) { err = "rANS Nx16 stream without a size"; return false; }
  if (!r.ok || len > g_stream_cap) { err = "malformed rANS Nx16 block"; return false; }
  const int N = (flags & 0x04) ? 32 : 4;
  if (flags & 0x08) {                     // STRIPE: byte i of the output comes from sub-stream i mod X
    const uint32_t X = r.u8();
    if (!X) { err = "malformed rANS Nx16 stripe"; return false; }
    std::vector<uint32_t> clen(X);
    for (auto &c : clen) c = r.u7();
    if (!r.ok) { err = "malformed rANS Nx16 stripe"; return false; }
    out.resize(len);
    std::vector<uint8_t> sub;
    for (uint32_t j = 0; j < X; ++j) {
      const size_t ul = len / X + ((len % X) > j ? 1 : 0);
      if (clen[j] > (size_t)(r.e - r.p)) { err = "truncated rANS Nx16 stripe"; return false; }
      In s{r.p, r.p + clen[j]};
      if (!rans_nx16(s, ul, true, sub, err, depth + 1)) return false;
      r.p += clen[j];
      for (size_t i = 0; i < ul; ++i) out[i * X + j] = sub[i];
    }
    return true;
  }
  // PACK metadata: the symbols that occur, and the size of the packed stream
  const size_t pack_len = len;
  uint8_t P[256];
  uint32_t nsym = 0;
  if (flags & 0x80) {
    nsym = r.u8();
    if (nsym > 16 || !nsym) {
      if (nsym == 0) { /* (no symbols: an empty stream) */ } else { err = "rANS Nx16 PACK with more than 16 symbols"; return false; }
    }
    for (uint32_t i = 0; i < nsym; ++i) P[i] = r.u8();
    len = r.u7();
    if (!r.ok || len > g_stream_cap) { err = "malformed rANS Nx16 PACK header"; return false; }
  }
  // RLE metadata: which symbols carry run lengths, the run lengths, and the size of the literal stream
  const size_t rle_len = len;
  std::vector<uint8_t> rle_meta;
  bool has_run[256];
  In runs{nullptr, nullptr};
  if (flags & 0x40) {
    const uint32_t mlen = r.u7();
    len = r.u7();
    if (!r.ok || len > g_stream_cap) { err = "malformed rANS Nx16 RLE header"; return false; }
    if (mlen & 1) {
      const size_t ml = mlen / 2;
      if (ml > (size_t)(r.e - r.p)) { err = "truncated rANS Nx16 RLE metadata"; return false; }
      rle_meta.assign(r.p, r.p + ml);
      r.p += ml;
    } else {
      const uint32_t cl = r.u7();
      if (!r.ok || cl > (size_t)(r.e - r.p) || mlen / 2 > (1u << 28)) { err = "truncated rANS Nx16 RLE metadata"; return false; }
      In s{r.p, r.p + cl};
      rle_meta.resize(mlen / 2);
      if (!decode_order0(s, rle_meta.data(), rle_meta.size(), 4, err)) return false;
      r.p += cl;
    }
    runs = In{rle_meta.data(), rle_meta.data() + rle_meta.size()};
    memset(has_run, 0, sizeof has_run);
    uint32_t m = runs.u8();
    if (m == 0) m = 256;
    for (uint32_t i = 0; i < m; ++i) has_run[runs.u8()] = true;
    if (!runs.ok) { err = "malformed rANS Nx16 RLE metadata"; return false; }
  }
  // the bytes themselves
  std::vector<uint8_t> data(len);
  if (flags & 0x20) {                     // CAT: stored
    if (len > (size_t)(r.e - r.p)) { err = "truncated rANS Nx16 block"; return false; }
    memcpy(data.data(), r.p, len);
    r.p += len;
  } else if (len) {
    if (!((flags & 1) ? decode_order1(r, data.data(), len, N, err) : decode_order0(r, data.data(), len, N, err))) return false;
  }
  if (flags & 0x40) {
    std::vector<uint8_t> un(rle_len);
    size_t o = 0;
    for (size_t i = 0; i < data.size(); ++i) {
      const uint8_t c = data[i];
      size_t rep = 1;
      if (has_run[c]) rep += runs.u7();
      if (!runs.ok || o + rep > rle_len) { err = "rANS Nx16 RLE: runs reach past the stream's size"; return false; }
      memset(un.data() + o, c, rep);
      o += rep;
    }
    if (o != rle_len) { err = "rANS Nx16 RLE: the runs do not add up to the stream's size"; return false; }
    data.swap(un);
  }
  if (flags & 0x80) {
    std::vector<uint8_t> un(pack_len);
    if (nsym <= 1) {
      memset(un.data(), nsym ? P[0] : 0, pack_len);
    } else {
      const int bits = nsym <= 2 ? 1 : nsym <= 4 ? 2 : 4, per = 8 / bits;
      if (data.size() < (pack_len + (size_t)per - 1) / (size_t)per) { err = "rANS Nx16 PACK: too few packed bytes"; return false; }
      const uint32_t mask = (1u << bits) - 1u;
      for (size_t i = 0, j = 0; i < pack_len; ++j) {
        uint32_t v = data[j];
        for (int k = 0; k < per && i < pack_len; ++k, ++i) {
          const uint32_t s = v & mask;
          if (s >= nsym) { err = "rANS Nx16 PACK: symbol outside the map"; return false; }
          un[i] = P[s];
          v >>= bits;
        }
      }
    }
    data.swap(un);
  }
  out.swap(data);
  return true;
}

// ---- name tokeniser (spec section 5) ---------------------------------------------------------------------------------------
enum { T_TYPE = 0, T_ALPHA = 1, T_CHAR = 2, T_DIGITS0 = 3, T_DZLEN = 4, T_DUP = 5, T_DIFF = 6, T_DIGITS = 7, T_DELTA = 8, T_DELTA0 = 9, T_MATCH = 10, T_NOP = 11, T_END = 12 };
struct Stream { std::vector<uint8_t> d; size_t at = 0; bool have = false; };
struct Tok { uint8_t type = 0; uint32_t val = 0, str = 0, len = 0; };      // str / len: ALPHA = offset / length in the output, DIGITS0 = width

}  // namespace

bool cram_rans_nx16_decode(const uint8_t *in, size_t in_len, size_t expect, std::vector<uint8_t> &out, std::string &err) {
  In r{in, in + in_len};
  if (!rans_nx16(r, expect, true, out, err, 0)) { if (err.empty()) err = "malformed rANS Nx16 block"; return false; }
  if (out.size() != expect) { err = "rANS Nx16 block of the wrong size"; return false; }
  return true;
}

bool cram_tok3_decode(const uint8_t *in, size_t in_len, size_t expect, std::vector<uint8_t> &out, std::string &err) {
  In r{in, in + in_len};
  const uint32_t ulen = r.u32le(), n_names = r.u32le();
  const uint8_t use_arith = r.u8();
  if (!r.ok || ulen != expect || ulen > MAX_OUT) { err = "malformed name tokeniser block"; return false; }
  // every name costs at least its NUL: a count beyond the block's size is a hostile header, and everything below is sized by it
  if (n_names > ulen) { err = "malformed name tokeniser block (more names than bytes)"; return false; }
  // A token stream holds at most a byte or a 32-bit number per name, or the names' own text: nothing useful is longer than this,
  // and a stream that claims more from its few input bytes is refused before anything is allocated for it.
  struct CapGuard { size_t was; ~CapGuard() { g_stream_cap = was; } } guard{g_stream_cap};
  g_stream_cap = std::min<size_t>(MAX_OUT, (size_t)ulen + 4 * (size_t)n_names + 64);
  if (use_arith) {
    err = "the CRAM's read names are compressed with the adaptive arithmetic coder (CRAM 3.1 `archive` / `small` profiles): not supported by this build "
          "(re-encode with `samtools view -C --output-fmt-option version=3.1` at the default profile, or `version=3.0`)";
    return false;
  }
  // token streams: [position][type]
  std::vector<std::vector<Stream>> S;
  int tnum = -1;
  while (r.p < r.e) {
    const uint8_t tt = r.u8();
    const int type = tt & 15;
    if (tt & 128) {
      if (++tnum > 128) { err = "name tokeniser block: more than 128 token positions"; return false; }
      S.emplace_back(16);
      if (type != T_TYPE) {               // every name has this token type at this position, except that from the second on it is "match"
        Stream &ts = S[(size_t)tnum][T_TYPE];
        ts.d.assign(std::max<uint32_t>(n_names, 1), (uint8_t)T_MATCH);
        ts.d[0] = (uint8_t)type;
        ts.have = true;
      }
    }
    if (tnum < 0) { err = "malformed name tokeniser block (no token position)"; return false; }
    Stream &dst = S[(size_t)tnum][(size_t)type];
    if (tt & 64) {                        // a copy of an earlier stream
      const uint32_t j = r.u8(), k = r.u8();
      if (!r.ok || j > (uint32_t)tnum || k > 15 || !S[j][k].have || (j == (uint32_t)tnum && k == (uint32_t)type)) { err = "malformed name tokeniser block (bad duplicate stream)"; return false; }
      dst.d = S[j][k].d; dst.at = 0; dst.have = true;
      continue;
    }
    const uint32_t clen = r.u7();
    if (!r.ok || clen > (size_t)(r.e - r.p)) { err = "truncated name tokeniser block"; return false; }
    In s{r.p, r.p + clen};
    if (!rans_nx16(s, 0, false, dst.d, err, 0)) return false;
    dst.at = 0; dst.have = true;
    r.p += clen;
  }
  out.clear();
  out.reserve(ulen);
  std::vector<std::vector<Tok>> hist(n_names);
  std::vector<std::pair<size_t, size_t>> where(n_names);          // each name's place in the output (without its NUL)
  auto byte_of = [&](int t, int type, uint32_t &v) -> bool {
    if ((size_t)t >= S.size()) return false;
    Stream &s = S[(size_t)t][(size_t)type];
    if (!s.have || s.at >= s.d.size()) return false;
    v = s.d[s.at++];
    return true;
  };
  auto int_of = [&](int t, int type, uint32_t &v) -> bool {
    if ((size_t)t >= S.size()) return false;
    Stream &s = S[(size_t)t][(size_t)type];
    if (!s.have || s.d.size() - s.at < 4 || s.at > s.d.size()) return false;
    memcpy(&v, s.d.data() + s.at, 4);
    s.at += 4;
    return true;
  };
  auto put_num = [&](uint32_t v, uint32_t width) {
    char buf[16];
    int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (uint32_t k = (uint32_t)n; k < width; ++k) out.push_back('0');
    while (n) out.push_back((uint8_t)buf[--n]);
  };
  const char *bad = "malformed name tokeniser block (token streams end early)";
  for (uint32_t c = 0; c < n_names; ++c) {
    uint32_t t0 = 0, dist = 0;
    if (!byte_of(0, T_TYPE, t0) || (t0 != T_DUP && t0 != T_DIFF) || !int_of(0, (int)t0, dist)) { err = bad; return false; }
    const uint32_t p = dist > c ? 0 : c - dist;
    const size_t start = out.size();
    std::vector<Tok> &H = hist[c];
    if (t0 == T_DUP) {
      if (p >= c) { err = "malformed name tokeniser block (a duplicate of nothing)"; return false; }
      const auto w = where[p];
      if (out.size() + w.second + 1 > ulen) { err = "name tokeniser block: names longer than the block says"; return false; }     // (and no reallocation below)
      H = hist[p];
      for (Tok &t : H) if (t.type == T_ALPHA) t.str = (uint32_t)(t.str - w.first + start);
      out.insert(out.end(), out.begin() + (long)w.first, out.begin() + (long)(w.first + w.second));   // (reserve()d: no reallocation while copying from itself)
      where[c] = {start, w.second};
      out.push_back(0);
      if (out.size() > ulen) { err = "name tokeniser block: names longer than the block says"; return false; }
      continue;
    }
    H.emplace_back();                     // token 0
    const std::vector<Tok> *PH = c ? &hist[p] : nullptr;
    for (int t = 1;; ++t) {
      if (t > 128) { err = "name tokeniser block: more than 128 tokens in a name"; return false; }
      uint32_t type = 0;
      if (!byte_of(t, T_TYPE, type)) { err = bad; return false; }
      Tok k;
      k.type = (uint8_t)type;
      bool end = false;
      switch (type) {
        case T_CHAR: { uint32_t v; if (!byte_of(t, T_CHAR, v)) { err = bad; return false; } out.push_back((uint8_t)v); k.val = v; break; }
        case T_ALPHA: {
          Stream &s = S[(size_t)t][T_ALPHA];
          if (!s.have) { err = bad; return false; }
          const uint8_t *b = s.d.data() + s.at, *z = static_cast<const uint8_t *>(memchr(b, 0, s.d.size() - s.at));
          if (!z) { err = bad; return false; }
          k.str = (uint32_t)out.size(); k.len = (uint32_t)(z - b);
          out.insert(out.end(), b, z);
          s.at += (size_t)(z - b) + 1;
          break;
        }
        case T_DIGITS0: { uint32_t v, w; if (!int_of(t, T_DIGITS0, v) || !byte_of(t, T_DZLEN, w)) { err = bad; return false; } put_num(v, w); k.val = v; k.str = w; break; }
        case T_DIGITS: { uint32_t v; if (!int_of(t, T_DIGITS, v)) { err = bad; return false; } put_num(v, 0); k.val = v; break; }
        case T_DELTA: case T_DELTA0: {
          uint32_t d;
          if (!byte_of(t, (int)type, d) || !PH || (size_t)t >= PH->size()) { err = "malformed name tokeniser block (a delta against nothing)"; return false; }
          const Tok &pt = (*PH)[(size_t)t];
          if (pt.type != (type == T_DELTA ? T_DIGITS : T_DIGITS0)) { err = "malformed name tokeniser block (a delta against a token that is not a number)"; return false; }
          k.val = pt.val + d;
          k.type = (uint8_t)(type == T_DELTA ? T_DIGITS : T_DIGITS0);
          k.str = pt.str;
          put_num(k.val, type == T_DELTA0 ? pt.str : 0);
          break;
        }
        case T_MATCH: {
          if (!PH || (size_t)t >= PH->size()) { err = "malformed name tokeniser block (a match against nothing)"; return false; }
          const Tok pt = (*PH)[(size_t)t];
          k = pt;
          switch (pt.type) {
            case T_CHAR: out.push_back((uint8_t)pt.val); break;
            case T_ALPHA: {
              if (out.size() + pt.len > ulen) { err = "name tokeniser block: names longer than the block says"; return false; }
              k.str = (uint32_t)out.size();
              const size_t a = pt.str;
              for (uint32_t q = 0; q < pt.len; ++q) { const uint8_t ch = out[a + q]; out.push_back(ch); }
              break;
            }
            case T_DIGITS: put_num(pt.val, 0); break;
            case T_DIGITS0: put_num(pt.val, pt.str); break;
            default: err = "malformed name tokeniser block (a match against a token without a value)"; return false;
          }
          break;
        }
        case T_NOP: break;
        case T_END: end = true; break;
        default: err = "malformed name tokeniser block (unknown token type)"; return false;
      }
      H.push_back(k);
      if (out.size() > ulen) { err = "name tokeniser block: names longer than the block says"; return false; }
      if (end) break;
    }
    where[c] = {start, out.size() - start};
    out.push_back(0);
  }
  if (out.size() != ulen) { err = "name tokeniser block: the names do not add up to the block's size"; return false; }
  return true;
}

}  // namespace strl
