// host_score.cpp -- the scorer for reads the kernels do not take (more than STRL_DEVICE_READ_LEN bases).
//
// The device scorer keeps one read per lane with byte-wide class counters, which is exact while a k-mer class cannot be seen
// more than 255 times: L / 2 <= 255.  The reference has no such bound -- its histograms are `Seqs[uint8]` whose `inc` wraps
// (utils.nim:192-195) -- so a longer read (a merged pair, a stray long record, a 2 x 600 library) is scored here, on the host,
// with exactly that arithmetic, and its words are merged into the device's results by record index (score.hip,
// long_reads_pass).  This is product code: it shares nothing with oracle/.
//
//   slide_by     utils.nim:10-34     one minimum-rotation code per NON-overlapping window of k bases
//   Seq.inc      utils.nim:192-195   uint8 bin, wrapping; the running arg-max moves when the bin just written EXCEEDS it
//   count        utils.nim:205-211
//   get_repeat   utils.nim:236-271   ladder k = 2..6, literal greedy recount (strutils.count), thresholds in float64
//   reduce_repeat utils.nim:220-233  homopolymer units collapse to one base, the count is multiplied
#include <algorithm>
#include <cstring>
#include <thread>
#include "common.h"
#include "host_score.h"

namespace strl {
namespace {

inline uint32_t code_of(char b) {        // brentp/nim-kmer's alphabet order C < A < T < G; anything else counts as A
  switch (b) { case 'C': return 0; case 'A': return 1; case 'T': return 2; case 'G': return 3; default: return 1; }
}

struct Ladder {
  uint8_t bins[4096];          // the largest table (k = 6); cleared per k over 4^k entries
  std::vector<uint8_t> codes;  // the read as 2-bit codes
};

// the class code of every window of k bases: the minimum over the window's k rotations (first base in the high bits)
template <int K> inline uint32_t window_class(const uint8_t *c) {
  constexpr uint32_t MASK = (1u << (2 * K)) - 1u;
  uint32_t f = 0;
  for (int j = 0; j < K; ++j) f = (f << 2) | c[j];
  uint32_t best = f;
  for (int j = 0; j < K - 1; ++j) {       // K - 1 further rotations; the K-th is the window again
    f = ((f << 2) | c[j]) & MASK;
    best = std::min(best, f);
  }
  return best;
}

template <int K> inline void count_k(Ladder &T, int L, int &count, long &imax) {
  constexpr int NB = 1 << (2 * K);
  memset(T.bins, 0, NB);
  imax = -1;
  const uint8_t *c = T.codes.data();
  for (int i = 0; i + K <= L; i += K) {
    const uint32_t e = window_class<K>(c + i);
    const uint8_t v = ++T.bins[e];                                   // wraps at 256: utils.nim:193 on a uint8
    if (imax < 0 || v > T.bins[imax]) imax = (long)e;                // :194-195
  }
  count = imax < 0 ? 0 : T.bins[imax];
}

// strutils.count(s, sub): greedy, left to right, non-overlapping, on the read's TEXT (an N or IUPAC letter matches nothing)
inline int literal_count(const char *s, int L, const char *u, int k) {
  int n = 0;
  for (int i = 0; i + k <= L;) {
    if (memcmp(s + i, u, (size_t)k) == 0) { ++n; i += k; }
    else ++i;
  }
  return n;
}

inline uint32_t pack(uint32_t code, int k, long count) {   // strling_amd.h "packed unit/count word", after reduce_repeat
  if (!k) return 0;
  const uint32_t b = code & 3u, rep = b * (((1u << (2 * k)) - 1u) / 3u);
  long c = count;
  uint32_t kk = (uint32_t)k, cc = code;
  if (code == rep) { kk = 1; cc = b; c = count * k; }     // utils.nim:220-233,271
  return cc | (kk << 12) | ((uint32_t)std::min<long>(c, 65535) << 16);     // (>= 256 is the doAssert of extract.nim:72 wherever it is looked at)
}

}  // namespace

void host_get_repeat(const char *read, int L, const double *p, int n_p, uint32_t *out) {
  for (int q = 0; q < n_p; ++q) out[q] = 0;
  int n_N = 0;
  for (int i = 0; i < L; ++i) n_N += read[i] == 'N';
  if (n_N > 20) return;                                              // utils.nim:238
  thread_local Ladder T;
  T.codes.resize((size_t)L + 8);
  for (int i = 0; i < L; ++i) T.codes[(size_t)i] = (uint8_t)code_of(read[i]);
  long best_score = -1;
  struct Res { uint32_t code = 0; int k = 0; long count = 0; } res[4];
  for (int k = 2; k <= 6; ++k) {
    int count = 0;
    long imax = -1;
    switch (k) {
      case 2: count_k<2>(T, L, count, imax); break;
      case 3: count_k<3>(T, L, count, imax); break;
      case 4: count_k<4>(T, L, count, imax); break;
      case 5: count_k<5>(T, L, count, imax); break;
      default: count_k<6>(T, L, count, imax); break;
    }
    // argmax of an untouched table is -1, i.e. all ones: decode() yields k times the last letter (utils.nim:197-198,245)
    const uint32_t code = (uint32_t)((uint64_t)imax & ((1ull << (2 * k)) - 1ull));
    char unit[8];
    for (int j = 0; j < k; ++j) unit[j] = "CATG"[(code >> (2 * (k - 1 - j))) & 3u];
    long score = (long)count * k;
    if (score <= best_score) {                                         // :250-253
      if (count < (int)((double)L * 0.12 / (double)k)) break;
      continue;
    }
    const int lit = literal_count(read, L, unit, k);                   // :254
    score = (long)lit * k;
    if (score < best_score) continue;
    best_score = score;
    for (int q = 0; q < n_p; ++q)
      if (lit > (int)((double)L * p[q] / (double)k)) { res[q].code = code; res[q].k = k; res[q].count = lit; }   // :259-263
  }
  for (int q = 0; q < n_p; ++q) out[q] = pack(res[q].code, res[q].k, res[q].count);
}

// one long read of a batch: the whole read under p, each clipped end add_soft would look at under both lowered thresholds
// (extract.nim:93-116, :207-211, :241-244).  seq4: the read's 4-bit SEQ as it sits in the BAM record.
void host_score_long_read(const uint8_t *seq4, uint32_t L, uint32_t clip_l, uint32_t clip_r, uint32_t cig, uint32_t mapq, const strl_opts &o, bool skipped,
                          uint32_t id, uint32_t &whole, strl_soft_rec soft[2], int &n_soft) {
  static const char NT16[] = "=ACMGRSVTWYHKDBN";                       // htslib seq_nt16_str (hts-nim's aln.sequence)
  n_soft = 0;
  whole = 0;
  if (skipped) return;                     // the skip predicate's word stays; a single-M cigar has no clipped end
  thread_local std::string text;
  text.resize((size_t)L);
  for (uint32_t j = 0; j < L; ++j) text[j] = NT16[(seq4[j >> 1] >> ((~j & 1u) << 2)) & 0xfu];
  const double p = o.proportion_repeat;
  host_get_repeat(text.data(), (int)L, &p, 1, &whole);
  // add_soft's gates, as the scorer kernel applies them (score.hip, score_kernel MODE 0)
  if (mapq < o.min_mapq || !(cig & (STRL_CIG_FIRST_S | STRL_CIG_LAST_S))) return;
  const bool has_unit = STRL_RES_K(whole) != 0;
  const double ps[2] = {p - 0.07, std::min(p, 0.6)};
  auto side = [&](uint32_t which, uint32_t first, uint32_t len) {
    strl_soft_rec &s = soft[n_soft++];
    s.read_side = (id << 1) | which;
    s.seg_len = len;
    uint32_t w[2];
    host_get_repeat(text.data() + first, (int)len, ps, 2, w);
    s.res_first = w[0];
    s.res_after = w[1];
  };
  if ((cig & STRL_CIG_FIRST_S) && (has_unit || clip_l > 16)) side(0, 0, std::min(clip_l, L));
  // (with a single cigar op both iterations of add_soft's loop look at op 0: the pair logic replays the duplicate)
  if ((cig & STRL_CIG_LAST_S) && !(cig & STRL_CIG_ONE_OP) && (has_unit || clip_r > 16)) { const uint32_t c = std::min(clip_r, L); side(1, L - c, c); }
}

}  // namespace strl

extern "C" int strl_score_read_host(const char *seq, int32_t l_seq, double proportion_repeat, uint32_t *word) {
  if ((!seq && l_seq) || l_seq < 0 || !word) { strl::set_error("bad argument"); return STRL_ERR_ARG; }
  strl::host_get_repeat(seq, l_seq, &proportion_repeat, 1, word);
  return STRL_OK;
}
