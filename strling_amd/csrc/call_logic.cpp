// call_logic.cpp -- host side of `strling call` above the clustering kernels: the evidence collected around one
// bound (collect.nim:36-182, spanning.nim:7-49, utils.nim:129-158) and the genotype record built from it
// (genotyper.nim:56-199, call.nim:29-48,264-281).  Sequential, hash-table shaped work on a few hundred records per
// bound: it stays on the host (the region's records come from an indexed BAM read, which is what bounds it).
// Nim stdlib order effects that reach the output text are reproduced with nim_tables.h.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>
#include "../../include/strling_amd.h"
#include "common.h"
#include "nim_tables.h"

using strl::set_error;
namespace nim { std::vector<int64_t> table_slot_order(const std::vector<uint64_t> &hcodes, uint64_t initial_size); }

namespace {

constexpr uint16_t F_UNMAP = 0x4, F_REVERSE = 0x10, F_SECONDARY = 0x100, F_DUP = 0x400, F_SUPPL = 0x800;

struct Rec {   // the hts-nim accessors collect.nim uses
  const strl_records *r;
  int ncig(int64_t i) const { return (int)(r->cigar_off[i + 1] - r->cigar_off[i]); }
  int op(int64_t i, int j) const { return (int)(r->cigar[r->cigar_off[i] + j] & 0xf); }
  int len(int64_t i, int j) const { return (int)(r->cigar[r->cigar_off[i] + j] >> 4); }
  static bool cons_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
  static bool cons_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
  int64_t start(int64_t i) const { return r->pos[i]; }
  int64_t stop(int64_t i) const {   // bam_endpos
    int64_t rl = 0;
    if (!(r->flag[i] & F_UNMAP))
      for (int j = 0, n = ncig(i); j < n; ++j) if (cons_ref(op(i, j))) rl += len(i, j);
    return r->pos[i] + (rl ? rl : 1);
  }
  std::string_view qname(int64_t i) const { return std::string_view(r->qnames + r->qname_off[i], (size_t)(r->qname_off[i + 1] - r->qname_off[i])); }
  char base(int64_t i, int64_t j) const { return "=ACMGRSVTWYHKDBN"[(r->seq4[r->seq_off[i] + (uint64_t)(j >> 1)] >> ((~j & 1) << 2)) & 0xf]; }
};

// spanning.nim:7-20
void cumulative(const uint32_t frag[4096], float cd[4096]) {
  for (int i = 0; i < 4096; ++i) {
    float s = 0;
    for (int j = std::max(0, i - 11); j <= std::min(i + 11, 4095); ++j) s += (float)frag[j];
    cd[i] = s;
  }
  for (int i = 1; i < 4096; ++i) cd[i] = cd[i] + cd[i - 1];
  const float fmax = cd[4095];
  for (int i = 0; i < 4096; ++i) cd[i] = cd[i] / fmax;
}
// spanning.nim:22-49
double expected_spanning_probability(const float cd[4096], int64_t start, int64_t stop, bool reverse, int64_t ev_start, int64_t ev_stop) {
  const int64_t msb = 20;
  int64_t dist;
  if (start < ev_stop - msb) {
    if (reverse) return 0;
    dist = ev_start - start;
  } else {
    if (!reverse) return 0;
    dist = stop - ev_stop;
  }
  if (dist < 0) return 0;
  if (dist + (ev_stop - ev_start) < msb) return 0;
  dist += msb + (ev_stop - ev_start);
  if (dist < 0 || dist > 4095) return 0;
  return (double)(1.0f - cd[dist]);
}
// What spanners() derives from the fragment-length histogram alone -- the smoothed cumulative distribution (spanning.nim:7-20:
// 4096 x 23 additions) and the prefix sums percentile() walks (utils.nim:129-137) -- kept per thread for the histogram it was made
// from: `strling call` asks for thousands of bounds with ONE histogram, and making them per bound (and the sum per spanning
// fragment) was a fifth of the function.  The values are the ones the functions below compute.
struct FragTables {
  uint32_t frag[4096];
  float cd[4096];
  uint64_t prefix[4096];       // prefix[i] = frag[0] + .. + frag[i]
  bool have = false;
};
void cumulative(const uint32_t frag[4096], float cd[4096]);
const FragTables &frag_tables(const uint32_t frag[4096]) {
  thread_local FragTables T;
  if (!T.have || memcmp(T.frag, frag, sizeof T.frag) != 0) {
    memcpy(T.frag, frag, sizeof T.frag);
    cumulative(frag, T.cd);
    uint64_t run = 0;
    for (int i = 0; i < 4096; ++i) { run += frag[i]; T.prefix[i] = run; }
    T.have = true;
  }
  return T;
}
// utils.nim:129-137 (total and partial sum wrap in uint32 / accumulate in int64 exactly like the loop this replaces)
double percentile_cached(const FragTables &T, int64_t fragment_length) {
  const uint32_t total = (uint32_t)T.prefix[4095];
  const int64_t upto = std::min<int64_t>(std::max<int64_t>(fragment_length, 0), 4095);
  const int64_t s = (int64_t)T.prefix[(size_t)upto];
  return (double)s / (double)std::max<uint32_t>(1u, total);
}
// utils.nim:129-137
double percentile(const uint32_t frag[4096], int64_t fragment_length) {
  uint32_t total = 0;
  for (int i = 0; i < 4096; ++i) total += frag[i];
  int64_t s = 0;
  for (int i = 0; i < 4096; ++i) { s += frag[i]; if (i >= fragment_length) break; }
  return (double)s / (double)std::max<uint32_t>(1u, total);
}
// utils.nim:148-158
int median_depth(const std::vector<int64_t> &D) {
  int64_t H[1048] = {0};
  for (int64_t d : D) H[std::min<int64_t>(d, 1047)] += 1;
  int64_t s = 0;
  for (int i = 0; i < 1048; ++i) { s += H[i]; if ((double)s > (double)D.size() / 2.0) return i; }
  return 0;
}
// collect.nim:50-72
int64_t find_read_position(const Rec &R, int64_t i, int64_t position) {
  int64_t r_off = R.start(i), q_off = 0;
  for (int j = 0, n = R.ncig(i); j < n; ++j) {
    if (r_off > position) return -1;
    const int op = R.op(i, j), len = R.len(i, j);
    if (Rec::cons_query(op)) q_off += len;
    if (Rec::cons_ref(op)) r_off += len;
    if (r_off < position) continue;
    const int64_t over = r_off - position;
    if (over > q_off) return -1;
    if (!Rec::cons_query(op)) return -1;
    return q_off - over;
  }
  return -1;
}
// collect.nim:75-93: greedy literal count of the unit inside the part of the read that lies over the bound
int count_in_bounds(const Rec &R, int64_t i, const strl_bounds &b) {
  if (b.right < b.left) return 0;
  const int64_t dlen = R.r->l_seq[i];
  int64_t rl = find_read_position(R, i, (int64_t)b.left), rr = find_read_position(R, i, (int64_t)b.right);
  if (rl >= 0 && rr < 0) rr = dlen;
  if (rl < 0 && rr < 0) return 0;
  if (rl < 0) rl = 0;
  const int64_t slen = std::max<int64_t>(0, rr - rl);
  const int k = (int)strnlen(b.repeat, 6);
  int result = 0;
  for (int64_t p = rl; p + k <= rl + slen;) {
    bool eq = true;
    for (int j = 0; j < k && eq; ++j) eq = R.base(i, p + j) == b.repeat[j];
    if (eq) { ++result; p += k; } else ++p;
  }
  if (result < (int)((double)slen * 0.7 / (double)k)) result = 0;
  return result;
}
int64_t bound_slop(const strl_bounds &b) {
  const int64_t width = (int64_t)b.right - (int64_t)b.left;
  int64_t slop = (int64_t)strnlen(b.repeat, 6) - 1;
  if (width < 5) slop += 5 - width;
  return slop;
}
// collect.nim:97-119 (+ Record/Bounds overlap, cluster.nim:104-108)
bool overlapping_read(const Rec &R, int64_t i, const strl_bounds &b, strl_support &s) {
  if (R.r->tid[i] != b.tid) return false;
  const int64_t start = R.start(i), stop = R.stop(i), slop = bound_slop(b);
  if (!(std::max<int64_t>(start, b.left) <= std::min<int64_t>(stop, b.right))) return false;
  s.type = STRL_OVERLAPPING_READ;
  s.repeat_count = (uint8_t)count_in_bounds(R, i, b);
  s.rec = i;
  if (start < (int64_t)b.left - slop && stop > (int64_t)b.right + slop) {
    s.type = STRL_SPANNING_READ;
    for (int j = 0, n = R.ncig(i); j < n; ++j) {
      if (R.op(i, j) == 1) s.cigar_ins = (uint8_t)(s.cigar_ins + (uint8_t)R.len(i, j));
      if (R.op(i, j) == 2) s.cigar_del = (uint8_t)(s.cigar_del + (uint8_t)R.len(i, j));
    }
  }
  return true;
}

// Nim CountTable over small integers, keys in inc() order -> (key, count) in slot order (tables.nim, 64 slots to start:
// `var t: CountTable[T]` is initialised with defaultInitialSize = 32 on the first inc)
std::vector<std::pair<int64_t, int64_t>> count_table(const std::vector<int64_t> &keys) {
  std::vector<int64_t> dk, dv;
  std::vector<uint64_t> hc;
  for (int64_t k : keys) {
    size_t j = 0;
    while (j < dk.size() && dk[j] != k) ++j;
    if (j < dk.size()) { ++dv[j]; continue; }
    dk.push_back(k); dv.push_back(1); hc.push_back(nim::hash_int((uint64_t)k));
  }
  std::vector<std::pair<int64_t, int64_t>> out;
  for (int64_t id : nim::table_slot_order(hc, 32)) out.push_back({dk[(size_t)id], dv[(size_t)id]});
  return out;
}
// utils.nim:165-177 most_frequent(t, 2) (CountTable.sort = stable sort of the slots, descending) / `largest`
void top_two(const std::vector<int64_t> &keys, double &a1, double &a2) {
  a1 = NAN; a2 = NAN;
  auto t = count_table(keys);
  if (t.size() >= 2) {
    std::stable_sort(t.begin(), t.end(), [](const auto &x, const auto &y) { return x.second > y.second; });
    a1 = (double)t[0].first; a2 = (double)t[1].first;
  } else if (t.size() == 1) a1 = (double)t[0].first;
}
// genotyper.nim:122-130
double anchored_lm(uint64_t sum_str_counts, double depth) {
  if (sum_str_counts == 0) return NAN;
  const double intercept = 4.3558142, cofficient = 0.7565329;
  const double y = log2((double)sum_str_counts / std::max(1.0, depth) + 1) * cofficient + intercept;
  return pow(2, y);
}
std::string nim_float(double v) {   // Nim `$float` for the values that occur (integral depths)
  if (std::isnan(v)) return "nan";
  char buf[64];
  snprintf(buf, sizeof buf, "%.16g", v);
  std::string s(buf);
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  return s;
}

}  // namespace

extern "C" {

// collect.nim:132-182.  `r` may hold more than the region query would return (e.g. everything read from the linear
// index offset on): htslib's iterator filter -- tid equal, pos < end, bam_endpos > beg -- is applied here.
int strl_spanners(const strl_records *r, const int32_t *isize, const strl_bounds *b, int32_t window, const uint32_t frag[4096],
                  uint8_t min_mapq, strl_support *out, uint64_t cap, strl_span_summary *sum) {
  if (!r || !b || !frag || !sum || (r->n && !isize)) { set_error("null argument"); return STRL_ERR_ARG; }
  if (b->left > b->right) { set_error("bound with left > right"); return STRL_ERR_ARG; }
  const Rec R{r};
  const int max_size = 5000;
  const int64_t window_left = (int64_t)b->left - window, window_right = (int64_t)b->right + window;
  const int64_t beg = std::max<int64_t>(0, window_left), end = window_right;
  const FragTables &FT = frag_tables(frag);
  const float *cd = FT.cd;
  std::vector<int64_t> depths((size_t)(window_right - window_left), 0);
  // One open-addressed index of the region's qnames for both of the reference's tables (the expected-spanner values and the
  // pairs), keyed by the Nim hash each table's slot order needs anyway: two std::unordered_map<string_view> look-ups and two
  // murmur hashes per record were a third of this function's 200 ns a record (3.3 CPU-seconds of `strling call` on a genome).
  struct Name { std::string_view qn; uint64_t hc; int32_t exp_id, pair_id; };
  std::vector<Name> names;
  std::vector<int32_t> slot(1024, -1);
  auto name_of = [&](std::string_view qn) -> Name & {
    const uint64_t hc = nim::hash_bytes(reinterpret_cast<const uint8_t *>(qn.data()), (int)qn.size());
    if (2 * (names.size() + 1) > slot.size()) {
      std::vector<int32_t> bigger(slot.size() * 4, -1);
      for (size_t k = 0; k < names.size(); ++k) {
        size_t at = (size_t)(names[k].hc * 0x9E3779B97F4A7C15ull >> 20) & (bigger.size() - 1);
        while (bigger[at] >= 0) at = (at + 1) & (bigger.size() - 1);
        bigger[at] = (int32_t)k;
      }
      slot.swap(bigger);
    }
    size_t at = (size_t)(hc * 0x9E3779B97F4A7C15ull >> 20) & (slot.size() - 1);
    for (; slot[at] >= 0; at = (at + 1) & (slot.size() - 1)) {
      Name &nm = names[(size_t)slot[at]];
      if (nm.hc == hc && nm.qn == qn) return nm;
    }
    slot[at] = (int32_t)names.size();
    names.push_back(Name{qn, hc, -1, -1});
    return names.back();
  };
  std::vector<double> exp_val;
  std::vector<uint64_t> exp_hc, pair_hc;
  struct Pair { int64_t first, second; int n; };
  std::vector<Pair> pairs;
  uint64_t n_out = 0;
  auto emit = [&](const strl_support &s) { if (out && n_out < cap) out[n_out] = s; ++n_out; };
  sum->median_depth = 0; sum->expected_spanners = 0; sum->n_support = 0;
  for (int64_t i = 0; i < r->n; ++i) {
    if (r->tid[i] != b->tid) continue;
    const int64_t start = R.start(i), stop = R.stop(i);
    if (!(start < end && stop > beg)) continue;
    const uint16_t f = r->flag[i];
    if (f & (F_SECONDARY | F_SUPPL | F_DUP)) continue;                               // :142
    if (r->mapq[i] < min_mapq) continue;                                            // :143
    const std::string_view qn = R.qname(i);
    const double prob = expected_spanning_probability(cd, start, stop, (f & F_REVERSE) != 0, b->left, b->right);
    Name *nm = nullptr;
    if (prob > 0) {                                                                  // :145-152
      nm = &name_of(qn);
      if (nm->exp_id >= 0) exp_val[(size_t)nm->exp_id] = 0.5 * (exp_val[(size_t)nm->exp_id] + prob);
      else {
        nm->exp_id = (int32_t)exp_val.size();
        exp_val.push_back(prob);
        exp_hc.push_back(nm->hc);
      }
    }
    depths[(size_t)std::max<int64_t>(0, start - window_left - 1)] += 1;              // :154-155
    depths[(size_t)std::min<int64_t>((int64_t)depths.size() - 1, stop - window_left - 1)] -= 1;
    strl_support s{};
    if (overlapping_read(R, i, *b, s)) emit(s);                                      // :157-159
    if (r->tid[i] != r->mtid[i]) continue;
    if (std::abs((int64_t)isize[i]) > max_size) continue;
    if (!nm) nm = &name_of(qn);
    if (nm->pair_id >= 0) { Pair &p = pairs[(size_t)nm->pair_id]; if (p.n == 1) p.second = i; ++p.n; }
    else {
      nm->pair_id = (int32_t)pairs.size();
      pairs.push_back(Pair{i, -1, 1});
      pair_hc.push_back(nm->hc);
    }
    if (pairs.size() > 20000) { sum->median_depth = -1; sum->expected_spanners = 0; sum->n_support = 0; return STRL_OK; }   // :171-174
  }
  float es = 0;                                                                      // :176-177: values() in slot order, float32 sum
  for (int64_t id : nim::table_slot_order(exp_hc, 32)) es += (float)exp_val[(size_t)id];
  sum->expected_spanners = es;
  const int64_t slop = bound_slop(*b);
  for (int64_t id : nim::table_slot_order(pair_hc, 32)) {                            // :179-183, spanning_fragment :36-48
    const Pair &p = pairs[(size_t)id];
    if (p.n != 2) continue;
    if (!(R.start(p.first) <= R.start(p.second))) { set_error("doAssert L.start <= R.start (collect.nim:37)"); return STRL_ERR_ASSERT; }
    if (R.start(p.first) < (int64_t)b->left - slop && R.stop(p.second) > (int64_t)b->right + slop) {
      strl_support s{};
      s.type = STRL_SPANNING_FRAGMENT;
      s.fragment_length = std::max<uint32_t>(1u, (uint32_t)std::abs((int64_t)isize[p.first]));
      s.fragment_percentile = percentile_cached(FT, (int64_t)s.fragment_length);
      s.rec = p.first;
      emit(s);
    }
  }
  for (size_t i = 1; i < depths.size(); ++i) depths[i] += depths[i - 1];
  sum->median_depth = median_depth(depths);
  sum->n_support = n_out;
  if (out && n_out > cap) { set_error("support capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)n_out); return STRL_ERR_CAPACITY; }
  return STRL_OK;
}

// genotyper.nim:150-199.  members = the cluster's treads; their qname strings are qnames[qname_off[t.qname_id] ..).
int strl_genotype(const strl_bounds *b, const strl_tread *members, uint64_t n_members, const uint64_t *qname_off, const char *qnames,
                  const strl_support *spanners, uint64_t n_spanners, const strl_call_opts *o, double depth, strl_call *c) {
  if (!b || !c || !o || (n_members && (!members || !qname_off || !qnames)) || (n_spanners && !spanners)) { set_error("null argument"); return STRL_ERR_ARG; }
  memset(c, 0, sizeof *c);
  c->tid = b->tid; c->start = b->left; c->stop = b->right; c->left_clips = b->n_left; c->right_clips = b->n_right;
  memcpy(c->repeat, b->repeat, 7);
  c->depth = depth;
  const int ru = std::max(1, (int)strnlen(c->repeat, 6));
  if (n_spanners == 0) c->allele1 = NAN;
  else {
    std::vector<int64_t> rc, indel;                                                 // spanning_read_est, :61-98
    uint32_t frags = 0;
    for (uint64_t i = 0; i < n_spanners; ++i) {
      if (spanners[i].type == STRL_SPANNING_READ) {
        rc.push_back((uint16_t)spanners[i].repeat_count);
        indel.push_back((int16_t)((int16_t)spanners[i].cigar_ins - (int16_t)spanners[i].cigar_del));
      }
      frags += spanners[i].type == STRL_SPANNING_FRAGMENT;
    }
    double a1bp, a2bp, a1ru, a2ru;
    top_two(rc, a1ru, a2ru);
    top_two(indel, a1bp, a2bp);
    if (!std::isnan(a1bp)) c->allele1 = a1bp / (double)ru;
    c->spanning_reads = (uint32_t)rc.size();
    c->spanning_pairs = frags;
  }
  // :175 -- evaluated before allele2 is assigned, so the last term compares 0.0 with the median fragment length
  c->is_large = b->n_left >= o->min_clip && b->n_right >= o->min_clip && (uint16_t)(b->n_left + b->n_right) >= o->min_clip_total &&
                (int64_t)n_members >= (int64_t)o->min_support && c->allele2 > (double)o->median_fragment_length;
  uint64_t sum = 0;
  for (uint64_t i = 0; i < n_members; ++i) sum += members[i].repeat_count;
  c->overlapping_reads = (uint32_t)n_members;
  c->sum_str_counts = (uint32_t)sum;
  c->allele2 = anchored_lm(sum, depth) / (double)ru;
  std::unordered_map<std::string_view, int> seen;                                    // :187-191 toHashSet(qnames).len
  for (uint64_t i = 0; i < n_members; ++i) {
    if (members[i].split != STRL_SOFT_NONE) continue;
    const uint64_t q = (uint64_t)members[i].qname_id;
    seen.emplace(std::string_view(qnames + qname_off[q], (size_t)(qname_off[q + 1] - qname_off[q])), 1);
  }
  c->anchored_reads = (uint32_t)seen.size();
  return STRL_OK;
}

// add_percentile (call.nim:29-48), the never-taken refinement of :264-276 (is_large is always false, see above) and the
// row order of -genotype.txt: Table[string, seq[Call]] keyed by the canonical unit, slot order, insertion order inside.
int strl_calls_finish(strl_call *calls, uint64_t n, const strl_unplaced *unplaced, uint64_t n_unplaced, uint64_t *order) {
  if ((n && (!calls || !order)) || (n_unplaced && !unplaced)) { set_error("null argument"); return STRL_ERR_ARG; }
  std::vector<float> oes((size_t)n);
  for (uint64_t i = 0; i < n; ++i) {
    const float obs = (float)calls[i].spanning_pairs, ex = calls[i].expected_spanning_fragments;
    oes[(size_t)i] = (1.0f + obs - ex) / (ex + 1.0f);
  }
  std::vector<float> sorted = oes;
  std::sort(sorted.begin(), sorted.end());
  for (uint64_t i = 0; i < n; ++i) {
    volatile float num = (float)(std::lower_bound(sorted.begin(), sorted.end(), oes[(size_t)i]) - sorted.begin());
    volatile float den = (float)((int64_t)n - 1);
    calls[i].spanning_fragments_oe_percentile = num / den;                           // 0/0 (one call) is a run-time NaN like the reference's
  }
  std::vector<std::string> keys;
  std::vector<uint64_t> hc;
  std::vector<std::vector<uint64_t>> groups;
  std::unordered_map<std::string, size_t> ix;
  for (uint64_t i = 0; i < n; ++i) {
    char in6[6] = {0}, out6[6];
    memcpy(in6, calls[i].repeat, strnlen(calls[i].repeat, 6));
    strl_canonical_repeat(in6, out6);
    std::string key(out6, strnlen(out6, 6));
    auto it = ix.find(key);
    if (it == ix.end()) {
      it = ix.emplace(key, keys.size()).first;
      keys.push_back(key);
      hc.push_back(nim::hash_bytes(reinterpret_cast<const uint8_t *>(key.data()), (int)key.size()));
      groups.emplace_back();
    }
    groups[it->second].push_back(i);
  }
  uint64_t k = 0;
  for (int64_t g : nim::table_slot_order(hc, 32)) {
    // call.nim:267-276: exactly one is_large call of a unit would take the unplaced count of that unit
    uint64_t n_large = 0, which = 0;
    for (uint64_t i : groups[(size_t)g]) if (calls[i].is_large) { if (n_large == 0) which = i; if (++n_large > 1) break; }
    if (n_large == 1) {
      int64_t cnt = 0;
      for (uint64_t u = 0; u < n_unplaced; ++u) if (keys[(size_t)g] == std::string(unplaced[u].repeat, strnlen(unplaced[u].repeat, 6))) cnt = unplaced[u].count;
      strl_call &c = calls[which];                                                   // update_genotype, genotyper.nim:201-205
      c.unplaced_reads = (int32_t)cnt;
      if (cnt > 2) {
        const double y = log2((double)cnt / c.depth + 1) * 0.7595562 + 8.9199168;
        c.allele2 = pow(2, y) / (double)strnlen(c.repeat, 6);
      }
    }
    for (uint64_t i : groups[(size_t)g]) order[k++] = i;
  }
  return STRL_OK;
}

// -unplaced.txt order (call.nim:280-281): CountTable[string] slots; `unplaced` in assignment order (what strl_cluster returns)
int strl_unplaced_order(const strl_unplaced *unplaced, uint64_t n, uint64_t *order) {
  if (n && (!unplaced || !order)) { set_error("null argument"); return STRL_ERR_ARG; }
  std::vector<uint64_t> hc;
  for (uint64_t i = 0; i < n; ++i) hc.push_back(nim::hash_bytes(reinterpret_cast<const uint8_t *>(unplaced[i].repeat), (int)strnlen(unplaced[i].repeat, 6)));
  uint64_t k = 0;
  for (int64_t id : nim::table_slot_order(hc, 32)) order[k++] = (uint64_t)id;
  return STRL_OK;
}

// assign_reads_locus (callclusters.nim:14-50) for every locus in order
int strl_assign_reads_loci(strl_tread *treads, uint64_t n, int mode, strl_locus *loci, uint64_t n_loci, uint64_t *assigned_off,
                           uint32_t *assigned, uint64_t cap) {
  if ((n && !treads) || (n_loci && (!loci || !assigned_off))) { set_error("null argument"); return STRL_ERR_ARG; }
  struct Key { int32_t tid; char rep[6]; bool operator==(const Key &o) const { return tid == o.tid && memcmp(rep, o.rep, 6) == 0; } };
  struct KeyHash { size_t operator()(const Key &k) const { return (size_t)nim::hash_tid_rep(k.tid, k.rep); } };
  std::unordered_map<Key, std::vector<uint32_t>, KeyHash> groups;                 // Table[tid_rep, seq[tread]], each sorted by position
  for (uint64_t i = 0; i < n; ++i) {
    if (treads[i].split == STRL_SOFT_TAKEN) continue;
    if (mode == STRL_MODE_MERGE && treads[i].tid < 0) continue;
    Key k{treads[i].tid, {0}};
    memcpy(k.rep, treads[i].repeat, 6);
    groups[k].push_back((uint32_t)i);
  }
  for (auto &g : groups)
    std::stable_sort(g.second.begin(), g.second.end(), [&](uint32_t a, uint32_t b) { return treads[a].position < treads[b].position; });
  uint64_t tot = 0;
  for (uint64_t j = 0; j < n_loci; ++j) {
    strl_bounds &L = loci[j].b;
    assigned_off[j] = tot;
    Key k{L.tid, {0}};
    memcpy(k.rep, L.repeat, strnlen(L.repeat, 6));
    L.n_total = 0; L.n_left = 0; L.n_right = 0;
    auto it = groups.find(k);
    if (it == groups.end() || it->second.empty()) continue;
    std::vector<uint32_t> &trs = it->second;
    const uint32_t left_most = L.left_most == 0 ? 0u : L.left_most - 1u;
    size_t li = 0, ri = 0;
    while (li < trs.size() && treads[trs[li]].position < left_most) ++li;          // lowerBound
    while (ri < trs.size() && treads[trs[ri]].position <= L.right_most) ++ri;      // upperBound
    if (ri < li) ri = li;
    for (size_t q = li; q < ri; ++q) {
      const strl_tread &t = treads[trs[q]];
      ++L.n_total;
      if (t.split == STRL_SOFT_RIGHT) ++L.n_right;
      else if (t.split == STRL_SOFT_LEFT) ++L.n_left;
      if (assigned && tot < cap) assigned[tot] = trs[q];
      ++tot;
    }
    // table[key] = trs[0..<li] & (if ri < trs.high: trs[ri+1..high]) -- trs[ri] is dropped with the assigned ones
    const size_t drop_end = std::min(trs.size(), ri + 1);
    if (ri > li || ri < trs.size()) {
      for (size_t q = li; q < drop_end; ++q) treads[trs[q]].split = STRL_SOFT_TAKEN;
      trs.erase(trs.begin() + (long)li, trs.begin() + (long)drop_end);
    }
  }
  if (n_loci) assigned_off[n_loci] = tot;
  if (assigned && tot > cap) { set_error("assigned capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)tot); return STRL_ERR_CAPACITY; }
  return STRL_OK;
}

// Table[(tid, repeat), seq[tread]] iteration order: keys in first-appearance order -> slot order (8192 initial slots)
int strl_group_order(const strl_tread *treads, uint64_t n, int mode, strl_group_key *out, uint64_t cap, uint64_t *n_groups) {
  if ((n && !treads) || !n_groups) { set_error("null argument"); return STRL_ERR_ARG; }
  struct Key { int32_t tid; char rep[6]; bool operator==(const Key &o) const { return tid == o.tid && memcmp(rep, o.rep, 6) == 0; } };
  struct KeyHash { size_t operator()(const Key &k) const { return (size_t)nim::hash_tid_rep(k.tid, k.rep); } };
  std::unordered_map<Key, size_t, KeyHash> seen;
  std::vector<Key> keys;
  std::vector<uint64_t> hc;
  for (uint64_t i = 0; i < n; ++i) {
    if (mode == STRL_MODE_MERGE && treads[i].tid < 0) continue;
    Key k{treads[i].tid, {0}};
    memcpy(k.rep, treads[i].repeat, 6);
    if (seen.emplace(k, keys.size()).second) { keys.push_back(k); hc.push_back(nim::hash_tid_rep(k.tid, k.rep)); }
  }
  *n_groups = keys.size();
  if (!out) return STRL_OK;
  if (keys.size() > cap) { set_error("group capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)keys.size()); return STRL_ERR_CAPACITY; }
  uint64_t q = 0;
  for (int64_t id : nim::table_slot_order(hc, 8192)) {
    out[q].tid = keys[(size_t)id].tid;
    memset(out[q].repeat, 0, sizeof out[q].repeat);
    memcpy(out[q].repeat, keys[(size_t)id].rep, 6);
    ++q;
  }
  return STRL_OK;
}

// genotyper.nim:54-57
int strl_call_row(char *buf, int cap, const strl_call *c, const char *chrom) {
  const std::string d = nim_float(c->depth);
  return snprintf(buf, (size_t)cap, "%s\t%u\t%u\t%s\t%.2f\t%.2f\t%u\t%u\t%u\t%.2f\t%.2f\t%u\t%u\t%d\t%s\t%u", chrom, c->start, c->stop, c->repeat,
                  c->allele1, c->allele2, c->anchored_reads, c->spanning_reads, c->spanning_pairs, (double)c->expected_spanning_fragments,
                  (double)c->spanning_fragments_oe_percentile, c->left_clips, c->right_clips, c->unplaced_reads, d.c_str(), c->sum_str_counts);
}

}  // extern "C"
