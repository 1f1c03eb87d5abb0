// host_logic.cpp -- host side of the C ABI that needs no device: SoA derivation from BAM-native
// records, the mate-pairing state machine of extract (Cache.add), fragment statistics, the .bin
// reader/writer and the -bounds.txt row formatter.  Everything here consumes the packed results the
// HIP kernels produced; none of it scores reads.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <atomic>
#include <vector>
#include "common.h"

using strl::set_error;

namespace {

constexpr uint16_t F_PROPER = 0x2, F_UNMAP = 0x4, F_REVERSE = 0x10, F_MREVERSE = 0x20, F_SECONDARY = 0x100, F_SUPPL = 0x800;
constexpr int OP_M = 0, OP_S = 4;

// kmer module alphabet (brentp/nim-kmer): 2-bit code -> base.  See DESIGN.md "kmer code order".
inline char code_base(uint32_t c) { return "CATG"[c & 3u]; }
inline uint32_t base_code(char b) {
  switch (b) { case 'C': return 0; case 'A': return 1; case 'T': return 2; case 'G': return 3; default: return 1; }
}

struct Unit {
  char s[6];
  int len;
};

inline Unit unpack_unit(uint32_t w) {
  Unit u{};
  u.len = (int)STRL_RES_K(w);
  const uint32_t code = STRL_RES_CODE(w);
  for (int j = 0; j < u.len; ++j) u.s[j] = code_base(code >> (2 * (u.len - 1 - j)));
  return u;
}

inline int unit_len(const char r[6]) {
  int l = 0;
  while (l < 6 && r[l]) ++l;
  return l;
}

// p_repeat template, extract.nim:56-58 (uint8 product)
inline double p_repeat(const strl_tread &t) {
  const uint8_t prod = (uint8_t)(t.repeat_count * (uint8_t)unit_len(t.repeat));
  const uint8_t al = t.align_length ? t.align_length : 1;
  return (double)prod / (double)al;
}

// minimum rotation of the reverse complement (utils.nim:61-80), on 2-bit codes:
// complement of a "CATG" code c is 3 - c.
void min_rev_complement(char rep[6]) {
  const int l = unit_len(rep);
  if (l == 0) return;
  uint32_t c = 0;
  for (int i = l - 1; i >= 0; --i) c = (c << 2) | (3u - base_code(rep[i]));  // only ACGT reach here (decoded units)
  const uint32_t mask = (1u << (2 * l)) - 1u;
  uint32_t best = c, f = c;
  for (int j = 0; j < l; ++j) {
    f = ((f << 2) | (f >> (2 * (l - 1)))) & mask;
    best = std::min(best, f);
  }
  for (int j = 0; j < l; ++j) rep[j] = code_base(best >> (2 * (l - 1 - j)));
}

// canonical_repeat, utils.nim:291-310: the rev-comp rotation minimum if it is ASCII-smaller
void canonical_repeat(char rep[6]) {
  char r[6];
  memcpy(r, rep, 6);
  min_rev_complement(r);
  if (memcmp(r, rep, 6) < 0) memcpy(rep, r, 6);
}

inline bool should_reverse(uint16_t f) {  // extract.nim:134-139
  bool r = !(f & F_MREVERSE);
  return (f & F_REVERSE) ? !r : r;
}

// adjust_by, extract.nim:141-179
bool adjust_by(strl_tread &A, const strl_tread &B, const strl_opts &o, uint32_t B_position) {
  if (A.repeat_count == 0) return false;
  const uint32_t half = (uint32_t)((double)A.align_length / 2.0 + 0.5);
  if (B.mapping_quality > o.min_mapq &&
      ((p_repeat(A) > o.proportion_repeat && p_repeat(B) < 0.2) || (!(A.flag & F_PROPER) && A.mapping_quality < o.min_mapq))) {
    if (B.flag & F_REVERSE) {
      A.position = B_position - (uint32_t)o.median_fragment_length + B.align_length + half;
      if (B.split == STRL_SOFT_NONE_LEFT) A.position = B_position;
    } else {
      A.position = B_position + (uint32_t)o.median_fragment_length - half;
      if (B.split == STRL_SOFT_NONE_RIGHT) A.position = B_position + (uint32_t)B.align_length;
    }
    A.split = STRL_SOFT_NONE;
    A.tid = B.tid;
    A.mapping_quality = std::max(A.mapping_quality, B.mapping_quality);
    if (should_reverse(A.flag)) min_rev_complement(A.repeat);
  } else if (A.mapping_quality >= o.min_mapq || (A.flag & F_PROPER)) {
    A.position += half;
    A.mapping_quality = std::max(A.mapping_quality, B.mapping_quality);
  }
  return true;
}

bool unplaced_pair(const strl_tread &A, const strl_tread &B, const strl_opts &o) {  // extract.nim:182-190
  if (p_repeat(A) > o.proportion_repeat && p_repeat(B) > o.proportion_repeat) return true;
  if (p_repeat(A) > o.proportion_repeat && B.mapping_quality < o.min_mapq) return true;
  if (p_repeat(B) > o.proportion_repeat && A.mapping_quality < o.min_mapq) return true;
  return false;
}

struct RecView {
  const strl_records *r;
  int ncig(int64_t i) const { return (int)(r->cigar_off[i + 1] - r->cigar_off[i]); }
  int op(int64_t i, int j) const { return (int)(r->cigar[r->cigar_off[i] + j] & 0xf); }
  int len(int64_t i, int j) const { return (int)(r->cigar[r->cigar_off[i] + j] >> 4); }
  int64_t stop(int64_t i) const {  // htslib bam_endpos
    int64_t rl = 0;
    if (!(r->flag[i] & F_UNMAP)) {
      const int n = ncig(i);
      for (int j = 0; j < n; ++j) {
        const int o = op(i, j);
        if (o == 0 || o == 2 || o == 3 || o == 7 || o == 8) rl += len(i, j);
      }
    }
    return (int64_t)r->pos[i] + (rl ? rl : 1);
  }
  std::string_view qname(int64_t i) const {
    return std::string_view(r->qnames + r->qname_off[i], (size_t)(r->qname_off[i + 1] - r->qname_off[i]));
  }
};

// Where first-seen treads wait for their mate and where emitted treads go.
//  BatchStore : one batch is the whole input; keys are views into the batch's qname buffer, emitted treads keep
//               qname_id = record index.
//  StreamStore: batches arrive one after another (the CLI); keys and emitted qnames are owned copies, emitted
//               treads get qname_id = index into the store's own qname arena.
struct BatchStore {
  std::unordered_map<std::string_view, strl_tread> tbl;
  strl_tread *out = nullptr;
  uint64_t cap = 0, n_out = 0;
  strl_tread *find(std::string_view q) { auto it = tbl.find(q); return it == tbl.end() ? nullptr : &it->second; }
  void erase(std::string_view q) { tbl.erase(q); }
  void insert(std::string_view q, const strl_tread &t) { tbl.emplace(q, t); }
  void emit(strl_tread t, std::string_view) { if (n_out < cap) out[n_out] = t; ++n_out; }
};
struct StreamStore {
  std::unordered_map<std::string, strl_tread> tbl;
  std::vector<strl_tread> out;
  std::vector<uint64_t> qoff{0};
  std::string qnames;
  std::string key;
  strl_tread *find(std::string_view q) { key.assign(q); auto it = tbl.find(key); return it == tbl.end() ? nullptr : &it->second; }
  void erase(std::string_view q) { key.assign(q); tbl.erase(key); }
  void insert(std::string_view q, const strl_tread &t) { tbl.emplace(std::string(q), t); }
  void emit(strl_tread t, std::string_view q) {
    t.qname_id = (int64_t)out.size();
    out.push_back(t);
    qnames.append(q);
    qoff.push_back(qnames.size());
  }
};

template <class Store> struct Pairer {
  RecView rv;
  const strl_opts *o;
  const uint32_t *whole;
  const strl_soft_rec *soft;
  uint64_t n_soft;
  Store &S;
  int err = 0;

  const strl_soft_rec *find_soft(int64_t i, int side) const {
    const uint32_t key = ((uint32_t)i << 1) | (uint32_t)side;
    const strl_soft_rec *e = soft + n_soft;
    const strl_soft_rec *it = std::lower_bound(soft, e, key, [](const strl_soft_rec &a, uint32_t k) { return a.read_side < k; });
    return (it != e && it->read_side == key) ? it : nullptr;
  }
  // to_tread, extract.nim:63-87, from the packed scorer word
  strl_tread to_tread(int64_t i) {
    const strl_records *r = rv.r;
    const uint32_t w = whole[i];
    strl_tread t{};
    const Unit u = unpack_unit(w);
    memcpy(t.repeat, u.s, 6);
    const uint32_t cnt = STRL_RES_COUNT(w);
    if (cnt >= 256) { err = STRL_ERR_ASSERT; set_error("repeat_count %u >= 256 for record %lld (doAssert extract.nim:72)", cnt, (long long)i); }
    const int align_length = (w & STRL_RES_SKIPPED) ? rv.len(i, 0) : r->l_seq[i];  // extract.nim:33 / :38
    t.tid = r->tid[i];
    t.position = (uint32_t)std::max(0, r->pos[i]);
    t.flag = r->flag[i];
    t.repeat_count = (uint8_t)cnt;
    t.align_length = (uint8_t)align_length;
    t.split = STRL_SOFT_NONE;
    t.mapping_quality = r->mapq[i];
    t.qname_id = i;
    const int L = rv.ncig(i);
    if (L > 1 && rv.op(i, 0) == OP_S && rv.len(i, 0) > 16) t.split = STRL_SOFT_NONE_LEFT;
    if (L > 1 && rv.op(i, L - 1) == OP_S && rv.len(i, L - 1) > 16) t.split = STRL_SOFT_NONE_RIGHT;
    return t;
  }
  // add_soft, extract.nim:93-132, consuming the device's soft-clip records
  void add_soft(int64_t i, bool first_seen, const char read_repeat[6], std::string_view qn) {
    const strl_records *r = rv.r;
    if (r->mapq[i] < o->min_mapq) return;
    const int L = rv.ncig(i);
    if (L == 0 || (rv.op(i, 0) != OP_S && rv.op(i, L - 1) != OP_S)) return;
    const int idxs[2] = {0, L - 1};
    for (int q = 0; q < 2; ++q) {
      const int ci = idxs[q];
      if (rv.op(i, ci) != OP_S) continue;
      const int clen = rv.len(i, ci);
      if (read_repeat[0] == 0 && clen <= 16) continue;
      const int side = (ci == 0) ? 0 : 1;
      const strl_soft_rec *s = find_soft(i, side);
      if (!s) { err = STRL_ERR_ARG; set_error("missing soft-clip result for record %lld side %d", (long long)i, side); return; }
      const uint32_t w = first_seen ? s->res_first : s->res_after;
      const uint32_t cnt = STRL_RES_COUNT(w);
      if (cnt == 0) continue;
      if (cnt >= 256) { err = STRL_ERR_ASSERT; set_error("soft repeat_count %u >= 256 for record %lld", cnt, (long long)i); }
      strl_tread t{};
      const Unit u = unpack_unit(w);
      memcpy(t.repeat, u.s, 6);
      t.tid = r->tid[i];
      const int64_t p = (ci == 0) ? (int64_t)r->pos[i] : rv.stop(i);
      t.position = (uint32_t)std::max<int64_t>(0, p);
      t.flag = r->flag[i];
      t.repeat_count = (uint8_t)cnt;
      t.align_length = (uint8_t)clen;
      t.split = (ci == 0) ? STRL_SOFT_LEFT : STRL_SOFT_RIGHT;
      t.mapping_quality = r->mapq[i];
      t.qname_id = i;
      if (p_repeat(t) < 0.9) continue;
      S.emit(t, qn);
    }
  }
  // Cache.add, extract.nim:192-248
  void add(int64_t i) {
    const strl_records *r = rv.r;
    const std::string_view qn = rv.qname(i);
    strl_tread *stored = S.find(qn);
    const int32_t tid = r->tid[i], mtid = r->mtid[i], start = r->pos[i], mpos = r->mpos[i];
    const bool after_mate = tid > mtid || (tid == mtid && (start > mpos || (start == mpos && stored != nullptr)));
    if (after_mate) {
      if (!stored) return;
      strl_tread mate = *stored;
      S.erase(qn);
      strl_tread self = to_tread(i);
      add_soft(i, false, self.repeat, qn);
      if (mate.repeat_count == 0 && self.repeat_count == 0) return;
      if (unplaced_pair(self, mate, *o)) {
        if (self.repeat[0] == 0 || mate.repeat[0] == 0) return;
        canonical_repeat(self.repeat);
        self.position = 0;
        self.tid = -1;
        canonical_repeat(mate.repeat);
        mate.position = 0;
        mate.tid = -1;
        S.emit(self, qn);
        S.emit(mate, qn);
        return;
      }
      const uint32_t mp = mate.position;
      if (adjust_by(mate, self, *o, self.position)) S.emit(mate, qn);
      if (adjust_by(self, mate, *o, mp)) S.emit(self, qn);
    } else {
      strl_tread tr = to_tread(i);
      add_soft(i, true, tr.repeat, qn);
      if (stored) S.erase(qn);  // hasKeyOrPut hit: warn + take, the new tread is not stored (:245-248)
      else S.insert(qn, tr);
    }
  }
};

inline uint64_t hash_bytes(std::string_view s) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
  return h ^ (h >> 29);
}

}  // namespace

extern "C" {

int strl_soa_from_records(const strl_records *rec, int32_t *end, uint32_t *seq_off16, uint16_t *l_seq, uint16_t *clip_l,
                          uint16_t *clip_r, uint8_t *cig, uint32_t *max_l_seq) {
  if (!rec || !end || !seq_off16 || !l_seq || !clip_l || !clip_r || !cig) { set_error("null argument"); return STRL_ERR_ARG; }
  RecView rv{rec};
  uint32_t mx = 0;
  for (int64_t i = 0; i < rec->n; ++i) {
    const int L = rv.ncig(i);
    uint8_t c = 0;
    uint32_t cl = 0, cr = 0;
    if (L == 0) c |= STRL_CIG_NONE;
    else {
      if (L == 1) c |= STRL_CIG_ONE_OP;
      if (L == 1 && rv.op(i, 0) == OP_M) { c |= STRL_CIG_SINGLE_M; cl = (uint32_t)rv.len(i, 0); }   // extract.nim:33 align_length
      if (rv.op(i, 0) == OP_S) { c |= STRL_CIG_FIRST_S; cl = (uint32_t)rv.len(i, 0); }
      if (rv.op(i, L - 1) == OP_S) { c |= STRL_CIG_LAST_S; cr = (uint32_t)rv.len(i, L - 1); }
    }
    const int32_t ls = rec->l_seq[i];
    if (ls < 0 || ls > STRL_MAX_READ_LEN) { set_error("record %lld: l_seq %d outside [0, %d]", (long long)i, ls, STRL_MAX_READ_LEN); return STRL_ERR_ARG; }
    if (rec->seq_off[i] & 15u) { set_error("record %lld: seq_off not 16-byte aligned", (long long)i); return STRL_ERR_ARG; }
    if ((rec->seq_off[i] >> 4) > 0xffffffffull) { set_error("SEQ buffer exceeds 64 GiB"); return STRL_ERR_ARG; }
    cig[i] = c;
    clip_l[i] = (uint16_t)std::min<uint32_t>(cl, 65535u);
    clip_r[i] = (uint16_t)std::min<uint32_t>(cr, 65535u);
    l_seq[i] = (uint16_t)ls;
    seq_off16[i] = (uint32_t)(rec->seq_off[i] >> 4);
    const int64_t e = rv.stop(i);
    end[i] = (int32_t)e;
    mx = std::max(mx, (uint32_t)ls);
  }
  if (max_l_seq) *max_l_seq = mx;
  return STRL_OK;
}

int strl_pair_rows(const strl_records *rec, const int32_t *end, const uint16_t *clip_l, const uint16_t *clip_r, const uint8_t *cig,
                   strl_pair_rec *out) {
  if (!rec || (rec->n && (!end || !clip_l || !clip_r || !cig || !out))) { set_error("null argument"); return STRL_ERR_ARG; }
  for (int64_t i = 0; i < rec->n; ++i) {
    strl_pair_rec &o = out[i];
    o.tid = rec->tid[i]; o.pos = rec->pos[i]; o.mtid = rec->mtid[i]; o.mpos = rec->mpos[i]; o.end = end[i];
    o.flag = rec->flag[i]; o.l_seq = (uint16_t)rec->l_seq[i]; o.clip_l = clip_l[i]; o.clip_r = clip_r[i];
    o.mapq = rec->mapq[i]; o.cig = cig[i]; o.pad = 0;
  }
  return STRL_OK;
}

int strl_pair_reads(const strl_records *rec, const strl_opts *opts, const uint32_t *whole, const strl_soft_rec *soft,
                    uint64_t n_soft, int64_t n_tail, strl_tread *out, uint64_t cap, uint64_t *n_out) {
  if (!rec || !opts || (!whole && rec->n) || (!out && cap)) { set_error("null argument"); return STRL_ERR_ARG; }
  BatchStore store;
  store.out = out;
  store.cap = cap;
  Pairer<BatchStore> P{RecView{rec}, opts, whole, soft, n_soft, store};
  const int64_t n = rec->n;
  // Qname groups never interact (the table is keyed by qname), and a group none of whose records
  // carries a repeat produces no output: only replay Cache.add for the groups that can emit.
  std::vector<uint8_t> hot((size_t)n, 0);
  for (int64_t i = 0; i < n; ++i) hot[(size_t)i] = STRL_RES_COUNT(whole[i]) != 0;
  for (uint64_t s = 0; s < n_soft; ++s)
    if (STRL_RES_COUNT(soft[s].res_first) || STRL_RES_COUNT(soft[s].res_after)) hot[soft[s].read_side >> 1] = 1;
  std::unordered_set<uint64_t> hot_names;
  for (int64_t i = 0; i < n; ++i) if (hot[(size_t)i]) hot_names.insert(hash_bytes(P.rv.qname(i)));
  std::vector<uint8_t> sel((size_t)n, 0);
  for (int64_t i = 0; i < n; ++i) {
    if (rec->flag[i] & (F_SECONDARY | F_SUPPL)) continue;  // extract.nim:309
    sel[(size_t)i] = hot[(size_t)i] || hot_names.count(hash_bytes(P.rv.qname(i)));
  }
  for (int64_t i = 0; i < n; ++i) if (sel[(size_t)i]) P.add(i);                      // extract.nim:308-322
  if (n_tail < 0) { n_tail = 0; while (n_tail < n && rec->tid[n - 1 - n_tail] < 0) ++n_tail; }
  for (int64_t i = n - n_tail; i < n; ++i) if (sel[(size_t)i]) P.add(i);             // extract.nim:326-329 (tail revisited)
  if (n_out) *n_out = store.n_out;
  if (P.err) return P.err;
  if (store.n_out > cap) { set_error("tread capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)store.n_out); return STRL_ERR_CAPACITY; }
  return STRL_OK;
}

// ---- streaming pairer: the Cache of extract.nim:298 kept alive across batches (what the CLI drives) ----
// Qname groups never interact in Cache.add (the table is keyed by qname and every emission happens while the record
// that triggers it is processed), so the table is split into shards by qname hash and the shards of a batch run on
// their own threads.  Every emission carries (batch, record index); merging the shards by that key -- stable within a
// shard -- gives exactly the order a single sequential pass produces.
struct alignas(256) ShardStore {   // own cache lines: `cur` is written for every record by the shard's thread
  std::unordered_map<std::string, strl_tread> tbl;
  std::vector<strl_tread> out;
  std::vector<uint64_t> key;       // batch << 32 | record index of the record being processed at emission time
  std::vector<uint64_t> qoff{0};
  std::string qnames;
  std::string tmp;
  uint64_t cur = 0;
  strl_tread *find(std::string_view q) { tmp.assign(q); auto it = tbl.find(tmp); return it == tbl.end() ? nullptr : &it->second; }
  void erase(std::string_view q) { tmp.assign(q); tbl.erase(tmp); }
  void insert(std::string_view q, const strl_tread &t) { tbl.emplace(std::string(q), t); }
  void emit(strl_tread t, std::string_view q) {
    out.push_back(t);
    key.push_back(cur);
    qnames.append(q);
    qoff.push_back(qnames.size());
  }
};

// Fixed team of threads: run(fn) executes fn(t) on thread t of the team (t = 0 is the caller).  The mapping is fixed so
// that shard t of the pairer is always touched by the same OS thread (its allocations stay in one malloc arena).
struct Team {
  int n;
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  const std::function<void(int)> *fn = nullptr;
  uint64_t gen = 0;
  int left = 0;
  bool stop = false;
  explicit Team(int n_) : n(n_) {
    for (int t = 1; t < n; ++t)
      th.emplace_back([this, t] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)> *f;
          {
            std::unique_lock<std::mutex> lk(m);
            cv_go.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            f = fn;
          }
          (*f)(t);
          {
            std::lock_guard<std::mutex> lk(m);
            if (--left == 0) cv_done.notify_one();
          }
        }
      });
  }
  ~Team() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_go.notify_all();
    for (auto &x : th) x.join();
  }
  void run(const std::function<void(int)> &f) {
    if (n == 1) { f(0); return; }
    { std::lock_guard<std::mutex> lk(m); fn = &f; left = n - 1; ++gen; }
    cv_go.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return left == 0; });
  }
};

struct strl_pairer {
  strl_opts opts;
  std::vector<ShardStore> shards;
  uint64_t batch_no = 0;
  std::unique_ptr<Team> team;
  // merged view, rebuilt by strl_pairer_result
  std::vector<strl_tread> out;
  std::vector<uint64_t> qoff;
  std::string qnames;
};

static int pair_threads() {
  const char *e = getenv("STRL_PAIR_THREADS");
  if (e && atoi(e) > 0) return std::min(atoi(e), 64);
  return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
}
int strl_pairer_create(const strl_opts *opts, strl_pairer **out) {
  if (!opts || !out) { set_error("null argument"); return STRL_ERR_ARG; }
  strl_pairer *p = new strl_pairer;
  p->opts = *opts;
  p->shards.resize((size_t)pair_threads());
  p->team.reset(new Team((int)p->shards.size()));
  *out = p;
  return STRL_OK;
}
void strl_pairer_destroy(strl_pairer *p) { delete p; }

int strl_pairer_add(strl_pairer *p, const strl_records *rec, const uint32_t *whole, const strl_soft_rec *soft, uint64_t n_soft) {
  if (!p || !rec || (!whole && rec->n)) { set_error("null argument"); return STRL_ERR_ARG; }
  if (rec->n >= (1ll << 32)) { set_error("batch too large"); return STRL_ERR_ARG; }
  const int P = (int)p->shards.size();
  const int64_t n = rec->n;
  const uint64_t batch = p->batch_no++;
  std::vector<int> errs((size_t)P, 0);
  if (P == 1) {
    ShardStore &S = p->shards[0];
    Pairer<ShardStore> pr{RecView{rec}, &p->opts, whole, soft, n_soft, S};
    for (int64_t i = 0; i < n; ++i) {
      if (rec->flag[i] & (F_SECONDARY | F_SUPPL)) continue;   // extract.nim:309,327
      S.cur = (batch << 32) | (uint64_t)i;
      pr.add(i);
    }
    return pr.err;
  }
  std::vector<uint8_t> owner((size_t)n);
  const RecView rv{rec};
  p->team->run([&](int t) {                                     // who owns which record
    const int64_t i0 = n * t / P, i1 = n * (t + 1) / P;
    for (int64_t i = i0; i < i1; ++i) owner[(size_t)i] = (uint8_t)(hash_bytes(rv.qname(i)) % (uint64_t)P);
  });
  p->team->run([&](int t) {
    ShardStore &S = p->shards[(size_t)t];
    Pairer<ShardStore> pr{RecView{rec}, &p->opts, whole, soft, n_soft, S};
    for (int64_t i = 0; i < n; ++i) {
      if (owner[(size_t)i] != (uint8_t)t) continue;
      if (rec->flag[i] & (F_SECONDARY | F_SUPPL)) continue;   // extract.nim:309,327
      S.cur = (batch << 32) | (uint64_t)i;
      pr.add(i);
    }
    errs[(size_t)t] = pr.err;
  });
  for (int e : errs) if (e) return e;
  return STRL_OK;
}

int strl_pairer_result(strl_pairer *p, const strl_tread **treads, uint64_t *n, const uint64_t **qname_off, const char **qnames,
                       uint64_t *n_pending) {
  if (!p) return STRL_ERR_ARG;
  // k-way merge of the shards' emissions by (batch, record); each shard is already in that order
  struct Ref { uint64_t key; uint32_t shard, idx; };
  std::vector<Ref> refs;
  uint64_t pending = 0;
  for (size_t sh = 0; sh < p->shards.size(); ++sh) {
    const ShardStore &S = p->shards[sh];
    pending += S.tbl.size();
    for (size_t k = 0; k < S.out.size(); ++k) refs.push_back(Ref{S.key[k], (uint32_t)sh, (uint32_t)k});
  }
  std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.key < b.key; });
  p->out.clear(); p->qoff.assign(1, 0); p->qnames.clear();
  p->out.reserve(refs.size());
  for (const Ref &r : refs) {
    const ShardStore &S = p->shards[r.shard];
    strl_tread t = S.out[r.idx];
    t.qname_id = (int64_t)p->out.size();
    p->out.push_back(t);
    p->qnames.append(S.qnames, S.qoff[r.idx], S.qoff[r.idx + 1] - S.qoff[r.idx]);
    p->qoff.push_back(p->qnames.size());
  }
  if (treads) *treads = p->out.data();
  if (n) *n = p->out.size();
  if (qname_off) *qname_off = p->qoff.data();
  if (qnames) *qnames = p->qnames.data();
  if (n_pending) *n_pending = pending;
  return STRL_OK;
}

int strl_extract(strl_ctx *ctx, const strl_records *rec, int64_t n_tail, strl_tread *out, uint64_t cap, uint64_t *n_out,
                 strl_score_stats *stats) {
  if (!ctx || !rec) { set_error("null argument"); return STRL_ERR_ARG; }
  const size_t n = (size_t)rec->n;
  std::vector<int32_t> end(n);
  std::vector<uint32_t> so(n);
  std::vector<uint16_t> ls(n), cl(n), cr(n);
  std::vector<uint8_t> cig(n);
  std::vector<uint64_t> qh(n);
  uint32_t mx = 0;
  int rc = strl_soa_from_records(rec, end.data(), so.data(), ls.data(), cl.data(), cr.data(), cig.data(), &mx);
  if (rc) return rc;
  if ((rc = strl_qname_hash(rec, qh.data()))) return rc;
  uint64_t seq_bytes = 32;
  for (size_t i = 0; i < n; ++i) seq_bytes = std::max<uint64_t>(seq_bytes, rec->seq_off[i] + (uint64_t)((rec->l_seq[i] + 1) / 2) + 32);
  strl_read_soa soa{};
  soa.n = n; soa.tid = rec->tid; soa.pos = rec->pos; soa.end = end.data(); soa.seq_off = so.data(); soa.l_seq = ls.data();
  soa.clip_l = cl.data(); soa.clip_r = cr.data(); soa.mapq = rec->mapq; soa.cig = cig.data(); soa.seq4 = rec->seq4;
  soa.seq4_bytes = seq_bytes; soa.max_l_seq = mx; soa.mem = STRL_MEM_HOST;
  std::vector<strl_pair_rec> rows(n);
  if ((rc = strl_pair_rows(rec, end.data(), cl.data(), cr.data(), cig.data(), rows.data()))) return rc;
  strl_pair_soa pp{rows.data(), qh.data()};
  if (n_tail < 0) { n_tail = 0; while ((size_t)n_tail < n && rec->tid[n - 1 - (size_t)n_tail] < 0) ++n_tail; }
  // scoring + pair logic on the device; capacities first from the defaults, then from the hard bounds
  for (int attempt = 0; attempt < 2; ++attempt) {
    const uint64_t icap = attempt ? 3 * (uint64_t)n + 16 : 0, tcap = attempt ? 8 * (uint64_t)n + 16 : 0;
    if ((rc = strl_extract_device(ctx, &soa, &pp, n_tail, icap, tcap))) return rc;
    uint64_t need = 0;
    rc = strl_treads_fetch(ctx, nullptr, 0, &need, stats);
    if (rc == STRL_ERR_CAPACITY && attempt == 0) continue;
    if (rc == STRL_ERR_FORMAT) break;                 // > 12 records under one qname hash: the host pair logic takes it
    if (rc) return rc;
    if (n_out) *n_out = need;
    if (need > cap) { set_error("tread capacity %llu too small, need %llu", (unsigned long long)cap, (unsigned long long)need); return STRL_ERR_CAPACITY; }
    return strl_treads_fetch(ctx, out, cap, n_out, nullptr);
  }
  std::vector<uint32_t> whole(n);
  std::vector<strl_soft_rec> soft(2 * n + 1);
  uint64_t ns = 0;
  rc = strl_score_reads(ctx, &soa, whole.data(), soft.data(), 2 * n, &ns, stats);
  if (rc) return rc;
  return strl_pair_reads(rec, &ctx->opts, whole.data(), soft.data(), ns, n_tail, out, cap, n_out);
}

int strl_qname_hash(const strl_records *rec, uint64_t *out) {
  if (!rec || (!out && rec->n)) { set_error("null argument"); return STRL_ERR_ARG; }
  RecView rv{rec};
  for (int64_t i = 0; i < rec->n; ++i) out[i] = hash_bytes(rv.qname(i));
  return STRL_OK;
}

int strl_frag_median(const uint32_t frag[4096], double pct) {  // utils.nim:139-146
  uint32_t n = 0;
  for (int i = 0; i < 4096; ++i) n += frag[i];
  const uint32_t want = (uint32_t)(0.5 + (double)n / (1.0 / pct));
  uint32_t c = 0;
  for (int i = 0; i < 4096; ++i) {
    c += frag[i];
    if (c >= want) return i;
  }
  return 4096;
}

// ---- the .bin writer: msgpack subset of the records, msgpack4nim picks the smallest encoding of every value (cluster.nim:38-50) ----
int strl_bin_write(const char *path, float proportion_repeat, uint8_t min_mapq, const uint32_t frag[4096], const char *sam_header,
                   int32_t header_len, const strl_tread *treads, uint64_t n, const uint64_t *qname_off, const char *qnames) {
  const auto t_begin = std::chrono::steady_clock::now();
  FILE *f = fopen(path, "wb");
  if (!f) { set_error("[strling] couldnt open binary output file %s", path); return STRL_ERR_IO; }
  std::string b;
  b.reserve(1 << 20);
  b.append("STR", 3);                                       // extract.nim:336
  const int16_t fmt = 0;                                    // version.nim:4
  b.append((const char *)&fmt, 2);
  char ver[9] = {0};
  memcpy(ver, "0.6.0", 5);                                  // version.nim:1
  b.append(ver, 9);
  b.append((const char *)&proportion_repeat, 4);
  b.push_back((char)min_mapq);
  b.append((const char *)frag, 4096 * 4);
  b.append((const char *)&header_len, 4);
  b.append(sam_header, (size_t)header_len);
  const int32_t n32 = (int32_t)n;
  b.append((const char *)&n32, 4);
  // pack_type, cluster.nim:38-50, through a raw cursor: a tread is ~35 values of one to five bytes, and appending them to a
  // std::string one push_back at a time (a capacity check each) was 0.4 - 0.7 us a tread -- the packing, not the file system,
  // was what `writing the .bin` took (0.24 - 0.29 s for a whole genome's 8e6 treads on 16 threads)
  struct W {
    static inline uint8_t *u(uint8_t *o, uint64_t v) {
      if (v < 128) { *o++ = (uint8_t)v; return o; }
      if (v < 256) { o[0] = 0xcc; o[1] = (uint8_t)v; return o + 2; }
      if (v < 65536) { o[0] = 0xcd; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)v; return o + 3; }
      o[0] = 0xce; o[1] = (uint8_t)(v >> 24); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 8); o[4] = (uint8_t)v;
      return o + 5;
    }
    static inline uint8_t *i(uint8_t *o, int32_t v) {
      if (v >= 0) return u(o, (uint64_t)v);
      if (v >= -32) { *o++ = (uint8_t)v; return o; }
      if (v >= -128) { o[0] = 0xd0; o[1] = (uint8_t)v; return o + 2; }
      if (v >= -32768) { o[0] = 0xd1; o[1] = (uint8_t)((uint16_t)v >> 8); o[2] = (uint8_t)v; return o + 3; }
      const uint32_t x = (uint32_t)v;
      o[0] = 0xd2; o[1] = (uint8_t)(x >> 24); o[2] = (uint8_t)(x >> 16); o[3] = (uint8_t)(x >> 8); o[4] = (uint8_t)x;
      return o + 5;
    }
    static inline uint8_t *str(uint8_t *o, const char *p, size_t l) {
      if (l < 32) *o++ = (uint8_t)(0xa0 | l);
      else if (l < 256) { o[0] = 0xd9; o[1] = (uint8_t)l; o += 2; }
      else if (l < 65536) { o[0] = 0xda; o[1] = (uint8_t)(l >> 8); o[2] = (uint8_t)l; o += 3; }
      else { o[0] = 0xdb; o[1] = (uint8_t)(l >> 24); o[2] = (uint8_t)(l >> 16); o[3] = (uint8_t)(l >> 8); o[4] = (uint8_t)l; o += 5; }
      memcpy(o, p, l);
      return o + l;
    }
  };
  constexpr size_t FIXED_MAX = 5 + 5 + 1 + 6 * 2 + 3 + 2 + 2 + 2 + 2 + 5 + 5;     // every value of a tread at its widest, without the name's bytes
  auto pack = [&](uint8_t *o, uint64_t i0, uint64_t i1) -> uint8_t * {
    for (uint64_t i = i0; i < i1; ++i) {
      const strl_tread &t = treads[i];
      o = W::i(o, t.tid);
      o = W::u(o, t.position);
      *o++ = 0x96;
      for (int j = 0; j < 6; ++j) o = W::u(o, (uint8_t)t.repeat[j]);
      o = W::u(o, t.flag);
      o = W::u(o, t.split);
      o = W::u(o, t.mapping_quality);
      o = W::u(o, t.repeat_count);
      o = W::u(o, t.align_length);
      const uint64_t q0 = qname_off[t.qname_id], q1 = qname_off[t.qname_id + 1];
      o = W::u(o, q1 - q0);
      o = W::str(o, qnames + q0, (size_t)(q1 - q0));
    }
    return o;
  };
  auto room = [&](uint64_t i0, uint64_t i1) {          // bytes the treads [i0, i1) take at most
    size_t names = 0;
    for (uint64_t i = i0; i < i1; ++i) names += (size_t)(qname_off[treads[i].qname_id + 1] - qname_off[treads[i].qname_id]);
    return (size_t)(i1 - i0) * FIXED_MAX + names;
  };
  // A whole genome leaves millions of treads (~35 bytes each).  Ranges of 2^16 treads are packed by a few threads into a
  // buffer each thread keeps (warm: no fresh pages per range) and written straight to their place in the file -- the sizes of
  // all ranges are known beforehand from a first, cheap pass of the same code.  The bytes are the sequential writer's.
  const unsigned hw = std::thread::hardware_concurrency();
  const uint64_t per = 1 << 16;
  const unsigned n_thr = n < 4 * per ? 1u : std::min<unsigned>({hw ? hw : 1u, 16u, (unsigned)((n + per - 1) / per)});
  if (fwrite(b.data(), 1, b.size(), f) != b.size()) { fclose(f); set_error("short write to %s", path); return STRL_ERR_IO; }
  b.clear();
  if (n_thr <= 1) {
    std::vector<uint8_t> buf;
    for (uint64_t i0 = 0; i0 < n; i0 += per) {
      const uint64_t i1 = std::min(n, i0 + per);
      buf.resize(room(i0, i1));
      const size_t len = (size_t)(pack(buf.data(), i0, i1) - buf.data());
      if (fwrite(buf.data(), 1, len, f) != len) { fclose(f); set_error("short write to %s", path); return STRL_ERR_IO; }
    }
  } else {
    const uint64_t n_parts = (n + per - 1) / per;
    std::vector<off_t> at((size_t)n_parts + 1, 0);
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    auto usz = [](uint64_t v) -> size_t { return v < 128 ? 1 : v < 256 ? 2 : v < 65536 ? 3 : 5; };
    auto isz = [&](int32_t v) -> size_t { return v >= 0 ? usz((uint64_t)v) : v >= -32 ? 1 : v >= -128 ? 2 : v >= -32768 ? 3 : 5; };
    for (unsigned k = 0; k < n_thr; ++k)          // pass 1: every range's size, by arithmetic (pass 2 checks it against what it packs)
      th.emplace_back([&] {
        for (uint64_t q; (q = next.fetch_add(1)) < n_parts;) {
          const uint64_t i0 = q * per, i1 = std::min(n, i0 + per);
          size_t bytes = 0;
          for (uint64_t i = i0; i < i1; ++i) {
            const strl_tread &t = treads[i];
            const uint64_t ql = qname_off[t.qname_id + 1] - qname_off[t.qname_id];
            size_t b = isz(t.tid) + usz(t.position) + 1 + usz(t.flag) + usz(t.split) + usz(t.mapping_quality) + usz(t.repeat_count) + usz(t.align_length) + usz(ql) +
                       (ql < 32 ? 1 : ql < 256 ? 2 : ql < 65536 ? 3 : 5) + (size_t)ql;
            for (int j = 0; j < 6; ++j) b += usz((uint8_t)t.repeat[j]);
            bytes += b;
          }
          at[(size_t)q + 1] = (off_t)bytes;
        }
      });
    for (auto &t : th) t.join();
    const auto t_packed = std::chrono::steady_clock::now();
    fflush(f);
    at[0] = ftello(f);
    for (uint64_t q = 0; q < n_parts; ++q) at[(size_t)q + 1] += at[(size_t)q];
    const int fd = fileno(f);
    // (the file at its final size first: the ranges then land in allocated space instead of each growing the file)
    if (ftruncate(fd, at[(size_t)n_parts]) != 0) { /* not fatal: pwrite extends the file as it goes */ }
    std::atomic<bool> bad{false};
    next = 0;
    th.clear();
    for (unsigned k = 0; k < n_thr; ++k)          // pass 2: packed again (the buffer is warm) and written where the range belongs
      th.emplace_back([&] {
        std::vector<uint8_t> buf;
        for (uint64_t q; (q = next.fetch_add(1)) < n_parts;) {
          const uint64_t i0 = q * per, i1 = std::min(n, i0 + per);
          const size_t need = room(i0, i1);
          if (buf.size() < need) buf.resize(need + need / 8);
          const size_t len = (size_t)(pack(buf.data(), i0, i1) - buf.data());
          if ((off_t)len != at[(size_t)q + 1] - at[(size_t)q]) { bad = true; break; }
          size_t done = 0;
          while (done < len) {
            const ssize_t w = pwrite(fd, buf.data() + done, len - done, at[(size_t)q] + (off_t)done);
            if (w <= 0) { bad = true; break; }
            done += (size_t)w;
          }
        }
      });
    for (auto &t : th) t.join();
    if (getenv("STRL_BIN_TIMING"))
      fprintf(stderr, "[strling] .bin: %u threads, sizes %.3f s, packing + writing %.3f s\n", n_thr, std::chrono::duration<double>(t_packed - t_begin).count(),
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t_packed).count());
    if (bad.load() || fseeko(f, at[(size_t)n_parts], SEEK_SET) != 0) { fclose(f); set_error("short write to %s", path); return STRL_ERR_IO; }
  }
  if (fwrite(b.data(), 1, b.size(), f) != b.size()) { fclose(f); set_error("short write to %s", path); return STRL_ERR_IO; }
  fclose(f);
  return STRL_OK;
}

namespace {
struct Rd {
  const uint8_t *p, *e;
  bool ok = true;
  uint64_t be(int n) { uint64_t v = 0; for (int i = 0; i < n; ++i) { if (p >= e) { ok = false; return 0; } v = (v << 8) | *p++; } return v; }
  int64_t integer() {
    if (p >= e) { ok = false; return 0; }
    const uint8_t t = *p++;
    if (t < 0x80) return t;
    if (t >= 0xe0) return (int8_t)t;
    switch (t) {
      case 0xcc: return (int64_t)be(1);
      case 0xcd: return (int64_t)be(2);
      case 0xce: return (int64_t)be(4);
      case 0xcf: return (int64_t)be(8);
      case 0xd0: return (int8_t)be(1);
      case 0xd1: return (int16_t)be(2);
      case 0xd2: return (int32_t)be(4);
      case 0xd3: return (int64_t)be(8);
      default: ok = false; return 0;
    }
  }
  size_t strhdr() {
    if (p >= e) { ok = false; return 0; }
    const uint8_t t = *p++;
    if ((t & 0xe0) == 0xa0) return t & 0x1f;
    if (t == 0xd9) return (size_t)be(1);
    if (t == 0xda) return (size_t)be(2);
    if (t == 0xdb) return (size_t)be(4);
    ok = false;
    return 0;
  }
};
}  // namespace

int strl_bin_peek(const char *path, strl_bin_info *info) {
  if (!path || !info) { set_error("null argument"); return STRL_ERR_ARG; }
  FILE *f = fopen(path, "rb");
  if (!f) { set_error("[strling] unable to open %s for reading. please check path", path); return STRL_ERR_IO; }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  const size_t fixed = 3 + 2 + 9 + 4 + 1 + 4096 * 4 + 4;
  std::vector<uint8_t> buf(fixed);
  if (sz < (long)fixed + 4 || fread(buf.data(), 1, fixed, f) != fixed || memcmp(buf.data(), "STR", 3) != 0) {
    fclose(f);
    set_error("[strling] expected bin file to start with \"STR\"");
    return STRL_ERR_FORMAT;
  }
  int16_t fmt;
  memcpy(&fmt, buf.data() + 3, 2);
  if (fmt != 0) { fclose(f); set_error("[strling] this bin file was generated using a different format"); return STRL_ERR_FORMAT; }
  size_t o = 3 + 2 + 9;
  memcpy(&info->proportion_repeat, buf.data() + o, 4); o += 4;
  info->min_mapq = buf[o++];
  memcpy(info->frag, buf.data() + o, 4096 * 4); o += 4096 * 4;
  memcpy(&info->header_len, buf.data() + o, 4);
  int32_t n = 0;
  const bool ok = info->header_len >= 0 && (long)fixed + info->header_len + 4 <= sz && fseek(f, (long)fixed + info->header_len, SEEK_SET) == 0 && fread(&n, 1, 4, f) == 4;
  fclose(f);
  if (!ok || n < 0) { set_error("truncated bin header"); return STRL_ERR_FORMAT; }
  info->n_reads = n;
  info->qnames_bytes = (uint64_t)sz - fixed - (uint64_t)info->header_len - 4;
  // (callers size their arrays from n_reads: a record is 15 bytes at the very least -- a count the file cannot hold is refused
  // here, not allocated for and then found wrong by the read, unpack.nim:130-131)
  if ((uint64_t)n * 15 > info->qnames_bytes) { set_error("[strling] expected %d got fewer: the file holds %llu bytes of records", n, (unsigned long long)info->qnames_bytes); return STRL_ERR_FORMAT; }
  return STRL_OK;
}

int strl_bin_read(const char *path, strl_bin_info *info, char *sam_header, strl_tread *treads, uint64_t *qname_off, char *qnames) {
  if (!path || !info) { set_error("null argument"); return STRL_ERR_ARG; }
  const auto t_open = std::chrono::steady_clock::now();
  FILE *f = fopen(path, "rb");
  if (!f) { set_error("[strling] unable to open %s for reading. please check path", path); return STRL_ERR_IO; }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  // (a whole genome's .bin is a quarter of a gigabyte: read in pieces by a few threads into memory nobody zeroes first)
  // ... mapped with huge pages where the kernel grants them (MADV_HUGEPAGE: a hundred faults instead of sixty thousand)
  struct Bytes {
    uint8_t *p = nullptr; size_t n = 0, mapped = 0; void *base = nullptr;
    uint8_t *data() { return p; }
    size_t size() const { return n; }
    uint8_t &operator[](size_t i) { return p[i]; }
    ~Bytes() { if (base) munmap(base, mapped); else delete[] p; }
  } buf;
  buf.n = (size_t)std::max(0L, sz);
  if (buf.n >= ((size_t)8 << 20)) {
    const size_t huge = (size_t)2 << 20;
    buf.mapped = ((buf.n + 16 + huge - 1) & ~(huge - 1)) + huge;
    void *m = mmap(nullptr, buf.mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m != MAP_FAILED) {
      buf.base = m;
      buf.p = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(m) + huge - 1) & ~(uintptr_t)(huge - 1));
      (void)madvise(buf.p, buf.mapped - huge, MADV_HUGEPAGE);
    }
  }
  if (!buf.p) buf.p = new uint8_t[buf.n + 16];
  {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = buf.n < ((size_t)32 << 20) ? 1 : std::min<size_t>({hw ? hw : 1u, 8u, buf.n >> 24});
    const int fd = fileno(f);
    std::atomic<bool> bad{false};
    auto part = [&](size_t k) {
      size_t a = buf.n / T * k;
      const size_t e = k + 1 == T ? buf.n : buf.n / T * (k + 1);
      while (a < e) {
        const ssize_t r = pread(fd, buf.data() + a, e - a, (off_t)a);
        if (r <= 0) { bad = true; return; }
        a += (size_t)r;
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < T; ++k) th.emplace_back(part, k);
    part(0);
    for (auto &t : th) t.join();
    if (bad.load()) { fclose(f); set_error("short read from %s", path); return STRL_ERR_IO; }
  }
  fclose(f);
  const bool tm = getenv("STRL_BIN_TIMING") != nullptr;
  const auto tm0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (tm) fprintf(stderr, "[strl_bin_read] %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count()); };
  if (tm) fprintf(stderr, "[strl_bin_read] file read into memory in %.3f s\n", std::chrono::duration<double>(tm0 - t_open).count());
  const size_t fixed = 3 + 2 + 9 + 4 + 1 + 4096 * 4 + 4;
  if (buf.size() < fixed + 4 || memcmp(buf.data(), "STR", 3) != 0) {                // unpack.nim:61-62
    set_error("[strling] expected bin file to start with \"STR\"");
    return STRL_ERR_FORMAT;
  }
  int16_t fmt;
  memcpy(&fmt, buf.data() + 3, 2);
  if (fmt != 0) { set_error("[strling] this bin file was generated using a different format"); return STRL_ERR_FORMAT; }   // :64-66
  size_t o = 3 + 2 + 9;
  memcpy(&info->proportion_repeat, buf.data() + o, 4); o += 4;
  info->min_mapq = buf[o++];
  memcpy(info->frag, buf.data() + o, 4096 * 4); o += 4096 * 4;
  memcpy(&info->header_len, buf.data() + o, 4); o += 4;
  if (info->header_len < 0 || o + (size_t)info->header_len + 4 > buf.size()) { set_error("truncated bin header"); return STRL_ERR_FORMAT; }
  if (sam_header) memcpy(sam_header, buf.data() + o, (size_t)info->header_len);
  o += (size_t)info->header_len;
  memcpy(&info->n_reads, buf.data() + o, 4); o += 4;
  // One record (unpack_type, unpack.nim:36-55) at rd.p; `strict` also asks for what the writers guarantee (value ranges, the name's
  // length equal to its length field): that is how a thread that starts in the middle of the file recognises a record start.
  auto one = [](Rd &rd, strl_tread &t, const uint8_t *&name, size_t &sl, bool strict) -> bool {
    const int64_t tid = rd.integer(), pos = rd.integer();
    t.tid = (int32_t)tid;
    t.position = (uint32_t)pos;
    if (!rd.ok || rd.p >= rd.e || *rd.p++ != 0x96) return false;
    int64_t v[11];
    for (int j = 0; j < 11; ++j) v[j] = rd.integer();
    for (int j = 0; j < 6; ++j) t.repeat[j] = (char)v[j];
    t.flag = (uint16_t)v[6]; t.split = (uint8_t)v[7]; t.mapping_quality = (uint8_t)v[8]; t.repeat_count = (uint8_t)v[9]; t.align_length = (uint8_t)v[10];
    const uint64_t L = (uint64_t)rd.integer();
    sl = 0;
    name = rd.p;
    if (L > 0) {                                                                     // :51-54 (qname read only when L > 0)
      sl = rd.strhdr();
      if (!rd.ok || rd.p + sl > rd.e) return false;
      name = rd.p;
      rd.p += sl;
    }
    if (!rd.ok) return false;
    if (strict) {
      if (tid < -1 || tid > INT32_MAX || pos < 0 || pos > (int64_t)UINT32_MAX || sl != L || L > 255) return false;
      for (int j = 0; j < 6; ++j) if (v[j] != 0 && v[j] != 'A' && v[j] != 'C' && v[j] != 'G' && v[j] != 'T') return false;
      if (v[6] < 0 || v[6] > 65535) return false;
      for (int j = 7; j < 11; ++j) if (v[j] < 0 || v[j] > 255) return false;
    }
    return true;
  };
  // A whole genome's records (8e6, 0.4 s of dependent token decoding on one thread) are parsed by several threads: thread k starts at
  // the first offset at or behind its share's start where EIGHT records in a row parse strictly, and parses up to the first record
  // boundary at or behind the next share's start.  The guess is verified, not trusted: only if every thread ended exactly where
  // the next one started (and the last one at the end of the file, and the count is the header's) are the parts put together --
  // by induction from the true first record every boundary then is a true one.  Anything else: the sequential walk below.
  const size_t body0 = o;
  bool done_parallel = false;
  const char *seq_env = getenv("STRL_BIN_READ");            // "seq": the sequential walk whatever the size; "parts=K": K parts whatever the size (tests)
  const size_t forced_parts = seq_env && !strncmp(seq_env, "parts=", 6) ? (size_t)std::min(64, std::max(0, atoi(seq_env + 6))) : 0;
  if (treads && qname_off && qnames && (buf.size() - body0 >= ((size_t)32 << 20) || (forced_parts >= 2 && buf.size() - body0 >= forced_parts)) && info->n_reads > 0 &&
      !(seq_env && !strcmp(seq_env, "seq"))) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t K = forced_parts >= 2 ? forced_parts : std::min<size_t>({hw ? hw : 1u, 16u, (buf.size() - body0) >> 22});
    if (K >= 2) {
      // (two passes per share -- count, then parse into place -- instead of per-thread copies: the second pass costs a thread
      // 30 ms, a few hundred megabytes of freshly mapped temporaries cost page faults by the ten thousand)
      struct Part { size_t start = 0, second = 0, end = 0; bool ok = false; uint64_t n = 0, name_bytes = 0, first_name = 0; };
      std::vector<Part> parts(K);
      const uint8_t *base = buf.data(), *fend = buf.data() + buf.size();
      auto share = [&](size_t k) { return body0 + (buf.size() - body0) / K * k; };
      auto work = [&](size_t k) {
        Part &P = parts[k];
        const size_t lo = share(k), hi = k + 1 == K ? buf.size() : share(k + 1);
        size_t s0 = lo;
        if (k > 0) {
          bool found = false;
          for (; s0 < hi && s0 < lo + 4096 && !found; ++s0) {
            Rd r{base + s0, fend};
            bool good = true;
            for (int q = 0; q < 8 && good && r.p < r.e; ++q) { strl_tread t{}; const uint8_t *nm; size_t sl; good = one(r, t, nm, sl, true); }
            if (good) { found = true; break; }
          }
          if (!found) return;
        }
        P.start = s0;
        Rd r{base + s0, fend};
        uint64_t n_here = 0, names_here = 0;         // (locals: the threads' Part records share cache lines -- counting in place made
                                                     // every record of every thread fight for them: 0.16 s of a 0.27 s read)
        while (r.p < r.e && (size_t)(r.p - base) < hi) {
          strl_tread t{};
          const uint8_t *nm;
          size_t sl;
          if (!one(r, t, nm, sl, false)) return;
          if (!n_here) { P.second = (size_t)(r.p - base); P.first_name = sl; }
          ++n_here;
          names_here += sl;
        }
        P.n = n_here;
        P.name_bytes = names_here;
        P.end = (size_t)(r.p - base);
        P.ok = true;
      };
      std::vector<std::thread> th;
      for (size_t k = 1; k < K; ++k) th.emplace_back(work, k);
      work(0);
      for (auto &t : th) t.join();
      lap("records counted");
      // One wrong guess is common enough to be taken care of (round 6: it had sent EVERY whole-genome file this far down the
      // sequential walk -- 0.2 s instead of 0.05): a share that starts on the last two bytes of a record's four- or eight-byte
      // position reads them as a one-byte tid and a one-byte position, finds the array marker behind them, and is in step with
      // the true records from there on -- eight strict records in a row, the first of them made up.  The part before it ends at
      // the true boundary behind the share's start, which then is where this part's SECOND record starts: the made-up first one
      // is dropped.  The induction is as before: a part starts where the part before it, which started at a true boundary, ended.
      bool linked = parts[0].ok;
      uint64_t total = 0;
      for (size_t k = 1; k < K && linked; ++k) {
        Part &P = parts[k];
        linked = P.ok;
        if (linked && parts[k - 1].end != P.start) {
          if (P.n >= 1 && parts[k - 1].end == P.second) { P.start = P.second; P.n -= 1; P.name_bytes -= P.first_name; }
          else linked = false;
        }
      }
      linked = linked && parts[K - 1].end == buf.size();
      for (size_t k = 0; k < K; ++k) total += parts[k].n;
      if (tm && !(linked && total == (uint64_t)info->n_reads))
        for (size_t k = 0; k < K; ++k) fprintf(stderr, "[strl_bin_read] part %zu: ok %d start %zu end %zu n %llu (share starts at %zu)\n", k, (int)parts[k].ok, parts[k].start, parts[k].end, (unsigned long long)parts[k].n, share(k));
      if (linked && total == (uint64_t)info->n_reads) {
        std::vector<uint64_t> r0(K + 1, 0), q0(K + 1, 0);
        for (size_t k = 0; k < K; ++k) { r0[k + 1] = r0[k] + parts[k].n; q0[k + 1] = q0[k] + parts[k].name_bytes; }
        auto place = [&](size_t k) {
          const Part &P = parts[k];
          Rd r{base + P.start, base + P.end};
          uint64_t i = r0[k], qb = q0[k];
          while (r.p < r.e) {
            strl_tread t{};
            const uint8_t *nm;
            size_t sl;
            if (!one(r, t, nm, sl, false)) break;          // (cannot happen: the same bytes parsed a moment ago)
            t.qname_id = (int64_t)i;
            treads[i] = t;
            qname_off[i] = qb;
            memcpy(qnames + qb, nm, sl);
            ++i;
            qb += sl;
          }
        };
        th.clear();
        for (size_t k = 1; k < K; ++k) th.emplace_back(place, k);
        place(0);
        for (auto &t : th) t.join();
        lap("records parsed into place");
        qname_off[total] = q0[K];
        info->qnames_bytes = q0[K];
        done_parallel = true;
      }
    }
  }
  if (done_parallel) return STRL_OK;
  Rd rd{buf.data() + o, buf.data() + buf.size()};
  uint64_t qbytes = 0;
  int64_t i = 0;
  while (rd.p < rd.e) {                                                              // unpack_type, unpack.nim:36-55
    strl_tread t{};
    t.tid = (int32_t)rd.integer();
    t.position = (uint32_t)rd.integer();
    if (rd.p >= rd.e || *rd.p++ != 0x96) { rd.ok = false; break; }
    for (int j = 0; j < 6; ++j) t.repeat[j] = (char)rd.integer();
    t.flag = (uint16_t)rd.integer();
    t.split = (uint8_t)rd.integer();
    t.mapping_quality = (uint8_t)rd.integer();
    t.repeat_count = (uint8_t)rd.integer();
    t.align_length = (uint8_t)rd.integer();
    const uint64_t L = (uint64_t)rd.integer();
    size_t sl = 0;
    if (L > 0) {                                                                     // :51-54 (qname read only when L > 0)
      sl = rd.strhdr();
      if (!rd.ok || rd.p + sl > rd.e) { rd.ok = false; break; }
      if (qnames && i < (int64_t)info->n_reads) memcpy(qnames + qbytes, rd.p, sl);
      rd.p += sl;
    }
    if (!rd.ok) break;
    t.qname_id = i;
    // (buffers are sized from the header's n_reads, strl_bin_peek: records beyond it are counted -- the mismatch is an error below -- not stored)
    if (treads && i < (int64_t)info->n_reads) treads[i] = t;
    if (qname_off && i < (int64_t)info->n_reads) qname_off[i] = qbytes;
    qbytes += sl;
    ++i;
  }
  if (!rd.ok) { set_error("malformed msgpack record %lld in %s", (long long)i, path); return STRL_ERR_FORMAT; }
  if (qname_off && i <= (int64_t)info->n_reads) qname_off[i] = qbytes;
  if (i != info->n_reads) { set_error("[strling] expected %d got %lld", info->n_reads, (long long)i); return STRL_ERR_FORMAT; }   // :130-131
  info->qnames_bytes = qbytes;
  return STRL_OK;
}

extern "C" int strl_pair_rule_device(strl_ctx *c, int op, strl_tread *A, const strl_tread *B, const strl_opts *o, uint32_t B_position, int *result);
int strl_pair_rule(strl_ctx *c, int op, strl_tread *A, const strl_tread *B, const strl_opts *o, uint32_t B_position, int *result) {
  if (c) return strl_pair_rule_device(c, op, A, B, o, B_position, result);
  if (!A || !B || !o || !result || op < 0 || op > 2) { set_error("bad argument"); return STRL_ERR_ARG; }
  if (op == STRL_RULE_ADJUST_BY) *result = adjust_by(*A, *B, *o, B_position) ? 1 : 0;
  else if (op == STRL_RULE_UNPLACED_PAIR) *result = unplaced_pair(*A, *B, *o) ? 1 : 0;
  else { canonical_repeat(A->repeat); *result = 0; }
  return STRL_OK;
}

void strl_canonical_repeat(const char in[6], char out[6]) {   // utils.nim:304-316
  memcpy(out, in, 6);
  canonical_repeat(out);
}

int strl_bounds_row(char *buf, int cap, const strl_bounds *b, const char *chrom) {   // cluster.nim:262-266
  return snprintf(buf, (size_t)cap, "%s\t%u\t%u\t%s\t\t%u\t%u\t%u\t%u\t%u\t%u", chrom, b->left, b->right, b->repeat, b->left_most,
                  b->right_most, b->center_mass, (unsigned)b->n_left, (unsigned)b->n_right, (unsigned)b->n_total);
}

}  // extern "C"

// ---- strling index: window merge + trim, genome_strs.nim:22-92 ------------------------------------------------
namespace {
inline uint32_t base_code_ci(char ch) {   // after toUpperAscii (genome_strs.nim:71); anything not ACGT counts as 'A'
  switch (ch) { case 'C': case 'c': return 0; case 'T': case 't': return 2; case 'G': case 'g': return 3; default: return 1; }
}
// minimum rotation of the k-mer at s[0..k) (utils.nim:10-34, one window)
inline uint32_t min_rot(const char *s, int k, bool reversed) {
  uint32_t f = 0;
  for (int j = 0; j < k; ++j) f = (f << 2) | base_code_ci(reversed ? s[-j] : s[j]);
  const uint32_t mask = (1u << (2 * k)) - 1u;
  uint32_t best = f;
  for (int j = 1; j < k; ++j) { f = ((f << 2) | (f >> (2 * k - 2))) & mask; best = std::min(best, f); }
  return best;
}
// trim, genome_strs.nim:22-59: walk k-sized steps in from both ends until a step is a rotation of the unit
bool trim_region(const char *seq, const Unit &u, uint64_t &start, uint64_t &stop) {
  const int k = u.len;
  const uint64_t len = stop - start, nwin = len / (uint64_t)k;
  const uint32_t fwd = min_rot(u.s, k, false);
  const uint32_t rev = min_rot(u.s + k - 1, k, true);
  const char *dna = seq + start;
  for (uint64_t i = 0; i < nwin; ++i) { if (min_rot(dna + i * k, k, false) != fwd) start += k; else break; }
  if (!(start < stop)) return false;
  const char *last = dna + len - 1;
  for (uint64_t i = 0; i < nwin; ++i) { if (min_rot(last - i * k, k, true) != rev) stop -= k; else break; }
  return start < stop;
}
}  // namespace

extern "C" int strl_index_regions(const char *seq, uint64_t n_bases, const uint32_t *words, uint64_t n_windows, uint32_t window,
                                  uint32_t step, strl_region *out, uint64_t cap, uint64_t *n_out) {
  if ((!seq && n_bases) || (!words && n_windows) || !n_out || !window || !step || step > window) { set_error("bad argument"); return STRL_ERR_ARG; }
  uint64_t n = 0;
  int err = STRL_OK;
  const uint64_t slack = window - step;
  bool have = false;
  uint64_t l_start = 0, l_stop = 0;
  uint32_t l_word = 0;
  auto flush = [&]() {     // genome_strs.nim:79-82 / :88-91
    if (!have || l_stop - l_start < slack) return;
    uint64_t a = l_start >= window ? l_start - window : 0, b = std::min<uint64_t>(l_stop + window, n_bases);
    const Unit u = unpack_unit(l_word);
    if (!trim_region(seq, u, a, b)) {
      if (!err) set_error("repeat %.6s not found in expected region %llu-%llu (doAssert genome_strs.nim:39/57)", u.s, (unsigned long long)l_start, (unsigned long long)l_stop);
      err = STRL_ERR_ASSERT;
      return;
    }
    if (n < cap && out) { out[n].start = a; out[n].stop = b; memset(out[n].unit, 0, sizeof out[n].unit); memcpy(out[n].unit, u.s, 6); }
    ++n;
  };
  for (uint64_t i = 0; i < n_windows; ++i) {
    const uint32_t w = words[i];
    if (STRL_RES_COUNT(w) == 0) continue;                       // :75
    const uint64_t start = i * step, stop = std::min<uint64_t>(start + window, n_bases);
    const uint32_t unit_bits = w & 0x7fffu;                     // code + k identify the unit string
    if (!have || unit_bits != (l_word & 0x7fffu) || start > l_stop + slack) {   // :78
      flush();
      have = true; l_start = start; l_stop = stop; l_word = w;
    } else l_stop = stop;
  }
  flush();
  *n_out = n;
  if (err) return err;
  if (n > cap && out) { set_error("region buffer too small: %llu needed", (unsigned long long)n); return STRL_ERR_CAPACITY; }
  return STRL_OK;
}
