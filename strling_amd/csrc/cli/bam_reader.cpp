// bam_reader.cpp -- see bam_reader.h.  BGZF = concatenated gzip members with a BC extra field carrying the
// compressed block size (SAM spec 4.1); each is inflated with raw zlib into a <= 64 KiB buffer.
#include "bam_reader.h"
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace strl {

void RecordBatch::clear() {
  tid.clear(); pos.clear(); mtid.clear(); mpos.clear(); isize.clear(); l_seq.clear(); flag.clear(); mapq.clear();
  cigar_off.assign(1, 0); cigar.clear(); seq_off.clear(); qname_off.assign(1, 0); seq4.clear(); qnames.clear();
}

strl_records RecordBatch::view() {
  seq4.resize(seq4.size() + 32, 0);   // slack the device may over-read (call view() once per filled batch)
  strl_records r{};
  r.n = (int64_t)tid.size();
  r.tid = tid.data(); r.pos = pos.data(); r.mtid = mtid.data(); r.mpos = mpos.data(); r.flag = flag.data(); r.mapq = mapq.data();
  r.cigar_off = cigar_off.data(); r.cigar = cigar.data(); r.seq_off = seq_off.data(); r.l_seq = l_seq.data();
  r.seq4 = seq4.data(); r.qname_off = qname_off.data(); r.qnames = qnames.data();
  return r;
}

BamReader::~BamReader() { close(); }
void BamReader::close() {
  if (f_) fclose(f_);
  f_ = nullptr;
}

bool BamReader::fill(std::string &err) {
  ubuf_.clear();
  upos_ = 0;
  for (;;) {  // skip empty blocks (e.g. the EOF marker in the middle of concatenated files)
    uint8_t h[18];
    block_start_ = next_block_;
    const size_t got = fread(h, 1, 18, f_);
    if (got == 0) { eof_ = true; return true; }
    if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
    const uint32_t xlen = h[10] | (h[11] << 8);
    // the BC subfield is first in every BGZF writer; tolerate others by scanning
    uint32_t bsize = 0;
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f_) != xlen - 6) { err = "truncated BGZF header"; return false; }
    for (uint32_t o = 0; o + 4 <= xlen;) {
      const uint32_t sl = extra[o + 2] | (extra[o + 3] << 8);
      if (o + 4 + sl > xlen) { err = "malformed BGZF extra field"; return false; }
      if (extra[o] == 'B' && extra[o + 1] == 'C' && sl == 2) bsize = (extra[o + 4] | (extra[o + 5] << 8)) + 1u;
      o += 4 + sl;
    }
    if (!bsize || bsize < 12 + xlen + 8) { err = "BGZF block without BC field"; return false; }   // (same guard as BamStream::load_chunk)
    const uint32_t clen = bsize - 12 - xlen - 8;
    cbuf_.resize(clen + 8);
    if (fread(cbuf_.data(), 1, clen + 8, f_) != clen + 8) { err = "truncated BGZF block"; return false; }
    next_block_ = block_start_ + bsize;
    const uint32_t isz = cbuf_[clen + 4] | (cbuf_[clen + 5] << 8) | (cbuf_[clen + 6] << 16) | ((uint32_t)cbuf_[clen + 7] << 24);
    if (isz == 0) continue;
    ubuf_.resize(isz);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; }
    zs.next_in = cbuf_.data(); zs.avail_in = clen; zs.next_out = ubuf_.data(); zs.avail_out = isz;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.total_out != isz) { err = "BGZF inflate failed"; return false; }
    return true;
  }
}

bool BamReader::get(void *dst, size_t n, std::string &err) {
  uint8_t *d = static_cast<uint8_t *>(dst);
  while (n) {
    if (upos_ == ubuf_.size()) {
      if (!fill(err)) return false;
      if (eof_) { if (err.empty()) err = "EOF"; return false; }
    }
    const size_t k = std::min(n, ubuf_.size() - upos_);
    memcpy(d, ubuf_.data() + upos_, k);
    d += k; upos_ += k; n -= k;
  }
  return true;
}

bool BamReader::open(const std::string &path, std::string &err) {
  close();
  f_ = fopen(path.c_str(), "rb");
  if (!f_) { err = "couldn't open bam"; return false; }   // extract.nim:276
  eof_ = false; next_block_ = 0; upos_ = 0; ubuf_.clear();
  char magic[4];
  int32_t l_text = 0, n_ref = 0;
  if (!get(magic, 4, err) || memcmp(magic, "BAM\1", 4) != 0) { err = "not a BAM file (CRAM is not supported by this build)"; return false; }
  if (!get(&l_text, 4, err) || l_text < 0) return false;
  text_.resize((size_t)l_text);
  if (l_text && !get(&text_[0], (size_t)l_text, err)) return false;
  while (!text_.empty() && text_.back() == '\0') text_.pop_back();
  if (!get(&n_ref, 4, err) || n_ref < 0) return false;
  targets_.clear();
  for (int32_t i = 0; i < n_ref; ++i) {
    int32_t l_name = 0, l_ref = 0;
    if (!get(&l_name, 4, err) || l_name <= 0) { err = "bad reference name"; return false; }
    std::string nm((size_t)l_name, '\0');
    if (!get(&nm[0], (size_t)l_name, err) || !get(&l_ref, 4, err)) return false;
    nm.pop_back();
    targets_.push_back(BamTarget{nm, (uint32_t)l_ref});
  }
  return true;
}

bool BamReader::seek(Pos p, std::string &err) {
  if (fseeko(f_, (off_t)p.block_off, SEEK_SET) != 0) { err = "seek failed"; return false; }
  next_block_ = p.block_off;
  eof_ = false;
  if (!fill(err)) return false;
  if (p.in_block > ubuf_.size()) { err = "bad virtual offset"; return false; }
  upos_ = p.in_block;
  return true;
}

// BAI (SAM spec 5.2): magic, n_ref, per reference { n_bin, { bin, n_chunk, { beg, end } }, n_intv, ioffset[] }
bool BamReader::load_index(const std::string &bam_path, std::string &err) {
  FILE *f = fopen((bam_path + ".bai").c_str(), "rb");
  if (!f && bam_path.size() > 4) f = fopen((bam_path.substr(0, bam_path.size() - 4) + ".bai").c_str(), "rb");
  if (!f) { err = "no .bai index next to " + bam_path; return false; }
  auto rd = [&](void *p, size_t n) { return fread(p, 1, n, f) == n; };
  char magic[4];
  int32_t n_ref = 0;
  bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
  lin_.assign((size_t)std::max(n_ref, 0), {});
  ref_beg_.assign((size_t)std::max(n_ref, 0), 0);
  for (int32_t r = 0; ok && r < n_ref; ++r) {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4);
    uint64_t first = 0;
    for (int32_t k = 0; ok && k < n_bin; ++k) {
      uint32_t bin = 0;
      int32_t n_chunk = 0;
      ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
      for (int32_t c = 0; ok && c < n_chunk; ++c) {
        uint64_t be[2];
        ok = rd(be, 16);
        if (ok && bin != 37450 && (first == 0 || be[0] < first)) first = be[0];   // 37450: the metadata pseudo-bin
      }
    }
    int32_t n_intv = 0;
    ok = ok && rd(&n_intv, 4) && n_intv >= 0;
    if (ok) {
      lin_[(size_t)r].resize((size_t)n_intv);
      ok = n_intv == 0 || rd(lin_[(size_t)r].data(), (size_t)n_intv * 8);
      ref_beg_[(size_t)r] = first;
    }
  }
  fclose(f);
  if (!ok) { lin_.clear(); err = "corrupt .bai index"; return false; }
  if (lin_.empty()) lin_.push_back({});   // has_index() for a BAM without references
  return true;
}

int64_t BamReader::read_region(RecordBatch &b, int32_t tid, int64_t beg, int64_t end, std::string &err) {
  if (tid < 0 || (size_t)tid >= ref_beg_.size() || ref_beg_[(size_t)tid] == 0 || end <= beg) return 0;
  const std::vector<uint64_t> &lin = lin_[(size_t)tid];
  // smallest offset of a record overlapping the 16 KiB window of `beg`; empty windows (0) fall back to the nearest
  // earlier filled one, windows beyond the last record-bearing one hold nothing that can overlap
  uint64_t off = 0;
  int64_t w = beg >> 14;
  if (w >= (int64_t)lin.size()) w = (int64_t)lin.size() - 1;
  for (; w >= 0 && off == 0; --w) off = lin[(size_t)w];
  if (off == 0) off = ref_beg_[(size_t)tid];
  if (!seek(Pos{off >> 16, (uint32_t)(off & 0xffff)}, err)) return -1;
  return read_until(b, INT64_MAX, tid, (int32_t)std::min<int64_t>(end, INT32_MAX), err);
}

int64_t BamReader::read_until(RecordBatch &b, int64_t max_records, int32_t stop_tid, int32_t stop_pos, std::string &err) {
  int64_t n = 0;
  std::vector<uint8_t> rec;
  while (n < max_records) {
    if (upos_ == ubuf_.size()) {
      if (!fill(err)) return -1;
      if (eof_) break;
    }
    int32_t bs = 0;
    if (!get(&bs, 4, err)) return -1;
    if (bs < 32) { err = "corrupt BAM record"; return -1; }
    rec.resize((size_t)bs);
    if (!get(rec.data(), (size_t)bs, err)) return -1;
    const uint8_t *p = rec.data();
    int32_t refID, pos, l_seq, next_ref, next_pos, tlen;
    uint8_t l_read_name, mapq;
    uint16_t n_cigar, flag;
    memcpy(&refID, p, 4); memcpy(&pos, p + 4, 4);
    l_read_name = p[8]; mapq = p[9];
    memcpy(&n_cigar, p + 12, 2); memcpy(&flag, p + 14, 2); memcpy(&l_seq, p + 16, 4);
    memcpy(&next_ref, p + 20, 4); memcpy(&next_pos, p + 24, 4); memcpy(&tlen, p + 28, 4);
    const size_t need = 32 + (size_t)l_read_name + 4u * n_cigar + (size_t)(l_seq + 1) / 2;
    if (l_seq < 0 || need > (size_t)bs) { err = "corrupt BAM record"; return -1; }
    if (stop_tid != INT32_MIN && (refID != stop_tid || pos >= stop_pos)) break;
    b.tid.push_back(refID); b.pos.push_back(pos); b.mtid.push_back(next_ref); b.mpos.push_back(next_pos);
    b.isize.push_back(tlen); b.l_seq.push_back(l_seq); b.flag.push_back(flag); b.mapq.push_back(mapq);
    const char *qn = reinterpret_cast<const char *>(p + 32);
    b.qnames.append(qn, l_read_name ? (size_t)l_read_name - 1 : 0);
    b.qname_off.push_back(b.qnames.size());
    const uint8_t *cg = p + 32 + l_read_name;
    for (int j = 0; j < n_cigar; ++j) { uint32_t c; memcpy(&c, cg + 4 * j, 4); b.cigar.push_back(c); }
    b.cigar_off.push_back((uint32_t)b.cigar.size());
    const size_t so = (b.seq4.size() + 15) & ~(size_t)15;
    const size_t sb = (size_t)(l_seq + 1) / 2;
    b.seq4.resize(so + sb, 0);
    memcpy(b.seq4.data() + so, cg + 4u * n_cigar, sb);
    b.seq_off.push_back(so);
    ++n;
  }
  return n;
}


// ---- ThreadPool ---------------------------------------------------------------------------------------------------------
struct ThreadPool::Impl {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  const std::function<void(size_t)> *fn = nullptr;
  size_t n = 0;
  std::atomic<size_t> next{0};
  uint64_t generation = 0;
  int active = 0;
  bool stop = false;
  void work() {
    for (;;) {
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) break;
      (*fn)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_start.wait(lk, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--active == 0) cv_done.notify_one();
      }
    }
  }
};
ThreadPool::ThreadPool(int threads) : impl_(new Impl), n_threads_(std::max(1, threads)) {
  for (int t = 1; t < n_threads_; ++t) impl_->workers.emplace_back([this] { impl_->loop(); });
}
ThreadPool::~ThreadPool() {
  { std::lock_guard<std::mutex> lk(impl_->m); impl_->stop = true; }
  impl_->cv_start.notify_all();
  for (auto &w : impl_->workers) w.join();
  delete impl_;
}
void ThreadPool::parallel_for(size_t n, const std::function<void(size_t)> &fn) {
  if (n == 0) return;
  if (n_threads_ == 1 || n == 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  {
    std::lock_guard<std::mutex> lk(impl_->m);
    impl_->fn = &fn; impl_->n = n; impl_->next.store(0); impl_->active = (int)impl_->workers.size(); ++impl_->generation;
  }
  impl_->cv_start.notify_all();
  impl_->work();
  std::unique_lock<std::mutex> lk(impl_->m);
  impl_->cv_done.wait(lk, [&] { return impl_->active == 0; });
}

// ---- BamStream ----------------------------------------------------------------------------------------------------------
namespace {
struct DecodeClock {   // STRL_DECODE_TIMING=1: where the reader's time goes, printed when the stream closes
  double walk = 0, inflate = 0, scan = 0, parse = 0;
  bool on = getenv("STRL_DECODE_TIMING") != nullptr;
  static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
} g_clk;
}  // namespace
BamStream::~BamStream() { close(); }
void BamStream::close() {
  if (g_clk.on && map_) fprintf(stderr, "[strling] decode seconds: walk+carry %.3f inflate %.3f record scan %.3f parse %.3f\n", g_clk.walk, g_clk.inflate, g_clk.scan, g_clk.parse);
  if (map_) munmap(const_cast<uint8_t *>(map_), map_len_);
  map_ = nullptr; map_len_ = 0;
  delete pool_;
  pool_ = nullptr;
}

bool BamStream::open(const std::string &path, int threads, std::string &err) {
  close();
  {   // header text + targets with the plain reader; it also tells where the first record starts
    BamReader hdr;
    if (!hdr.open(path, err)) return false;
    text_ = hdr.header_text();
    targets_ = hdr.targets();
    const BamReader::Pos p = hdr.tell();
    cpos_ = (size_t)p.block_off;
    skip_ = p.in_block;
  }
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) { err = "couldn't open bam"; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0) { ::close(fd); err = "couldn't stat bam"; return false; }
  map_len_ = (size_t)st.st_size;
  void *m = map_len_ ? mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
  ::close(fd);
  if (map_len_ && m == MAP_FAILED) { map_len_ = 0; err = "couldn't map bam"; return false; }
  map_ = static_cast<const uint8_t *>(m);
  if (map_len_) madvise(const_cast<uint8_t *>(map_), map_len_, MADV_SEQUENTIAL);
  pool_ = new ThreadPool(threads);
  eof_ = false; rec_next_ = 0; recs_.clear(); u_.clear(); prev_.clear();
  return true;
}

// Next superchunk: [leftover bytes of the previous one][blocks inflated in parallel] -> record table
bool BamStream::load_chunk(std::string &err) {
  struct Blk { const uint8_t *c; uint32_t clen, isize; size_t out; };
  static const char *env_blocks = getenv("STRL_CHUNK_BLOCKS");   // tests: tiny superchunks to exercise the carry path
  const size_t max_blocks = env_blocks && atoi(env_blocks) > 0 ? (size_t)atoi(env_blocks) : 64 * (size_t)pool_->size() + 64, max_bytes = (size_t)96 << 20;
  std::vector<Blk> blks;
  const double t0 = DecodeClock::now();
  // bytes behind the last complete record of the previous superchunk
  size_t carry = 0;
  if (!u_.empty()) {
    const size_t done = recs_.empty() ? skip_ : (size_t)(recs_.back().off + 4 + [&] { uint32_t bs; memcpy(&bs, u_.data() + recs_.back().off, 4); return bs; }());
    carry = u_.size() - done;
    prev_.assign(u_.begin() + (long)done, u_.end());
    skip_ = 0;
  }
  size_t total = carry;
  while (cpos_ < map_len_ && blks.size() < max_blocks && total < max_bytes) {
    if (cpos_ + 18 > map_len_) { err = "truncated BGZF header"; return false; }
    const uint8_t *h = map_ + cpos_;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
    const uint32_t xlen = h[10] | (h[11] << 8);
    if (cpos_ + 12 + xlen > map_len_) { err = "truncated BGZF header"; return false; }
    uint32_t bsize = 0;
    for (uint32_t o = 0; o + 4 <= xlen;) {
      const uint8_t *x = h + 12 + o;
      const uint32_t sl = x[2] | (x[3] << 8);
      if (o + 4 + sl > xlen) { err = "malformed BGZF extra field"; return false; }
      if (x[0] == 'B' && x[1] == 'C' && sl == 2) bsize = (x[4] | (x[5] << 8)) + 1u;
      o += 4 + sl;
    }
    if (!bsize || bsize < 12 + xlen + 8) { err = "BGZF block without BC field"; return false; }
    if (cpos_ + bsize > map_len_) { err = "truncated BGZF block"; return false; }
    const uint8_t *f = h + bsize - 4;
    const uint32_t isz = f[0] | (f[1] << 8) | (f[2] << 16) | ((uint32_t)f[3] << 24);
    if (isz) { blks.push_back(Blk{h + 12 + xlen, bsize - 12 - xlen - 8, isz, total}); total += isz; }
    cpos_ += bsize;
  }
  if (cpos_ >= map_len_) eof_ = true;
  u_.resize(total);
  if (carry) memcpy(u_.data(), prev_.data(), carry);
  std::atomic<bool> bad{false};
  const double t1 = DecodeClock::now();
  // Each task inflates one block and then walks the block_size chain through it -- while the data is still in that
  // core's cache -- starting where the previous block's walk stopped (start[k], published by task k - 1; tasks are
  // claimed in increasing order, so the task a walk waits for is always already running).  A record whose fixed
  // 36-byte part is not complete inside the blocks done so far is handed on to the next block.
  const size_t nb = blks.size(), total_n = u_.size();
  struct alignas(64) Link { std::atomic<int64_t> v{-1}; int64_t load(std::memory_order o = std::memory_order_seq_cst) const { return v.load(o); }
                            void store(int64_t x, std::memory_order o = std::memory_order_seq_cst) { v.store(x, o); } };
  std::vector<Link> start(nb + 1);                       // one cache line each: every link has one writer and one spinning reader
  start[0].store((int64_t)skip_, std::memory_order_release);
  std::vector<std::vector<RecMeta>> found(nb);
  pool_->parallel_for(nb, [&](size_t k) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    bool ok = inflateInit2(&zs, -15) == Z_OK;
    if (ok) {
      zs.next_in = const_cast<uint8_t *>(blks[k].c); zs.avail_in = blks[k].clen;
      zs.next_out = u_.data() + blks[k].out; zs.avail_out = blks[k].isize;
      const int rc = inflate(&zs, Z_FINISH);
      ok = rc == Z_STREAM_END && zs.total_out == blks[k].isize;
      inflateEnd(&zs);
    }
    if (!ok) bad = true;
    int64_t p;
    for (unsigned spins = 0; (p = start[k].load(std::memory_order_acquire)) < 0; ++spins) {
      if (bad) { start[k + 1].store(0, std::memory_order_release); return; }
      if (spins < 4096) __builtin_ia32_pause(); else std::this_thread::yield();
    }
    const size_t end_k = blks[k].out + blks[k].isize;
    std::vector<RecMeta> &out = found[k];
    out.reserve(blks[k].isize / 128 + 4);
    size_t q = (size_t)p;
    while (ok && q + 36 <= end_k) {
      uint32_t bs;
      memcpy(&bs, u_.data() + q, 4);
      const uint8_t *r = u_.data() + q + 4;
      RecMeta m;
      m.off = q;
      m.l_qname = r[8];
      memcpy(&m.n_cigar, r + 12, 2);
      memcpy(&m.l_seq, r + 16, 4);
      if (bs < 32 || m.l_seq < 0 || 32 + (size_t)m.l_qname + 4u * m.n_cigar + (size_t)(m.l_seq + 1) / 2 > bs) { bad = true; ok = false; break; }
      out.push_back(m);
      q += 4 + (size_t)bs;
    }
    start[k + 1].store((int64_t)q, std::memory_order_release);
  });
  if (bad) { err = "BGZF inflate failed or corrupt BAM record"; return false; }
  const double t2 = DecodeClock::now();
  // record table of the superchunk: the per-block lists in order, minus the records that are not complete yet
  recs_.clear();
  rec_next_ = 0;
  for (size_t k = 0; k < nb; ++k) recs_.insert(recs_.end(), found[k].begin(), found[k].end());
  const size_t n = total_n;
  auto rec_end = [&](const RecMeta &m) { uint32_t bs; memcpy(&bs, u_.data() + m.off, 4); return (size_t)m.off + 4 + (size_t)bs; };
  if (!recs_.empty() && rec_end(recs_.back()) > n) recs_.pop_back();   // only the last one can reach past the superchunk
  const size_t p = recs_.empty() ? skip_ : rec_end(recs_.back());      // what lies behind it is carried into the next superchunk
  if (eof_ && p != n) { err = "truncated BAM record at end of file"; return false; }
  const double t3 = DecodeClock::now();
  g_clk.walk += t1 - t0; g_clk.inflate += t2 - t1; g_clk.scan += t3 - t2;
  return true;
}

int64_t BamStream::read(RecordBatch &b, int64_t max_records, std::string &err) {
  int64_t n = 0;
  while (n < max_records) {
    if (rec_next_ == recs_.size()) {
      if (eof_) break;
      if (!load_chunk(err)) return -1;
      if (recs_.empty()) { if (eof_) break; continue; }
    }
    const size_t m = (size_t)std::min<int64_t>(max_records - n, (int64_t)(recs_.size() - rec_next_));
    const double tp0 = DecodeClock::now();
    const RecMeta *rm = recs_.data() + rec_next_;
    // output offsets of this part
    const size_t base = b.tid.size();
    std::vector<uint32_t> cig_at(m + 1);
    std::vector<uint64_t> qn_at(m + 1), seq_at(m + 1);
    cig_at[0] = b.cigar_off.back();
    qn_at[0] = b.qname_off.back();
    size_t so = b.seq4.size();
    for (size_t i = 0; i < m; ++i) {
      cig_at[i + 1] = cig_at[i] + rm[i].n_cigar;
      qn_at[i + 1] = qn_at[i] + (rm[i].l_qname ? rm[i].l_qname - 1u : 0u);
      so = (so + 15) & ~(size_t)15;
      seq_at[i] = so;
      so += (size_t)(rm[i].l_seq + 1) / 2;
    }
    seq_at[m] = so;
    b.tid.resize(base + m); b.pos.resize(base + m); b.mtid.resize(base + m); b.mpos.resize(base + m); b.isize.resize(base + m);
    b.l_seq.resize(base + m); b.flag.resize(base + m); b.mapq.resize(base + m); b.seq_off.resize(base + m);
    b.cigar_off.resize(base + m + 1); b.qname_off.resize(base + m + 1);
    b.cigar.resize(cig_at[m]); b.qnames.resize(qn_at[m]); b.seq4.resize(so);
    const size_t parts = std::min<size_t>(m, (size_t)pool_->size() * 4);
    pool_->parallel_for(parts, [&](size_t part) {
      const size_t i0 = m * part / parts, i1 = m * (part + 1) / parts;
      for (size_t i = i0; i < i1; ++i) {
        const uint8_t *p = u_.data() + rm[i].off + 4;
        const size_t o = base + i;
        memcpy(&b.tid[o], p, 4); memcpy(&b.pos[o], p + 4, 4);
        b.mapq[o] = p[9];
        memcpy(&b.flag[o], p + 14, 2);
        b.l_seq[o] = rm[i].l_seq;
        memcpy(&b.mtid[o], p + 20, 4); memcpy(&b.mpos[o], p + 24, 4); memcpy(&b.isize[o], p + 28, 4);
        const size_t ql = rm[i].l_qname ? rm[i].l_qname - 1u : 0u;
        if (ql) memcpy(&b.qnames[qn_at[i]], p + 32, ql);
        b.qname_off[o + 1] = qn_at[i + 1];
        const uint8_t *cg = p + 32 + rm[i].l_qname;
        if (rm[i].n_cigar) memcpy(&b.cigar[cig_at[i]], cg, 4u * rm[i].n_cigar);
        b.cigar_off[o + 1] = cig_at[i + 1];
        const size_t sb = (size_t)(rm[i].l_seq + 1) / 2;
        if (sb) memcpy(b.seq4.data() + seq_at[i], cg + 4u * rm[i].n_cigar, sb);
        b.seq_off[o] = seq_at[i];
      }
    });
    rec_next_ += m;
    n += (int64_t)m;
    g_clk.parse += DecodeClock::now() - tp0;
  }
  return n;
}

}  // namespace strl
