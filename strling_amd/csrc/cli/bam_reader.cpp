// bam_reader.cpp -- see bam_reader.h.  BGZF = concatenated gzip members with a BC extra field carrying the
// compressed block size (SAM spec 4.1); each is inflated into a <= 64 KiB buffer (fast_inflate.cpp, zlib behind it).
#include "bam_reader.h"
#include "cram_reader.h"
#include "fast_inflate.h"
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

std::string g_cram_fasta;

// One BGZF block: cli/fast_inflate.cpp first, zlib when it declines (zlib's verdict on a bad stream is the one reported).
// c[clen .. clen + 8) must be readable: it is the block's CRC32 + ISIZE trailer.  STRL_INFLATE=zlib forces zlib.
static bool inflate_block(const uint8_t *c, uint32_t clen, uint8_t *out, uint32_t isize) {
  static const bool zlib_only = getenv("STRL_INFLATE") && !strcmp(getenv("STRL_INFLATE"), "zlib");
  if (!zlib_only && fast_inflate(c, clen, out, isize) == 0) return true;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<uint8_t *>(c); zs.avail_in = clen;
  zs.next_out = out; zs.avail_out = isize;
  const int rc = inflate(&zs, Z_FINISH);
  const bool ok = rc == Z_STREAM_END && zs.total_out == isize;
  inflateEnd(&zs);
  return ok;
}

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
    if (!inflate_block(cbuf_.data(), clen, ubuf_.data(), isz)) { err = "BGZF inflate failed"; return false; }
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
  cram_.reset();
  if (CramFile::is_cram(path)) {
    cram_.reset(new CramFile());
    if (!cram_->open(path, g_cram_fasta, 1, err)) { cram_.reset(); return false; }
    path_ = path;
    text_ = cram_->header_text();
    targets_ = cram_->targets();
    return true;
  }
  f_ = fopen(path.c_str(), "rb");
  if (!f_) { err = "couldn't open bam"; return false; }   // extract.nim:276
  path_ = path;
  eof_ = false; next_block_ = 0; upos_ = 0; ubuf_.clear();
  char magic[4];
  int32_t l_text = 0, n_ref = 0;
  if (!get(magic, 4, err) || memcmp(magic, "BAM\1", 4) != 0) { err = "not a BAM file"; return false; }
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

bool BamReader::open_like(const BamReader &o, std::string &err) {
  close();
  if (o.cram_) {                       // a reader of its own on the same CRAM (the index is parsed again: small)
    cram_.reset(new CramFile());
    if (!cram_->open(o.path_, g_cram_fasta, 1, err, o.cram_->ref_cache()) || !cram_->load_index(err)) { cram_.reset(); return false; }     // (one reference cache for all of them)
    path_ = o.path_; text_ = o.text_; targets_ = o.targets_;
    lin_.assign(1, {});
    return true;
  }
  f_ = fopen(o.path_.c_str(), "rb");
  if (!f_) { err = "couldn't open bam"; return false; }
  path_ = o.path_;
  eof_ = false; next_block_ = 0; upos_ = 0; ubuf_.clear();
  text_ = o.text_; targets_ = o.targets_; lin_ = o.lin_; ref_beg_ = o.ref_beg_;
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
  if (cram_) {
    if (!cram_->load_index(err)) return false;
    lin_.assign(1, {});                // has_index()
    return true;
  }
  FILE *f = fopen((bam_path + ".bai").c_str(), "rb");
  if (!f && bam_path.size() > 4) f = fopen((bam_path.substr(0, bam_path.size() - 4) + ".bai").c_str(), "rb");
  if (!f) { err = "no .bai index next to " + bam_path; return false; }
  auto rd = [&](void *p, size_t n) { return fread(p, 1, n, f) == n; };
  fseeko(f, 0, SEEK_END);
  const uint64_t f_size = (uint64_t)std::max<off_t>(ftello(f), 0);
  fseeko(f, 0, SEEK_SET);
  auto left = [&]() -> uint64_t { const off_t at = ftello(f); return at < 0 || (uint64_t)at > f_size ? 0 : f_size - (uint64_t)at; };
  char magic[4];
  int32_t n_ref = 0;
  // (counts the file cannot hold -- a reference takes 8 bytes at least, an interval 8 -- are a corrupt index, not allocations)
  bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0 && (uint64_t)n_ref * 8 <= left();
  if (!ok) n_ref = 0;
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
    ok = ok && rd(&n_intv, 4) && n_intv >= 0 && (uint64_t)n_intv * 8 <= left();
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
  if (cram_) return cram_->read_region(b, tid, beg, end, err);
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

bool BamReader::region_span(int32_t tid, int64_t beg, int64_t end, uint64_t &c_beg, uint32_t &in_block, uint64_t &c_hint) const {
  if (cram_ || tid < 0 || (size_t)tid >= ref_beg_.size() || ref_beg_[(size_t)tid] == 0 || end <= beg) return false;
  const std::vector<uint64_t> &lin = lin_[(size_t)tid];
  uint64_t off = 0;                                    // as read_region
  int64_t w = beg >> 14;
  if (w >= (int64_t)lin.size()) w = (int64_t)lin.size() - 1;
  for (; w >= 0 && off == 0; --w) off = lin[(size_t)w];
  if (off == 0) off = ref_beg_[(size_t)tid];
  uint64_t hint = 0;
  for (size_t v = (size_t)((end - 1) >> 14) + 1; v < lin.size() && hint == 0; ++v) hint = lin[v];
  if (hint == 0 || (hint >> 16) < (off >> 16)) return false;
  c_beg = off >> 16;
  in_block = (uint32_t)(off & 0xffff);
  c_hint = hint >> 16;
  return true;
}

// one record (block_size bytes at p, behind its length field) into the batch
static bool append_record(RecordBatch &b, const uint8_t *p, int32_t bs, std::string &err) {
  int32_t refID, pos, l_seq, next_ref, next_pos, tlen;
  uint8_t l_read_name, mapq;
  uint16_t n_cigar, flag;
  memcpy(&refID, p, 4); memcpy(&pos, p + 4, 4);
  l_read_name = p[8]; mapq = p[9];
  memcpy(&n_cigar, p + 12, 2); memcpy(&flag, p + 14, 2); memcpy(&l_seq, p + 16, 4);
  memcpy(&next_ref, p + 20, 4); memcpy(&next_pos, p + 24, 4); memcpy(&tlen, p + 28, 4);
  const size_t need = 32 + (size_t)l_read_name + 4u * n_cigar + (size_t)(l_seq + 1) / 2;
  if (l_seq < 0 || need > (size_t)bs) { err = "corrupt BAM record"; return false; }
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
  return true;
}

int64_t BamReader::append_records(RecordBatch &b, const uint8_t *p, size_t n, std::string &err) {
  int64_t k = 0;
  size_t at = 0;
  while (at < n) {
    int32_t bs = 0;
    if (at + 4 > n) { err = "truncated BAM record"; return -1; }
    memcpy(&bs, p + at, 4);
    if (bs < 32 || at + 4 + (size_t)bs > n) { err = "corrupt BAM record"; return -1; }
    if (!append_record(b, p + at + 4, bs, err)) return -1;
    at += 4 + (size_t)bs;
    ++k;
  }
  return k;
}

int64_t BamReader::read_until(RecordBatch &b, int64_t max_records, int32_t stop_tid, int32_t stop_pos, std::string &err) {
  if (cram_) {
    if (stop_tid != INT32_MIN) { err = "read_until on a CRAM"; return -1; }
    return cram_->read(b, max_records, err);
  }
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
    int32_t refID, pos;
    memcpy(&refID, p, 4); memcpy(&pos, p + 4, 4);
    if (stop_tid != INT32_MIN && (refID != stop_tid || pos >= stop_pos)) break;
    if (!append_record(b, p, bs, err)) return -1;
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
  { std::lock_guard<std::mutex> lk(w_mu_); w_stop_ = true; }
  w_cv_.notify_all();
  if (loading_ && load_.valid()) (void)load_.get();      // a load in flight sees w_stop_ and returns
  loading_ = false;
  if (walker_.joinable()) walker_.join();
  w_stop_ = false; w_state_ = 0; wblks_.clear(); w_taken_ = 0; w_err_.clear();
  if (map_) munmap(const_cast<uint8_t *>(map_), map_len_);
  map_ = nullptr; map_len_ = 0;
  delete pool_;
  pool_ = nullptr;
  delete pool2_;
  pool2_ = nullptr;
}

bool BamStream::open(const std::string &path, int threads, std::string &err) {
  close();
  cram_.reset();
  if (CramFile::is_cram(path)) {       // extract.nim:278-279: the same open() takes a CRAM (with the FASTA it was written against)
    cram_.reset(new CramFile());
    if (!cram_->open(path, g_cram_fasta, threads, err)) { cram_.reset(); return false; }
    text_ = cram_->header_text();
    targets_ = cram_->targets();
    return true;
  }
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
  pool2_ = new ThreadPool(std::max(1, threads / 2));
  w_end_ = cpos_;
  walker_ = std::thread([this] {
    size_t pos = w_end_;
    std::vector<WBlk> local;
    int state = 0;
    std::string werr;
    auto publish = [&]() -> bool {      // hand over what was found; false = asked to stop
      std::unique_lock<std::mutex> lk(w_mu_);
      wblks_.insert(wblks_.end(), local.begin(), local.end());
      local.clear();
      w_end_ = pos; w_state_ = state; w_err_ = werr;
      w_cv_.notify_all();
      // stay at most ~64 Ki blocks (about a gigabyte of BAM) ahead of the decoder
      w_cv_.wait(lk, [&] { return w_stop_ || state != 0 || wblks_.size() - w_taken_ < 65536; });
      return !w_stop_;
    };
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
    size_t mapped = (pos / 4096) * 4096;
    while (state == 0) {
      if (pos >= map_len_) { state = 1; break; }
      if (pos + ((size_t)16 << 20) > mapped && mapped < map_len_) {
        // Map the file ahead of the walk in 64 MB pieces with ONE call each: otherwise every inflate thread takes page
        // faults on the shared address space (a quarter of a million for a 1 GB file) and the threads serialise in the kernel.
        const size_t len = std::min<size_t>((size_t)64 << 20, map_len_ - mapped);
        if (madvise(const_cast<uint8_t *>(map_) + mapped, len, MADV_POPULATE_READ) != 0) (void)madvise(const_cast<uint8_t *>(map_) + mapped, len, MADV_WILLNEED);
        mapped += len;
      }
      if (pos + 18 > map_len_) { state = 2; werr = "truncated BGZF header"; break; }
      const uint8_t *h = map_ + pos;
      if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { state = 2; werr = "not a BGZF block"; break; }
      const uint32_t xlen = h[10] | (h[11] << 8);
      if (pos + 12 + xlen > map_len_) { state = 2; werr = "truncated BGZF header"; break; }
      uint32_t bsize = 0;
      bool bad_extra = false;
      for (uint32_t o = 0; o + 4 <= xlen;) {
        const uint8_t *x = h + 12 + o;
        const uint32_t sl = x[2] | (x[3] << 8);
        if (o + 4 + sl > xlen) { bad_extra = true; break; }
        if (x[0] == 'B' && x[1] == 'C' && sl == 2) bsize = (x[4] | (x[5] << 8)) + 1u;
        o += 4 + sl;
      }
      if (bad_extra) { state = 2; werr = "malformed BGZF extra field"; break; }
      if (!bsize || bsize < 12 + xlen + 8) { state = 2; werr = "BGZF block without BC field"; break; }
      if (pos + bsize > map_len_) { state = 2; werr = "truncated BGZF block"; break; }
      const uint8_t *f = h + bsize - 4;
      const uint32_t isz = f[0] | (f[1] << 8) | (f[2] << 16) | ((uint32_t)f[3] << 24);
      pos += bsize;
      if (isz) local.push_back(WBlk{(size_t)(h + 12 + xlen - map_), bsize - 12 - xlen - 8, isz, pos});
      if (local.size() >= 512 && !publish()) return;
    }
    (void)publish();
  });
  eof_ = false; rec_next_ = 0; recs_.clear(); u_.clear(); prev_.clear(); have_prev_ = false;
  cu_.clear(); crecs_.clear(); ceof_ = false; loading_ = false;
  return true;
}

// Next superchunk: [leftover bytes of the previous one][blocks inflated in parallel] -> record table
bool BamStream::load_chunk(std::string &err) {
  struct Blk { const uint8_t *c; uint32_t clen, isize; size_t out; };
  static const char *env_blocks = getenv("STRL_CHUNK_BLOCKS");   // tests: tiny superchunks to exercise the carry path
  const size_t max_blocks = env_blocks && atoi(env_blocks) > 0 ? (size_t)atoi(env_blocks) : 64 * (size_t)pool_->size() + 64, max_bytes = (size_t)256 << 20;
  std::vector<Blk> blks;
  const double t0 = DecodeClock::now();
  // bytes behind the last complete record of the previous superchunk
  size_t carry = 0;
  if (have_prev_) {                  // stashed at the end of the previous load (u_ itself has gone to the parser since)
    carry = prev_.size();
    skip_ = 0;
  }
  size_t total = carry;
  {
    // block descriptors from the walker thread: as many as this superchunk takes, waiting only if the walker is behind
    std::unique_lock<std::mutex> lk(w_mu_);
    for (;;) {
      while (w_taken_ < wblks_.size() && blks.size() < max_blocks && total < max_bytes) {
        const WBlk &w = wblks_[w_taken_++];
        blks.push_back(Blk{map_ + w.c_off, w.clen, w.isize, total});
        total += w.isize;
        cpos_ = w.next;
      }
      if (blks.size() >= max_blocks || total >= max_bytes) break;
      if (w_state_ == 2) {
        // blocks before the damaged one are decoded first, like the sequential reader would; the error comes with the next call
        if (blks.empty()) { err = w_err_; return false; }
        break;
      }
      if (w_state_ == 1) { cpos_ = map_len_; break; }
      if (w_stop_) { err = "reader closed"; return false; }
      w_cv_.wait(lk);
    }
    if (w_taken_ > 32768) { wblks_.erase(wblks_.begin(), wblks_.begin() + (long)w_taken_); w_taken_ = 0; }
    if (w_state_ == 1 && w_taken_ == wblks_.size()) cpos_ = map_len_;
  }
  w_cv_.notify_all();
  if (cpos_ >= map_len_) eof_ = true;
  const size_t u_cap_before = u_.capacity();
  u_.resize(total);
  if (u_.capacity() != u_cap_before) (void)madvise(u_.data(), u_.capacity(), MADV_HUGEPAGE);   // fewer, larger first-touch faults
  if (carry) memcpy(u_.data(), prev_.data(), carry);
  std::atomic<bool> bad{false};
  const double t1 = DecodeClock::now();
  // The blocks are split into contiguous groups, one task each.  A task inflates its blocks one after the other and walks
  // the block_size chain of the records that start in them while the data is still in its core's cache.  Nothing waits for
  // anything: a group other than the first does not know where its first record starts, so it GUESSES -- the first offset
  // from which a chain of plausible BAM records runs (sizes, refIDs, name length and terminator, cigar / sequence lengths
  // that fit block_size) -- and the guesses are verified afterwards: the exact chain of the groups before it must arrive
  // exactly at a group's guessed start, or that group is walked again from where the chain really arrives.  (A chain of
  // per-block walks that wait for one another, the previous design, collapsed beyond ~32 threads: one descheduled
  // thread stalled all the spinning ones.)
  const size_t nb = blks.size(), total_n = u_.size();
  const int32_t n_ref = (int32_t)targets_.size();
  const size_t G = std::max<size_t>(1, std::min<size_t>(nb, (size_t)pool_->size() * 2));
  struct Group { size_t b0, b1; int64_t start = -1, end = -1; bool ok = true; std::vector<RecMeta> recs, junction; };
  std::vector<Group> groups(G);
  for (size_t g = 0; g < G; ++g) { groups[g].b0 = nb * g / G; groups[g].b1 = nb * (g + 1) / G; }
  const uint8_t *U = u_.data();
  // one record header at q (all 36 bytes readable): fields + plausibility
  auto header = [&](size_t q, RecMeta &m, uint32_t &bs) -> bool {
    memcpy(&bs, U + q, 4);
    const uint8_t *r = U + q + 4;
    m.off = q;
    m.l_qname = r[8];
    memcpy(&m.n_cigar, r + 12, 2);
    memcpy(&m.l_seq, r + 16, 4);
    return !(bs < 32 || m.l_seq < 0 || 32 + (size_t)m.l_qname + 4u * m.n_cigar + (size_t)(m.l_seq + 1) / 2 > bs);
  };
  auto plausible = [&](size_t q, size_t avail, size_t &next) -> int {   // 1 plausible record, 0 not a record, -1 cannot tell (too few bytes)
    if (q + 36 > avail) return -1;
    RecMeta m;
    uint32_t bs;
    if (!header(q, m, bs) || bs > (1u << 26)) return 0;
    int32_t ref, pos, nref, npos;
    memcpy(&ref, U + q + 4, 4); memcpy(&pos, U + q + 8, 4); memcpy(&nref, U + q + 24, 4); memcpy(&npos, U + q + 28, 4);
    if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos < -1 || npos < -1 || m.l_qname < 1) return 0;
    if ((size_t)m.l_seq + (size_t)(m.l_seq + 1) / 2 + 32 + m.l_qname + 4u * m.n_cigar > bs) return 0;   // the quality string has to fit as well
    const size_t name_end = q + 36 + m.l_qname - 1;
    if (name_end < avail && U[name_end] != 0) return 0;
    for (size_t c = q + 36; c < name_end && c < avail; ++c) if (U[c] < 33 || U[c] > 126) return 0;
    next = q + 4 + (size_t)bs;
    return 1;
  };
  auto guess_start = [&](size_t from, size_t avail) -> int64_t {
    for (size_t o = from; o + 36 <= avail && o < from + (1u << 20); ++o) {
      size_t c = o, nx = 0;
      int n_ok = 0, verdict = 1;
      while (n_ok < 8) {
        const int r = plausible(c, avail, nx);
        if (r == 0) { verdict = 0; break; }
        if (r < 0) { verdict = n_ok >= 3 ? 1 : -1; break; }
        ++n_ok;
        c = nx;
      }
      if (verdict == 1) return (int64_t)o;
      if (verdict < 0) return -1;          // ran out of inflated bytes before the chain was long enough: cannot tell yet
    }
    return -1;
  };
  auto walk = [&](Group &gr, size_t &q, size_t limit) {   // records whose 36-byte header lies inside [.., limit)
    while (q + 36 <= limit) {
      RecMeta m;
      uint32_t bs;
      if (!header(q, m, bs)) { gr.ok = false; return; }
      gr.recs.push_back(m);
      q += 4 + (size_t)bs;
    }
  };
  pool_->parallel_for(G, [&](size_t g) {
    Group &gr = groups[g];
    if (gr.b0 == gr.b1) return;
    const size_t g_begin = blks[gr.b0].out, g_end = blks[gr.b1 - 1].out + blks[gr.b1 - 1].isize;
    gr.recs.reserve((g_end - g_begin) / 160 + 8);
    size_t q = 0;
    bool have = false;
    if (g == 0) { q = skip_; have = true; gr.start = (int64_t)skip_; }
    for (size_t k = gr.b0; k < gr.b1 && !bad; ++k) {
      if (!inflate_block(blks[k].c, blks[k].clen, u_.data() + blks[k].out, blks[k].isize)) { bad = true; break; }
      const size_t avail = blks[k].out + blks[k].isize;
      if (!have && gr.ok && (avail - g_begin >= 8192 || k + 1 == gr.b1)) {
        const int64_t s0 = guess_start(g_begin, avail);
        if (s0 >= 0) { q = (size_t)s0; gr.start = s0; have = true; }
      }
      if (have && gr.ok) walk(gr, q, avail);
    }
    gr.end = have && gr.ok ? (int64_t)q : -1;
  });
  if (bad) { err = "BGZF inflate failed"; return false; }
  const double t2 = DecodeClock::now();
  // ---- stitch: follow the exact chain through the junctions, verify every guess, walk again what was guessed wrong ----
  size_t cur = 0;
  for (size_t g = 0; g < G; ++g) {
    Group &gr = groups[g];
    if (gr.b0 == gr.b1) continue;
    const size_t g_end = blks[gr.b1 - 1].out + blks[gr.b1 - 1].isize;
    if (g == 0) {
      if (!gr.ok) { err = "corrupt BAM record"; return false; }
      cur = (size_t)gr.end;
      continue;
    }
    const size_t g_begin = blks[gr.b0].out;
    Group &prev = groups[g - 1];
    // records that start before this group's data but whose header reaches into it (the previous walk could not read them)
    while (cur < g_begin && cur + 36 <= total_n) {
      RecMeta m;
      uint32_t bs;
      if (!header(cur, m, bs)) { err = "corrupt BAM record"; return false; }
      prev.junction.push_back(m);
      cur += 4 + (size_t)bs;
    }
    if (cur < g_begin) {                                    // the last record of the superchunk is not complete yet:
      for (size_t h = g; h < G; ++h) groups[h].recs.clear();  // nothing that was guessed behind it counts
      break;
    }
    if (gr.ok && gr.start == (int64_t)cur) { cur = (size_t)gr.end; continue; }
    gr.recs.clear();                                        // guessed wrong (or not at all): the chain arrives at `cur`
    gr.ok = true;
    gr.start = (int64_t)cur;
    size_t q = cur;
    walk(gr, q, g_end);
    if (!gr.ok) { err = "corrupt BAM record"; return false; }
    gr.end = (int64_t)q;
    cur = q;
  }
  // record table of the superchunk: the groups' lists in order, minus a last record that is not complete yet
  recs_.clear();
  {
    std::vector<size_t> at(2 * G + 1, 0);
    for (size_t g = 0; g < G; ++g) { at[2 * g + 1] = at[2 * g] + groups[g].recs.size(); at[2 * g + 2] = at[2 * g + 1] + groups[g].junction.size(); }
    recs_.resize(at[2 * G]);
    pool_->parallel_for(G, [&](size_t g) {
      if (!groups[g].recs.empty()) memcpy(recs_.data() + at[2 * g], groups[g].recs.data(), groups[g].recs.size() * sizeof(RecMeta));
      if (!groups[g].junction.empty()) memcpy(recs_.data() + at[2 * g + 1], groups[g].junction.data(), groups[g].junction.size() * sizeof(RecMeta));
    });
  }
  const size_t n = total_n;
  auto rec_end = [&](const RecMeta &m) { uint32_t bs; memcpy(&bs, u_.data() + m.off, 4); return (size_t)m.off + 4 + (size_t)bs; };
  if (!recs_.empty() && rec_end(recs_.back()) > n) recs_.pop_back();   // only the last one can reach past the superchunk
  const size_t p = recs_.empty() ? skip_ : rec_end(recs_.back());      // what lies behind it is carried into the next superchunk
  if (eof_ && p != n) { err = "truncated BAM record at end of file"; return false; }
  prev_.assign(u_.begin() + (long)p, u_.end());     // what the next superchunk starts with
  have_prev_ = true;
  const double t3 = DecodeClock::now();
  g_clk.walk += t1 - t0; g_clk.inflate += t2 - t1; g_clk.scan += t3 - t2;
  return true;
}

int64_t BamStream::read(RecordBatch &b, int64_t max_records, std::string &err) {
  if (cram_) return cram_->read(b, max_records, err);
  int64_t n = 0;
  while (n < max_records) {
    if (rec_next_ == crecs_.size()) {
      if (ceof_) break;
      // the next superchunk: loaded in the background while the previous one was parsed
      if (!loading_) { loading_ = true; load_ = std::async(std::launch::async, [this] { load_err_.clear(); return load_chunk(load_err_); }); }
      const bool ok = load_.get();
      loading_ = false;
      if (!ok) { err = load_err_; return -1; }
      cu_.swap(u_);
      crecs_.swap(recs_);
      ceof_ = eof_;
      rec_next_ = 0;
      if (!ceof_) { loading_ = true; load_ = std::async(std::launch::async, [this] { load_err_.clear(); return load_chunk(load_err_); }); }
      if (crecs_.empty()) { if (ceof_) break; continue; }
    }
    const size_t m = (size_t)std::min<int64_t>(max_records - n, (int64_t)(crecs_.size() - rec_next_));
    const double tp0 = DecodeClock::now();
    const RecMeta *rm = crecs_.data() + rec_next_;
    // output offsets of this part
    const size_t base = b.tid.size();
    rvec<uint32_t> cig_at(m + 1);
    rvec<uint64_t> qn_at(m + 1), seq_at(m + 1);
    size_t so;
    {
      // output offsets of every record: sums per part in parallel, a scan over the parts, offsets per part in parallel.
      // (SEQ is padded to 16 bytes per record, so a part's SEQ size does not depend on where it starts once the start is
      // 16-byte aligned, which the first record's padding guarantees.)
      const size_t P = std::min<size_t>(std::max<size_t>(m / 8192, 1), (size_t)pool2_->size() * 4);
      std::vector<uint64_t> pc(P + 1, 0), pq(P + 1, 0), ps(P + 1, 0);
      pool2_->parallel_for(P, [&](size_t part) {
        const size_t i0 = m * part / P, i1 = m * (part + 1) / P;
        uint64_t c = 0, q = 0, sq = 0;
        for (size_t i = i0; i < i1; ++i) {
          c += rm[i].n_cigar;
          q += rm[i].l_qname ? rm[i].l_qname - 1u : 0u;
          sq += ((uint64_t)(rm[i].l_seq + 1) / 2 + 15) & ~(uint64_t)15;
        }
        pc[part + 1] = c; pq[part + 1] = q; ps[part + 1] = sq;
      });
      pc[0] = b.cigar_off.back(); pq[0] = b.qname_off.back(); ps[0] = (b.seq4.size() + 15) & ~(size_t)15;
      for (size_t k = 0; k < P; ++k) { pc[k + 1] += pc[k]; pq[k + 1] += pq[k]; ps[k + 1] += ps[k]; }
      pool2_->parallel_for(P, [&](size_t part) {
        const size_t i0 = m * part / P, i1 = m * (part + 1) / P;
        uint64_t c = pc[part], q = pq[part], sq = ps[part];
        for (size_t i = i0; i < i1; ++i) {
          cig_at[i] = (uint32_t)c; qn_at[i] = q; seq_at[i] = sq;
          c += rm[i].n_cigar;
          q += rm[i].l_qname ? rm[i].l_qname - 1u : 0u;
          sq += ((uint64_t)(rm[i].l_seq + 1) / 2 + 15) & ~(uint64_t)15;
        }
      });
      cig_at[m] = (uint32_t)pc[P]; qn_at[m] = pq[P]; seq_at[m] = ps[P];
      // the last record keeps its true length (no padding behind it), like the sequential layout
      so = m ? seq_at[m - 1] + (size_t)(rm[m - 1].l_seq + 1) / 2 : b.seq4.size();
    }
    b.tid.resize(base + m); b.pos.resize(base + m); b.mtid.resize(base + m); b.mpos.resize(base + m); b.isize.resize(base + m);
    b.l_seq.resize(base + m); b.flag.resize(base + m); b.mapq.resize(base + m); b.seq_off.resize(base + m);
    b.cigar_off.resize(base + m + 1); b.qname_off.resize(base + m + 1);
    b.cigar.resize(cig_at[m]); b.qnames.resize(qn_at[m]); b.seq4.resize(so);
    const size_t parts = std::min<size_t>(m, (size_t)pool2_->size() * 4);
    pool2_->parallel_for(parts, [&](size_t part) {
      const size_t i0 = m * part / parts, i1 = m * (part + 1) / parts;
      for (size_t i = i0; i < i1; ++i) {
        const uint8_t *p = cu_.data() + rm[i].off + 4;
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
