// bgzf_feed.cpp -- see bgzf_feed.h
#include "bgzf_feed.h"
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>

namespace strl {

BgzfFeed::~BgzfFeed() { close(); }

void BgzfFeed::close() {
  { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
  cv_.notify_all();
  if (walker_.joinable()) walker_.join();
  stop_ = false; state_ = 0; blks_.clear(); taken_ = 0; werr_.clear(); trim_ = 0;
  stop_zapper();                   // (before the mapping goes: the thread names addresses inside it)
  if (map_ && map_owned_) (void)munmap(const_cast<uint8_t *>(map_), map_len_);
  map_ = nullptr; map_owned_ = false;
  if (fd_ >= 0) ::close(fd_);
  fd_ = -1; map_len_ = 0;
}

void BgzfFeed::halt() {
  { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
  cv_.notify_all();
  if (walker_.joinable()) walker_.join();
  stop_ = false;
  std::lock_guard<std::mutex> lk(mu_);
  state_ = 1; blks_.clear(); taken_ = 0;
}

bool BgzfFeed::read_at(void *dst, size_t off, size_t n) const {
  uint8_t *d = static_cast<uint8_t *>(dst);
  while (n) {
    const ssize_t got = pread(fd_, d, n, (off_t)off);
    if (got <= 0) return false;
    d += got; off += (size_t)got; n -= (size_t)got;
  }
  return true;
}

bool BgzfFeed::copy_at(void *dst, size_t off, size_t n) const {
  if (!map_) return read_at(dst, off, n);
  if (off > map_len_ || n > map_len_ - off) return false;
  memcpy(dst, map_ + off, n);         // (a file truncated under a running extraction ends it with SIGBUS here; pread would report a short read)
  return true;
}

void BgzfFeed::done_with(size_t off, size_t n) const {
  if (!map_ || !n) return;
  const size_t page = 4096, lo = (off + page - 1) & ~(page - 1), hi = (off + n) & ~(page - 1);
  if (hi <= lo) return;
  std::lock_guard<std::mutex> lk(zmu_);
  zq_.push_back({lo, hi - lo});
  if (!zapper_.joinable())
    zapper_ = std::thread([this] {
      for (;;) {
        std::vector<std::pair<size_t, size_t>> todo;
        {
          std::unique_lock<std::mutex> lk2(zmu_);
          zcv_.wait(lk2, [&] { return zstop_ || !zq_.empty(); });
          if (zstop_) return;                 // (what is left goes with the mapping)
          todo.swap(zq_);
        }
        for (const auto &r : todo) (void)madvise(const_cast<uint8_t *>(map_) + r.first, r.second, MADV_DONTNEED);
      }
    });
  zcv_.notify_one();
}

void BgzfFeed::stop_zapper() {
  { std::lock_guard<std::mutex> lk(zmu_); zstop_ = true; }
  zcv_.notify_all();
  if (zapper_.joinable()) zapper_.join();
  std::lock_guard<std::mutex> lk(zmu_);
  zstop_ = false; zq_.clear();
}

bool BgzfFeed::open(const std::string &path, std::string &err) {
  close();
  size_t start = 0;
  {
    BamReader hdr;
    if (!hdr.open(path, err)) return false;
    text_ = hdr.header_text();
    targets_ = hdr.targets();
    const BamReader::Pos p = hdr.tell();
    start = (size_t)p.block_off;
    first_off_ = p.in_block;
  }
  fd_ = ::open(path.c_str(), O_RDONLY);
  if (fd_ < 0) { err = "couldn't open bam"; return false; }
  struct stat st;
  if (fstat(fd_, &st) != 0) { ::close(fd_); fd_ = -1; err = "couldn't stat bam"; return false; }
  map_len_ = (size_t)st.st_size;
  path_ = path;
  (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
  {
    const char *how = getenv("STRL_FEED");
    if (!(how && strcmp(how, "pread") == 0) && map_len_) {
      void *m = mmap(nullptr, map_len_, PROT_READ, MAP_SHARED, fd_, 0);
      if (m != MAP_FAILED) { map_ = static_cast<const uint8_t *>(m); map_owned_ = true; (void)madvise(m, map_len_, MADV_SEQUENTIAL); }
    }
  }
  start_ = start;
  walk_from(start, 0, 0);
  return true;
}

bool BgzfFeed::open_share(const BgzfFeed &whole, uint64_t start_coff, uint32_t first_off, uint64_t end_coff, uint32_t end_uoff, std::string &err) {
  close();
  if (whole.fd_ < 0 || start_coff >= whole.map_len_ || (end_coff && (end_coff < start_coff || end_coff >= whole.map_len_))) { err = "bad share"; return false; }
  fd_ = ::open(whole.path_.c_str(), O_RDONLY);
  if (fd_ < 0) { err = "couldn't open bam"; return false; }
  map_len_ = whole.map_len_;
  map_ = whole.map_; map_owned_ = false;      // (the mapping is the whole feed's, which outlives its shares)
  path_ = whole.path_;
  text_ = whole.text_;
  targets_ = whole.targets_;
  first_off_ = first_off;
  start_ = start_coff;
  walk_from((size_t)start_coff, end_coff, end_uoff);
  return true;
}

// BAI (SAM spec 5.2): every ioffset of the linear indices and every chunk start of the bins is the virtual offset of a record
std::vector<uint64_t> BgzfFeed::split_points(const std::string &path) {
  std::vector<uint64_t> v;
  FILE *f = fopen((path + ".bai").c_str(), "rb");
  if (!f && path.size() > 4) f = fopen((path.substr(0, path.size() - 4) + ".bai").c_str(), "rb");
  if (!f) return v;
  auto rd = [&](void *p, size_t n) { return fread(p, 1, n, f) == n; };
  fseeko(f, 0, SEEK_END);
  const uint64_t f_size = (uint64_t)std::max<off_t>(ftello(f), 0);
  fseeko(f, 0, SEEK_SET);
  auto left = [&]() -> uint64_t { const off_t at = ftello(f); return at < 0 || (uint64_t)at > f_size ? 0 : f_size - (uint64_t)at; };
  char magic[4];
  int32_t n_ref = 0;
  bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
  std::vector<uint64_t> buf;
  for (int32_t r = 0; ok && r < n_ref; ++r) {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4) && n_bin >= 0;
    for (int32_t k = 0; ok && k < n_bin; ++k) {
      uint32_t bin = 0;
      int32_t n_chunk = 0;
      ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0 && (uint64_t)n_chunk * 16 <= left();      // (a count the file cannot hold: not an allocation)
      if (!ok) break;
      buf.resize((size_t)n_chunk * 2);
      ok = n_chunk == 0 || rd(buf.data(), (size_t)n_chunk * 16);
      if (ok && bin != 37450) for (int32_t c = 0; c < n_chunk; ++c) v.push_back(buf[(size_t)c * 2]);   // 37450: the metadata pseudo-bin
    }
    int32_t n_intv = 0;
    ok = ok && rd(&n_intv, 4) && n_intv >= 0 && (uint64_t)n_intv * 8 <= left();
    if (ok && n_intv) {
      buf.resize((size_t)n_intv);
      ok = rd(buf.data(), (size_t)n_intv * 8);
      if (ok) for (uint64_t x : buf) if (x) v.push_back(x);
    }
  }
  fclose(f);
  if (!ok) { v.clear(); return v; }
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  return v;
}

bool BgzfFeed::indexed_records(const std::string &path, uint64_t &n) {
  n = 0;
  FILE *f = fopen((path + ".bai").c_str(), "rb");
  if (!f && path.size() > 4) f = fopen((path.substr(0, path.size() - 4) + ".bai").c_str(), "rb");
  if (!f) return false;
  auto rd = [&](void *p, size_t k) { return fread(p, 1, k, f) == k; };
  fseeko(f, 0, SEEK_END);
  const uint64_t f_size = (uint64_t)std::max<off_t>(ftello(f), 0);
  fseeko(f, 0, SEEK_SET);
  auto left = [&]() -> uint64_t { const off_t at = ftello(f); return at < 0 || (uint64_t)at > f_size ? 0 : f_size - (uint64_t)at; };
  char magic[4];
  int32_t n_ref = 0;
  bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
  uint64_t total = 0;
  std::vector<uint64_t> buf;
  for (int32_t r = 0; ok && r < n_ref; ++r) {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4) && n_bin >= 0;
    bool meta = false;
    for (int32_t k = 0; ok && k < n_bin; ++k) {
      uint32_t bin = 0;
      int32_t n_chunk = 0;
      ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0 && (uint64_t)n_chunk * 16 <= left();      // (a count the file cannot hold: not an allocation)
      if (!ok) break;
      buf.resize((size_t)n_chunk * 2);
      ok = n_chunk == 0 || rd(buf.data(), (size_t)n_chunk * 16);
      if (ok && bin == 37450) {
        if (n_chunk != 2) ok = false;
        else { total += buf[2] + buf[3]; meta = true; }      // chunk 1 = (n_mapped, n_unmapped)
      }
    }
    if (ok && n_bin > 0 && !meta) ok = false;                 // a reference with records and no counts: an index written without them
    int32_t n_intv = 0;
    ok = ok && rd(&n_intv, 4) && n_intv >= 0 && (uint64_t)n_intv * 8 <= left() && fseeko(f, (off_t)n_intv * 8, SEEK_CUR) == 0;
  }
  uint64_t no_coor = 0;
  ok = ok && rd(&no_coor, 8);                                 // (optional in the format; samtools writes it -- without it the count is not known)
  fclose(f);
  if (!ok) return false;
  n = total + no_coor;
  return true;
}

// the header walker: from the block at `start` to the end of the file, or (end_coff != 0) to the share's end -- the block at
// end_coff is included when end_uoff > 0 (its first end_uoff bytes are the share's; trim_ = the rest)
void BgzfFeed::walk_from(size_t start, uint64_t end_coff, uint32_t end_uoff) {
  trim_ = 0;
  walker_ = std::thread([this, start, end_coff, end_uoff] {
    size_t pos = start;
    std::vector<Block> local;
    int state = 0;
    std::string werr;
    auto publish = [&]() -> bool {
      std::unique_lock<std::mutex> lk(mu_);
      blks_.insert(blks_.end(), local.begin(), local.end());
      local.clear();
      state_ = state; werr_ = werr;
      cv_.notify_all();
      cv_.wait(lk, [&] { return stop_ || state != 0 || blks_.size() - taken_ < (1u << 17); });   // stay <= ~2 GB of BAM ahead
      return !stop_;
    };
    // h = the 18 fixed bytes of the header of the block at pos (the next block's arrive with this block's trailer: one pread)
    uint8_t h[18], tail[8 + 18];
    bool have = pos + 18 <= map_len_ && read_at(h, pos, 18);
    while (state == 0) {
      if (pos >= map_len_) { state = end_coff ? 2 : 1; if (end_coff) werr = "the index names an offset behind the last BGZF block"; break; }
      if (end_coff && pos >= end_coff) {
        if (pos > end_coff) { state = 2; werr = "the index names an offset that is not a BGZF block's"; break; }
        if (end_uoff == 0) { state = 1; break; }
      }
      if (!have) { state = 2; werr = "truncated BGZF header"; break; }
      if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { state = 2; werr = "not a BGZF block"; break; }
      const uint32_t xlen = h[10] | (h[11] << 8);
      if (pos + 12 + xlen > map_len_) { state = 2; werr = "truncated BGZF header"; break; }
      uint8_t extra_buf[256];
      std::vector<uint8_t> extra_big;
      const uint8_t *extra = h + 12;                       // the BC subfield is first in every BGZF writer: the 18 bytes hold it
      if (xlen > 6) {
        uint8_t *e = extra_buf;
        if (xlen > sizeof extra_buf) { extra_big.resize(xlen); e = extra_big.data(); }
        if (!read_at(e, pos + 12, xlen)) { state = 2; werr = "truncated BGZF header"; break; }
        extra = e;
      }
      uint32_t bsize = 0;
      bool bad_extra = false;
      for (uint32_t o = 0; o + 4 <= xlen;) {
        const uint8_t *x = extra + o;
        const uint32_t sl = x[2] | (x[3] << 8);
        if (o + 4 + sl > xlen) { bad_extra = true; break; }
        if (x[0] == 'B' && x[1] == 'C' && sl == 2) bsize = (x[4] | (x[5] << 8)) + 1u;
        o += 4 + sl;
      }
      if (bad_extra) { state = 2; werr = "malformed BGZF extra field"; break; }
      if (!bsize || bsize < 12 + xlen + 8) { state = 2; werr = "BGZF block without BC field"; break; }
      if (pos + bsize > map_len_) { state = 2; werr = "truncated BGZF block"; break; }
      const size_t next = pos + bsize;
      const size_t want = std::min<size_t>(8 + 18, map_len_ - (next - 8));
      if (!read_at(tail, next - 8, want)) { state = 2; werr = "truncated BGZF block"; break; }
      const uint32_t crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
      const uint32_t isz = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
      if (isz > 65536u) { state = 2; werr = "BGZF block inflates to more than 64 KiB"; break; }
      if (end_coff && pos == end_coff) {               // the share's last block: the record boundary lies inside it
        if (end_uoff > isz) { state = 2; werr = "the index names an offset behind its block's end"; break; }
        if (isz) local.push_back(Block{pos + 12 + xlen, bsize - 12 - xlen - 8, isz, crc});
        trim_ = isz - end_uoff;                        // (published with state_ under the lock below)
        state = 1;
        break;
      }
      if (isz) local.push_back(Block{pos + 12 + xlen, bsize - 12 - xlen - 8, isz, crc});
      have = want == 8 + 18;
      if (have) memcpy(h, tail + 8, 18);
      pos = next;
      if (local.size() >= 1024 && !publish()) return;
    }
    (void)publish();
  });
}

int64_t BgzfFeed::next(std::vector<Block> &out, size_t max_blocks, size_t max_bytes, std::string &err, bool *last) {
  out.clear();
  std::unique_lock<std::mutex> lk(mu_);
  for (;;) {
    while (taken_ < blks_.size() && out.size() < max_blocks) {
      const Block &b = blks_[taken_];
      if (!out.empty() && b.c_off + b.clen - out.front().c_off > max_bytes) break;
      out.push_back(b);
      ++taken_;
    }
    if (out.size() >= max_blocks) break;
    if (taken_ < blks_.size()) break;              // the byte limit stopped the run
    if (state_ == 2) {
      if (out.empty()) { err = werr_; return -1; }  // blocks before the damaged one are delivered first
      break;
    }
    if (state_ == 1) break;
    cv_.wait(lk);
  }
  if (last) {      // is anything behind this run?  (waits for the walker's next batch or its end)
    while (taken_ == blks_.size() && state_ == 0) { cv_.notify_all(); cv_.wait(lk); }
    *last = taken_ == blks_.size();
  }
  if (taken_ > (1u << 16)) { blks_.erase(blks_.begin(), blks_.begin() + (long)taken_); taken_ = 0; }
  lk.unlock();
  cv_.notify_all();
  return (int64_t)out.size();
}

}  // namespace strl
