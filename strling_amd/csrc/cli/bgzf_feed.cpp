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
  stop_ = false; state_ = 0; blks_.clear(); taken_ = 0; werr_.clear();
  if (map_) munmap(const_cast<uint8_t *>(map_), map_len_);
  map_ = nullptr; map_len_ = 0;
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
  walker_ = std::thread([this, start] {
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
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
    size_t mapped = (pos / 4096) * 4096;
    while (state == 0) {
      if (pos >= map_len_) { state = 1; break; }
      if (pos + ((size_t)32 << 20) > mapped && mapped < map_len_) {
        // page-table entries for the next piece of the mapping with ONE call (the copy threads would otherwise take a
        // fault per page on the shared address space)
        const size_t len = std::min<size_t>((size_t)128 << 20, map_len_ - mapped);
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
      if (isz > 65536u) { state = 2; werr = "BGZF block inflates to more than 64 KiB"; break; }
      const uint8_t *cf = h + bsize - 8;
      const uint32_t crc = cf[0] | (cf[1] << 8) | (cf[2] << 16) | ((uint32_t)cf[3] << 24);
      if (isz) local.push_back(Block{(size_t)(h + 12 + xlen - map_), bsize - 12 - xlen - 8, isz, crc});
      pos += bsize;
      if (local.size() >= 1024 && !publish()) return;
    }
    (void)publish();
  });
  return true;
}

int64_t BgzfFeed::next(std::vector<Block> &out, size_t max_blocks, size_t max_bytes, std::string &err) {
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
  if (taken_ > (1u << 16)) { blks_.erase(blks_.begin(), blks_.begin() + (long)taken_); taken_ = 0; }
  lk.unlock();
  cv_.notify_all();
  return (int64_t)out.size();
}

}  // namespace strl
