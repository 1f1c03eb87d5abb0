// bgzf_feed.h -- the host side of the device BAM front end (strl_front_*): a thread walks the BGZF block headers ahead of the
// consumer (one 26-byte pread per block: the trailer of a block and the header of the next sit side by side), and the consumer
// reads whole runs of blocks -- compressed -- straight into page-locked buffers for the GPU (pread on several threads).  The file
// is NOT mapped: a 57 GB mapping cost 2 s of page-table population in front of a cold run and 0.9 s to release at exit
// (profiles/r04/e2e_full_mmap.json).  No inflate, no record ever touched on the host (extract.nim:275-329 does both
// through htslib on one thread).
#pragma once
#include <stdint.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "bam_reader.h"

namespace strl {

class BgzfFeed {
 public:
  ~BgzfFeed();
  // header text + targets (read with the plain reader), maps the file, starts the walker at the block the first record is in
  bool open(const std::string &path, std::string &err);
  void close();
  const std::string &header_text() const { return text_; }
  const std::vector<BamTarget> &targets() const { return targets_; }
  uint64_t first_record_offset() const { return first_off_; }   // bytes into the first block's inflated data
  size_t file_bytes() const { return map_len_; }
  // n bytes of the file at offset off into dst (any thread); false on a short read
  bool read_at(void *dst, size_t off, size_t n) const;
  struct Block { size_t c_off; uint32_t clen, isize, crc; };     // DEFLATE payload at file offset c_off; CRC-32 of the inflated bytes (trailer)
  // Next run of consecutive non-empty blocks: at most max_blocks and max_bytes of file (first block's payload to the last
  // block's end).  Returns the number of blocks (0 at the end of the file), -1 on a malformed file.
  int64_t next(std::vector<Block> &out, size_t max_blocks, size_t max_bytes, std::string &err);

 private:
  int fd_ = -1;
  size_t map_len_ = 0;          // file size
  uint64_t first_off_ = 0;
  std::string text_;
  std::vector<BamTarget> targets_;
  std::thread walker_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Block> blks_;
  size_t taken_ = 0;
  int state_ = 0;        // 0 walking, 1 end of file, 2 error
  std::string werr_;
  bool stop_ = false;
};

}  // namespace strl
