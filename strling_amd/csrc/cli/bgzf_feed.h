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
  // One contiguous SHARE of a file another feed has open (`strling extract --gpus N`: a share per context, each with its own
  // header walker): blocks from the one at file offset start_coff up to the record boundary at the virtual offset
  // (end_coff, end_uoff) -- the block at end_coff is the share's last when end_uoff > 0, and its bytes from end_uoff on belong
  // to the next share (tail_trim()).  end_coff = 0: to the end of the file.  first_off = offset of the share's first record
  // in its first block's inflated bytes.  Both ends are record starts taken from the .bai (split_points).
  bool open_share(const BgzfFeed &whole, uint64_t start_coff, uint32_t first_off, uint64_t end_coff, uint32_t end_uoff, std::string &err);
  void close();
  // Virtual offsets (coffset << 16 | uoffset) of record starts named by the .bai next to `path` (linear index + bin chunk
  // starts), ascending and distinct; empty when there is no usable index.
  static std::vector<uint64_t> split_points(const std::string &path);
  // the number of records of an indexed BAM from the index's metadata pseudo-bins (bin 37450 of every reference: mapped /
  // placed-unmapped records) and the count of unplaced records behind the last reference -- what `samtools idxstats` adds up.
  // false: no index, or one written without them
  static bool indexed_records(const std::string &path, uint64_t &n);
  // inflated bytes at the end of the share's last block that belong to the next share (known once next() has said `last`)
  uint32_t tail_trim() const { return trim_; }
  const std::string &path() const { return path_; }
  const std::string &header_text() const { return text_; }
  const std::vector<BamTarget> &targets() const { return targets_; }
  uint64_t first_record_offset() const { return first_off_; }   // bytes into the first block's inflated data
  uint64_t first_block_offset() const { return start_; }        // file offset of that block
  void halt();                                                  // stops the header walker (the file stays open: read_at, open_share)
  size_t file_bytes() const { return map_len_; }
  // n bytes of the file at offset off into dst (any thread); false on a short read
  bool read_at(void *dst, size_t off, size_t n) const;
  // The same for the BULK of the file -- the compressed bytes on their way into the page-locked rings: a user-space copy out of a
  // read-only mapping of the file where there is one (measured 1.5 x the rate of pread's copy inside the kernel, 147 against
  // 100 GB/s on 12 threads: profiles/r06/feed_probe_mmapcopy.log), else read_at.  STRL_FEED=pread keeps pread everywhere.
  bool copy_at(void *dst, size_t off, size_t n) const;
  // [off, off + n) has been copied and will not be read again: its pages leave the mapping's page tables now (MADV_DONTNEED on
  // a file mapping drops the translations, the page cache keeps the pages).  Without this the process ended with 57 GB worth of
  // translations to tear down: +0.8 s between the last line of `strling extract` and its caller getting control back
  // (profiles/r06/full_size_feed_and_shares.log: real 3.30 s against 2.49 s with pread, both 2.3 s inside).
  // (Done by a thread of the feed's own, in the background: MADV_DONTNEED costs ~90 ns a page -- 7 ms per 328 MB chunk on the
  // thread that stages the next chunk made the feed the bottleneck of a one-device run (loop 2.0 -> 2.5 s); inside the copy threads it
  // took an eighth off the rate of eight CPU-bound shares.)
  void done_with(size_t off, size_t n) const;
  struct Block { size_t c_off; uint32_t clen, isize, crc; };     // DEFLATE payload at file offset c_off; CRC-32 of the inflated bytes (trailer)
  // Next run of consecutive non-empty blocks: at most max_blocks and max_bytes of file (first block's payload to the last
  // block's end).  Returns the number of blocks (0 at the end of the file), -1 on a malformed file.
  // *last (if given): this run reaches the end of the file / share (the call waits until the walker can tell).
  int64_t next(std::vector<Block> &out, size_t max_blocks, size_t max_bytes, std::string &err, bool *last = nullptr);

 private:
  void walk_from(size_t start, uint64_t end_coff, uint32_t end_uoff);
  int fd_ = -1;
  std::string path_;
  uint32_t trim_ = 0;
  uint64_t start_ = 0;
  size_t map_len_ = 0;          // file size
  const uint8_t *map_ = nullptr;   // read-only mapping of the whole file (copy_at), or null
  bool map_owned_ = false;
  // ranges whose translations are to be dropped, and the thread that drops them
  mutable std::thread zapper_;
  mutable std::mutex zmu_;
  mutable std::condition_variable zcv_;
  mutable std::vector<std::pair<size_t, size_t>> zq_;
  mutable bool zstop_ = false;
  void stop_zapper();
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
