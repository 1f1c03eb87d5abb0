// bam_reader.h -- minimal sequential BGZF/BAM reader for the `strling` CLI (htslib is not available in
// this image; zlib is).  It exposes exactly what src/strpkg/extract.nim and utils.nim:86-111 read through
// hts-nim: header text + targets, and per record tid/pos/mapq/flag/mate tid+pos/isize/cigar/4-bit SEQ/qname,
// written straight into the structure-of-arrays batch layout of include/strling_amd.h (SEQ 16-byte aligned).
#pragma once
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../../include/strling_amd.h"

namespace strl {

struct BamTarget {
  std::string name;
  uint32_t length;
};

// One batch of records in the strl_records layout (owning storage).
struct RecordBatch {
  std::vector<int32_t> tid, pos, mtid, mpos, isize, l_seq;
  std::vector<uint16_t> flag;
  std::vector<uint8_t> mapq;
  std::vector<uint32_t> cigar_off{0}, cigar;
  std::vector<uint64_t> seq_off, qname_off{0};
  std::vector<uint8_t> seq4;
  std::string qnames;
  void clear();
  size_t size() const { return tid.size(); }
  strl_records view();  // pads seq4 with the 32 bytes of slack the kernels may read
};

class BamReader {
 public:
  ~BamReader();
  bool open(const std::string &path, std::string &err);
  void close();
  const std::string &header_text() const { return text_; }
  const std::vector<BamTarget> &targets() const { return targets_; }
  // Appends up to max_records records to `b`.  Returns number appended (0 at EOF), -1 on error.
  // keep_secondary = false drops secondary/supplementary records (extract.nim:309) before they are batched.
  int64_t read(RecordBatch &b, int64_t max_records, std::string &err) { return read_until(b, max_records, INT32_MIN, 0, err); }
  // Same, but stops (without appending) at the first record that is not on stop_tid or starts at/after stop_pos.
  int64_t read_until(RecordBatch &b, int64_t max_records, int32_t stop_tid, int32_t stop_pos, std::string &err);
  // Region read through the .bai linear index (hts-nim `b.query(tid, beg, end)`, collect.nim:141): appends every record
  // of `tid` from the first one that can overlap [beg, end) up to the first one starting at or after `end`, in file
  // order.  Records ending before `beg` may be included; consumers apply the overlap filter (strl_spanners does).
  bool load_index(const std::string &bam_path, std::string &err);
  bool has_index() const { return !lin_.empty(); }
  int64_t read_region(RecordBatch &b, int32_t tid, int64_t beg, int64_t end, std::string &err);
  // position of the NEXT record, to come back to it later (used to revisit the unmapped tail)
  struct Pos { uint64_t block_off; uint32_t in_block; };
  Pos tell() const { return Pos{block_start_, (uint32_t)upos_}; }
  bool seek(Pos p, std::string &err);

 private:
  bool fill(std::string &err);                       // inflate the next BGZF block into ubuf_
  bool get(void *dst, size_t n, std::string &err);   // copy n decompressed bytes, crossing blocks
  FILE *f_ = nullptr;
  std::vector<uint8_t> cbuf_, ubuf_;
  size_t upos_ = 0;
  uint64_t block_start_ = 0, next_block_ = 0;
  bool eof_ = false;
  std::string text_;
  std::vector<BamTarget> targets_;
  std::vector<std::vector<uint64_t>> lin_;   // per reference: linear index (virtual offset per 16 KiB window)
  std::vector<uint64_t> ref_beg_;            // per reference: smallest chunk start of any bin (0 = no records)
};

}  // namespace strl
