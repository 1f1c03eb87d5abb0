// cram_reader.h -- CRAM 3.0 input for the `strling` CLI (extract.nim:253,278-279 and call.nim:90,106 open BAM or CRAM through
// htslib with `fai=FASTA`).  htslib is not available in this image, so this is a reader of its own:
// the container / slice / block structure, the compression header's encodings (EXTERNAL, HUFFMAN, BYTE_ARRAY_LEN,
// BYTE_ARRAY_STOP, BETA, GAMMA, SUBEXP), block methods raw / gzip / rANS 4x8 order 0 and 1, the record layout of CRAMv3
// section 10, reference-based reconstruction of SEQ and CIGAR from the read features, mate links inside a slice.
// Refused with a precise message: bzip2 / lzma blocks, CRAM 2.x and 3.1 (other codecs), GOLOMB / GOLOMB_RICE encodings,
// slices with an embedded reference, reference-less slices without full bases.
// Output: the same RecordBatch the BAM readers fill -- scoring, pairing and clustering run on the GPU as for a BAM.
#pragma once
#include <stdint.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "bam_reader.h"

namespace strl {

// Reference sequences by name, loaded contig by contig on first use (FASTA + .fai: random access; otherwise the whole file is
// read once, plain or gzip).  Upper-cased; bytes other than ACGT become N.
class RefCache {
 public:
  bool open(const std::string &fasta, std::string &err);
  // nullptr if the FASTA has no such sequence
  // (bases upper-cased, IUPAC codes kept: what htslib hands out and what the slices' MD5s are made of).  One cache serves every
  // reader of the FASTA (`strling call`'s workers share it); a FASTA with a .fai keeps the few contigs used last, not the genome
  std::shared_ptr<const std::string> get(const std::string &name, std::string &err);

 private:
  struct Fai { uint64_t len, off; uint32_t line_bases, line_width; };
  std::string path_;
  std::vector<std::pair<std::string, Fai>> fai_;
  std::vector<std::pair<std::string, std::shared_ptr<const std::string>>> loaded_;
  bool all_loaded_ = false, cur_line_continues_ = false;
  size_t keep_ = 6;             // contigs kept when they can be read again (a .fai): callers move along the genome together
  std::mutex mu_;
  bool load_all(std::string &err);
};

class CramFile {
 public:
  ~CramFile();
  static bool is_cram(const std::string &path);
  // share: the reference cache of another reader of the same FASTA (open_like); null = a cache of its own
  bool open(const std::string &path, const std::string &fasta, int threads, std::string &err, std::shared_ptr<RefCache> share = nullptr);
  std::shared_ptr<RefCache> ref_cache() const { return ref_; }
  const std::string &header_text() const { return text_; }
  const std::vector<BamTarget> &targets() const { return targets_; }
  // sequential read, file order: appends up to ~max_records records (whole containers); 0 at the end, -1 on error
  int64_t read(RecordBatch &b, int64_t max_records, std::string &err);
  // .crai region read: every record of `tid` in slices overlapping [beg, end) that starts before `end`, file order
  bool load_index(std::string &err);
  bool has_index() const { return have_index_; }
  int64_t read_region(RecordBatch &b, int32_t tid, int64_t beg, int64_t end, std::string &err);

 private:
  struct Container { uint64_t off, data_off; uint32_t len; int32_t ref_id, n_records; std::vector<int32_t> landmarks; uint64_t counter; };
  struct CraiEntry { int32_t tid; int64_t start, span; uint64_t c_off; uint32_t s_off, s_size; };
  bool parse_container_header(uint64_t off, Container &c, std::string &err) const;
  // decode the slices of one container (all of them, or the one at `only_landmark`) into `out`
  bool decode_container(const Container &c, int64_t only_landmark, RecordBatch &out, std::string &err);
  bool decode_container_body(const Container &c, int64_t only_landmark, RecordBatch &out, std::string &err);
  const uint8_t *map_ = nullptr;
  size_t map_len_ = 0;
  std::string path_, text_;
  std::vector<BamTarget> targets_;
  std::shared_ptr<RefCache> ref_;
  uint64_t next_off_ = 0;       // next container of the sequential read
  bool eof_ = false;
  int threads_ = 1;
  ThreadPool *pool_ = nullptr;
  std::vector<CraiEntry> crai_;                 // sorted by (tid, start)
  std::vector<int64_t> crai_max_end_;           // running maximum of start + span within a tid: where a backward scan may stop
  bool have_index_ = false, saw_eof_container_ = false, warned_eof_ = false;
  // the slice a region read decoded last (consecutive bounds fall into the same slice)
  uint64_t last_c_off_ = ~0ull;
  uint32_t last_s_off_ = 0;
  RecordBatch last_slice_;
  std::vector<RecordBatch> parts_;               // read(): the containers decoded side by side
};

}  // namespace strl
