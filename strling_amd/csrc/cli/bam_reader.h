// bam_reader.h -- minimal sequential BGZF/BAM reader for the `strling` CLI (htslib is not available in
// this image; zlib is).  It exposes exactly what src/strpkg/extract.nim and utils.nim:86-111 read through
// hts-nim: header text + targets, and per record tid/pos/mapq/flag/mate tid+pos/isize/cigar/4-bit SEQ/qname,
// written straight into the structure-of-arrays batch layout of include/strling_amd.h (SEQ 16-byte aligned).
#pragma once
#include <limits.h>
#include <stdint.h>
#include <sys/mman.h>
#include <stdio.h>
#include <functional>
#include <memory>
#include <new>
#include <type_traits>
#include <utility>
#include <condition_variable>
#include <mutex>
#include <future>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/strling_amd.h"

namespace strl {

class CramFile;
// `-f FASTA` of the command line: a CRAM needs the reference it was written against (extract.nim:278-279, call.nim:90); the
// readers below hand a file that starts with "CRAM" to cram_reader.h's CramFile and behave the same otherwise
extern std::string g_cram_fasta;

struct BamTarget {
  std::string name;
  uint32_t length;
};

// std::vector whose resize() leaves new trivially-constructible elements uninitialised: the batch arrays are hundreds of
// megabytes that the parser overwrites completely, value-initialising them first would cost a second pass over memory.
template <class T> struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  using std::allocator<T>::allocator;
  template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using rvec = std::vector<T, default_init_allocator<T>>;

// ... and for the few arrays of a quarter of a gigabyte (a whole genome's treads and their names): the same, in a mapping of
// its own with MADV_HUGEPAGE -- a hundred page faults instead of a hundred thousand when several threads fill it at once
// (transparent huge pages are "madvise" on the machines this runs on).  Smaller requests go to the heap.
template <class T> struct huge_allocator {
  using value_type = T;
  huge_allocator() = default;
  template <class U> huge_allocator(const huge_allocator<U> &) {}
  template <class U> struct rebind { using other = huge_allocator<U>; };
  static constexpr size_t HUGE = (size_t)2 << 20, MIN_BYTES = (size_t)8 << 20;
  T *allocate(size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= MIN_BYTES) {
      const size_t len = (bytes + HUGE - 1) & ~(HUGE - 1);
      void *m = mmap(nullptr, len + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m != MAP_FAILED) {
        // keep the mapping 2 MB aligned by trimming its ends
        const uintptr_t a = (reinterpret_cast<uintptr_t>(m) + HUGE - 1) & ~(uintptr_t)(HUGE - 1);
        if (a > reinterpret_cast<uintptr_t>(m)) munmap(m, a - reinterpret_cast<uintptr_t>(m));
        const uintptr_t end = reinterpret_cast<uintptr_t>(m) + len + HUGE;
        if (end > a + len) munmap(reinterpret_cast<void *>(a + len), end - (a + len));
        (void)madvise(reinterpret_cast<void *>(a), len, MADV_HUGEPAGE);
        return reinterpret_cast<T *>(a);
      }
      throw std::bad_alloc();        // (never the heap for a size deallocate() unmaps)
    }
    return static_cast<T *>(::operator new(bytes));
  }
  void deallocate(T *p, size_t n) {
    const size_t bytes = n * sizeof(T);
    if (bytes >= MIN_BYTES) { munmap(p, (bytes + HUGE - 1) & ~(HUGE - 1)); return; }
    ::operator delete(p);
  }
  template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const huge_allocator<U> &) const { return true; }
  template <class U> bool operator!=(const huge_allocator<U> &) const { return false; }
};
template <class T> using hvec = std::vector<T, huge_allocator<T>>;

// One batch of records in the strl_records layout (owning storage).
struct RecordBatch {
  rvec<int32_t> tid, pos, mtid, mpos, isize, l_seq;
  rvec<uint16_t> flag;
  rvec<uint8_t> mapq;
  rvec<uint32_t> cigar_off{0}, cigar;
  rvec<uint64_t> seq_off, qname_off{0};
  rvec<uint8_t> seq4;
  std::string qnames;
  void clear();
  size_t size() const { return tid.size(); }
  strl_records view();  // pads seq4 with the 32 bytes of slack the kernels may read
};

class BamReader {
 public:
  ~BamReader();
  bool open(const std::string &path, std::string &err);
  // a second handle on a file another reader has open (header, targets and index are copied from it, not parsed again):
  // `strling call` reads the regions of its bounds on several threads, one reader each
  bool open_like(const BamReader &other, std::string &err);
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
  // Where read_region(tid, beg, end) starts and, as far as the linear index can tell, where it is over -- for callers that
  // inflate the blocks of many regions elsewhere (strl_regions_fetch): c_beg = file offset of the BGZF block the walk starts
  // in, in_block = the first record's offset in that block's inflated bytes, c_hint = file offset of the block holding the
  // first record that overlaps a LATER 16 KiB window than `end - 1` lies in (the walk ends in that block or the one behind
  // it unless a record spans more than a window).  false: the region holds nothing / the index has no later window (BAM
  // files only; the caller then uses read_region).
  bool region_span(int32_t tid, int64_t beg, int64_t end, uint64_t &c_beg, uint32_t &in_block, uint64_t &c_hint) const;
  bool is_cram() const { return (bool)cram_; }
  const std::string &path() const { return path_; }
  // BAM records in memory (block_size-prefixed, back to back: what read_until walks through the BGZF layer) appended to `b`.
  // Returns the number appended, -1 on a malformed record.
  static int64_t append_records(RecordBatch &b, const uint8_t *p, size_t n, std::string &err);
  // position of the NEXT record, to come back to it later (used to revisit the unmapped tail)
  struct Pos { uint64_t block_off; uint32_t in_block; };
  Pos tell() const { return Pos{block_start_, (uint32_t)upos_}; }
  bool seek(Pos p, std::string &err);

 private:
  bool fill(std::string &err);                       // inflate the next BGZF block into ubuf_
  bool get(void *dst, size_t n, std::string &err);   // copy n decompressed bytes, crossing blocks
  FILE *f_ = nullptr;
  std::string path_;
  std::shared_ptr<CramFile> cram_;
  std::vector<uint8_t> cbuf_, ubuf_;
  size_t upos_ = 0;
  uint64_t block_start_ = 0, next_block_ = 0;
  bool eof_ = false;
  std::string text_;
  std::vector<BamTarget> targets_;
  std::vector<std::vector<uint64_t>> lin_;   // per reference: linear index (virtual offset per 16 KiB window)
  std::vector<uint64_t> ref_beg_;            // per reference: smallest chunk start of any bin (0 = no records)
};

// Minimal fork-join pool: parallel_for(n, fn) runs fn(i) for every i in [0, n) on `threads` threads (the caller is one).
class ThreadPool {
 public:
  explicit ThreadPool(int threads);
  ~ThreadPool();
  int size() const { return n_threads_; }
  void parallel_for(size_t n, const std::function<void(size_t)> &fn);

 private:
  struct Impl;
  Impl *impl_;
  int n_threads_;
};

// Whole-file sequential BAM reader for `extract` and the fragment-length pass: the file is
// mapped, BGZF blocks are inflated in parallel a superchunk (a few hundred blocks) at a time, record boundaries are
// found in one light sequential scan, and the records of a batch are parsed into the SoA layout in parallel.
// Same batch contract as BamReader::read.  htslib in the reference decodes on one thread (threads=0, extract.nim:275).
class BamStream {
 public:
  ~BamStream();
  bool open(const std::string &path, int threads, std::string &err);
  void close();
  const std::string &header_text() const { return text_; }
  const std::vector<BamTarget> &targets() const { return targets_; }
  int64_t read(RecordBatch &b, int64_t max_records, std::string &err);

 private:
  struct RecMeta { uint64_t off; int32_t l_seq; uint16_t n_cigar; uint8_t l_qname; };
  bool load_chunk(std::string &err);
  std::shared_ptr<CramFile> cram_;
  const uint8_t *map_ = nullptr;
  size_t map_len_ = 0, cpos_ = 0;
  // The superchunk being LOADED (inflate + record scan, load_chunk, on a thread of its own) ...
  rvec<uint8_t> u_, prev_;            // decompressed superchunk (leftover of the previous one in front); prev_ = that leftover
  std::vector<RecMeta> recs_;         // complete records of that superchunk
  size_t skip_ = 0;
  bool eof_ = false, have_prev_ = false;
  ThreadPool *pool_ = nullptr;
  // ... and the one being PARSED into batches by read(): the two run side by side (inflate is compute-bound, the parse is
  // bound by memory traffic), swapping buffers when read() runs out of records.
  rvec<uint8_t> cu_;
  std::vector<RecMeta> crecs_;
  size_t rec_next_ = 0;
  bool ceof_ = false, loading_ = false;
  ThreadPool *pool2_ = nullptr;       // the parse's own pool
  std::future<bool> load_;
  std::string load_err_;
  std::string text_;
  std::vector<BamTarget> targets_;
  // The BGZF header walk (18 bytes of every ~16 KB block: two or three dependent cache misses per block, 0.1 s per GB of
  // BAM) runs on a thread of its own ahead of the decoder, which takes finished block descriptors from `wblks_`.
  struct WBlk { size_t c_off; uint32_t clen, isize; size_t next; };   // deflate data at map_ + c_off; `next` = offset behind the block
  std::thread walker_;
  std::mutex w_mu_;
  std::condition_variable w_cv_;
  std::vector<WBlk> wblks_;           // blocks found and not yet handed out (isize > 0 only)
  size_t w_taken_ = 0;                // how many of wblks_ the decoder has consumed (compacted away now and then)
  size_t w_end_ = 0;                  // file offset the walker has reached
  int w_state_ = 0;                   // 0 walking, 1 reached the end of the file, 2 error (w_err_)
  std::string w_err_;
  bool w_stop_ = false;
};

}  // namespace strl
