// strling -- command-line front end over libstrling_amd.so, keeping the reference's CLI surface and files:
//   strling extract [-f FASTA] [-g STR.bed] [-p 0.8] [-q 40] [-v] BAM BIN      (src/strpkg/extract.nim:250-350)
//   strling merge   [-w -1] [-m 5] [-c 0] [-t 0] [-q 40] [-o PREFIX] [-v] BIN...  (src/strpkg/merge.nim:47-191)
//   strling index   [-g STR.bed] [-p 0.8] FASTA                                  (src/strpkg/genome_strs.nim:61-135,175-205)
//   strling call    [-m 5] [-c 0] [-t 0] [-q 40] [-l BED] [-b BOUNDS] [-o PREFIX] [-v] BAM BIN   (src/strpkg/call.nim:51-285)
// The BAM is decoded on the host (own multi-threaded BGZF/BAM reader), batches go through the C ABI into the HIP kernels
// as they are decoded, the pair logic (Cache.add) runs on the device once the whole file has been scored, and the
// .bin / -bounds.txt / -genotype.txt / -unplaced.txt writers are byte-compatible with the reference's.
// Not in this build: CRAM input (needs htslib's codec stack; the reference's error text is kept).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <map>
#include <sys/stat.h>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/strling_amd.h"
#include "bam_reader.h"
#include "bgzf_feed.h"
#include "cram_reader.h"
#include "cram_codecs.h"

using namespace strl;

static std::thread *g_bg_init = nullptr;   // device bring-up running beside the first host pass; exit() waits for it
static std::function<void()> g_bg_abort;   // ... after telling it to stop waiting for what the quitting thread will never publish
// --device K (or STRL_DEVICE=K): the device the first context of this process sits on; --gpus N takes K, K + 1, ... (mod the
// devices there are).  A pipeline that runs one `strling` process per sample (pipelines/bpipe.config:4) gives every process its
// own GPU this way.
static double since_exec();
static double g_main_at = -1;      // seconds between exec and main()
static int g_device0 = 0;
static void set_device0(const std::string &v) {
  const char *e = getenv("STRL_DEVICE");
  g_device0 = std::max(0, atoi(!v.empty() ? v.c_str() : (e ? e : "0")));
}
static int device_of(int g) { return (g_device0 + g) % std::max(1, strl_device_count()); }

[[noreturn]] static void quit(const char *fmt, ...) {   // Nim `quit msg`: message on stderr, exit code 1
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  if (g_bg_abort) g_bg_abort();
  if (g_bg_init && g_bg_init->joinable() && g_bg_init->get_id() != std::this_thread::get_id()) g_bg_init->join();
  exit(1);
}
#define CHECK(call)                                                        \
  do {                                                                     \
    const int rc__ = (call);                                               \
    if (rc__ != 0) quit("[strling] %s (status %d)", strl_last_error(), rc__); \
  } while (0)

struct Args {
  std::map<std::string, std::string> opt;
  std::vector<std::string> pos;
  bool flag(const char *k) const { return opt.count(k) != 0; }
  std::string get(const char *k, const char *dflt) const { auto it = opt.find(k); return it == opt.end() ? dflt : it->second; }
};

// long name -> (short name, takes value)
struct OptSpec { const char *longn; char shortn; bool value; };

static Args parse(int argc, char **argv, int first, const std::vector<OptSpec> &specs, const char *usage) {
  Args a;
  for (int i = first; i < argc; ++i) {
    std::string s = argv[i];
    if (s == "-h" || s == "--help") { fputs(usage, stdout); exit(0); }
    const OptSpec *sp = nullptr;
    if (s.size() > 2 && s[0] == '-' && s[1] == '-') {
      for (auto &x : specs) if (s.substr(2) == x.longn) sp = &x;
      if (!sp) quit("unknown option %s\n%s", s.c_str(), usage);
    } else if (s.size() == 2 && s[0] == '-' && s[1] != '-') {
      for (auto &x : specs) if (s[1] == x.shortn) sp = &x;
      if (!sp) quit("unknown option %s\n%s", s.c_str(), usage);
    }
    if (!sp) { a.pos.push_back(s); continue; }
    if (sp->value) {
      if (i + 1 >= argc) quit("option %s needs a value", s.c_str());
      a.opt[sp->longn] = argv[++i];
    } else a.opt[sp->longn] = "1";
  }
  return a;
}

static std::vector<BamTarget> targets_from_header(const std::string &text) {
  std::vector<BamTarget> t;
  size_t p = 0;
  while (p < text.size()) {
    size_t e = text.find('\n', p);
    if (e == std::string::npos) e = text.size();
    const std::string line = text.substr(p, e - p);
    if (line.rfind("@SQ", 0) == 0) {
      BamTarget bt{"", 0};
      size_t q = 0;
      while (q < line.size()) {
        size_t f = line.find('\t', q);
        if (f == std::string::npos) f = line.size();
        const std::string fld = line.substr(q, f - q);
        if (fld.rfind("SN:", 0) == 0) bt.name = fld.substr(3);
        if (fld.rfind("LN:", 0) == 0) bt.length = (uint32_t)strtoul(fld.c_str() + 3, nullptr, 10);
        q = f + 1;
      }
      t.push_back(bt);
    }
    p = e + 1;
  }
  return t;
}

static void append_record(RecordBatch &dst, const RecordBatch &src, size_t i) {
  dst.tid.push_back(src.tid[i]); dst.pos.push_back(src.pos[i]); dst.mtid.push_back(src.mtid[i]); dst.mpos.push_back(src.mpos[i]);
  dst.isize.push_back(src.isize[i]); dst.l_seq.push_back(src.l_seq[i]); dst.flag.push_back(src.flag[i]); dst.mapq.push_back(src.mapq[i]);
  for (uint32_t c = src.cigar_off[i]; c < src.cigar_off[i + 1]; ++c) dst.cigar.push_back(src.cigar[c]);
  dst.cigar_off.push_back((uint32_t)dst.cigar.size());
  dst.qnames.append(src.qnames, src.qname_off[i], src.qname_off[i + 1] - src.qname_off[i]);
  dst.qname_off.push_back(dst.qnames.size());
  const size_t so = (dst.seq4.size() + 15) & ~(size_t)15, sb = (size_t)(src.l_seq[i] + 1) / 2;
  dst.seq4.resize(so + sb, 0);
  memcpy(dst.seq4.data() + so, src.seq4.data() + src.seq_off[i], sb);
  dst.seq_off.push_back(so);
}

// utils.nim:86-111
// decode threads: STRL_THREADS, else what the machine (or the container's CPU quota) has, at most 64 (the reference decodes on one: threads=0, extract.nim:275)
static int cpu_quota() {   // CPUs this process may actually use: the cgroup CPU quota when there is one (containers), else the hardware
  unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  long long quota = -1, period = 100000;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
    if (fscanf(g, "%lld", &quota) != 1) quota = -1;
    fclose(g);
    if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
  }
  if (quota > 0 && period > 0) hw = (unsigned)std::min<long long>(hw, std::max<long long>(1, (quota + period - 1) / period));
  return (int)hw;
}
// open() failed: the reference's message for a file it cannot open; a CRAM the reader refuses says why
static void quit_open(const std::string &path, const std::string &err) {
  if (CramFile::is_cram(path) && !err.empty()) quit("[strling] %s: %s", path.c_str(), err.c_str());
  quit("couldn't open bam");
}

static int decode_threads() {
  const char *e = getenv("STRL_THREADS");
  if (e && atoi(e) > 0) return atoi(e);
  // 1.5 threads per granted CPU: decode threads stall on page faults of the file mapping and on memory, so a modest
  // oversubscription pays (measured under a 16-CPU quota on a 256-thread box: 16 threads 2.4e7 reads/s, 20-48 threads
  // 2.8-3.0e7); far more threads than the quota only get throttled.
  static const int n = std::max(1, std::min({64, (int)std::thread::hardware_concurrency() > 0 ? (int)std::thread::hardware_concurrency() : 64,
                                              (3 * cpu_quota() + 1) / 2}));
  return n;
}

static void fragment_length_distribution(const std::string &bam, uint32_t frag[4096]) {
  const int64_t n_reads = 2000000, skip_reads = 100000;
  memset(frag, 0, 4096 * sizeof(uint32_t));
  BamStream rd;
  std::string err;
  if (!rd.open(bam, decode_threads(), err)) quit_open(bam, err);
  RecordBatch b;
  std::vector<int32_t> skipped;
  int64_t i = -1, counted = 0;
  bool done = false;
  while (!done) {
    b.clear();
    const int64_t got = rd.read(b, 1 << 16, err);
    if (got < 0) quit("[strling] error reading %s: %s", bam.c_str(), err.c_str());
    if (got == 0) break;
    for (int64_t k = 0; k < got; ++k) {
      ++i;
      const uint16_t f = b.flag[(size_t)k];
      if (!(f & 0x2)) continue;
      if (f & (0x800 | 0x100)) continue;
      const int32_t is = b.isize[(size_t)k];
      if (is < 0 || is > 4095) continue;
      if (i < skip_reads) { skipped.push_back(is); continue; }
      skipped.clear();
      frag[is]++;
      if (++counted > n_reads) { done = true; break; }
    }
  }
  uint64_t sum = 0;
  for (int k = 0; k < 4096; ++k) sum += frag[k];
  if ((uint32_t)sum == 0) {
    fprintf(stderr, "using first reads in fragment_length_distribution calculation as there were not enough\n");
    for (int32_t is : skipped) frag[is]++;
  }
}


// ---- genome STR index: repeat_windows + genome_repeats' writer (genome_strs.nim:61-92, :116-137) -------------------
// The FASTA is read front to back with zlib (plain, gzip or bgzip), one contig in memory at a time, in file order --
// the order hts-nim's Fai enumerates.  Windows are scored on the device (strl_index_chrom); merging + trimming is
// strl_index_regions.  Returns the number of rows written.
static bool file_exists(const std::string &p) { return access(p.c_str(), F_OK) == 0; }

static uint64_t build_genome_index(strl_ctx *ctx, const std::string &fasta, const std::string &bed_path) {
  gzFile in = gzopen(fasta.c_str(), "rb");
  if (!in) quit("[strling] couldn't open fasta %s make sure file is present and has a .fai index", fasta.c_str());
  gzbuffer(in, 1 << 20);
  FILE *fh = fopen(bed_path.c_str(), "w");
  if (!fh) quit("[strling] couldn't open bed file: %s for writing", bed_path.c_str());
  uint64_t n_rows = 0;
  std::string name, seq;
  std::vector<uint32_t> words;
  std::vector<strl_region> regions;
  bool have = false;
  const uint32_t window_size = 100, step = 60;                                     // genome_strs.nim:131-132
  auto finish = [&]() {
    if (!have) return;
    if (seq.size() > 2000000) fprintf(stderr, "[strling] finding STR regions on reference chromosome: %s\n", name.c_str());
    uint64_t nw = 0, nr = 0;
    CHECK(strl_index_chrom(ctx, seq.data(), seq.size(), window_size, step, nullptr, &nw));
    words.assign((size_t)nw + 1, 0);
    CHECK(strl_index_chrom(ctx, seq.data(), seq.size(), window_size, step, words.data(), &nw));
    regions.resize((size_t)nw + 1);
    CHECK(strl_index_regions(seq.data(), seq.size(), words.data(), nw, window_size, step, regions.data(), regions.size(), &nr));
    for (uint64_t i = 0; i < nr; ++i)
      fprintf(fh, "%s\t%llu\t%llu\t%s\n", name.c_str(), (unsigned long long)regions[(size_t)i].start, (unsigned long long)regions[(size_t)i].stop, regions[(size_t)i].unit);
    n_rows += nr;
  };
  std::vector<char> line(1 << 16);
  bool in_header = false;   // a header line longer than the buffer continues on the next gzgets
  while (gzgets(in, line.data(), (int)line.size())) {
    size_t len = strlen(line.data());
    const bool complete = len && line[len - 1] == '\n';
    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) --len;
    if (in_header) { in_header = !complete; continue; }
    if (len && line[0] == '>') {
      finish();
      have = true;
      seq.clear();
      size_t e = 1;
      while (e < len && line[e] != ' ' && line[e] != '\t') ++e;
      name.assign(line.data() + 1, e - 1);
      in_header = !complete;
      continue;
    }
    if (have) seq.append(line.data(), len);
  }
  finish();
  gzclose(in);
  fclose(fh);
  fprintf(stderr, "[strling] found %llu STR-like regions in the genome\n", (unsigned long long)n_rows);
  return n_rows;
}

// read_bed.nim:18-50, flattened per BAM tid
struct Genome {
  std::vector<uint8_t> has;
  std::vector<int64_t> off;
  std::vector<int32_t> st, en;
};
static Genome read_genome_bed(const std::string &path, const std::vector<BamTarget> &targets) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) quit("[strling] couldn't open bed file: %s", path.c_str());
  std::map<std::string, int> tid;
  for (size_t i = 0; i < targets.size(); ++i) tid[targets[i].name] = (int)i;
  std::vector<std::vector<std::pair<int32_t, int32_t>>> per(targets.size());
  Genome g;
  g.has.assign(targets.size(), 0);
  char line[1 << 16];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#' || strncmp(line, "track ", 6) == 0) continue;
    char chrom[4096];
    long a, b;
    if (sscanf(line, "%4095[^\t]\t%ld\t%ld", chrom, &a, &b) != 3) { fprintf(stderr, "[slivar] skipping bad bed line:%s", line); continue; }
    auto it = tid.find(chrom);
    if (it == tid.end()) continue;
    g.has[(size_t)it->second] = 1;
    per[(size_t)it->second].push_back({(int32_t)a, (int32_t)b});
  }
  fclose(f);
  g.off.assign(targets.size() + 1, 0);
  for (size_t t = 0; t < targets.size(); ++t) {
    for (auto &iv : per[t]) { g.st.push_back(iv.first); g.en.push_back(iv.second); }
    g.off[t + 1] = (int64_t)g.st.size();
  }
  return g;
}

// genome_repeats, genome_strs.nim:107-146: existing file, or build it from the FASTA (into a temporary file when
// no -g was given).  An existing -g is accepted without -f here; the reference insists on opening the FASTA first.
// The host half: the table read and flattened per tid.  `need_ctx` hands over a context when the table has to be BUILT (the
// window scorer runs on the device); with an existing file no device is touched.
template <typename F> static Genome genome_host_half(const Args &a, const std::vector<BamTarget> &targets, F need_ctx) {
  std::string bed_path = a.get("genome-repeats", "");
  const bool is_tmp = bed_path.empty();
  if (is_tmp) {
    const char *td = getenv("TMPDIR");
    bed_path = std::string(td ? td : "/tmp") + "/strling." + std::to_string((long)getpid()) + ".bed";
  }
  if (is_tmp || !file_exists(bed_path)) {
    if (!a.flag("fasta")) quit("[strling] couldn't open fasta %s make sure file is present and has a .fai index", a.get("fasta", "").c_str());
    strl_ctx *ctx = need_ctx();                  // (with its options set: the window scorer reads -p)
    build_genome_index(ctx, a.get("fasta", ""), bed_path);
  } else {
    fprintf(stderr, "[strling] using existing file %s for genome repeats\n", bed_path.c_str());
  }
  Genome g = read_genome_bed(bed_path, targets);
  fprintf(stderr, "[strling] got STR repeats from genome into an interval tree\n");
  if (is_tmp) remove(bed_path.c_str());
  return g;
}
static void setup_genome(strl_ctx *ctx, const Args &a, const std::vector<BamTarget> &targets, const std::vector<strl_ctx *> &more = {}) {
  const Genome g = genome_host_half(a, targets, [&] { return ctx; });
  strl_genome_str gs{(int32_t)targets.size(), g.has.data(), g.off.data(), g.st.data(), g.en.data()};
  CHECK(strl_ctx_set_genome(ctx, &gs));
  for (strl_ctx *m : more) CHECK(strl_ctx_set_genome(m, &gs));
}

static int extract_front(const Args &a, const std::string &bam, const std::string &bin, double p, uint8_t min_mapq, bool verbose);
constexpr int EXTRACT_AGAIN_ON_HOST = -77, EXTRACT_AGAIN_HOST_FRONT = -78, EXTRACT_AGAIN_BY_CHUNKS = -79;

static int extract_main(int argc, char **argv) {
  const char *usage =
      "strling extract\n\nUsage:\n  strling extract [options] bam bin\n\nArguments:\n  bam              path to bam file\n"
      "  bin              path bin to output bin file to be created\n\nOptions:\n  -f, --fasta=FASTA          path to fasta file (required for CRAM)\n"
      "  -g, --genome-repeats=GENOME_REPEATS\n                             optional path to genome repeats file. if it does not exist, it will be created\n"
      "  -p, --proportion-repeat=PROPORTION_REPEAT\n                             proportion of read that is repetitive to be considered as STR (default: 0.8)\n"
      "  -q, --min-mapq=MIN_MAPQ    minimum mapping quality (does not apply to STR reads) (default: 40)\n"
      "  --gpus=N                   spread the file over N GPUs (a contiguous share each: inflate, parse and scoring there; the pair logic on the first) (default: 1)\n"
      "  --device=K                 the GPU this process uses (the first of --gpus N) (default: 0; STRL_DEVICE)\n"
      "  -v, --verbose\n  -h, --help                 Show this help\n";
  if (argc <= 2) { fputs(usage, stdout); return 0; }
  const Args a = parse(argc, argv, 2, {{"fasta", 'f', true}, {"genome-repeats", 'g', true}, {"proportion-repeat", 'p', true},
                                       {"min-mapq", 'q', true}, {"verbose", 'v', false}, {"batch", 'B', true}, {"gpus", 'G', true}, {"device", 'D', true}}, usage);
  if (a.pos.size() != 2) quit("expected 2 arguments (bam, bin)\n%s", usage);
  set_device0(a.get("device", ""));
  const std::string bam = a.pos[0], bin = a.pos[1];
  const double p = atof(a.get("proportion-repeat", "0.8").c_str());
  const uint8_t min_mapq = (uint8_t)atoi(a.get("min-mapq", "40").c_str());
  const bool verbose = a.flag("verbose");
  const int64_t batch = atoll(a.get("batch", "1048576").c_str());
  g_cram_fasta = a.get("fasta", "");
  const bool is_cram = CramFile::is_cram(bam);     // decoded by host threads (cli/cram_reader.cpp); scoring + pairing on the device as for a BAM
  if (!is_cram) {
    // Default: the whole BAM front end on the device (inflate, record scan, parse: strl_front_*).  STRL_FRONT=host keeps the
    // host reader (threads inflate and parse, the device scores); STRL_PAIR=host (the host's streaming Cache) implies it.
    const char *fe = getenv("STRL_FRONT"), *pe = getenv("STRL_PAIR");
    if (!(fe && strcmp(fe, "host") == 0) && !(pe && strcmp(pe, "host") == 0)) {
      int r = extract_front(a, bam, bin, p, min_mapq, verbose);
      // (EXTRACT_AGAIN_BY_CHUNKS: --gpus N found the .bai's record starts not to be record starts; the same front end again,
      // chunk by chunk over the contexts, which needs no index)
      if (r == EXTRACT_AGAIN_BY_CHUNKS) {
        setenv("STRL_SHARES", "0", 1);
        r = extract_front(a, bam, bin, p, min_mapq, verbose);
      }
      if (r != EXTRACT_AGAIN_ON_HOST && r != EXTRACT_AGAIN_HOST_FRONT) return r;
      if (r == EXTRACT_AGAIN_ON_HOST) setenv("STRL_PAIR", "host", 1);   // the device join passed (hash collision / one qname on hundreds of records): the string-keyed Cache
      // (EXTRACT_AGAIN_HOST_FRONT: the device front end refused the file -- a block its decoder does not take, a record of more
      // than a megabyte: the host reader, whose verdict on the file is zlib's, takes over)
    }
  }

  // The HIP runtime + device context come up (a few hundred ms of driver work on one thread) while the host threads
  // run the fragment-length pass.
  strl_ctx *ctx = nullptr;
  int ctx_rc = 0;
  std::string ctx_err;
  std::thread ctx_thread([&] {
    ctx_rc = strl_ctx_create(device_of(0), &ctx);
    if (ctx_rc) ctx_err = strl_last_error();     // (the error text is thread-local in the library)
  });
  g_bg_init = &ctx_thread;
  uint32_t frag[4096];
  fragment_length_distribution(bam, frag);                                        // extract.nim:281
  const int frag_median = strl_frag_median(frag, 0.5);
  if (verbose) {
    fprintf(stderr, "Calculated median fragment length:%d\n", frag_median);
    fprintf(stderr, "10th, 90th percentile of fragment length:%d %d\n", strl_frag_median(frag, 0.1), strl_frag_median(frag, 0.9));
  }
  BamStream rd;
  std::string err;
  if (!rd.open(bam, decode_threads(), err)) quit_open(bam, err);
  ctx_thread.join();
  g_bg_init = nullptr;
  if (ctx_rc) quit("[strling] %s (status %d)", ctx_err.c_str(), ctx_rc);
  strl_opts opts{frag_median, p, min_mapq};
  CHECK(strl_ctx_set_opts(ctx, &opts));
  setup_genome(ctx, a, rd.targets());
  // Pair logic: on the device over the whole file (default), or the host's streaming Cache (STRL_PAIR=host; also the
  // way out -- taken automatically -- for inputs the device join passes on: one qname on hundreds of primary records).
  const char *pair_env = getenv("STRL_PAIR");
  const bool host_pair = pair_env && strcmp(pair_env, "host") == 0;
  strl_pairer *pairer = nullptr;
  if (host_pair) CHECK(strl_pairer_create(&opts, &pairer));
  else CHECK(strl_extract_begin(ctx, 0));

  ThreadPool prep(std::min(decode_threads(), 16));   // SoA preparation is light; the decoder keeps the other cores
  rvec<int32_t> end;
  rvec<uint32_t> so;
  std::vector<uint32_t> whole;
  rvec<uint16_t> ls, cl, cr;
  rvec<uint8_t> cig;
  rvec<strl_pair_rec> rows;
  rvec<uint64_t> qh;
  std::vector<strl_soft_rec> soft;
  // qnames of every record seen so far (device path): a tread names its record by index
  struct QChunk { uint64_t first; std::string names; rvec<uint64_t> off; };
  std::vector<QChunk> qchunks;
  uint64_t n_seen = 0;       // records handed to the device so far (secondary / supplementary included: they keep their index)
  int64_t n_tail = 0;        // length of the trailing run of unplaced records: the "*" region extract.nim:326 visits again
  int64_t tail_primary = 0;  // primary records among them
  double t_read = 0, t_soa = 0, t_score = 0, t_pair = 0;   // -v: where the wall time of the loop goes
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  bool over_limit = false;   // more records than one device pass takes: the file goes to the streaming host Cache (STRL_ERR_LIMIT)
  auto run_batch = [&](RecordBatch &b) {
    const size_t n = b.size();
    if (!n || over_limit) return;
    const auto ta = now();
    strl_records rec = b.view();
    end.resize(n); so.resize(n); ls.resize(n); cl.resize(n); cr.resize(n); cig.resize(n);
    if (!host_pair) { rows.resize(n); qh.resize(n); }
    // SoA derivation (+ pair rows + qname hashes) of disjoint record ranges in parallel
    const size_t parts = std::min<size_t>(std::max<size_t>(n / 4096, 1), (size_t)prep.size() * 4);
    std::vector<uint32_t> mxs(parts, 0);
    std::vector<int> rcs(parts, 0);
    std::vector<std::string> errs(parts);
    prep.parallel_for(parts, [&](size_t part) {
      const size_t i0 = n * part / parts, i1 = n * (part + 1) / parts;
      strl_records sub = rec;
      sub.n = (int64_t)(i1 - i0);
      sub.tid += i0; sub.pos += i0; sub.mtid += i0; sub.mpos += i0; sub.flag += i0; sub.mapq += i0; sub.cigar_off += i0; sub.seq_off += i0;
      sub.l_seq += i0; sub.qname_off += i0;
      int rc = strl_soa_from_records(&sub, end.data() + i0, so.data() + i0, ls.data() + i0, cl.data() + i0, cr.data() + i0, cig.data() + i0, &mxs[part]);
      if (!rc && !host_pair) rc = strl_pair_rows(&sub, end.data() + i0, cl.data() + i0, cr.data() + i0, cig.data() + i0, rows.data() + i0);
      if (!rc && !host_pair) rc = strl_qname_hash(&sub, qh.data() + i0);
      rcs[part] = rc;
      if (rc) errs[part] = strl_last_error();
    });
    uint32_t mx = 0;
    for (size_t k = 0; k < parts; ++k) { if (rcs[k]) quit("[strling] %s (status %d)", errs[k].c_str(), rcs[k]); mx = std::max(mx, mxs[k]); }
    strl_read_soa soa{};
    soa.n = n; soa.tid = rec.tid; soa.pos = rec.pos; soa.end = end.data(); soa.seq_off = so.data(); soa.l_seq = ls.data();
    soa.clip_l = cl.data(); soa.clip_r = cr.data(); soa.mapq = rec.mapq; soa.cig = cig.data(); soa.seq4 = rec.seq4;
    soa.seq4_bytes = b.seq4.size(); soa.max_l_seq = mx; soa.mem = STRL_MEM_HOST;
    const auto tb = now();
    if (host_pair) {
      whole.resize(n); soft.resize(2 * n + 2);
      uint64_t ns = 0;
      CHECK(strl_score_reads(ctx, &soa, whole.data(), soft.data(), 2 * n, &ns, nullptr));
      const auto tc = now();
      CHECK(strl_pairer_add(pairer, &rec, whole.data(), soft.data(), ns));
      t_score += secs(tb, tc); t_pair += secs(tc, now());
    } else {
      const strl_pair_soa pp{rows.data(), qh.data()};
      const int rc_add = strl_extract_add(ctx, &soa, &pp);
      if (rc_add == STRL_ERR_LIMIT) { over_limit = true; return; }
      if (rc_add) quit("[strling] %s (status %d)", strl_last_error(), rc_add);
      CHECK(strl_ctx_sync(ctx));                       // the batch's buffers are reused by the decoder
      QChunk q;
      q.first = n_seen;
      q.names.swap(b.qnames);
      q.off.swap(b.qname_off);
      qchunks.push_back(std::move(q));
      for (size_t i = 0; i < n; ++i) {
        if (rec.tid[i] < 0) { ++n_tail; if (!(rec.flag[i] & (0x100 | 0x800))) ++tail_primary; }
        else { n_tail = 0; tail_primary = 0; }
      }
      n_seen += n;
      t_score += secs(tb, now());
    }
    t_soa += secs(ta, tb);
  };

  fprintf(stderr, "[strling] collecting str-like reads\n");
  const auto t0 = std::chrono::steady_clock::now();
  // the decoder runs one batch ahead of scoring + pairing (two slots)
  struct Slot { RecordBatch b; int64_t got = 0; std::string err; };
  Slot slots[2];
  std::mutex mu;
  std::condition_variable cv;
  int filled = 0;
  bool stop = false;
  std::thread producer([&] {
    for (int w = 0;; w ^= 1) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return filled < 2 || stop; });
        if (stop) return;
      }
      Slot &sl = slots[w];
      sl.b.clear();
      sl.got = rd.read(sl.b, batch, sl.err);
      {
        std::lock_guard<std::mutex> lk(mu);
        ++filled;
      }
      cv.notify_all();
      if (sl.got <= 0) return;
    }
  });
  RecordBatch tail;
  int64_t nreads = 0, last_tid = -1;
  for (int r = 0;; r ^= 1) {
    const auto tr0 = now();
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return filled > 0; });
    }
    t_read += secs(tr0, now());
    RecordBatch &b = slots[r].b;
    const int64_t got = slots[r].got;
    if (got < 0) { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); producer.join(); quit("[strling] error reading %s: %s", bam.c_str(), slots[r].err.c_str()); }
    if (got == 0) break;
    for (size_t i = 0; i < (size_t)got; ++i) {
      const uint16_t f = b.flag[i];
      if (host_pair) { if (b.tid[i] >= 0) { if (tail.size()) tail.clear(); } else append_record(tail, b, i); }   // the "*" region: unplaced records at the end
      if (f & (0x100 | 0x800)) continue;
      if (b.tid[i] != last_tid && b.tid[i] >= 0) {
        if (rd.targets()[(size_t)b.tid[i]].length > 2000000u) fprintf(stderr, "[strling] extracting chromosome:%s\n", rd.targets()[(size_t)b.tid[i]].name.c_str());
        last_tid = b.tid[i];
      }
      ++nreads;
    }
    run_batch(b);
    if (over_limit) {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_all();
      producer.join();
      fprintf(stderr, "[strling] %s: repeating the extraction with the host pair logic\n", strl_last_error());
      strl_ctx_destroy(ctx);
      setenv("STRL_PAIR", "host", 1);
      return extract_main(argc, argv);
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      --filled;
    }
    cv.notify_all();
    if (verbose) {
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      fprintf(stderr, "%lld %.1f reads/sec\n", (long long)nreads, (double)nreads / std::max(s, 1e-9));
    }
  }
  producer.join();
  fprintf(stderr, "[strling] extracting unmapped reads\n");

  std::vector<strl_tread> dev_treads;
  std::vector<uint64_t> dev_qoff;
  std::string dev_qn;
  const strl_tread *treads = nullptr;
  const uint64_t *qoff = nullptr;
  const char *qn = nullptr;
  uint64_t nt = 0, pending = 0;
  if (host_pair) {
    for (size_t i = 0; i < tail.size(); ++i) if (!(tail.flag[i] & (0x100 | 0x800))) ++nreads;
    run_batch(tail);                                                                // extract.nim:326-329
    CHECK(strl_pairer_result(pairer, &treads, &nt, &qoff, &qn, &pending));
  } else {
    nreads += tail_primary;   // the "*" region is counted a second time by the reference's progress counter (extract.nim:326-329)
    const auto tp0 = now();
    int rc = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      CHECK(strl_extract_finish(ctx, n_tail, attempt ? std::min<uint64_t>(3 * n_seen + 16, 0x7ffffff0ull) : 0, attempt ? std::min<uint64_t>(8 * n_seen + 16, 0x7ffffff0ull) : 0));
      rc = strl_treads_fetch(ctx, nullptr, 0, &nt, nullptr);
      if (rc != STRL_ERR_CAPACITY) break;
    }
    if (rc == STRL_ERR_FORMAT) {
      fprintf(stderr, "[strling] %s: repeating the extraction with the host pair logic\n", strl_last_error());
      strl_ctx_destroy(ctx);
      setenv("STRL_PAIR", "host", 1);
      return extract_main(argc, argv);
    }
    if (rc) quit("[strling] %s (status %d)", strl_last_error(), rc);
    dev_treads.resize((size_t)nt + 1);
    CHECK(strl_treads_fetch(ctx, dev_treads.data(), nt, &nt, nullptr));
    t_pair += secs(tp0, now());
    dev_qoff.assign(1, 0);
    for (uint64_t i = 0; i < nt; ++i) {
      strl_tread &t = dev_treads[(size_t)i];
      const uint64_t g = (uint64_t)t.qname_id;
      size_t lo = 0, hi = qchunks.size();
      while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (qchunks[mid].first <= g) lo = mid; else hi = mid; }
      const QChunk &q = qchunks[lo];
      const uint64_t k = g - q.first;
      dev_qn.append(q.names, (size_t)q.off[(size_t)k], (size_t)(q.off[(size_t)k + 1] - q.off[(size_t)k]));
      dev_qoff.push_back(dev_qn.size());
      t.qname_id = (int64_t)i;
    }
    treads = dev_treads.data(); qoff = dev_qoff.data(); qn = dev_qn.data();
  }
  fprintf(stderr, "[strling] writing binary file:%s\n", bin.c_str());
  CHECK(strl_bin_write(bin.c_str(), (float)p, min_mapq, frag, rd.header_text().data(), (int32_t)rd.header_text().size(), treads, nt, qoff, qn));
  fprintf(stderr, "[strling] finished extraction\n");
  if (verbose) {
    fprintf(stderr, "[strling] %lld reads, %llu STR reads, %llu reads still waiting for a mate\n", (long long)nreads, (unsigned long long)nt, (unsigned long long)pending);
    fprintf(stderr, "[strling] seconds: total %.3f  waiting for the decoder %.3f  soa %.3f  device scoring (incl. copies) %.3f  pair logic %.3f\n",
            secs(t0, now()), t_read, t_soa, t_score, t_pair);
  }
  if (pairer) strl_pairer_destroy(pairer);
  strl_ctx_destroy(ctx);
  return 0;
}

// `strling extract` with the BAM front end on the device: this thread walks BGZF headers and copies compressed bytes into
// page-locked buffers; inflate, record scan, parse, scorer, pair logic all run on the GPU (extract.nim:275-348).
// --gpus N: the file's chunks go round-robin over N contexts (one per device, round-robin over the devices there are);
// what the pair logic needs of every record is gathered on the first one at the end (strl_ctxs_extract_gather).
// `extract --gpus G`: where the file is cut into contiguous shares, as virtual offsets (coffset << 16 | uoffset).  cut[0] = the
// first record; every further cut is a record start the .bai names, the first one at or behind g / G of the bytes behind the
// header.  Fewer than G entries when the index has too few points (a contig-free tail, a tiny file); one entry: no shares.
static std::vector<uint64_t> share_cuts(const BgzfFeed &feed, const std::string &bam, int G) {
  std::vector<uint64_t> cut;
  const std::vector<uint64_t> pts = BgzfFeed::split_points(bam);
  const uint64_t lo = feed.first_block_offset(), span = feed.file_bytes() - lo;
  cut.push_back((lo << 16) | feed.first_record_offset());
  for (int g = 1; g < G; ++g) {
    const uint64_t want = (lo + (uint64_t)((double)span * g / G)) << 16;
    auto it = std::lower_bound(pts.begin(), pts.end(), std::max(want, cut.back() + 1));
    if (it == pts.end() || (*it >> 16) >= feed.file_bytes()) break;     // (an index of a longer / another file: no cut there)
    cut.push_back(*it);
  }
  return cut;
}

static int extract_front(const Args &a, const std::string &bam, const std::string &bin, double p, uint8_t min_mapq, bool verbose) {
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double>(y - x).count(); };
  static const char *env_blocks = getenv("STRL_CHUNK_BLOCKS");     // tests: tiny chunks put records across chunk borders
  // Chunk = one inflate launch: 8192 blocks (~0.5 GB inflated) keep a 1 GB file's pipeline fine-grained (fill, drain and the
  // page-locked buffers all grow with the chunk); a whole-genome BAM takes 16384.  Consecutive chunks' inflates overlap (two
  // streams), so a launch's tail costs nothing and larger chunks no longer inflate faster (6.7e7-read file, loop seconds at
  // 8192 / 16384 / 24576 / 32768 blocks: 0.40 / 0.36 / 0.37 / 0.37) -- but three page-locked buffers of 24576 blocks took
  // 0.39 - 0.44 s to allocate, longer than the device context beside them (0.30 s at 16384; profiles/r04/extract_chunk_sweep.txt)
  size_t auto_blocks = 8192;
  {
    struct stat st;
    if (stat(bam.c_str(), &st) == 0) auto_blocks = std::min<size_t>(16384, std::max<size_t>(8192, (size_t)st.st_size / 16384 / 12));
  }
  const int G = std::max(1, atoi(a.get("gpus", "1").c_str()));
  // --gpus N: every context takes one contiguous share of the file (below) unless STRL_SHARES=0 / there is no usable .bai
  const char *env_shares = getenv("STRL_SHARES");        // (read per call: extract_main sets it to 0 for the chunk-by-chunk repeat)
  bool use_shares = G > 1 && !(env_shares && strcmp(env_shares, "0") == 0);
  if (use_shares) {
    struct stat st;
    // (a share's ring is four page-locked buffers of a chunk each, and page-locking goes through the driver at ~20 GB/s whatever
    // the number of threads (profiles/r06/feed_probe_shm.log): 8 shares x 4 x 16384 blocks were 10 GB = half a second and more of
    // every start; 4096 blocks -- a 2.5 ms inflate launch of 4096 workgroups -- keep 8 rings at 2.6 GB)
    const size_t cap = G >= 4 ? 4096 : 8192;
    if (stat(bam.c_str(), &st) == 0) auto_blocks = std::min<size_t>(cap, std::max<size_t>(2048, (size_t)st.st_size / (size_t)G / 16384 / 12));
  }
  const size_t chunk_blocks = env_blocks && atoi(env_blocks) > 0 ? (size_t)atoi(env_blocks) : auto_blocks;
  const size_t chunk_bytes = std::max<size_t>((size_t)1 << 20, chunk_blocks * 20000);       // compressed bytes one chunk may span
  std::vector<strl_ctx *> ctxs((size_t)G, nullptr);
  std::vector<int> ctx_rc((size_t)G, 0);
  std::vector<std::string> ctx_err((size_t)G);
  // four per context: [4 g + (its chunk count % 4)] -- while chunk k is handed over, chunk k+2 is read from the file, chunk k+1's
  // copy to the device has just been queued and chunk k's may still be going; the buffer chunk k+2 takes is chunk k-2's, whose
  // record scan this thread has waited for.  Block tables: coff u64 | clen | isize | crc u32
  const size_t RING = 4;
  std::vector<uint8_t *> pin(RING * G, nullptr), pin_meta(RING * G, nullptr);
  const auto t_start = now();
  double t_ctx = 0, t_pin = 0;
  uint32_t *fw_early = nullptr;          // flag / isize words of the first records (fragment lengths)
  const uint64_t early_n = 2400000;
  // Bring-up, one thread per context, side by side (round 6; before: one after the other, 0.1 - 0.2 s each, then options, genome
  // table and per-read state of every context in turn on the main thread -- more than a second of fixed cost at 8 contexts):
  //   create the context (the first HIP call of the process starts the runtime: one thread does that alone, the others wait
  //   for it and then run side by side) -> options -> [the genome table's host half is ready] -> genome table -> [the file's
  //   header is walked, the shares are cut] -> per-read state + front-end buffers of THIS context.
  // The main thread walks the header and parses the BED meanwhile and only waits where it needs a context.
  // STRL_SERIAL_CTX=1: the contexts one after the other, as before.
  strl_opts opts{0, p, min_mapq};              // the fragment-length median is only needed by the pair logic: set before strl_extract_finish
  struct BringUp {
    std::mutex mu;
    std::condition_variable cv;
    bool runtime_up = false, genome_ready = false, plan_ready = false, pin_done = false, give_up = false;
    std::vector<uint8_t> created, done;
    strl_genome_str gs{};
    int32_t n_ref = 0;
    std::vector<uint64_t> first_off, hint;
    std::vector<double> t_create, t_state;
  } bu;
  bu.created.assign((size_t)G, 0); bu.done.assign((size_t)G, 0);
  bu.first_off.assign((size_t)G, 0); bu.hint.assign((size_t)G, 0);
  bu.t_create.assign((size_t)G, 0.0); bu.t_state.assign((size_t)G, 0.0);
  // the HIP runtime, the contexts and the page-locked buffers come up on threads beside the header walk
  std::thread pin_thread([&] {
    const auto c0 = now();
    std::vector<std::thread> each;            // (the time is the kernel's, faulting and locking the pages: a thread per buffer)
    for (size_t k = 1; k < pin.size(); ++k) each.emplace_back([&, k] { pin[k] = static_cast<uint8_t *>(strl_pinned_alloc(chunk_bytes + 64)); });
    pin[0] = static_cast<uint8_t *>(strl_pinned_alloc(chunk_bytes + 64));
    for (auto &q : pin_meta) q = static_cast<uint8_t *>(strl_pinned_alloc(chunk_blocks * 20 + 64));
    fw_early = static_cast<uint32_t *>(strl_pinned_alloc(early_n * 4));     // (allocating page-locked memory inside the loop stalls the device)
    for (auto &t : each) t.join();
    t_pin = secs(c0, now());
    { std::lock_guard<std::mutex> lk(bu.mu); bu.pin_done = true; }
    bu.cv.notify_all();
  });
  static const bool serial_ctx = getenv("STRL_SERIAL_CTX") != nullptr;
  auto bring_up = [&](int g) {
    auto wait_for = [&](auto pred) {             // -> false: the main thread gave up (an error exit is under way)
      std::unique_lock<std::mutex> lk(bu.mu);
      bu.cv.wait(lk, [&] { return pred() || bu.give_up; });
      return !bu.give_up;
    };
    auto finish = [&](int rc) {
      if (rc) { ctx_rc[(size_t)g] = rc; ctx_err[(size_t)g] = strl_last_error(); }
      { std::lock_guard<std::mutex> lk(bu.mu); bu.created[(size_t)g] = 1; bu.done[(size_t)g] = 1; bu.runtime_up = true; }
      bu.cv.notify_all();
    };
    if (g > 0 && !wait_for([&] { return serial_ctx ? bu.created[(size_t)g - 1] != 0 : bu.runtime_up; })) return finish(0);
    const auto c0 = now();
    if (g == 0) {
      (void)strl_device_count();               // the runtime starts here, on one thread
      { std::lock_guard<std::mutex> lk(bu.mu); bu.runtime_up = true; }
      bu.cv.notify_all();
    }
    int rc = strl_ctx_create(device_of(g), &ctxs[(size_t)g]);
    bu.t_create[(size_t)g] = secs(c0, now());
    if (rc) return finish(rc);
    strl_ctx *c = ctxs[(size_t)g];
    if ((rc = strl_ctx_set_opts(c, &opts))) return finish(rc);
    if (G > 1 && (rc = strl_ctx_blocking_waits(c, 1))) return finish(rc);       // N feeding threads: waits sleep instead of spinning
    { std::lock_guard<std::mutex> lk(bu.mu); bu.created[(size_t)g] = 1; }       // (usable: created, options set)
    bu.cv.notify_all();
    if (!wait_for([&] { return bu.genome_ready; })) return finish(0);
    if ((rc = strl_ctx_set_genome(c, &bu.gs))) return finish(rc);
    if (!wait_for([&] { return bu.plan_ready; })) return finish(0);
    static const bool state_on_main = getenv("STRL_STATE_ON_MAIN") != nullptr;      // (diagnosis)
    if (state_on_main) return finish(0);
    const auto c1 = now();
    rc = strl_front_begin(c, bu.n_ref, bu.first_off[(size_t)g], bu.hint[(size_t)g]);
    if (!rc) rc = strl_front_reserve(c, (uint32_t)chunk_blocks, chunk_bytes);
    bu.t_state[(size_t)g] = secs(c1, now());
    finish(rc);
  };
  std::vector<std::thread> bring_threads;
  for (int g = 0; g < G; ++g) bring_threads.emplace_back(bring_up, g);
  // (an error exit -- quit() -- first tells the bring-up threads to give up where they wait, then waits for ONE background
  // thread: this one, which collects them all, so that no thread is inside the HIP runtime when the exit handlers run)
  std::thread reaper([&] { for (auto &t : bring_threads) t.join(); });
  g_bg_init = &reaper;
  g_bg_abort = [&bu] { { std::lock_guard<std::mutex> lk(bu.mu); bu.give_up = true; } bu.cv.notify_all(); };
  BgzfFeed feed;
  std::string err;
  const bool opened = feed.open(bam, err);
  if (!opened) quit("couldn't open bam");
  const double t_open = secs(t_start, now());
  // the genome table's host half (the BED read and flattened per tid; built from the FASTA on the first context when there is none)
  const auto tg0 = now();
  auto wait_created = [&](int g) {
    std::unique_lock<std::mutex> lk(bu.mu);
    bu.cv.wait(lk, [&] { return bu.created[(size_t)g] != 0; });
  };
  const Genome genome = genome_host_half(a, feed.targets(), [&]() -> strl_ctx * {
    wait_created(0);
    if (ctx_rc[0]) quit("[strling] %s (status %d)", ctx_err[0].c_str(), ctx_rc[0]);
    return ctxs[0];
  });
  {
    std::lock_guard<std::mutex> lk(bu.mu);
    bu.gs = strl_genome_str{(int32_t)feed.targets().size(), genome.has.data(), genome.off.data(), genome.st.data(), genome.en.data()};
    bu.genome_ready = true;
  }
  bu.cv.notify_all();
  const double t_genome = secs(tg0, now());
  const int32_t n_ref = (int32_t)feed.targets().size();
  // Reads the per-read state is sized for at the start (it grows geometrically beyond): a record of 150 bases with qualities
  // and a few tags takes 90 - 110 bytes of a level-6 BAM, 60 - 70 without qualities.  (The first sizing was file bytes / 48:
  // 124 GB of device memory and seconds of set-up for a 57 GB file of 5.4e8 reads.)
  // An index beside the file knows the count (the metadata pseudo-bins `samtools idxstats` reads): the state is then sized for
  // what the file holds -- a 30 x genome at 60 - 70 bytes a read is not under-sized and grown, the synthetic 57 GB file is not
  // over-sized by a fifth (71.6 -> 60.5 GB of device memory).  A stale index costs what a wrong guess costs: the state grows.
  static const char *env_hint = getenv("STRL_READS_HINT");
  uint64_t indexed = 0;
  // (a count of more records than the file has bytes is not believed: the state would be allocated for it)
  const bool have_count = !env_hint && !getenv("STRL_NO_INDEX_COUNT") && BgzfFeed::indexed_records(bam, indexed) && indexed > 0 && indexed <= feed.file_bytes();
  const uint64_t reads_hint = env_hint     ? strtoull(env_hint, nullptr, 10)
                              : have_count ? (G > 1 ? (indexed + indexed / 20) / (uint64_t)G + 65536 : indexed + indexed / 256 + 65536)
                                           : feed.file_bytes() / 88 / (size_t)G;
  if (verbose && have_count) fprintf(stderr, "[strling] %llu records by the index's counts\n", (unsigned long long)indexed);
  // Shares of the file, one per context: [cut[g], cut[g + 1]) in virtual offsets, every cut a record start the .bai names
  // (the one nearest to g / G of the bytes behind the header).  extract.nim:308-329 is one loop over the file in file order;
  // the shares are gathered in that order afterwards (strl_ctxs_extract_gather), so nothing downstream can tell.
  std::vector<uint64_t> cut;
  if (use_shares) {
    cut = share_cuts(feed, bam, G);
    if (cut.size() < 2) {
      use_shares = false;
      if (verbose) fprintf(stderr, "[strling] no .bai record starts to cut the file at: its chunks go over the contexts in turn\n");
    }
  }
  const int n_shares = use_shares ? (int)cut.size() : 0;
  {
    std::lock_guard<std::mutex> lk(bu.mu);
    bu.n_ref = n_ref;
    for (int g = 0; g < G; ++g) {
      // (shares: the first context is sized for the whole file -- the other shares' per-read state is appended to its own in the end)
      bu.first_off[(size_t)g] = use_shares && (size_t)g < cut.size() ? (cut[(size_t)g] & 0xffff) : feed.first_record_offset();
      bu.hint[(size_t)g] = use_shares && g == 0 ? reads_hint * (uint64_t)G : reads_hint;
    }
    bu.plan_ready = true;
  }
  bu.cv.notify_all();
  // everything below needs the contexts' buffers; the threads that feed a share only wait for THEIR context (below), the
  // one-context loop for the one there is
  auto wait_done = [&](int g) {
    std::unique_lock<std::mutex> lk(bu.mu);
    bu.cv.wait(lk, [&] { return bu.done[(size_t)g] != 0; });
  };
  bool brought_up = false;
  double t_begin = 0;
  auto bring_up_finish = [&]() -> int {        // -> 0, or EXTRACT_AGAIN_ON_HOST
    if (brought_up) return 0;
    brought_up = true;
    const auto tb0 = now();
    for (int g = 0; g < G; ++g) wait_done(g);
    if (reaper.joinable()) reaper.join();
    pin_thread.join();
    g_bg_init = nullptr; g_bg_abort = nullptr;
    if (getenv("STRL_STATE_ON_MAIN"))
      for (int g = 0; g < G && !ctx_rc[(size_t)g]; ++g) {
        const auto c1 = now();
        int rc = strl_front_begin(ctxs[(size_t)g], bu.n_ref, bu.first_off[(size_t)g], bu.hint[(size_t)g]);
        if (!rc) rc = strl_front_reserve(ctxs[(size_t)g], (uint32_t)chunk_blocks, chunk_bytes);
        if (rc) { ctx_rc[(size_t)g] = rc; ctx_err[(size_t)g] = strl_last_error(); }
        bu.t_state[(size_t)g] = secs(c1, now());
      }
    t_begin = secs(tb0, now());
    for (double v : bu.t_create) t_ctx = std::max(t_ctx, v);
    bool nomem = false;
    for (int g = 0; g < G; ++g) if (ctx_rc[(size_t)g] == STRL_ERR_NOMEM) nomem = true;
    if (nomem) {
      for (int g = 0; g < G; ++g) if (ctx_rc[(size_t)g] == STRL_ERR_NOMEM) { fprintf(stderr, "[strling] %s: repeating the extraction with the host pair logic\n", ctx_err[(size_t)g].c_str()); break; }
      for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
      for (strl_ctx *cc : ctxs) if (cc) strl_ctx_destroy(cc);
      return EXTRACT_AGAIN_ON_HOST;
    }
    for (int g = 0; g < G; ++g) if (ctx_rc[(size_t)g]) quit("[strling] %s (status %d)", ctx_err[(size_t)g].c_str(), ctx_rc[(size_t)g]);
    for (size_t k = 0; k < pin.size(); ++k) if (!pin[k] || !pin_meta[k]) quit("[strling] could not allocate page-locked memory");
    return 0;
  };
  { const int br = bring_up_finish(); if (br) return br; }
  strl_ctx *ctx = ctxs[0];
  if (verbose) {
    std::string devs;
    for (int g = 0; g < G; ++g) devs += (g ? " " : "") + std::to_string(device_of(g));
    fprintf(stderr, "[strling] %d context(s) on device(s) %s of %d\n", G, devs.c_str(), std::max(1, strl_device_count()));
  }

  fprintf(stderr, "[strling] collecting str-like reads\n");
  const auto t0 = now();
  // threads that read the compressed bytes into the page-locked buffers: 12 keep one device fed (0.8 - 1.1 s per 57 GB beside a
  // 3 s loop); several devices take what the CPU quota gives
  ThreadPool copy_pool(std::min(decode_threads(), G > 1 ? 48 : 12));
  std::vector<BgzfFeed::Block> blks;
  int64_t nreads = 0, n_tail = 0, tail_primary = 0;
  uint64_t n_seen = 0, slow_segments = 0;
  double t_walk = 0, t_copy = 0, t_push = 0, t_stage_wait = 0;
  // summaries arrive per context in the order of ITS chunks; the file order is what counts
  std::vector<strl_front_chunk> summary;                 // by chunk of the file
  std::vector<uint32_t> chunk_owner;
  std::vector<std::vector<uint64_t>> waiting((size_t)G);  // per context: its chunks without a summary yet
  std::vector<size_t> waiting_at((size_t)G, 0);
  uint64_t accounted = 0;
  auto got = [&](int g, const strl_front_chunk *done, int n_done) {
    for (int k = 0; k < n_done; ++k) summary[(size_t)waiting[(size_t)g][waiting_at[(size_t)g]++]] = done[k];
  };
  std::vector<uint8_t> have;
  auto account = [&] {           // the leading run of chunks whose summaries are in, in file order
    while (accounted < summary.size() && have[(size_t)accounted]) {
      const strl_front_chunk &d = summary[(size_t)accounted++];
      nreads += (int64_t)d.n_primary;
      n_seen += d.n_records;
      slow_segments += d.scan_slow_segments;
      if (d.last_placed >= 0) { n_tail = (int64_t)d.n_records - 1 - d.last_placed; tail_primary = (int64_t)d.tail_primary; }
      else { n_tail += (int64_t)d.n_records; tail_primary += (int64_t)d.n_primary; }
      if (verbose) fprintf(stderr, "%lld %.1f reads/sec\n", (long long)nreads, (double)nreads / std::max(secs(t0, now()), 1e-9));
    }
  };
  auto mark = [&](int g, size_t before) { for (size_t k = before; k < waiting_at[(size_t)g]; ++k) have[(size_t)waiting[(size_t)g][k]] = 1; };
  std::vector<uint64_t> pushes((size_t)G, 0);
  std::thread frag_thread, ahead;          // ahead: reads the next chunk's bytes (below)
  // a file the device front end refuses (STRL_ERR_FORMAT) goes to the host reader instead of ending the run
  auto give_up_front = [&]() -> int {
    fprintf(stderr, "[strling] %s: repeating the extraction with the host reader\n", strl_last_error());
    if (ahead.joinable()) ahead.join();
    if (frag_thread.joinable()) frag_thread.join();
    for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
    for (strl_ctx *c : ctxs) strl_ctx_destroy(c);
    return EXTRACT_AGAIN_HOST_FRONT;
  };
  // more records than one device pass takes (2^31 - 16; the reference has no cap, extract.nim:308), or more than the device's
  // memory holds the per-read state of (STRL_ERR_NOMEM: ~130 B per read): the streaming host Cache, which keeps nothing per
  // read on the device
  auto over_limit = [&]() -> int {
    fprintf(stderr, "[strling] %s: repeating the extraction with the host pair logic\n", strl_last_error());
    if (ahead.joinable()) ahead.join();
    if (frag_thread.joinable()) frag_thread.join();
    for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
    for (strl_ctx *c : ctxs) strl_ctx_destroy(c);
    return EXTRACT_AGAIN_ON_HOST;
  };
#define FRONT_CHECK(call)                                                                    \
  do {                                                                                       \
    const int rc__ = (call);                                                                 \
    if (rc__ == STRL_ERR_FORMAT) return give_up_front();                                     \
    if (rc__ == STRL_ERR_LIMIT || rc__ == STRL_ERR_NOMEM) return over_limit();              \
    if (rc__ == STRL_ERR_CRC) quit("[strling] error reading %s: %s", bam.c_str(), strl_last_error());   \
    if (rc__ != STRL_OK) quit("[strling] %s (status %d)", strl_last_error(), rc__);          \
  } while (0)
  // fragment_length_distribution (utils.nim:86-111, extract.nim:281) from the flag / isize words the parse kept of every record.
  // It needs the first ~2.1 M records only: as soon as they are parsed their words are copied out (behind the parse, no wait)
  // and a thread makes the histogram beside the rest of the file.
  struct FragState {
    uint32_t frag[4096];
    std::vector<int32_t> skipped;
    int64_t counted = 0;
    uint64_t next = 0;         // first record not looked at yet
    bool done = false;
  } fs;
  memset(fs.frag, 0, sizeof fs.frag);
  auto frag_feed = [&fs](const uint32_t *fw, uint64_t first, uint64_t n) {
    const int64_t n_reads = 2000000, skip_reads = 100000;
    for (uint64_t k = 0; k < n && !fs.done; ++k) {
      const int64_t i = (int64_t)(first + k);
      const uint32_t f = fw[k] & 0xffffu, is = fw[k] >> 16;
      if (!(f & 0x2)) continue;
      if (f & (0x800 | 0x100)) continue;
      if (is > 4095u) continue;
      if (i < skip_reads) { fs.skipped.push_back((int32_t)is); continue; }
      fs.skipped.clear();
      fs.frag[is]++;
      if (++fs.counted > n_reads) fs.done = true;
    }
    fs.next = first + n;
  };
  // Two chunks ahead: while chunk k is handed to the device (strl_front_collect returns when chunk k-1 has been inflated and
  // scanned), a thread walks the headers of chunk k+2 and reads its bytes, and chunk k+1 -- read during the previous turn --
  // has its copy to the device queued first thing.  (Reading only after the push had returned put the read AND the copy
  // between "chunk k-1 done" and "chunk k+1 may start": 0.40 s of 3.03 s with the inflate stream idle on the 57 GB file;
  // reading one ahead and queueing the copy when the read was done still left it 0.3 s late in all: a 323 MB chunk takes
  // 5 ms to read and 6.5 ms to copy, an inflate 12.)
  struct Staged {
    int64_t nb = 0;
    size_t lo = 0, hi = 0, slot = 0;
    int g = 0;
    std::string err;
    bool short_read = false;
  };
  Staged ring[3];                 // chunk c in ring[c % 3]: handed over | copy queued | being read
  std::vector<uint64_t> staged((size_t)G, 0);
  auto stage = [&](uint64_t ci, Staged &S) {
    const auto ta = now();
    S = Staged{};
    // a short first chunk gets the device going while the second is being copied
    S.nb = feed.next(blks, ci == 0 ? std::min<size_t>(chunk_blocks, 2048) : chunk_blocks, chunk_bytes, S.err);
    if (S.nb <= 0) return;
    const auto tb = now();
    S.g = (int)(ci % (uint64_t)G);
    S.lo = blks.front().c_off; S.hi = blks.back().c_off + blks.back().clen;
    S.slot = RING * (size_t)S.g + (size_t)(staged[(size_t)S.g]++ % RING);
    uint8_t *dst = pin[S.slot];
    const size_t lo = S.lo, hi = S.hi, piece = (size_t)4 << 20, pieces = (hi - lo + piece - 1) / piece;
    std::atomic<int> short_reads{0};
    copy_pool.parallel_for(pieces, [&](size_t k) { if (!feed.copy_at(dst + k * piece, lo + k * piece, std::min(piece, hi - lo - k * piece))) ++short_reads; });
    feed.done_with(lo, hi - lo);        // (once per chunk, by this thread: per 4 MB piece it was a round of TLB shoot-downs per piece on every copying CPU)
    S.short_read = short_reads.load() != 0;
    uint64_t *coff = reinterpret_cast<uint64_t *>(pin_meta[S.slot]);
    uint32_t *clen = reinterpret_cast<uint32_t *>(coff + chunk_blocks), *isz = clen + chunk_blocks, *crc = isz + chunk_blocks;
    for (size_t k = 0; k < (size_t)S.nb; ++k) { coff[k] = blks[k].c_off - lo; clen[k] = blks[k].clen; isz[k] = blks[k].isize; crc[k] = blks[k].crc; }
    t_walk += secs(ta, tb); t_copy += secs(tb, now());
  };
  auto queue_copy = [&](const Staged &S) -> int {     // the chunk's copy to the device, ahead of its turn
    uint64_t *ncoff = reinterpret_cast<uint64_t *>(pin_meta[S.slot]);
    uint32_t *nclen = reinterpret_cast<uint32_t *>(ncoff + chunk_blocks), *nisz = nclen + chunk_blocks, *ncrc = nisz + chunk_blocks;
    return strl_front_stage(ctxs[(size_t)S.g], pin[S.slot], S.hi - S.lo, ncoff, nclen, nisz, ncrc, (uint32_t)S.nb);
  };
  auto tf = now();
  if (use_shares) {
    // One feeding thread per share, each with its own header walker, its own ring of page-locked buffers, its own copy
    // threads and its own context: the pipeline of the one-GPU loop below, N times side by side.  Nothing is carried from a
    // share to the next; what makes that exact is checked when the shares are done (a share must end exactly where the
    // next begins).
    feed.halt();
    struct Share {
      BgzfFeed fd;
      std::vector<strl_front_chunk> sums;
      int rc = 0;
      bool fallback = false;
      std::string err;
      uint32_t tail = 0;
      double t_walk = 0, t_copy = 0, t_wait = 0, t_all = 0;
      uint64_t bytes = 0, chunks = 0;
    };
    std::vector<Share> shares((size_t)n_shares);
    for (int g = 0; g < n_shares; ++g) {
      const uint64_t end = g + 1 < n_shares ? cut[(size_t)g + 1] : 0;
      if (!shares[(size_t)g].fd.open_share(feed, cut[(size_t)g] >> 16, (uint32_t)(cut[(size_t)g] & 0xffff), end >> 16, (uint32_t)(end & 0xffff), err)) {
        // an offset of the index that is no block of THIS file (a stale .bai): like every later check, the way out is the
        // chunk-by-chunk run, which needs no index -- `--gpus 1` reads the file fine
        fprintf(stderr, "[strling] share %d of %d: %s; repeating the extraction chunk by chunk\n", g, n_shares, err.c_str());
        for (Share &Z : shares) Z.fd.close();
        for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
        for (strl_ctx *c : ctxs) strl_ctx_destroy(c);
        return EXTRACT_AGAIN_BY_CHUNKS;
      }
    }
    static const bool feed_only = getenv("STRL_FEED_ONLY") != nullptr;     // measurement: the host side alone, no device stage
    const int per_share = std::max(2, std::min(12, decode_threads() / n_shares));
    auto feeder = [&](int g) {
      Share &Z = shares[(size_t)g];
      strl_ctx *cx = ctxs[(size_t)g];
      const auto z0 = now();
      ThreadPool pool(per_share);
      std::vector<BgzfFeed::Block> bl;
      struct St { int64_t nb = 0; size_t lo = 0, hi = 0, slot = 0; bool last = false, short_read = false; std::string err; };
      St ring[3];
      uint64_t staged = 0;
      auto stage = [&](uint64_t ci, St &S) {
        const auto ta = now();
        S = St{};
        S.nb = Z.fd.next(bl, ci == 0 ? std::min<size_t>(chunk_blocks, 2048) : chunk_blocks, chunk_bytes, S.err, &S.last);
        if (S.nb <= 0) return;
        const auto tb = now();
        S.lo = bl.front().c_off; S.hi = bl.back().c_off + bl.back().clen;
        S.slot = RING * (size_t)g + (size_t)(staged++ % RING);
        uint8_t *dst = pin[S.slot];
        const size_t lo = S.lo, hi = S.hi, piece = (size_t)4 << 20, pieces = (hi - lo + piece - 1) / piece;
        std::atomic<int> short_reads{0};
        pool.parallel_for(pieces, [&](size_t k) { if (!Z.fd.copy_at(dst + k * piece, lo + k * piece, std::min(piece, hi - lo - k * piece))) ++short_reads; });
        Z.fd.done_with(lo, hi - lo);
        S.short_read = short_reads.load() != 0;
        uint64_t *coff = reinterpret_cast<uint64_t *>(pin_meta[S.slot]);
        uint32_t *clen = reinterpret_cast<uint32_t *>(coff + chunk_blocks), *isz = clen + chunk_blocks, *crc = isz + chunk_blocks;
        for (size_t k = 0; k < (size_t)S.nb; ++k) { coff[k] = bl[k].c_off - lo; clen[k] = bl[k].clen; isz[k] = bl[k].isize; crc[k] = bl[k].crc; }
        Z.t_walk += secs(ta, tb); Z.t_copy += secs(tb, now());
        Z.bytes += hi - lo; ++Z.chunks;
      };
      auto fail = [&](int rc) { Z.rc = rc; Z.err = strl_last_error(); };
      auto tables = [&](const St &S, uint64_t *&coff, uint32_t *&clen, uint32_t *&isz, uint32_t *&crc) {
        coff = reinterpret_cast<uint64_t *>(pin_meta[S.slot]);
        clen = reinterpret_cast<uint32_t *>(coff + chunk_blocks); isz = clen + chunk_blocks; crc = isz + chunk_blocks;
      };
      auto queue_copy = [&](const St &S) -> int {     // the chunk's copy to the device, ahead of its turn
        uint64_t *coff; uint32_t *clen, *isz, *crc;
        tables(S, coff, clen, isz, crc);
        int rc = S.last && Z.fd.tail_trim() ? strl_front_trim_next(cx, Z.fd.tail_trim()) : 0;
        return rc ? rc : strl_front_stage(cx, pin[S.slot], S.hi - S.lo, coff, clen, isz, crc, (uint32_t)S.nb);
      };
      std::thread ahead;
      int rc = 0;
      stage(0, ring[0]);
      if (!feed_only && ring[0].nb > 0 && !ring[0].short_read && (rc = queue_copy(ring[0]))) { fail(rc); return; }
      if (ring[0].nb > 0) stage(1, ring[1]);
      for (uint64_t ci = 0;; ++ci) {
        const St cur = ring[ci % 3];
        St &nxt = ring[(ci + 1) % 3];
        if (cur.nb < 0) { Z.fallback = true; Z.err = cur.err; break; }     // (a damaged file says so again in the chunk-by-chunk run)
        if (cur.nb == 0) break;
        if (cur.short_read) { Z.rc = STRL_ERR_ARG; Z.err = "short read"; break; }
        if (nxt.nb > 0) ahead = std::thread([&, ci] { stage(ci + 2, ring[(ci + 2) % 3]); });
        else ring[(ci + 2) % 3] = St{};
        if (!feed_only) {
          if (nxt.nb > 0 && !nxt.short_read && (rc = queue_copy(nxt))) { fail(rc); break; }
          uint64_t *coff; uint32_t *clen, *isz, *crc;
          tables(cur, coff, clen, isz, crc);
          strl_front_chunk done[2];
          int n_done = 0;
          const auto tc = now();
          if ((rc = strl_front_enqueue_after(cx, nullptr, pin[cur.slot], cur.hi - cur.lo, coff, clen, isz, crc, (uint32_t)cur.nb, done, &n_done))) { fail(rc); break; }
          for (int k = 0; k < n_done; ++k) Z.sums.push_back(done[k]);
          if ((rc = strl_front_collect(cx))) { fail(rc); break; }
          Z.t_wait += secs(tc, now());
        }
        if (ahead.joinable()) ahead.join();
      }
      if (ahead.joinable()) ahead.join();
      if (!feed_only && !Z.rc && !Z.fallback) {
        strl_front_chunk done[2];
        int n_done = 0;
        if ((rc = strl_front_finish(cx, done, &n_done))) fail(rc);
        else {
          for (int k = 0; k < n_done; ++k) Z.sums.push_back(done[k]);
          if ((rc = strl_front_tail_bytes(cx, &Z.tail))) fail(rc);
        }
      }
      Z.t_all = secs(z0, now());
    };
    std::vector<std::thread> feeders;
    for (int g = 1; g < n_shares; ++g) feeders.emplace_back(feeder, g);
    feeder(0);
    for (auto &t : feeders) t.join();
    tf = now();
    if (feed_only) {
      uint64_t bytes = 0;
      for (const Share &Z : shares) bytes += Z.bytes;
      const double dt = secs(t0, now());
      fprintf(stderr, "[strling] feed only: %d shares, %d copy threads each (+ walker, + read-ahead), %.3f s, %.2f GB of BAM, %.2f GB/s\n", n_shares, per_share, dt, (double)bytes / 1e9, (double)bytes / 1e9 / dt);
      for (int g = 0; g < n_shares; ++g)
        fprintf(stderr, "[strling]   share %d: %.2f GB in %llu chunks, %.3f s (block headers %.3f, reading compressed bytes %.3f)\n", g, (double)shares[(size_t)g].bytes / 1e9,
                (unsigned long long)shares[(size_t)g].chunks, shares[(size_t)g].t_all, shares[(size_t)g].t_walk, shares[(size_t)g].t_copy);
      fprintf(stderr, "[strling] feed only: waited %.3f s for the bring-up threads (slowest context %.3f, page-locked buffers %.3f); main() entered %.2f s after exec, now %.2f s after exec\n", t_begin, t_ctx,
              t_pin, g_main_at, since_exec());
      fflush(stderr);
      _exit(0);
    }
    bool again = false, nomem = false;
    for (int g = 0; g < n_shares; ++g) {
      const Share &Z = shares[(size_t)g];
      if (Z.rc == STRL_ERR_NOMEM) nomem = true;
      if (Z.rc == STRL_ERR_FORMAT || Z.rc == STRL_ERR_LIMIT || Z.rc == STRL_ERR_NOMEM || Z.fallback || (g + 1 < n_shares && Z.tail)) again = true;
      else if (Z.rc == STRL_ERR_CRC) quit("[strling] error reading %s: %s", bam.c_str(), Z.err.c_str());
      else if (Z.rc) quit("[strling] %s (status %d)", Z.err.c_str(), Z.rc);
    }
    if (again) {
      // a share that did not end on the next one's first record (a stale index), a block or record the device front end
      // refuses, too many records for one pass: the chunk-by-chunk run sorts out which, with its own ways out
      for (int g = 0; g < n_shares; ++g)
        if (shares[(size_t)g].rc || shares[(size_t)g].fallback || (g + 1 < n_shares && shares[(size_t)g].tail))
          fprintf(stderr, "[strling] share %d of %d: %s; repeating the extraction chunk by chunk\n", g, n_shares,
                  shares[(size_t)g].rc || shares[(size_t)g].fallback ? shares[(size_t)g].err.c_str() : "the .bai's record start is not where the share's records end");
      for (Share &Z : shares) Z.fd.close();
      for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
      for (strl_ctx *c : ctxs) strl_ctx_destroy(c);
      return nomem ? EXTRACT_AGAIN_ON_HOST : EXTRACT_AGAIN_BY_CHUNKS;      // (out of device memory: the chunk-by-chunk run would need as much)
    }
    for (int g = 0; g < n_shares; ++g)
      for (const strl_front_chunk &d : shares[(size_t)g].sums) { summary.push_back(d); have.push_back(1); chunk_owner.push_back((uint32_t)g); }
    account();
    for (int g = 0; g < n_shares; ++g) { t_walk += shares[(size_t)g].t_walk; t_copy += shares[(size_t)g].t_copy; t_push += shares[(size_t)g].t_wait; }
    if (verbose)
      for (int g = 0; g < n_shares; ++g)
        fprintf(stderr, "[strling] share %d: %.2f GB of BAM from offset %llu, %zu chunks, %.3f s (block headers %.3f, reading compressed bytes %.3f, enqueueing + waiting for the device %.3f)\n", g,
                (double)shares[(size_t)g].bytes / 1e9, (unsigned long long)(cut[(size_t)g] >> 16), shares[(size_t)g].sums.size(), shares[(size_t)g].t_all, shares[(size_t)g].t_walk,
                shares[(size_t)g].t_copy, shares[(size_t)g].t_wait);
  } else {
  stage(0, ring[0]);
  if (ring[0].nb > 0 && !ring[0].short_read) FRONT_CHECK(queue_copy(ring[0]));
  if (ring[0].nb > 0) stage(1, ring[1]);
  for (uint64_t ci = 0;; ++ci) {
    const Staged cur = ring[ci % 3];
    Staged &nxt = ring[(ci + 1) % 3];
    const int64_t nb = cur.nb;
    if (nb < 0) quit("[strling] error reading %s: %s", bam.c_str(), cur.err.c_str());
    if (nb == 0) break;
    if (cur.short_read) quit("[strling] error reading %s: short read", bam.c_str());
    const int g = cur.g;
    const size_t lo = cur.lo, hi = cur.hi;
    uint8_t *dst = pin[cur.slot];
    uint64_t *coff = reinterpret_cast<uint64_t *>(pin_meta[cur.slot]);
    uint32_t *clen = reinterpret_cast<uint32_t *>(coff + chunk_blocks), *isz = clen + chunk_blocks, *crc = isz + chunk_blocks;
    if (nxt.nb > 0) ahead = std::thread([&, ci] { stage(ci + 2, ring[(ci + 2) % 3]); });
    else ring[(ci + 2) % 3] = Staged{};
    if (nxt.nb > 0 && !nxt.short_read) FRONT_CHECK(queue_copy(nxt));
    const auto tc = now();
    strl_front_chunk done[2];
    int n_done = 0;
    summary.push_back(strl_front_chunk{});
    have.push_back(0);
    chunk_owner.push_back((uint32_t)g);
    waiting[(size_t)g].push_back(ci);
    const size_t before = waiting_at[(size_t)g];
    FRONT_CHECK(strl_front_enqueue_after(ctxs[(size_t)g], ci ? ctxs[(size_t)((ci - 1) % (uint64_t)G)] : nullptr, dst, hi - lo, coff, clen, isz, crc, (uint32_t)nb, done, &n_done));
    ++pushes[(size_t)g];
    got(g, done, n_done);
    mark(g, before);
    FRONT_CHECK(strl_front_collect(ctxs[(size_t)g]));
    account();
    if (G == 1 && !frag_thread.joinable()) {
      uint64_t parsed = 0;
      CHECK(strl_front_records(ctx, &parsed));
      if (parsed >= early_n) {
        void *ev = nullptr;
        if (fw_early) {
          CHECK(strl_front_fragwords_async(ctx, 0, early_n, fw_early, &ev));
          frag_thread = std::thread([&, ev] { if (strl_event_wait(ev) == STRL_OK) frag_feed(fw_early, 0, early_n); });
        }
      }
    }
    const auto td0 = now();
    if (ahead.joinable()) ahead.join();
    t_stage_wait += secs(td0, now());
    t_push += secs(tc, now());
  }
  if (frag_thread.joinable()) frag_thread.join();
  tf = now();
  for (int g = 0; g < G; ++g) {
    strl_front_chunk done[2];
    int n_done = 0;
    const size_t before = waiting_at[(size_t)g];
    FRONT_CHECK(strl_front_finish(ctxs[(size_t)g], done, &n_done));
    got(g, done, n_done);
    mark(g, before);
  }
  account();
  }   // (chunk by chunk)
  const auto tf2 = now();
  if (G > 1) {
    std::vector<uint64_t> recs(summary.size());
    for (size_t k = 0; k < summary.size(); ++k) recs[k] = summary[k].n_records;
    FRONT_CHECK(strl_ctxs_extract_gather(ctxs.data(), G, chunk_owner.data(), recs.data(), recs.size()));
    if (verbose) fprintf(stderr, "[strling] %zu chunks over %d contexts on %d device(s)%s; per-read state gathered on the first in %.3f s\n", summary.size(), G, std::min(G, std::max(1, strl_device_count())),
                         use_shares ? ", a contiguous share of the file each" : " in turn", secs(tf2, now()));
  }
  const double t_drain = secs(tf, now());
  {   // extract.nim:310-313: one line per large contig that has reads (here: once the whole file has been through)
    std::vector<uint8_t> seen((size_t)n_ref + 1, 0);
    CHECK(strl_front_tids(ctx, seen.data(), n_ref));
    for (int32_t t = 0; t < n_ref; ++t)
      if (seen[(size_t)t] && feed.targets()[(size_t)t].length > 2000000u) fprintf(stderr, "[strling] extracting chromosome:%s\n", feed.targets()[(size_t)t].name.c_str());
  }
  fprintf(stderr, "[strling] extracting unmapped reads\n");
  nreads += tail_primary;   // the "*" region is counted a second time by the reference's progress counter (extract.nim:326-329)
  // the rest of the fragment-length pass, if the early part did not finish it (few proper pairs, small files)
  const auto tq = now();
  double t_frag_copy = 0;
  {
    uint32_t *fw = reinterpret_cast<uint32_t *>(pin[0]);          // (page-locked: the copy needs no staging)
    const uint64_t fw_cap = chunk_bytes / 4;
    while (!fs.done && fs.next < n_seen) {
      const uint64_t first = fs.next, m = std::min<uint64_t>({n_seen - first, fw_cap, (uint64_t)8000000});
      const auto tw0 = now();
      CHECK(strl_front_fragwords(ctx, first, m, fw));
      t_frag_copy += secs(tw0, now());
      frag_feed(fw, first, m);
    }
    uint64_t sum = 0;
    for (int k = 0; k < 4096; ++k) sum += fs.frag[k];
    if ((uint32_t)sum == 0) {
      fprintf(stderr, "using first reads in fragment_length_distribution calculation as there were not enough\n");
      for (int32_t is : fs.skipped) fs.frag[is]++;
    }
  }
  uint32_t *frag = fs.frag;
  const int frag_median = strl_frag_median(frag, 0.5);
  if (verbose) {
    fprintf(stderr, "Calculated median fragment length:%d\n", frag_median);
    fprintf(stderr, "10th, 90th percentile of fragment length:%d %d\n", strl_frag_median(frag, 0.1), strl_frag_median(frag, 0.9));
  }
  opts.median_fragment_length = frag_median;
  const auto ts0 = now();
  CHECK(strl_ctx_set_opts(ctx, &opts));
  const double t_setopts = secs(ts0, now());
  const double t_frag = secs(tq, now());
  const auto tp0 = now();
  uint64_t nt = 0;
  int rc = 0;
  double t_finish = 0;
  auto cap_of = [](uint64_t v) { return std::min<uint64_t>(v, 0x7ffffff0ull); };
  for (int attempt = 0; attempt < 2; ++attempt) {
    rc = strl_extract_finish(ctx, n_tail, attempt ? cap_of(3 * n_seen + 16) : 0, attempt ? cap_of(8 * n_seen + 16) : 0);
    if (rc == STRL_ERR_NOMEM) break;
    CHECK(rc);
    rc = strl_treads_fetch(ctx, nullptr, 0, &nt, nullptr);
    if (rc != STRL_ERR_CAPACITY) break;
  }
  t_finish = secs(tp0, now());
  if (rc == STRL_ERR_FORMAT || rc == STRL_ERR_NOMEM) {
    fprintf(stderr, "[strling] %s: repeating the extraction with the host pair logic\n", strl_last_error());
    for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
    for (strl_ctx *c : ctxs) strl_ctx_destroy(c);
    return EXTRACT_AGAIN_ON_HOST;
  }
  if (rc) quit("[strling] %s (status %d)", strl_last_error(), rc);
  // treads + names through the page-locked chunk buffers (free now): no staging copies; vectors when they are too small
  std::vector<strl_tread> tv;
  std::vector<uint64_t> qv;
  std::string nv;
  strl_tread *treads_p = reinterpret_cast<strl_tread *>(pin[0]);
  uint64_t *qoff_p = reinterpret_cast<uint64_t *>(pin[1]);
  const uint64_t names_room = chunk_bytes > (nt + 1) * 8 + 4096 ? chunk_bytes - (nt + 1) * 8 - 64 : 0;
  char *names_p = reinterpret_cast<char *>(pin[1]) + (nt + 1) * 8 + 64;
  // (names: the room behind the offsets serves unless the names average more than 24 bytes -- then the call reports what it
  // needs and is repeated with vectors; sizing for the 255-byte maximum made a whole genome's 8e6 treads zero-fill 2 GB)
  if ((nt + 1) * sizeof(strl_tread) > chunk_bytes || names_room < nt * 24) {
    tv.resize((size_t)nt + 1); qv.resize((size_t)nt + 1);
    nv.resize((size_t)std::min<uint64_t>(nt * 255, std::max<uint64_t>(nt * 32, 1 << 20)) + 16);
    treads_p = tv.data(); qoff_p = qv.data(); names_p = &nv[0];
  }
  uint64_t need = 0, ngot = 0;
  rc = strl_front_treads_named(ctx, treads_p, nt + 1, &ngot, qoff_p, names_p, treads_p == tv.data() ? nv.size() : names_room, &need);
  if (rc == STRL_ERR_CAPACITY) {       // names longer than the room given: plain vectors of the size the call reported
    tv.resize((size_t)nt + 1); qv.resize((size_t)nt + 1); nv.resize((size_t)need + 16);
    treads_p = tv.data(); qoff_p = qv.data(); names_p = &nv[0];
    rc = strl_front_treads_named(ctx, treads_p, nt + 1, &ngot, qoff_p, names_p, nv.size(), &need);
  }
  if (rc) quit("[strling] %s (status %d)", strl_last_error(), rc);
  {   // (the .bin writer wants the name's index, not the record's: 8e6 strided stores, by the feed's idle copy threads)
    const uint64_t per = 1 << 17;
    copy_pool.parallel_for((size_t)((nt + per - 1) / per), [&](size_t q) { for (uint64_t i = q * per, e = std::min(nt, i + per); i < e; ++i) treads_p[i].qname_id = (int64_t)i; });
  }
  const double t_pair = secs(tp0, now());
  fprintf(stderr, "[strling] writing binary file:%s\n", bin.c_str());
  const auto tw0 = now();
  // The rings the .bin is not written out of go back meanwhile: un-registering page-locked memory the DMA engines have used
  // is ~0.1 s per GB (profiles/r06/exit_where.log) whoever does it -- this thread beside the writer's, or the kernel behind
  // the process' last line, on the caller's clock.  (STRL_NO_EARLY_UNPIN=1: left to the exit.)
  std::thread early_unpin;
  if (!getenv("STRL_NO_EARLY_UNPIN"))
    early_unpin = std::thread([&] {
      for (size_t k = 0; k < pin.size(); ++k) {
        if (pin[k] && (void *)pin[k] != (void *)treads_p && (void *)pin[k] != (void *)qoff_p) { strl_pinned_free(pin[k]); pin[k] = nullptr; }
        if (pin_meta[k]) { strl_pinned_free(pin_meta[k]); pin_meta[k] = nullptr; }
      }
    });
  const int bin_rc = strl_bin_write(bin.c_str(), (float)p, min_mapq, frag, feed.header_text().data(), (int32_t)feed.header_text().size(), treads_p, nt, qoff_p, names_p);
  if (early_unpin.joinable()) early_unpin.join();
  CHECK(bin_rc);
  const double t_write = secs(tw0, now());
  fprintf(stderr, "[strling] finished extraction\n");
  if (verbose) {
    fprintf(stderr, "[strling] %lld reads, %llu STR reads, 0 reads still waiting for a mate\n", (long long)nreads, (unsigned long long)nt);
    fprintf(stderr, "[strling] seconds: total %.3f  (beside the device: block headers %.3f, reading compressed bytes %.3f)  enqueueing + waiting for the device %.3f  waiting for the next chunk's bytes %.3f  "
                    "draining the device %.3f  fragment lengths %.3f (copy %.3f, set_opts %.3f)  pair logic + names %.3f (pair logic over all reads + .bin order %.3f)  (device front end; %llu scan segments walked twice)\n",
            secs(t0, now()), t_walk, t_copy, t_push - t_stage_wait, t_stage_wait, t_drain, t_frag, t_frag_copy, t_setopts, t_pair, t_finish, (unsigned long long)slow_segments);
  }
  if (verbose) {
    uint64_t mf = 0, mt = 0;
    if (strl_ctx_mem_info(ctx, &mf, &mt) == STRL_OK)
      fprintf(stderr, "[strling] device memory in use at the end (all chunks' per-read state resident): %.2f GB of %.1f GB\n", (double)(mt - mf) / 1e9, (double)mt / 1e9);
  }
  if (verbose)
  {
    double t_state = 0;
    for (double v : bu.t_state) t_state = std::max(t_state, v);
    fprintf(stderr, "[strling] seconds before the loop: header walk %.3f, genome table (host half) %.3f, then waiting for the bring-up threads %.3f -- beside all that, a thread "
                    "per context: slowest context + its options %.3f, slowest per-read state for %llu reads + front-end buffers %.3f; page-locked buffers %.3f; writing the .bin %.3f; whole run %.3f\n",
            t_open, t_genome, t_begin, t_ctx, (unsigned long long)reads_hint, t_state, t_pin, t_write, secs(t_start, now()));
    fprintf(stderr, "[strling] process: main() entered %.2f s after exec, now %.2f s after exec (what the caller's clock adds behind this line is the kernel "
                    "reclaiming the process' device and page-locked memory)\n", g_main_at, since_exec());
  }
  // the process ends here: the driver reclaims device and page-locked memory faster than freeing them piece by piece would
  // (STRL_TEARDOWN=1 frees them explicitly)
  if (getenv("STRL_TEARDOWN")) {
    const auto td0 = now();
    for (size_t k = 0; k < pin.size(); ++k) { strl_pinned_free(pin[k]); strl_pinned_free(pin_meta[k]); }
    const auto td1 = now();
    for (strl_ctx *c : ctxs) strl_ctx_destroy(c);
    const auto td2 = now();
    feed.close();
    if (verbose) fprintf(stderr, "[strling] teardown by hand: page-locked buffers %.3f s, contexts (device memory, streams) %.3f s, the feed (mapping, threads) %.3f s; now %.2f s after exec\n",
                         secs(td0, td1), secs(td1, td2), secs(td2, now()), since_exec());
    return 0;
  }
  // ... and so does the runtime's own shutdown (~0.15 s with gigabytes of device memory mapped).  Under a profiler the normal
  // exit path is kept: its tool library writes its files from an exit handler.
  const char *pre = getenv("LD_PRELOAD");
  if ((pre && strstr(pre, "rocprof")) || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_LIBRARY_CTOR")) return 0;
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}

// The end of a `strling` process whose outputs are closed: the driver reclaims device and page-locked memory faster than the
// runtime's own shutdown does piece by piece (~0.15 s and more with gigabytes mapped).  Under a profiler the normal exit path is
// kept -- its tool library writes its files from an exit handler -- and STRL_TEARDOWN=1 asks for it explicitly.
static int end_process(strl_ctx *ctx) {
  const char *pre = getenv("LD_PRELOAD");
  if (getenv("STRL_TEARDOWN") || (pre && strstr(pre, "rocprof")) || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_LIBRARY_CTOR")) {
    if (ctx) strl_ctx_destroy(ctx);
    return 0;
  }
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}

// ---- loci given on the command line: cluster.nim:96-169, call.nim:160-183 --------------------------------------------------------
static int get_tid(const std::string &name, const std::vector<BamTarget> &targets) {      // utils.nim:214-218
  for (size_t t = 0; t < targets.size(); ++t) if (targets[t].name == name) return (int)t;
  return -1;
}
static void check_unit(const std::string &rep, const std::string &line) {
  for (char x : rep) if (!strchr("ATCG", x)) quit("Error reading loci bed file. Expected DNA (ATCG only) in the 4th field, and got an unexpected character on line: %s", line.c_str());
}
static std::vector<std::string> read_lines(const std::string &path) {
  std::vector<std::string> out;
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return out;
  std::string cur;
  int ch;
  while ((ch = fgetc(f)) != EOF) { if (ch == '\n') { if (!cur.empty() && cur.back() == '\r') cur.pop_back(); out.push_back(cur); cur.clear(); } else cur.push_back((char)ch); }
  if (!cur.empty()) out.push_back(cur);
  fclose(f);
  return out;
}
static strl_locus make_locus(const std::vector<std::string> &f, const std::vector<BamTarget> &targets, const std::string &line) {
  strl_locus L;
  memset(&L, 0, sizeof L);
  L.b.tid = get_tid(f[0], targets);
  if (L.b.tid < 0) quit("[strling] chromosome %s of a locus is not in the bam header: %s", f[0].c_str(), line.c_str());
  L.b.left = (uint32_t)atoll(f[1].c_str());
  L.b.right = (uint32_t)atoll(f[2].c_str());
  if (f[3].size() > 6) quit("ERROR: STRling currently only supports 1-6 bp repeat units. Input bed contains repeat unit length %zu\n%s", f[3].size(), line.c_str());
  memcpy(L.b.repeat, f[3].data(), f[3].size());
  check_unit(f[3], line);
  return L;
}
// parse_bed, cluster.nim:111-141
static std::vector<strl_locus> parse_bed(const std::string &path, const std::vector<BamTarget> &targets, uint32_t window) {
  std::vector<strl_locus> out;
  for (const std::string &line : read_lines(path)) {
    std::vector<std::string> f;
    for (size_t i = 0; i < line.size();) {
      while (i < line.size() && isspace((unsigned char)line[i])) ++i;
      size_t j = i;
      while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
      if (j > i) f.push_back(line.substr(i, j - i));
      i = j;
    }
    if (f.size() != 4 && f.size() != 5) quit("Error reading loci bed file. Expected 4 or 5 fields and got %zu on line: %s", f.size(), line.c_str());
    strl_locus L = make_locus(f, targets, line);
    if (f.size() == 5) snprintf(L.name, sizeof L.name, "%s", f[4].c_str());
    L.b.left_most = (uint32_t)std::max<int32_t>((int32_t)L.b.left - (int32_t)window, 0);
    L.b.right_most = std::min<uint32_t>(L.b.right + window, targets[(size_t)L.b.tid].length);
    if (!(L.b.left <= L.b.right) || !(L.b.left_most <= L.b.right_most)) quit("[strling] inverted locus (doAssert cluster.nim:133-134): %s", line.c_str());
    out.push_back(L);
  }
  return out;
}
// parse_bounds, cluster.nim:143-169
static std::vector<strl_locus> parse_bounds(const std::string &path, const std::vector<BamTarget> &targets) {
  std::vector<strl_locus> out;
  for (const std::string &line : read_lines(path)) {
    if (!line.empty() && line[0] == '#') continue;
    std::vector<std::string> f;
    size_t i = 0;
    for (;;) { const size_t j = line.find('\t', i); f.push_back(line.substr(i, j == std::string::npos ? std::string::npos : j - i)); if (j == std::string::npos) break; i = j + 1; }
    if (f.size() != 11) quit("Error reading loci bed file. Expected 11 fields and got %zu on line: %s", f.size(), line.c_str());
    strl_locus L = make_locus(f, targets, line);
    snprintf(L.name, sizeof L.name, "%s", f[4].c_str());
    L.b.left_most = (uint32_t)atoll(f[5].c_str()); L.b.right_most = (uint32_t)atoll(f[6].c_str()); L.b.center_mass = (uint32_t)atoll(f[7].c_str());
    L.b.n_left = (uint16_t)atoll(f[8].c_str()); L.b.n_right = (uint16_t)atoll(f[9].c_str()); L.b.n_total = (uint16_t)atoll(f[10].c_str());
    if (!(L.b.left <= L.b.right) || !(L.b.left_most <= L.b.right_most)) quit("[strling] inverted bounds (doAssert cluster.nim:162-163): %s", line.c_str());
    out.push_back(L);
  }
  return out;
}
static bool loci_overlap(const strl_bounds &a, const strl_bounds &b) {                 // cluster.nim:96-100
  return a.tid == b.tid && strncmp(a.repeat, b.repeat, 7) == 0 && std::max(a.left, b.left) <= std::min(a.right, b.right);
}
static int locus_row(char *buf, int cap, const strl_locus &L, const char *chrom) {     // cluster.nim:262-266 with a name
  const strl_bounds &b = L.b;
  return snprintf(buf, (size_t)cap, "%s\t%u\t%u\t%s\t%s\t%u\t%u\t%u\t%u\t%u\t%u", chrom, b.left, b.right, b.repeat, L.name, b.left_most, b.right_most,
                  b.center_mass, (unsigned)b.n_left, (unsigned)b.n_right, (unsigned)b.n_total);
}

// targets.fill(fasta), merge.nim:27-34: names and lengths of the FASTA's .fai in file order
static std::vector<BamTarget> targets_from_fai(const std::string &fasta) {
  FILE *f = fopen((fasta + ".fai").c_str(), "r");
  if (!f || !file_exists(fasta)) { if (f) fclose(f); quit("could not open fasta:%s", fasta.c_str()); }
  std::vector<BamTarget> t;
  char line[1 << 16];
  while (fgets(line, sizeof line, f)) {
    char name[4096];
    unsigned long long len = 0;
    if (sscanf(line, "%4095[^\t]\t%llu", name, &len) == 2) t.push_back(BamTarget{name, (uint32_t)len});
  }
  fclose(f);
  return t;
}

static int merge_main(int argc, char **argv) {
  const char *usage =
      "strling merge\n\nUsage:\n  strling merge [options] [bin ...]\n\nOptions:\n  -w, --window=WINDOW        Number of bp within which to search for reads supporting the other side of a bound. "
      "Estimated from the insert size distribution by default. (default: -1)\n  -m, --min-support=MIN_SUPPORT\n"
      "                             minimum number of supporting reads required in at least one individual for a locus to be reported (default: 5)\n"
      "  -c, --min-clip=MIN_CLIP    minimum number of supporting clipped reads for each side of a locus (default: 0)\n"
      "  -t, --min-clip-total=MIN_CLIP_TOTAL\n                             minimum total number of supporting clipped reads for a locus (default: 0)\n"
      "  -q, --min-mapq=MIN_MAPQ    minimum mapping quality (does not apply to STR reads) (default: 40)\n"
      "  -o, --output-prefix=OUTPUT_PREFIX\n                             prefix for output files. Suffix will be -bounds.txt (default: strling)\n"
      "  -f, --fasta=FASTA          path to fasta file (required if using CRAM input)\n"
      "  --chromosome=CHROMOSOME    chromosome to restrict parsing. helps with memory/parallelization for large cohorts (default: -2)\n"
      "  -l, --bed=BED              Annoated bed file specifying additional STR loci to genotype. Format is: chr start stop repeatunit [name]\n"
      "  -d, --diff-refs            allow bin files generated on a mixture of reference genomes (by default differing references will produce an error). "
      "Reports chromosomes in the first bin or -f if provided\n"
      "  --gpus=N                   cluster on N GPUs: the samples' reads shard over them, RCCL all-gather, every GPU clusters the (chromosome, repeat unit) "
      "groups it owns (default: 1)\n  -v, --verbose\n  -h, --help                 Show this help\n";
  if (argc <= 2) { fputs(usage, stdout); return 0; }
  const Args a = parse(argc, argv, 2, {{"fasta", 'f', true}, {"window", 'w', true}, {"min-support", 'm', true}, {"chromosome", 'C', true},
                                       {"min-clip", 'c', true}, {"min-clip-total", 't', true}, {"min-mapq", 'q', true}, {"bed", 'l', true},
                                       {"output-prefix", 'o', true}, {"diff-refs", 'd', false}, {"verbose", 'v', false}, {"gpus", 'G', true}, {"device", 'D', true}}, usage);
  set_device0(a.get("device", ""));
  if (a.flag("bed") && !file_exists(a.get("bed", ""))) quit("couldn't open bed file");     // merge.nim:80-82
  const bool allow_diff = a.flag("diff-refs");
  std::vector<BamTarget> targets;
  if (a.flag("fasta") && allow_diff) targets = targets_from_fai(a.get("fasta", ""));          // merge.nim:84-86
  // --chromosome: the tid of that name IN THE FASTA (merge.nim:36-45,89); reads of other tids are dropped while a bin is read
  bool have_req = false;
  int32_t requested_tid = 0;
  if (a.flag("chromosome") && a.get("chromosome", "-2") != "-2") {
    const std::string chrom = a.get("chromosome", "");
    const std::vector<BamTarget> ft = targets_from_fai(a.get("fasta", ""));
    if (ft.empty()) quit("Error: unhandled exception: [strling merge] chromosome: %s specified, but no targets found in fasta. Specify a valid fasta file. [ValueError]", chrom.c_str());
    int found = -1;
    for (size_t k = 0; k < ft.size() && found < 0; ++k) if (ft[k].name == chrom) found = (int)k;
    if (found < 0) quit("Error: unhandled exception: [strling merge] chromosome: %s not found in fasta, check name and 'chr' prefix [ValueError]", chrom.c_str());
    have_req = true;
    requested_tid = found;
  }
  int window = atoi(a.get("window", "-1").c_str());
  const int min_support = atoi(a.get("min-support", "5").c_str());
  const uint16_t min_clip = (uint16_t)atoi(a.get("min-clip", "0").c_str());
  const uint16_t min_clip_total = (uint16_t)atoi(a.get("min-clip-total", "0").c_str());
  const bool verbose = a.flag("verbose");
  const std::string prefix = a.get("output-prefix", "strling");

  const auto tm0 = std::chrono::steady_clock::now();
  // the HIP runtime + the first device context come up beside the reading of the .bin files
  const int gpus_early = std::max(1, atoi(a.get("gpus", "1").c_str()));
  strl_ctx *ctx_early = nullptr;
  int ctx_early_rc = 0;
  std::string ctx_early_err;
  std::thread ctx_thread([&] { if (gpus_early == 1) { ctx_early_rc = strl_ctx_create(device_of(0), &ctx_early); if (ctx_early_rc) ctx_early_err = strl_last_error(); } });
  g_bg_init = &ctx_thread;
  uint32_t frag[4096] = {0};
  hvec<strl_tread> all;                                 // (huge pages, uninitialised: strl_bin_read fills a sample's share on several threads)
  for (size_t si = 0; si < a.pos.size(); ++si) {
    const std::string &path = a.pos[si];
    if (verbose) fprintf(stderr, "[strling] reading bin file: %s\n", path.c_str());
    strl_bin_info info;
    CHECK(strl_bin_peek(path.c_str(), &info));        // sizes from the header (names: an upper bound): the records are walked once
    std::string hdr((size_t)info.header_len, '\0');
    // the sample's treads are read straight behind the ones held so far and the dropped ones squeezed out in place: no second
    // copy of a whole genome's quarter gigabyte (names are not kept: merge.nim:118-125 replaces them by the sample's index)
    const size_t old_n = all.size();
    all.resize(old_n + (size_t)std::max(1, info.n_reads));
    strl_tread *t = all.data() + old_n;
    hvec<uint64_t> qo((size_t)info.n_reads + 1);
    hvec<char> qn((size_t)info.qnames_bytes + 1);
    CHECK(strl_bin_read(path.c_str(), &info, &hdr[0], t, qo.data(), qn.data()));
    const std::vector<BamTarget> tg = targets_from_header(hdr);
    if (targets.empty()) targets = tg;                                            // merge.nim:104-105
    else {
      bool same = tg.size() == targets.size();
      for (size_t k = 0; same && k < tg.size(); ++k) same = tg[k].name == targets[k].name && tg[k].length == targets[k].length;
      if (!same && !allow_diff) quit("[strling] Error: inconsistent bam header for %s. Were all samples run on the same reference genome?", path.c_str());
    }
    for (int k = 0; k < 4096; ++k) {                                               // merge.nim:112-115
      const uint32_t before = frag[k];
      frag[k] += info.frag[k];
      if (frag[k] < before) quit("overflow");
    }
    uint64_t kept = 0;
    for (int32_t k = 0; k < info.n_reads; ++k) {
      if (have_req && t[(size_t)k].tid != requested_tid) continue;                 // unpack.nim:126 requested_tid
      if (t[(size_t)k].tid < 0) continue;                                          // unpack_file(drop_unplaced=true)
      t[(size_t)k].qname_id = (int64_t)si;                                         // merge.nim:118-125: qname := sample index
      if (kept != (uint64_t)k) t[(size_t)kept] = t[(size_t)k];
      ++kept;
    }
    all.resize(old_n + (size_t)kept);
    fprintf(stderr, "[strling] read %llu STR reads from file: %s\n", (unsigned long long)kept, path.c_str());
  }
  if (verbose) {
    fprintf(stderr, "[strling] read %llu STR reads across all samples.\n", (unsigned long long)all.size());
    fprintf(stderr, "[strling] Calculated median fragment length accross all samples:%d\n", strl_frag_median(frag, 0.5));
    fprintf(stderr, "[strling] 10th, 90th percentile of fragment length:%d %d\n", strl_frag_median(frag, 0.1), strl_frag_median(frag, 0.9));
  }
  if (window < 0) window = strl_frag_median(frag, 0.98);                           // merge.nim:151-152
  const uint16_t max_clip_dist = (uint16_t)(0.5 * (double)strl_frag_median(frag, 0.5));   // merge.nim:181

  // -l BED: those loci take their reads first (merge.nim:154-167)
  std::vector<strl_locus> loci;
  if (a.flag("bed")) {
    loci = parse_bed(a.get("bed", ""), targets, (uint32_t)window);
    if (have_req) loci.erase(std::remove_if(loci.begin(), loci.end(), [&](const strl_locus &L) { return L.b.tid != requested_tid; }), loci.end());   // cluster.nim:139
    std::vector<uint64_t> aoff(loci.size() + 1);
    CHECK(strl_assign_reads_loci(all.data(), all.size(), STRL_MODE_MERGE, loci.data(), loci.size(), aoff.data(), nullptr, 0));
  }
  strl_ctx *ctx = nullptr;
  rvec<strl_bounds> bounds(std::max<size_t>(all.size(), 16));      // (left uninitialised: rows [0, nb) are written by the pass)
  uint64_t nb = 0, nu = 0;
  const int gpus = std::max(1, atoi(a.get("gpus", "1").c_str()));
  const auto tm1 = std::chrono::steady_clock::now();
  ctx_thread.join();
  g_bg_init = nullptr;
  const auto tm2 = std::chrono::steady_clock::now();
  if (gpus == 1) {
    if (ctx_early_rc) quit("[strling] %s (status %d)", ctx_early_err.c_str(), ctx_early_rc);
    ctx = ctx_early;
    CHECK(strl_cluster(ctx, all.data(), all.size(), STRL_MODE_MERGE, (uint32_t)window, min_support, min_clip, min_clip_total, max_clip_dist,
                       bounds.data(), bounds.size(), &nb, nullptr, 0, &nu, nullptr));
  } else {
    // contiguous shares of the reads (sample order is kept: rank-major = input order) go to the contexts,
    // one per device -- round-robin over the devices there are, so `--gpus 2` also runs on a one-GPU box --, the tread buffers
    // are all-gathered (RCCL over xGMI between different devices) and every context clusters the groups it owns; the rows come
    // back in the reference's order of the groups (first appearance, Nim table order: strl_group_order).
    const int n_dev = std::max(1, strl_device_count());
    std::vector<strl_ctx *> ctxs((size_t)gpus, nullptr);
    for (int r = 0; r < gpus; ++r) CHECK(strl_ctx_create(device_of(r), &ctxs[(size_t)r]));
    CHECK(strl_ctxs_comm_init(ctxs.data(), gpus));
    const size_t per = (all.size() + (size_t)gpus - 1) / (size_t)gpus;
    for (int r = 0; r < gpus; ++r) {
      const size_t lo = std::min(all.size(), per * (size_t)r), hi = std::min(all.size(), lo + per);
      CHECK(strl_ctx_set_treads(ctxs[(size_t)r], all.data() + lo, hi - lo));
    }
    CHECK(strl_ctxs_cluster_exchange(ctxs.data(), gpus, (uint32_t)std::max<size_t>(per, 1), STRL_MODE_MERGE, (int32_t)targets.size(), 0, (uint32_t)window, min_support,
                                     min_clip, min_clip_total, max_clip_dist));
    std::vector<strl_bounds> part(bounds.size());
    std::vector<strl_bounds> got;
    for (int r = 0; r < gpus; ++r) {
      uint64_t k = 0, ku = 0;
      CHECK(strl_cluster_collect(ctxs[(size_t)r], part.data(), part.size(), &k, nullptr, 0, &ku, nullptr));
      got.insert(got.end(), part.begin(), part.begin() + (long)k);
    }
    std::vector<strl_group_key> keys(all.size() + 1);
    uint64_t ng = 0;
    CHECK(strl_group_order(all.data(), all.size(), STRL_MODE_MERGE, keys.data(), keys.size(), &ng));
    std::map<std::pair<int32_t, std::string>, uint64_t> rank_of;
    for (uint64_t k = 0; k < ng; ++k) rank_of[{keys[(size_t)k].tid, std::string(keys[(size_t)k].repeat, strnlen(keys[(size_t)k].repeat, 6))}] = k;
    std::stable_sort(got.begin(), got.end(), [&](const strl_bounds &x, const strl_bounds &y) {
      return rank_of[{x.tid, std::string(x.repeat, strnlen(x.repeat, 6))}] < rank_of[{y.tid, std::string(y.repeat, strnlen(y.repeat, 6))}];
    });
    nb = got.size();
    std::copy(got.begin(), got.end(), bounds.begin());
    if (verbose) {
      int w = 0, rk = 0, rccl = 0;
      (void)strl_ctx_comm_info(ctxs[0], &w, &rk, &rccl);
      fprintf(stderr, "[strling] clustered on %d contexts over %d device(s), exchange by %s\n", gpus, std::min(gpus, n_dev), rccl ? "RCCL all-gather" : "device copies (contexts share a device)");
    }
    for (int r = 1; r < gpus; ++r) strl_ctx_destroy(ctxs[(size_t)r]);
    ctx = ctxs[0];
  }
  const auto tm3 = std::chrono::steady_clock::now();
  const std::string outp = prefix + "-bounds.txt";
  FILE *fo = fopen(outp.c_str(), "w");
  if (!fo) quit("couldn't open output file");
  fputs("#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total\n", fo);   // cluster.nim:89
  char row[1024];
  for (const strl_locus &L : loci) {                                                 // merge.nim:165-167: reported as they are
    locus_row(row, sizeof row, L, targets[(size_t)L.b.tid].name.c_str());
    fputs(row, fo);
    fputc('\n', fo);
  }
  for (uint64_t k = 0; k < nb; ++k) {
    const strl_bounds &b = bounds[(size_t)k];
    strl_bounds_row(row, sizeof row, &b, targets[(size_t)b.tid].name.c_str());
    fputs(row, fo);
    fputc('\n', fo);
  }
  fclose(fo);
  if (verbose) {
    fprintf(stderr, "[strling] Wrote merged str bounds to %s\n", outp.c_str());
    fprintf(stderr, "[strling] seconds: .bin files (beside the device context) %.3f  waiting for the device context %.3f  clustering %.3f  rows %.3f\n",
            std::chrono::duration<double>(tm1 - tm0).count(), std::chrono::duration<double>(tm2 - tm1).count(), std::chrono::duration<double>(tm3 - tm2).count(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - tm3).count());
  }
  return end_process(ctx);
}

// The run of BGZF blocks the walk of one region query (BamReader::read_region) passes through, as far as the .bai linear
// index can bound it: from the block of the index offset of `beg`'s window up to the block that holds the first record
// overlapping a later window than `end`'s -- and the one behind it --, their sizes read from the block headers and trailers
// (18 + 8 bytes per block).  ok = false: the index does not bound the region, the file is laid out unusually (other extra
// subfields than BC), the run is longer than 8 MB: such a region is read by the host reader.
// fragment_length_distribution (utils.nim:86-111; call.nim:92) through the device front end: the first ~2.2 M records' flag / isize
// words come from the parse kernel (the words `strling extract` makes its histogram from), so the host inflates nothing -- on
// 16 CPUs the host's inflate of that sample was 3 of the 4.7 CPU-seconds `strling call`'s start-up consists of (context, `.bin`
// and sample side by side: ~0.3 s each; the context alone is 0.11 s).  pin / pin_meta: three page-locked chunk buffers of
// chunk_blocks blocks each (allocated beside the context).  false: the caller takes the host's pass (the verdict on a file the
// front end refuses is the host reader's).
static bool fragment_lengths_on_device(strl_ctx *ctx, const std::string &bam, uint32_t frag[4096], uint8_t *const pin[3], uint8_t *const pin_meta[3], size_t chunk_blocks,
                                       size_t chunk_bytes, std::string &why) {
  const int64_t n_reads = 2000000, skip_reads = 100000;
  memset(frag, 0, 4096 * sizeof(uint32_t));
  const bool tm = getenv("STRL_FRAG_TIMING") != nullptr;
  const auto tm0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (tm) fprintf(stderr, "[fragment lengths] %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tm0).count()); };
  BgzfFeed feed;
  std::string err;
  if (!feed.open(bam, err)) { why = err; return false; }
  lap("file open, header walked");
  strl_opts o{0, 0.8, 40};                                  // (the chunks are scored too -- a few milliseconds nobody looks at)
  auto fail = [&](const char *what) { why = std::string(what) + ": " + strl_last_error(); (void)strl_front_end(ctx); return false; };
  if (strl_ctx_set_opts(ctx, &o) || strl_front_begin(ctx, (int32_t)feed.targets().size(), feed.first_record_offset(), 3u << 20) ||
      strl_front_reserve(ctx, (uint32_t)chunk_blocks, chunk_bytes))
    return fail("front end");
  std::vector<int32_t> skipped;
  int64_t counted = 0;
  uint64_t next = 0;
  bool done = false;
  std::vector<uint32_t> fw;
  auto take = [&](uint64_t upto) -> bool {                    // the words of records [next, upto)
    while (next < upto && !done) {
      const uint64_t m = std::min<uint64_t>(upto - next, 1u << 20);
      fw.resize((size_t)m);
      if (strl_front_fragwords(ctx, next, m, fw.data())) return false;
      for (uint64_t k = 0; k < m && !done; ++k) {
        const int64_t i = (int64_t)(next + k);
        const uint32_t f = fw[(size_t)k] & 0xffffu, is = fw[(size_t)k] >> 16;
        if (!(f & 0x2)) continue;
        if (f & (0x800 | 0x100)) continue;
        if (is > 4095u) continue;
        if (i < skip_reads) { skipped.push_back((int32_t)is); continue; }
        skipped.clear();
        frag[is]++;
        if (++counted > n_reads) done = true;
      }
      next += m;
    }
    return true;
  };
  lap("front end begun, buffers reserved");
  ThreadPool pool(std::min(decode_threads(), 12));
  // chunk ci + 1 is read (header walk, copy out of the file's mapping) by a thread beside the push of chunk ci and the wait for
  // chunk ci - 1's parse: one after the other the seven chunks of a sample took 12 ms each, 3 of them the device's
  struct St { int64_t nb = 0; size_t lo = 0, hi = 0; bool short_read = false; std::string err; };
  St ring[3];
  std::vector<BgzfFeed::Block> bl;
  auto stage = [&](uint64_t ci, St &S) {
    S = St{};
    S.nb = feed.next(bl, chunk_blocks, chunk_bytes, S.err);
    if (S.nb <= 0) return;
    uint8_t *dst = pin[ci % 3];
    S.lo = bl.front().c_off; S.hi = bl.back().c_off + bl.back().clen;
    const size_t lo = S.lo, hi = S.hi, piece = (size_t)4 << 20, pieces = (hi - lo + piece - 1) / piece;
    std::atomic<int> short_reads{0};
    pool.parallel_for(pieces, [&](size_t k) { if (!feed.copy_at(dst + k * piece, lo + k * piece, std::min(piece, hi - lo - k * piece))) ++short_reads; });
    feed.done_with(lo, hi - lo);
    S.short_read = short_reads.load() != 0;
    uint64_t *coff = reinterpret_cast<uint64_t *>(pin_meta[ci % 3]);
    uint32_t *clen = reinterpret_cast<uint32_t *>(coff + chunk_blocks), *isz = clen + chunk_blocks, *crc = isz + chunk_blocks;
    for (size_t k = 0; k < (size_t)S.nb; ++k) { coff[k] = bl[k].c_off - lo; clen[k] = bl[k].clen; isz[k] = bl[k].isize; crc[k] = bl[k].crc; }
  };
  stage(0, ring[0]);
  std::thread ahead;
  bool file_ended = false;
  for (uint64_t ci = 0; !done; ++ci) {
    const St cur = ring[ci % 3];
    if (cur.nb < 0 || cur.short_read) { why = cur.nb < 0 ? cur.err : "short read"; (void)strl_front_end(ctx); return false; }
    if (cur.nb == 0) { file_ended = true; break; }
    ahead = std::thread([&, ci] { stage(ci + 1, ring[(ci + 1) % 3]); });
    uint64_t *coff = reinterpret_cast<uint64_t *>(pin_meta[ci % 3]);
    uint32_t *clen = reinterpret_cast<uint32_t *>(coff + chunk_blocks), *isz = clen + chunk_blocks, *crc = isz + chunk_blocks;
    strl_front_chunk dn[2];
    int n_dn = 0;
    uint64_t parsed = 0;
    const bool ok = !strl_front_push(ctx, pin[ci % 3], cur.hi - cur.lo, coff, clen, isz, crc, (uint32_t)cur.nb, dn, &n_dn) && !strl_front_records(ctx, &parsed) && take(parsed);
    ahead.join();
    if (!ok) return fail("front end");
    lap("chunk pushed, words of the parsed records taken");
  }
  (void)file_ended;
  if (!done) {                                              // the file ended first: what is still in the pipeline
    strl_front_chunk dn[2];
    int n_dn = 0;
    uint64_t parsed = 0;
    if (strl_front_finish(ctx, dn, &n_dn) || strl_front_records(ctx, &parsed) || !take(parsed)) return fail("front end");
  }
  lap("sample complete");
  if (strl_front_end(ctx)) { why = strl_last_error(); return false; }
  lap("front end given up");
  uint64_t sum = 0;
  for (int k = 0; k < 4096; ++k) sum += frag[k];
  if ((uint32_t)sum == 0) {
    fprintf(stderr, "using first reads in fragment_length_distribution calculation as there were not enough\n");
    for (int32_t is : skipped) frag[is]++;
  }
  return true;
}

struct RegionPlan { bool ok = false; uint64_t c_beg = 0, c_end = 0; uint32_t in_block = 0; std::vector<uint32_t> hdr, bsize, isz, crc; };
static void plan_region(const BamReader &rd, int fd, int32_t tid, int64_t beg, int64_t end, RegionPlan &P) {
  uint64_t c_hint = 0;
  P.ok = false;
  if (!rd.region_span(tid, beg, end, P.c_beg, P.in_block, c_hint)) return;
  uint64_t o = P.c_beg, infl = 0;
  int beyond = 0;
  bool good = true;
  while (beyond < 2 && infl < ((uint64_t)8 << 20) && P.bsize.size() < 4096) {   // (a window of 30x data is ~1 MB)
    uint8_t h[18], tr[8];
    if (pread(fd, h, 18, (off_t)o) != 18) break;                                  // end of the file
    const uint32_t xlen = h[10] | (h[11] << 8);
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || xlen != 6 || h[12] != 'B' || h[13] != 'C') { good = false; break; }
    const uint32_t bsize = (h[16] | (h[17] << 8)) + 1u;
    if (bsize < 26 || pread(fd, tr, 8, (off_t)(o + bsize - 8)) != 8) { good = false; break; }
    uint32_t crc, isz;
    memcpy(&crc, tr, 4); memcpy(&isz, tr + 4, 4);
    if (isz > 65536u) { good = false; break; }
    P.hdr.push_back(18); P.bsize.push_back(bsize); P.isz.push_back(isz); P.crc.push_back(crc);
    infl += isz;
    if (o >= c_hint) ++beyond;
    o += bsize;
  }
  P.c_end = o;
  // the first block must hold the index offset; a run that did not reach the hinted block goes to the host reader
  P.ok = good && beyond >= 1 && !P.bsize.empty() && P.in_block <= P.isz[0];
}

// call.nim:51-285
static int call_main(int argc, char **argv) {
  const char *usage =
      "strling call\n\nUsage:\n  strling call [options] bam bin\n\nArguments:\n  bam              path to bam file\n"
      "  bin              bin file previously created by `strling extract`\n\nOptions:\n  -f, --fasta=FASTA          path to fasta file\n"
      "  -m, --min-support=MIN_SUPPORT\n                             minimum number of supporting reads for a locus to be reported (default: 5)\n"
      "  -c, --min-clip=MIN_CLIP    minimum number of supporting clipped reads for each side of a locus (default: 0)\n"
      "  -t, --min-clip-total=MIN_CLIP_TOTAL\n                             minimum total number of supporting clipped reads for a locus (default: 0)\n"
      "  -q, --min-mapq=MIN_MAPQ    minimum mapping quality (does not apply to STR reads) (default: 40)\n"
      "  -l, --loci=LOCI            Annoated bed file specifying additional STR loci to genotype. Format is: chr start stop repeatunit [name]\n"
      "  -b, --bounds=BOUNDS        STRling -bounds.txt file (usually produced by strling merge) specifying additional STR loci to genotype.\n"
      "  -o, --output-prefix=OUTPUT_PREFIX\n                             prefix for output files (default: strling)\n  -v, --verbose\n  -h, --help                 Show this help\n";
  if (argc <= 2) { fputs(usage, stdout); return 0; }
  const Args a = parse(argc, argv, 2, {{"fasta", 'f', true}, {"min-support", 'm', true}, {"min-clip", 'c', true}, {"min-clip-total", 't', true},
                                       {"min-mapq", 'q', true}, {"loci", 'l', true}, {"bounds", 'b', true}, {"output-prefix", 'o', true},
                                       {"verbose", 'v', false}, {"device", 'D', true}}, usage);
  if (a.pos.size() != 2) quit("expected 2 arguments (bam, bin)\n%s", usage);
  set_device0(a.get("device", ""));
  if (a.flag("loci") && !file_exists(a.get("loci", ""))) quit("couldn't open loci file");          // call.nim:81-87
  if (a.flag("bounds") && !file_exists(a.get("bounds", ""))) quit("couldn't open bounds file");
  const std::string bam = a.pos[0], bin = a.pos[1], prefix = a.get("output-prefix", "strling");
  const int min_support = atoi(a.get("min-support", "5").c_str());
  const uint16_t min_clip = (uint16_t)atoi(a.get("min-clip", "0").c_str());
  const uint16_t min_clip_total = (uint16_t)atoi(a.get("min-clip-total", "0").c_str());
  const uint8_t min_mapq = (uint8_t)atoi(a.get("min-mapq", "40").c_str());
  const bool verbose = a.flag("verbose");

  g_cram_fasta = a.get("fasta", "");
  const auto t_call0 = std::chrono::steady_clock::now();
  // three things that do not depend on each other, side by side: the HIP runtime + device context (0.2 - 0.4 s of driver work),
  // the .bin of extract (call.nim:118-123; a whole genome's is a quarter of a gigabyte of msgpack), the fragment lengths
  strl_ctx *ctx = nullptr;
  int ctx_rc = 0;
  std::string ctx_err;
  double t_up_ctx = 0, t_up_bin = 0, t_up_bam = 0;
  std::thread ctx_thread([&] { ctx_rc = strl_ctx_create(device_of(0), &ctx); if (ctx_rc) ctx_err = strl_last_error(); t_up_ctx = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call0).count(); });
  g_bg_init = &ctx_thread;
  strl_bin_info info;
  std::string hdr;
  hvec<strl_tread> treads;                             // (hvec: huge pages, resize leaves the elements uninitialised -- strl_bin_read fills them on several threads)
  hvec<uint64_t> qoff;
  hvec<char> qnames;
  int bin_rc = 0;
  std::string bin_err;
  std::thread bin_thread([&] {
    bin_rc = strl_bin_peek(bin.c_str(), &info);          // sizes from the header (names: an upper bound): the records are walked once
    if (!bin_rc) {
      hdr.assign((size_t)info.header_len, '\0');
      treads.resize((size_t)std::max(1, info.n_reads));
      qoff.resize((size_t)info.n_reads + 1);
      qnames.resize((size_t)info.qnames_bytes + 1);
      bin_rc = strl_bin_read(bin.c_str(), &info, &hdr[0], treads.data(), qoff.data(), qnames.data());
    }
    if (bin_rc) bin_err = strl_last_error();
    t_up_bin = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call0).count();
  });
  // The fragment-length sample (call.nim:92): through the device front end once the context is up (a BAM; STRL_CALL_FRAG=host and
  // CRAM input keep the host's pass, which also is the way out when the front end refuses the file).  Its three page-locked
  // chunk buffers come up beside the context.
  uint32_t frag[4096];
  const char *frag_env = getenv("STRL_CALL_FRAG");
  bool frag_on_device = !CramFile::is_cram(bam) && !(frag_env && !strcmp(frag_env, "host"));
  const size_t fr_blocks = 4096, fr_bytes = fr_blocks * 20000;
  uint8_t *fr_pin[3] = {nullptr, nullptr, nullptr}, *fr_meta[3] = {nullptr, nullptr, nullptr};
  std::thread fr_pin_thread;
  if (frag_on_device)
    fr_pin_thread = std::thread([&] {
      for (int k = 0; k < 3; ++k) { fr_pin[k] = static_cast<uint8_t *>(strl_pinned_alloc(fr_bytes + 64)); fr_meta[k] = static_cast<uint8_t *>(strl_pinned_alloc(fr_blocks * 20 + 64)); }
    });
  if (!frag_on_device) fragment_length_distribution(bam, frag);                       // call.nim:92
  BamReader rd;
  std::string err;
  const bool opened = rd.open(bam, err) && rd.load_index(bam, err);                   // index=true, call.nim:101-102
  ctx_thread.join();
  if (fr_pin_thread.joinable()) fr_pin_thread.join();
  if (frag_on_device && opened && !ctx_rc) {
    bool have = true;
    for (int k = 0; k < 3; ++k) have = have && fr_pin[k] && fr_meta[k];
    std::string why = "page-locked memory";
    if (!have || !fragment_lengths_on_device(ctx, bam, frag, fr_pin, fr_meta, fr_blocks, fr_bytes, why)) {
      if (verbose) fprintf(stderr, "[strling] fragment lengths on the host (%s)\n", why.c_str());
      fragment_length_distribution(bam, frag);
    }
  } else if (frag_on_device) fragment_length_distribution(bam, frag);                 // (errors of the open / the context are reported below)
  // (the three page-locked buffers are left to the end of the process: unlocking them here would stall the device in front of the clustering)
  const int frag_median = strl_frag_median(frag, 0.5);
  if (verbose) {
    fprintf(stderr, "Calculated median fragment length:%d\n", frag_median);
    fprintf(stderr, "10th, 90th percentile of fragment length:%d %d\n", strl_frag_median(frag, 0.1), strl_frag_median(frag, 0.9));
  }
  t_up_bam = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call0).count();
  bin_thread.join();
  g_bg_init = nullptr;
  if (!opened) quit_open(bam, err);
  if (bin_rc) quit("[strling] %s (status %d)", bin_err.c_str(), bin_rc);
  if (ctx_rc) quit("[strling] %s (status %d)", ctx_err.c_str(), ctx_rc);
  const double t_start_up = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call0).count();
  const int window = strl_frag_median(frag, 0.99);                                  // call.nim:109
  const strl_call_opts copts{frag_median, min_support, min_clip, min_clip_total};
  {
    const std::vector<BamTarget> tg = targets_from_header(hdr);
    bool same = tg.size() == rd.targets().size();
    for (size_t k = 0; same && k < tg.size(); ++k) same = tg[k].name == rd.targets()[k].name && tg[k].length == rd.targets()[k].length;
    if (!same) quit("[strling] the bin file and the bam do not have the same reference sequences (doAssert call.nim:121)");
  }
  const std::string pb = prefix + "-bounds.txt", pg = prefix + "-genotype.txt", pu = prefix + "-unplaced.txt";
  FILE *gt_fh = fopen(pg.c_str(), "w"), *bounds_fh = fopen(pb.c_str(), "w"), *unplaced_fh = fopen(pu.c_str(), "w");
  if (!gt_fh || !bounds_fh || !unplaced_fh) quit("couldn't open output file");
  fputs("#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total\tdepth\n", bounds_fh);   // call.nim:145
  fputs("#chrom\tleft\tright\trepeatunit\tallele1_est\tallele2_est\tanchored_reads\tspanning_reads\tspanning_pairs\texpected_spanning_pairs\t"
        "spanning_pairs_pctl\tleft_clips\tright_clips\tunplaced_pairs\tdepth\tsum_str_counts\n", gt_fh);                          // genotyper.nim:54

  // evidence + genotype of one bound (call.nim:196-218 / :237-255): indexed region read, spanners(), genotype(), one row.
  // The reference walks its bounds one after the other; a bound's evidence depends on nothing but the bound, its reads and
  // the file, so the bounds are worked on by a pool of threads (a reader each) and their rows written in the reference's order.
  std::vector<strl_call> calls;
  struct Task { strl_bounds b; const char *name; const std::vector<uint32_t> *idx; uint64_t i0, i1; const strl_tread *src; };
  struct Done { bool keep = false; strl_call c; std::string row; int depth = 0; };
  struct Worker { BamReader rd; RecordBatch region; std::vector<strl_support> sup; std::vector<strl_tread> cl; bool open = false; };
  const int n_workers = std::max(1, std::min(decode_threads(), 32));
  std::vector<std::unique_ptr<Worker>> workers;
  for (int k = 0; k < n_workers; ++k) workers.emplace_back(new Worker());
  ThreadPool ev_pool(n_workers);
  double t_evidence = 0;
  std::atomic<uint64_t> ns_region{0}, ns_rules{0};      // summed over the workers: region reads (seek + inflate + parse) / spanners + genotype
  // What a worker does with a bound once its region's records are in w.region: spanners(), genotype(), the row.
  auto rules = [&](Worker &w, const Task &t, Done &d, std::string &werr) -> bool {
    const strl_bounds &b = t.b;
    char row[2048];
    const strl_records rv = w.region.view();
    w.sup.resize(2 * w.region.size() + 16);
    strl_span_summary sm{};
    if (strl_spanners(&rv, w.region.isize.data(), &b, window, frag, min_mapq, w.sup.data(), w.sup.size(), &sm) != STRL_OK) { werr = std::string("[strling] ") + strl_last_error(); return false; }
    if (sm.n_support > 5000) return true;                                              // spans.len > 5_000
    if (sm.median_depth == -1) return true;
    w.cl.clear();
    for (uint64_t k = t.i0; k < t.i1; ++k) w.cl.push_back(t.src[(*t.idx)[(size_t)k]]);
    memset(&d.c, 0, sizeof d.c);
    if (strl_genotype(&b, w.cl.data(), w.cl.size(), qoff.data(), qnames.data(), w.sup.data(), sm.n_support, &copts, (double)sm.median_depth, &d.c) != STRL_OK) {
      werr = std::string("[strling] ") + strl_last_error();
      return false;
    }
    d.c.expected_spanning_fragments = sm.expected_spanners;
    strl_locus L{};
    L.b = b;
    if (t.name) snprintf(L.name, sizeof L.name, "%s", t.name);
    locus_row(row, sizeof row, L, rd.targets()[(size_t)b.tid].name.c_str());
    d.row = row;
    d.depth = sm.median_depth;
    d.keep = true;
    return true;
  };
  // The regions of many bounds through the device (strl_regions_fetch): the .bai linear index gives every region's run of
  // BGZF blocks; their compressed bytes are read into page-locked memory, inflated on the GPU -- which also cuts out the
  // records the query returns -- and the workers are left with parsing a few hundred records and the rules per bound.  The
  // host path inflates ~10 blocks per bound on a CPU (2 - 3 ms): that was the whole evidence step.  Batches of 384 MB of
  // inflated bytes (128 ... 768 MB measured alike); one batch is being read while two are on the device and one with the workers.  STRL_CALL_REGIONS=host keeps the
  // host path; a region whose blocks the index cannot bound, and CRAM input, take it by themselves.
  struct Batch {
    size_t t0 = 0, t1 = 0;
    std::vector<uint32_t> which;            // tasks of [t0, t1) that go through the device
    std::vector<uint64_t> coff, out_off, out_len;
    std::vector<uint32_t> clen, isize, crc;
    std::vector<strl_region_req> req;
    std::vector<uint8_t> status;
    uint64_t comp_bytes = 0, inflated = 0;
    int rc = 0;
    std::string err;
  };
  const char *regions_env = getenv("STRL_CALL_REGIONS");
  const bool device_regions = !rd.is_cram() && !(regions_env && !strcmp(regions_env, "host"));
  const uint64_t batch_inflated = getenv("STRL_CALL_BATCH_MB") ? (uint64_t)atoll(getenv("STRL_CALL_BATCH_MB")) << 20 : (uint64_t)384 << 20;
  int region_fd = -1;
  constexpr size_t NSETS = 4;      // batches in flight: one being read, two on the device (its two region slots), one with the workers
  uint8_t *pin_comp[NSETS] = {}, *pin_out[NSETS] = {};
  uint64_t pin_comp_cap[NSETS] = {}, pin_out_cap[NSETS] = {};
  double t_plan = 0, t_fetch = 0, t_wait_fetch = 0, t_pin = 0, t_pread = 0;
  uint64_t n_dev_regions = 0, n_host_regions = 0, dev_comp_bytes = 0, dev_inflated = 0, dev_kept = 0;
  // the batches' page-locked buffers come up beside the clustering (its device pass is 3 ms; the rest is the host's): sized for a
  // batch of the default size, grown by the reader if a batch needs more -- they were 0.03 s in front of the first batches
  std::thread region_pin_thread;
  if (device_regions)
    region_pin_thread = std::thread([&] {
      for (size_t k = 0; k < NSETS; ++k) {
        pin_comp_cap[k] = batch_inflated / 2 + (16 << 20);
        pin_out_cap[k] = (uint64_t)((double)batch_inflated * 0.75) + (16 << 20);
        pin_comp[k] = static_cast<uint8_t *>(strl_pinned_alloc(pin_comp_cap[k]));
        pin_out[k] = static_cast<uint8_t *>(strl_pinned_alloc(pin_out_cap[k]));
        if (!pin_comp[k]) pin_comp_cap[k] = 0;
        if (!pin_out[k]) pin_out_cap[k] = 0;
      }
    });
  auto run_tasks = [&](const std::vector<Task> &tasks) {
    if (region_pin_thread.joinable()) region_pin_thread.join();
    if (tasks.empty()) return;
    const auto te0 = std::chrono::steady_clock::now();
    std::vector<Done> done(tasks.size());
    std::mutex wm;
    std::vector<int> free_w;
    for (int k = n_workers - 1; k >= 0; --k) free_w.push_back(k);
    std::atomic<bool> failed{false};
    std::string fail_msg;
    auto fail = [&](const std::string &m) { std::lock_guard<std::mutex> lk(wm); if (!failed.exchange(true)) fail_msg = m; };
    // one bound on the host: indexed region read by the worker's own reader, then the rules
    auto host_task = [&](Worker &w, size_t j) -> bool {
      const Task &t = tasks[j];
      std::string werr;
      if (!w.open) { if (!w.rd.open_like(rd, werr)) { fail("couldn't open bam"); return false; } w.open = true; }
      w.region.clear();
      const int64_t wl = (int64_t)t.b.left - window, wr = (int64_t)t.b.right + window;
      const auto tw0 = std::chrono::steady_clock::now();
      if (w.rd.read_region(w.region, t.b.tid, std::max<int64_t>(0, wl), wr, werr) < 0) { fail("[strling] error reading " + bam + ": " + werr); return false; }
      const auto tw1 = std::chrono::steady_clock::now();
      ns_region += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(tw1 - tw0).count();
      const bool ok = rules(w, t, done[j], werr);
      ns_rules += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw1).count();
      if (!ok) fail(werr);
      return ok;
    };
    auto with_worker = [&](const std::function<void(Worker &)> &fn) {
      int wi;
      { std::lock_guard<std::mutex> lk(wm); wi = free_w.back(); free_w.pop_back(); }
      fn(*workers[(size_t)wi]);
      { std::lock_guard<std::mutex> lk(wm); free_w.push_back(wi); }
    };
    const size_t per = 8;
    if (!device_regions) {
      ev_pool.parallel_for((tasks.size() + per - 1) / per, [&](size_t blk) {
        with_worker([&](Worker &w) {
          for (size_t j = blk * per; j < std::min(tasks.size(), (blk + 1) * per) && !failed.load(); ++j) if (!host_task(w, j)) break;
        });
      });
    } else {
      // ---- plan: every region's run of blocks (index lookup + a walk over the block headers, 18 + 8 bytes read per block)
      const auto tp0 = std::chrono::steady_clock::now();
      if (region_fd < 0) region_fd = open(bam.c_str(), O_RDONLY);
      if (region_fd < 0) quit("couldn't open bam");
      std::vector<RegionPlan> plan(tasks.size());
      ev_pool.parallel_for((tasks.size() + 63) / 64, [&](size_t blk) {
        for (size_t j = blk * 64; j < std::min(tasks.size(), (blk + 1) * 64); ++j) {
          const Task &t = tasks[j];
          RegionPlan &P = plan[j];
          const int64_t wl = std::max<int64_t>(0, (int64_t)t.b.left - window), wr = (int64_t)t.b.right + window;
          plan_region(rd, region_fd, t.b.tid, wl, std::min<int64_t>(wr, INT32_MAX), P);
        }
      });
      t_plan += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count();
      // ---- batches of consecutive tasks
      std::vector<Batch> batches;
      for (size_t j = 0; j < tasks.size();) {
        Batch B;
        B.t0 = j;
        while (j < tasks.size() && (B.inflated < batch_inflated || B.which.empty())) {
          const RegionPlan &P = plan[j];
          if (P.ok) {
            strl_region_req q{};
            q.first_block = (uint32_t)B.clen.size(); q.n_blocks = (uint32_t)P.bsize.size(); q.in_block = P.in_block;
            q.tid = tasks[j].b.tid;
            q.beg = (int32_t)std::max<int64_t>(0, (int64_t)tasks[j].b.left - window);
            q.end = (int32_t)std::min<int64_t>((int64_t)tasks[j].b.right + window, INT32_MAX);
            uint64_t o = 0;
            for (size_t k = 0; k < P.bsize.size(); ++k) {
              B.coff.push_back(B.comp_bytes + o + P.hdr[k]);
              B.clen.push_back(P.bsize[k] - P.hdr[k] - 8);
              B.isize.push_back(P.isz[k]);
              B.crc.push_back(P.crc[k]);
              B.inflated += P.isz[k];
              o += P.bsize[k];
            }
            B.comp_bytes += (o + 15) & ~(uint64_t)15;
            B.req.push_back(q);
            B.which.push_back((uint32_t)j);
          }
          ++j;
        }
        B.t1 = j;
        batches.push_back(std::move(B));
      }
      // ---- three stages side by side over NSETS buffer sets: a reader (compressed bytes of batch i + 2 into page-locked
      // memory), the device (strl_regions_fetch of batch i + 1), the workers (batch i)
      ThreadPool io_pool(std::max(2, std::min(n_workers / 3, 8)));
      std::mutex pm;
      std::condition_variable pcv;
      size_t n_read = 0, n_done = 0;                     // batches that left the reader / the workers
      std::vector<char> fetched(batches.size(), 0);      // ... the device (two threads, alternate batches)
      bool stop_stages = false;
      auto read_stage = [&] {
        for (size_t bi = 0; bi < batches.size(); ++bi) {
          { std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return stop_stages || bi < n_done + NSETS; }); if (stop_stages) return; }
          Batch &B = batches[bi];
          const int s = (int)(bi % NSETS);
          if (!B.which.empty()) {
            const auto tr0 = std::chrono::steady_clock::now();
            // what the queries return is a fifth to a half of what the index makes one read (narrow / wide bounds): sized by a
            // guess, and by what it took if the guess proves short (STRL_ERR_CAPACITY: one more call with the bytes it names)
            const uint64_t want_out = (uint64_t)((double)B.inflated * 0.6) + 32 * B.which.size() + (1 << 20);
            if (pin_comp_cap[s] < B.comp_bytes + 64) {
              if (pin_comp[s]) strl_pinned_free(pin_comp[s]);
              pin_comp_cap[s] = B.comp_bytes + B.comp_bytes / 4 + 64;
              pin_comp[s] = static_cast<uint8_t *>(strl_pinned_alloc(pin_comp_cap[s]));
            }
            if (pin_out_cap[s] < want_out) {
              if (pin_out[s]) strl_pinned_free(pin_out[s]);
              pin_out_cap[s] = want_out + want_out / 4;
              pin_out[s] = static_cast<uint8_t *>(strl_pinned_alloc(pin_out_cap[s]));
            }
            const auto tr1 = std::chrono::steady_clock::now();
            t_pin += std::chrono::duration<double>(tr1 - tr0).count();
            if (!pin_comp[s] || !pin_out[s]) { B.rc = STRL_ERR_HIP; B.err = "page-locked memory for the region reads"; }
            else {
              std::atomic<bool> io_bad{false};
              io_pool.parallel_for((B.which.size() + 15) / 16, [&](size_t blk) {
                for (size_t k = blk * 16; k < std::min(B.which.size(), (blk + 1) * 16); ++k) {
                  const RegionPlan &P = plan[B.which[k]];
                  const uint64_t at = B.coff[B.req[k].first_block] - P.hdr[0], len = P.c_end - P.c_beg;
                  uint64_t got = 0;
                  while (got < len) {
                    const ssize_t r = pread(region_fd, pin_comp[s] + at + got, (size_t)(len - got), (off_t)(P.c_beg + got));
                    if (r <= 0) { io_bad = true; break; }
                    got += (uint64_t)r;
                  }
                }
              });
              if (io_bad.load()) { B.rc = STRL_ERR_IO; B.err = "reading " + bam; }
            }
            t_pread += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr1).count();
          }
          { std::lock_guard<std::mutex> lk(pm); n_read = bi + 1; }
          pcv.notify_all();
        }
      };
      auto device_stage = [&](size_t first) {
        for (size_t bi = first; bi < batches.size(); bi += 2) {
          { std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return stop_stages || bi < n_read; }); if (stop_stages) return; }
          Batch &B = batches[bi];
          const int s = (int)(bi % NSETS);
          if (!B.which.empty() && !B.rc) {
            const auto tf0 = std::chrono::steady_clock::now();
            B.out_off.assign(B.which.size(), 0); B.out_len.assign(B.which.size(), 0); B.status.assign(B.which.size(), 1);
            B.rc = strl_regions_fetch(ctx, pin_comp[s], B.comp_bytes, B.coff.data(), B.clen.data(), B.isize.data(), B.crc.data(), (uint32_t)B.clen.size(), B.req.data(),
                                      (uint32_t)B.req.size(), pin_out[s], pin_out_cap[s], B.out_off.data(), B.out_len.data(), B.status.data());
            if (B.rc == STRL_ERR_CAPACITY && !B.out_off.empty() && B.out_off[0] > pin_out_cap[s]) {
              const uint64_t need = B.out_off[0] + B.out_off[0] / 16 + 64;
              strl_pinned_free(pin_out[s]);
              pin_out[s] = static_cast<uint8_t *>(strl_pinned_alloc(need));
              pin_out_cap[s] = pin_out[s] ? need : 0;
              if (pin_out[s])
                B.rc = strl_regions_fetch(ctx, pin_comp[s], B.comp_bytes, B.coff.data(), B.clen.data(), B.isize.data(), B.crc.data(), (uint32_t)B.clen.size(), B.req.data(),
                                          (uint32_t)B.req.size(), pin_out[s], pin_out_cap[s], B.out_off.data(), B.out_len.data(), B.status.data());
            }
            if (B.rc) B.err = strl_last_error();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
            { std::lock_guard<std::mutex> lk(pm); t_fetch += dt; }
          }
          { std::lock_guard<std::mutex> lk(pm); fetched[bi] = 1; }
          pcv.notify_all();
        }
      };
      std::thread reader(read_stage), fetcher(device_stage, (size_t)0), fetcher2(device_stage, (size_t)1);
      for (size_t bi = 0; bi < batches.size() && !failed.load(); ++bi) {
        const auto tq0 = std::chrono::steady_clock::now();
        { std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return fetched[bi] != 0; }); }
        t_wait_fetch += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count();
        Batch &B = batches[bi];
        if (B.rc == STRL_ERR_FORMAT || B.rc == STRL_ERR_CRC || B.rc == STRL_ERR_CAPACITY) {
          // a block the device decoder does not take, damaged data, more records than the buffer was sized for: the host
          // reader, zlib behind it, has the last word on every region of the batch
          B.status.assign(B.which.size(), 1);
          B.rc = 0;
        }
        if (B.rc) {
          { std::lock_guard<std::mutex> lk(pm); stop_stages = true; }
          pcv.notify_all();
          reader.join(); fetcher.join(); fetcher2.join();
          quit("[strling] error reading %s: %s (status %d)", bam.c_str(), B.err.c_str(), B.rc);
        }
        const int s = (int)(bi % NSETS);
        // position of every task of the batch among its device regions (-1: host)
        std::vector<int64_t> slot(B.t1 - B.t0, -1);
        for (size_t k = 0; k < B.which.size(); ++k) {
          if (B.status[k] == 0) { slot[B.which[k] - B.t0] = (int64_t)k; ++n_dev_regions; dev_kept += B.out_len[k]; }
        }
        dev_comp_bytes += B.comp_bytes; dev_inflated += B.inflated;
        ev_pool.parallel_for((B.t1 - B.t0 + per - 1) / per, [&](size_t blk) {
          with_worker([&](Worker &w) {
            for (size_t j = B.t0 + blk * per; j < std::min(B.t1, B.t0 + (blk + 1) * per) && !failed.load(); ++j) {
              const int64_t k = slot[j - B.t0];
              if (k < 0) { { std::lock_guard<std::mutex> lk(wm); ++n_host_regions; } if (!host_task(w, j)) break; continue; }
              std::string werr;
              w.region.clear();
              const auto tw0 = std::chrono::steady_clock::now();
              if (BamReader::append_records(w.region, pin_out[s] + B.out_off[(size_t)k], (size_t)B.out_len[(size_t)k], werr) < 0) { fail("[strling] error reading " + bam + ": " + werr); break; }
              const auto tw1 = std::chrono::steady_clock::now();
              ns_region += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(tw1 - tw0).count();
              const bool ok = rules(w, tasks[j], done[j], werr);
              ns_rules += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw1).count();
              if (!ok) { fail(werr); break; }
            }
          });
        });
        { std::lock_guard<std::mutex> lk(pm); n_done = bi + 1; }
        pcv.notify_all();
      }
      { std::lock_guard<std::mutex> lk(pm); stop_stages = true; }
      pcv.notify_all();
      reader.join();
      fetcher.join();
      fetcher2.join();
    }
    if (failed.load()) quit("%s", fail_msg.c_str());
    for (const Done &d : done) {
      if (!d.keep) continue;
      calls.push_back(d.c);
      fprintf(bounds_fh, "%s\t%d\n", d.row.c_str(), d.depth);
    }
    t_evidence += std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count();
  };
  const uint64_t nt = (uint64_t)info.n_reads;
  hvec<strl_tread> taken_copy;                         // assigned reads are genotyped with the split they came with (copied only when loci are given)

  // loci handed in with -l / -b are genotyped first and take their reads out of the table (call.nim:150-218)
  std::vector<strl_locus> given;
  std::vector<uint32_t> assigned;
  {
    std::vector<strl_locus> loci;
    if (a.flag("loci")) { loci = parse_bed(a.get("loci", ""), rd.targets(), (uint32_t)window); fprintf(stderr, "Read %zu loci from %s\n", loci.size(), a.get("loci", "").c_str()); }
    if (a.flag("bounds")) { given = parse_bounds(a.get("bounds", ""), rd.targets()); fprintf(stderr, "Read %zu bounds from %s\n", given.size(), a.get("bounds", "").c_str()); }
    for (strl_locus &bound : given)                                                  // loci overwrite the bound they overlap (:160-169)
      for (size_t i = 0; i < loci.size(); ++i)
        if (loci_overlap(loci[i].b, bound.b)) {
          memcpy(bound.name, loci[i].name, sizeof bound.name);
          bound.b.left = loci[i].b.left; bound.b.right = loci[i].b.right;
          loci[i] = loci.back(); loci.pop_back();                                    // seq.del: the last element fills the hole
          break;
        }
    for (const strl_locus &l : loci) given.push_back(l);
    if (!given.empty()) {
      taken_copy = treads;
      std::vector<uint64_t> aoff(given.size() + 1);
      assigned.resize((size_t)std::max<uint64_t>(nt, 1));
      CHECK(strl_assign_reads_loci(treads.data(), nt, STRL_MODE_CALL, given.data(), given.size(), aoff.data(), assigned.data(), assigned.size()));
      // the reads keep the split they had: strl_assign_reads_loci only re-marks its own copies in `treads`
      std::vector<Task> tasks;
      for (size_t j = 0; j < given.size(); ++j) {
        const strl_locus &L = given[j];
        if (L.b.right - L.b.left > 1000u) { fprintf(stderr, "large bounds: %s:%u-%u skipping\n", rd.targets()[(size_t)L.b.tid].name.c_str(), L.b.left, L.b.right); continue; }
        tasks.push_back(Task{L.b, L.name, &assigned, aoff[j], aoff[j + 1], taken_copy.data()});
      }
      run_tasks(tasks);
    }
  }

  // discovery: group, sort, cluster, bounds on the device (call.nim:118-130,221-235)
  const auto tc0 = std::chrono::steady_clock::now();
  const auto tc1 = tc0;
  const uint16_t max_clip_dist = (uint16_t)(0.5 * (double)frag_median);             // call.nim:232
  rvec<strl_bounds> bounds(std::max<size_t>((size_t)nt, 16));        // (left uninitialised: the pass writes rows [0, nb) / [0, nu))
  rvec<strl_unplaced> unplaced(std::max<size_t>((size_t)nt, 16));
  uint64_t nb = 0, nu = 0;
  CHECK(strl_cluster(ctx, treads.data(), nt, STRL_MODE_CALL, (uint32_t)window, min_support, min_clip, min_clip_total, max_clip_dist, bounds.data(),
                     bounds.size(), &nb, unplaced.data(), unplaced.size(), &nu, nullptr));
  std::vector<uint64_t> moff((size_t)nb + 1);
  uint64_t nm = 0;
  CHECK(strl_cluster_members(ctx, moff.data(), nullptr, 0, &nm));
  std::vector<uint32_t> members((size_t)std::max<uint64_t>(nm, 1));
  CHECK(strl_cluster_members(ctx, moff.data(), members.data(), members.size(), &nm));
  const auto tc2 = std::chrono::steady_clock::now();
  {
    std::vector<Task> tasks;
    tasks.reserve((size_t)nb);
    for (uint64_t j = 0; j < nb; ++j) tasks.push_back(Task{bounds[(size_t)j], nullptr, &members, moff[(size_t)j], moff[(size_t)j + 1], treads.data()});
    run_tasks(tasks);
  }
  if (verbose)
    fprintf(stderr, "[strling] seconds: device context + .bin + index side by side, the fragment-length sample behind the context %.3f (context %.3f | .bin %.3f | context + sample + index %.3f)  clustering (upload, sort, sweep, bounds, members) %.3f  evidence + genotypes of %llu bounds on %d threads %.3f (summed over the threads: region records %.3f, spanners + genotype %.3f; regions through the device %llu, on the host %llu: "
            "index + block headers %.3f, page-locked buffers %.3f, reads %.3f, device fetch %.3f (two at a time), the workers waited %.3f for them, %.1f MB compressed -> %.1f MB inflated -> %.1f MB of records)  since the start %.3f\n",
            t_start_up, t_up_ctx, t_up_bin, t_up_bam, std::chrono::duration<double>(tc2 - tc1).count(), (unsigned long long)nb, n_workers, t_evidence, (double)ns_region.load() * 1e-9, (double)ns_rules.load() * 1e-9,
            (unsigned long long)n_dev_regions, (unsigned long long)n_host_regions, t_plan, t_pin, t_pread, t_fetch, t_wait_fetch, (double)dev_comp_bytes / 1e6, (double)dev_inflated / 1e6, (double)dev_kept / 1e6,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call0).count());
  char row[2048];
  std::vector<uint64_t> order(std::max<size_t>(calls.size(), 1)), uorder(std::max<size_t>((size_t)nu, 1));
  CHECK(strl_calls_finish(calls.data(), calls.size(), unplaced.data(), nu, order.data()));   // :264-278
  for (size_t k = 0; k < calls.size(); ++k) {
    const strl_call &c = calls[(size_t)order[k]];
    strl_call_row(row, sizeof row, &c, rd.targets()[(size_t)c.tid].name.c_str());
    fprintf(gt_fh, "%s\n", row);
  }
  CHECK(strl_unplaced_order(unplaced.data(), nu, uorder.data()));                   // :280-281
  for (uint64_t k = 0; k < nu; ++k) fprintf(unplaced_fh, "%s\t%lld\n", unplaced[(size_t)uorder[k]].repeat, (long long)unplaced[(size_t)uorder[k]].count);
  fclose(gt_fh); fclose(bounds_fh); fclose(unplaced_fh);
  if (verbose) {
    fprintf(stderr, "Supporting evidence used to make the genotype calls:\n");
    fprintf(stderr, "wrote putative str bounds to %s\n", pb.c_str());
    fprintf(stderr, "wrote counts of unplaced reads with STR content to %s\n", pu.c_str());
    fprintf(stderr, "Main results file:\n");
    fprintf(stderr, "wrote genotypes to %s\n", pg.c_str());
  }
  return end_process(ctx);
}

// `strling _dump BAM`: SAM-like text of every record as the reader decoded it (reader self-check; needs no GPU)
static int dump_main(int argc, char **argv) {
  if (argc < 3) quit("usage: strling _dump BAM [stream [BATCH]]");
  if (getenv("STRL_CRAM_FASTA")) g_cram_fasta = getenv("STRL_CRAM_FASTA");
  const bool stream = argc > 3 && std::string(argv[3]) == "stream";   // the multi-threaded whole-file reader instead of the plain one
  const int64_t batch = argc > 4 ? atoll(argv[4]) : 4096;
  BamReader rd;
  BamStream rs;
  std::string err;
  if (stream ? !rs.open(argv[2], decode_threads(), err) : !rd.open(argv[2], err)) quit_open(argv[2], err);
  fputs((stream ? rs.header_text() : rd.header_text()).c_str(), stdout);
  RecordBatch b;
  for (;;) {
    b.clear();
    const int64_t got = stream ? rs.read(b, batch, err) : rd.read(b, batch, err);
    if (got < 0) quit("[strling] error reading %s: %s", argv[2], err.c_str());
    if (got == 0) break;
    for (size_t i = 0; i < (size_t)got; ++i) {
      std::string cig, seq;
      for (uint32_t c = b.cigar_off[i]; c < b.cigar_off[i + 1]; ++c) cig += std::to_string(b.cigar[c] >> 4) + "MIDNSHP=X"[b.cigar[c] & 15];
      for (int j = 0; j < b.l_seq[i]; ++j) seq += "=ACMGRSVTWYHKDBN"[(b.seq4[b.seq_off[i] + (size_t)(j >> 1)] >> ((~j & 1) << 2)) & 15];
      printf("%s\t%u\t%d\t%d\t%u\t%s\t%d\t%d\t%d\t%s\n", b.qnames.substr(b.qname_off[i], b.qname_off[i + 1] - b.qname_off[i]).c_str(), b.flag[i],
             b.tid[i], b.pos[i], b.mapq[i], cig.empty() ? "*" : cig.c_str(), b.mtid[i], b.mpos[i], b.isize[i], seq.empty() ? "*" : seq.c_str());
    }
  }
  return 0;
}

// `strling _region BAM TID BEG END`: qname and start of every record the indexed region read returns that passes htslib's
// iterator filter (reader self-check; needs no GPU)
// strling _decode BAM [BATCH]: the multi-threaded reader alone (inflate + record scan + parse into batches), for timing
static int decode_main(int argc, char **argv) {
  if (argc < 3) quit("usage: strling _decode BAM [BATCH] [nosum]");
  const bool nosum = argc > 4 && strcmp(argv[4], "nosum") == 0;      // rate measurements: the single-threaded checksum below is left out
  if (getenv("STRL_CRAM_FASTA")) g_cram_fasta = getenv("STRL_CRAM_FASTA");
  const int64_t batch = argc > 3 ? atoll(argv[3]) : 1048576;
  BamStream rs;
  std::string err;
  if (!rs.open(argv[2], decode_threads(), err)) quit("couldn't open bam");
  RecordBatch b;
  int64_t n = 0;
  uint64_t sum = 0xcbf29ce484222325ull;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    b.clear();
    const int64_t got = rs.read(b, batch, err);
    if (got < 0) quit("[strling] error reading %s: %s", argv[2], err.c_str());
    if (got == 0) break;
    n += got;
    // order-sensitive checksum over every field of every record (FNV-1a over the values, not over the batch layout: the
    // same file must give the same sum whatever the batch size, thread count, superchunk size or inflate engine)
    auto mix = [&](uint64_t v) { sum = (sum ^ v) * 0x100000001b3ull; };
    for (size_t i = 0; i < (nosum ? 0 : (size_t)got); ++i) {
      mix((uint64_t)(uint32_t)b.tid[i]); mix((uint64_t)(uint32_t)b.pos[i]); mix((uint64_t)(uint32_t)b.mtid[i]); mix((uint64_t)(uint32_t)b.mpos[i]);
      mix((uint64_t)(uint32_t)b.isize[i]); mix(b.flag[i]); mix(b.mapq[i]); mix((uint64_t)(uint32_t)b.l_seq[i]);
      for (uint32_t c = b.cigar_off[i]; c < b.cigar_off[i + 1]; ++c) mix(b.cigar[c]);
      const uint8_t *sq = b.seq4.data() + b.seq_off[i];
      for (size_t k = 0; k < (size_t)(b.l_seq[i] + 1) / 2; ++k) mix(sq[k]);
      for (uint64_t k = b.qname_off[i]; k < b.qname_off[i + 1]; ++k) mix((uint8_t)b.qnames[(size_t)k]);
    }
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "[strling] decoded %lld records in %.3f s with %d threads: %.1f reads/s (checksum %llu)\n", (long long)n, s, decode_threads(),
          (double)n / std::max(s, 1e-9), (unsigned long long)sum);
  return 0;
}

static int region_main(int argc, char **argv) {
  if (argc < 6) quit("usage: strling _region BAM TID BEG END [plan]");
  BamReader rd;
  std::string err;
  if (getenv("STRL_CRAM_FASTA")) g_cram_fasta = getenv("STRL_CRAM_FASTA");
  if (!rd.open(argv[2], err) || !rd.load_index(argv[2], err)) quit("couldn't open bam: %s", err.c_str());
  const int32_t tid = atoi(argv[3]);
  const int64_t beg = atoll(argv[4]), end = atoll(argv[5]);
  RecordBatch b;
  int64_t got = -1;
  if (argc > 6 && !strcmp(argv[6], "plan")) {
    // what `strling call` does for the bounds' evidence, with zlib in the place of the device: the region's run of blocks from
    // the index (plan_region), inflated, walked by strl_regions_fetch's rule (keep from the first record whose cigar could
    // reach past `beg`, stop at the first record on another reference or at / behind `end`), parsed by append_records
    const int fd = open(argv[2], O_RDONLY);
    if (fd < 0) quit("couldn't open bam");
    RegionPlan P;
    plan_region(rd, fd, tid, beg, end, P);
    if (!P.ok) { fprintf(stderr, "plan: host reader\n"); got = rd.read_region(b, tid, beg, end, err); }
    else {
      std::vector<uint8_t> comp((size_t)(P.c_end - P.c_beg)), u;
      if (pread(fd, comp.data(), comp.size(), (off_t)P.c_beg) != (ssize_t)comp.size()) quit("short read");
      size_t o = 0;
      for (size_t k = 0; k < P.bsize.size(); ++k) {
        const size_t at = u.size();
        u.resize(at + P.isz[k]);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) quit("zlib");
        zs.next_in = comp.data() + o + P.hdr[k]; zs.avail_in = P.bsize[k] - P.hdr[k] - 8;
        zs.next_out = u.data() + at; zs.avail_out = P.isz[k];
        if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.total_out != P.isz[k]) quit("inflate");
        inflateEnd(&zs);
        if ((uint32_t)crc32(0, u.data() + at, P.isz[k]) != P.crc[k]) quit("crc");
        o += P.bsize[k];
      }
      size_t p = P.in_block, keep = SIZE_MAX;
      bool stopped = false;
      while (p + 36 <= u.size()) {
        int32_t bs, ref, pos;
        memcpy(&bs, &u[p], 4); memcpy(&ref, &u[p + 4], 4); memcpy(&pos, &u[p + 8], 4);
        if (bs < 32) break;
        if (ref != tid || pos >= end) { stopped = true; break; }
        if (p + 4 + (size_t)bs > u.size()) break;
        if (keep == SIZE_MAX) {
          const uint32_t l_name = u[p + 12];
          uint16_t n_cig;
          memcpy(&n_cig, &u[p + 16], 2);
          int64_t span = 1;
          for (uint32_t k = 0; k < n_cig; ++k) { uint32_t c; memcpy(&c, &u[p + 36 + l_name + 4 * k], 4); span += c >> 4; }
          if ((int64_t)pos + span > beg) keep = p;
        }
        p += 4 + (size_t)bs;
      }
      if (!stopped) { fprintf(stderr, "plan: blocks end before the query does, host reader\n"); got = rd.read_region(b, tid, beg, end, err); }
      else {
        if (keep == SIZE_MAX) keep = p;
        fprintf(stderr, "plan: %zu blocks, %zu inflated bytes, records in [%zu, %zu)\n", P.bsize.size(), u.size(), keep, p);
        got = BamReader::append_records(b, u.data() + keep, p - keep, err);
      }
    }
    close(fd);
  } else got = rd.read_region(b, tid, beg, end, err);
  if (got < 0) quit("[strling] error reading %s: %s", argv[2], err.c_str());
  int64_t kept = 0;
  for (size_t i = 0; i < b.size(); ++i) {
    int64_t rl = 0;
    if (!(b.flag[i] & 4)) for (uint32_t c = b.cigar_off[i]; c < b.cigar_off[i + 1]; ++c) { const uint32_t op = b.cigar[c] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += b.cigar[c] >> 4; }
    const int64_t stop = b.pos[i] + (rl ? rl : 1);
    if (!(b.tid[i] == tid && b.pos[i] < end && stop > beg)) continue;
    printf("%s\t%d\t%u\n", b.qnames.substr(b.qname_off[i], b.qname_off[i + 1] - b.qname_off[i]).c_str(), b.pos[i], b.flag[i]);
    ++kept;
  }
  fprintf(stderr, "read %lld records, %lld in region\n", (long long)got, (long long)kept);
  return 0;
}

// genome_strs.nim:175-205
static int index_main(int argc, char **argv) {
  const char *usage =
      "str index\n\nUsage:\n  str index [options] fasta\n\nArguments:\n  fasta            path to fasta file\n\nOptions:\n"
      "  -g, --genome-repeats=GENOME_REPEATS\n                             optional path to output genome repeats file. if it does not exist, it will be created (default: ./<FASTA>.str)\n"
      "  -p, --proportion-repeat=PROPORTION_REPEAT\n                             proportion of read that is repetitive to be considered as STR (default: 0.8)\n  -h, --help                 Show this help\n";
  if (argc <= 2) { fputs(usage, stdout); return 0; }
  const Args a = parse(argc, argv, 2, {{"genome-repeats", 'g', true}, {"proportion-repeat", 'p', true}}, usage);
  if (a.pos.size() != 1) quit("expected 1 argument (fasta)\n%s", usage);
  const std::string fasta = a.pos[0];
  std::string out = a.get("genome-repeats", "");
  if (out.empty()) {                                                               // lastPathPart(fasta) & ".str"
    const size_t sl = fasta.find_last_of('/');
    out = (sl == std::string::npos ? fasta : fasta.substr(sl + 1)) + ".str";
  }
  if (!file_exists(fasta)) quit("[strling] couldn't open fasta %s make sure file is present and has a .fai index", fasta.c_str());
  fprintf(stderr, "Writing genome str index to: %s\n", out.c_str());
  if (file_exists(out)) {                                                          // genome_strs.nim:139-140
    fprintf(stderr, "[strling] using existing file %s for genome repeats\n", out.c_str());
    return 0;
  }
  strl_ctx *ctx = nullptr;
  CHECK(strl_ctx_create(0, &ctx));
  strl_opts opts{0, atof(a.get("proportion-repeat", "0.8").c_str()), 0};
  CHECK(strl_ctx_set_opts(ctx, &opts));
  build_genome_index(ctx, fasta, out);
  strl_ctx_destroy(ctx);
  return 0;
}

// strling _codec nx16|tok3 IN OUT EXPECTED_SIZE  (tests): one CRAM 3.1 block payload through cli/cram_codecs.cpp
static int codec_main(int argc, char **argv) {
  if (argc < 6) quit("usage: strling _codec nx16|tok3|rans4x8|itf8|ltf8 IN OUT EXPECTED_SIZE");
  FILE *f = fopen(argv[3], "rb");
  if (!f) quit("couldn't open %s", argv[3]);
  std::vector<uint8_t> in, out;
  uint8_t buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + got);
  fclose(f);
  std::string err;
  const size_t expect = (size_t)strtoull(argv[5], nullptr, 10);
  const std::string kind = argv[2];
  const bool ok = kind == "tok3" ? cram_tok3_decode(in.data(), in.size(), expect, out, err)
                  : kind == "nx16" ? cram_rans_nx16_decode(in.data(), in.size(), expect, out, err)
                                   : cram_selftest_decode(kind, in.data(), in.size(), expect, out, err);
  if (!ok) quit("[strling] %s", err.c_str());
  f = fopen(argv[4], "wb");
  if (!f || (out.size() && fwrite(out.data(), 1, out.size(), f) != out.size())) quit("couldn't write %s", argv[4]);
  fclose(f);
  return 0;
}

// strling _shares BAM G  (tests): the shares `extract --gpus G` cuts the file into -- per share its first record's virtual
// offset, its blocks, their inflated bytes, the bytes of its last block that are the next share's -- as the walkers deliver them
static int shares_main(int argc, char **argv) {
  if (argc < 4) quit("usage: strling _shares BAM G [max_blocks]");
  const std::string bam = argv[2];
  const int G = std::max(1, atoi(argv[3]));
  const size_t max_blocks = argc > 4 ? (size_t)atoll(argv[4]) : 4096;
  BgzfFeed feed;
  std::string err;
  if (!feed.open(bam, err)) quit("[strling] %s", err.c_str());
  feed.halt();
  const std::vector<uint64_t> cut = share_cuts(feed, bam, G);
  printf("shares\t%zu\n", cut.size());
  for (size_t g = 0; g < cut.size(); ++g) {
    const uint64_t end = g + 1 < cut.size() ? cut[g + 1] : 0;
    BgzfFeed fd;
    if (!fd.open_share(feed, cut[g] >> 16, (uint32_t)(cut[g] & 0xffff), end >> 16, (uint32_t)(end & 0xffff), err)) quit("[strling] %s", err.c_str());
    std::vector<BgzfFeed::Block> bl;
    uint64_t nb = 0, isz = 0, runs = 0, first_c = 0, last_end = 0;
    bool last = false, saw_last = false;
    for (;;) {
      const int64_t n = fd.next(bl, max_blocks, (size_t)1 << 30, err, &last);
      if (n < 0) { printf("share\t%zu\terror\t%s\n", g, err.c_str()); break; }
      if (n == 0) break;
      if (saw_last) quit("[strling] blocks behind the run that was called the last");
      if (!nb) first_c = bl.front().c_off;
      for (const auto &b : bl) { ++nb; isz += b.isize; last_end = b.c_off + b.clen + 8; }
      ++runs;
      saw_last = last;
    }
    printf("share\t%zu\t%llu\t%u\t%llu\t%llu\t%u\t%llu\t%llu\t%llu\t%d\n", g, (unsigned long long)(cut[g] >> 16), (unsigned)(cut[g] & 0xffff), (unsigned long long)nb,
           (unsigned long long)isz, fd.tail_trim(), (unsigned long long)runs, (unsigned long long)first_c, (unsigned long long)last_end, saw_last ? 1 : 0);
  }
  return 0;
}

// seconds since the kernel started this process (its start time in /proc/self/stat against the uptime clock; 10 ms ticks):
// what -v reports of the time in front of main() -- the loader mapping the HIP runtime and this program's device code
static double since_exec() {
  FILE *f = fopen("/proc/self/stat", "r");
  if (!f) return -1;
  char buf[2048];
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = 0;
  const char *q = strrchr(buf, ')');          // (the command name may hold spaces)
  if (!q) return -1;
  unsigned long long start = 0;
  int field = 2;
  for (const char *t = q + 1; *t && field < 22; ++t) if (*t == ' ') { if (++field == 22) start = strtoull(t + 1, nullptr, 10); }
  timespec ts;
  if (!start || clock_gettime(CLOCK_BOOTTIME, &ts) != 0) return -1;
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec - (double)start / (double)sysconf(_SC_CLK_TCK);
}
int main(int argc, char **argv) {
  g_main_at = since_exec();
  const char *top =
      "strling version: 0.6.0 (MI355X-native hot path)\n\nCommands:\n  extract  :   extract informative STR reads from a BAM/CRAM. This is a required first step.\n"
      "  merge    :   merge putitive STR loci from multiple samples. Only required for joint calling.\n  call     :   call STRs\n"
      "  index    :   identify large STRs in the reference genome, to produce ref.fasta.str.\n";      // strling.nim:19-22 (pull_region, a debugging writer, is out of scope)
  if (argc < 2) { fputs(top, stdout); return 1; }
  const std::string cmd = argv[1];
  if (cmd == "extract") return extract_main(argc, argv);
  if (cmd == "merge") return merge_main(argc, argv);
  if (cmd == "index") return index_main(argc, argv);
  if (cmd == "call") return call_main(argc, argv);
  if (cmd == "_dump") return dump_main(argc, argv);
  if (cmd == "_decode") return decode_main(argc, argv);
  if (cmd == "_region") return region_main(argc, argv);
  if (cmd == "_shares") return shares_main(argc, argv);
  if (cmd == "_codec") return codec_main(argc, argv);
  if (cmd == "_indexed_records") {     // (tests) the record count the index beside a BAM gives, or "unknown"
    uint64_t n = 0;
    if (argc > 2 && BgzfFeed::indexed_records(argv[2], n)) printf("%llu\n", (unsigned long long)n);
    else printf("unknown\n");
    return 0;
  }
  if (cmd == "pull_region")
    quit("[strling] `%s` is not part of this build (the MI355X path covers index, extract, merge and call; see DESIGN.md section 9)", cmd.c_str());
  fputs(top, stdout);
  return 1;
}
