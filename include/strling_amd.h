/*
 * strling_amd.h -- C ABI of the MI355X-native STRling extract + cluster hot path.
 *
 * The reference (quinlan-lab/STRling v0.6.0) is a single statically linked Nim program with no
 * plugin / FFI seam; its drop-in contract is CLI + files (.bin, -bounds.txt).  This header is the
 * seam a Nim (or any) host binds instead of the reference's in-process procs: each entry point
 * names the reference code it replaces (paths relative to the STRling repository root).  Plain
 * pointers and sizes only; no C++ or torch types.  All functions return 0 (STRL_OK) or a negative
 * status; strl_last_error() gives the message (thread-local).  The library has NO CPU fallback:
 * every compute entry point needs a HIP device and fails with STRL_ERR_NO_DEVICE without one.
 */
#ifndef STRLING_AMD_H
#define STRLING_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STRL_OK 0
#define STRL_ERR_NO_DEVICE (-1)
#define STRL_ERR_HIP (-2)
#define STRL_ERR_ARG (-3)
#define STRL_ERR_CAPACITY (-4)
#define STRL_ERR_IO (-5)
#define STRL_ERR_FORMAT (-6)
#define STRL_ERR_ASSERT (-7) /* a doAssert of the reference would have fired (e.g. extract.nim:72) */
#define STRL_ERR_LIMIT (-9)  /* more records than one device pass over a whole input takes (2^31 - 16: record indices travel in 31 bits);
                                * the reference has no such cap (extract.nim:308) -- the CLI routes such a file to the streaming host Cache */
#define STRL_ERR_NOMEM (-10) /* device memory exhausted (the per-read state of a whole input is resident: ~130 B per read with the device
                               * front end; 288 GB hold ~2e9 reads).  The CLI repeats the extraction with the host pair logic, which keeps
                               * nothing per read on the device. */
#define STRL_ERR_CRC (-8)    /* a BGZF block inflates, but not to the bytes its CRC-32 names (htslib stops there too) */

#define STRL_MEM_HOST 0
#define STRL_MEM_DEVICE 1

/* Longest read the KERNELS score: a lane's byte-wide class counters are exact while no class can be seen more than 255 times.
 * The reference's uint8 histograms (utils.nim:113-117) simply wrap beyond that (utils.nim:192-195), so a longer read is scored by
 * the library's host twin of the scorer with that arithmetic (csrc/host_score.cpp) and its words are merged into the device's
 * results by record index -- every scoring entry point does so by itself.  STRL_MAX_READ_LEN is what the 16-bit length
 * columns of strl_read_soa / strl_pair_rec hold; a longer record is STRL_ERR_ARG. */
#define STRL_DEVICE_READ_LEN 510
#define STRL_MAX_READ_LEN 65534

typedef struct strl_ctx strl_ctx;

int strl_version(void);
const char *strl_last_error(void);
int strl_device_count(void);

/* One context per GPU / stream (re-entrant per handle; the reference is single-threaded). */
int strl_ctx_create(int device_ordinal, strl_ctx **ctx);
void strl_ctx_destroy(strl_ctx *ctx);
/* The HIP stream (hipStream_t) all kernels of this context are launched on. */
void *strl_ctx_stream(strl_ctx *ctx);
int strl_ctx_sync(strl_ctx *ctx);
/* Free / total memory of the context's device right now (hipMemGetInfo).  No counterpart in the reference (a host program);
 * `strling extract -v` prints what the whole-file resident state takes. */
int strl_ctx_mem_info(strl_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- Options (utils.nim:119-127; extract.nim:255-256,299-300) ---- */
typedef struct {
  int32_t median_fragment_length; /* frag_dist.median, extract.nim:283 */
  double proportion_repeat;       /* -p, default 0.8 */
  uint8_t min_mapq;               /* -q, default 40 */
} strl_opts;
int strl_ctx_set_opts(strl_ctx *ctx, const strl_opts *opts);

/* ---- genome STR intervals: the table genome_repeats() returns (genome_strs.nim:107-141,
 * read_bed.nim:30-50), flattened per BAM tid.  has_chrom[tid] != 0 iff the chromosome is a key of
 * the table (extract.nim:30 `aln.chrom in genome_str`).  Intervals need not be sorted. ---- */
typedef struct {
  int32_t n_tid;
  const uint8_t *has_chrom; /* [n_tid] */
  const int64_t *iv_off;    /* [n_tid+1] */
  const int32_t *iv_start;  /* [iv_off[n_tid]] */
  const int32_t *iv_stop;
} strl_genome_str;
int strl_ctx_set_genome(strl_ctx *ctx, const strl_genome_str *g); /* g == NULL: empty table */

/* ---- BAM records as hts-nim hands them to extract.nim (tid/start/mate_*, flag, mapq, cigar,
 * 4-bit SEQ, qname).  seq_off is a byte offset and must be a multiple of 16; seq4 must have 32
 * readable bytes of slack after the last record. ---- */
typedef struct {
  int64_t n;
  const int32_t *tid, *pos, *mtid, *mpos;
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint32_t *cigar_off; /* [n+1] */
  const uint32_t *cigar;     /* BAM encoding len<<4|op */
  const uint64_t *seq_off;   /* [n] */
  const int32_t *l_seq;      /* [n] */
  const uint8_t *seq4;
  const uint64_t *qname_off; /* [n+1] */
  const char *qnames;
} strl_records;

/* ---- structure-of-arrays read batch the kernels consume (24 B of metadata per read + SEQ) ---- */
#define STRL_CIG_SINGLE_M 1u /* n_cigar == 1 and op == M          (extract.nim:30) */
#define STRL_CIG_FIRST_S 2u  /* n_cigar >= 1 and cigar[0] is S     (extract.nim:98,104) */
#define STRL_CIG_LAST_S 4u   /* n_cigar >= 1 and cigar[last] is S */
#define STRL_CIG_ONE_OP 8u   /* n_cigar == 1 */
#define STRL_CIG_NONE 16u    /* n_cigar == 0 */
typedef struct {          /* == words y, z, w of a scorer queue entry */
  uint32_t seq_off;        /* 16-byte units */
  uint16_t l_seq, clip_l;
  uint16_t clip_r;
  uint8_t cig, mapq;
  uint32_t pad;
} strl_read_meta;
typedef struct {
  uint64_t n;
  const int32_t *tid;      /* [n] */
  const int32_t *pos;      /* [n] aln.start */
  const int32_t *end;      /* [n] aln.stop (bam_endpos) */
  const uint32_t *seq_off; /* [n] offset of the read's 4-bit SEQ in 16-byte units */
  const uint16_t *l_seq;   /* [n] */
  const uint16_t *clip_l;  /* [n] length of cigar[0] if it is S, or if the cigar is one single M op (the align_length of a read the
                            * skip predicate removes, extract.nim:33); else 0 */
  const uint16_t *clip_r;  /* [n] length of cigar[last] if it is S else 0 */
  const uint8_t *mapq;     /* [n] */
  const uint8_t *cig;      /* [n] STRL_CIG_* bits */
  const uint8_t *seq4;     /* BAM nibble packing; every read starts 16-byte aligned */
  uint64_t seq4_bytes;     /* including >= 32 bytes of slack */
  uint32_t max_l_seq;
  int32_t mem;             /* STRL_MEM_HOST or STRL_MEM_DEVICE: where ALL pointers above live */
  const strl_read_meta *meta; /* optional (may be NULL), same memory as the rest: seq_off | l_seq | clip_l | clip_r | cig | mapq of
                            * read i once more as ONE 16-byte row.  The skip-predicate pass removes ~90 % of the reads without
                            * looking at these six fields and gathers them for the kept ones: one row = one cache line per kept
                            * read where the five columns are five.  Host batches get their rows made on the device. */
} strl_read_soa;

/* What the pair logic (Cache.add, extract.nim:192-248 with to_tread, add_soft, adjust_by) reads of one record, as ONE
 * 32-byte row: the join touches a few per cent of the records at random, and a row is one memory transaction where the
 * column arrays of strl_read_soa would be a dozen.  Same `mem` as the read batch the rows belong to. */
typedef struct {
  int32_t tid, pos;      /* aln.tid, aln.start */
  int32_t mtid, mpos;    /* aln.mate_tid, aln.mate_pos */
  int32_t end;           /* aln.stop */
  uint16_t flag, l_seq;
  uint16_t clip_l, clip_r; /* as in strl_read_soa */
  uint8_t mapq, cig;
  uint16_t pad;
} strl_pair_rec;
typedef struct {
  const strl_pair_rec *rec; /* [n] */
  const uint64_t *qhash;    /* [n] 64-bit hash of aln.qname (strl_qname_hash); the Cache is keyed by it */
} strl_pair_soa;
/* Host: the rows of a batch from its records and the arrays strl_soa_from_records derived (out[n]). */
int strl_pair_rows(const strl_records *rec, const int32_t *end, const uint16_t *clip_l, const uint16_t *clip_r, const uint8_t *cig,
                   strl_pair_rec *out);

/* Host: derive the SoA metadata arrays from BAM-native records (replaces the hts-nim accessors
 * used at extract.nim:30-38,83-87,98-119: cigar ops, aln.stop, clip lengths).  Caller provides the
 * output arrays ([n] each); SEQ is shared with `rec` (seq_off/16). */
int strl_soa_from_records(const strl_records *rec, int32_t *end, uint32_t *seq_off16, uint16_t *l_seq, uint16_t *clip_l,
                          uint16_t *clip_r, uint8_t *cig, uint32_t *max_l_seq);

/* ---- scorer results ----
 * packed unit/count word:  bits 0-11 unit code (kmer 2-bit "CATG" code, first base in the high bits),
 * bits 12-14 unit length (0 = empty), bit 15 = read removed by the skip predicate (extract.nim:30-34),
 * bits 16-31 repeat_count (after reduce_repeat, utils.nim:271). */
#define STRL_RES_SKIPPED 0x8000u
#define STRL_RES_K(w) (((w) >> 12) & 7u)
#define STRL_RES_CODE(w) ((w) & 0xfffu)
#define STRL_RES_COUNT(w) ((w) >> 16)
typedef struct {
  uint32_t read_side; /* read index << 1 | side (0 = left clip / cigar[0], 1 = right clip / cigar[last]) */
  uint32_t res_first; /* get_repeat(soft_seq) with p - 0.07      (first-seen branch, extract.nim:241-244) */
  uint32_t res_after; /* get_repeat(soft_seq) with min(p, 0.6)   (after-mate branch,  extract.nim:207-211) */
  uint32_t seg_len;   /* c.len: number of soft-clipped bases scored */
} strl_soft_rec;

typedef struct {
  uint64_t n_reads, n_skipped, n_scored, n_soft_items;
  float ms_classify, ms_score, ms_soft; /* HIP-event kernel times on the context stream (0 if timing off) */
  uint32_t n_stage_b_whole, n_stage_b_soft; /* items whose ladder reaches k = 5 (second scorer launch) */
} strl_score_stats;

/* Score a batch: per read the skip predicate + utils.get_repeat on the whole read
 * (extract.nim:20-40 via to_tread :66), and the soft-clip repeat scan of add_soft
 * (extract.nim:93-116) for every clipped end that add_soft would look at, under both lowered
 * thresholds.  whole[n] and soft[soft_cap] live where soa->mem says.  Soft records come back in
 * unspecified order when mem == DEVICE and sorted by read_side when mem == HOST.
 * Requires strl_ctx_set_opts (and optionally strl_ctx_set_genome) first. */
int strl_score_reads(strl_ctx *ctx, const strl_read_soa *soa, uint32_t *whole, strl_soft_rec *soft, uint64_t soft_cap,
                     uint64_t *n_soft, strl_score_stats *stats);
/* `strling index` scoring (genome_strs.nim:61-92): utils.get_repeat on every window [i*step, min(n, i*step+window))
 * of one chromosome (ASCII, any case; hts-nim fai.get + toUpperAscii), with the context's proportion_repeat.
 * words[n_windows] receives the packed unit/count word of each window (host array; pass NULL to only get
 * *n_windows = ceil(n_bases / step)).  window <= 160. */
int strl_index_chrom(strl_ctx *ctx, const char *seq, uint64_t n_bases, uint32_t window, uint32_t step, uint32_t *words,
                     uint64_t *n_windows);
/* The sequential half of `strling index` (host): merge consecutive windows that carry the same unit (allowing one
 * skipped window), pad by one window on both sides and trim to the first/last unit-sized step that is a rotation of
 * the unit -- repeat_windows genome_strs.nim:76-91 and trim :22-59.  One BED row of <fasta>.str per region
 * (genome_strs.nim:137: chrom, start, stop, unit).  STRL_ERR_ASSERT where trim's doAssert would fire. */
typedef struct { uint64_t start, stop; char unit[8]; } strl_region;
int strl_index_regions(const char *seq, uint64_t n_bases, const uint32_t *words, uint64_t n_windows, uint32_t window,
                       uint32_t step, strl_region *out, uint64_t cap, uint64_t *n_out);

/* The host twin of the scorer (csrc/host_score.cpp): utils.get_repeat (utils.nim:236-271) on ONE read given as text (what
 * hts-nim's aln.sequence returns: "=ACMGRSVTWYHKDBN" letters), of any length, with the reference's own arithmetic -- uint8
 * histogram bins that wrap (utils.nim:192-195), float64 thresholds.  *word = the packed unit/count word.  No device is
 * touched.  This is what scores records of more than STRL_DEVICE_READ_LEN bases inside every scoring entry point. */
int strl_score_read_host(const char *seq, int32_t l_seq, double proportion_repeat, uint32_t *word);

/* Kernel timing with HIP events recorded on the context stream around every kernel of
 * strl_score_reads (a ring of 256 launches).  enable_timing(ctx, 1) resets the ring;
 * strl_ctx_kernel_times synchronises the stream and returns the SUM of the classify / score / soft
 * kernel durations (ms) over the launches recorded since then. */
int strl_ctx_enable_timing(strl_ctx *ctx, int on);
int strl_ctx_kernel_times(strl_ctx *ctx, double ms_sum[3], uint64_t *n_launches);
/* The same per launch of a kernel: classify | stage A, survivor compaction, stage B of the whole reads | soft-item
 * compaction | stage A, compaction, stage B of the segments. */
int strl_ctx_kernel_times_detail(strl_ctx *ctx, double ms_sum[8], uint64_t *n_launches);

/* ---- tread (cluster.nim:23-32); qname is carried as an index (record index in extract, sample
 * index in merge -- merge.nim:118-125 overwrites qname with the sample number) ---- */
#define STRL_SOFT_LEFT 0
#define STRL_SOFT_RIGHT 1
#define STRL_SOFT_BOTH 2
#define STRL_SOFT_NONE 3
#define STRL_SOFT_NONE_RIGHT 4
#define STRL_SOFT_NONE_LEFT 5
#define STRL_SOFT_TAKEN 255 /* not a Soft value: a tread strl_assign_reads_loci took out of the table (see there) */
typedef struct {
  int32_t tid;
  uint32_t position;
  char repeat[6];
  uint16_t flag;
  uint8_t split;
  uint8_t mapping_quality;
  uint8_t repeat_count;
  uint8_t align_length;
  int64_t qname_id;
} strl_tread;

/* Host pair logic: Cache.add over the record stream (extract.nim:192-248 driven by :308-329,
 * including the second visit of the unmapped tail by ibam.query("*")), fed by the scorer outputs.
 * soft must be sorted by read_side.  Writes at most cap treads; *n_out gets the total produced. */
int strl_pair_reads(const strl_records *rec, const strl_opts *opts, const uint32_t *whole, const strl_soft_rec *soft,
                    uint64_t n_soft, int64_t n_tail, strl_tread *out, uint64_t cap, uint64_t *n_out);

/* Streaming form of the pair logic: the reference's Cache (extract.nim:89-91,298) kept alive across batches fed in
 * file order.  Emitted treads accumulate inside the pairer; their qname_id indexes the pairer's own qname arena.
 * The caller replays the unmapped tail a second time itself (extract.nim:326-329). */
typedef struct strl_pairer strl_pairer;
int strl_pairer_create(const strl_opts *opts, strl_pairer **pairer);
void strl_pairer_destroy(strl_pairer *pairer);
int strl_pairer_add(strl_pairer *pairer, const strl_records *rec, const uint32_t *whole, const strl_soft_rec *soft, uint64_t n_soft);
int strl_pairer_result(strl_pairer *pairer, const strl_tread **treads, uint64_t *n, const uint64_t **qname_off,
                       const char **qnames, uint64_t *n_pending);

/* ---- the extract hot loop on the device, end to end (replaces extract.nim:308-329 for one batch that is the whole input):
 * skip predicate + scorer + soft-clip scan (strl_score_reads) and the pair logic (Cache.add with to_tread, add_soft,
 * adjust_by, unplaced_pair; the second visit of the last n_tail records, extract.nim:326-329), all as kernels on the context's
 * streams.  Asynchronous when soa->mem == STRL_MEM_DEVICE: nothing is copied to the host, the treads stay resident in the
 * context in the order of the reference's .bin file (qname_id = record index) for strl_treads_fetch / strl_cluster_resident.
 * Consecutive calls on device-resident input overlap: the pair logic of a batch runs on a side stream of the context while
 * the scorer of the next call's batch runs on the main stream (the context keeps two sets of the buffers involved); every
 * entry point that reads the results waits for the side streams first, so the caller sees the order of its calls.  The
 * input arrays of a call must stay valid and unchanged until a later synchronising call (strl_treads_fetch, strl_ctx_sync).
 * item_cap bounds the records that take part in the join (reads of qname groups with a repeat + their soft-clip records),
 * tread_cap the treads; 0 = defaults from n.  Exceeding either is reported by strl_treads_fetch (STRL_ERR_CAPACITY).
 * Qname groups are keyed by the 64-bit hash alone (two different qnames with equal hashes would be treated as one group). */
int strl_extract_device(strl_ctx *ctx, const strl_read_soa *soa, const strl_pair_soa *pair, int64_t n_tail, uint64_t item_cap,
                        uint64_t tread_cap);
/* The same for an input that arrives in chunks (a BAM file being decoded): the chunks are scored as they come, in file
 * order; per-read rows, qname hashes and scorer results of ALL chunks stay resident in HBM (44 B per read: a 30x genome is
 * ~27 GB of the 288 GB) and the pair logic runs once, at strl_extract_finish, over the whole input -- exactly the
 * reference's single Cache over the whole file (extract.nim:298), with no per-chunk state to carry.  At most 2^31 - 16
 * records.  n_reads_hint (may be 0) pre-sizes the buffers.  Then strl_treads_fetch as above (qname_id = record index over
 * all chunks). */
int strl_extract_begin(strl_ctx *ctx, uint64_t n_reads_hint);
int strl_extract_add(strl_ctx *ctx, const strl_read_soa *chunk, const strl_pair_soa *pair);
int strl_extract_finish(strl_ctx *ctx, int64_t n_tail, uint64_t item_cap, uint64_t tread_cap);
/* Wait for the last strl_extract_device call and copy its treads to the host (out may be NULL to only get the count).
 * STRL_ERR_CAPACITY: a capacity was exceeded (n_out = treads needed when known); STRL_ERR_ASSERT: the reference's
 * doAssert repeat_count < 256 (extract.nim:72) would have fired; STRL_ERR_FORMAT: more than 512 join items share the low 32
 * bits of their qname hash (one qname on hundreds of primary records), or -- checked where the device holds the qnames
 * (strl_front_*) -- two different qnames share one 64-bit hash: repeat with the host pair logic (strl_pair_reads), which
 * keys on the qname string.  Secondary / supplementary records do not take part in the join. */
int strl_treads_fetch(strl_ctx *ctx, strl_tread *out, uint64_t cap, uint64_t *n_out, strl_score_stats *stats);
/* HIP-event times (ms) of the last strl_extract_device call when timing is enabled: soft-clip join items | probe |
 * join sort | replay | order sort + gather. */
int strl_ctx_pair_times(strl_ctx *ctx, double ms[5]);

/* Stable LSD radix sort of (key, value) pairs by key bits [bit_lo, bit_lo + bits) on the device (the sort of the
 * clustering and pairing paths: call.nim:127-130 / merge.nim:132-135 `sort` by position inside a (tid, repeat) group
 * becomes one stable keyed sort).  Host arrays in, sorted in place; n_max >= n sizes the launch like a pipeline that
 * only knows an upper bound of the count would. */
int strl_sort_pairs(strl_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t n, uint64_t n_max, int bit_lo, int bits);

/* The pair rules on single treads, so that known-answer vectors (the reference's tests/test_extract.nim:7-19,
 * tests/test_strling.nim:91-107, tests/test_utils.nim:66-74) can be run through the product's own code: with a context the
 * DEVICE functions the replay kernel calls, with ctx == NULL the host twins the streaming pairer (strl_pairer_*) calls.
 * ADJUST_BY: A.adjust_by(B, opts, B_position), extract.nim:141-179 -> *result = its return value, A updated.
 * UNPLACED_PAIR: extract.nim:182-190 -> *result.  CANONICAL: A.repeat := canonical_repeat(A.repeat), utils.nim:304-316. */
#define STRL_RULE_ADJUST_BY 0
#define STRL_RULE_UNPLACED_PAIR 1
#define STRL_RULE_CANONICAL 2
int strl_pair_rule(strl_ctx *ctx, int op, strl_tread *A, const strl_tread *B, const strl_opts *opts, uint32_t B_position, int *result);

/* 64-bit hash of every record's qname (out[n]).  Qname groups never interact in the pair logic, so a multi-GPU
 * run only has to bring together the records whose hash belongs to a group that can emit (strling_amd/dist.py). */
int strl_qname_hash(const strl_records *rec, uint64_t *out);

/* The extract hot loop end to end on one batch: SoA derivation + device scoring + pair logic
 * (replaces extract.nim:308-329). */
int strl_extract(strl_ctx *ctx, const strl_records *rec, int64_t n_tail, strl_tread *out, uint64_t cap, uint64_t *n_out,
                 strl_score_stats *stats);

/* ---- clustering (cluster.nim:323-374 + :175-250, callclusters.nim:52-66) ---- */
typedef struct {
  int32_t tid;
  uint32_t left, left_most, right, right_most, center_mass;
  uint16_t n_left, n_right, n_total;
  char repeat[7];
} strl_bounds;
typedef struct {
  char repeat[7];
  int64_t count;
} strl_unplaced;
typedef struct {
  uint64_t n_treads, n_groups, n_clusters, n_bounds, n_tie_fixups;
  float ms_sort, ms_sweep, ms_bounds;
} strl_cluster_stats;

#define STRL_MODE_MERGE 0 /* merge.nim:172-187: tid<0 dropped on load, has_per_sample_reads gate, qname_id = sample */
#define STRL_MODE_CALL 1  /* call.nim:223-235: unplaced groups reported, no per-sample gate */
/* Group treads by (tid, repeat), stable-sort by position (call.nim:118-130 / merge.nim:121-135),
 * cluster every group and derive the gated Bounds.  Output order = the reference's order (Nim Table
 * slot order of the groups, clusters in position order within a group).  treads/out are host arrays. */
int strl_cluster(strl_ctx *ctx, const strl_tread *treads, uint64_t n, int mode, uint32_t window, int32_t min_support,
                 uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap,
                 uint64_t *n_out, strl_unplaced *unplaced, uint64_t unplaced_cap, uint64_t *n_unplaced,
                 strl_cluster_stats *stats);

/* The same over the treads the last strl_extract_device call left resident in the context (extract -> call without a
 * host round trip; STRL_MODE_CALL only).  n_tid = number of contigs of the BAM header (every tid < n_tid); pos_bits: width
 * of the position field of the sort key (0 = 32; else >= 2): 1 + the bits that hold every position on a contig, e.g. 29 for
 * a genome whose longest contig is < 2^28 bases -- fewer key bits, fewer sort passes.  The upper half of the field takes
 * the positions adjust_by wrapped below zero (uint32 arithmetic, utils.nim:304-310), in the reference's uint32 order.  With out, n_out, n_unplaced and stats all NULL the call only enqueues the kernels (asynchronous, on the context's side
 * stream; see strl_cluster_collect). */
int strl_cluster_resident(strl_ctx *ctx, int mode, int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support, uint16_t min_clip,
                          uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap, uint64_t *n_out,
                          strl_unplaced *unplaced, uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats);

/* Results of the last asynchronous strl_cluster_resident (all outputs NULL there).  Such a pass runs on a side stream of the
 * context: it overlaps whatever the caller enqueues next -- typically strl_extract_device of the NEXT batch, whose VALU-bound
 * scorer hides the clustering's many small launches -- and may be collected after that call has been issued; the pair logic
 * of the next batch waits for it on the device.  Same outputs and order as strl_cluster. */
int strl_cluster_collect(strl_ctx *ctx, strl_bounds *out, uint64_t cap, uint64_t *n_out, strl_unplaced *unplaced, uint64_t unplaced_cap,
                         uint64_t *n_unplaced, strl_cluster_stats *stats);

/* ---- multi-GPU clustering (one process per GPU) ----
 * The device buffer strl_extract_device left its treads in: *treads (strl_tread[*cap]) and *count (uint32 on the device).
 * A host framework (torch.distributed over RCCL) all-gathers these buffers; nothing is copied to the host. */
int strl_ctx_treads_device(strl_ctx *ctx, void **treads, uint64_t *cap, void **count);
/* The HIP stream (hipStream_t) the tail of the last strl_extract_device call runs on -- the context's main stream, or a side
 * stream when that call overlapped its pair logic with the next batch's scorer.  Enqueue the all-gather there: it is then
 * ordered behind the batch's pair logic (and the .bin-order sort strl_ctx_treads_device adds) and before an asynchronous
 * strl_cluster_gathered, and the whole exchange step overlaps the next strl_extract_device. */
void *strl_ctx_tail_stream(strl_ctx *ctx);
/* `gathered` (device) = world x pad treads, rank-major: rank r's treads are gathered[r * pad .. r * pad + counts[r]) with
 * counts (device, uint32[world]) -- what all_gather_into_tensor over the ranks' padded tread buffers produces.  This rank
 * keeps the treads of the (tid, unit) groups it owns (a hash of the key modulo world, computed on the device), in global
 * (rank, .bin) order, and clusters them like strl_cluster_resident.  Rows of different ranks are disjoint sets of groups;
 * strl_group_order over all treads gives the reference's order of the groups. */
int strl_cluster_gathered(strl_ctx *ctx, const strl_tread *gathered, const uint32_t *counts, int world, uint32_t pad, int rank, int mode,
                          int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support, uint16_t min_clip, uint16_t min_clip_total,
                          uint16_t max_clip_dist, strl_bounds *out, uint64_t cap, uint64_t *n_out, strl_unplaced *unplaced,
                          uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats);

/* ---- the same exchange inside the library: RCCL over xGMI (comm.hip).  The reference is single-threaded per sample
 * (merge.nim:52,89 is its only sharding knob); this is north_star's "RCCL all-gather ... before clustering".
 * Ranks = one process per GPU: rank 0 calls strl_comm_unique_id, the bytes travel by the host framework's own means
 * (torch.distributed broadcast, MPI_Bcast), every rank calls strl_ctx_comm_init; then strl_cluster_exchange per step:
 * .bin-order sort of the resident treads, ncclAllGather of the padded tread buffers and their counts on the stream the
 * batch's tail runs on, strl_cluster_gathered.  `pad` = the most treads a rank contributes, equal on all ranks.
 * Ranks = one process, n contexts (the CLI's --gpus N): strl_ctxs_comm_init (ncclCommInitAll on n different devices; where
 * contexts share a device -- RCCL refuses two ranks per device -- the exchange is made with ordered device copies), then
 * strl_ctxs_cluster_exchange for all of them and strl_cluster_collect per context.
 * strl_exchange_treads: all ranks' treads of the last exchange in (rank, .bin) order, for strl_group_order. */
/* host treads -> the context's resident treads (as if strl_extract_device had left them): how `strling merge --gpus N`
 * hands every context its share of the .bin files' treads before strl_ctxs_cluster_exchange */
int strl_ctx_set_treads(strl_ctx *ctx, const strl_tread *treads, uint64_t n);
#define STRL_COMM_ID_BYTES 128
int strl_comm_unique_id(uint8_t id[STRL_COMM_ID_BYTES]);
int strl_ctx_comm_init(strl_ctx *ctx, int world, int rank, const uint8_t id[STRL_COMM_ID_BYTES]);
int strl_ctxs_comm_init(strl_ctx **ctxs, int n);
int strl_ctx_comm_info(strl_ctx *ctx, int *world, int *rank, int *uses_rccl);
int strl_cluster_exchange(strl_ctx *ctx, uint32_t pad, int mode, int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support, uint16_t min_clip,
                          uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap, uint64_t *n_out, strl_unplaced *unplaced,
                          uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats);
int strl_ctxs_cluster_exchange(strl_ctx **ctxs, int n, uint32_t pad, int mode, int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support,
                               uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist);
int strl_exchange_treads(strl_ctx *ctx, strl_tread *out, uint64_t cap, uint64_t *n_out);

/* bounds() (cluster.nim:175-250) + the gate of callclusters.nim:52-66 on one bare cluster -- reads sorted by position,
 * Cluster.left_most = right_most = 0 -- run by the device function the clustering kernels call.  *good = the gate's verdict. */
int strl_bounds_bare(strl_ctx *ctx, const uint32_t *positions, const uint8_t *splits, uint32_t n, uint16_t min_clip, uint16_t min_clip_total,
                     uint16_t max_clip_dist, strl_bounds *out, int *good);

/* HIP-event times (ms) of the last clustering pass when timing is enabled: keys + sort + group tables | ends + walk | bounds */
int strl_ctx_cluster_times(strl_ctx *ctx, double ms[3]);

/* The (tid, unit) groups of `treads` in the order the reference iterates its Table (call.nim:223, merge.nim:172) -- the
 * order strl_cluster emits the groups' rows in.  Host only.  A multi-GPU run that clusters disjoint sets of groups on
 * different ranks puts the gathered rows back into the reference's order with it (strling_amd/dist.py). */
typedef struct {
  int32_t tid;
  char repeat[8];
} strl_group_key;
int strl_group_order(const strl_tread *treads, uint64_t n, int mode, strl_group_key *out, uint64_t cap, uint64_t *n_groups);

/* The reads of every bound the last strl_cluster call on this context returned, in cluster order (position-sorted,
 * stable): bound j holds treads[members[member_off[j]]] .. treads[members[member_off[j+1] - 1]] (indices into the
 * array given to strl_cluster; c.reads of call.nim:246).  member_off is [n_bounds + 1]. */
int strl_cluster_members(strl_ctx *ctx, uint64_t *member_off, uint32_t *members, uint64_t cap, uint64_t *n_members);

/* Loci given on the command line (-l BED / -b BOUNDS) take their reads before clustering: assign_reads_locus,
 * callclusters.nim:14-50, for every locus in order.  The reads of the locus' (tid, unit) group whose position lies in
 * [left_most - 1, right_most] go to the locus (assigned[assigned_off[j] .. assigned_off[j+1]), position order), the read
 * right behind that range is lost like in the reference (:34-36), n_total/n_left/n_right of the locus are recounted.
 * Taken and lost treads get split = STRL_SOFT_TAKEN in place: strl_cluster then ignores them as reads but still
 * counts them as keys of the reference's Table, which keeps its row order.  mode as for strl_cluster. */
typedef struct {
  strl_bounds b;
  char name[128]; /* Bounds.name, column 5 of -bounds.txt */
} strl_locus;
int strl_assign_reads_loci(strl_tread *treads, uint64_t n, int mode, strl_locus *loci, uint64_t n_loci, uint64_t *assigned_off,
                           uint32_t *assigned, uint64_t cap);

/* Re-run the device side of the last strl_cluster call (sorts, sweep, bounds) over the treads still resident on
 * the device, asynchronously on the context stream and without host synchronisation.  For timing. */
int strl_cluster_replay(strl_ctx *ctx);

/* ---- `strling call` evidence around one bound and the genotype record (host; collect.nim, spanning.nim,
 * genotyper.nim).  The caller performs the indexed BAM read (hts-nim `b.query(tid, start, stop)`, collect.nim:141)
 * and hands the records over; `isize` is aln.isize of each record. ---- */
#define STRL_SPANNING_FRAGMENT 0 /* collect.nim:10-13 SupportType */
#define STRL_SPANNING_READ 1
#define STRL_OVERLAPPING_READ 2
typedef struct { /* collect.nim:15-31 Support; repeat = the bound's unit, qname = qname of record `rec` */
  uint8_t type, repeat_count, cigar_ins, cigar_del;
  uint32_t fragment_length;
  double fragment_percentile;
  int64_t rec;
} strl_support;
typedef struct {
  int32_t median_depth;    /* -1: more than 20 000 read pairs in the window (collect.nim:171-174) */
  float expected_spanners;
  uint64_t n_support;
} strl_span_summary;
/* spanners(), collect.nim:132-182: out receives the Support list in the reference's order; records outside the
 * queried region [max(0, left - window), right + window) are ignored, so `r` may be a superset of the query result. */
int strl_spanners(const strl_records *r, const int32_t *isize, const strl_bounds *b, int32_t window, const uint32_t frag[4096],
                  uint8_t min_mapq, strl_support *out, uint64_t cap, strl_span_summary *sum);

typedef struct { /* the Options fields genotype() reads (utils.nim:119-127, call.nim:106-110) */
  int32_t median_fragment_length;
  int32_t min_support;
  uint16_t min_clip, min_clip_total;
} strl_call_opts;
typedef struct { /* genotyper.nim:29-52 Call */
  int32_t tid;
  uint32_t start, stop;
  char repeat[7];
  double allele1, allele2;
  uint32_t overlapping_reads, anchored_reads, spanning_reads, spanning_pairs, left_clips, right_clips, sum_str_counts;
  float expected_spanning_fragments, spanning_fragments_oe_percentile;
  int32_t unplaced_reads;
  double depth;
  int32_t is_large;
} strl_call;
/* genotype(), genotyper.nim:150-199 */
int strl_genotype(const strl_bounds *b, const strl_tread *members, uint64_t n_members, const uint64_t *qname_off, const char *qnames,
                  const strl_support *spanners, uint64_t n_spanners, const strl_call_opts *opts, double depth, strl_call *call);
/* After every bound: add_percentile (call.nim:38-48), the single-large-expansion refinement (:264-276) and the row
 * order of -genotype.txt (Table[string, seq[Call]] by canonical unit).  calls in the order they were made;
 * order[n] receives the indices in output order. */
int strl_calls_finish(strl_call *calls, uint64_t n, const strl_unplaced *unplaced, uint64_t n_unplaced, uint64_t *order);
/* Row order of -unplaced.txt (CountTable[string], call.nim:280-281) for the unplaced list strl_cluster returned */
int strl_unplaced_order(const strl_unplaced *unplaced, uint64_t n, uint64_t *order);
/* one -genotype.txt row (genotyper.nim:54-57), without newline */
int strl_call_row(char *buf, int cap, const strl_call *call, const char *chrom);
/* canonical_repeat, utils.nim:304-316 */
void strl_canonical_repeat(const char in[6], char out[6]);

/* ---- BGZF / BAM front end on the device (the reference reads the BAM through htslib on one thread, extract.nim:275,289) ----
 * Inflate n raw DEFLATE streams (the payloads of BGZF blocks: RFC 1951, <= 64 KiB inflated each) on the GPU, one wavefront
 * per stream.  comp = all compressed bytes, coff/clen = where each stream sits in it, isize = its inflated size (the BGZF
 * footer's ISIZE); out receives the streams back to back.  STRL_ERR_FORMAT for invalid data or a size mismatch. */
int strl_inflate_blocks(strl_ctx *ctx, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen,
                        const uint32_t *isize, uint32_t n_blocks, uint8_t *out, uint64_t out_bytes);
/* HIP-event time (ms) of the inflate kernel of the last strl_inflate_blocks call (copies excluded). */
int strl_ctx_inflate_ms(strl_ctx *ctx, double *ms);

/* ---- `strling call`'s evidence reads on the device (replaces the per-bound `for aln in ibam.query(tid, left - window,
 * right + window)` of call.nim:196-218 / collect.nim:132-141, i.e. htslib's indexed iterator: bgzf seek + inflate + bam_read1
 * from the .bai linear-index offset up to the first record at or behind `end`) for many bounds at once.
 * The host looks the regions up in the index and hands over, per region, the consecutive BGZF blocks from the linear-index
 * offset on (first_block .. first_block + n_blocks - 1 of the block arrays, which are laid out as for strl_inflate_blocks;
 * crc32 = the blocks' trailers, may be NULL) and in_block = where inside the first block's inflated bytes the index points.
 * On return out[out_off[r], out_off[r] + out_len[r]) holds region r's BAM records (block_size-prefixed, back to back, file
 * order) from the first record that may reach past `beg` up to, not including, the first record with another refID or
 * pos >= end -- a superset of htslib's iterator filter (tid equal, pos < end, bam_endpos > beg), which strl_spanners applies.
 * status[r] = 0: complete; 1: the blocks handed over end before such a record (or do not parse): read that region on the host.
 * out_cap = the sum over the regions of their blocks' ISIZE + 32 bytes per region always suffices; STRL_ERR_CAPACITY leaves the
 * bytes needed in out_off[0].  STRL_ERR_FORMAT / STRL_ERR_CRC as for the front end. */
typedef struct {
  uint32_t first_block, n_blocks;
  uint32_t in_block;
  int32_t tid, beg, end;
} strl_region_req;
int strl_regions_fetch(strl_ctx *ctx, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize,
                       const uint32_t *crc32, uint32_t n_blocks, const strl_region_req *req, uint32_t n_regions, uint8_t *out, uint64_t out_cap,
                       uint64_t *out_off, uint64_t *out_len, uint8_t *status);

/* ---- `strling extract` with the whole BAM front end on the device (replaces extract.nim:275-329: bam open / `for aln in ibam`
 * / `query("*")`, i.e. htslib's inflate + bam_read1 + the hts-nim accessors, for the whole file).  The host walks the BGZF
 * block headers and hands over compressed bytes; inflate, record boundaries, record parsing, the fragment-length words and
 * the qnames stay on the device.  Chunks are pipelined: strl_front_push(i) enqueues copy + inflate + record scan of chunk i
 * on a stream of its own, then parses and scores chunk i - 1 on the context's stream (strl_extract_add semantics: rows,
 * hashes and scorer words of ALL chunks stay resident; finish with strl_extract_finish + strl_treads_fetch).
 * A BGZF block may be handed over only once and blocks must come in file order; records may straddle blocks and chunks.
 *   n_ref                 targets in the BAM header (bounds refID / next_refID of a plausible record)
 *   first_record_offset   bytes between the start of the first pushed block's inflated data and the first record
 *   comp                  the chunk's compressed bytes; PINNED host memory (strl_pinned_alloc) makes the copy asynchronous; it
 *                         may be overwritten once the call after the next one has returned
 *   coff / clen / isize   per block: offset of its DEFLATE payload in comp, payload length, inflated size (BGZF ISIZE)
 *   crc32                 per block: the CRC-32 its BGZF trailer states (checked on the device like htslib checks it), or NULL
 *   done / n_done         summaries of chunks whose scoring has COMPLETED, in file order (push: 0 or 1, finish: up to 2)
 * Errors: STRL_ERR_FORMAT invalid DEFLATE data / ISIZE / malformed record; STRL_ERR_CRC; STRL_ERR_ARG a record's l_seq > STRL_MAX_READ_LEN (records of
 * more than STRL_DEVICE_READ_LEN bases are scored by the host twin, host_score.cpp). */
typedef struct {
  uint64_t n_records;       /* records of the chunk (secondary / supplementary included) */
  uint64_t n_primary;       /* those that are neither (the reference's progress counter, extract.nim:309,315) */
  int64_t last_placed;      /* index within the chunk of the last record with tid >= 0; -1: none */
  uint64_t tail_primary;    /* primary records behind it (the "*" region extract.nim:326 visits again) */
  uint32_t max_l_seq;
  uint32_t scan_slow_segments; /* 16 KiB segments whose guessed record start was wrong (walked again sequentially) */
} strl_front_chunk;
int strl_front_begin(strl_ctx *ctx, int32_t n_ref, uint64_t first_record_offset, uint64_t n_reads_hint);
int strl_front_push(strl_ctx *ctx, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize,
                    const uint32_t *crc32, uint32_t n_blocks, strl_front_chunk *done, int *n_done);
int strl_front_finish(strl_ctx *ctx, strl_front_chunk done[2], int *n_done);
/* Optional, for a caller that reads the file one chunk ahead (the CLI; the loop of extract.nim:308 has no counterpart):
 *   strl_front_reserve   (after _begin) sizes the buffers of both chunks in flight for chunks of up to max_blocks blocks and
 *                        max_comp_bytes compressed bytes, so none is reallocated in the middle of the file
 *   strl_front_stage     starts the copy to the device of the chunk the NEXT push of this context will hand over (which must
 *                        pass the same pointers and sizes) -- or, when that one is staged already, of the chunk after it: up to
 *                        two chunks may be staged, in file order.  `comp` and the tables of a staged chunk live until the
 *                        second push after its own has returned
 *   strl_front_enqueue_after + strl_front_collect = strl_front_push_after in two halves: the first queues the chunk's copy
 *                        (unless staged), inflate and record scan and returns the summary of the chunk two back; the second
 *                        waits for the PREVIOUS chunk's record scan and queues its parse + scoring.  Between them the caller
 *                        stages the chunk after this one: its copy is then in the device's queue before the wait, not behind it. */
int strl_front_reserve(strl_ctx *ctx, uint32_t max_blocks, uint64_t max_comp_bytes);
int strl_front_stage(strl_ctx *ctx, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize,
                     const uint32_t *crc32, uint32_t n_blocks);
int strl_front_enqueue_after(strl_ctx *ctx, strl_ctx *prev, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen,
                             const uint32_t *isize, const uint32_t *crc32, uint32_t n_blocks, strl_front_chunk *done, int *n_done);
int strl_front_collect(strl_ctx *ctx);
/* One file on several GPUs (`strling extract --gpus N`; the reference has no counterpart, extract.nim:275 is one thread): the
 * chunks go round-robin over n contexts, each with its own strl_front_begin (only the context that gets the file's first
 * chunk uses first_record_offset).  strl_front_push_after: like strl_front_push, `prev` = the context the file's previous
 * chunk went to (the partial record in front of this chunk comes from there, over xGMI when that is another device).
 * After every context's strl_front_finish, strl_ctxs_extract_gather moves what the pair logic needs of every record
 * (52 B: row, qname hash, scorer word, name reference; + soft-clip records, names, Bloom bits) to ctxs[0] in file order;
 * ctxs[0] then continues like a one-GPU run: strl_front_fragwords, strl_extract_finish, strl_treads_fetch, strl_front_qnames.
 * chunk_owner[k] / chunk_records[k]: context index and strl_front_chunk.n_records of the file's k-th chunk. */
int strl_front_push_after(strl_ctx *ctx, strl_ctx *prev, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen,
                          const uint32_t *isize, const uint32_t *crc32, uint32_t n_blocks, strl_front_chunk *done, int *n_done);
int strl_ctxs_extract_gather(strl_ctx **ctxs, int n, const uint32_t *chunk_owner, const uint64_t *chunk_records, uint64_t n_chunks);
/* The other way to spread one file (`strling extract --gpus N`, the default when the BAM has an index; extract.nim:308-329 is
 * one loop over the whole file): every context takes ONE CONTIGUOUS SHARE of the blocks, both ends at record starts the .bai
 * names, so a share is a BAM of its own -- strl_front_begin with the first record's offset in the share's first block, plain
 * strl_front_push (no `prev`, nothing carried between contexts, each context fed by its own host thread), and
 *   strl_front_trim_next   before the share's LAST chunk is handed over (strl_front_stage or the push itself): its last
 *                          tail_bytes inflated bytes -- the part of the last block behind the next share's first record --
 *                          are not the share's (the block is inflated and CRC-checked whole; the record scan ends in front of them);
 *   strl_front_tail_bytes  after strl_front_finish: bytes behind the last complete record of the last chunk.  0 for a share
 *                          that ended exactly where the next begins (anything else: the index lied; the caller repeats the
 *                          extraction chunk by chunk, strl_front_push_after).
 * strl_ctxs_extract_gather then takes the shares in order (chunk_owner non-decreasing). */
int strl_front_trim_next(strl_ctx *ctx, uint32_t tail_bytes);
int strl_front_tail_bytes(strl_ctx *ctx, uint32_t *tail_bytes);
/* Host waits of this context's front end (record scan, parse) block in the kernel instead of spinning: for N feeding threads
 * that share fewer than 2 N CPUs.  Call before strl_front_begin. */
int strl_ctx_blocking_waits(strl_ctx *ctx, int on);
/* flag | (isize in [0, 4095] ? isize : 0xffff) << 16 of records [first, first + n) of the file: what
 * fragment_length_distribution (utils.nim:86-111) reads of a record.  Synchronises the context's stream. */
int strl_front_fragwords(strl_ctx *ctx, uint64_t first, uint64_t n, uint32_t *out);
/* ... without waiting: enqueued behind the parse of every chunk handed over so far (strl_front_records of them); `out` must be
 * page-locked; *done_event is waited for (and released) with strl_event_wait, from any thread. */
int strl_front_fragwords_async(strl_ctx *ctx, uint64_t first, uint64_t n, uint32_t *out, void **done_event);
int strl_event_wait(void *event);
int strl_front_records(strl_ctx *ctx, uint64_t *n);
/* Gives up the extraction the front end fed; the context stays usable (its front-end buffers stay allocated until the context
 * goes or the next strl_front_begin).  For a caller that ran the front end over a PREFIX of a file -- `strling call`'s
 * fragment-length sample (call.nim:92, utils.nim:86-111: the first ~2.1 M records with a positive template length) -- and goes on to
 * cluster and fetch regions on the same context.  Everything the front end had in flight has completed on return. */
int strl_front_end(strl_ctx *ctx);
/* seen[tid] != 0: the contig has had a primary record so far (extract.nim:310-313 prints a line per large one) */
int strl_front_tids(strl_ctx *ctx, uint8_t *seen, int32_t n_ref);
/* qnames of the given records (a tread's qname_id) from the device's name arena: names[qname_off[i], qname_off[i + 1]).
 * STRL_ERR_CAPACITY with *need set when `cap` is too small. */
int strl_front_qnames(strl_ctx *ctx, const int64_t *record_ids, uint64_t n, uint64_t *qname_off, char *names, uint64_t cap, uint64_t *need);
/* strl_treads_fetch + strl_front_qnames in one go (no host round trips in between): the treads in .bin order (qname_id = record
 * index) and names[qname_off[i], qname_off[i + 1]) of tread i.  STRL_ERR_CAPACITY with *n_out / *names_need set when a buffer is too small. */
int strl_front_treads_named(strl_ctx *ctx, strl_tread *treads, uint64_t cap, uint64_t *n_out, uint64_t *qname_off, char *names, uint64_t names_cap,
                            uint64_t *names_need);
/* page-locked host memory for strl_front_push's compressed bytes */
void *strl_pinned_alloc(uint64_t bytes);
void strl_pinned_free(void *p);

/* ---- fragment-length statistics (utils.nim:139-146) ---- */
int strl_frag_median(const uint32_t frag[4096], double pct);

/* ---- files ---- */
/* .bin writer (extract.nim:332-348 + cluster.nim:38-50).  qname_off/qnames index by tread.qname_id. */
int strl_bin_write(const char *path, float proportion_repeat, uint8_t min_mapq, const uint32_t frag[4096],
                   const char *sam_header, int32_t header_len, const strl_tread *treads, uint64_t n,
                   const uint64_t *qname_off, const char *qnames);
/* .bin reader (unpack.nim:58-133).  Two-call protocol: with treads == NULL only the counts/sizes are
 * returned.  qname bytes are concatenated into qnames with offsets in qname_off ([n+1]). */
typedef struct {
  float proportion_repeat;
  uint8_t min_mapq;
  uint32_t frag[4096];
  int32_t header_len;
  int32_t n_reads;
  uint64_t qnames_bytes;
} strl_bin_info;
int strl_bin_read(const char *path, strl_bin_info *info, char *sam_header, strl_tread *treads, uint64_t *qname_off,
                  char *qnames);
/* The header part alone (unpack.nim:61-110 up to n_reads), without walking the records: info->qnames_bytes receives an UPPER
 * BOUND of the names' bytes (what is left of the file), so that one strl_bin_read call with buffers of these sizes reads a .bin of
 * millions of treads in a single pass. */
int strl_bin_peek(const char *path, strl_bin_info *info);
/* one -bounds.txt row (cluster.nim:262-266), without newline; returns length or <0 */
int strl_bounds_row(char *buf, int cap, const strl_bounds *b, const char *chrom);

#ifdef __cplusplus
}
#endif
#endif
