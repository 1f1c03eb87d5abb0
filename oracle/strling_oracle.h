/*
 * strling_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A literal, single-threaded, plain-C restatement of the STRling (v0.6.0)
 * extract + cluster hot path, used as the parity checker for the HIP path and
 * as the `cpu_baseline` leg of bench.py.  Nothing under strling_amd/ may
 * include, link or call this file.
 *
 * PARITY STATUS: pinned against every known-answer test the reference holds
 * for this path (tests/test_strling.nim, test_utils.nim, test_extract.nim,
 * test_cluster.nim -- transcribed in tests/golden/reference_kats.json).  The
 * real Nim binary cannot be built in this image (no nim / htslib / nimble
 * packages), so behaviours that no reference test pins -- the kmer module's
 * base->2bit table for non-ACGT symbols, msgpack4nim integer widths and Nim
 * 1.6 Table/CountTable slot order -- are "parity unpinned" and each lives
 * behind one small function here.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the STRling repository root).
 */
#ifndef STRLING_ORACLE_H
#define STRLING_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Soft enum, src/strpkg/cluster.nim:14-20 ---- */
enum { ORC_SOFT_LEFT = 0, ORC_SOFT_RIGHT = 1, ORC_SOFT_BOTH = 2, ORC_SOFT_NONE = 3,
       ORC_SOFT_NONE_RIGHT = 4, ORC_SOFT_NONE_LEFT = 5 };

/* ---- tread, src/strpkg/cluster.nim:23-32 ---- */
typedef struct {
  int32_t  tid;
  uint32_t position;
  char     repeat[6];
  uint16_t flag;
  uint8_t  split;
  uint8_t  mapping_quality;
  uint8_t  repeat_count;
  uint8_t  align_length;
  int64_t  qname_id;   /* index of the record whose qname this tread carries, or (merge) the sample index */
  int64_t  src;        /* bookkeeping: record index that produced it (not part of the reference type) */
} orc_tread;

/* ---- Options, src/strpkg/utils.nim:119-127 ---- */
typedef struct {
  int     median_fragment_length;
  double  proportion_repeat;
  uint8_t min_mapq;
} orc_opts;

/* ---- a batch of BAM records, fields as hts-nim exposes them to extract.nim ---- */
typedef struct {
  int64_t n;
  const int32_t  *tid, *pos, *mtid, *mpos;
  const uint16_t *flag;
  const uint8_t  *mapq;
  const uint32_t *cigar_off;   /* n+1 */
  const uint32_t *cigar;       /* BAM encoding: len<<4 | op */
  const uint64_t *seq_off;     /* n, byte offset of each record's 4-bit packed SEQ */
  const int32_t  *l_seq;       /* n */
  const uint8_t  *seq4;        /* BAM nibble packing, high nibble first */
  const uint64_t *qname_off;   /* n+1 */
  const char     *qnames;
} orc_records;

/* genome STR intervals (ref.fasta.str), grouped per tid; iv_off[tid]..iv_off[tid+1];
 * has_chrom[tid] != 0 iff the chromosome name is a key of the table (read_bed.nim:30-50) */
typedef struct {
  int32_t n_tid;
  const uint8_t *has_chrom;
  const int64_t *iv_off;      /* n_tid+1 */
  const int32_t *iv_start, *iv_stop; /* sorted by start within each tid */
  const int32_t *max_len;            /* [n_tid] longest interval of the tid (Lapper.max_len) */
} orc_genome_str;

/* kmer module (brentp/nim-kmer, not vendored): base -> 2-bit code and back. */
unsigned orc_kmer_code(char c);
char     orc_kmer_base(unsigned code);

/* utils.nim:10-34 */
int  orc_slide_by(const char *s, int len, int k, uint64_t *out);
/* utils.nim:236-271 */
void orc_get_repeat(const char *read, int len, double proportion_repeat, char rep[6], int *repeat_count);
/* utils.nim:220-233 */
int  orc_reduce_repeat(char rep[6]);
/* utils.nim:61-80 */
void orc_min_rev_complement(char rep[6]);
/* utils.nim:304-310 */
void orc_canonical_repeat(const char in[6], char out[6]);
/* utils.nim:139-146 */
int  orc_median(const uint32_t frag[4096], double pct);
/* extract.nim:56-58 */
double orc_p_repeat(const orc_tread *t);
/* extract.nim:141-179 */
int  orc_adjust_by(orc_tread *A, const orc_tread *B, const orc_opts *o, uint32_t B_position);
/* extract.nim:182-190 */
int  orc_unplaced_pair(const orc_tread *A, const orc_tread *B, const orc_opts *o);

/* extract.nim:20-40 + 63-87 (to_tread) for record i; also returns the decoded whole-read result */
void orc_to_tread(const orc_records *r, int64_t i, const orc_genome_str *g, const orc_opts *o, orc_tread *out);

/* extract.nim:93-132: soft-clip scan of record i with the (already lowered) proportion p.
 * Writes 0..2 treads to out, returns how many. */
int  orc_add_soft(const orc_records *r, int64_t i, const orc_opts *o, double p, const char read_repeat[6], orc_tread out[2]);

/* extract.nim:192-248 + 308-329: the whole extract loop over `r` (records already filtered to
 * what `for aln in ibam` yields, in file order); n_tail = number of trailing records that
 * `ibam.query("*")` would revisit (tid == -1 block at the end).  Returns number of treads written
 * (at most cap); *n_needed gets the total. */
int64_t orc_extract(const orc_records *r, int64_t n_tail, const orc_genome_str *g, const orc_opts *o,
                    orc_tread *out, int64_t cap, int64_t *n_needed);

/* per-record scorer outputs, for kernel parity (whole read + both soft ends under both lowered
 * thresholds).  Each result = unit (6 chars) + count (after reduce_repeat) + align_length. */
typedef struct { char rep[6]; int32_t count; int32_t align_length; } orc_seg_result;
void orc_score_record(const orc_records *r, int64_t i, const orc_genome_str *g, const orc_opts *o,
                      orc_seg_result *whole, orc_seg_result soft[4] /* L-first, L-after, R-first, R-after */,
                      int *skipped);

/* ---- clustering ---- */
typedef struct {
  int32_t  tid;
  uint32_t left, left_most, right, right_most, center_mass;
  uint16_t n_left, n_right, n_total;
  char     repeat[7];
} orc_bounds;

/* cluster.nim:175-250; reads must be the cluster's reads in order; left_most/right_most = Cluster fields */
void orc_bounds_of(const orc_tread *reads, int64_t n, uint32_t cl_left_most, uint32_t cl_right_most,
                   uint16_t max_clip_dist, orc_bounds *b);

/* a locus read from a file: Bounds with its name column (cluster.nim:75-87) */
typedef struct { orc_bounds b; char name[128]; } orc_locus;

/* One emitted cluster: [first, first+n) indexes into the *sorted group array* handed to the callback. */
typedef void (*orc_cluster_cb)(void *ud, const orc_tread *reads, int64_t n, uint32_t left_most, uint32_t right_most);
/* cluster.nim:323-374 (cluster -> trcluster -> split_cluster) on one (tid, repeat) group sorted by position */
void orc_cluster_group(const orc_tread *reps, int64_t n, uint32_t max_dist, int min_supporting_reads,
                       orc_cluster_cb cb, void *ud);

/* merge.nim:91-187 / call.nim:118-130,221-262 restricted to the bounds they derive from treads:
 * groups by (tid, repeat) in Nim Table slot order, stable sort by position, cluster, gate through
 * callclusters.nim:52-66.  mode 0 = merge (drops tid<0 on load, has_per_sample_reads with
 * qname_id = sample), mode 1 = call (unplaced groups are reported in unplaced[] instead).
 * Returns number of bounds (at most cap).  */
typedef struct { char repeat[7]; int64_t count; } orc_unplaced;
int64_t orc_call_bounds(const orc_tread *treads, int64_t n, int mode, uint32_t window, int min_support,
                        uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist,
                        orc_bounds *out, int64_t cap, orc_unplaced *unpl, int64_t unpl_cap, int64_t *n_unpl);

/* ---- strling index for one chromosome (genome_strs.nim:22-92); -1 if a doAssert of trim() would fire ---- */
int64_t orc_index_chrom(const char *seq, int64_t L, double p, int window_size, int step, int64_t *starts, int64_t *stops,
                        char (*units)[7], int64_t cap);

/* ---- strling call evidence + genotype (collect.nim, spanning.nim, genotyper.nim, call.nim) ---- */
enum { ORC_SPANNING_FRAGMENT = 0, ORC_SPANNING_READ = 1, ORC_OVERLAPPING_READ = 2 };   /* collect.nim:10-13 */
typedef struct {                       /* collect.nim:15-31; repeat = bounds.repeat, qname = qname of record `rec` */
  uint8_t  type, repeat_count, cigar_ins, cigar_del;
  uint32_t frag_len;
  double   frag_pct;
  int64_t  rec;
} orc_support;
typedef struct {                       /* genotyper.nim:29-52 */
  int32_t tid; uint32_t start, stop; char repeat[7];
  double allele1, allele2;
  uint32_t overlapping_reads, anchored_reads, spanning_reads, spanning_pairs, left_clips, right_clips, sum_str_counts;
  float expected_spanning_fragments, pctile;
  int32_t unplaced_reads;
  double depth;
  int is_large;
} orc_gt;
void   orc_cumulative(const uint32_t frag[4096], float cd[4096]);                                     /* spanning.nim:7-20 */
double orc_expected_spanning_probability(const float cd[4096], int64_t start, int64_t stop, int reverse, int64_t event_start,
                                         int64_t event_stop);                                         /* spanning.nim:22-49 */
double orc_percentile(const uint32_t frag[4096], int64_t fragment_length);                            /* utils.nim:129-137 */
int    orc_median_depth(const int64_t *D, int64_t n);                                                 /* utils.nim:148-158 */
int    orc_overlapping_read(const orc_records *r, int64_t i, const orc_bounds *b, orc_support *s);    /* collect.nim:97-119 */
int    orc_spanning_fragment(const orc_records *r, const int32_t *isize, int64_t L, int64_t R, const orc_bounds *b, orc_support *s,
                             const uint32_t frag[4096]);                                              /* collect.nim:36-48 */
int64_t orc_spanners(const orc_records *r, const int32_t *isize, const orc_bounds *b, int window, const uint32_t frag[4096],
                     uint8_t min_mapq, orc_support *out, int64_t cap, int *median_depth, float *expected_spanners);   /* collect.nim:132-182 */
void   orc_spanning_read_est(const orc_support *reads, int64_t n, double *allele1_bp, double *allele2_bp, double *allele1_ru,
                             double *allele2_ru, uint32_t *supporting);                               /* genotyper.nim:61-98 */
void   orc_genotype(const orc_bounds *b, const orc_tread *tandems, int64_t nt, const uint64_t *tq_off, const char *tqnames,
                    const orc_support *spanners, int64_t ns, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                    int median_fragment_length, double depth, orc_gt *c);                           /* genotyper.nim:150-199 */
int    orc_call_row(char *buf, int cap, const orc_gt *c, const char *chrom);                        /* genotyper.nim:56-57 */
/* call.nim:111-285 the three output files as text (returns 0; bn/gn/un receive the bytes needed) */
int    orc_call(const orc_tread *treads, int64_t n, const uint64_t *tq_off, const char *tqnames, const orc_records *r, const int32_t *isize,
                const uint32_t frag[4096], const char *const *target_names, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                uint8_t min_mapq, char *bounds_buf, int64_t bcap, char *gt_buf, int64_t gcap, char *unpl_buf, int64_t ucap,
                int64_t *bn, int64_t *gn, int64_t *un, const char *loci_text /* -l, or NULL */, const char *bounds_text /* -b, or NULL */,
                const uint32_t *target_lengths, int n_targets);
/* cluster.nim:111-169 (-1 where the reference quits) and the merge of -b bounds with -l loci, call.nim:160-183 */
int64_t orc_parse_bed(const char *text, const char *const *names, const uint32_t *lengths, int n_targets, uint32_t window, orc_locus *out, int64_t cap);
int64_t orc_parse_bounds(const char *text, const char *const *names, int n_targets, orc_locus *out, int64_t cap);
int64_t orc_merge_loci_bounds(orc_locus *bounds, int64_t nb, orc_locus *loci, int64_t nl, orc_locus *out);
/* merge.nim:154-187 as the text of -bounds.txt, with optional -l loci */
int    orc_merge_text(const orc_tread *treads, int64_t n, uint32_t window, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                      uint16_t max_clip_dist, const char *loci_text, const char *const *target_names, const uint32_t *target_lengths, int n_targets,
                      char *buf, int64_t cap, int64_t *need);

/* reads (indices into treads) of every bound orc_call_bounds(mode 1) returns, in cluster order; member_off is [nb + 1] */
int64_t orc_call_members(const orc_tread *treads, int64_t n, uint32_t window, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                         uint16_t max_clip_dist, int64_t *member_off, int64_t *members, int64_t cap, int64_t *n_members);

/* ---- Nim 1.6 stdlib emulation (hashes.nim / tables.nim) ---- */
uint64_t orc_nim_hash_int(uint64_t x);                 /* hashWangYi1 */
uint64_t orc_nim_hash_bytes(const uint8_t *p, int n);  /* murmurHash */
uint64_t orc_nim_hash_tidrep(int32_t tid, const char rep[6]);
/* CountTable[uint32] built by inc() in the given order; returns key/val of `largest` */
void orc_counttable_largest(const uint32_t *keys, int64_t n, int initial_size, uint32_t *key, int64_t *val, int64_t *n_distinct);

/* ---- .bin (extract.nim:336-346, cluster.nim:38-50, unpack.nim:36-133) ---- */
/* writes header+records to buf (cap bytes); returns bytes needed. qname via callback-free arrays */
int64_t orc_bin_write(uint8_t *buf, int64_t cap, float proportion_repeat, uint8_t min_mapq,
                      const uint32_t frag[4096], const char *sam_header, int32_t hdr_len,
                      const orc_tread *treads, int64_t n, const uint64_t *qname_off, const char *qnames);
/* msgpack one tread (returns bytes written) */
int orc_pack_tread(uint8_t *buf, const orc_tread *t, const char *qname, uint32_t qlen);

/* text row, cluster.nim:262-266 */
int orc_bounds_row(char *buf, int cap, const orc_bounds *b, const char *chrom);

#ifdef __cplusplus
}
#endif
#endif
