/*
 * strling_oracle.c -- TEST INFRASTRUCTURE ONLY (see strling_oracle.h).
 *
 * Literal CPU restatement of STRling v0.6.0's extract + cluster path.  Written to be read
 * side by side with the Nim source; it keeps the reference's data layout (ASCII read string,
 * uint8 per-k histograms with a running argmax, greedy substring recount, sequential sweeps)
 * on purpose -- it is the checker and the timed "port" CPU baseline, never the product.
 */
#include "strling_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------
 * kmer module (brentp/nim-kmer >= 0.2.2; strling.nimble:20 -- third party, NOT in the tree).
 * Published algorithm: 2 bits per base, first base in the highest bits, code order "CATG"
 * (C=0, A=1, T=2, G=3; every other byte maps to 1 = 'A').  The order is pinned by the
 * reference's own regression vector at src/strpkg/genome_strs.nim:203-206: a window whose
 * get_repeat unit is "CACGAT" -- the minimum rotation of that hexamer under C<A<T<G, whereas
 * A<C<G<T would have produced "ACGATC".  The non-ACGT mapping is unpinned by any test.
 * ---------------------------------------------------------------------------------------- */
unsigned orc_kmer_code(char c) {
  switch (c) {
    case 'C': case 'c': return 0;
    case 'A': case 'a': return 1;
    case 'T': case 't': return 2;
    case 'G': case 'g': return 3;
    default: return 1;
  }
}
char orc_kmer_base(unsigned code) { return "CATG"[code & 3]; }

static uint64_t kmer_encode(const char *s, int k) {
  uint64_t f = 0;
  for (int i = 0; i < k; i++) f = (f << 2) | orc_kmer_code(s[i]);
  return f;
}
static inline uint64_t forward_add(uint64_t f, char base, int k) {
  uint64_t mask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  return ((f << 2) | orc_kmer_code(base)) & mask;
}
static void kmer_decode(uint64_t e, char *s, int k) {
  for (int i = k; i > 0; i--) { s[i - 1] = orc_kmer_base((unsigned)(e & 3)); e >>= 2; }
}

/* utils.nim:10-34  slide_by: one minimum-rotation code per NON-overlapping window */
int orc_slide_by(const char *s, int len, int k, uint64_t *out) {
  int n = 0;
  if (k <= len) {
    uint64_t f = kmer_encode(s, k);
    uint64_t kmin = f;
    for (int j = 0; j < k; j++) {            /* :17-20 rotate, no new bases */
      f = forward_add(f, s[j], k);
      if (f < kmin) kmin = f;
    }
    out[n++] = kmin;
    for (int i = k; i <= (len - 1) - k + 1; i += k) {   /* :25 countup(k, s.high - k + 1, k) */
      for (int m = 0; m < k; m++) f = forward_add(f, s[i + m], k);
      kmin = f;
      for (int j = 0; j < k; j++) {
        f = forward_add(f, s[i + j], k);
        if (f < kmin) kmin = f;
      }
      out[n++] = kmin;
    }
  }
  return n;
}

/* utils.nim:113-117,181-203  Seq[uint8] with running argmax */
typedef struct { long imax; uint8_t *A; int n; } orc_seq;
static uint8_t g_A2[16], g_A3[64], g_A4[256], g_A5[1024], g_A6[4096];
static orc_seq g_counts[7] = {
  {-1, NULL, 0}, {-1, NULL, 0}, {-1, g_A2, 16}, {-1, g_A3, 64}, {-1, g_A4, 256}, {-1, g_A5, 1024}, {-1, g_A6, 4096}};

static inline void seq_inc(orc_seq *s, uint64_t enc) {      /* :192-195 */
  s->A[enc]++;                                               /* uint8 wraps */
  if (s->imax == -1 || s->A[enc] > s->A[s->imax]) s->imax = (long)enc;
}
static inline void seq_clear(orc_seq *s) {                   /* :200-203 */
  if (s->imax == -1) return;
  memset(s->A, 0, (size_t)s->n);
  s->imax = -1;
}
/* utils.nim:205-211 */
static int count_k(const char *read, int len, int k, orc_seq *c) {
  static uint64_t codes[1 << 15];
  seq_clear(c);
  int n = orc_slide_by(read, len, k, codes);
  for (int i = 0; i < n; i++) seq_inc(c, codes[i]);
  if (c->imax == -1) return 0;
  return c->A[c->imax];
}

/* Nim strutils.count(s, sub): greedy, non-overlapping, left to right (lib/pure/strutils.nim) */
static int str_count(const char *s, int len, const char *sub, int k) {
  int c = 0, i = 0;
  while (i + k <= len) {
    if (memcmp(s + i, sub, (size_t)k) == 0) { c++; i += k; } else i++;
  }
  return c;
}

/* utils.nim:220-233 */
int orc_reduce_repeat(char rep[6]) {
  int result = 1;
  if (rep[0] == '\0') return result;
  char seen = rep[0];
  for (int i = 1; i < 6; i++) {
    if (rep[i] == '\0') break;
    if (rep[i] != seen) return result;
  }
  for (int i = 1; i < 6; i++) {
    if (rep[i] == '\0') break;
    result++;
    rep[i] = '\0';
  }
  return result;
}

/* utils.nim:236-271 */
void orc_get_repeat(const char *read, int len, double proportion_repeat, char rep[6], int *repeat_count) {
  memset(rep, 0, 6);
  *repeat_count = 0;
  int nN = 0;
  for (int i = 0; i < len; i++) nN += (read[i] == 'N');
  if (nN > 20) return;                                       /* :238 */
  char s[8];
  int best_score = -1;
  for (int k = 2; k <= 6; k++) {
    int count = count_k(read, len, k, &g_counts[k]);         /* :243 */
    uint64_t am = (uint64_t)g_counts[k].imax;                /* :197-198, -1 -> all ones */
    kmer_decode(am, s, k);                                   /* :245 */
    int score = count * k;
    if (score <= best_score) {                               /* :250-253 */
      if (count < (int)((double)len * 0.12 / (double)k)) break;
      continue;
    }
    count = str_count(read, len, s, k);                      /* :254 */
    score = count * k;
    if (score < best_score) continue;
    best_score = score;
    if (count > (int)((double)len * proportion_repeat / (double)k)) {   /* :259-263 */
      memcpy(rep, s, (size_t)k);
      *repeat_count = count;
    }
  }
  *repeat_count *= orc_reduce_repeat(rep);                   /* :271 */
}

/* utils.nim:37-53 */
static char complement(char c) {
  switch (c) { case 'C': return 'G'; case 'G': return 'C'; case 'A': return 'T'; case 'T': return 'A'; default: return c; }
}
/* utils.nim:61-80 */
void orc_min_rev_complement(char rep[6]) {
  char s[16]; int l = 0;
  for (int i = 0; i < 6; i++) { if (rep[i] == 0) break; s[l++] = rep[i]; }
  if (l == 0) return;            /* never reached by the reference (callers guarantee a unit) */
  char rc[16];
  for (int i = 0; i < l; i++) rc[l - 1 - i] = complement(s[i]);
  char dbl[16];
  memcpy(dbl, rc, (size_t)l); memcpy(dbl + l, rc, (size_t)l);
  uint64_t mv = ~0ULL, codes[4];
  int n = orc_slide_by(dbl, 2 * l, l, codes);                /* l == 0 -> k <= len holds; encode("")==0 */
  for (int i = 0; i < n; i++) if (codes[i] < mv) mv = codes[i];
  char ms[8];
  kmer_decode(mv, ms, l);
  for (int i = 0; i < l; i++) rep[i] = ms[i];
}
/* utils.nim:291-310 */
void orc_canonical_repeat(const char in[6], char out[6]) {
  char r[6];
  memcpy(r, in, 6);
  orc_min_rev_complement(r);
  int lt = 0;
  for (int i = 0; i < 6; i++) {
    if (i < 5) { if (r[i] != in[i]) { lt = ((unsigned char)r[i] < (unsigned char)in[i]); break; } }
    else lt = ((unsigned char)r[5] < (unsigned char)in[5]);
  }
  memcpy(out, lt ? r : in, 6);
}

/* utils.nim:139-146 */
int orc_median(const uint32_t frag[4096], double pct) {
  uint32_t n = 0;
  for (int i = 0; i < 4096; i++) n += frag[i];
  uint32_t count = 0;
  for (int i = 0; i < 4096; i++) {
    count += frag[i];
    if (count >= (uint32_t)(0.5 + (double)n / (1.0 / pct))) return i;
  }
  return 4096;
}

/* extract.nim:51-58 */
static uint8_t repeat_length(const orc_tread *t) {
  uint8_t r = 0;
  for (int i = 0; i < 6; i++) { if (t->repeat[i] == 0) return r; r++; }
  return r;
}
double orc_p_repeat(const orc_tread *t) {
  uint8_t prod = (uint8_t)(t->repeat_count * repeat_length(t));     /* uint8 arithmetic */
  uint8_t al = t->align_length < 1 ? 1 : t->align_length;
  return (double)prod / (double)al;
}

#define FLAG_PAIRED 0x1
#define FLAG_PROPER 0x2
#define FLAG_REVERSE 0x10
#define FLAG_MREVERSE 0x20
#define FLAG_SECONDARY 0x100
#define FLAG_SUPPL 0x800
#define FLAG_UNMAP 0x4

/* extract.nim:134-139 */
static int should_reverse(uint16_t f) {
  int result = !(f & FLAG_MREVERSE);
  if (f & FLAG_REVERSE) result = !result;
  return result;
}

/* extract.nim:141-179 */
int orc_adjust_by(orc_tread *A, const orc_tread *B, const orc_opts *o, uint32_t B_position) {
  if (A->repeat_count == 0) return 0;
  uint32_t half = (uint32_t)((double)((float)A->align_length / 2.0f) + 0.5);   /* uint32(A.align_length.float / 2'f + 0.5) */
  if (B->mapping_quality > o->min_mapq &&
      ((orc_p_repeat(A) > o->proportion_repeat && orc_p_repeat(B) < 0.2) ||
       (!(A->flag & FLAG_PROPER) && A->mapping_quality < o->min_mapq))) {
    if (B->flag & FLAG_REVERSE) {
      A->position = B_position - (uint32_t)o->median_fragment_length + B->align_length + half;
      if (B->split == ORC_SOFT_NONE_LEFT) A->position = B_position;
    } else {
      A->position = B_position + (uint32_t)o->median_fragment_length - half;
      if (B->split == ORC_SOFT_NONE_RIGHT) A->position = B_position + (uint32_t)B->align_length;
    }
    A->split = ORC_SOFT_NONE;
    A->tid = B->tid;
    if (B->mapping_quality > A->mapping_quality) A->mapping_quality = B->mapping_quality;
    if (should_reverse(A->flag)) orc_min_rev_complement(A->repeat);
  } else if (A->mapping_quality >= o->min_mapq || (A->flag & FLAG_PROPER)) {
    A->position += half;
    if (B->mapping_quality > A->mapping_quality) A->mapping_quality = B->mapping_quality;
  }
  return 1;
}

/* extract.nim:182-190 */
int orc_unplaced_pair(const orc_tread *A, const orc_tread *B, const orc_opts *o) {
  if (orc_p_repeat(A) > o->proportion_repeat && orc_p_repeat(B) > o->proportion_repeat) return 1;
  if (orc_p_repeat(A) > o->proportion_repeat && B->mapping_quality < o->min_mapq) return 1;
  if (orc_p_repeat(B) > o->proportion_repeat && A->mapping_quality < o->min_mapq) return 1;
  return 0;
}

/* ---- hts-nim record accessors used by extract.nim ---- */
static const char NT16[] = "=ACMGRSVTWYHKDBN";               /* htslib seq_nt16_str */
static int rec_sequence(const orc_records *r, int64_t i, char *out) {
  const uint8_t *p = r->seq4 + r->seq_off[i];
  int L = r->l_seq[i];
  for (int j = 0; j < L; j++) out[j] = NT16[(p[j >> 1] >> ((~j & 1) << 2)) & 0xf];
  return L;
}
static int cig_n(const orc_records *r, int64_t i) { return (int)(r->cigar_off[i + 1] - r->cigar_off[i]); }
static int cig_op(const orc_records *r, int64_t i, int j) { return (int)(r->cigar[r->cigar_off[i] + j] & 0xf); }
static int cig_len(const orc_records *r, int64_t i, int j) { return (int)(r->cigar[r->cigar_off[i] + j] >> 4); }
#define CIG_M 0
#define CIG_S 4
/* htslib bam_endpos */
static int64_t rec_stop(const orc_records *r, int64_t i) {
  int64_t rlen = 0;
  if (!(r->flag[i] & FLAG_UNMAP)) {
    int n = cig_n(r, i);
    for (int j = 0; j < n; j++) {
      int op = cig_op(r, i, j);
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cig_len(r, i, j);   /* M D N = X */
    }
  }
  if (rlen == 0) rlen = 1;
  return (int64_t)r->pos[i] + rlen;
}
/* lapper find(): any interval with iv.start < stop and iv.stop > start (brentp/nim-lapper) */
/* Lapper.find (nim-lapper src/lapper.nim): intervals sorted by start, lowerBound(start - max_len), then a forward
 * scan that stops at the first interval starting at or after `stop`.  Callers hand the intervals over sorted by
 * start per tid (lapify sorts them, read_bed.nim:45-47); max_len[tid] = longest interval. */
static int genome_find(const orc_genome_str *g, int32_t tid, int64_t start, int64_t stop) {
  int64_t lo = g->iv_off[tid], hi = g->iv_off[tid + 1];
  const int64_t end = hi, key = start - (int64_t)g->max_len[tid];
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)g->iv_start[mid] < key) lo = mid + 1; else hi = mid;
  }
  for (int64_t j = lo; j < end; j++) {
    if ((int64_t)g->iv_start[j] < stop && (int64_t)g->iv_stop[j] > start) return 1;
    else if ((int64_t)g->iv_start[j] >= stop) break;
  }
  return 0;
}

/* extract.nim:20-40 */
static void rec_get_repeat(const orc_records *r, int64_t i, const orc_genome_str *g, const orc_opts *o,
                           char rep[6], int *repeat_count, int *align_length, int *skipped) {
  memset(rep, 0, 6);
  *repeat_count = 0;
  *skipped = 0;
  int32_t tid = r->tid[i];
  int in_tbl = (g != NULL && tid >= 0 && tid < g->n_tid && g->has_chrom[tid]);
  if (cig_n(r, i) == 1 && cig_op(r, i, 0) == CIG_M && in_tbl) {              /* :30 */
    if (!genome_find(g, tid, (int64_t)r->pos[i], rec_stop(r, i))) {          /* :32 */
      *align_length = cig_len(r, i, 0);
      *skipped = 1;
      return;
    }
  }
  static char read[1 << 16];
  int L = rec_sequence(r, i, read);
  *align_length = L;
  orc_get_repeat(read, L, o->proportion_repeat, rep, repeat_count);          /* :40 */
}

/* extract.nim:63-87 */
void orc_to_tread(const orc_records *r, int64_t i, const orc_genome_str *g, const orc_opts *o, orc_tread *t) {
  int repeat_count, align_length, skipped;
  char rep[6];
  rec_get_repeat(r, i, g, o, rep, &repeat_count, &align_length, &skipped);
  if (repeat_count >= 256) { fprintf(stderr, "oracle: doAssert repeat_count < 256 (extract.nim:72)\n"); abort(); }
  memset(t, 0, sizeof *t);
  t->tid = r->tid[i];
  t->position = (uint32_t)(r->pos[i] < 0 ? 0 : r->pos[i]);
  memcpy(t->repeat, rep, 6);
  t->flag = r->flag[i];
  t->repeat_count = (uint8_t)repeat_count;
  t->align_length = (uint8_t)align_length;
  t->split = ORC_SOFT_NONE;
  t->mapping_quality = r->mapq[i];
  t->qname_id = i;
  t->src = i;
  int L = cig_n(r, i);
  if (L > 1 && cig_op(r, i, 0) == CIG_S && cig_len(r, i, 0) > 16) t->split = ORC_SOFT_NONE_LEFT;
  if (L > 1 && cig_op(r, i, L - 1) == CIG_S && cig_len(r, i, L - 1) > 16) t->split = ORC_SOFT_NONE_RIGHT;
}

/* extract.nim:93-132 */
int orc_add_soft(const orc_records *r, int64_t i, const orc_opts *o, double p, const char read_repeat[6], orc_tread out[2]) {
  int nout = 0;
  if (r->mapq[i] < o->min_mapq) return 0;                                    /* :97 */
  int L = cig_n(r, i);
  if (L == 0 || (cig_op(r, i, 0) != CIG_S && cig_op(r, i, L - 1) != CIG_S)) return 0;   /* :98 */
  static char seq[1 << 16];
  int idxs[2] = {0, L - 1};
  for (int q = 0; q < 2; q++) {                                              /* :102 */
    int ci = idxs[q];
    if (cig_op(r, i, ci) != CIG_S) continue;
    int clen = cig_len(r, i, ci);
    if (read_repeat[0] == 0 && clen <= 16) continue;                         /* :106 */
    int sl = rec_sequence(r, i, seq);
    const char *soft = (ci == 0) ? seq : seq + (sl - clen);                  /* :109-112 */
    char rep[6]; int repeat_count;
    orc_get_repeat(soft, clen, p, rep, &repeat_count);                       /* :114 */
    if (repeat_count == 0) continue;
    int64_t posn = (ci == 0) ? (int64_t)r->pos[i] : rec_stop(r, i);          /* :119 */
    if (posn < 0) posn = 0;
    orc_tread tr;
    memset(&tr, 0, sizeof tr);
    tr.tid = r->tid[i];
    tr.position = (uint32_t)posn;
    tr.flag = r->flag[i];
    memcpy(tr.repeat, rep, 6);
    tr.repeat_count = (uint8_t)repeat_count;
    tr.align_length = (uint8_t)clen;
    tr.split = (ci == 0) ? ORC_SOFT_LEFT : ORC_SOFT_RIGHT;
    tr.mapping_quality = r->mapq[i];
    tr.qname_id = i;
    tr.src = i;
    if (orc_p_repeat(&tr) < 0.9) continue;                                   /* :131 */
    out[nout++] = tr;
  }
  return nout;
}

void orc_score_record(const orc_records *r, int64_t i, const orc_genome_str *g, const orc_opts *o,
                      orc_seg_result *whole, orc_seg_result soft[4], int *skipped) {
  int rc, al;
  rec_get_repeat(r, i, g, o, whole->rep, &rc, &al, skipped);
  whole->count = rc; whole->align_length = al;
  memset(soft, 0, 4 * sizeof(orc_seg_result));
  int L = cig_n(r, i);
  if (L == 0) return;
  static char seq[1 << 16];
  int sl = rec_sequence(r, i, seq);
  double p_first = o->proportion_repeat - 0.07;                               /* extract.nim:242 */
  double p_after = o->proportion_repeat < 0.6 ? o->proportion_repeat : 0.6;  /* extract.nim:208 */
  int idxs[2] = {0, L - 1};
  for (int q = 0; q < 2; q++) {
    int ci = idxs[q];
    if (cig_op(r, i, ci) != CIG_S) continue;
    int clen = cig_len(r, i, ci);
    const char *s = (q == 0) ? seq : seq + (sl - clen);
    for (int v = 0; v < 2; v++) {
      orc_seg_result *d = &soft[q * 2 + v];
      int c;
      orc_get_repeat(s, clen, v == 0 ? p_first : p_after, d->rep, &c);
      d->count = c; d->align_length = clen;
    }
  }
}

/* ---- Cache (extract.nim:89-91): qname -> tread table; only membership semantics matter ---- */
typedef struct { int64_t rec; orc_tread t; int used; } cache_ent;
typedef struct { cache_ent *e; int64_t cap; int64_t n; const orc_records *r; } cache_tbl;

static uint64_t fnv(const char *p, int64_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (int64_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ULL; }
  return h;
}
static int qname_eq(const orc_records *r, int64_t a, int64_t b) {
  int64_t la = (int64_t)(r->qname_off[a + 1] - r->qname_off[a]), lb = (int64_t)(r->qname_off[b + 1] - r->qname_off[b]);
  return la == lb && memcmp(r->qnames + r->qname_off[a], r->qnames + r->qname_off[b], (size_t)la) == 0;
}
static void tbl_grow(cache_tbl *t);
static int64_t tbl_find(cache_tbl *t, int64_t rec) {
  uint64_t h = fnv(t->r->qnames + t->r->qname_off[rec], (int64_t)(t->r->qname_off[rec + 1] - t->r->qname_off[rec]));
  int64_t i = (int64_t)(h & (uint64_t)(t->cap - 1));
  while (t->e[i].used) {
    if (t->e[i].used == 1 && qname_eq(t->r, t->e[i].rec, rec)) return i;
    i = (i + 1) & (t->cap - 1);
  }
  return -1;
}
static void tbl_put(cache_tbl *t, int64_t rec, const orc_tread *tr) {
  if ((t->n + 1) * 2 > t->cap) tbl_grow(t);
  uint64_t h = fnv(t->r->qnames + t->r->qname_off[rec], (int64_t)(t->r->qname_off[rec + 1] - t->r->qname_off[rec]));
  int64_t i = (int64_t)(h & (uint64_t)(t->cap - 1));
  while (t->e[i].used == 1) i = (i + 1) & (t->cap - 1);
  if (t->e[i].used == 0) t->n++;            /* tombstones (2) are reused without growing n */
  t->e[i].used = 1; t->e[i].rec = rec; t->e[i].t = *tr;
}
static void tbl_grow(cache_tbl *t) {
  cache_ent *old = t->e; int64_t oc = t->cap;
  t->cap *= 2; t->n = 0;
  t->e = (cache_ent *)calloc((size_t)t->cap, sizeof(cache_ent));
  for (int64_t i = 0; i < oc; i++) if (old[i].used == 1) tbl_put(t, old[i].rec, &old[i].t);
  free(old);
}

typedef struct { orc_tread *out; int64_t cap, n; } out_vec;
static void push(out_vec *v, const orc_tread *t) { if (v->n < v->cap) v->out[v->n] = *t; v->n++; }

/* extract.nim:192-248 */
static void cache_add(cache_tbl *tbl, out_vec *ov, const orc_records *r, int64_t i, const orc_genome_str *g, const orc_opts *o) {
  int32_t tid = r->tid[i], mtid = r->mtid[i];
  int32_t start = r->pos[i], mpos = r->mpos[i];
  int after_mate = (tid > mtid) || (tid == mtid && ((start > mpos) || (start == mpos && tbl_find(tbl, i) >= 0)));   /* :60-61 */
  orc_tread soft[2];
  if (after_mate) {
    int64_t slot = tbl_find(tbl, i);
    if (slot < 0) return;                                                     /* :198 */
    orc_tread mate = tbl->e[slot].t;
    tbl->e[slot].used = 2;                                                    /* take() */
    orc_tread self;
    orc_to_tread(r, i, g, o, &self);                                          /* :204 */
    double p = o->proportion_repeat < 0.6 ? o->proportion_repeat : 0.6;       /* :207-208 */
    int ns = orc_add_soft(r, i, o, p, self.repeat, soft);                     /* :209 */
    for (int k = 0; k < ns; k++) push(ov, &soft[k]);
    if (mate.repeat_count == 0 && self.repeat_count == 0) return;             /* :212 */
    if (orc_unplaced_pair(&self, &mate, o)) {                                 /* :215 */
      if (self.repeat[0] == 0 || mate.repeat[0] == 0) return;                 /* :218 */
      char c[6];
      orc_canonical_repeat(self.repeat, c); memcpy(self.repeat, c, 6);
      self.position = 0; self.tid = -1;
      orc_canonical_repeat(mate.repeat, c); memcpy(mate.repeat, c, 6);
      mate.position = 0; mate.tid = -1;
      push(ov, &self); push(ov, &mate);                                       /* :229-230 */
      return;
    }
    uint32_t mp = mate.position;                                              /* :233 */
    if (orc_adjust_by(&mate, &self, o, self.position)) push(ov, &mate);       /* :234-235 */
    if (orc_adjust_by(&self, &mate, o, mp)) push(ov, &self);                  /* :236-237 */
  } else {
    orc_tread tr;
    orc_to_tread(r, i, g, o, &tr);                                            /* :240 */
    double p = o->proportion_repeat - 0.07;                                   /* :242 */
    int ns = orc_add_soft(r, i, o, p, tr.repeat, soft);
    for (int k = 0; k < ns; k++) push(ov, &soft[k]);
    int64_t slot = tbl_find(tbl, i);                                          /* :245 hasKeyOrPut */
    if (slot >= 0) tbl->e[slot].used = 2;                                     /* :246-248 warn + take; new one NOT stored */
    else tbl_put(tbl, i, &tr);
  }
}

/* extract.nim:308-329 */
int64_t orc_extract(const orc_records *r, int64_t n_tail, const orc_genome_str *g, const orc_opts *o,
                    orc_tread *out, int64_t cap, int64_t *n_needed) {
  cache_tbl tbl; tbl.cap = 8192; tbl.n = 0; tbl.r = r;
  tbl.e = (cache_ent *)calloc((size_t)tbl.cap, sizeof(cache_ent));
  out_vec ov = {out, cap, 0};
  for (int64_t i = 0; i < r->n; i++) {                                        /* :308 for aln in ibam (to EOF) */
    if (r->flag[i] & (FLAG_SECONDARY | FLAG_SUPPL)) continue;
    cache_add(&tbl, &ov, r, i, g, o);
  }
  if (n_tail < 0) { n_tail = 0; while (n_tail < r->n && r->tid[r->n - 1 - n_tail] < 0) n_tail++; }
  for (int64_t i = r->n - n_tail; i < r->n; i++) {                            /* :326 ibam.query("*") revisits the tail */
    if (r->flag[i] & (FLAG_SECONDARY | FLAG_SUPPL)) continue;
    cache_add(&tbl, &ov, r, i, g, o);
  }
  free(tbl.e);
  if (n_needed) *n_needed = ov.n;
  return ov.n < cap ? ov.n : cap;
}

/* ==========================================================================================
 * strling index: src/strpkg/genome_strs.nim:22-92 (one chromosome; seq already upper-cased like fai.get().toUpperAscii)
 * ======================================================================================== */
typedef struct { int64_t start, stop; char repeat[7]; int valid; } idx_window;

/* genome_strs.nim:22-59; returns 0 when one of its doAsserts would fire */
static int window_trim(idx_window *w, const char *dna, int64_t dlen) {
  int k = (int)strlen(w->repeat);
  uint64_t *codes = (uint64_t *)malloc(((size_t)dlen + 2) * sizeof(uint64_t));
  uint64_t expected = 0, one[8];
  if (orc_slide_by(w->repeat, k, k, one) > 0) expected = one[0];
  int n = orc_slide_by(dna, (int)dlen, k, codes);                    /* trim left */
  for (int i = 0; i < n; i++) { if (expected != codes[i]) w->start += k; else break; }
  if (!(w->start < w->stop)) { free(codes); return 0; }
  char *dnar = (char *)malloc((size_t)dlen + 1), rep[8];
  for (int64_t i = 0; i < dlen; i++) dnar[dlen - 1 - i] = dna[i];
  for (int i = 0; i < k; i++) rep[k - 1 - i] = w->repeat[i];
  if (orc_slide_by(rep, k, k, one) > 0) expected = one[0];
  n = orc_slide_by(dnar, (int)dlen, k, codes);                       /* trim right */
  for (int i = 0; i < n; i++) { if (expected != codes[i]) w->stop -= k; else break; }
  free(dnar);
  free(codes);
  return w->start < w->stop;
}

/* genome_strs.nim:61-92 for one chromosome.  out: (start, stop, unit) triples; returns count or -1 on a doAssert */
int64_t orc_index_chrom(const char *seq, int64_t L, double p, int window_size, int step, int64_t *starts, int64_t *stops,
                        char (*units)[7], int64_t cap) {
  int64_t n_out = 0;
  idx_window last; memset(&last, 0, sizeof last); last.stop = -1;
#define YIELD_LAST()                                                                                   \
  do {                                                                                                 \
    if (last.stop != -1 && last.stop - last.start >= (window_size - step)) {                           \
      last.start = last.start - window_size < 0 ? 0 : last.start - window_size;                        \
      last.stop = last.stop + window_size < L ? last.stop + window_size : L;                           \
      idx_window t = last;                                                                             \
      if (!window_trim(&t, seq + last.start, last.stop - last.start)) return -1;                       \
      if (n_out < cap) { starts[n_out] = t.start; stops[n_out] = t.stop; memcpy(units[n_out], t.repeat, 7); } \
      n_out++;                                                                                         \
    }                                                                                                  \
  } while (0)
  for (int64_t start = 0; start < L; start += step) {
    int64_t dl = start + window_size <= L ? window_size : L - start;
    char rep[6]; int rc;
    orc_get_repeat(seq + start, (int)dl, p, rep, &rc);
    if (rc > 0) {
      idx_window w; memset(&w, 0, sizeof w);
      w.start = start; w.stop = start + dl; memcpy(w.repeat, rep, 6);
      if (strcmp(last.repeat, w.repeat) != 0 || w.start > last.stop + (window_size - step)) {
        YIELD_LAST();
        last = w;
      } else last.stop = w.stop;
    }
  }
  YIELD_LAST();
#undef YIELD_LAST
  return n_out;
}

/* ==========================================================================================
 * Nim 1.6 stdlib emulation: lib/pure/hashes.nim, lib/pure/collections/{tables,tableimpl,hashcommon}.nim
 * (Nim 1.6.10 is what the reference CI pins: .github/workflows/ci.yml:11).  Unpinned by any
 * reference test; isolated here.
 * ======================================================================================== */
static uint64_t hi_xor_lo(uint64_t a, uint64_t b) { __uint128_t r = a; r *= b; return (uint64_t)(r >> 64) ^ (uint64_t)r; }
uint64_t orc_nim_hash_int(uint64_t x) {          /* hashWangYi1 */
  const uint64_t P0 = 0xa0761d6478bd642fULL, P1 = 0xe7037ed1a0b428dbULL, P58 = 0xeb44accab455d165ULL ^ 8ULL;
  return hi_xor_lo(hi_xor_lo(P0, x ^ P1), P58);
}
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint64_t orc_nim_hash_bytes(const uint8_t *x, int size) {   /* murmurHash (MurmurHash3_x86_32, seed 0) */
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u, n1 = 0xe6546b64u, m1 = 0x85ebca6bu, m2 = 0xc2b2ae35u;
  int n = size / 4, i = 0;
  uint32_t h1 = 0;
  while (i < n * 4) {
    uint32_t k1 = 0;
    for (int j = 3; j >= 0; j--) k1 = (k1 << 8) | x[i + j];
    i += 4;
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
    h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + n1;
  }
  uint32_t k1 = 0; int rem = size % 4;
  while (rem > 0) { rem--; k1 = (k1 << 8) | x[i + rem]; }
  k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
  h1 ^= (uint32_t)size;
  h1 ^= h1 >> 16; h1 *= m1; h1 ^= h1 >> 13; h1 *= m2; h1 ^= h1 >> 16;
  return (uint64_t)h1;
}
static uint64_t nim_mix(uint64_t h, uint64_t val) {   /* `!&` */
  uint64_t res = h + val; res = res + (res << 10); res = res ^ (res >> 6); return res;
}
static uint64_t nim_finish(uint64_t h) {              /* `!$` */
  uint64_t res = h + (h << 3); res = res ^ (res >> 11); res = res + (res << 15); return res;
}
uint64_t orc_nim_hash_tidrep(int32_t tid, const char rep[6]) {   /* hash(tuple[tid:int32, repeat:array[6,char]]) */
  uint64_t h = 0;
  h = nim_mix(h, orc_nim_hash_int((uint64_t)(int64_t)tid));
  h = nim_mix(h, orc_nim_hash_bytes((const uint8_t *)rep, 6));
  return nim_finish(h);
}
static int64_t next_pow2(int64_t x) { int64_t p = 1; while (p < x) p <<= 1; return p; }
static int64_t slots_needed(int64_t count) { return next_pow2(count * 3 / 2 + 4); }
static int must_rehash(int64_t len, int64_t counter) { return (len * 2 < counter * 3) || (len - counter < 4); }

/* CountTable[uint32]: tables.nim ctRawInsert / rawGet / enlarge / inc / largest */
void orc_counttable_largest(const uint32_t *keys, int64_t n, int initial_size, uint32_t *key, int64_t *val, int64_t *n_distinct) {
  int64_t len = slots_needed(initial_size), counter = 0;
  uint32_t *k = (uint32_t *)calloc((size_t)len, 4);
  int64_t *v = (int64_t *)calloc((size_t)len, 8);
  for (int64_t q = 0; q < n; q++) {
    uint64_t hc = orc_nim_hash_int((uint64_t)keys[q]);
    int64_t h = (int64_t)(hc & (uint64_t)(len - 1));
    int found = 0;
    while (v[h] != 0) { if (k[h] == keys[q]) { found = 1; break; } h = (h + 1) & (len - 1); }
    if (found) { v[h]++; continue; }
    if (must_rehash(len, counter)) {                   /* enlarge: reinsert in slot order */
      int64_t nl = len * 2;
      uint32_t *nk = (uint32_t *)calloc((size_t)nl, 4);
      int64_t *nv = (int64_t *)calloc((size_t)nl, 8);
      for (int64_t i = 0; i < len; i++) if (v[i] != 0) {
        int64_t j = (int64_t)(orc_nim_hash_int((uint64_t)k[i]) & (uint64_t)(nl - 1));
        while (nv[j] != 0) j = (j + 1) & (nl - 1);
        nk[j] = k[i]; nv[j] = v[i];
      }
      free(k); free(v); k = nk; v = nv; len = nl;
    }
    h = (int64_t)(hc & (uint64_t)(len - 1));
    while (v[h] != 0) h = (h + 1) & (len - 1);
    k[h] = keys[q]; v[h] = 1; counter++;
  }
  int64_t mi = 0;                                       /* largest: first max in slot order */
  for (int64_t h = 1; h < len; h++) if (v[mi] < v[h]) mi = h;
  *key = k[mi]; *val = v[mi];
  if (n_distinct) *n_distinct = counter;
  free(k); free(v);
}

/* ==========================================================================================
 * clustering: src/strpkg/cluster.nim
 * ======================================================================================== */
/* cluster.nim:59-62 */
static uint32_t posmed(const orc_tread *reads, int64_t n) {
  int64_t m = n < 9 ? n : 9;
  int mid = (int)((double)m / 2.0 - 0.5);
  return reads[mid].position;
}

/* cluster.nim:175-250 */
void orc_bounds_of(const orc_tread *reads, int64_t n, uint32_t cl_left_most, uint32_t cl_right_most,
                   uint16_t max_clip_dist, orc_bounds *b) {
  memset(b, 0, sizeof *b);
  int rl = 0;
  for (int i = 0; i < 6; i++) { if (reads[0].repeat[i] == 0) break; b->repeat[rl++] = reads[0].repeat[i]; }
  b->tid = reads[0].tid;
  if (n > 65535) { fprintf(stderr, "oracle: doAssert cl.reads.len <= uint16.high (cluster.nim:185)\n"); abort(); }
  b->center_mass = reads[(int64_t)((double)n / 2.0)].position;          /* posns[int(posns.len / 2)] */
  uint32_t *lefts = (uint32_t *)malloc((size_t)n * 4), *rights = (uint32_t *)malloc((size_t)n * 4);
  int64_t nl = 0, nr = 0;
  for (int64_t i = 0; i < n; i++) {                                      /* :192-202 */
    const orc_tread *r = &reads[i];
    if (r->split == ORC_SOFT_LEFT && (int32_t)r->position < (int32_t)b->center_mass + (int32_t)max_clip_dist) {
      lefts[nl++] = r->position; b->n_left++; b->n_total++;
    } else if (r->split == ORC_SOFT_RIGHT && (int32_t)r->position > (int32_t)b->center_mass - (int32_t)max_clip_dist) {
      rights[nr++] = r->position; b->n_right++; b->n_total++;
    } else b->n_total++;
  }
  uint32_t key; int64_t val;
  if (nl > 0) { orc_counttable_largest(lefts, nl, 8, &key, &val, NULL); if (val > 1) b->left = key; }
  if (nr > 0) { orc_counttable_largest(rights, nr, 8, &key, &val, NULL); if (val > 1) b->right = key; }
  free(lefts); free(rights);
  if (b->left == 0) b->left = b->center_mass;                             /* :213-217 (posns.len > 0 always) */
  if (b->right == 0) b->right = b->left + 1;
  if (b->left >= b->right) {                                              /* :227-231 */
    if (b->n_left > 0 && b->n_right > 0) { uint32_t t = b->left; b->left = b->right; b->right = t; }
    else b->left = b->right - 1;
  }
  uint32_t pmin = reads[0].position, pmax = reads[0].position;
  for (int64_t i = 1; i < n; i++) { if (reads[i].position < pmin) pmin = reads[i].position; if (reads[i].position > pmax) pmax = reads[i].position; }
  b->left_most = ((int64_t)cl_left_most > 0) ? cl_left_most : pmin;       /* :234-241 */
  b->right_most = ((int64_t)cl_right_most > 0) ? cl_right_most : pmax;
  if (b->left_most > b->left) b->left_most = b->left;                     /* :244-247 */
  if (b->right_most < b->right) b->right_most = b->right;
}

/* cluster.nim:283-320 */
static void split_cluster(const orc_tread *reads, int64_t n, uint32_t left_most, uint32_t right_most,
                          int min_supporting_reads, orc_cluster_cb cb, void *ud) {
  uint32_t *lefts = (uint32_t *)malloc((size_t)(n ? n : 1) * 4), *rights = (uint32_t *)malloc((size_t)(n ? n : 1) * 4);
  int64_t nl = 0, nr = 0;
  for (int64_t i = 0; i < n; i++) {
    if (reads[i].split == ORC_SOFT_LEFT) lefts[nl++] = reads[i].position;
    else if (reads[i].split == ORC_SOFT_RIGHT) rights[nr++] = reads[i].position;
  }
  if (nr == 0 || nl == 0) { cb(ud, reads, n, left_most, right_most); free(lefts); free(rights); return; }
  uint32_t rlk, llk; int64_t rlv, llv, rdist, ldist;
  orc_counttable_largest(rights, nr, 8, &rlk, &rlv, &rdist);
  orc_counttable_largest(lefts, nl, 8, &llk, &llv, &ldist);
  free(lefts); free(rights);
  if (rlk < llk && rlv >= min_supporting_reads && llv >= min_supporting_reads &&
      (double)llv / (double)ldist > 0.5 && (double)rlv / (double)rdist > 0.5) {          /* :303 */
    uint32_t mid = (uint32_t)(0.5 + ((double)rlk + (double)llk) / 2.0);
    orc_tread *c1 = (orc_tread *)malloc((size_t)n * sizeof(orc_tread)), *c2 = (orc_tread *)malloc((size_t)n * sizeof(orc_tread));
    int64_t n1 = 0, n2 = 0;
    for (int64_t i = 0; i < n; i++) { if (reads[i].position < mid) c1[n1++] = reads[i]; else c2[n2++] = reads[i]; }
    cb(ud, c1, n1, 0, mid - 1);                                           /* :313 c1.right_most = mid - 1 */
    cb(ud, c2, n2, mid, 0);                                               /* :314 c2.left_most = mid */
    free(c1); free(c2);
  } else cb(ud, reads, n, left_most, right_most);
}

static int has_anchor(const orc_tread *reads, int64_t n) {               /* cluster.nim:275-281 */
  for (int64_t i = 0; i < n; i++) if (reads[i].split == ORC_SOFT_NONE) return 1;
  return 0;
}

/* cluster.nim:323-362 -- keeps the explicit growing `c.reads` array (as [s, s+n) of reps) */
static void trcluster(const orc_tread *reps, int64_t len, uint32_t max_dist, int min_supporting_reads, orc_cluster_cb cb, void *ud) {
  int64_t i = 0;
  const orc_tread *c = NULL; int64_t cn = 0;
  while (i < len) {
    c = &reps[i]; cn = 1;                                                 /* :330 */
    i += 1;
    for (int64_t j = i; j <= len - 1; j++) {
      if (reps[j].position <= posmed(c, cn) + max_dist + 100) {           /* :336 uint32 arithmetic */
        cn++;                                                             /* reps are contiguous, c.reads == reps[s .. j] */
        i = j + 1;
        continue;
      }
      /* :342 trim(max_dist + 100) */
      { uint32_t md = max_dist + 100;
        int64_t lo64 = (int64_t)posmed(c, cn) - (int64_t)md; if (lo64 < 0) lo64 = 0;
        uint32_t lo = (uint32_t)lo64;
        while (cn > 1 && c[0].position < lo) { c++; cn--; } }
      uint32_t rm = c[cn - 1].position, pm = posmed(c, cn);
      uint32_t right_most = rm > pm + max_dist ? rm : pm + max_dist;      /* :343 */
      uint32_t lm = c[0].position;
      uint32_t left_most = lm < pm - max_dist ? lm : pm - max_dist;       /* :344 uint32 wrap */
      if (cn >= min_supporting_reads && has_anchor(c, cn)) split_cluster(c, cn, left_most, right_most, min_supporting_reads, cb, ud);
      break;
    }
  }
  if (c == NULL) return;
  /* :354-362 tail: c is the last cluster built (if the loop above broke, `while` restarted and rebuilt it) */
  { uint32_t md = max_dist + 100;
    int64_t lo64 = (int64_t)posmed(c, cn) - (int64_t)md; if (lo64 < 0) lo64 = 0;
    uint32_t lo = (uint32_t)lo64;
    while (cn > 1 && c[0].position < lo) { c++; cn--; } }
  uint32_t rm = c[cn - 1].position, pm = posmed(c, cn);
  uint32_t right_most = rm > pm + max_dist ? rm : pm + max_dist;
  uint32_t lm = c[0].position;
  uint32_t left_most = lm < pm - max_dist ? lm : pm - max_dist;
  if (cn >= min_supporting_reads && has_anchor(c, cn)) split_cluster(c, cn, left_most, right_most, min_supporting_reads, cb, ud);
}

/* cluster.nim:364-374 */
void orc_cluster_group(const orc_tread *reps, int64_t n, uint32_t max_dist, int min_supporting_reads, orc_cluster_cb cb, void *ud) {
  if (n <= 0) return;
  if (reps[0].tid < 0) { cb(ud, reps, n, 0, 0); return; }
  trcluster(reps, n, max_dist, min_supporting_reads, cb, ud);
}

/* ---- driver: merge.nim:91-187 / call.nim:118-130,221-262 ---- */
typedef struct {
  int mode; int min_support; uint16_t min_clip, min_clip_total, max_clip_dist;
  orc_bounds *out; int64_t cap, n;
  orc_unplaced *unpl; int64_t unpl_cap, n_unpl;
  int (*on_bound)(void *ud, orc_bounds *b, const orc_tread *reads, int64_t n);   /* call.nim:237-255; 0 = bound dropped */
  void *hook_ud;
} drv;

/* merge.nim:18-25 */
static int has_per_sample_reads(const orc_tread *reads, int64_t n, int supporting) {
  int64_t best = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t c = 0;
    for (int64_t j = 0; j < n; j++) c += (reads[j].qname_id == reads[i].qname_id);
    if (c > best) best = c;
  }
  return best >= supporting;
}

static void on_cluster(void *ud, const orc_tread *reads, int64_t n, uint32_t left_most, uint32_t right_most) {
  drv *d = (drv *)ud;
  if (reads[0].tid == -1) {                                               /* call.nim:226-228 / merge.nim:175-176 */
    if (d->mode == 1) {
      if (d->n_unpl < d->unpl_cap) {
        orc_unplaced *u = &d->unpl[d->n_unpl];
        memset(u, 0, sizeof *u);
        for (int i = 0; i < 6 && reads[0].repeat[i]; i++) u->repeat[i] = reads[0].repeat[i];
        u->count = n;
      }
      d->n_unpl++;
    }
    return;
  }
  if (d->mode == 0 && !has_per_sample_reads(reads, n, d->min_support)) return;    /* merge.nim:177 */
  /* callclusters.nim:52-66 */
  if (n >= 65535) return;
  orc_bounds b;
  orc_bounds_of(reads, n, left_most, right_most, d->max_clip_dist, &b);
  if (b.right - b.left > 1000u) return;
  if (b.n_left < d->min_clip) return;
  if (b.n_right < d->min_clip) return;
  if ((uint16_t)(b.n_right + b.n_left) < d->min_clip_total) return;
  if (d->on_bound && !d->on_bound(d->hook_ud, &b, reads, n)) return;
  if (d->n < d->cap) d->out[d->n] = b;
  d->n++;
}

/* Nim Table[tid_rep, seq[tread]] slot order: tables.nim mgetOrPut / enlarge; initial size 8192 */
typedef struct { uint64_t hcode; int32_t tid; char rep[6]; int64_t gid; } tslot;

static int cmp_pos_stable(const void *a, const void *b) {
  const orc_tread *x = *(const orc_tread *const *)a, *y = *(const orc_tread *const *)b;
  if (x->position != y->position) return x->position < y->position ? -1 : 1;
  return x < y ? -1 : (x > y ? 1 : 0);        /* algorithm.sort is a stable merge sort */
}

/* loci handed to call/merge on the command line (-l / -b); set by the drivers around call_bounds_impl */
static orc_locus *g_loci = NULL;
static int64_t g_n_loci = 0;
static void (*g_on_locus)(void *ud, orc_locus *L, const orc_tread *reads, int64_t n) = NULL;
static void *g_locus_ud = NULL;
static int64_t call_bounds_impl(const orc_tread *treads, int64_t n, int mode, uint32_t window, int min_support,
                        uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist,
                        orc_bounds *out, int64_t cap, orc_unplaced *unpl, int64_t unpl_cap, int64_t *n_unpl,
                        int (*on_bound)(void *, orc_bounds *, const orc_tread *, int64_t), void *hook_ud);
int64_t orc_call_bounds(const orc_tread *treads, int64_t n, int mode, uint32_t window, int min_support,
                        uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist,
                        orc_bounds *out, int64_t cap, orc_unplaced *unpl, int64_t unpl_cap, int64_t *n_unpl) {
  return call_bounds_impl(treads, n, mode, window, min_support, min_clip, min_clip_total, max_clip_dist, out, cap, unpl, unpl_cap, n_unpl, NULL, NULL);
}
static int64_t call_bounds_impl(const orc_tread *treads, int64_t n, int mode, uint32_t window, int min_support,
                        uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist,
                        orc_bounds *out, int64_t cap, orc_unplaced *unpl, int64_t unpl_cap, int64_t *n_unpl,
                        int (*on_bound)(void *, orc_bounds *, const orc_tread *, int64_t), void *hook_ud) {
  int64_t len = slots_needed(8192), counter = 0;
  tslot *tb = (tslot *)calloc((size_t)len, sizeof(tslot));
  int64_t *gid_of = (int64_t *)malloc((size_t)(n ? n : 1) * 8);
  int64_t ngroups = 0;
  for (int64_t q = 0; q < n; q++) {
    const orc_tread *t = &treads[q];
    if (mode == 0 && t->tid < 0) { gid_of[q] = -1; continue; }          /* unpack_file(drop_unplaced=true), merge.nim:101 */
    uint64_t hc = orc_nim_hash_tidrep(t->tid, t->repeat);
    if (hc == 0) hc = 314159265;
    int64_t h = (int64_t)(hc & (uint64_t)(len - 1));
    int found = 0;
    while (tb[h].hcode != 0) {
      if (tb[h].hcode == hc && tb[h].tid == t->tid && memcmp(tb[h].rep, t->repeat, 6) == 0) { found = 1; break; }
      h = (h + 1) & (len - 1);
    }
    if (!found) {
      if (must_rehash(len, counter)) {
        int64_t nl = len * 2;
        tslot *nt = (tslot *)calloc((size_t)nl, sizeof(tslot));
        for (int64_t i = 0; i < len; i++) if (tb[i].hcode != 0) {
          int64_t j = (int64_t)(tb[i].hcode & (uint64_t)(nl - 1));
          while (nt[j].hcode != 0) j = (j + 1) & (nl - 1);
          nt[j] = tb[i];
        }
        free(tb); tb = nt; len = nl;
        h = (int64_t)(hc & (uint64_t)(len - 1));
        while (tb[h].hcode != 0) h = (h + 1) & (len - 1);
      }
      tb[h].hcode = hc; tb[h].tid = t->tid; memcpy(tb[h].rep, t->repeat, 6); tb[h].gid = ngroups++;
      counter++;
    }
    gid_of[q] = tb[h].gid;
  }
  /* bucket the treads per group preserving input order, then stable sort by position */
  int64_t *gcount = (int64_t *)calloc((size_t)(ngroups + 1), 8);
  for (int64_t q = 0; q < n; q++) if (gid_of[q] >= 0) gcount[gid_of[q] + 1]++;
  for (int64_t gi = 0; gi < ngroups; gi++) gcount[gi + 1] += gcount[gi];
  const orc_tread **ptr = (const orc_tread **)malloc((size_t)(n ? n : 1) * sizeof(void *));
  int64_t *fill = (int64_t *)malloc((size_t)(ngroups + 1) * 8);
  memcpy(fill, gcount, (size_t)(ngroups + 1) * 8);
  for (int64_t q = 0; q < n; q++) if (gid_of[q] >= 0) ptr[fill[gid_of[q]]++] = &treads[q];
  drv d = {mode, min_support, min_clip, min_clip_total, max_clip_dist, out, cap, 0, unpl, unpl_cap, 0, on_bound, hook_ud};
  orc_tread *buf = (orc_tread *)malloc((size_t)(n ? n : 1) * sizeof(orc_tread));
  int64_t *glen = (int64_t *)malloc((size_t)(ngroups + 1) * 8);
  for (int64_t gi = 0; gi < ngroups; gi++) {                              /* call.nim:127-130 / merge.nim:132-135 */
    glen[gi] = gcount[gi + 1] - gcount[gi];
    qsort(ptr + gcount[gi], (size_t)glen[gi], sizeof(void *), cmp_pos_stable);
  }
  /* loci given on the command line take their reads out of the table first (callclusters.nim:14-50) */
  for (int64_t li = 0; li < g_n_loci; li++) {
    orc_locus *L = &g_loci[li];
    char rep6[6] = {0};
    memcpy(rep6, L->b.repeat, strlen(L->b.repeat));
    uint64_t hc = orc_nim_hash_tidrep(L->b.tid, rep6);
    if (hc == 0) hc = 314159265;
    int64_t h = (int64_t)(hc & (uint64_t)(len - 1)), gi = -1;
    while (tb[h].hcode != 0) {
      if (tb[h].hcode == hc && tb[h].tid == L->b.tid && memcmp(tb[h].rep, rep6, 6) == 0) { gi = tb[h].gid; break; }
      h = (h + 1) & (len - 1);
    }
    int64_t nres = 0;
    if (gi >= 0 && glen[gi] > 0) {
      const orc_tread **trs = ptr + gcount[gi];
      int64_t tl = glen[gi];
      uint32_t left_most = L->b.left_most == 0 ? 0u : L->b.left_most - 1u;
      int64_t lo = 0, ri = 0;
      while (lo < tl && trs[lo]->position < left_most) lo++;               /* lowerBound */
      while (ri < tl && trs[ri]->position <= L->b.right_most) ri++;          /* upperBound */
      if (ri < lo) ri = lo;
      nres = ri - lo;
      for (int64_t j = 0; j < nres; j++) buf[j] = *trs[lo + j];
      /* table[key] = trs[0..<li] & (if ri < trs.high: trs[ri+1..high]): trs[ri] itself is lost */
      int64_t w = lo;
      if (ri < tl - 1) for (int64_t j = ri + 1; j < tl; j++) trs[w++] = trs[j];
      glen[gi] = w;
    }
    L->b.n_total = 0; L->b.n_left = 0; L->b.n_right = 0;
    for (int64_t j = 0; j < nres; j++) {
      L->b.n_total++;
      if (buf[j].split == ORC_SOFT_RIGHT) L->b.n_right++;
      else if (buf[j].split == ORC_SOFT_LEFT) L->b.n_left++;
    }
    if (g_on_locus) g_on_locus(g_locus_ud, L, buf, nres);
  }
  for (int64_t h = 0; h < len; h++) {                                      /* mpairs: slot order */
    if (tb[h].hcode == 0) continue;
    int64_t gi = tb[h].gid, a = gcount[gi], e = a + glen[gi];
    for (int64_t j = a; j < e; j++) buf[j - a] = *ptr[j];
    orc_cluster_group(buf, e - a, window, min_support, on_cluster, &d);
  }
  free(glen);
  free(buf); free(fill); free(ptr); free(gcount); free(gid_of); free(tb);
  if (n_unpl) *n_unpl = d.n_unpl;
  return d.n < cap ? d.n : cap;
}

/* ==========================================================================================
 * .bin : extract.nim:336-346 + cluster.nim:38-50.  Integer widths follow msgpack4nim
 * (jangko/msgpack4nim >= 0.4.2, third party, not vendored): always the smallest encoding.
 * ======================================================================================== */
static int mp_uint(uint8_t *b, uint64_t v) {
  if (v < 128) { b[0] = (uint8_t)v; return 1; }
  if (v < 256) { b[0] = 0xcc; b[1] = (uint8_t)v; return 2; }
  if (v < 65536) { b[0] = 0xcd; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)v; return 3; }
  b[0] = 0xce; b[1] = (uint8_t)(v >> 24); b[2] = (uint8_t)(v >> 16); b[3] = (uint8_t)(v >> 8); b[4] = (uint8_t)v; return 5;
}
static int mp_int32(uint8_t *b, int32_t v) {
  if (v >= 0) return mp_uint(b, (uint64_t)v);
  if (v >= -32) { b[0] = (uint8_t)v; return 1; }
  if (v >= -128) { b[0] = 0xd0; b[1] = (uint8_t)v; return 2; }
  if (v >= -32768) { b[0] = 0xd1; b[1] = (uint8_t)((uint16_t)v >> 8); b[2] = (uint8_t)v; return 3; }
  uint32_t u = (uint32_t)v;
  b[0] = 0xd2; b[1] = (uint8_t)(u >> 24); b[2] = (uint8_t)(u >> 16); b[3] = (uint8_t)(u >> 8); b[4] = (uint8_t)u; return 5;
}
int orc_pack_tread(uint8_t *b, const orc_tread *t, const char *qname, uint32_t qlen) {
  int o = 0;
  o += mp_int32(b + o, t->tid);
  o += mp_uint(b + o, t->position);
  b[o++] = 0x96;                                                  /* array[6, char] */
  for (int i = 0; i < 6; i++) o += mp_uint(b + o, (uint8_t)t->repeat[i]);
  o += mp_uint(b + o, t->flag);
  o += mp_uint(b + o, t->split);
  o += mp_uint(b + o, t->mapping_quality);
  o += mp_uint(b + o, t->repeat_count);
  o += mp_uint(b + o, t->align_length);
  o += mp_uint(b + o, qlen);                                      /* L */
  if (qlen < 32) b[o++] = (uint8_t)(0xa0 | qlen);
  else if (qlen < 256) { b[o++] = 0xd9; b[o++] = (uint8_t)qlen; }
  else if (qlen < 65536) { b[o++] = 0xda; b[o++] = (uint8_t)(qlen >> 8); b[o++] = (uint8_t)qlen; }
  else { b[o++] = 0xdb; b[o++] = (uint8_t)(qlen >> 24); b[o++] = (uint8_t)(qlen >> 16); b[o++] = (uint8_t)(qlen >> 8); b[o++] = (uint8_t)qlen; }
  memcpy(b + o, qname, qlen); o += (int)qlen;
  return o;
}
int64_t orc_bin_write(uint8_t *buf, int64_t cap, float proportion_repeat, uint8_t min_mapq,
                      const uint32_t frag[4096], const char *sam_header, int32_t hdr_len,
                      const orc_tread *treads, int64_t n, const uint64_t *qname_off, const char *qnames) {
  int64_t o = 0;
  uint8_t tmp[70000];
#define PUT(p, l) do { if (o + (int64_t)(l) <= cap) memcpy(buf + o, (p), (size_t)(l)); o += (int64_t)(l); } while (0)
  PUT("STR", 3);
  int16_t fmt = 0; PUT(&fmt, 2);                                   /* version.nim:4 */
  char ver[9] = {0}; memcpy(ver, "0.6.0", 5); PUT(ver, 9);         /* version.nim:1,6-8 */
  PUT(&proportion_repeat, 4);
  PUT(&min_mapq, 1);
  PUT(frag, 4096 * 4);
  PUT(&hdr_len, 4);
  PUT(sam_header, hdr_len);
  int32_t n32 = (int32_t)n; PUT(&n32, 4);
  for (int64_t i = 0; i < n; i++) {
    int64_t q = treads[i].qname_id;
    uint32_t ql = (uint32_t)(qname_off[q + 1] - qname_off[q]);
    int l = orc_pack_tread(tmp, &treads[i], qnames + qname_off[q], ql);
    PUT(tmp, l);
  }
#undef PUT
  return o;
}

/* cluster.nim:262-266 */
int orc_bounds_row(char *buf, int cap, const orc_bounds *b, const char *chrom) {
  return snprintf(buf, (size_t)cap, "%s\t%u\t%u\t%s\t%s\t%u\t%u\t%u\t%u\t%u\t%u", chrom, b->left, b->right, b->repeat, "",
                  b->left_most, b->right_most, b->center_mass, (unsigned)b->n_left, (unsigned)b->n_right, (unsigned)b->n_total);
}

/* ==========================================================================================
 * strling call: evidence around one bound (collect.nim, spanning.nim), genotype (genotyper.nim),
 * driver (call.nim:111-285).  Literal and slow: every region query is a scan over the whole
 * in-memory record set with htslib's iterator filter (tid equal, pos < end, bam_endpos > beg).
 * PARITY: the reference's own KATs for this part (tests/test_collect.nim, test_genotyper.nim,
 * test_utils.nim:10-13,29-35) are transcribed in tests/golden/reference_kats.json; Nim stdlib
 * behaviours that leak into the text (Table/CountTable slot order for string, uint16 and int16
 * keys, CountTable.sort = stable merge sort of the slot array, C printf of NaN) are unpinned.
 * ======================================================================================== */
/* generic Nim 1.6 Table/CountTable slot bookkeeping over pre-hashed distinct keys ------------- */
typedef struct { int64_t len, counter; uint64_t *hc; int64_t *id; } nim_slots;
static void ns_init(nim_slots *t, int64_t initial_size) {
  t->len = slots_needed(initial_size); t->counter = 0;
  t->hc = (uint64_t *)calloc((size_t)t->len, 8); t->id = (int64_t *)malloc((size_t)t->len * 8);
  for (int64_t i = 0; i < t->len; i++) t->id[i] = -1;
}
static void ns_free(nim_slots *t) { free(t->hc); free(t->id); }
/* insert a key known to be absent; hc = its hash (already remapped 0 -> 314159265 for Table) */
static void ns_insert(nim_slots *t, uint64_t hc, int64_t id) {
  if (must_rehash(t->len, t->counter)) {
    int64_t nl = t->len * 2;
    uint64_t *nh = (uint64_t *)calloc((size_t)nl, 8); int64_t *ni = (int64_t *)malloc((size_t)nl * 8);
    for (int64_t i = 0; i < nl; i++) ni[i] = -1;
    for (int64_t i = 0; i < t->len; i++) if (t->id[i] >= 0) {
      int64_t j = (int64_t)(t->hc[i] & (uint64_t)(nl - 1));
      while (ni[j] >= 0) j = (j + 1) & (nl - 1);
      nh[j] = t->hc[i]; ni[j] = t->id[i];
    }
    free(t->hc); free(t->id); t->hc = nh; t->id = ni; t->len = nl;
  }
  int64_t h = (int64_t)(hc & (uint64_t)(t->len - 1));
  while (t->id[h] >= 0) h = (h + 1) & (t->len - 1);
  t->hc[h] = hc; t->id[h] = id; t->counter++;
}
static uint64_t str_hash(const char *s, int n) { return orc_nim_hash_bytes((const uint8_t *)s, n); }

/* spanning.nim:7-20 */
void orc_cumulative(const uint32_t frag[4096], float cd[4096]) {
  for (int i = 0; i < 4096; i++) {
    cd[i] = 0;
    int lo = i - 11 < 0 ? 0 : i - 11, hi = i + 11 > 4095 ? 4095 : i + 11;
    for (int j = lo; j <= hi; j++) cd[i] += (float)frag[j];
  }
  for (int i = 1; i < 4096; i++) cd[i] = cd[i] + cd[i - 1];          /* math.cumsum */
  float fmax = cd[4095];
  for (int i = 0; i < 4096; i++) cd[i] = cd[i] / fmax;
}
/* spanning.nim:22-49 */
double orc_expected_spanning_probability(const float cd[4096], int64_t start, int64_t stop, int reverse, int64_t event_start,
                                         int64_t event_stop) {
  const int64_t min_spanning_bases = 20;
  int64_t dist;
  if (start < event_stop - min_spanning_bases) {
    if (reverse) return 0;
    dist = event_start - start;
    if (dist < 0) return 0;
    if (dist + (event_stop - event_start) < min_spanning_bases) return 0;
  } else {
    if (!reverse) return 0;
    dist = stop - event_stop;
    if (dist < 0) return 0;
    if (dist + (event_stop - event_start) < min_spanning_bases) return 0;
  }
  dist += min_spanning_bases;
  dist += (event_stop - event_start);
  if (dist < 0 || dist > 4095) return 0;
  return (double)(1.0f - cd[dist]);
}
/* utils.nim:129-137 */
double orc_percentile(const uint32_t frag[4096], int64_t fragment_length) {
  uint32_t total = 0;
  for (int i = 0; i < 4096; i++) total += frag[i];
  int64_t s = 0;
  for (int i = 0; i < 4096; i++) { s += frag[i]; if (i >= fragment_length) break; }
  return (double)s / (double)(total > 1 ? total : 1);
}
/* utils.nim:148-158 */
int orc_median_depth(const int64_t *D, int64_t n) {
  int64_t H[1048];
  memset(H, 0, sizeof H);
  for (int64_t i = 0; i < n; i++) H[D[i] < 1047 ? D[i] : 1047] += 1;
  int64_t s = 0;
  for (int i = 0; i < 1048; i++) { s += H[i]; if ((double)s > (double)n / 2.0) return i; }
  return 0;
}

static int cig_consumes_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
static int cig_consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
/* collect.nim:50-72 */
static int64_t find_read_position(const orc_records *r, int64_t i, int64_t position) {
  int64_t r_off = r->pos[i], q_off = 0;
  int n = cig_n(r, i);
  for (int j = 0; j < n; j++) {
    if (r_off > position) return -1;
    int op = cig_op(r, i, j), len = cig_len(r, i, j);
    if (cig_consumes_query(op)) q_off += len;
    if (cig_consumes_ref(op)) r_off += len;
    if (r_off < position) continue;
    int64_t over = r_off - position;
    if (over > q_off) return -1;
    if (!cig_consumes_query(op)) return -1;
    return q_off - over;
  }
  return -1;
}
/* collect.nim:75-93 */
static int count_in_bounds(const orc_records *r, int64_t i, const orc_bounds *b) {
  if (b->right < b->left) return 0;
  static char dna[1 << 16];
  int dlen = rec_sequence(r, i, dna);
  int64_t read_left = find_read_position(r, i, (int64_t)b->left);
  int64_t read_right = find_read_position(r, i, (int64_t)b->right);
  if (read_left >= 0 && read_right < 0) read_right = dlen;
  if (read_left < 0 && read_right < 0) return 0;
  if (read_left < 0) read_left = 0;
  int64_t slen = read_right - read_left;
  if (slen < 0) slen = 0;
  int k = (int)strlen(b->repeat);
  int result = str_count(dna + read_left, (int)slen, b->repeat, k);
  if (result < (int)((double)slen * 0.7 / (double)k)) result = 0;
  return result;
}
static void bound_slop(const orc_bounds *b, int64_t *slop) {
  int64_t bound_width = (int64_t)b->right - (int64_t)b->left;
  *slop = (int64_t)strlen(b->repeat) - 1;
  if (bound_width < 5) *slop += (5 - bound_width);
}
/* collect.nim:97-119; record overlap: cluster.nim:104-108 */
int orc_overlapping_read(const orc_records *r, int64_t i, const orc_bounds *b, orc_support *s) {
  int64_t slop; bound_slop(b, &slop);
  int64_t start = r->pos[i], stop = rec_stop(r, i);
  if (r->tid[i] != b->tid) return 0;
  int64_t ileft = start > (int64_t)b->left ? start : (int64_t)b->left, iright = stop < (int64_t)b->right ? stop : (int64_t)b->right;
  if (!(ileft <= iright)) return 0;
  s->type = ORC_OVERLAPPING_READ;
  s->repeat_count = (uint8_t)count_in_bounds(r, i, b);
  s->rec = i;
  if (start < ((int64_t)b->left - slop) && stop > ((int64_t)b->right + slop)) {
    s->type = ORC_SPANNING_READ;
    int n = cig_n(r, i);
    for (int j = 0; j < n; j++) {
      if (cig_op(r, i, j) == 1) s->cigar_ins = (uint8_t)(s->cigar_ins + (uint8_t)cig_len(r, i, j));
      if (cig_op(r, i, j) == 2) s->cigar_del = (uint8_t)(s->cigar_del + (uint8_t)cig_len(r, i, j));
    }
  }
  return 1;
}
/* collect.nim:36-48 */
int orc_spanning_fragment(const orc_records *r, const int32_t *isize, int64_t L, int64_t R, const orc_bounds *b, orc_support *s,
                          const uint32_t frag[4096]) {
  if (!(r->pos[L] <= r->pos[R])) { fprintf(stderr, "oracle: doAssert L.start <= R.start (collect.nim:37)\n"); abort(); }
  int64_t slop; bound_slop(b, &slop);
  if (r->pos[L] < ((int64_t)b->left - slop) && rec_stop(r, R) > ((int64_t)b->right + slop)) {
    s->type = ORC_SPANNING_FRAGMENT;
    int64_t a = isize[L] < 0 ? -(int64_t)isize[L] : isize[L];
    s->frag_len = (uint32_t)a > 1u ? (uint32_t)a : 1u;
    s->frag_pct = orc_percentile(frag, (int64_t)s->frag_len);
    s->rec = L;
    return 1;
  }
  return 0;
}

/* collect.nim:132-182.  out gets the Support list in the reference's order (overlapping reads in file order, then
 * spanning fragments in Table[string, seq[Record]] slot order); returns the number (also when > cap). */
int64_t orc_spanners(const orc_records *r, const int32_t *isize, const orc_bounds *b, int window, const uint32_t frag[4096],
                     uint8_t min_mapq, orc_support *out, int64_t cap, int *median_depth, float *expected_spanners) {
  const int max_size = 5000;
  int64_t window_left = (int64_t)b->left - window, window_right = (int64_t)b->right + window;
  static float cd[4096];
  orc_cumulative(frag, cd);
  int64_t nd = window_right - window_left;
  int64_t *depths = (int64_t *)calloc((size_t)nd, 8);
  int64_t beg = window_left > 0 ? window_left : 0, end = window_right;
  int64_t n_out = 0;
  /* distinct qnames seen: id -> first record; both tables hold qname strings */
  nim_slots exp_t, pair_t;
  ns_init(&exp_t, 32); ns_init(&pair_t, 32);
  int64_t cap_q = 1024, nq_exp = 0, nq_pair = 0;
  int64_t *exp_rec = (int64_t *)malloc((size_t)cap_q * 8); double *exp_val = (double *)malloc((size_t)cap_q * 8);
  int64_t *pair_rec = (int64_t *)malloc((size_t)cap_q * 8), *pair_n = (int64_t *)malloc((size_t)cap_q * 8), *pair_second = (int64_t *)malloc((size_t)cap_q * 8);
  int early = 0;
  for (int64_t i = 0; i < r->n && !early; i++) {
    if (r->tid[i] != b->tid) continue;                                   /* hts iterator: region filter */
    int64_t start = r->pos[i], stop = rec_stop(r, i);
    if (!(start < end && stop > beg)) continue;
    uint16_t f = r->flag[i];
    if ((f & 0x100) || (f & 0x800) || (f & 0x400)) continue;             /* :142 */
    if (r->mapq[i] < min_mapq) continue;
    double prob = orc_expected_spanning_probability(cd, start, stop, (f & 0x10) != 0, (int64_t)b->left, (int64_t)b->right);
    const char *qn = r->qnames + r->qname_off[i];
    int qlen = (int)(r->qname_off[i + 1] - r->qname_off[i]);
    if (prob > 0) {
      int64_t id = -1;
      for (int64_t k = 0; k < nq_exp; k++) {
        int64_t rr = exp_rec[k];
        if ((int)(r->qname_off[rr + 1] - r->qname_off[rr]) == qlen && memcmp(r->qnames + r->qname_off[rr], qn, (size_t)qlen) == 0) { id = k; break; }
      }
      if (id >= 0) exp_val[id] = 0.5 * (exp_val[id] + prob);
      else {
        if (nq_exp == cap_q) { cap_q *= 2; exp_rec = realloc(exp_rec, (size_t)cap_q * 8); exp_val = realloc(exp_val, (size_t)cap_q * 8);
                               pair_rec = realloc(pair_rec, (size_t)cap_q * 8); pair_n = realloc(pair_n, (size_t)cap_q * 8); pair_second = realloc(pair_second, (size_t)cap_q * 8); }
        uint64_t hc = str_hash(qn, qlen); if (hc == 0) hc = 314159265;
        ns_insert(&exp_t, hc, nq_exp);
        exp_rec[nq_exp] = i; exp_val[nq_exp] = prob; nq_exp++;
      }
    }
    { int64_t a = start - window_left - 1; if (a < 0) a = 0; depths[a] += 1;
      int64_t z = stop - window_left - 1; if (z > nd - 1) z = nd - 1; depths[z] -= 1; }
    orc_support s; memset(&s, 0, sizeof s);
    if (orc_overlapping_read(r, i, b, &s)) { if (n_out < cap) out[n_out] = s; n_out++; }
    if (r->tid[i] != r->mtid[i]) continue;
    { int64_t a = isize[i] < 0 ? -(int64_t)isize[i] : isize[i]; if (a > max_size) continue; }
    int64_t id = -1;
    for (int64_t k = 0; k < nq_pair; k++) {
      int64_t rr = pair_rec[k];
      if ((int)(r->qname_off[rr + 1] - r->qname_off[rr]) == qlen && memcmp(r->qnames + r->qname_off[rr], qn, (size_t)qlen) == 0) { id = k; break; }
    }
    if (id >= 0) { if (pair_n[id] == 1) pair_second[id] = i; pair_n[id]++; }
    else {
      if (nq_pair == cap_q) { cap_q *= 2; exp_rec = realloc(exp_rec, (size_t)cap_q * 8); exp_val = realloc(exp_val, (size_t)cap_q * 8);
                              pair_rec = realloc(pair_rec, (size_t)cap_q * 8); pair_n = realloc(pair_n, (size_t)cap_q * 8); pair_second = realloc(pair_second, (size_t)cap_q * 8); }
      uint64_t hc = str_hash(qn, qlen); if (hc == 0) hc = 314159265;
      ns_insert(&pair_t, hc, nq_pair);
      pair_rec[nq_pair] = i; pair_n[nq_pair] = 1; pair_second[nq_pair] = -1; nq_pair++;
    }
    if (nq_pair > 20000) early = 1;                                       /* :171-174 */
  }
  if (early) { n_out = 0; *median_depth = -1; *expected_spanners = 0; }
  else {
    float es = 0;
    for (int64_t h = 0; h < exp_t.len; h++) if (exp_t.id[h] >= 0) es += (float)exp_val[exp_t.id[h]];   /* :176-177, values in slot order */
    *expected_spanners = es;
    for (int64_t h = 0; h < pair_t.len; h++) {                            /* :179-183 */
      int64_t id = pair_t.id[h];
      if (id < 0 || pair_n[id] != 2) continue;
      orc_support s; memset(&s, 0, sizeof s);
      if (orc_spanning_fragment(r, isize, pair_rec[id], pair_second[id], b, &s, frag)) { if (n_out < cap) out[n_out] = s; n_out++; }
    }
    for (int64_t i = 1; i < nd; i++) depths[i] += depths[i - 1];
    *median_depth = orc_median_depth(depths, nd);
  }
  free(depths); free(exp_rec); free(exp_val); free(pair_rec); free(pair_n); free(pair_second); ns_free(&exp_t); ns_free(&pair_t);
  return n_out;
}

/* ---- genotyper.nim ---- */
/* CountTable over small integer keys: keys[] in inc() order -> distinct keys in SLOT order with their counts */
static int64_t count_table_slots(const int64_t *keys, int64_t n, int64_t *okeys, int64_t *ovals) {
  nim_slots t; ns_init(&t, 32);                       /* `var x: CountTable[T]` -> initImpl(defaultInitialSize = 32) on first inc */
  int64_t nd = 0;
  int64_t *dk = (int64_t *)malloc((size_t)(n ? n : 1) * 8), *dv = (int64_t *)malloc((size_t)(n ? n : 1) * 8);
  for (int64_t i = 0; i < n; i++) {
    int64_t id = -1;
    for (int64_t k = 0; k < nd; k++) if (dk[k] == keys[i]) { id = k; break; }
    if (id >= 0) { dv[id]++; continue; }
    ns_insert(&t, orc_nim_hash_int((uint64_t)keys[i]), nd);
    dk[nd] = keys[i]; dv[nd] = 1; nd++;
  }
  int64_t m = 0;
  for (int64_t h = 0; h < t.len; h++) if (t.id[h] >= 0) { okeys[m] = dk[t.id[h]]; ovals[m] = dv[t.id[h]]; m++; }
  free(dk); free(dv); ns_free(&t);
  return m;
}
/* utils.nim:165-177 most_frequent(2) = CountTable.sort (stable, descending) then the first two keys; `largest` = first
 * maximum in slot order.  a1/a2 receive NaN when absent. */
static void top_two(const int64_t *keys, int64_t n, double *a1, double *a2) {
  *a1 = NAN; *a2 = NAN;
  if (n == 0) return;
  int64_t *k = (int64_t *)malloc((size_t)n * 8), *v = (int64_t *)malloc((size_t)n * 8);
  int64_t m = count_table_slots(keys, n, k, v);
  if (m >= 2) {
    for (int64_t i = 1; i < m; i++) {                 /* stable insertion sort, descending by count */
      int64_t kk = k[i], vv = v[i], j = i - 1;
      while (j >= 0 && v[j] < vv) { k[j + 1] = k[j]; v[j + 1] = v[j]; j--; }
      k[j + 1] = kk; v[j + 1] = vv;
    }
    *a1 = (double)k[0]; *a2 = (double)k[1];
  } else if (m == 1) *a1 = (double)k[0];
  free(k); free(v);
}
/* genotyper.nim:61-98 */
void orc_spanning_read_est(const orc_support *reads, int64_t n, double *allele1_bp, double *allele2_bp, double *allele1_ru,
                           double *allele2_ru, uint32_t *supporting) {
  int64_t *rc = (int64_t *)malloc((size_t)(n ? n : 1) * 8), *ind = (int64_t *)malloc((size_t)(n ? n : 1) * 8), m = 0;
  for (int64_t i = 0; i < n; i++) if (reads[i].type == ORC_SPANNING_READ) {
    rc[m] = (uint16_t)reads[i].repeat_count;
    ind[m] = (int16_t)((int16_t)reads[i].cigar_ins - (int16_t)reads[i].cigar_del);
    m++;
  }
  *supporting = (uint32_t)m;
  top_two(rc, m, allele1_ru, allele2_ru);
  top_two(ind, m, allele1_bp, allele2_bp);
  free(rc); free(ind);
}
/* genotyper.nim:122-130 */
static double anchored_lm(uint64_t sum_str_counts, double depth) {
  if (sum_str_counts == 0) return NAN;
  const double intercept = 4.3558142, cofficient = 0.7565329;
  double y = log2((double)sum_str_counts / (depth > 1.0 ? depth : 1.0) + 1) * cofficient + intercept;
  return pow(2, y);
}
/* genotyper.nim:150-199; tq_off/tqnames resolve tread.qname_id to its qname string */
void orc_genotype(const orc_bounds *b, const orc_tread *tandems, int64_t nt, const uint64_t *tq_off, const char *tqnames,
                  const orc_support *spanners, int64_t ns, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                  int median_fragment_length, double depth, orc_gt *c) {
  memset(c, 0, sizeof *c);
  c->tid = b->tid; c->start = b->left; c->stop = b->right; c->left_clips = b->n_left; c->right_clips = b->n_right;
  memcpy(c->repeat, b->repeat, 7);
  c->depth = depth;
  int RUlen = (int)strlen(c->repeat);
  if (ns == 0) c->allele1 = NAN;
  else {
    double a1bp, a2bp, a1ru, a2ru; uint32_t sup;
    orc_spanning_read_est(spanners, ns, &a1bp, &a2bp, &a1ru, &a2ru, &sup);
    if (!isnan(a1bp)) c->allele1 = a1bp / (double)(RUlen > 1 ? RUlen : 1);
    c->spanning_reads = sup;
    uint32_t sp = 0;
    for (int64_t i = 0; i < ns; i++) sp += (spanners[i].type == ORC_SPANNING_FRAGMENT);
    c->spanning_pairs = sp;
  }
  c->is_large = (uint16_t)b->n_left >= min_clip && (uint16_t)b->n_right >= min_clip && (uint16_t)(b->n_left + b->n_right) >= min_clip_total &&
                nt >= min_support && c->allele2 > (double)median_fragment_length;        /* allele2 is still 0.0 here (:175) */
  uint64_t sum = 0;
  for (int64_t i = 0; i < nt; i++) sum += tandems[i].repeat_count;
  c->overlapping_reads = (uint32_t)nt;
  c->sum_str_counts = (uint32_t)sum;
  c->allele2 = anchored_lm(sum, depth) / (double)(RUlen > 1 ? RUlen : 1);
  uint32_t distinct = 0;                                                               /* :187-191 */
  for (int64_t i = 0; i < nt; i++) {
    if (tandems[i].split != ORC_SOFT_NONE) continue;
    int dup = 0;
    const char *qi = tqnames + tq_off[tandems[i].qname_id]; uint64_t li = tq_off[tandems[i].qname_id + 1] - tq_off[tandems[i].qname_id];
    for (int64_t j = 0; j < i && !dup; j++) {
      if (tandems[j].split != ORC_SOFT_NONE) continue;
      const char *qj = tqnames + tq_off[tandems[j].qname_id]; uint64_t lj = tq_off[tandems[j].qname_id + 1] - tq_off[tandems[j].qname_id];
      dup = (li == lj && memcmp(qi, qj, (size_t)li) == 0);
    }
    distinct += !dup;
  }
  c->anchored_reads = distinct;
}
/* Nim `$float` for the values that occur here */
static int nim_float_str(char *buf, int cap, double v) {
  if (isnan(v)) return snprintf(buf, (size_t)cap, "nan");
  int n = snprintf(buf, (size_t)cap, "%.16g", v);
  if (!strpbrk(buf, ".eEn")) n += snprintf(buf + n, (size_t)(cap - n), ".0");
  return n;
}
/* genotyper.nim:56-57 */
int orc_call_row(char *buf, int cap, const orc_gt *c, const char *chrom) {
  char d[64];
  nim_float_str(d, sizeof d, c->depth);
  return snprintf(buf, (size_t)cap, "%s\t%u\t%u\t%s\t%.2f\t%.2f\t%u\t%u\t%u\t%.2f\t%.2f\t%u\t%u\t%d\t%s\t%u", chrom, c->start, c->stop, c->repeat,
                  c->allele1, c->allele2, c->anchored_reads, c->spanning_reads, c->spanning_pairs, (double)c->expected_spanning_fragments,
                  (double)c->pctile, c->left_clips, c->right_clips, c->unplaced_reads, d, c->sum_str_counts);
}

/* ---- call.nim:111-285 without -l/-b ---- */
typedef struct { char *p; int64_t cap, n; } sbuf;
static void sb_add(sbuf *s, const char *line) {
  int64_t l = (int64_t)strlen(line);
  if (s->n + l + 1 < s->cap) { memcpy(s->p + s->n, line, (size_t)l); s->p[s->n + l] = '\n'; }
  s->n += l + 1;
}
typedef struct {
  const orc_records *r; const int32_t *isize; const uint32_t *frag; const uint64_t *tq_off; const char *tqnames; const char *const *targets;
  int window, min_support, frag_median; uint16_t min_clip, min_clip_total; uint8_t min_mapq;
  orc_gt *calls; int64_t ncalls, ccap; char (*canon)[7];
  sbuf *bounds;
} call_ctx;
static int call_on_bound(void *ud, orc_bounds *b, const orc_tread *reads, int64_t n) {
  call_ctx *c = (call_ctx *)ud;
  int64_t cap = 1 << 16;
  orc_support *sp = (orc_support *)malloc((size_t)cap * sizeof(orc_support));
  int md; float es;
  int64_t ns = orc_spanners(c->r, c->isize, b, c->window, c->frag, c->min_mapq, sp, cap, &md, &es);
  if (ns > 5000 || md == -1) { free(sp); return 0; }                                   /* call.nim:239-244 */
  if (c->ncalls == c->ccap) { c->ccap *= 2; c->calls = realloc(c->calls, (size_t)c->ccap * sizeof(orc_gt)); c->canon = realloc(c->canon, (size_t)c->ccap * 7); }
  orc_gt *gt = &c->calls[c->ncalls];
  orc_genotype(b, reads, n, c->tq_off, c->tqnames, sp, ns, c->min_support, c->min_clip, c->min_clip_total, c->frag_median, (double)md, gt);
  gt->expected_spanning_fragments = es;
  char in6[6] = {0}, out6[6];
  memcpy(in6, b->repeat, strlen(b->repeat));
  orc_canonical_repeat(in6, out6);
  memset(c->canon[c->ncalls], 0, 7); memcpy(c->canon[c->ncalls], out6, 6);
  c->ncalls++;
  char row[512], line[600];
  orc_bounds_row(row, sizeof row, b, c->targets[b->tid]);
  snprintf(line, sizeof line, "%s\t%d", row, md);
  sb_add(c->bounds, line);
  free(sp);
  return 1;
}
/* cluster.nim:262-266 with the name column filled */
static int locus_row(char *buf, int cap, const orc_locus *L, const char *chrom) {
  const orc_bounds *b = &L->b;
  return snprintf(buf, (size_t)cap, "%s\t%u\t%u\t%s\t%s\t%u\t%u\t%u\t%u\t%u\t%u", chrom, b->left, b->right, b->repeat, L->name, b->left_most,
                  b->right_most, b->center_mass, (unsigned)b->n_left, (unsigned)b->n_right, (unsigned)b->n_total);
}
/* call.nim:190-218: a bound given with -l/-b, after assign_reads_locus took its reads */
static void call_on_locus(void *ud, orc_locus *L, const orc_tread *reads, int64_t n) {
  call_ctx *c = (call_ctx *)ud;
  orc_bounds *b = &L->b;
  if (b->right - b->left > 1000u) return;                                             /* :193-195 */
  int64_t cap = 1 << 16;
  orc_support *sp = (orc_support *)malloc((size_t)cap * sizeof(orc_support));
  int md; float es;
  int64_t ns = orc_spanners(c->r, c->isize, b, c->window, c->frag, c->min_mapq, sp, cap, &md, &es);
  if (ns > 5000 || md == -1) { free(sp); return; }
  if (c->ncalls == c->ccap) { c->ccap *= 2; c->calls = realloc(c->calls, (size_t)c->ccap * sizeof(orc_gt)); c->canon = realloc(c->canon, (size_t)c->ccap * 7); }
  orc_gt *gt = &c->calls[c->ncalls];
  orc_genotype(b, reads, n, c->tq_off, c->tqnames, sp, ns, c->min_support, c->min_clip, c->min_clip_total, c->frag_median, (double)md, gt);
  gt->expected_spanning_fragments = es;
  char in6[6] = {0}, out6[6];
  memcpy(in6, b->repeat, strlen(b->repeat));
  orc_canonical_repeat(in6, out6);
  memset(c->canon[c->ncalls], 0, 7); memcpy(c->canon[c->ncalls], out6, 6);
  c->ncalls++;
  char row[768], line[800];
  locus_row(row, sizeof row, L, c->targets[b->tid]);
  snprintf(line, sizeof line, "%s\t%d", row, md);
  sb_add(c->bounds, line);
  free(sp);
}

static int get_tid(const char *name, int nlen, const char *const *names, int n_targets) {   /* utils.nim:214-218 */
  for (int t = 0; t < n_targets; t++) if ((int)strlen(names[t]) == nlen && memcmp(names[t], name, (size_t)nlen) == 0) return t;
  return -1;
}
/* cluster.nim:111-141 parse_bed: whitespace separated chrom start stop unit [name]; returns count or -1 where the reference quits */
int64_t orc_parse_bed(const char *text, const char *const *names, const uint32_t *lengths, int n_targets, uint32_t window, orc_locus *out, int64_t cap) {
  int64_t n = 0;
  const char *p = text;
  while (*p) {
    const char *e = strchr(p, '\n'); if (!e) e = p + strlen(p);
    const char *f[8]; int fl[8], nf = 0;
    const char *q = p;
    while (q < e) {
      while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
      if (q >= e) break;
      const char *s0 = q;
      while (q < e && !(*q == ' ' || *q == '\t' || *q == '\r')) q++;
      if (nf < 8) { f[nf] = s0; fl[nf] = (int)(q - s0); }
      nf++;
    }
    if (nf != 4 && nf != 5) return -1;
    orc_locus L; memset(&L, 0, sizeof L);
    if (nf == 5) memcpy(L.name, f[4], (size_t)(fl[4] < 127 ? fl[4] : 127));
    L.b.tid = get_tid(f[0], fl[0], names, n_targets);
    L.b.left = (uint32_t)strtoll(f[1], NULL, 10); L.b.right = (uint32_t)strtoll(f[2], NULL, 10);
    if (fl[3] > 6 || L.b.tid < 0) return -1;
    memcpy(L.b.repeat, f[3], (size_t)fl[3]);
    int32_t lm = (int32_t)L.b.left - (int32_t)window;
    L.b.left_most = (uint32_t)(lm > 0 ? lm : 0);
    L.b.right_most = L.b.right + window < lengths[L.b.tid] ? L.b.right + window : lengths[L.b.tid];
    for (int i = 0; i < fl[3]; i++) if (!strchr("ATCG", f[3][i])) return -1;
    if (!(L.b.left <= L.b.right) || !(L.b.left_most <= L.b.right_most)) return -1;
    if (n < cap) out[n] = L;
    n++;
    p = *e ? e + 1 : e;
  }
  return n;
}
/* cluster.nim:143-169 parse_bounds: exactly 11 tab separated fields, '#' lines skipped */
int64_t orc_parse_bounds(const char *text, const char *const *names, int n_targets, orc_locus *out, int64_t cap) {
  int64_t n = 0;
  const char *p = text;
  while (*p) {
    const char *e = strchr(p, '\n'); if (!e) e = p + strlen(p);
    if (*p != '#') {
      const char *f[12]; int fl[12], nf = 0;
      const char *q = p;
      for (;;) {
        const char *t = q;
        while (t < e && *t != '\t') t++;
        if (nf < 12) { f[nf] = q; fl[nf] = (int)(t - q); }
        nf++;
        if (t >= e) break;
        q = t + 1;
      }
      if (nf != 11) return -1;
      orc_locus L; memset(&L, 0, sizeof L);
      L.b.tid = get_tid(f[0], fl[0], names, n_targets);
      if (L.b.tid < 0 || fl[3] > 6) return -1;
      L.b.left = (uint32_t)strtoll(f[1], NULL, 10); L.b.right = (uint32_t)strtoll(f[2], NULL, 10);
      memcpy(L.b.repeat, f[3], (size_t)fl[3]);
      memcpy(L.name, f[4], (size_t)(fl[4] < 127 ? fl[4] : 127));
      L.b.left_most = (uint32_t)strtoll(f[5], NULL, 10); L.b.right_most = (uint32_t)strtoll(f[6], NULL, 10);
      L.b.center_mass = (uint32_t)strtoll(f[7], NULL, 10);
      L.b.n_left = (uint16_t)strtoll(f[8], NULL, 10); L.b.n_right = (uint16_t)strtoll(f[9], NULL, 10); L.b.n_total = (uint16_t)strtoll(f[10], NULL, 10);
      for (int i = 0; i < fl[3]; i++) if (!strchr("ATCG", f[3][i])) return -1;
      if (!(L.b.left <= L.b.right) || !(L.b.left_most <= L.b.right_most)) return -1;
      if (n < cap) out[n] = L;
      n++;
    }
    p = *e ? e + 1 : e;
  }
  return n;
}
static int loci_overlap(const orc_bounds *a, const orc_bounds *b) {                    /* cluster.nim:96-100 */
  if (a->tid == b->tid && strcmp(a->repeat, b->repeat) == 0) {
    uint32_t il = a->left > b->left ? a->left : b->left, ir = a->right < b->right ? a->right : b->right;
    return il <= ir;
  }
  return 0;
}
/* call.nim:160-183: bounds first (loci overwrite the bound they overlap and are then removed with seq.del, which moves the
 * last element into the hole), then the loci that are left.  Returns the merged list length. */
int64_t orc_merge_loci_bounds(orc_locus *bounds, int64_t nb, orc_locus *loci, int64_t nl, orc_locus *out) {
  for (int64_t i = 0; i < nb; i++) {
    for (int64_t j = 0; j < nl; j++) {
      if (loci_overlap(&loci[j].b, &bounds[i].b)) {
        memcpy(bounds[i].name, loci[j].name, sizeof bounds[i].name);
        bounds[i].b.left = loci[j].b.left; bounds[i].b.right = loci[j].b.right;
        loci[j] = loci[nl - 1]; nl--;                                               /* loci.del(i) */
        break;
      }
    }
    out[i] = bounds[i];
  }
  for (int64_t j = 0; j < nl; j++) out[nb + j] = loci[j];
  return nb + nl;
}

static int cmp_f32(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : (x > y ? 1 : 0); }

int orc_call(const orc_tread *treads, int64_t n, const uint64_t *tq_off, const char *tqnames, const orc_records *r, const int32_t *isize,
             const uint32_t frag[4096], const char *const *target_names, int min_support, uint16_t min_clip, uint16_t min_clip_total,
             uint8_t min_mapq, char *bounds_buf, int64_t bcap, char *gt_buf, int64_t gcap, char *unpl_buf, int64_t ucap,
             int64_t *bn, int64_t *gn, int64_t *un, const char *loci_text, const char *bounds_text, const uint32_t *target_lengths, int n_targets) {
  sbuf sb = {bounds_buf, bcap, 0}, sg = {gt_buf, gcap, 0}, su = {unpl_buf, ucap, 0};
  sb_add(&sb, "#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total\tdepth");
  sb_add(&sg, "#chrom\tleft\tright\trepeatunit\tallele1_est\tallele2_est\tanchored_reads\tspanning_reads\tspanning_pairs\texpected_spanning_pairs\tspanning_pairs_pctl\tleft_clips\tright_clips\tunplaced_pairs\tdepth\tsum_str_counts");
  call_ctx c; memset(&c, 0, sizeof c);
  c.r = r; c.isize = isize; c.frag = frag; c.tq_off = tq_off; c.tqnames = tqnames; c.targets = target_names;
  c.frag_median = orc_median(frag, 0.5); c.window = orc_median(frag, 0.99);
  c.min_support = min_support; c.min_clip = min_clip; c.min_clip_total = min_clip_total; c.min_mapq = min_mapq;
  c.ccap = 256; c.calls = malloc((size_t)c.ccap * sizeof(orc_gt)); c.canon = malloc((size_t)c.ccap * 7); c.bounds = &sb;
  uint16_t max_clip_dist = (uint16_t)(0.5 * (double)orc_median(frag, 0.5));
  int64_t ucap_n = n + 1, nu = 0;
  orc_unplaced *unpl = (orc_unplaced *)calloc((size_t)ucap_n, sizeof(orc_unplaced));
  orc_bounds *tmp = (orc_bounds *)malloc((size_t)(n + 1) * sizeof(orc_bounds));
  int64_t lcap = 1 << 16, nl = 0, nbd = 0;
  orc_locus *loci = (orc_locus *)malloc((size_t)lcap * sizeof(orc_locus)), *bnds = (orc_locus *)malloc((size_t)lcap * sizeof(orc_locus));
  orc_locus *merged = (orc_locus *)malloc((size_t)2 * lcap * sizeof(orc_locus));
  if (loci_text) nl = orc_parse_bed(loci_text, target_names, target_lengths, n_targets, (uint32_t)c.window, loci, lcap);
  if (bounds_text) nbd = orc_parse_bounds(bounds_text, target_names, n_targets, bnds, lcap);
  if (nl < 0 || nbd < 0) return -1;
  g_n_loci = orc_merge_loci_bounds(bnds, nbd, loci, nl, merged);
  g_loci = merged; g_on_locus = call_on_locus; g_locus_ud = &c;
  call_bounds_impl(treads, n, 1, (uint32_t)c.window, min_support, min_clip, min_clip_total, max_clip_dist, tmp, n + 1, unpl, ucap_n, &nu, call_on_bound, &c);
  g_loci = NULL; g_n_loci = 0; g_on_locus = NULL; g_locus_ud = NULL;
  free(tmp); free(loci); free(bnds); free(merged);
  /* add_percentile, call.nim:38-48 */
  float *oes = (float *)malloc((size_t)(c.ncalls ? c.ncalls : 1) * sizeof(float));
  for (int64_t i = 0; i < c.ncalls; i++) {
    float obs = (float)c.calls[i].spanning_pairs, ex = c.calls[i].expected_spanning_fragments;
    oes[i] = (1.0f + obs - ex) / (ex + 1.0f);
  }
  float *sorted = (float *)malloc((size_t)(c.ncalls ? c.ncalls : 1) * sizeof(float));
  memcpy(sorted, oes, (size_t)c.ncalls * sizeof(float));
  qsort(sorted, (size_t)c.ncalls, sizeof(float), cmp_f32);
  for (int64_t i = 0; i < c.ncalls; i++) {
    int64_t lb = 0;
    while (lb < c.ncalls && sorted[lb] < oes[i]) lb++;
    volatile float num = (float)lb, den = (float)(c.ncalls - 1);
    c.calls[i].pctile = num / den;
  }
  /* unplaced_counts: CountTable[string], `[]=` in group order; file in slot order (call.nim:280-281) */
  nim_slots ut; ns_init(&ut, 32);
  for (int64_t i = 0; i < nu; i++) ns_insert(&ut, str_hash(unpl[i].repeat, (int)strlen(unpl[i].repeat)), i);
  /* genotypes_by_repeat: Table[string, seq[Call]] keyed by the canonical unit; rows in slot order, insertion order inside */
  nim_slots gt; ns_init(&gt, 32);
  int64_t *first_of = (int64_t *)malloc((size_t)(c.ncalls ? c.ncalls : 1) * 8), ng = 0;
  for (int64_t i = 0; i < c.ncalls; i++) {
    int seen = 0;
    for (int64_t k = 0; k < ng && !seen; k++) seen = strcmp(c.canon[first_of[k]], c.canon[i]) == 0;
    if (seen) continue;
    uint64_t hc = str_hash(c.canon[i], (int)strlen(c.canon[i])); if (hc == 0) hc = 314159265;
    ns_insert(&gt, hc, i);
    first_of[ng++] = i;
  }
  for (int64_t h = 0; h < gt.len; h++) {
    if (gt.id[h] < 0) continue;
    const char *key = c.canon[gt.id[h]];
    /* is_large is always false (see orc_genotype), so update_genotype never runs: call.nim:267-276 */
    for (int64_t i = 0; i < c.ncalls; i++) {
      if (strcmp(c.canon[i], key) != 0) continue;
      char row[1024];
      orc_call_row(row, sizeof row, &c.calls[i], target_names[c.calls[i].tid]);
      sb_add(&sg, row);
    }
  }
  for (int64_t h = 0; h < ut.len; h++) {
    if (ut.id[h] < 0) continue;
    char row[64];
    snprintf(row, sizeof row, "%s\t%lld", unpl[ut.id[h]].repeat, (long long)unpl[ut.id[h]].count);
    sb_add(&su, row);
  }
  *bn = sb.n; *gn = sg.n; *un = su.n;
  ns_free(&ut); ns_free(&gt); free(first_of); free(oes); free(sorted); free(unpl); free(c.calls); free(c.canon);
  return 0;
}

/* the reads (indices into treads) of every bound orc_call_bounds(mode 1) returns, cluster order */
typedef struct { int64_t *off, *mem, cap, nb, nm; } mem_ctx;
static int members_on_bound(void *ud, orc_bounds *b, const orc_tread *reads, int64_t n) {
  mem_ctx *m = (mem_ctx *)ud;
  m->off[m->nb++] = m->nm;
  for (int64_t i = 0; i < n; i++) { if (m->nm < m->cap) m->mem[m->nm] = reads[i].src; m->nm++; }
  return 1;
}
int64_t orc_call_members(const orc_tread *treads, int64_t n, uint32_t window, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                         uint16_t max_clip_dist, int64_t *member_off, int64_t *members, int64_t cap, int64_t *n_members) {
  orc_tread *cp = (orc_tread *)malloc((size_t)(n ? n : 1) * sizeof(orc_tread));
  for (int64_t i = 0; i < n; i++) { cp[i] = treads[i]; cp[i].src = i; }
  orc_bounds *tmp = (orc_bounds *)malloc((size_t)(n + 1) * sizeof(orc_bounds));
  mem_ctx m = {member_off, members, cap, 0, 0};
  int64_t nb = call_bounds_impl(cp, n, 1, window, min_support, min_clip, min_clip_total, max_clip_dist, tmp, n + 1, NULL, 0, NULL, members_on_bound, &m);
  member_off[nb] = m.nm;
  *n_members = m.nm;
  free(tmp); free(cp);
  return nb;
}

/* merge.nim:154-187 as text: header, one row per -l locus (ungated, after assign_reads_locus), then the clustered bounds */
static void merge_on_locus(void *ud, orc_locus *L, const orc_tread *reads, int64_t n) {
  call_ctx *c = (call_ctx *)ud;
  char row[768];
  locus_row(row, sizeof row, L, c->targets[L->b.tid]);
  sb_add(c->bounds, row);
}
int orc_merge_text(const orc_tread *treads, int64_t n, uint32_t window, int min_support, uint16_t min_clip, uint16_t min_clip_total,
                   uint16_t max_clip_dist, const char *loci_text, const char *const *target_names, const uint32_t *target_lengths, int n_targets,
                   char *buf, int64_t cap, int64_t *need) {
  sbuf sb = {buf, cap, 0};
  sb_add(&sb, "#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total");
  call_ctx c; memset(&c, 0, sizeof c);
  c.targets = target_names; c.bounds = &sb;
  int64_t lcap = 1 << 16, nl = 0;
  orc_locus *loci = (orc_locus *)malloc((size_t)lcap * sizeof(orc_locus));
  if (loci_text) nl = orc_parse_bed(loci_text, target_names, target_lengths, n_targets, window, loci, lcap);
  if (nl < 0) return -1;
  g_loci = loci; g_n_loci = nl; g_on_locus = merge_on_locus; g_locus_ud = &c;
  orc_bounds *tmp = (orc_bounds *)malloc((size_t)(n + 1) * sizeof(orc_bounds));
  int64_t nb = call_bounds_impl(treads, n, 0, window, min_support, min_clip, min_clip_total, max_clip_dist, tmp, n + 1, NULL, 0, NULL, NULL, NULL);
  g_loci = NULL; g_n_loci = 0; g_on_locus = NULL; g_locus_ud = NULL;
  for (int64_t i = 0; i < nb; i++) {
    char row[768];
    orc_bounds_row(row, sizeof row, &tmp[i], target_names[tmp[i].tid]);
    sb_add(&sb, row);
  }
  free(tmp); free(loci);
  *need = sb.n;
  return 0;
}
