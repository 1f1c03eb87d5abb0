/* bamgen.c -- test / benchmark infrastructure, NOT part of the product: turns record arrays (strling_amd.records.RecordBatch)
 * into BAM record bytes (SAM spec 4.2) and BGZF blocks (SAM spec 4.1, zlib deflate at a chosen level), so that a
 * whole-genome-sized synthetic BAM can be written in minutes instead of hours (strling_amd/bamio.py::write_bam_slabs
 * drives it from a pool of processes).  The bytes are the ones bamio.write_bam produces record for record, plus, on
 * request, per-base qualities drawn from four Illumina-style bins and a few aux tags (what a real aligner output carries
 * and what makes the DEFLATE streams literal-heavy like htslib's level-6 output).
 * gcc -O2 -shared -fPIC -o libbamgen.so bamgen.c -lz */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static int reg2bin(int64_t beg, int64_t end) { /* SAM spec 5.3 */
  --end;
  if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
  if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
  if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
  if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
  if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
  return 0;
}

static inline uint64_t splitmix(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* Serialise records [0, n) into out (capacity cap).  rec_off[n + 1] receives the byte offset of every record, ref_end[n]
 * the end of its alignment on the reference (for the index), bin_out[n] its BAI bin.
 * quals: 0 = 0xff (absent, as bamio.write_bam writes them), 1 = per-base draws from the bins {2, 12, 23, 37} with
 * probabilities {.03, .07, .15, .75} (the model tools/inflate_bench.py calls "realistic"), seeded by `seed`.
 * aux: 0 = none, 1 = NM:C MD:Z AS:C XS:C RG:Z (27 - 29 bytes per record).
 * Returns the bytes written, or -1 if cap is too small. */
int64_t bamgen_serialize(int64_t n, const int32_t *tid, const int32_t *pos, const int32_t *mtid, const int32_t *mpos, const uint16_t *flag,
                         const uint8_t *mapq, const int32_t *isize, const uint32_t *cigar_off, const uint32_t *cigar, const uint64_t *seq_off,
                         const int32_t *l_seq, const uint8_t *seq4, const uint64_t *qname_off, const uint8_t *qnames, int quals, int aux,
                         uint64_t seed, uint8_t *out, int64_t cap, uint64_t *rec_off, int32_t *ref_end, uint32_t *bin_out) {
  static const uint8_t bins[4] = {2, 12, 23, 37};
  int64_t o = 0;
  uint64_t st = seed * 0x2545F4914F6CDD1Dull + 12345;
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t c0 = cigar_off[i], c1 = cigar_off[i + 1];
    const uint32_t nc = c1 - c0;
    const int32_t L = l_seq[i];
    const uint64_t q0 = qname_off[i], q1 = qname_off[i + 1];
    const uint32_t ql = (uint32_t)(q1 - q0) + 1;
    int64_t rl = 0;
    for (uint32_t k = c0; k < c1; ++k) {
      const uint32_t op = cigar[k] & 15;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cigar[k] >> 4;
    }
    if ((flag[i] & 4) || rl == 0) rl = 1;
    const int64_t need = 36 + ql + 4ll * nc + (L + 1) / 2 + L + (aux ? 40 : 0);
    if (o + need > cap) return -1;
    rec_off[i] = (uint64_t)o;
    if (ref_end) ref_end[i] = (int32_t)(pos[i] + rl);
    const int bin = tid[i] < 0 ? 4680 : reg2bin(pos[i] < 0 ? 0 : pos[i], (pos[i] < 0 ? 0 : pos[i]) + rl);
    if (bin_out) bin_out[i] = (uint32_t)bin;
    uint8_t *p = out + o + 4;
    memcpy(p, &tid[i], 4); memcpy(p + 4, &pos[i], 4);
    p[8] = (uint8_t)ql; p[9] = mapq[i];
    const uint16_t b16 = 4680, nc16 = (uint16_t)nc;      /* bamio.write_bam writes the constant 4680; kept: byte-identical records */
    memcpy(p + 10, &b16, 2); memcpy(p + 12, &nc16, 2); memcpy(p + 14, &flag[i], 2);
    memcpy(p + 16, &L, 4); memcpy(p + 20, &mtid[i], 4); memcpy(p + 24, &mpos[i], 4);
    const int32_t tl = isize ? isize[i] : 0;
    memcpy(p + 28, &tl, 4);
    p += 32;
    memcpy(p, qnames + q0, ql - 1); p[ql - 1] = 0; p += ql;
    memcpy(p, cigar + c0, 4ull * nc); p += 4ull * nc;
    const int sb = (L + 1) / 2;
    memcpy(p, seq4 + seq_off[i], (size_t)sb);
    if ((L & 1) && sb) p[sb - 1] &= 0xF0;
    p += sb;
    if (!quals) memset(p, 0xff, (size_t)L);
    else {
      for (int32_t j = 0; j < L; j += 8) {       /* 8 bases per 64 random bits */
        uint64_t r = splitmix(&st);
        for (int32_t k = j; k < L && k < j + 8; ++k, r >>= 8) {
          const unsigned u = (unsigned)(r & 255);
          p[k] = bins[u < 8 ? 0 : u < 26 ? 1 : u < 64 ? 2 : 3];      /* 3.1 % / 7.0 % / 14.8 % / 75 % */
        }
      }
    }
    p += L;
    if (aux) {
      const uint64_t r = splitmix(&st);
      const unsigned nm = (unsigned)(r & 255) < 200 ? 0 : (unsigned)((r >> 8) & 3) + 1;
      p[0] = 'N'; p[1] = 'M'; p[2] = 'C'; p[3] = (uint8_t)nm; p += 4;
      p[0] = 'M'; p[1] = 'D'; p[2] = 'Z'; p += 3;
      {
        char md[16];
        int m = 0, v = L - (int)nm;
        char tmp[12];
        int t = 0;
        if (v <= 0) tmp[t++] = '0';
        for (; v > 0; v /= 10) tmp[t++] = (char)('0' + v % 10);
        while (t) md[m++] = tmp[--t];
        md[m++] = 0;
        memcpy(p, md, (size_t)m); p += m;
      }
      p[0] = 'A'; p[1] = 'S'; p[2] = 'C'; p[3] = (uint8_t)(L > 255 ? 255 : L - 5 * (int)nm < 0 ? 0 : L - 5 * (int)nm); p += 4;
      p[0] = 'X'; p[1] = 'S'; p[2] = 'C'; p[3] = (uint8_t)((r >> 16) & 63); p += 4;
      memcpy(p, "RGZgrp1", 8); p += 8;        /* "RG" 'Z' "grp1" NUL */
    }
    const int32_t bs = (int32_t)(p - (out + o + 4));
    memcpy(out + o, &bs, 4);
    o += 4 + bs;
  }
  rec_off[n] = (uint64_t)o;
  return o;
}

/* BGZF: cut raw[0, n) every `block` bytes, deflate every piece (raw deflate, zlib `level`) into a block with the BC extra
 * field, CRC-32 and ISIZE.  csize[k] receives the size of block k.  Returns the bytes written, -1 if cap is too small,
 * -2 on a zlib error. */
int64_t bamgen_bgzf(const uint8_t *raw, int64_t n, int level, int block, uint8_t *out, int64_t cap, uint32_t *csize, int64_t csize_cap) {
  int64_t o = 0, kb = 0;
  z_stream z;
  memset(&z, 0, sizeof z);
  if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -2;
  for (int64_t a = 0; a < n; a += block, ++kb) {
    const int64_t len = n - a < block ? n - a : block;
    if (o + 18 + len + len / 1000 + 64 + 8 > cap || kb >= csize_cap) { deflateEnd(&z); return -1; }
    uint8_t *h = out + o;
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(h, hdr, 16);
    deflateReset(&z);
    z.next_in = (Bytef *)(raw + a); z.avail_in = (uInt)len;
    z.next_out = h + 18; z.avail_out = (uInt)(cap - o - 18 - 8 > 0xffff - 26 ? 0xffff - 26 : cap - o - 18 - 8);
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) { deflateEnd(&z); return -2; }    /* (a block of <= 0xff00 bytes always fits 64 KiB) */
    const uint32_t clen = (uint32_t)z.total_out;
    const uint16_t bsize = (uint16_t)(clen + 25);
    memcpy(h + 16, &bsize, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), raw + a, (uInt)len), isz = (uint32_t)len;
    memcpy(h + 18 + clen, &crc, 4); memcpy(h + 18 + clen + 4, &isz, 4);
    csize[kb] = clen + 26;
    o += clen + 26;
  }
  deflateEnd(&z);
  return o;
}
