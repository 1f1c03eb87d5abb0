// bam_reader.cpp -- see bam_reader.h.  BGZF = concatenated gzip members with a BC extra field carrying the
// compressed block size (SAM spec 4.1); each is inflated with raw zlib into a <= 64 KiB buffer.
#include "bam_reader.h"
#include <string.h>
#include <zlib.h>
#include <algorithm>

namespace strl {

void RecordBatch::clear() {
  tid.clear(); pos.clear(); mtid.clear(); mpos.clear(); isize.clear(); l_seq.clear(); flag.clear(); mapq.clear();
  cigar_off.assign(1, 0); cigar.clear(); seq_off.clear(); qname_off.assign(1, 0); seq4.clear(); qnames.clear();
}

strl_records RecordBatch::view() {
  seq4.resize(seq4.size() + 32, 0);   // slack the device may over-read (call view() once per filled batch)
  strl_records r{};
  r.n = (int64_t)tid.size();
  r.tid = tid.data(); r.pos = pos.data(); r.mtid = mtid.data(); r.mpos = mpos.data(); r.flag = flag.data(); r.mapq = mapq.data();
  r.cigar_off = cigar_off.data(); r.cigar = cigar.data(); r.seq_off = seq_off.data(); r.l_seq = l_seq.data();
  r.seq4 = seq4.data(); r.qname_off = qname_off.data(); r.qnames = qnames.data();
  return r;
}

BamReader::~BamReader() { close(); }
void BamReader::close() {
  if (f_) fclose(f_);
  f_ = nullptr;
}

bool BamReader::fill(std::string &err) {
  ubuf_.clear();
  upos_ = 0;
  for (;;) {  // skip empty blocks (e.g. the EOF marker in the middle of concatenated files)
    uint8_t h[18];
    block_start_ = next_block_;
    const size_t got = fread(h, 1, 18, f_);
    if (got == 0) { eof_ = true; return true; }
    if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
    const uint32_t xlen = h[10] | (h[11] << 8);
    // the BC subfield is first in every BGZF writer; tolerate others by scanning
    uint32_t bsize = 0;
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f_) != xlen - 6) { err = "truncated BGZF header"; return false; }
    for (uint32_t o = 0; o + 4 <= xlen;) {
      const uint32_t sl = extra[o + 2] | (extra[o + 3] << 8);
      if (extra[o] == 'B' && extra[o + 1] == 'C' && sl == 2) bsize = (extra[o + 4] | (extra[o + 5] << 8)) + 1u;
      o += 4 + sl;
    }
    if (!bsize) { err = "BGZF block without BC field"; return false; }
    const uint32_t clen = bsize - 12 - xlen - 8;
    cbuf_.resize(clen + 8);
    if (fread(cbuf_.data(), 1, clen + 8, f_) != clen + 8) { err = "truncated BGZF block"; return false; }
    next_block_ = block_start_ + bsize;
    const uint32_t isz = cbuf_[clen + 4] | (cbuf_[clen + 5] << 8) | (cbuf_[clen + 6] << 16) | ((uint32_t)cbuf_[clen + 7] << 24);
    if (isz == 0) continue;
    ubuf_.resize(isz);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; }
    zs.next_in = cbuf_.data(); zs.avail_in = clen; zs.next_out = ubuf_.data(); zs.avail_out = isz;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.total_out != isz) { err = "BGZF inflate failed"; return false; }
    return true;
  }
}

bool BamReader::get(void *dst, size_t n, std::string &err) {
  uint8_t *d = static_cast<uint8_t *>(dst);
  while (n) {
    if (upos_ == ubuf_.size()) {
      if (!fill(err)) return false;
      if (eof_) { if (err.empty()) err = "EOF"; return false; }
    }
    const size_t k = std::min(n, ubuf_.size() - upos_);
    memcpy(d, ubuf_.data() + upos_, k);
    d += k; upos_ += k; n -= k;
  }
  return true;
}

bool BamReader::open(const std::string &path, std::string &err) {
  close();
  f_ = fopen(path.c_str(), "rb");
  if (!f_) { err = "couldn't open bam"; return false; }   // extract.nim:276
  eof_ = false; next_block_ = 0; upos_ = 0; ubuf_.clear();
  char magic[4];
  int32_t l_text = 0, n_ref = 0;
  if (!get(magic, 4, err) || memcmp(magic, "BAM\1", 4) != 0) { err = "not a BAM file (CRAM is not supported by this build)"; return false; }
  if (!get(&l_text, 4, err) || l_text < 0) return false;
  text_.resize((size_t)l_text);
  if (l_text && !get(&text_[0], (size_t)l_text, err)) return false;
  while (!text_.empty() && text_.back() == '\0') text_.pop_back();
  if (!get(&n_ref, 4, err) || n_ref < 0) return false;
  targets_.clear();
  for (int32_t i = 0; i < n_ref; ++i) {
    int32_t l_name = 0, l_ref = 0;
    if (!get(&l_name, 4, err) || l_name <= 0) { err = "bad reference name"; return false; }
    std::string nm((size_t)l_name, '\0');
    if (!get(&nm[0], (size_t)l_name, err) || !get(&l_ref, 4, err)) return false;
    nm.pop_back();
    targets_.push_back(BamTarget{nm, (uint32_t)l_ref});
  }
  return true;
}

bool BamReader::seek(Pos p, std::string &err) {
  if (fseeko(f_, (off_t)p.block_off, SEEK_SET) != 0) { err = "seek failed"; return false; }
  next_block_ = p.block_off;
  eof_ = false;
  if (!fill(err)) return false;
  if (p.in_block > ubuf_.size()) { err = "bad virtual offset"; return false; }
  upos_ = p.in_block;
  return true;
}

// BAI (SAM spec 5.2): magic, n_ref, per reference { n_bin, { bin, n_chunk, { beg, end } }, n_intv, ioffset[] }
bool BamReader::load_index(const std::string &bam_path, std::string &err) {
  FILE *f = fopen((bam_path + ".bai").c_str(), "rb");
  if (!f && bam_path.size() > 4) f = fopen((bam_path.substr(0, bam_path.size() - 4) + ".bai").c_str(), "rb");
  if (!f) { err = "no .bai index next to " + bam_path; return false; }
  auto rd = [&](void *p, size_t n) { return fread(p, 1, n, f) == n; };
  char magic[4];
  int32_t n_ref = 0;
  bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && n_ref >= 0;
  lin_.assign((size_t)std::max(n_ref, 0), {});
  ref_beg_.assign((size_t)std::max(n_ref, 0), 0);
  for (int32_t r = 0; ok && r < n_ref; ++r) {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4);
    uint64_t first = 0;
    for (int32_t k = 0; ok && k < n_bin; ++k) {
      uint32_t bin = 0;
      int32_t n_chunk = 0;
      ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
      for (int32_t c = 0; ok && c < n_chunk; ++c) {
        uint64_t be[2];
        ok = rd(be, 16);
        if (ok && bin != 37450 && (first == 0 || be[0] < first)) first = be[0];   // 37450: the metadata pseudo-bin
      }
    }
    int32_t n_intv = 0;
    ok = ok && rd(&n_intv, 4) && n_intv >= 0;
    if (ok) {
      lin_[(size_t)r].resize((size_t)n_intv);
      ok = n_intv == 0 || rd(lin_[(size_t)r].data(), (size_t)n_intv * 8);
      ref_beg_[(size_t)r] = first;
    }
  }
  fclose(f);
  if (!ok) { lin_.clear(); err = "corrupt .bai index"; return false; }
  if (lin_.empty()) lin_.push_back({});   // has_index() for a BAM without references
  return true;
}

int64_t BamReader::read_region(RecordBatch &b, int32_t tid, int64_t beg, int64_t end, std::string &err) {
  if (tid < 0 || (size_t)tid >= ref_beg_.size() || ref_beg_[(size_t)tid] == 0 || end <= beg) return 0;
  const std::vector<uint64_t> &lin = lin_[(size_t)tid];
  // smallest offset of a record overlapping the 16 KiB window of `beg`; empty windows (0) fall back to the nearest
  // earlier filled one, windows beyond the last record-bearing one hold nothing that can overlap
  uint64_t off = 0;
  int64_t w = beg >> 14;
  if (w >= (int64_t)lin.size()) w = (int64_t)lin.size() - 1;
  for (; w >= 0 && off == 0; --w) off = lin[(size_t)w];
  if (off == 0) off = ref_beg_[(size_t)tid];
  if (!seek(Pos{off >> 16, (uint32_t)(off & 0xffff)}, err)) return -1;
  return read_until(b, INT64_MAX, tid, (int32_t)std::min<int64_t>(end, INT32_MAX), err);
}

int64_t BamReader::read_until(RecordBatch &b, int64_t max_records, int32_t stop_tid, int32_t stop_pos, std::string &err) {
  int64_t n = 0;
  std::vector<uint8_t> rec;
  while (n < max_records) {
    if (upos_ == ubuf_.size()) {
      if (!fill(err)) return -1;
      if (eof_) break;
    }
    int32_t bs = 0;
    if (!get(&bs, 4, err)) return -1;
    if (bs < 32) { err = "corrupt BAM record"; return -1; }
    rec.resize((size_t)bs);
    if (!get(rec.data(), (size_t)bs, err)) return -1;
    const uint8_t *p = rec.data();
    int32_t refID, pos, l_seq, next_ref, next_pos, tlen;
    uint8_t l_read_name, mapq;
    uint16_t n_cigar, flag;
    memcpy(&refID, p, 4); memcpy(&pos, p + 4, 4);
    l_read_name = p[8]; mapq = p[9];
    memcpy(&n_cigar, p + 12, 2); memcpy(&flag, p + 14, 2); memcpy(&l_seq, p + 16, 4);
    memcpy(&next_ref, p + 20, 4); memcpy(&next_pos, p + 24, 4); memcpy(&tlen, p + 28, 4);
    const size_t need = 32 + (size_t)l_read_name + 4u * n_cigar + (size_t)(l_seq + 1) / 2;
    if (l_seq < 0 || need > (size_t)bs) { err = "corrupt BAM record"; return -1; }
    if (stop_tid != INT32_MIN && (refID != stop_tid || pos >= stop_pos)) break;
    b.tid.push_back(refID); b.pos.push_back(pos); b.mtid.push_back(next_ref); b.mpos.push_back(next_pos);
    b.isize.push_back(tlen); b.l_seq.push_back(l_seq); b.flag.push_back(flag); b.mapq.push_back(mapq);
    const char *qn = reinterpret_cast<const char *>(p + 32);
    b.qnames.append(qn, l_read_name ? (size_t)l_read_name - 1 : 0);
    b.qname_off.push_back(b.qnames.size());
    const uint8_t *cg = p + 32 + l_read_name;
    for (int j = 0; j < n_cigar; ++j) { uint32_t c; memcpy(&c, cg + 4 * j, 4); b.cigar.push_back(c); }
    b.cigar_off.push_back((uint32_t)b.cigar.size());
    const size_t so = (b.seq4.size() + 15) & ~(size_t)15;
    const size_t sb = (size_t)(l_seq + 1) / 2;
    b.seq4.resize(so + sb, 0);
    memcpy(b.seq4.data() + so, cg + 4u * n_cigar, sb);
    b.seq_off.push_back(so);
    ++n;
  }
  return n;
}

}  // namespace strl
