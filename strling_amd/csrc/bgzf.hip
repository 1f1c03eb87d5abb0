// bgzf.hip -- BGZF inflate + BAM record parsing on the GPU (gfx950): the front end of `strling extract`
// (extract.nim:275-329 reads the BAM through htslib on one thread).
//
// A BAM file is a sequence of independent BGZF blocks (RFC 1951 DEFLATE streams of <= 64 KiB each).  DEFLATE decoding is
// serial inside a stream, so the parallelism is ACROSS blocks: one lane per BGZF block, 64 blocks per wave, every wave's
// Huffman tables in LDS in a [entry][lane] layout (each lane its own bank).  A 1 GiB superchunk holds ~16 Ki blocks, a whole
// 30x genome ~2.6 M: far more independent streams than the chip has lanes.
//   inflate_kernel : stored / fixed / dynamic blocks.  Canonical-Huffman decoding bit by bit against per-length code counts
//                    kept in registers (10 bits per length, three per VGPR) -- the only table access per symbol is the
//                    final symbol lookup in LDS.  LZ77 matches are copied byte by byte through the lane's own output
//                    region in global memory (same-thread program order makes overlapping copies correct).
//   The kernels below turn the inflated bytes into the SoA batch the scorer consumes, with no host parsing:
//   rec_first_kernel : one lane per BGZF block finds the first record that STARTS in its block: a candidate offset is
//                    accepted when a chain of BAM records starting there stays plausible (sizes, refIDs, name lengths and
//                    terminators, cigar / sequence lengths consistent with block_size) beyond the block's end,
//   rec_walk_kernel  : each lane walks the records starting in its block (count, then offsets after a prefix sum) and the
//                    junctions are VERIFIED: the walk of block k must end exactly where block k+1's chain starts, else the
//                    batch is flagged and the host decodes it instead -- the guess never changes a result,
//   rec_parse_kernel : one lane per record -> strl_read_soa columns, strl_pair_rec row, qname hash, packed SEQ, qname bytes.
#include <string.h>
#include <algorithm>
#include "common.h"
#include "device_util.h"

namespace strl {

constexpr int INF_ERR_DATA = 1, INF_ERR_SIZE = 2;

__device__ const uint16_t d_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t d_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t d_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t d_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t d_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct InflateParams {
  const uint8_t *comp;       // compressed bytes of the superchunk (8 readable bytes of slack behind the last block)
  const uint64_t *coff;      // [n] offset of each block's DEFLATE data in comp
  const uint32_t *clen;      // [n] its length
  const uint64_t *uoff;      // [n] offset of the block's output in out
  const uint32_t *isize;     // [n] inflated size (BGZF footer)
  uint32_t n_blocks;
  uint8_t *out;
  uint32_t *err;             // [1] flags, [1 + ...] unused
};

// LDS layout of one wave (rows of 64 lanes)
constexpr int L_SYMLL = 288, L_SYMD = 32, L_LENS = 320, L_CNT = 16;
constexpr int INF_LDS_BYTES = (L_SYMLL + L_SYMD) * 64 * 2 + L_LENS * 64 + 2 * L_CNT * 64 * 2;

struct BitReader {
  const uint32_t *wp;
  uint64_t buf;
  int cnt;
  int64_t left;   // bits of the stream not yet loaded into buf
  __device__ void init(const uint8_t *p, uint32_t nbytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    wp = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const int skip = (int)(a & 3);
    buf = (uint64_t)(*wp++) >> (8 * skip);
    cnt = 32 - 8 * skip;
    left = (int64_t)nbytes * 8 - cnt;
  }
  __device__ __forceinline__ void refill() {
    if (cnt <= 32) { buf |= (uint64_t)(*wp++) << cnt; cnt += 32; left -= 32; }
  }
  __device__ __forceinline__ uint32_t bits(int n) {   // n <= 16, caller keeps cnt >= n
    const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
    buf >>= n;
    cnt -= n;
    return v;
  }
  __device__ __forceinline__ uint32_t bit() { const uint32_t v = (uint32_t)buf & 1u; buf >>= 1; --cnt; return v; }
  __device__ bool overrun() const { return left + cnt < 0; }   // consumed more bits than the stream holds
};

// code counts per length, 10 bits each, three per register: c[(len - 1) / 3] >> 10 * ((len - 1) % 3)
struct Counts { uint32_t c[5]; };
#define CNT_OF(C, len) (((C).c[((len) - 1) / 3] >> (10 * (((len) - 1) % 3))) & 0x3ffu)

// canonical Huffman decode (one bit at a time, codes are packed most-significant bit first): <= 15 steps of register work,
// then ONE table access.  Returns -1 for an invalid code.
__device__ __forceinline__ int huff_decode(BitReader &br, const Counts &C, const uint16_t *sym /* LDS column, stride 64 */) {
  int code = 0, first = 0, index = 0;
#pragma unroll
  for (int len = 1; len <= 15; ++len) {
    code |= (int)br.bit();
    const int count = (int)CNT_OF(C, len);
    if (code - count < first) return (int)sym[(index + (code - first)) * 64];
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// build the decoding table of `n` symbols whose code lengths sit in lens (LDS column of bytes, stride 64): counts per
// length -> C, symbols ordered by (length, value) -> sym.  cnt / offs: scratch LDS columns of 16 u16.  Returns false for an
// over-subscribed set of lengths (incomplete sets are allowed: a single distance code is legal).
__device__ bool huff_build(const uint8_t *lens, int n, uint16_t *cnt, uint16_t *offs, uint16_t *sym, Counts &C) {
  for (int l = 0; l < 16; ++l) cnt[l * 64] = 0;
  for (int s = 0; s < n; ++s) { const int l = lens[s * 64]; cnt[l * 64] = (uint16_t)(cnt[l * 64] + 1); }
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= (int)cnt[l * 64];
    if (left < 0) return false;
  }
  offs[1 * 64] = 0;
  for (int l = 1; l < 15; ++l) offs[(l + 1) * 64] = (uint16_t)(offs[l * 64] + cnt[l * 64]);
  for (int s = 0; s < n; ++s) {
    const int l = lens[s * 64];
    if (l) { const int o = offs[l * 64]; sym[o * 64] = (uint16_t)s; offs[l * 64] = (uint16_t)(o + 1); }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) C.c[q] = 0;
#pragma unroll
  for (int l = 1; l <= 15; ++l) C.c[(l - 1) / 3] |= (uint32_t)cnt[l * 64] << (10 * ((l - 1) % 3));
  return true;
}

__global__ __launch_bounds__(64) void inflate_kernel(InflateParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t inf_lds[];
  const int lane = threadIdx.x;
  uint16_t *sym_ll = reinterpret_cast<uint16_t *>(inf_lds) + lane;
  uint16_t *sym_d = sym_ll + L_SYMLL * 64;
  uint16_t *cnt = sym_d + L_SYMD * 64;
  uint16_t *offs = cnt + L_CNT * 64;
  uint8_t *lens = reinterpret_cast<uint8_t *>(reinterpret_cast<uint16_t *>(inf_lds) + (L_SYMLL + L_SYMD + 2 * L_CNT) * 64) + lane;
  const uint32_t b = blockIdx.x * 64u + (uint32_t)lane;
  if (b >= P.n_blocks) return;
  const uint32_t isize = P.isize[b];
  uint8_t *out = P.out + P.uoff[b];
  BitReader br;
  br.init(P.comp + P.coff[b], P.clen[b]);
  uint32_t o = 0;
  int err = 0;
  bool last = false;
  while (!last && !err) {
    br.refill();
    last = br.bit() != 0;
    const uint32_t type = br.bits(2);
    if (type == 0) {                       // stored
      br.bits(br.cnt & 7);                 // to the byte boundary (cnt and the stream position are congruent mod 8)
      br.refill();
      const uint32_t len = br.bits(16);
      br.refill();
      const uint32_t nlen = br.bits(16);
      if ((len ^ 0xffffu) != nlen || o + len > isize) { err = INF_ERR_DATA; break; }
      for (uint32_t i = 0; i < len; ++i) { br.refill(); out[o++] = (uint8_t)br.bits(8); }
      continue;
    }
    if (type == 3) { err = INF_ERR_DATA; break; }
    Counts CL{}, CD{};
    if (type == 1) {                       // fixed codes
      for (int s = 0; s < 144; ++s) lens[s * 64] = 8;
      for (int s = 144; s < 256; ++s) lens[s * 64] = 9;
      for (int s = 256; s < 280; ++s) lens[s * 64] = 7;
      for (int s = 280; s < 288; ++s) lens[s * 64] = 8;
      huff_build(lens, 288, cnt, offs, sym_ll, CL);
      for (int s = 0; s < 30; ++s) lens[s * 64] = 5;
      huff_build(lens, 30, cnt, offs, sym_d, CD);
    } else {                               // dynamic codes
      const int nlen = (int)br.bits(5) + 257;
      const int ndist = (int)br.bits(5) + 1;
      const int ncode = (int)br.bits(4) + 4;
      if (nlen > 286 || ndist > 30) { err = INF_ERR_DATA; break; }
      for (int i = 0; i < 19; ++i) lens[i * 64] = 0;
      for (int i = 0; i < ncode; ++i) { br.refill(); lens[d_clorder[i] * 64] = (uint8_t)br.bits(3); }
      Counts CC{};
      if (!huff_build(lens, 19, cnt, offs, sym_d, CC)) { err = INF_ERR_DATA; break; }   // the code-length code lives in sym_d for now
      int idx = 0;
      while (idx < nlen + ndist) {
        br.refill();
        const int s = huff_decode(br, CC, sym_d);
        if (s < 0) { err = INF_ERR_DATA; break; }
        if (s < 16) lens[(idx++) * 64] = (uint8_t)s;
        else {
          int prev = 0, rep;
          if (s == 16) {
            if (idx == 0) { err = INF_ERR_DATA; break; }
            prev = lens[(idx - 1) * 64];
            rep = 3 + (int)br.bits(2);
          } else if (s == 17) rep = 3 + (int)br.bits(3);
          else rep = 11 + (int)br.bits(7);
          if (idx + rep > nlen + ndist) { err = INF_ERR_DATA; break; }
          while (rep--) lens[(idx++) * 64] = (uint8_t)prev;
        }
      }
      if (err) break;
      if (lens[256 * 64] == 0) { err = INF_ERR_DATA; break; }                     // no end-of-block code
      if (!huff_build(lens, nlen, cnt, offs, sym_ll, CL)) { err = INF_ERR_DATA; break; }
      if (!huff_build(lens + nlen * 64, ndist, cnt, offs, sym_d, CD)) { err = INF_ERR_DATA; break; }
    }
    // ---- the symbols of this block ----
    for (;;) {
      br.refill();
      int s = huff_decode(br, CL, sym_ll);
      if (s < 0) { err = INF_ERR_DATA; break; }
      if (s < 256) {
        if (o >= isize) { err = INF_ERR_SIZE; break; }
        out[o++] = (uint8_t)s;
      } else if (s == 256) break;
      else {
        s -= 257;
        if (s >= 29) { err = INF_ERR_DATA; break; }
        const uint32_t len = d_lbase[s] + br.bits(d_lext[s]);
        br.refill();
        const int ds = huff_decode(br, CD, sym_d);
        if (ds < 0 || ds >= 30) { err = INF_ERR_DATA; break; }
        const uint32_t dist = d_dbase[ds] + br.bits(d_dext[ds]);
        if (dist > o || o + len > isize) { err = INF_ERR_DATA; break; }
        const uint8_t *src = out + o - dist;
        uint8_t *dst = out + o;
        for (uint32_t i = 0; i < len; ++i) dst[i] = src[i];
        o += len;
      }
      if (br.overrun()) { err = INF_ERR_DATA; break; }
    }
  }
  if (!err && o != isize) err = INF_ERR_SIZE;
  if (err) atomicOr(P.err, (uint32_t)err);
}

}  // namespace strl

using namespace strl;

// Inflate n DEFLATE streams (device arrays as in InflateParams); asynchronous on the context stream.
int strl_inflate_device(strl_ctx *c, const uint8_t *d_comp, const uint64_t *d_coff, const uint32_t *d_clen, const uint64_t *d_uoff,
                        const uint32_t *d_isize, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_err) {
  if (!n_blocks) return STRL_OK;
  InflateParams P{d_comp, d_coff, d_clen, d_uoff, d_isize, n_blocks, d_out, d_err};
  static bool attr_done = false;
  if (!attr_done) {
    STRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(inflate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, INF_LDS_BYTES));
    attr_done = true;
  }
  hipLaunchKernelGGL(inflate_kernel, dim3((n_blocks + 63) / 64), dim3(64), INF_LDS_BYTES, c->stream, P);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}

// C ABI for tests and hosts that hold the compressed blocks in memory: inflate n raw DEFLATE streams on the device.
extern "C" int strl_inflate_blocks(strl_ctx *c, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen,
                                   const uint32_t *isize, uint32_t n_blocks, uint8_t *out, uint64_t out_bytes) {
  if (!c || (n_blocks && (!comp || !coff || !clen || !isize || !out))) { set_error("null argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  std::vector<uint64_t> uoff(n_blocks);
  uint64_t tot = 0;
  for (uint32_t i = 0; i < n_blocks; ++i) {
    if (coff[i] + clen[i] > comp_bytes) { set_error("block %u reaches past the compressed buffer", i); return STRL_ERR_ARG; }
    uoff[i] = tot;
    tot += isize[i];
  }
  if (tot > out_bytes) { set_error("output buffer too small: %llu needed", (unsigned long long)tot); return STRL_ERR_CAPACITY; }
  DevBuf d_comp, d_meta, d_out;
  int rc;
  const size_t meta = (size_t)n_blocks * (8 + 4 + 8 + 4) + 64;
  if ((rc = d_comp.reserve(comp_bytes + 16)) || (rc = d_meta.reserve(meta)) || (rc = d_out.reserve(tot + 16))) return rc;
  uint64_t *m_coff = d_meta.as<uint64_t>(), *m_uoff = m_coff + n_blocks;
  uint32_t *m_clen = reinterpret_cast<uint32_t *>(m_uoff + n_blocks), *m_isize = m_clen + n_blocks, *m_err = m_isize + n_blocks;
  hipStream_t st = c->stream;
  STRL_HIP(hipMemcpyAsync(d_comp.p, comp, comp_bytes, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemsetAsync(static_cast<uint8_t *>(d_comp.p) + comp_bytes, 0, 16, st));
  STRL_HIP(hipMemcpyAsync(m_coff, coff, (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_uoff, uoff.data(), (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_clen, clen, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_isize, isize, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemsetAsync(m_err, 0, 4, st));
  if ((rc = strl_inflate_device(c, d_comp.as<uint8_t>(), m_coff, m_clen, m_uoff, m_isize, n_blocks, d_out.as<uint8_t>(), m_err))) return rc;
  uint32_t err = 0;
  STRL_HIP(hipMemcpyAsync(&err, m_err, 4, hipMemcpyDeviceToHost, st));
  if (tot) STRL_HIP(hipMemcpyAsync(out, d_out.p, tot, hipMemcpyDeviceToHost, st));
  STRL_HIP(hipStreamSynchronize(st));
  d_comp.release(); d_meta.release(); d_out.release();
  if (err) { set_error("device inflate: %s", (err & INF_ERR_DATA) ? "invalid DEFLATE data" : "inflated size differs from the block's ISIZE"); return STRL_ERR_FORMAT; }
  return STRL_OK;
}
