// bgzf.hip -- BGZF inflate + BAM record parsing on the GPU (gfx950): the front end of `strling extract`
// (extract.nim:275-329 reads the BAM through htslib on one thread).
//
// A BAM file is a sequence of independent BGZF blocks (RFC 1951 DEFLATE streams of <= 64 KiB each).  DEFLATE decoding is
// serial inside a stream, so the parallelism is ACROSS blocks: one lane per BGZF block, 64 blocks per wave, every wave's
// Huffman tables in LDS in a [entry][lane] layout (each lane its own bank).  A 1 GiB superchunk holds ~16 Ki blocks, a whole
// 30x genome ~2.6 M: far more independent streams than the chip has lanes.
//   inflate_kernel : stored / fixed / dynamic blocks.  Canonical-Huffman decoding bit by bit against per-length code counts
//                    kept in registers (10 bits per length, three per VGPR) -- the only table access per symbol is the
//                    final symbol lookup in LDS.  Output goes through a per-lane 1 KiB ring in LDS: its older half
//                    leaves as 16-byte stores, LZ77 matches read their source from the ring, 4 bytes at a time.
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
#include "inflate_core.h"

namespace strl {

struct InflateParams {
  const uint8_t *comp;       // compressed bytes of the superchunk (8 readable bytes of slack behind the last block)
  const uint64_t *coff;      // [n] offset of each block's DEFLATE data in comp
  const uint32_t *clen;      // [n] its length
  const uint64_t *uoff;      // [n] offset of the block's output in out
  const uint32_t *isize;     // [n] inflated size (BGZF footer)
  uint32_t n_blocks;
  uint8_t *out;
  uint32_t *err;             // [1] flags
};

__global__ __launch_bounds__(64) void inflate_kernel(InflateParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t inf_lds[];
  const int lane = threadIdx.x;
  uint16_t *sym_ll = reinterpret_cast<uint16_t *>(inf_lds) + lane;
  uint16_t *sym_d = sym_ll + L_SYMLL * 64;
  uint16_t *cnt = sym_d + L_SYMD * 64;
  uint16_t *offs = cnt + L_CNT * 64;
  uint8_t *win = inf_lds + INF_TAB_BYTES;
  const uint32_t b = blockIdx.x * 64u + (uint32_t)lane;
  if (b >= P.n_blocks) return;
  const int err = inflate_lane(P.comp + P.coff[b], P.clen[b], P.out + P.uoff[b], P.isize[b], sym_ll, sym_d, cnt, offs,
                               reinterpret_cast<uint32_t *>(win), reinterpret_cast<uint32_t *>(win + INF_WIN_BYTES), lane);
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
  STRL_HIP(hipMemsetAsync(static_cast<uint8_t *>(d_comp.p) + comp_bytes, 0, 16, st));   // the bit reader loads one dword ahead
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
