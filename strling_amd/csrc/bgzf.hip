// bgzf.hip -- the BGZF / BAM front end of `strling extract` on the GPU (gfx950): extract.nim:275-329 reads the BAM through
// htslib on one thread; SURVEY section 8f N3.
//
// A BAM file is a sequence of independent BGZF blocks (RFC 1951 DEFLATE streams of <= 64 KiB each).
//   inflate_kernel : ONE WAVE per BGZF block (inflate_wave.h): wave-uniform symbol loop split between the scalar and the vector unit, first-level
//                    Huffman tables in LDS built by the 64 lanes together, input through a lane-register window, literals
//                    through a lane register, LZ77 matches copied by the whole wave.  stored / fixed / dynamic blocks,
//                    multi-block streams.  A block the decoder refuses is flagged in status[] (the host hands exactly those
//                    to zlib, whose verdict stands).
#include <string.h>
#include <algorithm>
#include "common.h"
#include "device_util.h"
#include "inflate_wave.h"
// The grouped form of the decoder (inflate_group.h: a block per 8 lanes) measured slower than the wave form everywhere it was
// tried (profiles/r05/inflate_group/): it is an experiment, kept with its logs and its CPU tests, and NOT part of the shipped
// library -- a second decoder of untrusted input nobody benefits from.  `STRL_WITH_INFLATE_GROUP=1 python -m strling_amd.build`
// compiles it in (then STRL_INFLATE_FORM=group selects it at run time).
#ifdef STRL_WITH_INFLATE_GROUP
#include "inflate_group.h"
#endif

namespace strl {

struct InflateParams {
  const uint8_t *comp;       // compressed bytes of the superchunk
  uint64_t readable;         // bytes of comp that may be loaded: a multiple of 4, >= the end of the last stream
  const uint64_t *coff;      // [n] offset of each block's DEFLATE data in comp
  const uint32_t *clen;      // [n] its length
  const uint64_t *uoff;      // [n] offset of the block's output in out
  const uint32_t *isize;     // [n] inflated size (BGZF footer)
  uint32_t n_blocks;
  uint8_t *out;
  uint32_t *err;             // [1] IW_ERR_* flags of all blocks
  uint8_t *status;           // [n] per block (may be null)
  uint64_t out_bytes;        // bytes of `out` the blocks lie in (the grouped form bounds its stores by it; 0: not known)
  uint8_t *work;             // the grouped form's workspace: IG_WORK_STRIDE bytes a block
  uint64_t work_bytes;
};

#ifndef STRL_INFLATE_WAVES
#define STRL_INFLATE_WAVES 6
#endif
#ifndef STRL_IG_WAVES
#define STRL_IG_WAVES 3                    // waves per SIMD the grouped form's registers are held to (G = 8)
#endif
#ifndef STRL_INFLATE_GROUP_DEFAULT
#define STRL_INFLATE_GROUP_DEFAULT 0     // which form runs when STRL_INFLATE_FORM is not set
#endif
// (waves per SIMD: the scalar unit bounds the kernel and half the wave-cycles are spent parked on s_waitcnt -- more waves, not fewer registers per se)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STRL_INFLATE_WAVES, 8))) void inflate_kernel(InflateParams P) {
  // (iw_run forms a table entry's address by a bit-field insert into the table's own address: the tables sit on multiples of their size)
  __shared__ __attribute__((aligned(4u << IW_LIT_ROOT))) IwLds lds;
  static_assert(offsetof(IwLds, lit_tab) == 0 && offsetof(IwLds, dist_tab) % (4u << IW_DIST_ROOT) == 0, "table alignment");
  const uint32_t b = blockIdx.x;
  const int rc = iw_inflate(P.comp, P.coff[b], P.clen[b], P.readable, P.out + P.uoff[b], P.isize[b], lds);
  if (threadIdx.x == 0) {
    if (P.status) P.status[b] = (uint8_t)rc;
    if (rc) atomicOr(P.err, (uint32_t)rc);
  }
}

#ifdef STRL_WITH_INFLATE_GROUP
// The grouped form (inflate_group.h): G lanes per block, 64 / G blocks per wave; one wave per workgroup, twelve workgroups per
// CU at G = 8 (13 KB of LDS each, <= 168 registers).  32-bit offsets into the buffer descriptors: the host only launches it for
// < 2 GiB each.
template <int G>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(G == 8 ? STRL_IG_WAVES : 1, 8))) void inflate_group_kernel(InflateParams P) {
  __shared__ IgLds<G> lds[64 / G];
  const int lane = (int)threadIdx.x, grp = lane / G, sub = lane % G;
  const uint32_t b = blockIdx.x * (uint32_t)(64 / G) + (uint32_t)grp;
  // (the match copies load dwords at byte offsets: their descriptor reaches three bytes past the output; every caller's buffer has >= 16 behind it)
  const IwBuf in = iw_make_buf(P.comp, P.readable), out = iw_make_buf(P.out, P.out_bytes), out_ld = iw_make_buf(P.out, P.out_bytes + 3);
  const IwBuf work = iw_make_buf(P.work, P.work_bytes);
  if (b < P.n_blocks) {
    const int rc = ig_inflate<G>(in, (uint32_t)P.coff[b], P.clen[b], out, out_ld, (uint32_t)P.uoff[b], P.isize[b], work, b * IG_WORK_STRIDE, lds[grp], sub);
    if (sub == 0) {
      if (P.status) P.status[b] = (uint8_t)rc;
      if (rc) atomicOr(P.err, (uint32_t)rc);
    }
  }
}
#endif

// ---- CRC-32 of every block's inflated bytes against the BGZF trailer (htslib refuses a block whose CRC differs: bgzf.c) ----
// The CRC is linear over GF(2): CRC(M) = sum over the dwords w_i of M of  w_i * x^(8 * bytes behind w_i)  (+ the start value's
// and the final complement's constants).  So the dwords may be dealt to the lanes in ANY way -- here lane j takes dwords
// j, j + 64, j + 128, ...: every load of the wave is one coalesced 256-byte read (contiguous KiB slices per lane were tried
// first: 64 cache lines per load instruction, 18 ms per 4.6 GB).  A lane's Horner step is "advance 256 bytes, add the next
// dword": the advance is four table lookups (one 256-entry table per byte of the state, in LDS); at the end every lane advances
// by the bytes behind its last dword (the "2^k zero bytes" operators zlib's crc32_combine squares) and the lanes XOR together.
struct CrcTables {
  uint32_t tab[256];          // the byte table of the reflected polynomial 0xedb88320
  uint32_t zop[17][32];       // operator of 2^k zero bytes, one column per state bit
  uint32_t adv[4][256];       // advance by 256 zero bytes: adv[t][v] = zop[8] applied to v << 8 t
};
__device__ __forceinline__ uint32_t crc_zeros(const uint32_t (*zop)[32], uint32_t v, uint32_t n_bytes) {
  for (int k = 0; k < 17 && v; ++k) {
    if (n_bytes & (1u << k)) {
      uint32_t r = 0;
      for (int b = 0; b < 32; ++b) r ^= (v >> b) & 1u ? zop[k][b] : 0u;
      v = r;
    }
  }
  return v;
}
__global__ __launch_bounds__(256) void crc32_kernel(const uint8_t *out, const uint64_t *uoff, const uint32_t *isize, const uint32_t *expect, uint32_t n_blocks,
                                                     const CrcTables *T, uint8_t *status, uint32_t *err) {
  __shared__ uint32_t tab[256];
  __shared__ uint32_t zop[17][32];
  __shared__ uint32_t adv[4][256];
  for (int i = threadIdx.x; i < 256; i += 256) tab[i] = T->tab[i];
  for (int i = threadIdx.x; i < 17 * 32; i += 256) (&zop[0][0])[i] = (&T->zop[0][0])[i];
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&adv[0][0])[i] = (&T->adv[0][0])[i];
  __syncthreads();
  const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (b >= n_blocks) return;
  const uint32_t n = isize[b], nd = n >> 2;
  const uint8_t *p = out + uoff[b];
  uint32_t c = 0, last = 0;
  bool any = false;
  for (uint32_t i = lane; i < nd; i += 64u) {
    uint32_t w;
    __builtin_memcpy(&w, p + 4u * i, 4);
    c = adv[0][c & 0xffu] ^ adv[1][(c >> 8) & 0xffu] ^ adv[2][(c >> 16) & 0xffu] ^ adv[3][c >> 24] ^ w;
    last = i;
    any = true;
  }
  if (any) c = crc_zeros(zop, c, n - 4u * last);       // through the dword's own four bytes and everything behind it
  if (lane == 0) {
    uint32_t t = 0;                                     // the block's last 0..3 bytes
    for (uint32_t i = 4u * nd; i < n; ++i) t = tab[(t ^ p[i]) & 0xffu] ^ (t >> 8);
    c ^= t ^ crc_zeros(zop, 0xffffffffu, n);            // ... and the all-ones start value, run through the whole block
  }
  for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d);
  if (lane == 0 && (~c) != expect[b]) {
    if (status) status[b] |= (uint8_t)IW_ERR_CRC;
    if (err) atomicOr(err, (uint32_t)IW_ERR_CRC);
  }
}

// ---- `strling call`'s evidence reads (call.nim:196-218: one bam.query(tid, left - window, right + window) per bound) for many
// bounds at once.  The host finds each region's BGZF blocks through the .bai linear index (a region's blocks are consecutive in
// the file and inflate to one contiguous piece of `u`); after the inflate ONE LANE per region walks its records from the index
// offset on -- htslib's iterator does the same walk -- and notes the byte range from the first record that can reach past `beg`
// to the first record at or behind `end` (or on another reference); a second launch copies exactly those bytes out.  What goes
// back to the host is the tenth of the inflated bytes the query returns, not all the index makes one read.
struct RegionWalk { uint64_t start, stop; };
__device__ __forceinline__ uint32_t rg_ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
// One WAVE per region.  The walk is serial (a record's place follows from the length of the one before it) and, lane per
// region straight out of global memory, every step was a dependent miss: 5 - 6 ms for ~4600 records.  Here the wave copies a
// 4 KiB window of the stream into LDS with one coalesced load per KiB and every lane walks the ~13 records inside it out of
// LDS (all lanes the same values: nothing diverges); a record whose cigar does not fit the window is read from global memory.
constexpr uint32_t RW_WIN = 4096;
__global__ __launch_bounds__(64) void region_walk_kernel(const uint8_t *u, uint64_t u_readable, const uint64_t *uoff, const uint32_t *isize, const strl_region_req *req, uint32_t n,
                                                          RegionWalk *range, uint8_t *status) {
  __shared__ __attribute__((aligned(16))) uint8_t win[RW_WIN];
  const uint32_t r = blockIdx.x, lane = threadIdx.x;
  if (r >= n) return;
  const strl_region_req q = req[r];
  const uint32_t last = q.first_block + q.n_blocks - 1u;
  const uint64_t base = uoff[q.first_block], lim = uoff[last] + isize[last];
  uint64_t p = base + q.in_block, keep = ~0ull, w0 = 0, w1 = 0;      // the window holds u[w0, w1)
  auto load_win = [&](uint64_t at) {
    __builtin_amdgcn_wave_barrier();
    w0 = at & ~(uint64_t)15;
    w1 = w0 + RW_WIN < u_readable ? w0 + RW_WIN : u_readable;        // (u is allocated 64 bytes past the last block; bytes past `lim` are never interpreted)
#pragma unroll
    for (uint32_t k = 0; k < RW_WIN / 1024; ++k) {
      const uint64_t o = w0 + 1024ull * k + 16ull * lane;
      if (o + 16 <= w1) *reinterpret_cast<uint4 *>(win + 1024u * k + 16u * lane) = *reinterpret_cast<const uint4 *>(u + o);
    }
    w1 &= ~(uint64_t)15;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  uint8_t st = 1;                                      // the inflated bytes end before a record that stops the query: the host reads this one itself
  while (p + 36 <= lim) {
    if (p < w0 || p + 36 > w1) {
      load_win(p);
      if (p + 36 > w1) break;
    }
    const uint8_t *h = win + (p - w0);
    const uint32_t bs = rg_ld32(h);
    if (bs < 32u || bs > (1u << 28)) break;            // not a record (st stays 1: the host's reader gives the verdict)
    const int32_t ref = (int32_t)rg_ld32(h + 4), pos = (int32_t)rg_ld32(h + 8);
    if (ref != q.tid || pos >= q.end) { st = 0; break; }
    if (p + 4 + bs > lim) break;
    if (keep == ~0ull) {
      // an upper bound of bam_endpos: every cigar operation counted as if it consumed the reference (a record that is dropped
      // here cannot overlap; one that is kept for nothing is filtered by strl_spanners like on the host path)
      const uint32_t l_name = h[12], n_cig = rg_ld32(h + 16) & 0xffffu;
      const uint64_t need = 36ull + l_name + 4ull * n_cig;
      int64_t span = 1;
      if (need <= 4ull + bs) {
        if (p + need > w1 && need <= RW_WIN - 16) { load_win(p); h = win + (p - w0); }
        if (p + need <= w1) for (uint32_t k = 0; k < n_cig; ++k) span += rg_ld32(h + 36 + l_name + 4u * k) >> 4;
        else for (uint32_t k = 0; k < n_cig; ++k) span += rg_ld32(u + p + 36 + l_name + 4u * k) >> 4;      // (a cigar of a thousand operations)
      } else span = 1ll << 40;                         // malformed: keep it, the host's parser reports it
      if ((int64_t)pos + span > (int64_t)q.beg) keep = p;
    }
    p += 4ull + bs;
  }
  if (st == 0 && keep == ~0ull) keep = p;
  if (lane == 0) {
    range[r] = RegionWalk{st ? 0ull : keep, st ? 0ull : p};
    status[r] = st;
  }
}
// a block per region: bytes [start, stop) of `u` to out + off[r]; off[r] = start (mod 16), so the body moves in 16-byte pieces
__global__ __launch_bounds__(256) void region_copy_kernel(const uint8_t *u, const RegionWalk *range, const uint64_t *off, uint32_t n, uint8_t *out) {
  for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
    const uint64_t s = range[r].start, e = range[r].stop;
    if (e <= s) continue;
    const uint8_t *src = u + s;
    uint8_t *dst = out + off[r];
    const uint64_t len = e - s;
    uint64_t head = (16u - (uint32_t)(s & 15u)) & 15u;
    if (head > len) head = len;
    const uint64_t body = (len - head) >> 4, tail0 = head + (body << 4);
    if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + head);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
    for (uint64_t i = threadIdx.x; i < body; i += 256) d4[i] = s4[i];
    if (tail0 + threadIdx.x < len) dst[tail0 + threadIdx.x] = src[tail0 + threadIdx.x];
  }
}

}  // namespace strl

using namespace strl;

// host: the tables of crc32_kernel (zlib's polynomial 0xedb88320, reflected), made once
static CrcTables make_crc_tables() {
  CrcTables T;
  {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      T.tab[i] = c;
    }
    for (int b = 0; b < 32; ++b) { const uint32_t v = 1u << b; T.zop[0][b] = T.tab[v & 0xffu] ^ (v >> 8); }   // one zero byte
    for (int k = 1; k < 17; ++k)
      for (int b = 0; b < 32; ++b) {
        uint32_t v = T.zop[k - 1][b], r = 0;
        for (int j = 0; j < 32; ++j) if ((v >> j) & 1u) r ^= T.zop[k - 1][j];
        T.zop[k][b] = r;
      }
    for (int t = 0; t < 4; ++t)
      for (uint32_t v = 0; v < 256; ++v) {
        const uint32_t x = v << (8 * t);
        uint32_t r = 0;
        for (int j = 0; j < 32; ++j) if ((x >> j) & 1u) r ^= T.zop[8][j];
        T.adv[t][v] = r;
      }
  }
  return T;
}
static const CrcTables &crc_tables() {
  static const CrcTables T = make_crc_tables();   // (initialised once, thread-safe: contexts of several feeding threads share it)
  return T;
}

// CRC-32 of the inflated blocks against `d_crc` (the BGZF trailers' values): mismatches set IW_ERR_CRC in d_status
int strl_crc_device(strl_ctx *c, const uint8_t *d_out, const uint64_t *d_uoff, const uint32_t *d_isize, const uint32_t *d_crc, uint32_t n_blocks, uint8_t *d_status,
                    uint32_t *d_err, hipStream_t st) {
  if (!n_blocks) return STRL_OK;
  if (!c->crc_tab.p) {
    int rc = c->crc_tab.reserve(sizeof(CrcTables));
    if (rc) return rc;
    STRL_HIP(hipMemcpy(c->crc_tab.p, &crc_tables(), sizeof(CrcTables), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(crc32_kernel, dim3((n_blocks + 3) / 4), dim3(256), 0, st, d_out, d_uoff, d_isize, d_crc, n_blocks, c->crc_tab.as<CrcTables>(), d_status, d_err);
  STRL_HIP(hipGetLastError());
  return STRL_OK;
}

// Inflate n DEFLATE streams (device arrays as in InflateParams); asynchronous on `st`.
// Which form a launch takes: 0 = the wave form, else the lanes per block of the grouped form.
// STRL_INFLATE_FORM = wave | group (G lanes per block, STRL_INFLATE_G = 4 | 8 | 16)
int strl_inflate_form() {
#ifndef STRL_WITH_INFLATE_GROUP
  return 0;
#else
  static const int form_g = [] {
    const char *f = getenv("STRL_INFLATE_FORM"), *g = getenv("STRL_INFLATE_G");
    if (f && !strcmp(f, "wave")) return 0;
    if (f && strcmp(f, "group")) return 0;
    if (!f && !STRL_INFLATE_GROUP_DEFAULT) return 0;
    const int G = g ? atoi(g) : 8;
    return G == 4 || G == 16 ? G : 8;
  }();
  return form_g;
#endif
}
// The grouped form's workspace for n_blocks blocks (0 bytes when the wave form runs): the caller keeps it beside the launch's
// other buffers -- one per stream that may hold a launch.
size_t strl_inflate_work_bytes(uint32_t n_blocks) {
#ifdef STRL_WITH_INFLATE_GROUP
  return strl_inflate_form() ? (size_t)n_blocks * IG_WORK_STRIDE + 64 : 0;
#else
  (void)n_blocks;
  return 0;
#endif
}

// out_bytes: the bytes of d_out the blocks' outputs lie in; d_work / work_bytes: strl_inflate_work_bytes(n_blocks) of device
// memory.  Without either (0 / null) the wave form, which needs neither, is launched.
int strl_inflate_device(strl_ctx *c, const uint8_t *d_comp, uint64_t readable, const uint64_t *d_coff, const uint32_t *d_clen, const uint64_t *d_uoff,
                        const uint32_t *d_isize, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_err, uint8_t *d_status, hipStream_t st, uint64_t out_bytes,
                        uint8_t *d_work, size_t work_bytes) {
  if (!n_blocks) return STRL_OK;
  InflateParams P{d_comp, readable & ~(uint64_t)3, d_coff, d_clen, d_uoff, d_isize, n_blocks, d_out, d_err, d_status, out_bytes, d_work, work_bytes};
  static const unsigned lds_pad = getenv("STRL_INFLATE_LDS_PAD") ? (unsigned)atoi(getenv("STRL_INFLATE_LDS_PAD")) : 0u;   // (occupancy experiments: unused dynamic LDS)
#ifdef STRL_WITH_INFLATE_GROUP
  const int form_g = strl_inflate_form();
  if (form_g && out_bytes && out_bytes < 0x7ffffff0ull && readable < 0x7ffffff0ull && d_work && work_bytes >= (size_t)n_blocks * IG_WORK_STRIDE) {
    const unsigned per = 64u / (unsigned)form_g, grid = (n_blocks + per - 1) / per;
    if (form_g == 4) hipLaunchKernelGGL(inflate_group_kernel<4>, dim3(grid), dim3(64), lds_pad, st, P);
    else if (form_g == 16) hipLaunchKernelGGL(inflate_group_kernel<16>, dim3(grid), dim3(64), lds_pad, st, P);
    else hipLaunchKernelGGL(inflate_group_kernel<8>, dim3(grid), dim3(64), lds_pad, st, P);
  } else
#endif
  {
    hipLaunchKernelGGL(inflate_kernel, dim3(n_blocks), dim3(64), lds_pad, st, P);
  }
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
  DevBuf d_comp, d_meta, d_out, d_work;
  int rc;
  const size_t meta = (size_t)n_blocks * (8 + 4 + 8 + 4) + 64, work = strl_inflate_work_bytes(n_blocks);
  const uint64_t readable = (comp_bytes + 3) & ~(uint64_t)3;
  if ((rc = d_comp.reserve(readable + 16)) || (rc = d_meta.reserve(meta)) || (rc = d_out.reserve(tot + 16)) || (work && (rc = d_work.reserve(work)))) return rc;
  uint64_t *m_coff = d_meta.as<uint64_t>(), *m_uoff = m_coff + n_blocks;
  uint32_t *m_clen = reinterpret_cast<uint32_t *>(m_uoff + n_blocks), *m_isize = m_clen + n_blocks, *m_err = m_isize + n_blocks;
  hipStream_t st = c->stream;
  STRL_HIP(hipMemcpyAsync(d_comp.p, comp, comp_bytes, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemsetAsync(static_cast<uint8_t *>(d_comp.p) + comp_bytes, 0, 16, st));   // the last dword may be partial
  STRL_HIP(hipMemcpyAsync(m_coff, coff, (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_uoff, uoff.data(), (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_clen, clen, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_isize, isize, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemsetAsync(m_err, 0, 4, st));
  hipEvent_t e0, e1;
  STRL_HIP(hipEventCreate(&e0));
  STRL_HIP(hipEventCreate(&e1));
  STRL_HIP(hipEventRecord(e0, st));
  if ((rc = strl_inflate_device(c, d_comp.as<uint8_t>(), readable, m_coff, m_clen, m_uoff, m_isize, n_blocks, d_out.as<uint8_t>(), m_err, nullptr, st, tot, d_work.as<uint8_t>(), work))) return rc;
  STRL_HIP(hipEventRecord(e1, st));
  uint32_t err = 0;
  STRL_HIP(hipMemcpyAsync(&err, m_err, 4, hipMemcpyDeviceToHost, st));
  if (tot) STRL_HIP(hipMemcpyAsync(out, d_out.p, tot, hipMemcpyDeviceToHost, st));
  STRL_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  c->inflate_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  d_comp.release(); d_meta.release(); d_out.release(); d_work.release();
  if (err) { set_error("device inflate: %s", (err & IW_ERR_DATA) ? "invalid DEFLATE data" : "inflated size differs from the block's ISIZE"); return STRL_ERR_FORMAT; }
  return STRL_OK;
}

// HIP-event time (ms) of the inflate kernel of the last strl_inflate_blocks call on this context (copies excluded).
extern "C" int strl_ctx_inflate_ms(strl_ctx *c, double *ms) {
  if (!c || !ms) return STRL_ERR_ARG;
  *ms = c->inflate_ms;
  return STRL_OK;
}

// C ABI: the records of many region queries of one BAM (call.nim:196-218), inflated and cut out on the device.
extern "C" int strl_regions_fetch(strl_ctx *c, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *coff, const uint32_t *clen, const uint32_t *isize,
                                  const uint32_t *crc32, uint32_t n_blocks, const strl_region_req *req, uint32_t n_regions, uint8_t *out, uint64_t out_cap,
                                  uint64_t *out_off, uint64_t *out_len, uint8_t *status) {
  if (!c || (n_blocks && (!comp || !coff || !clen || !isize)) || (n_regions && (!req || !out_off || !out_len || !status)) || (out_cap && !out)) { set_error("null argument"); return STRL_ERR_ARG; }
  if (!n_regions) return STRL_OK;
  STRL_HIP(hipSetDevice(c->device));
  std::vector<uint64_t> uoff(n_blocks);
  uint64_t tot = 0;
  for (uint32_t i = 0; i < n_blocks; ++i) {
    if (coff[i] + clen[i] > comp_bytes) { set_error("block %u reaches past the compressed buffer", i); return STRL_ERR_ARG; }
    if (isize[i] > 65536u) { set_error("block %u: ISIZE %u", i, isize[i]); return STRL_ERR_FORMAT; }
    uoff[i] = tot;
    tot += isize[i];
  }
  for (uint32_t r = 0; r < n_regions; ++r)
    if (!req[r].n_blocks || (uint64_t)req[r].first_block + req[r].n_blocks > n_blocks) { set_error("region %u names blocks that were not handed over", r); return STRL_ERR_ARG; }
  // a slot of the context: its stream, its buffers (kept between calls: freeing gigabytes synchronises the device)
  strl_ctx::RegionSlot *slot = nullptr;
  {
    std::unique_lock<std::mutex> lk(c->rg_mu);
    c->rg_cv.wait(lk, [&] { return !c->rg[0].busy || !c->rg[1].busy; });
    slot = !c->rg[0].busy ? &c->rg[0] : &c->rg[1];
    slot->busy = true;
    if (crc32 && !c->crc_tab.p) {                       // (made once, by whichever call comes first)
      if (c->crc_tab.reserve(sizeof(CrcTables)) != STRL_OK || hipMemcpy(c->crc_tab.p, &crc_tables(), sizeof(CrcTables), hipMemcpyHostToDevice) != hipSuccess) {
        slot->busy = false;
        set_error("CRC tables");
        return STRL_ERR_HIP;
      }
    }
  }
  // (the slot's stream is drained before the slot -- and the stack memory its copies read and write: uoff, err, range -- is given
  // up, whichever way the function is left: an early error return must not leave DMA pending on destroyed memory; round-4 advisor)
  struct Rel {
    strl_ctx *c; strl_ctx::RegionSlot *s;
    ~Rel() {
      if (s->st) (void)hipStreamSynchronize(s->st);
      { std::lock_guard<std::mutex> lk(c->rg_mu); s->busy = false; }
      c->rg_cv.notify_all();
    }
  };
  std::vector<RegionWalk> range(n_regions);          // declared ahead of `rel`: destroyed after its destructor has drained the stream
  uint32_t err = 0;
  Rel rel{c, slot};
  if (!slot->st) STRL_HIP(hipStreamCreateWithFlags(&slot->st, hipStreamNonBlocking));
  DevBuf &d_comp = slot->comp, &d_meta = slot->meta, &d_u = slot->u, &d_out = slot->out, &d_rq = slot->rq, &d_work = slot->work;
  int rc;
  const uint64_t readable = (comp_bytes + 3) & ~(uint64_t)3;
  const size_t meta = (size_t)n_blocks * (8 + 8 + 4 + 4 + 4) + 64;
  const size_t rq_bytes = (size_t)n_regions * (sizeof(strl_region_req) + sizeof(RegionWalk) + 8 + 1) + 64;
  const size_t work = strl_inflate_work_bytes(n_blocks);
  if ((rc = d_comp.reserve(readable + 16)) || (rc = d_meta.reserve(meta)) || (rc = d_u.reserve(tot + 64)) || (rc = d_rq.reserve(rq_bytes)) || (work && (rc = d_work.reserve(work)))) return rc;
  uint64_t *m_coff = d_meta.as<uint64_t>(), *m_uoff = m_coff + n_blocks;
  uint32_t *m_clen = reinterpret_cast<uint32_t *>(m_uoff + n_blocks), *m_isize = m_clen + n_blocks, *m_crc = m_isize + n_blocks, *m_err = m_crc + n_blocks;
  RegionWalk *d_range = d_rq.as<RegionWalk>();
  uint64_t *d_off = reinterpret_cast<uint64_t *>(d_range + n_regions);
  strl_region_req *d_req = reinterpret_cast<strl_region_req *>(d_off + n_regions);
  uint8_t *d_status = reinterpret_cast<uint8_t *>(d_req + n_regions);
  hipStream_t st = slot->st;
  STRL_HIP(hipMemcpyAsync(d_comp.p, comp, comp_bytes, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemsetAsync(static_cast<uint8_t *>(d_comp.p) + comp_bytes, 0, 16, st));
  STRL_HIP(hipMemcpyAsync(m_coff, coff, (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_uoff, uoff.data(), (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_clen, clen, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(m_isize, isize, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  if (crc32) STRL_HIP(hipMemcpyAsync(m_crc, crc32, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemcpyAsync(d_req, req, (size_t)n_regions * sizeof(strl_region_req), hipMemcpyHostToDevice, st));
  STRL_HIP(hipMemsetAsync(m_err, 0, 4, st));
  if ((rc = strl_inflate_device(c, d_comp.as<uint8_t>(), readable, m_coff, m_clen, m_uoff, m_isize, n_blocks, d_u.as<uint8_t>(), m_err, nullptr, st, tot, d_work.as<uint8_t>(), work))) return rc;
  if (crc32 && (rc = strl_crc_device(c, d_u.as<uint8_t>(), m_uoff, m_isize, m_crc, n_blocks, nullptr, m_err, st))) return rc;
  hipLaunchKernelGGL(region_walk_kernel, dim3(n_regions), dim3(64), 0, st, d_u.as<uint8_t>(), (tot + 64) & ~(uint64_t)15, m_uoff, m_isize, d_req, n_regions, d_range, d_status);
  STRL_HIP(hipGetLastError());
  STRL_HIP(hipMemcpyAsync(&err, m_err, 4, hipMemcpyDeviceToHost, st));
  STRL_HIP(hipMemcpyAsync(range.data(), d_range, (size_t)n_regions * sizeof(RegionWalk), hipMemcpyDeviceToHost, st));
  STRL_HIP(hipMemcpyAsync(status, d_status, n_regions, hipMemcpyDeviceToHost, st));
  STRL_HIP(hipStreamSynchronize(st));
  if (err & IW_ERR_CRC) { set_error("a BGZF block inflates, but not to the bytes its CRC-32 names"); return STRL_ERR_CRC; }
  if (err) { set_error("device inflate: %s", (err & IW_ERR_DATA) ? "invalid DEFLATE data" : "inflated size differs from the block's ISIZE"); return STRL_ERR_FORMAT; }
  // where each region's bytes go: back to back, every piece starting at its source's offset modulo 16 (16-byte copies)
  uint64_t at = 0;
  for (uint32_t r = 0; r < n_regions; ++r) {
    const uint64_t len = range[r].stop - range[r].start;
    if (len) at = ((at + 15) & ~(uint64_t)15) + (range[r].start & 15u);
    out_off[r] = at;
    out_len[r] = len;
    at += len;
  }
  if (at > out_cap) {            // (out_off[0] = the bytes needed: a caller that sized `out` by a guess asks again with that much)
    out_off[0] = at;
    set_error("region records: %llu bytes, room for %llu", (unsigned long long)at, (unsigned long long)out_cap);
    return STRL_ERR_CAPACITY;
  }
  if (at) {
    if ((rc = d_out.reserve(at + 64))) return rc;
    STRL_HIP(hipMemcpyAsync(d_off, out_off, (size_t)n_regions * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(region_copy_kernel, dim3(std::min<uint32_t>(n_regions, 4096u)), dim3(256), 0, st, d_u.as<uint8_t>(), d_range, d_off, n_regions, d_out.as<uint8_t>());
    STRL_HIP(hipGetLastError());
    STRL_HIP(hipMemcpyAsync(out, d_out.p, at, hipMemcpyDeviceToHost, st));
    STRL_HIP(hipStreamSynchronize(st));
  }
  return STRL_OK;
}
