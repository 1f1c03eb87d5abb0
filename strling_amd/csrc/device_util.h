// device_util.h -- small device-side helpers shared by the .hip translation units
#pragma once
#include <algorithm>
#include "common.h"

namespace strl {

// Bloom bitmap over mixed qname hashes (pair.hip): two bits per key
__device__ __forceinline__ void bloom_set(uint32_t *bloom, uint32_t mask, uint64_t m) {
  const uint32_t b0 = (uint32_t)m & mask, b1 = (uint32_t)(m >> 32) & mask;
  atomicOr(&bloom[b0 >> 5], 1u << (b0 & 31u));
  atomicOr(&bloom[b1 >> 5], 1u << (b1 & 31u));
}
__device__ __forceinline__ bool bloom_test(const uint32_t *bloom, uint32_t mask, uint64_t m) {
  const uint32_t b0 = (uint32_t)m & mask;
  if (!((bloom[b0 >> 5] >> (b0 & 31u)) & 1u)) return false;
  const uint32_t b1 = (uint32_t)(m >> 32) & mask;
  return (bloom[b1 >> 5] >> (b1 & 31u)) & 1u;
}


// Zero-fill as a kernel of our own instead of hipMemsetAsync: the runtime's fill goes through its blit path (a kernel
// launch plus bookkeeping on the runtime's side per call); here it is one plain launch on the caller's stream.
static __global__ void zero_words_kernel(uint32_t *p, size_t n_words) {
  const size_t n4 = n_words / 4, stride = (size_t)gridDim.x * blockDim.x;
  uint4 *q = reinterpret_cast<uint4 *>(p);                 // (hipMalloc'd buffers: 16-byte aligned)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) q[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[n4 * 4 + threadIdx.x] = 0;
}
// two regions in one launch (counters + a sort's chunk tables): under a second stream that saturates the chip every small
// launch of a dependent chain costs tens of microseconds, so launches are worth saving
static __global__ void zero_words2_kernel(uint32_t *p, size_t n_words, uint32_t *q, size_t m_words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = i0; i < n_words; i += stride) p[i] = 0;
  for (size_t i = i0; i < m_words; i += stride) q[i] = 0;
}
static inline hipError_t zero_words2(void *p, size_t bytes, void *q, size_t qbytes, hipStream_t st) {
  const size_t w = std::max(bytes, qbytes) / 4;
  if (!w) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<size_t>((w + 255) / 256, 1024);
  hipLaunchKernelGGL(zero_words2_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint32_t *>(p), bytes / 4, static_cast<uint32_t *>(q), qbytes / 4);
  return hipGetLastError();
}
static inline hipError_t zero_words(void *p, size_t bytes, hipStream_t st) {   // p 16-byte aligned, bytes a multiple of 4
  const size_t words = bytes / 4;
  if (!words) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<size_t>((words / 4 + 255) / 256 + 1, 1024);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint32_t *>(p), words);
  return hipGetLastError();
}
}  // namespace strl
