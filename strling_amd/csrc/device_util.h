// device_util.h -- small device-side helpers shared by the .hip translation units
#pragma once
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

}  // namespace strl
