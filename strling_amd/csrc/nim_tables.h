// nim_tables.h -- the slice of Nim 1.6 stdlib behaviour that leaks into STRling's output:
//   * hashes.nim: hashWangYi1 (integers), murmurHash (char arrays), `!&` / `!$` (tuples)
//   * tables.nim: Table slot order under linear probing + enlarge (order of the (tid, repeat)
//     groups in -bounds.txt: call.nim:223, merge.nim:172) and CountTable[uint32].largest tie order
//     (modal clip position: cluster.nim:204-211,300-301).
// Host+device inline so the clustering kernels can reproduce `largest` on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NIM_HD __host__ __device__ inline
#else
#define NIM_HD inline
#endif

namespace nim {

NIM_HD uint64_t hi_xor_lo(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b) ^ (a * b);
#else
  const __uint128_t r = (__uint128_t)a * b;
  return (uint64_t)(r >> 64) ^ (uint64_t)r;
#endif
}
// hashWangYi1, lib/pure/hashes.nim
NIM_HD uint64_t hash_int(uint64_t x) {
  const uint64_t P0 = 0xa0761d6478bd642full, P1 = 0xe7037ed1a0b428dbull, P58 = 0xeb44accab455d165ull ^ 8ull;
  return hi_xor_lo(hi_xor_lo(P0, x ^ P1), P58);
}
NIM_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
// murmurHash (MurmurHash3_x86_32, seed 0), lib/pure/hashes.nim
NIM_HD uint64_t hash_bytes(const uint8_t *x, int size) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h1 = 0;
  int i = 0;
  for (; i + 4 <= size; i += 4) {
    uint32_t k1 = (uint32_t)x[i] | ((uint32_t)x[i + 1] << 8) | ((uint32_t)x[i + 2] << 16) | ((uint32_t)x[i + 3] << 24);
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
    h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5u + 0xe6546b64u;
  }
  uint32_t k1 = 0;
  for (int r = size - i; r > 0; --r) k1 = (k1 << 8) | x[i + r - 1];
  k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
  h1 ^= k1;
  h1 ^= (uint32_t)size;
  h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
  return h1;
}
NIM_HD uint64_t mix(uint64_t h, uint64_t v) { uint64_t r = h + v; r += r << 10; r ^= r >> 6; return r; }   // `!&`
NIM_HD uint64_t finish(uint64_t h) { uint64_t r = h + (h << 3); r ^= r >> 11; r += r << 15; return r; }     // `!$`
// hash(tuple[tid: int32, repeat: array[6, char]])  (cluster.nim:12)
NIM_HD uint64_t hash_tid_rep(int32_t tid, const char rep[6]) {
  uint64_t h = mix(0, hash_int((uint64_t)(int64_t)tid));
  h = mix(h, hash_bytes(reinterpret_cast<const uint8_t *>(rep), 6));
  return finish(h);
}
NIM_HD bool must_rehash(uint64_t len, uint64_t counter) { return (len * 2 < counter * 3) || (len - counter < 4); }
NIM_HD uint64_t slots_needed(uint64_t count) {
  uint64_t want = count * 3 / 2 + 4, p = 1;
  while (p < want) p <<= 1;
  return p;
}

// CountTable[uint32] as STRling uses it: keys arrive in ascending order (reads are position
// sorted), so only first insertions move slots; `val` of a key is its total count.
// Storage: two buffers of `cap` entries each (keys|vals), cap >= max(16, 3 * distinct).
struct CountTable {
  uint32_t *ka, *va, *kb, *vb;
  uint32_t len, counter;
  NIM_HD void init(uint32_t *scratch, uint32_t cap) {
    ka = scratch; va = scratch + cap; kb = scratch + 2 * (uint64_t)cap; vb = scratch + 3 * (uint64_t)cap;
    len = (uint32_t)slots_needed(8);   // initCountTable[uint32](8), cluster.nim:177-178,289-290
    counter = 0;
    for (uint32_t i = 0; i < len; ++i) va[i] = 0;
  }
  NIM_HD void insert_new(uint32_t key, uint32_t val) {
    if (must_rehash(len, counter)) {   // enlarge: reinsert in slot order into a table twice the size
      const uint32_t nl = len * 2;
      for (uint32_t i = 0; i < nl; ++i) vb[i] = 0;
      for (uint32_t i = 0; i < len; ++i)
        if (va[i]) {
          uint32_t j = (uint32_t)(hash_int(ka[i]) & (nl - 1));
          while (vb[j]) j = (j + 1) & (nl - 1);
          kb[j] = ka[i]; vb[j] = va[i];
        }
      uint32_t *t = ka; ka = kb; kb = t;
      t = va; va = vb; vb = t;
      len = nl;
    }
    uint32_t h = (uint32_t)(hash_int(key) & (len - 1));
    while (va[h]) h = (h + 1) & (len - 1);
    ka[h] = key; va[h] = val;
    ++counter;
  }
  NIM_HD void largest(uint32_t &key, uint32_t &val) const {   // first maximum in slot order
    uint32_t mi = 0;
    for (uint32_t h = 1; h < len; ++h)
      if (va[mi] < va[h]) mi = h;
    key = ka[mi]; val = va[mi];
  }
};

}  // namespace nim
