// sort.h -- hand-written stable LSD radix sort of (u64 key, u32 value) pairs for gfx950 (sort.hip).
//
// Used for the (tid, unit) << 32 | position keys of the clustering path (call.nim:118-130, merge.nim:121-135:
// group + stable sort by position) and for the qname-hash join / emission order of the pair logic
// (extract.nim:192-248).  The element count lives on the DEVICE (*d_n, bounded by n_max on the host), so a whole
// pipeline can be enqueued without a host round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace strl {

constexpr uint32_t SORT_THREADS = 256;                       // 4 waves per tile
constexpr uint32_t SORT_KPT = 8;                             // keys per thread
constexpr uint32_t SORT_TILE = SORT_THREADS * SORT_KPT;      // 2048 keys per tile
constexpr uint32_t SORT_CHUNK = 32;                          // tiles per chunk of the two-level histogram tables

// bytes of scratch radix_sort_pairs needs for n_max elements and `bits` key bits
size_t radix_sort_scratch_bytes(uint32_t n_max, int bits);

// Stable sort of (keys, vals)[0, *d_n) by key bits [bit_lo, bit_lo + bits).  Ping-pongs between (keys, vals) and
// (keys_alt, vals_alt); *out_keys / *out_vals receive the buffers that hold the result.  Everything is enqueued on
// `st`; nothing synchronises.  bits == 0: nothing to do (result = input).  Returns a hipError_t as int (0 = ok).
int radix_sort_pairs(hipStream_t st, const uint32_t *d_n, uint32_t n_max, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt,
                     uint32_t *vals_alt, void *scratch, size_t scratch_bytes, int bit_lo, int bits, uint64_t **out_keys,
                     uint32_t **out_vals, bool tables_zeroed = false);

// The part of `scratch` a sort accumulates into with atomics and therefore needs zeroed first (its chunk tables).  A caller
// that zeroes other things on the same stream anyway can include this region in its own fill (one launch instead of two)
// and pass tables_zeroed = true.
void radix_sort_tables(void *scratch, uint32_t n_max, int bits, void **p, size_t *bytes);

}  // namespace strl
