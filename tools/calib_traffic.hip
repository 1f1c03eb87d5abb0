// calib_traffic.hip -- known-byte-count kernels with classify_kernel's access shapes, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 for exactly those shapes (MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide
// coalesced read; other widths are uncalibrated).  Build + run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/calib tools/calib_traffic.hip && rocprofv3 --pmc FETCH_SIZE ... -- /tmp/calib
// Every kernel prints the bytes it must move; tools/summarise_calibration.py divides the counters by them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) classify's coordinate streams: 16-byte loads, 4 consecutive reads per lane
__global__ void calib_stream16(const int4 *a, uint32_t n4, int *sink) {
  int acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) { const int4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x7fffffff) *sink = acc;
}
// (b) classify's cig stream: 4-byte loads (4 reads per lane)
__global__ void calib_stream4(const uint32_t *a, uint32_t n, int *sink) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc ^= a[i];
  if (acc == 0x7fffffffu) *sink = (int)acc;
}
// (c) the flush gathers: 2-byte / 1-byte / 4-byte elements at the indices of the queued reads (ascending, ~8.5 % density)
template <typename T> __global__ void calib_gather(const T *a, const uint32_t *idx, uint32_t m, int *sink) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) acc += (uint32_t)a[idx[i]];
  if (acc == 0x7fffffffu) *sink = (int)acc;
}
// (d) the result words: 16-byte stores
__global__ void calib_store16(uint4 *a, uint32_t n4) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) a[i] = make_uint4(i, 0, 0, 0);
}
// (e) the queue entries: 16-byte stores at consecutive slots (same as d, kept separate for the table)
// (f) pair_probe's stream: 8-byte loads
__global__ void calib_stream8(const uint64_t *a, uint32_t n, int *sink) {
  uint64_t acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc ^= a[i];
  if (acc == 0x7fffffffull) *sink = 1;
}

int main() {
  const uint32_t n = 1u << 25;                 // reads
  void *buf = nullptr, *sink = nullptr;
  uint32_t *idx = nullptr;
  CK(hipMalloc(&buf, (size_t)n * 16));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, (size_t)n * 16));
  std::vector<uint32_t> h;
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; if ((s & 0xffff) < 5570) h.push_back(i); }   // 8.5 %
  const uint32_t m = (uint32_t)h.size();
  CK(hipMalloc(&idx, (size_t)m * 4));
  CK(hipMemcpy(idx, h.data(), (size_t)m * 4, hipMemcpyHostToDevice));
  const dim3 g(2048), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_stream16, g, b, 0, 0, (const int4 *)buf, n / 4 * 3, (int *)sink);     // tid, pos, end: 12 B per read
    hipLaunchKernelGGL(calib_stream4, g, b, 0, 0, (const uint32_t *)buf, n / 4, (int *)sink);      // cig: 1 B per read
    hipLaunchKernelGGL(calib_stream8, g, b, 0, 0, (const uint64_t *)buf, n, (int *)sink);          // qhash: 8 B per read
    hipLaunchKernelGGL(calib_gather<uint16_t>, g, b, 0, 0, (const uint16_t *)buf, idx, m, (int *)sink);
    hipLaunchKernelGGL(calib_gather<uint8_t>, g, b, 0, 0, (const uint8_t *)buf, idx, m, (int *)sink);
    hipLaunchKernelGGL(calib_gather<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, idx, m, (int *)sink);
    hipLaunchKernelGGL(calib_store16, g, b, 0, 0, (uint4 *)buf, n / 4);                            // 4 B per read
  }
  CK(hipDeviceSynchronize());
  // bytes per launch: what the kernel must read / write; for the gathers both the element bytes and the bytes of the
  // 64-byte lines they touch (what a cache-line-granular memory system has to move)
  auto lines = [&](uint32_t esz) { uint64_t c = 0, last = ~0ull; for (uint32_t i : h) { const uint64_t l = (uint64_t)i * esz / 64; if (l != last) { ++c; last = l; } } return c * 64; };
  printf("{\"n\": %u, \"m\": %u, \"bytes\": {\"calib_stream16\": %llu, \"calib_stream4\": %llu, \"calib_stream8\": %llu, "
         "\"calib_gather<unsigned short>\": {\"elements\": %llu, \"lines64\": %llu, \"index\": %llu}, "
         "\"calib_gather<unsigned char>\": {\"elements\": %llu, \"lines64\": %llu, \"index\": %llu}, "
         "\"calib_gather<unsigned int>\": {\"elements\": %llu, \"lines64\": %llu, \"index\": %llu}, \"calib_store16\": %llu}}\n",
         n, m, (unsigned long long)n * 12, (unsigned long long)n, (unsigned long long)n * 8,
         (unsigned long long)m * 2, (unsigned long long)lines(2), (unsigned long long)m * 4,
         (unsigned long long)m, (unsigned long long)lines(1), (unsigned long long)m * 4,
         (unsigned long long)m * 4, (unsigned long long)lines(4), (unsigned long long)m * 4, (unsigned long long)n * 4);
  return 0;
}
