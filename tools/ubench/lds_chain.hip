// Micro-benchmark behind the inflate design: how long does ONE dependent table lookup take a wave, alone and
// with the CU full of waves doing the same?  (a) ds_read_b32 pointer chase, (b) the same through v_readlane from a lane table,
// (c) the chase with a compare + taken branch per step (the symbol loop's shape).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_chain tools/ubench/lds_chain.hip && /tmp/lds_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE> __global__ __launch_bounds__(64) void chase(uint32_t *out, uint64_t *cyc, int iters, int lds_pad) {
  extern __shared__ uint32_t tab[];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 2654435761u) >> 7;
  __syncthreads();
  uint32_t e = blockIdx.x * 7u + 1u, acc = 0;
  uint32_t lanetab = (threadIdx.x * 40503u) ^ 0x5bd1e995u;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      e = tab[e & 1023u];
    } else if (MODE == 1) {
      const uint32_t k = __builtin_amdgcn_readfirstlane(e & 63u);
      e = (uint32_t)__builtin_amdgcn_readlane((int)lanetab, (int)k) + (e >> 6);
    } else {
      e = tab[e & 1023u];
      if (e & 0x800u) { acc += e >> 3; asm volatile("" ::: "memory"); } else { acc ^= e; asm volatile("" ::: "memory"); }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = e + acc; cyc[blockIdx.x] = t1 - t0; }
}
template <int MODE> static void run(const char *name, int blocks, int lds_bytes) {
  uint32_t *out; uint64_t *cyc;
  hipMalloc(&out, blocks * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(64), lds_bytes, 0, out, cyc, iters, 0);
  hipEventRecord(a);
  hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(64), lds_bytes, 0, out, cyc, iters, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("%-28s blocks %6d lds/wave %6d B: %.1f counter ticks per step per wave (s_memtime, 100 MHz), kernel %.3f ms -> %.1f ns per step per wave\n", name, blocks, lds_bytes,
         s / blocks / iters, ms, ms * 1e6 / iters / ((blocks + 256 * (160 * 1024 / lds_bytes > 32 ? 32 : 160 * 1024 / lds_bytes) - 1) / (256 * (160 * 1024 / lds_bytes > 32 ? 32 : 160 * 1024 / lds_bytes))));
}
int main() {
  for (int lds : {65536, 20480, 6656, 5000}) {
    const int per_cu = 160 * 1024 / lds > 32 ? 32 : 160 * 1024 / lds;
    run<0>("ds_read chase", 256 * per_cu, lds);
    run<2>("ds_read chase + branch", 256 * per_cu, lds);
    run<1>("readlane chase", 256 * per_cu, lds);
  }
  return 0;
}
