// What does device memory cost to obtain, and can it be obtained beside running kernels?  hipMalloc of N GB; the same through
// the virtual-memory API (reserve once, map in 1 GB steps); both from a second thread while the first keeps launching kernels.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/vmm_probe tools/ubench/vmm_probe.hip -lpthread && /tmp/vmm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(uint32_t *p, int n) { uint32_t v = threadIdx.x; for (int i = 0; i < n; ++i) v = v * 1664525u + 1013904223u; if (v == 42) p[0] = v; }
int main() {
  int vmm = 0;
  CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
  printf("virtual memory management supported: %d\n", vmm);
  const size_t GB = 1ull << 30, N = 16;
  { double t = now(); void *p; CK(hipMalloc(&p, N * GB)); double t1 = now(); CK(hipMemset(p, 0, N * GB)); CK(hipDeviceSynchronize()); double t2 = now(); CK(hipFree(p));
    printf("hipMalloc %zu GB: %.3f s (%.1f GB/s), memset %.3f s, free %.3f s\n", N, t1 - t, N / (t1 - t), t2 - t1, now() - t2); }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu\n", gran);
  void *va = nullptr;
  double t = now();
  CK(hipMemAddressReserve(&va, 256 * GB, 0, nullptr, 0));
  printf("reserve 256 GB of addresses: %.4f s\n", now() - t);
  hipMemAccessDesc acc = {};
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> hs;
  t = now();
  for (size_t k = 0; k < N; ++k) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, GB, &prop, 0));
    CK(hipMemMap((char *)va + k * GB, GB, 0, h, 0));
    CK(hipMemSetAccess((char *)va + k * GB, GB, &acc, 1));
    hs.push_back(h);
  }
  printf("create + map + access %zu x 1 GB: %.3f s (%.1f GB/s)\n", N, now() - t, N / (now() - t));
  CK(hipMemset(va, 1, N * GB)); CK(hipDeviceSynchronize());
  // beside kernels: a thread launches 20 us kernels back to back and counts them; the main thread maps 16 more GB
  uint32_t *d; CK(hipMalloc(&d, 64));
  std::atomic<bool> stop{false};
  std::atomic<long> launched{0};
  std::thread th([&] { hipStream_t s; hipStreamCreate(&s); while (!stop) { hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, s, d, 2000); hipStreamSynchronize(s); ++launched; } });
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  long l0 = launched; double t0 = now();
  std::this_thread::sleep_for(std::chrono::milliseconds(500));
  printf("kernels alone: %.0f per s\n", (launched - l0) / (now() - t0));
  l0 = launched; t0 = now();
  for (size_t k = N; k < 2 * N; ++k) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, GB, &prop, 0)); CK(hipMemMap((char *)va + k * GB, GB, 0, h, 0)); CK(hipMemSetAccess((char *)va + k * GB, GB, &acc, 1));
    hs.push_back(h);
  }
  printf("mapping 16 GB beside them: %.3f s, kernels %.0f per s\n", now() - t0, (launched - l0) / (now() - t0));
  l0 = launched; t0 = now();
  { void *p; CK(hipMalloc(&p, N * GB)); printf("hipMalloc 16 GB beside them: %.3f s, kernels %.0f per s\n", now() - t0, (launched - l0) / (now() - t0)); CK(hipFree(p)); }
  stop = true; th.join();
  return 0;
}
