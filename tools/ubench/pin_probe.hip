// How long page-locked host memory takes to get: hipHostMalloc against an aligned allocation with MADV_HUGEPAGE, touched, then
// hipHostRegister.  usage: pin_probe [MB] [buffers]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 328, n = argc > 2 ? atoi(argv[2]) : 4, bytes = mb << 20;
  hipFree(nullptr);
  std::vector<void *> p(n);
  double t = now();
  for (size_t i = 0; i < n; ++i) hipHostMalloc(&p[i], bytes, hipHostMallocDefault);
  printf("hipHostMalloc %zu x %zu MB sequential: %.3f s\n", n, mb, now() - t);
  for (void *q : p) hipHostFree(q);
  t = now();
  { std::vector<std::thread> th; for (size_t i = 0; i < n; ++i) th.emplace_back([&, i] { hipHostMalloc(&p[i], bytes, hipHostMallocDefault); }); for (auto &x : th) x.join(); }
  printf("hipHostMalloc %zu x %zu MB on %zu threads: %.3f s\n", n, mb, n, now() - t);
  for (void *q : p) hipHostFree(q);
  for (int huge = 0; huge < 2; ++huge) {
    t = now();
    double t_touch = 0, t_reg = 0;
    for (size_t i = 0; i < n; ++i) {
      void *q = mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      char *a = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(q) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
      if (huge) madvise(a, bytes, MADV_HUGEPAGE);
      double t0 = now();
      const int T = 8;
      std::vector<std::thread> th;
      for (int k = 0; k < T; ++k) th.emplace_back([&, k] { for (size_t o = bytes / T * k; o < bytes / T * (k + 1); o += 4096) a[o] = 0; });
      for (auto &x : th) x.join();
      t_touch += now() - t0; t0 = now();
      hipError_t e = hipHostRegister(a, bytes, hipHostRegisterDefault);
      t_reg += now() - t0;
      if (e != hipSuccess) printf("register failed: %s\n", hipGetErrorString(e));
      p[i] = a;
    }
    printf("%s + touch (8 threads) + hipHostRegister, %zu x %zu MB: %.3f s (touch %.3f, register %.3f)\n", huge ? "MADV_HUGEPAGE" : "4 KB pages", n, mb, now() - t, t_touch, t_reg);
    // a copy from it, to see it behaves like page-locked memory
    void *d; hipMalloc(&d, bytes);
    hipStream_t st; hipStreamCreate(&st);
    double t0 = now();
    hipMemcpyAsync(d, p[0], bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
    hipMemcpyAsync(d, p[0], bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
    printf("   2 copies to the device: %.1f GB/s\n", 2.0 * bytes / (now() - t0) / 1e9);
    hipFree(d);
    for (void *q : p) hipHostUnregister(q);
  }
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP: %s", b);
  return 0;
}
