// What the kernel takes to reclaim a process that ends with device memory and page-locked host memory in hand (the 0.05 - 0.35 s
// the caller's clock adds behind `strling`'s last line).  usage: exit_probe <device GB> <page-locked MB> [unpin]   -- allocates
// (the device memory in 8 pieces, written once; the host memory from a huge-page mapping, touched and registered, as
// strl_pinned_alloc does), runs a kernel, prints its clock and ends with _exit(0); the caller times the whole process.
// "unpin": hipHostUnregister before the exit, timed.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint32_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
int main(int argc, char **argv) {
  const double t0 = now();
  const double dev_gb = argc > 1 ? atof(argv[1]) : 0;
  const size_t pin_mb = argc > 2 ? (size_t)atoi(argv[2]) : 0;
  const bool unpin = argc > 3 && !strcmp(argv[3], "unpin");
  hipFree(nullptr);
  const double t_ctx = now();
  std::vector<void *> dev;
  for (int k = 0; k < 8 && dev_gb > 0; ++k) {
    void *p = nullptr;
    const size_t bytes = (size_t)(dev_gb / 8 * (1ull << 30));
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, static_cast<uint32_t *>(p), bytes / 4);
    dev.push_back(p);
  }
  hipDeviceSynchronize();
  const double t_dev = now();
  char *a = nullptr;
  const size_t bytes = pin_mb << 20;
  if (bytes) {
    void *q = mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    a = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(q) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    madvise(a, bytes, MADV_HUGEPAGE);
    std::vector<std::thread> th;
    for (int k = 0; k < 8; ++k) th.emplace_back([=] { for (size_t o = bytes / 8 * k; o < bytes / 8 * (k + 1); o += 4096) a[o] = 0; });
    for (auto &x : th) x.join();
    if (hipHostRegister(a, bytes, hipHostRegisterPortable) != hipSuccess) { printf("hipHostRegister failed\n"); return 1; }
  }
  const double t_pin = now();
  double t_unpin = 0;
  int unpin_rc = 0;
  if (unpin && a) { unpin_rc = (int)hipHostUnregister(a); t_unpin = now() - t_pin; }
  printf("context %.3f s, %.1f GB of device memory written %.3f s, %zu MB page-locked %.3f s%s; in the process %.3f s\n", t_ctx - t0, dev_gb, t_dev - t_ctx, pin_mb, t_pin - t_dev,
         unpin ? (std::string(", unregistered in ") + std::to_string(t_unpin) + " s (status " + std::to_string(unpin_rc) + ")").c_str() : "", now() - t0);
  fflush(stdout);
  _exit(0);
}
