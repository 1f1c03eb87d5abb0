// Platform probe: does a chain of small kernels on stream B (fenced to stream A by events, like the pair logic + clustering
// of batch i) run beside big kernels on stream A (the scorer of batch i + 1)?  Kernels stamp wall_clock64() at start / end.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ovp tools/overlap_probe.hip && /tmp/ovp [memset]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
__global__ void big(unsigned long long *t, int slot, int iters, float *sink) {
  if (blockIdx.x == 0 && threadIdx.x == 0) t[2 * slot] = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) sink[0] = a;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) t[2 * slot + 1] = wall_clock64();
}
__global__ void small(unsigned long long *t, int slot, float *sink) {
  if (blockIdx.x == 0 && threadIdx.x == 0) t[2 * slot] = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) sink[0] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) t[2 * slot + 1] = wall_clock64();
}
int main(int argc, char **argv) {
  const bool with_memset = argc > 1 && strstr(argv[1], "memset");
  const bool nowait = argc > 1 && strstr(argv[1], "nowait");       // stream A never waits for stream B
  const bool split = argc > 1 && strstr(argv[1], "split");         // the chain in two parts with a second A->B fence between them
  hipEvent_t mid;
  CK(hipEventCreateWithFlags(&mid, hipEventDisableTiming));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  if (argc > 1 && strstr(argv[1], "prio")) {
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("priority range: least %d greatest %d\n", lo, hi);
    CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
  } else CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int STEPS = 6, NS = 40;
  unsigned long long *t;
  float *sink;
  uint32_t *buf[2];
  CK(hipMalloc(&t, 2 * 8 * STEPS * (NS + 1)));
  CK(hipMalloc(&sink, 4));
  for (auto &b : buf) CK(hipMalloc(&b, 1 << 22));
  hipEvent_t head[STEPS], freeb[STEPS];
  for (auto &e : head) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : freeb) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int s = 0; s < STEPS; ++s) {
    if (s >= 2 && !nowait) CK(hipStreamWaitEvent(A, freeb[s - 2], 0));
    if (with_memset) CK(hipMemsetAsync(buf[s & 1], 0, 1 << 22, A));
    hipLaunchKernelGGL(big, dim3(8192), dim3(256), 0, A, t, s * (NS + 1), 6000, sink);
    CK(hipEventRecord(head[s], A));
    CK(hipStreamWaitEvent(B, head[s], 0));
    for (int k = 0; k < NS; ++k) {
      if (split && k == NS / 2) { CK(hipEventRecord(freeb[s], B)); CK(hipEventRecord(mid, A)); CK(hipStreamWaitEvent(B, mid, 0)); }
      hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, B, t, s * (NS + 1) + 1 + k, sink);
    }
    if (!split) CK(hipEventRecord(freeb[s], B));
  }
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * STEPS * (NS + 1));
  CK(hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost));
  const double tick = 1e6 / 100e6;   // wall_clock64: 100 MHz
  const unsigned long long t0 = h[0];
  for (int s = 0; s < STEPS; ++s) {
    const int b = s * (NS + 1);
    printf("step %d: big %.1f .. %.1f us | small chain %.1f .. %.1f us\n", s, (h[2 * b] - t0) * tick, (h[2 * b + 1] - t0) * tick, (h[2 * (b + 1)] - t0) * tick,
           (h[2 * (b + NS) + 1] - t0) * tick);
  }
  return 0;
}
