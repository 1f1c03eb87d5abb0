// How the compressed bytes of a BAM can reach the device: the shipped feed `pread`s page-cache bytes
// into page-locked rings (one CPU copy of every byte, ~6 GB/s per granted CPU) and DMAs from there.  This probe times the
// alternatives on a real file, chunk by chunk like the feed:
//   pread     T threads pread() a chunk into a page-locked buffer, then one hipMemcpyAsync (the shipped way)
//   register  the file mmap()ed read-only; every chunk of the mapping is hipHostRegister()ed (by a helper thread, one chunk
//             ahead of the copy), DMAed straight from the page cache, unregistered -- no CPU copy
//   mmapcopy  like pread, but the copy is a user-space memcpy from the file's mapping
//   pageable  hipMemcpyAsync straight from the unregistered mapping (the runtime stages it itself)
//   direct    O_DIRECT reads into the page-locked buffer (a file that is NOT in the page cache), then the DMA
// usage: feed_probe FILE [chunk_MB=320] [max_GB=16] [threads=12] [modes=pread,register,pageable,direct]
// hipcc --offload-arch=gfx950 -O3 -o /tmp/feed_probe tools/ubench/feed_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); (void)hipGetLastError(); failed = true; } } while (0)

static void *pinned(size_t bytes) {
  const size_t huge = (size_t)2 << 20, len = ((bytes + huge - 1) & ~(huge - 1)) + huge;
  void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  char *a = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(base) + huge - 1) & ~(uintptr_t)(huge - 1));
  madvise(a, len - huge, MADV_HUGEPAGE);
  for (size_t o = 0; o < len - huge; o += 4096) a[o] = 0;
  if (hipHostRegister(a, len - huge, hipHostRegisterPortable) != hipSuccess) { fprintf(stderr, "hipHostRegister of an anonymous buffer failed\n"); exit(2); }
  return a;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: feed_probe FILE [chunk_MB] [max_GB] [threads] [modes]\n"); return 1; }
  const char *path = argv[1];
  const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 320) << 20;
  const double max_gb = argc > 3 ? atof(argv[3]) : 16.0;
  const int T = argc > 4 ? atoi(argv[4]) : 12;
  const std::string modes = argc > 5 ? argv[5] : "pread,mmapcopy,register,pageable,direct";
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { perror("open"); return 1; }
  struct stat st;
  fstat(fd, &st);
  const size_t total = std::min<size_t>((size_t)st.st_size, (size_t)(max_gb * 1e9)) / chunk * chunk;
  const size_t n_chunks = total / chunk;
  if (!n_chunks) { fprintf(stderr, "file smaller than one chunk\n"); return 1; }
  bool failed = false;
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t *dev[2];
  CK(hipMalloc((void **)&dev[0], chunk));
  CK(hipMalloc((void **)&dev[1], chunk));
  printf("feed_probe: %s, %.2f GB in %zu chunks of %zu MB, %d copy threads\n", path, total / 1e9, n_chunks, chunk >> 20, T);

  if (modes.find("pread") != std::string::npos) {
    uint8_t *ring[2] = {(uint8_t *)pinned(chunk), (uint8_t *)pinned(chunk)};
    hipEvent_t ev[2];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double t_read = 0;
    const double t0 = now();
    for (size_t c = 0; c < n_chunks; ++c) {
      if (c >= 2) CK(hipEventSynchronize(ev[c & 1]));
      const double a = now();
      std::vector<std::thread> th;
      const size_t piece = (size_t)4 << 20, pieces = chunk / piece;
      std::atomic<size_t> next{0};
      for (int t = 0; t < T; ++t) th.emplace_back([&] { for (size_t k; (k = next.fetch_add(1)) < pieces;) if (pread(fd, ring[c & 1] + k * piece, piece, (off_t)(c * chunk + k * piece)) != (ssize_t)piece) abort(); });
      for (auto &x : th) x.join();
      t_read += now() - a;
      CK(hipMemcpyAsync(dev[c & 1], ring[c & 1], chunk, hipMemcpyHostToDevice, s));
      CK(hipEventRecord(ev[c & 1], s));
    }
    CK(hipStreamSynchronize(s));
    const double dt = now() - t0;
    printf("pread    : %.3f s = %.1f GB/s to the device (pread alone %.3f s = %.1f GB/s on %d threads)\n", dt, total / 1e9 / dt, t_read, total / 1e9 / t_read, T);
  }

  for (int pass = 0; pass < 3 && modes.find("register") != std::string::npos; ++pass) {
    // pass 0: hipHostRegisterDefault, 1: Portable | ReadOnly (0x08), 2: the registration on R helper threads, several chunks ahead
    const unsigned flags = pass == 0 ? hipHostRegisterDefault : (hipHostRegisterPortable | 0x08u);
    const int R = pass == 2 ? 4 : 1;
    void *map = mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); break; }
    uint8_t *m = (uint8_t *)map;
    failed = false;
    std::vector<double> t_reg(n_chunks, 0), t_unreg(n_chunks, 0);
    std::vector<std::atomic<int>> ready(n_chunks);
    for (auto &r : ready) r = 0;
    std::atomic<size_t> next{0}, copied{0};
    std::atomic<bool> bad{false};
    const size_t ahead = (size_t)R + 1;
    const double t0 = now();
    std::vector<std::thread> reg;
    for (int r = 0; r < R; ++r)
      reg.emplace_back([&] {
        (void)hipSetDevice(0);
        for (size_t c; (c = next.fetch_add(1)) < n_chunks;) {
          while (c >= copied.load() + ahead && !bad) std::this_thread::yield();
          const double a = now();
          const hipError_t e = hipHostRegister(m + c * chunk, chunk, flags);
          t_reg[c] = now() - a;
          if (e != hipSuccess) { fprintf(stderr, "hipHostRegister(mapping of the file, flags %#x): %s\n", flags, hipGetErrorString(e)); (void)hipGetLastError(); bad = true; ready[c] = -1; return; }
          ready[c] = 1;
        }
      });
    double t_copy_wait = 0;
    for (size_t c = 0; c < n_chunks && !bad; ++c) {
      while (!ready[c].load()) std::this_thread::yield();
      if (ready[c] < 0) break;
      const double a = now();
      CK(hipMemcpyAsync(dev[c & 1], m + c * chunk, chunk, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      t_copy_wait += now() - a;
      const double b = now();
      CK(hipHostUnregister(m + c * chunk));
      t_unreg[c] = now() - b;
      copied = c + 1;
    }
    bad = bad.load() || failed;
    copied = n_chunks;
    for (auto &x : reg) x.join();
    const double dt = now() - t0;
    double sr = 0, su = 0;
    for (size_t c = 0; c < n_chunks; ++c) { sr += t_reg[c]; su += t_unreg[c]; }
    if (!bad) printf("register : flags %#x, %d registering thread(s): %.3f s = %.1f GB/s to the device (register %.3f s = %.1f GB/s per thread, copy + wait %.3f s = %.1f GB/s, unregister %.3f s)\n", flags, R, dt,
                     total / 1e9 / dt, sr, total / 1e9 / sr, t_copy_wait, total / 1e9 / t_copy_wait, su);
    else printf("register : flags %#x: not possible on this box (see stderr)\n", flags);
    munmap(map, total);
  }

  if (modes.find("mmapcopy") != std::string::npos) {
    // the CPU copy in user space: memcpy from the file's mapping into the page-locked ring (glibc streams large copies past
    // the cache), instead of the kernel's copy_to_user inside pread
    void *map = mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
    uint8_t *m = (uint8_t *)map;
    uint8_t *ring[2] = {(uint8_t *)pinned(chunk), (uint8_t *)pinned(chunk)};
    hipEvent_t ev[2];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    double t_read = 0;
    const double t0 = now();
    for (size_t c = 0; c < n_chunks; ++c) {
      if (c >= 2) CK(hipEventSynchronize(ev[c & 1]));
      const double a = now();
      std::vector<std::thread> th;
      const size_t piece = (size_t)4 << 20, pieces = chunk / piece;
      std::atomic<size_t> next{0};
      for (int t = 0; t < T; ++t) th.emplace_back([&] { for (size_t k; (k = next.fetch_add(1)) < pieces;) memcpy(ring[c & 1] + k * piece, m + c * chunk + k * piece, piece); });
      for (auto &x : th) x.join();
      t_read += now() - a;
      CK(hipMemcpyAsync(dev[c & 1], ring[c & 1], chunk, hipMemcpyHostToDevice, s));
      CK(hipEventRecord(ev[c & 1], s));
    }
    CK(hipStreamSynchronize(s));
    const double dt = now() - t0;
    printf("mmapcopy : %.3f s = %.1f GB/s to the device (memcpy from the mapping alone %.3f s = %.1f GB/s on %d threads)\n", dt, total / 1e9 / dt, t_read, total / 1e9 / t_read, T);
    munmap(map, total);
  }

  if (modes.find("pageable") != std::string::npos) {
    void *map = mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
    uint8_t *m = (uint8_t *)map;
    const size_t lim = std::min<size_t>(n_chunks, 12);
    const double t0 = now();
    for (size_t c = 0; c < lim; ++c) CK(hipMemcpyAsync(dev[c & 1], m + c * chunk, chunk, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    const double dt = now() - t0;
    printf("pageable : %.3f s = %.1f GB/s to the device (%zu chunks straight from the unregistered mapping)\n", dt, lim * chunk / 1e9 / dt, lim);
    munmap(map, total);
  }

  if (modes.find("direct") != std::string::npos) {
    const int fdd = open(path, O_RDONLY | O_DIRECT);
    if (fdd < 0) printf("direct   : O_DIRECT not possible on this file system (%s)\n", strerror(errno));
    else {
      uint8_t *ring[2] = {(uint8_t *)pinned(chunk), (uint8_t *)pinned(chunk)};
      hipEvent_t ev[2];
      for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      double t_read = 0;
      bool ok = true;
      const double t0 = now();
      for (size_t c = 0; c < n_chunks && ok; ++c) {
        if (c >= 2) CK(hipEventSynchronize(ev[c & 1]));
        const double a = now();
        std::vector<std::thread> th;
        const size_t piece = (size_t)4 << 20, pieces = chunk / piece;
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        for (int t = 0; t < T; ++t) th.emplace_back([&] { for (size_t k; (k = next.fetch_add(1)) < pieces;) if (pread(fdd, ring[c & 1] + k * piece, piece, (off_t)(c * chunk + k * piece)) != (ssize_t)piece) bad = true; });
        for (auto &x : th) x.join();
        if (bad) { printf("direct   : O_DIRECT read failed (%s)\n", strerror(errno)); ok = false; break; }
        t_read += now() - a;
        CK(hipMemcpyAsync(dev[c & 1], ring[c & 1], chunk, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[c & 1], s));
      }
      CK(hipStreamSynchronize(s));
      const double dt = now() - t0;
      if (ok) printf("direct   : %.3f s = %.1f GB/s to the device (O_DIRECT reads alone %.3f s = %.1f GB/s on %d threads)\n", dt, total / 1e9 / dt, t_read, total / 1e9 / t_read, T);
      close(fdd);
    }
  }
  return 0;
}
