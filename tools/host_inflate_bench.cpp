// Single-thread inflate throughput on the BGZF blocks of a BAM file: zlib vs cli/fast_inflate.cpp (and a byte-wise check).
//   g++ -O3 -std=c++17 -Istrling_amd/csrc/cli tools/host_inflate_bench.cpp strling_amd/csrc/cli/fast_inflate.cpp -lz -o /tmp/hib && /tmp/hib x.bam
#include <zlib.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "fast_inflate.h"
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  std::vector<uint8_t> file;
  uint8_t buf[1 << 16];
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + k);
  fclose(f);
  file.resize(file.size() + 16);
  struct B { size_t c; uint32_t clen, isize; size_t out; };
  std::vector<B> blks;
  size_t p = 0, tot = 0;
  const size_t n = file.size() - 16;
  while (p + 18 <= n) {
    const uint32_t xlen = file[p + 10] | (file[p + 11] << 8), bsize = (file[p + 16] | (file[p + 17] << 8)) + 1u;
    const uint32_t clen = bsize - 12 - xlen - 8;
    const uint8_t *t = &file[p + bsize - 4];
    const uint32_t isz = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
    blks.push_back({p + 12 + xlen, clen, isz, tot});
    tot += isz;
    p += bsize;
    if (blks.size() >= 4000) break;
  }
  std::vector<uint8_t> a(tot + 16), b(tot + 16);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  for (auto &x : blks) {
    z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
    zs.next_in = &file[x.c]; zs.avail_in = x.clen; zs.next_out = &a[x.out]; zs.avail_out = x.isize;
    inflate(&zs, Z_FINISH); inflateEnd(&zs);
  }
  auto t1 = now();
  int bad = 0;
  for (auto &x : blks) bad += strl::fast_inflate(&file[x.c], x.clen, &b[x.out], x.isize);
  auto t2 = now();
  const double tz = std::chrono::duration<double>(t1 - t0).count(), tf = std::chrono::duration<double>(t2 - t1).count();
  printf("%zu blocks, %.1f MB inflated (ratio %.2f): zlib %.3f GB/s, fast %.3f GB/s (x%.2f), %d blocks refused, identical: %s\n", blks.size(), tot / 1e6,
         (double)tot / p, tot / tz / 1e9, tot / tf / 1e9, tz / tf, bad, memcmp(a.data(), b.data(), tot) == 0 ? "yes" : "NO");
  return 0;
}
