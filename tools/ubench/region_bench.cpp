// region_bench.cpp -- where `strling call`'s evidence step spends its time on the host: BamReader::read_region
// (linear-index seek, BGZF inflate, record parse) against strl_spanners on the records, for bounds-sized regions at
// random places of an indexed BAM.  usage: region_bench file.bam [n_regions] [window]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include "../../strling_amd/csrc/cli/bam_reader.h"
#include "../../include/strling_amd.h"
using namespace strl;
int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const int n = argc > 2 ? atoi(argv[2]) : 500, window = argc > 3 ? atoi(argv[3]) : 700;
  BamReader rd;
  std::string err;
  if (!rd.open(argv[1], err) || !rd.load_index(argv[1], err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  std::mt19937_64 g(7);
  RecordBatch b;
  uint32_t frag[4096] = {0};
  for (int i = 300; i < 700; ++i) frag[i] = 100;
  std::vector<strl_support> sup;
  double t_read = 0, t_span = 0;
  uint64_t recs = 0, support = 0;
  for (int i = 0; i < n; ++i) {
    const int32_t tid = (int32_t)(g() % rd.targets().size());
    const int64_t len = rd.targets()[(size_t)tid].length;
    const int64_t left = (int64_t)(g() % (uint64_t)(len - 2000)) + 1000;
    strl_bounds bd{};
    bd.tid = tid; bd.left = (uint32_t)left; bd.right = (uint32_t)left + 40;
    b.clear();
    auto t0 = std::chrono::steady_clock::now();
    if (rd.read_region(b, tid, left - window, left + 40 + window, err) < 0) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    auto t1 = std::chrono::steady_clock::now();
    const strl_records rv = b.view();
    sup.resize(2 * b.size() + 16);
    strl_span_summary sm{};
    if (strl_spanners(&rv, b.isize.data(), &bd, window, frag, 40, sup.data(), sup.size(), &sm) != STRL_OK) { fprintf(stderr, "%s\n", strl_last_error()); return 1; }
    auto t2 = std::chrono::steady_clock::now();
    t_read += std::chrono::duration<double>(t1 - t0).count();
    t_span += std::chrono::duration<double>(t2 - t1).count();
    recs += b.size(); support += sm.n_support;
  }
  printf("%d regions: read_region %.3f ms each (%.0f records), spanners %.3f ms each (%.1f support)\n", n, 1e3 * t_read / n, (double)recs / n, 1e3 * t_span / n, (double)support / n);
  return 0;
}
