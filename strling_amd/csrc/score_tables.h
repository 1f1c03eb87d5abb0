// score_tables.h -- host-side builders of the two small device tables the scorer reads:
// the window -> minimum-rotation code LUT (slide_by, utils.nim:10-34) and the integer thresholds
// of the score ladder (utils.nim:251,259 with the float64 expressions of the reference).
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "score_core.h"
#include "../../include/strling_amd.h"

namespace strl {

inline void build_lut(std::vector<uint16_t> &lut) {
  lut.assign(LUT_ENTRIES, 0);
  const int off[7] = {0, 0, LUT_OFF2, LUT_OFF3, LUT_OFF4, LUT_OFF5, LUT_OFF6};
  for (int k = 2; k <= 6; ++k) {
    const uint32_t mask = (1u << (2 * k)) - 1u;
    for (uint32_t v = 0; v <= mask; ++v) {
      uint32_t c = 0;  // reference orientation: first base in the high bits
      for (int j = 0; j < k; ++j) c |= ((v >> (2 * j)) & 3u) << (2 * (k - 1 - j));
      uint32_t best = c, f = c;
      for (int j = 0; j < k; ++j) {  // slide_by: rotate, keep the minimum (utils.nim:17-20)
        f = ((f << 2) | (f >> (2 * (k - 1)))) & mask;
        best = std::min(best, f);
      }
      lut[off[k] + v] = (uint16_t)best;
    }
    if (k <= 4) {   // canonical code -> dense class id (rank), and the way back
      const int coff = k == 2 ? LUT_C2 : k == 3 ? LUT_C3 : LUT_C4;
      std::vector<uint16_t> codes(lut.begin() + off[k], lut.begin() + off[k] + mask + 1);
      std::sort(codes.begin(), codes.end());
      codes.erase(std::unique(codes.begin(), codes.end()), codes.end());
      if (codes.size() != (k == 2 ? 10u : k == 3 ? 24u : 70u)) abort();   // necklaces of length k over 4 letters
      for (size_t id = 0; id < codes.size(); ++id) lut[coff + (int)id] = codes[id];
      for (uint32_t v = 0; v <= mask; ++v)
        lut[off[k] + v] = (uint16_t)(std::lower_bound(codes.begin(), codes.end(), lut[off[k] + v]) - codes.begin());
    }
  }
}

// stage A's tables (score_core.h TA_*), from the same class ids build_lut assigns
inline void build_stage_a_tables(const std::vector<uint16_t> &lut, std::vector<uint32_t> &ta) {
  ta.assign(TA_WORDS, 0);
  for (uint32_t b = 0; b < 256; ++b) {
    ta[TA_K2 + b] = (uint32_t)lut[LUT_OFF2 + (b & 15u)] * ROWB | ((uint32_t)lut[LUT_OFF2 + (b >> 4)] * ROWB) << 16;
    const uint32_t c4 = lut[LUT_OFF4 + b];
    ta[TA_K4 + b] = (c4 >> 2) * ROWB | (8u * (c4 & 3u)) << 16;
  }
  uint8_t *t3 = reinterpret_cast<uint8_t *>(ta.data() + TA_K3);
  for (uint32_t v = 0; v < 64; ++v) t3[v] = (uint8_t)lut[LUT_OFF3 + v];
  uint16_t *tc = reinterpret_cast<uint16_t *>(ta.data() + TA_C);
  for (int i = 0; i < 10; ++i) tc[TaCls<2>::off + i] = lut[LUT_C2 + i];
  for (int i = 0; i < 24; ++i) tc[TaCls<3>::off + i] = lut[LUT_C3 + i];
  for (int i = 0; i < 70; ++i) tc[TaCls<4>::off + i] = lut[LUT_C4 + i];
}

inline void build_thr(const strl_opts &o, std::vector<uint64_t> &thr) {
  thr.assign(4 * THR_LMAX, 0);
  const double p = o.proportion_repeat;
  const double ps[4] = {0.12, p, p - 0.07, std::min(p, 0.6)};  // utils.nim:251,259; extract.nim:242,208
  for (int row = 0; row < 4; ++row)
    for (int L = 0; L < THR_LMAX; ++L) {
      uint64_t w = 0;
      for (int k = 2; k <= 6; ++k) {
        int v = (int)((double)L * ps[row] / (double)k);
        v = std::max(0, std::min(v, 255));   // L <= 510: only p > 1 could exceed a byte, and then nothing passes anyway
        w |= (uint64_t)v << (8 * (k - 2));
      }
      thr[row * THR_LMAX + L] = w;
    }
}

// conv8_lut's table: one BAM byte (base 2m in the high nibble, base 2m+1 in the low nibble) -> codes | flags << 16.
// Built FROM conv8 so that the two can never disagree.
inline void build_conv_lut(std::vector<uint32_t> &t) {
  t.assign(256, 0);
  for (uint32_t b = 0; b < 256; ++b) {
    uint32_t pairs, flags;
    conv8(b, pairs, flags);          // the byte sits in the low byte of the dword: bases 0 and 1
    t[conv_swizzle(b) & 0xffu] = (pairs & 0xfu) | ((flags & 0xfu) << 16);     // (indexed by the swizzled byte: score_core.h)
  }
}

}  // namespace strl
