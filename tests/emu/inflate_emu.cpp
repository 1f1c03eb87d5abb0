// Host build of the device DEFLATE decoder (strling_amd/csrc/inflate_wave.h with STRL_EMU: the wave's 64 lanes become
// loops) for the CPU-only tests.
#define STRL_EMU 1
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../strling_amd/csrc/inflate_wave.h"
#include "../../strling_amd/csrc/inflate_group.h"

// comp holds the stream at byte offset `lead` (any alignment), `tail` readable bytes follow it.
extern "C" int emu_inflate_at(const uint8_t *comp, uint32_t clen, uint32_t lead, uint8_t *out, uint32_t isize) {
  using namespace strl;
  static IwLds lds;
  memset(&lds, 0xA5, sizeof lds);
  const size_t readable = ((size_t)lead + clen + 3) & ~(size_t)3;
  std::vector<uint8_t> in(readable + 64, 0xEE);        // bytes behind `readable` must never be touched: poisoned
  memcpy(in.data() + lead, comp, clen);
  std::vector<uint8_t> o((size_t)isize + 128, 0xAA);
  const int rc = iw_inflate(in.data(), lead, clen, readable, o.data() + 21, isize, lds);
  memcpy(out, o.data() + 21, isize);
  for (int i = 0; i < 21; ++i) if (o[i] != 0xAA) return 100;                             // nothing before the stream's first byte
  for (size_t i = 21 + (size_t)isize; i < o.size(); ++i) if (o[i] != 0xAA) return 101;   // nothing behind its last
  return rc;
}
extern "C" int emu_inflate(const uint8_t *comp, uint32_t clen, uint8_t *out, uint32_t isize) { return emu_inflate_at(comp, clen, 5, out, isize); }
extern "C" int emu_lds_bytes(void) { return (int)sizeof(strl::IwLds); }

// The grouped form (inflate_group.h) with G = 1: the same logic as the device's groups of 8 lanes, one "lane" doing every share.
extern "C" int emu_group_inflate_at(const uint8_t *comp, uint32_t clen, uint32_t lead, uint8_t *out, uint32_t isize) {
  using namespace strl;
  static IgLds<1> lds;
  memset(&lds, 0xA5, sizeof lds);
  const size_t readable = ((size_t)lead + clen + 3) & ~(size_t)3;
  std::vector<uint8_t> in(readable + 64, 0xEE);
  memcpy(in.data() + lead, comp, clen);
  std::vector<uint8_t> o((size_t)isize + 128, 0xAA);
  const IwBuf ib = iw_make_buf(in.data(), readable), ob = iw_make_buf(o.data(), o.size());
  std::vector<uint8_t> wk(IG_WORK_STRIDE + 64, 0xCC);
  const IwBuf wb = iw_make_buf(wk.data() + 32, IG_WORK_STRIDE);
  const int rc = ig_inflate<1>(ib, lead, clen, ob, ob, 21u, isize, wb, 0u, lds, 0);
  for (int i = 0; i < 32; ++i) if (wk[i] != 0xCC || wk[32 + IG_WORK_STRIDE + i] != 0xCC) return 102;                 // nothing around the workspace slice
  memcpy(out, o.data() + 21, isize);
  for (int i = 0; i < 21; ++i) if (o[i] != 0xAA) return 100;
  for (size_t i = 21 + (size_t)isize; i < o.size(); ++i) if (o[i] != 0xAA) return 101;
  return rc;
}
extern "C" int emu_group_inflate(const uint8_t *comp, uint32_t clen, uint8_t *out, uint32_t isize) { return emu_group_inflate_at(comp, clen, 5, out, isize); }
extern "C" int emu_group_lds_bytes(int g) { return g == 8 ? (int)sizeof(strl::IgLds<8>) : g == 16 ? (int)sizeof(strl::IgLds<16>) : g == 4 ? (int)sizeof(strl::IgLds<4>) : (int)sizeof(strl::IgLds<1>); }
