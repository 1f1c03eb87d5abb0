// Host build of the device DEFLATE decoder (strling_amd/csrc/inflate_core.h with STRL_EMU) for the CPU-only tests.
#define STRL_EMU 1
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../strling_amd/csrc/inflate_core.h"

extern "C" int emu_inflate(const uint8_t *comp, uint32_t clen, uint8_t *out, uint32_t isize) {
  using namespace strl;
  std::vector<uint16_t> sym_ll(L_SYMLL), sym_d(L_SYMD), cnt(L_CNT), offs(L_CNT);
  std::vector<uint32_t> win(INF_R / 4), lens(L_LENS / 4);
  // the decoder reads whole aligned dwords around the stream: give it an aligned copy with slack
  std::vector<uint8_t> in((size_t)clen + 16, 0);
  memcpy(in.data() + 4, comp, clen);
  std::vector<uint8_t> o((size_t)isize + 64, 0xAA);
  const int rc = inflate_lane(in.data() + 4, clen, o.data() + 21, isize, sym_ll.data(), sym_d.data(), cnt.data(), offs.data(), win.data(), lens.data(), 0);
  memcpy(out, o.data() + 21, isize);
  for (int i = 0; i < 21; ++i) if (o[i] != 0xAA) return 100;                       // nothing before the stream's first byte
  for (size_t i = 21 + (size_t)isize; i < o.size(); ++i) if (o[i] != 0xAA) return 101;   // nothing behind its last
  return rc;
}
