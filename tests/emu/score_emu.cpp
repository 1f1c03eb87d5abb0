// score_emu.cpp -- TEST-ONLY host build of the device scorer (strling_amd/csrc/score_core.h with
// STRL_EMU: one lane, "LDS" in a plain array).  It lets the CPU test-suite exercise the exact
// device logic (bit packing, LUT, tie rule, bit-parallel recount, threshold ladder) against the
// oracle without a GPU.  The product never links this file.
#define STRL_EMU 1
#include "../../strling_amd/csrc/score_core.h"
#include "../../strling_amd/csrc/score_tables.h"
#include <string.h>

using namespace strl;

static std::vector<uint16_t> g_lut;
static std::vector<uint64_t> g_thr;
static std::vector<uint32_t> g_clut, g_ta;
static bool g_last_alive = false;

template <int NW, int SLOTS>
static void run(const uint8_t *seq4, int s0, int len, int row0, int row1, bool whole, uint32_t *o0, uint32_t *o1) {
  static uint32_t tab[SLOTS + 8];   // SLOTS rows + the dummy row
  constexpr int MAXCH = (16 * NW + 62) / 32;
  const int s0l = s0 & 31;
  const int nch = (s0l + len + 31) >> 5;
  const uint8_t *src = seq4 + (size_t)(s0 >> 5) * 16;
  for (int c = 0; c < MAXCH && c < nch; ++c) memcpy(&tab[4 * c], src + 16 * c, 16);
  Seg<NW> sg;
  uint32_t inv_mem[NW];
  static uint32_t inv_lds[INV_SLOTS * NW];
  sg.inv_lds = inv_lds;
  sg.inv = inv_mem;
  sg.inv_stride = 1;
  sg.inv_nslots = INV_SLOTS;
  const LenBounds lb = len_bounds(true, len);
  if (whole && s0 == 0) {   // the whole-read kernel converts straight from the loaded registers
    uint32_t raw[4 * MAXCH];
    for (int i = 0; i < 4 * MAXCH; ++i) raw[i] = i < 4 * nch ? tab[i] : 0u;
    seg_from_words<NW>(raw, g_clut.data(), len, lb, sg);
  } else {
    seg_from_raw<NW>(tab, g_clut.data(), s0l, len, sg);
  }
  // run the two stages the way the kernels do: stage A, hand the state over, re-stage the bases, stage B
  ScoreState st;
  LaneThr lt;
  load_thr(g_thr.data(), row0, row1, len, lt);
  score_stage_a<NW, SLOTS>(sg, true, tab, tab, 0, g_ta.data(), lt, lb, st);
  g_last_alive = st.alive;
  if (st.alive) {
    for (int c = 0; c < MAXCH && c < nch; ++c) memcpy(&tab[4 * c], src + 16 * c, 16);
    Seg<NW> sg2;
    sg2.inv_lds = inv_lds;
    sg2.inv = inv_mem;
    sg2.inv_stride = 1;
    sg2.inv_nslots = INV_SLOTS;
    seg_from_raw<NW>(tab, g_clut.data(), s0l, len, sg2);
    score_stage_b<NW, SLOTS>(sg2, tab, 0, g_lut.data(), lt, lb, st);
  }
  *o0 = reduce_packed(st.res0);
  *o1 = reduce_packed(st.res1);
}

extern "C" {
int emu_last_alive() { return g_last_alive ? 1 : 0; }
void emu_set_p(double p) {
  strl_opts o{};
  o.proportion_repeat = p;
  build_lut(g_lut);
  build_thr(o, g_thr);
  build_conv_lut(g_clut);
  build_stage_a_tables(g_lut, g_ta);
}
// mode 0: whole read (threshold p); mode 1: soft clip (p-0.07 / min(p,0.6)).  seq4 must have 32 B slack.
void emu_score(const uint8_t *seq4, int s0, int len, int mode, int klass, uint32_t *o0, uint32_t *o1) {
  const int r0 = mode == 0 ? 1 : 2, r1 = mode == 0 ? 1 : 3;
  if (klass == 0) run<10, 64>(seq4, s0, len, r0, r1, mode == 0, o0, o1);
  else if (klass == 1) run<16, 128>(seq4, s0, len, r0, r1, mode == 0, o0, o1);
  else run<32, 256>(seq4, s0, len, r0, r1, mode == 0, o0, o1);
}
}
