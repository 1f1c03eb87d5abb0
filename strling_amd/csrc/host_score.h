// host_score.h -- the host's scorer for reads longer than the kernels take (host_score.cpp)
#pragma once
#include <stdint.h>
#include "../../include/strling_amd.h"

namespace strl {

// utils.get_repeat (utils.nim:236-271) on an ASCII read of any length, under n_p <= 4 thresholds at once (the ladder's path does
// not depend on the threshold, only what it keeps): out[q] = packed unit/count word after reduce_repeat
void host_get_repeat(const char *read, int L, const double *p, int n_p, uint32_t *out);

// to_tread's get_repeat + add_soft's clip scans of one record (extract.nim:20-40, 93-116)
void host_score_long_read(const uint8_t *seq4, uint32_t L, uint32_t clip_l, uint32_t clip_r, uint32_t cig, uint32_t mapq, const strl_opts &o, bool skipped,
                          uint32_t id, uint32_t &whole, strl_soft_rec soft[2], int &n_soft);

}  // namespace strl
