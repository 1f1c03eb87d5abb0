// C entry for tests/test_fast_inflate.py
#include "fast_inflate.h"
extern "C" int fi_inflate(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) { return strl::fast_inflate(in, in_len, out, out_len); }
