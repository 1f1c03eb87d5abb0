// cram_codecs.h -- CRAM 3.1 block codecs (cram_codecs.cpp): compression methods 5 (rANS Nx16) and 8 (name tokeniser)
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace strl {

// `in_len` bytes of a block's payload -> `expect` bytes (the block's raw size).  false + err on anything malformed.
bool cram_rans_nx16_decode(const uint8_t *in, size_t in_len, size_t expect, std::vector<uint8_t> &out, std::string &err);
bool cram_tok3_decode(const uint8_t *in, size_t in_len, size_t expect, std::vector<uint8_t> &out, std::string &err);

// test hooks (cram_reader.cpp): "rans4x8" block, "itf8" / "ltf8" value lists -> decimal lines
bool cram_selftest_decode(const std::string &kind, const uint8_t *in, size_t in_len, size_t expect, std::vector<uint8_t> &out, std::string &err);

}  // namespace strl
