// fast_inflate.h -- raw DEFLATE (RFC 1951) decoder for BGZF blocks, written for throughput per host thread.
//
// `strling extract` reads the BAM through htslib on one thread (extract.nim:275-329); here the host front end inflates
// BGZF blocks on every CPU the container grants, and the inflate itself is the phase the end-to-end loop waits for
// (DESIGN section 7).  zlib's inflate() decodes ~0.65 GB/s per thread on BAM blocks; this decoder keeps a 64-bit bit
// buffer refilled with one unaligned 8-byte load, resolves literal/length codes through an 11-bit table (+ subtables),
// decodes up to three literals per refill and copies matches a word at a time.  It decodes whole blocks only (input and
// output sizes known, as in BGZF) and never writes outside [out, out + out_len).
//
// Any stream it does not decode to exactly out_len bytes returns non-zero; callers then hand the block to zlib, whose
// verdict (and error text) stays authoritative.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace strl {

// in[0, in_len) = the DEFLATE stream; the caller guarantees that in[in_len .. in_len + 8) is READABLE memory (a BGZF block
// carries an 8-byte CRC32 + ISIZE trailer there).  Returns 0 when the stream ends with its final block having produced
// exactly out_len bytes.
int fast_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

}  // namespace strl
