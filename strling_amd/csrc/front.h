// front.h -- internal interface of the device BAM front end (front.hip): BGZF blocks -> inflated bytes -> record table ->
// the structure-of-arrays batch + pair rows + qname hashes the scorer and the pair logic consume, with no host parsing.
#pragma once
#include "common.h"
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace strl {

constexpr uint32_t FRONT_SEG = 16384;            // bytes of inflated data whose records one lane chains through
constexpr uint32_t FRONT_CARRY_MAX = 1u << 20;   // room in front of a chunk's inflated bytes for the partial record the previous chunk ended in
constexpr uint32_t FRONT_NONE = 0xffffffffu;
constexpr uint32_t FRONT_ERR_INFLATE = 1, FRONT_ERR_RECORD = 2, FRONT_ERR_LSEQ = 4, FRONT_ERR_CARRY = 8, FRONT_ERR_CRC = 16;

// device-resident summary of one chunk; the host reads it back after the record scan and again after the parse
struct FrontInfo {
  uint32_t start0;        // offset of the first record in the chunk's inflated buffer (set by the carry copy / the host)
  uint32_t end;           // end of the inflated bytes
  uint32_t n_records;     // complete records (secondary / supplementary included: they keep their index)
  uint32_t carry_off, carry_len;   // the trailing partial record, carried into the next chunk
  uint32_t max_l_seq;
  uint32_t err;           // FRONT_ERR_*
  uint32_t inflate_err;   // IW_ERR_* of all blocks
  uint64_t seq_bytes;     // sum of the records' SEQ sizes, each padded to 16 bytes
  uint64_t qname_bytes;   // sum of the qname lengths (without the NUL)
  uint32_t all_ok;        // record scan: every segment's chain arrived exactly at the next segment's guessed start
  uint32_t slow_segments; // segments the sequential fix walked again
  uint32_t n_primary;     // parse: records that are neither secondary nor supplementary
  int32_t last_placed;    // parse: index of the last record with tid >= 0, -1 if none
  uint32_t tail_primary;  // parse: primary records behind it
  uint32_t pad;
};

struct FrontSeg {         // one FRONT_SEG-byte segment of the inflated bytes
  uint32_t guess;         // guessed first record start inside the segment (FRONT_NONE: none found)
  uint32_t entry;         // where the walk of this segment started
  uint32_t exit;          // where it ended (first record start at or behind the segment's end, or the record that does not fit)
  uint32_t cnt;           // complete records starting in the segment
  uint32_t seq16;         // their SEQ sizes in 16-byte units
  uint32_t qn;            // their qname bytes
  uint32_t max_l_seq;
  uint32_t flags;         // 1 = the walk stopped inside the data (partial record / end of data), 2 = malformed record
};

// buffers of one chunk in flight (the context keeps two)
struct FrontSlot {
  DevBuf comp, infl, coff, clen, uoff, isize, crc, status, seg, recoff, seqoff, qoff, info, base3, carry_stage, iwork;   // (iwork: the grouped inflate's workspace)
  uint32_t n_blocks = 0, n_seg = 0;
  uint64_t infl_bytes = 0, comp_bytes = 0;
  hipEvent_t ev_carry = nullptr;  // owned (created on this context's device): recorded behind THIS slot's copy of another context's tail
  hipEvent_t wait_read = nullptr; // not owned: the ev_carry of the context that copied this slot's tail (multi-GPU carry); an event is
  bool read_pending = false;      // recorded on a stream of its own device only, a wait on it is legal from any device
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_h2d = nullptr;   // record scan done | parse + scoring done (slot reusable) | compressed bytes on the device
  hipEvent_t ev_i = nullptr;                                      // inflate + CRC done (compressed bytes and block tables reusable)
  bool b_pending = false, a_pending = false, i_pending = false, h2d_pending = false;
  bool staged = false;            // front_copy has run for the slot's next chunk (strl_front_stage); the push must hand over the same chunk
  const uint8_t *staged_comp = nullptr;
  uint64_t staged_bytes = 0, staged_tot = 0;
  uint32_t staged_blocks = 0;
  uint32_t staged_trim = 0;       // inflated bytes at the end of the staged chunk that belong to the next share (strl_front_trim_next)
  FrontInfo *h_info = nullptr;   // pinned: [0] as of the record scan, [1] as of the parse, [2] the initial values
  uint64_t *h_uoff = nullptr;    // pinned: output offsets of the blocks
  uint32_t h_uoff_cap = 0;
};

// The per-read state of a whole-genome file (80 B per read: rows, hashes, scorer words, name references, fragment words, names)
// is tens of gigabytes of hipMalloc -- 0.15 s for 1.6e8 reads, 0.6 s for 6.5e8, in front of the first chunk.  strl_front_begin
// allocates it for the first eighth of the hint and leaves the full-size buffers to a thread; the front end moves over to them
// (a device copy of what is filled so far) when they are ready or when the small ones are full.
struct FrontBigAlloc {
  std::thread th;
  std::atomic<int> done{0};
  int rc = 0;
  std::string err;
  DevBuf rows, qhash, whole, qref, fragw, qarena;
};

struct strl_front {
  FrontBigAlloc *big = nullptr;
  uint64_t small_reads = 0;          // reads the synchronously allocated buffers hold
  std::vector<DevBuf> trash;         // the small buffers after the move (kernels in flight may still read them): freed with the front end
  hipStream_t st_i[2] = {nullptr, nullptr};   // inflate + CRC of the chunk in slot 0 / 1, lowest priority: a launch fills every CU for ~17 ms; the scan,
                                     // parse and scorer kernels of the neighbouring chunks take the slots its waves give up.  Two streams: chunk k+1's inflate
                                     // needs nothing of chunk k (its partial first record comes through carry_buf, behind the inflate), so its waves take the
                                     // slots chunk k's last waves leave -- one stream had the device drain to a handful of waves between launches
  DevBuf carry_buf[2];               // [slot]: u32 length (64-byte header) + the partial record the slot's chunk ended in, written behind its record scan
  hipStream_t st_a = nullptr;        // record scan (behind the chunk's inflate, beside the next chunk's)
  hipStream_t st_c = nullptr;        // copies of the compressed bytes to the device: the next chunk's copy runs beside this chunk's inflate
  FrontSlot slot[2];
  int n_ref = 0;
  uint32_t hint_blocks = 0;          // front_reserve: blocks per chunk the buffers were sized for
  uint64_t b_issued = 0;             // chunks whose parse + scoring have been enqueued (strl_front_collect)
  uint64_t first_off = 0;            // offset of the first record in the first chunk's inflated bytes
  uint64_t chunks = 0;               // chunks pushed so far
  bool not_first = false;            // (multi-GPU) this context's first chunk is not the file's first
  int last_slot = -1;                // slot of the chunk pushed last
  uint32_t last_end = 0;             // end of its inflated bytes
  uint64_t comp_total = 0, infl_total = 0;   // bytes handed over / inflated so far
  int pending = -1;                  // slot whose stage B has not been enqueued yet
  uint32_t next_trim = 0;            // strl_front_trim_next: taken by the next front_copy
  // per-read state of all chunks (beside x_rows / x_qhash / x_whole of the chunked extract)
  DevBuf qref, qarena, fragw, tidflag, tid_seen;   // tid_seen[n_ref]: contigs with a primary record so far
  uint64_t qarena_used = 0;
  // SoA of the chunk being scored (chunk-temporary)
  DevBuf s_tid, s_pos, s_end, s_seqoff, s_lseq, s_clipl, s_clipr, s_mapq, s_cig, s_seq4, s_meta;
  double ms_inflate = 0, ms_scan = 0, ms_parse = 0;
  std::vector<hipEvent_t> tev;       // timing events (STRL_FRONT_TIMING)
};

struct FrontChunkDesc {   // host view of a chunk handed to front_stage_a
  const uint8_t *comp;    // compressed bytes (pinned host memory for an asynchronous copy)
  uint64_t comp_bytes;
  const uint64_t *coff;   // [n] offset of each block's DEFLATE payload in comp
  const uint32_t *clen, *isize;
  const uint32_t *crc;    // [n] CRC-32 of each block's inflated bytes (BGZF trailer), or null: not checked
  uint32_t n_blocks;
};

// where the partial record in front of a chunk comes from when the previous chunk went to ANOTHER context (multi-GPU
// extract: chunks go round-robin over the contexts): that context's slot, its device, the event behind its record scan
struct FrontCarrySrc {
  const uint8_t *infl;
  const FrontInfo *info;
  uint32_t end;          // end of the inflated bytes in that slot
  int device;
  hipEvent_t ev_a;
  hipEvent_t *wait_read; // the reader stores ITS event (recorded behind its copy) here: the owner waits for it before it overwrites the slot
  bool *read_pending;
};

// front.hip
int front_stage_a(strl_ctx *c, strl_front *F, int slot, const FrontChunkDesc &d, bool first, const FrontCarrySrc *carry = nullptr);
int front_copy(strl_ctx *c, strl_front *F, int slot, const FrontChunkDesc &d);
int front_reserve(strl_ctx *c, strl_front *F, uint32_t max_blocks, uint64_t max_comp_bytes);
struct FrontParseOut {
  int32_t *tid, *pos, *end;
  uint32_t *seq_off;
  uint16_t *l_seq, *clip_l, *clip_r;
  uint8_t *mapq, *cig, *seq4;
  uint4 *meta;             // [n] strl_read_meta rows
  strl_pair_rec *rows;     // [n] persistent rows of this chunk's records
  uint64_t *qhash;
  uint64_t *qref;          // [n] (qname arena offset << 8) | length
  uint8_t *qarena;         // arena base
  uint64_t qarena_at;      // where this chunk's names start
  uint32_t *fragw;         // [n] flag | (isize in [0, 4095] ? isize : 0xffff) << 16
  uint8_t *tidflag;        // [n] bit 0 primary, bit 1 placed (tid >= 0)
};
int front_parse(strl_ctx *c, strl_front *F, int slot, uint32_t n, const FrontParseOut &o, hipStream_t st);
int front_gather_names(strl_ctx *c, strl_front *F, const uint32_t *d_ids, uint32_t n, uint64_t *d_ref_out, hipStream_t st);
int front_copy_names(strl_ctx *c, strl_front *F, const uint64_t *d_ref, const uint64_t *d_dst_off, uint32_t n, uint8_t *d_out, hipStream_t st);
int front_tread_names(strl_ctx *c, strl_front *F, const strl_tread *d_treads, const uint32_t *d_n, uint32_t cap, uint64_t *d_ref, uint32_t *d_len, uint64_t *d_off,
                      uint8_t *d_out, uint64_t out_cap, uint64_t *d_tile_sums, hipStream_t st);
size_t front_name_tiles(uint32_t cap);      // words of d_tile_sums for `cap` names
void front_destroy(strl_front *F);

}  // namespace strl
