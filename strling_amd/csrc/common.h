// common.h -- internal declarations shared by the C-ABI translation units.
#pragma once
#include <condition_variable>
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/strling_amd.h"

namespace strl {

void set_error(const char *fmt, ...);

#define STRL_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t e__ = (call);                                                                        \
    if (e__ != hipSuccess) {                                                                        \
      strl::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);  \
      return STRL_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

// growable device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);
  // like reserve, but the first keep_bytes survive a reallocation (copied on `st`, which is synchronised before the old
  // block is freed); grows geometrically
  int grow(size_t bytes, size_t keep_bytes, hipStream_t st);
  void release();
  template <typename T> T *as() { return static_cast<T *>(p); }
};

// counters of one scoring pass (strl_ctx::counters), 64 B apart
constexpr int CNT_STRIDE = 16;
constexpr int CNT_QUEUE = 0, CNT_SOFT = 16, CNT_SKIP = 32, CNT_SBW = 48, CNT_SBS = 64, CNT_WORDS = 80;
// counters of the device pair logic (strl_ctx::pair_cnt)
constexpr int PC_ITEMS = 0, PC_EMIT = 16, PC_ERR = 32, PC_SPILL = 40, PC_WORDS = 48;
// counters of a chunked extract (strl_ctx::x_cnt): soft records appended so far, then the sums of the chunks' CNT_* counters
constexpr int XC_SOFT = 0, XC_SKIP = 1, XC_QUEUE = 2, XC_SBW = 3, XC_SBS = 4, XC_SOFT_ITEMS = 5, XC_OVERFLOW = 6, XC_WORDS = 16;
constexpr int PAIR_LONG_MAX_ITEMS = 512;   // longest hash run the device pair logic replays (pair.hip PAIR_LONG_MAX)
constexpr uint32_t PAIR_ERR_RUN = 1u, PAIR_ERR_ASSERT = 2u, PAIR_ERR_ITEMS = 4u, PAIR_ERR_EMIT = 8u, PAIR_ERR_LOCAL = 16u, PAIR_ERR_COLLISION = 32u;

// murmur3 finaliser: a bijection on 64-bit words, so equality of mixed hashes == equality of hashes
__host__ __device__ inline uint64_t fmix64(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return h;
}

}  // namespace strl

// the last clustering pass of a context: what strl_cluster_replay / strl_cluster_members need
struct ClusterRun {
  uint32_t n_max = 0;              // launch bound of the pass (treads the buffers are sized for)
  uint32_t n = 0, n_groups = 0, n_clusters = 0;   // known after the results were collected
  int32_t n_tid = 0;
  int kbits = 0, pos_bits = 32, mode = 0;
  bool fold = false;   // pos_bits is a caller's bound (resident / gathered treads): the field's upper half takes wrapped positions
  bool composite = false;
  uint32_t window = 0;
  int32_t min_support = 0;
  uint32_t min_clip = 0, min_clip_total = 0, max_clip_dist = 0;
  const strl_tread *treads = nullptr;   // device
  const uint32_t *d_n = nullptr;        // device
  const uint32_t *perm = nullptr;       // device: sorted index -> input index
  const uint64_t *first_key = nullptr;  // device: per-tread key that orders first appearances (nullptr: the input index)
  const uint32_t *part_err = nullptr;   // device: error word of the owner partition in front of the pass (a rank sent more treads than `pad`)
  // members of the bounds the last pass returned: [first, first + count) in sorted order; `kept` maps the
  // uploaded (filtered) treads back to the caller's indices when merge mode dropped unplaced ones
  std::vector<uint32_t> b_first, b_count, kept;
};

// Records one device pass over a whole input takes (join items and soft-clip records name a record in 31 bits).
// STRL_RECORD_LIMIT lowers it (tests of the route the CLI takes beyond it).
inline uint64_t strl_record_limit() {
  static const uint64_t v = [] {
    const char *e = getenv("STRL_RECORD_LIMIT");
    const unsigned long long x = e ? strtoull(e, nullptr, 10) : 0;
    return x > 0 && x < 0x7ffffff0ull ? (uint64_t)x : (uint64_t)0x7ffffff0ull;
  }();
  return v;
}

namespace strl { struct strl_front; struct strl_comm; void comm_destroy(strl_comm *m); }
struct strl_ctx;
int side_join(strl_ctx *c);   // main stream waits for the side streams' pending work (score.hip)
int side_streams(strl_ctx *c);   // the side streams of the overlapped mode, made at its first use (score.hip)
void rotate_tail(strl_ctx *c);  // make the least recently used set of pair-logic / clustering state the current one (score.hip)
constexpr int N_SETS = 2;        // batches in flight on a context: buffer sets of the scorer's output and of the tail (3 measured no faster than 2)

// Everything the pair logic and the clustering of ONE batch own (the "tail" of a step).  A context has N_SETS: the members of
// strl_ctx with these names are the current set, `alt[]` the others (most recently used first).  The overlapped
// strl_extract_device rotates through them, each with its own side stream, so the tails of consecutive batches -- chains of small dependent launches that
// crawl while the main stream saturates the chip -- make progress side by side.
struct TailSet {
  strl::DevBuf c_buf[16];
  ClusterRun cl_run;
  strl::DevBuf p_key0, p_key1, p_val0, p_val1, p_emit, sort_scratch, pair_cnt, treads;
  uint32_t *n_treads_dev = nullptr;
  uint32_t tread_cap = 0, pair_item_cap = 0;
  uint64_t *po_key = nullptr, *po_key_alt = nullptr;
  uint32_t *po_val = nullptr, *po_val_alt = nullptr;
  int po_bits = 0;
  bool pair_ordered = false;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_side_done = nullptr;
  bool side_pending = false, pair_on_side = false;
};

struct strl_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // Side stream: an asynchronous strl_cluster_resident (results stay on the device) runs here, so that the clustering of
  // one batch -- 20 small, latency-bound launches -- overlaps the VALU-bound scorer of the next batch.  Whatever touches
  // the treads or the cluster state afterwards calls side_join() first.
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_main_done = nullptr, ev_side_done = nullptr;
  bool side_pending = false;
  // strl_extract_device on device-resident input goes further: the pair logic of batch i (latency-bound: Bloom probes,
  // sorts, gathers) runs on the side stream too, beside classify + scorer of batch i + 1 on the main stream.  What the
  // pair logic reads of the scorer's output exists N_SETS times (whole[], soft-clip records, counters, Bloom bitmap); a call
  // rotates the sets and waits (on the device) until the side stream is done with the set it is about to overwrite.
  struct HeadSet { strl::DevBuf st_whole, st_soft, counters, bloom; uint32_t bloom_mask = 0; } head_alt[N_SETS - 1];   // most recently used first
  hipEvent_t ev_head_done = nullptr, ev_set_free[N_SETS] = {};
  bool set_used[N_SETS] = {};
  int set = 0;
  bool pair_on_side = false;       // the last pair logic ran on the side stream (its treads are ordered there already)
  TailSet alt[N_SETS - 1];         // the other tail sets (see TailSet), most recently used first
  int cl_where = 0;                // the last clustering pass lives in: 0 = the current tail set, k = alt[k - 1]
  bool timing = false;
  bool blocking_waits = false;     // strl_ctx_blocking_waits: the front end's events are created with hipEventBlockingSync
  hipEvent_t ev[8] = {};
  std::vector<hipEvent_t> ring;   // 4 events per recorded strl_score_reads launch
  uint64_t ring_pos = 0;
  // options / tables
  bool have_opts = false;
  strl_opts opts{};
  strl::DevBuf lut;  // uint16[LUT_ENTRIES]
  strl::DevBuf inv_spill;   // score.hip: Seg::inv of the waves that hold a base that is not ACGT
  strl::DevBuf thr;  // uint16[4][5][512]
  // genome STR intervals, per tid sorted by start, with prefix max of stop
  int32_t n_tid = 0;
  uint64_t n_iv = 0;
  strl::DevBuf g_tid, g_bins, g_start, g_pmax;
  // scratch
  strl::DevBuf queue, soft_queue, counters, soft_tmp, sb_whole, sb_soft, soft_dense, sb_state_w, sb_state_s, queue_r;
  // staging for host-memory batches
  strl::DevBuf st_tid, st_pos, st_end, st_seqoff, st_lseq, st_clipl, st_clipr, st_mapq, st_cig, st_seq4, st_whole, st_soft, st_text, st_meta;
  // reads the host twin scores (score.hip long_reads_pass): their list, their SEQ bytes / the words that come back
  strl::DevBuf long_list, long_seq;
  // clustering scratch
  strl::DevBuf c_buf[16];
  ClusterRun cl_run;
  // device pair logic (pair.hip): join items / emission keys (ping-pong), emitted treads, Bloom bitmap, counters
  strl::DevBuf p_key0, p_key1, p_val0, p_val1, p_emit, sort_scratch, pair_cnt, bloom;
  uint32_t bloom_mask = 0;
  // the treads the last strl_extract_device call produced, in .bin order: treads[0, *n_treads_dev)
  strl::DevBuf treads;
  uint32_t *n_treads_dev = nullptr;
  uint32_t tread_cap = 0, pair_item_cap = 0;
  // the last pair pass' treads before ordering: p_emit[i] with emission key po_key[i] (and the sort's other buffers)
  uint64_t *po_key = nullptr, *po_key_alt = nullptr;
  uint32_t *po_val = nullptr, *po_val_alt = nullptr;
  int po_bits = 0;
  bool pair_ordered = false;
  uint64_t ex_n = 0, ex_soft_cap = 0;
  // chunked extract (strl_extract_begin / _add / _finish): per-read state of all chunks so far
  strl::DevBuf x_rows, x_qhash, x_whole, x_soft, x_cnt, g_aux;
  uint64_t x_n = 0, x_soft_cap = 0;
  // host's knowledge of the device-side soft-clip record count: `x_soft_known` records after `x_soft_known_at` reads
  uint64_t x_soft_known = 0, x_soft_known_at = 0, x_soft_seen_at = 0;
  uint32_t *x_soft_seen = nullptr;        // pinned
  hipEvent_t x_soft_seen_ev = nullptr;
  bool x_soft_pending = false;
  bool x_open = false, x_mode = false;
  bool x_front = false;            // the chunks came through the device front end: qnames of all records sit in its arena
  strl::DevBuf crc_tab;            // tables of the BGZF CRC-32 check (bgzf.hip)
  // strl_regions_fetch (bgzf.hip): two calls of different host threads run side by side, each on its slot's stream with its
  // slot's buffers (kept between calls) -- the copies of one batch of regions pass beside the inflate of the other
  struct RegionSlot { hipStream_t st = nullptr; strl::DevBuf comp, meta, u, out, rq, work; bool busy = false; };
  RegionSlot rg[2];
  std::mutex rg_mu;
  std::condition_variable rg_cv;
  strl::DevBuf p_spill;            // pair logic: first items of the hash runs too long for the in-block replay
  bool pg_attr_done = false;
  hipEvent_t pev[6] = {};
  double inflate_ms = 0;           // kernel time of the last strl_inflate_blocks call
  strl::strl_front *front = nullptr;   // device BAM front end (front.h), created by strl_front_begin
  strl::strl_comm *comm = nullptr;     // multi-GPU exchange (comm.hip): RCCL communicator / local group of this context
  // staging of the pairing arrays for host-memory batches
  strl::DevBuf st_mtid, st_mpos, st_flag, st_qhash;
};

// pair.hip: enqueue the device pair logic behind a scoring pass of the same batch
int strl_pair_order(strl_ctx *c, hipStream_t on_stream = nullptr);   // nullptr = the main stream (after a side_join)
int strl_pair_device(strl_ctx *c, uint64_t n, const strl_pair_soa *pp, const uint32_t *whole, const strl_soft_rec *soft,
                     const uint32_t *d_n_soft, uint64_t soft_cap, int64_t n_tail, uint64_t item_cap, uint64_t tread_cap,
                     hipStream_t on_stream = nullptr);   // nullptr = the main stream
