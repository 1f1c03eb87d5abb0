// comm.hip -- the exchange step of the multi-GPU path inside the library: the ranks' resident tread
// buffers are all-gathered with RCCL over xGMI on the stream the batch's tail runs on, and every rank clusters the
// (tid, unit) groups it owns (strl_cluster_gathered).  Two ways to form the ranks:
//   * one process per GPU (bench.py under torch.distributed.run, a Nim host under mpirun): strl_comm_unique_id on rank 0,
//     the 128 bytes travel by whatever the host framework offers, strl_ctx_comm_init on every rank (ncclCommInitRank);
//   * one process, n contexts (the CLI's --gpus N): strl_ctxs_comm_init -- ncclCommInitAll when the contexts sit on n
//     different devices; when contexts SHARE a device (RCCL refuses two ranks on one device; also how the path is tested
//     on a one-GPU box) the same exchange is made with device-to-device copies ordered by events.
// The reference has no counterpart: it is single-threaded per sample (merge.nim:52,89 is its only sharding knob).
#include <rccl/rccl.h>      // types and prototypes only: the library is not linked, see `Rccl` below
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include "common.h"
#include "device_util.h"

using namespace strl;

extern "C" int strl_ctx_treads_device(strl_ctx *c, void **treads, uint64_t *cap, void **count);
extern "C" void *strl_ctx_tail_stream(strl_ctx *c);

namespace strl {

struct strl_comm {
  ncclComm_t nccl = nullptr;
  int world = 1, rank = 0;
  std::vector<strl_ctx *> peers;        // one process, shared devices: the group's contexts (copies instead of RCCL)
  bool owner = false;                   // this context's comm object owns `peers`' bookkeeping (rank 0 of a local group)
  struct Set {
    hipStream_t st = nullptr;
    DevBuf t_local, t_all, c_all;
    hipEvent_t ready = nullptr;      // shared devices: the send buffer is complete (recorded on the sender's stream)
    hipEvent_t copied = nullptr;     // shared devices: THIS rank's copies out of its peers' send buffers are done (recorded on its stream)
    bool copied_pending = false;
  };
  std::vector<Set *> sets;              // exchange buffers per tail stream (a step's buffers outlive its asynchronous clustering)
  uint32_t pad = 0;
};

// RCCL is bound at the first use, not at link time (round-3 advisor finding): a one-GPU build / run needs no librccl at
// all, and inside a process that already holds one (torch ships its own librccl.so) THAT copy is used instead of loading a
// second RCCL from /opt/rocm beside it.
struct Rccl {
  decltype(&::ncclGetErrorString) GetErrorString = nullptr;
  decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&::ncclCommInitRank) CommInitRank = nullptr;
  decltype(&::ncclCommInitAll) CommInitAll = nullptr;
  decltype(&::ncclCommDestroy) CommDestroy = nullptr;
  decltype(&::ncclAllGather) AllGather = nullptr;
  decltype(&::ncclGroupStart) GroupStart = nullptr;
  decltype(&::ncclGroupEnd) GroupEnd = nullptr;
  std::string origin, error;
  bool ok = false;
};
static Rccl &rccl() {
  static Rccl R = [] {
    Rccl r;
    void *h = nullptr;
    const char *env = getenv("STRL_RCCL_LIB");
    const char *loaded[] = {"librccl.so.1", "librccl.so"};
    const char *fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (env && *env) { h = dlopen(env, RTLD_NOW | RTLD_GLOBAL); r.origin = env; }
    for (const char *n : loaded) if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) r.origin = std::string(n) + " (already in the process)";
    if (!h && dlsym(RTLD_DEFAULT, "ncclAllGather")) { h = RTLD_DEFAULT; r.origin = "symbols already in the process"; }
    for (const char *n : fresh) if (!h && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) r.origin = n;
    if (!h) { r.error = "RCCL not found (librccl.so; set STRL_RCCL_LIB)"; return r; }
#define STRL_RCCL_SYM(name) r.name = reinterpret_cast<decltype(r.name)>(dlsym(h, "nccl" #name)); if (!r.name) { r.error = "RCCL symbol nccl" #name " missing in " + r.origin; return r; }
    STRL_RCCL_SYM(GetErrorString) STRL_RCCL_SYM(GetUniqueId) STRL_RCCL_SYM(CommInitRank) STRL_RCCL_SYM(CommInitAll)
    STRL_RCCL_SYM(CommDestroy) STRL_RCCL_SYM(AllGather) STRL_RCCL_SYM(GroupStart) STRL_RCCL_SYM(GroupEnd)
#undef STRL_RCCL_SYM
    r.ok = true;
    return r;
  }();
  return R;
}
#define STRL_NEED_RCCL()                                                         \
  do {                                                                           \
    if (!rccl().ok) { strl::set_error("%s", rccl().error.c_str()); return STRL_ERR_HIP; } \
  } while (0)
#define ncclGetErrorString rccl().GetErrorString
#define ncclGetUniqueId rccl().GetUniqueId
#define ncclCommInitRank rccl().CommInitRank
#define ncclCommInitAll rccl().CommInitAll
#define ncclCommDestroy rccl().CommDestroy
#define ncclAllGather rccl().AllGather
#define ncclGroupStart rccl().GroupStart
#define ncclGroupEnd rccl().GroupEnd

#define STRL_NCCL(call)                                                                             \
  do {                                                                                              \
    ncclResult_t r__ = (call);                                                                      \
    if (r__ != ncclSuccess) {                                                                       \
      strl::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r__), __FILE__, __LINE__); \
      return STRL_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)

void comm_destroy(strl_comm *m) {
  if (!m) return;
  for (auto *s : m->sets) {
    s->t_local.release(); s->t_all.release(); s->c_all.release();
    if (s->ready) (void)hipEventDestroy(s->ready);
    if (s->copied) (void)hipEventDestroy(s->copied);
    delete s;
  }
  if (m->nccl && rccl().ok) (void)ncclCommDestroy(m->nccl);
  delete m;
}

static strl_comm::Set *set_for(strl_comm *m, hipStream_t st, uint32_t pad) {
  strl_comm::Set *s = nullptr;
  for (auto *x : m->sets) if (x->st == st) s = x;
  if (!s) { s = new strl_comm::Set(); s->st = st; m->sets.push_back(s); }
  if (s->t_local.reserve((size_t)pad * sizeof(strl_tread)) || s->t_all.reserve((size_t)m->world * pad * sizeof(strl_tread)) || s->c_all.reserve((size_t)m->world * 4 + 64)) return nullptr;
  if (!s->ready && hipEventCreateWithFlags(&s->ready, hipEventDisableTiming) != hipSuccess) return nullptr;
  if (!s->copied && hipEventCreateWithFlags(&s->copied, hipEventDisableTiming) != hipSuccess) return nullptr;
  return s;
}

// this rank's part of the gather: order the treads, make the padded send buffer; returns the set and the stream
static int gather_prepare(strl_ctx *c, uint32_t pad, strl_comm::Set **set, hipStream_t *st, void **count) {
  strl_comm *m = c->comm;
  void *treads = nullptr;
  uint64_t cap = 0;
  int rc = strl_ctx_treads_device(c, &treads, &cap, count);        // (the .bin-order sort, on the tail's stream)
  if (rc) return rc;
  STRL_HIP(hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(strl_ctx_tail_stream(c));
  if (!s) s = c->stream;
  strl_comm::Set *S = set_for(m, s, pad);
  if (!S) { set_error("exchange buffers: out of memory"); return STRL_ERR_HIP; }
  // shared devices: the peers of the previous step read this rank's send buffer on THEIR streams -- the refill waits for them
  // (round-3 advisor finding: only the sender -> reader `ready` event existed)
  for (strl_ctx *q : m->peers)
    if (q && q != c && q->comm)
      for (auto *x : q->comm->sets)
        if (x->copied_pending) STRL_HIP(hipStreamWaitEvent(s, x->copied, 0));
  const size_t mbytes = (size_t)std::min<uint64_t>(pad, cap) * sizeof(strl_tread);
  if (mbytes) STRL_HIP(hipMemcpyAsync(S->t_local.p, treads, mbytes, hipMemcpyDeviceToDevice, s));
  *set = S; *st = s;
  return STRL_OK;
}

}  // namespace strl

extern "C" {

int strl_comm_unique_id(uint8_t id[STRL_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= STRL_COMM_ID_BYTES, "ncclUniqueId larger than STRL_COMM_ID_BYTES");
  if (!id) { set_error("null argument"); return STRL_ERR_ARG; }
  STRL_NEED_RCCL();
  ncclUniqueId u;
  STRL_NCCL(ncclGetUniqueId(&u));
  memset(id, 0, STRL_COMM_ID_BYTES);
  memcpy(id, &u, sizeof u);
  return STRL_OK;
}

int strl_ctx_comm_init(strl_ctx *c, int world, int rank, const uint8_t id[STRL_COMM_ID_BYTES]) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) { set_error("strl_ctx_comm_init: bad argument"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  if (c->comm) { comm_destroy(c->comm); c->comm = nullptr; }
  STRL_NEED_RCCL();
  strl_comm *m = new strl_comm();
  m->world = world; m->rank = rank;
  ncclUniqueId u;
  memset(&u, 0, sizeof u);
  if (id) memcpy(&u, id, sizeof u);
  else STRL_NCCL(ncclGetUniqueId(&u));
  const ncclResult_t r = ncclCommInitRank(&m->nccl, world, u, rank);
  if (r != ncclSuccess) { delete m; set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r)); return STRL_ERR_HIP; }
  c->comm = m;
  return STRL_OK;
}

int strl_ctxs_comm_init(strl_ctx **ctxs, int n) {
  if (!ctxs || n < 1) { set_error("strl_ctxs_comm_init: bad argument"); return STRL_ERR_ARG; }
  bool distinct = true;
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) { set_error("strl_ctxs_comm_init: null context"); return STRL_ERR_ARG; }
    for (int j = 0; j < i; ++j) if (ctxs[j]->device == ctxs[i]->device) distinct = false;
  }
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  if (distinct && n > 1) {
    STRL_NEED_RCCL();
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
    STRL_NCCL(ncclCommInitAll(comms.data(), n, devs.data()));
  }
  for (int i = 0; i < n; ++i) {
    if (ctxs[i]->comm) { comm_destroy(ctxs[i]->comm); ctxs[i]->comm = nullptr; }
    strl_comm *m = new strl_comm();
    m->world = n; m->rank = i; m->nccl = comms[(size_t)i];
    m->peers.assign(ctxs, ctxs + n);
    ctxs[i]->comm = m;
  }
  return STRL_OK;
}

int strl_ctx_comm_info(strl_ctx *c, int *world, int *rank, int *uses_rccl) {
  if (!c || !c->comm) { set_error("no communicator on this context"); return STRL_ERR_ARG; }
  if (world) *world = c->comm->world;
  if (rank) *rank = c->comm->rank;
  if (uses_rccl) *uses_rccl = c->comm->nccl != nullptr;
  return STRL_OK;
}

// One process per GPU: collective over the context's communicator.  pad = treads every rank contributes at most (equal on
// all ranks).  Outputs as strl_cluster_gathered (all null: asynchronous, on the tail's stream).
int strl_cluster_exchange(strl_ctx *c, uint32_t pad, int mode, int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support, uint16_t min_clip,
                          uint16_t min_clip_total, uint16_t max_clip_dist, strl_bounds *out, uint64_t cap, uint64_t *n_out, strl_unplaced *unplaced,
                          uint64_t unplaced_cap, uint64_t *n_unplaced, strl_cluster_stats *stats) {
  if (!c || !c->comm || !c->comm->nccl || pad == 0) { set_error("strl_cluster_exchange: no RCCL communicator on this context (strl_ctx_comm_init) or pad == 0"); return STRL_ERR_ARG; }
  strl_comm *m = c->comm;
  strl_comm::Set *S = nullptr;
  hipStream_t st = nullptr;
  void *count = nullptr;
  int rc = gather_prepare(c, pad, &S, &st, &count);
  if (rc) return rc;
  m->pad = pad;
  STRL_NCCL(ncclGroupStart());
  STRL_NCCL(ncclAllGather(S->t_local.p, S->t_all.p, (size_t)pad * sizeof(strl_tread), ncclUint8, m->nccl, st));
  STRL_NCCL(ncclAllGather(count, S->c_all.p, 1, ncclUint32, m->nccl, st));
  STRL_NCCL(ncclGroupEnd());
  return strl_cluster_gathered(c, S->t_all.as<strl_tread>(), S->c_all.as<uint32_t>(), m->world, pad, m->rank, mode, n_tid, pos_bits, window, min_support,
                               min_clip, min_clip_total, max_clip_dist, out, cap, n_out, unplaced, unplaced_cap, n_unplaced, stats);
}

// One process, n contexts (strl_ctxs_comm_init): the same step for all of them, enqueued by this thread; fetch every rank's
// rows with strl_cluster_collect.  pad = 0: the largest tread capacity of the group.
int strl_ctxs_cluster_exchange(strl_ctx **ctxs, int n, uint32_t pad, int mode, int32_t n_tid, int pos_bits, uint32_t window, int32_t min_support,
                               uint16_t min_clip, uint16_t min_clip_total, uint16_t max_clip_dist) {
  if (!ctxs || n < 1) { set_error("strl_ctxs_cluster_exchange: bad argument"); return STRL_ERR_ARG; }
  for (int i = 0; i < n; ++i)
    if (!ctxs[i] || !ctxs[i]->comm || ctxs[i]->comm->world != n || ctxs[i]->comm->rank != i) { set_error("strl_ctxs_cluster_exchange: contexts are not one group (strl_ctxs_comm_init)"); return STRL_ERR_ARG; }
  if (!pad) for (int i = 0; i < n; ++i) pad = std::max(pad, ctxs[i]->tread_cap);
  if (!pad) { set_error("strl_ctxs_cluster_exchange: no treads on any context"); return STRL_ERR_ARG; }
  std::vector<strl_comm::Set *> S((size_t)n, nullptr);
  std::vector<hipStream_t> st((size_t)n, nullptr);
  std::vector<void *> count((size_t)n, nullptr);
  int rc;
  for (int i = 0; i < n; ++i) {
    if ((rc = gather_prepare(ctxs[i], pad, &S[(size_t)i], &st[(size_t)i], &count[(size_t)i]))) return rc;
    ctxs[i]->comm->pad = pad;
  }
  const bool use_rccl = ctxs[0]->comm->nccl != nullptr;
  if (use_rccl) {
    STRL_NCCL(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
      STRL_NCCL(ncclAllGather(S[(size_t)i]->t_local.p, S[(size_t)i]->t_all.p, (size_t)pad * sizeof(strl_tread), ncclUint8, ctxs[i]->comm->nccl, st[(size_t)i]));
      STRL_NCCL(ncclAllGather(count[(size_t)i], S[(size_t)i]->c_all.p, 1, ncclUint32, ctxs[i]->comm->nccl, st[(size_t)i]));
    }
    STRL_NCCL(ncclGroupEnd());
  } else {
    // contexts share a device: every rank copies every rank's send buffer, behind an event on the sender's stream
    for (int i = 0; i < n; ++i) { STRL_HIP(hipSetDevice(ctxs[i]->device)); STRL_HIP(hipEventRecord(S[(size_t)i]->ready, st[(size_t)i])); }
    for (int i = 0; i < n; ++i) {
      STRL_HIP(hipSetDevice(ctxs[i]->device));
      for (int p = 0; p < n; ++p) {
        if (p != i) STRL_HIP(hipStreamWaitEvent(st[(size_t)i], S[(size_t)p]->ready, 0));
        STRL_HIP(hipMemcpyAsync(S[(size_t)i]->t_all.as<uint8_t>() + (size_t)p * pad * sizeof(strl_tread), S[(size_t)p]->t_local.p, (size_t)pad * sizeof(strl_tread),
                                hipMemcpyDeviceToDevice, st[(size_t)i]));
        STRL_HIP(hipMemcpyAsync(S[(size_t)i]->c_all.as<uint32_t>() + p, count[(size_t)p], 4, hipMemcpyDeviceToDevice, st[(size_t)i]));
      }
      STRL_HIP(hipEventRecord(S[(size_t)i]->copied, st[(size_t)i]));
      S[(size_t)i]->copied_pending = true;
    }
  }
  for (int i = 0; i < n; ++i)
    if ((rc = strl_cluster_gathered(ctxs[i], S[(size_t)i]->t_all.as<strl_tread>(), S[(size_t)i]->c_all.as<uint32_t>(), n, pad, i, mode, n_tid, pos_bits, window,
                                    min_support, min_clip, min_clip_total, max_clip_dist, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr)))
      return rc;
  return STRL_OK;
}

// all ranks' treads of the last exchange in global (rank, .bin) order on the host (the final row order: strl_group_order)
int strl_exchange_treads(strl_ctx *c, strl_tread *out, uint64_t cap, uint64_t *n_out) {
  if (!c || !c->comm || c->comm->sets.empty()) { set_error("strl_exchange_treads: no exchange on this context"); return STRL_ERR_ARG; }
  STRL_HIP(hipSetDevice(c->device));
  strl_comm *m = c->comm;
  hipStream_t st = static_cast<hipStream_t>(strl_ctx_tail_stream(c));
  if (!st) st = c->stream;
  strl_comm::Set *S = nullptr;
  for (auto *x : m->sets) if (x->st == st) S = x;
  if (!S) S = m->sets.back();
  STRL_HIP(hipStreamSynchronize(S->st));
  std::vector<uint32_t> cnt((size_t)m->world);
  STRL_HIP(hipMemcpy(cnt.data(), S->c_all.p, (size_t)m->world * 4, hipMemcpyDeviceToHost));
  const uint64_t pad = m->pad;
  uint64_t tot = 0;
  for (int r = 0; r < m->world; ++r) tot += std::min<uint64_t>(cnt[(size_t)r], pad);
  if (n_out) *n_out = tot;
  if (!out) return STRL_OK;
  if (tot > cap) { set_error("strl_exchange_treads: %llu treads, capacity %llu", (unsigned long long)tot, (unsigned long long)cap); return STRL_ERR_CAPACITY; }
  uint64_t at = 0;
  for (int r = 0; r < m->world; ++r) {
    const uint64_t k = std::min<uint64_t>(cnt[(size_t)r], pad);
    if (k) STRL_HIP(hipMemcpy(out + at, S->t_all.as<strl_tread>() + (uint64_t)r * pad, (size_t)k * sizeof(strl_tread), hipMemcpyDeviceToHost));
    at += k;
  }
  return STRL_OK;
}

}  // extern "C"
