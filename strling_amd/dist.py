"""Multi-GPU extract: reads shard by record, one process per GPU (torch.distributed: "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

Scoring needs no communication.  The pair logic (extract.nim:192-248) is keyed by qname, qname groups never
interact, and a group none of whose records carries a repeat emits nothing -- so the only exchange the path needs
is small:
  1. all-gather of the qname hashes of each rank's "hot" records (whole-read or soft-clip count > 0; ~2 % of reads);
  2. all-gather of the compact records (coordinates, flag, mapq, cigar, qname, scorer words -- no SEQ) whose hash is
     in the union, i.e. the hot groups with their mates wherever those were scored.
Every rank then replays Cache.add over the merged records in global file order and obtains the same treads as a
single-GPU run; clustering follows on the gathered treads.  Payload ~ a few % of the records, tens of MB per 30x
sample: latency-bound on xGMI, so one fused all-gather per step.
"""
import numpy as np
import torch.distributed as dist

from . import api
from .records import RecordBatch


def shard_bounds(n, world):
    """contiguous record ranges [lo, hi) per rank"""
    step = (n + world - 1) // world
    return [(min(r * step, n), min((r + 1) * step, n)) for r in range(world)]


def slice_records(rec, lo, hi):
    """sub-batch [lo, hi) of a RecordBatch (keeps SEQ so the shard can be scored)"""
    c0, c1 = int(rec.cigar_off[lo]), int(rec.cigar_off[hi])
    q0, q1 = int(rec.qname_off[lo]), int(rec.qname_off[hi])
    s0 = int(rec.seq_off[lo]) if hi > lo else 0
    s1 = int(rec.seq_off[hi - 1]) + (int(rec.l_seq[hi - 1]) + 1) // 2 if hi > lo else 0
    seq4 = np.concatenate([rec.seq4[s0:s1], np.zeros(32, np.uint8)])
    return RecordBatch(rec.tid[lo:hi], rec.pos[lo:hi], rec.mtid[lo:hi], rec.mpos[lo:hi], rec.flag[lo:hi], rec.mapq[lo:hi],
                       (rec.cigar_off[lo:hi + 1] - c0).astype(np.uint32), rec.cigar[c0:c1], (rec.seq_off[lo:hi] - np.uint64(s0)),
                       rec.l_seq[lo:hi], seq4, (rec.qname_off[lo:hi + 1] - np.uint64(q0)), rec.qnames[q0:q1],
                       None if rec.isize is None else rec.isize[lo:hi], rec.targets)


def _compact(rec, idx, whole, soft):
    """SEQ-free payload of the selected records (local indices idx, ascending)"""
    clen = (rec.cigar_off[idx + 1] - rec.cigar_off[idx]).astype(np.int64)
    cig = np.concatenate([rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])] for i in idx]) if idx.size else np.zeros(0, np.uint32)
    qn = [rec.qname(int(i)) for i in idx]
    side = soft["read_side"] & 1
    sread = (soft["read_side"] >> 1).astype(np.int64)
    keep = np.isin(sread, idx)
    return dict(idx=idx, tid=rec.tid[idx], pos=rec.pos[idx], mtid=rec.mtid[idx], mpos=rec.mpos[idx], flag=rec.flag[idx], mapq=rec.mapq[idx],
                l_seq=rec.l_seq[idx], clen=clen, cigar=cig, qnames=qn, whole=whole[idx],
                soft_read=sread[keep], soft_side=side[keep], soft_first=soft["res_first"][keep], soft_after=soft["res_after"][keep],
                soft_len=soft["seg_len"][keep])


def exchange_hot_groups(rec, whole, soft, global_offset, group=None):
    """Steps 1+2 of the module docstring.  rec/whole/soft are this rank's shard and its scorer outputs (soft sorted by
    read_side, local indices).  Returns (merged RecordBatch in global file order, whole words, soft records re-indexed
    to the merged batch, global index of every merged record)."""
    world = dist.get_world_size(group)
    n = rec.n
    h = api.qname_hash(rec)
    hot = (whole >> 16) != 0
    if soft.size:
        nz = ((soft["res_first"] >> 16) != 0) | ((soft["res_after"] >> 16) != 0)
        hot[(soft["read_side"][nz] >> 1).astype(np.int64)] = True
    gathered = [None] * world
    dist.all_gather_object(gathered, np.unique(h[hot]), group=group)
    hot_all = np.unique(np.concatenate(gathered)) if gathered else np.zeros(0, np.uint64)
    primary = (rec.flag & 0x900) == 0
    idx = np.nonzero(np.isin(h, hot_all) & primary)[0]
    payload = _compact(rec, idx, whole, soft)
    payload["goff"] = int(global_offset)
    parts = [None] * world
    dist.all_gather_object(parts, payload, group=group)
    parts.sort(key=lambda p: p["goff"])
    # ---- merge in global order ----
    gidx = np.concatenate([p["idx"] + p["goff"] for p in parts]).astype(np.int64)
    cat = lambda k, dt: np.concatenate([np.asarray(p[k], dt) for p in parts])
    clen = cat("clen", np.int64)
    cig_off = np.zeros(gidx.size + 1, np.uint32)
    cig_off[1:] = np.cumsum(clen)
    qn = [q for p in parts for q in p["qnames"]]
    qoff = np.zeros(gidx.size + 1, np.uint64)
    qoff[1:] = np.cumsum([len(q) for q in qn])
    merged = RecordBatch(cat("tid", np.int32), cat("pos", np.int32), cat("mtid", np.int32), cat("mpos", np.int32), cat("flag", np.uint16),
                         cat("mapq", np.uint8), cig_off, cat("cigar", np.uint32), np.zeros(gidx.size, np.uint64), cat("l_seq", np.int32),
                         np.zeros(64, np.uint8), qoff, b"".join(qn), None, rec.targets)
    whole_m = cat("whole", np.uint32)
    sread = np.concatenate([p["soft_read"] + p["goff"] for p in parts]).astype(np.int64)
    pos_of = np.searchsorted(gidx, sread)
    soft_m = np.zeros(sread.size, api.SOFT_DTYPE)
    soft_m["read_side"] = (pos_of.astype(np.uint32) << 1) | cat("soft_side", np.uint32)
    soft_m["res_first"], soft_m["res_after"], soft_m["seg_len"] = cat("soft_first", np.uint32), cat("soft_after", np.uint32), cat("soft_len", np.uint32)
    soft_m.sort(order="read_side")
    return merged, whole_m, soft_m, gidx


def extract_sharded(score_fn, rec, global_offset, n_total, tail_start, opts, group=None):
    """One rank's part of a multi-GPU extract.
    score_fn(rec_shard) -> (whole, soft)   (api.Context.score_reads on the GPU box)
    tail_start = global index of the first record of the unplaced tail the reference visits twice (extract.nim:326).
    Returns the treads of the WHOLE input (identical on every rank), qname_id = global record index."""
    whole, soft = score_fn(rec)[:2]
    merged, whole_m, soft_m, gidx = exchange_hot_groups(rec, whole, soft, global_offset, group)
    n_tail = int((gidx >= tail_start).sum())
    t = api.pair_reads(merged, opts, whole_m, soft_m, n_tail=n_tail)
    t["qname_id"] = gidx[t["qname_id"]]
    return t


def group_owner(treads, world):
    """rank that clusters each tread's (tid, unit) group: any function of the key works, groups never interact"""
    rep = np.ascontiguousarray(treads["repeat"]).view(np.uint8).reshape(-1, 6).astype(np.uint64)
    h = treads["tid"].astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    for j in range(6):
        h = (h ^ rep[:, j]) * np.uint64(0x100000001B3)
    return ((h >> np.uint64(17)) % np.uint64(world)).astype(np.int64)


def cluster_sharded(cluster_fn, treads_local, mode, group=None):
    """Multi-GPU clustering: all-gather the compact tread arrays (32 B per STR read), every rank
    clusters the (tid, unit) groups it owns, the rows are gathered and put back into the reference's row order.
    cluster_fn(treads) -> (bounds, unplaced)   (api.Context.cluster on the GPU box; rows of one group in position order)
    treads_local: this rank's treads; ranks are concatenated in rank order (= sample order for merge, file order for call).
    Returns (bounds, unplaced) of the whole input, identical on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    parts = [None] * world
    dist.all_gather_object(parts, np.ascontiguousarray(treads_local), group=group)
    all_t = np.concatenate(parts)
    mine = group_owner(all_t, world) == rank
    b, u = cluster_fn(all_t[mine])[:2]
    rows = [None] * world
    dist.all_gather_object(rows, (b, u), group=group)
    order = {k: i for i, k in enumerate(api.group_order(all_t, mode))}
    bs = np.concatenate([r[0] for r in rows])
    us = np.concatenate([r[1] for r in rows])
    # numpy hands out the NUL-padded unit fields without their padding, on both sides
    key_b = np.array([order[(int(x["tid"]), bytes(x["repeat"]))] for x in bs], np.int64) if len(bs) else np.zeros(0, np.int64)
    key_u = np.array([order[(-1, bytes(x["repeat"]))] for x in us], np.int64) if len(us) else np.zeros(0, np.int64)
    return bs[np.argsort(key_b, kind="stable")], us[np.argsort(key_u, kind="stable")]


class _DevArray:
    """a device buffer of the C ABI as something torch.as_tensor can wrap (zero copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class DeviceClusterExchange:
    """The multi-GPU clustering step on DEVICE buffers: every rank's resident treads (what
    strl_extract_device left in its context) are all-gathered as padded device tensors -- RCCL over xGMI with the "nccl"
    backend; with "gloo" (tests, dry runs) the same tensors are staged through the host, since gloo has no device all-gather --
    and every rank clusters the (tid, unit) groups it owns: the owner filter, the order-preserving compaction and the
    clustering all run on the device (strl_cluster_gathered).  The same object serves bench.py and tests/test_dist.py."""

    def __init__(self, ctx, world, rank, n_treads_hint, dev, group=None):
        import torch
        self.ctx, self.world, self.rank, self.group, self.dev = ctx, world, rank, group, dev
        pad = torch.tensor([int(n_treads_hint * 1.25) + 4096], dtype=torch.int64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(pad, op=dist.ReduceOp.MAX, group=group)          # ranks hold different samples: pad to the largest
        self.pad = int(pad.item())
        # One set of exchange buffers per stream the context runs a batch's tail on (an overlapped extract alternates between
        # side streams; the buffers of a step must outlive the asynchronous clustering that reads them).
        self._sets = {}
        self._use(ctx.stream)

    def _new_set(self, handle):
        import torch
        st = dict(t_local=torch.zeros(self.pad * 32, dtype=torch.uint8, device=self.dev),
                  t_all=torch.zeros(self.world * self.pad * 32, dtype=torch.uint8, device=self.dev),
                  c_all=torch.zeros(self.world, dtype=torch.int32, device=self.dev),
                  stream=torch.cuda.ExternalStream(handle))                 # the context's own stream: kernels and collectives in order
        self._sets[handle] = st
        return st

    def _use(self, handle):
        st = self._sets.get(handle) or self._new_set(handle)
        self.t_local, self.t_all, self.c_all, self.stream = st["t_local"], st["t_all"], st["c_all"], st["stream"]

    def gather(self):
        import torch
        ptr, cap, cnt = self.ctx.treads_device()                             # (orders the treads on the tail's stream)
        self._use(self.ctx.tail_stream() or self.ctx.stream)
        src = torch.as_tensor(_DevArray(ptr, cap * 32), device=self.dev)
        c_local = torch.as_tensor(_DevArray(cnt, 4), device=self.dev).view(torch.int32)
        with torch.cuda.stream(self.stream):
            m = min(self.pad, cap) * 32
            self.t_local[:m].copy_(src[:m])
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(self.t_all, self.t_local, group=self.group)
                dist.all_gather_into_tensor(self.c_all, c_local, group=self.group)
            else:                                                            # gloo: no device all-gather -- stage through the host
                th, ch = self.t_local.cpu(), c_local.cpu()
                ta, ca = torch.empty(self.world * th.numel(), dtype=torch.uint8), torch.empty(self.world, dtype=torch.int32)
                dist.all_gather_into_tensor(ta, th, group=self.group)
                dist.all_gather_into_tensor(ca, ch, group=self.group)
                self.t_all.copy_(ta)
                self.c_all.copy_(ca)

    def step(self, n_tid, window, min_support, max_clip_dist, pos_bits=0, fetch=False):
        """gather + cluster my share; fetch=False: everything stays enqueued on the context stream"""
        self.gather()
        return self.ctx.cluster_gathered(self.t_all.data_ptr(), self.c_all.data_ptr(), self.world, self.pad, self.rank, n_tid, window,
                                         min_support=min_support, max_clip_dist=max_clip_dist, pos_bits=pos_bits, fetch=fetch)

    def gathered_treads(self):
        """all ranks' treads in global (rank, .bin) order as a host array (for the final row order, strl_group_order)"""
        import torch
        torch.cuda.synchronize()
        c = self.c_all.cpu().numpy()
        raw = self.t_all.cpu().numpy().view(api.TREAD_DTYPE)
        return np.concatenate([raw[r * self.pad: r * self.pad + int(c[r])] for r in range(self.world)])


class NativeClusterExchange:
    """The same step with the collective INSIDE the library (comm.hip): strl_cluster_exchange = .bin-order sort, ncclAllGather of
    the padded tread buffers on the tail's stream, strl_cluster_gathered.  torch.distributed only carries the 128-byte
    communicator id to the ranks and agrees on the padding.  Interface of DeviceClusterExchange."""

    def __init__(self, ctx, world, rank, n_treads_hint, dev, group=None):
        import torch
        self.ctx, self.world, self.rank, self.group, self.dev = ctx, world, rank, group, dev
        on_dev = dist.get_backend(group) == "nccl"
        pad = torch.tensor([int(n_treads_hint * 1.25) + 4096], dtype=torch.int64, device=dev if on_dev else "cpu")
        dist.all_reduce(pad, op=dist.ReduceOp.MAX, group=group)
        self.pad = int(pad.item())
        box = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        ctx.comm_init(world, rank, box[0])

    def step(self, n_tid, window, min_support, max_clip_dist, pos_bits=0, fetch=False):
        return self.ctx.cluster_exchange(self.pad, n_tid, window, min_support=min_support, max_clip_dist=max_clip_dist, pos_bits=pos_bits, fetch=fetch)

    def gathered_treads(self):
        return self.ctx.exchange_treads()


def cluster_sharded_device(ex, mode_call_args, n_tid):
    """Whole multi-GPU clustering through a DeviceClusterExchange: my share on the device, then the rows of all ranks in the
    reference's row order (identical on every rank).  mode_call_args = dict(window=, min_support=, max_clip_dist=, pos_bits=)."""
    b, u, _ = ex.step(n_tid, mode_call_args["window"], mode_call_args["min_support"], mode_call_args["max_clip_dist"],
                      mode_call_args.get("pos_bits", 0), fetch=True)
    rows = [None] * ex.world
    dist.all_gather_object(rows, (b, u), group=ex.group)
    all_t = ex.gathered_treads()
    order = {k: i for i, k in enumerate(api.group_order(all_t, api.MODE_CALL))}
    bs = np.concatenate([r[0] for r in rows])
    us = np.concatenate([r[1] for r in rows])
    key_b = np.array([order[(int(x["tid"]), bytes(x["repeat"]))] for x in bs], np.int64) if len(bs) else np.zeros(0, np.int64)
    key_u = np.array([order[(-1, bytes(x["repeat"]))] for x in us], np.int64) if len(us) else np.zeros(0, np.int64)
    return bs[np.argsort(key_b, kind="stable")], us[np.argsort(key_u, kind="stable")]
