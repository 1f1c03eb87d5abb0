#!/usr/bin/env python
"""bench.py -- reads/s through the extract + cluster hot path on N MI355X (one process per GPU).

A "step" = one pass of the WHOLE device hot path over one HBM-resident batch of 2^25 DISTINCT synthetic 150 bp
paired-end WGS records (input S1):
    classify (skip predicate) -> scorer (whole reads) -> soft-clip scan          strl_extract_device, scoring half
    -> pair logic on the device (Bloom mark, probe, hash join, Cache.add replay, .bin order)      ... pairing half
    -> clustering of the treads THAT STEP produced (keys, radix sort, sweep, bounds)               strl_cluster_resident
Nothing crosses to the host inside a step and no stage is replayed from precomputed data.  Reads shard by record, so with
N > 1 every rank runs the extract half on its own batch; before clustering the ranks all-gather their treads (RCCL) and
every rank clusters the (tid, unit) groups it owns (strling_amd/dist.py).

Prints ONE JSON line on rank 0: value = reads of ALL ranks / max-rank time; `roofline` for the slowest kernel launch
(HIP events on the kernels' own stream); `cpu_baseline` = the oracle's extract + cluster over the same stages, 1 thread;
`end_to_end` = the `strling extract` -> `strling call` / `strling merge` processes on a BAM FILE on this box (zlib level 6, qualities,
aux tags, .bai), whole-process wall clock, with a share of the .bin and of -bounds.txt checked against the oracle.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# profiles/<round>/traffic.json, pmc_counters.json: the PMC passes the roofline block quotes -- this round's when they were collected
PROFILE_ROUND = next((r for r in ("r06", "r05") if os.path.exists(os.path.join(ROOT, "profiles", r, "traffic.json"))), "r06")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads-per-gpu", type=int, default=2 ** 25, help="reads resident per GPU (2^24 pairs, SURVEY S1)")
    ap.add_argument("--chunks", type=int, default=64, help="independent synthetic sub-samples the batch is generated from (in parallel)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-pairs", type=int, default=0, help="pairs of the BAM file the end-to-end leg runs `strling extract` / `call` / `merge` on.  0 = choose: 2^28 pairs (5.4e8 reads, "
                    "the 30x sample BASELINE.json names; ~8 min to write at zlib level 6 on 16 cores) when the box has the room (>= 75 GB free in the work directory, >= 12 CPUs) and N = 1, "
                    "else 2^26 pairs (1.3e8 reads, ~2 min to write); the line's end_to_end.input says which ran")
    ap.add_argument("--cache", default="", help="directory to keep the generated batch in (profiling runs reload it instead of forking generators)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: become N ranks (one per GPU) under torch.distributed.run, the launch the
        # driver's own N > 1 command makes.  exec: the ranks' single JSON line is this process' output.
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- end to end first (it forks generators and runs the CLI, which wants the GPU to itself): BAM file -> .bin -----
    e2e = None
    if rank == 0 and not args.no_e2e:
        # N > 1: rank 0 runs `strling extract --gpus N` (ONE process, a context per device, a contiguous share of the file each)
        # on all N devices while the other ranks wait at the rendezvous below -- they have not touched their device yet
        try:
            e2e = end_to_end(args.e2e_pairs or pick_e2e_pairs(world), gpus=world)
        except Exception as e:
            e2e = {"error": str(e)[:300]}

    # When this run's own end_to_end leg was NOT the whole-genome sized file (no room / N > 1), the builder's last run of that size
    # is quoted -- under a key that says it is a quotation, not a measurement of this run
    e2e_full = None
    if rank == 0 and world == 1 and not (e2e and e2e.get("reads", 0) >= 5e8):
        for rnd in ("r05", "r04"):
            try:
                e2e_full = json.load(open(os.path.join(ROOT, "profiles", rnd, "e2e_full.json")))
                e2e_full = {k: e2e_full[k] for k in e2e_full if k not in ("note",)}
                e2e_full["from_committed_profile"] = f"profiles/{rnd}/e2e_full.json (python tools/e2e_bench.py 268435456 --check-slabs 64 --repeats 2 on the GPU box; ~8 min to write the file): NOT measured in this run"
                break
            except Exception:
                e2e_full = None

    # ---- synthetic S1 batch: DISTINCT reads, generated before the GPU runtime starts (worker processes fork) ----
    from strling_amd import synth
    t_gen = time.perf_counter()
    n_chunks = max(1, min(args.chunks, args.reads_per_gpu // 2048))
    pairs_per_chunk = max(1, args.reads_per_gpu // 2 // n_chunks)
    cpus = _cpu_quota() or (os.cpu_count() or 2)       # the container's CPU quota, not the box's thread count: N ranks share it
    procs = max(1, min(n_chunks, int(cpus * 1.5) // max(1, world)))
    cache = os.path.join(args.cache, f"s1x30_{args.reads_per_gpu}_{n_chunks}_{rank}.pkl") if args.cache else ""
    if cache and os.path.exists(cache):
        import pickle
        rec, g = pickle.load(open(cache, "rb"))
    else:
        rec, g = synth.synth_wgs_30x(n_chunks, pairs_per_chunk, seed=1234 + 1000 * rank, procs=procs)
        if cache:
            import pickle
            pickle.dump((rec, g), open(cache, "wb"), protocol=4)
    t_gen = time.perf_counter() - t_gen

    import torch
    import torch.distributed as dist
    from strling_amd import api

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    shared_device = world > n_dev                   # more ranks than GPUs (a 2-rank dry run on a 1-GPU box): ranks share devices
    local = local % n_dev
    torch.cuda.set_device(local)
    backend = None
    if world > 1:
        import datetime
        # (rank 0 arrives late: it has run the end-to-end leg on all devices first)
        # nccl == RCCL on ROCm.  RCCL refuses two ranks on one device, so ranks that share a device rendezvous over gloo and
        # exchange through the host: a dry run of the N > 1 code path, not a measurement of the collective
        backend = os.environ.get("BENCH_BACKEND", "gloo" if shared_device else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=40))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=40))
    dev = torch.device("cuda", local)

    soa = api.Soa(rec)
    n = soa.n
    L = int(soa.max_l_seq)
    rows, qh = soa.pair_rows()
    n_tail = int((rec.tid < 0).sum())
    n_tid = len(rec.targets)
    frag = synth.frag_hist(rec)
    med = api.frag_median(frag)
    window = api.frag_median(frag, 0.99)                     # call.nim:114
    max_clip_dist = int(0.5 * api.frag_median(frag, 0.5))    # call.nim:232
    pos_bits = max(int(max(ln for _, ln in rec.targets)) + 8192, 2).bit_length() + 1   # + the half that holds adjust_by's wrapped positions

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    d = dict(tid=up(soa.tid), pos=up(soa.pos), end=up(soa.end), seq_off=up(soa.seq_off.view(np.int32)), l_seq=up(soa.l_seq.view(np.int16)),
             clip_l=up(soa.clip_l.view(np.int16)), clip_r=up(soa.clip_r.view(np.int16)), mapq=up(soa.mapq), cig=up(soa.cig),
             seq4=up(soa.seq4), rows=up(rows.view(np.uint8)), qhash=up(qh.view(np.int64)), meta=up(soa.meta_rows().view(np.int32)))
    cs = api.CReadSoa(n, d["tid"].data_ptr(), d["pos"].data_ptr(), d["end"].data_ptr(), d["seq_off"].data_ptr(), d["l_seq"].data_ptr(),
                      d["clip_l"].data_ptr(), d["clip_r"].data_ptr(), d["mapq"].data_ptr(), d["cig"].data_ptr(), d["seq4"].data_ptr(),
                      d["seq4"].numel(), L, api.MEM_DEVICE, d["meta"].data_ptr())
    cp = api.CPairSoa(d["rows"].data_ptr(), d["qhash"].data_ptr())
    torch.cuda.synchronize()
    del soa, qh, rows
    rec.seq4 = None

    ctx = api.Context(local)
    ctx.set_opts(0.8, 40, med)      # reference defaults: -p 0.8 -q 40 (extract.nim:255-256)
    ctx.set_genome(g)
    item_cap, tread_cap = n // 8 + 65536, n // 16 + 65536

    # ---- one synchronous pass: unit counts of every stage, and the result of the step -------------------------------
    ctx.extract_device(cs, cp, n_tail, item_cap, tread_cap)
    bounds, unplaced, cst = ctx.cluster_resident(n_tid, window, min_support=5, max_clip_dist=max_clip_dist, pos_bits=pos_bits)
    treads, st = ctx.treads_fetch()
    n_treads = int(treads.size)

    def agree(flag):
        """min over the ranks of a host-side flag (a CPU tensor under gloo, a device tensor under nccl)"""
        t = torch.tensor([int(flag)], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    exchange = None
    if world > 1:
        from strling_amd import dist as sdist
        # the collective inside the library (comm.hip: ncclAllGather on the tail's stream, the entry point the CLI's --gpus N
        # uses too); STRL_TORCH_COMM=1, or ranks that share a device, keep torch.distributed's all_gather_into_tensor.
        # ncclCommInitRank is itself a collective: a rank that cannot take part would leave the others hanging inside it, so
        # the ranks agree on every precondition BEFORE anybody calls it (round-3 advisor finding).
        native_ok = agree(backend == "nccl" and not shared_device and not os.environ.get("STRL_TORCH_COMM") and hasattr(ctx, "comm_init"))
        for cls in ([sdist.NativeClusterExchange] if native_ok else []) + [sdist.DeviceClusterExchange]:
            try:
                exchange = cls(ctx, world, rank, n_treads, dev)
                exchange.step(n_tid * 1, window, 5, max_clip_dist, pos_bits)
                torch.cuda.synchronize()
                ok1 = 1
            except Exception as e:                    # keep the shard-only measurement if the collective is unavailable
                print(f"[bench] rank {rank}: {cls.__name__} unavailable: {e}", file=sys.stderr)
                exchange, ok1 = None, 0
            if agree(ok1) == 1:                       # all ranks must take the same path
                break
            exchange = None
        if agree(exchange is not None) == 0:          # all ranks must agree, or the collective would hang
            exchange = None

    def step():
        ctx.extract_device(cs, cp, n_tail, item_cap, tread_cap)
        if exchange is not None:
            exchange.step(n_tid, window, 5, max_clip_dist, pos_bits)
        else:
            ctx.cluster_resident(n_tid, window, min_support=5, max_clip_dist=max_clip_dist, pos_bits=pos_bits, fetch=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    gc.collect()
    gc.disable()                 # no collector pause inside the timed region (the loop allocates next to nothing)
    for _ in range(args.warmup):
        step()
    ctx.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    barrier()
    el = time.perf_counter() - t0
    gc.enable()
    # the last timed step's results, collected after the fact, are the synchronous pass' results (same batch every step):
    # the pipelined steps computed what the un-pipelined one did
    verified = None
    if exchange is None and args.steps > 0:
        b_last, u_last, _ = ctx.cluster_collect()
        t_last, _ = ctx.treads_fetch()
        verified = bool(np.array_equal(b_last, bounds) and np.array_equal(u_last, unplaced) and np.array_equal(t_last, treads))
        if not verified:
            raise SystemExit("bench.py: the pipelined steps did not reproduce the synchronous pass' treads / bounds")
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- per-launch times: a few instrumented steps with HIP events on the kernels' own stream ----------------------
    ctx.enable_timing(True)
    n_inst = 5
    pair_ms, cl_ms = {}, {}
    for _ in range(n_inst):
        ctx.extract_device(cs, cp, n_tail, item_cap, tread_cap)
        for k, v in ctx.pair_times().items():
            pair_ms[k] = pair_ms.get(k, 0.0) + v / n_inst
        ctx.cluster_resident(n_tid, window, min_support=5, max_clip_dist=max_clip_dist, pos_bits=pos_bits, fetch=False)
        for k, v in ctx.cluster_times().items():
            cl_ms[k] = cl_ms.get(k, 0.0) + v / n_inst
    detail, launches = ctx.kernel_times_detail()
    ctx.enable_timing(False)
    launches = max(1, launches)
    ms = {k: v / launches for k, v in detail.items()}
    ms.update(pair_ms)
    ms.update(cl_ms)

    # ---- roofline of the slowest launch -----------------------------------------------------------------------------
    # Algorithmic bytes per launch (DESIGN.md "Kernels"):
    #   classify   13 B of coordinates read + 4 B result word written per read, 4 B id written per kept read
    #   stage A    4 B id + 16 B row gathered + ceil(L/2) B SEQ read, 16 B queue entry + 4 B result written per item
    #   stage B    32 B item + ceil(L/2) B SEQ + 4 B result per item that reaches k = 5
    #   segments   16 B item + the clipped bases (counted as ceil(L/4) B on average) + 16 B result record
    #   compaction 16 B state read per slot + 32 B item written per survivor; soft items: 1 B flag per slot, 16 B entry
    #              read + 16 B item written per clipped end
    #   pair mark + probe  16 B per soft-clip record + 8 B qname hash per read (the one full pass the pair logic adds)
    #                      + 12 B written per join item
    #   join sort / order sort  (12 B read + 12 B written) per item per 8-bit pass (4 passes each)
    #   replay     12 B item + ~40 B of metadata gathered per read item + 40 B per emitted tread
    #   cluster    32 B tread read + 25 B keys/payload written, 6 sort passes x 24 B, 41 B gathered + written per tread, sweep 16 B
    seq_b = (L + 1) // 2
    nbw, nbs = int(st.n_stage_b_whole), int(st.n_stage_b_soft)
    n_items = int(min(item_cap, 2.3 * n_treads + st.n_soft_items * 0.4))
    key_passes = (pos_bits + max(n_tid, 1).bit_length() + 15 + 7) // 8
    alg = {
        "classify_kernel": 17.0 * n + 4.0 * st.n_scored,          # 13 B read + 4 B written per read, 4 B id per kept read
        "score_kernel<whole,A>": (40.0 + seq_b) * st.n_scored,     # 4 B id + 16 B row gathered + SEQ + 16 B entry + 4 B result written
        "compact_kernel<whole>": 16.0 * st.n_scored + 32.0 * nbw,
        "score_kernel<whole,B>": (36.0 + seq_b) * nbw,
        "soft_compact_kernel": 1.0 * st.n_scored + 32.0 * st.n_soft_items,
        "score_kernel<segment,A>": (32.0 + (L + 3) // 4) * st.n_soft_items,
        "compact_kernel<segment>": 16.0 * st.n_soft_items + 32.0 * nbs,
        "score_kernel<segment,B>": (48.0 + (L + 3) // 4) * nbs,
        "pair_soft_items_kernel": 16.0 * st.n_soft_items + 0.4 * 20.0 * st.n_soft_items,
        "pair_probe_kernel": 8.0 * n + 20.0 * n_items,
        "pair_join_sort": 4 * 24.0 * n_items,
        "pair_groups_kernel": 52.0 * n_items + 44.0 * n_treads,
        "pair_order": 0.0,   # the .bin-order sort of the treads is deferred to a fetch: clustering works from the emission keys
        "cluster_keys_sort_groups": (57.0 + key_passes * 24.0 + 41.0) * n_treads,
        "cluster_sweep": 16.0 * n_treads,
        "cluster_bounds": 8.0 * n_treads + 44.0 * len(bounds),
    }
    if L <= 160 and not os.environ.get("STRL_SPLIT_SEGMENTS"):
        # segments of the short-read class: stage A and B in ONE launch (no hand-over, no compaction, one fetch of the bases);
        # the library reports it in stage A's slot, the two slots behind it hold nothing but the gap between two events
        ms["score_kernel<segment,A+B>"] = ms.pop("score_kernel<segment,A>") + ms.pop("compact_kernel<segment>") + ms.pop("score_kernel<segment,B>")
        alg["score_kernel<segment,A+B>"] = alg.pop("score_kernel<segment,A>")      # 16 B item + the clipped bases + 16 B result record
        alg.pop("compact_kernel<segment>"); alg.pop("score_kernel<segment,B>")
    grouped = ("pair_join_sort", "pair_order", "cluster_keys_sort_groups", "cluster_sweep", "cluster_bounds")   # several launches each
    kernels = {k: (ms[k], alg[k]) for k in ms if k not in grouped}
    groups = {"classify_kernel": ms["classify_kernel"],
              "score_kernel<whole>": ms["score_kernel<whole,A>"] + ms["compact_kernel<whole>"] + ms["score_kernel<whole,B>"],
              "score_kernel<soft>": ms["soft_compact_kernel"] + sum(v for k, v in ms.items() if "segment" in k),
              "pair_logic": sum(pair_ms.values()), "cluster_pass": sum(cl_ms.values())}
    groups.update({k: ms[k] for k in grouped})
    dom = max(kernels, key=lambda k: kernels[k][0])     # slowest single launch
    dom_ms, dom_bytes = kernels[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    tj, pick = {}, {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic.json")))["kernels"]
        pick = {"classify_kernel": "classify_kernel", "score_kernel<whole,A>": "<10, 64, 0, 0", "score_kernel<segment,A>": "<10, 64, 1, 0", "score_kernel<segment,A+B>": "<10, 64, 1, 2",
                "pair_probe_kernel": "pair_probe_kernel", "pair_groups_kernel": "pair_groups_kernel"}
        if dom in pick and n == 2 ** 25 and L == 150:
            traffic = sum(v["hbm_bytes_corrected"] for name, v in tj.items() if pick[dom] in name) or None
    except Exception:
        traffic = None
    # The slowest launch may be one that HBM does not bound at all (the scorer stages issue integer VALU instructions ~88 % of
    # the time): report its VALU issue utilisation from the committed PMC pass next to the HBM figure, and the slowest launch
    # that IS bound by HBM traffic (the skip-predicate pass) separately.
    valu_frac = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_counters.json")))
        if dom in pick and n == 2 ** 25 and L == 150:
            insts = [v["SQ_INSTS_VALU"] for name, v in pj.items() if pick[dom] in name and "SQ_INSTS_VALU" in v]
            if insts and dom_ms > 0:      # a wave64 VALU instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
                valu_frac = round(insts[0] * 4.0 / (dom_ms * 1e-3 * 2.4e9 * 1024), 3)
    except Exception:
        valu_frac = None
    c_ms, c_bytes = kernels["classify_kernel"]
    c_traffic = None
    try:
        if n == 2 ** 25 and L == 150:
            c_traffic = sum(v["hbm_bytes_corrected"] for name, v in tj.items() if "classify_kernel" in name) or None
    except Exception:
        c_traffic = None
    c_ach = c_bytes / (c_ms * 1e-3) / 1e9 if c_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "traffic_note": f"HBM bytes per launch of that kernel from profiles/{PROFILE_ROUND}/traffic.json (rocprofv3 PMC passes of the same workload and build, committed; not measured in this run)", "from_committed_profile": True,
                "valu_issue_frac": valu_frac,
                "valu_note": f"SQ_INSTS_VALU (profiles/{PROFILE_ROUND}/pmc_counters.json, committed) x 4 cycles / (launch time x 1024 SIMDs x 2.4 GHz): the scorer launches are bound by integer VALU issue, not by HBM",
                "hbm_bound_launch": {"kernel": "classify_kernel", "achieved": round(c_ach, 2), "frac": round(c_ach / HBM_PEAK_GBPS, 5), "traffic": c_traffic,
                                     "note": "slowest launch that HBM traffic bounds (13 B read + 4 B written per read, algorithmic)"},
                "times_note": "kernel_ms / group_ms: un-overlapped launch times from instrumented steps (HIP events, one stream); ms_per_step is the "
                              "pipelined period: pair logic + clustering of step i run on side streams beside classify + scorer of step i + 1 and the tail of step i - 1",
                "kernel_ms": {k: round(v[0], 4) for k, v in kernels.items()},
                "kernel_alg_bytes": {k: int(v[1]) for k, v in kernels.items()},
                "kernel_alg_GBps": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0.0 for k, v in kernels.items()},
                "group_ms": {k: round(v, 4) for k, v in groups.items()},
                "pipeline_alg_GBps": round(sum(alg.values()) / (el / args.steps) / 1e9, 2),
                "survey_115B_per_read_GBps": round(115.0 * n / (el / args.steps) / 1e9, 2)}

    # ---- CPU baseline: the oracle ("port" of the reference algorithm) over the SAME stages, 1 thread, bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        srec, sg = synth.synth_wgs(2 ** 17, seed=1234)
        sfrag = synth.frag_hist(srec)
        smed = O.median(sfrag)
        opts = O.make_opts(smed, 0.8, 40)
        reads_done, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds:
            t1 = time.perf_counter()
            et = O.extract(srec, sg, opts)
            O.call_bounds(et, 1, api.frag_median(sfrag, 0.99), min_support=5, max_clip_dist=int(0.5 * smed))
            t_cpu += time.perf_counter() - t1
            reads_done += srec.n
        cpu = {"value": round(reads_done / t_cpu, 1), "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": f"oracle extract loop (skip predicate + get_repeat + add_soft + pair logic) + cluster/bounds over the S1 mix, "
                         f"{srec.n} reads x {reads_done // srec.n} passes, {t_cpu:.1f} s, single thread like the reference (threads=0)"}

    # ---- ... the same on every core the box grants (process-sharded: the reference is single-threaded per sample, a pipeline
    #      runs one process per sample), and a decode-inclusive single-thread leg to put beside end_to_end ----
    cpu_nproc, cpu_e2e = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_nproc = cpu_baseline_nproc(max(4.0, args.cpu_seconds / 2))
            cpu_e2e = cpu_baseline_e2e(cpu["value"])
            for blk in (e2e, e2e_full):
                if blk and blk.get("reads_per_s_extract_plus_call") and cpu_e2e:
                    blk["vs_cpu_baseline_e2e_extract_plus_call"] = round(blk["reads_per_s_extract_plus_call"] / cpu_e2e["with_call"], 1)
                if blk and blk.get("value") and cpu_e2e and blk is not e2e:
                    blk["vs_cpu_baseline_e2e_wall"] = round(blk["value"] / cpu_e2e["value"], 1)
            if e2e and "runs" in e2e and cpu_e2e:
                e2e["vs_cpu_baseline_e2e_wall"] = round(e2e["value"] / cpu_e2e["value"], 1)
                best = max(e2e["runs"], key=lambda x: x["reads_per_s_wall"])
                if best.get("reads_per_s_loop"):
                    e2e["vs_cpu_baseline_e2e_loop"] = round(best["reads_per_s_loop"] / cpu_e2e["value"], 1)
        except Exception as e:
            cpu_nproc = cpu_nproc or {"error": str(e)[:200]}

    if rank == 0:
        total_reads = n * world * args.steps
        out = {
            "metric": "reads/sec through extract+cluster, 30x 150 bp WGS; 1/2/4/8 MI355X + CPU ref",
            "value": round(total_reads / el, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 integer", "data": "synthetic",
            "exchange": "none" if exchange is None else ("native" if type(exchange).__name__ == "NativeClusterExchange" else "torch"),
            "rccl_ranks": world if (exchange is not None and backend == "nccl") else 0, "backend": backend, "devices_visible": n_dev,
            "config": {"workload": f"{world}xMI355X: 30x 150 bp PE synthetic WGS, k=2-6 repeat-unit scorer + soft-clip scan + device pair logic + on-GPU radix-sort/segmented clustering (BASELINE.json configs[1]+[2])",
                       "reads_per_gpu": n, "read_len": L, "unique_reads_per_gpu": n, "tiles": 1, "generate_s": round(t_gen, 1),
                       "skipped_frac": round(st.n_skipped / n, 4), "scored_reads": int(st.n_scored), "soft_items": int(st.n_soft_items),
                       "str_reads_clustered": n_treads, "clusters": int(cst.n_clusters), "bounds": int(len(bounds)),
                       "last_step_equals_synchronous_pass": verified,
                       "timed_region": "every kernel of the path on HBM-resident records: classify + score + soft-clip scan, the pair logic (Cache.add) on the device, "
                                       "then keys + radix sort + sweep + bounds over the treads the same step produced; consecutive steps are pipelined on the "
                                       "context's streams (side streams run pair logic + clustering of a step while the next step's scorer runs); BAM decode, PCIe, the host-side row order "
                                       "(Nim table order) and file writing are in end_to_end, not here",
                       "parallelism": (f"records sharded over {world} GPU(s), no data-path collective" if exchange is None else
                                       f"records sharded over {world} GPUs; per step one RCCL all-gather of the tread arrays [{type(exchange).__name__}] "
                                       f"({exchange.pad * 32} B per rank) before clustering, every rank clusters the (tid, unit) groups it owns")},
            "build": build_info(),
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_nproc": cpu_nproc, "cpu_baseline_e2e": cpu_e2e, "end_to_end": e2e, "quoted_not_measured_end_to_end_full_size": e2e_full,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _cpu_quota():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else int(q) / int(per)
    except Exception:
        return None


def _nproc_worker(args):
    seed, seconds = args
    from oracle import oracle as O
    from strling_amd import api, synth
    srec, sg = synth.synth_wgs(2 ** 16, seed=seed)
    sfrag = synth.frag_hist(srec)
    smed = O.median(sfrag)
    opts = O.make_opts(smed, 0.8, 40)
    done, t = 0, 0.0
    t0 = time.perf_counter()
    while t < seconds:
        et = O.extract(srec, sg, opts)
        O.call_bounds(et, 1, api.frag_median(sfrag, 0.99), min_support=5, max_clip_dist=int(0.5 * smed))
        done += srec.n
        t = time.perf_counter() - t0
    return done, t


def cpu_baseline_nproc(seconds):
    """the oracle's extract + cluster loop in P processes at once (one sample each, like a pipeline runs the single-threaded
    reference), P = the CPUs this container may use"""
    import multiprocessing as mp
    quota = _cpu_quota()
    hw = os.cpu_count() or 1
    procs = max(1, min(hw, int(quota) if quota else hw, 64))
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_nproc_worker, [(4321 + k, seconds) for k in range(procs)])
    reads = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {"value": round(reads / wall, 1), "unit": "reads/s", "cores": procs, "kind": "port", "host_threads_visible": hw, "cgroup_cpu_quota": quota,
            "sample": f"{procs} processes, each the oracle's extract + cluster loop over its own 2^17-read S1 sample for {seconds:.0f} s (decode excluded, "
                      f"like cpu_baseline); aggregate reads / slowest process' time"}


def cpu_baseline_e2e(oracle_reads_per_s):
    """what the reference's threads=0 run does per read, on one core: inflate the BGZF blocks with zlib, then the extract loop.
    Inflate is timed on the blocks of a synthetic BAM written like end_to_end's (zlib level 6, binned qualities, aux tags), the
    loop rate is cpu_baseline's; record parsing (htslib bam_read1) is not charged.  `with_call` adds what `strling call` does on
    the .bin per read of the file (the oracle's call over an in-memory slab: its region reads cost no inflate, the reference's do)."""
    import zlib
    from strling_amd import bamio
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import inflate_bench
    d = os.environ.get("TMPDIR", "/tmp")
    path = os.path.join(d, "cpu_e2e_sample.bam")
    info = bamio.write_bam_slabs(path, 1, 2 ** 16, seed=77, level=6, quals=True, aux=True, index=False, procs=1)
    streams, sizes = inflate_bench.bam_blocks(path)
    os.remove(path)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3.0:
        for s_ in streams:
            zlib.decompress(s_, -15)
        n += 1
    t_inf = (time.perf_counter() - t0) / n
    inflate_rps = info["reads"] / t_inf
    value = 1.0 / (1.0 / inflate_rps + 1.0 / oracle_reads_per_s)
    # `strling call` on the CPU: cluster + evidence + genotypes over the slab's own treads
    rec, g = bamio.slab_records(0, 1, 2 ** 16, 77)
    from strling_amd import synth
    frag = synth.frag_hist(rec)
    t = O.extract(rec, g, O.make_opts(O.median(frag), 0.8, 40))
    t1 = time.perf_counter()
    k = 0
    while time.perf_counter() - t1 < 2.0:
        O.call(t, rec, frag)
        k += 1
    call_s_per_read = (time.perf_counter() - t1) / k / rec.n
    with_call = 1.0 / (1.0 / value + call_s_per_read)
    return {"value": round(value, 1), "unit": "reads/s", "cores": 1, "kind": "port", "with_call": round(with_call, 1),
            "parts": {"zlib_inflate_reads_per_s": round(inflate_rps, 1), "zlib_inflate_GBps": round(sum(sizes) / t_inf / 1e9, 3), "extract_loop_reads_per_s": oracle_reads_per_s,
                      "call_s_per_read": call_s_per_read},
            "sample": f"single thread: zlib inflate of the {len(streams)} BGZF blocks of a {info['reads']}-read synthetic BAM (zlib level 6, binned qualities, aux tags: the "
                      f"end_to_end writer) + cpu_baseline's extract/cluster rate, combined per read (1 / (1/inflate + 1/loop)); the like-for-like denominator of "
                      f"end_to_end.  with_call: + the oracle's `call` (cluster, spanning evidence, genotypes) per read of the file"}


def pick_e2e_pairs(world):
    """2^28 pairs (5.4e8 reads: the 30x sample of BASELINE.json's metric, a 57 GB BAM) when this is the one-GPU run and the box
    has the room and the cores to write it in ~8 minutes; else 2^26 pairs"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_bench
    if os.environ.get("BENCH_E2E_SMALL"):
        return 2 ** 26
    # (N > 1 takes the same full-size file: the 1.3e8-read one is start-up bound even on one GPU -- a curve over it would measure
    # process start, not the shares.  The file is cached in the work directory: a 1 / 2 / 4 / 8 sweep on one node writes it once.)
    cpus = _cpu_quota() or (os.cpu_count() or 1)
    d = e2e_bench.work_dir(2 ** 28 * 2 * 115)
    st = os.statvfs(d)
    return 2 ** 28 if (st.f_bavail * st.f_frsize >= 75e9 and cpus >= 12) else 2 ** 26


def end_to_end(n_pairs, check_slabs=0, gpus=1):
    """`strling extract` -> .bin -> `strling call` and `strling merge`, from a coordinate-sorted, indexed BAM of 2 * n_pairs distinct
    reads (zlib level 6, binned random qualities, aux tags) written to local disk / shm (page cache warm): wall clock of whole
    processes, all host threads; a share of the outputs checked against the oracle (tools/e2e_bench.py).  gpus > 1: the same
    with `--gpus N` (one process, a share of the file per device), the host feed alone, N concurrent per-sample replicas
    (`--device k`), and the strong scaling against the N = 1 figures kept beside the cached input."""
    from strling_amd import build
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_bench
    inp = e2e_bench.make_input(n_pairs)
    full = inp["reads"] >= 5e8
    check_slabs = check_slabs or (16 if full else 4)      # 16 of 1024 slabs at the headline size (~10 s), 4 of 256 below it
    try:
        res = e2e_bench.run(inp, build.CLI, repeats=3 if full else 2, gpus=gpus)      # the first process behind the writer reads a cold file
        res["input_reused_from_cache"] = bool(inp.get("reused_cached_input"))
        if "error" not in res:
            if gpus == 1:
                e2e_bench.n1_record(inp, res)
            else:
                n1 = e2e_bench.n1_record(inp)
                if n1 and n1.get("extract_s") and n1.get("reads") == inp["reads"]:
                    res["strong_scaling_vs_n1"] = {"n1_extract_s": n1["extract_s"], "wall": round(n1["extract_s"] / res["extract_s"], 3),
                                                   "loop": round(res["reads_per_s_loop"] / n1["reads_per_s_loop"], 3) if res.get("reads_per_s_loop") and n1.get("reads_per_s_loop") else None,
                                                   "note": "this run's `extract --gpus N` against the N = 1 run of the same cached file on this node (wall: whole processes; loop: inside the loop)"}
                try:
                    res["feed_only"] = e2e_bench.feed_only(inp, build.CLI, gpus)
                except Exception as e:
                    res["feed_only"] = {"error": str(e)[:200]}
                try:
                    res["replicas"] = e2e_bench.replicas(inp, build.CLI, gpus, _visible_devices())
                except Exception as e:
                    res["replicas"] = {"error": str(e)[:200]}
        if check_slabs and "error" not in res:
            res["check"] = e2e_bench.check_in_subprocess(inp, e2e_bench.pick_slabs(inp["n_slabs"], check_slabs), call=res.get("call_rc") == 0)
        return res
    finally:
        # the input stays cached for the next N of a sweep / the next run, unless BENCH_E2E_CLEAN=1
        e2e_bench.cleanup(inp, keep_input=not os.environ.get("BENCH_E2E_CLEAN"))


def _visible_devices():
    """GPUs this process may use, WITHOUT starting the HIP runtime in it (the generators fork later): rocm-smi / the env"""
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(k)
        if v:
            return max(1, len([x for x in v.split(",") if x.strip()]))
    try:
        return max(1, len([d for d in os.listdir("/sys/class/kfd/kfd/topology/nodes")
                           if int(open(f"/sys/class/kfd/kfd/topology/nodes/{d}/properties").read().split("simd_count")[1].split()[0]) > 0]))
    except Exception:
        return 1


def build_info():
    """what the line was measured on: the compiler that built the library, whether the shipped .so is newer than its sources"""
    from strling_amd import build as b
    info = {"lib": os.path.relpath(b.LIB, ROOT)}
    try:
        info["lib_mtime"] = int(os.path.getmtime(b.LIB))
        new = []
        for target, names in ((b.LIB, b.SOURCES + b.HEADERS), (b.CLI, b.CLI_SOURCES + b.HEADERS)):
            for s_ in names:
                src = os.path.join(b.CSRC, s_)
                if os.path.exists(src) and os.path.exists(target) and os.path.getmtime(src) > os.path.getmtime(target):
                    new.append(os.path.relpath(src, ROOT))
        info["sources_newer_than_their_binary"] = sorted(set(new))      # empty: build() had nothing to recompile
        info["with_inflate_group"] = b.WITH_INFLATE_GROUP
    except Exception as e:
        info["error"] = str(e)[:100]
    try:
        v = subprocess.run([b._hipcc(), "--version"], capture_output=True, text=True, timeout=30).stdout.splitlines()
        info["hipcc"] = "; ".join(l.strip() for l in v[:2])
    except Exception as e:
        info["hipcc"] = f"unavailable ({str(e)[:60]})"
    return info


if __name__ == "__main__":
    main()
