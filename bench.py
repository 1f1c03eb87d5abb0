#!/usr/bin/env python
"""bench.py -- reads/s through the extract hot path on N MI355X (one process per GPU).

A "step" = one pass of the device hot path over one HBM-resident batch of synthetic 150 bp paired-end WGS
records (input S1): the extract kernels (classify -> score -> soft-clip scan, the whole
`strl_score_reads` entry point) followed by the clustering pass (stable radix sorts -> sweep -> bounds,
`strl_cluster_replay`) over the STR reads such a batch yields.  Reads shard by record, so with N > 1 every rank scores
its own batch and no collective sits on the data path (weak scaling); the only collectives are
the barrier + MAX of the elapsed time the contract asks for.

Prints ONE JSON line on rank 0 (see the task contract): value = reads of ALL ranks / max-rank time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads-per-gpu", type=int, default=2 ** 25, help="reads resident per GPU (2^24 pairs, SURVEY S1)")
    ap.add_argument("--base-pairs", type=int, default=2 ** 18, help="unique synthetic pairs generated on the host, tiled in HBM")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from strling_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()       # (a 2-rank dry run on a 1-GPU box maps both ranks to cuda:0)
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get("BENCH_BACKEND", "nccl")   # nccl == RCCL on ROCm; gloo only for dry runs of the N > 1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)

    # ---- synthetic S1 base sample on the host, tiled into HBM -------------------------------------
    rec, g = synth.synth_wgs(args.base_pairs, seed=1234 + rank)
    soa = api.Soa(rec)
    n_base = soa.n
    tiles = max(1, args.reads_per_gpu // n_base)
    n = n_base * tiles
    stride16 = int(soa.seq_off[1] - soa.seq_off[0]) if n_base > 1 else 5
    seq_bytes_base = n_base * stride16 * 16

    def tile(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev).view(dt).repeat(tiles)

    d = dict(tid=tile(soa.tid, torch.int32), pos=tile(soa.pos, torch.int32), end=tile(soa.end, torch.int32),
             l_seq=tile(soa.l_seq.view(np.int16), torch.int16), clip_l=tile(soa.clip_l.view(np.int16), torch.int16),
             clip_r=tile(soa.clip_r.view(np.int16), torch.int16), mapq=tile(soa.mapq, torch.uint8), cig=tile(soa.cig, torch.uint8))
    so = torch.from_numpy(soa.seq_off.astype(np.int64)).to(dev)
    d["seq_off"] = (so[None, :] + (torch.arange(tiles, device=dev, dtype=torch.int64) * (seq_bytes_base // 16))[:, None]).reshape(-1).to(torch.int32)
    seq_base = torch.from_numpy(soa.seq4[:seq_bytes_base]).to(dev)
    d["seq4"] = torch.cat([seq_base.repeat(tiles), torch.zeros(64, dtype=torch.uint8, device=dev)])
    whole = torch.zeros(n, dtype=torch.int32, device=dev)
    soft_cap = max(1024, n // 8)
    soft = torch.zeros((soft_cap, 4), dtype=torch.int32, device=dev)
    cs = api.CReadSoa(n, d["tid"].data_ptr(), d["pos"].data_ptr(), d["end"].data_ptr(), d["seq_off"].data_ptr(), d["l_seq"].data_ptr(),
                      d["clip_l"].data_ptr(), d["clip_r"].data_ptr(), d["mapq"].data_ptr(), d["cig"].data_ptr(), d["seq4"].data_ptr(),
                      d["seq4"].numel(), soa.max_l_seq, api.MEM_DEVICE)
    torch.cuda.synchronize()

    ctx = api.Context(local)
    med = api.frag_median(synth.frag_hist(rec))
    ctx.set_opts(0.8, 40, med)      # reference defaults: -p 0.8 -q 40 (extract.nim:255-256)
    ctx.set_genome(g)

    # the STR reads of the batch: extract the unique sample through the full path (GPU scoring + host pair logic) and
    # give every tile its own contigs, as if the genome were `tiles` times larger (no artificial pile-ups)
    base_treads, _ = ctx.extract(rec)
    treads = np.tile(base_treads, tiles)
    n_contigs = len(rec.targets)
    placed = treads["tid"] >= 0
    treads["tid"] = np.where(placed, treads["tid"] + np.repeat(np.arange(tiles, dtype=np.int32) * n_contigs, base_treads.size), -1)
    frag = synth.frag_hist(rec)
    window = api.frag_median(frag, 0.99)                     # call.nim:114
    max_clip_dist = int(0.5 * api.frag_median(frag, 0.5))    # call.nim:232
    bounds, unplaced, cst = ctx.cluster(treads, api.MODE_CALL, window, min_support=5, max_clip_dist=max_clip_dist)

    # N > 1: the exchange step of SURVEY section 8(e) -- one RCCL all-gather of the compact tread arrays (32 B per STR read)
    # before clustering, after which every rank clusters an equal share of the (tid, unit) groups.  The synthetic ranks
    # hold equally sized tread sets, so the share a rank clusters is as large as its own set: the replayed pass.
    exchange = None
    cstream = torch.cuda.ExternalStream(ctx.stream)
    if world > 1:
        try:
            t_mine = torch.from_numpy(np.ascontiguousarray(treads).view(np.uint8).copy()).to(dev)
            n_max = torch.tensor([t_mine.numel()], dtype=torch.int64, device=dev)
            dist.all_reduce(n_max, op=dist.ReduceOp.MAX)      # ranks hold different samples: pad to the largest tread array
            t_local = torch.zeros(int(n_max.item()), dtype=torch.uint8, device=dev)
            t_local[:t_mine.numel()] = t_mine
            t_all = torch.empty(world * t_local.numel(), dtype=torch.uint8, device=dev)
            gather_done = torch.cuda.Event()

            def exchange():
                dist.all_gather_into_tensor(t_all, t_local)
                gather_done.record()
                cstream.wait_event(gather_done)      # the clustering kernels of this step start after the gather

            exchange()
            torch.cuda.synchronize()
        except Exception as e:                        # keep the shard-only measurement if the collective is unavailable
            print(f"[bench] rank {rank}: tread all-gather disabled: {e}", file=sys.stderr)
            exchange = None
        ok = torch.tensor([1 if exchange is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)     # all ranks must agree, or the collective would hang
        if int(ok.item()) == 0:
            exchange = None

    # one synchronous pass for the unit counts of each kernel
    n_soft, st = ctx.score_device(cs, whole.data_ptr(), soft.data_ptr(), soft_cap, sync=True)
    for _ in range(args.warmup):
        ctx.score_device(cs, whole.data_ptr(), soft.data_ptr(), soft_cap)
        if exchange:
            exchange()
        ctx.cluster_replay()
    ctx.sync()
    # clustering pass alone, HIP events on the context stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(cstream):
        ev0.record()
        for _ in range(5):
            ctx.cluster_replay()
        ev1.record()
    ctx.sync()
    ms_cluster = ev0.elapsed_time(ev1) / 5
    ctx.enable_timing(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.score_device(cs, whole.data_ptr(), soft.data_ptr(), soft_cap)
        if exchange:
            exchange()
        ctx.cluster_replay()
    ctx.sync()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    detail, launches = ctx.kernel_times_detail()
    ctx.enable_timing(False)

    # ---- roofline of the dominant kernel (HIP-event durations on the kernels' own stream) ----------
    # One entry per kernel LAUNCH of a scoring pass; algorithmic bytes per launch as in DESIGN.md "Kernels":
    #   classify   13 B of coordinates read + 4 B written (result word or queue slot) per read
    #   stage A    16 B queue entry + ceil(L/2) B SEQ read + 4 B result written per item
    #   stage B    32 B item + ceil(L/2) B SEQ + 4 B result per item that reaches k = 5
    #   segments   16 B item + the clipped bases (counted as ceil(L/4) B on average) + 16 B result record
    #   compaction 16 B state read per slot + 32 B item written per survivor; soft items: 1 B flag per slot, 16 B entry
    #              read + 16 B item written per clipped end
    launches = max(1, launches)
    L = int(soa.max_l_seq)
    seq_b = (L + 1) // 2
    ms = {k: v / launches for k, v in detail.items()}
    nbw, nbs = int(st.n_stage_b_whole), int(st.n_stage_b_soft)
    alg = {
        "classify_kernel": 17.0 * n,
        "score_kernel<whole,A>": (20.0 + seq_b) * st.n_scored,
        "compact_kernel<whole>": 16.0 * st.n_scored + 32.0 * nbw,
        "score_kernel<whole,B>": (36.0 + seq_b) * nbw,
        "soft_compact_kernel": 1.0 * st.n_scored + 32.0 * st.n_soft_items,
        "score_kernel<segment,A>": (32.0 + (L + 3) // 4) * st.n_soft_items,
        "compact_kernel<segment>": 16.0 * st.n_soft_items + 32.0 * nbs,
        "score_kernel<segment,B>": (48.0 + (L + 3) // 4) * nbs,
    }
    kernels = {k: (ms[k], alg[k]) for k in ms}
    groups = {"classify_kernel": ms["classify_kernel"],
              "score_kernel<whole>": ms["score_kernel<whole,A>"] + ms["compact_kernel<whole>"] + ms["score_kernel<whole,B>"],
              "score_kernel<soft>": ms["soft_compact_kernel"] + ms["score_kernel<segment,A>"] + ms["compact_kernel<segment>"] + ms["score_kernel<segment,B>"],
              # clustering: 36 small launches (merge sort of the composite key, scans, sweep, bounds); reported as one pass
              "cluster_pass": ms_cluster}
    cluster_alg = (2 * (4 + 4) * 2 + 2 * (8 + 4) * 2 + 24.0) * treads.size + 44.0 * len(bounds)
    # HBM traffic per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs of tools/prof_run.py, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if absent
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "f_traffic.json")))["kernels"]
        pick = {"classify_kernel": "classify_kernel", "score_kernel<whole,A>": "<10, 64, 0, 0", "compact_kernel<whole>": "compact_kernel<1, 0>",
                "score_kernel<whole,B>": "<10, 64, 0, 1", "soft_compact_kernel": "soft_compact_kernel", "score_kernel<segment,A>": "<10, 64, 1, 0",
                "compact_kernel<segment>": "compact_kernel<1, 1>", "score_kernel<segment,B>": "<10, 64, 1, 1"}
        for k, pat in pick.items():
            traffic[k] = sum(v["hbm_bytes_corrected"] for name, v in tj.items() if pat in name)
    except Exception:
        traffic = {}
    dom = max(kernels, key=lambda k: kernels[k][0])
    dom_ms, dom_bytes = kernels[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic.get(dom) if (n == 2 ** 25 and L == 150) else None,
                "traffic_note": "HBM bytes per launch of that kernel from profiles/r01/f_traffic.json (PMC, same workload)",
                "kernel_ms": {k: round(v[0], 4) for k, v in kernels.items()},
                "kernel_alg_bytes": {k: int(v[1]) for k, v in kernels.items()},
                "kernel_alg_GBps": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0.0 for k, v in kernels.items()},
                "group_ms": {k: round(v, 4) for k, v in groups.items()},
                "pipeline_alg_GBps": round((sum(v[1] for v in kernels.values()) + cluster_alg) / (el / args.steps) / 1e9, 2),
                "survey_115B_per_read_GBps": round(115.0 * n / (el / args.steps) / 1e9, 2)}

    # ---- CPU baseline: the oracle ("port" of the reference algorithm), 1 thread, bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        sample_pairs = min(args.base_pairs, 2 ** 17)
        srec, sg = synth.synth_wgs(sample_pairs, seed=1234)
        opts = O.make_opts(med, 0.8, 40)
        reads_done, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds:
            t1 = time.perf_counter()
            O.extract(srec, sg, opts)
            t_cpu += time.perf_counter() - t1
            reads_done += srec.n
        cpu = {"value": round(reads_done / t_cpu, 1), "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": f"oracle extract loop (skip predicate + get_repeat + add_soft + pair logic) over the S1 mix, "
                         f"{srec.n} reads x {reads_done // srec.n} passes, {t_cpu:.1f} s, single thread like the reference (threads=0)"}

    if rank == 0:
        total_reads = n * world * args.steps
        out = {
            "metric": "reads/sec through extract+cluster, 30x 150 bp WGS; 1/2/4/8 MI355X + CPU ref",
            "value": round(total_reads / el, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 integer", "data": "synthetic",
            "config": {"workload": f"{world}xMI355X: 30x 150 bp PE synthetic WGS, k=2-6 repeat-unit scorer + soft-clip scan + on-GPU radix-sort/segmented clustering (BASELINE.json configs[1]+[2])",
                       "reads_per_gpu": n, "read_len": L, "unique_reads_per_gpu": n_base, "tiles": tiles,
                       "skipped_frac": round(st.n_skipped / n, 4), "scored_reads": int(st.n_scored), "soft_items": int(st.n_soft_items),
                       "str_reads_clustered": int(treads.size), "clusters": int(cst.n_clusters), "bounds": int(len(bounds)),
                       "timed_region": "classify + score + soft-clip kernels, then radix-sort + sweep + bounds clustering kernels, all on HBM-resident data; BAM decode, PCIe and the host pair logic between the two are excluded",
                       "parallelism": (f"records sharded over {world} GPU(s), no data-path collective" if exchange is None else
                                       f"records sharded over {world} GPUs; per step one RCCL all-gather of the compact tread arrays "
                                       f"({treads.size * 32} B per rank) before clustering, every rank clusters an equal share of the groups")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
