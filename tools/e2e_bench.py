"""End to end at the size BASELINE.json names: `strling extract` -> `.bin` -> `strling call` / `strling merge` on a synthetic
coordinate-sorted, indexed BAM of N distinct 150 bp reads of a 30x sample -- zlib level 6, binned random base qualities, aux
tags (bamio.write_bam_slabs) -- with the CLI's own phase breakdown, and a check of a deterministic share of the outputs
against the oracle (test infrastructure: the oracle is the checker, never the thing timed).

usage: python tools/e2e_bench.py [n_pairs] [--dir D] [--check-slabs K] [--level L] [--keep]        (GPU box)
As a module (bench.py): make_input(n_pairs) BEFORE the GPU runtime starts in this process (it forks), run(inp, cli),
check(inp, res, slabs)."""
import json
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

PAIRS_PER_SLAB = 1 << 18      # the slab bench.py's resident batch is made of (64 of them = 2^25 reads)


def work_dir(need_bytes):
    """a directory with room for the file: TMPDIR / /tmp if its filesystem has it, else /dev/shm (RAM-backed; the GPU box has
    79 GB of disk and 1.5 TB of shm)"""
    cands = [os.environ.get("STRL_E2E_DIR"), os.environ.get("TMPDIR", "/tmp"), "/dev/shm"]
    for d in cands:
        if not d or not os.path.isdir(d):
            continue
        st = os.statvfs(d)
        if st.f_bavail * st.f_frsize > need_bytes * 1.3 + (4 << 30):
            return d
    return cands[1]


def make_input(n_pairs, d=None, level=6, seed=99, quals=True, aux=True, progress=False, reuse=True):
    """-> dict(bam, bed, out, prefix, reads, n_slabs, ...).  S1 mix, distinct reads, coordinate sorted, .bai + ref.fasta.str.
    The file is CACHED in the work directory: a side-car <tag>.input.json written behind the .bai names its size, seed, level and the
    BAM's byte count; a later call with the same parameters (the next N of a 1 / 2 / 4 / 8 sweep on one node; a second bench
    run) finds it and writes nothing.  cleanup() removes outputs only, unless asked for the input too."""
    from strling_amd import bamio
    n_slabs = max(1, n_pairs // PAIRS_PER_SLAB)
    pairs = n_pairs // n_slabs
    d = d or work_dir(n_pairs * 2 * 115)
    tag = f"e2e_{n_pairs}_{level}"
    bam, bed, side = f"{d}/{tag}.bam", f"{d}/{tag}.str", f"{d}/{tag}.input.json"
    key = {"n_pairs": n_pairs, "level": level, "seed": seed, "quals": bool(quals), "aux": bool(aux), "writer": 2}
    if reuse and os.path.exists(side):
        try:
            j = json.load(open(side))
            if j.get("key") == key and os.path.getsize(bam) == j["bam_bytes"] and os.path.exists(bam + ".bai") and os.path.exists(bed):
                inp = j["inp"]
                inp["reused_cached_input"] = True
                return inp
        except Exception:
            pass
    if os.path.exists(side):
        os.remove(side)
    pr = (lambda k, n, s: print(f"[e2e] slab {k}/{n} {s:.0f} s", file=sys.stderr, flush=True) if k % 64 == 0 else None) if progress else None
    r = bamio.write_bam_slabs(bam, n_slabs, pairs, seed=seed, level=level, quals=quals, aux=aux, index=True, bed=bed, progress=pr)
    inp = {"bam": bam, "bed": bed, "out": f"{d}/{tag}.bin", "prefix": f"{d}/{tag}", "reads": r["reads"], "n_slabs": n_slabs, "pairs_per_slab": pairs,
           "seed": seed, "level": level, "bam_MB": round(r["bytes"] / 1e6, 1), "make_s": round(r["seconds"], 1), "make_procs": r["procs"], "dir": d,
           "targets": r["targets"], "tag": tag,
           "input": f"{r['reads']} distinct reads in {n_slabs} slabs of a 30x sample, coordinate sorted + unmapped tail, zlib level {level}, "
                    f"{'binned random' if quals else 'absent'} qualities, {'NM MD AS XS RG' if aux else 'no'} aux tags"}
    try:
        with open(side + ".tmp", "w") as f:
            json.dump({"key": key, "bam_bytes": os.path.getsize(bam), "inp": inp}, f)
        os.replace(side + ".tmp", side)
    except Exception:
        pass
    return inp


def _quota():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 1)
    except Exception:
        return None


SETTLE_S = 0.0      # set by run(): pause in front of every timed process
PROCESS_TIMEOUT_S = 900


def _timed(cmd, env):
    # The driver reclaims the previous process' device memory (tens of GB at the headline size) after that process has gone;
    # a process started right behind it pays for it in its own context creation (measured: 0.4 -> 2.1 s).  Separate runs of
    # `strling` are not back to back like that: every timed process starts on a device that has settled.
    if SETTLE_S:
        time.sleep(SETTLE_S)
    t = time.time()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=PROCESS_TIMEOUT_S)
    except subprocess.TimeoutExpired as e:        # (a process that hangs must not hang the bench line with it)
        r = subprocess.CompletedProcess(cmd, 124, stdout="", stderr=f"{(e.stderr or b'').decode(errors='replace') if isinstance(e.stderr, bytes) else (e.stderr or '')}\n[e2e] killed after {PROCESS_TIMEOUT_S} s")
    return r, time.time() - t


def run(inp, cli, threads=(0,), call=True, merge=True, repeats=1, gpus=1, extra_env=None):
    """`strling extract -v` once per thread count (0 = the CLI's default), then `strling call` and `strling merge` on the
    .bin -> the end_to_end block of bench.py's line.  gpus > 1: `strling extract --gpus N` / `strling merge --gpus N` (one process,
    N contexts: a contiguous share of the file per device, each fed by its own host threads)."""
    global SETTLE_S
    # (5 s were not always enough at the headline size: the driver clears what the previous process held -- 84 GB at ~48 GB/s -- and
    # a process whose large allocations arrive before that is done pays for the clearing itself, 0.5 s in front of its loop
    # (profiles/r06/state_alloc_diag.log, bench_default_full_size_r6h.json))
    SETTLE_S = 1.0 + 7.0 * min(1.0, inp["bam_MB"] / 50000.0)
    res = {"unit": "reads/s", "reads": inp["reads"], "bam_MB": inp["bam_MB"], "settle_s_before_each_process": round(SETTLE_S, 1), "input": inp.get("input"), "make_s": inp.get("make_s"),
           "host_threads_available": os.cpu_count(), "cgroup_cpu_quota": _quota(), "gpus": gpus, "runs": []}
    g_args = ["--gpus", str(gpus)] if gpus > 1 else []
    for t in list(threads) * repeats:
        env = dict(os.environ, STRL_DECODE_TIMING="1", STRL_FRONT_TIMING="1")
        env.update(extra_env or {})
        if t:
            env["STRL_THREADS"] = str(t)
        r, wall = _timed([cli, "extract", "-v", "-g", inp["bed"]] + g_args + [inp["bam"], inp["out"]], env)
        err = r.stderr.splitlines()
        line = [l for l in err if "seconds: total" in l]
        loop_s = float(line[-1].split("total")[1].split()[0]) if line else None
        front = None
        fl = [l for l in err if "device front end, ms over" in l]
        if fl:
            m = re.search(r"over (\d+) chunks: copies to the device ([\d.]+)  inflate ([\d.]+)  record scan ([\d.]+)  \(([\d.]+) MB compressed -> ([\d.]+) MB inflated", fl[-1])
            if m:
                ch, h2d, inf, scan, cmb, imb = (float(x) for x in m.groups())
                front = {"chunks": int(ch), "copy_ms": h2d, "inflate_ms": inf, "record_scan_ms": scan, "compressed_MB": cmb, "inflated_MB": imb,
                         "inflate_GBps": round(imb / inf, 1) if inf else None}
        mem = [l for l in err if "device memory in use" in l]
        mem_gb = float(re.search(r": ([\d.]+) GB of", mem[-1]).group(1)) if mem else None
        n_str = [l for l in err if " STR reads, " in l]
        su = [l for l in err if "seconds before the loop" in l]
        sh = [l.split("[strling] ")[1] for l in err if l.startswith("[strling] share ")]
        gl = [l.split("[strling] ")[1] for l in err if "per-read state gathered" in l]
        run_ = {"decode_threads": t or "default", "rc": r.returncode, "shares": sh or None, "gather": gl[-1] if gl else None, "wall_s": round(wall, 3), "reads_per_s_wall": round(inp["reads"] / wall),
                "loop_s": loop_s, "reads_per_s_loop": round(inp["reads"] / loop_s) if loop_s else None, "device_front_end": front,
                "device_mem_GB": mem_gb, "str_reads": int(n_str[-1].split(" reads, ")[1].split()[0]) if n_str else None,
                "phases": line[-1].split("seconds:")[1].strip() if line else r.stderr[-400:],
                "outside_the_loop": su[-1].split("seconds before the loop:")[1].strip() if su else None}
        # the progress lines ("<primary reads so far> <rate> reads/sec", one per chunk summary): seconds between consecutive chunks
        at = []
        for l in err:
            m = re.match(r"^(\d+) ([\d.]+) reads/sec$", l.strip())
            if m and float(m.group(2)) > 0:
                at.append(int(m.group(1)) / float(m.group(2)))
        if len(at) > 4:
            d = sorted(1e3 * (b - a) for a, b in zip(at, at[1:]))
            run_["chunk_summaries"] = {"n": len(at), "first_at_ms": round(1e3 * at[0], 1), "last_at_ms": round(1e3 * at[-1], 1),
                                       "between_ms": {"min": round(d[0], 1), "median": round(d[len(d) // 2], 1), "p90": round(d[int(len(d) * 0.9)], 1), "max": round(d[-1], 1)}}
        if r.returncode != 0:
            run_["stderr_tail"] = r.stderr[-600:]
        res["runs"].append(run_)
    ok = [x for x in res["runs"] if x["rc"] == 0]
    if not ok:
        res["error"] = "strling extract failed"
        return res
    best = max(ok, key=lambda x: x["reads_per_s_wall"])
    res["value"] = best["reads_per_s_wall"]
    res["extract_s"] = best["wall_s"]
    res["reads_per_s_loop"] = best.get("reads_per_s_loop")
    # the FIRST run reads a file no process has read since it was written (or since the cache was filled): what a user sees on a
    # file that is not in the page cache is this number, not the best one
    res["first_run_wall_s"] = res["runs"][0]["wall_s"]
    res["first_run_note"] = ("the first extract process behind the writer" if not inp.get("reused_cached_input") else
                             "the first extract process of this bench run on a cached input file (page cache state unknown)")
    res["bin_MB"] = round(os.path.getsize(inp["out"]) / 1e6, 1)
    env = dict(os.environ)
    if call:
        r, wall = _timed([cli, "call", "-v", "-o", inp["prefix"], inp["bam"], inp["out"]], env)
        res["call_s"] = round(wall, 3)
        res["call_rc"] = r.returncode
        if r.returncode == 0:
            res["call_bounds_rows"] = sum(1 for _ in open(inp["prefix"] + "-bounds.txt")) - 1
            res["call_genotype_rows"] = sum(1 for _ in open(inp["prefix"] + "-genotype.txt")) - 1
            ph = [l for l in r.stderr.splitlines() if "seconds:" in l]
            if ph:
                res["call_phases"] = ph[-1].split("seconds:")[1].strip()
            res["extract_plus_call_s"] = round(res["extract_s"] + wall, 3)
            res["reads_per_s_extract_plus_call"] = round(inp["reads"] / (res["extract_s"] + wall))
        else:
            res["call_stderr_tail"] = r.stderr[-600:]
    if merge:
        r, wall = _timed([cli, "merge", "-v"] + g_args + ["-o", inp["prefix"] + "-joint", inp["out"]], env)
        res["merge_s"] = round(wall, 3)
        res["merge_rc"] = r.returncode
        if r.returncode == 0:
            res["merge_bounds_rows"] = sum(1 for _ in open(inp["prefix"] + "-joint-bounds.txt")) - 1
            ph = [l for l in r.stderr.splitlines() if "seconds:" in l]
            if ph:
                res["merge_phases"] = ph[-1].split("seconds:")[1].strip()
        else:
            res["merge_stderr_tail"] = r.stderr[-600:]
    res["note"] = ("`strling extract` BAM file (page cache) -> .bin: whole process wall clock incl. start-up (HIP context, page-locked buffers), copies of the "
                   "compressed bytes, BGZF inflate + CRC + record scan + parse + scorer + pair logic on the device, fragment lengths, .bin writing; "
                   "reads_per_s_loop excludes process start-up and the .bin write.  call_s / merge_s: whole `strling call` (clustering on the device, the bounds' "
                   ".bai region reads inflated and cut out on the device, spanning evidence + genotypes on the host's threads) and `strling merge` processes on that .bin")
    return res


def feed_only(inp, cli, gpus):
    """the host side of `extract --gpus N` alone (STRL_FEED_ONLY=1: header walkers + copy threads into the page-locked rings, no
    device stage): what the feeding threads of N shares sustain on this box's CPUs"""
    if gpus < 2:
        return None
    env = dict(os.environ, STRL_FEED_ONLY="1")
    r, wall = _timed([cli, "extract", "-v", "-g", inp["bed"], "--gpus", str(gpus), inp["bam"], inp["out"] + ".feed"], env)
    m = re.search(r"feed only: (\d+) shares, (\d+) copy threads each.*?, ([\d.]+) s, ([\d.]+) GB of BAM, ([\d.]+) GB/s", r.stderr)
    if not m:
        return {"rc": r.returncode, "stderr_tail": r.stderr[-300:]}
    return {"shares": int(m.group(1)), "copy_threads_per_share": int(m.group(2)), "feed_s": float(m.group(3)), "GB": float(m.group(4)), "GBps": float(m.group(5)),
            "reads_per_s": round(inp["reads"] / float(m.group(3))), "process_wall_s": round(wall, 3)}


def replicas(inp, cli, n, devices):
    """N concurrent `strling extract --device k` processes, one per GPU, each on a whole sample of its own -- how the reference's
    pipelines scale (one single-threaded process per sample: pipelines/bpipe.config:4, strling-joint-bychrom.groovy:8-14) and
    BASELINE configs[4]'s real shape.  Every replica reads the same file (its own .bin out); aggregate = N x reads / the slowest
    process' wall."""
    import threading
    if SETTLE_S:
        time.sleep(SETTLE_S)
    outs = [inp["out"] + f".rep{k}" for k in range(n)]
    res = [None] * n

    def one(k):
        env = dict(os.environ)
        # the box's CPUs are shared: every replica takes its share of the decode / copy threads
        q = _quota() or os.cpu_count() or 1
        env["STRL_THREADS"] = str(max(2, int(q // n)))
        t = time.time()
        try:
            r = subprocess.run([cli, "extract", "-v", "-g", inp["bed"], "--device", str(k % max(1, devices)), inp["bam"], outs[k]], capture_output=True, text=True, env=env,
                               timeout=PROCESS_TIMEOUT_S)
            rc, err = r.returncode, r.stderr
        except subprocess.TimeoutExpired:
            rc, err = 124, "timeout"
        wall = time.time() - t
        line = [l for l in err.splitlines() if "seconds: total" in l]
        res[k] = {"device": k % max(1, devices), "rc": rc, "wall_s": round(wall, 3), "loop_s": float(line[-1].split("total")[1].split()[0]) if line else None}

    th = [threading.Thread(target=one, args=(k,)) for k in range(n)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.time() - t0
    same = None
    if all(r["rc"] == 0 for r in res) and os.path.exists(inp["out"]):
        ref = open(inp["out"], "rb").read() if os.path.getsize(inp["out"]) < (1 << 31) else None
        same = all(os.path.getsize(o) == os.path.getsize(inp["out"]) for o in outs) and (ref is None or all(open(o, "rb").read() == ref for o in outs))
    for o in outs:
        if os.path.exists(o):
            os.remove(o)
    return {"replicas": n, "devices": devices, "wall_s": round(wall, 3), "aggregate_reads_per_s": round(n * inp["reads"] / wall) if all(r["rc"] == 0 for r in res) else None,
            "per_replica": res, "bins_identical_to_the_single_run": same,
            "what": "N concurrent `strling extract --device k` processes (one sample each, here the same file; STRL_THREADS = CPU quota / N each): N x reads / the slowest process' wall clock"}


def _slab_check(a):
    """oracle over ONE slab regenerated from its seed: its treads in .bin order, its `call` bounds rows"""
    c, n_slabs, pairs, seed, frag, want_call = a
    import numpy as np
    from oracle import oracle as O
    from strling_amd import bamio
    rec, g = bamio.slab_records(c, n_slabs, pairs, seed)
    med = O.median(frag)
    t = O.extract(rec, g, O.make_opts(med, 0.8, 40))
    names = [rec.qname(int(i)) for i in t["qname_id"]]
    rows = None
    if want_call:
        b, _, _ = O.call(t, rec, frag)
        mine = {rec.targets[2 * c][0], rec.targets[2 * c + 1][0]}
        rows = sorted(l for l in b.splitlines()[1:] if l.split("\t")[0] in mine)
    keep = ("tid", "position", "repeat", "flag", "split", "mapping_quality", "repeat_count", "align_length")
    return c, {f: np.ascontiguousarray(t[f]) for f in keep}, names, rows, rec.n


def check(inp, slabs, call=True, procs=None):
    """The .bin's treads of the chosen slabs (every tread whose qname belongs to one of them, in file order) against the oracle
    run over those slabs alone -- qname groups never interact, so a slab's treads are a subsequence of the file's -- and the
    `-bounds.txt` rows on those slabs' contigs against the oracle's `call` over the slab (depth column included)."""
    import multiprocessing as mp
    import numpy as np
    from strling_amd import api
    t0 = time.time()
    b = api.bin_read(inp["out"])
    t, qo, qn = b["treads"], b["qname_off"].astype(np.int64), np.frombuffer(b["qnames"], np.uint8)
    # qname "q<pair id>": slab = pair id // pairs_per_slab
    pid = np.zeros(len(t), np.int64)
    ln = qo[1:] - qo[:-1]
    for k in range(1, int(ln.max()) if len(t) else 0):
        has = ln > k
        pid[has] = pid[has] * 10 + (qn[qo[:-1][has] + k] - 48)
    slab_of = pid // inp["pairs_per_slab"]
    frag = b["frag"]
    rows_by_chrom = {}
    if call and os.path.exists(inp["prefix"] + "-bounds.txt"):
        for l in open(inp["prefix"] + "-bounds.txt").read().splitlines()[1:]:
            rows_by_chrom.setdefault(l.split("\t")[0], []).append(l)
    jobs = [(c, inp["n_slabs"], inp["pairs_per_slab"], inp["seed"], frag, bool(rows_by_chrom)) for c in slabs]
    procs = procs or max(1, min(len(jobs), int(_quota() or os.cpu_count() or 1)))
    bad, n_t, n_rows, n_reads = [], 0, 0, 0
    with mp.get_context("spawn").Pool(procs) as pool:
        for c, exp, names, rows, n in pool.imap_unordered(_slab_check, jobs):
            sel = np.nonzero(slab_of == c)[0]
            n_reads += n
            n_t += len(names)
            ok = sel.size == len(names)
            if ok:
                for f, v in exp.items():
                    ok = ok and np.array_equal(t[f][sel], v)
                got_names = [qn[qo[i]:qo[i + 1]].tobytes() for i in sel]
                ok = ok and got_names == names
            if not ok:
                bad.append(f"slab {c}: treads differ ({sel.size} vs {len(names)})")
            if rows is not None:
                tg = inp["targets"]
                got = sorted(rows_by_chrom.get(tg[2 * c][0], []) + rows_by_chrom.get(tg[2 * c + 1][0], []))
                n_rows += len(rows)
                if got != rows:
                    bad.append(f"slab {c}: bounds rows differ ({len(got)} vs {len(rows)})")
    return {"slabs": len(slabs), "reads_checked": n_reads, "treads_checked": n_t, "bounds_rows_checked": n_rows, "mismatches": bad[:8], "ok": not bad,
            "seconds": round(time.time() - t0, 1),
            "what": "every tread of the .bin whose qname belongs to one of the checked slabs, field by field and in order, against the oracle's extract "
                    "over the slab regenerated from its seed; the `strling call` -bounds.txt rows on those slabs' contigs (depth column included) "
                    "against the oracle's call over the slab"}


def check_in_subprocess(inp, slabs, call=True):
    """check() in a process of its own: the caller (bench.py) has not started the GPU runtime yet and forks generators later --
    loading the library (bin_read) into it first made its later hipInit find no device"""
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"inp": inp, "slabs": list(slabs), "call": bool(call)}, f)
        path = f.name
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--check-json", path], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if r.returncode == 0 and lines else {"ok": False, "error": (r.stderr or r.stdout)[-400:]}
    finally:
        os.remove(path)


def pick_slabs(n_slabs, k):
    """a deterministic spread of k slabs over the file (first, last, evenly between)"""
    k = max(1, min(k, n_slabs))
    return sorted({int(round(j * (n_slabs - 1) / max(1, k - 1))) for j in range(k)})


def cleanup(inp, keep_input=False):
    """outputs always; the input file (BAM, .bai, BED, side-car) unless it is to stay cached for the next run"""
    ps = [inp["out"], inp["out"] + ".feed", inp["prefix"] + "-bounds.txt", inp["prefix"] + "-genotype.txt", inp["prefix"] + "-unplaced.txt", inp["prefix"] + "-joint-bounds.txt"]
    if not keep_input:
        ps += [inp["bam"], inp["bam"] + ".bai", inp["bed"], inp["prefix"] + ".input.json"]
    for p in ps:
        if os.path.exists(p):
            os.remove(p)


def n1_record(inp, res=None):
    """the N = 1 figures of this input, kept beside the cached file: a later N > 1 run of the sweep quotes its strong scaling
    against them (bench.py end_to_end.strong_scaling_vs_n1)"""
    path = inp["prefix"] + ".n1.json"
    if res is not None:
        try:
            json.dump({"extract_s": res.get("extract_s"), "reads_per_s_wall": res.get("value"), "reads_per_s_loop": res.get("reads_per_s_loop"), "reads": inp["reads"]}, open(path, "w"))
        except Exception:
            pass
        return None
    try:
        return json.load(open(path))
    except Exception:
        return None


if __name__ == "__main__":
    import argparse
    from strling_amd import build
    ap = argparse.ArgumentParser()
    ap.add_argument("n_pairs", type=int, nargs="?", default=1 << 22)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--check-slabs", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=0, help="also run N concurrent `extract --device k` processes (one per GPU)")
    ap.add_argument("--feed-only", action="store_true", help="also time the host side of --gpus N alone")
    ap.add_argument("--out", default="")
    ap.add_argument("--check-json", default="", help="(internal) run check() on the input described by this file and print its result")
    a = ap.parse_args()
    if a.check_json:
        j = json.load(open(a.check_json))
        print(json.dumps(check(j["inp"], j["slabs"], call=j["call"])))
        sys.exit(0)
    inp = make_input(a.n_pairs, d=a.dir, level=a.level, progress=True)
    print(f"[e2e] wrote {inp['bam']} ({inp['bam_MB']} MB, {inp['reads']} reads) in {inp['make_s']} s", file=sys.stderr, flush=True)
    res = run(inp, build.CLI, repeats=a.repeats, gpus=a.gpus)
    if a.feed_only:
        res["feed_only"] = feed_only(inp, build.CLI, a.gpus)
    if a.replicas:
        import torch
        res["replicas"] = replicas(inp, build.CLI, a.replicas, max(1, torch.cuda.device_count()))
    if a.check_slabs and "error" not in res:
        res["check"] = check(inp, pick_slabs(inp["n_slabs"], a.check_slabs), call=res.get("call_rc") == 0)
    cleanup(inp, keep_input=a.keep)
    s = json.dumps(res)
    if a.out:
        open(a.out, "w").write(s + "\n")
    print(s)
