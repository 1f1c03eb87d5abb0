"""End-to-end `strling extract` rate (BAM bytes -> .bin) on a synthetic BAM, with the CLI's own phase breakdown.
usage: python tools/e2e_bench.py [n_pairs] [threads...]     (GPU box)
As a module: make_input(n_pairs) before the GPU runtime starts in this process (it forks), run(paths, cli, threads)."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def make_input(n_pairs, d=None, level=1):
    """-> dict(bam, bed, out, reads, bam_MB, make_s).  S1 mix, distinct reads, coordinate sorted, no index."""
    from strling_amd import bamio, synth
    d = d or os.environ.get("TMPDIR", "/tmp")
    bam, bed, out = f"{d}/e2e_{n_pairs}.bam", f"{d}/e2e_{n_pairs}.str", f"{d}/e2e_{n_pairs}.bin"
    t0 = time.time()
    chunks = max(1, min(64, n_pairs // 65536))
    rec, g = synth.synth_wgs_30x(chunks, n_pairs // chunks, seed=99)
    bamio.write_bam_parallel(bam, rec, level=level)
    bamio.write_genome_bed(bed, g, rec.targets)
    return {"bam": bam, "bed": bed, "out": out, "reads": rec.n, "bam_MB": round(os.path.getsize(bam) / 1e6, 1), "make_s": round(time.time() - t0, 1)}


def run(inp, cli, threads=(0,)):
    """runs `strling extract -v` on the prepared input once per thread count (0 = the CLI's default) -> end_to_end block"""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 1)
    except Exception:
        pass
    res = {"unit": "reads/s", "reads": inp["reads"], "bam_MB": inp["bam_MB"], "host_threads_available": os.cpu_count(), "cgroup_cpu_quota": quota,
           "runs": []}
    for t in threads:
        env = dict(os.environ, STRL_DECODE_TIMING="1", STRL_FRONT_TIMING="1")
        if t:
            env["STRL_THREADS"] = str(t)
        t1 = time.time()
        r = subprocess.run([cli, "extract", "-v", "-g", inp["bed"], inp["bam"], inp["out"]], capture_output=True, text=True, env=env)
        wall = time.time() - t1
        line = [l for l in r.stderr.splitlines() if "seconds: total" in l]
        dec = [l.split("decode seconds:")[1].strip() for l in r.stderr.splitlines() if "decode seconds:" in l]
        loop_s = float(line[-1].split("total")[1].split()[0]) if line else None
        front = None
        fl = [l for l in r.stderr.splitlines() if "device front end, ms over" in l]
        if fl:
            import re
            m = re.search(r"over (\d+) chunks: copies to the device ([\d.]+)  inflate ([\d.]+)  record scan ([\d.]+)  \(([\d.]+) MB compressed -> ([\d.]+) MB inflated", fl[-1])
            if m:
                ch, h2d, inf, scan, cmb, imb = (float(x) for x in m.groups())
                front = {"chunks": int(ch), "copy_ms": h2d, "inflate_ms": inf, "record_scan_ms": scan, "compressed_MB": cmb, "inflated_MB": imb,
                         "inflate_GBps": round(imb / inf, 1) if inf else None,
                         "note": "HIP events on the front end's own stream (copies overlap the previous chunk's inflate): BGZF inflate, record-boundary scan "
                                 "and BAM parse run on the device; the host walks block headers and copies compressed bytes"}
        res["runs"].append({"decode_threads": t or "default: min(64, 1.5 x CPU quota)", "rc": r.returncode, "wall_s": round(wall, 3),
                            "reads_per_s_wall": round(inp["reads"] / wall), "loop_s": loop_s,
                            "reads_per_s_loop": round(inp["reads"] / loop_s) if loop_s else None, "decode": dec, "device_front_end": front,
                            "phases": line[-1].split("seconds:")[1].strip() if line else r.stderr[-300:]})
    best = max(res["runs"], key=lambda x: x["reads_per_s_wall"])
    res["value"] = best["reads_per_s_wall"]
    res["note"] = ("`strling extract` BAM file (page cache) -> .bin, whole process wall clock incl. start-up (HIP context, page-locked buffers), copies of the "
                   "compressed bytes, BGZF inflate + record scan + parse + scorer + pair logic on the device, fragment lengths, .bin writing; reads_per_s_loop "
                   "excludes process start-up and the .bin write")
    return res


if __name__ == "__main__":
    from strling_amd import build
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    threads = [int(x) for x in sys.argv[2:]] or [0]
    inp = make_input(n_pairs)
    print(json.dumps(run(inp, build.CLI, threads)))
