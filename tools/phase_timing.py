"""Per-phase wave-cycle breakdown of the scorer kernels (debug build with -DSTRL_PHASE_TIMING).

  hipcc ... -DSTRL_PHASE_TIMING -o tools/ab/libstrl_phase.so   (see `build()` below)
  STRL_LIB=tools/ab/libstrl_phase.so python tools/phase_timing.py
"""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "load(queue+meta+seq->LDS)", 1: "conv 4bit->2bit", 2: "k2 clear+hist", 3: "k2 decide+recount", 4: "k3 clear+hist",
         5: "k3 decide+recount", 6: "k4 clear+hist", 7: "k4 decide+recount", 8: "k5 clear+hist", 9: "k5 decide+recount",
         10: "k6 clear+hist", 11: "k6 decide+recount", 12: "finalize+queues"}


def build():
    out = os.path.join(ROOT, "tools", "ab", "libstrl_phase.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    from strling_amd import build as b
    src = [os.path.join(ROOT, "strling_amd", "csrc", f) for f in b.SOURCES]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSTRL_PHASE_TIMING", "-o", out] + src +
                          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        print(build()); sys.exit(0)
    import numpy as np, torch
    from strling_amd import api, synth
    L = api.load()
    if os.environ.get("BENCH_DATA"):      # the generator bench.py uses (a slab of a 30x WGS)
        n = int(os.environ.get("READS", 2 ** 21))
        rec, g = synth.synth_wgs_30x(max(1, n // 2 ** 17), 2 ** 16, seed=1234)
    else:
        rec, g = synth.synth_wgs(int(os.environ.get("READS", 2 ** 18)), seed=1234, with_qnames=False)
    ctx = api.Context(0); ctx.set_opts(0.8, 40, 350); ctx.set_genome(g)
    soa = api.Soa(rec)
    for rep in range(2):
        L.strl_debug_phase(None, 1)
        whole, soft, st = ctx.score_reads(soa)
        ph = (C.c_ulonglong * 32)()
        L.strl_debug_phase(ph, 0)
    tot = sum(ph)
    print(f"reads {soa.n} scored {st.n_scored} soft {st.n_soft_items}; total wave-cycles {tot/1e6:.1f} M")
    print(f"  segments with a base that is not ACGT: {ph[20]} lanes, {ph[21]} of {ph[22]} wave-items")
    print("  recount executions (waves, lanes) k2 / k3 / k4 (k5, k6 share the k2 / k3 slots):", [(ph[23 + 2 * i], ph[24 + 2 * i]) for i in range(3)])
    tot = sum(ph[:13])
    for i in range(13):
        print(f"  {NAMES[i]:28s} {ph[i]/1e6:10.2f} M  {100*ph[i]/tot:5.1f} %")
