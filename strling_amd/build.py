"""Builds libstrling_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs on the CPU-only build box as well; the built
.so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libstrling_amd.so")
CLI = os.path.join(LIBDIR, "strling")
SOURCES = ["score.hip", "pair.hip", "sort.hip", "cluster.hip", "bgzf.hip", "front.hip", "comm.hip", "host_logic.cpp", "host_score.cpp", "call_logic.cpp", "nim_tables.cpp"]
CLI_SOURCES = ["cli/main.cpp", "cli/bam_reader.cpp", "cli/fast_inflate.cpp", "cli/bgzf_feed.cpp", "cli/cram_reader.cpp", "cli/cram_codecs.cpp"]
HEADERS = ["common.h", "device_util.h", "sort.h", "inflate_wave.h", "inflate_group.h", "front.h", "score_core.h", "score_tables.h", "nim_tables.h", "host_score.h", "cli/bam_reader.h", "cli/fast_inflate.h", "cli/bgzf_feed.h", "cli/cram_reader.h", "cli/cram_codecs.h", "../../include/strling_amd.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# the grouped form of the device inflate (csrc/inflate_group.h): an experiment that measured slower (profiles/r05/inflate_group/);
# not in the shipped library unless asked for
WITH_INFLATE_GROUP = os.environ.get("STRL_WITH_INFLATE_GROUP") == "1"
if WITH_INFLATE_GROUP:
    FLAGS = FLAGS + ["-DSTRL_WITH_INFLATE_GROUP"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def _compile_objects(names, objdir, deps_common, force, verbose):
    """one hipcc -c per translation unit, stale ones only, side by side (the kernels are independent TUs: no -fgpu-rdc)"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in names:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + deps_common):
            cmd = [hipcc] + FLAGS + ["-c", "-o", obj, src]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            list(ex.map(run, jobs))
    return objs, bool(jobs)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, fresh = _compile_objects([s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))], objdir, hdrs, force, verbose)
    if fresh or force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]   # RCCL is bound at first use (comm.hip)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    cli_srcs = [os.path.join(CSRC, s) for s in CLI_SOURCES]
    if all(os.path.exists(s) for s in cli_srcs):
        cobjs, cfresh = _compile_objects(CLI_SOURCES, objdir, hdrs, force, verbose)
        if cfresh or force or _stale(CLI, cobjs + [LIB]):
            cmd = [_hipcc(), "-o", CLI] + cobjs + ["-L" + LIBDIR, "-lstrling_amd", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
