"""Builds a variant of libstrling_amd.so with extra compiler flags for ONE translation unit (timing experiments: A/B on one box).
usage: python tools/build_variant.py NAME file.hip -DFOO=1 ...   ->  strling_amd/lib/libstrling_amd_NAME.so (use with STRL_LIB=...)"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strling_amd import build as b
name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
objdir = os.path.join(b.LIBDIR, "obj")
objs = [os.path.join(objdir, s.replace("/", "_") + ".o") for s in b.SOURCES]
var = os.path.join(objdir, f"{unit}.{name}.o")
subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ["-c", "-o", var, os.path.join(b.CSRC, unit)], cwd=b.CSRC)
objs = [var if o.endswith("/" + unit + ".o") else o for o in objs]
out = os.path.join(b.LIBDIR, f"libstrling_amd_{name}.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], cwd=b.CSRC)
print(out)
