"""Device scorer logic (strling_amd/csrc/score_core.h compiled for the host, one lane) against the
oracle on random and adversarial segments.  CPU only: this exercises the bit packing, LUT,
tie rule, bit-parallel recount and threshold ladder that the HIP kernels run, without a GPU."""
import ctypes as C
import os
import random
import subprocess

import pytest

from strling_amd.records import pack_seq4, unpack_result

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(HERE, "emu", "libscore_emu.so")
    src = os.path.join(HERE, "emu", "score_emu.cpp")
    core = os.path.join(HERE, "..", "strling_amd", "csrc", "score_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    L = C.CDLL(so)
    L.emu_set_p.argtypes = [C.c_double]
    L.emu_score.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return L


def _reads(rng, n):
    out = []
    for _ in range(n):
        kind = rng.random()
        L = rng.choice([150, 150, 151, 100, 76, 36, 17, 16, 6, 5, 3, 2, 1, 0, 250, 255, 300, 510]) if rng.random() < 0.5 else rng.randint(0, 160)
        if kind < 0.3:
            s = "".join(rng.choice("ACGT") for _ in range(L))
        elif kind < 0.8:
            k = rng.randint(1, 6)
            u = "".join(rng.choice("ACGT") for _ in range(k))
            pur = rng.choice([1.0, 0.98, 0.95, 0.9, 0.85, 0.7])
            ph = rng.randint(0, k)
            s = (u * (L // k + 3))[ph:ph + L]
            s = "".join(c if rng.random() < pur else rng.choice("ACGT") for c in s)
        elif kind < 0.9:
            k1, k2 = rng.randint(2, 6), rng.randint(2, 6)
            u1 = "".join(rng.choice("ACGT") for _ in range(k1))
            u2 = "".join(rng.choice("AC") for _ in range(k2))
            cut = rng.randint(0, L)
            s = ((u1 * 200)[:cut] + (u2 * 200))[:L]
        else:
            s = "".join(rng.choice("ACGTNNMR=") for _ in range(L))
        if rng.random() < 0.15 and L > 0:
            s = list(s)
            for _ in range(rng.randint(1, 25)):
                s[rng.randrange(L)] = rng.choice("NNNMRY")
            s = "".join(s)
        out.append(s)
    return out


@pytest.mark.parametrize("p", [0.8, 0.6, 0.9])
def test_emulated_device_scorer_matches_oracle(emu, oracle, p):
    rng = random.Random(int(p * 100))
    emu.emu_set_p(p)
    o0, o1 = C.c_uint32(), C.c_uint32()
    for s in _reads(rng, 6000):
        seq4, _, _ = pack_seq4([s])
        klass = 0 if len(s) <= 160 else (1 if len(s) <= 256 else 2)
        if rng.random() < 0.2:
            klass = max(klass, rng.choice([1, 2]))
        emu.emu_score(seq4.ctypes.data, 0, len(s), 0, klass, C.byref(o0), C.byref(o1))
        assert unpack_result(o0.value)[:2] == oracle.get_repeat(s, p), s
        if len(s) > 1:   # a soft-clipped end: arbitrary sub-segment, both lowered thresholds (extract.nim:208,242)
            a = 0 if rng.random() < 0.3 else rng.randint(0, len(s) - 1)
            b = len(s) if rng.random() < 0.5 else rng.randint(a, len(s))
            emu.emu_score(seq4.ctypes.data, a, b - a, 1, klass, C.byref(o0), C.byref(o1))
            exp = (oracle.get_repeat(s[a:b], p - 0.07), oracle.get_repeat(s[a:b], min(p, 0.6)))
            assert (unpack_result(o0.value)[:2], unpack_result(o1.value)[:2]) == exp, (s, a, b)
