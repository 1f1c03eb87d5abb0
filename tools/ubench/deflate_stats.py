"""Symbol statistics of DEFLATE streams (pure Python, a few blocks): literals, matches, match length histogram, code-length
histogram of the literal codes, refills.  usage: python tools/ubench/deflate_stats.py  (the two inputs of tools/inflate_bench.py)"""
import os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
ORDER = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]

class Bits:
    def __init__(s, d): s.d = d; s.p = 0
    def get(s, n):
        v = 0
        for i in range(n):
            v |= ((s.d[s.p >> 3] >> (s.p & 7)) & 1) << i; s.p += 1
        return v

def build(lens):
    codes, code = {}, 0
    for l in range(1, 16):
        for sym, ll in enumerate(lens):
            if ll == l: codes[(l, code)] = sym; code += 1
        code <<= 1
    return codes

def dec(b, codes):
    c = 0
    for l in range(1, 16):
        c = (c << 1) | b.get(1)
        if (l, c) in codes: return codes[(l, c)], l
    raise ValueError

def stats(stream, st):
    b = Bits(stream)
    while True:
        fin, typ = b.get(1), b.get(2)
        st["blocks"] += 1
        if typ == 0:
            b.p = (b.p + 7) & ~7; n = b.get(16); b.get(16); b.p += 8 * n; st["stored"] += n
        else:
            if typ == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8; dl = [5] * 32
            else:
                hl, hd, hc = b.get(5) + 257, b.get(5) + 1, b.get(4) + 4
                cl = [0] * 19
                for i in range(hc): cl[ORDER[i]] = b.get(3)
                cc = build(cl); lens = []
                while len(lens) < hl + hd:
                    s, _ = dec(b, cc)
                    if s < 16: lens.append(s)
                    elif s == 16: lens += [lens[-1]] * (3 + b.get(2))
                    elif s == 17: lens += [0] * (3 + b.get(3))
                    else: lens += [0] * (11 + b.get(7))
                ll, dl = lens[:hl], lens[hl:]
            lc, dc = build(ll), build(dl)
            while True:
                s, l = dec(b, lc)
                if s < 256: st["lit"] += 1; st["lit_len"][l] += 1
                elif s == 256: break
                else:
                    L = LBASE[s - 257] + b.get(LEXT[s - 257]); d, dl_ = dec(b, dc); D = DBASE[d] + b.get(DEXT[d])
                    st["match"] += 1; st["match_bytes"] += L; st["mlen"][min(L, 64)] += 1; st["len_code_len"][l] += 1; st["dist_code_len"][dl_] += 1
                    st["overlap"] += D < L
        if fin: break
    st["bits"] += b.p

def report(name, streams):
    st = dict(blocks=0, stored=0, lit=0, match=0, match_bytes=0, overlap=0, bits=0, lit_len=[0] * 16, mlen=[0] * 65, len_code_len=[0] * 16, dist_code_len=[0] * 16)
    for s in streams: stats(s, st)
    tot = st["lit"] + st["match_bytes"] + st["stored"]
    print(f"{name}: {len(streams)} streams, {tot} bytes; literals {st['lit']} ({st['lit'] / tot:.3f} of the bytes), matches {st['match']} (mean length {st['match_bytes'] / max(st['match'], 1):.1f}, "
          f"{st['overlap']} overlapping), symbols per output byte {(st['lit'] + st['match']) / tot:.3f}, bits per symbol {st['bits'] / (st['lit'] + st['match']):.2f}")
    print("   literal code lengths 1..15:", st["lit_len"][1:], " length-code lengths:", st["len_code_len"][1:], " distance-code lengths:", st["dist_code_len"][1:])
    m = st["mlen"]; print("   match lengths 3..16:", m[3:17], " 17-32:", sum(m[17:33]), " 33-63:", sum(m[33:64]), " >=64:", m[64])

if __name__ == "__main__":
    from strling_amd import bamio, synth
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import inflate_bench as ib
    bam = "/tmp/deflate_stats.bam"
    rec, g = synth.synth_wgs_30x(1, 1 << 14, seed=5)
    bamio.write_bam_parallel(bam, rec, level=1)
    streams, sizes = ib.bam_blocks(bam)
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    report("level 1, constant qualities", streams[2:2 + k])
    real, _ = ib.realistic(streams[2:2 + k], np.random.default_rng(1))
    report("level 6, binned random qualities", real)
