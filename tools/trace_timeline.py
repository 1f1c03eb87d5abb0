"""Timeline of one `strling extract` from a rocprofv3 --kernel-trace CSV: per kernel name the busy time, how much of it
ran with no other kernel on the device, and the gaps of the inflate stream.  usage: python tools/trace_timeline.py run_kernel_trace.csv"""
import csv, sys, collections

def main(path):
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("strl::", "").replace("void ", "")[:40], r.get("Queue_Id", "")))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    print(f"{len(ev)} launches over {(t1 - t0) / 1e6:.1f} ms")
    # device busy (union), and per-name exclusive time by a sweep over the boundaries
    pts = sorted({e[0] for e in ev} | {e[1] for e in ev})
    idx = {p: i for i, p in enumerate(pts)}
    cover = [[] for _ in range(len(pts) - 1)]
    for k, e in enumerate(ev):
        for i in range(idx[e[0]], idx[e[1]]):
            cover[i].append(k)
    busy = sum(pts[i + 1] - pts[i] for i in range(len(cover)) if cover[i])
    print(f"device busy (any kernel) {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms")
    tot = collections.Counter(); alone = collections.Counter(); cnt = collections.Counter()
    for e in ev:
        tot[e[2]] += e[1] - e[0]; cnt[e[2]] += 1
    for i, c in enumerate(cover):
        if len(c) == 1:
            alone[ev[c[0]][2]] += pts[i + 1] - pts[i]
    print(f"{'kernel':42s} {'calls':>6s} {'total ms':>9s} {'alone ms':>9s}")
    for n, t in tot.most_common(14):
        print(f"{n:42s} {cnt[n]:6d} {t / 1e6:9.2f} {alone[n] / 1e6:9.2f}")
    print("the first launches (ms from the first):")
    for e in ev[:int(sys.argv[2]) if len(sys.argv) > 2 else 0]:
        print(f"   {(e[0] - t0) / 1e6:8.2f} .. {(e[1] - t0) / 1e6:8.2f}  {e[2]}  q{e[3]}")
    inf = [e for e in ev if e[2].startswith("inflate_kernel")]
    if len(inf) > 2:
        gaps = [(inf[i + 1][0] - inf[i][1]) / 1e6 for i in range(len(inf) - 1)]
        print("inflate launches: ms " + " ".join(f"{(e[1] - e[0]) / 1e6:.1f}" for e in inf[:24]))
        print("gaps between them: ms " + " ".join(f"{g:.1f}" for g in gaps[:24]) + f"   (sum {sum(gaps):.1f})")
        print(f"first inflate starts {(inf[0][0] - t0) / 1e6:.1f} ms after the first kernel; last ends {(t1 - inf[-1][1]) / 1e6:.1f} ms before the last kernel's end")
        # what runs inside one mid-file gap
        k = len(inf) // 2
        a, b = inf[k][1], inf[k + 1][0]
        print(f"between inflate {k} and {k + 1} ({(b - a) / 1e6:.2f} ms):")
        for e in ev:
            if e[1] > a and e[0] < b and not e[2].startswith("inflate"):
                print(f"   +{(e[0] - a) / 1e6:7.2f} .. +{(e[1] - a) / 1e6:7.2f}  {e[2]}  q{e[3]}")

if __name__ == "__main__":
    main(sys.argv[1])
