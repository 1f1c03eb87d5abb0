"""Reduce a rocprofv3 kernel trace (run_kernel_trace.csv) to a timeline summary: busy time (union of kernel intervals),
sum of kernel durations, concurrency, and the per-queue order of the steady-state steps.  usage: timeline.py TRACE.csv [N_LAST [N_PRINT [SKIP_AT_END]]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 600
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # kernels to drop from the end first (bench.py ends with instrumented steps)
ev = ev[-(n_last + skip):len(ev) - skip]
t0 = ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
span = max(e for _, e, _, _ in ev) - t0
print(f"{len(ev)} kernels: span {span/1e3:.1f} us, busy (union) {busy/1e3:.1f} us, sum of durations {tot/1e3:.1f} us, mean concurrency {tot/busy:.2f}")
qs = {}
for s, e, k, q in ev: qs.setdefault(q, []).append((s, e, k))
for q, v in qs.items(): print("queue", q, len(v), "kernels, sum", sum(e - s for s, e, _ in v) / 1e3, "us")
if len(sys.argv) > 3:
    for s, e, k, q in ev[-int(sys.argv[3]):]:
        print(f"{(s - t0)/1e3:10.1f} {(e - s)/1e3:8.1f}  q{q}  {k[:70]}")
