import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["x", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--cache", "/tmp"]
import bench
# monkeypatch: run bench.main up to context creation is complex; replicate minimal loop
import pickle, numpy as np, torch
from strling_amd import api, synth
rec, g = pickle.load(open("/tmp/s1x30_33554432_64_0.pkl", "rb"))
dev = torch.device("cuda", 0)
soa = api.Soa(rec); n = soa.n
rows, qh = soa.pair_rows()
frag = synth.frag_hist(rec); med = api.frag_median(frag); window = api.frag_median(frag, 0.99); mcd = int(0.5 * api.frag_median(frag, 0.5))
def up(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = {k: up(getattr(soa, k)) for k in ("tid", "pos", "end", "seq_off", "l_seq", "clip_l", "clip_r", "mapq", "cig", "seq4")}
drows, dqh = up(rows.view(np.uint8)), up(qh)
cs = api.CReadSoa(n, d["tid"].data_ptr(), d["pos"].data_ptr(), d["end"].data_ptr(), d["seq_off"].data_ptr(), d["l_seq"].data_ptr(), d["clip_l"].data_ptr(),
                  d["clip_r"].data_ptr(), d["mapq"].data_ptr(), d["cig"].data_ptr(), d["seq4"].data_ptr(), d["seq4"].numel(), soa.max_l_seq, api.MEM_DEVICE)
cp = api.CPairSoa(drows.data_ptr(), dqh.data_ptr())
torch.cuda.synchronize()
ctx = api.Context(0); ctx.set_opts(0.8, 40, med); ctx.set_genome(g)
n_tail = int((rec.tid < 0).sum()); n_tid = len(rec.targets)
pos_bits = max(int(max(ln for _, ln in rec.targets)) + 8192, 2).bit_length() + 1
ic, tc = n // 8 + 65536, n // 16 + 65536
def step():
    ctx.extract_device(cs, cp, n_tail, ic, tc)
    ctx.cluster_resident(n_tid, window, min_support=5, max_clip_dist=mcd, pos_bits=pos_bits, fetch=False)
for _ in range(3): step()
ctx.sync()
MODE = os.environ.get("STEP_MODE", "full")     # full | extract (no clustering) : what each part costs in the overlapped pipeline
if MODE == "extract":
    for K in (50, 200):
        t0 = time.perf_counter()
        for _ in range(K):
            ctx.extract_device(cs, cp, n_tail, ic, tc)
        ctx.sync()
        print("extract only: ms/step %.3f" % ((time.perf_counter() - t0) / K * 1e3))
    sys.exit(0)
for K in (1, 2, 5, 20, 200):
    t0 = time.perf_counter()
    ts = []
    for _ in range(K):
        a = time.perf_counter(); ctx.extract_device(cs, cp, n_tail, ic, tc); b = time.perf_counter()
        ctx.cluster_resident(n_tid, window, min_support=5, max_clip_dist=mcd, pos_bits=pos_bits, fetch=False); c = time.perf_counter()
        ts.append((b - a, c - b))
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print(K, "enqueue ms/step %.3f  total ms/step %.3f" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3), " per call (extract, cluster) us:", [(round(x * 1e6), round(y * 1e6)) for x, y in ts[:6]])
