"""Armed for a box with TWO OR MORE MI355X: everything the N > 1 paths do between devices that one GPU cannot show --
ncclCommInitAll over real devices, a multi-rank ncclAllGather over xGMI, peer DMA (the carried partial record of the
chunk-by-chunk extract; the gather of the shares' per-read state), `strling extract --gpus N` with a context per device fed by
its own thread, `bench.py --gpus 2` on the native exchange.  On a one-GPU box every test here skips (the same code paths run
there with N contexts on the one device: tests/test_cli.py, tests/test_comm_native.py, tests/test_bench_launch.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from strling_amd import api, bamio, build, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = build.CLI


def _n_dev():
    # (through torch, like conftest's check: the library's own hipInit in this process ahead of torch's leaves torch without a device)
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


need2 = pytest.mark.skipif(_n_dev() < 2, reason="needs two GPUs")


def _run(args, **kw):
    return subprocess.run([CLI] + args, capture_output=True, text=True, **kw)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("md")
    rec, g = synth.synth_wgs(40000, seed=91, contig_len=3_000_000)
    bam = str(d / "s.bam")
    bamio.write_bam(bam, rec, block=6000)
    bed = str(d / "ref.fa.str")
    bamio.write_genome_bed(bed, g, rec.targets)
    one = str(d / "one.bin")
    r = _run(["extract", "-g", bed, bam, one])
    assert r.returncode == 0, r.stderr
    return dict(dir=d, bam=bam, bed=bed, one=one, rec=rec, g=g)


@need2
@pytest.mark.parametrize("shares", ["1", "0"])
def test_extract_over_real_devices(sample, shares):
    """a context per DEVICE: shares (own feeder threads; the gather of the per-read state over xGMI, one stream per source) and
    chunk by chunk (strl_front_push_after: the partial record in front of a chunk comes from the previous device by peer DMA)"""
    n = min(_n_dev(), 8)
    for gpus in sorted({2, n}):
        out = str(sample["dir"] / f"dev{gpus}_{shares}.bin")
        r = _run(["extract", "-g", sample["bed"], "-v", "--gpus", str(gpus), sample["bam"], out], env=dict(os.environ, STRL_CHUNK_BLOCKS="24", STRL_SHARES=shares))
        assert r.returncode == 0, r.stderr
        assert f"over {gpus} contexts on {gpus} device(s)" in r.stderr, r.stderr
        assert ("a contiguous share of the file each" in r.stderr) == (shares == "1")
        assert open(out, "rb").read() == open(sample["one"], "rb").read()


@need2
def test_merge_and_call_over_real_devices(sample):
    """`strling merge --gpus N` on N devices: ncclCommInitAll + the tread all-gather on every context's tail stream"""
    one = str(sample["dir"] / "m1")
    r = _run(["merge", "-m", "2", "-o", one, sample["one"]])
    assert r.returncode == 0, r.stderr
    n = min(_n_dev(), 8)
    for gpus in sorted({2, n}):
        pg = str(sample["dir"] / f"m{gpus}")
        r = _run(["merge", "-m", "2", "-v", "--gpus", str(gpus), "-o", pg, sample["one"]])
        assert r.returncode == 0, r.stderr
        assert f"clustered on {gpus} contexts" in r.stderr
        assert open(pg + "-bounds.txt").read() == open(one + "-bounds.txt").read()


@need2
def test_group_exchange_is_rccl_between_devices():
    """strl_ctxs_comm_init over contexts on DIFFERENT devices = ncclCommInitAll; the exchange a real all-gather"""
    from oracle import oracle as O
    from test_comm_native import _extract
    recs = [synth.synth_wgs(12000, seed=600 + r, contig_len=1_500_000) for r in range(2)]
    frag = synth.frag_hist(recs[0][0])
    med, window, mcd = O.median(frag), O.median(frag, 0.99), int(0.5 * O.median(frag, 0.5))
    ctxs = [api.Context(0), api.Context(1)]
    try:
        keep = [_extract(c, r, g, med) for c, (r, g) in zip(ctxs, recs)]
        api.group_comm_init(ctxs)
        assert [c.comm_info() for c in ctxs] == [(2, 0, True), (2, 1, True)]        # (world, rank, rccl)
        n_tid = len(recs[0][0].targets)
        api.group_cluster_exchange(ctxs, n_tid, window, min_support=3, max_clip_dist=mcd, pos_bits=22)
        parts = [c.cluster_collect() for c in ctxs]
        all_t = ctxs[0].exchange_treads()
        assert np.array_equal(all_t, ctxs[1].exchange_treads())
        opts = O.make_opts(med, 0.8, 40)
        exp_t = np.concatenate([O.extract(r, g, opts) for r, g in recs])
        for f in ("tid", "position", "repeat", "flag", "split", "repeat_count"):
            assert np.array_equal(all_t[f], exp_t[f]), f
        eb, _ = O.call_bounds(exp_t, 1, window, min_support=3, max_clip_dist=mcd)
        got = sorted(api.bounds_row(x, "c") for p in parts for x in p[0])
        assert got == sorted(O.bounds_row(x, "c") for x in eb) and len(eb) > 10
        assert len(keep) == 2
    finally:
        for c in ctxs:
            c.close()


@need2
def test_bench_two_ranks_native_exchange_and_file_leg():
    """`python bench.py --gpus 2` on two devices: one rank per GPU over RCCL, the exchange the library's own (`native`), and the
    end_to_end leg = `strling extract --gpus 2` on a file"""
    env = dict(os.environ, BENCH_E2E_SMALL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reads-per-gpu", str(2 ** 20),
                        "--no-cpu-baseline", "--e2e-pairs", str(2 ** 20)], capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["exchange"] == "native" and out["rccl_ranks"] == 2 and out["backend"] == "nccl"
    e = out["end_to_end"]
    assert e and e.get("gpus") == 2 and e["check"]["ok"], e
    assert any("a contiguous share" in (x.get("gather") or "") for x in e["runs"])


def test_file_leg_of_a_two_rank_bench_on_whatever_devices_there_are():
    """the N > 1 end_to_end leg (rank 0 runs `strling extract --gpus N`) -- on a one-GPU box the N contexts share the device"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_bench
    inp = e2e_bench.make_input(2 ** 19, level=1)
    try:
        res = e2e_bench.run(inp, CLI, repeats=1, gpus=2)
        assert "error" not in res, res
        assert res["gpus"] == 2 and res["runs"][0]["rc"] == 0
        assert res["runs"][0]["shares"] and len(res["runs"][0]["shares"]) == 2, res["runs"][0]
        chk = e2e_bench.check_in_subprocess(inp, e2e_bench.pick_slabs(inp["n_slabs"], 2), call=res.get("call_rc") == 0)
        assert chk["ok"], chk
    finally:
        e2e_bench.cleanup(inp)


@need2
def test_one_process_per_sample_on_its_own_device(sample):
    """pipelines/bpipe.config:4: one `strling` process per sample.  `--device K` / STRL_DEVICE put a process' context on device K
    (`--gpus N --device K`: K, K + 1, ...); two concurrent extractions on two devices both write the one-device .bin; the
    replica leg of bench.py (tools/e2e_bench.py::replicas) over the real devices"""
    n = _n_dev()
    procs = []
    for k in range(min(n, 8)):
        out = str(sample["dir"] / f"rep{k}.bin")
        procs.append((k, out, subprocess.Popen([CLI, "extract", "-g", sample["bed"], "-v", "--device", str(k), sample["bam"], out], stderr=subprocess.PIPE, text=True)))
    for k, out, p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err
        assert f"1 context(s) on device(s) {k} of {n}" in err, err
        assert open(out, "rb").read() == open(sample["one"], "rb").read()
    out = str(sample["dir"] / "g2d1.bin")
    r = _run(["extract", "-g", sample["bed"], "-v", "--gpus", "2", "--device", str(n - 1), sample["bam"], out])
    assert r.returncode == 0 and f"2 context(s) on device(s) {n - 1} 0 of {n}" in r.stderr, r.stderr
    assert open(out, "rb").read() == open(sample["one"], "rb").read()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import e2e_bench
    inp = dict(bam=sample["bam"], bed=sample["bed"], out=sample["one"], reads=sample["rec"].n)
    res = e2e_bench.replicas(inp, CLI, min(n, 8), n)
    assert res["bins_identical_to_the_single_run"] and res["aggregate_reads_per_s"] and {r["device"] for r in res["per_replica"]} == set(range(min(n, 8)))
