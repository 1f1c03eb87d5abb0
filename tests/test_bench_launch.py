"""bench.py's own launch of N ranks (round-3 review: `python bench.py --gpus N` used to run ONE rank and report n_gpus 1).
CPU half: the parent re-executes itself under torch.distributed.run with the contract's arguments.  GPU half: the N > 1
path end to end with two ranks on the one device a test box has (gloo rendezvous, exchange staged through the host: a dry
run of the code path -- RCCL refuses two ranks per device -- not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_execs_torch_distributed_run(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, argv
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in a and "--nproc-per-node=4" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert a[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and a[-7].endswith("bench.py")


def test_single_rank_does_not_relaunch(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    monkeypatch.setattr(os, "execv", lambda *a: (_ for _ in ()).throw(AssertionError("must not exec")))
    monkeypatch.setenv("WORLD_SIZE", "2")           # already a rank of somebody's launch (the driver's torch.distributed.run)
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--reads-per-gpu", "0", "--no-e2e"])
    # the rank goes on past the launch block; stop it at the first thing behind it (the synthetic batch of 0 reads)
    with pytest.raises((Exception, SystemExit)):
        bench.main()


@pytest.mark.gpu
def test_bench_gpus_2_on_one_device():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads-per-gpu", str(2 ** 18),
                        "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["exchange"] in ("native", "torch") and out["config"]["reads_per_gpu"] == 2 ** 18
    assert out["value"] > 0 and "all-gather" in out["config"]["parallelism"]
