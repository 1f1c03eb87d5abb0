"""bench.py prints ONE JSON line with the contract's keys (run on a reduced batch so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--reads-per-gpu",
                        str(2 ** 21), "--cpu-seconds", "1", "--e2e-pairs", str(2 ** 20)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["reads_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["kernel"] in rf["kernel_ms"] and rf["kernel_ms"][rf["kernel"]] == max(rf["kernel_ms"].values())
    e2e = d["end_to_end"]
    assert e2e["reads"] == 2 ** 21 and e2e["call_rc"] == 0 and e2e["merge_rc"] == 0 and e2e["check"]["ok"] and e2e["check"]["treads_checked"] > 1000
    assert e2e["value"] > 1e6 and e2e["reads_per_s_extract_plus_call"] > 1e5 and e2e["runs"][0]["device_mem_GB"] > 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 1e5
