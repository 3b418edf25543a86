"""The line bench.py prints: run as the driver runs it (a child process, rank 0 prints ONE JSON line), at a reduced size so that the
CPU leg takes seconds; the keys the driver and the judge read are there and consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract_configs3_reduced():
    d = _run(["--n", "2000000", "--cpu-sample", "2000000"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "weights/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    # value = weights of all ranks / time of exactly `steps` steps
    assert abs(d["value"] - 2_000_000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["k"] == 257
    assert cfg["input"].startswith("a batch of its own per step")          # every step had its own resident batch
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 10 * 2_000_000             # 4 B read + 2 B index + 4 B value a weight
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "weights/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    p = d["parity"]
    assert p["mask_equal"] is True and p["n_iter_gpu"] >= 1 and p["n_iter_cpu"] >= 1


@pytest.mark.gpu
def test_bench_input_pool_fallback_says_so():
    d = _run(["--n", "1000000", "--no-cpu-baseline", "--no-streaming-leg", "--input-pool-gb", "0"])
    assert d["config"]["input"].startswith("one resident vector, copied")
