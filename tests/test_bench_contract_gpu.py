"""bench.py prints ONE JSON line with the fields the driver reads (tiny workload, one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--taxa", "24", "--patterns", "3000", "--steps", "2", "--warmup", "1",
                          "--cpu-sample-patterns", "300", "--cpu-budget-s", "1"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "evals/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # achieved / frac are MEASURED HBM bytes (PMC) over the live kernel time: only present when profiles/traffic_latest.json
    # describes this very shape -- never for this tiny workload; the three-pass ratio has its own name and is not a fraction
    assert r["achieved"] is None and r["frac"] is None and r["traffic"] is None
    assert r["vs_three_pass"]["GB/s"] > 0 and "valu" in r
    assert "NOT the headline shape" in d["metric"] and "NOT de-duplicated" in d["config"]["workload"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0


def test_bench_secondary_modes():
    assert _run(["--no-cpu-baseline", "--subst-gradient"])["value"] > 0
    d = _run(["--no-cpu-baseline", "--rescale", "always"])
    assert d["config"]["rescaling"] is True


def test_bench_two_rank_launch_path():
    """the driver's N > 1 launch line (torch.distributed.run, one rank per GPU) rehearsed on one GPU: both ranks use cuda:0
    and a gloo group (PHYAMD_BENCH_REHEARSAL); rank 0 prints the one line, with the whole job's value and n_gpus = 2, and
    the sharded lnL equals the single-rank one"""
    one = _run(["--no-cpu-baseline"])
    env = dict(os.environ, PHYAMD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--taxa", "24", "--patterns", "3000", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    from physher_amd.sharding import shard_range
    lo, hi = shard_range(3000, 0, 2)  # whole 64-pattern blocks by bisection: rank 0 holds a subtree of the one-GPU summation
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["patterns_per_gpu"] == hi - lo == 1472
    assert d["config"]["lnL"] == one["config"]["lnL"]  # ... so the sum of the two ranks is bit for bit the one-rank result
