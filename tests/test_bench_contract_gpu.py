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


def test_bench_line_carries_step_statistics_and_collective_fields():
    d = _run(["--no-cpu-baseline"])
    s = d["step_ms"]
    assert s["min"] <= s["median"] <= s["max"] and s["min"] > 0
    c = d["config"]["collective"]
    assert c["world"] == 1 and c["backend"] is None and c["all_reduce_us"] is None and c["host_epilogue_us"] > 0
    assert c["rank_ms_per_step"]["min"] == c["rank_ms_per_step"]["max"] > 0


def test_bench_cpu_baseline_checks_the_timed_engine_against_the_reference():
    """cpu_baseline.gradient_check: per-pattern lnL and the branch gradient of the TIMED engine against the reference's own values
    for the baseline's shards (present when oracle/_ref is built; a miss makes bench.py exit non-zero, which _run asserts against)"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_driver")):
        pytest.skip("oracle/_ref is not built (needs the reference tree)")
    d = _run([])
    gc = d["cpu_baseline"]["gradient_check"]
    assert gc["ok"] and len(gc["shards"]) >= 1
    for sh in gc["shards"]:
        assert sh["ok"] and sh["per_pattern_lnl_max_rel_err"] <= 1e-11 and sh["lnl_rel_err"] <= 1e-10
        assert sh["gradient_max_abs_err"] <= 1e-9 * max(1.0, sh["gradient_inf_norm"]) and sh["branches_compared"] == 2 * 24 - 3


def test_bench_rccl_path_in_a_world_of_one():
    """The N > 1 code path with the REAL backend on one GPU: launched as the driver launches N ranks (torch.distributed.run, here
    --nproc-per-node 1), --force-collective creates the RCCL ("nccl") process group and runs the all-reduce of the result vector
    on the engine's stream in a world of one.  (A child process: nothing here re-executes a process that has touched the GPU.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--taxa", "24", "--patterns", "3000",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    c = d["config"]["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["kind"] == "all_reduce(SUM)" and c["bytes"] == 8 * (1 + 47 * 4)
    assert c["all_reduce_us"] is not None and c["all_reduce_us"] > 0 and c["host_epilogue_us"] > 0
    one = _run(["--no-cpu-baseline"])
    assert d["config"]["lnL"] == one["config"]["lnL"] and d["n_gpus"] == 1  # the collective of one rank changes nothing
