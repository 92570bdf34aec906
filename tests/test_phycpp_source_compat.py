"""A C++ caller written against the reference's wrapper classes (tests/cpp/phycpp_usage.cpp): compiles and links against
libphycpp_amd.so on the CPU; on the GPU box it runs and reproduces the golden lnL and gradient of gtr_g4_t16."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from golden_util import GOLDEN, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "physher_amd")


def _build(tmp_path):
    exe = tmp_path / "phycpp_usage"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "physher_amd", "csrc", "host"),
                           "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "phycpp_usage.cpp"),
                           "-L" + LIBDIR, "-lphycpp_amd", "-lphysher_amd", "-Wl,-rpath," + LIBDIR])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists(os.path.join(LIBDIR, "libphycpp_amd.so")), reason="needs g++ and the built host library")
def test_reference_style_caller_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_reference_style_caller_runs(tmp_path):
    exe = _build(tmp_path)
    case = os.path.join(GOLDEN, "gtr_g4_t16")
    out = subprocess.run([str(exe), os.path.join(case, "aln.fa"), os.path.join(case, "tree.nwk")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    gold = load("gtr_g4_t16")
    lnl = float(lines[0].split()[1])
    n = int(lines[1].split()[1])
    g = np.array([float(x) for x in lines[2: 2 + n]])
    N = gold["node_count"]
    assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    assert n == N - 2 + 1 + 9
    np.testing.assert_allclose(g[N - 2:], gold["gradient_all"][N:], rtol=2e-7, atol=1e-7)  # shape, 5 rates, 4 frequencies
    assert abs(g[: N - 2] - gold["gradient_all"][: N - 2]).max() <= 1e-9 * np.abs(gold["gradient_all"]).max()
    lnl2 = float(lines[2 + n].split()[1])
    assert np.isfinite(lnl2) and abs(lnl2 - lnl) > 1e-3
