"""Seam A: the REFERENCE's own object graph -- JSON model, SingleTreeLikelihood, gradient epilogue, its
C++ wrapper -- evaluated on the HIP engine through integration/physher_device.c.

Everything executed here was built by oracle/Makefile (target `device`) from the reference's sources where they lie, into
oracle/_ref/ (prebuilt files travel to the GPU box; /root/reference does not and is not read):
  libphyc_ref.so            the compiled reference
  libphysher_device.so      the binding (this repository's code against the reference's headers), in front of libphyc
  ref_driver                this repository's dump driver over the reference's API
  test_tree_likelihood_ref  the reference's own known-answer test, unmodified
  libphycpp_ref.so          the reference's own C++ wrapper, unmodified; phycpp_usage_ref = tests/cpp/phycpp_usage.cpp over it
Expected values are the committed fixtures (outputs of the reference's CPU path) and the constants inside the reference's test.
"""
import gzip
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from golden_util import GOLDEN, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
SHIM = os.path.join(REFDIR, "libphysher_device.so")
DRIVER = os.path.join(REFDIR, "ref_driver")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(SHIM) and os.path.exists(DRIVER)), reason="oracle/_ref is not built (needs the reference tree)")]

SPEC_CASES = sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "spec.txt")))
JSON_CASES = {"fluA_jc69_time": "jc69-time.json", "fluA_hky_g4_time": "hky-g4-time.json", "fluA_hky_g4_branch_rates": "hky-g4-branch-rates.json"}


def device_env(**extra):
    env = dict(os.environ)
    env["LD_PRELOAD"] = SHIM  # the binding in front of libphyc in the symbol lookup order
    env["PHYSHER_DEVICE_VERBOSE"] = "1"
    env.update({k: str(v) for k, v in extra.items()})
    return env


def device_work(stderr):
    m = re.search(r"physher device backend: (\d+) likelihood, (\d+) gradient, (\d+) single-branch evaluations on the device", stderr)
    assert m, "the binding did not report: " + stderr[-2000:]
    return tuple(int(x) for x in m.groups())


def run(cmd, cwd, env):
    out = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, f"{cmd}: rc {out.returncode}\n{out.stdout[-3000:]}\n{out.stderr[-3000:]}"
    return out


def close_where_finite(got, want, tol, what, underflows=False):
    """underflows: the reference's rescaled gradient divides by one category's own site likelihood and is NaN where that
    underflows (60 % of the branches of gtr_g4_t700_autorescale); the engine's reproduction of that
    arithmetic underflows in slightly different places, so entries are compared where both are finite (at least half of the
    reference's finite ones) and to the looser tolerance tests/test_engine_gpu.py uses for the same case."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    ok = np.isfinite(want)
    if underflows:
        both = ok & np.isfinite(got)
        assert both.sum() >= 0.5 * ok.sum(), (what, both.sum(), ok.sum())
        ok = both
    scale = max(1.0, np.abs(want[ok]).max()) if ok.any() else 1.0
    err = np.abs(got[ok] - want[ok]).max() if ok.any() else 0.0
    assert err <= tol * scale, f"{what}: max |diff| {err:.3e} > {tol:g} * {scale:.3e}"


@pytest.mark.parametrize("case", SPEC_CASES)
def test_reference_object_graph_on_device_matches_cpu_fixture(case, tmp_path):
    """ref_driver builds the model exactly as for the fixture (JSON -> new_TreeLikelihoodModel_from_json, or the C
    constructors for MG94) -- PHYSHER_DEVICE=1 moves the hot path to the GPU; lnL, per-pattern lnL, the TREE_MODEL gradient
    and the full (tree + site model + substitution model) gradient must equal the CPU reference's."""
    gold = load(case)
    out_json = tmp_path / "device.json"
    out = run([DRIVER, "dump", "spec.txt", str(out_json)], os.path.join(GOLDEN, case), device_env(PHYSHER_DEVICE=1))
    lik, grad, _ = device_work(out.stderr)
    assert lik >= 1 and grad >= 2, (lik, grad)
    with open(out_json) as f:
        dev = json.load(f)
    assert abs(dev["lnl"] - gold["lnl"]) <= 1e-10 * abs(gold["lnl"]), (dev["lnl"], gold["lnl"])
    # the same allowances as tests/test_engine_gpu.py::test_golden_generic_states makes for the same fixtures: codon P(t)
    # entries of multi-nucleotide changes come out of sums that cancel to 1e-17 absolute (relative accuracy ~1e-7 in the
    # reference and here alike); the reference's rescaled multi-category gradient divides by per-category likelihoods
    codon = gold["state_count"] > 20
    multi = gold["rescaled"] and gold["category_count"] > 1
    lossy = case == "gtr_g4_t700_autorescale"
    np.testing.assert_allclose(np.array(dev["pattern_lk"]), gold["pattern_lk"], rtol=1e-9 if codon else 1e-10, atol=1e-10)
    factor = (10 if codon else 1) * (1e3 if multi else 1)
    close_where_finite(dev["gradient_tree"], gold["gradient_tree"], 1e-5 if lossy else 1e-9 * factor, "gradient_tree", underflows=lossy or multi)
    close_where_finite(dev["gradient_all"], gold["gradient_all"], 1e-5 if lossy else 1e-8 * factor, "gradient_all", underflows=lossy or multi)


TRAIT_CASES = sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "trait_spec.txt")))


@pytest.mark.parametrize("case", TRAIT_CASES)
def test_state_counts_without_kernels_are_padded(case, tmp_path):
    """Discrete-trait models (general data type: 2, 5 and 7 states, named ambiguity sets, with and without gamma categories) built by
    the reference's own constructors (ref_driver attr = physher.cpp:53-75, 321-350, 594-629).  The engine has kernels for 4, 20,
    60 and 61 states; the binding pads other counts with inert states, as the reference's generic kernels take any count
    (treelikelihoodX.c:43-576): lnL and every gradient block must equal the CPU reference's."""
    gold = load(case)
    out_json = tmp_path / "device.json"
    out = run([DRIVER, "attr", "trait_spec.txt", str(out_json)], os.path.join(GOLDEN, case), device_env(PHYSHER_DEVICE=1))
    lik, grad, _ = device_work(out.stderr)
    assert lik >= 1 and grad >= 1, (lik, grad)
    with open(out_json) as f:
        dev = json.load(f)
    assert dev["state_count"] == gold["state_count"] and gold["state_count"] not in (4, 20, 60, 61)
    assert abs(dev["lnl"] - gold["lnl"]) <= 1e-10 * abs(gold["lnl"]), (dev["lnl"], gold["lnl"])
    np.testing.assert_allclose(np.array(dev["pattern_lk"]), gold["pattern_lk"], rtol=1e-10, atol=1e-10)
    close_where_finite(dev["gradient_tree"], gold["gradient_tree"], 1e-9, "gradient_tree")
    close_where_finite(dev["gradient_all"], gold["gradient_all"], 1e-8, "gradient_all")


@pytest.mark.parametrize("case", ["jc69_t12", "gtr_g4_t16"])
def test_brent_call_pattern_on_device(case, tmp_path):
    """The optimiser's fast path through the binding (serial_brent_optimize_tree's call pattern, optimizer.c:112-153; ref_driver
    brent): use_upper on, three trial lengths per branch as single-branch device evaluations, every last trial kept, then the plain
    lnL and the TREE_MODEL gradient.  A closed-form model (JC69: every node carries the model's own P(t), which must follow an
    accepted length) and the eigen route, against what the reference's CPU path printed for the same calls."""
    with open(os.path.join(GOLDEN, case, "brent_trials.json")) as f:
        gold = json.load(f)
    out_json = tmp_path / "brent.json"
    out = run([DRIVER, "brent", "spec.txt", str(out_json)], os.path.join(GOLDEN, case), device_env(PHYSHER_DEVICE=1))
    lik, grad, branch = device_work(out.stderr)
    assert branch >= len(gold["trials"]) // 2 and grad >= 1, (lik, grad, branch)
    with open(out_json) as f:
        dev = json.load(f)
    assert len(dev["trials"]) == len(gold["trials"])
    for got, want in zip(dev["trials"], gold["trials"]):
        assert got["node"] == want["node"] and got["length"] == want["length"]
        assert abs(got["lnl"] - want["lnl"]) <= 1e-9 * abs(want["lnl"]), (got, want)
    for key in ("lnl_start", "lnl_end", "lnl_end_recomputed"):
        assert abs(dev[key] - gold[key]) <= 1e-10 * abs(gold[key]), (key, dev[key], gold[key])
    close_where_finite(dev["gradient_tree_end"], gold["gradient_tree_end"], 1e-9, "gradient after the accepted lengths")


@pytest.mark.parametrize("case", sorted(JSON_CASES))
def test_json_device_key(case, tmp_path):
    """The reference's JSON surface with one more key: "device": true inside the treelikelihood node
    (new_TreeLikelihoodModel_from_json, treelikelihood.c:819-942).  fluA_jc69_time is the model file of the reference's own
    test; the other two add HKY + G4 and one clock rate per branch (ratio / clock / site / substitution gradient blocks)."""
    src = os.path.join(GOLDEN, case)
    with open(os.path.join(src, JSON_CASES[case])) as f:
        doc = json.load(f)
    doc["model"]["device"] = True
    with open(tmp_path / "model.json", "w") as f:
        json.dump(doc, f, indent=1)
    shutil.copy(os.path.join(src, "fluA.fa"), tmp_path / "fluA.fa")
    out = run([DRIVER, "json", "model.json", str(tmp_path / "device.json")], str(tmp_path), device_env())  # no PHYSHER_DEVICE: the key decides
    lik, grad, _ = device_work(out.stderr)
    assert lik >= 2 and grad >= 2, (lik, grad)
    with open(tmp_path / "device.json") as f:
        dev = json.load(f)
    with gzip.open(os.path.join(src, "expected.json.gz"), "rt") as f:
        gold = json.load(f)
    for jac in (0, 1):
        assert abs(dev[f"lnl_jacobian{jac}"] - gold[f"lnl_jacobian{jac}"]) <= 1e-10 * abs(gold[f"lnl_jacobian{jac}"])
        # the reference's own test accepts 1e-8 absolute on these (tests/test_tree_likelihood.c:31-131); the clock entry is ~3e5
        close_where_finite(dev[f"gradient_tree_clock_jacobian{jac}"], gold[f"gradient_tree_clock_jacobian{jac}"], 1e-10, f"gradient jacobian{jac}")
    if "gradient_all_time" in gold:
        close_where_finite(dev["gradient_all_time"], gold["gradient_all_time"], 1e-8, "gradient_all_time")
    np.testing.assert_allclose(np.array(dev["pattern_lk"]), np.array(gold["pattern_lk"]), rtol=1e-10, atol=1e-10)


def test_json_device_false_stays_on_cpu(tmp_path):
    src = os.path.join(GOLDEN, "fluA_jc69_time")
    with open(os.path.join(src, "jc69-time.json")) as f:
        doc = json.load(f)
    doc["model"]["device"] = False
    with open(tmp_path / "model.json", "w") as f:
        json.dump(doc, f)
    shutil.copy(os.path.join(src, "fluA.fa"), tmp_path / "fluA.fa")
    out = run([DRIVER, "json", "model.json", str(tmp_path / "cpu.json")], str(tmp_path), device_env(PHYSHER_DEVICE=1))  # the key wins over the environment
    assert "evaluations on the device" not in out.stderr
    with open(tmp_path / "cpu.json") as f:
        assert abs(json.load(f)["lnl_jacobian0"] - (-4777.616349713985)) < 1e-8


def test_reference_known_answer_test_passes_on_device():
    """tests/test_tree_likelihood.c of the reference, compiled unmodified: lnL with and without the Jacobian, the clock
    gradient and 67 + 1 ratio / root-height gradient entries at 1e-8 -- computed by the HIP engine."""
    exe = os.path.join(REFDIR, "test_tree_likelihood_ref")
    if not os.path.exists(exe):
        pytest.skip("test_tree_likelihood_ref is not built")
    out = run([exe], os.path.join(GOLDEN, "fluA_jc69_time"), device_env(PHYSHER_DEVICE=1))
    assert "ALL TESTS" in out.stdout and "PASSED" in out.stdout, out.stdout[-2000:]
    assert "FAILED" not in out.stdout
    lik, grad, _ = device_work(out.stderr)
    assert lik >= 2 and grad >= 2, (lik, grad)


def test_reference_phycpp_wrapper_on_device():
    """The reference's own src/phycpp/physher.cpp (compiled unmodified into libphycpp_ref.so) under a caller written against
    physher.hpp: TreeLikelihoodInterface never sees JSON, PHYSHER_DEVICE moves it (new_TreeLikelihoodModel)."""
    exe = os.path.join(REFDIR, "phycpp_usage_ref")
    if not os.path.exists(exe):
        pytest.skip("phycpp_usage_ref is not built")
    case = os.path.join(GOLDEN, "gtr_g4_t16")
    cpu = run([exe, "aln.fa", "tree.nwk"], case, dict(os.environ))
    dev = run([exe, "aln.fa", "tree.nwk"], case, device_env(PHYSHER_DEVICE=1))
    lik, grad, _ = device_work(dev.stderr)
    assert lik >= 2 and grad >= 1, (lik, grad)

    def parse(text):
        lines = text.strip().splitlines()
        n = int(lines[1].split()[1])
        return float(lines[0].split()[1]), np.array([float(x) for x in lines[2:2 + n]]), float(lines[2 + n].split()[1])

    l0, g0, l0b = parse(cpu.stdout)
    l1, g1, l1b = parse(dev.stdout)
    gold = load("gtr_g4_t16")
    assert abs(l1 - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    assert abs(l1 - l0) <= 1e-10 * abs(l0) and abs(l1b - l0b) <= 1e-10 * abs(l0b)
    assert g0.shape == g1.shape
    assert np.abs(g1 - g0).max() <= 1e-8 * max(1.0, np.abs(g0).max())


@pytest.mark.parametrize("case", ["gtr_g4_t16", "wag_g4_t12", "gtr_g4_t96_rescale"])
def test_two_shards_through_the_binding(case, tmp_path):
    """PHYSHER_DEVICE=2: the binding asks for phyamd_create_sharded; with one GPU on the box both shards sit on device 0."""
    gold = load(case)
    out = run([DRIVER, "dump", "spec.txt", str(tmp_path / "device.json")], os.path.join(GOLDEN, case),
              device_env(PHYSHER_DEVICE=2, PHYSHER_DEVICE_IDS="0,0"))
    device_work(out.stderr)
    with open(tmp_path / "device.json") as f:
        dev = json.load(f)
    assert abs(dev["lnl"] - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    np.testing.assert_allclose(np.array(dev["pattern_lk"]), gold["pattern_lk"], rtol=1e-10, atol=1e-10)
    close_where_finite(dev["gradient_tree"], gold["gradient_tree"], 1e-9, "gradient_tree")
    close_where_finite(dev["gradient_all"], gold["gradient_all"], 1e-8, "gradient_all")


def _synthetic_spec(tmp_path, T, L, tipstates, seed=5):
    from physher_amd import synth
    rng = np.random.default_rng(seed)
    tree = synth.random_tree(T, rng)
    states = np.ascontiguousarray(synth.evolve(tree, L, 4, rng))
    (tmp_path / "aln.fa").write_text(synth.to_fasta(tree.names, states, "nucleotide"))
    (tmp_path / "tree.nwk").write_text(tree.newick() + "\n")
    (tmp_path / "spec.txt").write_text(f"fasta {tmp_path}/aln.fa\nnewick {tmp_path}/tree.nwk\ndatatype nucleotide\nmodel gtr\nrates 1.2,3.1,0.7,0.9,2.8\n"
                                       f"freqs 0.3,0.2,0.2,0.3\ncategories 4\nalpha 0.5\ntipstates {tipstates}\nsse 1\n")
    return str(tmp_path / "spec.txt")


def test_device_objects_are_built_without_host_partial_arrays(tmp_path):
    """The wrapper's default tip mode ("tipstates" off): the reference's constructor fills one [C][P][S] array per tip
    (treelikelihood.c:1106-1117: 128 GB at the headline shape).  With the device asked for, the binding's allocate_storage hands
    every tip ONE shared scratch array and internal nodes none: the process's peak resident memory stays far below the full
    arrays' size (PHYSHER_DEVICE_LEAN=0 = the reference's own storage, same numbers)."""
    T, L = 120, 60000  # tips alone: 120 x 60000 x 16 doubles = 0.92 GB
    spec = _synthetic_spec(tmp_path, T, L, 0)
    res = {}
    for lean in ("1", "0"):
        out = run([DRIVER, "bench", spec, "2", "1"], str(tmp_path), device_env(PHYSHER_DEVICE=1, PHYSHER_DEVICE_LEAN=lean))
        r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        res[lean] = (r["lnl"], r["peak_rss_kb"] * 1024 / 1e9)  # VmHWM of the driver process itself
        lik, grad, _ = device_work(out.stderr)
        assert lik >= 2 and grad >= 2
    assert res["1"][0] == res["0"][0]
    assert res["0"][1] > 0.9, res  # the reference's own storage: every tip array touched
    assert res["1"][1] < res["0"][1] - 0.7, res


@pytest.mark.parametrize("tipstates", [0, 1])
def test_disabling_the_device_on_a_live_lean_object_restores_the_cpu_path(tmp_path, tipstates):
    spec = _synthetic_spec(tmp_path, 30, 400, tipstates, seed=8)
    out = run([DRIVER, "toggle", spec, str(tmp_path / "t.json")], str(tmp_path), device_env(PHYSHER_DEVICE=1))
    d = json.loads((tmp_path / "t.json").read_text())
    assert d["on_device_before"] == 1 and d["on_device_after"] == 0
    assert abs(d["lnl0"] - d["lnl1"]) <= 1e-10 * abs(d["lnl1"])
    g0, g1 = np.array(d["gradient0"]), np.array(d["gradient1"])
    assert np.abs(g0 - g1).max() <= 1e-9 * max(1.0, np.abs(g1).max())


def test_patched_reference_runs_on_the_device_without_interposition():
    """integration/physher-device.patch: the reference as a STATIC library with the hook table, the binding built with
    -DPHYSHER_DEVICE_PATCHED -- no LD_PRELOAD, no interposable symbol -- and the reference's own known-answer test on the engine."""
    exe = os.path.join(REFDIR, "test_tree_likelihood_patched")
    if not os.path.exists(exe):
        pytest.skip("test_tree_likelihood_patched is not built")
    env = dict(os.environ, PHYSHER_DEVICE="1", PHYSHER_DEVICE_VERBOSE="1")
    env.pop("LD_PRELOAD", None)
    out = run([exe], os.path.join(GOLDEN, "fluA_jc69_time"), env)
    assert "ALL TESTS" in out.stdout and "PASSED" in out.stdout and "FAILED" not in out.stdout, out.stdout[-2000:]
    lik, grad, _ = device_work(out.stderr)
    assert lik >= 2 and grad >= 2, (lik, grad)
