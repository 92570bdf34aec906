"""CPU-only checks of the seam-A binding (integration/physher_device.c, built into oracle/_ref/ where the reference tree
exists): it exports what INTEGRATION.md says, leaves objects that were not moved to the device exactly on the reference's CPU
path, and fails loudly -- never a silent CPU run -- when a device is asked for and there is none."""
import json
import os
import subprocess

import pytest

from golden_util import GOLDEN, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
SHIM = os.path.join(REFDIR, "libphysher_device.so")
DRIVER = os.path.join(REFDIR, "ref_driver")

pytestmark = pytest.mark.skipif(not (os.path.exists(SHIM) and os.path.exists(DRIVER)), reason="oracle/_ref is not built (needs the reference tree)")


def test_binding_exports_the_hooks_and_the_control_functions():
    out = subprocess.run(["nm", "-D", "--defined-only", SHIM], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    for sym in ("SingleTreeLikelihood_enable_device", "SingleTreeLikelihood_disable_device", "SingleTreeLikelihood_on_device",
                "SingleTreeLikelihood_device_is_rescaling",
                # the reference's functions it stands in front of (treelikelihood.c)
                "update_upper_partials", "gradient_cat_branch_lengths", "calculate_dlnl_dQ", "gradient_pinv_sitemodel", "gradient_pinv_W_sitemodel",
                "free_SingleTreeLikelihood_internals", "new_TreeLikelihoodModel", "new_TreeLikelihoodModel_from_json"):
        assert sym in names, sym
    undefined = subprocess.run(["nm", "-D", "--undefined-only", SHIM], capture_output=True, text=True, check=True).stdout
    for sym in ("phyamd_create", "phyamd_create_sharded", "phyamd_gradient", "phyamd_parameter_gradient", "phyamd_branch_log_likelihood", "phyamd_store"):
        assert sym in undefined, sym  # the binding talks to the product through the C ABI only


def test_preloaded_binding_forwards_untouched_objects_to_the_cpu_path(tmp_path):
    """LD_PRELOAD without PHYSHER_DEVICE / "device": every hooked function must hand over to the reference's own."""
    gold = load("gtr_g4i_mu_t14")  # site-model gradient (+I root term) and substitution gradient go through the hooks
    env = dict(os.environ, LD_PRELOAD=SHIM)
    env.pop("PHYSHER_DEVICE", None)
    out = tmp_path / "cpu.json"
    subprocess.run([DRIVER, "dump", "spec.txt", str(out)], cwd=os.path.join(GOLDEN, "gtr_g4i_mu_t14"), env=env, check=True, capture_output=True, timeout=300)
    with open(out) as f:
        got = json.load(f)
    assert got["lnl"] == gold["lnl"]
    assert got["gradient_all"] == list(gold["gradient_all"]) and got["gradient_tree"] == list(gold["gradient_tree"])


def test_requested_device_without_gpu_is_a_loud_error(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = dict(os.environ, LD_PRELOAD=SHIM, PHYSHER_DEVICE="1")
    out = subprocess.run([DRIVER, "dump", "spec.txt", str(tmp_path / "x.json")], cwd=os.path.join(GOLDEN, "gtr_g4_t16"), env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 2
    assert "device was requested and could not be enabled" in out.stderr


# ---- the maintainer's form: integration/physher-device.patch (a hook table instead of symbol interposition) ---------------------

PATCH = os.path.join(ROOT, "integration", "physher-device.patch")
PATCHED_TEST = os.path.join(REFDIR, "test_tree_likelihood_patched")
REF = os.environ.get("PHYSHER_REF", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "phyc")), reason="the reference tree is not here")
def test_patch_is_current_and_applies_to_the_reference(tmp_path):
    """the committed patch is what integration/make_patch.py produces from the reference as it lies, and `patch` applies it to a
    scratch copy of the two files it touches without fuzz or rejects"""
    import sys
    made = subprocess.run([sys.executable, os.path.join(ROOT, "integration", "make_patch.py"), REF], capture_output=True, text=True, check=True).stdout
    with open(PATCH) as f:
        assert f.read() == made
    scratch = tmp_path / "src" / "phyc"
    scratch.mkdir(parents=True)
    for name in ("treelikelihood.c", "treelikelihood.h"):
        (scratch / name).write_bytes(open(os.path.join(REF, "src", "phyc", name), "rb").read())
    out = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0 and "fuzz" not in out.stdout and "FAILED" not in out.stdout, out.stdout + out.stderr
    hooks = [ln for ln in open(PATCH) if ln.startswith("+\tDEVICE_HOOK")]
    assert len(hooks) == 11, hooks


@pytest.mark.skipif(not os.path.exists(PATCHED_TEST), reason="oracle/_ref/test_tree_likelihood_patched is not built (make -C oracle patched)")
def test_patched_reference_passes_its_own_test_through_the_hook_table():
    """The reference's unmodified tests/test_tree_likelihood.c linked against the PATCHED reference as a static library and the
    binding built with -DPHYSHER_DEVICE_PATCHED: nothing is interposed (the binding defines none of the reference's symbols), every
    hooked call goes hook -> binding -> back into the function's own body, and the known answers hold on the CPU path."""
    shim = os.path.join(REFDIR, "libphysher_device_patched.so")
    names = {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--defined-only", shim], capture_output=True, text=True, check=True).stdout.splitlines() if ln.strip()}
    for sym in ("gradient_cat_branch_lengths", "update_upper_partials", "new_TreeLikelihoodModel", "allocate_storage", "free_SingleTreeLikelihood_internals"):
        assert sym not in names, sym
    assert "pd_gradient_cat_branch_lengths" in names and "SingleTreeLikelihood_enable_device" in names
    env = dict(os.environ)
    env.pop("PHYSHER_DEVICE", None)
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([PATCHED_TEST], cwd=os.path.join(GOLDEN, "fluA_jc69_time"), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL TESTS" in out.stdout and "PASSED" in out.stdout and "FAILED" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
