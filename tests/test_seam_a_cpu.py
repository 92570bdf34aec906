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
