"""The hot kernels' register budgets, read from the code object of the built library (profiles/kernel_resources.py): the
occupancies DESIGN.md section 3 relies on -- four waves per SIMD for the streamed pre-order walk (<= 128 registers, nothing
spilled; three for its rescaled / ambiguity instantiations), five for the streamed post-order walk (<= 96), seven for the table-gather one (<= 72), four for the 20-state walk -- are a build-time property, checked without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "physher_amd", "libphysher_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("built library or llvm tools missing")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "kernel_resources.py"), "k_"], check=True, capture_output=True, text=True).stdout
    table = json.loads(out)
    assert len(table["library_sha256"]) == 64
    return table["kernels"]


@pytest.mark.parametrize("name,max_vgpr", [("k_upper4_stream<false, 0, false, true>", 128), ("k_upper4_stream<true, 0, false, true>", 128),
                                           ("k_upper4_stream<false, 0, false, false>", 128),
                                           ("k_lower4_stream<false, 0, true>", 96), ("k_lower4_stream<false, 0, false>", 96), ("k_lower4_stream<false, 2, true>", 96),
                                           ("k_lower4_walk<4, 1, false, true>", 72),
                                           ("k_lower_gen_walk<2, 5>", 128)])
def test_hot_kernels_keep_their_occupancy(kernels, name, max_vgpr):
    k = kernels[name]
    assert k["vgpr_count"] <= max_vgpr, k
    assert k["vgpr_spill_count"] == 0 and k["scratch_bytes"] == 0, k


def test_every_streamed_variant_is_in_the_library_and_spills_nothing(kernels):
    """the pre-order walk's prefetches are loads written out in assembly whose results must not be moved before the op's explicit
    wait: a spilled register set would be saved before the load has written it (phyamd_walk4s.inc, prefetch4)"""
    for fold in ("false", "true"):
        # plain (stored children as partials / carried through their branches), the reference's rescaling, powers of two per category
        for scale, tf in (("0", "false"), ("0", "true"), ("1", "false"), ("2", "true")):
            for ambig in ("false", "true"):
                k = kernels[f"k_upper4_stream<{fold}, {scale}, {ambig}, {tf}>"]
                assert k["vgpr_spill_count"] == 0 and k["scratch_bytes"] == 0, k
                assert k["vgpr_count"] <= (128 if scale == "0" and ambig == "false" else 168), k
