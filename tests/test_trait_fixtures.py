"""CPU: discrete-trait fixtures from the compiled reference (general data type with K states, one attribute per taxon --
the model src/phycpp/physher.cpp:594-629 assembles) against the host library's general substitution model and the oracle.

The reference evaluates these with its scalar generic-state kernels and the eigen system of its non-symmetric solver;
here Q, its eigen system and dQ/dtheta come from physher_amd/csrc/host/models.cpp (product code, no GPU needed) and the
pruning / gradient from oracle/phyoracle.py.  The GPU side of the same fixtures is tests/test_phycpp_gpu.py.
"""
import numpy as np
import pytest

from golden_util import TRAIT_CASES, load, read_trait_case
from oracle import phyoracle as po
from physher_amd import _phycpp_amd as pc


def trait_partials(tc, gold):
    """[T][1][K] tip partials from the reference's pattern codes (datatype.c:184-240): state, named set, or unknown."""
    K = len(tc["states"])
    sets = list(tc["ambiguities"].values())
    T = gold["tip_count"]
    tp = np.zeros((T, 1, K))
    for tip in range(T):
        code = int(gold["patterns"][gold["mapping"][tip]][0])
        if code < K:
            tp[tip, 0, code] = 1.0
        elif code < K + len(sets):
            for s in sets[code - K]:
                tp[tip, 0, tc["states"].index(s)] = 1.0
        else:
            tp[tip, 0, :] = 1.0
    return tp


def host_model(tc):
    dt = pc.GeneralDataTypeInterface(tc["states"], tc["ambiguities"] or None)
    return dt, pc.GeneralSubstitutionModelInterface(dt, tc["rates"], tc["freqs"], tc["structure"], bool(tc.get("normalize", 1)))


def test_trait_cases_present():
    assert {"trait_k2_t9", "trait_k5_g3_t14", "trait_k7_sets_t12"} <= set(TRAIT_CASES)


@pytest.mark.parametrize("case", TRAIT_CASES)
def test_pattern_codes(case):
    """new_AttributePattern (sitepattern.c:321-353): one pattern of weight 1; codes as _encoding_string assigns them."""
    tc, gold = read_trait_case(case), load(case)
    K = len(tc["states"])
    assert gold["pattern_count"] == 1 and gold["weights"].tolist() == [1.0] and gold["taxa"] == tc["taxa"]
    names = list(tc["ambiguities"])
    for i, v in enumerate(tc["values"]):
        want = tc["states"].index(v) if v in tc["states"] else (K + names.index(v) if v in names else K + len(names))
        assert int(gold["patterns"][i][0]) == want, (i, v)


@pytest.mark.parametrize("case", TRAIT_CASES)
def test_general_model_and_oracle_match_reference(case):
    tc, gold = read_trait_case(case), load(case)
    K = len(tc["states"])
    dt, subst = host_model(tc)
    ev, U, Ui, Q = subst.eigen_system()
    # same rate matrix as the reference's (its eigenvectors are scaled differently; U diag(ev) U^-1 is what must agree)
    ref_Q = (gold["evec"] * gold["eval"][None, :]) @ gold["ivec"]
    np.testing.assert_allclose(Q, ref_Q, atol=1e-12)
    np.testing.assert_allclose((U * ev[None, :]) @ Ui, Q, atol=1e-12)
    pb = po.Problem(gold["left"], gold["right"], gold["root"], gold["weights"], ev, U, Ui, gold["frequencies"],
                    gold["cat_rates"], gold["cat_proportions"], gold["distance"],
                    tip_states=np.zeros((gold["tip_count"], 1), dtype=np.uint8), tip_partials=trait_partials(tc, gold))
    res = pb.gradient()
    assert abs(res["lnl"] - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    N = gold["node_count"]
    bg = po.branch_gradient_from_cat(res["cat_grad"], gold["cat_rates"], gold["cat_proportions"])
    ref = gold["gradient_all"]
    skip = {gold["root"], int(gold["right"][gold["root"]])}  # the reference leaves these two entries alone
    for i in range(N):
        if i not in skip:
            assert abs(bg[i] - ref[i]) <= 1e-9 * max(1.0, np.abs(ref[:N]).max()), (i, bg[i], ref[i])
    # substitution block: rates, then the K frequencies as free coordinates
    assert gold["gradient_all_flags"] & 4
    n_rates = len(tc["rates"])
    dQ = subst.rate_matrix_derivatives()
    assert dQ.shape == (n_rates + K, K, K)
    _, g = po.parameter_gradient(pb, dQ)
    g[n_rates:] += po.root_frequency_term(pb)
    np.testing.assert_allclose(g, ref[-(n_rates + K):], rtol=2e-8, atol=1e-7)


def test_structure_layouts_agree():
    """The general model's structure in its three layouts (packed upper triangle, upper + lower triangles, full matrix)."""
    tc = read_trait_case("trait_k5_g3_t14")
    K = 5
    tri = K * (K - 1) // 2
    dt = pc.GeneralDataTypeInterface(tc["states"])
    both = tc["structure"]
    assert len(both) == 2 * tri
    full = np.zeros((K, K), dtype=int)
    t = 0
    for i in range(K):
        for j in range(i + 1, K):
            full[i, j] = full[j, i] = both[t]
            t += 1
    Qs = [pc.GeneralSubstitutionModelInterface(dt, tc["rates"], tc["freqs"], s, True).eigen_system()[3]
          for s in (both[:tri], both, [int(x) for x in full.ravel()])]
    np.testing.assert_array_equal(Qs[0], Qs[1])
    np.testing.assert_array_equal(Qs[0], Qs[2])
    broken = list(both)
    broken[-1] = (broken[-1] + 1) % len(tc["rates"])  # lower triangle no longer mirrors the upper one
    with pytest.raises(pc.PhyamdError):
        pc.GeneralSubstitutionModelInterface(dt, tc["rates"], tc["freqs"], broken, True).eigen_system()
