"""CPU: pin the substitution-model gradient (calculate_dlnl_dQ) before the GPU path is trusted with it.

Two pieces are checked against `gradient_all` of the compiled reference (tests/golden/*, flags TREE|SITE|SUBSTITUTION_MODEL,
i.e. constrained-value derivatives: 5 GTR rates / kappa, then the 4 frequencies as free coordinates):
  * the host library's dQ/dtheta (physher_amd/csrc/host/models.cpp, product code, no GPU needed), and
  * the oracle's numpy restatement of the branch sum and of the root-frequency term (oracle/phyoracle.py).
"""
import numpy as np
import pytest

from golden_util import UNROOTED_CASES, load, oracle_problem, read_spec
from oracle import phyoracle as po
from physher_amd import _phycpp_amd as pc


def subst_cases():
    out = []
    for c in UNROOTED_CASES:
        spec = read_spec(c)
        if spec["datatype"] == "nucleotide" and spec["model"] in ("gtr", "hky"):
            out.append(c)
    return out


def model_from_spec(spec, freqs):
    if spec["model"] == "hky":
        return pc.HKYInterface(float(spec["rates"]), list(freqs))
    return pc.GTRInterface([float(x) for x in spec["rates"].split(",")], list(freqs))


def reference_tail(gold):
    """[rates..., frequencies...] of gradient_all; the tree block is node_count long, then the site-model block."""
    assert gold["gradient_all_flags"] & 4
    n_rates = {"gtr": 5, "hky": 1}[gold["_model"]]
    return gold["gradient_all"][-(n_rates + 4):], n_rates


@pytest.mark.parametrize("case", subst_cases())
def test_parameter_gradient_matches_reference(case):
    gold = load(case)
    spec = read_spec(case)
    gold["_model"] = spec["model"]
    ref, n_rates = reference_tail(gold)
    m = model_from_spec(spec, gold["frequencies"])
    dQ = m.rate_matrix_derivatives()
    assert dQ.shape == (n_rates + 4, 4, 4)
    np.testing.assert_allclose(dQ.sum(axis=2), 0.0, atol=1e-13)  # rows of Q sum to 0 for every parameter value
    pb = oracle_problem(case, gold)
    lnl, g = po.parameter_gradient(pb, dQ)
    assert abs(lnl - gold["lnl"]) <= 1e-9 * abs(gold["lnl"])
    g[n_rates:] += po.root_frequency_term(pb)
    # the reference accumulates per-pattern sums in a different order and its HKY P(t) is closed-form
    np.testing.assert_allclose(g, ref, rtol=2e-8, atol=1e-7)


def test_rate_matrix_derivatives_are_derivatives():
    """dQ/dtheta against central differences of the host Q, all model families including a 6-rate simplex GTR."""
    f = np.array([0.3, 0.2, 0.15, 0.35])
    r5 = np.array([1.2, 3.1, 0.7, 0.9, 2.8])
    r6 = np.array([0.1, 0.3, 0.05, 0.15, 0.25, 0.15])
    dt = pc.GeneralDataTypeInterface(["A", "C", "G", "T"])
    makers = {
        "hky": lambda r, fr: pc.HKYInterface(float(r[0]), list(fr)),
        "gtr5": lambda r, fr: pc.GTRInterface(list(r), list(fr)),
        "gtr6": lambda r, fr: pc.GTRInterface(list(r), list(fr)),
        "general": lambda r, fr: pc.GeneralSubstitutionModelInterface(dt, list(r), list(fr), [0, 1, 0, 2, 1, 0], True),
        "general_raw": lambda r, fr: pc.GeneralSubstitutionModelInterface(dt, list(r), list(fr), [0, 1, 0, 2, 1, 0], False),
    }
    rates = {"hky": np.array([2.5]), "gtr5": r5, "gtr6": r6, "general": np.array([1.0, 2.0, 0.5]), "general_raw": np.array([1.0, 2.0, 0.5])}
    h = 1e-6
    for name, make in makers.items():
        r = rates[name]
        dQ = make(r, f).rate_matrix_derivatives()
        assert dQ.shape[0] == len(r) + 4
        Qof = lambda rr, ff: make(rr, ff).eigen_system()[3]
        for i in range(len(r)):
            e = np.zeros(len(r))
            e[i] = h
            fd = (Qof(r + e, f) - Qof(r - e, f)) / (2 * h)
            np.testing.assert_allclose(dQ[i], fd, atol=2e-9, err_msg=f"{name} rate {i}")
        for i in range(4):  # frequencies move as free coordinates (the reference's grad_wrt_reparam = false convention)
            e = np.zeros(4)
            e[i] = h
            fd = (Qof(r, f + e) - Qof(r, f - e)) / (2 * h)
            np.testing.assert_allclose(dQ[len(r) + i], fd, atol=2e-9, err_msg=f"{name} freq {i}")
    assert pc.JC69Interface().rate_matrix_derivatives().shape[0] == 0


def test_oracle_parameter_gradient_is_a_derivative():
    """The oracle restatement against central differences of the oracle's own lnL (GTR rates and frequencies)."""
    case = "gtr_g4_t16"
    gold = load(case)
    spec = read_spec(case)
    r0 = np.array([float(x) for x in spec["rates"].split(",")])
    f0 = gold["frequencies"].copy()

    def lnl_of(r, f):
        m = pc.GTRInterface(list(r), list(f))
        ev, U, Ui, _ = m.eigen_system()
        g = dict(gold)
        g.update(eval=ev, evec=U, ivec=Ui, frequencies=f)
        return oracle_problem(case, g).log_likelihood()["lnl"]

    m = pc.GTRInterface(list(r0), list(f0))
    ev, U, Ui, _ = m.eigen_system()
    g = dict(gold)
    g.update(eval=ev, evec=U, ivec=Ui)
    pb = oracle_problem(case, g)
    _, grad = po.parameter_gradient(pb, m.rate_matrix_derivatives())
    grad[5:] += po.root_frequency_term(pb)
    h = 1e-5
    for i in range(5):
        e = np.zeros(5)
        e[i] = h
        fd = (lnl_of(r0 + e, f0) - lnl_of(r0 - e, f0)) / (2 * h)
        assert abs(grad[i] - fd) < 1e-5 * max(1.0, abs(fd)), (i, grad[i], fd)
    for i in range(4):
        e = np.zeros(4)
        e[i] = h
        fd = (lnl_of(r0, f0 + e) - lnl_of(r0, f0 - e)) / (2 * h)
        assert abs(grad[5 + i] - fd) < 1e-5 * max(1.0, abs(fd)), (i, grad[5 + i], fd)
