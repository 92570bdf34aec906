"""CPU-only tests of the host-side model code (physher_amd/csrc/host) against the reference's golden vectors.

The host library is product code: Newick/ids, pattern compression, eigen systems, site-model rates, the
node-height ratio transform.  Nothing here touches the GPU (TreeLikelihoodInterface itself is tested in
tests/test_phycpp_gpu.py).
"""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, UNROOTED_CASES, load, read_fasta, read_spec
from physher_amd import _phycpp_amd as pc


def _model_from_spec(spec, gold):
    f = list(map(float, gold["frequencies"]))
    if spec["model"] == "jc69":
        return pc.JC69Interface()
    if spec["model"] == "hky":
        return pc.HKYInterface(float(spec["rates"]), f)
    if spec["model"] == "gtr":
        return pc.GTRInterface([float(x) for x in spec["rates"].split(",")], f)
    return None


@pytest.mark.parametrize("case", UNROOTED_CASES)
def test_pattern_compression_bit_exact(case):
    gold = load(case)
    names, seqs = read_fasta(os.path.join(GOLDEN, case, "aln.fa"))
    states, weights = pc.compress_patterns(read_spec(case)["datatype"], names, seqs)
    assert np.array_equal(states, gold["patterns"])
    assert np.array_equal(weights, gold["weights"])


def test_pattern_compression_fluA():
    gold = load("fluA_jc69_time")
    names, seqs = read_fasta(os.path.join(GOLDEN, "fluA_jc69_time", "fluA.fa"))
    states, weights = pc.compress_patterns("nucleotide", names, seqs)
    assert states.shape == (69, 238)
    assert np.array_equal(states, gold["patterns"]) and np.array_equal(weights, gold["weights"])


def test_pattern_compression_rejects_ragged_alignment():
    with pytest.raises(pc.PhyamdError):
        pc.compress_patterns("nucleotide", ["a", "b"], ["ACGT", "ACG"])


@pytest.mark.parametrize("case", UNROOTED_CASES)
def test_tree_ids_and_branch_lengths_match_reference(case):
    """Node ids (tips by taxon index, clades in post-order), children, root-branch folding (tree.c:183-224, 1438-1443)."""
    gold = load(case)
    with open(os.path.join(GOLDEN, case, "tree.nwk")) as f:
        newick = f.read().strip()
    tm = pc.UnRootedTreeModelInterface(newick, gold["taxa"])
    d = tm.describe()
    assert d["root"] == gold["root"]
    assert d["left"] == gold["left"].tolist() and d["right"] == gold["right"].tolist()
    np.testing.assert_array_equal(np.array(d["distance"]), gold["distance"])
    assert tm.parameter_count == gold["node_count"] - 2
    assert list(tm.node_map) == list(range(gold["node_count"]))
    # parameters round-trip through the reference's indexing
    p = tm.get_parameters()
    tm.set_parameters(p * 2.0)
    np.testing.assert_allclose(tm.get_parameters(), p * 2.0)


def test_newick_polytomy_and_quotes():
    # trifurcating root (the usual unrooted newick): the third child hangs under an inserted node on the right
    tm = pc.UnRootedTreeModelInterface("('A b':0.1,B:0.2,(C:0.3,D:0.4):0.5);", ["A b", "B", "C", "D"])
    d = tm.describe()
    assert tm.get_node_count() == 7 and d["root"] == 6
    # post-order: A(0) | B(1) (C,D)->4 inserted->5 | root 6
    assert d["left"][6] == 0 and d["right"][6] == 5
    assert d["left"][5] == 1 and d["right"][5] == 4
    assert d["left"][4] == 2 and d["right"][4] == 3
    # inserted node had BL_MIN, folded onto root->left; root->right = 0
    assert d["distance"][5] == 0.0 and abs(d["distance"][0] - (0.1 + 1e-8)) < 1e-15
    with pytest.raises(pc.PhyamdError):
        pc.UnRootedTreeModelInterface("((A:1,B:1):1,C:1);", ["A", "B", "X"])
    with pytest.raises(pc.PhyamdError):
        pc.UnRootedTreeModelInterface("((A:1,B:1):1,C:1;", ["A", "B", "C"])


@pytest.mark.parametrize("case", [c for c in UNROOTED_CASES if read_spec(c)["datatype"] == "nucleotide" and "pt" in load(c)])
def test_transition_matrices_from_model_parameters(case):
    """Q construction + eigen system (own Jacobi solver) give the reference's P(t) and dP/dt."""
    gold = load(case)
    spec = read_spec(case)
    m = _model_from_spec(spec, gold)
    for q, node in enumerate(gold["pt_nodes"] if "pt" in gold else []):  # (the slim rescaled fixtures carry no matrices)
        for c in range(gold["category_count"]):
            t = gold["distance"][node] * gold["cat_rates"][c]
            np.testing.assert_allclose(m.transition_matrix(t), gold["pt"][q, c], rtol=1e-11, atol=1e-14)
            np.testing.assert_allclose(m.transition_matrix(t, derivative=True), gold["dpt"][q, c], rtol=1e-10, atol=1e-13)
    ev, U, Ui, Q = m.eigen_system()
    np.testing.assert_allclose(U @ Ui, np.eye(4), atol=1e-13)
    np.testing.assert_allclose(U @ np.diag(ev) @ Ui, Q, atol=1e-13)
    np.testing.assert_allclose(Q.sum(axis=1), 0, atol=1e-14)
    assert abs(-(np.diag(Q) * gold["frequencies"]).sum() - 1.0) < 1e-14  # normalised


def test_gtr_simplex_and_general_model_agree():
    f = [0.3, 0.2, 0.2, 0.3]
    r5 = np.array([1.2, 3.1, 0.7, 0.9, 2.8])
    r6 = np.append(r5, 1.0)
    a = pc.GTRInterface(list(r5), f)
    b = pc.GTRInterface(list(r6 / r6.sum()), f)  # 6-rate simplex: same Q after normalisation
    dt = pc.GeneralDataTypeInterface(["A", "C", "G", "T"])
    g = pc.GeneralSubstitutionModelInterface(dt, list(r6), f, [0, 1, 2, 3, 4, 5], True)
    for t in (0.01, 0.3, 2.0):
        np.testing.assert_allclose(a.transition_matrix(t), b.transition_matrix(t), rtol=1e-12)
        np.testing.assert_allclose(a.transition_matrix(t), g.transition_matrix(t), rtol=1e-12)
    a.set_rates(r5 * 1.5)
    assert not np.allclose(a.transition_matrix(0.3), b.transition_matrix(0.3), rtol=1e-6)


@pytest.mark.parametrize("case", [c for c in UNROOTED_CASES if int(read_spec(c)["categories"]) > 1])
def test_gamma_rates_match_reference(case):
    gold = load(case)
    spec = read_spec(case)
    pinv = float(spec["pinv"]) if "pinv" in spec else None
    mu = float(spec["mu"]) if "mu" in spec else None
    cls = pc.WeibullSiteModelInterface if spec.get("sitedist", "gamma") == "weibull" else pc.GammaSiteModelInterface
    sm = cls(float(spec["alpha"]), int(spec["categories"]), pinv, mu)
    np.testing.assert_allclose(sm.rates(), gold["cat_rates_without_mu"], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(sm.proportions(), gold["cat_proportions"], rtol=1e-15, atol=0)


def test_site_model_variants():
    # quantile function is the inverse of the regularised incomplete gamma function
    for a in (0.05, 0.5, 1.0, 3.7, 50.0):
        for p in (1e-6, 0.125, 0.5, 0.875, 0.999999):
            x = pc.gamma_quantile(p, a, a)
            assert abs(pc.reg_lower_gamma(a, x * a) - p) < 1e-12 * max(1.0, p / 1e-3)
    g = pc.GammaSiteModelInterface(0.5, 4, proportion_invariant=0.2)
    r, w = g.rates(), g.proportions()
    assert g.get_category_count() == 5 and r[0] == 0 and abs(w[0] - 0.2) < 1e-15
    assert abs((r * w).sum() - 1.0) < 1e-14 and abs(w.sum() - 1.0) < 1e-15
    wb = pc.WeibullSiteModelInterface(1.3, 6)
    assert abs((wb.rates() * wb.proportions()).sum() - 1.0) < 1e-14
    inv = pc.InvariantSiteModelInterface(0.25)
    np.testing.assert_allclose(inv.rates(), [0.0, 1.0 / 0.75])
    c = pc.ConstantSiteModelInterface(mu=2.5)
    assert c.rates().tolist() == [2.5] and c.parameter_count == 1
    g.set_parameters(np.array([0.9, 0.1]))
    np.testing.assert_allclose(g.get_parameters(), [0.9, 0.1])


def _fluA():
    with open(os.path.join(GOLDEN, "fluA_jc69_time", "jc69-time.json")) as f:
        js = json.load(f)
    tree = js["model"]["tree"]
    gold = load("fluA_jc69_time")
    # the fixture came through the reference's JSON route, which numbers tips by post-order rank (tree.c:183-200);
    # handing the taxa in that order to the phycpp-style constructor (ids = index in the taxon list) gives the same ids
    taxa = [gold["node_names"][i] for i in range(gold["tip_count"])]
    dates = [float(tree["dates"][t]) for t in taxa]
    return gold, tree["newick"], taxa, dates


def test_time_tree_heights_and_ratio_transform():
    gold, newick, taxa, dates = _fluA()
    tm = pc.ReparameterizedTimeTreeModelInterface(newick, taxa, dates, pc.TreeTransformFlags.RATIO)
    d = tm.describe()
    assert d["left"] == gold["left"].tolist() and d["right"] == gold["right"].tolist()
    np.testing.assert_allclose(d["height"], gold["heights"], rtol=1e-12, atol=1e-12)
    # the reference's own constants: log|J| = lnL(with Jacobian) - lnL(without)  (tests/test_tree_likelihood.c:29,88)
    assert abs(tm.transform_jacobian() - ((-4786.867701371271) - (-4777.616349713985))) < 1e-8
    # ratios -> heights round trip
    r = tm.get_parameters()
    assert r.shape == (68,) and np.all((r[:-1] > 0) & (r[:-1] <= 1))
    tm.set_parameters(r)
    np.testing.assert_allclose(tm.describe()["height"], gold["heights"], rtol=1e-12, atol=1e-12)
    # JVP against central differences of  f(r) = sum_i c_i h_i(r)
    rng = np.random.default_rng(3)
    c = rng.normal(size=68)
    g = tm.gradient_transform_jvp(c)

    def f(rr):
        tm.set_parameters(rr)
        return float(np.dot(c, tm.get_node_heights()))

    for i in rng.choice(68, size=8, replace=False):
        h = 1e-6
        rp, rm = r.copy(), r.copy()
        rp[i] += h
        rm[i] -= h
        fd = (f(rp) - f(rm)) / (2 * h)
        assert abs(fd - g[i]) < 1e-6 * max(1.0, abs(g[i]))
    tm.set_parameters(r)
    # Jacobian gradient against central differences of log|J|
    gj = tm.gradient_transform_jacobian()
    for i in rng.choice(68, size=8, replace=False):
        h = 1e-6
        rp, rm = r.copy(), r.copy()
        rp[i] += h
        rm[i] -= h
        tm.set_parameters(rp)
        up = tm.transform_jacobian()
        tm.set_parameters(rm)
        dn = tm.transform_jacobian()
        assert abs((up - dn) / (2 * h) - gj[i]) < 1e-5 * max(1.0, abs(gj[i]))
    tm.set_parameters(r)


def test_clock_models():
    gold, newick, taxa, dates = _fluA()
    tm = pc.TimeTreeModelInterface(newick, taxa, dates)
    sc = pc.StrictClockModelInterface(0.001, tm)
    assert sc.parameter_count == 1 and sc.get_parameters().tolist() == [0.001]
    sc.set_rate(0.002)
    assert sc.get_parameters().tolist() == [0.002]
    rates = list(np.linspace(1e-3, 2e-3, tm.get_node_count() - 1))
    pc.SimpleClockModelInterface(rates, tm)
    with pytest.raises(pc.PhyamdError):
        pc.SimpleClockModelInterface(rates[:-1], tm)


# ---------------------------------------------------------------------------------------------------------
# M3: the fixed-exchangeability models of the 20- and 61-state configurations (wag.c:23-36, lg.c:23-36, mg94.c:62-138)
# ---------------------------------------------------------------------------------------------------------
def _spec_model(case):
    spec = read_spec(case)
    name = {"wag": "WAG", "lg": "LG", "mg94": "MG94"}[spec["model"]]
    rates = [float(x) for x in spec["rates"].split(",")] if "rates" in spec else []
    freqs = [float(x) for x in spec["freqs"].split(",")] if "freqs" in spec else None
    return name, rates, freqs


@pytest.mark.parametrize("case", [c for c in UNROOTED_CASES if read_spec(c)["model"] in ("wag", "lg", "mg94")])
def test_empirical_and_codon_rate_matrices_match_reference(case):
    """WAG / LG exchangeability tables and the MG94 codon rule (universal code; kappa, alpha, beta) against the normalised Q,
    the frequencies and P(t) of three branches that the compiled reference produced for the same model."""
    gold = load(case)
    name, rates, freqs = _spec_model(case)
    if name == "MG94" and freqs is None:
        freqs = list(gold["frequencies"])  # new_MG94_with_values starts from a uniform 61-simplex
    m = pc.substitution_model(name, rates, freqs)
    S = gold["state_count"]
    assert m["state_count"] == S
    np.testing.assert_allclose(m["frequencies"], gold["frequencies"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(m["Q"], gold["Q"], rtol=1e-12, atol=1e-15)
    # the eigen system is a different basis (symmetrised Jacobi here, orthes + hqr2 there): compare what it is used for
    Q = m["evec"] @ np.diag(m["eval"]) @ m["ivec"]
    np.testing.assert_allclose(Q, gold["Q"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(m["evec"] @ m["ivec"], np.eye(S), atol=1e-12)
    for q, node in enumerate(gold["pt_nodes"] if "pt" in gold else []):  # (the slim rescaled fixtures carry no matrices)
        for c in range(gold["category_count"]):
            t = gold["distance"][node] * gold["cat_rates"][c]
            P = np.abs(m["evec"] @ np.diag(np.exp(m["eval"] * t)) @ m["ivec"])  # substmodel.c:518-557
            np.testing.assert_allclose(P, gold["pt"][q][c], rtol=1e-9, atol=1e-14)


def test_wag_default_frequencies_and_mg94_structure():
    m = pc.substitution_model("WAG")
    assert abs(m["frequencies"].sum() - 1.0) < 1e-12 and m["Q"].shape == (20, 20)
    np.testing.assert_allclose(m["Q"].sum(axis=1), 0.0, atol=1e-14)
    assert abs(-(m["frequencies"] * np.diag(m["Q"])).sum() - 1.0) < 1e-13  # one expected substitution per unit time
    mg = pc.substitution_model("MG94", [2.0, 1.0, 0.5])
    Q = mg["Q"]
    assert Q.shape == (61, 61)
    # every sense codon has 9 single-nucleotide neighbours minus those that are stops: between 7 and 9 non-zero rates per row
    nz = (Q != 0).sum(axis=1) - 1
    assert nz.min() >= 7 and nz.max() == 9 and nz.sum() == 2 * 263  # 263 sense-codon pairs one substitution apart (universal code)
    with pytest.raises(Exception):
        pc.substitution_model("MG94", [2.0, 1.0])
