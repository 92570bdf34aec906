"""GPU end-to-end tests through the phycpp-compatible classes (what torchtree-physher would drive):
alignment + newick + model parameters in, lnL and gradient out, against the reference's golden vectors
and its own known-answer constants (tests/test_tree_likelihood.c)."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN, TRAIT_CASES, UNROOTED_CASES, load, read_fasta, read_spec, read_trait_case

pytestmark = pytest.mark.gpu

SITE_CASES = ["gtr_g4i_mu_t14", "gtr_w3_t12", "hky_w4i_t10", "jc69_inv_t10"]
CASES4 = [c for c in UNROOTED_CASES if read_spec(c)["datatype"] == "nucleotide" and read_spec(c)["rescale"] == "0"
          and not load(c)["rescaled"]]


def _build(case, pc):
    gold = load(case)
    spec = read_spec(case)
    names, seqs = read_fasta(os.path.join(GOLDEN, case, "aln.fa"))
    with open(os.path.join(GOLDEN, case, "tree.nwk")) as f:
        newick = f.read().strip()
    tree = pc.UnRootedTreeModelInterface(newick, names)
    f = list(map(float, gold["frequencies"]))
    if spec["model"] == "jc69":
        subst = pc.JC69Interface()
    elif spec["model"] == "hky":
        subst = pc.HKYInterface(float(spec["rates"]), f)
    else:
        subst = pc.GTRInterface([float(x) for x in spec["rates"].split(",")], f)
    C = int(spec["categories"])
    pinv = float(spec["pinv"]) if "pinv" in spec else None
    mu = float(spec["mu"]) if "mu" in spec else None
    dist = spec.get("sitedist", "gamma")
    if dist == "discrete":
        site = pc.InvariantSiteModelInterface(pinv, mu)
    elif C > 1 and dist == "weibull":
        site = pc.WeibullSiteModelInterface(float(spec["alpha"]), C, pinv, mu)
    elif C > 1:
        site = pc.GammaSiteModelInterface(float(spec["alpha"]), C, pinv, mu)
    else:
        site = pc.ConstantSiteModelInterface(mu)
    tlk = pc.TreeLikelihoodInterface(list(zip(names, seqs)), tree, subst, site, None, use_tip_states=spec["tipstates"] == "1")
    return gold, tree, subst, site, tlk


@pytest.mark.parametrize("case", CASES4)
def test_unrooted_likelihood_and_gradient(case):
    from physher_amd import _phycpp_amd as pc
    gold, tree, subst, site, tlk = _build(case, pc)
    N = gold["node_count"]
    assert tlk.get_pattern_count() == gold["pattern_count"]
    assert np.array_equal(tlk.pattern_states(), gold["patterns"]) and np.array_equal(tlk.pattern_weights(), gold["weights"])
    lnl = tlk.log_likelihood()
    assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    tlk.request_gradient([pc.TreeLikelihoodGradientFlags.TREE_HEIGHT])
    assert tlk.gradient_length == N - 2  # physher.cpp:639-641
    # reference arithmetic (include_root_freqs = true): what physher returns for TREE_MODEL-only requests
    tlk.set_reference_compatibility(True)
    g = tlk.gradient()
    ref = gold["gradient_tree"][: N - 2]
    assert np.abs(g - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # default: exact derivative (= the reference's own numbers when it also differentiates the substitution model)
    tlk.set_reference_compatibility(False)
    g = tlk.gradient()
    rr, rl = gold["right"][gold["root"]], gold["left"][gold["root"]]
    mu = float(read_spec(case).get("mu", 1.0))
    if gold["gradient_all_flags"] & 4:
        ref = gold["gradient_all"][: N - 2].copy() * mu  # the reference's branch gradient leaves mu out (treelikelihood.c:3134)
        if rr < gold["tip_count"]:
            # bifurcating-root newick with a tip on the right: the root branch is parameter rr; the reference reports 0
            # for it (treelikelihood.c:3249-3255), the default mode reports its derivative = that of root->left
            ref[rr] = gold["gradient_all"][rl]
        assert np.abs(g - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # ... and it IS the derivative: central differences through SetParameters
    p = tree.get_parameters()
    rng = np.random.default_rng(0)
    for i in rng.choice(N - 2, size=3, replace=False):
        h = 1e-6
        pp, pm = p.copy(), p.copy()
        pp[i] += h
        pm[i] -= h
        tree.set_parameters(pp)
        up = tlk.log_likelihood()
        tree.set_parameters(pm)
        dn = tlk.log_likelihood()
        assert abs((up - dn) / (2 * h) - g[i]) <= 2e-5 * max(1.0, abs(g[i]))
    tree.set_parameters(p)
    assert abs(tlk.log_likelihood() - lnl) <= 1e-12 * abs(lnl)


def test_parameter_updates_propagate():
    from physher_amd import _phycpp_amd as pc
    gold, tree, subst, site, tlk = _build("gtr_g4_t16", pc)
    l0 = tlk.log_likelihood()
    site.set_shape(1.7)
    l1 = tlk.log_likelihood()
    subst.set_rates(np.array([1.0, 2.0, 1.0, 1.0, 2.0]))
    l2 = tlk.log_likelihood()
    subst.set_frequencies(np.array([0.25, 0.25, 0.25, 0.25]))
    l3 = tlk.log_likelihood()
    assert len({round(x, 6) for x in (l0, l1, l2, l3)}) == 4
    site.set_shape(0.5)
    subst.set_rates(np.array([1.2, 3.1, 0.7, 0.9, 2.8]))
    subst.set_frequencies(np.array(gold["frequencies"]))
    assert abs(tlk.log_likelihood() - l0) <= 1e-11 * abs(l0)


def test_substitution_gradient_refused_without_parameters():
    from physher_amd import _phycpp_amd as pc
    gold, tree, subst, site, tlk = _build("jc69_t12", pc)
    with pytest.raises(pc.PhyamdError):  # JC69 has no dPdp in the reference either
        tlk.request_gradient([pc.TreeLikelihoodGradientFlags.SUBSTITUTION_MODEL])


SUBST_CASES = [c for c in CASES4 if read_spec(c)["model"] in ("gtr", "hky")]


@pytest.mark.parametrize("case", SUBST_CASES)
def test_substitution_model_gradient(case):
    """SUBSTITUTION_MODEL / _RATES / _FREQUENCIES blocks of the reference's gradient (gradient_PMatrix,
    treelikelihood.c:3077-3110; calculate_dlnl_dQ :2337-2583), computed here in the same two passes as the branch gradient."""
    from physher_amd import _phycpp_amd as pc
    F = pc.TreeLikelihoodGradientFlags
    gold, tree, subst, site, tlk = _build(case, pc)
    N = gold["node_count"]
    n_rates = {"gtr": 5, "hky": 1}[read_spec(case)["model"]]
    n_site = gold["site_rate_parameters"] + int(gold["site_has_pinv"]) + int(gold["site_has_mu"])
    ref = gold["gradient_all"]
    assert gold["gradient_all_flags"] & 4 and len(ref) == N + n_site + n_rates + 4
    tol = dict(rtol=2e-8, atol=1e-7)
    tlk.request_gradient([F.TREE_HEIGHT, F.SITE_MODEL, F.SUBSTITUTION_MODEL])
    assert tlk.gradient_length == N - 2 + n_site + n_rates + 4
    g = tlk.gradient()
    np.testing.assert_allclose(g[N - 2 + n_site:], ref[N + n_site:], **tol)
    np.testing.assert_allclose(g[N - 2: N - 2 + n_site], ref[N: N + n_site], rtol=2e-7, atol=1e-7)
    tlk.request_gradient([F.SUBSTITUTION_MODEL_RATES])
    assert tlk.gradient_length == n_rates
    np.testing.assert_allclose(tlk.gradient(), ref[N + n_site: N + n_site + n_rates], **tol)
    tlk.request_gradient([F.SUBSTITUTION_MODEL_FREQUENCIES])
    assert tlk.gradient_length == 4
    np.testing.assert_allclose(tlk.gradient(), ref[N + n_site + n_rates:], **tol)
    # default request = everything differentiable (TreeLikelihood_initialize_gradient(flags = 0), treelikelihood.c:255-270)
    tlk.request_gradient()
    assert tlk.gradient_length == N - 2 + n_site + n_rates + 4
    # reference-compatibility mode keeps the substitution block (the reference clears include_root_freqs itself)
    tlk.request_gradient([F.TREE_HEIGHT, F.SUBSTITUTION_MODEL])
    tlk.set_reference_compatibility(True)
    gc = tlk.gradient()
    np.testing.assert_allclose(gc[N - 2:], ref[N + n_site:], **tol)


def test_substitution_gradient_is_a_derivative():
    """Central differences of the GPU lnL through the wrapper's setters (rates, kappa; frequencies as free coordinates)."""
    from physher_amd import _phycpp_amd as pc
    F = pc.TreeLikelihoodGradientFlags
    for case, n_rates in (("gtr_g4_t16", 5), ("hky_g3_t10", 1)):
        gold, tree, subst, site, tlk = _build(case, pc)
        tlk.request_gradient([F.SUBSTITUTION_MODEL])
        g = tlk.gradient()
        spec = read_spec(case)
        r0 = np.array([float(x) for x in spec["rates"].split(",")])
        set_rates = (lambda r: subst.set_kappa(float(r[0]))) if n_rates == 1 else subst.set_rates
        h = 1e-5
        for i in range(n_rates):
            e = np.zeros(n_rates)
            e[i] = h
            set_rates(r0 + e)
            up = tlk.log_likelihood()
            set_rates(r0 - e)
            dn = tlk.log_likelihood()
            fd = (up - dn) / (2 * h)
            assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(fd)), (case, i, fd, g[i])
        set_rates(r0)
        # the frequency entries are partial derivatives with the four values free; along a direction that keeps the sum
        # at 1 they combine into the derivative the simplex setter can realise
        f0 = np.array(gold["frequencies"])
        d = np.array([1.0, -1.0, 0.5, -0.5])
        subst.set_frequencies(f0 + h * d)
        up = tlk.log_likelihood()
        subst.set_frequencies(f0 - h * d)
        dn = tlk.log_likelihood()
        subst.set_frequencies(f0)
        fd = (up - dn) / (2 * h)
        assert abs(fd - g[n_rates:] @ d) <= 2e-5 * max(1.0, abs(fd)), (case, fd, g[n_rates:] @ d)


def _fluA(pc, include_jacobian):
    d = os.path.join(GOLDEN, "fluA_jc69_time")
    with open(os.path.join(d, "jc69-time.json")) as f:
        js = json.load(f)
    tree_js = js["model"]["tree"]
    names, seqs = read_fasta(os.path.join(d, "fluA.fa"))
    dates = [float(tree_js["dates"][t]) for t in names]
    tree = pc.ReparameterizedTimeTreeModelInterface(tree_js["newick"], names, dates, pc.TreeTransformFlags.RATIO)
    clock = pc.StrictClockModelInterface(0.001, tree)
    subst = pc.JC69Interface()
    site = pc.ConstantSiteModelInterface()
    tlk = pc.TreeLikelihoodInterface(list(zip(names, seqs)), tree, subst, site, clock, use_tip_states=True, include_jacobian=include_jacobian)
    return tree, clock, tlk


def test_fluA_reference_known_answers():
    """The reference's only known-answer test (tests/test_tree_likelihood.c:29-131), constants copied from it."""
    from physher_amd import _phycpp_amd as pc
    gold = load("fluA_jc69_time")
    tree, clock, tlk = _fluA(pc, include_jacobian=False)
    assert abs(tlk.log_likelihood() - (-4777.616349713985)) < 1e-8
    tlk.request_gradient([pc.TreeLikelihoodGradientFlags.TREE_HEIGHT, pc.TreeLikelihoodGradientFlags.BRANCH_MODEL])
    assert tlk.gradient_length == 69
    g = tlk.gradient()
    assert abs(g[68] - 328017.6732813406) < 1e-9 * 328017.67  # clock rate (:38)
    assert abs(g[67] - 17.492484957839924) < 1e-8            # root height (:77)
    for i, v in {0: -0.5936536642214764, 1: 6.441289658869611, 13: 96.69564894572747, 43: 152.27137882559083, 66: 6.802521820384058}.items():
        assert abs(g[i] - v) < 1e-8, (i, g[i], v)                # ratios (:53-75)
    ref = np.array(gold["gradient_tree_clock_jacobian0"])
    assert np.abs(g - ref).max() <= 1e-10 * np.abs(ref).max()

    tree, clock, tlk = _fluA(pc, include_jacobian=True)
    assert abs(tlk.log_likelihood() - (-4786.867701371271)) < 1e-8  # (:88)
    tlk.request_gradient([pc.TreeLikelihoodGradientFlags.TREE_HEIGHT, pc.TreeLikelihoodGradientFlags.BRANCH_MODEL])
    g = tlk.gradient()
    assert abs(g[67] - 19.936860572419484) < 1e-8                   # (:117)
    assert abs(g[2] - 11.202945298115116) < 1e-8 and abs(g[43] - 188.80951477041825) < 1e-8  # (:93, :107)
    ref = np.array(gold["gradient_tree_clock_jacobian1"])
    assert np.abs(g - ref).max() <= 1e-10 * np.abs(ref).max()
    # cache / update behaviour exercised by the reference test (:37-50): changing the clock and restoring it
    clock.set_rate(0.002)
    assert abs(tlk.log_likelihood() - (-4786.867701371271)) > 1.0
    clock.set_rate(0.001)
    assert abs(tlk.log_likelihood() - (-4786.867701371271)) < 1e-8


@pytest.mark.parametrize("case", SITE_CASES)
def test_site_model_gradient(case):
    """shape (Gamma: the reference's own central difference on the quantiles; Weibull: closed form), proportion of
    invariant sites, mu -- the SITE_MODEL block of the reference's gradient (treelikelihood.c:3010-3052, 3277-3303)."""
    from physher_amd import _phycpp_amd as pc
    gold, tree, subst, site, tlk = _build(case, pc)
    N = gold["node_count"]
    np.testing.assert_allclose(site.rates(), gold["cat_rates_without_mu"], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(site.proportions(), gold["cat_proportions"], rtol=1e-14)
    lnl = tlk.log_likelihood()
    assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    n_site = gold["site_rate_parameters"] + int(gold["site_has_pinv"]) + int(gold["site_has_mu"])
    tlk.request_gradient([pc.TreeLikelihoodGradientFlags.TREE_HEIGHT, pc.TreeLikelihoodGradientFlags.SITE_MODEL])
    assert tlk.gradient_length == N - 2 + n_site
    g = tlk.gradient()
    mu = float(read_spec(case).get("mu", 1.0))
    ref_tree = gold["gradient_all"][: N - 2] * mu if gold["gradient_all_flags"] & 4 else None
    if ref_tree is not None:
        assert np.abs(g[: N - 2] - ref_tree).max() <= 1e-9 * max(1.0, np.abs(ref_tree).max())
    ref_site = gold["gradient_all"][N: N + n_site]
    if not (gold["gradient_all_flags"] & 4):
        tlk.set_reference_compatibility(True)  # JC69: the reference ran with folded root frequencies (exact for uniform pi)
        g = tlk.gradient()
    np.testing.assert_allclose(g[N - 2:], ref_site, rtol=2e-7, atol=1e-7)
    # and each is a derivative of lnL: central differences through the wrapper's SetParameters
    p0 = site.get_parameters()
    for i in range(n_site):
        h = 1e-6 * max(1.0, abs(p0[i]))
        pp, pm = p0.copy(), p0.copy()
        pp[i] += h
        pm[i] -= h
        site.set_parameters(pp)
        up = tlk.log_likelihood()
        site.set_parameters(pm)
        dn = tlk.log_likelihood()
        fd = (up - dn) / (2 * h)
        assert abs(fd - g[N - 2 + i]) <= 1e-4 * max(1.0, abs(fd)), (i, fd, g[N - 2 + i])
    site.set_parameters(p0)


@pytest.mark.parametrize("K,tip_states", [(2, False), (3, True), (4, False), (7, False), (7, True), (20, False), (33, False), (61, True)])
def test_attribute_patterns_any_state_count(K, tip_states):
    """The second constructor (taxa + one attribute each, new_AttributePattern): discrete-trait likelihoods with a general
    data type of K states.  The engine has kernels for 4 / 20 / 60 / 61 states; other K are padded with states nothing can
    enter or leave.  Checked against the CPU oracle run on the unpadded K-state problem."""
    from oracle import phyoracle as po
    from physher_amd import _phycpp_amd as pc, synth
    rng = np.random.default_rng(40 + K)
    T = 11
    tree_s = synth.random_tree(T, rng, bl_low=0.05, bl_high=0.6)
    names = list(tree_s.names)
    states = [f"loc{i}" for i in range(K)]
    tip_codes = rng.integers(0, K, size=T)
    attrs = [states[c] for c in tip_codes]
    attrs[3] = "?"  # unknown: all states
    dt = pc.GeneralDataTypeInterface(states)
    n_pairs = K * (K - 1) // 2
    structure = [int(i % 3) for i in range(n_pairs)]
    rates = [0.7, 1.9, 1.1]
    freqs = rng.dirichlet(np.full(K, 4.0))
    subst = pc.GeneralSubstitutionModelInterface(dt, rates, list(freqs), structure, True)
    site = pc.GammaSiteModelInterface(0.8, 3, None, None)
    tree = pc.UnRootedTreeModelInterface(tree_s.newick(), names)
    tlk = pc.TreeLikelihoodInterface(names, attrs, tree, subst, site, None, use_tip_states=tip_states)
    assert tlk.get_pattern_count() == 1
    lnl = tlk.log_likelihood()
    tlk.request_gradient([pc.TreeLikelihoodGradientFlags.TREE_HEIGHT])
    g = tlk.gradient()
    # oracle on the K-state problem, node ids / branch lengths as the wrapper's tree has them
    ev, U, Ui, _ = subst.eigen_system()
    d = tree.describe()
    left, right, root, dist = np.array(d["left"]), np.array(d["right"]), d["root"], np.array(d["distance"])
    dist[root] = 0.0
    codes = np.array([K if a == "?" else states.index(a) for a in attrs], dtype=np.uint8)
    order = [names.index(n) for n in d["name"][:T]]
    pb = po.Problem(left, right, root, np.ones(1), ev, U, Ui, freqs, site.rates(), site.proportions(), dist,
                    tip_states=codes[order][:, None])
    ref = pb.gradient()
    assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
    bg = po.branch_gradient_from_cat(ref["cat_grad"], site.rates(), site.proportions())
    N = 2 * T - 1
    rr = right[root]
    node_map = tree.node_map
    for i in range(N - 2):  # physher.cpp:649-655: gradient[nodeMap_[i]] = g[i]
        if i != rr:
            assert abs(g[node_map[i]] - bg[i]) <= 1e-9 * max(1.0, np.abs(bg).max()), (i, g[node_map[i]], bg[i])
    p0 = tree.get_parameters()
    for i in rng.choice(N - 2, size=3, replace=False):  # and it is the derivative of the wrapper's own lnL
        h = 1e-6
        pp, pm = p0.copy(), p0.copy()
        pp[i] += h
        pm[i] -= h
        tree.set_parameters(pp)
        up = tlk.log_likelihood()
        tree.set_parameters(pm)
        dn = tlk.log_likelihood()
        assert abs((up - dn) / (2 * h) - g[i]) <= 2e-5 * max(1.0, abs(g[i]))
    tree.set_parameters(p0)


def test_attribute_patterns_with_ambiguity_sets():
    """Named ambiguity sets of the general data type (GenericDataType_add_ambiguity, datatype.c:212-262) are exact on the
    4-state engine: the set becomes the tip's mask."""
    from oracle import phyoracle as po
    from physher_amd import _phycpp_amd as pc, synth
    rng = np.random.default_rng(5)
    T, K = 9, 3
    tree_s = synth.random_tree(T, rng, bl_low=0.05, bl_high=0.6)
    names = list(tree_s.names)
    states = ["a", "b", "c"]
    attrs = [states[c] for c in rng.integers(0, K, size=T)]
    attrs[1], attrs[4], attrs[6] = "ab", "bc", "-"
    dt = pc.GeneralDataTypeInterface(states, {"ab": ["a", "b"], "bc": ["b", "c"]})
    subst = pc.GeneralSubstitutionModelInterface(dt, [0.5, 1.5, 1.0], [0.2, 0.5, 0.3], [0, 1, 2], True)
    site = pc.ConstantSiteModelInterface(None)
    tree = pc.UnRootedTreeModelInterface(tree_s.newick(), names)
    tlk = pc.TreeLikelihoodInterface(names, attrs, tree, subst, site, None)
    ev, U, Ui, _ = subst.eigen_system()
    d = tree.describe()
    dist = np.array(d["distance"])
    dist[d["root"]] = 0.0
    sets = {"a": [1, 0, 0], "b": [0, 1, 0], "c": [0, 0, 1], "ab": [1, 1, 0], "bc": [0, 1, 1], "-": [1, 1, 1]}
    tp = np.array([[sets[attrs[names.index(n)]]] for n in d["name"][:T]], dtype=np.float64)  # [T][1][3]
    pb = po.Problem(d["left"], d["right"], d["root"], np.ones(1), ev, U, Ui, [0.2, 0.5, 0.3], [1.0], [1.0], dist,
                    tip_states=np.zeros((T, 1), dtype=np.uint8), tip_partials=tp)
    ref = pb.log_likelihood()
    assert abs(tlk.log_likelihood() - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])


@pytest.mark.parametrize("tip_states", [False, True])
@pytest.mark.parametrize("case", TRAIT_CASES)
def test_discrete_traits_match_reference(case, tip_states):
    """Discrete-trait fixtures generated from the compiled reference (tests/golden/make_golden.py::run_attr_case): K = 2, 5
    and 7 states (padded here to the 4- and 20-state kernels), named ambiguity sets and unknowns, with and without gamma
    categories -- lnL and every gradient block (branches, gamma shape, rates, frequencies) in the reference's order."""
    from physher_amd import _phycpp_amd as pc
    F = pc.TreeLikelihoodGradientFlags
    tc, gold = read_trait_case(case), load(case)
    K, n_rates = len(tc["states"]), len(tc["rates"])
    has_sets = bool(tc["ambiguities"])
    if tip_states and has_sets:
        pytest.skip("named ambiguity sets need tip partials (the wrapper raises, tested below)")
    dt = pc.GeneralDataTypeInterface(tc["states"], tc["ambiguities"] or None)
    subst = pc.GeneralSubstitutionModelInterface(dt, tc["rates"], tc["freqs"], tc["structure"], True)
    site = pc.GammaSiteModelInterface(tc["alpha"], tc["categories"], None, None) if tc["categories"] > 1 else pc.ConstantSiteModelInterface(None)
    tree = pc.UnRootedTreeModelInterface(tc["newick"], tc["taxa"])
    tlk = pc.TreeLikelihoodInterface(tc["taxa"], tc["values"], tree, subst, site, None, use_tip_states=tip_states)
    assert tlk.get_pattern_count() == 1
    lnl = tlk.log_likelihood()
    assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    N = gold["node_count"]
    n_site = 1 if tc["categories"] > 1 else 0
    ref = gold["gradient_all"]
    assert len(ref) == N + n_site + n_rates + K
    flags = [F.TREE_HEIGHT, F.SUBSTITUTION_MODEL] + ([F.SITE_MODEL] if n_site else [])
    tlk.request_gradient(flags)
    assert tlk.gradient_length == N - 2 + n_site + n_rates + K
    g = tlk.gradient()
    node_map = tree.node_map
    scale = max(1.0, np.abs(ref[:N]).max())
    skipped = {gold["root"], int(gold["right"][gold["root"]])}
    for i in range(N - 2):  # physher.cpp:649-655
        if i not in skipped:
            assert abs(g[node_map[i]] - ref[i]) <= 1e-9 * scale, (i, g[node_map[i]], ref[i])
    if n_site:
        assert abs(g[N - 2] - ref[N]) <= 2e-7 * max(1.0, abs(ref[N]))  # the reference's own central difference
    np.testing.assert_allclose(g[N - 2 + n_site:], ref[N + n_site:], rtol=2e-8, atol=1e-7)


def test_fluA_time_tree_hky_gamma_all_gradient_blocks():
    """Time tree + strict clock + HKY + G4 on the fluA data (fixture generated from the compiled reference by
    tests/golden/make_golden.py::run_fluA_hky_g4): lnL and the whole gradient -- 68 ratio / root-height entries, gamma shape,
    clock rate, kappa and the four frequencies -- in the reference's order (treelikelihood.c:3205-3361)."""
    from physher_amd import _phycpp_amd as pc
    F = pc.TreeLikelihoodGradientFlags
    d = os.path.join(GOLDEN, "fluA_hky_g4_time")
    with open(os.path.join(d, "hky-g4-time.json")) as f:
        js = json.load(f)["model"]
    gold = load("fluA_hky_g4_time")
    names, seqs = read_fasta(os.path.join(d, "fluA.fa"))
    dates = [float(js["tree"]["dates"][t]) for t in names]
    tree = pc.ReparameterizedTimeTreeModelInterface(js["tree"]["newick"], names, dates, pc.TreeTransformFlags.RATIO)
    clock = pc.StrictClockModelInterface(js["branchmodel"]["rate"]["value"], tree)
    sm = js["sitemodel"]["substitutionmodel"]
    subst = pc.HKYInterface(sm["rates"]["kappa"]["value"], sm["frequencies"]["values"])
    site = pc.GammaSiteModelInterface(js["sitemodel"]["distribution"]["parameters"]["alpha"]["value"], 4, None, None)
    tlk = pc.TreeLikelihoodInterface(list(zip(names, seqs)), tree, subst, site, clock, use_tip_states=True, include_jacobian=False)
    lnl = tlk.log_likelihood()
    assert abs(lnl - gold["lnl_jacobian0"]) <= 1e-10 * abs(lnl)
    ref = np.array(gold["gradient_all_time"])
    assert gold["gradient_all_time_flags"] == 1 | 2 | 4 | 64 and len(ref) == 68 + 1 + 1 + 5
    tlk.request_gradient([F.TREE_HEIGHT, F.SITE_MODEL, F.BRANCH_MODEL, F.SUBSTITUTION_MODEL])
    assert tlk.gradient_length == len(ref)
    g = tlk.gradient()
    assert np.abs(g[:68] - ref[:68]).max() <= 1e-9 * np.abs(ref[:68]).max()            # ratios, root height
    assert abs(g[68] - ref[68]) <= 2e-7 * max(1.0, abs(ref[68]))                          # gamma shape (the reference's own central difference)
    assert abs(g[69] - ref[69]) <= 1e-9 * abs(ref[69])                                    # clock rate
    np.testing.assert_allclose(g[70:], ref[70:], rtol=2e-8, atol=1e-7)                   # kappa, frequencies
    # the jacobian variant of lnL and of the tree block, as in the reference's test for JC69
    tree2 = pc.ReparameterizedTimeTreeModelInterface(js["tree"]["newick"], names, dates, pc.TreeTransformFlags.RATIO)
    clock2 = pc.StrictClockModelInterface(js["branchmodel"]["rate"]["value"], tree2)
    tlk2 = pc.TreeLikelihoodInterface(list(zip(names, seqs)), tree2, subst, site, clock2, use_tip_states=True, include_jacobian=True)
    assert abs(tlk2.log_likelihood() - gold["lnl_jacobian1"]) <= 1e-10 * abs(gold["lnl_jacobian1"])
    tlk2.request_gradient([F.TREE_HEIGHT, F.BRANCH_MODEL])
    tlk2.set_reference_compatibility(True)  # tree + clock only: the reference folds the root frequencies (inexact for this pi)
    ref1 = np.array(gold["gradient_tree_clock_jacobian1"])
    assert np.abs(tlk2.gradient() - ref1).max() <= 1e-9 * np.abs(ref1).max()


def test_fluA_time_tree_per_branch_clock_rates():
    """One clock rate per branch (the reference's "discrete" branch model = SimpleClockModelInterface, physher.cpp:188-217) on
    the fluA time tree under HKY + G4; fixture from the compiled reference (make_golden.py::run_fluA_discrete_clock).  The
    reference numbers its rates by node id, the wrapper by taxon index (tips) / class id + T (internal nodes)."""
    from physher_amd import _phycpp_amd as pc
    F = pc.TreeLikelihoodGradientFlags
    d = os.path.join(GOLDEN, "fluA_hky_g4_branch_rates")
    with open(os.path.join(d, "hky-g4-branch-rates.json")) as f:
        js = json.load(f)["model"]
    gold = load("fluA_hky_g4_branch_rates")
    names, seqs = read_fasta(os.path.join(d, "fluA.fa"))
    T = len(names)
    dates = [float(js["tree"]["dates"][t]) for t in names]
    ref_rates = js["branchmodel"]["parameters"]["values"]
    index_of = {}  # reference node id -> the wrapper's rate index
    for nd in gold["nodes"]:
        if nd["id"] != gold["root"]:
            index_of[nd["id"]] = names.index(nd["name"]) if nd["left"] < 0 else nd["class_id"] + T
    assert sorted(index_of.values()) == list(range(2 * T - 2)) and sorted(index_of) == list(range(2 * T - 2))
    rates = np.zeros(2 * T - 2)
    for node, idx in index_of.items():
        rates[idx] = ref_rates[node]
    tree = pc.ReparameterizedTimeTreeModelInterface(js["tree"]["newick"], names, dates, pc.TreeTransformFlags.RATIO)
    clock = pc.SimpleClockModelInterface(list(rates), tree)
    sm = js["sitemodel"]["substitutionmodel"]
    subst = pc.HKYInterface(sm["rates"]["kappa"]["value"], sm["frequencies"]["values"])
    site = pc.GammaSiteModelInterface(js["sitemodel"]["distribution"]["parameters"]["alpha"]["value"], 4, None, None)
    tlk = pc.TreeLikelihoodInterface(list(zip(names, seqs)), tree, subst, site, clock, use_tip_states=True, include_jacobian=False)
    lnl = tlk.log_likelihood()
    assert abs(lnl - gold["lnl_jacobian0"]) <= 1e-10 * abs(lnl)
    ref = np.array(gold["gradient_all_time"])
    n_clock = 2 * T - 2
    assert len(ref) == (T - 1) + 1 + n_clock + 5
    tlk.request_gradient([F.TREE_HEIGHT, F.SITE_MODEL, F.BRANCH_MODEL, F.SUBSTITUTION_MODEL])
    assert tlk.gradient_length == len(ref)
    g = tlk.gradient()
    assert np.abs(g[:T - 1] - ref[:T - 1]).max() <= 1e-9 * np.abs(ref[:T - 1]).max()          # ratios, root height
    assert abs(g[T - 1] - ref[T - 1]) <= 2e-7 * max(1.0, abs(ref[T - 1]))                       # gamma shape
    gc, rc = g[T: T + n_clock], ref[T: T + n_clock]
    scale = np.abs(rc).max()
    for node, idx in index_of.items():                                                          # one rate per branch
        assert abs(gc[idx] - rc[node]) <= 1e-9 * scale, (node, idx, gc[idx], rc[node])
    np.testing.assert_allclose(g[T + n_clock:], ref[T + n_clock:], rtol=2e-8, atol=1e-7)         # kappa, frequencies
