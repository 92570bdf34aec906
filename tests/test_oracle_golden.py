"""Pin the CPU oracle (oracle/phyoracle.c) against the compiled reference's golden vectors.

CPU only.  Tolerances: pattern compression exact; lnL 1e-10 relative; per-pattern lnL and partials
1e-11 relative; gradient 1e-9 * max(1, |g|_inf).
"""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, UNROOTED_CASES, load, oracle_problem, read_fasta, read_spec, reversible_eigen
from oracle import phyoracle as po


@pytest.mark.parametrize("case", UNROOTED_CASES)
def test_pattern_compression_bit_exact(case):
    gold = load(case)
    spec = read_spec(case)
    names, seqs = read_fasta(os.path.join(GOLDEN, case, "aln.fa"))
    assert names == gold["taxa"]
    cols = po.encode_alignment(spec["datatype"], seqs)
    patterns, weights = po.compress_patterns(cols)
    assert patterns.shape == gold["patterns"].shape
    assert np.array_equal(patterns, gold["patterns"])
    assert np.array_equal(weights, gold["weights"])


def test_pattern_compression_fluA_bit_exact():
    gold = load("fluA_jc69_time")
    names, seqs = read_fasta(os.path.join(GOLDEN, "fluA_jc69_time", "fluA.fa"))
    assert names == gold["taxa"]
    patterns, weights = po.compress_patterns(po.encode_alignment("nucleotide", seqs))
    assert patterns.shape == (69, 238)
    assert np.array_equal(patterns, gold["patterns"])
    assert np.array_equal(weights, gold["weights"])
    assert weights.sum() == 987


def test_pattern_compression_edge_cases():
    # a single column, all-identical columns, and enough distinct columns to force table growth
    p, w = po.compress_patterns(np.array([[0, 1, 2]], dtype=np.uint8))
    assert p.shape == (3, 1) and w.tolist() == [1.0]
    p, w = po.compress_patterns(np.zeros((50, 4), dtype=np.uint8))
    assert p.shape == (4, 1) and w.tolist() == [50.0]
    rng = np.random.default_rng(0)
    cols = rng.integers(0, 4, size=(5000, 12), dtype=np.uint8)
    p, w = po.compress_patterns(cols)
    uniq = np.unique(cols, axis=0)
    assert p.shape[1] == len(uniq) and w.sum() == 5000
    assert {tuple(c) for c in p.T} == {tuple(c) for c in uniq}


@pytest.mark.parametrize("case", [c for c in UNROOTED_CASES if "pt" in load(c)])
def test_transition_matrices(case):
    gold = load(case)
    S, C = gold["state_count"], gold["category_count"]
    for q, node in enumerate(gold["pt_nodes"]):
        for c in range(C):
            t = gold["distance"][node] * gold["cat_rates"][c]
            P = po.p_t(S, gold["eval"], gold["evec"], gold["ivec"], t)
            dP = po.p_t(S, gold["eval"], gold["evec"], gold["ivec"], t, derivative=True)
            np.testing.assert_allclose(P, gold["pt"][q, c], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(dP, gold["dpt"][q, c], rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("case", UNROOTED_CASES)
def test_log_likelihood(case):
    gold = load(case)
    pb = oracle_problem(case, gold)
    r = pb.log_likelihood(want_lower=True)
    assert r["rescaled"] == gold["rescaled"]
    assert abs(r["lnl"] - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    np.testing.assert_allclose(r["pattern_lk"], gold["pattern_lk"], rtol=1e-11, atol=1e-11)
    if "partials_root" in gold:
        T = gold["tip_count"]
        np.testing.assert_allclose(r["lower"][gold["root"]], gold["partials_root"], rtol=1e-9, atol=_atol(case))
        np.testing.assert_allclose(r["lower"][T], gold["partials_first_internal"], rtol=1e-9, atol=_atol(case))


def _atol(case):
    # with an invariant class (rate 0) the reference's closed-form JC69/HKY matrices are exactly the identity while an
    # eigen-system P(0) has 1e-17 off-diagonals: partial entries that are exactly 0 there are ~1e-17 here
    return 1e-15 if "pinv" in read_spec(case) else 1e-300


def _branch_gradient(gold, res):
    return po.branch_gradient_from_cat(res["cat_grad"], gold["cat_rates_without_mu"], gold["cat_proportions"],
                                       zero_node=gold["right"][gold["root"]])


@pytest.mark.parametrize("case", UNROOTED_CASES)
@pytest.mark.parametrize("fold", [1, 0])
def test_branch_gradient(case, fold):
    """Oracle vs the reference's branch-length gradient, in both of the reference's modes.

    fold=1: gradient requested with TREELIKELIHOOD_FLAG_TREE_MODEL only -> include_root_freqs = true
            (fixture key gradient_tree).  NOTE: that mode of the reference is only correct for uniform
            frequencies (it propagates pi-weighted uppers with P instead of P^T); reproduced here to pin the
            restatement, not shipped as the default.
    fold=0: substitution-model gradient requested too -> include_root_freqs = false (gradient_all[:N]);
            this is d lnL / d t (see the finite-difference test below).
    Under rescaling the reference divides by the per-category site likelihood (treelikelihood.c:2851-2870);
    compat_scaled_gradient reproduces that for the comparison.
    """
    gold = load(case)
    spec = read_spec(case)
    N = gold["node_count"]
    if not fold and not (gold["gradient_all_flags"] & 4):
        pytest.skip("model has no dPdp: the reference never runs include_root_freqs = false for it")
    ref = gold["gradient_tree"] if fold else gold["gradient_all"][:N]
    scaled_multi_cat = gold["rescaled"] and gold["category_count"] > 1
    pb = oracle_problem(case, gold, compat_scaled_gradient=1 if scaled_multi_cat else 0, fold_root_freqs=fold)
    r = pb.gradient(want_partials=True)
    g = _branch_gradient(gold, r)
    finite = np.isfinite(ref)  # the reference's rescaled gradient is NaN where a category's likelihood underflows
    assert finite.sum() >= len(ref) // 4
    scale = max(1.0, np.abs(ref[finite]).max())
    assert np.abs(g[finite] - ref[finite]).max() <= 1e-9 * scale
    if fold and "upper_first_internal" in gold:
        T = gold["tip_count"]
        np.testing.assert_allclose(r["upper"][T], gold["upper_first_internal"], rtol=1e-9, atol=_atol(case))
        if spec["tipstates"] == "0":
            np.testing.assert_allclose(r["upper"][0], gold["upper_tip0"], rtol=1e-9, atol=_atol(case))


def test_reference_folded_gradient_is_wrong_for_nonuniform_frequencies():
    """Documents the reference quirk: its TREE_MODEL-only gradient differs from its own full-flag gradient."""
    gold = load("gtr_g4_t16")
    N = gold["node_count"]
    assert np.abs(gold["gradient_tree"] - gold["gradient_all"][:N]).max() > 1.0
    gold = load("jc69_t12")  # uniform frequencies: both agree
    N = gold["node_count"]
    assert np.abs(gold["gradient_tree"] - gold["gradient_all"][:N]).max() < 1e-9


@pytest.mark.parametrize("case", ["gtr_g4_t16", "gtr_g4_t96_rescale", "wag_g4_t12"])
def test_gradient_matches_finite_differences(case):
    """The consistent (non-compat) gradient is d lnL / d branch length, also under rescaling."""
    gold = load(case)
    pb = oracle_problem(case, gold)
    g = _branch_gradient(gold, pb.gradient())
    root, rr = gold["root"], gold["right"][gold["root"]]
    rng = np.random.default_rng(1)
    nodes = [n for n in rng.permutation(gold["node_count"]) if n not in (root, rr)][:6]
    for n in nodes:
        bl = pb.branch_lengths.copy()
        h = 1e-6
        pb.branch_lengths = bl.copy(); pb.branch_lengths[n] += h
        up = pb.log_likelihood()["lnl"]
        pb.branch_lengths = bl.copy(); pb.branch_lengths[n] -= h
        dn = pb.log_likelihood()["lnl"]
        pb.branch_lengths = bl
        fd = (up - dn) / (2 * h)
        assert abs(fd - g[n]) <= 2e-5 * max(1.0, abs(g[n])), (n, fd, g[n])


def test_fluA_reference_constants():
    """The reference's own known-answer constants (tests/test_tree_likelihood.c:29,88) as re-emitted by the driver."""
    gold = load("fluA_jc69_time")
    assert abs(gold["lnl_jacobian0"] - (-4777.616349713985)) < 1e-8
    assert abs(gold["lnl_jacobian1"] - (-4786.867701371271)) < 1e-8
    g = np.array(gold["gradient_tree_clock_jacobian0"])
    assert abs(g[67] - 17.492484957839924) < 1e-8       # root height (test_tree_likelihood.c:77)
    assert abs(g[68] - 328017.6732813406) < 1e-8        # clock rate (:38)
    assert abs(g[0] - (-0.5936536642214764)) < 1e-8     # first ratio (:53)


def test_fluA_lnl_from_branch_lengths():
    """Oracle lnL on the reference's known-answer case, fed with the time-tree's branch lengths (rate * elapsed time)."""
    gold = load("fluA_jc69_time")
    T = gold["tip_count"]
    order = gold["mapping"][:T]
    pb = po.Problem(gold["left"], gold["right"], gold["root"], gold["weights"], gold["eval"], gold["evec"], gold["ivec"],
                    gold["frequencies"], gold["cat_rates"], gold["cat_proportions"], np.array(gold["branch_lengths"]),
                    tip_states=gold["patterns"][order])
    pb.eval, pb.evec, pb.ivec = [np.ascontiguousarray(a) for a in reversible_eigen(np.ones((4, 4)), gold["frequencies"])]
    r = pb.log_likelihood()
    assert abs(r["lnl"] - (-4777.616349713985)) < 1e-8
