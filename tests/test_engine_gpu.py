"""GPU parity tests: the HIP engine (through the C ABI) against the reference's golden vectors and the CPU oracle.

Tolerances: lnL 1e-10 relative; per-pattern lnL 1e-11 (+1e-11 abs); partials 1e-9 relative;
gradient 1e-9 * max(1, |g|_inf).
"""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, UNROOTED_CASES, load, oracle_problem, read_spec
from gpu_util import engine_from_problem, random_problem
from oracle import phyoracle as po
from physher_amd.engine import GRAD_COMPAT_SCALED, GRAD_FOLD_ROOT_FREQS, RESCALE_ALWAYS, RESCALE_AUTO, RESCALE_NEVER, Engine, EngineError

pytestmark = pytest.mark.gpu


def _atol(case):
    # with an invariant class (rate 0) the reference's closed-form JC69/HKY matrices are exactly the identity while an
    # eigen-system P(0) has 1e-17 off-diagonals: partial entries that are exactly 0 there are ~1e-17 here
    return 1e-15 if "pinv" in read_spec(case) else 1e-300

CASES4 = [c for c in UNROOTED_CASES if read_spec(c)["datatype"] == "nucleotide"]


def _tip_mode(case):
    return "states" if read_spec(case)["tipstates"] == "1" else "partials"


def _rescale(case):
    return RESCALE_ALWAYS if read_spec(case)["rescale"] == "1" else RESCALE_AUTO


def _grad_tol(ref):
    return 1e-9 * max(1.0, np.abs(ref[np.isfinite(ref)]).max())


@pytest.mark.parametrize("case", CASES4)
def test_golden_log_likelihood(case):
    gold = load(case)
    pb = oracle_problem(case, gold)
    with engine_from_problem(pb, rescale=_rescale(case), tip_mode=_tip_mode(case)) as e:
        lnl = e.log_likelihood()
        assert e.rescaling == gold["rescaled"]
        assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
        np.testing.assert_allclose(e.pattern_log_likelihoods(), gold["pattern_lk"], rtol=1e-11, atol=1e-11)
        if "partials_root" in gold:
            e.set_keep_partials(True)  # store every node (the default schedule fuses cherries into their parents)
            assert abs(e.log_likelihood() - lnl) <= 1e-12 * abs(lnl)
            np.testing.assert_allclose(e.partials(gold["root"]), gold["partials_root"], rtol=1e-9, atol=_atol(case))
            np.testing.assert_allclose(e.partials(gold["tip_count"]), gold["partials_first_internal"], rtol=1e-9, atol=_atol(case))
        if "pt" in gold:
            for q, node in enumerate(gold["pt_nodes"]):
                np.testing.assert_allclose(e.node_matrices(node), gold["pt"][q], rtol=1e-12, atol=1e-15)
                np.testing.assert_allclose(e.node_matrices(node, derivative=True), gold["dpt"][q], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("case", CASES4)
@pytest.mark.parametrize("fold", [0, 1])
def test_golden_branch_gradient(case, fold):
    """fold=1 / COMPAT reproduce the reference's two quirks (see tests/test_oracle_golden.py); fold=0 is the default."""
    gold = load(case)
    N = gold["node_count"]
    if not fold and not (gold["gradient_all_flags"] & 4):
        pytest.skip("reference never runs include_root_freqs = false for this model")
    ref = gold["gradient_tree"] if fold else gold["gradient_all"][:N]
    pb = oracle_problem(case, gold)
    flags = (GRAD_FOLD_ROOT_FREQS if fold else 0) | (GRAD_COMPAT_SCALED if gold["rescaled"] and gold["category_count"] > 1 else 0)
    with engine_from_problem(pb, rescale=_rescale(case), tip_mode=_tip_mode(case)) as e:
        lnl_f, cg_f = e.gradient(flags)  # default schedule: cherries / cherry+tip nodes fused into their parents
        e.set_keep_partials(True)        # unfused schedule, every node stored
        lnl, cg = e.gradient(flags)
        assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"]) and abs(lnl_f - lnl) <= 1e-12 * abs(lnl)
        both = np.isfinite(cg) & np.isfinite(cg_f)
        # (per-category ratios of denormal numbers -- see below -- depend on the arithmetic path at the 1e-6 level)
        assert np.abs(cg[both] - cg_f[both]).max() <= (1e-5 if case == "gtr_g4_t700_autorescale" else 1e-10) * max(1.0, np.abs(cg[both]).max())
        g = po.branch_gradient_from_cat(cg, gold["cat_rates_without_mu"], gold["cat_proportions"], zero_node=gold["right"][gold["root"]])
        # the reference's rescaled multi-category gradient is NaN wherever one category underflows (60 % of the
        # branches of gtr_g4_t700_autorescale); the compat mode underflows in slightly different places
        finite = np.isfinite(ref) & np.isfinite(g)
        assert finite.sum() >= np.isfinite(ref).sum() * 0.5
        tol = _grad_tol(ref)
        if case == "gtr_g4_t700_autorescale":
            # per-category ratios of denormal numbers (1e-310 / 1e-310): a handful of entries keep only ~6 digits in
            # the reference and here alike; the default (non-compat) mode is checked at full precision below
            tol *= 1e4
        assert np.abs(g[finite] - ref[finite]).max() <= tol
        # the C-ABI epilogue gives the same numbers (root->right is not zeroed there: that is the caller's convention)
        lnl2, bg = e.branch_gradient(flags, rates_without_mu=gold["cat_rates_without_mu"])
        keep = finite & np.isfinite(bg)
        keep[gold["right"][gold["root"]]] = False
        assert np.abs(bg[keep] - ref[keep]).max() <= tol
        if fold and "upper_first_internal" in gold and not gold["rescaled"]:
            np.testing.assert_allclose(e.partials(gold["tip_count"], upper=True), gold["upper_first_internal"], rtol=1e-9, atol=_atol(case))


def test_autorescale_default_gradient_is_finite_and_matches_oracle():
    """Where the reference's rescaled gradient is NaN/ill-conditioned, the default mode is finite and exact."""
    gold = load("gtr_g4_t700_autorescale")
    pb = oracle_problem("gtr_g4_t700_autorescale", gold)
    ref = pb.gradient()
    assert ref["rescaled"] and np.all(np.isfinite(ref["cat_grad"]))
    with engine_from_problem(pb, rescale=RESCALE_AUTO, tip_mode="partials") as e:
        lnl, cg = e.gradient()
    assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    assert np.all(np.isfinite(cg))
    assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())


def _compare_with_oracle(pb, rescale, flags=0, tip_mode="states", check_partials=False):
    ref = pb.gradient(want_partials=check_partials)
    with engine_from_problem(pb, rescale=rescale, tip_mode=tip_mode) as e:
        if check_partials:
            # first the default (fused) schedule, then the one that stores every node
            lnl_f, cg_f = e.gradient(flags)
            assert abs(lnl_f - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
            assert np.abs(cg_f - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
            e.set_keep_partials(True)
        lnl_only = e.log_likelihood()
        lnl, cg = e.gradient(flags)
        if (flags & GRAD_COMPAT_SCALED) and e.rescaling:
            # the reference's per-category arithmetic reads stored partials in the reference's own rescaling convention: the post-order
            # pass has run again (the streamed walks keep powers of two per category between them), other factors, the same lnL to rounding
            assert abs(lnl - lnl_only) <= 1e-13 * abs(lnl)
        else:
            assert lnl == lnl_only  # fixed-order reductions: bitwise reproducible
        assert e.rescaling == ref["rescaled"]
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        np.testing.assert_allclose(e.pattern_log_likelihoods(), ref["pattern_lk"], rtol=1e-11, atol=1e-11)
        tol = 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
        assert np.abs(cg - ref["cat_grad"]).max() <= tol
        if check_partials:
            for n in range(pb.T, pb.N):
                np.testing.assert_allclose(e.partials(n), ref["lower"][n], rtol=1e-9, atol=1e-300)
            if not ref["rescaled"]:
                for n in range(pb.N):
                    if n != pb.root:
                        np.testing.assert_allclose(e.partials(n, upper=True), ref["upper"][n], rtol=1e-9, atol=1e-300)
        return lnl, cg


@pytest.mark.parametrize("T,P,C", [(2, 1, 1), (3, 63, 1), (5, 64, 2), (8, 65, 3), (16, 1000, 4), (33, 257, 5), (12, 129, 6), (9, 300, 8),
                                   (10, 77, 16)])
def test_against_oracle_shapes(T, P, C):
    """ragged pattern counts (not multiples of the wave / workgroup), every category count incl. the 16-wave maximum"""
    pb = random_problem(T, P, C, seed=100 + T + P + C)
    _compare_with_oracle(pb, RESCALE_NEVER, check_partials=True)


@pytest.mark.parametrize("shape", ["caterpillar", "balanced", "random"])
def test_against_oracle_tree_shapes(shape):
    pb = random_problem(40, 500, 4, seed=7, shape=shape, gaps=0.03)
    _compare_with_oracle(pb, RESCALE_NEVER, check_partials=True)


def test_against_oracle_fold_flag():
    pb = random_problem(20, 300, 4, seed=9, fold_root_freqs=1)
    _compare_with_oracle(pb, RESCALE_NEVER, flags=GRAD_FOLD_ROOT_FREQS)


@pytest.mark.parametrize("C", [1, 4])
def test_against_oracle_forced_rescaling(C):
    pb = random_problem(150, 200, C, seed=11, bl=(0.3, 0.9), rescale=1)
    _compare_with_oracle(pb, RESCALE_ALWAYS, check_partials=True)
    pb.compat_scaled_gradient = 1
    _compare_with_oracle(pb, RESCALE_ALWAYS, flags=GRAD_COMPAT_SCALED)


def test_lazy_rescaling_switch():
    """+-inf lnL without rescaling turns it on for good and recomputes (treelikelihood.c:1496-1519)."""
    pb = random_problem(900, 64, 4, seed=13, bl=(0.5, 1.5), rescale=2)
    ref = pb.gradient()
    assert ref["rescaled"] and np.isfinite(ref["lnl"])
    with engine_from_problem(pb, rescale=RESCALE_AUTO) as e:
        assert not e.rescaling
        lnl, cg = e.gradient()
        assert e.rescaling
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:  # in-band failure like the reference: -inf and NaN gradient
        lnl, cg = e.gradient()
        assert np.isinf(lnl) and lnl < 0 and np.all(np.isnan(cg))


@pytest.mark.parametrize("shape,T,P,C", [("caterpillar", 5000, 200, 2), ("random", 8000, 150, 4), ("balanced", 4096, 100, 1)])
def test_very_large_trees(shape, T, P, C):
    """thousands of taxa (deepest possible, random, and perfectly balanced shapes): schedules, parked-upper slots and the lazy
    rescaling switch at sizes where every pattern underflows without it; against the oracle"""
    pb = random_problem(T, P, C, seed=T, shape=shape, gaps=0.02, rescale=2)
    ref = pb.gradient()
    assert ref["rescaled"]
    with engine_from_problem(pb, rescale=RESCALE_AUTO) as e:
        lnl, cg = e.gradient()
        assert e.rescaling
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())


def test_medium_problem_and_shard_additivity():
    """cfg2-like shape at reduced size vs the oracle, then the sharding property used for multi-GPU:
    lnL and the per-category gradient are sums over disjoint pattern shards."""
    pb = random_problem(120, 6000, 4, seed=21)
    lnl, cg = _compare_with_oracle(pb, RESCALE_NEVER)
    cut = 2500
    parts = []
    for sl in (slice(0, cut), slice(cut, pb.P)):
        sub = po.Problem(pb.left, pb.right, pb.root, pb.weights[sl], pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props,
                         pb.branch_lengths, tip_states=pb.tip_states[:, sl])
        with engine_from_problem(sub, rescale=RESCALE_NEVER) as e:
            parts.append(e.gradient())
    assert abs(parts[0][0] + parts[1][0] - lnl) <= 1e-11 * abs(lnl)
    assert np.abs(parts[0][1] + parts[1][1] - cg).max() <= 1e-10 * max(1.0, np.abs(cg).max())


def test_full_size_properties():
    """BASELINE configs[1] shape (500 taxa x 1e5 patterns x 4 states x 4 categories) -- too big for the oracle in
    seconds, so size-independent properties: determinism, weight linearity, duplicated-pattern invariance,
    gradient vs central finite differences of lnL."""
    from physher_amd import synth
    rng = np.random.default_rng(5)
    T, P, C = 500, 100_000, 4
    tree = synth.random_tree(T, rng)
    states, weights = synth.distinct_patterns(tree, P, 4, rng)
    pb = random_problem(T, 8, C, seed=5)  # borrow a model
    with Engine(T, P, 4, C, rescale=RESCALE_AUTO) as e:
        e.set_topology(tree.left, tree.right, tree.root)
        e.set_branch_lengths(tree.length)
        e.set_eigen(pb.eval, pb.evec, pb.ivec)
        e.set_frequencies(pb.freqs)
        e.set_category_rates(pb.cat_rates, pb.cat_props)
        e.set_pattern_weights(weights)
        for t in range(T):
            e.set_tip_states(t, states[t])
        lnl, cg = e.gradient()
        lnl2, cg2 = e.gradient()
        assert lnl == lnl2 and np.array_equal(cg, cg2)
        plk = e.pattern_log_likelihoods()
        assert abs(np.dot(plk, weights) - lnl) <= 1e-11 * abs(lnl)
        # linearity in the weights
        e.set_pattern_weights(2.0 * weights)
        lnl3, cg3 = e.gradient()
        assert abs(lnl3 - 2 * lnl) <= 1e-12 * abs(lnl) and np.abs(cg3 - 2 * cg).max() <= 1e-11 * np.abs(cg).max()
        e.set_pattern_weights(weights)
        # finite differences on a few branches
        bg = po.branch_gradient_from_cat(cg, pb.cat_rates, pb.cat_props)
        for n in rng.choice([i for i in range(2 * T - 1) if i != tree.root], size=3, replace=False):
            h = 1e-6
            bl = tree.length.copy(); bl[n] += h
            e.set_branch_lengths(bl); up = e.log_likelihood()
            bl[n] -= 2 * h
            e.set_branch_lengths(bl); dn = e.log_likelihood()
            fd = (up - dn) / (2 * h)
            assert abs(fd - bg[n]) <= 1e-4 * max(1.0, abs(bg[n])), (n, fd, bg[n])
        e.set_branch_lengths(tree.length)


def test_headline_size_properties():
    """BASELINE's metric shape itself -- 1000 taxa x 1e6 patterns x 4 states x 4 categories on one GPU (48.7 GB resident):
    bitwise reproducibility, sum of the per-pattern lnL, additivity over two half-size shards (what multi-GPU sharding
    relies on), equality with the tiled engine under a 16 GB cap, and the gradient against a central difference of lnL."""
    import os
    from physher_amd import synth
    T, P, C = 1000, 1_000_000, 4
    rng = np.random.default_rng(1)
    tree = synth.random_tree(T, rng)
    # 1e5 evolved columns repeated ten times with independent weights: the kernels see 1e6 patterns either way, and host-side
    # generation stays at seconds (torch is not used here: its bundled HIP runtime cannot initialise after the engine's)
    states = np.tile(synth.evolve(tree, P // 10, 4, rng), (1, 10))
    model = random_problem(T, 8, C, seed=5)  # borrow a GTR-like model and gamma rates
    ev, U, Ui, freqs, rates, props = model.eval, model.evec, model.ivec, model.freqs, model.cat_rates, model.cat_props
    weights = rng.integers(1, 4, size=P).astype(np.float64)

    def make(lo, hi, **kw):
        e = Engine(T, hi - lo, 4, C, rescale=RESCALE_AUTO, **kw)
        e.set_topology(tree.left, tree.right, tree.root)
        e.set_branch_lengths(tree.length)
        e.set_eigen(ev, U, Ui)
        e.set_frequencies(freqs)
        e.set_category_rates(rates, props)
        e.set_pattern_weights(weights[lo:hi])
        for t in range(T):
            e.set_tip_states(t, np.ascontiguousarray(states[t, lo:hi]))
        return e

    with make(0, P) as e:
        if not any(v in os.environ for v in ("PHYAMD_FUSE", "PHYAMD_DEEP", "PHYAMD_WALK")):  # (A/B switches change what is stored)
            assert 40e9 < e.profile()["device_bytes"] < 60e9
        lnl, cg = e.gradient()
        lnl2, cg2 = e.gradient()
        assert np.isfinite(lnl) and lnl == lnl2 and np.array_equal(cg, cg2)
        assert not e.rescaling
        assert abs(np.dot(e.pattern_log_likelihoods(), weights) - lnl) <= 1e-11 * abs(lnl)
        bg = po.branch_gradient_from_cat(cg, rates, props)
        n = int(tree.left[tree.root])
        h = 1e-6
        bl = tree.length.copy()
        bl[n] += h
        e.set_branch_lengths(bl)
        up = e.log_likelihood()
        bl[n] -= 2 * h
        e.set_branch_lengths(bl)
        dn = e.log_likelihood()
        assert abs((up - dn) / (2 * h) - bg[n]) <= 1e-4 * max(1.0, abs(bg[n]))
    halves = []
    for lo, hi in ((0, P // 2), (P // 2, P)):
        with make(lo, hi) as h_:
            halves.append(h_.gradient())
    assert abs(halves[0][0] + halves[1][0] - lnl) <= 1e-12 * abs(lnl)
    assert np.abs(halves[0][1] + halves[1][1] - cg).max() <= 1e-11 * np.abs(cg).max()
    with make(0, P, max_device_bytes=16_000_000_000) as tiled:
        assert tiled.profile()["tiles"] >= 4
        if "PHYAMD_FUSE" not in os.environ:  # (the unfused A/B schedule stores every node: more than the estimate behind the tile count)
            assert tiled.profile()["device_bytes"] < 16e9
        lnl_t, cg_t = tiled.gradient()
        assert abs(lnl_t - lnl) <= 1e-12 * abs(lnl)
        assert np.abs(cg_t - cg).max() <= 1e-11 * np.abs(cg).max()


def test_fusion_switch_and_memory():
    """PHYAMD_FUSE=0 gives the unfused schedule; results agree and the fused engine stores about half the arrays."""
    import os
    pb = random_problem(200, 3000, 4, seed=31, gaps=0.02)
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        lnl_f, cg_f = e.gradient()
        e.set_profiling(True)
        bytes_f = e.profile()["device_bytes"]
        with pytest.raises(EngineError):
            e.partials(int(np.argmax((pb.left[pb.T:] < pb.T) & (pb.right[pb.T:] < pb.T))) + pb.T)  # a cherry: not stored
    os.environ["PHYAMD_FUSE"] = "0"
    try:
        with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
            lnl_u, cg_u = e.gradient()
            bytes_u = e.profile()["device_bytes"]
    finally:
        del os.environ["PHYAMD_FUSE"]
    assert abs(lnl_f - lnl_u) <= 1e-12 * abs(lnl_u)
    assert np.abs(cg_f - cg_u).max() <= 1e-10 * max(1.0, np.abs(cg_u).max())
    assert bytes_f < 0.75 * bytes_u


@pytest.mark.parametrize("T,P,C,shape", [(300, 2500, 4, "random"), (120, 777, 1, "caterpillar"), (64, 1000, 2, "balanced"), (50, 300, 3, "random"),
                                         (40, 500, 8, "random"), (33, 129, 16, "random"), (3, 70, 4, "random"), (2, 5, 4, "random")])
def test_tree_walk_kernels_match_level_kernels(T, P, C, shape):
    """The depth-first tree-walk kernels (default for unscaled 4-state evaluations: one launch per pass, child/parent
    partials handed on in registers / LDS) against the level-batched kernels (PHYAMD_WALK=0) and the oracle; fused and
    unfused fringe; the walk parks far fewer upper partials."""
    import os
    pb = random_problem(T, P, C, seed=900 + T + C, shape=shape, gaps=0.03)
    ref = pb.gradient()
    res = {}
    for walk, fuse in ((1, 1), (0, 1), (1, 0)):
        os.environ["PHYAMD_WALK"], os.environ["PHYAMD_FUSE"] = str(walk), str(fuse)
        try:
            with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
                lnl0 = e.log_likelihood()
                lnl, cg = e.gradient()
                lnl2, cg2 = e.gradient()
                assert lnl == lnl2 and np.array_equal(cg, cg2)  # fixed-order reductions in both forms
                assert abs(lnl - lnl0) <= 1e-13 * abs(lnl)
                e.set_profiling(True)
                res[(walk, fuse)] = (lnl, cg, e.pattern_log_likelihoods(), e.profile()["device_bytes"])
        finally:
            del os.environ["PHYAMD_WALK"], os.environ["PHYAMD_FUSE"]
    tol = 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
    for key, (lnl, cg, plk, _) in res.items():
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]), key
        assert np.abs(cg - ref["cat_grad"]).max() <= tol, key
        np.testing.assert_allclose(plk, ref["pattern_lk"], rtol=1e-11, atol=1e-11)
    assert np.abs(res[(1, 1)][1] - res[(0, 1)][1]).max() <= 1e-11 * max(1.0, np.abs(ref["cat_grad"]).max())
    if T >= 100 and P >= 2000:
        assert res[(1, 1)][3] < res[(0, 1)][3]


def test_error_behaviour():
    with Engine(4, 10, 4, 2) as e:
        with pytest.raises(EngineError) as ei:
            e.log_likelihood()
        assert "topology" in str(ei.value)
        with pytest.raises(EngineError):
            e.set_topology([-1, -1, -1, -1, 0, 1, 2], [-1, -1, -1, -1, 1, 2, 3], 6)  # node 1 and 2 with two parents
        e.set_topology([-1, -1, -1, -1, 0, 4, 5], [-1, -1, -1, -1, 1, 2, 3], 6)
        with pytest.raises(EngineError) as ei:
            e.log_likelihood()
        assert "branch_lengths" in str(ei.value)
    with pytest.raises(EngineError):
        Engine(4, 10, 4, 17)  # more categories than waves in a workgroup
    with pytest.raises(EngineError) as ei:
        Engine(4, 10, 7, 4)  # no kernels for 7 states
    assert ei.value.code == -4  # loud, not a silent CPU fallback


def test_explicit_matrices_jc69():
    """Closed-form models hand P(t) matrices and Q to the engine instead of an eigen system."""
    gold = load("jc69_t12")
    pb = oracle_problem("jc69_t12", gold)
    with engine_from_problem(pb, rescale=RESCALE_NEVER, tip_mode="partials") as e:
        lnl_eig, cg_eig = e.gradient()
    with Engine(pb.T, pb.P, 4, 1, rescale=RESCALE_NEVER) as e:
        e.set_topology(pb.left, pb.right, pb.root)
        e.set_branch_lengths(pb.branch_lengths)
        e.set_frequencies(pb.freqs)
        e.set_category_rates(pb.cat_rates, pb.cat_props)
        e.set_pattern_weights(pb.weights)
        for t in range(pb.T):
            e.set_tip_states(t, pb.tip_states[t])
        for n in range(pb.N):
            if n == pb.root:
                continue
            t = pb.branch_lengths[n]
            ex = np.exp(-4.0 / 3.0 * t)  # jc69.c:73-79
            Pm = np.full((4, 4), 0.25 - 0.25 * ex) + np.eye(4) * ex
            e.set_node_matrices(n, Pm[None])
        with pytest.raises(EngineError):
            e.gradient()  # no rate matrix yet
        Q = np.full((4, 4), 1.0 / 3.0) - np.eye(4) * (4.0 / 3.0)
        e.set_rate_matrix(Q)
        lnl, cg = e.gradient()
    assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
    assert np.abs(cg - cg_eig).max() <= 1e-9 * max(1.0, np.abs(cg_eig).max())


# ------------------------------------------------------------------------------------------------------------
# 20 / 60 / 61 states: the fp64-MFMA kernels (K9)
# ------------------------------------------------------------------------------------------------------------
CASES_GEN = [c for c in UNROOTED_CASES if read_spec(c)["datatype"] in ("aa", "codon")]


@pytest.mark.parametrize("case", CASES_GEN)
def test_golden_generic_states(case):
    """WAG / LG (20 states) and MG94 (61 states) against the compiled reference."""
    gold = load(case)
    N = gold["node_count"]
    pb = oracle_problem(case, gold)
    with engine_from_problem(pb, rescale=_rescale(case), tip_mode=_tip_mode(case)) as e:
        e.set_keep_partials(True)
        lnl = e.log_likelihood()
        assert e.rescaling == gold["rescaled"]
        assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
        # codon P(t) entries for multi-nucleotide changes are ~1e-10 and come out of sums that cancel to 1e-17 absolute:
        # their RELATIVE accuracy is ~1e-7 in the reference and here alike, and carries into the small partial entries
        codon = gold["state_count"] > 20
        np.testing.assert_allclose(e.pattern_log_likelihoods(), gold["pattern_lk"], rtol=1e-9 if codon else 1e-11, atol=1e-11)
        prtol = 1e-6 if codon else 1e-9
        if "partials_root" in gold:  # the rescaled fixtures are slim (no partial arrays)
            np.testing.assert_allclose(e.partials(gold["root"]), gold["partials_root"], rtol=prtol, atol=1e-300)
            np.testing.assert_allclose(e.partials(gold["tip_count"]), gold["partials_first_internal"], rtol=prtol, atol=1e-300)
            for q, node in enumerate(gold["pt_nodes"]):
                np.testing.assert_allclose(e.node_matrices(node), gold["pt"][q], rtol=1e-11, atol=1e-15)
        # the reference differentiates these models with include_root_freqs = true only (no dPdp): FOLD reproduces it;
        # rescaled with several categories it divides by per-category likelihoods (COMPAT) and is NaN where one underflows
        multi = gold["rescaled"] and gold["category_count"] > 1
        lnl2, cg = e.gradient(GRAD_FOLD_ROOT_FREQS | (GRAD_COMPAT_SCALED if multi else 0))
        assert lnl2 == lnl
        g = po.branch_gradient_from_cat(cg, gold["cat_rates_without_mu"], gold["cat_proportions"], zero_node=gold["right"][gold["root"]])
        ref = gold["gradient_tree"]
        finite = np.isfinite(ref) & np.isfinite(g)
        assert finite.sum() >= 0.5 * np.isfinite(ref).sum()
        # the reference's 61-state eigen system (orthes + hqr2) satisfies U U^-1 = I only to ~1e-9, so U L e^{Lt} U^-1 p
        # (reference) and Q (P p) with Q = U L U^-1 (here) agree to ~2e-9 relative instead of 1e-9
        assert np.abs(g[finite] - ref[finite]).max() <= _grad_tol(ref) * (10 if codon else 1) * (1e3 if multi else 1)
        if "upper_first_internal" in gold and not gold["rescaled"]:
            np.testing.assert_allclose(e.partials(gold["tip_count"], upper=True), gold["upper_first_internal"], rtol=prtol, atol=1e-300)
        if gold["rescaled"]:  # the default mode (mixture likelihood in the branch's units) is finite everywhere and exact
            orc = oracle_problem(case, gold).gradient()
            _, cgd = e.gradient()
            assert np.all(np.isfinite(cgd))
            assert np.abs(cgd - orc["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(orc["cat_grad"]).max()) * (10 if codon else 1)


@pytest.mark.parametrize("S,T,P,C", [(20, 9, 1, 1), (20, 17, 130, 4), (20, 30, 515, 2), (61, 7, 33, 1), (61, 12, 200, 2), (60, 8, 70, 3)])
def test_generic_states_against_oracle(S, T, P, C):
    """ragged tile counts (P not a multiple of 16 or of the workgroup's patterns), gaps, every built state count;
    default gradient (pi applied in the final state sum) including all lower and upper partials"""
    pb = random_problem(T, P, C, seed=300 + S + T, S=S, gaps=0.05)
    _compare_with_oracle(pb, RESCALE_NEVER, check_partials=True)
    pb = random_problem(T, P, C, seed=300 + S + T, S=S, gaps=0.05, fold_root_freqs=1)
    _compare_with_oracle(pb, RESCALE_NEVER, flags=GRAD_FOLD_ROOT_FREQS)


@pytest.mark.parametrize("S,T,P,C", [(20, 60, 130, 4), (20, 90, 33, 1), (61, 40, 70, 2), (60, 30, 17, 3)])
def test_generic_states_forced_rescaling(S, T, P, C):
    """R1 for the MFMA kernels: per-category workgroups publish per-pattern maxima, a level kernel rescales; lower
    partials, scale factors (through the per-pattern lnL) and the three gradient modes against the oracle."""
    pb = random_problem(T, P, C, seed=500 + S + T, S=S, gaps=0.05, bl=(0.3, 0.9), rescale=1)
    _compare_with_oracle(pb, RESCALE_ALWAYS, check_partials=True)
    pb.compat_scaled_gradient = 1
    _compare_with_oracle(pb, RESCALE_ALWAYS, flags=GRAD_COMPAT_SCALED)
    pb.compat_scaled_gradient = 0
    pb.fold_root_freqs = 1
    _compare_with_oracle(pb, RESCALE_ALWAYS, flags=GRAD_FOLD_ROOT_FREQS)


@pytest.mark.parametrize("T,P,C,shape,rescale", [(40, 130, 4, "random", RESCALE_NEVER), (300, 70, 4, "random", RESCALE_NEVER), (120, 65, 2, "caterpillar", RESCALE_NEVER),
                                                 (1000, 64, 4, "random", RESCALE_NEVER), (200, 90, 4, "random", RESCALE_ALWAYS)])
def test_nucleotide_ambiguity_codes_in_the_streamed_walk(T, P, C, shape, rescale):
    """IUPAC-style partial ambiguity masks (R, Y, ... : two or three of the four states) at 4 states: the streamed pre-order walk
    keeps five rows per tip and forms such a tip's rows as the sum of its states' rows (k_upper4_stream<.., AMBIG>,
    SX::tip_rows).  One tip cell in eight carries one of the ten partial masks, some are unknown; lnL, the stored partials
    and the gradient against the oracle on the same 0/1 tip vectors, for the default engine and for the table-gather walk."""
    forced = rescale == RESCALE_ALWAYS
    pb = random_problem(T, P, C, seed=4100 + T, shape=shape, gaps=0.03, bl=(0.2, 0.6) if forced else (0.01, 0.1), rescale=1 if forced else 0)
    rng = np.random.default_rng(T * 7 + P)
    masks = [m for m in range(1, 15) if bin(m).count("1") in (2, 3)]
    assert len(masks) == 10
    tp = np.zeros((T, P, 4))
    for t in range(T):
        for k in range(P):
            code = pb.tip_states[t, k]
            if code >= 4:
                tp[t, k, :] = 1.0
            elif rng.random() < 0.125:
                m = masks[rng.integers(10)] | (1 << int(code))  # keep the observed state possible
                tp[t, k, :] = [(m >> i) & 1 for i in range(4)]
            else:
                tp[t, k, code] = 1.0
    pb2 = po.Problem(pb.left, pb.right, pb.root, pb.weights, pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props,
                     pb.branch_lengths, tip_states=pb.tip_states, tip_partials=tp, rescale=1 if forced else 0)
    _compare_with_oracle(pb2, rescale, tip_mode="partials", check_partials=not forced)


@pytest.mark.parametrize("S,T,P,C,rescale", [(20, 14, 150, 2, RESCALE_NEVER), (61, 9, 40, 1, RESCALE_NEVER), (20, 50, 70, 2, RESCALE_ALWAYS)])
def test_generic_states_ambiguity_sets(S, T, P, C, rescale):
    """Tip partials that are sets of states (named ambiguities of a general data type, datatype.c:212-262) on the MFMA
    engines: the tip's message is the sum of the set's columns.  A fifth of the tip cells carry one of a handful of random
    sets, some are unknown, the rest one state; lnL, all partials and the gradient against the oracle run on the same
    0/1 tip vectors, and the tip partial reads back as it was given."""
    forced = rescale == RESCALE_ALWAYS
    pb = random_problem(T, P, C, seed=900 + S + T, S=S, gaps=0.03, bl=(0.3, 0.9) if forced else (0.01, 0.1), rescale=1 if forced else 0)
    rng = np.random.default_rng(S * T)
    sets = [rng.random(S) < rng.uniform(0.1, 0.6) for _ in range(7)]
    sets = [s for s in sets if 1 < s.sum() < S]
    tp = np.zeros((T, P, S))
    for t in range(T):
        for k in range(P):
            code = pb.tip_states[t, k]
            if code >= S:
                tp[t, k, :] = 1.0
            elif rng.random() < 0.2:
                tp[t, k, sets[rng.integers(len(sets))]] = 1.0
                tp[t, k, code] = 1.0  # keep the observed state inside the set (several distinct sets arise this way)
            else:
                tp[t, k, code] = 1.0
    pb2 = po.Problem(pb.left, pb.right, pb.root, pb.weights, pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props,
                     pb.branch_lengths, tip_states=pb.tip_states, tip_partials=tp, rescale=1 if forced else 0)
    _compare_with_oracle(pb2, rescale, tip_mode="partials", check_partials=True)
    with engine_from_problem(pb2, rescale=rescale, tip_mode="partials") as e:
        for t in (0, T - 1):
            got = e.partials(t)
            for c in range(C):
                np.testing.assert_array_equal(got[c], tp[t])


def test_generic_states_lazy_rescaling_switch():
    pb = random_problem(400, 20, 2, seed=21, S=20, bl=(0.5, 1.5), rescale=2)
    ref = pb.gradient()
    assert ref["rescaled"] and np.isfinite(ref["lnl"])
    with engine_from_problem(pb, rescale=RESCALE_AUTO) as e:
        assert not e.rescaling
        lnl, cg = e.gradient()
        assert e.rescaling
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
        assert e.log_likelihood() == lnl


@pytest.mark.parametrize("S,T,P,C", [(20, 200, 50_000, 4), (61, 100, 20_000, 1)])
def test_generic_full_size_properties(S, T, P, C):
    """BASELINE configs[2] (WAG-like 20 states) and configs[3] (codon, 61 states) shapes: determinism, weight
    linearity, sum over patterns, gradient vs central differences."""
    from physher_amd import synth
    rng = np.random.default_rng(S)
    tree = synth.random_tree(T, rng)
    states = synth.evolve(tree, P, S, rng)
    weights = rng.integers(1, 4, size=P).astype(np.float64)
    model = random_problem(4, 8, C, seed=S, S=S)
    with Engine(T, P, S, C, rescale=RESCALE_AUTO) as e:
        e.set_topology(tree.left, tree.right, tree.root)
        e.set_branch_lengths(tree.length)
        e.set_eigen(model.eval, model.evec, model.ivec)
        e.set_frequencies(model.freqs)
        e.set_category_rates(model.cat_rates, model.cat_props)
        e.set_pattern_weights(weights)
        for t in range(T):
            e.set_tip_states(t, states[t])
        lnl, cg = e.gradient()
        lnl2, cg2 = e.gradient()
        assert np.isfinite(lnl) and lnl == lnl2 and np.array_equal(cg, cg2)
        assert abs(np.dot(e.pattern_log_likelihoods(), weights) - lnl) <= 1e-11 * abs(lnl)
        e.set_pattern_weights(2.0 * weights)
        lnl3, cg3 = e.gradient()
        assert abs(lnl3 - 2 * lnl) <= 1e-12 * abs(lnl) and np.abs(cg3 - 2 * cg).max() <= 1e-11 * np.abs(cg).max()
        e.set_pattern_weights(weights)
        bg = po.branch_gradient_from_cat(cg, model.cat_rates, model.cat_props)
        for n in rng.choice([i for i in range(2 * T - 1) if i != tree.root], size=2, replace=False):
            h = 1e-6
            bl = tree.length.copy(); bl[n] += h
            e.set_branch_lengths(bl); up = e.log_likelihood()
            bl[n] -= 2 * h
            e.set_branch_lengths(bl); dn = e.log_likelihood()
            assert abs((up - dn) / (2 * h) - bg[n]) <= 1e-4 * max(1.0, abs(bg[n]))


# ---------------------------------------------------------------------------------------------------------
# substitution-model gradient: phyamd_parameter_gradient + phyamd_root_frequency_term
# ---------------------------------------------------------------------------------------------------------
SUBST_CASES = [c for c in CASES4 if read_spec(c)["model"] in ("gtr", "hky")]


def _host_dq(case, gold):
    from physher_amd import _phycpp_amd as pc
    spec = read_spec(case)
    f = list(gold["frequencies"])
    m = pc.HKYInterface(float(spec["rates"]), f) if spec["model"] == "hky" else pc.GTRInterface([float(x) for x in spec["rates"].split(",")], f)
    return m.rate_matrix_derivatives()


@pytest.mark.parametrize("case", SUBST_CASES)
def test_golden_parameter_gradient(case):
    """Engine-level: all parameters in one pre-order pass, fused and unfused schedules, rescaled cases included,
    against the reference's gradient_all tail and the oracle restatement."""
    gold = load(case)
    dQ = _host_dq(case, gold)
    n_rates = len(dQ) - 4
    ref = gold["gradient_all"][-len(dQ):]
    pb = oracle_problem(case, gold)
    _, og = po.parameter_gradient(pb, dQ)
    orf = po.root_frequency_term(pb)
    with engine_from_problem(pb, rescale=_rescale(case), tip_mode=_tip_mode(case)) as e:
        e.set_rate_matrix_derivatives(dQ)
        lnl, cg, pg = e.parameter_gradient()
        rf = e.root_frequency_term()
        assert e.rescaling == gold["rescaled"]
        assert abs(lnl - gold["lnl"]) <= 1e-10 * abs(gold["lnl"])
        np.testing.assert_allclose(rf, orf, rtol=1e-11)
        np.testing.assert_allclose(pg, og, rtol=1e-9, atol=1e-9)
        full = pg.copy()
        full[n_rates:] += rf
        np.testing.assert_allclose(full, ref, rtol=2e-8, atol=1e-7)
        # the branch gradient that comes with it is the one phyamd_gradient gives
        lnl2, cg2 = e.gradient()  # (the tree-walk kernels: another summation order)
        assert lnl2 == lnl and np.abs(cg - cg2).max() <= 1e-11 * max(1.0, np.abs(cg).max())
        # unfused schedule (every node stored) gives the same sums up to rounding
        e.set_keep_partials(True)
        _, _, pg_u = e.parameter_gradient()
        np.testing.assert_allclose(pg_u, pg, rtol=1e-11, atol=1e-10)
        # a subset of the parameters, and bitwise reproducibility
        e.set_rate_matrix_derivatives(dQ[:1])
        _, _, p1 = e.parameter_gradient()
        _, _, p1b = e.parameter_gradient()
        assert p1.shape == (1,) and p1[0] == p1b[0] and abs(p1[0] - pg_u[0]) <= 1e-11 * max(1.0, abs(pg_u[0]))


@pytest.mark.parametrize("T,P,C,shape,rescale", [(33, 1500, 4, "random", 0), (64, 700, 1, "caterpillar", 0), (40, 333, 8, "balanced", 0),
                                                 (150, 400, 4, "random", 1), (21, 200, 16, "random", 0), (12, 90, 2, "random", 1)])
def test_parameter_gradient_random_problems(T, P, C, shape, rescale):
    """Seeded synthetic problems (ragged pattern counts, 1..16 categories, gaps) against the oracle; 9 parameters with 16
    categories need more LDS accumulators than one workgroup holds, which exercises the chunked passes."""
    pb = random_problem(T, P, C, seed=T * 1000 + P, shape=shape, gaps=0.05, rescale=rescale)
    rng = np.random.default_rng(7)
    dQ = rng.normal(size=(9, 4, 4))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(4)[None]  # rows sum to zero like any dQ/dtheta
    _, og = po.parameter_gradient(pb, dQ)
    orf = po.root_frequency_term(pb)
    with engine_from_problem(pb, rescale=rescale) as e:
        e.set_rate_matrix_derivatives(dQ)
        lnl, cg, pg = e.parameter_gradient()
        assert e.rescaling == bool(rescale)
        scale = max(1.0, np.abs(og).max())
        assert np.abs(pg - og).max() <= 1e-9 * scale
        np.testing.assert_allclose(e.root_frequency_term(), orf, rtol=1e-10)
        ref = pb.gradient()
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())


@pytest.mark.parametrize("S,count", [(4, 150), (20, 100)])
def test_parameter_gradient_many_parameters(S, count):
    """more parameters than states squared (a 12-location symmetric trait model has 66 rates + 12 frequencies): the eigen-basis
    sums do not depend on the count, only the final contraction does"""
    pb = random_problem(10, 40, 2, seed=S + count, S=S, gaps=0.05)
    rng = np.random.default_rng(count)
    dQ = rng.normal(size=(count, S, S))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(S)[None]
    _, og = po.parameter_gradient(pb, dQ)
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        e.set_rate_matrix_derivatives(dQ)
        _, _, pg = e.parameter_gradient()
        assert np.abs(pg - og).max() <= 1e-9 * max(1.0, np.abs(og).max())


def test_parameter_gradient_argument_checks():
    pb = random_problem(8, 100, 2, seed=3)
    with engine_from_problem(pb) as e:
        with pytest.raises(EngineError):
            e.parameter_gradient()  # no dQ set
        e.set_rate_matrix_derivatives(np.zeros((2, 4, 4)))
        with pytest.raises(EngineError):
            e.parameter_gradient(GRAD_FOLD_ROOT_FREQS)
        lnl, cg, pg = e.parameter_gradient()
        assert np.all(pg == 0.0)
        with pytest.raises(EngineError):
            e.set_rate_matrix_derivatives(np.zeros((2049, 4, 4)))


@pytest.mark.parametrize("S,T,P,C,rescale", [(20, 11, 1, 1, 0), (20, 17, 90, 3, 0), (61, 8, 37, 2, 0), (60, 6, 5, 1, 0), (20, 70, 40, 2, 1), (61, 40, 9, 1, 1)])
def test_parameter_gradient_generic_states(S, T, P, C, rescale):
    """G2 for 20 / 60 / 61 states (general K-state rate matrices of discrete-trait models, gensubst.c:284-323): the engine
    keeps every partial and forms the per-branch outer products in the eigen basis; against the oracle, with gaps, unscaled
    and rescaled, and together with the per-category branch gradient of the same call."""
    pb = random_problem(T, P, C, seed=S * 100 + T, S=S, gaps=0.05, bl=(0.3, 0.9) if rescale else (0.01, 0.2), rescale=rescale)
    rng = np.random.default_rng(S + T)
    dQ = rng.normal(size=(5, S, S))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(S)[None]
    _, og = po.parameter_gradient(pb, dQ)
    orf = po.root_frequency_term(pb)
    ref = pb.gradient()
    with engine_from_problem(pb, rescale=RESCALE_ALWAYS if rescale else RESCALE_NEVER) as e:
        e.set_rate_matrix_derivatives(dQ)
        lnl, cg, pg = e.parameter_gradient()
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert e.rescaling == bool(rescale)
        assert np.abs(pg - og).max() <= 1e-9 * max(1.0, np.abs(og).max())
        np.testing.assert_allclose(e.root_frequency_term(), orf, rtol=1e-10)
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
        lnl2, cg2 = e.gradient()  # the plain gradient still runs on the same engine, now in keep-partials mode
        assert lnl2 == lnl and np.array_equal(cg2, cg)
        lnl3, _, pg3 = e.parameter_gradient()
        assert lnl3 == lnl and np.array_equal(pg3, pg)  # fixed-order sums: reproducible


# ---------------------------------------------------------------------------------------------------------
# D1: incremental (dirty-node) post-order updates
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,T,P,C,rescale", [(4, 60, 700, 4, RESCALE_NEVER), (4, 60, 300, 2, RESCALE_ALWAYS), (20, 25, 90, 2, RESCALE_NEVER),
                                             (61, 12, 40, 1, RESCALE_ALWAYS)])
def test_single_branch_updates_recompute_only_the_path_to_the_root(S, T, P, C, rescale):
    """phyamd_set_branch_length marks one node like update_nodes[] does (treelikelihood.c:73-114): the next evaluation
    recomputes only the stored nodes above it.  Results equal a full recomputation; the launch count shows the saving."""
    pb = random_problem(T, P, C, seed=70 + S + T, S=S, gaps=0.03, rescale=1 if rescale == RESCALE_ALWAYS else 0,
                        bl=(0.3, 0.9) if rescale == RESCALE_ALWAYS else (0.01, 0.1))
    rng = np.random.default_rng(3)
    with engine_from_problem(pb, rescale=rescale) as e, engine_from_problem(pb, rescale=rescale) as full:
        e.set_profiling(True)
        e.log_likelihood()
        full_launches = None
        bl = pb.branch_lengths.copy()
        for step in range(6):
            nodes = rng.choice([n for n in range(pb.N) if n != pb.root], size=1 + step % 2, replace=False)
            for n in nodes:
                bl[n] *= rng.uniform(0.5, 1.5)
                e.set_branch_length(int(n), bl[n])
            lnl = e.log_likelihood()
            launches = e.profile()["lower_launches"]
            full.set_branch_lengths(bl)
            ref = full.log_likelihood()
            if full_launches is None:
                full.set_profiling(True)
                full.set_branch_lengths(bl)
                full.log_likelihood()
                full_launches = full.profile()["lower_launches"]
            assert abs(lnl - ref) <= 1e-12 * abs(ref), (step, lnl, ref)
            np.testing.assert_allclose(e.pattern_log_likelihoods(), full.pattern_log_likelihoods(), rtol=1e-12, atol=1e-12)
            assert e.log_likelihood() == lnl and e.profile()["lower_launches"] == 0  # nothing changed: cached
            # the gradient after an incremental update is the gradient of the current branch lengths
            if step in (1, 4):
                l1, cg = e.gradient()
                l2, cg_ref = full.gradient()
                assert abs(l1 - l2) <= 1e-12 * abs(l2)
                assert np.abs(cg - cg_ref).max() <= 1e-10 * max(1.0, np.abs(cg_ref).max())
        # against the oracle at the final branch lengths
        pb.branch_lengths[:] = bl
        orc = pb.log_likelihood()
        assert abs(e.log_likelihood() - orc["lnl"]) <= 1e-10 * abs(orc["lnl"])
        # any other setter, or update_all_nodes, falls back to the full pass
        e.update_all_nodes()
        assert abs(e.log_likelihood() - orc["lnl"]) <= 1e-10 * abs(orc["lnl"])
        assert e.profile()["lower_launches"] >= 1


# ---------------------------------------------------------------------------------------------------------
# f.2: the optimiser's fast path -- lnL and two derivatives along ONE branch from the resident partials
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,T,P,C,fold,rescale", [(4, 20, 333, 4, 0, 0), (4, 9, 64, 1, 0, 0), (4, 14, 100, 3, 1, 0), (4, 60, 150, 4, 0, 1), (4, 30, 70, 1, 0, 1),
                                                  (20, 12, 90, 2, 0, 0), (61, 8, 33, 1, 0, 0), (20, 50, 40, 2, 0, 1), (60, 7, 300, 1, 1, 0)])
def test_single_branch_evaluation(S, T, P, C, fold, rescale):
    """phyamd_branch_log_likelihood against the CPU oracle evaluated at the trial length: lnL(t) and d1 for trial lengths of tip
    and internal branches; d2 against central differences of d1 (and against the reference's own d2lnldt2_uppper values in
    test_single_branch_evaluation_matches_reference_fixture below)
    (_calculate_uppper / dlnldt_uppper / d2lnldt2_uppper, treelikelihood.c:2196-2335, 2592-2686).  Every state count;
    rescaled evaluations anchor the stored (scaled) partials on the per-pattern lnL they belong to."""
    pb = random_problem(T, P, C, seed=60 + T + S, S=S, gaps=0.03, fold_root_freqs=fold, bl=(0.3, 0.9) if rescale else (0.01, 0.1), rescale=rescale)
    flags = GRAD_FOLD_ROOT_FREQS if fold else 0
    mode = RESCALE_ALWAYS if rescale else RESCALE_NEVER
    oracle0 = pb.gradient()
    with engine_from_problem(pb, rescale=mode) as e:
        e.set_keep_partials(True)
        lnl0, cg = e.gradient(flags)
        if not fold:
            assert abs(lnl0 - oracle0["lnl"]) <= 1e-10 * abs(oracle0["lnl"])
        bg = po.branch_gradient_from_cat(cg, pb.cat_rates, pb.cat_props)
        rng = np.random.default_rng(1)
        nodes = [0, T - 1] + list(rng.choice([n for n in range(T, pb.N) if n != pb.root], size=3, replace=False))
        for n in nodes:
            t0 = pb.branch_lengths[n]
            l, d1, d2 = e.branch_log_likelihood(n, t0)
            if not fold:  # (the folded form is the reference's inexact arithmetic for non-uniform pi: only self-consistency below)
                assert abs(l - lnl0) <= 1e-11 * abs(lnl0)
                assert abs(d1 - bg[n]) <= 1e-9 * max(1.0, abs(bg).max())
            for t in (0.5 * t0, 1.7 * t0, 0.3):
                lt, d1t, d2t = e.branch_log_likelihood(n, t)
                if not fold:  # the CPU oracle evaluated at the trial length: lnL(t) and d lnL / dt
                    pb.branch_lengths[n] = t
                    o = pb.gradient()
                    pb.branch_lengths[n] = t0
                    assert abs(lt - o["lnl"]) <= 1e-10 * abs(lt), (n, t)
                    o_d1 = po.branch_gradient_from_cat(o["cat_grad"], pb.cat_rates, pb.cat_props)[n]
                    assert abs(d1t - o_d1) <= 1e-9 * max(1.0, np.abs(o["cat_grad"]).max()), (n, t)
                h = 1e-5 * max(t, 1e-3)
                lp, d1p, _ = e.branch_log_likelihood(n, t + h)
                lm, d1m, _ = e.branch_log_likelihood(n, t - h)
                assert abs((lp - lm) / (2 * h) - d1t) <= 1e-5 * max(1.0, abs(d1t))
                assert abs((d1p - d1m) / (2 * h) - d2t) <= 1e-5 * max(1.0, abs(d2t))
        with pytest.raises(EngineError):
            e.branch_log_likelihood(pb.root, 0.1)


@pytest.mark.parametrize("S,T,P,C,rescale", [(4, 60, 333, 4, 0), (4, 33, 100, 1, 0), (4, 70, 150, 4, 1), (20, 14, 90, 2, 0), (61, 9, 33, 1, 0), (20, 50, 40, 2, 1)])
def test_single_branch_evaluation_without_resident_uppers(S, T, P, C, rescale):
    """The optimiser's loop as the reference runs it (optimizer.c:116-150): for branch after branch, trial lengths, then the
    accepted length is set and the next branch follows.  No keep-partials gradient in between: the one upper partial a
    branch needs is rebuilt by a walk down its path from the root (fringe, DEEP and stored siblings alike), pending single-branch
    changes recompute only their paths to the root.  Every value against the CPU oracle at the same lengths."""
    pb = random_problem(T, P, C, seed=90 + T + S, S=S, gaps=0.03, bl=(0.3, 0.9) if rescale else (0.01, 0.1), rescale=rescale)
    mode = RESCALE_ALWAYS if rescale else RESCALE_NEVER
    rng = np.random.default_rng(T)
    with engine_from_problem(pb, rescale=mode) as e:
        bl = pb.branch_lengths.copy()
        order = [n for n in range(pb.N) if n != pb.root]
        rng.shuffle(order)
        for n in order[:12]:  # tips, cherries' tips, fringe and DEEP nodes, stored nodes: whatever the shuffle brings
            for t in (0.6 * bl[n], 1.5 * bl[n] + 0.01):
                lt, d1, d2 = e.branch_log_likelihood(n, t)
                pb.branch_lengths[:] = bl
                pb.branch_lengths[n] = t
                o = pb.gradient()  # the CPU oracle at the trial length
                ref_lnl, ref_cg = o["lnl"], o["cat_grad"]
                assert abs(lt - ref_lnl) <= 1e-10 * abs(ref_lnl), (n, t)
                ref_bg = po.branch_gradient_from_cat(ref_cg, pb.cat_rates, pb.cat_props)
                assert abs(d1 - ref_bg[n]) <= 1e-9 * max(1.0, np.abs(ref_bg).max()), (n, t)
            bl[n] = 1.5 * bl[n] + 0.01  # accept the last trial: the next branch sees it
            e.set_branch_length(n, bl[n])
        assert abs(e.log_likelihood() - ref_lnl) <= 1e-10 * abs(ref_lnl)
        e.set_profiling(True)
        n = order[20 % len(order)]
        e.branch_log_likelihood(n, bl[n])
        e.branch_log_likelihood(n, 1.1 * bl[n])  # same branch again: the rebuilt upper is reused, nothing is recomputed
        assert e.profile()["lower_launches"] == 0


@pytest.mark.parametrize("S", [4, 20, 61])
def test_tiny_trees_and_ragged_pattern_counts(S):
    """Two to nine taxa, 1 / 2 / 15 / 16 / 17 / 65 patterns (below, at and just past a wave's 16-pattern MFMA tile and 64-lane
    wave), one and four categories, plain and rescaled: lnL, gradient and a tip's single-branch evaluation against the oracle."""
    for T in (2, 3, 4, 5, 9):
        for P in (1, 2, 15, 16, 17, 65):
            for C in (1, 4):
                for resc in (0, 1):
                    pb = random_problem(T, P, C, seed=7 * T + P + S, S=S, gaps=0.1, rescale=resc)
                    o = pb.gradient()
                    with engine_from_problem(pb, rescale=RESCALE_ALWAYS if resc else RESCALE_NEVER) as e:
                        lnl, cg = e.gradient()
                        assert abs(lnl - o["lnl"]) <= 1e-10 * abs(o["lnl"]), (T, P, C, resc)
                        np.testing.assert_allclose(cg, o["cat_grad"], rtol=1e-8, atol=1e-9 * max(1e-300, np.abs(o["cat_grad"]).max()))
                        if T > 2:
                            lt, _, _ = e.branch_log_likelihood(0, pb.branch_lengths[0])
                            assert abs(lt - o["lnl"]) <= 1e-10 * abs(o["lnl"]), (T, P, C, resc)


def _caterpillar(T):
    N = 2 * T - 1
    left, right = np.full(N, -1, np.int32), np.full(N, -1, np.int32)
    left[T], right[T] = 0, 1
    for i in range(1, T - 1):
        left[T + i], right[T + i] = T + i - 1, i + 1
    return left, right, N - 1


def _balanced(T):
    N = 2 * T - 1
    left, right = np.full(N, -1, np.int32), np.full(N, -1, np.int32)
    level, nxt = list(range(T)), T
    while len(level) > 1:
        new = []
        for a in range(0, len(level) - 1, 2):
            left[nxt], right[nxt] = level[a], level[a + 1]
            new.append(nxt)
            nxt += 1
        if len(level) % 2:
            new.append(level[-1])
        level = new
    return left, right, N - 1


@pytest.mark.parametrize("shape", [_caterpillar, _balanced])
@pytest.mark.parametrize("T,rescale", [(8, 0), (65, 1), (256, 0), (300, 1)])
def test_extreme_tree_shapes(shape, T, rescale):
    """A ladder (every op has one stored child: the chunked walks find nothing to cut, nothing is ever parked) and a balanced tree
    (cuts at every level, deepest park nesting), plain and rescaled, against the oracle."""
    pb = random_problem(T, 130, 4, seed=T + 3, S=4, gaps=0.05, bl=(0.3, 0.9) if rescale else (0.01, 0.1), rescale=rescale)
    left, right, root = shape(T)
    pb.left[:] = left
    pb.right[:] = right
    pb.root = root
    o = pb.gradient()
    with engine_from_problem(pb, rescale=RESCALE_ALWAYS if rescale else RESCALE_NEVER) as e:
        lnl, cg = e.gradient()
        assert abs(lnl - o["lnl"]) <= 1e-10 * abs(o["lnl"])
        np.testing.assert_allclose(cg, o["cat_grad"], rtol=1e-8, atol=1e-9 * np.abs(o["cat_grad"]).max())


@pytest.mark.parametrize("shape", ["random", "caterpillar", "balanced"])
@pytest.mark.parametrize("T", [1000, 2000])
def test_default_walks_at_headline_taxa_match_oracle(T, shape):
    """The DEFAULT unscaled schedule at the headline tree size and twice that (the chunked post-order walk and the streamed, chunked
    pre-order walk: six cut subtrees, dozens of HBM park slots, LDS parks, table blocks) against the CPU oracle: lnL, per-pattern
    lnL, the full per-category gradient and the lower partial of every node the walk stores.  A pattern count that is not a
    multiple of the wave, gaps, and branches short enough that 2000 taxa stay above the underflow that switches rescaling on."""
    bl = (0.004, 0.04) if T == 1000 else (0.002, 0.02)
    pb = random_problem(T, 200, 4, seed=7000 + T, shape=shape, gaps=0.03, bl=bl)  # (the sites are evolved on the tree they are evaluated on)
    o = pb.gradient(want_partials=True)
    assert not o["rescaled"] and np.isfinite(o["lnl"])
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        lnl, cg = e.gradient()
        assert not e.rescaling
        assert abs(lnl - o["lnl"]) <= 1e-10 * abs(o["lnl"])
        np.testing.assert_allclose(e.pattern_log_likelihoods(), o["pattern_lk"], rtol=1e-11, atol=1e-11)
        assert np.abs(cg - o["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(o["cat_grad"]).max())
        stored = 0
        for n in range(pb.T, pb.N):
            try:
                got = e.partials(n)
            except Exception:  # fused into its parent (cherry, cherry + tip, DEEP): never stored by this schedule
                continue
            stored += 1
            np.testing.assert_allclose(got, o["lower"][n], rtol=1e-9, atol=1e-300)
        assert stored >= T // 5  # about a third of the internal nodes of a random tree, a quarter of a balanced one, every second one of a ladder
        p = e.profile()
        assert p["lower_launches"] <= 2 and p["upper_launches"] <= 2  # the walks ran (a level schedule takes one launch per tree level)


def test_fused_cherries_at_20_states(monkeypatch):
    """20 states: cherries are fused into their parents' ops (never stored; their uppers stay in registers).  The fused schedule
    and the one that stores every node (PHYAMD_GEN_FUSION=0) must both equal the CPU oracle -- lnL, per-pattern lnL, gradient --
    and the single-branch evaluation must work on a cherry, on its tips and beside it (the cherry's partial is then formed on the
    side); a one-branch change under a cherry recomputes through its first stored ancestor."""
    pb = random_problem(40, 150, 3, seed=321, S=20, gaps=0.05)
    o = pb.gradient()
    cherries = [n for n in range(pb.T, pb.N) if pb.left[n] < pb.T and pb.right[n] < pb.T and n != pb.root]
    assert len(cherries) >= 5
    parent = np.full(pb.N, -1)
    for n in range(pb.T, pb.N):
        parent[pb.left[n]] = parent[pb.right[n]] = n
    results = {}
    for fusion in ("1", "0"):
        monkeypatch.setenv("PHYAMD_GEN_FUSION", fusion)
        with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
            lnl, cg = e.gradient()
            assert abs(lnl - o["lnl"]) <= 1e-10 * abs(o["lnl"])
            np.testing.assert_allclose(cg, o["cat_grad"], rtol=1e-9, atol=1e-9 * np.abs(o["cat_grad"]).max())
            np.testing.assert_allclose(e.pattern_log_likelihoods(), o["pattern_lk"], rtol=1e-10, atol=1e-10)
            results[fusion] = (lnl, cg.copy())
            ref_bg = po.branch_gradient_from_cat(o["cat_grad"], pb.cat_rates, pb.cat_props)
            c0 = cherries[0]
            sib = pb.right[parent[c0]] if pb.left[parent[c0]] == c0 else pb.left[parent[c0]]
            for n in (c0, pb.left[c0], pb.right[c0], sib):
                lt, d1, _ = e.branch_log_likelihood(n, pb.branch_lengths[n])
                assert abs(lt - o["lnl"]) <= 1e-10 * abs(o["lnl"]), (fusion, n)
                assert abs(d1 - ref_bg[n]) <= 1e-9 * max(1.0, np.abs(ref_bg).max()), (fusion, n)
            # one changed tip branch under a cherry
            t = pb.left[cherries[1]]
            bl = pb.branch_lengths.copy()
            bl[t] *= 1.7
            e.set_branch_length(t, bl[t])
            saved = pb.branch_lengths.copy()
            pb.branch_lengths[:] = bl
            o2 = pb.gradient()
            pb.branch_lengths[:] = saved
            lnl2, cg2 = e.gradient()
            assert abs(lnl2 - o2["lnl"]) <= 1e-10 * abs(o2["lnl"])
            np.testing.assert_allclose(cg2, o2["cat_grad"], rtol=1e-9, atol=1e-9 * np.abs(o2["cat_grad"]).max())
    assert abs(results["1"][0] - results["0"][0]) <= 1e-12 * abs(results["0"][0])
    np.testing.assert_allclose(results["1"][1], results["0"][1], rtol=1e-11, atol=1e-11 * np.abs(results["0"][1]).max())


def test_tip_rate_products_follow_the_model_at_20_states():
    """20 states: the branch term of a tip child is a column of the image of Qf P(t) (k_tip_rate_products; Qf = diag(pi) Q, or Q
    when pi is folded into the uppers), built once per state of the model.  One engine asked for both conventions in turn, then
    with other branch lengths, other frequencies (a new eigen system with them) and other category rates must follow every time:
    a stale image shows in the gradient of the tip branches only (lnL does not use it)."""
    from golden_util import reversible_eigen
    pb = random_problem(26, 210, 2, seed=911, S=20, gaps=0.04)
    tips = np.arange(pb.T)

    def check(e, fold):
        pb.fold_root_freqs = 1 if fold else 0
        o = pb.gradient()
        lnl, cg = e.gradient(GRAD_FOLD_ROOT_FREQS if fold else 0)
        assert abs(lnl - o["lnl"]) <= 1e-10 * abs(o["lnl"])
        scale = np.abs(o["cat_grad"]).max()
        np.testing.assert_allclose(cg, o["cat_grad"], rtol=1e-9, atol=1e-9 * scale)
        assert np.abs(o["cat_grad"][tips]).max() > 1e-3 * scale  # the tip branches carry weight in this check

    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        check(e, False)
        check(e, True)   # Qf changes from diag(pi) Q to Q
        check(e, False)  # and back
        pb.branch_lengths[:] = pb.branch_lengths * np.random.default_rng(5).uniform(0.5, 2.0, size=pb.N)
        e.set_branch_lengths(pb.branch_lengths)
        check(e, False)
        rng = np.random.default_rng(6)
        pb.freqs = rng.dirichlet(np.full(20, 4.0))
        r = rng.uniform(0.5, 3.0, size=(20, 20))
        pb.eval, pb.evec, pb.ivec = (np.ascontiguousarray(a, dtype=np.float64) for a in reversible_eigen(0.5 * (r + r.T), pb.freqs))
        e.set_eigen(pb.eval, pb.evec, pb.ivec)
        e.set_frequencies(pb.freqs)
        check(e, True)
        check(e, False)
        pb.cat_rates = np.array([0.3, 1.7])
        e.set_category_rates(pb.cat_rates, pb.cat_props)
        check(e, False)


@pytest.mark.parametrize("case", ["gtr_g4_t16", "gtr_g4_t24_gaps_tipstates", "wag_g4_t12", "mg94_t8"])
@pytest.mark.parametrize("resident", [False, True])
def test_single_branch_evaluation_matches_reference_fixture(case, resident):
    """lnL(t), d lnL/dt and d2 lnL/dt2 of single branches at trial lengths against values the compiled reference produced with
    its own upper-partial protocol (full evaluation at the trial length, update_upper_partials, calculate_dldt_uppper,
    d2lnldt2_uppper: treelikelihood.c:469-530, 2196-2335; fixtures tests/golden/<case>/branch_trials.json, written by
    make_golden.py through `ref_driver branch`): 4, 20 and 61 states, tip partials and tip states, a tip, two inner branches
    and a branch next to the root."""
    import json
    gold = load(case)
    spec = read_spec(case)
    with open(os.path.join(GOLDEN, case, "branch_trials.json")) as f:
        trials = json.load(f)["trials"]
    pb = oracle_problem(case, gold)
    with engine_from_problem(pb, tip_mode="states" if spec["tipstates"] == "1" else "partials") as e:
        if resident:
            e.set_keep_partials(True)
            e.gradient()
        for tr in trials:
            lnl, d1, d2 = e.branch_log_likelihood(tr["node"], tr["length"])
            assert abs(lnl - tr["lnl"]) <= 1e-10 * abs(tr["lnl"]), tr
            assert abs(d1 - tr["d1"]) <= 1e-8 * max(1.0, abs(tr["d1"])), (tr, d1)
            assert abs(d2 - tr["d2"]) <= 1e-7 * max(1.0, abs(tr["d2"])), (tr, d2)


@pytest.mark.parametrize("S,T,P,C", [(4, 30, 400, 4), (20, 10, 120, 2)])
def test_newton_branch_length_optimisation(S, T, P, C):
    """What the fast path is for: maximum-likelihood branch lengths by safeguarded Newton steps on one branch at a time
    (the reference does the same with Brent's method, optimizer.c:116-150), driven only by phyamd_branch_log_likelihood and
    phyamd_set_branch_length.  lnL never decreases, and after a few sweeps the branch gradient has shrunk a hundredfold."""
    pb = random_problem(T, P, C, seed=123 + S, S=S, gaps=0.02)
    with engine_from_problem(pb, rescale=RESCALE_AUTO) as e:
        bl = pb.branch_lengths * np.random.default_rng(3).uniform(0.3, 3.0, size=pb.N)  # start away from the generating lengths
        bl[pb.root] = 0.0
        e.set_branch_lengths(bl)
        lnl_start, cg = e.gradient()
        g_start = np.abs(po.branch_gradient_from_cat(cg, pb.cat_rates, pb.cat_props)).max()
        branches = [n for n in range(pb.N) if n != pb.root]
        lnl = lnl_start
        for sweep in range(6):
            for n in branches:
                t = bl[n]
                for _ in range(6):
                    l, d1, d2 = e.branch_log_likelihood(n, t)
                    step = -d1 / d2 if d2 < 0 else (0.5 * t if d1 > 0 else -0.5 * t)  # Newton where concave, else a cautious move uphill
                    t_new = min(max(t + step, 1e-8), 10.0)
                    l_new = e.branch_log_likelihood(n, t_new)[0]
                    while l_new < l - 1e-12 * abs(l) and abs(t_new - t) > 1e-12:  # backtrack
                        t_new = 0.5 * (t + t_new)
                        l_new = e.branch_log_likelihood(n, t_new)[0]
                    if abs(t_new - t) <= 1e-9 * max(t, 1e-6):
                        t = t_new
                        break
                    t = t_new
                l_acc = e.branch_log_likelihood(n, t)[0]
                assert l_acc >= lnl - 1e-10 * abs(lnl), (sweep, n, l_acc, lnl)
                lnl = l_acc
                bl[n] = t
                e.set_branch_length(n, t)
        lnl_end, cg = e.gradient()
        assert abs(lnl_end - lnl) <= 1e-10 * abs(lnl)  # the trial values were those of real evaluations
        live = [n for n in branches if bl[n] > 1e-7 and n != pb.right[pb.root]]  # (lengths pinned at the lower bound keep a negative slope)
        g_end = np.abs(po.branch_gradient_from_cat(cg, pb.cat_rates, pb.cat_props)[live]).max()
        assert lnl_end > lnl_start and g_end < 1e-2 * g_start, (lnl_start, lnl_end, g_start, g_end)  # (coordinate ascent: linear convergence)


# ---------------------------------------------------------------------------------------------------------
# pattern tiling under a device-memory cap (SURVEY 8d "memory feasibility")
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,T,P,C,rescale,shape", [(4, 40, 3000, 4, RESCALE_NEVER, "random"), (4, 30, 3000, 2, RESCALE_ALWAYS, "random"),
                                                   (4, 40, 3000, 4, RESCALE_NEVER, "caterpillar"), (20, 12, 1500, 2, RESCALE_NEVER, "random"),
                                                   (20, 12, 1500, 2, RESCALE_NEVER, "caterpillar"), (61, 8, 2000, 1, RESCALE_NEVER, "random")])
def test_pattern_tiling_under_a_memory_cap(S, T, P, C, rescale, shape):
    """max_device_bytes below the untiled working set: the engine walks the patterns in tiles through one set of partial
    arrays and adds the per-tile sums in tile order.  lnL, per-pattern lnL and the gradient equal the ORACLE's (ragged last tile
    included), the substitution-parameter sums the untiled engine's; the engine never holds more than the cap (the tile size
    follows the real tree at phyamd_set_topology: a ladder-like tree stores twice what a random one does); calls that need
    resident partials are refused."""
    forced = rescale == RESCALE_ALWAYS
    pb = random_problem(T, P, C, seed=4000 + S + T, S=S, shape=shape, gaps=0.04, bl=(0.3, 0.9) if forced else (0.01, 0.1), rescale=1 if forced else 0)
    ref = pb.gradient()
    rng = np.random.default_rng(S)
    dQ = rng.normal(size=(3, S, S))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(S)[None]
    with engine_from_problem(pb, rescale=rescale) as whole:
        l3, cg_ref = whole.gradient()
        untiled_bytes = whole.profile()["device_bytes"]
        cap = int(0.45 * untiled_bytes)
        with engine_from_problem(pb, rescale=rescale, max_device_bytes=cap) as e:
            assert e.profile()["tiles"] >= 2 and whole.profile()["tiles"] == 1
            lnl = e.log_likelihood()
            assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
            np.testing.assert_allclose(e.pattern_log_likelihoods(), ref["pattern_lk"], rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(e.pattern_log_likelihoods(), whole.pattern_log_likelihoods(), rtol=1e-13, atol=1e-13)
            l2, cg = e.gradient()
            assert abs(l2 - l3) <= 1e-12 * abs(l3)
            assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
            assert np.abs(cg - cg_ref).max() <= 1e-10 * max(1.0, np.abs(cg_ref).max())
            if C >= 2:  # the +I root term is summed over the tiles while each tile's root partial is resident
                assert abs(e.root_invariant_term() - whole.root_invariant_term()) <= 1e-10 * max(1.0, abs(whole.root_invariant_term()))
            e.set_rate_matrix_derivatives(dQ)
            whole.set_rate_matrix_derivatives(dQ)
            if S == 4:
                _, _, pg = e.parameter_gradient()
                _, _, pg_ref = whole.parameter_gradient()
                assert np.abs(pg - pg_ref).max() <= 1e-10 * max(1.0, np.abs(pg_ref).max())
                _, pg_oracle = po.parameter_gradient(pb, dQ)
                assert np.abs(pg - pg_oracle).max() <= 1e-8 * max(1.0, np.abs(pg_oracle).max())
                np.testing.assert_allclose(e.root_frequency_term(), po.root_frequency_term(pb), rtol=1e-9)
            e.set_branch_length(0, 0.2)  # a changed branch: every tile is recomputed
            pb.branch_lengths[0] = 0.2
            assert abs(e.log_likelihood() - pb.log_likelihood()["lnl"]) <= 1e-10 * abs(ref["lnl"])
            for call in (lambda: e.partials(T), lambda: e.set_keep_partials(True), lambda: e.store(), lambda: e.branch_log_likelihood(0, 0.1)):
                with pytest.raises(EngineError):
                    call()
            assert e.profile()["device_bytes"] <= cap, (e.profile(), cap)  # the guarantee of max_device_bytes
    with pytest.raises(EngineError):
        Engine(T, P, S, C, max_device_bytes=1000)  # below the smallest tile


def test_memory_cap_is_never_exceeded_by_later_requests():
    """20 states, tiled: a substitution-parameter gradient wants every node's lower and upper partial resident (about three
    times the tile's working set).  Under an explicit cap that is refused with PHYAMD_ENOMEM instead of silently overshooting."""
    S, T, P, C = 20, 12, 1500, 2
    pb = random_problem(T, P, C, seed=4100, S=S)
    with engine_from_problem(pb) as whole:
        whole.gradient()
        cap = int(0.45 * whole.profile()["device_bytes"])
    rng = np.random.default_rng(5)
    dQ = rng.normal(size=(2, S, S))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(S)[None]
    with engine_from_problem(pb, max_device_bytes=cap) as e:
        e.gradient()
        e.set_rate_matrix_derivatives(dQ)
        try:
            e.parameter_gradient()
        except EngineError as err:
            assert err.code == -3, err  # PHYAMD_ENOMEM
        assert e.profile()["device_bytes"] <= cap


# ---------------------------------------------------------------------------------------------------------
# f.4: MCMC store / restore
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,T,P,C,rescale,keep", [(4, 40, 500, 4, RESCALE_NEVER, False), (4, 25, 130, 2, RESCALE_NEVER, True),
                                                  (4, 70, 200, 4, RESCALE_ALWAYS, False), (20, 15, 60, 2, RESCALE_NEVER, False),
                                                  (61, 30, 20, 1, RESCALE_ALWAYS, False)])
def test_store_restore_mcmc_walk(S, T, P, C, rescale, keep):
    """A short Metropolis-style walk: proposals change one branch, a few branches, every branch, or the category rates; each
    is evaluated, then accepted (store) or rejected (restore).  Every evaluation equals the CPU ORACLE at the same
    parameters; after a restore the stored lnL and gradient come back and only the root is re-integrated
    (_singleTreeLikelihood_store / _treelikelihood_handle_restore, treelikelihood.c:116-161)."""
    forced = rescale == RESCALE_ALWAYS
    pb = random_problem(T, P, C, seed=7000 + S + T, S=S, gaps=0.03, bl=(0.3, 0.9) if forced else (0.01, 0.1), rescale=1 if forced else 0)
    rng = np.random.default_rng(S + T)
    with engine_from_problem(pb, rescale=rescale) as e:
        with pytest.raises(EngineError):
            e.restore()  # nothing stored yet
        if keep:
            e.set_keep_partials(True)
        e.set_profiling(True)
        bl, rates = pb.branch_lengths.copy(), pb.cat_rates.copy()
        lnl_cur, _ = e.gradient()
        e.store()
        non_root = [n for n in range(pb.N) if n != pb.root]

        def reference(bl_, rates_):  # the CPU oracle at these parameters (accepted and rejected states alike)
            o = po.Problem(pb.left, pb.right, pb.root, pb.weights, pb.eval, pb.evec, pb.ivec, pb.freqs, rates_, pb.cat_props, bl_,
                           tip_states=pb.tip_states, rescale=pb.rescale).gradient()
            return o["lnl"], o["cat_grad"]

        accepted = rejected = 0
        for step in range(14):
            kind = ["one", "few", "all", "rates"][step % 4]
            nbl, nrates = bl.copy(), rates.copy()
            if kind == "one":
                n = int(rng.choice(non_root))
                nbl[n] *= rng.uniform(0.5, 2.0)
                e.set_branch_length(n, nbl[n])
            elif kind == "few":
                for n in rng.choice(non_root, size=3, replace=False):
                    nbl[n] *= rng.uniform(0.5, 2.0)
                    e.set_branch_length(int(n), nbl[n])
            elif kind == "all":
                nbl = bl * rng.uniform(0.8, 1.25, size=pb.N)
                e.set_branch_lengths(nbl)
            else:
                nrates = rates * rng.uniform(0.8, 1.25, size=C)
                e.set_category_rates(nrates, pb.cat_props)
            lnl_new = e.log_likelihood()
            ref_lnl, ref_cg = reference(nbl, nrates)
            assert abs(lnl_new - ref_lnl) <= 1e-10 * abs(ref_lnl), (step, kind)
            if step % 3 == 0:  # accept
                _, cg = e.gradient()
                assert np.abs(cg - ref_cg).max() <= 1e-9 * max(1.0, np.abs(ref_cg).max())
                bl, rates, lnl_cur = nbl, nrates, lnl_new
                e.store()
                accepted += 1
            else:  # reject
                e.restore()
                back = e.log_likelihood()
                assert abs(back - lnl_cur) <= 1e-13 * abs(lnl_cur), (step, kind)
                assert e.profile()["lower_launches"] <= 1  # the root alone
                ref_lnl, ref_cg = reference(bl, rates)
                lnl_g, cg = e.gradient()
                assert abs(lnl_g - ref_lnl) <= 1e-10 * abs(ref_lnl)
                assert np.abs(cg - ref_cg).max() <= 1e-9 * max(1.0, np.abs(ref_cg).max())
                rejected += 1
        assert accepted >= 4 and rejected >= 8
        # data changes drop the stored state
        e.set_pattern_weights(pb.weights)
        with pytest.raises(EngineError):
            e.restore()


@pytest.mark.parametrize("S,T,P,C", [(4, 40, 300, 4), (20, 10, 50, 2)])
def test_store_after_the_partial_storage_has_shrunk(S, T, P, C):
    """keep_partials on (every internal node stored) and off again (the fused schedule stores about a third of them, the
    allocation is kept): the first phyamd_store must move only the live slots into its two-slot buffer."""
    pb = random_problem(T, P, C, seed=7700 + S, S=S, gaps=0.02)
    ref = pb.gradient()
    with engine_from_problem(pb) as e:
        e.set_keep_partials(True)
        e.gradient()
        e.set_keep_partials(False)
        assert abs(e.log_likelihood() - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        e.store()
        bl = pb.branch_lengths * 1.3
        e.set_branch_lengths(bl)
        pb2 = po.Problem(pb.left, pb.right, pb.root, pb.weights, pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props, bl, tip_states=pb.tip_states)
        assert abs(e.log_likelihood() - pb2.log_likelihood()["lnl"]) <= 1e-10 * abs(ref["lnl"])
        e.restore()
        lnl, cg = e.gradient()
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())


def test_root_terms_agree_between_rescaled_and_unscaled_evaluations():
    """The +I root term and the frequency root term form L_k from the root partial itself, so they are the same numbers
    whether the evaluation was rescaled or not."""
    pb = random_problem(40, 300, 4, seed=77, gaps=0.02)
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as a, engine_from_problem(pb, rescale=RESCALE_ALWAYS) as b:
        la, lb = a.log_likelihood(), b.log_likelihood()
        assert b.rescaling and abs(la - lb) <= 1e-11 * abs(la)
        assert abs(a.root_invariant_term() - b.root_invariant_term()) <= 1e-10 * max(1.0, abs(a.root_invariant_term()))
        np.testing.assert_allclose(a.root_frequency_term(), b.root_frequency_term(), rtol=1e-10)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_driver")),
                    reason="oracle/_ref is not built (needs the reference tree)")
def test_headline_taxa_gradient_vector_against_the_reference_itself(tmp_path):
    """1000 taxa x 8000 sites of the bench's generator: the compiled reference (ref_driver bench ... details: the protocol of
    examples/benchmarking.c:485-503) yields its gradient vector, per-pattern lnL and compressed patterns; the engine holds the
    sites uncompressed, as the bench does, and is compared node by node through the clade map (bench.py::check_against_reference,
    the routine behind cpu_baseline.gradient_check)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from physher_amd import synth
    T, L = 1000, 8000
    rng = np.random.default_rng(77)
    tree = synth.random_tree(T, rng)
    states = np.ascontiguousarray(synth.evolve(tree, L, 4, rng))
    rates, props = bench.category_rates(4)
    ev, U, Ui = bench.gtr_eigen()
    (tmp_path / "aln.fa").write_text(synth.to_fasta(tree.names, states, "nucleotide"))
    (tmp_path / "tree.nwk").write_text(tree.newick() + "\n")
    (tmp_path / "spec.txt").write_text(f"fasta {tmp_path}/aln.fa\nnewick {tmp_path}/tree.nwk\ndatatype nucleotide\nmodel gtr\n"
                                       f"rates {','.join(map(str, bench.GTR_RATES[:5]))}\nfreqs {','.join(map(str, bench.GTR_FREQS))}\n"
                                       f"categories 4\nalpha {bench.ALPHA}\ntipstates 0\nsse 1\n")
    driver = os.path.join(root, "oracle", "_ref", "ref_driver")
    subprocess.run([driver, "bench", str(tmp_path / "spec.txt"), "1", "0", str(tmp_path / "details.json")], check=True, capture_output=True, timeout=600)
    details = json.loads((tmp_path / "details.json").read_text())
    weights = np.ones(L)
    with Engine(T, L, 4, 4) as e:
        e.set_topology(tree.left, tree.right, tree.root)
        e.set_branch_lengths(tree.length)
        e.set_eigen(ev, U, Ui)
        e.set_frequencies(np.array(bench.GTR_FREQS))
        e.set_category_rates(rates, props)
        e.set_pattern_weights(weights)
        for t in range(T):
            e.set_tip_states(t, states[t])
        e.gradient()
        plk = e.pattern_log_likelihoods()
        gc = bench.check_against_reference(e, tree, states, weights, [dict(first_site=0, details=details)], rates, props, plk)
    assert gc["ok"], gc
    sh = gc["shards"][0]
    assert sh["branches_compared"] == 2 * T - 3 and sh["sites"] == L and sh["reference_patterns"] <= L


@pytest.mark.parametrize("T,P,C,shape", [(400, 700, 4, "random"), (900, 130, 3, "caterpillar"), (256, 257, 1, "balanced"), (300, 200, 6, "random")])
def test_power_of_two_rescaling_of_the_streamed_walks(monkeypatch, T, P, C, shape):
    """Rescaled evaluations of the streamed walks scale every category by its own powers of two (integer exponents, no exchange
    between the category waves: phyamd_walk4s.inc, exp2_rescale) unless a caller needs stored partials in the reference's
    convention.  lnL, per-pattern lnL and the gradient against the oracle; the same numbers (to rounding) with
    PHYAMD_SCALE_EXP2=0; then the callers that force the reference's convention -- a single-branch evaluation, a changed branch
    (incremental update), stored partials read back -- each still equal to the oracle, and the gradient after the switch too."""
    pb = random_problem(T, P, C, seed=31 * T + P, shape=shape, gaps=0.03, bl=(0.3, 0.9), rescale=1)
    ref = pb.gradient(want_partials=True)
    assert ref["rescaled"]
    tol = 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
    with engine_from_problem(pb, rescale=RESCALE_ALWAYS) as e:
        lnl, cg = e.gradient()
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]) and np.abs(cg - ref["cat_grad"]).max() <= tol
        np.testing.assert_allclose(e.pattern_log_likelihoods(), ref["pattern_lk"], rtol=1e-11, atol=1e-11)
        assert e.log_likelihood() == lnl  # the post-order walk alone: the same bits
        lnl2, cg2 = e.gradient()
        assert lnl2 == lnl and np.array_equal(cg, cg2)  # reproducible
        # a single-branch evaluation needs the reference's convention: the post-order pass runs again, the numbers stay
        node = T + 3
        lt, d1, _ = e.branch_log_likelihood(node, pb.branch_lengths[node])
        assert abs(lt - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        lnl3, cg3 = e.gradient()
        assert abs(lnl3 - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]) and np.abs(cg3 - ref["cat_grad"]).max() <= tol
    monkeypatch.setenv("PHYAMD_SCALE_EXP2", "0")
    with engine_from_problem(pb, rescale=RESCALE_ALWAYS) as e:
        lnl0, cg0 = e.gradient()
        assert abs(lnl0 - lnl) <= 1e-12 * abs(lnl) and np.abs(cg0 - cg).max() <= tol
    monkeypatch.delenv("PHYAMD_SCALE_EXP2")
    with engine_from_problem(pb, rescale=RESCALE_ALWAYS) as e:
        e.gradient()
        # a changed branch after a power-of-two pass: the incremental update cannot read those partials, everything is recomputed
        bl = pb.branch_lengths.copy()
        bl[5] *= 1.7
        e.set_branch_length(5, bl[5])
        q = po.Problem(pb.left, pb.right, pb.root, pb.weights, pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props, bl, tip_states=pb.tip_states, rescale=1)
        r2 = q.gradient()
        lnl4, cg4 = e.gradient()
        assert abs(lnl4 - r2["lnl"]) <= 1e-10 * abs(r2["lnl"]) and np.abs(cg4 - r2["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(r2["cat_grad"]).max())
    with engine_from_problem(pb, rescale=RESCALE_ALWAYS) as e:
        e.gradient()
        # stored partials read back are the reference's (its scale factors), whatever the walks used in between
        stored = [n for n in range(T, pb.N) if _is_stored(e, n)]
        for n in stored[:5] + [pb.root]:
            np.testing.assert_allclose(e.partials(n), ref["lower"][n], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("T,P,C,shape", [(400, 700, 4, "random"), (900, 130, 3, "caterpillar"), (256, 257, 1, "balanced")])
def test_carried_storage_of_the_streamed_walks(monkeypatch, T, P, C, shape):
    """Between the two streamed 4-state walks a stored node is t_n = P_n p_n (k_lower4_stream TF: the pre-order walk takes a stored
    child as its message).  lnL, per-pattern lnL and both gradient conventions against the oracle; the same numbers to rounding with
    PHYAMD_STREAM_TFORM=0; then every caller that reads stored partials -- partials read back, a single-branch evaluation, a changed
    branch (incremental update), the substitution-parameter gradient, store / restore -- gets the partials themselves (the post-order
    pass runs again once) and still equals the oracle, as does the gradient afterwards."""
    pb = random_problem(T, P, C, seed=17 * T + P, shape=shape, gaps=0.03)
    ref = pb.gradient(want_partials=True)
    tol = 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        lnl, cg = e.gradient()
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]) and np.abs(cg - ref["cat_grad"]).max() <= tol
        np.testing.assert_allclose(e.pattern_log_likelihoods(), ref["pattern_lk"], rtol=1e-11, atol=1e-11)
        assert e.log_likelihood() == lnl
        pb.fold_root_freqs = 1
        rf = pb.gradient()
        pb.fold_root_freqs = 0
        lnl_f, cg_f = e.gradient(GRAD_FOLD_ROOT_FREQS)  # stays on the streamed walks (no post-order pass in between)
        assert lnl_f == lnl and np.abs(cg_f - rf["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(rf["cat_grad"]).max())
        # stored partials read back are the partials themselves
        stored = [n for n in range(T, pb.N) if _is_stored(e, n)]
        assert len(stored) >= 3
        for n in stored[:4] + [pb.root]:
            np.testing.assert_allclose(e.partials(n), ref["lower"][n], rtol=1e-9, atol=1e-300)
        lnl2, cg2 = e.gradient()
        assert abs(lnl2 - lnl) <= 1e-13 * abs(lnl) and np.abs(cg2 - ref["cat_grad"]).max() <= tol
    monkeypatch.setenv("PHYAMD_STREAM_TFORM", "0")
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        lnl0, cg0 = e.gradient()
        assert abs(lnl0 - lnl) <= 1e-13 * abs(lnl) and np.abs(cg0 - cg).max() <= 1e-11 * max(1.0, np.abs(cg).max())
    monkeypatch.delenv("PHYAMD_STREAM_TFORM")
    ref_bg = po.branch_gradient_from_cat(ref["cat_grad"], pb.cat_rates, pb.cat_props)
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        e.gradient()
        node = stored[1]
        lt, d1, _ = e.branch_log_likelihood(node, pb.branch_lengths[node])  # a stored node's own branch: needs p_node
        assert abs(lt - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]) and abs(d1 - ref_bg[node]) <= 1e-9 * max(1.0, np.abs(ref_bg).max())
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        e.gradient()
        bl = pb.branch_lengths.copy()
        bl[stored[2]] *= 1.6  # an incremental update reads stored children
        e.set_branch_length(stored[2], bl[stored[2]])
        q = po.Problem(pb.left, pb.right, pb.root, pb.weights, pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props, bl, tip_states=pb.tip_states)
        r2 = q.gradient()
        lnl4, cg4 = e.gradient()
        assert abs(lnl4 - r2["lnl"]) <= 1e-10 * abs(r2["lnl"]) and np.abs(cg4 - r2["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(r2["cat_grad"]).max())
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as e:
        lnl5, _ = e.gradient()
        e.store()  # the stored state is kept in the compatible form
        e.set_branch_length(stored[0], 2.0 * pb.branch_lengths[stored[0]])
        assert e.log_likelihood() != lnl5
        e.restore()
        lnl6, cg6 = e.gradient()
        assert abs(lnl6 - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]) and np.abs(cg6 - ref["cat_grad"]).max() <= tol
    # the substitution-parameter sums after a plain gradient: their kernels read stored partials
    rng = np.random.default_rng(3)
    dQ = rng.normal(size=(2, 4, 4))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(4)[None]
    with engine_from_problem(pb, rescale=RESCALE_NEVER) as a, engine_from_problem(pb, rescale=RESCALE_NEVER) as b:
        a.gradient()  # a: the streamed walks first, then the parameter gradient; b: the parameter gradient at once
        a.set_rate_matrix_derivatives(dQ)
        b.set_rate_matrix_derivatives(dQ)
        ra, rb = a.parameter_gradient(), b.parameter_gradient()
        for x, y in zip(ra, rb):
            np.testing.assert_allclose(np.asarray(x), np.asarray(y), rtol=1e-10, atol=1e-10 * max(1.0, float(np.abs(np.asarray(y)).max())))


def _is_stored(e, node):
    try:
        e.partials(node)
        return True
    except EngineError:
        return False
