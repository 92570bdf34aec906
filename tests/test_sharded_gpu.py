"""GPU: the engine sharded over several devices inside ONE process (phyamd_create_sharded) against the CPU
oracle.  A one-GPU box lists the same ordinal several times -- every shard is still a complete engine with its own stream and
its own host thread, so the slicing of per-pattern arguments, the replication of tree and model, the concurrent evaluation and
the fixed-order host sum are the ones an 8-GPU node runs; with more than one device visible the shards really spread out."""
import numpy as np
import pytest
import torch

from gpu_util import engine_from_problem, random_problem
from oracle import phyoracle as po
from physher_amd.engine import RESCALE_ALWAYS, RESCALE_AUTO, RESCALE_NEVER, Engine, EngineError

pytestmark = pytest.mark.gpu


def device_list(n):
    have = max(1, torch.cuda.device_count())
    return [i % have for i in range(n)]


@pytest.mark.parametrize("S,T,P,C,n,rescale", [(4, 40, 1001, 4, 2, RESCALE_NEVER), (4, 25, 517, 2, 3, RESCALE_AUTO), (4, 60, 300, 4, 2, RESCALE_ALWAYS),
                                               (20, 12, 203, 2, 2, RESCALE_NEVER), (61, 8, 70, 1, 3, RESCALE_NEVER), (4, 12, 8, 1, 8, RESCALE_NEVER)])
def test_sharded_engine_matches_oracle(S, T, P, C, n, rescale):
    forced = rescale == RESCALE_ALWAYS
    pb = random_problem(T, P, C, seed=9100 + S + T + n, S=S, gaps=0.03, bl=(0.3, 0.9) if forced else (0.01, 0.1), rescale=1 if forced else 0)
    ref = pb.gradient(want_partials=True)
    with engine_from_problem(pb, rescale=rescale, devices=device_list(n)) as e, engine_from_problem(pb, rescale=rescale) as one:
        assert e.shard_count == n and one.shard_count == 1
        lnl = e.log_likelihood()
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        np.testing.assert_allclose(e.pattern_log_likelihoods(), ref["pattern_lk"], rtol=1e-10, atol=1e-10)  # concatenated in shard order
        l2, cg = e.gradient()
        assert abs(l2 - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
        l1, cg1 = one.gradient()
        assert abs(l2 - l1) <= 1e-12 * abs(l1) and np.abs(cg - cg1).max() <= 1e-11 * max(1.0, np.abs(cg1).max())  # (another summation order)
        _, bg = e.branch_gradient()
        np.testing.assert_allclose(bg, po.branch_gradient_from_cat(cg, pb.cat_rates, pb.cat_props), rtol=1e-13, atol=1e-13)
        if C >= 2:
            assert abs(e.root_invariant_term() - one.root_invariant_term()) <= 1e-10 * max(1.0, abs(one.root_invariant_term()))
        np.testing.assert_allclose(e.root_frequency_term(), po.root_frequency_term(pb), rtol=1e-9)
        # the optimiser's fast path: sums over patterns, shard by shard
        node = T + 1 if T + 1 != pb.root else T
        t = 1.3 * pb.branch_lengths[node]
        lt, d1, d2 = e.branch_log_likelihood(node, t)
        lt1, d11, d21 = one.branch_log_likelihood(node, t)
        pb.branch_lengths[node] = t
        o = pb.gradient()
        assert abs(lt - o["lnl"]) <= 1e-10 * abs(o["lnl"])
        assert abs(d1 - po.branch_gradient_from_cat(o["cat_grad"], pb.cat_rates, pb.cat_props)[node]) <= 1e-9 * max(1.0, np.abs(o["cat_grad"]).max())
        assert abs(d2 - d21) <= 1e-9 * max(1.0, abs(d21))
        # incremental update + store / restore are replicated to every shard
        e.store()
        e.set_branch_length(node, t)
        assert abs(e.log_likelihood() - o["lnl"]) <= 1e-10 * abs(o["lnl"])
        e.restore()
        assert abs(e.log_likelihood() - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        pb.branch_lengths[node] = t / 1.3
        # partials come back in the reference's [C][P][S] layout, stitched from the shards' pattern ranges
        e.set_keep_partials(True)
        e.gradient()
        inner = [x for x in range(T, pb.N) if x != pb.root][0]
        np.testing.assert_allclose(e.partials(inner), ref["lower"][inner], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(e.partials(0, upper=True), ref["upper"][0], rtol=1e-9, atol=1e-300)
        # device-resident outputs belong to ONE device
        with pytest.raises(EngineError):
            e.gradient_device(0)


@pytest.mark.parametrize("P", [6000, 4133, 100_003])
def test_result_does_not_depend_on_the_shard_count(P):
    """SURVEY 8e "deterministic alternative": lnL and the gradient of the default (unscaled, 4-state) path are sums over blocks
    of 64 patterns in an order fixed by the pattern list alone -- eight bisection segments, added pairwise -- and 2, 4 or 8
    shards are subtrees of that bisection whose results the host adds pairwise: every shard count returns the SAME BITS."""
    pb = random_problem(40, P, 4, seed=9400 + P % 97, gaps=0.02)
    results = {}
    for n in (1, 2, 4, 8):
        with (engine_from_problem(pb, rescale=RESCALE_NEVER, devices=device_list(n)) if n > 1 else engine_from_problem(pb, rescale=RESCALE_NEVER)) as e:
            lnl0 = e.log_likelihood()
            lnl, cg = e.gradient()
            assert lnl0 == lnl
            results[n] = (lnl, cg.copy())
    ref = pb.gradient()
    assert abs(results[1][0] - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
    assert np.abs(results[1][1] - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
    for n in (2, 4, 8):
        assert results[n][0] == results[1][0], (n, results[n][0], results[1][0])
        assert np.array_equal(results[n][1], results[1][1]), n


def test_sharded_parameter_gradient_matches_oracle():
    S, T, P, C, n = 4, 30, 700, 4, 2
    pb = random_problem(T, P, C, seed=9300, S=S, gaps=0.02)
    rng = np.random.default_rng(2)
    dQ = rng.normal(size=(5, S, S))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(S)[None]
    _, want = po.parameter_gradient(pb, dQ)
    with engine_from_problem(pb, devices=device_list(n)) as e:
        e.set_rate_matrix_derivatives(dQ)
        lnl, cg, pg = e.parameter_gradient()
        assert np.abs(pg - want).max() <= 1e-8 * max(1.0, np.abs(want).max())
        ref = pb.gradient()
        assert abs(lnl - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())


def test_sharded_engine_argument_checks():
    with pytest.raises(EngineError):
        Engine(4, 3, devices=[0, 0, 0, 0])  # fewer patterns than shards
    with pytest.raises(EngineError):
        Engine(4, 100, devices=[0, 99])  # no such device
    with pytest.raises(EngineError):
        Engine(4, 100, devices=[])


def test_sharded_evaluations_overlap_in_time():
    """Not a speed assertion: two shards of a mid-size problem on one GPU run on two streams from two host threads; the test
    log records one shard alone against both together (on a multi-GPU node the second number is the per-GPU time)."""
    import time
    pb = random_problem(200, 60000, 4, seed=9400, gaps=0.0)
    ref = pb.gradient()
    with engine_from_problem(pb, rescale=RESCALE_NEVER, devices=device_list(2)) as e, engine_from_problem(pb, rescale=RESCALE_NEVER) as one:
        for eng in (e, one):
            eng.gradient()
        t0 = time.perf_counter()
        for _ in range(5):
            one.update_all_nodes()
            l1, _ = one.gradient()
        t1 = time.perf_counter()
        for _ in range(5):
            e.update_all_nodes()
            l2, cg = e.gradient()
        t2 = time.perf_counter()
        assert abs(l2 - ref["lnl"]) <= 1e-10 * abs(ref["lnl"]) and abs(l1 - ref["lnl"]) <= 1e-10 * abs(ref["lnl"])
        assert np.abs(cg - ref["cat_grad"]).max() <= 1e-9 * max(1.0, np.abs(ref["cat_grad"]).max())
        print(f"\n[sharded, 200 x 60000] one engine {1e3 * (t1 - t0) / 5:.2f} ms / evaluation, two shards on {device_list(2)} {1e3 * (t2 - t1) / 5:.2f} ms")
