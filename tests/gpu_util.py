"""Helpers shared by the GPU parity tests: build an Engine from a golden case / an oracle Problem."""
import numpy as np

from physher_amd.engine import Engine


def engine_from_problem(pb, rescale=2, tip_mode="states", **kw):
    e = Engine(pb.T, pb.P, pb.S, pb.C, rescale=rescale, **kw)
    e.set_topology(pb.left, pb.right, pb.root)
    e.set_branch_lengths(pb.branch_lengths)
    e.set_eigen(pb.eval, pb.evec, pb.ivec)
    e.set_frequencies(pb.freqs)
    e.set_category_rates(pb.cat_rates, pb.cat_props)
    e.set_pattern_weights(pb.weights)
    for t in range(pb.T):
        if pb.tip_partials is not None and tip_mode != "states":
            e.set_tip_partials(t, pb.tip_partials[t])
        else:
            e.set_tip_states(t, pb.tip_states[t])
    return e


def random_problem(T, P, C, seed, S=4, shape="random", gaps=0.0, bl=(0.01, 0.1), rescale=0, **kw):
    """Seeded synthetic problem (oracle Problem object) with a GTR-like reversible model."""
    from oracle import phyoracle as po
    from physher_amd import synth
    from golden_util import reversible_eigen
    rng = np.random.default_rng(seed)
    tree = synth.random_tree(T, rng, shape=shape, bl_low=bl[0], bl_high=bl[1])
    states = synth.evolve(tree, P, S, rng)
    if gaps > 0:
        states = np.where(rng.random(states.shape) < gaps, S + 13, states).astype(np.uint8)
    weights = rng.integers(1, 5, size=P).astype(np.float64)
    freqs = rng.dirichlet(np.full(S, 5.0))
    r = rng.uniform(0.5, 3.0, size=(S, S))
    r = 0.5 * (r + r.T)
    ev, U, Ui = reversible_eigen(r, freqs)
    rates = np.sort(rng.gamma(0.5, 2.0, size=C)) + 0.05
    props = np.full(C, 1.0 / C)
    rates = rates / (rates * props).sum()
    return po.Problem(tree.left, tree.right, tree.root, weights, ev, U, Ui, freqs, rates, props, tree.length,
                      tip_states=states, rescale=rescale, **kw)
