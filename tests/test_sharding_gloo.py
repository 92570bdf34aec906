"""world_size-2 gloo test (CPU) of the multi-GPU path: shard ranges, the single all-reduce, the epilogue.

The per-shard evaluator is the CPU oracle here (no GPU in this container); on the GPU box the same
ShardedLikelihood drives Engine.gradient_device (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpu_util import random_problem
    from oracle import phyoracle as po
    from physher_amd.sharding import ShardedLikelihood, shard_range

    pb = random_problem(T=24, P=501, C=4, seed=77, gaps=0.02)  # same problem on every rank
    lo, hi = shard_range(pb.P, rank, world)
    sub = po.Problem(pb.left, pb.right, pb.root, pb.weights[lo:hi], pb.eval, pb.evec, pb.ivec, pb.freqs, pb.cat_rates, pb.cat_props,
                     pb.branch_lengths, tip_states=pb.tip_states[:, lo:hi])

    def evaluate_shard(out):
        r = sub.gradient()
        out[0] = r["lnl"]
        out[1:] = torch.from_numpy(r["cat_grad"].reshape(-1))

    buf = torch.zeros(1 + pb.N * pb.C, dtype=torch.float64)
    lnl, bg = ShardedLikelihood(evaluate_shard, pb.N, pb.cat_rates, pb.cat_props, world, buf)()
    full = pb.gradient()
    ref_bg = po.branch_gradient_from_cat(full["cat_grad"], pb.cat_rates, pb.cat_props)
    ok = abs(lnl - full["lnl"]) <= 1e-11 * abs(full["lnl"]) and np.abs(bg - ref_bg).max() <= 1e-10 * max(1.0, np.abs(ref_bg).max())
    # the substitution-parameter sums and the root frequency term are per-pattern sums too: they ride in the same all-reduce
    rng = np.random.default_rng(5)
    dQ = rng.normal(size=(3, 4, 4))
    dQ -= dQ.sum(axis=2, keepdims=True) * np.eye(4)[None]

    def evaluate_shard_params(out):
        evaluate_shard(out[: 1 + pb.N * pb.C])
        out[1 + pb.N * pb.C: 1 + pb.N * pb.C + 3] = torch.from_numpy(po.parameter_gradient(sub, dQ)[1])
        out[1 + pb.N * pb.C + 3:] = torch.from_numpy(po.root_frequency_term(sub))

    buf2 = torch.zeros(1 + pb.N * pb.C + 3 + 4, dtype=torch.float64)
    lnl2, bg2, tail = ShardedLikelihood(evaluate_shard_params, pb.N, pb.cat_rates, pb.cat_props, world, buf2, tail=7)()
    ref_tail = np.concatenate([po.parameter_gradient(pb, dQ)[1], po.root_frequency_term(pb)])
    ok = ok and lnl2 == lnl and np.array_equal(bg2, bg) and np.abs(tail - ref_tail).max() <= 1e-10 * max(1.0, np.abs(ref_tail).max())
    # the deterministic form: one all-gather, then the pairwise sum every rank forms for itself
    buf3 = torch.zeros(1 + pb.N * pb.C, dtype=torch.float64)
    lnl3, bg3 = ShardedLikelihood(evaluate_shard, pb.N, pb.cat_rates, pb.cat_props, world, buf3, deterministic=True)()
    ok = ok and abs(lnl3 - full["lnl"]) <= 1e-11 * abs(full["lnl"]) and np.abs(bg3 - ref_bg).max() <= 1e-10 * max(1.0, np.abs(ref_bg).max())
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate([[float(ok), lnl, hi - lo], bg]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_likelihood(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert r0[0] == 1.0 and r1[0] == 1.0          # both ranks match the unsharded oracle
    assert r0[1] == r1[1]                           # and hold the identical all-reduced lnL
    assert np.array_equal(r0[3:], r1[3:])
    assert r0[2] + r1[2] == 501 and abs(r0[2] - r1[2]) <= 64  # whole 64-pattern blocks


def test_tree_sum_is_the_bisection_order():
    from physher_amd.sharding import reduction_levels, tree_sum
    rng = np.random.default_rng(3)
    parts = [rng.normal(size=5) * 10.0 ** rng.integers(-8, 8, size=5) for _ in range(8)]
    want = ((parts[0] + parts[1]) + (parts[2] + parts[3])) + ((parts[4] + parts[5]) + (parts[6] + parts[7]))
    assert np.array_equal(tree_sum(parts), want)
    assert np.array_equal(tree_sum(parts[:2]), parts[0] + parts[1])
    assert np.array_equal(tree_sum(parts[:3]), (parts[0] + parts[1]) + parts[2])
    assert [reduction_levels(1_000_000, w) for w in (1, 2, 4, 8, 3)] == [3, 2, 1, 0, 3]


@pytest.mark.parametrize("P,world", [(1_000_000, 8), (1001, 4), (7, 8), (5, 2)])
def test_shard_ranges_partition_the_patterns(P, world):
    from physher_amd.sharding import shard_range
    edges = [shard_range(P, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == P
    assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 127  # 2 / 4 / 8 ranks: whole 64-pattern blocks by bisection (the engine's summation segments)
    if world in (2, 4, 8) and (P + 63) // 64 >= world:
        assert all(a % 64 == 0 for a, _ in edges)
        # every rank's range is a subtree of the bisection of the whole block range: halves of halves
        half = shard_range(P, 0, 2)[1]
        assert any(b == half for _, b in edges)
    with pytest.raises(ValueError):
        shard_range(P, world, world)
