"""Site-pattern sharding across the GPUs of one node.

Patterns are independent given the tree and the parameters, so rank r owns the contiguous range
[r*P/R, (r+1)*P/R) of the compressed pattern list and evaluates it with its own engine.  The only exchange
per evaluation is one SUM all-reduce of the vector [lnL, g[0][0..C-1], g[1][..], ...] (1 + N*C doubles,
64 KB at 1000 taxa x 4 categories) -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
The O(N*C) epilogue (treelikelihood.c:3129-3143) then runs on every rank.
"""
from __future__ import annotations

import numpy as np


def shard_range(pattern_count: int, rank: int, world: int):
    """[lo, hi) of rank's patterns; ranges are contiguous, disjoint, cover everything, differ by at most 1 in size."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * pattern_count // world, (rank + 1) * pattern_count // world


def all_reduce_result(result, world: int, via_host: bool = False):
    """In-place SUM of the per-shard [lnL, cat-gradient] vector (a torch tensor on the engine's device).

    via_host: rehearsal mode for a one-GPU box (several ranks share the card, gloo group): the 64 KB vector makes a
    round trip through host memory because gloo cannot reduce device tensors on ROCm."""
    if world > 1:
        import torch.distributed as dist
        if via_host and result.is_cuda:
            tmp = result.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
            result.copy_(tmp)
        else:
            dist.all_reduce(result, op=dist.ReduceOp.SUM)
    return result


def epilogue(result: np.ndarray, node_count: int, cat_rates, cat_props):
    """[lnL, g[node][cat]] -> (lnL, branch gradient [node]): gradient_branch_length_from_cat_inplace."""
    C = len(cat_rates)
    cg = np.asarray(result[1:1 + node_count * C]).reshape(node_count, C)
    if C == 1:
        return float(result[0]), cg[:, 0].copy()
    return float(result[0]), (cg * (np.asarray(cat_props) * np.asarray(cat_rates))[None, :]).sum(axis=1)


class ShardedLikelihood:
    """lnL + branch gradient of the whole alignment from this rank's shard.

    evaluate_shard(out) must write this shard's [lnL, cat-gradient] into `out` (a torch tensor); on GPUs that is
    Engine.gradient_device(out.data_ptr()) -- the HIP kernels -- and in the CPU tests a stand-in.
    The engine writes `out` asynchronously on ITS stream: create it with stream = a non-default torch stream that is also
    torch's current stream (torch.cuda.set_stream), so that the all-reduce and the host copy below are ordered behind the
    kernels.  (Handle 0, torch's default stream, means "engine-owned stream" to phyamd_create and is NOT ordered with torch.)
    """

    def __init__(self, evaluate_shard, node_count, cat_rates, cat_props, world, result_buffer, via_host=False, tail=0):
        """tail > 0: the vector carries `tail` more per-shard sums after the cat-gradient (Engine.parameter_gradient_device:
        substitution-parameter sums, then the root frequency term); they ride in the same all-reduce and are returned third."""
        self.tail = tail
        self.via_host = via_host
        self.evaluate_shard = evaluate_shard
        self.N = node_count
        self.cat_rates = np.asarray(cat_rates, dtype=np.float64)
        self.cat_props = np.asarray(cat_props, dtype=np.float64)
        self.world = world
        self.result = result_buffer

    def __call__(self):
        self.evaluate_shard(self.result)
        all_reduce_result(self.result, self.world, self.via_host)
        host = self.result.detach().cpu().numpy()
        lnl, grad = epilogue(host, self.N, self.cat_rates, self.cat_props)
        if self.tail:
            return lnl, grad, host[1 + self.N * len(self.cat_rates):][:self.tail].copy()
        return lnl, grad
