"""Site-pattern sharding across the GPUs of one node.

Patterns are independent given the tree and the parameters, so rank r owns the contiguous range
[r*P/R, (r+1)*P/R) of the compressed pattern list and evaluates it with its own engine.  The only exchange
per evaluation is one SUM all-reduce of the vector [lnL, g[0][0..C-1], g[1][..], ...] (1 + N*C doubles,
64 KB at 1000 taxa x 4 categories) -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
The O(N*C) epilogue (treelikelihood.c:3129-3143) then runs on every rank.
"""
from __future__ import annotations

import numpy as np


WAVE = 64  # patterns per block of the engine's per-block sums


def _bisect(lo, hi, levels, out):
    if levels == 0:
        out.append(lo)
        return
    mid = lo + (hi - lo) // 2
    _bisect(lo, mid, levels - 1, out)
    _bisect(mid, hi, levels - 1, out)


def shard_range(pattern_count: int, rank: int, world: int):
    """[lo, hi) of rank's patterns; ranges are contiguous, disjoint and cover everything.

    world = 2, 4, 8: the range of 64-pattern blocks is bisected 1, 2, 3 times (mid = lo + (hi - lo) // 2) -- the segments the engine
    itself sums by (phyamd_set_reduction_levels) -- so every rank holds a subtree of the one-GPU summation and the pairwise sum
    of the ranks' results (tree_sum) is bit for bit the one-GPU result.  Other world sizes: equal parts."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    nb = (pattern_count + WAVE - 1) // WAVE
    if world in (2, 4, 8) and nb >= world:
        bounds = []
        _bisect(0, nb, {2: 1, 4: 2, 8: 3}[world], bounds)
        bounds.append(nb)
        return bounds[rank] * WAVE, min(pattern_count, bounds[rank + 1] * WAVE)
    return rank * pattern_count // world, (rank + 1) * pattern_count // world


def reduction_levels(pattern_count: int, world: int) -> int:
    """levels for Engine.set_reduction_levels on one rank of `world` (see shard_range)."""
    nb = (pattern_count + WAVE - 1) // WAVE
    return 3 - {2: 1, 4: 2, 8: 3}[world] if world in (2, 4, 8) and nb >= world else 3


def tree_sum(parts):
    """pairwise sum of the ranks' vectors up the bisection tree (2, 4 or 8 of them), in rank order otherwise"""
    parts = list(parts)
    if len(parts) in (2, 4, 8):
        while len(parts) > 1:
            parts = [parts[i] + parts[i + 1] for i in range(0, len(parts), 2)]
        return parts[0]
    total = parts[0]
    for p in parts[1:]:
        total = total + p
    return total


def all_reduce_result(result, world: int, via_host: bool = False, deterministic: bool = False, force: bool = False):
    """In-place SUM of the per-shard [lnL, cat-gradient] vector (a torch tensor on the engine's device).
    force: run the collective even in a world of one (a process group must exist): the RCCL path of a one-GPU rehearsal.

    Default: ONE all-reduce (RCCL over xGMI on GPUs) -- the order of the additions is the collective's.
    deterministic: ONE all-gather of the same 64 KB vectors and the pairwise tree_sum on every rank -- with shard_range's
    ranges the result is bit for bit that of one GPU.
    via_host: rehearsal mode for a one-GPU box (several ranks share the card, gloo group): the 64 KB vector makes a
    round trip through host memory because gloo cannot reduce device tensors on ROCm."""
    if world > 1 or force:
        import torch
        import torch.distributed as dist
        src = result.cpu() if (via_host and result.is_cuda) else result
        if deterministic:
            parts = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(parts, src)
            src = tree_sum(parts)
        else:
            dist.all_reduce(src, op=dist.ReduceOp.SUM)
        if src is not result:
            result.copy_(src)
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

    def __init__(self, evaluate_shard, node_count, cat_rates, cat_props, world, result_buffer, via_host=False, tail=0, deterministic=False,
                 force_collective=False, timers=None):
        """tail > 0: the vector carries `tail` more per-shard sums after the cat-gradient (Engine.parameter_gradient_device:
        substitution-parameter sums, then the root frequency term); they ride in the same all-reduce and are returned third.
        timers: a dict that receives, per call, "all_reduce_us" (device time of the collective between two events on the current
        stream -- the engine's) and "host_epilogue_us"."""
        self.tail = tail
        self.force = force_collective
        self.timers = timers
        self.via_host = via_host
        self.deterministic = deterministic
        self.evaluate_shard = evaluate_shard
        self.N = node_count
        self.cat_rates = np.asarray(cat_rates, dtype=np.float64)
        self.cat_props = np.asarray(cat_props, dtype=np.float64)
        self.world = world
        self.result = result_buffer

    def __call__(self):
        import time
        self.evaluate_shard(self.result)
        timed = self.timers is not None and (self.world > 1 or self.force) and self.result.is_cuda and not self.via_host
        if timed:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        all_reduce_result(self.result, self.world, self.via_host, self.deterministic, self.force)
        if timed:
            e1.record()
        host = self.result.detach().cpu().numpy()
        t0 = time.perf_counter()
        lnl, grad = epilogue(host, self.N, self.cat_rates, self.cat_props)
        if self.timers is not None:
            self.timers.setdefault("host_epilogue_us", []).append(1e6 * (time.perf_counter() - t0))
            if timed:
                e1.synchronize()
                self.timers.setdefault("all_reduce_us", []).append(1e3 * e0.elapsed_time(e1))
        if self.tail:
            return lnl, grad, host[1 + self.N * len(self.cat_rates):][:self.tail].copy()
        return lnl, grad
