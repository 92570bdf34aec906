#!/usr/bin/env python3
"""Benchmark of the tree-likelihood hot path: lnL + branch-length-gradient evaluations per second.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1: launched by torch.distributed.run, one
rank per GPU).  A step is one FULL re-evaluation, the protocol of the reference's own harness
(examples/benchmarking.c:498-503): every branch's P(t) is rebuilt from the eigen system, then the
post-order pass, the pre-order pass and the gradient are recomputed; the result (lnL and the
per-branch gradient) ends up on the host.  Inputs (tip states, weights) are resident in HBM before the
timed region.  With N > 1 the site patterns are sharded across ranks and one RCCL all-reduce of the
[lnL, gradient] vector joins them (strong scaling: the workload is the same 1e6 patterns for every N).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the `roofline` and `cpu_baseline` objects).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

GTR_RATES = (1.2, 3.1, 0.7, 0.9, 2.8, 1.0)  # ac ag at cg ct gt
GTR_FREQS = (0.3, 0.2, 0.2, 0.3)
ALPHA = 0.5
# physher's own discrete gamma(alpha = 0.5), 4 categories (quantile medians rescaled to mean 1: sitemodel.c / gamma.c; read off
# `ref_driver dump`'s cat_rates): the engine takes rates as inputs, and the CPU baseline's lnL of the same sample is compared
# with the engine's (cpu_baseline.lnl_check), so both must discretise alike
GAMMA4_RATES_05 = (0.029077754761923313, 0.28071453713995315, 0.9247730651141549, 2.7654346429839682)


def gtr_eigen():
    """Eigen system of the normalised GTR rate matrix (host-side numpy; O(1) work, outside the timed region)."""
    pi = np.array(GTR_FREQS)
    ac, ag, at, cg, ct, gt = GTR_RATES
    r = np.array([[0, ac, ag, at], [ac, 0, cg, ct], [ag, cg, 0, gt], [at, ct, gt, 0]], dtype=np.float64)
    Q = r * pi[None, :]
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q /= -(pi * np.diag(Q)).sum()
    d = np.sqrt(pi)
    B = (d[:, None] * Q) / d[None, :]
    w, V = np.linalg.eigh(0.5 * (B + B.T))
    return w, V / d[:, None], V.T * d[None, :]


def evolve_on_device(tree, site_count, seed, device, S=4):
    """Sequences for every tip, evolved down the tree on the GPU (equal-rates model, S states).  uint8 [T][sites] torch tensor."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N = tree.node_count
    seqs = [None] * N
    seqs[tree.root] = torch.randint(0, S, (site_count,), dtype=torch.uint8, device=device, generator=g)
    for n in range(N - 1, -1, -1):
        if tree.left[n] < 0:
            continue
        for c in (int(tree.left[n]), int(tree.right[n])):
            p_same = 1.0 / S + (1.0 - 1.0 / S) * np.exp(-S / (S - 1.0) * tree.length[c])
            change = torch.rand(site_count, device=device, generator=g) >= p_same
            new = torch.randint(0, S, (site_count,), dtype=torch.uint8, device=device, generator=g)
            seqs[c] = torch.where(change, new, seqs[n])
        seqs[n] = None
    return torch.stack(seqs[: tree.tip_count])


def algorithmic_flops(T, P, C, S):
    """lower / upper node 2S(2S-1)+S flops per (pattern, category), branch gradient S(2S-1)+2S."""
    return P * C * ((3 * T - 3) * (2 * S * (2 * S - 1) + S) + (2 * T - 2) * (S * (2 * S - 1) + 2 * S))


# BASELINE.json configs[1..4]; the headline metric is quoted on configs[4]'s shape, which fits one GPU (160 GB)
WORKLOADS = {
    "cfg2": dict(taxa=500, patterns=100_000, states=4, categories=4, name="GTR+G4 DNA"),
    "cfg3": dict(taxa=200, patterns=50_000, states=20, categories=4, name="WAG+G4 amino-acid"),
    "cfg4": dict(taxa=100, patterns=20_000, states=61, categories=1, name="MG94 codon"),
    "cfg5": dict(taxa=1000, patterns=1_000_000, states=4, categories=4, name="GTR+G4 DNA"),
}
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X spec sheet (fp64 matrix = fp64 vector on CDNA4)


def algorithmic_bytes(T, P, C, S):
    """SURVEY.md 8(d) three-pass accounting, split per kernel family."""
    B = 8.0 * P * C * S
    lower = (2 * T - 3) * B + T * P
    upper_grad = (8 * T - 12) * B + 2 * T * P + 16 * (T - 1) * P
    return lower, upper_grad


def cpu_baseline(tree, states, weights, cat_rates, sample_patterns, budget_s):
    """Time the CPU path on a bounded sample (the first `sample_patterns` patterns) of the same workload.

    Preferred: the compiled reference itself (oracle/_ref/ref_driver: physher's SSE kernels, 1 thread, the protocol
    of examples/benchmarking.c).  Fallback: this repository's CPU port (oracle/libphyoracle.so).  The measured
    per-evaluation time is scaled linearly in the pattern count (the cost is linear in patterns).
    """
    from physher_amd import synth
    T = tree.tip_count
    sp = min(sample_patterns, states.shape[1])
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    sub = states[:, :sp]
    if os.path.exists(driver):
        # Three shards of the workload (its first, middle and last `sp` sites), one after the other on one core.  Whole shards of 1e5
        # patterns are out of reach of a bounded baseline: one gradient evaluation of 1000 taxa x 1e5 patterns costs ~16 s on a core and
        # the reference holds ~51 GB of partial arrays for it (2 x 1999 nodes x 1e5 x 16 doubles).
        L = states.shape[1]
        starts = sorted({0, max(0, (L - sp) // 2), max(0, L - sp)})
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "tree.nwk"), "w") as f:
                f.write(tree.newick() + "\n")
            specs = []
            for j, st in enumerate(starts):
                with open(os.path.join(d, f"aln{j}.fa"), "w") as f:
                    f.write(synth.to_fasta(tree.names, states[:, st:st + sp], "nucleotide"))
                with open(os.path.join(d, f"spec{j}.txt"), "w") as f:
                    f.write(f"fasta {d}/aln{j}.fa\nnewick {d}/tree.nwk\ndatatype nucleotide\nmodel gtr\n"
                            f"rates {','.join(map(str, GTR_RATES[:5]))}\nfreqs {','.join(map(str, GTR_FREQS))}\n"
                            f"categories {len(cat_rates)}\nalpha {ALPHA}\ntipstates 0\nsse 1\n")
                specs.append(os.path.join(d, f"spec{j}.txt"))
            # one gradient evaluation costs 16-26 ns per (branch, pattern, category) per core; the driver also times as many
            # lnL-only evaluations (~40 % of that) and one warm-up of each
            est = 20e-9 * (2 * T - 2) * sp * len(cat_rates)
            iters = max(2, min(40, int(budget_s / (1.5 * est * len(specs)))))
            try:
                runs = []
                for j, spec in enumerate(specs):
                    det_path = os.path.join(d, f"details{j}.json")
                    out = subprocess.run([driver, "bench", spec, str(iters), "1", det_path], capture_output=True, text=True,
                                         timeout=max(120, 20 * budget_s), check=True).stdout
                    runs.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
                    with open(det_path) as f:  # the gradient vector, per-pattern lnL and compressed patterns of the timed protocol
                        runs[-1]["details"] = json.load(f)
                r = runs[0]
                compressed = sum(x["patterns"] for x in runs) / len(runs)
                t_eval = sum(x["grad_ms_per_eval"] for x in runs) / len(runs) / 1e3
                # physher has no threading inside one likelihood: the host's full capability is K independent
                # instances, each owning a share of the patterns -- timed here as K concurrent copies of the first shard
                multi = None
                try:
                    K = max(1, min(len(os.sched_getaffinity(0)), 16))  # a one-GPU share of the host
                    if K > 1:
                        it2 = max(2, iters // 2)
                        procs = [subprocess.Popen([driver, "bench", specs[0], str(it2), "1"], stdout=subprocess.PIPE, text=True)
                                 for _ in range(K)]
                        outs = [pr.communicate(timeout=max(180, 30 * budget_s))[0] for pr in procs]
                        ts = [json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1])["grad_ms_per_eval"] / 1e3 for o in outs]
                        multi = dict(cores=K, patterns_per_second=sum(r["patterns"] / t for t in ts), slowest_t_eval=max(ts), iters=it2)
                except Exception as exc:
                    print(f"[bench] concurrent CPU timing skipped ({exc})", file=sys.stderr)
                return dict(kind="reference", cores=1, t_eval=t_eval, patterns=compressed, iters=iters, lnl_ms=r["lnl_ms_per_eval"], multi=multi, lnl=r.get("lnl"),
                            shards=[dict(first_site=st, patterns=x["patterns"], seconds_per_eval=x["grad_ms_per_eval"] / 1e3, details=x.get("details"))
                                    for st, x in zip(starts, runs)],
                            sample=f"physher SSE path (oracle/_ref/ref_driver bench), {T} taxa, {len(runs)} shards of {sp} sites (first, middle and last of the "
                                   f"workload; {int(compressed)} patterns each on average), {iters} gradient evals per shard after 1 warm-up, one core, mean "
                                   f"seconds per evaluation scaled linearly to the full pattern count; whole 1e5-pattern shards are not timed: ~16 s per "
                                   f"evaluation and ~51 GB of host memory each")
            except Exception as exc:  # fall through to the port, but say why
                print(f"[bench] reference driver failed ({exc}); timing the CPU port instead", file=sys.stderr)
    from oracle import phyoracle as po
    ev, U, Ui = gtr_eigen()
    pb = po.Problem(tree.left, tree.right, tree.root, weights[:sp], ev, U, Ui, GTR_FREQS, cat_rates, np.full(len(cat_rates), 1.0 / len(cat_rates)),
                    tree.length, tip_states=np.ascontiguousarray(sub))
    t0 = time.perf_counter()
    pb.gradient()
    t_eval = time.perf_counter() - t0
    return dict(kind="port", cores=1, t_eval=t_eval, patterns=sp, iters=1, lnl_ms=None,
                sample=f"scalar CPU port (oracle/phyoracle.c), {T} taxa x {sp} patterns, 1 evaluation, scaled linearly")


def reference_node_map(tree, details):
    """bench node id -> the reference's node id, by clade (both trees come from the same Newick string; the reference numbers
    its nodes while parsing).  Returns (ref_of [N], the reference's right-of-root node: it reports 0 for that branch and the
    sum on its sibling, treelikelihood.c:3249-3255)."""
    tip_of = {name: i for i, name in enumerate(tree.names)}
    nodes = {nd["id"]: nd for nd in details["nodes"]}
    ref_clade = {}
    order, stack = [], [details["root"]]
    while stack:
        i = stack.pop()
        order.append(i)
        if nodes[i]["left"] >= 0:
            stack += [nodes[i]["left"], nodes[i]["right"]]
    for i in reversed(order):
        nd = nodes[i]
        ref_clade[i] = frozenset([tip_of[nd["name"]]]) if nd["left"] < 0 else ref_clade[nd["left"]] | ref_clade[nd["right"]]
    by_clade = {c: i for i, c in ref_clade.items()}
    N = tree.node_count
    clade = [None] * N
    order, stack = [], [int(tree.root)]
    while stack:
        i = stack.pop()
        order.append(i)
        if tree.left[i] >= 0:
            stack += [int(tree.left[i]), int(tree.right[i])]
    ref_of = np.full(N, -1, dtype=np.int64)
    for i in reversed(order):
        clade[i] = frozenset([i]) if tree.left[i] < 0 else clade[int(tree.left[i])] | clade[int(tree.right[i])]
        ref_of[i] = by_clade[clade[i]]
    return ref_of, nodes[details["root"]]["right"]


def reference_site_map(tree, details, states_slice):
    """site of the slice -> index of its column among the reference's compressed patterns (hash-table order)"""
    tip_of = {name: i for i, name in enumerate(tree.names)}
    rows = np.stack([np.frombuffer(r.encode(), dtype=np.uint8) - ord("0") for r in details["patterns"]])
    pat = np.empty_like(rows)
    for r, name in enumerate(details["taxa"]):
        pat[tip_of[name]] = rows[r]
    index = {pat[:, k].tobytes(): k for k in range(pat.shape[1])}
    cols = np.ascontiguousarray(states_slice.T)
    return np.array([index[cols[j].tobytes()] for j in range(cols.shape[0])], dtype=np.int64)


def check_against_reference(eng, tree, states, weights, shards, cat_rates, cat_props, plk_all):
    """Hold the TIMED engine (same object, same 1e6-pattern launches) against what the reference computed for the CPU baseline's
    shards (ref_driver bench ... details: examples/benchmarking.c:485-503 protocol):
      (i)  its per-pattern lnL of the last timed evaluation on each shard's sites (rtol 1e-11);
      (ii) its branch gradient of each shard -- pattern weights 1 on the shard's sites and 0 elsewhere, so that the full-size pass
           yields the shard's gradient -- against the reference's TreeLikelihood_gradient vector, node by node through the clade
           map (1e-9 * max(1, |g|_inf)), and the shard's lnL (1e-10 relative).  The reference's TREE_MODEL-only gradient folds
           the root frequencies into the uppers (treelikelihood.c:241, 2147-2153): the engine runs the same
           convention here (GRAD_FOLD_ROOT_FREQS).
    Returns the gradient_check object of the bench line."""
    from physher_amd.engine import GRAD_FOLD_ROOT_FREQS
    from physher_amd.sharding import epilogue
    N = tree.node_count
    out = {"shards": [], "per_pattern_lnl_rtol": 1e-11, "gradient_tol": "1e-9 * max(1, |g|_inf)", "lnl_rtol": 1e-10,
           "what": "the timed engine's per-pattern lnL and (pattern weights 1 on the shard, 0 elsewhere) its branch gradient, against the "
                   "reference's own values for the cpu_baseline shards (TreeLikelihood_gradient, TREE_MODEL flag; root-frequency folding as the reference does it)"}
    ok = True
    try:
        for sh in shards:
            det, st = sh["details"], sh["first_site"]
            sp = int(round(sum(det["weights"])))
            idx = reference_site_map(tree, det, states[:, st:st + sp])
            ref_plk = np.array(det["pattern_lk"])[idx]
            plk_err = float(np.max(np.abs(plk_all[st:st + sp] - ref_plk) / np.maximum(1e-300, np.abs(ref_plk))))
            w = np.zeros(states.shape[1])
            w[st:st + sp] = 1.0
            eng.set_pattern_weights(w)
            ref_of, ref_zero = reference_node_map(tree, det)
            # the reference's own branch lengths: reading the rooted Newick as an unrooted tree it moves the root's right branch onto
            # the left one (same lnL), and its folded gradient -- unlike the exact one -- depends on where that length sits
            ref_nodes = {nd["id"]: nd for nd in det["nodes"]}
            eng.set_branch_lengths(np.array([0.0 if n == tree.root else ref_nodes[int(ref_of[n])]["distance"] for n in range(N)]))
            lnl, cg = eng.gradient(GRAD_FOLD_ROOT_FREQS)
            _, g = epilogue(np.concatenate([[lnl], cg.reshape(-1)]), N, cat_rates, cat_props)
            ref_g = np.array(det["gradient"])
            keep = np.array([n != tree.root and ref_of[n] != ref_zero for n in range(N)])
            gerr = float(np.max(np.abs(g[keep] - ref_g[ref_of[keep]])))
            gmax = float(np.max(np.abs(ref_g[ref_of[keep]])))
            lerr = abs(lnl - det["lnl"]) / abs(det["lnl"])
            good = plk_err <= 1e-11 and gerr <= 1e-9 * max(1.0, gmax) and lerr <= 1e-10
            ok = ok and good
            out["shards"].append({"first_site": st, "sites": sp, "reference_patterns": len(det["weights"]), "per_pattern_lnl_max_rel_err": plk_err,
                                  "gradient_max_abs_err": gerr, "gradient_inf_norm": gmax, "branches_compared": int(keep.sum()),
                                  "lnl_rel_err": lerr, "ok": bool(good)})
    finally:
        eng.set_pattern_weights(weights)
        eng.set_branch_lengths(tree.length)
    out["ok"] = bool(ok)
    return out


def count_distinct_patterns(states):
    """Number of distinct columns of the synthetic alignment, by the product's own device pattern compressor
    (phyamd_compress_patterns: bit-exact with the reference's new_SitePattern).  None if it declines."""
    import ctypes as C
    from physher_amd import _lib
    lib = _lib.load()
    T, L = states.shape
    rows = (C.c_void_p * T)(*[states.ctypes.data + t * states.strides[0] for t in range(T)])
    n = C.c_int32()
    out = np.empty((T, L), dtype=np.uint8)
    w = np.empty(L, dtype=np.float64)
    rc = lib.phyamd_compress_patterns(-1, T, L, rows, None, C.byref(n), out.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
    return int(n.value) if rc == 0 else None


def named_model(config, seed):
    """(frequencies, eval, evec, ivec, label) of the model BASELINE.json names for a configuration, from the host model library."""
    from physher_amd import _phycpp_amd as pc
    if config in ("cfg2", "cfg5"):
        ev, U, Ui = gtr_eigen()
        return np.array(GTR_FREQS), ev, U, Ui, "GTR"
    if config == "cfg3":  # WAG with its own equilibrium frequencies (wag.c:38-44)
        m = pc.substitution_model("WAG")
        return m["frequencies"], m["eval"], m["evec"], m["ivec"], "WAG"
    m = pc.substitution_model("MG94", [2.0, 1.0, 0.5])  # kappa 2, alpha 1, beta 0.5, uniform codon frequencies
    return m["frequencies"], m["eval"], m["evec"], m["ivec"], "MG94"


def category_rates(C):
    if C == 4:
        r = np.array(GAMMA4_RATES_05)
    else:
        r = np.linspace(0.2, 1.8, C) if C > 1 else np.ones(1)
    return r / r.mean(), np.full(C, 1.0 / C)


def generic_counters(path):
    """profiles/generic_latest.json (profiles/collect_generic.sh) if it describes the loaded build of the library, else (None, why)"""
    import hashlib
    if not os.path.exists(path):
        return None, "no counter profile (profiles/generic_latest.json)"
    try:
        with open(path) as f:
            gj = json.load(f)
        with open(os.path.join(ROOT, "physher_amd", "libphysher_amd.so"), "rb") as f:
            if hashlib.sha256(f.read()).hexdigest() != gj.get("library_sha256"):
                return None, f"the counter profile ({gj.get('tag')}) was collected with another build of libphysher_amd.so: re-run profiles/collect_generic.sh"
        return gj, None
    except Exception as exc:
        return None, f"counter profile unreadable: {exc}"


def mfma_object(config, T, P, C, S, kernel_s, generic_json):
    """The matrix-core side of a 20- / 61-state workload.  Counter-based (sha-tied profile): busy_frac = SQ_VALU_MFMA_BUSY_CYCLES per
    evaluation / 1024 SIMDs / (kernel time x 2.4 GHz: the nominal clock -- under sustained f64 MFMA load the part holds ~2.05 GHz, so
    the pipe's true occupancy is up to 17 % higher); issued flops from the instruction counters (tip children skip the MFMA: a
    gathered column / row sums); HBM bytes from FETCH_SIZE / WRITE_SIZE.  vs_algorithmic_flops is SURVEY 8(d)'s per-node flop count
    over the same time: a ratio against the spec-sheet peak, NOT a utilisation (it counts tip products the kernels never issue)."""
    fl = algorithmic_flops(T, P, C, S)
    tf = fl / kernel_s / 1e12
    out = {"vs_algorithmic_flops": {"achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "ratio": tf / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_eval": fl},
           "busy_frac": None, "issued_flops_per_eval": None, "issued_tflops": None, "hbm": None, "pmc_profile": None, "pmc_note": None}
    gj, why = generic_counters(generic_json)
    if gj is None or not gj.get(config):
        out["pmc_note"] = why or f"the counter profile holds no {config}"
        return out
    g = gj[config]
    out["pmc_profile"] = gj.get("tag")
    if g.get("mfma_busy_cycles_per_eval"):
        out["busy_frac"] = g["mfma_busy_cycles_per_eval"] / 1024.0 / (kernel_s * 2.4e9)
    if g.get("mfma_f64_mops_per_eval"):  # one MOP = 512 flops (a v_mfma_f64_4x4x4_4b; a 16x16x4 is four)
        out["issued_flops_per_eval"] = 512.0 * g["mfma_f64_mops_per_eval"]
        out["issued_tflops"] = out["issued_flops_per_eval"] / kernel_s / 1e12
    if g.get("hbm_bytes_per_eval"):
        gbs = g["hbm_bytes_per_eval"] / kernel_s / 1e9
        out["hbm"] = {"bytes_per_eval": g["hbm_bytes_per_eval"], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    return out


def time_other_config(config, device, stream, seed, evals, generic_json=None):
    """One of BASELINE.json's smaller configurations (cfg2..cfg4) on the same GPU: ms per lnL + gradient evaluation with full
    recompute, timed around synchronous calls (result on the host), plus the engine's HIP-event split."""
    import torch
    from physher_amd import synth
    from physher_amd.engine import RESCALE_AUTO, Engine
    wl = WORKLOADS[config]
    T, P, C, S = wl["taxa"], wl["patterns"], wl["categories"], wl["states"]
    tree = synth.random_tree(T, np.random.default_rng(seed))
    states = np.ascontiguousarray(evolve_on_device(tree, P, seed * 100003, device, S).cpu().numpy())
    weights = np.random.default_rng(seed + 17).integers(1, 4, size=P).astype(np.float64)
    freqs, ev, U, Ui, model = named_model(config, seed)
    rates, props = category_rates(C)
    eng = Engine(T, P, S, C, device=device.index, rescale=RESCALE_AUTO, stream=stream.cuda_stream)
    eng.set_topology(tree.left, tree.right, tree.root)
    eng.set_branch_lengths(tree.length)
    eng.set_eigen(ev, U, Ui)
    eng.set_frequencies(freqs)
    eng.set_category_rates(rates, props)
    eng.set_pattern_weights(weights)
    for t in range(T):
        eng.set_tip_states(t, states[t])
    eng.set_profiling(True)
    for _ in range(2):
        eng.set_branch_lengths(tree.length)
        lnl, _ = eng.gradient()
    torch.cuda.synchronize(device)
    lower = upper = 0.0
    t0 = time.perf_counter()
    for _ in range(evals):
        eng.set_branch_lengths(tree.length)  # full recompute (benchmarking.c:498-500)
        lnl, _ = eng.gradient()
        pr = eng.profile()
        lower += pr["lower_ms"]
        upper += pr["upper_ms"]
    dt = (time.perf_counter() - t0) / evals
    out = {"workload": f"{model}+G{C} ({wl['name']}), {T} taxa x {P} patterns x {S} states x {C} categories (BASELINE configs[{int(config[3]) - 1}])",
           "evals_per_s": 1.0 / dt, "ms_per_eval": 1e3 * dt, "lower_ms": lower / evals, "upper_ms": upper / evals, "lnL": lnl,
           "rescaling": eng.rescaling, "evals": evals}
    if S != 4:
        out["mfma"] = mfma_object(config, T, P, C, S, (lower + upper) / evals * 1e-3, generic_json or os.path.join(ROOT, "profiles", "generic_latest.json"))
    eng.close()
    del states
    torch.cuda.empty_cache()
    return out


def drop_in_measure(config, device, seed, iters, tipstates=1):
    """What a user of the REFERENCE gets from the binding (INTEGRATION.md seam A): physher's own object graph, built by its own JSON
    parser from a FASTA file and a Newick string, with integration/physher_device.c in front of libphyc -- timed by the protocol of
    examples/benchmarking.c:498-503 (oracle/_ref/ref_driver bench: invalidate every node and the eigen system, then
    TreeLikelihood_gradient with the TREE_MODEL flag), in a process of its own.  tipstates = 0 is the wrapper's default tip mode
    (src/phycpp/physher.hpp:366-367: use_tip_states = false, 0/1 partials per tip).  Also reported: the process's peak resident
    host memory (wait4's ru_maxrss) and its wall time, which is mostly construction (FASTA parsing, pattern compression, the
    object graph, tips to the device).  Test infrastructure built where the reference tree exists (oracle/_ref travels to the GPU
    box prebuilt); None where it is absent."""
    from physher_amd import synth
    refdir = os.path.join(ROOT, "oracle", "_ref")
    driver, shim = os.path.join(refdir, "ref_driver"), os.path.join(refdir, "libphysher_device.so")
    if not (os.path.exists(driver) and os.path.exists(shim)):
        return None
    wl = WORKLOADS[config]
    T, P, C, S = wl["taxa"], wl["patterns"], wl["categories"], wl["states"]
    if S != 4:
        return None
    tree = synth.random_tree(T, np.random.default_rng(seed))
    BLK = 125000
    with tempfile.TemporaryDirectory() as d:
        rows = [[] for _ in range(T)]
        for b0 in range(0, P, BLK):  # block-wise: the same sites as the main workload (which generates them in blocks of 125000)
            st = evolve_on_device(tree, min(BLK, P - b0), seed * 100003 + (b0 if config == "cfg5" else 0), device, S).cpu().numpy()
            table = np.frombuffer(b"ACGT", dtype="S1")
            for t in range(T):
                rows[t].append(table[st[t]].tobytes())
            del st
        with open(os.path.join(d, "aln.fa"), "wb") as f:
            for t in range(T):
                f.write(b">" + tree.names[t].encode() + b"\n" + b"".join(rows[t]) + b"\n")
        del rows
        with open(os.path.join(d, "tree.nwk"), "w") as f:
            f.write(tree.newick() + "\n")
        with open(os.path.join(d, "spec.txt"), "w") as f:
            f.write(f"fasta {d}/aln.fa\nnewick {d}/tree.nwk\ndatatype nucleotide\nmodel gtr\n"
                    f"rates {','.join(map(str, GTR_RATES[:5]))}\nfreqs {','.join(map(str, GTR_FREQS))}\n"
                    f"categories {C}\nalpha {ALPHA}\ntipstates {tipstates}\nsse 1\n")
        env = dict(os.environ)
        env["LD_PRELOAD"] = shim
        env["PHYSHER_DEVICE"] = "1"
        env["PHYSHER_DEVICE_VERBOSE"] = "1"
        t0 = time.perf_counter()
        with open(os.path.join(d, "out.txt"), "w") as fo, open(os.path.join(d, "err.txt"), "w") as fe:
            proc = subprocess.Popen([driver, "bench", os.path.join(d, "spec.txt"), str(iters), "2"], stdout=fo, stderr=fe, env=env)
            _, status, ru = os.wait4(proc.pid, 0)
            proc.returncode = os.waitstatus_to_exitcode(status)
        wall = time.perf_counter() - t0
        out_text, err_text = open(os.path.join(d, "out.txt")).read(), open(os.path.join(d, "err.txt")).read()
    if proc.returncode != 0:
        return {"error": (err_text or out_text)[-400:], "tipstates": tipstates}
    r = json.loads([ln for ln in out_text.splitlines() if ln.startswith("{")][-1])
    timed_s = iters * (r["grad_ms_per_eval"] + r["lnl_ms_per_eval"]) / 1e3
    return {"workload": f"{wl['name']}, {T} taxa x {P} sites -> {r['patterns']} patterns x {S} states x {C} categories (BASELINE configs[{int(config[3]) - 1}]), "
                        f"physher's own JSON model + SingleTreeLikelihood + TreeLikelihood_gradient (TREE_MODEL) with the device binding preloaded",
            "tipstates": tipstates,
            "evals_per_s": 1e3 / r["grad_ms_per_eval"], "ms_per_eval": r["grad_ms_per_eval"], "lnl_only_ms_per_eval": r["lnl_ms_per_eval"], "lnL": r["lnl"],
            "patterns": r["patterns"], "iters": iters, "process_wall_s": wall, "construction_s": max(0.0, wall - timed_s),
            "host_rss_gb": r["peak_rss_kb"] * 1024.0 / 1e9,  # VmHWM of the driver process itself (wait4's ru_maxrss includes what this process held at fork)
            "device_work": (err_text.strip().splitlines() or [""])[-1][-160:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="cfg5", help="BASELINE.json configs[1..4]; default = the metric's shape")
    ap.add_argument("--taxa", type=int, default=None)
    ap.add_argument("--patterns", type=int, default=None)
    ap.add_argument("--categories", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample-patterns", type=int, default=8000)
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg2..cfg4 and lnL-only secondary measurements (N = 1 only)")
    ap.add_argument("--no-distinct-check", action="store_true", help="skip counting the distinct columns of the synthetic alignment")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the drop_in measurement (the reference's object graph through the binding)")
    ap.add_argument("--drop-in-config", choices=("cfg2", "cfg5"), default="cfg2", help="workload of the drop_in measurement (cfg5: a 1 GB FASTA file, minutes)")
    ap.add_argument("--rescale", choices=("auto", "always", "never"), default="auto",
                    help="rescaling policy (auto = the reference's lazy switch; always: NOT the headline configuration, measures the rescaled kernels)")
    ap.add_argument("--max-device-gb", type=float, default=0.0,
                    help="NOT the headline configuration: cap the engine's device memory; below the working set the patterns are "
                         "processed in tiles through one set of partial arrays (per-kernel roofline figures then describe the last tile)")
    ap.add_argument("--subst-gradient", action="store_true",
                    help="NOT the headline metric: each step also yields d lnL / d(5 GTR rates, 4 frequencies) (SURVEY 8f.1) in the same two passes")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs inside ONE process (phyamd_create_sharded: one engine, stream and host thread per GPU, results added pairwise on the "
                         "host) instead of one process per GPU + RCCL; launch WITHOUT torch.distributed.run.  PHYAMD_BENCH_DEVICE_IDS=0,0,.. rehearses "
                         "it on fewer GPUs")
    ap.add_argument("--deterministic-sum", action="store_true",
                    help="N > 1 ranks: one all-gather + the pairwise sum of physher_amd/sharding.py::tree_sum instead of one all-reduce: bit for bit the one-GPU result")
    ap.add_argument("--generic-json", default=os.path.join(ROOT, "profiles", "generic_latest.json"),
                    help="per-evaluation MFMA / HBM counters of the cfg3 / cfg4 workloads from profiles/collect_generic.sh (used only for the same build of the library)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 under torch.distributed.run: create the RCCL process group and run the all-reduce in a world of one "
                         "(the code path of N > 1 on a one-GPU box; the timed loop then includes the collective)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="per-launch HBM bytes of the dominant kernel from a rocprofv3 --pmc run of this workload")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    single = args.single_process and args.gpus > 1
    if single and world != 1:
        raise SystemExit("--single-process runs every GPU from ONE process: launch it without torch.distributed.run")
    if not single and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    # PHYAMD_BENCH_REHEARSAL=1: every rank uses cuda:0 and a gloo group -- to rehearse the N > 1 code path on a one-GPU box
    rehearsal = os.environ.get("PHYAMD_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collective
    if args.force_collective and "RANK" not in os.environ:
        raise SystemExit("--force-collective needs a rendezvous: launch with torch.distributed.run --nproc-per-node 1")
    if collective:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from physher_amd import synth
    from physher_amd.engine import RESCALE_ALWAYS, RESCALE_AUTO, RESCALE_NEVER, Engine
    from physher_amd.sharding import ShardedLikelihood, epilogue, reduction_levels, shard_range

    wl = WORKLOADS[args.config]
    T = args.taxa or wl["taxa"]
    P = args.patterns or wl["patterns"]
    C = args.categories or wl["categories"]
    S = wl["states"]
    rng = np.random.default_rng(args.seed)
    tree = synth.random_tree(T, rng)  # identical on every rank
    # contiguous shard of the pattern list; generated block-wise so the data do not depend on N
    lo, hi = shard_range(P, rank, world)
    BLK = int(os.environ.get("PHYAMD_BENCH_BLOCK", "125000"))  # profiling runs use one big block: fewer generator kernels
    chunks = []
    for b0 in range(0, P, BLK):
        b1 = min(P, b0 + BLK)
        if b1 <= lo or b0 >= hi:
            continue
        blk = evolve_on_device(tree, b1 - b0, args.seed * 100003 + b0, device, S)
        chunks.append(blk[:, max(lo, b0) - b0: min(hi, b1) - b0].cpu().numpy())
    states = np.ascontiguousarray(np.concatenate(chunks, axis=1))
    del chunks
    torch.cuda.empty_cache()
    Pl = hi - lo
    wrng = np.random.default_rng(args.seed + 17)
    weights_all = wrng.integers(1, 4, size=P).astype(np.float64)
    weights = weights_all[lo:hi]
    cat_rates, cat_props = category_rates(C)
    freqs, ev, U, Ui, model_name = named_model(args.config, args.seed)
    distinct_note = "distinctness not checked on this rank"
    if world == 1 and not single and not args.no_distinct_check:
        nd = count_distinct_patterns(states)
        distinct_note = "device pattern compressor declined: not checked" if nd is None else f"{nd} of the {P} sites are distinct columns"

    # ONE stream for the engine's kernels and for everything torch does with their output (device-to-host copies, the
    # RCCL all-reduce): torch's default stream has handle 0, which the engine would take as "create your own stream", and a
    # stream of its own is not ordered with torch's work -- the reduction could read the result vector before it is written
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    rescale_policy = {"auto": RESCALE_AUTO, "always": RESCALE_ALWAYS, "never": RESCALE_NEVER}[args.rescale]
    if single:  # one handle = a group of shards, one per GPU (the same ordinal may repeat: a rehearsal on fewer GPUs)
        ids = os.environ.get("PHYAMD_BENCH_DEVICE_IDS")
        devices = [int(x) for x in ids.split(",")] if ids else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit(f"PHYAMD_BENCH_DEVICE_IDS lists {len(devices)} devices, --gpus is {args.gpus}")
        eng = Engine(T, Pl, S, C, rescale=rescale_policy, devices=devices)
    else:
        eng = Engine(T, Pl, S, C, device=local_rank, rescale=rescale_policy, stream=stream.cuda_stream, max_device_bytes=int(args.max_device_gb * 1e9))
        if world > 1:
            eng.set_reduction_levels(reduction_levels(P, world))  # this rank's patterns are a subtree of the one-GPU summation
    eng.set_topology(tree.left, tree.right, tree.root)
    eng.set_branch_lengths(tree.length)
    eng.set_eigen(ev, U, Ui)
    eng.set_frequencies(freqs)
    eng.set_category_rates(cat_rates, cat_props)
    eng.set_pattern_weights(weights)
    for t in range(T):
        eng.set_tip_states(t, states[t])
    eng.set_profiling(True)

    N = 2 * T - 1
    n_tail = 0
    if args.subst_gradient:
        if S != 4:
            raise SystemExit("--subst-gradient: 4-state workloads only")
        from physher_amd import _phycpp_amd as pc  # host model code: dQ/dtheta of GTR (5 rates relative to GT, then 4 frequencies)
        dQ = pc.GTRInterface(list(GTR_RATES[:5]), list(GTR_FREQS)).rate_matrix_derivatives()
        eng.set_rate_matrix_derivatives(dQ)
        n_tail = len(dQ) + S
    result = torch.zeros(1 + N * C + n_tail, dtype=torch.float64, device=device)

    def evaluate_shard(out):
        eng.set_branch_lengths(tree.length)  # invalidates every P(t): full recompute (benchmarking.c:498-500)
        if args.subst_gradient:
            eng.parameter_gradient_device(out.data_ptr())  # [lnL | g[node][cat] | 9 parameter sums | root frequency term]
        else:
            eng.gradient_device(out.data_ptr())  # HIP kernels on torch's current stream; [lnL, g[node][cat]] stays on the device

    if single:
        def step():  # every shard evaluates on its own GPU at once; the 64 KB results are added pairwise on the host (phyamd_abi.inc)
            eng.set_branch_lengths(tree.length)
            lnl_, cg_ = eng.gradient()
            return epilogue(np.concatenate([[lnl_], cg_.reshape(-1)]), N, cat_rates, cat_props)
    else:
        timers = {}
        step = ShardedLikelihood(evaluate_shard, N, cat_rates, cat_props, world, result, via_host=rehearsal, tail=n_tail,
                                 deterministic=args.deterministic_sum, force_collective=args.force_collective,
                                 timers=timers)  # + one RCCL all-reduce (or all-gather) + host epilogue

    def fence():
        torch.cuda.synchronize(device)
        if collective:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        lnl, bg = step()[:2]
    fence()
    prof = dict(lower_ms=0.0, upper_ms=0.0, matrices_ms=0.0, reduce_ms=0.0)
    if not single:
        timers.clear()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        lnl, bg = step()[:2]  # (ends in a device-to-host copy of the result: nothing is left in flight)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        p = eng.profile()  # HIP-event times of this evaluation, recorded on the engine's stream
        for k in prof:
            prof[k] += p[k]
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = [1e3 * elapsed / args.steps]
    if collective:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        every = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)  # every rank's own clock over the same K steps: the line reports the slowest (and the spread)
        rank_ms = [1e3 * float(x.item()) / args.steps for x in every]
        elapsed = max(float(x.item()) for x in every)
    plk_timed = eng.pattern_log_likelihoods() if (world == 1 and not single and not args.no_cpu_baseline) else None  # of the last timed evaluation
    for k in prof:
        prof[k] /= max(1, args.steps)
    p = eng.profile()

    failed = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.steps / elapsed
        lower_b, upper_b = algorithmic_bytes(T, Pl, C, S)  # this rank's shard
        launches = max(1, p["upper_launches"])
        # 4 states: the tree-walk kernels (each pass is two launches: cut subtrees and the top of the tree); otherwise one launch
        # per tree level
        kern = ("4_walk" if launches <= 2 and T > 4 else "4") if S == 4 else "_gen"
        upper_s = prof["upper_ms"] * 1e-3
        three_pass = upper_b / upper_s / 1e9 if upper_s > 0 else None
        # PMC figures of the same workload (profiles/collect.sh -> profiles/traffic_latest.json): HBM bytes and vector-ALU
        # wave-instructions per launch of the dominant kernel.  They describe THIS shape only: another shape reports null.
        traffic = valu_insts = None
        pmc_source = pmc_note = None
        tj = {}
        if os.path.exists(args.traffic_json):
            try:
                import hashlib
                with open(args.traffic_json) as f:
                    tj = json.load(f)
                with open(os.path.join(ROOT, "physher_amd", "libphysher_amd.so"), "rb") as f:
                    lib_hash = hashlib.sha256(f.read()).hexdigest()
                same_shape = tj.get("states", 4) == S and tj.get("taxa") == T and tj.get("patterns") == Pl and tj.get("categories") == C and tj.get("launches_per_eval") == launches
                if not same_shape:
                    pmc_note = "the PMC profile describes another workload shape"
                elif tj.get("library_sha256") != lib_hash:
                    pmc_note = f"the PMC profile ({tj.get('tag')}) was collected with another build of libphysher_amd.so: re-run profiles/collect.sh"
                else:
                    traffic = tj["upper_bytes_per_launch"]
                    valu_insts = tj.get("upper_valu_insts_per_launch")
                    pmc_source = tj.get("tag")
            except Exception as exc:
                traffic = None
                pmc_note = f"PMC profile unreadable: {exc}"
        else:
            pmc_note = "no PMC profile (profiles/traffic_latest.json)"
        achieved = traffic * launches / upper_s / 1e9 if traffic is not None and upper_s > 0 else None
        valu = None
        if valu_insts is not None and upper_s > 0:
            # a wave instruction occupies its SIMD's 16-lane vector ALU for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
            valu_s = valu_insts * launches * 4.0 / (1024 * 2.4e9)
            valu = {"wave_instructions_per_launch": valu_insts, "busy_frac": valu_s / upper_s,
                    "note": "VALU wave-instructions (PMC SQ_INSTS_VALU) x 4 cycles / (1024 SIMDs x 2.4 GHz) / kernel time"}
            # every instruction class counts against a SIMD's issue rate (both walks are bound by it): all wave-instructions
            # of the launch against the SIMD-cycles it took
            every = sum(tj.get(f"upper_{k}_insts_per_launch") or 0 for k in ("valu", "salu", "smem", "lds"))
            if every > valu_insts:
                valu["all_wave_instructions_per_launch"] = every
                valu["simd_cycles_per_instruction"] = upper_s / launches * 1024 * 2.4e9 / every
                # a SIMD has one issue slot every 4 cycles; no variant of either walk has been seen to issue faster than one
                # instruction per slot, so this is the fraction of that (empirical) roof
                valu["issue_slot_frac"] = 4.0 / valu["simd_cycles_per_instruction"]
        upper_kernel = ((tj.get("kernels") or {}).get("upper") or {}).get("kernel") if traffic is not None else None
        if not upper_kernel:  # (the streamed walk runs plain and rescaled 4-state evaluations with <= 4 categories and one pattern tile)
            upper_kernel = "k_upper4_stream" if kern == "4_walk" and C <= 4 and p["tiles"] == 1 and not os.environ.get("PHYAMD_WALK_STREAM") == "0" else f"k_upper{kern}"
        lower_kernel = ((tj.get("kernels") or {}).get("lower") or {}).get("kernel") if traffic is not None else None
        if not lower_kernel:  # (the streamed post-order walk runs plain 4-state evaluations with <= 4 categories)
            lower_kernel = ("k_lower4_stream" if kern == "4_walk" and (C <= 4 or os.environ.get("PHYAMD_SCALE_EXP2") != "0") and os.environ.get("PHYAMD_LOWER_STREAM") != "0" else
                            (f"k_lower{kern}" if p["lower_launches"] <= 2 or S != 4 else "k_lower4"))
        workload_label = f"{T}-taxon {wl['name'].split(' (')[0]} fp64, {P:.0e} site patterns".replace("e+0", "e").replace("e+", "e")
        out = {
            "metric": ("lnL+gradient evals/sec, 1000-taxon GTR+G4 fp64, 1e6 site patterns" if (args.config == "cfg5" and T == 1000 and P == 1_000_000 and C == 4)
                       else f"lnL+gradient evals/sec, {workload_label} [NOT the headline shape]")
                      + (" [+ 9 substitution-parameter gradients per eval: NOT the headline metric]" if args.subst_gradient else ""),
            "value": value,
            "unit": "evals/s",
            "n_gpus": args.gpus if single else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            # this rank's wall clock per step (host loop included): the box-to-box spread is +-5 %, the step-to-step spread is here
            "step_ms": {"min": float(np.min(step_ms)), "median": float(np.median(step_ms)), "mean": float(np.mean(step_ms)), "max": float(np.max(step_ms))},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{wl['name']}, {T} taxa x {P} patterns x {S} states x {C} categories, unrooted, full recompute per eval "
                                   f"(BASELINE configs[{int(args.config[3]) - 1}] shape; {Pl} patterns on this rank); "
                                   f"patterns = sites evolved down the tree on the GPU with integer weights 1..3 standing in for multiplicities, "
                                   f"NOT de-duplicated ({distinct_note})",
                       "parallelism": (f"{args.gpus} GPUs in one process (phyamd_create_sharded), host sum" if single else
                                       f"{world} process(es), one per GPU" + ("" if world == 1 else (", one all-gather + pairwise sum" if args.deterministic_sum else ", one RCCL all-reduce"))),
                       "taxa": T, "patterns": P, "categories": C, "states": S, "patterns_per_gpu": Pl // args.gpus if single else Pl, "lnL": lnl,
                       "rescaling": eng.rescaling, "device_bytes": p["device_bytes"], "tiles": p["tiles"],
                       # where a step's time goes besides the kernels: the collective as the engine's stream saw it (two events around
                       # it), the O(N C) host epilogue, and every rank's own ms per step
                       "collective": None if single else {
                           "backend": (dist.get_backend() if collective else None), "world": (dist.get_world_size() if collective else 1),
                           "kind": None if not collective else ("all_gather + pairwise sum" if args.deterministic_sum else "all_reduce(SUM)"),
                           "bytes": int(result.numel() * 8),
                           "all_reduce_us": (float(np.mean(timers["all_reduce_us"])) if timers.get("all_reduce_us") else None),
                           "host_epilogue_us": (float(np.mean(timers["host_epilogue_us"])) if timers.get("host_epilogue_us") else None),
                           "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)}}},
            "roofline": {"bound": "hbm", "kernel": f"{upper_kernel} (pre-order pass + fused branch gradient)",
                         # achieved / frac: MEASURED HBM bytes of the launch (PMC, per the guide's FETCH_SIZE / WRITE_SIZE recipe)
                         # over the live HIP-event time of the same kernel; null when no PMC file matches this shape
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "pmc_profile": pmc_source, "pmc_note": pmc_note, "kernel_resources": tj.get("kernels") if traffic is not None else None,
                         "valu": valu,
                         # SURVEY 8(d)'s algorithmic flops of the WHOLE evaluation over the step time, against the fp64 vector peak
                         # (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz): the kernels recompute fringe / DEEP nodes, so they issue more
                         "fp64_vector": {"achieved": algorithmic_flops(T, Pl, C, S) / (ms_per_step * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": algorithmic_flops(T, Pl, C, S) / (ms_per_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
                         # the reference's three-pass algorithmic bytes over the same time: the fused kernels move about
                         # a tenth of them, so this ratio exceeds 1 and is NOT a fraction of any roof
                         "vs_three_pass": None if three_pass is None else
                                          {"GB/s": three_pass, "ratio_to_hbm_peak": three_pass / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": upper_b / launches},
                         "launches_per_eval": launches,
                         "avg_launch_ms": prof["upper_ms"] / launches,
                         "lower_kernel": {"kernel": lower_kernel,
                                          "launches_per_eval": p["lower_launches"], "ms_per_eval": prof["lower_ms"]},
                         "ms_per_eval": {k: prof[k] for k in prof}},
        }
        if traffic is not None and tj.get("lower_bytes_per_launch") and prof["lower_ms"] > 0 and p["lower_launches"] == tj.get("lower_launches_per_eval", 1):
            lb = tj["lower_bytes_per_launch"] * p["lower_launches"] / (prof["lower_ms"] * 1e-3) / 1e9
            out["roofline"]["lower_kernel"].update({"achieved": lb, "frac": lb / HBM_PEAK_GBS, "traffic": tj["lower_bytes_per_launch"]})
        if S != 4 and prof["upper_ms"] > 0:  # the 20-/61-state contraction runs on the fp64 matrix cores: report that side of the roofline too
            out["roofline"]["mfma"] = mfma_object(args.config, T, Pl, C, S, (prof["lower_ms"] + prof["upper_ms"]) * 1e-3, args.generic_json)
            if S > 20:
                out["roofline"]["bound"] = "mfma"
        if p["tiles"] > 1:  # the engine's per-kernel timings describe the last tile only: no roofline claim for a tiled run
            out["roofline"] = {"bound": "hbm", "kernel": out["roofline"]["kernel"], "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                               "traffic": None, "note": f"patterns processed in {p['tiles']} tiles (--max-device-gb): per-launch figures not comparable"}
        if world == 1 and not single and not args.no_cpu_baseline and S == 4:
            cb = cpu_baseline(tree, states, weights, cat_rates, args.cpu_sample_patterns, args.cpu_budget_s)
            scaled = cb["t_eval"] * (P / cb["patterns"])
            out["cpu_baseline"] = {"value": 1.0 / scaled, "unit": "evals/s", "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
                                   "sample_seconds_per_eval": cb["t_eval"], "sample_patterns": cb["patterns"]}
            if cb.get("shards"):
                out["cpu_baseline"]["shards"] = [{k: v for k, v in sh.items() if k != "details"} for sh in cb["shards"]]
                if all(sh.get("details") for sh in cb["shards"]) and p["tiles"] == 1 and not args.subst_gradient:
                    gc = check_against_reference(eng, tree, states, weights, cb["shards"], cat_rates, cat_props, plk_timed)
                    out["cpu_baseline"]["gradient_check"] = gc
                    if not gc["ok"]:
                        failed = "the timed engine disagrees with the reference (cpu_baseline.gradient_check)"
            if cb.get("lnl") is not None:
                # the same sample (the first sites of the workload, every site with weight 1) through the engine: the reference's lnL of
                # the driver's own run is the yardstick (relative difference; the parity tests hold 1e-10)
                sp = min(args.cpu_sample_patterns, states.shape[1])
                with Engine(T, sp, S, C, device=local_rank, rescale=RESCALE_AUTO, stream=stream.cuda_stream) as chk:
                    chk.set_topology(tree.left, tree.right, tree.root)
                    chk.set_branch_lengths(tree.length)
                    chk.set_eigen(ev, U, Ui)
                    chk.set_frequencies(freqs)
                    chk.set_category_rates(cat_rates, cat_props)
                    chk.set_pattern_weights(np.ones(sp))
                    for t in range(T):
                        chk.set_tip_states(t, np.ascontiguousarray(states[t, :sp]))
                    lnl_gpu = chk.log_likelihood()
                rel = abs(lnl_gpu - cb["lnl"]) / abs(cb["lnl"])
                out["cpu_baseline"]["lnl_check"] = {"lnL_reference": cb["lnl"], "lnL_engine": lnl_gpu, "relative_difference": rel, "within_1e-10": bool(rel <= 1e-10)}
                if rel > 1e-10:
                    print(f"[bench] WARNING: the engine's lnL of the CPU baseline's first shard differs from the reference's by {rel:.3e} (relative)", file=sys.stderr)
            if cb.get("multi"):  # every host core busy with its own pattern shard (the reference's only way to use them)
                m = cb["multi"]
                out["cpu_baseline"]["all_cores"] = {"value": m["patterns_per_second"] / P, "unit": "evals/s", "cores": m["cores"],
                                                    "sample": f"{m['cores']} concurrent single-thread instances of the same sample, {m['iters']} gradient evals each; "
                                                              f"aggregate pattern throughput scaled to the full pattern count"}
        if (world == 1 and not single and not args.no_other_configs and args.config == "cfg5" and args.taxa is None and args.patterns is None and p["tiles"] == 1
                and not args.subst_gradient):
            # SURVEY 8d secondary metric (examples/benchmarking.c:466-471): lnL only, full recompute, result on the host
            evals = max(3, args.steps)
            for _ in range(2):
                eng.set_branch_lengths(tree.length)
                eng.log_likelihood()
            t1 = time.perf_counter()
            for _ in range(evals):
                eng.set_branch_lengths(tree.length)
                lnl_only = eng.log_likelihood()
            dt = (time.perf_counter() - t1) / evals
            out["lnl_only"] = {"evals_per_s": 1.0 / dt, "ms_per_eval": 1e3 * dt, "lnL": lnl_only, "evals": evals,
                               "note": "post-order pass + root integration only, same workload and protocol"}
            out["other_configs"] = []
            for cfg in ("cfg2", "cfg3", "cfg4"):
                try:
                    out["other_configs"].append(time_other_config(cfg, device, stream, args.seed, 10, args.generic_json))
                except Exception as exc:  # a secondary measurement must not lose the headline line
                    out["other_configs"].append({"workload": cfg, "error": str(exc)})
        if world == 1 and not single and not args.no_drop_in and args.config == "cfg5" and args.taxa is None and args.patterns is None and p["tiles"] == 1 and not args.subst_gradient:
            try:
                di = drop_in_measure(args.drop_in_config, device, args.seed, 20 if args.drop_in_config == "cfg2" else 5, 1)
                if di is not None and "evals_per_s" in di:  # ... and in the wrapper's default tip mode (0/1 partials per tip)
                    d0 = drop_in_measure(args.drop_in_config, device, args.seed, 20 if args.drop_in_config == "cfg2" else 5, 0)
                    di["tip_partials_mode"] = {k: d0.get(k) for k in ("tipstates", "evals_per_s", "ms_per_eval", "lnL", "process_wall_s", "construction_s",
                                                                       "host_rss_gb", "error") if k in d0}
            except Exception as exc:
                di = {"error": str(exc)}
            if di is not None:
                if "evals_per_s" in di:  # against the raw engine on the same workload (same tree, same sites)
                    if args.drop_in_config == "cfg5":
                        raw = value
                    else:
                        same = [o for o in out.get("other_configs", []) if "configs[1]" in o.get("workload", "") and "evals_per_s" in o]
                        raw = same[0]["evals_per_s"] if same else None
                    di["engine_evals_per_s"] = raw
                    di["ratio_to_engine"] = None if not raw else di["evals_per_s"] / raw
                out["drop_in"] = di
        print(json.dumps(out), flush=True)
    eng.close()
    if collective:
        dist.destroy_process_group()
    if failed:
        print(f"[bench] FAILED: {failed}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
