"""ad-hoc: time of a single-branch incremental update at the cfg5 shape (one GPU)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from physher_amd import synth
from physher_amd.engine import Engine, RESCALE_AUTO
T, P, C = 1000, 1_000_000, 4
rng = np.random.default_rng(1)
tree = synth.random_tree(T, rng)
dev = torch.device("cuda", 0)
states = bench.evolve_on_device(tree, P, 5, dev, 4).cpu().numpy()
ev, U, Ui = bench.gtr_eigen()
rates = np.array(bench.GAMMA4_RATES_05); rates /= rates.mean()
e = Engine(T, P, 4, C, rescale=RESCALE_AUTO)
e.set_topology(tree.left, tree.right, tree.root); e.set_branch_lengths(tree.length); e.set_eigen(ev, U, Ui)
e.set_frequencies(np.array(bench.GTR_FREQS)); e.set_category_rates(rates, np.full(C, 0.25)); e.set_pattern_weights(np.ones(P))
for t in range(T): e.set_tip_states(t, states[t])
e.set_profiling(True)
l0 = e.log_likelihood(); l0 = e.log_likelihood()
t0 = time.perf_counter(); e.update_all_nodes(); lf = e.log_likelihood(); tfull = time.perf_counter() - t0
out = {"full_ms": tfull * 1e3, "lnL": lf, "single": []}
bl = tree.length.copy()
for n in rng.choice([i for i in range(2 * T - 1) if i != tree.root], size=12, replace=False):
    bl[n] *= 1.1
    t0 = time.perf_counter(); e.set_branch_length(int(n), bl[n]); l = e.log_likelihood(); dt = time.perf_counter() - t0
    out["single"].append({"node": int(n), "ms": dt * 1e3, "launches": e.profile()["lower_launches"], "kernel_ms": e.profile()["lower_ms"]})
e.update_all_nodes(); lchk = e.log_likelihood()
out["check_rel"] = abs(l - lchk) / abs(lchk)
# MCMC store / restore: propose one branch, evaluate, reject
t0 = time.perf_counter(); e.store(); out["first_store_ms"] = (time.perf_counter() - t0) * 1e3   # allocates the second slots
out["device_bytes_after_store"] = e.profile()["device_bytes"]
out["mcmc"] = []
for n in rng.choice([i for i in range(2 * T - 1) if i != tree.root], size=8, replace=False):
    t0 = time.perf_counter(); e.store(); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); e.set_branch_length(int(n), bl[n] * 1.3); lp = e.log_likelihood(); tp = time.perf_counter() - t0
    t0 = time.perf_counter(); e.restore(); lb = e.log_likelihood(); tr = time.perf_counter() - t0
    out["mcmc"].append({"node": int(n), "store_ms": ts * 1e3, "propose_eval_ms": tp * 1e3, "restore_eval_ms": tr * 1e3,
                        "restore_launches": e.profile()["lower_launches"], "restored_rel": abs(lb - lchk) / abs(lchk)})
# the optimiser's loop (optimizer.c:116-150): per branch a first trial (pending change evaluated on its path to the
# root + the branch's upper rebuilt by a walk down from the root), further trials of the same branch, then the accepted length
e2 = Engine(T, P, 4, C, rescale=RESCALE_AUTO)
e2.set_topology(tree.left, tree.right, tree.root); e2.set_branch_lengths(tree.length); e2.set_eigen(ev, U, Ui)
e2.set_frequencies(np.array(bench.GTR_FREQS)); e2.set_category_rates(rates, np.full(C, 0.25)); e2.set_pattern_weights(np.ones(P))
for t in range(T): e2.set_tip_states(t, states[t])
e2.log_likelihood()
out["optimizer"] = []
bl2 = tree.length.copy()
for n in rng.choice([i for i in range(2 * T - 1) if i != tree.root], size=10, replace=False):
    t0 = time.perf_counter(); a = e2.branch_log_likelihood(int(n), bl2[n]); tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    for f in (0.7, 1.3, 1.1, 0.95): e2.branch_log_likelihood(int(n), bl2[n] * f)
    tt = (time.perf_counter() - t0) / 4
    bl2[n] *= 0.95
    t0 = time.perf_counter(); e2.set_branch_length(int(n), bl2[n]); ts = time.perf_counter() - t0
    out["optimizer"].append({"node": int(n), "first_trial_ms": tf * 1e3, "further_trial_ms": tt * 1e3, "accept_ms": ts * 1e3})
e2.update_all_nodes(); chk = e2.log_likelihood()
a = e2.branch_log_likelihood(5, bl2[5])
out["optimizer_check_rel"] = abs(a[0] - chk) / abs(chk)
out["median_first_trial_ms"] = float(np.median([m["first_trial_ms"] for m in out["optimizer"][1:]]))
out["median_further_trial_ms"] = float(np.median([m["further_trial_ms"] for m in out["optimizer"]]))
out["median_restore_eval_ms"] = float(np.median([m["restore_eval_ms"] for m in out["mcmc"]]))
out["median_single_ms"] = float(np.median([s["ms"] for s in out["single"]]))
print(json.dumps(out))
