#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only where /root/reference exists (this container): builds oracle/_ref (the reference's own
sources compiled where they lie, see oracle/Makefile) and drives it through oracle/ref_driver.c.
Inputs are seeded synthetic alignments/trees from physher_amd/synth.py, plus the reference's own
test data files (tests/data/fluA.fa + jc69-time.json: data fixtures of its known-answer test).

Each case directory holds: aln.fa, tree.nwk, spec.txt (inputs) and expected.json.gz (outputs of the
reference); the discrete-trait cases hold traits.txt, tree.nwk, trait_spec.txt instead.  Re-running this script must reproduce the committed files byte for byte.
"""
import gzip
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from physher_amd import synth  # noqa: E402

REF = os.environ.get("PHYSHER_REF", "/root/reference")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

GTR_RATES = "1.2,3.1,0.7,0.9,2.8"
GTR_FREQS = "0.3,0.2,0.2,0.3"

AA_FREQS = ",".join(repr(x) for x in (np.arange(1, 21) / 210.0).tolist())

CASES = [
    # name, dict(options)
    ("gtr_g4_t16", dict(T=16, sites=400, seed=1, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5)),
    ("gtr_g4_t16_tipstates", dict(T=16, sites=400, seed=1, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5, tipstates=1)),
    ("gtr_g4_t16_scalar", dict(T=16, sites=400, seed=1, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5, sse=0)),
    ("gtr_g4_t24_gaps", dict(T=24, sites=500, seed=2, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.7, gaps=0.05)),
    ("gtr_g4_t24_gaps_tipstates", dict(T=24, sites=500, seed=2, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.7, gaps=0.05, tipstates=1)),
    ("jc69_t12", dict(T=12, sites=300, seed=3, datatype="nucleotide", model="jc69", freqs="0.25,0.25,0.25,0.25", categories=1)),
    ("hky_g3_t10", dict(T=10, sites=300, seed=4, datatype="nucleotide", model="hky", rates="2.5", freqs="0.1,0.2,0.3,0.4", categories=3, alpha=1.3)),
    ("gtr_g4_caterpillar_t20", dict(T=20, sites=300, seed=5, shape="caterpillar", datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5)),
    ("gtr_g1_t96_rescale", dict(T=96, sites=150, seed=6, bl=(0.3, 0.9), datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=1, rescale=1)),
    ("gtr_g4_t96_rescale", dict(T=96, sites=150, seed=6, bl=(0.3, 0.9), datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5, rescale=1)),
    ("gtr_g4_t700_autorescale", dict(T=700, sites=20, seed=7, bl=(0.5, 1.5), datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5, slim=1)),
    ("wag_g4_t12", dict(T=12, sites=80, seed=8, datatype="aa", model="wag", categories=4, alpha=0.5)),
    ("wag_g4_t12_tipstates", dict(T=12, sites=80, seed=8, datatype="aa", model="wag", categories=4, alpha=0.5, tipstates=1)),
    # LG without explicit frequencies crashes in the reference (lg.c:47 builds a 0-dimensional simplex): pass them
    ("lg_g1_t9_gaps", dict(T=9, sites=60, seed=9, datatype="aa", model="lg", categories=1, gaps=0.05, freqs=AA_FREQS)),
    # site-model gradient cases (shape, pinv, mu): SITE_MODEL block of gradient_all
    ("gtr_g4i_mu_t14", dict(T=14, sites=400, seed=12, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.6,
                            pinv=0.15, mu=1.7)),
    ("gtr_w3_t12", dict(T=12, sites=300, seed=13, datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=3, alpha=1.4,
                        sitedist="weibull")),
    ("hky_w4i_t10", dict(T=10, sites=300, seed=14, datatype="nucleotide", model="hky", rates="2.0", freqs="0.1,0.2,0.3,0.4", categories=4, alpha=0.9,
                         sitedist="weibull", pinv=0.1)),
    ("jc69_inv_t10", dict(T=10, sites=300, seed=15, datatype="nucleotide", model="jc69", freqs="0.25,0.25,0.25,0.25", categories=1,
                          sitedist="discrete", pinv=0.2)),
    ("mg94_t8", dict(T=8, sites=40, seed=10, datatype="codon", model="mg94", rates="2.0,1.0,0.5", categories=1)),
    ("wag_g4_t60_rescale", dict(T=60, sites=40, seed=16, bl=(0.3, 0.9), datatype="aa", model="wag", categories=4, alpha=0.5, rescale=1, slim=1)),
    ("wag_g2_t500_autorescale", dict(T=500, sites=10, seed=17, bl=(0.5, 1.5), datatype="aa", model="wag", categories=2, alpha=0.5, slim=1)),
    ("mg94_t40_rescale", dict(T=40, sites=12, seed=18, bl=(0.3, 0.9), datatype="codon", model="mg94", rates="2.0,1.0,0.5", categories=1, rescale=1, slim=1)),
    # the headline tree size, unscaled (branches short enough to stay above the underflow that switches rescaling on): the default
    # chunked walks of the engine and the binding (integration/physher_device.c) against the reference at 1000 taxa
    ("gtr_g4_t1000", dict(T=1000, sites=150, seed=19, bl=(0.004, 0.04), datatype="nucleotide", model="gtr", rates=GTR_RATES, freqs=GTR_FREQS, categories=4, alpha=0.5,
                          tipstates=1, slim=1)),
    ("mg94_g2_t6_tipstates", dict(T=6, sites=30, seed=11, datatype="codon", model="mg94", rates="2.0,1.0,0.5", categories=2, alpha=0.8, tipstates=1)),
]

STATE_COUNT = {"nucleotide": 4, "aa": 20, "codon": 61}
# bulky arrays dropped from "slim" cases
SLIM_DROP = ("partials_first_internal", "partials_root", "upper_tip0", "upper_first_internal", "pt", "dpt")


def build_ref():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", f"REF={REF}"])


def run_case(name, o):
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(o["seed"])
    lo, hi = o.get("bl", (0.01, 0.1))
    tree = synth.random_tree(o["T"], rng, shape=o.get("shape", "random"), bl_low=lo, bl_high=hi)
    S = STATE_COUNT[o["datatype"]]
    states = synth.evolve(tree, o["sites"], S, rng)
    fasta = synth.to_fasta(tree.names, states, o["datatype"], o.get("gaps", 0.0), rng)
    with open(os.path.join(d, "aln.fa"), "w") as f:
        f.write(fasta)
    with open(os.path.join(d, "tree.nwk"), "w") as f:
        f.write(tree.newick() + "\n")
    spec = [f"fasta {d}/aln.fa", f"newick {d}/tree.nwk", f"datatype {o['datatype']}", f"model {o['model']}"]
    for k in ("rates", "freqs", "categories", "alpha", "tipstates", "sse", "rescale", "sitedist", "pinv", "mu"):
        if k in o:
            spec.append(f"{k} {o[k]}")
    tmp_spec = os.path.join(d, "spec.abs.txt")
    with open(tmp_spec, "w") as f:
        f.write("\n".join(spec) + "\n")
    # the committed spec uses paths relative to the case directory
    with open(os.path.join(d, "spec.txt"), "w") as f:
        f.write("\n".join(s.replace(d + "/", "") for s in spec) + "\n")
    out = os.path.join(d, "expected.json")
    subprocess.check_call([DRIVER, "dump", tmp_spec, out], stdout=subprocess.DEVNULL)
    os.remove(tmp_spec)
    with open(out) as f:
        data = json.load(f)
    if o.get("slim"):
        for k in SLIM_DROP:
            data.pop(k, None)
    os.remove(out)
    with gzip.GzipFile(out + ".gz", "w", mtime=0) as f:
        f.write(json.dumps(data, separators=(",", ":")).encode())
    print(f"{name}: T={data['tip_count']} P={data['pattern_count']} S={data['state_count']} C={data['category_count']} "
          f"lnL={data['lnl']!r} rescaled={data['rescaled']}")


def run_fluA():
    """The reference's only known-answer case (tests/test_tree_likelihood.c) through the driver's json mode."""
    d = os.path.join(HERE, "fluA_jc69_time")
    os.makedirs(d, exist_ok=True)
    for fn in ("fluA.fa", "jc69-time.json"):
        shutil.copyfile(os.path.join(REF, "tests", "data", fn), os.path.join(d, fn))
    out = os.path.join(d, "expected.json")
    subprocess.check_call([DRIVER, "json", "jc69-time.json", out], cwd=d, stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)
    os.remove(out)
    for k in SLIM_DROP:
        data.pop(k, None)
    with gzip.GzipFile(out + ".gz", "w", mtime=0) as f:
        f.write(json.dumps(data, separators=(",", ":")).encode())
    print(f"fluA_jc69_time: lnL={data['lnl_jacobian0']!r} (reference test constant -4777.616349713985), "
          f"lnL+jac={data['lnl_jacobian1']!r} (-4786.867701371271)")


def run_fluA_hky_g4():
    """The fluA time tree (data fixture of fluA_jc69_time) under HKY + G4 with a strict clock: a configuration of ours in the
    reference's JSON schema, run through the driver's json mode -- pins the ratio / clock / site / substitution blocks of
    the time-tree gradient beyond the reference's own JC69 test."""
    src = os.path.join(HERE, "fluA_jc69_time")
    d = os.path.join(HERE, "fluA_hky_g4_time")
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(os.path.join(src, "fluA.fa"), os.path.join(d, "fluA.fa"))
    with open(os.path.join(src, "jc69-time.json")) as f:
        js = json.load(f)
    model = js["model"]
    model["sitemodel"]["substitutionmodel"] = {
        "id": "sm", "type": "substitutionmodel", "model": "hky", "datatype": "nucleotide",
        "frequencies": {"id": "freqs", "type": "Simplex", "values": [0.33, 0.19, 0.23, 0.25]},
        "rates": {"kappa": {"id": "kappa", "type": "parameter", "value": 3.0, "lower": 0, "upper": "infinity"}},
    }
    model["sitemodel"]["distribution"] = {
        "distribution": "gamma", "categories": 4,
        "parameters": {"alpha": {"id": "alpha", "type": "parameter", "value": 0.6, "lower": 0, "upper": "infinity"}},
    }
    model["branchmodel"]["rate"]["value"] = 0.0015
    with open(os.path.join(d, "hky-g4-time.json"), "w") as f:
        json.dump({"model": model}, f, indent=1)
    out = os.path.join(d, "expected.json")
    subprocess.check_call([DRIVER, "json", "hky-g4-time.json", out], cwd=d, stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)
    os.remove(out)
    for k in SLIM_DROP:
        data.pop(k, None)
    with gzip.GzipFile(out + ".gz", "w", mtime=0) as f:
        f.write(json.dumps(data, separators=(",", ":")).encode())
    print(f"fluA_hky_g4_time: lnL={data['lnl_jacobian0']!r}, gradient blocks={len(data['gradient_all_time'])} flags={data['gradient_all_time_flags']}")


def run_fluA_discrete_clock():
    """fluA under HKY + G4 with one clock rate per branch (the reference's "discrete" branch model = phycpp's
    SimpleClockModelInterface, physher.cpp:188-217): pins the per-branch clock block of the time-tree gradient."""
    src = os.path.join(HERE, "fluA_jc69_time")
    d = os.path.join(HERE, "fluA_hky_g4_branch_rates")
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(os.path.join(src, "fluA.fa"), os.path.join(d, "fluA.fa"))
    with open(os.path.join(src, "jc69-time.json")) as f:
        js = json.load(f)
    model = js["model"]
    model["sitemodel"]["substitutionmodel"] = {
        "id": "sm", "type": "substitutionmodel", "model": "hky", "datatype": "nucleotide",
        "frequencies": {"id": "freqs", "type": "Simplex", "values": [0.33, 0.19, 0.23, 0.25]},
        "rates": {"kappa": {"id": "kappa", "type": "parameter", "value": 3.0, "lower": 0, "upper": "infinity"}},
    }
    model["sitemodel"]["distribution"] = {
        "distribution": "gamma", "categories": 4,
        "parameters": {"alpha": {"id": "alpha", "type": "parameter", "value": 0.6, "lower": 0, "upper": "infinity"}},
    }
    n_branches = 2 * 69 - 2
    rng = np.random.default_rng(31)
    rates = np.round(rng.uniform(0.0008, 0.0025, size=n_branches), 7)
    model["branchmodel"] = {"id": "bm", "type": "branchmodel", "model": "discrete", "tree": "&tree",
                            "parameters": {"id": "clock_rates", "type": "parameter", "dimension": n_branches,
                                           "values": [float(x) for x in rates], "lower": 0}}
    with open(os.path.join(d, "hky-g4-branch-rates.json"), "w") as f:
        json.dump({"model": model}, f, indent=1)
    out = os.path.join(d, "expected.json")
    subprocess.check_call([DRIVER, "json", "hky-g4-branch-rates.json", out], cwd=d, stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)
    os.remove(out)
    for k in SLIM_DROP:
        data.pop(k, None)
    with gzip.GzipFile(out + ".gz", "w", mtime=0) as f:
        f.write(json.dumps(data, separators=(",", ":")).encode())
    print(f"fluA_hky_g4_branch_rates: lnL={data['lnl_jacobian0']!r}, gradient entries={len(data['gradient_all_time'])} flags={data['gradient_all_time_flags']}")


ATTR_CASES = [
    # discrete traits (one attribute per taxon, general data type): the wrapper's second constructor (physher.cpp:594-629)
    ("trait_k5_g3_t14", dict(T=14, K=5, seed=21, categories=3, alpha=0.8, n_rates=3, unknown=(3,), ambiguities={})),
    ("trait_k7_sets_t12", dict(T=12, K=7, seed=22, categories=1, n_rates=21, unknown=(5,),
                               ambiguities={"north": (0, 1, 2), "coast": (2, 5)}, ambiguous_tips={1: "north", 8: "coast"})),
    ("trait_k2_t9", dict(T=9, K=2, seed=23, categories=1, n_rates=1, unknown=(), ambiguities={})),
]


def reference_consistent_structure(K, n_rates):
    """A symmetric rate-class assignment in the only layout the reference's general model reads consistently.

    new_GeneralModel_with_parameters (gensubst.c:177-200) takes the S(S-1)/2 packed form to _reversible_update_Q, which
    indexes it as a full matrix (gensubst.c:130-151: out-of-bounds reads).  The S(S-1) form goes through
    _nonreversible_update_Q (upper triangle row by row, then lower triangle row by row, gensubst.c:60-79), but _general_dQdp
    (gensubst.c:227-260) walks the lower triangle in the order of the mirrored upper one.  The two orders agree when pairs
    sharing a position share a rate class: merge those pairs, then deal classes out to the rates."""
    lower_rows = [(i, j) for i in range(1, K) for j in range(i)]
    lower_mirrored = [(j, i) for i in range(K) for j in range(i + 1, K)]
    parent = {p: p for p in lower_rows}

    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x
    for a, b in zip(lower_rows, lower_mirrored):
        parent[find(a)] = find(b)
    roots = sorted({find(p) for p in lower_rows})
    cls = {p: roots.index(find(p)) % n_rates for p in lower_rows}
    upper = [cls[(j, i)] for i in range(K) for j in range(i + 1, K)]
    lower = [cls[p] for p in lower_rows]
    assert lower == [cls[p] for p in lower_mirrored]
    return upper + lower, len(roots)


def run_attr_case(name, o):
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(o["seed"])
    T, K = o["T"], o["K"]
    tree = synth.random_tree(T, rng, bl_low=0.05, bl_high=0.6)
    states = [f"loc{i}" for i in range(K)]
    values = [states[c] for c in rng.integers(0, K, size=T)]
    for i in o["unknown"]:
        values[i] = "?"
    for i, a in o.get("ambiguous_tips", {}).items():
        values[i] = a
    with open(os.path.join(d, "traits.txt"), "w") as f:
        f.write("".join(f"{n} {v}\n" for n, v in zip(tree.names, values)))
    with open(os.path.join(d, "tree.nwk"), "w") as f:
        f.write(tree.newick() + "\n")
    structure, n_classes = reference_consistent_structure(K, o["n_rates"])
    rates = np.round(rng.uniform(0.3, 2.5, size=o["n_rates"]), 3)
    freqs = rng.dirichlet(np.full(K, 4.0))
    spec = ["states " + ",".join(states)]
    spec += [f"ambiguity {a}=" + "|".join(states[i] for i in m) for a, m in o["ambiguities"].items()]
    spec += [f"traits {d}/traits.txt", f"newick {d}/tree.nwk", "structure " + ",".join(map(str, structure)),
             "rates " + ",".join(repr(float(x)) for x in rates), "freqs " + ",".join(repr(float(x)) for x in freqs),
             "normalize 1", f"categories {o['categories']}"]
    if "alpha" in o:
        spec.append(f"alpha {o['alpha']}")
    tmp_spec = os.path.join(d, "spec.abs.txt")
    with open(tmp_spec, "w") as f:
        f.write("\n".join(spec) + "\n")
    with open(os.path.join(d, "trait_spec.txt"), "w") as f:
        f.write("\n".join(s.replace(d + "/", "") for s in spec) + "\n")
    out = os.path.join(d, "expected.json")
    subprocess.check_call([DRIVER, "attr", tmp_spec, out], stdout=subprocess.DEVNULL)
    os.remove(tmp_spec)
    with open(out) as f:
        data = json.load(f)
    os.remove(out)
    with gzip.GzipFile(out + ".gz", "w", mtime=0) as f:
        f.write(json.dumps(data, separators=(",", ":")).encode())
    print(f"{name}: K={data['state_count']} lnL={data['lnl']!r} gradient_all={len(data['gradient_all'])} flags={data['gradient_all_flags']}")


# single-branch trial evaluations (the optimiser's fast path): lnL(t), d lnL/dt, d2 lnL/dt2 by the reference's
# upper-partial protocol (ref_driver branch mode) on four unscaled cases -- 4, 20 and 61 states, tip partials and tip states
BRANCH_TRIAL_CASES = ("gtr_g4_t16", "gtr_g4_t24_gaps_tipstates", "wag_g4_t12", "mg94_t8")


def run_branch_trials(name):
    d = os.path.join(HERE, name)
    out = os.path.join(d, "branch_trials.json")
    subprocess.check_call([DRIVER, "branch", "spec.txt", out], cwd=d, stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)
    with open(out, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    print(name, "branch trials:", len(data["trials"]))


# the optimiser's call pattern (serial_brent_optimize_tree, optimizer.c:112-153) by the reference's CPU path: single-branch trial
# evaluations with use_upper on, accepted lengths, then the plain lnL and gradient -- a closed-form model (JC69: the binding hands
# the engine the model's own P(t) per node) and the eigen route (GTR + G4)
BRENT_CASES = ("jc69_t12", "gtr_g4_t16")


def run_brent(name):
    d = os.path.join(HERE, name)
    out = os.path.join(d, "brent_trials.json")
    subprocess.check_call([DRIVER, "brent", "spec.txt", out], cwd=d, stdout=subprocess.DEVNULL)
    with open(out) as f:
        data = json.load(f)
    with open(out, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    print(name, "brent trials:", len(data["trials"]), "lnl", data["lnl_start"], "->", data["lnl_end"])


if __name__ == "__main__":
    build_ref()
    only = sys.argv[1:]
    if only == ["brent"]:
        for name in BRENT_CASES:
            run_brent(name)
        sys.exit(0)
    if only == ["branch_trials"]:
        for name in BRANCH_TRIAL_CASES:
            run_branch_trials(name)
        sys.exit(0)
    for name, opts in CASES:
        if not only or name in only:
            run_case(name, opts)
            if name in BRANCH_TRIAL_CASES:
                run_branch_trials(name)
    if not only or "fluA_jc69_time" in only:
        run_fluA()
    if not only or "fluA_hky_g4_time" in only:
        run_fluA_hky_g4()
    if not only or "fluA_hky_g4_branch_rates" in only:
        run_fluA_discrete_clock()
    for name, opts in ATTR_CASES:
        if not only or name in only:
            run_attr_case(name, opts)
