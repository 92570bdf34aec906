"""Load golden fixtures (tests/golden/<case>/) written by tests/golden/make_golden.py."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

UNROOTED_CASES = sorted(
    d for d in os.listdir(GOLDEN)
    if os.path.isfile(os.path.join(GOLDEN, d, "spec.txt")))

# discrete-trait cases (one attribute per taxon, general data type): trait_spec.txt + traits.txt + tree.nwk
TRAIT_CASES = sorted(
    d for d in os.listdir(GOLDEN)
    if os.path.isfile(os.path.join(GOLDEN, d, "trait_spec.txt")))


def read_trait_case(case):
    """Inputs of a discrete-trait case: states, named ambiguity sets (in the order they were added), taxa and their
    attribute values (tip id == line index), newick, and the general model's structure / rates / frequencies."""
    d = os.path.join(GOLDEN, case)
    out = {"ambiguities": {}, "categories": 1, "alpha": 0.5}
    with open(os.path.join(d, "trait_spec.txt")) as f:
        for line in f:
            k, v = line.split()
            if k == "states":
                out["states"] = v.split(",")
            elif k == "ambiguity":
                name, members = v.split("=")
                out["ambiguities"][name] = members.split("|")
            elif k == "structure":
                out["structure"] = [int(x) for x in v.split(",")]
            elif k in ("rates", "freqs"):
                out[k] = [float(x) for x in v.split(",")]
            elif k in ("categories", "normalize"):
                out[k] = int(v)
            elif k == "alpha":
                out[k] = float(v)
    with open(os.path.join(d, "traits.txt")) as f:
        rows = [line.split() for line in f if line.strip()]
    out["taxa"] = [r[0] for r in rows]
    out["values"] = [r[1] for r in rows]
    with open(os.path.join(d, "tree.nwk")) as f:
        out["newick"] = f.read().strip()
    return out


def read_fasta(path):
    names, seqs = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                names.append(line[1:])
                seqs.append("")
            else:
                seqs[-1] += line
    return names, seqs


def read_spec(case):
    spec = {}
    with open(os.path.join(GOLDEN, case, "spec.txt")) as f:
        for line in f:
            k, v = line.split()
            spec[k] = v
    spec.setdefault("categories", "1")
    spec.setdefault("tipstates", "0")
    spec.setdefault("rescale", "0")
    return spec


def load(case):
    with gzip.open(os.path.join(GOLDEN, case, "expected.json.gz")) as f:
        exp = json.loads(f.read().decode())
    out = dict(exp)
    for k in ("weights", "pattern_lk", "cat_rates", "cat_proportions", "cat_rates_without_mu", "frequencies", "eval", "gradient_tree", "gradient_all"):
        if k in exp:
            out[k] = np.array(exp[k], dtype=np.float64)
    S = exp["state_count"]
    for k in ("evec", "ivec", "Q"):
        if k in exp:
            out[k] = np.array(exp[k], dtype=np.float64).reshape(S, S)
    out["patterns"] = np.array(exp["patterns"], dtype=np.uint8)
    N = exp["node_count"]
    left = np.full(N, -1, dtype=np.int32)
    right = np.full(N, -1, dtype=np.int32)
    dist = np.zeros(N)
    mapping = np.full(N, -1, dtype=np.int32)
    names = [""] * N
    for nd in exp["nodes"]:
        i = nd["id"]
        left[i], right[i], dist[i], mapping[i], names[i] = nd["left"], nd["right"], nd["distance"], nd["mapping"], nd["name"]
    dist[exp["root"]] = 0.0
    out.update(left=left, right=right, distance=dist, mapping=mapping, node_names=names)
    if os.path.isfile(os.path.join(GOLDEN, case, "spec.txt")):
        fill_closed_form_eigen(case, out)
    C, P = exp["category_count"], exp["pattern_count"]
    for k in ("partials_first_internal", "partials_root", "upper_tip0", "upper_first_internal"):
        if k in exp:
            out[k] = np.array(exp[k], dtype=np.float64).reshape(C, P, S)
    for k in ("pt", "dpt"):
        if k in exp:
            out[k] = np.array(exp[k], dtype=np.float64).reshape(3, C, S, S)
    return out


def reversible_eigen(rates_sym, freqs):
    """Eigen system of a normalised reversible rate matrix Q_ij = r_ij * pi_j (test-side numpy helper).

    Used for the closed-form models (JC69, HKY) whose reference objects carry no eigendecomposition
    (jc69.c:73-79, hky.c:230-273 compute P(t) analytically)."""
    pi = np.asarray(freqs, dtype=np.float64)
    Q = np.asarray(rates_sym, dtype=np.float64) * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q /= -(pi * np.diag(Q)).sum()
    d = np.sqrt(pi)
    B = (d[:, None] * Q) / d[None, :]
    B = 0.5 * (B + B.T)
    w, V = np.linalg.eigh(B)
    return w, V / d[:, None], V.T * d[None, :]


def fill_closed_form_eigen(case, gold):
    """JC69/HKY fixtures have an all-zero eigen system: rebuild it from the spec's parameters."""
    if np.any(gold["eval"] != 0.0):
        return gold
    spec = read_spec(case)
    r = np.ones((4, 4))
    if spec["model"] == "hky":
        kappa = float(spec["rates"])
        r[0, 2] = r[2, 0] = r[1, 3] = r[3, 1] = kappa
    elif spec["model"] != "jc69":
        raise ValueError(spec["model"])
    gold["eval"], gold["evec"], gold["ivec"] = reversible_eigen(r, gold["frequencies"])
    return gold


def oracle_problem(case, gold=None, **kw):
    """Build an oracle Problem from a golden case: tip data in tip-id order, tipstates as in the spec."""
    from oracle import phyoracle as po
    gold = gold or load(case)
    spec = read_spec(case)
    T = gold["tip_count"]
    order = gold["mapping"][:T]  # tip id -> sequence index
    states = gold["patterns"][order]
    tip_partials = None
    if spec["tipstates"] == "0":
        tip_partials = po.state_partials(spec["datatype"], gold["state_count"], states)
    kw.setdefault("rescale", 2 if spec["rescale"] == "0" else 1)
    return po.Problem(gold["left"], gold["right"], gold["root"], gold["weights"], gold["eval"], gold["evec"], gold["ivec"],
                      gold["frequencies"], gold["cat_rates"], gold["cat_proportions"], gold["distance"],
                      tip_states=states, tip_partials=tip_partials, **kw)
