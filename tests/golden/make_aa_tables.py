#!/usr/bin/env python3
"""Recover the WAG / LG exchangeability tables of physher_amd/csrc/host/aa_models.inc from the reference-generated fixtures
(s_ij = Q_ij / pi_j of the normalised rate matrix in expected.json.gz; wag_g4_t12 ran with the model's own frequencies, which
become WAG_FREQUENCIES).  Re-running reproduces the committed file."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from golden_util import load  # noqa: E402


def table(case):
    g = load(case)
    S = g["Q"] / g["frequencies"][None, :]
    np.fill_diagonal(S, 0.0)
    return 0.5 * (S + S.T), g["frequencies"]


def emit(name, S):
    S = S / S[np.triu_indices(20, 1)].mean()
    rows = ["\t" + ", ".join(repr(float(S[i, j])) for j in range(i + 1, 20)) + "," for i in range(19)]
    return f"static const double {name}[190] = {{  // upper triangle, row by row\n" + "\n".join(rows) + "\n};\n"


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(HERE)), "physher_amd", "csrc", "host", "aa_models.inc")
    with open(path) as f:
        header = f.read().split("static const double")[0]
    Sw, piw = table("wag_g4_t12")
    Sl, _ = table("lg_g1_t9_gaps")
    with open(path, "w") as f:
        f.write(header + emit("WAG_EXCHANGEABILITIES", Sw) + "static const double WAG_FREQUENCIES[20] = {" + ", ".join(repr(float(x)) for x in piw) + "};\n"
                + emit("LG_EXCHANGEABILITIES", Sl))
    print("wrote", path)
