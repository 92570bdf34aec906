"""Seeded synthetic trees and alignments for the tree-likelihood hot path.

Own code (SURVEY.md §8d recipe): random binary topology by random joins, branch lengths
U(0.01, 0.1), uniform root sequence, Jukes-Cantor-style evolution down the tree so columns look
like real data.  State codes follow the reference's encodings (datatype.c:55-89:
nucleotides "ACGT", amino acids "ACDEFGHIKLMNPQRSTVWY", codons = 64 triplets minus the stops of
the universal code, sitepattern.c:808-819).
"""
from __future__ import annotations

import numpy as np

NUC = "ACGT"
AA = "ACDEFGHIKLMNPQRSTVWY"
_STOPS = (48, 50, 56)  # TAA, TAG, TGA in n1*16+n2*4+n3 with A,C,G,T = 0..3
CODONS = [NUC[i >> 4] + NUC[(i >> 2) & 3] + NUC[i & 3] for i in range(64) if i not in _STOPS]


class SynthTree:
    """Rooted binary tree in a plain array form.

    Nodes 0..T-1 are tips (named t0..), T..2T-2 internals in creation order, root = 2T-2.
    ``left/right`` are child indices (-1 for tips); ``length`` is the branch above each node.
    """

    def __init__(self, left, right, length, names):
        self.left = np.asarray(left, dtype=np.int32)
        self.right = np.asarray(right, dtype=np.int32)
        self.length = np.asarray(length, dtype=np.float64)
        self.names = list(names)
        self.tip_count = len(names)
        self.node_count = len(self.left)
        self.root = self.node_count - 1

    def newick(self) -> str:
        out = {}
        # iterative post-order so 1e5-taxon trees do not hit the recursion limit
        stack = [(self.root, False)]
        while stack:
            n, done = stack.pop()
            if self.left[n] < 0:
                out[n] = f"{self.names[n]}:{float(self.length[n])!r}"
            elif not done:
                stack.append((n, True))
                stack.append((int(self.right[n]), False))
                stack.append((int(self.left[n]), False))
            else:
                s = f"({out.pop(int(self.left[n]))},{out.pop(int(self.right[n]))})"
                out[n] = s + (";" if n == self.root else f":{float(self.length[n])!r}")
        return out[self.root]


def random_tree(tip_count: int, rng: np.random.Generator, shape: str = "random",
                bl_low: float = 0.01, bl_high: float = 0.1) -> SynthTree:
    """shape: 'random' (random joins), 'caterpillar' (ladder) or 'balanced'."""
    T = tip_count
    N = 2 * T - 1
    left = -np.ones(N, dtype=np.int32)
    right = -np.ones(N, dtype=np.int32)
    active = list(range(T))
    nxt = T
    while len(active) > 1:
        if shape == "random":
            i, j = rng.choice(len(active), size=2, replace=False)
        elif shape == "caterpillar":
            i, j = 0, 1
        elif shape == "balanced":
            i, j = 0, 1
        else:
            raise ValueError(shape)
        a, b = active[i], active[j]
        for idx in sorted((i, j), reverse=True):
            active.pop(idx)
        left[nxt], right[nxt] = a, b
        if shape == "caterpillar":
            active.insert(0, nxt)
        else:
            active.append(nxt)
        nxt += 1
    length = rng.uniform(bl_low, bl_high, size=N)
    length[N - 1] = 0.0
    return SynthTree(left, right, length, [f"t{i}" for i in range(T)])


def evolve(tree: SynthTree, site_count: int, state_count: int, rng: np.random.Generator,
           scale: float = 1.0) -> np.ndarray:
    """Return uint8 states [tip_count][site_count] evolved under an equal-rates model."""
    S = state_count
    N = tree.node_count
    seqs = [None] * N
    seqs[tree.root] = rng.integers(0, S, size=site_count, dtype=np.uint8)
    # parents have larger indices than children: walk down from the root
    for n in range(N - 1, -1, -1):
        if tree.left[n] < 0:
            continue
        for c in (int(tree.left[n]), int(tree.right[n])):
            p_same = 1.0 / S + (1.0 - 1.0 / S) * np.exp(-S / (S - 1.0) * tree.length[c] * scale)
            change = rng.random(site_count) >= p_same
            new = rng.integers(0, S, size=site_count, dtype=np.uint8)
            seqs[c] = np.where(change, new, seqs[n]).astype(np.uint8)
        if n != tree.root or True:
            pass
    return np.stack([seqs[i] for i in range(tree.tip_count)])


def distinct_patterns(tree: SynthTree, pattern_count: int, state_count: int,
                      rng: np.random.Generator, scale: float = 1.0):
    """Generate sites until exactly ``pattern_count`` distinct columns exist.

    Returns (states uint8 [T][P], weights float64 [P]).  Used by the benchmark, where the
    compressed pattern count -- not the raw site count -- is the workload size.
    """
    cols = None
    weights = None
    want = pattern_count
    while True:
        batch = evolve(tree, int(want * 1.3) + 64, state_count, rng, scale)
        allc = batch if cols is None else np.concatenate([cols, batch], axis=1)
        allw = np.ones(batch.shape[1]) if weights is None else np.concatenate([weights, np.ones(batch.shape[1])])
        uniq, inv = np.unique(allc.T, axis=0, return_inverse=True)
        w = np.bincount(inv.ravel(), weights=allw, minlength=len(uniq))
        cols, weights = uniq.T.copy(), w
        if cols.shape[1] >= pattern_count:
            sel = rng.permutation(cols.shape[1])[:pattern_count]
            return np.ascontiguousarray(cols[:, sel]), np.ascontiguousarray(weights[sel])
        want = pattern_count - cols.shape[1]


def to_fasta(names, states: np.ndarray, datatype: str = "nucleotide", gap_fraction: float = 0.0,
             rng: np.random.Generator | None = None) -> str:
    lines = []
    if datatype in ("nucleotide", "aa") and gap_fraction == 0.0:  # one symbol per state, no gaps: a table lookup per row
        table = np.frombuffer((NUC if datatype == "nucleotide" else AA).encode() if isinstance(NUC, str) else "".join(NUC if datatype == "nucleotide" else AA).encode(), dtype="S1")
        return "".join(f">{name}\n{table[np.asarray(row)].tobytes().decode()}\n" for name, row in zip(names, states))
    for name, row in zip(names, states):
        if datatype == "nucleotide":
            sym = [NUC[s] for s in row]
            gap = "-"
        elif datatype == "aa":
            sym = [AA[s] for s in row]
            gap = "-"
        elif datatype == "codon":
            sym = [CODONS[s] for s in row]
            gap = "---"
        else:
            raise ValueError(datatype)
        if gap_fraction > 0.0:
            mask = rng.random(len(sym)) < gap_fraction
            sym = [gap if m else s for s, m in zip(sym, mask)]
        lines.append(f">{name}\n{''.join(sym)}")
    return "\n".join(lines) + "\n"
