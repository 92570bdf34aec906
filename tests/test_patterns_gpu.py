"""GPU: site-pattern compression on the device (phyamd_compress_patterns) against the ORACLE's sequential hash
table (oracle/phyoracle.c::phyo_compress_patterns, pinned to the reference's new_SitePattern by tests/test_oracle_golden.py)
and against the patterns / weights the compiled reference itself produced for every committed alignment (gold["patterns"],
gold["weights"]).  The product's own host table is compared too, but is never the only yardstick.
Bit-exact: the same columns, in the same order, with the same weights."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from golden_util import GOLDEN, UNROOTED_CASES, load, read_fasta, read_spec
from oracle import phyoracle as po

pytestmark = pytest.mark.gpu


def _random_alignment(rng, T, L, alphabet, p_dup=0.5):
    """columns drawn from a small pool (so duplicates abound) mixed with fresh random columns"""
    pool = rng.integers(0, len(alphabet), size=(T, max(1, L // 7)))
    cols = np.where(rng.random(L) < p_dup, rng.integers(0, pool.shape[1], size=L), -1)
    codes = rng.integers(0, len(alphabet), size=(T, L))
    for s in np.nonzero(cols >= 0)[0]:
        codes[:, s] = pool[:, cols[s]]
    return ["".join(alphabet[c] for c in row) for row in codes]


@pytest.mark.parametrize("T,L,alphabet,datatype", [
    (1, 1, "ACGT", "nucleotide"), (2, 7, "ACGT", "nucleotide"), (5, 120, "ACGT-", "nucleotide"), (9, 1000, "ACGTRYN-?acgt", "nucleotide"),
    (16, 5000, "ACGT", "nucleotide"), (7, 40000, "ACGT-", "nucleotide"), (30, 200000, "ACGT", "nucleotide"),
    (6, 3000, "ACDEFGHIKLMNPQRSTVWYBZX*?-", "aa"), (12, 30000, "ACDEFGHIKLMNPQRSTVWY", "aa")])
def test_device_compression_is_bit_exact(T, L, alphabet, datatype):
    from physher_amd import _phycpp_amd as pc
    rng = np.random.default_rng(T * 1000 + L)
    seqs = _random_alignment(rng, T, L, alphabet)
    names = [f"t{i}" for i in range(T)]
    ds, dw = pc.compress_patterns_device(datatype, names, seqs)
    os_, ow = po.compress_patterns(po.encode_alignment(datatype, seqs))  # the oracle's chained table, column by column
    assert ds.shape == os_.shape
    np.testing.assert_array_equal(ds, os_)
    np.testing.assert_array_equal(dw, ow)
    hs, hw = pc.compress_patterns(datatype, names, seqs)  # and the host library's table agrees with both
    np.testing.assert_array_equal(ds, hs)
    np.testing.assert_array_equal(dw, hw)
    assert dw.sum() == L


def test_device_compression_codons_and_growth_boundaries():
    """codon triplets (coded on the host, grouped on the device) and pattern counts right at the table's growth limits
    (126 = ceil(0.65 * 193) distinct columns fill the first table; the 127th grows it)"""
    from physher_amd import _phycpp_amd as pc
    rng = np.random.default_rng(3)
    seqs = _random_alignment(rng, 5, 3 * 4000, "ACGT", p_dup=0.3)
    names = [f"t{i}" for i in range(5)]
    os_, ow = po.compress_patterns(po.encode_alignment("codon", seqs))
    ds, dw = pc.compress_patterns_device("codon", names, seqs)
    np.testing.assert_array_equal(ds, os_)
    np.testing.assert_array_equal(dw, ow)
    T = 6
    cols = set()
    while len(cols) < 260:
        cols.add(tuple(rng.integers(0, 4, size=T)))
    cols = np.array(sorted(cols), dtype=int).T  # [T][260] distinct columns
    names = [f"t{i}" for i in range(T)]
    for n in (125, 126, 127, 128, 252, 253, 254, 260):
        order = rng.permutation(n)
        picks = np.concatenate([order, rng.integers(0, n, size=50)])  # every column once, then repeats
        seqs = ["".join("ACGT"[c] for c in cols[t, picks]) for t in range(T)]
        os_, ow = po.compress_patterns(po.encode_alignment("nucleotide", seqs))
        ds, dw = pc.compress_patterns_device("nucleotide", names, seqs)
        assert os_.shape[1] == n
        np.testing.assert_array_equal(ds, os_)
        np.testing.assert_array_equal(dw, ow)


def test_device_compression_golden_alignments():
    """every committed alignment, including the reference's own fluA data: the device's patterns and weights are the ones
    the compiled reference's new_SitePattern produced (fixtures), byte for byte"""
    from physher_amd import _phycpp_amd as pc
    cases = [(c, os.path.join(GOLDEN, c, "aln.fa"), read_spec(c)["datatype"]) for c in UNROOTED_CASES]
    cases.append(("fluA_jc69_time", os.path.join(GOLDEN, "fluA_jc69_time", "fluA.fa"), "nucleotide"))
    for case, path, datatype in cases:
        names, seqs = read_fasta(path)
        gold = load(case)
        ds, dw = pc.compress_patterns_device(datatype, names, seqs)
        np.testing.assert_array_equal(ds, gold["patterns"], err_msg=case)
        np.testing.assert_array_equal(dw, gold["weights"], err_msg=case)


def test_c_abi_argument_checks():
    from physher_amd import _lib
    lib = _lib.load()
    n = C.c_int32()
    row = (C.c_uint8 * 4)(0, 1, 2, 3)
    rows = (C.c_void_p * 1)(C.addressof(row))
    out = (C.c_uint8 * 4)()
    w = (C.c_double * 4)()
    assert lib.phyamd_compress_patterns(-1, 0, 4, rows, None, C.byref(n), out, w) == _lib.EINVAL
    assert lib.phyamd_compress_patterns(-1, 1, 0, rows, None, C.byref(n), out, w) == _lib.EINVAL
    assert lib.phyamd_compress_patterns(99, 1, 4, rows, None, C.byref(n), out, w) == _lib.EINVAL
    assert lib.phyamd_compress_patterns(-1, 1, 4, rows, None, C.byref(n), out, w) == _lib.OK
    assert n.value == 4 and list(w) == [1.0] * 4 and sorted(out) == [0, 1, 2, 3]
    lut = (C.c_uint8 * 256)(*([7] * 256))  # every symbol -> code 7: one pattern of weight 4
    assert lib.phyamd_compress_patterns(-1, 1, 4, rows, lut, C.byref(n), out, w) == _lib.OK
    assert n.value == 1 and w[0] == 4.0 and out[0] == 7


def test_wrapper_uses_the_device_for_long_alignments():
    """alignments of >= 50000 sites are compressed on the device inside the TreeLikelihoodInterface constructor: same patterns,
    same lnL as an engine fed with the host-compressed patterns"""
    from physher_amd import _phycpp_amd as pc, synth
    rng = np.random.default_rng(11)
    T, L = 12, 60000
    tree_s = synth.random_tree(T, rng)
    states = synth.evolve(tree_s, L, 4, rng)
    seqs = ["".join("ACGT"[c] for c in row) for row in states]
    names = list(tree_s.names)
    hs, hw = pc.compress_patterns("nucleotide", names, seqs)
    tree = pc.UnRootedTreeModelInterface(tree_s.newick(), names)
    subst = pc.HKYInterface(2.0, [0.3, 0.2, 0.2, 0.3])
    site = pc.ConstantSiteModelInterface(None)
    tlk = pc.TreeLikelihoodInterface(list(zip(names, seqs)), tree, subst, site, None)
    np.testing.assert_array_equal(tlk.pattern_states(), hs)
    np.testing.assert_array_equal(tlk.pattern_weights(), hw)
    assert np.isfinite(tlk.log_likelihood())


def test_device_compression_timing_report(capsys):
    """not an assertion on speed: prints host vs device time for a 200 x 1e6 alignment into the test log"""
    from physher_amd import _phycpp_amd as pc
    rng = np.random.default_rng(5)
    T, L = 200, 1_000_000
    pool = rng.integers(0, 4, size=(T, 300_000), dtype=np.uint8)
    pick = rng.integers(0, pool.shape[1], size=L)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [lut[pool[t, pick]].tobytes().decode() for t in range(T)]
    names = [f"t{i}" for i in range(T)]
    t0 = time.perf_counter()
    hs, hw = pc.compress_patterns("nucleotide", names, seqs)
    t1 = time.perf_counter()
    ds, dw = pc.compress_patterns_device("nucleotide", names, seqs)
    t2 = time.perf_counter()
    ds, dw = pc.compress_patterns_device("nucleotide", names, seqs)
    t3 = time.perf_counter()
    np.testing.assert_array_equal(ds, hs)
    np.testing.assert_array_equal(dw, hw)
    with capsys.disabled():
        print(f"\n[pattern compression {T} x {L} -> {hs.shape[1]} patterns] host {t1 - t0:.2f} s, device {t2 - t1:.2f} s (first call), {t3 - t2:.2f} s")
