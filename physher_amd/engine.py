"""numpy-facing wrapper over the C ABI (include/physher_amd.h).  No computation happens in Python."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (GRAD_COMPAT_SCALED, GRAD_FOLD_ROOT_FREQS, RESCALE_ALWAYS, RESCALE_AUTO, RESCALE_NEVER,  # noqa: F401
                   EngineError)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One tree-likelihood instance on one GPU (the device analogue of physher's SingleTreeLikelihood)."""

    def __init__(self, tip_count, pattern_count, state_count=4, category_count=1, device=-1, rescale=RESCALE_AUTO,
                 max_device_bytes=0, stream=None, devices=None):
        """devices: list of HIP device ordinals -> the patterns are sharded over them inside this process
        (phyamd_create_sharded; the same ordinal may repeat); None -> one engine on `device`."""
        self._lib = _lib.load()
        self.T, self.P, self.S, self.C = int(tip_count), int(pattern_count), int(state_count), int(category_count)
        self.N = 2 * self.T - 1
        cfg = _lib.Config(self.T, self.P, self.S, self.C, device, rescale, max_device_bytes, stream)
        h = C.c_void_p()
        self._h = None
        if devices is None:
            self._check(self._lib.phyamd_create(C.byref(cfg), C.byref(h)))
        else:
            ids = np.ascontiguousarray(devices, dtype=np.int32)
            self._check(self._lib.phyamd_create_sharded(C.byref(cfg), len(ids), _ptr(ids), C.byref(h)))
        self._h = h

    @property
    def shard_count(self):
        return self._lib.phyamd_shard_count(self._h)

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self._lib.phyamd_last_error().decode())

    def close(self):
        if self._h is not None:
            self._lib.phyamd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- data
    def set_tip_states(self, tip, states):
        a = np.ascontiguousarray(states, dtype=np.uint8)
        assert a.shape == (self.P,)
        self._check(self._lib.phyamd_set_tip_states(self._h, tip, _ptr(a)))

    def set_tip_partials(self, tip, partials):
        a = _f64(partials)
        assert a.shape == (self.P, self.S)
        self._check(self._lib.phyamd_set_tip_partials(self._h, tip, _ptr(a)))

    def set_pattern_weights(self, w):
        a = _f64(w)
        assert a.shape == (self.P,)
        self._check(self._lib.phyamd_set_pattern_weights(self._h, _ptr(a)))

    def set_topology(self, left, right, root):
        l = np.ascontiguousarray(left, dtype=np.int32)
        r = np.ascontiguousarray(right, dtype=np.int32)
        assert l.shape == (self.N,) and r.shape == (self.N,)
        self._check(self._lib.phyamd_set_topology(self._h, _ptr(l), _ptr(r), int(root)))

    def set_branch_lengths(self, bl):
        a = _f64(bl)
        assert a.shape == (self.N,)
        self._check(self._lib.phyamd_set_branch_lengths(self._h, _ptr(a)))

    def set_branch_length(self, node, length):
        """One branch; the next evaluation only recomputes the path from `node` to the root."""
        self._check(self._lib.phyamd_set_branch_length(self._h, int(node), float(length)))

    def update_all_nodes(self):
        self._check(self._lib.phyamd_update_all_nodes(self._h))

    def set_eigen(self, eval_, evec, ivec):
        a, b, c = _f64(eval_), _f64(evec), _f64(ivec)
        assert a.shape == (self.S,) and b.shape == (self.S, self.S) and c.shape == (self.S, self.S)
        self._check(self._lib.phyamd_set_eigen(self._h, _ptr(a), _ptr(b), _ptr(c)))

    def set_frequencies(self, f):
        a = _f64(f)
        assert a.shape == (self.S,)
        self._check(self._lib.phyamd_set_frequencies(self._h, _ptr(a)))

    def set_category_rates(self, rates, props):
        a, b = _f64(rates), _f64(props)
        assert a.shape == (self.C,) and b.shape == (self.C,)
        self._check(self._lib.phyamd_set_category_rates(self._h, _ptr(a), _ptr(b)))

    def set_node_matrices(self, node, mats):
        a = _f64(mats)
        assert a.shape == (self.C, self.S, self.S)
        self._check(self._lib.phyamd_set_node_matrices(self._h, node, _ptr(a)))

    def set_matrices(self, mats):
        """explicit P(t) of every node at once: [N][C][S][S] by node id (the root's entry is ignored)"""
        a = _f64(mats)
        assert a.shape == (self.N, self.C, self.S, self.S)
        self._check(self._lib.phyamd_set_matrices(self._h, _ptr(a)))

    def set_rate_matrix(self, Q):
        a = _f64(Q)
        assert a.shape == (self.S, self.S)
        self._check(self._lib.phyamd_set_rate_matrix(self._h, _ptr(a)))

    # --- evaluation
    def log_likelihood(self):
        v = C.c_double()
        self._check(self._lib.phyamd_log_likelihood(self._h, C.byref(v)))
        return v.value

    def gradient(self, flags=0):
        """Returns (lnL, cat_gradient [N][C])."""
        v = C.c_double()
        g = np.empty((self.N, self.C))
        self._check(self._lib.phyamd_gradient(self._h, flags, C.byref(v), _ptr(g)))
        return v.value, g

    def branch_gradient(self, flags=0, rates_without_mu=None):
        v = C.c_double()
        g = np.empty(self.N)
        r = None if rates_without_mu is None else _f64(rates_without_mu)
        self._check(self._lib.phyamd_branch_gradient(self._h, flags, None if r is None else _ptr(r), C.byref(v), _ptr(g)))
        return v.value, g

    def log_likelihood_device(self, device_ptr):
        """post-order pass only; lnL -> device_ptr[0] on the engine's stream"""
        self._check(self._lib.phyamd_log_likelihood_device(self._h, C.c_void_p(device_ptr)))

    def gradient_device(self, device_ptr, flags=0):
        self._check(self._lib.phyamd_gradient_device(self._h, flags, C.c_void_p(device_ptr)))

    def root_invariant_term(self):
        v = C.c_double()
        self._check(self._lib.phyamd_root_invariant_term(self._h, C.byref(v)))
        return v.value

    # --- substitution-model gradient (calculate_dlnl_dQ, treelikelihood.c:2337-2583)
    def set_rate_matrix_derivatives(self, dQ):
        """dQ [count][S][S]: d(normalised Q)/d(parameter); an empty array clears."""
        dQ = np.ascontiguousarray(dQ, dtype=np.float64).reshape(-1, self.S, self.S)
        self._np = dQ.shape[0]
        self._check(self._lib.phyamd_set_rate_matrix_derivatives(self._h, self._np, _ptr(dQ)))

    def parameter_gradient(self, flags=0):
        """(lnL, cat_gradient [N][C], parameter_gradient [count]) from one post-order + one pre-order pass."""
        lnl = C.c_double()
        g = np.empty((self.N, self.C))
        pg = np.empty(getattr(self, "_np", 0))
        self._check(self._lib.phyamd_parameter_gradient(self._h, flags, C.byref(lnl), _ptr(g), _ptr(pg)))
        return lnl.value, g, pg

    def parameter_gradient_device(self, device_ptr, flags=0):
        """[lnL | cat gradient | parameter gradient | root frequency term] -> device buffer (1 + N*C + count + S doubles)."""
        self._check(self._lib.phyamd_parameter_gradient_device(self._h, flags, C.c_void_p(device_ptr)))

    def root_frequency_term(self):
        a = np.empty(self.S)
        self._check(self._lib.phyamd_root_frequency_term(self._h, _ptr(a)))
        return a

    def branch_log_likelihood(self, node, length):
        """(lnL, d lnL/dt, d2 lnL/dt2) of one branch at a trial length, from the resident upper/lower partials
        (needs set_keep_partials(True) and a gradient() call for the current parameters)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.phyamd_branch_log_likelihood(self._h, int(node), float(length), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def store(self):
        """Remember the current (evaluated) state: parameters, lnL and partials (MCMC store)."""
        self._check(self._lib.phyamd_store(self._h))

    def restore(self):
        """Back to the stored state without recomputing the tree (MCMC reject)."""
        self._check(self._lib.phyamd_restore(self._h))

    def synchronize(self):
        self._check(self._lib.phyamd_synchronize(self._h))

    # --- inspection
    def pattern_log_likelihoods(self):
        a = np.empty(self.P)
        self._check(self._lib.phyamd_get_pattern_log_likelihoods(self._h, _ptr(a)))
        return a

    def partials(self, node, upper=False):
        a = np.empty((self.C, self.P, self.S))
        self._check(self._lib.phyamd_get_partials(self._h, node, int(upper), _ptr(a)))
        return a

    def node_matrices(self, node, derivative=False):
        a = np.empty((self.C, self.S, self.S))
        self._check(self._lib.phyamd_get_node_matrices(self._h, node, int(derivative), _ptr(a)))
        return a

    @property
    def rescaling(self):
        return bool(self._lib.phyamd_is_rescaling(self._h))

    def set_rescaling(self, policy):
        self._check(self._lib.phyamd_set_rescaling(self._h, int(policy)))

    def set_keep_partials(self, on=True):
        self._check(self._lib.phyamd_set_keep_partials(self._h, int(on)))

    def set_reduction_levels(self, levels):
        """3 (default): this engine sums its pattern blocks over eight bisection segments; an engine holding 1 / 2^k of a larger
        pattern list cut by sharding.shard_range runs with 3 - k (see include/physher_amd.h)."""
        self._check(self._lib.phyamd_set_reduction_levels(self._h, int(levels)))

    def set_profiling(self, on=True):
        self._check(self._lib.phyamd_set_profiling(self._h, int(on)))

    def profile(self):
        p = _lib.Profile()
        self._check(self._lib.phyamd_get_profile(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in p._fields_}
