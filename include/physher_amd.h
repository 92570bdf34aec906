/*
 * physher_amd.h -- C ABI of the MI355X (gfx950) tree-likelihood engine.
 *
 * This is the drop-in boundary for physher's Felsenstein-pruning hot path.  Every entry point is
 * plain C (pointers + sizes, no C++/torch types) and replaces one piece of what
 * `struct _SingleTreeLikelihood` and its five kernel function pointers do on the CPU in the
 * reference (src/phyc/treelikelihood.h:46-124).  The reference-side binding a maintainer would add
 * is shown in INTEGRATION.md.
 *
 * Conventions shared with the reference:
 *   - node ids: tips 0..T-1, internal nodes T..2T-2, children before parents not required
 *     (src/phyc/tree.c:183-224); the root may be any internal node.
 *   - host-side array layouts are the reference's: partials [C][P][S] (state fastest,
 *     treelikelihood.c:1028), matrices [C][S][S] row-major P[parent state][child state]
 *     (substmodel.c:547-555), tip states uint8, code >= S = unknown/gap (sitepattern.h:68-82).
 *   - numerical trouble is reported in-band like the reference does: NaN/inf lnL and an all-NaN
 *     gradient (treelikelihood.c:327-332, 1489-1519).  API misuse and device errors return a
 *     negative PHYAMD_E* code; phyamd_last_error() holds the message.  Nothing here calls exit().
 *
 * All functions are single-caller per engine (the reference's objects are not thread-safe either,
 * SURVEY.md section 8b); different engines may be used from different threads.
 */
#ifndef PHYSHER_AMD_H
#define PHYSHER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct phyamd_engine phyamd_engine;

enum {
	PHYAMD_OK = 0,
	PHYAMD_EINVAL = -1,   /* bad argument / call order */
	PHYAMD_EDEVICE = -2,  /* HIP runtime error (message in phyamd_last_error) */
	PHYAMD_ENOMEM = -3,   /* device memory exhausted */
	PHYAMD_EUNSUPPORTED = -4
};

/* rescaling policy: SingleTreeLikelihood_use_rescaling + the lazy switch of treelikelihood.c:1496-1519 */
enum { PHYAMD_RESCALE_NEVER = 0, PHYAMD_RESCALE_ALWAYS = 1, PHYAMD_RESCALE_AUTO = 2 };

/* gradient flags */
enum {
	/* Multiply the root frequencies into the upper partials of the root's children and drop them from
	 * the final state sum: the reference's `include_root_freqs = true` mode (treelikelihood.c:241,
	 * 2147-2153), which it uses when no substitution-model gradient is requested.  That mode is only
	 * exact for uniform frequencies; the default (0) is the reference's `include_root_freqs = false`
	 * arithmetic (treelikelihood.c:2715-2751), exact for every reversible model. */
	PHYAMD_GRAD_FOLD_ROOT_FREQS = 1,
	/* Under rescaling divide each category's derivative by that category's own site likelihood, as
	 * treelikelihood.c:2851-2870 does, instead of by the mixture likelihood. */
	PHYAMD_GRAD_COMPAT_SCALED = 2
};

typedef struct {
	int32_t tip_count;      /* T */
	int32_t pattern_count;  /* P (this engine's shard of the compressed patterns) */
	int32_t state_count;    /* S: 4 (nucleotides), 20 (amino acids), 60 / 61 (codons); others -> PHYAMD_EUNSUPPORTED */
	int32_t category_count; /* C */
	int32_t device;         /* HIP device ordinal; -1 = current device */
	int32_t rescale;        /* PHYAMD_RESCALE_* (all state counts) */
	int64_t max_device_bytes; /* 0 = 92 % of the device memory that is free when the engine is created.  Below the (estimated)
	                             working set of all patterns the engine processes them in
	                             tiles through ONE set of partial arrays: tip data, weights and per-pattern lnL of all tiles
	                             stay resident, per-tile sums are added in tile order (phyamd_profile.tiles tells how many).
	                             lnL, gradients and parameter gradients work as usual; calls that need resident partials
	                             (phyamd_get_partials, phyamd_set_keep_partials, phyamd_store, phyamd_branch_log_likelihood)
	                             return PHYAMD_EUNSUPPORTED, and every evaluation recomputes every
	                             tile.  PHYAMD_ENOMEM if even a 256-pattern tile does not fit. */
	void *stream;           /* hipStream_t to run on, NULL = engine-owned stream */
} phyamd_config;

/* --- life cycle: new_SingleTreeLikelihood / free_SingleTreeLikelihood (treelikelihood.c:1007-1185) --- */
int phyamd_create(const phyamd_config *cfg, phyamd_engine **out);
/* The same engine with its site patterns sharded over `device_count` GPUs of this node, inside ONE process: shard s
 * is a complete engine on device_ids[s] (NULL: devices 0 .. device_count-1) that owns the contiguous pattern range
 * [s P / n, (s+1) P / n); tree, branch lengths, eigen system and rates are replicated to every shard, per-pattern arguments
 * (tip data, weights, per-pattern lnL, partials) are sliced, and every evaluation runs on all shards at once (one host thread per
 * shard) and returns the per-shard sums added in shard order -- lnL, the [node][category] gradient, parameter sums, root terms:
 * the "single all-reduce of the per-block lnL and gradient vector", done on the host where the caller wants the 64 KB anyway.
 * Every other call of this header works on the handle unchanged, except the *_device evaluations (their output lives on one
 * device: PHYAMD_EUNSUPPORTED for device_count > 1; the one-process-per-GPU form of the same sharding -- bench.py under
 * torchrun -- is what uses them, with one RCCL all-reduce).  cfg->device is ignored, cfg->stream must be NULL,
 * cfg->max_device_bytes applies per shard.  The same device may be listed more than once. */
int phyamd_create_sharded(const phyamd_config *cfg, int32_t device_count, const int32_t *device_ids, phyamd_engine **out);
/* number of shards behind a handle (1 for phyamd_create) */
int phyamd_shard_count(phyamd_engine *e);
void phyamd_destroy(phyamd_engine *e);
const char *phyamd_last_error(void);
/* version of this ABI (bumped on any signature change) */
int phyamd_abi_version(void);

/* --- data: sp->patterns / sp->weights / tlk->partials of tips (sitepattern.h:68-82, treelikelihood.c:1106-1117) --- */
/* states[P] codes of one tip ("tipstates": true semantics; code >= S => all ones). */
int phyamd_set_tip_states(phyamd_engine *e, int tip, const uint8_t *states);
/* partials[P][S] of one tip ("tipstates": false semantics), replicated over categories like treelikelihood.c:1111-1114.
 * Built for the 0/1 vectors every data type of the reference produces (datatype.c:212-240, datatype.h:26-66): one state,
 * all states, or a set of states (nucleotide ambiguity codes; named sets of a general data type -- at most 255 - S
 * distinct sets per engine for 20 / 60 / 61 states); other values are refused with PHYAMD_EUNSUPPORTED. */
int phyamd_set_tip_partials(phyamd_engine *e, int tip, const double *partials);
int phyamd_set_pattern_weights(phyamd_engine *e, const double *weights /* [P] */);

/* new_SitePattern (sitepattern.c:186-251) on the device: de-duplicates the columns of an alignment and returns them in the
 * reference's order -- the iteration order of its chained hash table (hashtable.c: 193 buckets, growth through the prime
 * table at load 0.65, chains prepended and reversed by growth), rebuilt from per-column hashes and first-occurrence ranks
 * with radix sorts instead of T*L sequential insertions.  Engine-independent and synchronous.
 *   rows[taxon_count]: host pointers to site_count bytes each (state codes; or raw one-byte symbols when symbol_codes is
 *   given: symbol_codes[256] maps a symbol to its state code, datatype.c:55-89);
 *   patterns: host buffer of taxon_count * site_count bytes, filled as [taxon][pattern] with *pattern_count columns;
 *   weights: host buffer of site_count doubles.
 * PHYAMD_EUNSUPPORTED if two different columns collide in both 32-bit hashes (nothing is folded: compress on the host). */
int phyamd_compress_patterns(int device /* -1: current */, int32_t taxon_count, int64_t site_count, const uint8_t *const *rows,
                             const uint8_t *symbol_codes /* [256] or NULL */, int32_t *pattern_count, uint8_t *patterns, double *weights);

/* --- tree: Tree/Node ids, Node_left/right (tree.c:183-224) --- */
int phyamd_set_topology(phyamd_engine *e, const int32_t *left, const int32_t *right /* [2T-1], -1 for tips */, int root);
/* branch length per node id, already multiplied by the clock rate for time trees
 * (treelikelihood.c:1652-1663); the root entry is ignored. */
int phyamd_set_branch_lengths(phyamd_engine *e, const double *lengths /* [2T-1] */);

/* One branch: Node_set_distance + update_nodes[node] = true (treelikelihood.c:73-92).  The next evaluation recomputes only
 * the stored partials on the path from `node` to the root (_calculate_partials' dirty walk, treelikelihood.c:1645-1734)
 * instead of the whole tree; every other setter (and phyamd_set_branch_lengths) implies a full recomputation. */
int phyamd_set_branch_length(phyamd_engine *e, int node, double length);
/* SingleTreeLikelihood_update_all_nodes (treelikelihood.c:1737-1771): force the next evaluation to recompute every node. */
int phyamd_update_all_nodes(phyamd_engine *e);

/* MCMC store / restore (_singleTreeLikelihood_store, _treelikelihood_handle_restore: treelikelihood.c:116-161; the
 * current/stored index pairs of allocate_storage, :947-1005).  phyamd_store evaluates anything pending and remembers the
 * engine's state: branch lengths, eigen system, frequencies, category rates and proportions, lnL, and the stored partials --
 * the first call gives every stored node a second slot (lower-partial memory doubles); evaluations after a store never
 * write the slot the stored state lives in.  phyamd_restore brings that state back without recomputing the tree: nodes
 * point at their stored slots again and only the root is re-integrated on the next evaluation.  The caller does NOT send
 * the old parameters again after a restore.  Not covered: topology, tip data, pattern weights, explicit node matrices
 * (changing the first three drops the stored state; restore then returns PHYAMD_EINVAL).  If slots were reassigned in between
 * (phyamd_set_keep_partials, the lazy rescaling switch) the restored parameters are simply recomputed in full. */
int phyamd_store(phyamd_engine *e);
int phyamd_restore(phyamd_engine *e);

/* --- models: SubstitutionModel eigen system + frequencies, SiteModel rates/proportions --- */
/* eval[S], evec[S][S], ivec[S][S]: m->eigendcmp after update_eigen_system (substmodel.c:1092-1115);
 * P(t) = |evec diag(exp(eval t)) ivec| is formed on the device (substmodel.c:518-557). */
int phyamd_set_eigen(phyamd_engine *e, const double *eval, const double *evec, const double *ivec);
int phyamd_set_frequencies(phyamd_engine *e, const double *freqs /* [S] */);
/* sm->get_rate(c) (includes mu) and sm->get_proportions (sitemodel.c:544-549) */
int phyamd_set_category_rates(phyamd_engine *e, const double *rates /* [C] */, const double *proportions /* [C] */);
/* Optional: explicit matrices [C][S][S] for one node (closed-form models: jc69.c:73-79, hky.c:230-273).
 * Cleared by phyamd_set_eigen. */
int phyamd_set_node_matrices(phyamd_engine *e, int node, const double *matrices);
/* The same for every node at once: matrices [2T-1][C][S][S] by node id (the root's entry is ignored), one upload.  What
 * _calculate_partials does per node with m->p_t (treelikelihood.c:1671-1691) when the caller wants the model's own closed-form
 * arithmetic bit for bit. */
int phyamd_set_matrices(phyamd_engine *e, const double *matrices);
/* Rate matrix Q [S][S] (rows sum to 0, normalised like substmodel.c:1135-1143).  The gradient kernels use
 * (dP/dt) p = Q (P p) instead of a second matrix per branch (the reference's dp_dt, substmodel.c:695-723, is the
 * same product).  phyamd_set_eigen derives Q itself; only explicit-matrix users need this call. */
int phyamd_set_rate_matrix(phyamd_engine *e, const double *Q);

/* --- evaluation --- */
/* lnL = sum_k w_k log L_k: _calculate_simple (treelikelihood.c:1454-1526). */
int phyamd_log_likelihood(phyamd_engine *e, double *lnl);
/* lnL and the per-category branch gradient g[node][c] = sum_k w_k (dL_kc/dt_node) / L_k:
 * update_upper_partials + gradient_cat_branch_lengths (treelikelihood.c:2129-2161, 2793-2941).
 * cat_gradient [2T-1][C]; the root row is 0.  NaN/inf lnL => all-NaN gradient. */
int phyamd_gradient(phyamd_engine *e, int flags, double *lnl, double *cat_gradient);
/* Same, then the epilogue gradient_branch_length_from_cat_inplace (treelikelihood.c:3129-3143):
 * branch_gradient[node] = sum_c g[node][c] w_c r_c  (r_c WITHOUT mu: pass them here).  [2T-1]. */
int phyamd_branch_gradient(phyamd_engine *e, int flags, const double *rates_without_mu /* [C] or NULL */, double *lnl,
                           double *branch_gradient);
/* lnL alone, left on the device (device_out[0]) on the engine's stream: the post-order pass without a host round trip
 * (with PHYAMD_RESCALE_AUTO an unscaled engine still reads lnL back once to test for +-inf, treelikelihood.c:1496-1519). */
int phyamd_log_likelihood_device(phyamd_engine *e, double *device_out);
/* Device-resident result for multi-GPU sharding: writes [lnL, g[0][0..C-1], g[1][..], ...]
 * (1 + (2T-1)*C doubles) to `device_out` on the engine's stream, no host synchronisation. */
int phyamd_gradient_device(phyamd_engine *e, int flags, double *device_out);
/* After an evaluation (rescaled or not): sum_k (w_k / L_k) sum_i pi_i ( p_root[cat 0] - mean of p_root[cat >= 1] ), the only part
 * of the +I site-model gradient that needs O(P) data (gradient_pinv_sitemodel / gradient_pinv_W_sitemodel,
 * treelikelihood.c:2943-3008); the rest of that gradient is O(N C) host arithmetic on phyamd_gradient's output. */
int phyamd_root_invariant_term(phyamd_engine *e, double *out);

/* --- substitution-model gradient: calculate_dlnl_dQ (treelikelihood.c:2337-2583) --- */
#define PHYAMD_MAX_PARAMETERS 2048 /* a 61-state symmetric model has 1830 rates + 61 frequencies */
/* dQ [count][S][S]: derivative of the (normalised) rate matrix with respect to each parameter, what the reference's
 * m->dQ holds after _gtr_dQdp / _hky_dQdp / _general_dQdp (gtr.c:256-326, hky.c:493-541, gensubst.c:216-279).  The engine
 * forms dP/dtheta = U ((U^-1 dQ U) o F(t)) U^-1 per branch and category itself (dPdp_with_dQdp, substmodel.c:469-489).
 * Needs phyamd_set_eigen; count = 0 clears. */
int phyamd_set_rate_matrix_derivatives(phyamd_engine *e, int count, const double *dQ);
/* One post-order + one pre-order pass giving lnL, the per-category branch gradient (cat_gradient may be NULL) and
 *   parameter_gradient[th] = sum_k (w_k / L_k) sum_branches sum_c w_c sum_i pi_i u_i (dP_th p)_i
 * i.e. the branch sum of calculate_dlnl_dQ for all parameters at once (the reference re-walks the tree per parameter).
 * For a frequency parameter add dpi_f/dtheta * phyamd_root_frequency_term()[f] (treelikelihood.c:2370-2401).
 * Works with rescaling; PHYAMD_GRAD_FOLD_ROOT_FREQS is refused (the reference clears include_root_freqs here).
 * 4 states: fused into the pre-order pass.  20 / 60 / 61 states (general K-state matrices of discrete-trait models,
 * dPdp_with_dQdp_general, gensubst.c:284-323): separate kernels on the stored partials -- the first call switches the
 * engine to phyamd_set_keep_partials(1). */
int phyamd_parameter_gradient(phyamd_engine *e, int flags, double *lnl, double *cat_gradient, double *parameter_gradient);
/* Device-resident form for multi-GPU sharding, like phyamd_gradient_device: writes
 * [lnL | g[node][cat] | parameter_gradient[count] | root frequency term[S]]  (1 + (2T-1)*C + count + S doubles, all of them
 * sums over this engine's patterns) to `device_out` on the engine's stream, no host synchronisation. */
int phyamd_parameter_gradient_device(phyamd_engine *e, int flags, double *device_out);
/* After an evaluation: out[f] = sum_k w_k (sum_c w_c p_root[c][k][f]) / (sum_i pi_i sum_c w_c p_root[c][k][i]), f < S:
 * d lnL / d pi_f through the root frequencies alone. */
int phyamd_root_frequency_term(phyamd_engine *e, double *out /* [S] */);
/* The optimiser's fast path (_calculate_uppper / dlnldt_uppper / d2lnldt2_uppper, treelikelihood.c:2196-2335, 2592-2686):
 * lnL and its first two derivatives with respect to the length of ONE branch, evaluated at a TRIAL length from the upper
 * and lower partials that meet on the branch -- O(patterns) work per trial instead of a tree sweep.  Every state count,
 * rescaled evaluations included (their stored partials are anchored on the per-pattern lnL they belong to).
 * The upper partial comes from the last phyamd_gradient if phyamd_set_keep_partials(1) left it resident; otherwise pending
 * changes are evaluated first (single changed branches: only their paths to the root) and the one upper the branch needs is
 * rebuilt by a walk down its path from the root (node_upper / update_upper of the reference, treelikelihood.c:1737-1771) and
 * kept until partials change again -- so the reference's loop "trials of one branch, accept (phyamd_set_branch_length), next
 * branch" (optimizer.c:116-150) costs a path per branch, not a sweep.  Any of lnl / d1 / d2 may be NULL.  The engine's branch
 * lengths are not changed by this call. */
int phyamd_branch_log_likelihood(phyamd_engine *e, int node, double length, double *lnl, double *d1, double *d2);
int phyamd_synchronize(phyamd_engine *e);

/* --- inspection (parity tests, debugging) --- */
int phyamd_get_pattern_log_likelihoods(phyamd_engine *e, double *out /* [P] */);
/* lower (upper=0) or upper (upper=1) partials of a node after the last evaluation, reference layout
 * [C][P][S]. Upper partials exist only after phyamd_gradient with keep_partials enabled. */
int phyamd_get_partials(phyamd_engine *e, int node, int upper, double *out);
int phyamd_get_node_matrices(phyamd_engine *e, int node, int derivative, double *out /* [C][S][S] */);
int phyamd_is_rescaling(phyamd_engine *e);
/* SingleTreeLikelihood_use_rescaling (treelikelihood.c:1410-1423) after construction: PHYAMD_RESCALE_ALWAYS / _NEVER switch
 * at once (the next evaluation recomputes every node), PHYAMD_RESCALE_AUTO keeps the current state and re-arms the lazy switch. */
int phyamd_set_rescaling(phyamd_engine *e, int policy);
/* keep every node's upper partials resident after a gradient call (costs memory; off by default) */
int phyamd_set_keep_partials(phyamd_engine *e, int on);

/* --- measurement --- */
typedef struct {
	double matrices_ms, lower_ms, upper_ms, reduce_ms; /* HIP-event time per kernel family, last evaluation */
	int32_t lower_launches, upper_launches;
	int64_t device_bytes; /* resident device memory of this engine */
	int32_t tiles;         /* pattern tiles per evaluation */
} phyamd_profile;
/* Sums over patterns (lnL, gradient rows of the default 4-state path) are formed over the engine's blocks of 64 patterns in an
 * order fixed by the pattern range alone: the block range is bisected `levels` times (default 3: eight segments), each segment
 * summed in block order, the segments added pairwise.  An engine that holds one half / quarter / eighth of a larger pattern list
 * cut by the same bisection (physher_amd/sharding.py::shard_range, phyamd_create_sharded with 2 / 4 / 8 devices) runs with
 * 2 / 1 / 0 levels; adding the shards' results pairwise then gives bit for bit the one-engine result.  Replaces nothing in the
 * reference (its sums are sequential, treelikelihood.c:1482-1487); SURVEY 8e "deterministic alternative". */
int phyamd_set_reduction_levels(phyamd_engine *e, int levels);
int phyamd_set_profiling(phyamd_engine *e, int on);
int phyamd_get_profile(phyamd_engine *e, phyamd_profile *out);

#ifdef __cplusplus
}
#endif
#endif
