/*
 * physher_device.c -- seam A of INTEGRATION.md as a real, compiled binding: physher's SingleTreeLikelihood on the
 * MI355X engine (include/physher_amd.h), with the reference's object graph, JSON surface, gradient epilogue and
 * wrapper classes left exactly as they are.
 *
 * REFERENCE SIDE OF THE BOUNDARY.  Built against physher's headers where they lie: `make -C integration PHYSHER_SRC=...` for a
 * maintainer, oracle/Makefile into oracle/_ref/libphysher_device.so for this repository's tests; nothing of the reference is
 * copied, nothing under physher_amd/ links this file.
 *
 * How it binds without editing the reference:
 *   - `tlk->calculate` is a function pointer (treelikelihood.h:92, set at treelikelihood.c:1064, used at :165, :327):
 *     SingleTreeLikelihood_enable_device() points it at _calculate_device, the device twin of _calculate /
 *     _calculate_simple (treelikelihood.c:1454-1610).
 *   - the gradient body is a chain of exported functions that libphyc calls through its PLT
 *     (TreeLikelihood_gradient -> update_upper_partials -> TreeLikelihood_calculate_gradient ->
 *     gradient_cat_branch_lengths / gradient_pinv_*_sitemodel / calculate_dlnl_dQ, treelikelihood.c:320-340,
 *     3205-3361).  This library defines the same symbols and sits in front of libphyc in the lookup order (link it
 *     first, or LD_PRELOAD it); each definition serves device-enabled objects and forwards everything else to the
 *     reference's own function (dlsym RTLD_NEXT).  A maintainer would instead add `if (tlk->device) ...` at the top of
 *     those five functions -- same code, no interposition.
 *   - the JSON switch: new_TreeLikelihoodModel_from_json (treelikelihood.c:819-942) gains the key "device"
 *     (true / a GPU count), read here and hidden from json_check_allowed; and, so that callers which never see JSON
 *     (src/phycpp's TreeLikelihoodInterface, physher.cpp:594-629) can be moved too, new_TreeLikelihoodModel honours the
 *     environment variable PHYSHER_DEVICE (=1, or a GPU count).
 *
 * Numbers: the engine is asked for the reference's own arithmetic -- PHYAMD_GRAD_FOLD_ROOT_FREQS whenever the
 * reference has include_root_freqs set, PHYAMD_GRAD_COMPAT_SCALED for rescaled gradients -- so results are the
 * reference's to rounding, quirks included.  PHYSHER_DEVICE_EXACT=1 selects the exact derivatives.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <math.h>
#include <signal.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

#include "phyc/branchmodel.h"
#include "phyc/eigen.h"
#include "phyc/hashtable.h"
#include "phyc/mjson.h"
#include "phyc/parameters.h"
#include "phyc/simplex.h"
#include "phyc/sitemodel.h"
#include "phyc/sitepattern.h"
#include "phyc/substmodel.h"
#include "phyc/tree.h"
#include "phyc/treelikelihood.h"

#include "physher_amd.h"
#include "physher_device.h"

/* ------------------------------------------------------------------------------------------------------------ */
/* one binding per device-enabled SingleTreeLikelihood (a maintainer's version is a member `tlk->device`)          */

typedef struct binding {
	SingleTreeLikelihood *tlk;
	Model *model;
	phyamd_engine *dev;
	int N, T, P, S, C;
	int Sp; /* the engine's state count: S, or the next count it has kernels for (4, 20, 60, 61) -- the extra states are inert */
	bool exact; /* PHYSHER_DEVICE_EXACT: exact derivatives instead of the reference's folded / per-category forms */
	/* what the engine holds now: parameters are pushed only when they differ (update_nodes[] + need_update semantics) */
	bool pushed;
	double *bl, *rates, *props, *freqs, *eigen; /* eigen = eval[S] | evec[S][S] | ivec[S][S] */
	/* the same at the last store (phyamd_store remembers the engine's side) */
	bool stored;
	double *st_bl, *st_rates, *st_props, *st_freqs, *st_eigen;
	bool restore_pending;
	/* 4-state models with a closed-form P(t) and no parameter derivatives (jc69.c:73-79, f81.c): the engine is handed the model's
	 * own m->p_t matrices for every node, as _calculate_partials fills tlk->matrices (treelikelihood.c:1671-1691) -- the
	 * eigen route is the same function of t but rounds differently in the last digits of the off-diagonals (1e-13 relative
	 * for short branches), and the reference's known-answer test pins the clock gradient to 3e-14 */
	bool closed_form;
	double *mats; /* [N][C][S][S] */
	EigenDecomposition *own_eigen; /* decomposition of m->Q (closed-form models keep none; the others' is left to them) */
	int upper_node;                /* Brent fast path: node whose trial lengths are being evaluated (-1 none) */
	int rescale_sent;              /* last policy sent (tlk->scale is the caller's request) */
	/* substitution-parameter gradient of the current evaluation (calculate_dlnl_dQ is asked for one index at a time) */
	unsigned long generation, pgrad_generation;
	double *pgrad, *rootf;
	size_t np;
	bool dq_sent; /* the engine holds dQ/dtheta of the current substitution parameters (they move only with the eigen system) */
	double *scratch; /* 2 * Sp * Sp */
	/* CPU entry points this binding replaced */
	double (*cpu_calculate)(SingleTreeLikelihood *);
	void (*cpu_store)(Model *);
	void (*cpu_handle_restore)(Model *, Model *, int);
	double (*cpu_d2logP)(Model *, const Parameter *);
	struct binding *next;
} binding;

static binding *g_bindings = NULL;

/* objects built without host partial arrays (see "host storage of an object that is going to the device" below) */
static bool is_lean(const SingleTreeLikelihood *tlk);
static void lean_to_full(SingleTreeLikelihood *tlk);
static void lean_unshare(SingleTreeLikelihood *tlk);
static void forget_lean(const SingleTreeLikelihood *tlk);

/* PHYSHER_DEVICE_VERBOSE=1: a line at exit saying how much work went to the device (the tests read it: a run that silently
 * stayed on the CPU kernels would print zeros) */
static unsigned long g_likelihood_calls = 0, g_gradient_calls = 0, g_branch_calls = 0;
static void report_at_exit(void) {
	if (getenv("PHYSHER_DEVICE_VERBOSE"))
		fprintf(stderr, "physher device backend: %lu likelihood, %lu gradient, %lu single-branch evaluations on the device\n", g_likelihood_calls, g_gradient_calls,
		        g_branch_calls);
}

/* PHYSHER_DEVICE_VERBOSE: a crash in a device-enabled process prints where (the GPU box has no debugger) */
static void crash_report(int sig) {
	void *frames[64];
	const char msg[] = "physher device backend: fatal signal, backtrace:\n";
	if (write(2, msg, sizeof msg - 1) < 0) {}
	const int n = backtrace(frames, 64);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}

__attribute__((constructor)) static void install_crash_report(void) {
	if (getenv("PHYSHER_DEVICE_VERBOSE")) signal(SIGSEGV, crash_report);
}

static binding *find_binding(const SingleTreeLikelihood *tlk) {
	for (binding *b = g_bindings; b; b = b->next)
		if (b->tlk == tlk) return b;
	return NULL;
}

static void die_on(int rc, const char *what) {
	if (rc == PHYAMD_OK) return;
	/* the reference's error convention on this path: message + exit (treelikelihood.c:252-253, 1099-1100, 1661) */
	fprintf(stderr, "physher device backend: %s: %s\n", what, phyamd_last_error());
	exit(2);
}

/* Two ways into the reference (see the header comment and integration/physher-device.patch):
 *   interposition (default): this library defines the reference's own symbols and sits in front of libphyc; "the reference's
 *     function" is the next definition in the lookup order;
 *   PHYSHER_DEVICE_PATCHED: the reference carries the hook table of physher-device.patch; the same functions are compiled under
 *     private names (pd_...) and installed in that table when the library is loaded; "the reference's function" is the function
 *     itself, which runs its own body when called back from its hook.  Nothing depends on symbol lookup order then: static
 *     libphyc, -Bsymbolic, -fno-semantic-interposition and LTO builds all work. */
#ifdef PHYSHER_DEVICE_PATCHED
#define HOOK(name) pd_##name
#else
#define HOOK(name) name
#endif

static void *next_symbol(const char *name) {
#ifdef PHYSHER_DEVICE_PATCHED
	void *p = dlsym(RTLD_DEFAULT, name);
#else
	void *p = dlsym(RTLD_NEXT, name);
#endif
	if (!p) {
		fprintf(stderr, "physher device backend: the reference's %s is not behind this library (link order / LD_PRELOAD)\n", name);
		exit(2);
	}
	return p;
}

/* branch length of a node as _calculate_partials forms it (treelikelihood.c:1652-1663) */
static double branch_length_of(SingleTreeLikelihood *tlk, Node *n) {
	if (Node_isroot(n)) return 0.0;
	if (tlk->bm == NULL || !Tree_is_time_mode(tlk->tree)) return Node_distance(n);
	double bl = tlk->bm->get(tlk->bm, n) * Node_time_elapsed(n);
	if (bl < 0) {
		fprintf(stderr, "calculate_partials: %s branch length = %E rate = %f height = %f - parent height [%s]= %f (%f)\n", n->name, bl, tlk->bm->get(tlk->bm, n),
		        Node_height(n), n->parent->name, Node_height(Node_parent(n)), Node_distance(n));
		exit(1);
	}
	return bl;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* parameters -> engine                                                                                           */

/* eval[Sp] | evec[Sp][Sp] | ivec[Sp][Sp]; padding states (Sp > S: a state count without kernels of its own, as the reference's
 * generic kernels take any, treelikelihoodX.c:43-576): eigenvalue 0 with unit eigenvectors -- P(t) is the identity on them, their
 * frequency and every tip partial on them is 0, so nothing enters or leaves them and lnL and all gradients are the S-state model's */
static void flatten_eigen(const binding *b, const EigenDecomposition *eg, double *out) {
	const int S = b->S, Sp = b->Sp;
	memset(out, 0, sizeof(double) * (Sp + 2 * (size_t)Sp * Sp));
	memcpy(out, eg->eval, sizeof(double) * S);
	for (int i = 0; i < Sp; i++) out[Sp + i * Sp + i] = out[Sp + Sp * Sp + i * Sp + i] = 1.0;
	for (int i = 0; i < S; i++)
		for (int j = 0; j < S; j++) {
			out[Sp + i * Sp + j] = eg->evec[i][j];
			out[Sp + Sp * Sp + i * Sp + j] = eg->Invevec[i][j];
		}
}

/* closed-form models: m->p_t per (node, category), all nodes in one upload or the few changed ones node by node */
static void explicit_matrices(binding *b, bool all, const bool *changed) {
	SingleTreeLikelihood *tlk = b->tlk;
	const int N = b->N, S = b->S, C = b->C;
	const size_t msz = (size_t)C * S * S;
	for (int i = 0; i < N; i++) {
		if (!all && !changed[i]) continue;
		for (int c = 0; c < C; c++) tlk->m->p_t(tlk->m, b->bl[i] * b->rates[c], b->mats + i * msz + (size_t)c * S * S);
		if (!all && !Node_isroot(Tree_node(tlk->tree, i))) die_on(phyamd_set_node_matrices(b->dev, i, b->mats + i * msz), "phyamd_set_node_matrices");
	}
	if (all) die_on(phyamd_set_matrices(b->dev, b->mats), "phyamd_set_matrices");
}

/* one branch's length to the engine; closed-form models also send that node's own m->p_t matrices -- every node carries explicit
 * matrices then, which the engine never rebuilds from the eigen system, so a length alone would leave P(old t) in place */
static void send_branch_length(binding *b, int node, double length) {
	SingleTreeLikelihood *tlk = b->tlk;
	b->bl[node] = length;
	die_on(phyamd_set_branch_length(b->dev, node, length), "phyamd_set_branch_length");
	if (b->closed_form && !Node_isroot(Tree_node(tlk->tree, node))) {
		const int S = b->S, C = b->C;
		const size_t msz = (size_t)C * S * S;
		for (int c = 0; c < C; c++) tlk->m->p_t(tlk->m, length * b->rates[c], b->mats + node * msz + (size_t)c * S * S);
		die_on(phyamd_set_node_matrices(b->dev, node, b->mats + node * msz), "phyamd_set_node_matrices");
	}
}

/* Site model, substitution model, frequencies, branch lengths: what _calculate_partials and p_t pull lazily per node on the
 * CPU (treelikelihood.c:1645-1696, substmodel.c:520-523) is pushed once per evaluation, and only what changed. */
static void push_model(binding *b, bool all_nodes) {
	SingleTreeLikelihood *tlk = b->tlk;
	const int N = b->N, S = b->S, C = b->C;
	bool full = !b->pushed;
	bool model_changed = full;                    /* rates, eigen system: every P(t) moves */
	bool *lengths_changed = calloc(N, sizeof(bool)); /* single branches */

	double *rates = b->scratch; /* C <= S*S always holds for the kernels' state counts (C <= 16) */
	for (int c = 0; c < C; c++) rates[c] = tlk->sm->get_rate(tlk->sm, c);
	const double *props = tlk->sm->get_proportions(tlk->sm);
	if (full || memcmp(rates, b->rates, sizeof(double) * C) || memcmp(props, b->props, sizeof(double) * C)) {
		memcpy(b->rates, rates, sizeof(double) * C);
		memcpy(b->props, props, sizeof(double) * C);
		die_on(phyamd_set_category_rates(b->dev, b->rates, b->props), "phyamd_set_category_rates");
		model_changed = true;
	}

	SubstitutionModel *m = tlk->m;
	if (m->need_update || full) {
		/* update_Q + eigen decomposition, the first three lines of _p_t (substmodel.c:520-523); closed-form models
		 * (jc69.c:73-79, hky.c) never decompose on the CPU, the engine always works from an eigen system of Q */
		m->p_t(m, 0.1, b->scratch); /* the model's own refresh, so that it stays coherent for CPU-side users of p_t / dPdp */
		m->update_Q(m);             /* (closed-form p_t never builds Q: jc69.c:73-79) */
		EigenDecomposition *eg = b->own_eigen;
		EigenDecomposition_decompose(m->Q, eg);
		if (eg->failed) {
			fprintf(stderr, "physher device backend: eigen decomposition of %s failed\n", m->name);
			exit(2);
		}
		for (int i = 0; i < S; i++)
			if (eg->evali[i] != 0.0) { /* (a non-reversible Q may have complex pairs; the CPU path works in complex arithmetic then) */
				fprintf(stderr, "physher device backend: %s has complex eigenvalues, the device engine takes real eigen systems\n", m->name);
				exit(2);
			}
		const int Sp = b->Sp;
		const size_t esz = (size_t)Sp + 2 * (size_t)Sp * Sp;
		double *flat = malloc(sizeof(double) * esz);
		flatten_eigen(b, eg, flat);
		if (full || memcmp(flat, b->eigen, sizeof(double) * esz)) {
			memcpy(b->eigen, flat, sizeof(double) * esz);
			die_on(phyamd_set_eigen(b->dev, b->eigen, b->eigen + Sp, b->eigen + Sp + Sp * Sp), "phyamd_set_eigen");
			if (b->closed_form) { /* the gradient's (dP/dt) p = Q (P p) takes the model's Q as it stands, not U L U^-1 */
				for (int i = 0; i < S; i++) memcpy(b->scratch + i * S, m->Q[i], sizeof(double) * S);
				die_on(phyamd_set_rate_matrix(b->dev, b->scratch), "phyamd_set_rate_matrix");
			}
			model_changed = true;
			b->dq_sent = false;
		}
		free(flat);
	}
	const double *freqs = tlk->get_root_frequencies(tlk);
	if (full || memcmp(freqs, b->freqs, sizeof(double) * S)) {
		memcpy(b->freqs, freqs, sizeof(double) * S); /* (entries S .. Sp-1 stay 0) */
		die_on(phyamd_set_frequencies(b->dev, b->freqs), "phyamd_set_frequencies");
	}

	/* branch lengths by node id; few changed ones go one by one so that only their paths to the root are recomputed
	 * (update_nodes[index] = true, treelikelihood.c:73-92), many or all as one vector */
	double *bl = malloc(sizeof(double) * N);
	int changed = 0;
	for (int i = 0; i < N; i++) {
		Node *n = Tree_node(tlk->tree, i);
		bl[Node_id(n)] = branch_length_of(tlk, n);
	}
	for (int i = 0; i < N; i++) changed += bl[i] != b->bl[i];
	if (full || changed > N / 8) {
		memcpy(b->bl, bl, sizeof(double) * N);
		die_on(phyamd_set_branch_lengths(b->dev, b->bl), "phyamd_set_branch_lengths");
		model_changed = true;
	} else {
		for (int i = 0; i < N; i++)
			if (bl[i] != b->bl[i]) {
				b->bl[i] = bl[i];
				lengths_changed[i] = true;
				die_on(phyamd_set_branch_length(b->dev, i, bl[i]), "phyamd_set_branch_length");
			}
		/* SingleTreeLikelihood_update_all_nodes asks for every node even when no value moved (examples/benchmarking.c:498-503) */
		if (all_nodes) die_on(phyamd_update_all_nodes(b->dev), "phyamd_update_all_nodes");
	}
	free(bl);
	if (b->closed_form) explicit_matrices(b, model_changed, lengths_changed);

	const int want = tlk->scale ? PHYAMD_RESCALE_ALWAYS : PHYAMD_RESCALE_AUTO; /* SingleTreeLikelihood_use_rescaling after construction */
	if (want != b->rescale_sent) {
		die_on(phyamd_set_rescaling(b->dev, want), "phyamd_set_rescaling");
		b->rescale_sent = want;
	}
	free(lengths_changed);
	b->pushed = true;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* tlk->calculate on the device: _calculate + _calculate_simple + _calculate_uppper (treelikelihood.c:1454-1610, 2592-2686) */

static double _calculate_device(SingleTreeLikelihood *tlk) {
	binding *b = find_binding(tlk);
	if (!tlk->update) return tlk->lk; /* :1458 */

	const int N = b->N;
	int dirty = 0, last_dirty = -1;
	for (int i = 0; i < N; i++)
		if (tlk->update_nodes[i]) {
			dirty++;
			if (b->upper_node != i) last_dirty = i;
		}

	if (!tlk->sm->update(tlk->sm)) { /* :1462-1466 */
		tlk->lk = NAN;
		return NAN;
	}
	if (Tree_is_time_mode(tlk->tree)) Tree_update_heights(tlk->tree);

	/* Brent on one branch at a time (optimizer.c:116-150 sets use_upper): trial lengths of that branch are single-branch
	 * evaluations from the upper and lower partial that meet on it (_calculate :1558-1606, _calculate_uppper :2592-2686).
	 * One dirty node = a trial on that node; more = the optimiser has moved on and the previous node still carries its flag
	 * (:1589-1605).  Moving on makes the previous branch's accepted length part of the engine's state: the engine then
	 * recomputes its path to the root before it rebuilds the next branch's upper partial. */
	if (tlk->use_upper && b->pushed && dirty >= 1) {
		const int idx = dirty == 1 ? (last_dirty >= 0 ? last_dirty : b->upper_node) : last_dirty;
		if (idx >= 0 && !Node_isroot(Tree_node(tlk->tree, idx))) {
			if (b->upper_node >= 0 && b->upper_node != idx) {
				Node *prev = Tree_node(tlk->tree, b->upper_node);
				send_branch_length(b, b->upper_node, branch_length_of(tlk, prev));
				tlk->update_nodes[b->upper_node] = false; /* :1601 */
			}
			Node *node = Tree_node(tlk->tree, idx);
			die_on(phyamd_branch_log_likelihood(b->dev, idx, branch_length_of(tlk, node), &tlk->lk, NULL, NULL), "phyamd_branch_log_likelihood");
			g_branch_calls++;
			b->upper_node = idx;
			tlk->node_upper = node;
			b->generation++;
			return tlk->lk; /* update and update_nodes[idx] stay set, as on the CPU (:1575-1580) */
		}
	}
	b->upper_node = -1;

	push_model(b, dirty == N);
	die_on(phyamd_log_likelihood(b->dev, &tlk->lk), "phyamd_log_likelihood"); /* lazy rescaling (:1496-1519) happens inside */
	g_likelihood_calls++;
	die_on(phyamd_get_pattern_log_likelihoods(b->dev, tlk->pattern_lk), "phyamd_get_pattern_log_likelihoods");
	b->generation++;
	b->restore_pending = false;

	if (isnan(tlk->lk)) { /* :1489-1495 */
		for (int i = 0; i < N; i++) tlk->update_nodes[i] = true;
		tlk->update = true;
		tlk->update_upper = true;
		return tlk->lk;
	}
	for (int i = 0; i < N; i++) tlk->update_nodes[i] = false;
	tlk->update = false;
	tlk->update_upper = true;
	return tlk->lk;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* gradient: the device twins of the functions TreeLikelihood_calculate_gradient calls (treelikelihood.c:3205-3361) */

static int gradient_flags(const binding *b) {
	if (b->exact) return 0;
	int flags = PHYAMD_GRAD_COMPAT_SCALED; /* only has an effect under rescaling (treelikelihood.c:2851-2870) */
	if (b->tlk->include_root_freqs) flags |= PHYAMD_GRAD_FOLD_ROOT_FREQS; /* :241, 2147-2153 */
	return flags;
}

/* number of substitution parameters gradient_PMatrix walks (:3077-3089) and where the frequencies start (:2363-2366) */
static size_t subst_parameter_count(const SubstitutionModel *m, size_t *rate_count) {
	size_t n = m->rates_simplex == NULL ? Parameters_count(m->rates) : (size_t)m->rates_simplex->K;
	if (m->rates_simplex != NULL && m->grad_wrt_reparam) n--;
	*rate_count = n;
	if (m->simplex != NULL) {
		n += m->simplex->K;
		if (m->grad_wrt_reparam) n--;
	}
	return n;
}

static bool wants_substitution_gradient(const binding *b) {
	const SingleTreeLikelihood *tlk = b->tlk;
	const int f = tlk->prepared_gradient;
	if (tlk->m->dPdp == NULL) return false;
	if (f & (TREELIKELIHOOD_FLAG_SUBSTITUTION_MODEL_UNCONSTRAINED | TREELIKELIHOOD_FLAG_SUBSTITUTION_MODEL)) return true;
	if (f & (TREELIKELIHOOD_FLAG_SUBSTITUTION_MODEL_RATES | TREELIKELIHOOD_FLAG_SUBSTITUTION_MODEL_FREQUENCIES)) {
		const double eps = b->model ? ((Model **)b->model->data)[1]->epsilon : 0.0; /* :3308: finite differences when epsilon > 0 */
		return !(eps > 0.0);
	}
	return false;
}

/* all parameters in one pre-order pass (the CPU does one sweep per parameter, gradient_PMatrix :3077-3110); the branch
 * gradient of the same pass goes to cat_gradient when the caller wants it */
static void device_parameter_gradient(binding *b, double *cat_gradient) {
	SingleTreeLikelihood *tlk = b->tlk;
	SubstitutionModel *m = tlk->m;
	const int S = b->S;
	size_t rate_count;
	const size_t np = subst_parameter_count(m, &rate_count);
	if (np > b->np) {
		b->pgrad = realloc(b->pgrad, sizeof(double) * np);
		b->np = np;
	}
	if (!b->dq_sent) { /* np calls of m->dPdp: once per parameter change, not once per gradient */
		const int Sp = b->Sp;
		double *dQ = calloc(np * (size_t)Sp * Sp, sizeof(double)); /* rows and columns of the padding states stay 0 */
		for (size_t p = 0; p < np; p++) {
			/* m->dQ after the first half of m->dPdp (_gtr_dQdp, _hky_dQdp, _general_dQdp) is d(normalised Q)/d(parameter p) */
			m->dQ_need_update = true;
			m->dPdp(m, (int)p, b->scratch, 0.1);
			const double *src = m->dQ;
			if (m->modeltype == REVERSIBLE || m->modeltype == NONREVERSIBLE) {
				/* the general model's dPdp leaves m->dQ rotated into its eigen basis, U^-1 dQ U (gensubst.c:308-319; the nucleotide
				 * models keep dQ itself, substmodel.c:469-489): rotate it back, the engine takes dQ */
				double *tmp = b->scratch, *raw = b->scratch + (size_t)S * S;
				double **U = m->eigendcmp->evec, **Ui = m->eigendcmp->Invevec;
				for (int i = 0; i < S; i++)
					for (int j = 0; j < S; j++) {
						double v = 0.0;
						for (int k = 0; k < S; k++) v += U[i][k] * m->dQ[(size_t)k * S + j];
						tmp[(size_t)i * S + j] = v;
					}
				for (int i = 0; i < S; i++)
					for (int j = 0; j < S; j++) {
						double v = 0.0;
						for (int k = 0; k < S; k++) v += tmp[(size_t)i * S + k] * Ui[k][j];
						raw[(size_t)i * S + j] = v;
					}
				src = raw;
			}
			for (int i = 0; i < S; i++) memcpy(dQ + p * Sp * Sp + (size_t)i * Sp, src + (size_t)i * S, sizeof(double) * S);
			if (getenv("PHYSHER_DEVICE_DEBUG")) {
				fprintf(stderr, "dQ[%zu] (modeltype %d):", p, (int)m->modeltype);
				for (int i = 0; i < S * S; i++) fprintf(stderr, " %.10g", src[i]);
				fprintf(stderr, "\n");
			}
		}
		m->dQ_need_update = true;
		die_on(phyamd_set_rate_matrix_derivatives(b->dev, (int)np, dQ), "phyamd_set_rate_matrix_derivatives");
		free(dQ);
		b->dq_sent = true;
	}
	double lnl;
	const int flags = b->exact ? 0 : PHYAMD_GRAD_COMPAT_SCALED; /* the reference clears include_root_freqs here (:291-305) */
	die_on(phyamd_parameter_gradient(b->dev, flags, &lnl, cat_gradient, b->pgrad), "phyamd_parameter_gradient");
	g_gradient_calls++;
	/* frequency parameters also move the root distribution (:2370-2401) */
	const double *freqs = tlk->get_root_frequencies(tlk);
	if (np > rate_count && freqs != tlk->root_frequencies) {
		die_on(phyamd_root_frequency_term(b->dev, b->rootf), "phyamd_root_frequency_term");
		double *dphi = calloc(S, sizeof(double));
		for (size_t p = rate_count; p < np; p++) {
			const size_t f = p - rate_count;
			memset(dphi, 0, sizeof(double) * S);
			if (m->grad_wrt_reparam) m->simplex->gradient(m->simplex, f, dphi);
			else dphi[f] = 1.0;
			for (int i = 0; i < S; i++) b->pgrad[p] += dphi[i] * b->rootf[i];
		}
		free(dphi);
	}
	b->pgrad_generation = b->generation;
}

/* TreeLikelihood_calculate_gradient opens with pattern_likelihoods[k] = exp(pattern_lk[k]) over all patterns
 * (treelikelihood.c:3207-3210) for the CPU kernels it then calls; the device twins below never read that array (w_k / L_k stays
 * on the device), and at 1e5 patterns the loop costs as much as the device's whole gradient.  The loop's bound is
 * tlk->sp->count, read per iteration: it is 0 for the duration of the call.  A maintainer writes `if (!tlk->device)` around
 * the loop instead.  (Nothing else the function reaches on a device-enabled object reads the pattern count.) */
void HOOK(TreeLikelihood_calculate_gradient)(Model *model, double *grads) {
	static void (*real)(Model *, double *);
	if (!real) real = next_symbol("TreeLikelihood_calculate_gradient");
	SingleTreeLikelihood *tlk = (SingleTreeLikelihood *)model->obj;
	if (!find_binding(tlk)) {
		real(model, grads);
		return;
	}
	const size_t count = tlk->sp->count;
	tlk->sp->count = 0;
	real(model, grads);
	tlk->sp->count = count;
}

/* SingleTreeLikelihood_update_uppers (treelikelihood.c:1530-1538: what serial_brent_optimize_tree calls first, optimizer.c:125)
 * runs the static CPU pass _calculate_simple and the CPU pre-order pass; on a device object it means "evaluate, then serve
 * single-branch trials": the engine rebuilds the one upper a trial needs on demand */
void HOOK(SingleTreeLikelihood_update_uppers)(SingleTreeLikelihood *tlk) {
	static void (*real)(SingleTreeLikelihood *);
	if (!find_binding(tlk)) {
		if (!real) real = next_symbol("SingleTreeLikelihood_update_uppers");
		real(tlk);
		return;
	}
	tlk->use_upper = false;
	tlk->calculate(tlk);
	tlk->update_upper = false;
	tlk->use_upper = true;
}

void HOOK(update_upper_partials)(SingleTreeLikelihood *tlk, Node *node, bool include_root_freqs) {
	static void (*real)(SingleTreeLikelihood *, Node *, bool);
	if (find_binding(tlk)) return; /* the pre-order pass is fused with the gradient (gradient_cat_branch_lengths below) */
	if (!real) real = next_symbol("update_upper_partials");
	real(tlk, node, include_root_freqs);
}

void HOOK(gradient_cat_branch_lengths)(SingleTreeLikelihood *tlk, double *branch_gradient, const double *pattern_likelihoods) {
	static void (*real)(SingleTreeLikelihood *, double *, const double *);
	binding *b = find_binding(tlk);
	if (!b) {
		if (!real) real = next_symbol("gradient_cat_branch_lengths");
		real(tlk, branch_gradient, pattern_likelihoods);
		return;
	}
	if (wants_substitution_gradient(b)) {
		device_parameter_gradient(b, branch_gradient);
		return;
	}
	double lnl;
	die_on(phyamd_gradient(b->dev, gradient_flags(b), &lnl, branch_gradient), "phyamd_gradient");
	g_gradient_calls++;
}

double HOOK(calculate_dlnl_dQ)(SingleTreeLikelihood *tlk, int index, const double *pattern_likelihoods) {
	static double (*real)(SingleTreeLikelihood *, int, const double *);
	binding *b = find_binding(tlk);
	if (!b) {
		if (!real) real = next_symbol("calculate_dlnl_dQ");
		return real(tlk, index, pattern_likelihoods);
	}
	if (b->pgrad_generation != b->generation || b->pgrad == NULL) device_parameter_gradient(b, NULL);
	return b->pgrad[index];
}

/* the +I term reads the root partial (treelikelihood.c:2943-3008): one O(P) reduction on the device, the rest is the
 * reference's O(N C) arithmetic on the per-category branch gradient */
void HOOK(gradient_pinv_sitemodel)(SingleTreeLikelihood *tlk, const double *branch_gradient, const double *branch_lengths, double *gradient) {
	static void (*real)(SingleTreeLikelihood *, const double *, const double *, double *);
	binding *b = find_binding(tlk);
	if (!b) {
		if (!real) real = next_symbol("gradient_pinv_sitemodel");
		real(tlk, branch_gradient, branch_lengths, gradient);
		return;
	}
	double discrete_grad[2] = {0, 0};
	for (int i = 0; i < b->N; i++) discrete_grad[1] += branch_gradient[i * 2 + 1] * branch_lengths[i];
	die_on(phyamd_root_invariant_term(b->dev, &discrete_grad[0]), "phyamd_root_invariant_term");
	gradient[0] = tlk->sm->derivative(tlk->sm, discrete_grad, Parameters_at(tlk->sm->proportions->parameters, 0));
}

void HOOK(gradient_pinv_W_sitemodel)(SingleTreeLikelihood *tlk, const double *branch_gradient, const double *branch_lengths, double *gradient) {
	static void (*real)(SingleTreeLikelihood *, const double *, const double *, double *);
	binding *b = find_binding(tlk);
	if (!b) {
		if (!real) real = next_symbol("gradient_pinv_W_sitemodel");
		real(tlk, branch_gradient, branch_lengths, gradient);
		return;
	}
	const int C = b->C;
	double *discrete_grad = calloc(C, sizeof(double));
	die_on(phyamd_root_invariant_term(b->dev, &discrete_grad[0]), "phyamd_root_invariant_term");
	for (int i = 0; i < b->N; i++)
		for (int j = 1; j < C; j++) discrete_grad[j] += branch_gradient[i * C + j] * branch_lengths[i];
	gradient[0] = tlk->sm->derivative(tlk->sm, discrete_grad, Parameters_at(tlk->sm->proportions->parameters, 0));
	free(discrete_grad);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Model hooks: second derivative of one branch, store / restore                                                  */

/* _singleTreeLikelihood_d2logP (treelikelihood.c:469-530): d2 lnL / dt2 of the branch whose distance parameter is p */
static double _d2logP_device(Model *self, const Parameter *p) {
	SingleTreeLikelihood *tlk = (SingleTreeLikelihood *)self->obj;
	binding *b = find_binding(tlk);
	Node *node = NULL;
	int i = 0;
	for (; i < b->N; i++) {
		node = Tree_node(tlk->tree, i);
		if (strcmp(node->distance->name, Parameter_name(p)) == 0) break;
	}
	if (i == b->N) return b->cpu_d2logP(self, p); /* not a branch: the reference's finite differences over logP (device) */
	double logP = tlk->calculate(tlk);
	if (isnan(logP) || isinf(logP)) return logP;
	double d2 = NAN;
	die_on(phyamd_branch_log_likelihood(b->dev, Node_id(node), branch_length_of(tlk, node), NULL, NULL, &d2), "phyamd_branch_log_likelihood");
	g_branch_calls++;
	if (isnan(d2)) SingleTreeLikelihood_update_all_nodes(tlk);
	return d2;
}

static void keep_copy(double **dst, const double *src, size_t n) {
	if (!*dst) *dst = malloc(sizeof(double) * n);
	memcpy(*dst, src, sizeof(double) * n);
}

/* _singleTreeLikelihood_store (treelikelihood.c:125-150): the CPU copies its index vectors, the engine remembers its
 * parameters, lnL and stored partials (second slot per stored node on the first call) */
static void _store_device(Model *self) {
	SingleTreeLikelihood *tlk = (SingleTreeLikelihood *)self->obj;
	binding *b = find_binding(tlk);
	if (b->closed_form) { /* phyamd_store covers the eigen route only: from here on P(t) comes from the eigen system of Q */
		b->closed_form = false;
		b->pushed = false;
		SingleTreeLikelihood_update_all_nodes(tlk);
	}
	if (tlk->update) tlk->calculate(tlk); /* the engine stores an evaluated state */
	b->cpu_store(self);
	die_on(phyamd_store(b->dev), "phyamd_store");
	const int S = b->S;
	keep_copy(&b->st_bl, b->bl, b->N);
	keep_copy(&b->st_rates, b->rates, b->C);
	keep_copy(&b->st_props, b->props, b->C);
	keep_copy(&b->st_freqs, b->freqs, b->Sp);
	keep_copy(&b->st_eigen, b->eigen, (size_t)b->Sp + 2 * (size_t)b->Sp * b->Sp);
	b->stored = true;
	b->restore_pending = false;
}

/* _treelikelihood_handle_restore (treelikelihood.c:116-124) is fired once per restored sub-model: the engine goes back
 * once, on the first of them after a store / evaluation */
static void _handle_restore_device(Model *self, Model *model, int index) {
	SingleTreeLikelihood *tlk = (SingleTreeLikelihood *)self->obj;
	binding *b = find_binding(tlk);
	b->cpu_handle_restore(self, model, index);
	if (!b->stored || b->restore_pending) return;
	die_on(phyamd_restore(b->dev), "phyamd_restore");
	const int S = b->S;
	memcpy(b->bl, b->st_bl, sizeof(double) * b->N);
	memcpy(b->rates, b->st_rates, sizeof(double) * b->C);
	memcpy(b->props, b->st_props, sizeof(double) * b->C);
	memcpy(b->freqs, b->st_freqs, sizeof(double) * b->Sp);
	memcpy(b->eigen, b->st_eigen, sizeof(double) * ((size_t)b->Sp + 2 * (size_t)b->Sp * b->Sp));
	b->dq_sent = false;
	b->restore_pending = true; /* cleared by the next store or evaluation; further sub-model restores of this cycle change nothing */
	b->generation++;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* enable / disable                                                                                               */

int SingleTreeLikelihood_on_device(const SingleTreeLikelihood *tlk) { return find_binding(tlk) != NULL; }

int SingleTreeLikelihood_device_is_rescaling(const SingleTreeLikelihood *tlk) {
	binding *b = find_binding(tlk);
	return b ? phyamd_is_rescaling(b->dev) : 0;
}

int SingleTreeLikelihood_enable_device(SingleTreeLikelihood *tlk, Model *model, int device_count, const int *device_ids) {
	if (find_binding(tlk)) return PHYAMD_OK;
	if (tlk->sm->site_category != NULL) {
		fprintf(stderr, "physher device backend: the CAT site model (per-site categories) stays on the CPU kernels\n");
		return PHYAMD_EUNSUPPORTED;
	}
	binding *b = calloc(1, sizeof(binding));
	b->tlk = tlk;
	b->model = model;
	b->N = Tree_node_count(tlk->tree);
	b->T = Tree_tip_count(tlk->tree);
	b->P = tlk->sp->count;
	b->S = tlk->m->nstate;
	b->C = tlk->cat_count;
	b->upper_node = -1;
	const char *ex = getenv("PHYSHER_DEVICE_EXACT");
	b->exact = ex && atoi(ex) != 0;
	const int N = b->N, S = b->S, C = b->C;
	if (S > 61) {
		fprintf(stderr, "physher device backend: %d states (more than 61) stay on the CPU kernels\n", S);
		free(b);
		return PHYAMD_EUNSUPPORTED;
	}
	const int Sp = b->Sp = S <= 4 ? 4 : S <= 20 ? 20 : S <= 60 ? 60 : 61;

	phyamd_config cfg = {b->T, b->P, Sp, C, device_count >= 1 && device_ids ? device_ids[0] : -1, tlk->scale ? PHYAMD_RESCALE_ALWAYS : PHYAMD_RESCALE_AUTO, 0, NULL};
	b->rescale_sent = cfg.rescale;
	int rc = device_count > 1 ? phyamd_create_sharded(&cfg, device_count, device_ids, &b->dev) : phyamd_create(&cfg, &b->dev);
	if (rc != PHYAMD_OK) {
		fprintf(stderr, "physher device backend: %s -- staying on the CPU kernels\n", phyamd_last_error());
		free(b);
		return rc;
	}
	/* node ids are the reference's own (tree.c:183-224): tips 0..T-1, internal nodes T..2T-2 */
	int32_t *left = malloc(sizeof(int32_t) * N), *right = malloc(sizeof(int32_t) * N);
	for (int i = 0; i < N; i++) {
		Node *n = Tree_node(tlk->tree, i);
		left[Node_id(n)] = Node_isleaf(n) ? -1 : Node_id(Node_left(n));
		right[Node_id(n)] = Node_isleaf(n) ? -1 : Node_id(Node_right(n));
	}
	die_on(phyamd_set_topology(b->dev, left, right, Node_id(Tree_root(tlk->tree))), "phyamd_set_topology");
	free(left);
	free(right);
	die_on(phyamd_set_pattern_weights(b->dev, tlk->sp->weights), "phyamd_set_pattern_weights");
	/* tips: state codes ("tipstates": true, kernels K3/K4) or the data type's 0/1 partials (treelikelihood.c:1094-1117) */
	double *tmp = tlk->use_tip_states ? NULL : malloc(sizeof(double) * (size_t)b->P * S);
	double *wide = !tlk->use_tip_states && Sp != S ? calloc((size_t)b->P * Sp, sizeof(double)) : NULL; /* padding states: partial 0 */
	uint8_t *codes = tlk->use_tip_states && Sp != S ? malloc(b->P) : NULL;
	for (int i = 0; i < N; i++) {
		Node *n = Tree_node(tlk->tree, i);
		if (!Node_isleaf(n)) continue;
		const int seq = tlk->mapping[Node_id(n)];
		if (tlk->use_tip_states) {
			const uint8_t *src = tlk->sp->patterns[seq];
			if (codes) { /* a code >= S reads as "all states" (sitepattern.h:68-82): that is code Sp on the engine, not a padding state */
				for (int k = 0; k < b->P; k++) codes[k] = src[k] >= S ? (uint8_t)Sp : src[k];
				src = codes;
			}
			die_on(phyamd_set_tip_states(b->dev, Node_id(n), src), "phyamd_set_tip_states");
		} else {
			tlk->sp->get_partials(tlk->sp, seq, tmp);
			if (wide) {
				for (int k = 0; k < b->P; k++) memcpy(wide + (size_t)k * Sp, tmp + (size_t)k * S, sizeof(double) * S);
				die_on(phyamd_set_tip_partials(b->dev, Node_id(n), wide), "phyamd_set_tip_partials");
			} else
				die_on(phyamd_set_tip_partials(b->dev, Node_id(n), tmp), "phyamd_set_tip_partials");
		}
	}
	free(tmp);
	free(wide);
	free(codes);

	b->bl = calloc(N, sizeof(double));
	b->rates = calloc(C, sizeof(double));
	b->props = calloc(C, sizeof(double));
	b->freqs = calloc(Sp, sizeof(double));
	b->eigen = calloc((size_t)Sp + 2 * (size_t)Sp * Sp, sizeof(double));
	b->rootf = calloc(Sp, sizeof(double));
	b->scratch = calloc((size_t)2 * Sp * Sp + 16, sizeof(double));
	b->own_eigen = new_EigenDecomposition(S);
	/* (models with parameter derivatives -- HKY: dPdp -- stay on the eigen route: the engine differentiates U F(t) U^-1) */
	b->closed_form = S == 4 && tlk->m->dPdp == NULL && (tlk->m->modeltype == JC69 || tlk->m->modeltype == K80 || (tlk->m->name && strcasecmp(tlk->m->name, "F81") == 0)) &&
	                 !getenv("PHYSHER_DEVICE_EIGEN_ONLY");
	if (b->closed_form) b->mats = calloc((size_t)N * C * S * S, sizeof(double));

	b->cpu_calculate = tlk->calculate;
	tlk->calculate = _calculate_device;
	if (model) {
		b->cpu_store = model->store;
		b->cpu_handle_restore = model->handle_restore;
		b->cpu_d2logP = model->d2logP;
		model->store = _store_device;
		model->handle_restore = _handle_restore_device;
		model->d2logP = _d2logP_device;
	}
	SingleTreeLikelihood_update_all_nodes(tlk);
	if (!g_bindings && !g_likelihood_calls) atexit(report_at_exit);
	b->next = g_bindings;
	g_bindings = b;
	return PHYAMD_OK;
}

/* unlink the binding and free what it owns; `alive`: tlk and its sub-models are still usable (not inside their destructor) */
static void release_binding(SingleTreeLikelihood *tlk, bool alive) {
	binding **pp = &g_bindings;
	for (; *pp && (*pp)->tlk != tlk; pp = &(*pp)->next) {}
	binding *b = *pp;
	if (!b) return;
	*pp = b->next;
	if (alive) {
		tlk->calculate = b->cpu_calculate;
		if (b->model) {
			b->model->store = b->cpu_store;
			b->model->handle_restore = b->cpu_handle_restore;
			b->model->d2logP = b->cpu_d2logP;
		}
	}
	phyamd_destroy(b->dev);
	if (b->own_eigen) free_EigenDecomposition(b->own_eigen);
	double *owned[] = {b->mats, b->bl, b->rates, b->props, b->freqs, b->eigen, b->st_bl, b->st_rates, b->st_props, b->st_freqs, b->st_eigen, b->pgrad, b->rootf, b->scratch};
	for (size_t i = 0; i < sizeof owned / sizeof owned[0]; i++) free(owned[i]);
	free(b);
	if (alive) {
		if (is_lean(tlk)) lean_to_full(tlk);
		SingleTreeLikelihood_update_all_nodes(tlk);
	}
}

void SingleTreeLikelihood_disable_device(SingleTreeLikelihood *tlk) { release_binding(tlk, true); }

/* _treeLikelihood_model_free (treelikelihood.c:694-713) frees the tree, the models and the site pattern BEFORE it calls this:
 * nothing of tlk's object graph may be touched here */
void HOOK(free_SingleTreeLikelihood_internals)(SingleTreeLikelihood *tlk) {
	static void (*real)(SingleTreeLikelihood *);
	if (!real) real = next_symbol("free_SingleTreeLikelihood_internals");
	release_binding(tlk, false);
	if (is_lean(tlk)) {
		lean_unshare(tlk);
		forget_lean(tlk);
	}
	real(tlk);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* switches: JSON key "device" and the environment variable PHYSHER_DEVICE                                        */

static int g_json_decision = -1; /* >= 0 while new_TreeLikelihoodModel_from_json is building an object: the JSON key wins */

/* ------------------------------------------------------------------------------------------------------------ */
/* host storage of an object that is going to the device                                                          */
/* new_SingleTreeLikelihood gives every node a [C][P][S] array, twice (lower and upper partials), and fills and      */
/* category-replicates one per tip when "tipstates" is off (treelikelihood.c:947-1005, 1047, 1106-1117): 256 GB of  */
/* address space and 128 GB of touched pages at 1000 taxa x 1e6 patterns x 4 categories, none of which a             */
/* device-enabled object reads.  allocate_storage is an exported function (treelikelihood.c:68), so the object is    */
/* built LEAN when the device has already been asked for (JSON key seen, or PHYSHER_DEVICE set; PHYSHER_DEVICE_LEAN=0 */
/* opts out): no arrays for internal nodes and uppers, ONE scratch array shared by all tips (the constructor's       */
/* get_partials writes land there; SingleTreeLikelihood_enable_device sends each tip to the engine from the site     */
/* pattern itself).  Disabling the device on a live object gives it the reference's full storage back.               */

typedef struct lean_rec {
	SingleTreeLikelihood *tlk;
	struct lean_rec *next;
} lean_rec;
static lean_rec *g_lean = NULL;

static bool is_lean(const SingleTreeLikelihood *tlk) {
	for (lean_rec *r = g_lean; r; r = r->next)
		if (r->tlk == tlk) return true;
	return false;
}

static void forget_lean(const SingleTreeLikelihood *tlk) {
	for (lean_rec **pp = &g_lean; *pp; pp = &(*pp)->next)
		if ((*pp)->tlk == tlk) {
			lean_rec *r = *pp;
			*pp = r->next;
			free(r);
			return;
		}
}

static bool device_wanted_now(void) {
	const char *lean = getenv("PHYSHER_DEVICE_LEAN");
	if (lean && atoi(lean) == 0) return false;
	if (g_json_decision >= 0) return g_json_decision > 0;
	const char *env = getenv("PHYSHER_DEVICE");
	return env && atoi(env) > 0;
}

/* one table of a lean object: every tip that holds partials points at `shared`, everything else is NULL */
static double **lean_table(SingleTreeLikelihood *tlk, double *shared) {
	double **rows = calloc(tlk->partials_dim, sizeof(double *));
	if (!tlk->use_tip_states)
		for (int i = 0; i < Tree_node_count(tlk->tree); i++) {
			Node *n = Tree_node(tlk->tree, i);
			if (Node_isleaf(n)) rows[Node_id(n)] = shared;
		}
	return rows;
}

static double *lean_scratch(const SingleTreeLikelihood *tlk) {
	void *p = NULL;
	if (tlk->use_tip_states) return NULL;
	if (posix_memalign(&p, 16, sizeof(double) * (size_t)tlk->partials_size) != 0) {
		fprintf(stderr, "physher device backend: out of host memory for the tip scratch array\n");
		exit(2);
	}
	return p;
}

static double **small_matrices(const SingleTreeLikelihood *tlk) {
	double **m = malloc(sizeof(double *) * tlk->matrix_dim);
	for (int i = 0; i < tlk->matrix_dim; i++) {
		void *p = NULL;
		if (posix_memalign(&p, 16, sizeof(double) * (size_t)tlk->matrix_size * tlk->sm->cat_count) != 0) exit(2);
		m[i] = p;
	}
	return m;
}

void HOOK(allocate_storage)(SingleTreeLikelihood *tlk, size_t index) {
	static void (*real)(SingleTreeLikelihood *, size_t);
	if (!real) real = next_symbol("allocate_storage");
	if (index == 0 ? !device_wanted_now() : !is_lean(tlk)) {
		real(tlk, index);
		return;
	}
	const size_t nodes = Tree_node_count(tlk->tree);
	if (index == 0) {
		lean_rec *r = malloc(sizeof(lean_rec));
		r->tlk = tlk;
		r->next = g_lean;
		g_lean = r;
		tlk->current_matrices_indexes = calloc(2 * nodes, sizeof(unsigned));
		tlk->current_partials_indexes = calloc(2 * nodes, sizeof(unsigned));
		tlk->stored_matrices_indexes = NULL;
		tlk->stored_partials_indexes = NULL;
		tlk->partials = malloc(2 * sizeof(double **));
		tlk->partials[0] = lean_table(tlk, lean_scratch(tlk));
		tlk->partials[1] = NULL;
		tlk->matrices = malloc(2 * sizeof(double **));
		tlk->matrices[0] = small_matrices(tlk);
		tlk->matrices[1] = NULL;
	} else { /* first store (treelikelihood.c:125-137) */
		tlk->stored_matrices_indexes = calloc(2 * nodes, sizeof(unsigned));
		tlk->stored_partials_indexes = calloc(2 * nodes, sizeof(unsigned));
		tlk->partials[1] = lean_table(tlk, lean_scratch(tlk));
		tlk->matrices[1] = small_matrices(tlk);
	}
}

/* the destructor frees every non-NULL row: leave the shared array in one of them */
static void lean_unshare(SingleTreeLikelihood *tlk) {
	for (int t = 0; t < 2; t++) {
		double **rows = tlk->partials[t];
		if (!rows) continue;
		double *shared = NULL;
		for (int i = 0; i < tlk->partials_dim; i++) {
			if (!rows[i]) continue;
			if (!shared) shared = rows[i];
			else if (rows[i] == shared) rows[i] = NULL;
		}
	}
}

/* a lean object goes back to the CPU kernels: the reference's own storage, tips refilled as its constructor fills them */
static void lean_to_full(SingleTreeLikelihood *tlk) {
	static void (*real)(SingleTreeLikelihood *, size_t);
	if (!real) real = next_symbol("allocate_storage");
	const bool had_store = tlk->partials[1] != NULL;
	lean_unshare(tlk);
	for (int t = 0; t < 2; t++) {
		if (!tlk->partials[t]) continue;
		for (int i = 0; i < tlk->partials_dim; i++) free(tlk->partials[t][i]);
		free(tlk->partials[t]);
		for (int i = 0; i < tlk->matrix_dim; i++) free(tlk->matrices[t][i]);
		free(tlk->matrices[t]);
	}
	free(tlk->partials);
	free(tlk->matrices);
	free(tlk->current_matrices_indexes);
	free(tlk->current_partials_indexes);
	free(tlk->stored_matrices_indexes);
	free(tlk->stored_partials_indexes);
	forget_lean(tlk);
	real(tlk, 0);
	if (had_store) real(tlk, 1);
	if (!tlk->use_tip_states)
		for (int t = 0; t < (had_store ? 2 : 1); t++)
			for (int i = 0; i < Tree_node_count(tlk->tree); i++) {
				Node *n = Tree_node(tlk->tree, i);
				if (!Node_isleaf(n)) continue;
				double *row = tlk->partials[t][Node_id(n)];
				tlk->sp->get_partials(tlk->sp, tlk->mapping[Node_id(n)], row);
				for (size_t k = 1; k < (size_t)tlk->cat_count; k++)
					memcpy(row + k * tlk->m->nstate * tlk->pattern_count, row, sizeof(double) * tlk->m->nstate * tlk->pattern_count);
			}
}

static void enable_n(SingleTreeLikelihood *tlk, Model *model, int n) {
	if (n <= 0) return;
	/* PHYSHER_DEVICE_IDS=0,2,5: the GPUs to shard over (default 0 .. n-1; an ordinal may repeat) */
	int ids[64], nids = 0;
	const char *list = getenv("PHYSHER_DEVICE_IDS");
	if (list) {
		char *tmp = strdup(list), *save = NULL;
		for (char *tok = strtok_r(tmp, ",", &save); tok && nids < 64; tok = strtok_r(NULL, ",", &save)) ids[nids++] = atoi(tok);
		free(tmp);
		if (nids != n) {
			fprintf(stderr, "physher device backend: PHYSHER_DEVICE_IDS lists %d devices, %d were asked for\n", nids, n);
			exit(2);
		}
	}
	if (SingleTreeLikelihood_enable_device(tlk, model, n, nids ? ids : NULL) != PHYAMD_OK) {
		/* asked for explicitly: a silent CPU run would be mistaken for a device run */
		fprintf(stderr, "physher device backend: the device was requested and could not be enabled\n");
		exit(2);
	}
}

Model *HOOK(new_TreeLikelihoodModel)(const char *name, SingleTreeLikelihood *tlk, Model *tree, Model *m, Model *sm, Model *bm) {
	static Model *(*real)(const char *, SingleTreeLikelihood *, Model *, Model *, Model *, Model *);
	if (!real) real = next_symbol("new_TreeLikelihoodModel");
	Model *model = real(name, tlk, tree, m, sm, bm);
	const char *env = getenv("PHYSHER_DEVICE");
	if (g_json_decision < 0 && env) enable_n(tlk, model, atoi(env));
	return model;
}

Model *HOOK(new_TreeLikelihoodModel_from_json)(json_node *node, Hashtable *hash) {
	static Model *(*real)(json_node *, Hashtable *);
	if (!real) real = next_symbol("new_TreeLikelihoodModel_from_json");
	/* "device": true | false | <number of GPUs>; taken out of the node while the reference checks its allowed keys (:820-832) */
	json_node *dev = NULL;
	size_t at = 0;
	for (size_t i = 0; i < node->child_count; i++)
		if (strcasecmp(node->children[i]->key, "device") == 0) {
			dev = node->children[i];
			at = i;
		}
	if (dev) {
		memmove(node->children + at, node->children + at + 1, sizeof(json_node *) * (node->child_count - at - 1));
		node->child_count--;
	}
	const int saved = g_json_decision;
	g_json_decision = dev ? atoi((char *)dev->value) : -1;
	const int n = g_json_decision;
	Model *model = real(node, hash);
	g_json_decision = saved;
	if (dev) {
		memmove(node->children + at + 1, node->children + at, sizeof(json_node *) * (node->child_count - at));
		node->children[at] = dev;
		node->child_count++;
		enable_n((SingleTreeLikelihood *)model->obj, model, n);
	}
	return model;
}

#ifdef PHYSHER_DEVICE_PATCHED
/* the table of physher-device.patch, installed when this library is loaded */
static const TreeLikelihoodDeviceHooks g_hooks = {
    pd_new_TreeLikelihoodModel, pd_new_TreeLikelihoodModel_from_json, pd_allocate_storage, pd_free_SingleTreeLikelihood_internals,
    pd_SingleTreeLikelihood_update_uppers, pd_update_upper_partials, pd_calculate_dlnl_dQ, pd_gradient_cat_branch_lengths,
    pd_gradient_pinv_sitemodel, pd_gradient_pinv_W_sitemodel, pd_TreeLikelihood_calculate_gradient};
__attribute__((constructor)) static void install_hooks(void) { treelikelihood_device_hooks = &g_hooks; }
#endif
