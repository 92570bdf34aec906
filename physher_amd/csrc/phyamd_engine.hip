// phyamd_engine.hip -- MI355X (gfx950) tree-likelihood engine behind include/physher_amd.h.
//
// Replaces the CPU hot path of physher's SingleTreeLikelihood (src/phyc/treelikelihood.c and the
// per-state-count kernel files) with hand-written HIP kernels.  Design:
//   * one thread owns one site pattern and loops over the rate categories, so per-pattern work
//     (rescaling max, mixture denominators, weights) never leaves the thread;
//   * transition matrices P(t) and dP/dt are built on the device from the cached eigen system and
//     reach the kernels through wave-uniform (scalar) loads;
//   * the tree is executed level by level: one launch covers every node of a level
//     (blockIdx.y = node, blockIdx.x = pattern block), so launch count = tree height, not node count;
//   * the pre-order pass computes BOTH children's upper partials from one read of the parent's upper
//     and fuses the branch-length gradient (dP/dt contraction, state sum, 1/L_k, weights, pattern
//     reduction) into the same kernel: the reference's spare_partials array never exists;
//   * reductions are fixed-order (wave shuffle -> LDS -> per-block slab -> reduction kernel), so
//     results are bitwise reproducible run to run.
//
// gfx950 only; no CUDA/compat paths.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "physher_amd.h"

#define PHYAMD_ABI_VERSION 3

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}

#define HIP_TRY(expr)                                                                                       \
	do {                                                                                                    \
		hipError_t err__ = (expr);                                                                          \
		if (err__ != hipSuccess)                                                                            \
			return fail(err__ == hipErrorOutOfMemory ? PHYAMD_ENOMEM : PHYAMD_EDEVICE, "%s: %s (%s:%d)", #expr, \
			            hipGetErrorString(err__), __FILE__, __LINE__);                                      \
	} while (0)

constexpr int WAVE = 64;             // lanes per wavefront (gfx950)
// tuning constants (A/B measured on MI355X at 1000 taxa x 1e6 patterns): pattern groups swept
// sequentially by one workgroup, and the occupancy the pre-order kernel is compiled for
#ifndef PHYAMD_PPT_LOWER
#define PHYAMD_PPT_LOWER 2
#endif
#ifndef PHYAMD_PPT_UPPER
#define PHYAMD_PPT_UPPER 8
#endif
#ifndef PHYAMD_UPPER_MIN_WAVES
#define PHYAMD_UPPER_MIN_WAVES 6
#endif
// patterns per thread of the tree-walk kernels are chosen per engine from the shard size (phyamd_create)
constexpr int PPT_LOWER = PHYAMD_PPT_LOWER, PPT_UPPER = PHYAMD_PPT_UPPER;
constexpr int MAX_WAVES = 16;        // 1024 threads
constexpr double SCALING_THRESHOLD = 1.0e-40;  // treelikelihood.c:1121

// how a child's lower partial is obtained: read (CORE), or recomputed in registers from tips (CHERRY, CHERRY_TIP: "fringe",
// which also has its inner branch gradients computed inside its parent's pre-order op) or from two fringe / tip children
// (DEEP: no stored lower either, but a pre-order op of its own)
enum { CH_TIP = 0, CH_CORE = 1, CH_DEEP = 2, CH_CHERRY = 3, CH_CHERRY_TIP = 4 };

// children of a DEEP node, by node id: what child_message needs to rebuild its partial
struct DeepDesc {
	int32_t kind_left, left, lt0, lt1, lt2, linner;
	int32_t kind_right, right, rt0, rt1, rt2, rinner;
};

struct NodeOp {
	int32_t parent;  // lower pass: destination node; upper pass: the node whose children are produced
	int32_t left, right;
	int32_t upper_slot_parent;  // upper pass: slot of the parent's upper (-1: parent is the root)
	int32_t upper_slot_left, upper_slot_right;  // slots to write (-1: nothing stored)
	int32_t core_parent, core_left, core_right;  // index of the stored lower array (-1: tip or fused fringe node)
	int32_t kind_left, kind_right;               // CH_*
	int32_t lt0, lt1, lt2, linner;               // left fringe: cherry tips, outer tip, inner cherry node
	int32_t rt0, rt1, rt2, rinner;               // right fringe
	// tree-walk kernels only (ops in depth-first order, values handed from one op to the next in registers):
	int32_t carry_in;   // lower walk: 1 / 2 = the left / right child's partial is the previous op's result;
	                    // upper walk: 1 = the parent's upper is the previous op's carried child upper
	int32_t carry_out;  // upper walk: 1 / 2 = the left / right child's upper goes to the next op in registers (not stored)
};

#include "phyamd_device.inc"
#include "phyamd_level4.inc"
#include "phyamd_walk4.inc"
#include "phyamd_general.inc"
#include "phyamd_patterns.inc"

}  // namespace

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------

struct phyamd_engine {
	phyamd_config cfg{};
	int T = 0, N = 0, P = 0, S = 0, C = 0, root = -1;
	int G = 1;  // pattern groups (waves along z) per workgroup
	bool generic = false;  // S != 4: MFMA kernels, plane layout [C][S][Pp]
	int Pp = 0;            // padded plane stride (generic)
	int nblk_root = 0;     // workgroups of k_root_finish (generic)
	int nblk_lower = 0;    // pattern blocks of the post-order kernels
	int nblk_walk = 0, nblk_walk_upper = 0;  // pattern blocks of the tree-walk kernels
	int ppt_walk_lower = 2;  // patterns per thread of the post-order walk
	int lnl_blocks = 0;    // entries of d_lnl_part the last post-order pass wrote
	int grad_blocks = 0;   // entries per row of d_gpart the last pre-order pass wrote
	size_t gpart_row = 0;  // allocated entries per row
	// incremental (dirty-node) post-order updates: D1, treelikelihood.c:73-114, 1645-1734
	bool lower_valid = false;         // stored lower partials are those of the current inputs except for `changed` branches
	bool all_dirty = true;            // something other than single branch lengths changed: recompute every node
	bool incremental_pass = false;
	std::vector<int> changed;         // nodes whose branch length changed since the last evaluation
	std::vector<NodeOp> inc_ops;      // ops of the dirty core nodes, by level
	std::vector<int> inc_level_off;
	NodeOp *d_inc_ops = nullptr;
	const std::vector<int> *act_level_off = nullptr;
	NodeOp *act_lower_ops = nullptr;
	bool level_upper_needed = false;  // a level-schedule pre-order pass (parameter gradients) has been requested
	// MCMC store / restore (_singleTreeLikelihood_store, _treelikelihood_handle_restore: treelikelihood.c:116-161): after a
	// store every stored node has two slots (slot = core index, + core_count for the second); an evaluation never writes the
	// slot the stored state lives in, so restore is an index flip (plus re-integrating the root), not a recomputation
	struct Stored {
		bool valid = false;
		std::vector<double> lengths, model, freqs, rates, props;
		std::vector<int32_t> core_index;  // node -> slot of the stored state
		bool have_eigen = false, scaling_on = false;
		unsigned long epoch = 0;
		double lnl = 0.0;
	} stored;
	// pattern tiling (cfg.max_device_bytes): P = patterns per tile (what every kernel sees), Ptot = the caller's count;
	// tip data, weights and per-pattern lnL of all tiles stay resident, the partial arrays are reused tile after tile
	int Ptot = 0, tiles = 1;
	bool tiled_eval_done = false;
	bool tiled_root_term = false;     // d_result holds the summed root frequency term of a tiled parameter gradient
	uint8_t *d_tip_all = nullptr;      // [T][Ptot]
	double *d_weights_all = nullptr, *d_plk_all = nullptr, *d_total = nullptr;
	unsigned long schedule_epoch = 0;  // bumped whenever slots are reassigned from scratch
	bool two_slots = false;            // d_lower / d_lscale hold 2 * core_count slots
	bool force_root = false;           // the root's outputs (lnL_k, w_k / L_k, lnL) belong to a discarded state
	bool walk_params_on = true;  // parameter gradients through the tree walk (PHYAMD_WALK_PARAMS = 0: level kernels)
	bool walk_lower_on = true, walk_upper_on = true;  // A/B switches (PHYAMD_WALK_LOWER / PHYAMD_WALK_UPPER = 0)
	bool walk_enabled = true, walking = false;  // tree-walk kernels (4 states, unscaled, not keep_partials)
	std::vector<NodeOp> walk_lower_ops, walk_upper_ops;  // depth-first op orders
	NodeOp *d_walk_lower_ops = nullptr, *d_walk_upper_ops = nullptr;
	int walk_upper_slots = 0;
	double *d_Lc = nullptr;  // [C][P] per-category site likelihoods at the root (generic)
	double *d_imgs = nullptr;  // generic: MFMA fragment images of P(t) per (node, category), then of Q (k_matrix_images)
	bool qimg_dirty = true;
	double *d_inv_part = nullptr;  // partial sums of k_root_invariant_term
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;

	std::vector<int32_t> left, right, parent;
	std::vector<double> lengths, model, freqs, rates, props;
	std::vector<uint8_t> explicit_host;
	bool have_topology = false, have_lengths = false, have_eigen = false, have_freqs = false, have_rates = false, have_weights = false;
	std::vector<uint8_t> tip_set;
	bool matrices_dirty = true;
	bool scaling_on = false;
	bool keep_partials = false;
	bool profiling = false;
	bool upper_valid = false;
	bool prof_pending = false, prof_with_upper = false;

	// schedule
	std::vector<NodeOp> lower_ops, upper_ops;
	std::vector<int> lower_level_off, upper_level_off;  // offsets into the op arrays, one past the last at the end
	std::vector<int32_t> upper_slot;                    // node -> slot of its upper partial in the last schedule (-1 none)
	int upper_slots = 0;
	std::vector<int32_t> core_index;  // node -> index of its stored lower array (-1: tip or fused)
	int core_count = 0;
	bool fusion_enabled = true, fused = false;
	bool deep_enabled = true;          // PHYAMD_DEEP = 0: every node above the fringe is stored
	std::vector<DeepDesc> deep_host;   // by node id (only DEEP nodes filled)
	int deep_count = 0;
	// (device copy: behind the tip-message table, see Ctx4::deep)
	size_t lower_alloc_cores = 0;

	// device memory
	uint8_t *d_tipmask = nullptr;
	// 20 / 60 / 61 states: tip code S + 1 + q = ambiguity set q, one bit per member state (tip partials that are neither
	// one state nor all states: named sets of a general data type, states of a padded state space)
	unsigned long long *d_tipsets = nullptr;
	std::vector<unsigned long long> tipsets_host;
	double *d_lower = nullptr, *d_upper = nullptr, *d_mats = nullptr, *d_dmats = nullptr;
	double *d_Q = nullptr;
	double *d_Qpi = nullptr;          // diag(pi) Q: the tree-walk gradient contracts u with (pi o Q b) in one mat-vec (4 states)
	std::vector<double> Q_host;
	bool qpi_dirty = true;
	bool have_Q = false;
	double *d_tiptab = nullptr;  // [T][C][16][4] tip messages (4-state)
	// substitution-parameter gradient (G2)
	int np = 0;                      // number of dQ/dtheta matrices set
	std::vector<double> dQ_host;     // [np][S][S]
	double *d_B = nullptr;           // [np][S][S]  U^-1 dQ U
	double *d_dpm = nullptr;         // [np][N][C][S][S]
	double *d_dptab = nullptr;       // [np][T][C][16][4]
	double *d_ppart = nullptr;       // [np][upper ops][nblk] per-workgroup parameter sums, then [np][upper ops]
	double *d_Bw = nullptr;          // tree-walk G2: [np][16] U^-1 dQ U
	double *d_pbuf = nullptr;        // tree-walk G2: [UTpi 16 | Uinv 16 | utab 64]
	double *d_Fw = nullptr;          // tree-walk G2: [N][C][20] w_c F_ab(t_n r_c), l_a e^{l_a t_n r_c}
	double *d_gacc = nullptr;        // tree-walk G2: [16][slabs * C] per-wave eigen-basis sums, then [16] totals
	// phyamd_branch_log_likelihood without resident uppers: the one upper it needs is rebuilt by a root-to-node path walk
	std::vector<int> node_kind;      // CH_* of every node in the current schedule
	PathStep *d_path_steps = nullptr;
	double *d_path_upper = nullptr, *d_path_tmp = nullptr, *d_path_lower = nullptr;  // one node partial each
	int path_node = -1;              // node whose upper d_path_upper holds (-1: none); dropped whenever partials are recomputed
	double *d_branch = nullptr;      // phyamd_branch_log_likelihood: [C][3][16] matrices | [3][blocks] partial sums | [3]
	bool upper_fold = false;         // the stored uppers carry the root frequencies (last gradient call used FOLD)
	double *d_rf_part = nullptr;     // [S][blocks] partial sums of k_root_frequency_term, then [S]
	double *d_gen_scratch = nullptr; // rescaled S != 4 path: per-level maxima / numerators / denominators
	size_t gen_scratch_alloc = 0;
	size_t np_alloc = 0, np_alloc_B = 0, ppart_alloc = 0;
	// 20 / 60 / 61 states (k_param_*_gen): branch nodes, node -> stored lower index, per-branch site likelihoods, G tables
	int *d_pg_nodes = nullptr, *d_pg_core = nullptr;
	double *d_pg_den = nullptr, *d_pg_Gw = nullptr, *d_pg_B = nullptr;
	size_t pg_np_alloc = 0;
	bool params_dirty = true;
	double *d_model = nullptr, *d_freqs = nullptr, *d_rates = nullptr, *d_props = nullptr, *d_lengths = nullptr, *d_weights = nullptr;
	double *d_wl = nullptr;  // [P] w_k / L_k from the root kernel (unscaled evaluations)
	double *d_plk = nullptr, *d_lscale = nullptr, *d_lnl_part = nullptr, *d_gpart = nullptr, *d_result = nullptr;
	uint8_t *d_explicit = nullptr, *d_row_valid = nullptr;
	NodeOp *d_lower_ops = nullptr, *d_upper_ops = nullptr;
	double *h_result = nullptr;  // pinned
	int nblk = 0;
	int64_t device_bytes = 0;
	size_t upper_alloc_slots = 0;

	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	phyamd_profile prof{};
};

namespace {

template <typename Tp>
int dev_alloc(phyamd_engine *e, Tp **p, size_t count) {
	if (count == 0) count = 1;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(Tp)));
	e->device_bytes += (int64_t)(count * sizeof(Tp));
	return PHYAMD_OK;
}

template <typename Tp>
void dev_free(phyamd_engine *e, Tp **p, size_t count) {
	if (*p) {
		(void)hipFree(*p);
		e->device_bytes -= (int64_t)((count ? count : 1) * sizeof(Tp));
		*p = nullptr;
	}
}

int bind_device(phyamd_engine *e) {
	HIP_TRY(hipSetDevice(e->device));
	return PHYAMD_OK;
}

size_t node_partial_doubles(const phyamd_engine *e) { return (size_t)e->C * e->S * (e->generic ? e->Pp : e->P); }

// Build the level schedule and the upper-slot assignment.
int build_schedule(phyamd_engine *e) {
	const int N = e->N, T = e->T;
	e->parent.assign(N, -1);
	std::vector<int> seen(N, 0);
	for (int n = 0; n < N; n++) {
		const int l = e->left[n], r = e->right[n];
		if (n < T) {
			if (l != -1 || r != -1) return fail(PHYAMD_EINVAL, "node %d is a tip (id < tip_count) but has children", n);
			continue;
		}
		if (l < 0 || r < 0 || l >= N || r >= N || l == r) return fail(PHYAMD_EINVAL, "internal node %d has invalid children (%d, %d)", n, l, r);
		if (++seen[l] > 1 || ++seen[r] > 1) return fail(PHYAMD_EINVAL, "node %d or %d has two parents", l, r);
		e->parent[l] = n;
		e->parent[r] = n;
	}
	if (e->root < T || e->root >= N || e->parent[e->root] != -1) return fail(PHYAMD_EINVAL, "root %d is not a parentless internal node", e->root);
	// depth (root = 0) by a stack walk; also detects unreachable nodes / cycles
	std::vector<int> depth(N, -1), order;
	order.reserve(N);
	std::vector<int> stack{e->root};
	depth[e->root] = 0;
	while (!stack.empty()) {
		const int n = stack.back();
		stack.pop_back();
		order.push_back(n);
		if (n >= T) {
			for (int ch : {e->left[n], e->right[n]}) {
				depth[ch] = depth[n] + 1;
				stack.push_back(ch);
			}
		}
	}
	if ((int)order.size() != N) return fail(PHYAMD_EINVAL, "topology is not a single binary tree over all %d nodes", N);
	// Fringe classification (4-state, not keep_partials): cherries (tip, tip) and cherry + tip nodes are fused
	// into their parent's work and never stored.  Everything else that is internal is a "core" node with an array in HBM.
	// (rescaled evaluations keep the fusion: fringe nodes never reach the rescaling threshold themselves)
	const bool fuse = e->fusion_enabled && !e->generic && !e->keep_partials;
	e->fused = fuse;
	std::vector<int> kind(N, CH_CORE);
	for (int n = 0; n < T; n++) kind[n] = CH_TIP;
	if (fuse) {
		for (int i = N - 1; i >= 0; i--) {  // children before parents
			const int n = order[i];
			if (n < T || n == e->root) continue;
			const int l = e->left[n], r = e->right[n];
			if (l < T && r < T) kind[n] = CH_CHERRY;
			else if ((l < T && kind[r] == CH_CHERRY) || (r < T && kind[l] == CH_CHERRY)) kind[n] = CH_CHERRY_TIP;
			else if (e->deep_enabled && kind[l] != CH_CORE && kind[l] != CH_DEEP && kind[r] != CH_CORE && kind[r] != CH_DEEP)
				kind[n] = CH_DEEP;  // both children are tips or fringe: 4-6 tips below, rebuilt in registers wherever its partial is needed
		}
	}
	e->node_kind = kind;
	e->path_node = -1;
	// stored lower arrays: core nodes in id order
	e->core_index.assign(N, -1);
	e->core_count = 0;
	for (int n = T; n < N; n++)
		if (kind[n] == CH_CORE) e->core_index[n] = e->core_count++;
	auto describe = [&](int ch, int32_t &k, int32_t &core, int32_t &t0, int32_t &t1, int32_t &t2, int32_t &inner) {
		k = kind[ch];
		core = e->core_index[ch];
		t0 = t1 = t2 = inner = -1;
		if (k == CH_CHERRY) {
			t0 = e->left[ch];
			t1 = e->right[ch];
		} else if (k == CH_CHERRY_TIP) {
			const int l = e->left[ch], r = e->right[ch];
			inner = l < T ? r : l;
			t2 = l < T ? l : r;
			t0 = e->left[inner];
			t1 = e->right[inner];
		}
	};
	e->deep_host.assign(N, DeepDesc{});
	e->deep_count = 0;
	for (int n = T; n < N; n++)
		if (kind[n] == CH_DEEP) {
			DeepDesc &d = e->deep_host[n];
			int32_t core_unused;
			d.left = e->left[n];
			d.right = e->right[n];
			describe(d.left, d.kind_left, core_unused, d.lt0, d.lt1, d.lt2, d.linner);
			describe(d.right, d.kind_right, core_unused, d.rt0, d.rt1, d.rt2, d.rinner);
			e->deep_count++;
		}
	auto make_op = [&](int n) {
		NodeOp op{};
		op.parent = n;
		op.left = e->left[n];
		op.right = e->right[n];
		op.upper_slot_parent = op.upper_slot_left = op.upper_slot_right = -1;
		op.core_parent = e->core_index[n];
		describe(op.left, op.kind_left, op.core_left, op.lt0, op.lt1, op.lt2, op.linner);
		describe(op.right, op.kind_right, op.core_right, op.rt0, op.rt1, op.rt2, op.rinner);
		return op;
	};
	// height over the core tree (tips and fused nodes = 0), children before parents: reverse of the pre-order list
	std::vector<int> height(N, 0);
	int H = 0, Dmax = 0;
	for (int i = N - 1; i >= 0; i--) {
		const int n = order[i];
		if (kind[n] == CH_CORE) height[n] = 1 + std::max(height[e->left[n]], height[e->right[n]]);
		H = std::max(H, height[n]);
		Dmax = std::max(Dmax, depth[n]);
	}
	// lower levels: core nodes by height 1..H (the root has the largest height and is alone on its level)
	e->lower_ops.clear();
	e->lower_level_off.assign(1, 0);
	for (int h = 1; h <= H; h++) {
		for (int n = T; n < N; n++)
			if (kind[n] == CH_CORE && height[n] == h) e->lower_ops.push_back(make_op(n));
		e->lower_level_off.push_back((int)e->lower_ops.size());
	}
	// upper levels: core parents by depth 0..Dmax-1.  Upper partials of depth-d nodes are only read while
	// depth d+1 is produced, so slots are recycled two levels later (keep_partials: slot = own index).
	e->upper_ops.clear();
	e->upper_level_off.assign(1, 0);
	e->upper_slot.assign(N, -1);
	std::vector<int> free_slots;
	int next_slot = 0;
	std::vector<std::vector<int>> slots_of_depth(Dmax + 2);
	for (int d = 0; d < Dmax; d++) {
		if (!e->keep_partials && d >= 2) {  // uppers of depth d-1 were consumed while producing depth d
			for (int s : slots_of_depth[d - 1]) free_slots.push_back(s);
			slots_of_depth[d - 1].clear();
		}
		for (int n = T; n < N; n++) {
			if (depth[n] != d || (kind[n] != CH_CORE && kind[n] != CH_DEEP)) continue;
			NodeOp op = make_op(n);
			op.upper_slot_parent = n == e->root ? -1 : e->upper_slot[n];
			for (int side = 0; side < 2; side++) {
				const int ch = side ? e->right[n] : e->left[n];
				if (kind[ch] != CH_CORE && kind[ch] != CH_DEEP && !e->keep_partials) continue;  // uppers of tips and fringe nodes stay in registers
				int s;
				if (e->keep_partials) s = ch;
				else if (!free_slots.empty()) {
					s = free_slots.back();
					free_slots.pop_back();
				} else
					s = next_slot++;
				e->upper_slot[ch] = s;
				slots_of_depth[d + 1].push_back(s);
				(side ? op.upper_slot_right : op.upper_slot_left) = s;
			}
			e->upper_ops.push_back(op);
		}
		e->upper_level_off.push_back((int)e->upper_ops.size());
	}
	e->upper_slots = e->keep_partials ? N : next_slot;

	// Depth-first op orders for the tree-walk kernels (see k_lower4_walk).  csize = core ops in the subtree.
	e->walking = e->walk_enabled && !e->generic && !e->keep_partials;
	e->walk_lower_ops.clear();
	e->walk_upper_ops.clear();
	e->walk_upper_slots = 0;
	if (e->walking) {
		std::vector<int> csize(N, 0), usize(N, 0);  // stored nodes / pre-order ops (stored + DEEP) in the subtree
		for (int i = N - 1; i >= 0; i--) {
			const int n = order[i];
			if (kind[n] == CH_CORE) csize[n] = 1 + csize[e->left[n]] + csize[e->right[n]];
			if (kind[n] == CH_CORE || kind[n] == CH_DEEP) usize[n] = 1 + usize[e->left[n]] + usize[e->right[n]];
		}
		// post-order, larger core subtree first: the op before a node is its second (smaller) core child, or its only one
		struct Frame {
			int node, stage;
		};
		std::vector<Frame> st{{e->root, 0}};
		while (!st.empty()) {
			Frame &f = st.back();
			const int n = f.node, l = e->left[n], r = e->right[n];
			const int first = csize[l] >= csize[r] ? l : r, second = first == l ? r : l;
			if (f.stage == 0) {
				f.stage = 1;
				if (kind[first] == CH_CORE) st.push_back({first, 0});
			} else if (f.stage == 1) {
				f.stage = 2;
				if (kind[second] == CH_CORE) st.push_back({second, 0});
			} else {
				NodeOp op = make_op(n);
				op.carry_in = 0;
				if (!e->walk_lower_ops.empty()) {
					const int prev = e->walk_lower_ops.back().parent;
					if (prev == l) op.carry_in = 1;
					else if (prev == r) op.carry_in = 2;
				}
				e->walk_lower_ops.push_back(op);
				st.pop_back();
			}
		}
		// pre-order, SMALLER core subtree first: the first-visited core child takes its upper in registers (never stored);
		// the other child's upper waits in a slot while the small subtree is walked (nesting depth <= log2 of the core count)
		std::vector<int> free_w;
		int next_w = 0;
		std::vector<int> slot_of(N, -1);
		std::vector<int> stack2{e->root};
		int carried_node = -1;  // node whose upper the previous op carried out
		while (!stack2.empty()) {
			const int n = stack2.back();
			stack2.pop_back();
			NodeOp op = make_op(n);
			op.carry_in = (n != e->root && carried_node == n) ? 1 : 0;
			op.upper_slot_parent = -1;
			if (n != e->root && !op.carry_in) {
				op.upper_slot_parent = slot_of[n];
				free_w.push_back(slot_of[n]);  // read by this op; reusable by ops after it
			}
			const int l = e->left[n], r = e->right[n];
			const bool lc = kind[l] == CH_CORE || kind[l] == CH_DEEP, rc2 = kind[r] == CH_CORE || kind[r] == CH_DEEP;  // children with ops of their own
			int first = -1, second = -1;
			if (lc && rc2) {
				first = usize[l] <= usize[r] ? l : r;
				second = first == l ? r : l;
			} else if (lc || rc2)
				first = lc ? l : r;
			op.carry_out = first < 0 ? 0 : (first == l ? 1 : 2);
			carried_node = first;
			if (second >= 0) {
				int sl;
				// a slot freed by THIS op (its own parent upper) must not be reused for its output: lanes of other waves may
				// still be reading it -- not an issue within a thread, but keep it simple and safe: take another one
				if (free_w.size() > 1 || (free_w.size() == 1 && free_w.back() != op.upper_slot_parent)) {
					size_t pick = free_w.size() - 1;
					if (free_w[pick] == op.upper_slot_parent) pick--;
					sl = free_w[pick];
					free_w.erase(free_w.begin() + pick);
				} else
					sl = next_w++;
				slot_of[second] = sl;
				(second == l ? op.upper_slot_left : op.upper_slot_right) = sl;
				stack2.push_back(second);
			}
			if (first >= 0) stack2.push_back(first);  // visited next
			e->walk_upper_ops.push_back(op);
		}
		e->walk_upper_slots = next_w;
	}
	return PHYAMD_OK;
}

int ensure_lower_storage(phyamd_engine *e) {
	const size_t need = (size_t)std::max(1, e->core_count) * (e->two_slots ? 2 : 1);
	if (e->d_lower && e->lower_alloc_cores >= need) return PHYAMD_OK;
	dev_free(e, &e->d_lower, e->lower_alloc_cores * node_partial_doubles(e));
	dev_free(e, &e->d_lscale, e->lower_alloc_cores * (size_t)e->P);
	e->lower_alloc_cores = 0;
	int rc = dev_alloc(e, &e->d_lower, need * node_partial_doubles(e));
	if (rc) return rc;
	e->lower_alloc_cores = need;
	return PHYAMD_OK;
}

// after slots moved (store / restore): the ops carry slot indices
void refresh_op_cores(phyamd_engine *e) {
	for (std::vector<NodeOp> *ops : {&e->lower_ops, &e->upper_ops, &e->walk_lower_ops, &e->walk_upper_ops})
		for (NodeOp &op : *ops) {
			op.core_parent = e->core_index[op.parent];
			op.core_left = e->core_index[op.left];
			op.core_right = e->core_index[op.right];
		}
}

int upload_schedule(phyamd_engine *e) {
	// op tables are a few KB: allocated once at the maximum size (N - T ops each)
	int rc;
	if (!e->d_lower_ops && (rc = dev_alloc(e, &e->d_lower_ops, (size_t)e->N))) return rc;
	if (!e->d_upper_ops && (rc = dev_alloc(e, &e->d_upper_ops, (size_t)e->N))) return rc;
	HIP_TRY(hipMemcpyAsync(e->d_lower_ops, e->lower_ops.data(), e->lower_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipMemcpyAsync(e->d_upper_ops, e->upper_ops.data(), e->upper_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
	static_assert(sizeof(DeepDesc) == 6 * sizeof(double), "DeepDesc is laid out in the tail of the tip-message table");
	if (!e->generic)
		HIP_TRY(hipMemcpyAsync(e->d_tiptab + (size_t)e->T * e->C * 64, e->deep_host.data(), e->deep_host.size() * sizeof(DeepDesc), hipMemcpyHostToDevice,
		                       e->stream));
	if (e->walking) {
		if (!e->d_walk_lower_ops && (rc = dev_alloc(e, &e->d_walk_lower_ops, (size_t)e->N))) return rc;
		if (!e->d_walk_upper_ops && (rc = dev_alloc(e, &e->d_walk_upper_ops, (size_t)e->N))) return rc;
		HIP_TRY(hipMemcpyAsync(e->d_walk_lower_ops, e->walk_lower_ops.data(), e->walk_lower_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(e->d_walk_upper_ops, e->walk_upper_ops.data(), e->walk_upper_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
	}
	// rows of the gradient slab that are produced by the upper pass (every non-root node)
	std::vector<uint8_t> valid((size_t)e->N * e->C, 1);
	for (int c = 0; c < e->C; c++) valid[(size_t)e->root * e->C + c] = 0;
	HIP_TRY(hipMemcpyAsync(e->d_row_valid, valid.data(), valid.size(), hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return PHYAMD_OK;
}

int ensure_upper_storage(phyamd_engine *e) {
	// the tree-walk schedule parks far fewer uppers than the level schedule keeps; parameter-gradient and inspection
	// calls still run the level kernels, so the larger of the two is held once either has been needed
	const bool level_path = !(e->walking && e->walk_upper_on) || e->level_upper_needed;
	const size_t need = (size_t)std::max(1, level_path ? std::max(e->upper_slots, e->walk_upper_slots) : e->walk_upper_slots);
	if (e->d_upper && e->upper_alloc_slots >= need) return PHYAMD_OK;
	dev_free(e, &e->d_upper, e->upper_alloc_slots * node_partial_doubles(e));
	e->upper_alloc_slots = 0;
	int rc = dev_alloc(e, &e->d_upper, need * node_partial_doubles(e));
	if (rc) return rc;
	e->upper_alloc_slots = need;
	return PHYAMD_OK;
}

int ensure_scaling_storage(phyamd_engine *e) {
	if (e->d_lscale) return PHYAMD_OK;
	return dev_alloc(e, &e->d_lscale, e->lower_alloc_cores * (size_t)e->P);
}

int check_ready(phyamd_engine *e) {
	if (!e->have_topology) return fail(PHYAMD_EINVAL, "phyamd_set_topology has not been called");
	if (!e->have_lengths) return fail(PHYAMD_EINVAL, "phyamd_set_branch_lengths has not been called");
	if (!e->have_freqs) return fail(PHYAMD_EINVAL, "phyamd_set_frequencies has not been called");
	if (!e->have_rates) return fail(PHYAMD_EINVAL, "phyamd_set_category_rates has not been called");
	if (!e->have_weights) return fail(PHYAMD_EINVAL, "phyamd_set_pattern_weights has not been called");
	for (int t = 0; t < e->T; t++)
		if (!e->tip_set[t]) return fail(PHYAMD_EINVAL, "tip %d has no data (phyamd_set_tip_states / phyamd_set_tip_partials)", t);
	if (!e->have_eigen) {
		for (int n = 0; n < e->N; n++)
			if (n != e->root && !e->explicit_host[n]) return fail(PHYAMD_EINVAL, "no eigen system and node %d has no explicit matrices", n);
	}
	return PHYAMD_OK;
}

size_t gen_image_doubles(const phyamd_engine *e) {
	return e->S == 20 ? MatImage<2, 5>::SIZE : e->S == 60 ? MatImage<4, 15>::SIZE : MatImage<4, 16>::SIZE;
}

template <int RT, int KT>
void launch_matrix_images(phyamd_engine *e, int count, const double *src, double *dst) {
	hipLaunchKernelGGL((k_matrix_images<RT, KT>), dim3(count), dim3(256), 0, e->stream, e->S, src, dst);
}

void build_matrix_images(phyamd_engine *e, int count, const double *src, double *dst) {
	if (e->S == 20) launch_matrix_images<2, 5>(e, count, src, dst);
	else if (e->S == 60) launch_matrix_images<4, 15>(e, count, src, dst);
	else launch_matrix_images<4, 16>(e, count, src, dst);
}

int update_matrices(phyamd_engine *e) {
	if (e->generic && e->qimg_dirty && e->have_Q) {  // the rate matrix's image sits behind the per-(node, category) ones
		build_matrix_images(e, 1, e->d_Q, e->d_imgs + (size_t)e->N * e->C * gen_image_doubles(e));
		HIP_TRY(hipGetLastError());
		e->qimg_dirty = false;
	}
	if (!e->matrices_dirty) return PHYAMD_OK;
	if (e->have_eigen) {
		const size_t total = (size_t)e->N * e->C * e->S * e->S;
		const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
		hipLaunchKernelGGL(k_transition_matrices, dim3(blocks), dim3(256), 0, e->stream, e->S, e->C, e->N, e->d_model, e->d_rates, e->d_lengths,
		                   e->d_explicit, e->root, e->d_mats, e->d_dmats);
		HIP_TRY(hipGetLastError());
	}
	if (!e->generic) {  // explicit matrices included: the tables are built from whatever d_mats holds
		const int n = e->T * e->C * 64;
		hipLaunchKernelGGL(k_tip_tables, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, e->C, e->d_mats, e->d_tiptab);
		HIP_TRY(hipGetLastError());
	} else {
		build_matrix_images(e, e->N * e->C, e->d_mats, e->d_imgs);
		HIP_TRY(hipGetLastError());
	}
	e->matrices_dirty = false;
	return PHYAMD_OK;
}

dim3 block_dims(const phyamd_engine *e) { return dim3(WAVE, e->C, e->G); }

template <int WAVES, bool SCALE>
int launch_lower_levels(phyamd_engine *e) {
	const std::vector<int> &level_off = *e->act_level_off;  // all core nodes, or only the dirty ones (incremental update)
	const int levels = (int)level_off.size() - 1;
	int launched = 0;
	const size_t lds = sizeof(double) * ((size_t)4 * e->G * e->C * WAVE + e->G);
	for (int lv = 0; lv < levels; lv++) {
		const int off = level_off[lv], cnt = level_off[lv + 1] - off;
		if (cnt == 0) continue;
		const bool is_root = lv == levels - 1;
		dim3 grid(e->nblk_lower, cnt);
		launched++;
		if (is_root)
			hipLaunchKernelGGL((k_lower4<WAVES, SCALE, true>), grid, block_dims(e), lds, e->stream, e->act_lower_ops + off, e->T, e->P, e->C,
			                   e->d_tipmask, e->d_lower, e->d_mats, e->d_tiptab, e->d_lscale, e->d_freqs, e->d_props, e->d_weights, e->d_plk, e->d_wl,
			                   e->d_lnl_part);
		else
			hipLaunchKernelGGL((k_lower4<WAVES, SCALE, false>), grid, block_dims(e), SCALE ? lds : 0, e->stream, e->act_lower_ops + off, e->T, e->P,
			                   e->C, e->d_tipmask, e->d_lower, e->d_mats, e->d_tiptab, e->d_lscale, e->d_freqs, e->d_props, e->d_weights, e->d_plk, e->d_wl,
			                   e->d_lnl_part);
	}
	HIP_TRY(hipGetLastError());
	e->prof.lower_launches = launched;
	e->lnl_blocks = e->nblk_lower;
	return PHYAMD_OK;
}

template <int WAVES, bool SCALE, int PPT>
void launch_lower_walk_ppt(phyamd_engine *e, size_t lds) {
	hipLaunchKernelGGL((k_lower4_walk<WAVES, PPT, SCALE>), dim3(e->nblk_walk), block_dims(e), lds, e->stream, e->d_walk_lower_ops, (int)e->walk_lower_ops.size(), e->T,
	                   e->P, e->C, e->d_tipmask, e->d_lower, e->d_mats, e->d_tiptab, e->d_lscale, e->d_freqs, e->d_props, e->d_weights, e->d_plk, e->d_wl,
	                   e->d_lnl_part);
}

template <int WAVES, bool SCALE>
int launch_lower_walk(phyamd_engine *e) {
	const size_t lds = sizeof(double) * ((size_t)e->G * e->C * WAVE * (SCALE ? 3 : 1) + e->G);
	if (e->ppt_walk_lower == 1) launch_lower_walk_ppt<WAVES, SCALE, 1>(e, lds);
	else launch_lower_walk_ppt<WAVES, SCALE, 2>(e, lds);
	HIP_TRY(hipGetLastError());
	e->prof.lower_launches = 1;
	e->lnl_blocks = e->nblk_walk;
	return PHYAMD_OK;
}

template <int WAVES>
int launch_lower_w(phyamd_engine *e) {
	if (e->walking && e->walk_lower_on && !e->incremental_pass) return e->scaling_on ? launch_lower_walk<WAVES, true>(e) : launch_lower_walk<WAVES, false>(e);
	return e->scaling_on ? launch_lower_levels<WAVES, true>(e) : launch_lower_levels<WAVES, false>(e);
}

template <typename K>
int allow_big_lds(K kernel, size_t bytes) {
	if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
	return PHYAMD_OK;
}

// one pre-order pass.  PARAMS: also accumulate the substitution-parameter sums of parameters [p0, p0 + pc)
template <int WAVES, bool SCALE, bool FOLD, bool COMPAT, bool PARAMS>
int launch_upper_levels(phyamd_engine *e, int p0 = 0, int pc = 0) {
	const int levels = (int)e->upper_level_off.size() - 1;
	int launched = 0;
	const size_t nw = (size_t)e->G * e->C, nacc = NACC + (PARAMS ? pc : 0);
	const size_t lds = sizeof(double) * ((SCALE ? 6 * nw * WAVE : 0) + nw * nacc * WAVE + nw * nacc);
	int rc;
	if ((rc = allow_big_lds(k_upper4<WAVES, SCALE, FOLD, COMPAT, PARAMS>, lds))) return rc;
	const int op_total = (int)e->upper_ops.size();
	const double *dpm = PARAMS ? e->d_dpm + (size_t)p0 * e->N * e->C * 16 : nullptr;
	const double *dptab = PARAMS ? e->d_dptab + (size_t)p0 * e->T * e->C * 64 : nullptr;
	double *ppart = PARAMS ? e->d_ppart + (size_t)p0 * op_total * e->nblk : nullptr;
	for (int lv = 0; lv < levels; lv++) {
		const int off = e->upper_level_off[lv], cnt = e->upper_level_off[lv + 1] - off;
		if (cnt == 0) continue;
		dim3 grid(e->nblk, cnt);
		launched++;
		hipLaunchKernelGGL((k_upper4<WAVES, SCALE, FOLD, COMPAT, PARAMS>), grid, block_dims(e), lds, e->stream, e->d_upper_ops + off, e->T, e->P, e->C,
		                   e->d_tipmask, e->d_lower, e->d_upper, e->d_mats, e->d_tiptab, e->d_Q, e->d_freqs, e->d_props, e->d_weights, e->d_wl, e->d_gpart,
		                   e->nblk, dpm, dptab, pc, e->N, ppart, off, op_total);
	}
	HIP_TRY(hipGetLastError());
	e->prof.upper_launches = launched;
	return PHYAMD_OK;
}

int upload_qpi(phyamd_engine *e) {
	if (!e->qpi_dirty) return PHYAMD_OK;  // diag(pi) Q, 16 doubles
	int rc;
	if (!e->d_Qpi && (rc = dev_alloc(e, &e->d_Qpi, 16))) return rc;
	double qpi[16];
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++) qpi[i * 4 + j] = e->freqs[i] * e->Q_host[i * 4 + j];
	HIP_TRY(hipMemcpyAsync(e->d_Qpi, qpi, sizeof(qpi), hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->qpi_dirty = false;
	return PHYAMD_OK;
}

template <int WAVES, bool FOLD, bool SCALE, bool COMPAT>
int launch_upper_walk_v(phyamd_engine *e) {
	const int ops = (int)e->walk_upper_ops.size(), nb = e->nblk_walk_upper * e->G;
	const size_t lds = sizeof(double) * ((size_t)e->G * e->C * NACC * WCOL + (SCALE ? (size_t)6 * e->G * e->C * WAVE : 0));
	hipLaunchKernelGGL((k_upper4_walk<WAVES, FOLD, false, SCALE, COMPAT>), dim3(e->nblk_walk_upper), block_dims(e), lds, e->stream, e->d_walk_upper_ops, ops, e->T,
	                   e->P, e->C, e->d_tipmask, e->d_lower, e->d_upper, e->d_mats, e->d_tiptab, FOLD ? e->d_Q : e->d_Qpi, e->d_freqs, e->d_wl, e->d_gpart, nb,
	                   (const double *)nullptr, (const double *)nullptr, (double *)nullptr, e->d_props, e->d_weights);
	HIP_TRY(hipGetLastError());
	e->prof.upper_launches = 1;
	e->grad_blocks = nb;
	return PHYAMD_OK;
}

template <int WAVES>
int launch_upper_walk(phyamd_engine *e, bool fold, bool compat) {
	int rc;
	if ((rc = upload_qpi(e))) return rc;
	if (e->scaling_on) {
		if (fold) return compat ? launch_upper_walk_v<WAVES, true, true, true>(e) : launch_upper_walk_v<WAVES, true, true, false>(e);
		return compat ? launch_upper_walk_v<WAVES, false, true, true>(e) : launch_upper_walk_v<WAVES, false, true, false>(e);
	}
	return fold ? launch_upper_walk_v<WAVES, true, false, false>(e) : launch_upper_walk_v<WAVES, false, false, false>(e);
}

// G2 through the tree walk: B = U^-1 dQ U per parameter, the eigen-basis tables, one walk, then 16 sums and a contraction
template <int WAVES, bool SCALE>
int launch_upper_walk_params(phyamd_engine *e) {
	const int ops = (int)e->walk_upper_ops.size(), nb = e->nblk_walk_upper * e->G, S = 4, np = e->np;
	const size_t lds = sizeof(double) * ((size_t)e->G * e->C * 16 * WCOL + (SCALE ? (size_t)6 * e->G * e->C * WAVE : 0));
	int rc;
	if ((rc = upload_qpi(e))) return rc;
	if ((size_t)np > e->np_alloc_B) {
		dev_free(e, &e->d_Bw, e->np_alloc_B * 16);
		e->np_alloc_B = 0;
		if ((rc = dev_alloc(e, &e->d_Bw, (size_t)np * 16))) return rc;
		e->np_alloc_B = np;
	}
	if (!e->d_pbuf && (rc = dev_alloc(e, &e->d_pbuf, 96))) return rc;
	if (!e->d_Fw && (rc = dev_alloc(e, &e->d_Fw, (size_t)e->N * e->C * 20))) return rc;
	if (!e->d_gacc && (rc = dev_alloc(e, &e->d_gacc, (size_t)16 * nb * e->C + 16))) return rc;
	{
		const double *evec = e->model.data() + S, *ivec = e->model.data() + S + S * S;
		std::vector<double> B((size_t)np * 16), tmp(16), pb(96);
		for (int th = 0; th < np; th++) {
			const double *dQ = e->dQ_host.data() + (size_t)th * 16;
			for (int a = 0; a < 4; a++)
				for (int j = 0; j < 4; j++) {
					double v = 0.0;
					for (int i = 0; i < 4; i++) v += ivec[a * 4 + i] * dQ[i * 4 + j];
					tmp[a * 4 + j] = v;
				}
			for (int a = 0; a < 4; a++)
				for (int b = 0; b < 4; b++) {
					double v = 0.0;
					for (int j = 0; j < 4; j++) v += tmp[a * 4 + j] * evec[j * 4 + b];
					B[(size_t)th * 16 + a * 4 + b] = v;
				}
		}
		for (int a = 0; a < 4; a++)
			for (int i = 0; i < 4; i++) {
				pb[a * 4 + i] = evec[i * 4 + a] * e->freqs[i];  // (diag(pi) U)^T
				pb[16 + a * 4 + i] = ivec[a * 4 + i];
			}
		for (int m = 0; m < 16; m++)
			for (int b = 0; b < 4; b++) {
				double v = 0.0;
				for (int j = 0; j < 4; j++)
					if (m >> j & 1) v += ivec[b * 4 + j];
				pb[32 + m * 4 + b] = v;  // U^-1 . mask
			}
		HIP_TRY(hipMemcpyAsync(e->d_Bw, B.data(), sizeof(double) * B.size(), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(e->d_pbuf, pb.data(), sizeof(double) * pb.size(), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
	}
	const int nf = e->N * e->C * 20;
	hipLaunchKernelGGL(k_eigen_weights, dim3((nf + 255) / 256), dim3(256), 0, e->stream, e->C, e->N, e->d_model, e->d_rates, e->d_props, e->d_lengths, e->d_explicit,
	                   e->root, e->d_Fw);
	hipLaunchKernelGGL((k_upper4_walk<WAVES, false, true, SCALE, false>), dim3(e->nblk_walk_upper), block_dims(e), lds, e->stream, e->d_walk_upper_ops, ops, e->T,
	                   e->P, e->C, e->d_tipmask, e->d_lower, e->d_upper, e->d_mats, e->d_tiptab, e->d_Qpi, e->d_freqs, e->d_wl, e->d_gpart, nb, e->d_pbuf, e->d_Fw,
	                   e->d_gacc, e->d_props, e->d_weights);
	double *gsum = e->d_gacc + (size_t)16 * nb * e->C;
	hipLaunchKernelGGL(k_reduce_rows, dim3(16), dim3(64), 0, e->stream, e->d_gacc, nb * e->C, (const uint8_t *)nullptr, gsum);
	hipLaunchKernelGGL(k_contract_parameters, dim3((np + 63) / 64), dim3(64), 0, e->stream, np, e->d_Bw, gsum, e->d_result + 1 + (size_t)e->N * e->C);
	HIP_TRY(hipGetLastError());
	e->prof.upper_launches = 1;
	e->grad_blocks = nb;
	return PHYAMD_OK;
}

template <int WAVES>
int launch_upper_w(phyamd_engine *e, int flags) {
	const bool fold = flags & PHYAMD_GRAD_FOLD_ROOT_FREQS, compat = (flags & PHYAMD_GRAD_COMPAT_SCALED) && e->scaling_on;
	e->grad_blocks = e->nblk;
	if (e->walking && e->walk_upper_on) return launch_upper_walk<WAVES>(e, fold, compat);
	if (e->scaling_on) {
		if (fold) return compat ? launch_upper_levels<WAVES, true, true, true, false>(e) : launch_upper_levels<WAVES, true, true, false, false>(e);
		return compat ? launch_upper_levels<WAVES, true, false, true, false>(e) : launch_upper_levels<WAVES, true, false, false, false>(e);
	}
	return fold ? launch_upper_levels<WAVES, false, true, false, false>(e) : launch_upper_levels<WAVES, false, false, false, false>(e);
}

// largest number of parameter accumulators one workgroup's LDS holds next to the NACC branch accumulators
int parameter_chunk(const phyamd_engine *e) {
	const size_t nw = (size_t)e->G * e->C, budget = 160 * 1024 / sizeof(double) - (e->scaling_on ? 6 * nw * WAVE : 0);
	const long cols = (long)(budget / (nw * (WAVE + 1))) - NACC;
	return (int)std::max(0L, std::min(32L, cols));
}

template <int WAVES>
int launch_upper_params_w(phyamd_engine *e, int flags) {
	const bool compat = (flags & PHYAMD_GRAD_COMPAT_SCALED) && e->scaling_on;
	const int chunk = parameter_chunk(e);
	if (chunk < 1) return fail(PHYAMD_EUNSUPPORTED, "%d categories leave no LDS for parameter accumulators", e->C);
	int rc = PHYAMD_OK;
	// more parameters than one workgroup can accumulate: repeat the pass (uppers and branch sums are rewritten with identical values)
	for (int p0 = 0; p0 < e->np && !rc; p0 += chunk) {
		const int pc = std::min(chunk, e->np - p0);
		if (e->scaling_on)
			rc = compat ? launch_upper_levels<WAVES, true, false, true, true>(e, p0, pc) : launch_upper_levels<WAVES, true, false, false, true>(e, p0, pc);
		else
			rc = launch_upper_levels<WAVES, false, false, false, true>(e, p0, pc);
	}
	return rc;
}

// ---- S != 4: MFMA kernels -----------------------------------------------------------------------------------
// scratch of the rescaled S != 4 path: per op of a level, 5 rows [C][P] (lower: 1 row of maxima; upper: 3 rows num_l,
// num_r, den + 2 rows of maxima)
int ensure_gen_scale_storage(phyamd_engine *e) {
	int widest = 1;
	for (size_t i = 0; i + 1 < e->lower_level_off.size(); i++) widest = std::max(widest, e->lower_level_off[i + 1] - e->lower_level_off[i]);
	for (size_t i = 0; i + 1 < e->upper_level_off.size(); i++) widest = std::max(widest, e->upper_level_off[i + 1] - e->upper_level_off[i]);
	const size_t need = (size_t)widest * 5 * e->C * e->P;
	if (e->d_gen_scratch && e->gen_scratch_alloc >= need) return PHYAMD_OK;
	dev_free(e, &e->d_gen_scratch, e->gen_scratch_alloc);
	e->gen_scratch_alloc = 0;
	int rc = dev_alloc(e, &e->d_gen_scratch, need);
	if (rc) return rc;
	e->gen_scratch_alloc = need;
	return PHYAMD_OK;
}

template <int RT, int KT, bool SCALE>
int launch_lower_gen(phyamd_engine *e) {
	const std::vector<int> &level_off = *e->act_level_off;
	const int levels = (int)level_off.size() - 1;
	const size_t lds = sizeof(double) * 2 * MatImage<RT, KT>::SIZE;
	int rc;
	if ((rc = allow_big_lds(k_lower_gen<RT, KT, true, SCALE>, lds)) || (rc = allow_big_lds(k_lower_gen<RT, KT, false, SCALE>, lds))) return rc;
	if (SCALE && (rc = ensure_gen_scale_storage(e))) return rc;
	const int pblocks = (e->P + 255) / 256;
	int launched = 0;
	for (int lv = 0; lv < levels; lv++) {
		const int off = level_off[lv], cnt = level_off[lv + 1] - off;
		if (cnt == 0) continue;
		launched++;
		dim3 grid(e->nblk, cnt, e->C);
		const bool is_root = lv == levels - 1;
		if (is_root)
			hipLaunchKernelGGL((k_lower_gen<RT, KT, true, SCALE>), grid, dim3(GenGeo<RT>::WAVES * 64), lds, e->stream, e->act_lower_ops + off, e->T, e->P, e->Pp, e->S, e->C,
			                   e->d_tipmask, e->d_tipsets, e->d_lower, e->d_imgs, e->d_freqs, e->d_props, e->d_Lc, e->d_gen_scratch);
		else
			hipLaunchKernelGGL((k_lower_gen<RT, KT, false, SCALE>), grid, dim3(GenGeo<RT>::WAVES * 64), lds, e->stream, e->act_lower_ops + off, e->T, e->P, e->Pp, e->S, e->C,
			                   e->d_tipmask, e->d_tipsets, e->d_lower, e->d_imgs, e->d_freqs, e->d_props, e->d_Lc, e->d_gen_scratch);
		if (SCALE)
			hipLaunchKernelGGL(k_scale_gen, dim3(pblocks, cnt), dim3(256), 0, e->stream, e->act_lower_ops + off, e->P, e->Pp, e->S, e->C, e->d_lower, e->d_gen_scratch,
			                   e->d_lscale, is_root ? e->d_Lc : (double *)nullptr);
	}
	const double *lscale_root = SCALE ? e->d_lscale + (size_t)e->core_index[e->root] * e->P : nullptr;
	hipLaunchKernelGGL(k_root_finish, dim3(e->nblk_root), dim3(256), 0, e->stream, e->P, e->C, e->d_Lc, e->d_weights, lscale_root, e->d_plk, e->d_wl, e->d_lnl_part);
	HIP_TRY(hipGetLastError());
	e->prof.lower_launches = launched;
	return PHYAMD_OK;
}

template <int RT, int KT, bool FOLD, bool SCALE>
int launch_upper_gen_v(phyamd_engine *e, bool compat) {
	const int levels = (int)e->upper_level_off.size() - 1;
	const size_t lds = sizeof(double) * (4 * MatImage<RT, KT>::SIZE + 2 * GenGeo<RT>::WAVES);
	int rc;
	if ((rc = allow_big_lds(k_upper_gen<RT, KT, FOLD, SCALE>, lds))) return rc;
	if (SCALE && (rc = ensure_gen_scale_storage(e))) return rc;
	const int pblocks = (e->P + 255) / 256;
	double *nd = e->d_gen_scratch, *mxu = SCALE ? e->d_gen_scratch : nullptr;
	for (int lv = 0; lv < levels; lv++) {
		const int off = e->upper_level_off[lv], cnt = e->upper_level_off[lv + 1] - off;
		if (cnt == 0) continue;
		dim3 grid(e->nblk, cnt, e->C);
		if (SCALE) mxu = e->d_gen_scratch + (size_t)cnt * 3 * e->C * e->P;
		hipLaunchKernelGGL((k_upper_gen<RT, KT, FOLD, SCALE>), grid, dim3(GenGeo<RT>::WAVES * 64), lds, e->stream, e->d_upper_ops + off, e->T, e->P, e->Pp, e->S, e->C,
		                   e->d_tipmask, e->d_tipsets, e->d_lower, e->d_upper, e->d_imgs, e->d_imgs + (size_t)e->N * e->C * MatImage<RT, KT>::SIZE, e->d_freqs, e->d_wl,
		                   e->d_gpart, e->nblk, nd, mxu);
		if (SCALE) {
			hipLaunchKernelGGL(k_scale_upper_gen, dim3(pblocks, cnt), dim3(256), 0, e->stream, e->d_upper_ops + off, e->P, e->Pp, e->S, e->C, e->d_upper, mxu);
			const int ppb = GenGeo<RT>::PATTERNS_PER_BLOCK;
			if (compat)
				hipLaunchKernelGGL(k_scaled_gradient_gen<true>, dim3(e->nblk, cnt), dim3(256), 0, e->stream, e->d_upper_ops + off, e->P, e->C, ppb, nd, e->d_weights,
				                   e->d_props, e->d_gpart, e->nblk);
			else
				hipLaunchKernelGGL(k_scaled_gradient_gen<false>, dim3(e->nblk, cnt), dim3(256), 0, e->stream, e->d_upper_ops + off, e->P, e->C, ppb, nd, e->d_weights,
				                   e->d_props, e->d_gpart, e->nblk);
		}
	}
	HIP_TRY(hipGetLastError());
	e->prof.upper_launches = levels;
	return PHYAMD_OK;
}

template <int RT, int KT>
int launch_upper_gen(phyamd_engine *e, int flags) {
	const bool fold = flags & PHYAMD_GRAD_FOLD_ROOT_FREQS, compat = flags & PHYAMD_GRAD_COMPAT_SCALED;
	if (e->scaling_on) return fold ? launch_upper_gen_v<RT, KT, true, true>(e, compat) : launch_upper_gen_v<RT, KT, false, true>(e, compat);
	return fold ? launch_upper_gen_v<RT, KT, true, false>(e, false) : launch_upper_gen_v<RT, KT, false, false>(e, false);
}

template <int RT, int KT>
int launch_lower_gen_s(phyamd_engine *e) {
	return e->scaling_on ? launch_lower_gen<RT, KT, true>(e) : launch_lower_gen<RT, KT, false>(e);
}

// workgroups hold C*G waves; the bound is a template parameter so small groups are not register-capped for 1024 threads
int launch_lower(phyamd_engine *e) {
	if (e->generic) return e->S == 20 ? launch_lower_gen_s<2, 5>(e) : e->S == 60 ? launch_lower_gen_s<4, 15>(e) : launch_lower_gen_s<4, 16>(e);
	const int waves = e->C * e->G;
	return waves <= 4 ? launch_lower_w<4>(e) : waves <= 8 ? launch_lower_w<8>(e) : launch_lower_w<16>(e);
}
int launch_upper(phyamd_engine *e, int flags) {
	if (e->generic) return e->S == 20 ? launch_upper_gen<2, 5>(e, flags) : e->S == 60 ? launch_upper_gen<4, 15>(e, flags) : launch_upper_gen<4, 16>(e, flags);
	const int waves = e->C * e->G;
	return waves <= 4 ? launch_upper_w<4>(e, flags) : waves <= 8 ? launch_upper_w<8>(e, flags) : launch_upper_w<16>(e, flags);
}

int launch_upper_params(phyamd_engine *e, int flags) {
	const int waves = e->C * e->G;
	return waves <= 4 ? launch_upper_params_w<4>(e, flags) : waves <= 8 ? launch_upper_params_w<8>(e, flags) : launch_upper_params_w<16>(e, flags);
}

int rebuild_schedule(phyamd_engine *e) {
	int rc;
	if ((rc = build_schedule(e))) return rc;
	if ((rc = upload_schedule(e))) return rc;
	if ((rc = ensure_lower_storage(e))) return rc;
	e->schedule_epoch++;  // slots start over: a stored state no longer maps onto them
	e->upper_valid = false;
	e->all_dirty = true;
	e->lower_valid = false;
	return PHYAMD_OK;
}

void record(phyamd_engine *e, int i) {
	if (e->profiling) (void)hipEventRecord(e->ev[i], e->stream);
}

// lower pass (+ lazy rescaling).  On return d_result[0] holds lnL on the device.
int run_lower(phyamd_engine *e, bool need_host_check) {
	int rc;
	if ((rc = bind_device(e))) return rc;
	if ((rc = check_ready(e))) return rc;
	record(e, 0);
	if ((rc = update_matrices(e))) return rc;
	record(e, 1);
	if (e->scaling_on && (rc = ensure_scaling_storage(e))) return rc;
	e->act_level_off = &e->lower_level_off;
	e->act_lower_ops = e->d_lower_ops;
	e->incremental_pass = false;
	const bool incremental = e->lower_valid && !e->all_dirty;
	if (incremental && e->changed.empty() && !e->force_root) {  // nothing changed: d_result[0] still holds lnL
		record(e, 2);
		e->prof.lower_launches = 0;
		e->prof_pending = e->profiling;
		e->prof_with_upper = false;
		return PHYAMD_OK;
	}
	e->path_node = -1;  // partials are about to change
	std::vector<uint8_t> dirty;
	if (incremental) {
		// only single branch lengths changed since the stored partials were computed: recompute the core nodes on the paths
		// from those branches to the root, in level order (update_nodes[] semantics, treelikelihood.c:73-114, 1645-1734)
		dirty.assign(e->N, 0);
		for (int n : e->changed)
			for (int a = e->parent[n]; a >= 0 && !dirty[a]; a = e->parent[a])
				if (e->core_index[a] >= 0) dirty[a] = 1;  // fused fringe nodes are recomputed inside their first stored ancestor
		if (e->force_root) dirty[e->root] = 1;
	}
	if (e->stored.valid && e->stored.epoch == e->schedule_epoch) {
		// nodes about to be written leave the slot the stored state lives in (current_partials_indexes flip, treelikelihood.c:1693-1700)
		bool moved = false;
		for (int n = e->T; n < e->N; n++)
			if (e->core_index[n] >= 0 && (!incremental || dirty[n]) && e->core_index[n] == e->stored.core_index[n]) {
				e->core_index[n] += e->core_index[n] < e->core_count ? e->core_count : -e->core_count;
				moved = true;
			}
		if (moved) {
			refresh_op_cores(e);
			if ((rc = upload_schedule(e))) return rc;
		}
	}
	if (incremental) {
		e->inc_ops.clear();
		e->inc_level_off.assign(1, 0);
		for (size_t lv = 0; lv + 1 < e->lower_level_off.size(); lv++) {
			for (int i = e->lower_level_off[lv]; i < e->lower_level_off[lv + 1]; i++)
				if (dirty[e->lower_ops[i].parent]) e->inc_ops.push_back(e->lower_ops[i]);
			e->inc_level_off.push_back((int)e->inc_ops.size());
		}
		if (!e->d_inc_ops && (rc = dev_alloc(e, &e->d_inc_ops, (size_t)e->N))) return rc;
		HIP_TRY(hipMemcpyAsync(e->d_inc_ops, e->inc_ops.data(), e->inc_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));  // inc_ops is reused by the next call
		e->act_level_off = &e->inc_level_off;
		e->act_lower_ops = e->d_inc_ops;
		e->incremental_pass = true;
	}
	for (int attempt = 0; attempt < 2; attempt++) {
		if ((rc = launch_lower(e))) return rc;
		hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(64), 0, e->stream, e->d_lnl_part, e->generic ? e->nblk_root : e->lnl_blocks, (const uint8_t *)nullptr,
		                   e->d_result);
		HIP_TRY(hipGetLastError());
		if (e->cfg.rescale != PHYAMD_RESCALE_AUTO || e->scaling_on || !need_host_check) break;
		// lazy switch (treelikelihood.c:1496-1519): +-inf lnL turns rescaling on for good and recomputes
		HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double), hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		if (!std::isinf(e->h_result[0])) break;
		e->scaling_on = true;
		if ((rc = rebuild_schedule(e))) return rc;  // rescaling runs the level kernels (no tree walk), fringe fusion stays
		if ((rc = ensure_scaling_storage(e))) return rc;
		e->act_level_off = &e->lower_level_off;  // and recomputes every node
		e->act_lower_ops = e->d_lower_ops;
		e->incremental_pass = false;
	}
	e->incremental_pass = false;
	e->lower_valid = true;
	e->all_dirty = false;
	e->force_root = false;
	e->changed.clear();
	record(e, 2);
	e->prof_pending = e->profiling;
	e->prof_with_upper = false;
	e->upper_valid = false;
	return PHYAMD_OK;
}

// B_theta = U^-1 dQ_theta U, dP/dtheta matrices and their tip tables (recomputed per call: O(np N C) work)
int update_parameter_matrices(phyamd_engine *e) {
	const int S = e->S, np = e->np;
	int rc;
	if ((size_t)np > e->np_alloc) {
		dev_free(e, &e->d_B, e->np_alloc * S * S);
		dev_free(e, &e->d_dpm, e->np_alloc * e->N * e->C * S * S);
		dev_free(e, &e->d_dptab, e->np_alloc * e->T * e->C * 64);
		e->np_alloc = 0;
		if ((rc = dev_alloc(e, &e->d_B, (size_t)np * S * S)) || (rc = dev_alloc(e, &e->d_dpm, (size_t)np * e->N * e->C * S * S)) ||
		    (rc = dev_alloc(e, &e->d_dptab, (size_t)np * e->T * e->C * 64)))
			return rc;
		e->np_alloc = np;
	}
	const size_t need = (size_t)np * e->upper_ops.size() * ((size_t)e->nblk + 1);
	if (need > e->ppart_alloc) {
		dev_free(e, &e->d_ppart, e->ppart_alloc);
		e->ppart_alloc = 0;
		if ((rc = dev_alloc(e, &e->d_ppart, need))) return rc;
		e->ppart_alloc = need;
	}
	if (e->params_dirty) {
		const double *evec = e->model.data() + S, *ivec = e->model.data() + S + S * S;
		std::vector<double> B((size_t)np * S * S), tmp((size_t)S * S);
		for (int th = 0; th < np; th++) {
			const double *dQ = e->dQ_host.data() + (size_t)th * S * S;
			for (int a = 0; a < S; a++)
				for (int j = 0; j < S; j++) {
					double v = 0.0;
					for (int i = 0; i < S; i++) v += ivec[a * S + i] * dQ[i * S + j];
					tmp[a * S + j] = v;
				}
			for (int a = 0; a < S; a++)
				for (int b = 0; b < S; b++) {
					double v = 0.0;
					for (int j = 0; j < S; j++) v += tmp[a * S + j] * evec[j * S + b];
					B[((size_t)th * S + a) * S + b] = v;
				}
		}
		HIP_TRY(hipMemcpyAsync(e->d_B, B.data(), sizeof(double) * B.size(), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));  // B is a stack-lifetime buffer
		e->params_dirty = false;
	}
	const size_t total = (size_t)np * e->N * e->C * S * S;
	hipLaunchKernelGGL(k_parameter_matrices, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, e->stream, S, e->C, e->N, np, e->d_model,
	                   e->d_B, e->d_rates, e->d_lengths, e->d_explicit, e->root, e->d_dpm);
	const int n = np * e->T * e->C * 64;
	hipLaunchKernelGGL(k_parameter_tip_tables, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->T, e->N, e->C, np, e->d_dpm, e->d_dptab);
	HIP_TRY(hipGetLastError());
	return PHYAMD_OK;
}

// substitution-parameter sums of the 20 / 60 / 61-state engines on the stored partials of a keep-partials gradient
// (k_param_den_gen ... k_param_contract_gen); dst: [np] on the device
int launch_parameters_gen(phyamd_engine *e, double *dst) {
	const int S = e->S, S2 = S * S, np = e->np, B = e->N - 1;
	int rc;
	if (!e->d_pg_nodes) {
		if ((rc = dev_alloc(e, &e->d_pg_nodes, (size_t)B)) || (rc = dev_alloc(e, &e->d_pg_core, (size_t)e->N)) ||
		    (rc = dev_alloc(e, &e->d_pg_den, (size_t)B * e->P)) || (rc = dev_alloc(e, &e->d_pg_Gw, ((size_t)B * e->C + 1) * S2)))
			return rc;
	}
	if ((size_t)np > e->pg_np_alloc) {
		dev_free(e, &e->d_pg_B, e->pg_np_alloc * S2);
		e->pg_np_alloc = 0;
		if ((rc = dev_alloc(e, &e->d_pg_B, (size_t)np * S2))) return rc;
		e->pg_np_alloc = np;
		e->params_dirty = true;
	}
	{  // the schedule may have been rebuilt since the last call: the two index tables are N ints
		std::vector<int> nodes;
		for (int n = 0; n < e->N; n++)
			if (n != e->root) nodes.push_back(n);
		HIP_TRY(hipMemcpyAsync(e->d_pg_nodes, nodes.data(), sizeof(int) * nodes.size(), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(e->d_pg_core, e->core_index.data(), sizeof(int) * e->N, hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
	}
	if (e->params_dirty) {  // B_theta = U^-1 dQ_theta U
		const double *evec = e->model.data() + S, *ivec = e->model.data() + S + S2;
		std::vector<double> Bm((size_t)np * S2), tmp((size_t)S2);
		for (int th = 0; th < np; th++) {
			const double *dQ = e->dQ_host.data() + (size_t)th * S2;
			for (int a = 0; a < S; a++)
				for (int j = 0; j < S; j++) {
					double v = 0.0;
					for (int i = 0; i < S; i++) v += ivec[a * S + i] * dQ[i * S + j];
					tmp[a * S + j] = v;
				}
			for (int a = 0; a < S; a++)
				for (int b = 0; b < S; b++) {
					double v = 0.0;
					for (int j = 0; j < S; j++) v += tmp[a * S + j] * evec[j * S + b];
					Bm[((size_t)th * S + a) * S + b] = v;
				}
		}
		HIP_TRY(hipMemcpyAsync(e->d_pg_B, Bm.data(), sizeof(double) * Bm.size(), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		e->params_dirty = false;
	}
	hipLaunchKernelGGL(k_param_den_gen, dim3((e->P + 255) / 256, B), dim3(256), 0, e->stream, e->d_pg_nodes, e->T, e->P, e->Pp, S, e->C, e->d_pg_core, e->d_tipmask,
	                   e->d_tipsets, e->d_lower, e->d_upper, e->d_mats, e->d_freqs, e->d_props, e->d_pg_den);
	const size_t lds = sizeof(double) * 2 * (size_t)std::max(S2, S * (PARAM_CHUNK + 1));
	hipLaunchKernelGGL(k_param_outer_gen, dim3(B, e->C), dim3(256), lds, e->stream, e->d_pg_nodes, e->T, e->P, e->Pp, S, e->C, e->d_pg_core, e->d_tipmask,
	                   e->d_tipsets, e->d_lower, e->d_upper, e->d_model, e->d_freqs, e->d_props, e->d_rates, e->d_lengths, e->d_explicit, e->d_weights,
	                   e->d_pg_den, e->d_pg_Gw);
	double *Gsum = e->d_pg_Gw + (size_t)B * e->C * S2;
	hipLaunchKernelGGL(k_param_sum_gen, dim3((S2 + 255) / 256), dim3(256), 0, e->stream, B * e->C, S2, e->d_pg_Gw, Gsum);
	hipLaunchKernelGGL(k_param_contract_gen, dim3(np), dim3(64), 0, e->stream, S2, e->d_pg_B, Gsum, dst);
	HIP_TRY(hipGetLastError());
	return PHYAMD_OK;
}

// d lnL / d pi_f through the root frequencies, f < S, written to dst (device) on the engine's stream
int launch_root_frequency_term(phyamd_engine *e, double *dst) {
	int rc;
	const int nb = (e->P + 255) / 256;
	if (!e->d_rf_part && (rc = dev_alloc(e, &e->d_rf_part, ((size_t)nb + 1) * e->S))) return rc;
	const double *root = e->d_lower + (size_t)e->core_index[e->root] * node_partial_doubles(e);
	const size_t cat_stride = e->generic ? (size_t)e->S * e->Pp : (size_t)e->P * e->S;
	const size_t pat_stride = e->generic ? 1 : (size_t)e->S, state_stride = e->generic ? (size_t)e->Pp : 1;
	hipLaunchKernelGGL(k_root_frequency_term, dim3(nb), dim3(256), 0, e->stream, e->P, e->S, e->C, root, cat_stride, pat_stride, state_stride, e->d_freqs,
	                   e->d_props, e->d_weights, e->d_rf_part);
	hipLaunchKernelGGL(k_reduce_rows, dim3(e->S), dim3(64), 0, e->stream, e->d_rf_part, nb, (const uint8_t *)nullptr, dst ? dst : e->d_rf_part + (size_t)nb * e->S);
	HIP_TRY(hipGetLastError());
	return PHYAMD_OK;
}

// sum_k (w_k / L_k) sum_i pi_i (p_root[cat 0] - mean of p_root[cat >= 1]) of the resident root partial -> dst (device)
int launch_root_invariant_term(phyamd_engine *e, double *dst) {
	int rc;
	const int nb = (e->P + 255) / 256;
	if (!e->d_inv_part && (rc = dev_alloc(e, &e->d_inv_part, (size_t)nb + 1))) return rc;
	const double *root = e->d_lower + (size_t)e->core_index[e->root] * node_partial_doubles(e);
	const size_t cat_stride = e->generic ? (size_t)e->S * e->Pp : (size_t)e->P * e->S;
	const size_t pat_stride = e->generic ? 1 : (size_t)e->S, state_stride = e->generic ? (size_t)e->Pp : 1;
	hipLaunchKernelGGL(k_root_invariant_term, dim3(nb), dim3(256), 0, e->stream, e->P, e->S, e->C, root, cat_stride, pat_stride, state_stride, e->d_freqs,
	                   e->d_props, e->d_weights, e->d_inv_part);
	hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(64), 0, e->stream, e->d_inv_part, nb, (const uint8_t *)nullptr, dst ? dst : e->d_inv_part + nb);
	HIP_TRY(hipGetLastError());
	return PHYAMD_OK;
}

int run_gradient(phyamd_engine *e, int flags, bool with_params = false) {
	int rc;
	e->upper_fold = (flags & PHYAMD_GRAD_FOLD_ROOT_FREQS) != 0;
	if (with_params) {
		if (e->np < 1) return fail(PHYAMD_EINVAL, "phyamd_set_rate_matrix_derivatives has not been called");
		if (!e->have_eigen) return fail(PHYAMD_EINVAL, "substitution-parameter gradients need the eigen system (phyamd_set_eigen)");
		if (flags & PHYAMD_GRAD_FOLD_ROOT_FREQS)
			return fail(PHYAMD_EINVAL, "PHYAMD_GRAD_FOLD_ROOT_FREQS cannot be combined with parameter gradients (the reference clears include_root_freqs, treelikelihood.c:291-305)");
	}
	if (with_params && e->generic && !e->keep_partials) {
		// the 20 / 60 / 61-state parameter kernels read every node's lower and upper partial: keep them from here on
		if ((rc = bind_device(e))) return rc;
		e->keep_partials = true;
		if ((rc = rebuild_schedule(e))) return rc;
	}
	if ((rc = run_lower(e, true))) return rc;
	if ((flags & PHYAMD_GRAD_FOLD_ROOT_FREQS) && e->scaling_on && e->fused) {
		// The reference's folded-frequency arithmetic is inexact for non-uniform pi: under rescaling every
		// branch then has its own "site likelihood" as denominator, which the fused fringe does not form.  Reproducing it takes
		// the unfused schedule (every internal node stored) from here on.
		e->fusion_enabled = false;
		if ((rc = rebuild_schedule(e))) return rc;
		if ((rc = run_lower(e, true))) return rc;
	}
	bool any_explicit = false;  // explicit matrices have no eigen system: the tree-walk's eigen-basis branch term does not cover them
	for (uint8_t x : e->explicit_host) any_explicit |= x != 0;
	const bool walk_params = with_params && !any_explicit && !e->generic && e->walking && e->walk_upper_on && e->walk_params_on &&
	                         !((flags & PHYAMD_GRAD_COMPAT_SCALED) && e->scaling_on);
	if (with_params && !e->generic && !walk_params) e->level_upper_needed = true;
	if ((rc = ensure_upper_storage(e))) return rc;
	if (!e->have_Q) return fail(PHYAMD_EINVAL, "the gradient needs the rate matrix: phyamd_set_eigen or phyamd_set_rate_matrix");
	e->grad_blocks = e->nblk;
	// (the compat flag changes only the branch terms; parameter sums always use the mixture denominator: level kernels then)
	if (walk_params) {
		const int waves = e->C * e->G;
		if (e->scaling_on)
			rc = waves <= 4 ? launch_upper_walk_params<4, true>(e) : waves <= 8 ? launch_upper_walk_params<8, true>(e) : launch_upper_walk_params<16, true>(e);
		else
			rc = waves <= 4 ? launch_upper_walk_params<4, false>(e) : waves <= 8 ? launch_upper_walk_params<8, false>(e) : launch_upper_walk_params<16, false>(e);
		if (rc) return rc;
	} else if (with_params && !e->generic) {
		if ((rc = update_parameter_matrices(e))) return rc;
		if ((rc = launch_upper_params(e, flags))) return rc;
	} else if ((rc = launch_upper(e, flags)))
		return rc;
	record(e, 3);
	hipLaunchKernelGGL(k_reduce_rows, dim3(e->N * e->C), dim3(64), 0, e->stream, e->d_gpart, e->grad_blocks, e->d_row_valid, e->d_result + 1);
	if (walk_params) {
		if ((rc = launch_root_frequency_term(e, e->d_result + 1 + (size_t)e->N * e->C + e->np))) return rc;
	} else if (with_params && e->generic) {
		if ((rc = launch_parameters_gen(e, e->d_result + 1 + (size_t)e->N * e->C))) return rc;
		if ((rc = launch_root_frequency_term(e, e->d_result + 1 + (size_t)e->N * e->C + e->np))) return rc;
	} else if (with_params) {  // [np][ops][nblk] -> [np][ops] -> [np], fixed order
		const int ops = (int)e->upper_ops.size();
		double *stage = e->d_ppart + (size_t)e->np * ops * e->nblk;
		hipLaunchKernelGGL(k_reduce_rows, dim3(e->np * ops), dim3(64), 0, e->stream, e->d_ppart, e->nblk, (const uint8_t *)nullptr, stage);
		hipLaunchKernelGGL(k_reduce_rows, dim3(e->np), dim3(64), 0, e->stream, stage, ops, (const uint8_t *)nullptr, e->d_result + 1 + (size_t)e->N * e->C);
		if ((rc = launch_root_frequency_term(e, e->d_result + 1 + (size_t)e->N * e->C + e->np))) return rc;
	}
	HIP_TRY(hipGetLastError());
	record(e, 4);
	e->prof_with_upper = true;
	e->upper_valid = true;
	return PHYAMD_OK;
}

__global__ void k_accumulate(int n, const double *__restrict__ src, double *__restrict__ dst) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] += src[i];
}

// One evaluation = every tile in turn through the same partial storage; the per-tile results ([lnL | gradient rows | parameter
// sums | root frequency term]: all of them sums over patterns) are added in tile order (fixed: reproducible).
// mode 0: post-order pass only; 1: + pre-order pass and branch gradient; 2: + substitution-parameter sums
int run_tiled(phyamd_engine *e, int mode, int flags) {
	int rc;
	if ((rc = bind_device(e))) return rc;
	const int n = mode == 0 ? 1 : 1 + e->N * e->C + (mode == 2 ? e->np + e->S : 0);
	HIP_TRY(hipMemsetAsync(e->d_total, 0, sizeof(double) * n, e->stream));
	double *inv_total = e->d_total + (size_t)e->N * e->C + 2 * PHYAMD_MAX_PARAMETERS;  // the last entry of the allocation: the +I root term
	HIP_TRY(hipMemsetAsync(inv_total, 0, sizeof(double), e->stream));
	const uint8_t unknown = e->generic ? (uint8_t)e->S : (uint8_t)0xF;
	for (int t = 0; t < e->tiles; t++) {
		const size_t off = (size_t)t * e->P;
		const size_t w = std::min<size_t>((size_t)e->P, (size_t)e->Ptot - off);
		HIP_TRY(hipMemcpy2DAsync(e->d_tipmask, (size_t)e->P, e->d_tip_all + off, (size_t)e->Ptot, w, (size_t)e->T, hipMemcpyDeviceToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(e->d_weights, e->d_weights_all + off, sizeof(double) * w, hipMemcpyDeviceToDevice, e->stream));
		if (w < (size_t)e->P) {  // ragged last tile: unknown tips of weight 0 (L = 1, log L = 0, no gradient)
			HIP_TRY(hipMemset2DAsync(e->d_tipmask + w, (size_t)e->P, unknown, (size_t)e->P - w, (size_t)e->T, e->stream));
			HIP_TRY(hipMemsetAsync(e->d_weights + w, 0, sizeof(double) * ((size_t)e->P - w), e->stream));
		}
		e->all_dirty = true;
		if ((rc = mode == 0 ? run_lower(e, true) : run_gradient(e, flags, mode == 2))) return rc;
		hipLaunchKernelGGL(k_accumulate, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, e->d_result, e->d_total);
		if (e->C >= 2) {  // the +I site-model gradient needs this tile's root partial while it is resident
			if ((rc = launch_root_invariant_term(e, nullptr))) return rc;
			hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(64), 0, e->stream, 1, e->d_inv_part + (e->P + 255) / 256, inv_total);
		}
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(e->d_plk_all + off, e->d_plk, sizeof(double) * w, hipMemcpyDeviceToDevice, e->stream));
	}
	HIP_TRY(hipMemcpyAsync(e->d_result, e->d_total, sizeof(double) * n, hipMemcpyDeviceToDevice, e->stream));
	e->tiled_root_term = mode == 2;
	e->tiled_eval_done = true;
	e->lower_valid = false;  // the resident partials are those of the last tile only
	e->all_dirty = true;
	e->upper_valid = false;
	return PHYAMD_OK;
}

int eval_lower(phyamd_engine *e) { return e->tiles > 1 ? run_tiled(e, 0, 0) : run_lower(e, true); }
int eval_gradient(phyamd_engine *e, int flags, bool with_params = false) {
	return e->tiles > 1 ? run_tiled(e, with_params ? 2 : 1, flags) : run_gradient(e, flags, with_params);
}

// destination of one tip's pattern codes: the engine's own table, or the all-tiles table
uint8_t *tip_row(phyamd_engine *e, int tip) { return e->tiles > 1 ? e->d_tip_all + (size_t)tip * e->Ptot : e->d_tipmask + (size_t)tip * e->P; }

#define NOT_TILED(e, what) \
	if ((e)->tiles > 1) return fail(PHYAMD_EUNSUPPORTED, what " is not available when the patterns are processed in tiles (max_device_bytes)")

void finish_profile(phyamd_engine *e, bool with_upper) {
	if (!e->profiling || !e->prof_pending) return;
	e->prof_pending = false;
	(void)hipEventSynchronize(e->ev[with_upper ? 4 : 2]);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]);
	e->prof.matrices_ms = ms;
	(void)hipEventElapsedTime(&ms, e->ev[1], e->ev[2]);
	e->prof.lower_ms = ms;
	e->prof.upper_ms = e->prof.reduce_ms = 0;
	if (with_upper) {
		(void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]);
		e->prof.upper_ms = ms;
		(void)hipEventElapsedTime(&ms, e->ev[3], e->ev[4]);
		e->prof.reduce_ms = ms;
	}
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------

extern "C" {

const char *phyamd_last_error(void) { return g_last_error.c_str(); }
int phyamd_abi_version(void) { return PHYAMD_ABI_VERSION; }

int phyamd_create(const phyamd_config *cfg, phyamd_engine **out) {
	if (!cfg || !out) return fail(PHYAMD_EINVAL, "null argument");
	*out = nullptr;
	if (cfg->tip_count < 2) return fail(PHYAMD_EINVAL, "tip_count must be >= 2 (got %d)", cfg->tip_count);
	if (cfg->pattern_count < 1) return fail(PHYAMD_EINVAL, "pattern_count must be >= 1 (got %d)", cfg->pattern_count);
	if (cfg->category_count < 1) return fail(PHYAMD_EINVAL, "category_count must be >= 1 (got %d)", cfg->category_count);
	if (cfg->state_count != 4 && cfg->state_count != 20 && cfg->state_count != 60 && cfg->state_count != 61)
		return fail(PHYAMD_EUNSUPPORTED, "state_count %d: kernels are built for 4, 20, 60 and 61 states", cfg->state_count);
	if (cfg->rescale < 0 || cfg->rescale > 2) return fail(PHYAMD_EINVAL, "rescale must be PHYAMD_RESCALE_*");
	int ndev = 0;
	HIP_TRY(hipGetDeviceCount(&ndev));
	if (ndev == 0) return fail(PHYAMD_EDEVICE, "no HIP device visible");
	phyamd_engine *e = new phyamd_engine();
	e->cfg = *cfg;
	e->T = cfg->tip_count;
	e->N = 2 * e->T - 1;
	e->P = e->Ptot = cfg->pattern_count;
	e->S = cfg->state_count;
	e->C = cfg->category_count;
	if (cfg->device >= 0) e->device = cfg->device;
	else if (hipGetDevice(&e->device) != hipSuccess) e->device = 0;
	if (e->device >= ndev) {
		delete e;
		return fail(PHYAMD_EINVAL, "device %d out of range (%d visible)", cfg->device, ndev);
	}
	auto bail = [&](int rc) {
		phyamd_destroy(e);
		return rc;
	};
	{
		hipError_t err = hipSetDevice(e->device);
		if (err != hipSuccess) return bail(fail(PHYAMD_EDEVICE, "hipSetDevice(%d): %s", e->device, hipGetErrorString(err)));
	}
	{
		// Tiling decision.  Working set of one tile of p patterns -- 4 states: about half of the internal nodes' partials are
		// stored (fringe / DEEP nodes are not), plus parked uppers and scratch; 20 / 60 / 61 states: every internal node's lower
		// partial and the level schedule's uppers (two levels' worth at a time).  Resident for all tiles: tip data, weights,
		// per-pattern lnL.  The cap is the caller's
		// max_device_bytes, or -- when none is given -- most of what the device has free right now, so that a problem larger than
		// the card runs in tiles instead of failing in hipMalloc.
		auto need = [&](double p) {
			const double pp = e->S == 4 ? p : std::ceil(p / 16.0) * 16.0, npd = (double)e->C * e->S * pp;
			return 8.0 * ((e->S == 4 ? 0.5 : 1.6) * (double)(e->N - e->T) * npd + 2.0 * npd) + (double)e->T * p;
		};
		double cap = (double)cfg->max_device_bytes;
		const bool automatic = cfg->max_device_bytes <= 0;
		if (automatic) {
			size_t free_bytes = 0, total_bytes = 0;
			cap = hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess ? 0.92 * (double)free_bytes : 0.0;
		}
		if (cap > 0 && need((double)e->Ptot) > cap) {
			const double resident = (double)e->T * e->Ptot + 16.0 * e->Ptot;
			int tiles = 2, per = 0;
			for (;; tiles++) {
				per = ((e->Ptot + tiles - 1) / tiles + 255) / 256 * 256;
				if (need((double)per) + resident <= cap) break;
				if (per <= 256)
					return bail(fail(PHYAMD_ENOMEM, "%s (%.3g bytes) is below the smallest tiled working set (%.3g bytes)",
					                 automatic ? "free device memory" : "max_device_bytes", cap, need(256.0) + resident));
			}
			e->P = per;
			e->tiles = (e->Ptot + per - 1) / per;
		}
	}
	if (cfg->stream) e->stream = (hipStream_t)cfg->stream;
	else {
		hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
		if (err != hipSuccess) return bail(fail(PHYAMD_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(err)));
		e->own_stream = true;
	}
	e->scaling_on = cfg->rescale == PHYAMD_RESCALE_ALWAYS;
	if (const char *env = std::getenv("PHYAMD_FUSE")) e->fusion_enabled = std::atoi(env) != 0;  // A/B switch for the fringe fusion
	if (const char *env = std::getenv("PHYAMD_DEEP")) e->deep_enabled = std::atoi(env) != 0;
	if (e->C > MAX_WAVES) {
		delete e;
		return fail(PHYAMD_EUNSUPPORTED, "category_count %d exceeds %d (one wave per category)", cfg->category_count, MAX_WAVES);
	}
	e->G = std::max(1, 4 / e->C);  // at least 4 waves per workgroup
	e->nblk = (e->P + WAVE * e->G * PPT_UPPER - 1) / (WAVE * e->G * PPT_UPPER);        // pre-order kernel / gradient slabs
	e->nblk_lower = (e->P + WAVE * e->G * PPT_LOWER - 1) / (WAVE * e->G * PPT_LOWER);  // post-order kernel / lnL slab
	{
		// Tree-walk geometry.  The pre-order walk always takes one pattern per thread (see k_upper4_walk); the post-order walk
		// two (measured on 125k..1M-pattern shards: 1 and 4 are 5-10 % slower).  PHYAMD_PPT_WALK_LOWER = 1 overrides (A/B runs).
		const int groups = (e->P + WAVE * e->G - 1) / (WAVE * e->G);  // workgroups at one pattern per thread
		e->ppt_walk_lower = 2;
		if (const char *env = std::getenv("PHYAMD_PPT_WALK_LOWER")) e->ppt_walk_lower = std::atoi(env) == 1 ? 1 : 2;
		e->nblk_walk = (groups + e->ppt_walk_lower - 1) / e->ppt_walk_lower;
		e->nblk_walk_upper = groups;
	}
	if (const char *env = std::getenv("PHYAMD_WALK")) e->walk_enabled = std::atoi(env) != 0;
	if (const char *env = std::getenv("PHYAMD_WALK_LOWER")) e->walk_lower_on = std::atoi(env) != 0;
	if (const char *env = std::getenv("PHYAMD_WALK_UPPER")) e->walk_upper_on = std::atoi(env) != 0;
	if (const char *env = std::getenv("PHYAMD_WALK_PARAMS")) e->walk_params_on = std::atoi(env) != 0;
	e->generic = e->S != 4;
	if (e->generic) {
		e->Pp = (e->P + 15) / 16 * 16;
		const int ppb = e->S == 20 ? GenGeo<2>::PATTERNS_PER_BLOCK : GenGeo<4>::PATTERNS_PER_BLOCK;
		e->nblk = (e->P + ppb - 1) / ppb;
		e->nblk_lower = e->nblk;
		e->nblk_root = (e->P + 255) / 256;
	}
	e->tip_set.assign(e->T, 0);
	e->explicit_host.assign(e->N, 0);
	const size_t msz = (size_t)e->N * e->C * e->S * e->S;
	int rc;
	if ((rc = dev_alloc(e, &e->d_tipmask, (size_t)e->T * e->P))) return bail(rc);
	if (e->tiles > 1) {
		if ((rc = dev_alloc(e, &e->d_tip_all, (size_t)e->T * e->Ptot)) || (rc = dev_alloc(e, &e->d_weights_all, (size_t)e->Ptot)) ||
		    (rc = dev_alloc(e, &e->d_plk_all, (size_t)e->Ptot)) || (rc = dev_alloc(e, &e->d_total, (size_t)1 + e->N * e->C + 2 * PHYAMD_MAX_PARAMETERS)))
			return bail(rc);
	}
	if (e->generic && (rc = dev_alloc(e, &e->d_tipsets, (size_t)256))) return bail(rc);
	if (e->generic && (rc = dev_alloc(e, &e->d_imgs, ((size_t)e->N * e->C + 1) * gen_image_doubles(e)))) return bail(rc);
	// d_lower is sized by the schedule (stored "core" nodes only): ensure_lower_storage
	if ((rc = dev_alloc(e, &e->d_mats, msz))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_dmats, msz))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_model, (size_t)e->S + 2 * e->S * e->S))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_Q, (size_t)e->S * e->S))) return bail(rc);
	if (!e->generic && (rc = dev_alloc(e, &e->d_tiptab, (size_t)e->T * e->C * 64 + (size_t)e->N * 6))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_freqs, (size_t)e->S))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_rates, (size_t)e->C))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_props, (size_t)e->C))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_lengths, (size_t)e->N))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_weights, (size_t)e->P))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_plk, (size_t)e->P))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_wl, (size_t)e->P))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_lnl_part, (size_t)std::max(std::max(std::max(e->nblk, e->nblk_lower), e->nblk_walk), e->nblk_root)))) return bail(rc);
	if (e->generic && (rc = dev_alloc(e, &e->d_Lc, (size_t)e->C * e->P))) return bail(rc);
	e->gpart_row = (size_t)std::max(e->nblk, e->generic ? 0 : e->nblk_walk_upper * e->G);  // the tree-walk kernels write one entry per wave-group
	if ((rc = dev_alloc(e, &e->d_gpart, (size_t)e->N * e->C * e->gpart_row))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_result, (size_t)1 + e->N * e->C + 2 * PHYAMD_MAX_PARAMETERS))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_explicit, (size_t)e->N))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_row_valid, (size_t)e->N * e->C))) return bail(rc);
	{
		hipError_t err = hipHostMalloc(reinterpret_cast<void **>(&e->h_result), sizeof(double) * ((size_t)1 + e->N * e->C + 2 * PHYAMD_MAX_PARAMETERS), hipHostMallocDefault);
		if (err != hipSuccess) return bail(fail(PHYAMD_EDEVICE, "hipHostMalloc: %s", hipGetErrorString(err)));
		err = hipMemsetAsync(e->d_explicit, 0, e->N, e->stream);
		if (err == hipSuccess) err = hipMemsetAsync(e->d_gpart, 0, sizeof(double) * (size_t)e->N * e->C * e->gpart_row, e->stream);
		if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
		if (err != hipSuccess) return bail(fail(PHYAMD_EDEVICE, "memset: %s", hipGetErrorString(err)));
		for (auto &ev : e->ev) {
			err = hipEventCreate(&ev);
			if (err != hipSuccess) return bail(fail(PHYAMD_EDEVICE, "hipEventCreate: %s", hipGetErrorString(err)));
		}
	}
	*out = e;
	return PHYAMD_OK;
}

void phyamd_destroy(phyamd_engine *e) {
	if (!e) return;
	(void)hipSetDevice(e->device);
	if (e->stream) (void)hipStreamSynchronize(e->stream);
	for (void *p : {(void *)e->d_branch, (void *)e->d_path_steps, (void *)e->d_path_upper, (void *)e->d_path_tmp, (void *)e->d_path_lower, (void *)e->d_Bw, (void *)e->d_pbuf, (void *)e->d_Fw, (void *)e->d_gacc, (void *)e->d_gen_scratch, (void *)e->d_rf_part, (void *)e->d_B, (void *)e->d_dpm, (void *)e->d_dptab, (void *)e->d_ppart, (void *)e->d_imgs, (void *)e->d_tipmask, (void *)e->d_tip_all, (void *)e->d_weights_all, (void *)e->d_plk_all, (void *)e->d_total, (void *)e->d_tipsets, (void *)e->d_pg_nodes, (void *)e->d_pg_core, (void *)e->d_pg_den, (void *)e->d_pg_Gw, (void *)e->d_pg_B, (void *)e->d_lower, (void *)e->d_upper, (void *)e->d_mats, (void *)e->d_dmats, (void *)e->d_model, (void *)e->d_Q, (void *)e->d_Lc, (void *)e->d_inv_part, (void *)e->d_tiptab,
	                (void *)e->d_freqs, (void *)e->d_rates, (void *)e->d_props, (void *)e->d_lengths, (void *)e->d_weights, (void *)e->d_plk, (void *)e->d_wl,
	                (void *)e->d_lscale, (void *)e->d_lnl_part, (void *)e->d_gpart, (void *)e->d_result, (void *)e->d_explicit, (void *)e->d_row_valid,
	                (void *)e->d_lower_ops, (void *)e->d_upper_ops, (void *)e->d_walk_lower_ops, (void *)e->d_walk_upper_ops, (void *)e->d_inc_ops, (void *)e->d_Qpi})
		if (p) (void)hipFree(p);
	if (e->h_result) (void)hipHostFree(e->h_result);
	for (auto &ev : e->ev)
		if (ev) (void)hipEventDestroy(ev);
	if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
	delete e;
}

#define CHECK_ENGINE(e) \
	if (!(e)) return fail(PHYAMD_EINVAL, "null engine")

int phyamd_set_tip_states(phyamd_engine *e, int tip, const uint8_t *states) {
	CHECK_ENGINE(e);
	if (tip < 0 || tip >= e->T || !states) return fail(PHYAMD_EINVAL, "bad tip %d or null states", tip);
	int rc;
	if ((rc = bind_device(e))) return rc;
	std::vector<uint8_t> mask(e->Ptot);
	if (e->generic)
		for (int k = 0; k < e->Ptot; k++) mask[k] = states[k] < e->S ? states[k] : (uint8_t)e->S;  // raw codes; S = unknown
	else
		for (int k = 0; k < e->Ptot; k++) mask[k] = states[k] < 4 ? (uint8_t)(1u << states[k]) : (uint8_t)0xF;  // code >= S: unknown (treelikelihood4.c:946-988)
	HIP_TRY(hipMemcpyAsync(tip_row(e, tip), mask.data(), e->Ptot, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->tip_set[tip] = 1;
	e->all_dirty = true;
	e->stored.valid = false;
	return PHYAMD_OK;
}

int phyamd_set_tip_partials(phyamd_engine *e, int tip, const double *partials) {
	CHECK_ENGINE(e);
	if (tip < 0 || tip >= e->T || !partials) return fail(PHYAMD_EINVAL, "bad tip %d or null partials", tip);
	int rc;
	if ((rc = bind_device(e))) return rc;
	std::vector<uint8_t> mask(e->Ptot);
	if (e->generic) {  // 0/1 vectors: one state, all states, or a set of states (datatype.c:212-240)
		const int S = e->S;
		bool grew = false;
		for (int k = 0; k < e->Ptot; k++) {
			int ones = 0, last = -1;
			unsigned long long members = 0;
			for (int s = 0; s < S; s++) {
				const double v = partials[(size_t)k * S + s];
				if (v == 1.0) ones++, last = s, members |= 1ull << s;
				else if (v != 0.0) return fail(PHYAMD_EUNSUPPORTED, "tip %d pattern %d: only 0/1 tip partials are built", tip, k);
			}
			if (ones == 1) mask[k] = (uint8_t)last;
			else if (ones == S) mask[k] = (uint8_t)S;
			else {
				size_t q = std::find(e->tipsets_host.begin(), e->tipsets_host.end(), members) - e->tipsets_host.begin();
				if (q == e->tipsets_host.size()) {
					if ((int)q + S + 1 > 255) return fail(PHYAMD_EUNSUPPORTED, "tip %d pattern %d: more than %d distinct ambiguity sets", tip, k, 255 - S);
					e->tipsets_host.push_back(members);
					grew = true;
				}
				mask[k] = (uint8_t)(S + 1 + q);
			}
		}
		if (grew)
			HIP_TRY(hipMemcpyAsync(e->d_tipsets, e->tipsets_host.data(), e->tipsets_host.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipMemcpyAsync(tip_row(e, tip), mask.data(), e->Ptot, hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		e->tip_set[tip] = 1;
		e->all_dirty = true;
		return PHYAMD_OK;
	}
	for (int k = 0; k < e->Ptot; k++) {
		unsigned m = 0;
		for (int s = 0; s < 4; s++) {
			const double v = partials[(size_t)k * 4 + s];
			if (v == 1.0) m |= 1u << s;
			else if (v != 0.0)
				return fail(PHYAMD_EUNSUPPORTED, "tip %d pattern %d: tip partials other than 0/1 ambiguity masks are not built in this revision", tip, k);
		}
		mask[k] = (uint8_t)m;
	}
	HIP_TRY(hipMemcpyAsync(tip_row(e, tip), mask.data(), e->Ptot, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->tip_set[tip] = 1;
	e->all_dirty = true;
	e->stored.valid = false;
	return PHYAMD_OK;
}

int phyamd_set_pattern_weights(phyamd_engine *e, const double *weights) {
	CHECK_ENGINE(e);
	if (!weights) return fail(PHYAMD_EINVAL, "null weights");
	int rc;
	if ((rc = bind_device(e))) return rc;
	HIP_TRY(hipMemcpyAsync(e->tiles > 1 ? e->d_weights_all : e->d_weights, weights, sizeof(double) * e->Ptot, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->have_weights = true;
	e->all_dirty = true;
	e->stored.valid = false;
	return PHYAMD_OK;
}

int phyamd_set_topology(phyamd_engine *e, const int32_t *left, const int32_t *right, int root) {
	CHECK_ENGINE(e);
	if (!left || !right) return fail(PHYAMD_EINVAL, "null topology arrays");
	int rc;
	if ((rc = bind_device(e))) return rc;
	std::vector<int32_t> old_l = e->left, old_r = e->right;
	const int old_root = e->root;
	e->left.assign(left, left + e->N);
	e->right.assign(right, right + e->N);
	e->root = root;
	if ((rc = build_schedule(e))) {
		e->left = old_l;
		e->right = old_r;
		e->root = old_root;
		if (e->have_topology) (void)build_schedule(e);
		return rc;
	}
	if ((rc = upload_schedule(e))) return rc;
	if ((rc = ensure_lower_storage(e))) return rc;
	e->have_topology = true;
	e->matrices_dirty = true;
	e->upper_valid = false;
	e->all_dirty = true;  // every partial belongs to the old tree
	e->lower_valid = false;
	e->schedule_epoch++;
	e->stored.valid = false;  // topology is not part of phyamd_store
	return PHYAMD_OK;
}

int phyamd_set_branch_lengths(phyamd_engine *e, const double *lengths) {
	CHECK_ENGINE(e);
	if (!lengths) return fail(PHYAMD_EINVAL, "null lengths");
	int rc;
	if ((rc = bind_device(e))) return rc;
	e->lengths.assign(lengths, lengths + e->N);
	if (e->have_topology) e->lengths[e->root] = 0.0;
	HIP_TRY(hipMemcpyAsync(e->d_lengths, e->lengths.data(), sizeof(double) * e->N, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->have_lengths = true;
	e->matrices_dirty = true;
	e->all_dirty = true;  // the whole vector: every node is recomputed (SingleTreeLikelihood_update_all_nodes)
	return PHYAMD_OK;
}

int phyamd_set_branch_length(phyamd_engine *e, int node, double length) {
	CHECK_ENGINE(e);
	if (!e->have_topology || !e->have_lengths) return fail(PHYAMD_EINVAL, "phyamd_set_topology and phyamd_set_branch_lengths come first");
	if (node < 0 || node >= e->N || node == e->root) return fail(PHYAMD_EINVAL, "node %d has no branch", node);
	int rc;
	if ((rc = bind_device(e))) return rc;
	if (e->lengths[node] == length) return PHYAMD_OK;
	e->lengths[node] = length;
	HIP_TRY(hipMemcpyAsync(e->d_lengths + node, &e->lengths[node], sizeof(double), hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->matrices_dirty = true;  // all P(t) are re-formed (microseconds); only the partials above `node` are recomputed
	e->changed.push_back(node);
	e->upper_valid = false;
	return PHYAMD_OK;
}

int phyamd_store(phyamd_engine *e) {
	CHECK_ENGINE(e);
	NOT_TILED(e, "phyamd_store");
	int rc;
	if ((rc = bind_device(e))) return rc;
	for (uint8_t x : e->explicit_host)
		if (x) return fail(PHYAMD_EUNSUPPORTED, "phyamd_store does not cover explicit node matrices");
	if ((rc = run_lower(e, true))) return rc;  // the state that is stored is an evaluated one (a no-op when nothing is pending)
	if (!e->two_slots) {  // first store: a second slot per stored node (allocate_storage(tlk, 1), treelikelihood.c:977-1003), contents kept
		const size_t npd = node_partial_doubles(e), old_slots = e->lower_alloc_cores, want = (size_t)std::max(1, e->core_count) * 2;
		double *lower = nullptr, *lscale = nullptr;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lower), want * npd * sizeof(double)));
		HIP_TRY(hipMemcpyAsync(lower, e->d_lower, old_slots * npd * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
		if (e->d_lscale) {
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lscale), want * e->P * sizeof(double)));
			HIP_TRY(hipMemcpyAsync(lscale, e->d_lscale, old_slots * e->P * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
		}
		HIP_TRY(hipStreamSynchronize(e->stream));
		e->device_bytes += (int64_t)(want * npd * sizeof(double)) + (lscale ? (int64_t)(want * e->P * sizeof(double)) : 0);
		dev_free(e, &e->d_lower, old_slots * npd);
		if (e->d_lscale) dev_free(e, &e->d_lscale, old_slots * (size_t)e->P);
		e->d_lower = lower;
		e->d_lscale = lscale;
		e->lower_alloc_cores = want;
		e->two_slots = true;
	}
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	auto &st = e->stored;
	st.lnl = e->h_result[0];
	st.lengths = e->lengths;
	st.model = e->model;
	st.freqs = e->freqs;
	st.rates = e->rates;
	st.props = e->props;
	st.have_eigen = e->have_eigen;
	st.scaling_on = e->scaling_on;
	st.core_index = e->core_index;
	st.epoch = e->schedule_epoch;
	st.valid = true;
	return PHYAMD_OK;
}

int phyamd_restore(phyamd_engine *e) {
	CHECK_ENGINE(e);
	if (!e->stored.valid) return fail(PHYAMD_EINVAL, "nothing is stored (phyamd_store has not been called, or tree / data changed since)");
	int rc;
	if ((rc = bind_device(e))) return rc;
	const phyamd_engine::Stored st = e->stored;  // the setters below write the engine's own copies
	const int S = e->S;
	if (st.have_eigen && (rc = phyamd_set_eigen(e, st.model.data(), st.model.data() + S, st.model.data() + S + S * S))) return rc;
	if ((rc = phyamd_set_frequencies(e, st.freqs.data()))) return rc;
	if ((rc = phyamd_set_category_rates(e, st.rates.data(), st.props.data()))) return rc;
	if ((rc = phyamd_set_branch_lengths(e, st.lengths.data()))) return rc;
	e->changed.clear();
	e->upper_valid = false;
	if (st.epoch == e->schedule_epoch && st.scaling_on == e->scaling_on && e->two_slots) {
		// the stored partials are still in their slots: point the nodes back at them (treelikelihood.c:116-124) and
		// re-integrate the root, whose per-pattern outputs belong to the discarded state
		bool moved = false;
		for (int n = e->T; n < e->N; n++)
			if (e->core_index[n] != st.core_index[n]) {
				e->core_index[n] = st.core_index[n];
				moved = true;
			}
		if (moved) {
			refresh_op_cores(e);
			if ((rc = upload_schedule(e))) return rc;
		}
		e->all_dirty = false;
		e->lower_valid = true;
		e->force_root = true;
	}  // else: slots were reassigned since (schedule rebuilt, rescaling switched on): the restored parameters are recomputed in full
	return PHYAMD_OK;
}

int phyamd_update_all_nodes(phyamd_engine *e) {
	CHECK_ENGINE(e);
	e->all_dirty = true;
	e->matrices_dirty = true;
	return PHYAMD_OK;
}

int phyamd_set_eigen(phyamd_engine *e, const double *eval, const double *evec, const double *ivec) {
	CHECK_ENGINE(e);
	if (!eval || !evec || !ivec) return fail(PHYAMD_EINVAL, "null eigen system");
	int rc;
	if ((rc = bind_device(e))) return rc;
	const int S = e->S;
	e->model.resize((size_t)S + 2 * S * S);
	std::copy(eval, eval + S, e->model.begin());
	std::copy(evec, evec + S * S, e->model.begin() + S);
	std::copy(ivec, ivec + S * S, e->model.begin() + S + S * S);
	HIP_TRY(hipMemcpyAsync(e->d_model, e->model.data(), sizeof(double) * e->model.size(), hipMemcpyHostToDevice, e->stream));
	// Q = evec diag(eval) ivec: the gradient kernels use (dP/dt) p = Q (P p)
	std::vector<double> Q((size_t)S * S, 0.0);
	for (int i = 0; i < S; i++)
		for (int j = 0; j < S; j++) {
			double q = 0.0;
			for (int k = 0; k < S; k++) q += evec[i * S + k] * eval[k] * ivec[k * S + j];
			Q[(size_t)i * S + j] = q;
		}
	HIP_TRY(hipMemcpyAsync(e->d_Q, Q.data(), sizeof(double) * Q.size(), hipMemcpyHostToDevice, e->stream));
	e->Q_host = Q;
	e->qpi_dirty = true;
	e->have_Q = true;
	e->qimg_dirty = true;
	std::fill(e->explicit_host.begin(), e->explicit_host.end(), 0);
	HIP_TRY(hipMemsetAsync(e->d_explicit, 0, e->N, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->have_eigen = true;
	e->matrices_dirty = true;
	e->params_dirty = true;
	e->all_dirty = true;
	return PHYAMD_OK;
}

int phyamd_set_frequencies(phyamd_engine *e, const double *freqs) {
	CHECK_ENGINE(e);
	if (!freqs) return fail(PHYAMD_EINVAL, "null freqs");
	int rc;
	if ((rc = bind_device(e))) return rc;
	e->freqs.assign(freqs, freqs + e->S);
	HIP_TRY(hipMemcpyAsync(e->d_freqs, e->freqs.data(), sizeof(double) * e->S, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->have_freqs = true;
	e->all_dirty = true;
	e->qpi_dirty = true;
	return PHYAMD_OK;
}

int phyamd_set_category_rates(phyamd_engine *e, const double *rates, const double *proportions) {
	CHECK_ENGINE(e);
	if (!rates || !proportions) return fail(PHYAMD_EINVAL, "null rates/proportions");
	int rc;
	if ((rc = bind_device(e))) return rc;
	e->rates.assign(rates, rates + e->C);
	e->props.assign(proportions, proportions + e->C);
	HIP_TRY(hipMemcpyAsync(e->d_rates, e->rates.data(), sizeof(double) * e->C, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipMemcpyAsync(e->d_props, e->props.data(), sizeof(double) * e->C, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->have_rates = true;
	e->matrices_dirty = true;
	e->all_dirty = true;
	return PHYAMD_OK;
}

int phyamd_set_node_matrices(phyamd_engine *e, int node, const double *matrices) {
	CHECK_ENGINE(e);
	if (node < 0 || node >= e->N || !matrices) return fail(PHYAMD_EINVAL, "bad node %d or null matrices", node);
	int rc;
	if ((rc = bind_device(e))) return rc;
	const size_t sz = (size_t)e->C * e->S * e->S;
	HIP_TRY(hipMemcpyAsync(e->d_mats + (size_t)node * sz, matrices, sizeof(double) * sz, hipMemcpyHostToDevice, e->stream));
	e->explicit_host[node] = 1;
	HIP_TRY(hipMemcpyAsync(e->d_explicit + node, &e->explicit_host[node], 1, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->matrices_dirty = true;  // the tip tables are built from d_mats
	if (e->have_topology && node != e->root) e->changed.push_back(node);  // like a branch-length change of this one node
	e->upper_valid = false;
	return PHYAMD_OK;
}

int phyamd_set_rate_matrix(phyamd_engine *e, const double *Q) {
	CHECK_ENGINE(e);
	if (!Q) return fail(PHYAMD_EINVAL, "null Q");
	int rc;
	if ((rc = bind_device(e))) return rc;
	HIP_TRY(hipMemcpyAsync(e->d_Q, Q, sizeof(double) * e->S * e->S, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->Q_host.assign(Q, Q + (size_t)e->S * e->S);
	e->qpi_dirty = true;
	e->have_Q = true;
	e->qimg_dirty = true;
	return PHYAMD_OK;
}

int phyamd_log_likelihood(phyamd_engine *e, double *lnl) {
	CHECK_ENGINE(e);
	if (!lnl) return fail(PHYAMD_EINVAL, "null lnl");
	int rc;
	if ((rc = eval_lower(e))) return rc;
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	finish_profile(e, false);
	*lnl = e->h_result[0];
	return PHYAMD_OK;
}

int phyamd_log_likelihood_device(phyamd_engine *e, double *device_out) {
	CHECK_ENGINE(e);
	if (!device_out) return fail(PHYAMD_EINVAL, "null device_out");
	int rc;
	if ((rc = eval_lower(e))) return rc;
	HIP_TRY(hipMemcpyAsync(device_out, e->d_result, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
	return PHYAMD_OK;
}

int phyamd_gradient_device(phyamd_engine *e, int flags, double *device_out) {
	CHECK_ENGINE(e);
	if (!device_out) return fail(PHYAMD_EINVAL, "null device_out");
	int rc;
	if ((rc = eval_gradient(e, flags))) return rc;
	HIP_TRY(hipMemcpyAsync(device_out, e->d_result, sizeof(double) * ((size_t)1 + e->N * e->C), hipMemcpyDeviceToDevice, e->stream));
	return PHYAMD_OK;
}

int phyamd_gradient(phyamd_engine *e, int flags, double *lnl, double *cat_gradient) {
	CHECK_ENGINE(e);
	if (!cat_gradient) return fail(PHYAMD_EINVAL, "null cat_gradient");
	int rc;
	if ((rc = eval_gradient(e, flags))) return rc;
	const size_t n = (size_t)1 + e->N * e->C;
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	finish_profile(e, true);
	const double l = e->h_result[0];
	if (lnl) *lnl = l;
	if (std::isnan(l) || std::isinf(l)) {  // treelikelihood.c:327-332
		for (size_t i = 0; i < n - 1; i++) cat_gradient[i] = NAN;
	} else
		std::memcpy(cat_gradient, e->h_result + 1, sizeof(double) * (n - 1));
	return PHYAMD_OK;
}

int phyamd_branch_gradient(phyamd_engine *e, int flags, const double *rates_without_mu, double *lnl, double *branch_gradient) {
	CHECK_ENGINE(e);
	if (!branch_gradient) return fail(PHYAMD_EINVAL, "null branch_gradient");
	std::vector<double> cg((size_t)e->N * e->C);
	int rc;
	if ((rc = phyamd_gradient(e, flags, lnl, cg.data()))) return rc;
	const double *r = rates_without_mu ? rates_without_mu : e->rates.data();
	for (int n = 0; n < e->N; n++) {  // gradient_branch_length_from_cat_inplace, treelikelihood.c:3129-3143
		if (e->C == 1) {
			branch_gradient[n] = cg[n];  // catCount == 1: no rate/weight factor (treelikelihood.c:3258-3266)
			continue;
		}
		double g = cg[(size_t)n * e->C] * e->props[0] * r[0];
		for (int c = 1; c < e->C; c++) g += cg[(size_t)n * e->C + c] * e->props[c] * r[c];
		branch_gradient[n] = g;
	}
	return PHYAMD_OK;
}

int phyamd_set_rate_matrix_derivatives(phyamd_engine *e, int count, const double *dQ) {
	CHECK_ENGINE(e);
	if (count < 0 || count > PHYAMD_MAX_PARAMETERS) return fail(PHYAMD_EINVAL, "count %d outside 0..%d", count, PHYAMD_MAX_PARAMETERS);
	if (count > 0 && !dQ) return fail(PHYAMD_EINVAL, "null dQ");
	e->np = count;
	e->dQ_host.assign(dQ, dQ + (size_t)count * e->S * e->S);
	e->params_dirty = true;
	return PHYAMD_OK;
}

int phyamd_parameter_gradient(phyamd_engine *e, int flags, double *lnl, double *cat_gradient, double *parameter_gradient) {
	CHECK_ENGINE(e);
	if (!parameter_gradient) return fail(PHYAMD_EINVAL, "null parameter_gradient");
	int rc;
	if ((rc = eval_gradient(e, flags, true))) return rc;
	const size_t ncat = (size_t)e->N * e->C, n = 1 + ncat + e->np;
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	finish_profile(e, true);
	const double l = e->h_result[0];
	if (lnl) *lnl = l;
	const bool bad = std::isnan(l) || std::isinf(l);  // treelikelihood.c:327-332
	for (size_t i = 0; cat_gradient && i < ncat; i++) cat_gradient[i] = bad ? NAN : e->h_result[1 + i];
	for (int i = 0; i < e->np; i++) parameter_gradient[i] = bad ? NAN : e->h_result[1 + ncat + i];
	return PHYAMD_OK;
}

int phyamd_parameter_gradient_device(phyamd_engine *e, int flags, double *device_out) {
	CHECK_ENGINE(e);
	if (!device_out) return fail(PHYAMD_EINVAL, "null device_out");
	int rc;
	if ((rc = eval_gradient(e, flags, true))) return rc;
	HIP_TRY(hipMemcpyAsync(device_out, e->d_result, sizeof(double) * ((size_t)1 + e->N * e->C + e->np + e->S), hipMemcpyDeviceToDevice, e->stream));
	return PHYAMD_OK;
}

int phyamd_root_frequency_term(phyamd_engine *e, double *out) {
	CHECK_ENGINE(e);
	if (!out) return fail(PHYAMD_EINVAL, "null out");
	int rc;
	if ((rc = bind_device(e))) return rc;
	if ((rc = check_ready(e))) return rc;
	if (e->tiles > 1) {  // the per-tile terms were summed by the last phyamd_parameter_gradient
		if (!e->tiled_root_term) return fail(PHYAMD_EINVAL, "with tiled patterns the root frequency term comes with phyamd_parameter_gradient: call that first");
		HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result + 1 + (size_t)e->N * e->C + e->np, sizeof(double) * e->S, hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		std::memcpy(out, e->h_result, sizeof(double) * e->S);
		return PHYAMD_OK;
	}
	if (e->core_index.empty() || e->core_index[e->root] < 0 || !e->d_lower) return fail(PHYAMD_EINVAL, "no evaluation has been run yet");
	if ((rc = launch_root_frequency_term(e, nullptr))) return rc;
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_rf_part + (size_t)((e->P + 255) / 256) * e->S, sizeof(double) * e->S, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	std::memcpy(out, e->h_result, sizeof(double) * e->S);
	return PHYAMD_OK;
}

// the sibling subtree `s` as child_message / PathStep take it
static void describe_subtree(const phyamd_engine *e, int s, PathStep &st) {
	st.kind = e->node_kind[s];
	st.node = s;
	st.core = e->core_index[s];
	st.t0 = st.t1 = st.t2 = st.inner = -1;
	if (st.kind == CH_CHERRY) {
		st.t0 = e->left[s];
		st.t1 = e->right[s];
	} else if (st.kind == CH_CHERRY_TIP) {
		const int l = e->left[s], r = e->right[s];
		st.inner = l < e->T ? r : l;
		st.t2 = l < e->T ? l : r;
		st.t0 = e->left[st.inner];
		st.t1 = e->right[st.inner];
	}
}

// upper partial of `node` into d_path_upper (and, for a node without a stored lower partial, that partial into d_path_lower)
static int rebuild_path_upper(phyamd_engine *e, int node) {
	int rc;
	const size_t npd = node_partial_doubles(e);
	if (!e->d_path_upper && ((rc = dev_alloc(e, &e->d_path_upper, npd)) || (rc = dev_alloc(e, &e->d_path_lower, npd)) ||
	                         (rc = dev_alloc(e, &e->d_path_steps, (size_t)e->N))))
		return rc;
	if (e->generic && !e->d_path_tmp && (rc = dev_alloc(e, &e->d_path_tmp, npd))) return rc;
	std::vector<int> path;  // node, parent, ..., root
	for (int a = node; a >= 0; a = e->parent[a]) path.push_back(a);
	const int m = (int)path.size() - 1;  // steps
	if (!e->generic) {
		std::vector<PathStep> steps(m);
		for (int j = 0; j < m; j++) {
			const int par = path[m - j], child = path[m - j - 1];
			steps[j].mat = par == e->root ? -1 : par;
			describe_subtree(e, e->left[par] == child ? e->right[par] : e->left[par], steps[j]);
		}
		HIP_TRY(hipMemcpyAsync(e->d_path_steps, steps.data(), sizeof(PathStep) * m, hipMemcpyHostToDevice, e->stream));
		const dim3 grid((e->P + WAVE - 1) / WAVE), block(WAVE, e->C);
		if (e->scaling_on)
			hipLaunchKernelGGL(k_path_upper4<true>, grid, block, sizeof(double) * e->C * WAVE, e->stream, e->d_path_steps, m, e->T, e->P, e->C, e->d_tipmask, e->d_lower,
			                   e->d_mats, e->d_tiptab, e->d_path_upper);
		else
			hipLaunchKernelGGL(k_path_upper4<false>, grid, block, 0, e->stream, e->d_path_steps, m, e->T, e->P, e->C, e->d_tipmask, e->d_lower, e->d_mats,
			                   e->d_tiptab, e->d_path_upper);
		if (node >= e->T && e->core_index[node] < 0) {  // fringe / DEEP node: its own partial is not stored either
			PathStep self;
			describe_subtree(e, node, self);
			hipLaunchKernelGGL(k_unstored_partial4, grid, block, 0, e->stream, self.kind, node, self.t0, self.t1, self.t2, self.inner, e->T, e->P, e->C, e->d_tipmask,
			                   e->d_mats, e->d_tiptab, e->d_path_lower);
		}
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(e->stream));  // `steps` is a stack-lifetime buffer
	} else {
		// one launch per step, ping-pong between two buffers so that the last step lands in d_path_upper
		const size_t msz = (size_t)e->C * e->S * e->S;
		const int pb = (e->P + 255) / 256;
		double *cur = nullptr;
		for (int j = 0; j < m; j++) {
			const int par = path[m - j], child = path[m - j - 1];
			const int sib = e->left[par] == child ? e->right[par] : e->left[par];
			double *dst = ((m - 1 - j) & 1) ? e->d_path_tmp : e->d_path_upper;
			const double *sp = sib < e->T ? nullptr : e->d_lower + (size_t)e->core_index[sib] * npd;
			hipLaunchKernelGGL(k_path_step_gen, dim3(pb), dim3(256), 0, e->stream, e->P, e->Pp, e->S, e->C, par == e->root ? (const double *)nullptr : e->d_mats + (size_t)par * msz,
			                   par == e->root ? (const double *)nullptr : cur, e->d_mats + (size_t)sib * msz, sp,
			                   sib < e->T ? e->d_tipmask + (size_t)sib * e->P : (const uint8_t *)nullptr, e->d_tipsets, dst);
			if (e->scaling_on) hipLaunchKernelGGL(k_path_scale_gen, dim3(pb), dim3(256), 0, e->stream, e->P, e->Pp, e->S, e->C, dst);
			cur = dst;
		}
		HIP_TRY(hipGetLastError());
	}
	e->path_node = node;
	return PHYAMD_OK;
}

int phyamd_branch_log_likelihood(phyamd_engine *e, int node, double length, double *lnl, double *d1, double *d2) {
	CHECK_ENGINE(e);
	NOT_TILED(e, "the single-branch evaluation");
	if (node < 0 || node >= e->N || node == e->root) return fail(PHYAMD_EINVAL, "node %d has no branch", node);
	if (!e->have_eigen) return fail(PHYAMD_EINVAL, "the single-branch evaluation needs the eigen system (phyamd_set_eigen)");
	int rc;
	if ((rc = bind_device(e))) return rc;
	const size_t npd = node_partial_doubles(e);
	// the two partials that meet on the branch: resident after a keep-partials gradient, else the upper one is rebuilt by a
	// walk down the path from the root (pending changes are evaluated first; the result is kept until partials change)
	const bool resident = e->keep_partials && e->upper_valid;
	const double *up, *low;
	int fold = 0;
	if (resident) {
		up = e->d_upper + (size_t)e->upper_slot[node] * npd;
		low = node < e->T ? nullptr : e->d_lower + (size_t)e->core_index[node] * npd;
		fold = e->upper_fold ? 1 : 0;
	} else {
		if ((rc = run_lower(e, true))) return rc;
		if (e->path_node != node && (rc = rebuild_path_upper(e, node))) return rc;
		up = e->d_path_upper;
		low = node < e->T ? nullptr : (e->core_index[node] >= 0 ? e->d_lower + (size_t)e->core_index[node] * npd : e->d_path_lower);
	}
	const int C = e->C, S = e->S, S2 = S * S;
	const int per_block = e->generic ? 256 : WAVE, nb = (e->P + per_block - 1) / per_block;
	const size_t msz = (size_t)C * (e->generic ? 4 : 3) * S2, need = msz + (size_t)3 * nb + 3;
	if (!e->d_branch && (rc = dev_alloc(e, &e->d_branch, need))) return rc;
	// P(t r_c), r_c Q P, r_c^2 Q Q P from the eigen system (host: S^3 per category)
	const double *ev = e->model.data(), *U = ev + S, *Ui = U + S2;
	std::vector<double> pm(msz), ex(S);
	const int stride = e->generic ? 4 * S2 : 48;
	for (int c = 0; c < C; c++) {
		const double r = e->rates[c], t = length * r;
		for (int a = 0; a < S; a++) ex[a] = std::exp(ev[a] * t);
		for (int i = 0; i < S; i++)
			for (int j = 0; j < S; j++) {
				double p0 = 0.0, p1 = 0.0, p2 = 0.0;
				for (int a = 0; a < S; a++) {
					const double w = U[i * S + a] * Ui[a * S + j] * ex[a];
					p0 += w;
					p1 += w * ev[a];
					p2 += w * ev[a] * ev[a];
				}
				pm[(size_t)c * stride + i * S + j] = std::fabs(p0);  // substmodel.c:552
				pm[(size_t)c * stride + S2 + i * S + j] = r * p1;
				pm[(size_t)c * stride + 2 * S2 + i * S + j] = r * r * p2;
			}
	}
	HIP_TRY(hipMemcpyAsync(e->d_branch, pm.data(), sizeof(double) * pm.size(), hipMemcpyHostToDevice, e->stream));
	// rescaled evaluations: the per-pattern lnL of the resident evaluation anchors the stored (scaled) partials
	const double *plk = e->scaling_on ? e->d_plk : nullptr;
	const double *m0 = e->d_mats + (size_t)node * C * S2;
	double *part = e->d_branch + msz;
	if (e->generic) {
		if (plk)
			for (int c = 0; c < C; c++)
				HIP_TRY(hipMemcpyAsync(e->d_branch + (size_t)c * stride + 3 * S2, m0 + (size_t)c * S2, sizeof(double) * S2, hipMemcpyDeviceToDevice, e->stream));
		hipLaunchKernelGGL(k_branch_eval_gen, dim3(nb), dim3(256), 0, e->stream, node, e->T, e->P, e->Pp, S, C, up, low, e->d_tipmask, e->d_tipsets, e->d_branch,
		                   e->d_freqs, fold, e->d_props, e->d_weights, plk, part);
	} else
		hipLaunchKernelGGL(k_branch_eval4, dim3(nb), dim3(WAVE, C), sizeof(double) * 4 * C * WAVE, e->stream, e->P, C, up, low,
		                   node < e->T ? e->d_tipmask + (size_t)node * e->P : (const uint8_t *)nullptr, e->d_branch, e->d_freqs, fold, e->d_props, e->d_weights, plk,
		                   m0, part);
	hipLaunchKernelGGL(k_reduce_rows, dim3(3), dim3(64), 0, e->stream, part, nb, (const uint8_t *)nullptr, part + (size_t)3 * nb);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_result, part + (size_t)3 * nb, sizeof(double) * 3, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));  // also covers pm (stack-lifetime buffer)
	if (lnl) *lnl = e->h_result[0];
	if (d1) *d1 = e->h_result[1];
	if (d2) *d2 = e->h_result[2];
	return PHYAMD_OK;
}

int phyamd_compress_patterns(int device, int32_t taxon_count, int64_t site_count, const uint8_t *const *rows, const uint8_t *symbol_codes,
                             int32_t *pattern_count, uint8_t *patterns, double *weights) {
	return compress_patterns_device(device, taxon_count, site_count, rows, symbol_codes, pattern_count, patterns, weights);
}

int phyamd_root_invariant_term(phyamd_engine *e, double *out) {
	CHECK_ENGINE(e);
	if (!out) return fail(PHYAMD_EINVAL, "null out");
	if (e->C < 2) return fail(PHYAMD_EINVAL, "the invariant-class term needs at least two categories");
	int rc;
	if ((rc = bind_device(e))) return rc;
	if ((rc = check_ready(e))) return rc;
	const double *src;
	if (e->tiles > 1) {  // summed over the tiles by the last evaluation (one entry behind everything else in the total)
		if (!e->tiled_eval_done) return fail(PHYAMD_EINVAL, "no evaluation has been run yet");
		src = e->d_total + (size_t)e->N * e->C + 2 * PHYAMD_MAX_PARAMETERS;
	} else {
		if ((rc = launch_root_invariant_term(e, nullptr))) return rc;
		src = e->d_inv_part + (e->P + 255) / 256;
	}
	HIP_TRY(hipMemcpyAsync(e->h_result, src, sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	*out = e->h_result[0];
	return PHYAMD_OK;
}

int phyamd_synchronize(phyamd_engine *e) {
	CHECK_ENGINE(e);
	int rc;
	if ((rc = bind_device(e))) return rc;
	HIP_TRY(hipStreamSynchronize(e->stream));
	return PHYAMD_OK;
}

int phyamd_get_pattern_log_likelihoods(phyamd_engine *e, double *out) {
	CHECK_ENGINE(e);
	if (!out) return fail(PHYAMD_EINVAL, "null out");
	int rc;
	if ((rc = bind_device(e))) return rc;
	HIP_TRY(hipMemcpyAsync(out, e->tiles > 1 ? e->d_plk_all : e->d_plk, sizeof(double) * e->Ptot, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return PHYAMD_OK;
}

int phyamd_get_partials(phyamd_engine *e, int node, int upper, double *out) {
	CHECK_ENGINE(e);
	NOT_TILED(e, "reading partials back");
	if (!out || node < 0 || node >= e->N) return fail(PHYAMD_EINVAL, "bad node %d or null out", node);
	int rc;
	if ((rc = bind_device(e))) return rc;
	const size_t np = node_partial_doubles(e);
	if (upper) {
		if (!e->keep_partials || !e->upper_valid) return fail(PHYAMD_EINVAL, "upper partials need phyamd_set_keep_partials(1) before phyamd_gradient");
		if (node == e->root) return fail(PHYAMD_EINVAL, "the root has no upper partial");
	}
	if (!upper && node < e->T) {  // rebuild the replicated tip partial from its mask / code
		std::vector<uint8_t> mask(e->P);
		HIP_TRY(hipMemcpyAsync(mask.data(), e->d_tipmask + (size_t)node * e->P, e->P, hipMemcpyDeviceToHost, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		const int S = e->S;
		for (int c = 0; c < e->C; c++)
			for (int k = 0; k < e->P; k++)
				for (int s = 0; s < S; s++)
					out[((size_t)c * e->P + k) * S + s] = !e->generic        ? ((mask[k] >> s) & 1 ? 1.0 : 0.0)
					                                      : mask[k] > S      ? (double)((e->tipsets_host[mask[k] - S - 1] >> s) & 1)
					                                      : (mask[k] == S || mask[k] == s) ? 1.0 : 0.0;
		return PHYAMD_OK;
	}
	if (!upper && e->core_index[node] < 0)
		return fail(PHYAMD_EINVAL, "node %d is fused into its parent (cherry / cherry+tip) and not stored: phyamd_set_keep_partials(1) first", node);
	const double *src = upper ? e->d_upper + (size_t)e->upper_slot[node] * np : e->d_lower + (size_t)e->core_index[node] * np;
	if (e->generic) {  // planes [C][S][Pp] -> the reference's [C][P][S]
		double *tmp = nullptr;
		const size_t cnt = (size_t)e->C * e->P * e->S;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), cnt * sizeof(double)));
		hipLaunchKernelGGL(k_planes_to_reference, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 4096)), dim3(256), 0, e->stream, e->P, e->Pp, e->S,
		                   e->C, src, tmp);
		hipError_t err = hipMemcpyAsync(out, tmp, cnt * sizeof(double), hipMemcpyDeviceToHost, e->stream);
		if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
		(void)hipFree(tmp);
		if (err != hipSuccess) return fail(PHYAMD_EDEVICE, "get_partials: %s", hipGetErrorString(err));
		return PHYAMD_OK;
	}
	HIP_TRY(hipMemcpyAsync(out, src, sizeof(double) * np, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return PHYAMD_OK;
}

int phyamd_get_node_matrices(phyamd_engine *e, int node, int derivative, double *out) {
	CHECK_ENGINE(e);
	if (!out || node < 0 || node >= e->N) return fail(PHYAMD_EINVAL, "bad node %d or null out", node);
	int rc;
	if ((rc = bind_device(e))) return rc;
	if ((rc = check_ready(e))) return rc;
	if ((rc = update_matrices(e))) return rc;
	const size_t sz = (size_t)e->C * e->S * e->S;
	HIP_TRY(hipMemcpyAsync(out, (derivative ? e->d_dmats : e->d_mats) + (size_t)node * sz, sizeof(double) * sz, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return PHYAMD_OK;
}

int phyamd_is_rescaling(phyamd_engine *e) {
	CHECK_ENGINE(e);
	return e->scaling_on ? 1 : 0;
}

int phyamd_set_keep_partials(phyamd_engine *e, int on) {
	CHECK_ENGINE(e);
	if (on) NOT_TILED(e, "keeping every partial");
	const bool want = on != 0;
	if (want == e->keep_partials) return PHYAMD_OK;
	e->keep_partials = want;
	e->upper_valid = false;
	if (e->have_topology) {
		int rc;
		if ((rc = bind_device(e))) return rc;
		if ((rc = rebuild_schedule(e))) return rc;
	}
	return PHYAMD_OK;
}

int phyamd_set_profiling(phyamd_engine *e, int on) {
	CHECK_ENGINE(e);
	e->profiling = on != 0;
	return PHYAMD_OK;
}

int phyamd_get_profile(phyamd_engine *e, phyamd_profile *out) {
	CHECK_ENGINE(e);
	if (!out) return fail(PHYAMD_EINVAL, "null out");
	finish_profile(e, e->prof_with_upper);  // waits for the last evaluation's events if they are still pending
	e->prof.device_bytes = e->device_bytes;
	e->prof.tiles = e->tiles;
	*out = e->prof;
	return PHYAMD_OK;
}

}  // extern "C"
