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

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "physher_amd.h"

#define PHYAMD_ABI_VERSION 2

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

enum { CH_TIP = 0, CH_CORE = 1, CH_CHERRY = 2, CH_CHERRY_TIP = 3 };  // how a child's lower partial is obtained

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

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------

struct d4 {
	double x, y, z, w;
};

__device__ __forceinline__ d4 load4(const double *p) {
	// 32 contiguous bytes per lane: two 16-byte loads (global_load_dwordx4)
	const double2 a = reinterpret_cast<const double2 *>(p)[0];
	const double2 b = reinterpret_cast<const double2 *>(p)[1];
	return d4{a.x, a.y, b.x, b.y};
}

__device__ __forceinline__ void store4(double *p, const d4 &v) {
	reinterpret_cast<double2 *>(p)[0] = double2{v.x, v.y};
	reinterpret_cast<double2 *>(p)[1] = double2{v.z, v.w};
}

// tip vector from a 4-bit ambiguity mask (one-hot for a known state, 0xF for a gap: datatype.h:26-66)
__device__ __forceinline__ d4 mask4(unsigned m) {
	return d4{(m & 1u) ? 1.0 : 0.0, (m & 2u) ? 1.0 : 0.0, (m & 4u) ? 1.0 : 0.0, (m & 8u) ? 1.0 : 0.0};
}

// Pointer into the AMDGPU constant address space: loads through it with a wave-uniform address become
// s_load_dwordx* into SGPRs (kernel inputs that no kernel of the same launch writes: matrices, Q, pi, weights).
typedef const __attribute__((address_space(4))) double *cptr;
__device__ __forceinline__ cptr as_const(const double *p) { return (cptr)p; }

// y = M v, M row-major 4x4 at a wave-uniform address (scalar loads)
__device__ __forceinline__ d4 matvec4(cptr M, const d4 &v) {
	d4 r;
	r.x = M[0] * v.x + M[1] * v.y + M[2] * v.z + M[3] * v.w;
	r.y = M[4] * v.x + M[5] * v.y + M[6] * v.z + M[7] * v.w;
	r.z = M[8] * v.x + M[9] * v.y + M[10] * v.z + M[11] * v.w;
	r.w = M[12] * v.x + M[13] * v.y + M[14] * v.z + M[15] * v.w;
	// Fence the scheduler: without it every s_load_dwordx16 of a basic block is hoisted to its top (10 matrices = 320
	// SGPRs in the fringe paths) and the overflow is parked in VGPR lanes (v_writelane / v_readlane around every use).
	__builtin_amdgcn_sched_barrier(0);
	return r;
}

__device__ __forceinline__ d4 mul4(const d4 &a, const d4 &b) { return d4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
__device__ __forceinline__ double max4(const d4 &a) { return fmax(fmax(a.x, a.y), fmax(a.z, a.w)); }
__device__ __forceinline__ double dot4(const d4 &a, const d4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// fixed-order sum over the 64 lanes of a wave; every lane ends with the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

// Make a wave-uniform pointer opaque to the optimiser.  The matrices are loop-invariant inside the
// pattern-group loop; without this LLVM hoists all of them out of the loop, runs out of SGPRs and
// parks them in VGPR lanes (v_writelane/v_readlane pairs around every use).  Re-issuing the scalar
// loads per group costs a few s_load_dwordx16 from the scalar cache instead.
__device__ __forceinline__ cptr opaque(cptr p) {
	asm volatile("" : "+s"(p));
	return p;
}

// ------------------------------------------------------------------------------------------------
// M1/M2: P(t) and dP/dt for every (node, category) from the eigen system (substmodel.c:518-557, 695-723)
// ------------------------------------------------------------------------------------------------
// model layout: eval[S] | evec[S*S] | ivec[S*S]
__global__ void k_transition_matrices(int S, int C, int node_count, const double *__restrict__ model, const double *__restrict__ rates,
                                      const double *__restrict__ lengths, const uint8_t *__restrict__ is_explicit, int root,
                                      double *__restrict__ mats, double *__restrict__ dmats) {
	const size_t total = (size_t)node_count * C * S * S;
	const double *eval = model, *evec = model + S, *ivec = model + S + S * S;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
		const int j = idx % S;
		const int i = (idx / S) % S;
		const int c = (idx / ((size_t)S * S)) % C;
		const int n = idx / ((size_t)S * S * C);
		if (n == root || is_explicit[n]) continue;
		const double t = lengths[n] * rates[c];
		double p = 0., dp = 0.;
		for (int k = 0; k < S; k++) {
			const double e = exp(eval[k] * t);
			const double w = ivec[k * S + j] * evec[i * S + k];
			p += w * e;
			dp += w * (eval[k] * e);
		}
		mats[idx] = fabs(p);  // substmodel.c:552
		dmats[idx] = dp;
	}
}

// Tip messages, 4 states: for every (tip, category) the 16 vectors  P(t) . mask  (mask = 4-bit ambiguity code) are
// tabulated once per evaluation, so "transition matrix times tip vector" is one 32-byte gather per (tip, pattern)
// instead of 16 multiply-adds: tiptab[tip][c][mask][i] = sum_j P[i][j] bit_j(mask).  2 MB at 1000 taxa x 4 categories.
__global__ void k_tip_tables(int T, int C, const double *__restrict__ mats, double *__restrict__ tiptab) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= T * C * 64) return;
	const int i = idx & 3, m = (idx >> 2) & 15, tc = idx >> 6;  // tc = tip * C + c
	const double *M = mats + (size_t)tc * 16 + i * 4;
	double s = 0.0;
	if (m & 1) s += M[0];
	if (m & 2) s += M[1];
	if (m & 4) s += M[2];
	if (m & 8) s += M[3];
	tiptab[idx] = s;
}

// G2: dP/dtheta for every (parameter, node, category) from B_theta = U^-1 (dQ/dtheta) U:
//   dP = U ( B o F(t) ) U^-1,  F_ab = (e^{l_a t} - e^{l_b t}) / (l_a - l_b)  or  t e^{l_a t} when l_a == l_b
// (dPdp_with_dQdp, substmodel.c:469-489).  dpm: [NP][N][C][S][S]; root and explicit-matrix nodes get zeros.
__global__ void k_parameter_matrices(int S, int C, int node_count, int np, const double *__restrict__ model, const double *__restrict__ B,
                                     const double *__restrict__ rates, const double *__restrict__ lengths,
                                     const uint8_t *__restrict__ is_explicit, int root, double *__restrict__ dpm) {
	const size_t per = (size_t)node_count * C * S * S, total = per * np;
	const double *eval = model, *evec = model + S, *ivec = model + S + S * S;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
		const int j = idx % S;
		const int i = (idx / S) % S;
		const int c = (idx / ((size_t)S * S)) % C;
		const int n = (idx / ((size_t)S * S * C)) % node_count;
		const int th = idx / per;
		double v = 0.0;
		if (n != root && !is_explicit[n]) {
			const double t = lengths[n] * rates[c];
			const double *Bt = B + (size_t)th * S * S;
			for (int a = 0; a < S; a++) {
				const double ea = exp(eval[a] * t);
				double row = 0.0;
				for (int b = 0; b < S; b++) {
					const double f = eval[a] != eval[b] ? (ea - exp(eval[b] * t)) / (eval[a] - eval[b]) : t * ea;
					row += Bt[a * S + b] * f * ivec[b * S + j];
				}
				v += evec[i * S + a] * row;
			}
		}
		dpm[idx] = v;
	}
}

// tip tables of dP/dtheta, like k_tip_tables: dptab[th][tip][c][mask][i] = sum_j dP_th[tip][c][i][j] bit_j(mask)
__global__ void k_parameter_tip_tables(int T, int N, int C, int np, const double *__restrict__ dpm, double *__restrict__ dptab) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= np * T * C * 64) return;
	const int i = idx & 3, m = (idx >> 2) & 15, tc = (idx >> 6) % (T * C), th = (idx >> 6) / (T * C);
	const double *M = dpm + ((size_t)th * N * C + tc) * 16 + i * 4;
	double s = 0.0;
	if (m & 1) s += M[0];
	if (m & 2) s += M[1];
	if (m & 4) s += M[2];
	if (m & 8) s += M[3];
	dptab[idx] = s;
}

// ------------------------------------------------------------------------------------------------
// 4-state kernels.  Workgroup = (64 lanes = patterns) x (C waves = categories) x (G pattern groups);
// a wave's category is uniform, so its 4x4 matrices live in SGPRs and feed v_fma_f64 directly.
// Cross-category quantities (rescaling max, mixture sums) go through a small LDS exchange.
//
// Fringe fusion: an internal node whose subtree is a cherry (tip, tip) or a cherry plus a tip is never written to
// HBM.  Wherever its partial is needed it is recomputed from 2-3 tip bytes in registers, and in the pre-order pass
// its upper partial is pushed down through it in registers, yielding the gradients of its 2-4 inner branches in the
// same kernel.  Half of the internal nodes of a random tree are of these two shapes, so both passes move about half
// the bytes.  (The unfused schedule of keep_partials runs the same kernels with only tip / stored children.)
// ------------------------------------------------------------------------------------------------
// lower: stored partials, array `core` at lower + core * C*P*4, layout [C][P][4] (the reference's)
// tipmask: [T][P] 4-bit ambiguity masks
// lscale: [cores][P] cumulative log scale factors (SCALE only)

struct Ctx4 {
	const uint8_t *__restrict__ tipmask;
	const double *__restrict__ mats;
	const double *__restrict__ tiptab;
	int P, C, c, k;
	// P_t . tip vector: one gather from the per-(tip, category) table of the 16 possible masks
	__device__ __forceinline__ d4 tipmsg(int t) const {
		const unsigned m = tipmask[(size_t)t * P + k];
		return load4(tiptab + (((size_t)t * C + c) * 16 + m) * 4);
	}
	__device__ __forceinline__ cptr M(int node) const { return opaque(as_const(mats + ((size_t)node * C + c) * 16)); }
};

// message of one child to its parent, P_child . p_child, at (pattern k, category c): a tip (table gather), a stored
// array, or a fringe subtree recomputed in registers
__device__ __forceinline__ d4 child_message(const Ctx4 &x, int kind, int node, int core, int t0, int t1, int t2, int inner,
                                            const double *__restrict__ lower, size_t plane) {
	if (kind == CH_TIP) return x.tipmsg(node);
	if (kind == CH_CORE) return matvec4(x.M(node), load4(lower + ((size_t)core * x.C + x.c) * plane + (size_t)x.k * 4));
	const d4 cherry = mul4(x.tipmsg(t0), x.tipmsg(t1));
	if (kind == CH_CHERRY) return matvec4(x.M(node), cherry);
	return matvec4(x.M(node), mul4(matvec4(x.M(inner), cherry), x.tipmsg(t2)));  // CH_CHERRY_TIP
}
// same, also handing back p_child itself, which the substitution-parameter gradient contracts with dP/dtheta
__device__ __forceinline__ d4 child_message_pre(const Ctx4 &x, int kind, int node, int core, int t0, int t1, int t2, int inner,
                                                const double *__restrict__ lower, size_t plane, d4 &pre) {
	if (kind == CH_TIP) return x.tipmsg(node);
	if (kind == CH_CORE) pre = load4(lower + ((size_t)core * x.C + x.c) * plane + (size_t)x.k * 4);
	else {
		pre = mul4(x.tipmsg(t0), x.tipmsg(t1));                                            // cherry
		if (kind == CH_CHERRY_TIP) pre = mul4(matvec4(x.M(inner), pre), x.tipmsg(t2));    // cherry + tip
	}
	return matvec4(x.M(node), pre);
}

// dynamic LDS: 4 * G*C*64 doubles (two double-buffered exchanges) + G doubles (reduction)
template <int WAVES, bool SCALE, bool ROOT>
__global__ __launch_bounds__(WAVES *WAVE) void k_lower4(const NodeOp *__restrict__ ops, int T, int P, int C,
                                                        const uint8_t *__restrict__ tipmask, double *__restrict__ lower,
                                                        const double *__restrict__ mats, const double *__restrict__ tiptab,
                                                        double *__restrict__ lscale,
                                                        const double *__restrict__ freqs, const double *__restrict__ props,
                                                        const double *__restrict__ weights, double *__restrict__ pattern_lk,
                                                        double *__restrict__ w_over_L, double *__restrict__ lnl_part) {
	extern __shared__ double sh[];
	// blockDim.x == 64: threadIdx.y/z are wave-uniform; readfirstlane tells the compiler so (SGPR addressing)
	const int lane = threadIdx.x, c = __builtin_amdgcn_readfirstlane(threadIdx.y), g = __builtin_amdgcn_readfirstlane(threadIdx.z), G = blockDim.z;
	const NodeOp op = ops[blockIdx.y];
	const size_t plane = (size_t)P * 4;  // one category of one node
	double *dst = lower + ((size_t)op.core_parent * C + c) * plane;
	const int xsz = G * C * WAVE;  // one exchange buffer
	double acc = 0.0;

#pragma unroll 1
	for (int q = 0; q < PPT_LOWER; q++) {
		const int k0 = ((blockIdx.x * PPT_LOWER + q) * G + g) * WAVE + lane;
		const bool valid = k0 < P;
		const Ctx4 x{tipmask, mats, tiptab, P, C, c, valid ? k0 : P - 1};
		const int k = x.k;
		const d4 a = child_message(x, op.kind_left, op.left, op.core_left, op.lt0, op.lt1, op.lt2, op.linner, lower, plane);
		const d4 b = child_message(x, op.kind_right, op.right, op.core_right, op.rt0, op.rt1, op.rt2, op.rinner, lower, plane);
		d4 out = mul4(a, b);
		double sf = 0.0;
		if (SCALE) {  // SingleTreeLikelihood_scalePartials (treelikelihood.c:1790-1836): max over categories and states
			double *xb = sh + (q & 1) * xsz;
			xb[(g * C + c) * WAVE + lane] = max4(out);
			__syncthreads();
			double m = 0.0;
			for (int cc = 0; cc < C; cc++) m = fmax(m, xb[(g * C + cc) * WAVE + lane]);
			if (m < SCALING_THRESHOLD) {
				out = d4{out.x / m, out.y / m, out.z / m, out.w / m};
				sf = log(m);
			}
			if (op.kind_left == CH_CORE) sf += lscale[(size_t)op.core_left * P + k];
			if (op.kind_right == CH_CORE) sf += lscale[(size_t)op.core_right * P + k];
			if (c == 0 && valid) lscale[(size_t)op.core_parent * P + k] = sf;
		}
		if (valid) store4(dst + (size_t)k * 4, out);
		if (ROOT) {  // integrate_partials + node_log_likelihoods + weighted sum (treelikelihood.c:1473-1487)
			double *yb = sh + (2 + (q & 1)) * xsz;
			yb[(g * C + c) * WAVE + lane] = props[c] * (freqs[0] * out.x + freqs[1] * out.y + freqs[2] * out.z + freqs[3] * out.w);
			__syncthreads();
			if (c == 0) {
				double L = 0.0;
				for (int cc = 0; cc < C; cc++) L += yb[(g * C + cc) * WAVE + lane];
				const double lk = log(L) + sf;
				if (valid) {
					const double w = weights[k];
					pattern_lk[k] = lk;
					if (!SCALE) w_over_L[k] = w / L;  // the gradient's w_k / L_k (treelikelihood.c:2879), formed once per pattern
					acc += lk * w;
				}
			}
		}
	}
	if (ROOT) {
		double *red = sh + 4 * xsz;
		const double s = wave_sum(acc);
		if (lane == 0 && c == 0) red[g] = s;
		__syncthreads();
		if (lane == 0 && c == 0 && g == 0) {
			double t = red[0];
			for (int gg = 1; gg < G; gg++) t += red[gg];
			lnl_part[blockIdx.x] = t;
		}
	}
}


// ------------------------------------------------------------------------------------------------
// Tree-walk form of the post-order pass (unscaled evaluations).  Patterns are independent, so instead of one launch
// per tree level a workgroup keeps its 64*G*PPT_WALK patterns (PPT_WALK = 1 or 2, by shard size) and walks ALL core nodes itself, in depth-first
// post-order (larger subtree first).  The op before a node is then always one of its children: that child's partial
// is taken from registers instead of being read back (it is still stored once for the pre-order pass), and the other
// child was written by this very thread a short subtree ago, often still in L2 / Infinity Cache.  One launch, no
// level barriers; the root's integration happens after the loop on the carried root partial.
// dynamic LDS: G*C*64 doubles (root exchange) + G doubles
// ------------------------------------------------------------------------------------------------
template <int WAVES, int PPT_WALK, bool SCALE>
__global__ __launch_bounds__(WAVES *WAVE) void k_lower4_walk(const NodeOp *__restrict__ ops, int nops, int T, int P, int C,
                                                             const uint8_t *__restrict__ tipmask, double *__restrict__ lower,
                                                             const double *__restrict__ mats, const double *__restrict__ tiptab,
                                                             double *__restrict__ lscale,
                                                             const double *__restrict__ freqs, const double *__restrict__ props,
                                                             const double *__restrict__ weights, double *__restrict__ pattern_lk,
                                                             double *__restrict__ w_over_L, double *__restrict__ lnl_part) {
	// dynamic LDS: root exchange G*C*64 + G doubles; SCALE: + two G*C*64 buffers for the per-op maximum over categories
	extern __shared__ double sh[];
	const int lane = threadIdx.x, c = __builtin_amdgcn_readfirstlane(threadIdx.y), g = __builtin_amdgcn_readfirstlane(threadIdx.z), G = blockDim.z;
	const size_t plane = (size_t)P * 4;
	const int xsz = G * C * WAVE;
	int kq[PPT_WALK];
	bool vq[PPT_WALK];
#pragma unroll
	for (int q = 0; q < PPT_WALK; q++) {
		const int k0 = ((blockIdx.x * PPT_WALK + q) * G + g) * WAVE + lane;
		vq[q] = k0 < P;
		kq[q] = vq[q] ? k0 : P - 1;
	}
	d4 carry[PPT_WALK];
	double sfc[PPT_WALK];  // SCALE: cumulative log scale factor of the carried partial
#pragma unroll
	for (int q = 0; q < PPT_WALK; q++) {
		carry[q] = d4{0., 0., 0., 0.};
		sfc[q] = 0.0;
	}
	int flip = 0;
#pragma unroll 1
	for (int i = 0; i < nops; i++) {
		const NodeOp *op = ops + i;  // wave-uniform: scalar loads
		const int cin = op->carry_in;
		double *dst = lower + ((size_t)op->core_parent * C + c) * plane;
#pragma unroll
		for (int q = 0; q < PPT_WALK; q++) {
			const Ctx4 x{tipmask, mats, tiptab, P, C, c, kq[q]};
			const d4 a = cin == 1 ? matvec4(x.M(op->left), carry[q])
			                      : child_message(x, op->kind_left, op->left, op->core_left, op->lt0, op->lt1, op->lt2, op->linner, lower, plane);
			const d4 b = cin == 2 ? matvec4(x.M(op->right), carry[q])
			                      : child_message(x, op->kind_right, op->right, op->core_right, op->rt0, op->rt1, op->rt2, op->rinner, lower, plane);
			d4 out = mul4(a, b);
			if (SCALE) {  // SingleTreeLikelihood_scalePartials (treelikelihood.c:1790-1836): max over categories and states
				double *xb = sh + xsz + G + (flip & 1) * xsz;
				flip++;
				xb[(g * C + c) * WAVE + lane] = max4(out);
				__syncthreads();
				double m = 0.0, sf = 0.0;
				for (int cc = 0; cc < C; cc++) m = fmax(m, xb[(g * C + cc) * WAVE + lane]);
				if (m < SCALING_THRESHOLD) {
					out = d4{out.x / m, out.y / m, out.z / m, out.w / m};
					sf = log(m);
				}
				if (op->kind_left == CH_CORE) sf += cin == 1 ? sfc[q] : lscale[(size_t)op->core_left * P + kq[q]];
				if (op->kind_right == CH_CORE) sf += cin == 2 ? sfc[q] : lscale[(size_t)op->core_right * P + kq[q]];
				if (c == 0 && vq[q]) lscale[(size_t)op->core_parent * P + kq[q]] = sf;
				sfc[q] = sf;
			}
			if (vq[q]) store4(dst + (size_t)kq[q] * 4, out);
			carry[q] = out;
		}
	}
	// the last op is the root: integrate_partials + node_log_likelihoods + weighted sum (treelikelihood.c:1473-1487)
	double acc = 0.0;
#pragma unroll
	for (int q = 0; q < PPT_WALK; q++) {
		const d4 out = carry[q];
		__syncthreads();
		sh[(g * C + c) * WAVE + lane] = props[c] * (freqs[0] * out.x + freqs[1] * out.y + freqs[2] * out.z + freqs[3] * out.w);
		__syncthreads();
		if (c == 0) {
			double L = 0.0;
			for (int cc = 0; cc < C; cc++) L += sh[(g * C + cc) * WAVE + lane];
			const double lk = log(L) + (SCALE ? sfc[q] : 0.0);
			if (vq[q]) {
				const double w = weights[kq[q]];
				pattern_lk[kq[q]] = lk;
				if (!SCALE) w_over_L[kq[q]] = w / L;
				acc += lk * w;
			}
		}
	}
	double *red = sh + xsz;
	const double s = wave_sum(acc);
	if (lane == 0 && c == 0) red[g] = s;
	__syncthreads();
	if (lane == 0 && c == 0 && g == 0) {
		double t = red[0];
		for (int gg = 1; gg < G; gg++) t += red[gg];
		lnl_part[blockIdx.x] = t;
	}
}

// ------------------------------------------------------------------------------------------------
// K7 + K8 fused, 4 states: one level of the pre-order pass
// ------------------------------------------------------------------------------------------------
// For a parent p with children l, r, per (pattern k, category c):
//   a   = P_p u_p                      (parent is the root: a = pi if FOLD else 1)
//   bl  = P_l p_l,  br = P_r p_r
//   u_l = a o br,   u_r = a o bl                                       (treelikelihood.c:2142-2147)
//   den_c  = sum_i f_i a_i bl_i br_i      = L_kc in this branch's (scaled) units
//   num_lc = sum_i f_i u_l,i (Q bl)_i     since (dP/dt) p = Q P p     (treelikelihood.c:2846-2939), f = 1 if FOLD else pi
//   g[l][c] += w_k num_lc / L_k.  Unscaled: w_k / L_k comes from the root kernel.  Rescaled: L_k in this branch's
//   units = sum_c' w_c' den_c' (COMPAT: den_c alone, treelikelihood.c:2851-2870)
// A fringe child continues in registers: its upper is pushed through its cherry (and inner cherry), producing the
// gradients of the 2-4 branches inside it (accumulators e[0..3]).
// upper: slot s at upper + s * C*P*4.  gpart: [(N*C)][nblk] per-block partial sums.
// dynamic LDS: SCALE: 6 * G*C*64 doubles (three double-buffered exchanges); then NACC * waves * 64 doubles (reduction)

constexpr int NACC = 10;  // gradient accumulators per thread: 2 children + 4 + 4 fringe branches
constexpr int WCOL = WAVE + 2;  // tree-walk kernel: LDS column stride (padding spreads the quarter-column readers over the banks)

struct Grad4 {
	cptr Q;
	d4 f;
	double wl;
	double *acc;  // this thread's accumulators in LDS: acc[i * WAVE] (lane-interleaved, conflict-free)
	// substitution-parameter side (PARAMS kernels only): dP/dtheta per (parameter, node, category) and their tip tables
	const double *__restrict__ dpm;    // [NP][N][C][16]
	const double *__restrict__ dptab;  // [NP][T][C][16][4]
	int np, N, T;
	double wc;                         // category proportion: parameter terms are summed over categories
	// acc[i] += w_k / L_k * sum_i f_i u_i (Q b)_i
	__device__ __forceinline__ void add(int i, const d4 &u, const d4 &b) const {
		acc[i * WAVE] += wl * dot4(mul4(f, u), matvec4(opaque(Q), b));
	}
	// acc[NACC + th] += w_c w_k / L_k sum_i f_i u_i (dP_th,node p)_i   (calculate_dlnl_dQ, treelikelihood.c:2472-2580)
	__device__ __forceinline__ void addp_vec(const Ctx4 &x, int node, const d4 &u, const d4 &p) const {
		const d4 fu = mul4(f, u);
		for (int th = 0; th < np; th++) {
			const cptr D = opaque(as_const(dpm + (((size_t)th * N + node) * x.C + x.c) * 16));
			acc[(NACC + th) * WAVE] += wc * wl * dot4(fu, matvec4(D, p));
		}
	}
	__device__ __forceinline__ void addp_tip(const Ctx4 &x, int tip, const d4 &u) const {
		const d4 fu = mul4(f, u);
		const unsigned m = x.tipmask[(size_t)tip * x.P + x.k];
		for (int th = 0; th < np; th++)
			acc[(NACC + th) * WAVE] += wc * wl * dot4(fu, load4(dptab + ((((size_t)th * T + tip) * x.C + x.c) * 16 + m) * 4));
	}
};

// push upper `u` of a fringe child (node `node`) down to its inner branches; accumulators base..base+3:
//   CH_CHERRY     : +0 -> t0, +1 -> t1
//   CH_CHERRY_TIP : +0 -> t0, +1 -> t1 (inside the inner cherry), +2 -> inner, +3 -> t2
// Ordered so that few vectors are live at once (the kernel is register-limited).
// tree-walk kernel: one pattern per thread, so every branch term is written exactly once per op (no read-modify-write);
// Q is diag(pi) Q (or Q when the frequencies are folded into the uppers), so there is no separate state weight.
// PARAMS: every gradient site also feeds the substitution-parameter gradient, accumulated in the eigen basis:
//   d lnL / d theta = sum_ab B_theta,ab G_ab,   G_ab = sum_sites w_c F_ab(t r_c) (w_k / L_k) a_a b_b,
//   a = U^T (pi o u),  b = U^-1 p,  B_theta = U^-1 dQ_theta U,  F as in dPdp_with_dQdp (substmodel.c:469-489)
// -- 16 accumulators per thread whatever the number of parameters, two extra mat-vecs and 16 multiply-adds per site
// (the reference walks the tree once per parameter; the level kernels contract with one dP/dtheta per parameter).
struct ParamCtx {
	cptr UTpi, Uinv;     // (diag(pi) U)^T and U^-1, 4x4 row-major
	const double *utab;  // [16][4]: U^-1 . tip mask
	const double *Fw;    // [N][C][16]: w_c F_ab(t_n r_c)
};
// SCALE (rescaled evaluations): the branch term is w_k (num / D_k) with D_k the site likelihood in the op's scaled units --
// quotient first, num and D can both be denormal -- and the parameter sums weight with wl = w_k / D_k.
template <bool PARAMS, bool SCALE>
struct GradWT {
	cptr Q;
	double wl;
	double *col;  // this thread's NACC slots in LDS, stride WCOL
	ParamCtx pc;
	double *G;    // [16], registers of the kernel
	double w, d;  // SCALE only
	__device__ __forceinline__ void add(int i, const d4 &u, const d4 &b) const {
		const double num = dot4(u, matvec4(opaque(Q), b));
		col[i * WCOL] = SCALE ? w * (num / d) : wl * num;
	}
	__device__ __forceinline__ void accumulate(const Ctx4 &x, int node, const d4 &a, const d4 &b) const {
		const cptr F = opaque(as_const(pc.Fw + ((size_t)node * x.C + x.c) * 16));
		const double t0 = wl * a.x, t1 = wl * a.y, t2 = wl * a.z, t3 = wl * a.w;
		G[0] += (F[0] * t0) * b.x;
		G[1] += (F[1] * t0) * b.y;
		G[2] += (F[2] * t0) * b.z;
		G[3] += (F[3] * t0) * b.w;
		G[4] += (F[4] * t1) * b.x;
		G[5] += (F[5] * t1) * b.y;
		G[6] += (F[6] * t1) * b.z;
		G[7] += (F[7] * t1) * b.w;
		G[8] += (F[8] * t2) * b.x;
		G[9] += (F[9] * t2) * b.y;
		G[10] += (F[10] * t2) * b.z;
		G[11] += (F[11] * t2) * b.w;
		G[12] += (F[12] * t3) * b.x;
		G[13] += (F[13] * t3) * b.y;
		G[14] += (F[14] * t3) * b.z;
		G[15] += (F[15] * t3) * b.w;
		__builtin_amdgcn_sched_barrier(0);
	}
	// a branch with a computed lower partial p (stored node, cherry product, ...) / a tip branch with mask m
	__device__ __forceinline__ void site_vec(const Ctx4 &x, int node, const d4 &u, const d4 &p) const {
		if (PARAMS) accumulate(x, node, matvec4(opaque(pc.UTpi), u), matvec4(opaque(pc.Uinv), p));
	}
	__device__ __forceinline__ void site_tip(const Ctx4 &x, int tip, const d4 &u, unsigned m) const {
		if (PARAMS) accumulate(x, tip, matvec4(opaque(pc.UTpi), u), load4(pc.utab + m * 4));
	}
};

// rescaled evaluations: the branch term is w_k num / D_k with D_k the site likelihood in the op's scaled units; the quotient
// is formed first (num and D can both be denormal when a category has underflowed)
struct GradS {
	cptr Q;
	d4 f;
	double w, d;
	double *acc;
	__device__ __forceinline__ void add(int i, const d4 &u, const d4 &b) const { acc[i * WAVE] += w * (dot4(mul4(f, u), matvec4(opaque(Q), b)) / d); }
	__device__ __forceinline__ void addp_vec(const Ctx4 &, int, const d4 &, const d4 &) const {}
	__device__ __forceinline__ void addp_tip(const Ctx4 &, int, const d4 &) const {}
};
// only the substitution-parameter terms of a Grad4 (used next to GradS)
struct GradPOnly {
	const Grad4 &g;
	__device__ __forceinline__ void add(int, const d4 &, const d4 &) const {}
	__device__ __forceinline__ void addp_vec(const Ctx4 &x, int node, const d4 &u, const d4 &p) const { g.addp_vec(x, node, u, p); }
	__device__ __forceinline__ void addp_tip(const Ctx4 &x, int tip, const d4 &u) const { g.addp_tip(x, tip, u); }
};

template <bool PARAMS, typename GradT>
__device__ __forceinline__ void descend_fringe(const Ctx4 &x, const GradT &gr, int base, int kind, int node, int t0, int t1, int t2, int inner,
                                               const d4 &u) {
	const d4 b0 = x.tipmsg(t0), b1 = x.tipmsg(t1);
	d4 a2 = matvec4(x.M(node), u);
	if (kind == CH_CHERRY_TIP) {
		const d4 pn = mul4(b0, b1);
		const d4 bn = matvec4(x.M(inner), pn);
		const d4 b2 = x.tipmsg(t2);
		const d4 un = mul4(a2, b2);
		gr.add(base + 2, un, bn);
		gr.add(base + 3, mul4(a2, bn), b2);
		if (PARAMS) {
			gr.addp_vec(x, inner, un, pn);
			gr.addp_tip(x, t2, mul4(a2, bn));
		}
		a2 = matvec4(x.M(inner), un);  // now the upper message entering the inner cherry
	}
	gr.add(base + 0, mul4(a2, b1), b0);
	gr.add(base + 1, mul4(a2, b0), b1);
	if (PARAMS) {
		gr.addp_tip(x, t0, mul4(a2, b1));
		gr.addp_tip(x, t1, mul4(a2, b0));
	}
}

template <int WAVES, bool SCALE, bool FOLD, bool COMPAT, bool PARAMS>
__global__ __launch_bounds__(WAVES *WAVE, (WAVES == 4 && !PARAMS) ? PHYAMD_UPPER_MIN_WAVES : 1) void k_upper4(const NodeOp *__restrict__ ops, int T, int P, int C,
                                                        const uint8_t *__restrict__ tipmask, const double *__restrict__ lower,
                                                        double *__restrict__ upper, const double *__restrict__ mats,
                                                        const double *__restrict__ tiptab, const double *__restrict__ Q,
                                                        const double *__restrict__ freqs,
                                                        const double *__restrict__ props, const double *__restrict__ weights,
                                                        const double *__restrict__ w_over_L, double *__restrict__ gpart, int nblk,
                                                        const double *__restrict__ dpm, const double *__restrict__ dptab, int np, int N,
                                                        double *__restrict__ ppart, int op_base, int op_total) {
	extern __shared__ double sh[];
	// blockDim.x == 64: threadIdx.y/z are wave-uniform; readfirstlane tells the compiler so (SGPR addressing)
	const int lane = threadIdx.x, c = __builtin_amdgcn_readfirstlane(threadIdx.y), g = __builtin_amdgcn_readfirstlane(threadIdx.z), G = blockDim.z;
	const NodeOp op = ops[blockIdx.y];
	const size_t plane = (size_t)P * 4;
	const bool proot = op.upper_slot_parent < 0;
	const int nacc = NACC + (PARAMS ? np : 0);  // accumulator columns per thread
	const double *up = proot ? nullptr : upper + ((size_t)op.upper_slot_parent * C + c) * plane;
	double *ul_dst = op.upper_slot_left < 0 ? nullptr : upper + ((size_t)op.upper_slot_left * C + c) * plane;
	double *ur_dst = op.upper_slot_right < 0 ? nullptr : upper + ((size_t)op.upper_slot_right * C + c) * plane;
	const d4 pi = d4{freqs[0], freqs[1], freqs[2], freqs[3]};
	const d4 one = d4{1., 1., 1., 1.};
	const int xsz = G * C * WAVE;
	// gradient accumulators live in LDS (one column per thread), not in registers: the kernel is VGPR-limited
	const int wv = g * C + c, nw = G * C;
	double *red = sh + (SCALE ? 6 * xsz : 0);
	Grad4 gr{as_const(Q), FOLD ? one : pi, 0.0, red + (size_t)wv * nacc * WAVE + lane, dpm, dptab, np, N, T, props[c]};
	for (int i = 0; i < nacc; i++) gr.acc[i * WAVE] = 0.0;

#pragma unroll 1
	for (int q = 0; q < PPT_UPPER; q++) {
		const int k0 = ((blockIdx.x * PPT_UPPER + q) * G + g) * WAVE + lane;
		const bool valid = k0 < P;
		const Ctx4 x{tipmask, mats, tiptab, P, C, c, valid ? k0 : P - 1};
		const int k = x.k;
		d4 prel = one, prer = one;  // the children's own partials (PARAMS only)
		const d4 bl = PARAMS ? child_message_pre(x, op.kind_left, op.left, op.core_left, op.lt0, op.lt1, op.lt2, op.linner, lower, plane, prel)
		                     : child_message(x, op.kind_left, op.left, op.core_left, op.lt0, op.lt1, op.lt2, op.linner, lower, plane);
		const d4 br = PARAMS ? child_message_pre(x, op.kind_right, op.right, op.core_right, op.rt0, op.rt1, op.rt2, op.rinner, lower, plane, prer)
		                     : child_message(x, op.kind_right, op.right, op.core_right, op.rt0, op.rt1, op.rt2, op.rinner, lower, plane);
		const d4 a = proot ? (FOLD ? pi : one) : matvec4(x.M(op.parent), load4(up + (size_t)k * 4));
		d4 ul = mul4(a, br), ur = mul4(a, bl);
		if (!SCALE) {
			// unscaled: divide by the site likelihood formed at the root, like the reference (treelikelihood.c:2879);
			// no cross-category exchange, no barrier, no division in this kernel
			gr.wl = valid ? w_over_L[k] : 0.0;
			gr.add(0, ul, bl);
			gr.add(1, ur, br);
			if (ul_dst && valid) store4(ul_dst + (size_t)k * 4, ul);
			if (ur_dst && valid) store4(ur_dst + (size_t)k * 4, ur);
			if (PARAMS) {
				if (op.kind_left == CH_TIP) gr.addp_tip(x, op.left, ul);
				else gr.addp_vec(x, op.left, ul, prel);
				if (op.kind_right == CH_TIP) gr.addp_tip(x, op.right, ur);
				else gr.addp_vec(x, op.right, ur, prer);
			}
			if (op.kind_left >= CH_CHERRY) descend_fringe<PARAMS>(x, gr, 2, op.kind_left, op.left, op.lt0, op.lt1, op.lt2, op.linner, ul);
			if (op.kind_right >= CH_CHERRY) descend_fringe<PARAMS>(x, gr, 6, op.kind_right, op.right, op.rt0, op.rt1, op.rt2, op.rinner, ur);
			continue;
		} else {
			// rescaled: L_k underflows by construction, so the mixture likelihood is re-formed
			// in this branch's scaled units from all categories' den (exchange through LDS); the scale factors cancel in num / D
			const double den = dot4(mul4(gr.f, a), mul4(bl, br));
			const double numl = dot4(mul4(gr.f, ul), matvec4(opaque(gr.Q), bl));
			const double numr = dot4(mul4(gr.f, ur), matvec4(opaque(gr.Q), br));
			double *xb = sh + (q & 1) * 3 * xsz;
			const int xi = (g * C + c) * WAVE + lane;
			xb[xi] = props[c] * den;
			xb[xsz + xi] = max4(ul);
			xb[2 * xsz + xi] = max4(ur);
			__syncthreads();
			double D = 0.0, ml = 0.0, mr = 0.0;
			for (int cc = 0; cc < C; cc++) {
				D += xb[(g * C + cc) * WAVE + lane];
				ml = fmax(ml, xb[xsz + (g * C + cc) * WAVE + lane]);
				mr = fmax(mr, xb[2 * xsz + (g * C + cc) * WAVE + lane]);
			}
			// num / L first: with COMPAT both can be denormal (a category that has underflowed) and 1 / den alone overflows
			const double w = valid ? weights[k] : 0.0, d = COMPAT ? den : D;
			gr.acc[0] += w * (numl / d);
			gr.acc[WAVE] += w * (numr / d);
			if (PARAMS) {  // mixture numerator over the mixture likelihood in this branch's units (treelikelihood.c:2545-2556)
				gr.wl = w / D;
				if (op.kind_left == CH_TIP) gr.addp_tip(x, op.left, ul);
				else gr.addp_vec(x, op.left, ul, prel);
				if (op.kind_right == CH_TIP) gr.addp_tip(x, op.right, ur);
				else gr.addp_vec(x, op.right, ur, prer);
			}
			// Fringe children (fused schedules run rescaled too): a cherry or cherry + tip never reaches the rescaling threshold
			// (products of two or three transition probabilities), so its likelihood is in this op's units and shares D
			if (op.kind_left >= CH_CHERRY || op.kind_right >= CH_CHERRY) {
				const GradS gs{gr.Q, gr.f, w, d, gr.acc};
				if (op.kind_left >= CH_CHERRY) descend_fringe<false>(x, gs, 2, op.kind_left, op.left, op.lt0, op.lt1, op.lt2, op.linner, ul);
				if (op.kind_right >= CH_CHERRY) descend_fringe<false>(x, gs, 6, op.kind_right, op.right, op.rt0, op.rt1, op.rt2, op.rinner, ur);
				if (PARAMS) {
					const GradPOnly gp{gr};  // gr.wl = w / D from above
					if (op.kind_left >= CH_CHERRY) descend_fringe<true>(x, gp, 2, op.kind_left, op.left, op.lt0, op.lt1, op.lt2, op.linner, ul);
					if (op.kind_right >= CH_CHERRY) descend_fringe<true>(x, gp, 6, op.kind_right, op.right, op.rt0, op.rt1, op.rt2, op.rinner, ur);
				}
			}
			// uppers are rescaled like lowers (treelikelihood.c:1414, 1795-1796)
			if (ml < SCALING_THRESHOLD) ul = d4{ul.x / ml, ul.y / ml, ul.z / ml, ul.w / ml};
			if (mr < SCALING_THRESHOLD) ur = d4{ur.x / mr, ur.y / mr, ur.z / mr, ur.w / mr};
		}
		if (ul_dst && valid) store4(ul_dst + (size_t)k * 4, ul);
		if (ur_dst && valid) store4(ur_dst + (size_t)k * 4, ur);
	}
	// Fixed-order reduction over the workgroup's patterns: the accumulators sit in LDS as [wave][acc][lane];
	// one lane per (wave, accumulator) adds the 64 entries in lane order; wave 0 of each category adds the pattern groups.
	__syncthreads();
	if (lane < nacc) {
		const double *src = red + ((size_t)wv * nacc + lane) * WAVE;
		double s0 = src[0], s1 = src[16], s2 = src[32], s3 = src[48];  // four chains of 16, then a fixed combine
		for (int j = 1; j < 16; j++) {
			s0 += src[j];
			s1 += src[16 + j];
			s2 += src[32 + j];
			s3 += src[48 + j];
		}
		const double s = (s0 + s1) + (s2 + s3);
		red[(size_t)nw * nacc * WAVE + wv * nacc + lane] = s;
	}
	__syncthreads();
	const double *tot = red + (size_t)nw * nacc * WAVE;
	if (g == 0 && lane < NACC) {
		double s = tot[c * nacc + lane];
		for (int gg = 1; gg < G; gg++) s += tot[(gg * C + c) * nacc + lane];
		// accumulator -> gradient row (node id); -1 = unused for this op
		const int kl = op.kind_left, kr = op.kind_right;
		int node = -1;
		switch (lane) {
			case 0: node = op.left; break;
			case 1: node = op.right; break;
			case 2: node = kl >= CH_CHERRY ? op.lt0 : -1; break;
			case 3: node = kl >= CH_CHERRY ? op.lt1 : -1; break;
			case 4: node = kl == CH_CHERRY_TIP ? op.linner : -1; break;
			case 5: node = kl == CH_CHERRY_TIP ? op.lt2 : -1; break;
			case 6: node = kr >= CH_CHERRY ? op.rt0 : -1; break;
			case 7: node = kr >= CH_CHERRY ? op.rt1 : -1; break;
			case 8: node = kr == CH_CHERRY_TIP ? op.rinner : -1; break;
			case 9: node = kr == CH_CHERRY_TIP ? op.rt2 : -1; break;
		}
		if (node >= 0) gpart[((size_t)node * C + c) * nblk + blockIdx.x] = s;
	}
	if (PARAMS && wv == 0 && lane < np) {  // parameter terms: sum over all waves (categories and pattern groups) in a fixed order
		double s = 0.0;
		for (int w = 0; w < nw; w++) s += tot[w * nacc + NACC + lane];
		ppart[((size_t)lane * op_total + op_base + blockIdx.y) * nblk + blockIdx.x] = s;
	}
}

// Variants for the pre-order tree walk: the mask bytes of an op's (up to six) tips are all requested at the top of the op,
// together with the parent's upper, so the op pays one memory round trip for them instead of one per child.
__device__ __forceinline__ d4 tip_gather(const Ctx4 &x, int t, unsigned m) { return load4(x.tiptab + (((size_t)t * x.C + x.c) * 16 + m) * 4); }
// `pre` receives the child's own partial (what the parameter gradient contracts): untouched for tips
__device__ __forceinline__ d4 child_message_m(const Ctx4 &x, int kind, int node, int core, int t0, int t1, int t2, int inner,
                                              const d4 &pcore, unsigned m0, unsigned m1, unsigned m2, d4 &pre) {
	if (kind == CH_TIP) return tip_gather(x, node, m0);
	if (kind == CH_CORE) pre = pcore;
	else {
		pre = mul4(tip_gather(x, t0, m0), tip_gather(x, t1, m1));                                    // cherry
		if (kind == CH_CHERRY_TIP) pre = mul4(matvec4(x.M(inner), pre), tip_gather(x, t2, m2));   // cherry + tip
	}
	return matvec4(x.M(node), pre);
}
template <typename GradT>
__device__ __forceinline__ void descend_fringe_m(const Ctx4 &x, const GradT &gr, int base, int kind, int node, int t0, int t1, int t2, int inner,
                                                 const d4 &u, unsigned m0, unsigned m1, unsigned m2) {
	const d4 b0 = tip_gather(x, t0, m0), b1 = tip_gather(x, t1, m1);
	d4 a2 = matvec4(x.M(node), u);
	if (kind == CH_CHERRY_TIP) {
		const d4 pn = mul4(b0, b1);
		const d4 bn = matvec4(x.M(inner), pn);
		const d4 b2 = tip_gather(x, t2, m2);
		const d4 un = mul4(a2, b2);
		gr.add(base + 2, un, bn);
		gr.add(base + 3, mul4(a2, bn), b2);
		gr.site_vec(x, inner, un, pn);
		gr.site_tip(x, t2, mul4(a2, bn), m2);
		a2 = matvec4(x.M(inner), un);  // now the upper message entering the inner cherry
	}
	gr.add(base + 0, mul4(a2, b1), b0);
	gr.add(base + 1, mul4(a2, b0), b1);
	gr.site_tip(x, t0, mul4(a2, b1), m0);
	gr.site_tip(x, t1, mul4(a2, b0), m1);
}

// sum 16 per-lane values over the 64 lanes of a wave in 17 exchange steps (instead of 16 x 6): after the xor-32 step a
// lane keeps only half of the values, after xor-16 a quarter, ...  Returns, in every lane, the wave total of value
// index ((lane >> 2) & 15) with bits taken as (bit5, bit4, bit3, bit2) -> (8, 4, 2, 1).  Fixed order: deterministic.
__device__ __forceinline__ double wave_sum16(const double (&v)[16], int lane) {
	double a8[8], a4[4], a2[2];
	const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
	for (int i = 0; i < 8; i++) a8[i] = (h5 ? v[i + 8] : v[i]) + __shfl_xor(h5 ? v[i] : v[i + 8], 32, 64);
#pragma unroll
	for (int i = 0; i < 4; i++) a4[i] = (h4 ? a8[i + 4] : a8[i]) + __shfl_xor(h4 ? a8[i] : a8[i + 4], 16, 64);
#pragma unroll
	for (int i = 0; i < 2; i++) a2[i] = (h3 ? a4[i + 2] : a4[i]) + __shfl_xor(h3 ? a4[i] : a4[i + 2], 8, 64);
	double a1 = (h2 ? a2[1] : a2[0]) + __shfl_xor(h2 ? a2[0] : a2[1], 4, 64);
	a1 += __shfl_xor(a1, 2, 64);
	a1 += __shfl_xor(a1, 1, 64);
	return a1;
}

// ------------------------------------------------------------------------------------------------
// Tree-walk form of the pre-order pass + branch gradient (unscaled evaluations), the counterpart of k_lower4_walk: a
// workgroup keeps its patterns (ONE per thread) and visits every core node in depth-first pre-order, smaller core subtree
// first.  The upper partial of the child visited next never leaves registers (no store, no load); the other core child's
// upper is parked in one of ~log2(core nodes) recycled slots and read back by the same thread after the small subtree.
// The branch terms of an op go to LDS columns (written once per op) and are reduced over the wave by wave_sum16, then
// written to the gradient slab: gpart[(node * C + c) * nblk + blockIdx.x * G + g], nblk = gridDim.x * G.
// Register-lean on purpose (<= 96 VGPRs, 5 waves per SIMD): the walk is bound by dependent latency chains (tip byte ->
// table gather -> mat-vec, scalar matrix loads), so resident waves matter more than per-op amortisation; a form with 4
// patterns per thread, carried uppers in LDS and accumulators in registers (126 VGPRs, 4 waves) measured 14 % slower.
// dynamic LDS: [waves][NACC][WCOL] doubles
// ------------------------------------------------------------------------------------------------
#ifndef PHYAMD_WALK_UPPER_MIN_WAVES
#define PHYAMD_WALK_UPPER_MIN_WAVES 5
#endif
// PARAMS: dynamic LDS holds 16 columns per wave (the eigen-basis sums are reduced once, after the walk); pbuf = [UTpi(16) |
// Uinv(16) | utab(64)], Fw as in ParamCtx, gacc [16][nblk] receives the per-wave sums.
// SCALE / COMPAT: rescaled evaluations (see k_upper4): one LDS exchange per op gives every category's wave the mixture
// denominator D_k and the maxima of the two new uppers; dynamic LDS grows by 6 * waves * 64 doubles (double-buffered).
template <int WAVES, bool FOLD, bool PARAMS, bool SCALE, bool COMPAT>
__global__ __launch_bounds__(WAVES *WAVE, WAVES == 4 ? (PARAMS ? 3 : (SCALE ? 4 : PHYAMD_WALK_UPPER_MIN_WAVES)) : 1) void k_upper4_walk(const NodeOp *__restrict__ ops, int nops, int T, int P, int C,
                                                              const uint8_t *__restrict__ tipmask, const double *__restrict__ lower,
                                                              double *__restrict__ upper, const double *__restrict__ mats,
                                                              const double *__restrict__ tiptab, const double *__restrict__ Q,
                                                              const double *__restrict__ freqs, const double *__restrict__ w_over_L,
                                                              double *__restrict__ gpart, int nblk, const double *__restrict__ pbuf,
                                                              const double *__restrict__ Fw, double *__restrict__ gacc,
                                                              const double *__restrict__ props, const double *__restrict__ weights) {
	extern __shared__ double sh[];
	constexpr int NCOL = PARAMS ? 16 : NACC;
	const int lane = threadIdx.x, c = __builtin_amdgcn_readfirstlane(threadIdx.y), g = __builtin_amdgcn_readfirstlane(threadIdx.z), G = blockDim.z;
	const size_t plane = (size_t)P * 4;
	const d4 pi = d4{freqs[0], freqs[1], freqs[2], freqs[3]};
	const d4 one = d4{1., 1., 1., 1.};
	const int wv = g * C + c;
	double *wave_cols = sh + (size_t)wv * NCOL * WCOL;  // this wave's columns of WCOL (= 64 + padding) doubles
	double *col = wave_cols + lane;                     // this thread's slot in each column, stride WCOL
	// reduction role of this lane: lanes 0..4*NACC-1 each add a quarter (16 entries) of one column
	const int my = lane >> 2, seg = lane & 3;
	const size_t slab = (size_t)blockIdx.x * G + g;
	const int k0 = (blockIdx.x * G + g) * WAVE + lane;
	const bool valid = k0 < P;
	const int k = valid ? k0 : P - 1;
	const Ctx4 x{tipmask, mats, tiptab, P, C, c, k};
	const double wl = SCALE ? 0.0 : (valid ? w_over_L[k] : 0.0);
	const double wk = SCALE ? (valid ? weights[k] : 0.0) : 0.0;
	const int xsz = G * C * WAVE;
	double *xbase = sh + (size_t)G * C * NCOL * WCOL;  // SCALE: exchange buffers behind the columns
	d4 carry = one;
	double Gab[16] = {0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0.};
	const ParamCtx pc{as_const(pbuf), as_const(pbuf + 16), pbuf + 32, Fw};
#pragma unroll 1
	for (int i = 0; i < nops; i++) {
		const NodeOp *op = ops + i;  // wave-uniform: scalar loads
		const bool proot = i == 0;   // pre-order: the root comes first
		const int cin = op->carry_in, cout = op->carry_out;
		const int kl = op->kind_left, kr = op->kind_right;
		GradWT<PARAMS, SCALE> gr{as_const(Q), wl, col, pc, Gab, 0.0, 1.0};  // Q is diag(pi) Q unless FOLD
		// every tip mask byte of the op up front: all in flight together
		unsigned ml0 = 0, ml1 = 0, ml2 = 0, mr0 = 0, mr1 = 0, mr2 = 0;
		if (kl == CH_TIP) ml0 = tipmask[(size_t)op->left * P + k];
		else if (kl >= CH_CHERRY) {
			ml0 = tipmask[(size_t)op->lt0 * P + k];
			ml1 = tipmask[(size_t)op->lt1 * P + k];
			if (kl == CH_CHERRY_TIP) ml2 = tipmask[(size_t)op->lt2 * P + k];
		}
		if (kr == CH_TIP) mr0 = tipmask[(size_t)op->right * P + k];
		else if (kr >= CH_CHERRY) {
			mr0 = tipmask[(size_t)op->rt0 * P + k];
			mr1 = tipmask[(size_t)op->rt1 * P + k];
			if (kr == CH_CHERRY_TIP) mr2 = tipmask[(size_t)op->rt2 * P + k];
		}
		d4 uin = carry;  // the parent's upper: carried in registers, or parked by an earlier op of this thread
		if (!proot && !cin) uin = load4(upper + ((size_t)op->upper_slot_parent * C + c) * plane + (size_t)k * 4);
		d4 pl = one, pr = one;  // stored children
		if (kl == CH_CORE) pl = load4(lower + ((size_t)op->core_left * C + c) * plane + (size_t)k * 4);
		if (kr == CH_CORE) pr = load4(lower + ((size_t)op->core_right * C + c) * plane + (size_t)k * 4);
		d4 prel = one, prer = one;  // the children's own partials (the parameter gradient contracts them)
		const d4 bl = child_message_m(x, kl, op->left, op->core_left, op->lt0, op->lt1, op->lt2, op->linner, pl, ml0, ml1, ml2, prel);
		const d4 br = child_message_m(x, kr, op->right, op->core_right, op->rt0, op->rt1, op->rt2, op->rinner, pr, mr0, mr1, mr2, prer);
		d4 a;
		if (proot) a = FOLD ? pi : one;
		else a = matvec4(x.M(op->parent), uin);
		d4 ul = mul4(a, br), ur = mul4(a, bl);
		double ml = 1.0, mr = 1.0;
		if (SCALE) {
			// the mixture likelihood in this op's scaled units from all categories (L_k itself underflows by construction)
			const double den = dot4(FOLD ? a : mul4(pi, a), mul4(bl, br));
			double *xb = xbase + (size_t)(i & 1) * 3 * xsz;
			const int xi = wv * WAVE + lane;
			xb[xi] = props[c] * den;
			xb[xsz + xi] = max4(ul);
			xb[2 * xsz + xi] = max4(ur);
			__syncthreads();
			double D = 0.0;
			ml = mr = 0.0;
			for (int cc = 0; cc < C; cc++) {
				D += xb[(g * C + cc) * WAVE + lane];
				ml = fmax(ml, xb[xsz + (g * C + cc) * WAVE + lane]);
				mr = fmax(mr, xb[2 * xsz + (g * C + cc) * WAVE + lane]);
			}
			gr.w = wk;
			gr.d = COMPAT ? den : D;
			gr.wl = wk / D;
		}
		gr.add(0, ul, bl);
		gr.add(1, ur, br);
		if (PARAMS) {
			if (kl == CH_TIP) gr.site_tip(x, op->left, ul, ml0);
			else gr.site_vec(x, op->left, ul, prel);
			if (kr == CH_TIP) gr.site_tip(x, op->right, ur, mr0);
			else gr.site_vec(x, op->right, ur, prer);
		}
		// fringe children continue in registers with the un-rescaled uppers (they share this op's units and denominator)
		if (kl >= CH_CHERRY) descend_fringe_m(x, gr, 2, kl, op->left, op->lt0, op->lt1, op->lt2, op->linner, ul, ml0, ml1, ml2);
		if (kr >= CH_CHERRY) descend_fringe_m(x, gr, 6, kr, op->right, op->rt0, op->rt1, op->rt2, op->rinner, ur, mr0, mr1, mr2);
		if (SCALE) {  // uppers are rescaled like lowers (treelikelihood.c:1414, 1795-1796)
			if (ml < SCALING_THRESHOLD) ul = d4{ul.x / ml, ul.y / ml, ul.z / ml, ul.w / ml};
			if (mr < SCALING_THRESHOLD) ur = d4{ur.x / mr, ur.y / mr, ur.z / mr, ur.w / mr};
		}
		if (op->upper_slot_left >= 0 && valid) store4(upper + ((size_t)op->upper_slot_left * C + c) * plane + (size_t)k * 4, ul);
		if (op->upper_slot_right >= 0 && valid) store4(upper + ((size_t)op->upper_slot_right * C + c) * plane + (size_t)k * 4, ur);
		carry = cout == 1 ? ul : ur;
		// Fixed-order sum of each column over the wave's 64 patterns: four lanes per column add 16 entries each in order,
		// then (s0 + s1) + (s2 + s3).  Slots an op did not write hold stale values; their rows are never stored.
		__builtin_amdgcn_wave_barrier();
		double tot = 0.0;
		if (my < NACC) {
			const double *src = wave_cols + my * WCOL + seg * 16;
			tot = src[0];
#pragma unroll
			for (int j = 1; j < 16; j++) tot += src[j];
		}
		tot += __shfl_xor(tot, 1, 64);
		tot += __shfl_xor(tot, 2, 64);
		__builtin_amdgcn_wave_barrier();
		if (seg == 0 && my < NACC) {
			int node = -1;  // accumulator -> gradient row (node id); -1 = unused for this op
			switch (my) {
				case 0: node = op->left; break;
				case 1: node = op->right; break;
				case 2: node = kl >= CH_CHERRY ? op->lt0 : -1; break;
				case 3: node = kl >= CH_CHERRY ? op->lt1 : -1; break;
				case 4: node = kl == CH_CHERRY_TIP ? op->linner : -1; break;
				case 5: node = kl == CH_CHERRY_TIP ? op->lt2 : -1; break;
				case 6: node = kr >= CH_CHERRY ? op->rt0 : -1; break;
				case 7: node = kr >= CH_CHERRY ? op->rt1 : -1; break;
				case 8: node = kr == CH_CHERRY_TIP ? op->rinner : -1; break;
				case 9: node = kr == CH_CHERRY_TIP ? op->rt2 : -1; break;
			}
			if (node >= 0) gpart[((size_t)node * C + c) * nblk + slab] = tot;
		}
	}
	if (PARAMS) {  // the 16 eigen-basis sums of this wave (all categories add into the same G_ab: one slab entry per wave)
#pragma unroll
		for (int a = 0; a < 16; a++) col[a * WCOL] = Gab[a];
		__builtin_amdgcn_wave_barrier();
		const double *src = wave_cols + my * WCOL + seg * 16;
		double tot = src[0];
#pragma unroll
		for (int j = 1; j < 16; j++) tot += src[j];
		tot += __shfl_xor(tot, 1, 64);
		tot += __shfl_xor(tot, 2, 64);
		if (seg == 0) gacc[(size_t)my * nblk * C + slab * C + c] = tot;
	}
}

// G2 tables of the tree-walk kernel: Fw[n][c][a*4+b] = w_c F_ab(t_n r_c), F_ab = (e^{l_a t} - e^{l_b t}) / (l_a - l_b) or
// t e^{l_a t} (dPdp_with_dQdp, substmodel.c:469-489); root and explicit-matrix nodes get zeros.  4 states.
__global__ void k_eigen_weights(int C, int node_count, const double *__restrict__ model, const double *__restrict__ rates,
                                const double *__restrict__ props, const double *__restrict__ lengths, const uint8_t *__restrict__ is_explicit, int root,
                                double *__restrict__ Fw) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= node_count * C * 16) return;
	const int b = idx & 3, a = (idx >> 2) & 3, c = (idx >> 4) % C, n = (idx >> 4) / C;
	double v = 0.0;
	if (n != root && !is_explicit[n]) {
		const double t = lengths[n] * rates[c], la = model[a], lb = model[b], ea = exp(la * t);
		v = props[c] * (la != lb ? (ea - exp(lb * t)) / (la - lb) : t * ea);
	}
	Fw[idx] = v;
}

// d lnL / d theta = sum_ab B_theta,ab G_ab (B = U^-1 dQ U, [np][16])
__global__ void k_contract_parameters(int np, const double *__restrict__ B, const double *__restrict__ Gsum, double *__restrict__ out) {
	const int th = blockIdx.x * blockDim.x + threadIdx.x;
	if (th >= np) return;
	double s = 0.0;
	for (int ab = 0; ab < 16; ab++) s += B[(size_t)th * 16 + ab] * Gsum[ab];
	out[th] = s;
}

// fixed-order reduction of per-block slabs: one wave per row. out[row_offset + row] = sum_b part[row][b]
__global__ __launch_bounds__(64) void k_reduce_rows(const double *__restrict__ part, int nblk, const uint8_t *__restrict__ row_valid,
                                                   double *__restrict__ out) {
	const int row = blockIdx.x;
	double s = 0.0;
	if (row_valid == nullptr || row_valid[row]) {
		for (int b = threadIdx.x; b < nblk; b += 64) s += part[(size_t)row * nblk + b];
		s = wave_sum(s);
	}
	if (threadIdx.x == 0) out[row] = s;
}

// sum_k (w_k / L_k) sum_i pi_i ( p_root[0][k][i] - mean_{c >= 1} p_root[c][k][i] ): the d lnL / d(proportion of the
// invariant class) term that needs the root partials (treelikelihood.c:2943-3008).  root: stored root partial;
// cat_stride / pat_stride / state_stride describe its layout ([C][P][4] or planes [C][S][Pp]).
__global__ __launch_bounds__(256) void k_root_invariant_term(int P, int S, int C, const double *__restrict__ root, size_t cat_stride, size_t pat_stride,
                                                            size_t state_stride, const double *__restrict__ freqs,
                                                            const double *__restrict__ w_over_L, double *__restrict__ part) {
	__shared__ double red[4];
	const int k = blockIdx.x * 256 + threadIdx.x;
	double acc = 0.0;
	if (k < P) {
		double s = 0.0;
		for (int i = 0; i < S; i++) {
			const double *p = root + (size_t)k * pat_stride + (size_t)i * state_stride;
			double others = 0.0;
			for (int c = 1; c < C; c++) others += p[(size_t)c * cat_stride];
			s += freqs[i] * (p[0] - others / (C - 1));
		}
		acc = s * w_over_L[k];
	}
	const double t = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
	__syncthreads();
	if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The root term of d lnL / d pi_f (calculate_dlnl_dQ, treelikelihood.c:2370-2401): for every state f
//   sum_k w_k  ( sum_c w_c p_root[c][k][f] )  /  ( sum_i pi_i sum_c w_c p_root[c][k][i] ).
// Scale factors are per pattern, so the same expression serves rescaled evaluations.  part: [S][gridDim.x]
__global__ __launch_bounds__(256) void k_root_frequency_term(int P, int S, int C, const double *__restrict__ root, size_t cat_stride, size_t pat_stride,
                                                            size_t state_stride, const double *__restrict__ freqs, const double *__restrict__ props,
                                                            const double *__restrict__ weights, double *__restrict__ part) {
	__shared__ double red[4];
	const int k = blockIdx.x * 256 + threadIdx.x;
	double like = 0.0;
	if (k < P)
		for (int i = 0; i < S; i++) {
			const double *p = root + (size_t)k * pat_stride + (size_t)i * state_stride;
			double m = 0.0;
			for (int c = 0; c < C; c++) m += props[c] * p[(size_t)c * cat_stride];
			like += freqs[i] * m;
		}
	const double wl = k < P ? weights[k] / like : 0.0;
	for (int f = 0; f < S; f++) {
		double m = 0.0;
		if (k < P) {
			const double *p = root + (size_t)k * pat_stride + (size_t)f * state_stride;
			for (int c = 0; c < C; c++) m += props[c] * p[(size_t)c * cat_stride];
		}
		const double t = wave_sum(m * wl);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
		__syncthreads();
		if (threadIdx.x == 0) part[(size_t)f * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
	}
}

#include "phyamd_general.inc"

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
	bool walk_params_on = true;  // parameter gradients through the tree walk (PHYAMD_WALK_PARAMS = 0: level kernels)
	bool walk_lower_on = true, walk_upper_on = true;  // A/B switches (PHYAMD_WALK_LOWER / PHYAMD_WALK_UPPER = 0)
	bool walk_enabled = true, walking = false;  // tree-walk kernels (4 states, unscaled, not keep_partials)
	std::vector<NodeOp> walk_lower_ops, walk_upper_ops;  // depth-first op orders
	NodeOp *d_walk_lower_ops = nullptr, *d_walk_upper_ops = nullptr;
	int walk_upper_slots = 0;
	double *d_Lc = nullptr;  // [C][P] per-category site likelihoods at the root (generic)
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
	size_t lower_alloc_cores = 0;

	// device memory
	uint8_t *d_tipmask = nullptr;
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
	double *d_Fw = nullptr;          // tree-walk G2: [N][C][16] w_c F_ab(t_n r_c)
	double *d_gacc = nullptr;        // tree-walk G2: [16][slabs * C] per-wave eigen-basis sums, then [16] totals
	double *d_rf_part = nullptr;     // [S][blocks] partial sums of k_root_frequency_term, then [S]
	double *d_gen_scratch = nullptr; // rescaled S != 4 path: per-level maxima / numerators / denominators
	size_t gen_scratch_alloc = 0;
	size_t np_alloc = 0, np_alloc_B = 0, ppart_alloc = 0;
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
		}
	}
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
			if (depth[n] != d || kind[n] != CH_CORE) continue;
			NodeOp op = make_op(n);
			op.upper_slot_parent = n == e->root ? -1 : e->upper_slot[n];
			for (int side = 0; side < 2; side++) {
				const int ch = side ? e->right[n] : e->left[n];
				if (kind[ch] != CH_CORE && !e->keep_partials) continue;  // uppers of tips and fused nodes stay in registers
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
		std::vector<int> csize(N, 0);
		for (int i = N - 1; i >= 0; i--) {
			const int n = order[i];
			if (kind[n] == CH_CORE) csize[n] = 1 + csize[e->left[n]] + csize[e->right[n]];
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
			const bool lc = kind[l] == CH_CORE, rc2 = kind[r] == CH_CORE;
			int first = -1, second = -1;
			if (lc && rc2) {
				first = csize[l] <= csize[r] ? l : r;
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
	const size_t need = (size_t)std::max(1, e->core_count);
	if (e->d_lower && e->lower_alloc_cores >= need) return PHYAMD_OK;
	dev_free(e, &e->d_lower, e->lower_alloc_cores * node_partial_doubles(e));
	dev_free(e, &e->d_lscale, e->lower_alloc_cores * (size_t)e->P);
	e->lower_alloc_cores = 0;
	int rc = dev_alloc(e, &e->d_lower, need * node_partial_doubles(e));
	if (rc) return rc;
	e->lower_alloc_cores = need;
	return PHYAMD_OK;
}

int upload_schedule(phyamd_engine *e) {
	// op tables are a few KB: allocated once at the maximum size (N - T ops each)
	int rc;
	if (!e->d_lower_ops && (rc = dev_alloc(e, &e->d_lower_ops, (size_t)e->N))) return rc;
	if (!e->d_upper_ops && (rc = dev_alloc(e, &e->d_upper_ops, (size_t)e->N))) return rc;
	HIP_TRY(hipMemcpyAsync(e->d_lower_ops, e->lower_ops.data(), e->lower_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipMemcpyAsync(e->d_upper_ops, e->upper_ops.data(), e->upper_ops.size() * sizeof(NodeOp), hipMemcpyHostToDevice, e->stream));
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

int update_matrices(phyamd_engine *e) {
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

template <int WAVES, bool SCALE>
int launch_lower_walk(phyamd_engine *e) {
	const size_t lds = sizeof(double) * ((size_t)e->G * e->C * WAVE * (SCALE ? 3 : 1) + e->G);
	if (e->ppt_walk_lower == 1)
		hipLaunchKernelGGL((k_lower4_walk<WAVES, 1, SCALE>), dim3(e->nblk_walk), block_dims(e), lds, e->stream, e->d_walk_lower_ops, (int)e->walk_lower_ops.size(),
		                   e->T, e->P, e->C, e->d_tipmask, e->d_lower, e->d_mats, e->d_tiptab, e->d_lscale, e->d_freqs, e->d_props, e->d_weights, e->d_plk, e->d_wl,
		                   e->d_lnl_part);
	else
		hipLaunchKernelGGL((k_lower4_walk<WAVES, 2, SCALE>), dim3(e->nblk_walk), block_dims(e), lds, e->stream, e->d_walk_lower_ops, (int)e->walk_lower_ops.size(),
		                   e->T, e->P, e->C, e->d_tipmask, e->d_lower, e->d_mats, e->d_tiptab, e->d_lscale, e->d_freqs, e->d_props, e->d_weights, e->d_plk, e->d_wl,
		                   e->d_lnl_part);
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
	if (!e->d_Fw && (rc = dev_alloc(e, &e->d_Fw, (size_t)e->N * e->C * 16))) return rc;
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
	const int nf = e->N * e->C * 16;
	hipLaunchKernelGGL(k_eigen_weights, dim3((nf + 255) / 256), dim3(256), 0, e->stream, e->C, e->N, e->d_model, e->d_rates, e->d_props, e->d_lengths, e->d_explicit,
	                   e->root, e->d_Fw);
	hipLaunchKernelGGL((k_upper4_walk<WAVES, false, true, SCALE, false>), dim3(e->nblk_walk_upper), block_dims(e), lds, e->stream, e->d_walk_upper_ops, ops, e->T,
	                   e->P, e->C, e->d_tipmask, e->d_lower, e->d_upper, e->d_mats, e->d_tiptab, e->d_Qpi, e->d_freqs, e->d_wl, e->d_gpart, nb, e->d_pbuf, e->d_Fw,
	                   e->d_gacc, e->d_props, e->d_weights);
	double *gsum = e->d_gacc + (size_t)16 * nb * e->C;
	hipLaunchKernelGGL(k_reduce_rows, dim3(16), dim3(64), 0, e->stream, e->d_gacc, nb * e->C, (const uint8_t *)nullptr, gsum);
	hipLaunchKernelGGL(k_contract_parameters, dim3(1), dim3(64), 0, e->stream, np, e->d_Bw, gsum, e->d_result + 1 + (size_t)e->N * e->C);
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
	for (int lv = 0; lv < levels; lv++) {
		const int off = level_off[lv], cnt = level_off[lv + 1] - off;
		if (cnt == 0) continue;
		dim3 grid(e->nblk, cnt, e->C);
		const bool is_root = lv == levels - 1;
		if (is_root)
			hipLaunchKernelGGL((k_lower_gen<RT, KT, true, SCALE>), grid, dim3(GenGeo<RT>::WAVES * 64), lds, e->stream, e->act_lower_ops + off, e->T, e->P, e->Pp, e->S, e->C,
			                   e->d_tipmask, e->d_lower, e->d_mats, e->d_freqs, e->d_props, e->d_Lc, e->d_gen_scratch);
		else
			hipLaunchKernelGGL((k_lower_gen<RT, KT, false, SCALE>), grid, dim3(GenGeo<RT>::WAVES * 64), lds, e->stream, e->act_lower_ops + off, e->T, e->P, e->Pp, e->S, e->C,
			                   e->d_tipmask, e->d_lower, e->d_mats, e->d_freqs, e->d_props, e->d_Lc, e->d_gen_scratch);
		if (SCALE)
			hipLaunchKernelGGL(k_scale_gen, dim3(pblocks, cnt), dim3(256), 0, e->stream, e->act_lower_ops + off, e->P, e->Pp, e->S, e->C, e->d_lower, e->d_gen_scratch,
			                   e->d_lscale, is_root ? e->d_Lc : (double *)nullptr);
	}
	const double *lscale_root = SCALE ? e->d_lscale + (size_t)e->core_index[e->root] * e->P : nullptr;
	hipLaunchKernelGGL(k_root_finish, dim3(e->nblk_root), dim3(256), 0, e->stream, e->P, e->C, e->d_Lc, e->d_weights, lscale_root, e->d_plk, e->d_wl, e->d_lnl_part);
	HIP_TRY(hipGetLastError());
	e->prof.lower_launches = levels;
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
		                   e->d_tipmask, e->d_lower, e->d_upper, e->d_mats, e->d_Q, e->d_freqs, e->d_wl, e->d_gpart, e->nblk, nd, mxu);
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
	if (e->lower_valid && !e->all_dirty) {
		// only single branch lengths changed since the stored partials were computed: recompute the core nodes on the paths
		// from those branches to the root, in level order (update_nodes[] semantics, treelikelihood.c:73-114, 1645-1734)
		if (e->changed.empty()) {  // nothing changed: d_result[0] still holds lnL
			record(e, 2);
			e->prof.lower_launches = 0;
			e->prof_pending = e->profiling;
			e->prof_with_upper = false;
			return PHYAMD_OK;
		}
		std::vector<uint8_t> dirty(e->N, 0);
		for (int n : e->changed)
			for (int a = e->parent[n]; a >= 0 && !dirty[a]; a = e->parent[a])
				if (e->core_index[a] >= 0) dirty[a] = 1;  // fused fringe nodes are recomputed inside their first stored ancestor
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

int run_gradient(phyamd_engine *e, int flags, bool with_params = false) {
	int rc;
	if (with_params) {
		if (e->generic) return fail(PHYAMD_EUNSUPPORTED, "substitution-parameter gradients are built for 4-state models only");
		if (e->np < 1) return fail(PHYAMD_EINVAL, "phyamd_set_rate_matrix_derivatives has not been called");
		if (!e->have_eigen) return fail(PHYAMD_EINVAL, "substitution-parameter gradients need the eigen system (phyamd_set_eigen)");
		if (flags & PHYAMD_GRAD_FOLD_ROOT_FREQS)
			return fail(PHYAMD_EINVAL, "PHYAMD_GRAD_FOLD_ROOT_FREQS cannot be combined with parameter gradients (the reference clears include_root_freqs, treelikelihood.c:291-305)");
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
	if (with_params && !(e->walking && e->walk_upper_on && e->walk_params_on && !((flags & PHYAMD_GRAD_COMPAT_SCALED) && e->scaling_on)))
		e->level_upper_needed = true;
	if ((rc = ensure_upper_storage(e))) return rc;
	if (!e->have_Q) return fail(PHYAMD_EINVAL, "the gradient needs the rate matrix: phyamd_set_eigen or phyamd_set_rate_matrix");
	e->grad_blocks = e->nblk;
	// (the compat flag changes only the branch terms; parameter sums always use the mixture denominator: level kernels then)
	const bool walk_params = with_params && e->walking && e->walk_upper_on && e->walk_params_on && !((flags & PHYAMD_GRAD_COMPAT_SCALED) && e->scaling_on);
	if (walk_params) {
		const int waves = e->C * e->G;
		if (e->scaling_on)
			rc = waves <= 4 ? launch_upper_walk_params<4, true>(e) : waves <= 8 ? launch_upper_walk_params<8, true>(e) : launch_upper_walk_params<16, true>(e);
		else
			rc = waves <= 4 ? launch_upper_walk_params<4, false>(e) : waves <= 8 ? launch_upper_walk_params<8, false>(e) : launch_upper_walk_params<16, false>(e);
		if (rc) return rc;
	} else if (with_params) {
		if ((rc = update_parameter_matrices(e))) return rc;
		if ((rc = launch_upper_params(e, flags))) return rc;
	} else if ((rc = launch_upper(e, flags)))
		return rc;
	record(e, 3);
	hipLaunchKernelGGL(k_reduce_rows, dim3(e->N * e->C), dim3(64), 0, e->stream, e->d_gpart, e->grad_blocks, e->d_row_valid, e->d_result + 1);
	if (walk_params) {
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
	e->P = cfg->pattern_count;
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
	if (cfg->stream) e->stream = (hipStream_t)cfg->stream;
	else {
		hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
		if (err != hipSuccess) return bail(fail(PHYAMD_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(err)));
		e->own_stream = true;
	}
	e->scaling_on = cfg->rescale == PHYAMD_RESCALE_ALWAYS;
	if (const char *env = std::getenv("PHYAMD_FUSE")) e->fusion_enabled = std::atoi(env) != 0;  // A/B switch for the fringe fusion
	if (e->C > MAX_WAVES) {
		delete e;
		return fail(PHYAMD_EUNSUPPORTED, "category_count %d exceeds %d (one wave per category)", cfg->category_count, MAX_WAVES);
	}
	e->G = std::max(1, 4 / e->C);  // at least 4 waves per workgroup
	e->nblk = (e->P + WAVE * e->G * PPT_UPPER - 1) / (WAVE * e->G * PPT_UPPER);        // pre-order kernel / gradient slabs
	e->nblk_lower = (e->P + WAVE * e->G * PPT_LOWER - 1) / (WAVE * e->G * PPT_LOWER);  // post-order kernel / lnL slab
	{
		// Tree-walk geometry.  The pre-order walk always takes one pattern per thread (see k_upper4_walk); the post-order walk
		// is write-bound and prefers one pattern per thread once the shard fills the chip, two below that (measured on
		// 125k..1M-pattern shards).  PHYAMD_PPT_WALK_LOWER overrides (A/B runs).
		const int groups = (e->P + WAVE * e->G - 1) / (WAVE * e->G);  // workgroups at one pattern per thread
		e->ppt_walk_lower = groups >= 3000 ? 1 : 2;
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
	const size_t np = node_partial_doubles(e);
	const size_t msz = (size_t)e->N * e->C * e->S * e->S;
	if (cfg->max_device_bytes > 0) {
		const double need = 8.0 * (0.5 * (double)(e->N - e->T) * np + 2.0 * np) + (double)e->T * e->P;
		if (need > (double)cfg->max_device_bytes)
			return bail(fail(PHYAMD_ENOMEM, "engine needs >= %.3g bytes, max_device_bytes is %lld (pattern tiling is not built in this revision)", need,
			                 (long long)cfg->max_device_bytes));
	}
	int rc;
	if ((rc = dev_alloc(e, &e->d_tipmask, (size_t)e->T * e->P))) return bail(rc);
	// d_lower is sized by the schedule (stored "core" nodes only): ensure_lower_storage
	if ((rc = dev_alloc(e, &e->d_mats, msz))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_dmats, msz))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_model, (size_t)e->S + 2 * e->S * e->S))) return bail(rc);
	if ((rc = dev_alloc(e, &e->d_Q, (size_t)e->S * e->S))) return bail(rc);
	if (!e->generic && (rc = dev_alloc(e, &e->d_tiptab, (size_t)e->T * e->C * 64))) return bail(rc);
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
	for (void *p : {(void *)e->d_Bw, (void *)e->d_pbuf, (void *)e->d_Fw, (void *)e->d_gacc, (void *)e->d_gen_scratch, (void *)e->d_rf_part, (void *)e->d_B, (void *)e->d_dpm, (void *)e->d_dptab, (void *)e->d_ppart, (void *)e->d_tipmask, (void *)e->d_lower, (void *)e->d_upper, (void *)e->d_mats, (void *)e->d_dmats, (void *)e->d_model, (void *)e->d_Q, (void *)e->d_Lc, (void *)e->d_inv_part, (void *)e->d_tiptab,
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
	std::vector<uint8_t> mask(e->P);
	if (e->generic)
		for (int k = 0; k < e->P; k++) mask[k] = states[k] < e->S ? states[k] : (uint8_t)e->S;  // raw codes; S = unknown
	else
		for (int k = 0; k < e->P; k++) mask[k] = states[k] < 4 ? (uint8_t)(1u << states[k]) : (uint8_t)0xF;  // code >= S: unknown (treelikelihood4.c:946-988)
	HIP_TRY(hipMemcpyAsync(e->d_tipmask + (size_t)tip * e->P, mask.data(), e->P, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->tip_set[tip] = 1;
	e->all_dirty = true;
	return PHYAMD_OK;
}

int phyamd_set_tip_partials(phyamd_engine *e, int tip, const double *partials) {
	CHECK_ENGINE(e);
	if (tip < 0 || tip >= e->T || !partials) return fail(PHYAMD_EINVAL, "bad tip %d or null partials", tip);
	int rc;
	if ((rc = bind_device(e))) return rc;
	std::vector<uint8_t> mask(e->P);
	if (e->generic) {  // one-hot or all-ones vectors only (what datatype.c:212-240 produces without ambiguity tables)
		const int S = e->S;
		for (int k = 0; k < e->P; k++) {
			int ones = 0, last = -1;
			for (int s = 0; s < S; s++) {
				const double v = partials[(size_t)k * S + s];
				if (v == 1.0) ones++, last = s;
				else if (v != 0.0) return fail(PHYAMD_EUNSUPPORTED, "tip %d pattern %d: only 0/1 tip partials are built", tip, k);
			}
			if (ones == 1) mask[k] = (uint8_t)last;
			else if (ones == S) mask[k] = (uint8_t)S;
			else return fail(PHYAMD_EUNSUPPORTED, "tip %d pattern %d: partial ambiguity sets are only built for 4 states", tip, k);
		}
		HIP_TRY(hipMemcpyAsync(e->d_tipmask + (size_t)tip * e->P, mask.data(), e->P, hipMemcpyHostToDevice, e->stream));
		HIP_TRY(hipStreamSynchronize(e->stream));
		e->tip_set[tip] = 1;
		e->all_dirty = true;
		return PHYAMD_OK;
	}
	for (int k = 0; k < e->P; k++) {
		unsigned m = 0;
		for (int s = 0; s < 4; s++) {
			const double v = partials[(size_t)k * 4 + s];
			if (v == 1.0) m |= 1u << s;
			else if (v != 0.0)
				return fail(PHYAMD_EUNSUPPORTED, "tip %d pattern %d: tip partials other than 0/1 ambiguity masks are not built in this revision", tip, k);
		}
		mask[k] = (uint8_t)m;
	}
	HIP_TRY(hipMemcpyAsync(e->d_tipmask + (size_t)tip * e->P, mask.data(), e->P, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->tip_set[tip] = 1;
	e->all_dirty = true;
	return PHYAMD_OK;
}

int phyamd_set_pattern_weights(phyamd_engine *e, const double *weights) {
	CHECK_ENGINE(e);
	if (!weights) return fail(PHYAMD_EINVAL, "null weights");
	int rc;
	if ((rc = bind_device(e))) return rc;
	HIP_TRY(hipMemcpyAsync(e->d_weights, weights, sizeof(double) * e->P, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->have_weights = true;
	e->all_dirty = true;
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
	return PHYAMD_OK;
}

int phyamd_log_likelihood(phyamd_engine *e, double *lnl) {
	CHECK_ENGINE(e);
	if (!lnl) return fail(PHYAMD_EINVAL, "null lnl");
	int rc;
	if ((rc = run_lower(e, true))) return rc;
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_result, sizeof(double), hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	finish_profile(e, false);
	*lnl = e->h_result[0];
	return PHYAMD_OK;
}

int phyamd_gradient_device(phyamd_engine *e, int flags, double *device_out) {
	CHECK_ENGINE(e);
	if (!device_out) return fail(PHYAMD_EINVAL, "null device_out");
	int rc;
	if ((rc = run_gradient(e, flags))) return rc;
	HIP_TRY(hipMemcpyAsync(device_out, e->d_result, sizeof(double) * ((size_t)1 + e->N * e->C), hipMemcpyDeviceToDevice, e->stream));
	return PHYAMD_OK;
}

int phyamd_gradient(phyamd_engine *e, int flags, double *lnl, double *cat_gradient) {
	CHECK_ENGINE(e);
	if (!cat_gradient) return fail(PHYAMD_EINVAL, "null cat_gradient");
	int rc;
	if ((rc = run_gradient(e, flags))) return rc;
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
	if (count > 0 && e->generic) return fail(PHYAMD_EUNSUPPORTED, "substitution-parameter gradients are built for 4-state models only");
	e->np = count;
	e->dQ_host.assign(dQ, dQ + (size_t)count * e->S * e->S);
	e->params_dirty = true;
	return PHYAMD_OK;
}

int phyamd_parameter_gradient(phyamd_engine *e, int flags, double *lnl, double *cat_gradient, double *parameter_gradient) {
	CHECK_ENGINE(e);
	if (!parameter_gradient) return fail(PHYAMD_EINVAL, "null parameter_gradient");
	int rc;
	if ((rc = run_gradient(e, flags, true))) return rc;
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
	if ((rc = run_gradient(e, flags, true))) return rc;
	HIP_TRY(hipMemcpyAsync(device_out, e->d_result, sizeof(double) * ((size_t)1 + e->N * e->C + e->np + e->S), hipMemcpyDeviceToDevice, e->stream));
	return PHYAMD_OK;
}

int phyamd_root_frequency_term(phyamd_engine *e, double *out) {
	CHECK_ENGINE(e);
	if (!out) return fail(PHYAMD_EINVAL, "null out");
	int rc;
	if ((rc = bind_device(e))) return rc;
	if ((rc = check_ready(e))) return rc;
	if (e->core_index.empty() || e->core_index[e->root] < 0 || !e->d_lower) return fail(PHYAMD_EINVAL, "no evaluation has been run yet");
	if ((rc = launch_root_frequency_term(e, nullptr))) return rc;
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_rf_part + (size_t)((e->P + 255) / 256) * e->S, sizeof(double) * e->S, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	std::memcpy(out, e->h_result, sizeof(double) * e->S);
	return PHYAMD_OK;
}

int phyamd_root_invariant_term(phyamd_engine *e, double *out) {
	CHECK_ENGINE(e);
	if (!out) return fail(PHYAMD_EINVAL, "null out");
	if (e->C < 2) return fail(PHYAMD_EINVAL, "the invariant-class term needs at least two categories");
	if (e->scaling_on) return fail(PHYAMD_EUNSUPPORTED, "the invariant-class term is not built for rescaled evaluations");
	int rc;
	if ((rc = bind_device(e))) return rc;
	if ((rc = check_ready(e))) return rc;
	const int nb = (e->P + 255) / 256;
	if (!e->d_inv_part && (rc = dev_alloc(e, &e->d_inv_part, (size_t)nb + 1))) return rc;
	const double *root = e->d_lower + (size_t)e->core_index[e->root] * node_partial_doubles(e);
	const size_t cat_stride = e->generic ? (size_t)e->S * e->Pp : (size_t)e->P * e->S;
	const size_t pat_stride = e->generic ? 1 : (size_t)e->S, state_stride = e->generic ? (size_t)e->Pp : 1;
	hipLaunchKernelGGL(k_root_invariant_term, dim3(nb), dim3(256), 0, e->stream, e->P, e->S, e->C, root, cat_stride, pat_stride, state_stride, e->d_freqs,
	                   e->d_wl, e->d_inv_part);
	hipLaunchKernelGGL(k_reduce_rows, dim3(1), dim3(64), 0, e->stream, e->d_inv_part, nb, (const uint8_t *)nullptr, e->d_inv_part + nb);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(e->h_result, e->d_inv_part + nb, sizeof(double), hipMemcpyDeviceToHost, e->stream));
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
	HIP_TRY(hipMemcpyAsync(out, e->d_plk, sizeof(double) * e->P, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return PHYAMD_OK;
}

int phyamd_get_partials(phyamd_engine *e, int node, int upper, double *out) {
	CHECK_ENGINE(e);
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
					out[((size_t)c * e->P + k) * S + s] = e->generic ? ((mask[k] >= S || mask[k] == s) ? 1.0 : 0.0) : ((mask[k] >> s) & 1 ? 1.0 : 0.0);
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
	e->prof.tiles = 1;
	*out = e->prof;
	return PHYAMD_OK;
}

}  // extern "C"
