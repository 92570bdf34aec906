// phyamd_engine.hip -- MI355X (gfx950) tree-likelihood engine behind include/physher_amd.h.
//
// One translation unit, assembled from:
//   phyamd_device.inc / _level4 / _walk4 / _walk4s / _general / _genwalk / _patterns   device code (kernels)
//   phyamd_shard.inc        state of one engine on one GPU (= one shard of the site patterns)
//   phyamd_schedule.inc     level and tree-walk schedules, device storage
//   phyamd_launch.inc       kernel launches per pass
//   phyamd_eval.inc         one evaluation (incremental updates, lazy rescaling, gradients, pattern tiling)
//   phyamd_shard_api.inc    per-device half of the C ABI
//   phyamd_abi.inc          extern "C" entry points: a handle is a group of 1..n shards on 1..n GPUs
//
// Replaces the CPU hot path of physher's SingleTreeLikelihood (src/phyc/treelikelihood.c and the per-state-count kernel files)
// with hand-written HIP kernels.  The shipped design:
//   * 4 states: a wave keeps 64 site patterns of one rate category (one per lane) and WALKS THE TREE -- the post-order pass and
//     the pre-order pass + branch gradient are each one depth-first walk over a host-built op list (two launches: cut subtrees,
//     then the top of the tree), values handed from op to op in registers, short-lived ones parked in LDS.  Cherries, cherry +
//     tip nodes and nodes above them with 4-6 tips below are never stored: they are rebuilt from 4-bit tip codes wherever needed.
//     Both walks are software pipelines: packed mask words, per-op table blocks (LDS-DMA) and stored operands of op i + 1 are
//     requested while op i computes (phyamd_walk4s.inc; phyamd_walk4.inc = the table-gather form for parameter gradients and
//     rescaled evaluations with more than four categories; phyamd_level4.inc = one launch per tree level for incremental updates,
//     keep_partials and the reference-compatible rescaled gradients);
//   * 20 / 60 / 61 states: P . partial on v_mfma_f64_16x16x4 from LDS matrix images, one launch per tree level (pre-order) or a
//     depth-first walk (20-state post-order), cherries fused (phyamd_general.inc, phyamd_genwalk.inc);
//   * a stored node holds t_n = P_n p_n, its partial carried through its own branch: the pre-order pass reads it where it would
//     repeat that product (20 / 60 / 61 states: always; 4 states: between the two streamed walks -- every other reader of stored
//     partials gets p_n back, ensure_compat_state);
//   * transition matrices are built on the device from the cached eigen system and reach the 4-state kernels through
//     wave-uniform (scalar) loads;
//   * the pre-order pass computes BOTH children's uppers from one read of the parent's upper and fuses the branch-length
//     gradient into the same kernel: the reference's spare_partials array never exists;
//   * reductions are fixed-order (matrix-pipe cross-lane sums, per-block slabs, bisection-ordered segment sums), so results are
//     bitwise reproducible run to run and independent of how many GPUs share the patterns.
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

#define PHYAMD_ABI_VERSION 5

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
	// upper walk, plain kernel: a parked upper whose waiting time holds no other park ("leaf park": two thirds of them in a random
	// tree) can wait in the wave's LDS slot instead of HBM.  bit 0: the parent's upper is read from LDS; bit 1 / 2: the left /
	// right child's upper is parked in LDS.  The HBM slot stays assigned (the rescaling and parameter variants use it).
	int32_t lds_park;
};

#include "phyamd_device.inc"
#include "phyamd_level4.inc"
#include "phyamd_walk4.inc"
#include "phyamd_walk4s.inc"
#include "phyamd_general.inc"
#include "phyamd_genwalk.inc"
#include "phyamd_patterns.inc"

#include "phyamd_shard.inc"

#include "phyamd_schedule.inc"
#include "phyamd_launch.inc"
#include "phyamd_eval.inc"
#include "phyamd_shard_api.inc"

}  // namespace

#include "phyamd_abi.inc"
