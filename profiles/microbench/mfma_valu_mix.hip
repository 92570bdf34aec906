// Microbenchmark: does the fp64 matrix pipe run BESIDE the vector ALU on gfx950?  (measurement aid for DESIGN.md; not product code)
// One "op" of the pre-order walk is ~6 4x4 mat-vecs + Hadamard products, dot products and ~FILL integer instructions.
//   L  lane = pattern, a vector = 4 registers, a mat-vec = 16 v_fma_f64 with the matrix in SGPRs          (k_upper4_stream today)
//   T  lane = (state, pattern slot), a vector = 4 registers of 16 patterns each, a mat-vec = 4 v_mfma_f64_4x4x4 (matrix = 1 VGPR)
// Both do the same arithmetic per 64 patterns.  Prints ns per op per wave and SIMD-cycles per op at 2.4 GHz for 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct V4 { double r[4]; };
__device__ __forceinline__ V4 had(const V4 &a, const V4 &b) { return V4{{a.r[0] * b.r[0], a.r[1] * b.r[1], a.r[2] * b.r[2], a.r[3] * b.r[3]}}; }
__device__ __forceinline__ double dot(const V4 &a, const V4 &b) { return fma(a.r[3], b.r[3], fma(a.r[2], b.r[2], fma(a.r[1], b.r[1], a.r[0] * b.r[0]))); }

// T layout: o.r[q] = M . v.r[q] on the matrix pipe
__device__ __forceinline__ V4 mx(double A, const V4 &v) {
	V4 o;
#pragma unroll
	for (int q = 0; q < 4; q++) o.r[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, v.r[q], 0.0, 0, 0, 0);
	return o;
}
// L layout: matrix in SGPRs (kernel arguments are uniform)
struct M16 { double m[16]; };
__device__ __forceinline__ V4 mv(const M16 &M, const V4 &v) {
	V4 o;
#pragma unroll
	for (int i = 0; i < 4; i++) o.r[i] = fma(M.m[4 * i + 3], v.r[3], fma(M.m[4 * i + 2], v.r[2], fma(M.m[4 * i + 1], v.r[1], M.m[4 * i] * v.r[0])));
	return o;
}

template <int FILL>
__device__ __forceinline__ unsigned filler(unsigned w, unsigned lane) {
#pragma unroll
	for (int j = 0; j < FILL; j++) w = ((w >> 3) ^ (w << 5)) + lane;  // 3 integer instructions each
	return w;
}

template <int FILL>
__global__ __launch_bounds__(256, 4) void k_T(double *out, int iters, const double *mats) {
	const unsigned lane = threadIdx.x & 63;
	const double Ap = mats[lane], Al = mats[64 + lane], Ar = mats[128 + lane], AQ = mats[192 + lane], Ad = mats[256 + lane];
	const double sel = (lane & 3) == 0 ? 1.0 : 0.0;
	V4 u{{1.0, 0.9, 0.8, 0.7}}, pA{{0.5, 0.6, 0.7, 0.8}}, pB{{0.9, 0.8, 0.7, 0.6}};
	double acc = 0.0;
	unsigned w = lane * 2654435761u;
#pragma unroll 1
	for (int it = 0; it < iters; it++) {
		const V4 a = mx(Ap, u);
		const V4 bl = mx(Al, pA);
		const V4 br = mx(Ar, had(pB, mx(Ad, pA)));
		const V4 ul = had(a, br), ur = had(a, bl);
		const double sL = dot(ul, mx(AQ, bl)), sR = dot(ur, mx(AQ, br));
		acc = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, sL, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, sR, acc, 0, 0, 0);
		w = filler<FILL>(w, lane);
		const double nz = 1.0 + (double)(w & 1u);  // keeps the filler alive and the values bounded
		u = V4{{ul.r[0] * nz + 0.25, ul.r[1] + 0.25, ul.r[2] + 0.25, ul.r[3] + 0.25}};
		pA = V4{{pA.r[1], pA.r[2], pA.r[3], pA.r[0]}};
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc + u.r[0] + w;
}

template <int FILL>
__global__ __launch_bounds__(256, 4) void k_L(double *out, int iters, M16 Mp, M16 Ml, M16 Mr, M16 MQ, M16 Md) {
	const unsigned lane = threadIdx.x & 63;
	const double sel = (lane & 3) == 0 ? 1.0 : 0.0;
	V4 u{{1.0, 0.9, 0.8, 0.7}}, pA{{0.5, 0.6, 0.7, 0.8}}, pB{{0.9, 0.8, 0.7, 0.6}};
	double acc = 0.0;
	unsigned w = lane * 2654435761u;
#pragma unroll 1
	for (int it = 0; it < iters; it++) {
		const V4 a = mv(Mp, u);
		const V4 bl = mv(Ml, pA);
		const V4 br = mv(Mr, had(pB, mv(Md, pA)));
		const V4 ul = had(a, br), ur = had(a, bl);
		const double sL = dot(ul, mv(MQ, bl)), sR = dot(ur, mv(MQ, br));
		acc = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, sL, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, sR, acc, 0, 0, 0);
		w = filler<FILL>(w, lane);
		const double nz = 1.0 + (double)(w & 1u);
		u = V4{{ul.r[0] * nz + 0.25, ul.r[1] + 0.25, ul.r[2] + 0.25, ul.r[3] + 0.25}};
		pA = V4{{pA.r[1], pA.r[2], pA.r[3], pA.r[0]}};
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc + u.r[0] + w;
}

// pure issue-rate probes: N independent MFMAs (no VALU) / dependent chain of 4-groups
__global__ __launch_bounds__(256, 4) void k_mfma_only(double *out, int iters, const double *mats) {
	const unsigned lane = threadIdx.x & 63;
	const double A = mats[lane];
	double v0 = 1.0 + lane, v1 = 2.0, v2 = 3.0, v3 = 4.0;
#pragma unroll 1
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 6; r++) {
			v0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, v0, 0.0, 0, 0, 0);
			v1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, v1, 0.0, 0, 0, 0);
			v2 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, v2, 0.0, 0, 0, 0);
			v3 = __builtin_amdgcn_mfma_f64_4x4x4f64(A, v3, 0.0, 0, 0, 0);
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
}

int main(int argc, char **argv) {
	const int iters = 20000;
	double hm[320];
	for (int i = 0; i < 320; i++) {
		const int l = i % 64, r = l & 3, c = l >> 4;  // lane l holds M[l & 3][l >> 4]; near-identity stochastic matrix
		hm[i] = (r == c ? 0.94 : 0.02);
	}
	M16 M;
	for (int i = 0; i < 16; i++) M.m[i] = (i / 4 == i % 4) ? 0.94 : 0.02;
	double *dm, *out;
	CK(hipMalloc(&dm, sizeof(hm)));
	CK(hipMemcpy(dm, hm, sizeof(hm), hipMemcpyHostToDevice));
	const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU = 4 waves per SIMD
	CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	auto report = [&](const char *name, float ms) {
		const double ns = ms * 1e6 / iters;  // per op per wave (all waves run concurrently, 4 per SIMD)
		printf("%-28s %8.2f ms  %7.1f ns per op-round = %6.0f SIMD-cycles per (wave, op) at 2.4 GHz\n", name, ms, ns, ns * 2.4 / 4.0);
	};
#define RUN(NAME, LAUNCH)                                  \
	do {                                                   \
		LAUNCH;                                            \
		CK(hipDeviceSynchronize());                        \
		CK(hipEventRecord(e0));                            \
		LAUNCH;                                            \
		CK(hipEventRecord(e1));                            \
		CK(hipEventSynchronize(e1));                       \
		float ms;                                          \
		CK(hipEventElapsedTime(&ms, e0, e1));              \
		report(NAME, ms);                                  \
	} while (0)
	RUN("mfma only (24 per op)", (k_mfma_only<<<blocks, 256>>>(out, iters, dm)));
	RUN("T layout, fill 0", (k_T<0><<<blocks, 256>>>(out, iters, dm)));
	RUN("T layout, fill 10 (30 int)", (k_T<10><<<blocks, 256>>>(out, iters, dm)));
	RUN("T layout, fill 30 (90 int)", (k_T<30><<<blocks, 256>>>(out, iters, dm)));
	RUN("L layout, fill 0", (k_L<0><<<blocks, 256>>>(out, iters, M, M, M, M, M)));
	RUN("L layout, fill 10 (30 int)", (k_L<10><<<blocks, 256>>>(out, iters, M, M, M, M, M)));
	RUN("L layout, fill 30 (90 int)", (k_L<30><<<blocks, 256>>>(out, iters, M, M, M, M, M)));
	return 0;
}
