// Microbenchmark: at what rate does a SIMD of gfx950 issue v_fma_f64 / v_mul_f64, by operand form, instruction-level parallelism
// and waves per SIMD?  (measurement aid for DESIGN.md; not product code)  The nominal rate is one wave instruction per 4 cycles.
// build: hipcc --offload-arch=gfx950 -O3 -o fma64_issue fma64_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// ILP independent chains, 64 FMAs per chain step; SG: the multiplier is an SGPR pair (kernel argument), else a VGPR pair
template <int ILP, bool SG>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double m_arg, double add) {
	double acc[ILP];
	const double mv = SG ? m_arg : m_arg + 1e-12 * threadIdx.x;  // (per-lane value: stays in VGPRs)
#pragma unroll
	for (int j = 0; j < ILP; j++) acc[j] = 1.0 + j + threadIdx.x;
#pragma unroll 1
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 64 / ILP; r++) {
#pragma unroll
			for (int j = 0; j < ILP; j++) acc[j] = fma(acc[j], mv, add);
		}
	}
	double s = 0;
#pragma unroll
	for (int j = 0; j < ILP; j++) s += acc[j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// a 4 x 4 mat-vec stream as the walk kernels do it: matrix in SGPRs (kernel arguments), 16 instructions per mat-vec, chains of 4
struct M16 { double m[16]; };
__global__ __launch_bounds__(256) void k_matvec(double *out, int iters, M16 M) {
	double v0 = 1.0 + threadIdx.x, v1 = 0.5, v2 = 0.25, v3 = 0.125;
#pragma unroll 1
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const double o0 = fma(M.m[3], v3, fma(M.m[2], v2, fma(M.m[1], v1, M.m[0] * v0)));
			const double o1 = fma(M.m[7], v3, fma(M.m[6], v2, fma(M.m[5], v1, M.m[4] * v0)));
			const double o2 = fma(M.m[11], v3, fma(M.m[10], v2, fma(M.m[9], v1, M.m[8] * v0)));
			const double o3 = fma(M.m[15], v3, fma(M.m[14], v2, fma(M.m[13], v1, M.m[12] * v0)));
			v0 = o0; v1 = o1; v2 = o2; v3 = o3;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
}

int main() {
	double *out;
	CK(hipMalloc(&out, (size_t)256 * 8 * 256 * 8));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const int iters = 20000;
	M16 M;
	for (int i = 0; i < 16; i++) M.m[i] = (i / 4 == i % 4) ? 0.94 : 0.02;
	for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD = workgroups of 4 waves per CU
		const int blocks = 256 * wps;
#define RUN(NAME, INSTR, LAUNCH)                                                                                          \
	do {                                                                                                                  \
		LAUNCH;                                                                                                           \
		CK(hipDeviceSynchronize());                                                                                       \
		CK(hipEventRecord(e0));                                                                                           \
		LAUNCH;                                                                                                           \
		CK(hipEventRecord(e1));                                                                                           \
		CK(hipEventSynchronize(e1));                                                                                      \
		float ms;                                                                                                         \
		CK(hipEventElapsedTime(&ms, e0, e1));                                                                             \
		printf("%d waves/SIMD  %-26s %7.2f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", wps, NAME, ms,    \
		       ms * 1e-3 * 2.4e9 / ((double)iters * (INSTR) * wps));                                                      \
	} while (0)
		RUN("fma vgpr ILP1", 64, (k_fma<1, false><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("fma vgpr ILP2", 64, (k_fma<2, false><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("fma vgpr ILP4", 64, (k_fma<4, false><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("fma vgpr ILP8", 64, (k_fma<8, false><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("fma sgpr ILP1", 64, (k_fma<1, true><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("fma sgpr ILP4", 64, (k_fma<4, true><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("fma sgpr ILP8", 64, (k_fma<8, true><<<blocks, 256>>>(out, iters, 0.999, 0.001)));
		RUN("matvec sgpr matrix", 64, (k_matvec<<<blocks, 256>>>(out, iters, M)));
	}
	return 0;
}
