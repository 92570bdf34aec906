#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
	f64x4 acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = f64x4{0., 0., 0., 0.};
	double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16 / NACC; r++)
#pragma unroll
			for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	double s = 0;
	for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(const char *name, int wgs, int threads) {
	double *out;
	hipMalloc(&out, sizeof(double) * wgs * threads);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 4000;
	k<NACC><<<wgs, threads>>>(out, 10, 1.0, 2.0);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<NACC><<<wgs, threads>>>(out, iters, 1.0, 2.0);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double mf = (double)iters * 16 * (threads / 64) * wgs;
	printf("%s wgs %d threads %d: %.3f ms, %.1f ns per MFMA per wave, %.2f TFLOP/s\n", name, wgs, threads, ms, ms * 1e6 / (iters * 16.0), mf * 2048 / (ms * 1e-3) / 1e12);
	hipFree(out);
}
int main() {
	run<1>("1 acc ", 256, 256);
	run<4>("4 acc ", 256, 256);
	run<16>("16 acc", 256, 256);
	run<4>("4 acc 2 waves/SIMD", 256, 512);
	run<4>("4 acc 1 CU", 1, 256);
	run<1>("1 acc 1 CU", 1, 256);
	return 0;
}
