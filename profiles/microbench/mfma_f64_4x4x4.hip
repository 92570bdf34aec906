#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// layout probe for v_mfma_f64_4x4x4_4b_f64: each lane supplies one A and one B double and receives one D double
__global__ void k(const double *A, const double *B, double *D) {
	const int l = threadIdx.x;
	D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void rate(double *out, int iters, double a0, double b0) {
	double acc = 0.0, a = a0 + threadIdx.x, b = b0 - threadIdx.x;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16; r++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
	std::vector<double> A(64), B(64), D(64);
	double *dA, *dB, *dD;
	hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
	// probe: A = one-hot at lane la, B = one-hot at lane lb -> which D lanes light up
	for (int la : {0, 1, 4, 5, 16, 21, 37}) {
		for (int lb : {0, 1, 4, 5, 16, 21, 37}) {
			for (int i = 0; i < 64; i++) { A[i] = 0; B[i] = 0; }
			A[la] = 1.0; B[lb] = 1.0;
			hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
			k<<<1, 64>>>(dA, dB, dD);
			hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
			printf("A@%2d B@%2d ->", la, lb);
			for (int i = 0; i < 64; i++) if (D[i] != 0.0) printf(" D[%d]=%g", i, D[i]);
			printf("\n");
		}
	}
	double *out; hipMalloc(&out, 8 * 256 * 256);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	rate<<<256, 256>>>(out, 10, 1.0, 2.0); hipDeviceSynchronize();
	hipEventRecord(e0); rate<<<256, 256>>>(out, 4000, 1.0, 2.0); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("4x4x4_4b: %.1f ns per MFMA per wave\n", ms * 1e6 / (4000.0 * 16));
	return 0;
}
