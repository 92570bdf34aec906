// Microbenchmark: what HBM bandwidth does the likelihood kernels' access pattern reach with NO arithmetic?
// (measurement aid for DESIGN.md; not part of the product).  Build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip
//   A  "lower-like": per (pattern, category) read two 32-byte vectors from two arrays, write one (layout [C][P][4])
//   B  same bytes, but each lane handles two adjacent patterns (64 contiguous bytes per lane per array)
//   C  plain 16-byte-per-lane grid-stride copy (the 6.3 TB/s calibration kernel of the guide)
//   D  "upper-like": read three arrays, write two
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct d4 { double x, y, z, w; };
__device__ __forceinline__ d4 load4(const double *p) { const double2 a = ((const double2 *)p)[0], b = ((const double2 *)p)[1]; return d4{a.x, a.y, b.x, b.y}; }
__device__ __forceinline__ void store4(double *p, d4 v) { ((double2 *)p)[0] = double2{v.x, v.y}; ((double2 *)p)[1] = double2{v.z, v.w}; }
__device__ __forceinline__ d4 mul(d4 a, d4 b) { return d4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

template <int NIN, int NOUT, int PPT>
__global__ __launch_bounds__(256) void k_pattern(const double *__restrict__ base, double *__restrict__ outb, int P, int C, size_t node_stride) {
	const int lane = threadIdx.x, c = threadIdx.y;
	const size_t plane = (size_t)P * 4;
	const double *in = base + (size_t)blockIdx.y * NIN * node_stride + (size_t)c * plane;
	double *out = outb + (size_t)blockIdx.y * NOUT * node_stride + (size_t)c * plane;
#pragma unroll 1
	for (int q = 0; q < PPT; q++) {
		const int k = (blockIdx.x * PPT + q) * 64 + lane;
		if (k >= P) break;
		d4 v = load4(in + (size_t)k * 4);
#pragma unroll
		for (int i = 1; i < NIN; i++) v = mul(v, load4(in + i * node_stride + (size_t)k * 4));
#pragma unroll
		for (int i = 0; i < NOUT; i++) store4(out + i * node_stride + (size_t)k * 4, v);
	}
}

template <int NIN, int NOUT, int PPT>
__global__ __launch_bounds__(256) void k_pattern2(const double *__restrict__ base, double *__restrict__ outb, int P, int C, size_t node_stride) {
	const int lane = threadIdx.x, c = threadIdx.y;
	const size_t plane = (size_t)P * 4;
	const double *in = base + (size_t)blockIdx.y * NIN * node_stride + (size_t)c * plane;
	double *out = outb + (size_t)blockIdx.y * NOUT * node_stride + (size_t)c * plane;
#pragma unroll 1
	for (int q = 0; q < PPT; q++) {
		const int k = ((blockIdx.x * PPT + q) * 64 + lane) * 2;
		if (k >= P) break;
		d4 v0 = load4(in + (size_t)k * 4), v1 = load4(in + (size_t)k * 4 + 4);
#pragma unroll
		for (int i = 1; i < NIN; i++) { v0 = mul(v0, load4(in + i * node_stride + (size_t)k * 4)); v1 = mul(v1, load4(in + i * node_stride + (size_t)k * 4 + 4)); }
#pragma unroll
		for (int i = 0; i < NOUT; i++) { store4(out + i * node_stride + (size_t)k * 4, v0); store4(out + i * node_stride + (size_t)k * 4 + 4, v1); }
	}
}

typedef const __attribute__((address_space(4))) double *cptr;
__device__ __forceinline__ cptr opaque(cptr p) { asm volatile("" : "+s"(p)); return p; }
template <bool FENCE>
__device__ __forceinline__ d4 matvec4(cptr M, d4 v) {
	d4 r;
	r.x = M[0] * v.x + M[1] * v.y + M[2] * v.z + M[3] * v.w;
	r.y = M[4] * v.x + M[5] * v.y + M[6] * v.z + M[7] * v.w;
	r.z = M[8] * v.x + M[9] * v.y + M[10] * v.z + M[11] * v.w;
	r.w = M[12] * v.x + M[13] * v.y + M[14] * v.z + M[15] * v.w;
	if (FENCE) __builtin_amdgcn_sched_barrier(0);
	return r;
}
// F: the product's lower-pass arithmetic on stored children: out = (M_l a) o (M_r b), matrices through scalar loads.
// MODE 0: opaque pointer + scheduler fence (the product); 1: opaque, no fence; 2: plain pointers (hoisted out of the loop)
template <int MODE, int NMV>
__global__ __launch_bounds__(256) void k_lower_math(const double *__restrict__ base, double *__restrict__ outb, const double *__restrict__ mats, int P, int C,
                                                    size_t node_stride) {
	const int lane = threadIdx.x, c = __builtin_amdgcn_readfirstlane(threadIdx.y);
	const size_t plane = (size_t)P * 4;
	const double *in = base + (size_t)blockIdx.y * 2 * node_stride + (size_t)c * plane;
	double *out = outb + (size_t)blockIdx.y * node_stride + (size_t)c * plane;
	const cptr M0 = (cptr)(mats + ((size_t)blockIdx.y * 8 + c) * 16);
#pragma unroll 1
	for (int q = 0; q < 4; q++) {
		const int k = (blockIdx.x * 4 + q) * 64 + lane;
		if (k >= P) break;
		d4 a = load4(in + (size_t)k * 4), b = load4(in + node_stride + (size_t)k * 4);
#pragma unroll
		for (int i = 0; i < NMV; i++) {
			cptr Ma = M0 + i * 64, Mb = M0 + i * 64 + 32 * 16;
			if (MODE < 2) { Ma = opaque(Ma); Mb = opaque(Mb); }
			a = matvec4<MODE == 0>(Ma, a);
			b = matvec4<MODE == 0>(Mb, b);
		}
		store4(out + (size_t)k * 4, mul(a, b));
	}
}

__global__ void k_copy(const double2 *__restrict__ in, double2 *__restrict__ out, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
	const int P = 1000000, C = 4, NODES = 40;
	const size_t node = (size_t)C * P * 4;  // doubles per node array (128 MB)
	double *in = nullptr, *out = nullptr;
	CHECK(hipMalloc(&in, sizeof(double) * node * NODES * 3));
	CHECK(hipMalloc(&out, sizeof(double) * node * NODES * 2));
	CHECK(hipMemset(in, 0, sizeof(double) * node * NODES * 3));
	CHECK(hipMemset(out, 0, sizeof(double) * node * NODES * 2));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	auto timeit = [&](const char *name, double bytes, auto launch) {
		launch();
		hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int r = 0; r < 3; r++) launch();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / 3, bytes / (ms / 3 * 1e-3) / 1e9);
	};
	const double nb = (double)node * 8 * NODES;
	timeit("A lower-like 2 in 1 out, PPT 4", 3 * nb, [&] { hipLaunchKernelGGL((k_pattern<2, 1, 4>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	timeit("A lower-like 2 in 1 out, PPT 1", 3 * nb, [&] { hipLaunchKernelGGL((k_pattern<2, 1, 1>), dim3((P + 63) / 64, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	timeit("A lower-like 2 in 1 out, PPT 16", 3 * nb, [&] { hipLaunchKernelGGL((k_pattern<2, 1, 16>), dim3((P + 1023) / 1024, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	timeit("B lower-like, 2 patterns per lane, PPT 2", 3 * nb, [&] { hipLaunchKernelGGL((k_pattern2<2, 1, 2>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	timeit("D upper-like 3 in 2 out, PPT 4", 5 * nb, [&] { hipLaunchKernelGGL((k_pattern<3, 2, 4>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	timeit("D upper-like, 2 patterns per lane, PPT 2", 5 * nb, [&] { hipLaunchKernelGGL((k_pattern2<3, 2, 2>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	timeit("E read-only-ish 3 in 0 out (1 tiny)", 3 * nb, [&] { hipLaunchKernelGGL((k_pattern<3, 0, 4>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, P, C, node); });
	double *mats = nullptr;
	CHECK(hipMalloc(&mats, sizeof(double) * 16 * 64 * (NODES * 8 + 64)));
	CHECK(hipMemset(mats, 0, sizeof(double) * 16 * 64 * (NODES * 8 + 64)));
	timeit("F0 lower math 1 matvec/child, opaque+fence", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<0, 1>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F1 lower math 1 matvec/child, opaque", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<1, 1>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F2 lower math 1 matvec/child, hoisted", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<2, 1>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F0x3 lower math 3 matvec/child, opaque+fence", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<0, 3>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F1x3 lower math 3 matvec/child, opaque", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<1, 3>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F2x3 lower math 3 matvec/child, hoisted", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<2, 3>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F0x6 lower math 6 matvec/child, opaque+fence", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<0, 6>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	timeit("F1x6 lower math 6 matvec/child, opaque", 3 * nb, [&] { hipLaunchKernelGGL((k_lower_math<1, 6>), dim3((P + 255) / 256, NODES), dim3(64, 4), 0, 0, in, out, mats, P, C, node); });
	const size_t n2 = node * NODES / 2;
	timeit("C copy 16 B/lane grid-stride (2048 blocks)", 2 * nb, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const double2 *)in, (double2 *)out, n2); });
	return 0;
}
