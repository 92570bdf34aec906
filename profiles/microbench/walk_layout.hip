// Microbenchmark: does the node-major layout ([node][C][P][4]: every op of a workgroup lands in a different 128 MB array)
// cost the tree-walk kernels address-translation / DRAM-page locality?  Each wave loops over NODES "ops"; per op it reads
// two earlier 2 KB chunks and writes one, like k_lower4_walk with stored children.  Layout 0: node-major; layout 1:
// block-major ([pattern block][node][...]: a workgroup's data for all nodes is contiguous).
// build: hipcc --offload-arch=gfx950 -O3 -o walk_layout walk_layout.hip ; run: ./walk_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int LAYOUT>
__global__ __launch_bounds__(256) void walk(double *buf, int nodes, size_t blocks, int C) {
	const int lane = threadIdx.x, c = threadIdx.y;
	const size_t blk = blockIdx.x;
	auto addr = [&](int node) -> double * {
		if (LAYOUT == 0) return buf + (((size_t)node * C + c) * blocks + blk) * 256 + lane * 4;      // [node][c][block][64][4]
		return buf + (((size_t)blk * nodes + node) * C + c) * 256 + lane * 4;                            // [block][node][c][64][4]
	};
	double4 carry = {1., 1., 1., 1.};
	for (int i = 2; i < nodes; i++) {
		const double4 a = *reinterpret_cast<const double4 *>(addr(i - 2));  // an earlier result (cold-ish)
		carry.x = carry.x * 0.5 + a.x;
		carry.y = carry.y * 0.5 + a.y;
		carry.z = carry.z * 0.5 + a.z;
		carry.w = carry.w * 0.5 + a.w;
		*reinterpret_cast<double4 *>(addr(i)) = carry;
	}
}

int main() {
	const int nodes = 360, C = 4;
	const size_t blocks = 15625;  // 1e6 patterns / 64
	const size_t n = (size_t)nodes * C * blocks * 256;
	double *buf;
	CK(hipMalloc(&buf, n * sizeof(double)));
	CK(hipMemset(buf, 0, n * sizeof(double)));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	for (int layout = 0; layout < 2; layout++)
		for (int rep = 0; rep < 3; rep++) {
			CK(hipEventRecord(e0));
			if (layout == 0) hipLaunchKernelGGL(walk<0>, dim3(blocks), dim3(64, C), 0, 0, buf, nodes, blocks, C);
			else hipLaunchKernelGGL(walk<1>, dim3(blocks), dim3(64, C), 0, 0, buf, nodes, blocks, C);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			const double gb = (double)(nodes - 2) * C * blocks * 2048.0 * 2 / 1e9;  // one read + one write of 2 KB per op and wave
			printf("layout %d (%s): %.2f ms, %.1f GB moved, %.2f TB/s\n", layout, layout ? "block-major" : "node-major", ms, gb, gb / ms);
		}
	return 0;
}
