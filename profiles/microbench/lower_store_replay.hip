// Microbenchmark: the store stream of k_lower4_stream with NO arithmetic (measurement aid for DESIGN.md; not product code).
// cfg5: 15 625 blocks of 64 patterns x 4 categories = 62 500 waves, each stores 360 results of 2 KB (two 1 KB whole-line
// non-temporal instructions) into [node][category][P][4] planes (node planes 128 MB apart, category planes 32 MB apart).
// Workgroup = four consecutive blocks of one category (blockIdx.x = group * C + c), as in the kernel.
//   WAITN   s_waitcnt vmcnt(WAITN) at the top of every op (-1: never wait); the kernel waits with vmcnt(2)
//   LAYOUT  0: node-major planes (the engine's); 1: [node][group][C][256 patterns][4] (8 KB contiguous per workgroup and node)
//   spin    dependent fp64 FMAs per op (emulates the op's arithmetic; 0 = pure store stream)
// build: hipcc --offload-arch=gfx950 -O3 -o lower_store_replay lower_store_replay.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double dv2 __attribute__((ext_vector_type(2)));

template <int WAITN, int LAYOUT, int MINW>
__global__ __launch_bounds__(256, MINW) void k_replay(double *buf, int nodes, int nb, int C, int spin, double seed) {
	const int lane = threadIdx.x, wv = threadIdx.y;
	const int c = blockIdx.x % C, grp = blockIdx.x / C, blk = grp * 4 + wv;
	if (blk >= nb) return;
	const size_t P = (size_t)nb * 64, plane = P * 4;
	dv2 lo{seed + lane, seed}, hi{seed, seed - lane};
#pragma unroll 1
	for (int i = 0; i < nodes; i++) {
		if (WAITN == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (WAITN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
		if (WAITN == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
		if (WAITN == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
		double x = lo.x;
		for (int s = 0; s < spin; s++) x = fma(x, 1.0000001, 1e-9);
		lo.x = x;
		char *dst;
		if (LAYOUT == 0) dst = reinterpret_cast<char *>(buf + ((size_t)i * C + c) * plane) + (size_t)blk * 2048 + lane * 16;
		else dst = reinterpret_cast<char *>(buf + (size_t)i * C * plane) + ((size_t)grp * C + c) * 8192 + wv * 2048 + lane * 16;
		__builtin_nontemporal_store(lo, reinterpret_cast<dv2 *>(dst));
		__builtin_nontemporal_store(hi, reinterpret_cast<dv2 *>(dst + 1024));
	}
}

int main(int argc, char **argv) {
	const int nodes = 360, C = 4, nb = 15625;
	const size_t bytes = (size_t)nodes * C * nb * 2048;
	double *buf;
	CK(hipMalloc(&buf, bytes));
	CK(hipMemset(buf, 0, bytes));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const dim3 grid(((nb + 3) / 4) * C), block(64, 4);
#define RUN(NAME, KERNEL, SPIN)                                                                   \
	do {                                                                                          \
		float best = 1e9;                                                                         \
		for (int rep = 0; rep < 3; rep++) {                                                       \
			CK(hipEventRecord(e0));                                                               \
			KERNEL<<<grid, block>>>(buf, nodes, nb, C, SPIN, 1.0);                                \
			CK(hipEventRecord(e1));                                                               \
			CK(hipEventSynchronize(e1));                                                          \
			float ms;                                                                             \
			CK(hipEventElapsedTime(&ms, e0, e1));                                                 \
			if (ms < best) best = ms;                                                             \
		}                                                                                         \
		printf("%-52s spin %4d: %7.2f ms  %5.2f TB/s\n", NAME, SPIN, best, bytes / 1e9 / best);   \
	} while (0)
	RUN("node-major, no wait, 6 waves", (k_replay<-1, 0, 6>), 0);
	RUN("node-major, vmcnt(2) per op, 6 waves", (k_replay<2, 0, 6>), 0);
	RUN("node-major, vmcnt(4) per op, 6 waves", (k_replay<4, 0, 6>), 0);
	RUN("node-major, vmcnt(0) per op, 6 waves", (k_replay<0, 0, 6>), 0);
	RUN("node-major, no wait, 8 waves", (k_replay<-1, 0, 8>), 0);
	RUN("node-major, no wait, 4 waves", (k_replay<-1, 0, 4>), 0);
	RUN("blocked 8 KB, no wait, 6 waves", (k_replay<-1, 1, 6>), 0);
	RUN("blocked 8 KB, vmcnt(2) per op, 6 waves", (k_replay<2, 1, 6>), 0);
	for (int spin : {50, 100, 200, 400}) {
		RUN("node-major, vmcnt(2) per op, 6 waves", (k_replay<2, 0, 6>), spin);
		RUN("node-major, vmcnt(6) per op, 6 waves", (k_replay<6, 0, 6>), spin);
		RUN("node-major, no wait, 6 waves", (k_replay<-1, 0, 6>), spin);
	}
	return 0;
}
