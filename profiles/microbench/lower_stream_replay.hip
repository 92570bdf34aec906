// Microbenchmark: the whole memory side of k_lower4_stream with emulated arithmetic (measurement aid for DESIGN.md; not product code).
// Per (wave, op): two LDS-DMA pieces of a 2 KB table block (L2-resident table, 360 x C blocks), two LDS-DMA mask-word rows
// (one dword per lane each, streamed from HBM), the previous result stored as two 1 KB non-temporal whole-line instructions after a
// transposition through LDS, a handful of LDS row reads and NFMA fp64 FMAs.  The requests of op i + DIST leave at the top of op i.
//   bits of `what`: 1 table DMA from the real table (else always block 0), 2 words from the real stream (else row 0), 4 stores,
//                   8 LDS transposition of the result, 16 tip-row reads
// build: hipcc --offload-arch=gfx950 -O3 -o lower_stream_replay lower_stream_replay.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double dv2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) char *lds_cptr;
typedef __attribute__((address_space(3))) dv2 lds_dv2;

__device__ __forceinline__ void dma16(const char *g, lds_cptr d) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)d, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const char *g, lds_cptr d) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)d, 4, 0, 0);
}

template <int DIST, int MINW>
__global__ __launch_bounds__(256, MINW) void k_replay(double *buf, const char *table, const unsigned *words, int nodes, int nb, int C, int nrows, int what, int nfma, int per_wave, int K) {
	extern __shared__ double sh[];
	constexpr int RING = DIST + 1;
	const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
	const int c = blockIdx.x % C, grp = blockIdx.x / C, blk = grp * 4 + wv;
	if (blk >= nb) return;
	const size_t P = (size_t)nb * 64, plane = P * 4, mrow = P * 4;
	const lds_cptr base = (lds_cptr)sh + (size_t)wv * per_wave;
	const lds_cptr blocks = base, wrows = base + RING * 2048, stage = wrows + RING * 1024;
	const char *tab_c = table + (size_t)c * nodes * 2048 + lane * 16;
	const char *wbase = reinterpret_cast<const char *>(words) + ((size_t)blk * 64 + lane) * 4;
	auto request = [&](int op) {
		const int slot = op % RING;
		const char *src = tab_c + ((what & 1) ? (size_t)op * 2048 : 0);
		dma16(src, blocks + slot * 2048);
		dma16(src + 1024, blocks + slot * 2048 + 1024);
		if (what & 32) {  // packed words: 16 bytes per lane (1 KB per wave, contiguous) every K ops; the other ops re-read a hot line
			const bool real = (op % K) == 0;
			dma16(real ? reinterpret_cast<const char *>(words) + ((size_t)(op / K) * P + (size_t)blk * 64 + lane) * 16 : tab_c, wrows + slot * 1024);
		} else {
			const size_t row = (what & 2) ? (size_t)((long)op * nrows / nodes) : 0;
			dma4(wbase + row * mrow, wrows + slot * 1024);
			dma4(wbase + (row + 1) * mrow, wrows + slot * 1024 + 256);
		}
	};
	for (int j = 0; j < DIST && j < nodes; j++) request(j);
	double r0 = 1.0 + lane, r1 = 0.5, r2 = 0.25, r3 = 0.125;
#pragma unroll 1
	for (int i = 0; i < nodes; i++) {
		// requests of op i are DIST ops old: younger = DIST - 1 request groups (4 each) + DIST store pairs
		if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		else if (DIST == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
		else if (DIST == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
		const int slot = i % RING;
		const unsigned w0 = *(const __attribute__((address_space(3))) unsigned *)(wrows + slot * 1024 + lane * 4);
		const unsigned w1 = *(const __attribute__((address_space(3))) unsigned *)(wrows + slot * 1024 + 256 + lane * 4);
		if (i + DIST < nodes) request(i + DIST);
		else { request(0); }  // (keeps the instruction count per op constant)
		if (what & 4) {
			dv2 lo{r0, r1}, hi{r2, r3};
			if (what & 8) {
				__attribute__((address_space(3))) dv2 *w = (__attribute__((address_space(3))) dv2 *)(stage + lane * 32);
				w[0] = lo;
				w[1] = hi;
				lo = *(const lds_dv2 *)(stage + lane * 16);
				hi = *(const lds_dv2 *)(stage + 1024 + lane * 16);
			}
			char *dst = reinterpret_cast<char *>(buf + ((size_t)i * C + c) * plane) + (size_t)blk * 2048 + lane * 16;
			__builtin_nontemporal_store(lo, reinterpret_cast<dv2 *>(dst));
			__builtin_nontemporal_store(hi, reinterpret_cast<dv2 *>(dst + 1024));
		} else {
			dma16(tab_c, stage);  // two vector-memory instructions in place of the stores keep the counts of the waits honest
			dma16(tab_c + 1024, stage + 1024);
		}
		double t0 = 1.0, t1 = 1.0, t2 = 1.0, t3 = 1.0;
		if (what & 16) {
			const lds_cptr tb = blocks + slot * 2048;
			const dv2 a = *(const lds_dv2 *)(tb + 64 + ((w0 << 5) & 0xE0u)), b = *(const lds_dv2 *)(tb + 64 + 16 + ((w0 << 5) & 0xE0u));
			const dv2 cc = *(const lds_dv2 *)(tb + 1024 + ((w1 << 1) & 0xE0u)), d = *(const lds_dv2 *)(tb + 1024 + 16 + ((w1 << 1) & 0xE0u));
			t0 = a.x + cc.x; t1 = a.y + cc.y; t2 = b.x + d.x; t3 = b.y + d.y;
		} else {
			t0 += (double)(w0 & 1u) + (double)(w1 & 1u);
		}
		for (int s = 0; s < nfma; s += 4) {
			r0 = fma(r0, 0.999, t0); r1 = fma(r1, 0.999, t1); r2 = fma(r2, 0.999, t2); r3 = fma(r3, 0.999, t3);
		}
	}
	if (r0 + r1 + r2 + r3 == 12345.678) buf[0] = r0;
}

int main(int argc, char **argv) {
	const int nodes = 360, C = 4, nb = 15625, nrows = 424;
	const size_t bytes = (size_t)nodes * C * nb * 2048, wbytes = (size_t)(nodes + 2) * nb * 64 * 16, tbytes = (size_t)nodes * C * 2048;
	double *buf;
	char *table;
	unsigned *words;
	CK(hipMalloc(&buf, bytes));
	CK(hipMalloc(&table, tbytes));
	CK(hipMalloc(&words, wbytes));
	CK(hipMemset(buf, 0, bytes));
	CK(hipMemset(table, 0, tbytes));
	CK(hipMemset(words, 0, wbytes));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const dim3 grid(((nb + 3) / 4) * C), block(64, 4);
	auto run = [&](const char *name, int dist, int minw, int what, int nfma, int K = 1) -> int {
		const int ring = dist + 1, per_wave = ring * 2048 + ring * 1024 + 2048;
		float best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			CK(hipEventRecord(e0));
			const size_t lds = (size_t)per_wave * 4;
			if (dist == 1 && minw == 6) k_replay<1, 6><<<grid, block, lds>>>(buf, table, words, nodes, nb, C, nrows, what, nfma, per_wave, K);
			else if (dist == 1 && minw == 4) k_replay<1, 4><<<grid, block, lds>>>(buf, table, words, nodes, nb, C, nrows, what, nfma, per_wave, K);
			else if (dist == 2) k_replay<2, 4><<<grid, block, lds>>>(buf, table, words, nodes, nb, C, nrows, what, nfma, per_wave, K);
			else k_replay<3, 4><<<grid, block, lds>>>(buf, table, words, nodes, nb, C, nrows, what, nfma, per_wave, K);
			CK(hipGetLastError());
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		printf("%-40s dist %d lds/wg %5d  what %2d  fma %3d K %2d: %7.2f ms\n", name, dist, per_wave * 4, what, nfma, K, best);
		return 0;
	};
	// (the packed modes issue ONE word instruction per op: vmcnt(2) then also leaves one table piece in flight -- timing only)
	for (int nfma : {64}) {
		run("everything", 1, 6, 31, nfma);
		run("stores + table, words hot", 1, 6, 4 + 8 + 1 + 16, nfma);
		for (int K : {1, 2, 4, 8, 16}) run("everything, packed words", 1, 6, 31 + 32, nfma, K);
		for (int K : {1, 4, 8}) run("no stores, packed words", 1, 6, 31 + 32 - 4 - 8, nfma, K);
	}
	return 0;
}
