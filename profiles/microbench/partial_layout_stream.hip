#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
// each lane owns 32 bytes (two 16-byte halves): the 4-state partial layout
template <bool NT>
__global__ __launch_bounds__(256) void k_write32(dv2 *out, size_t n32) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	dv2 v = {1.0 + threadIdx.x, 2.0}, w = {3.0, 4.0 + threadIdx.x};
	for (; i < n32; i += stride) {
		if (NT) {
			__builtin_nontemporal_store(v, out + 2 * i);
			__builtin_nontemporal_store(w, out + 2 * i + 1);
		} else {
			out[2 * i] = v;
			out[2 * i + 1] = w;
		}
	}
}
// split planes: halves in two separate arrays, each instruction contiguous
template <bool NT>
__global__ __launch_bounds__(256) void k_write_split(dv2 *out, size_t n32) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	dv2 v = {1.0 + threadIdx.x, 2.0}, w = {3.0, 4.0 + threadIdx.x};
	for (; i < n32; i += stride) {
		if (NT) {
			__builtin_nontemporal_store(v, out + i);
			__builtin_nontemporal_store(w, out + n32 + i);
		} else {
			out[i] = v;
			out[n32 + i] = w;
		}
	}
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read32(const dv2 *in, size_t n32, double *sink) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	dv2 acc = {0, 0};
	for (; i < n32; i += stride) {
		dv2 a = NT ? __builtin_nontemporal_load(in + 2 * i) : in[2 * i];
		dv2 b = NT ? __builtin_nontemporal_load(in + 2 * i + 1) : in[2 * i + 1];
		acc += a + b;
	}
	if (acc.x == 12345.678) sink[0] = acc.y;
}
// the walk's pattern: every workgroup writes its own 64-pattern slice of successive "node planes"
template <bool NT, bool SPLIT>
__global__ __launch_bounds__(256) void k_walk_write(dv2 *out, size_t P, int nodes) {
	// 4 waves = 4 categories; block owns 64 patterns
	const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
	const size_t k = (size_t)blockIdx.x * 64 + lane;
	if (k >= P) return;
	dv2 v = {1.0 + threadIdx.x, 2.0}, w = {3.0, 4.0 + threadIdx.x};
	for (int n = 0; n < nodes; n++) {
		dv2 *plane = out + ((size_t)n * 4 + c) * P * 2;
		if (SPLIT) {
			if (NT) { __builtin_nontemporal_store(v, plane + k); __builtin_nontemporal_store(w, plane + P + k); }
			else { plane[k] = v; plane[P + k] = w; }
		} else {
			if (NT) { __builtin_nontemporal_store(v, plane + 2 * k); __builtin_nontemporal_store(w, plane + 2 * k + 1); }
			else { plane[2 * k] = v; plane[2 * k + 1] = w; }
		}
		v.x += 1.0;
	}
}
int main() {
	const size_t bytes = (size_t)16 << 30, n32 = bytes / 32;
	dv2 *a;
	double *sink;
	hipMalloc(&a, bytes);
	hipMalloc(&sink, 8);
	hipMemset(a, 0, bytes);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	auto timeit = [&](const char *name, auto launch, double gb) {
		launch();
		hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int r = 0; r < 3; r++) launch();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("%-28s %.2f TB/s\n", name, gb * 3 / (ms * 1e-3) / 1e3);
	};
	const int grid = 256 * 32;
	timeit("write32 (2x16B stride 32)", [&] { k_write32<false><<<grid, 256>>>(a, n32); }, bytes / 1e9);
	timeit("write32 nt", [&] { k_write32<true><<<grid, 256>>>(a, n32); }, bytes / 1e9);
	timeit("write split planes", [&] { k_write_split<false><<<grid, 256>>>(a, n32); }, bytes / 1e9);
	timeit("write split planes nt", [&] { k_write_split<true><<<grid, 256>>>(a, n32); }, bytes / 1e9);
	timeit("read32", [&] { k_read32<false><<<grid, 256>>>(a, n32, sink); }, bytes / 1e9);
	timeit("read32 nt", [&] { k_read32<true><<<grid, 256>>>(a, n32, sink); }, bytes / 1e9);
	const size_t P = 1000000;
	const int nodes = (int)(bytes / (P * 32 * 4));
	const double gb = (double)nodes * 4 * P * 32 / 1e9;
	const int wg = (int)((P + 63) / 64);
	timeit("walk write", [&] { k_walk_write<false, false><<<wg, 256>>>(a, P, nodes); }, gb);
	timeit("walk write nt", [&] { k_walk_write<true, false><<<wg, 256>>>(a, P, nodes); }, gb);
	timeit("walk write split", [&] { k_walk_write<false, true><<<wg, 256>>>(a, P, nodes); }, gb);
	timeit("walk write split nt", [&] { k_walk_write<true, true><<<wg, 256>>>(a, P, nodes); }, gb);
	return 0;
}
