#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void k_write(dv2 *out, size_t n) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	dv2 v = {1.0 + threadIdx.x, 2.0};
	for (; i < n; i += stride) {
		if (NT) __builtin_nontemporal_store(v, out + i);
		else out[i] = v;
	}
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const dv2 *in, size_t n, double *sink) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	dv2 acc = {0, 0};
	for (; i < n; i += stride) {
		dv2 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
		acc += v;
	}
	if (acc.x == 12345.678) sink[0] = acc.y;
}
__global__ __launch_bounds__(256) void k_copy(const dv2 *in, dv2 *out, size_t n) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
int main() {
	const size_t bytes = (size_t)16 << 30, n = bytes / 16;
	dv2 *a, *b;
	double *sink;
	hipMalloc(&a, bytes);
	hipMalloc(&b, bytes);
	hipMalloc(&sink, 8);
	hipMemset(a, 0, bytes);
	hipMemset(b, 0, bytes);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int grid : {256 * 8, 256 * 32, 256 * 128}) {
		auto timeit = [&](const char *name, auto launch, double gb) {
			launch();
			hipDeviceSynchronize();
			hipEventRecord(e0);
			for (int r = 0; r < 3; r++) launch();
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			printf("grid %6d %-10s %.2f TB/s\n", grid, name, gb * 3 / (ms * 1e-3) / 1e3);
		};
		timeit("write", [&] { k_write<false><<<grid, 256>>>(a, n); }, bytes / 1e9);
		timeit("write nt", [&] { k_write<true><<<grid, 256>>>(a, n); }, bytes / 1e9);
		timeit("read", [&] { k_read<false><<<grid, 256>>>(a, n, sink); }, bytes / 1e9);
		timeit("read nt", [&] { k_read<true><<<grid, 256>>>(a, n, sink); }, bytes / 1e9);
		timeit("copy nt", [&] { k_copy<<<grid, 256>>>(a, b, n); }, 2 * bytes / 1e9);
	}
	return 0;
}
