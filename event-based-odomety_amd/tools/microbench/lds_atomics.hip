// Microbenchmark: LDS atomic-add throughput on gfx950 by data type and address pattern.
// Every CU runs WG workgroups of 256 threads; each lane issues ITERS atomics.
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int LDS_ELEMS = 4096;  // 32 KB of 8-byte slots

// pattern 0: lane-linear (conflict free), 1: pseudo-random per lane, 2: all lanes one address,
// 3: 8 lanes share an address (8-way same-address), 4: stride 2 slots,
// 5: groups of 7 consecutive lanes on 7 consecutive slots from a random base per group (the
//    'seven lanes per event' layout), 6: a random base per lane plus a common tap offset (today's)
template <typename T, int PATTERN>
__global__ void k_atomic(T* out, int iters)
{
	__shared__ T lds[LDS_ELEMS];
	for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = T(0);
	__syncthreads();
	unsigned idx = threadIdx.x;
	unsigned rnd = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	T v = T(1);
	for (int it = 0; it < iters; ++it)
	{
		unsigned a;
		if (PATTERN == 0) a = (idx + it * 64) & (LDS_ELEMS - 1);
		else if (PATTERN == 1) { rnd = rnd * 1664525u + 1013904223u; a = (rnd >> 10) & (LDS_ELEMS - 1); }
		else if (PATTERN == 2) a = (it * 7) & (LDS_ELEMS - 1);
		else if (PATTERN == 3) a = ((idx >> 3) + it * 8) & (LDS_ELEMS - 1);
		else if (PATTERN == 4) a = (idx * 2 + it * 128) & (LDS_ELEMS - 1);
		else if (PATTERN == 5)
		{
			// per step a fresh base per 7-lane group (hash of group id and step), lane adds its column
			const unsigned g = (threadIdx.x & 63) / 7 + (threadIdx.x >> 6) * 9;
			unsigned h = (g * 2654435761u) ^ ((it / 7) * 40503u + blockIdx.x * 97u);
			h = h * 1664525u + 1013904223u;
			a = ((h >> 10) + (threadIdx.x & 63) % 7 + (it % 7) * 41) & (LDS_ELEMS - 1);
		}
		else if (PATTERN == 6)
		{
			if (it % 49 == 0) { rnd = rnd * 1664525u + 1013904223u; }
			a = ((rnd >> 10) + (it % 7) + ((it / 7) % 7) * 41) & (LDS_ELEMS - 1);
		}
		else
		{
			// 7 / 8: as 6, but the random bases of the lanes of one group of 16 (7) or 32 (8) lanes are pairwise distinct
			// modulo the group's size: what an ideal dealing of events over lanes would reach
			constexpr unsigned G = PATTERN == 7 ? 16u : 32u;
			if (it % 49 == 0) { rnd = rnd * 1664525u + 1013904223u; }
			const unsigned base = (((rnd >> 10) & ~(G - 1u)) | (threadIdx.x & (G - 1u)));
			a = (base + (it % 7) + ((it / 7) % 7) * 41) & (LDS_ELEMS - 1);
		}
		atomicAdd(&lds[a], v);
	}
	__syncthreads();
	T s = T(0);
	for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) s += lds[i];
	if (s == T(12345)) out[blockIdx.x] = s;
}

// The kernels' own shape: per "event" a random base per lane, then 7 x 7 taps at IMMEDIATE offsets (row pitch 41 slots).
// MODE 0: any base; 1 / 2: the bases of the lanes of a group of 16 / 32 lanes pairwise distinct modulo the group size --
// what an ideal dealing of a wave's events over its lanes would reach.  (Patterns 6-8 above compute `it % 7`, `it / 7` per
// operation and are bound by that arithmetic, not by the LDS.)
template <typename T, int MODE, bool ATOMIC>
__global__ void k_taps(T* out, int iters)
{
	__shared__ T lds[LDS_ELEMS + 7 * 41];
	for (int i = threadIdx.x; i < LDS_ELEMS + 7 * 41; i += blockDim.x) lds[i] = T(1);
	__syncthreads();
	unsigned rnd = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	T acc = T(0);
	for (int ev = 0; ev < iters / 49; ++ev)
	{
		rnd = rnd * 1664525u + 1013904223u;
		unsigned base = (rnd >> 10) & (LDS_ELEMS - 1);
		if (MODE == 1) base = (base & ~15u) | (threadIdx.x & 15u);
		if (MODE == 2) base = (base & ~31u) | (threadIdx.x & 31u);
		T* p = lds + base;
#pragma unroll
		for (int j = 0; j < 7; ++j)
		{
#pragma unroll
			for (int i = 0; i < 7; ++i)
			{
				if (ATOMIC)
				{
					atomicAdd(p + j * 41 + i, T(1));
				}
				else
				{
					acc += __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + j * 41 + i),
																 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
				}
			}
		}
	}
	__syncthreads();
	if (acc == T(12345)) out[blockIdx.x] = acc + lds[threadIdx.x];
}

// plain 8-byte LDS reads with the same address patterns (the gather pass)
template <int PATTERN>
__global__ void k_read(double* out, int iters)
{
	__shared__ double lds[LDS_ELEMS];
	for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = double(i);
	__syncthreads();
	unsigned rnd = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	double acc = 0.0;
	for (int it = 0; it < iters; ++it)
	{
		unsigned a;
		if (PATTERN == 0) a = (threadIdx.x + it * 64) & (LDS_ELEMS - 1);
		else if (PATTERN == 1) { rnd = rnd * 1664525u + 1013904223u; a = (rnd >> 10) & (LDS_ELEMS - 1); }
		else if (PATTERN == 5)
		{
			const unsigned g = (threadIdx.x & 63) / 7 + (threadIdx.x >> 6) * 9;
			unsigned h = (g * 2654435761u) ^ ((it / 7) * 40503u + blockIdx.x * 97u);
			h = h * 1664525u + 1013904223u;
			a = ((h >> 10) + (threadIdx.x & 63) % 7 + (it % 7) * 41) & (LDS_ELEMS - 1);
		}
		else if (PATTERN == 6)
		{
			if (it % 49 == 0) { rnd = rnd * 1664525u + 1013904223u; }
			a = ((rnd >> 10) + (it % 7) + ((it / 7) % 7) * 41) & (LDS_ELEMS - 1);
		}
		else
		{
			constexpr unsigned G = PATTERN == 7 ? 16u : 32u;
			if (it % 49 == 0) { rnd = rnd * 1664525u + 1013904223u; }
			const unsigned base = (((rnd >> 10) & ~(G - 1u)) | (threadIdx.x & (G - 1u)));
			a = (base + (it % 7) + ((it / 7) % 7) * 41) & (LDS_ELEMS - 1);
		}
		acc += lds[a];
	}
	if (acc == -1.0) out[blockIdx.x] = acc;  // never true (all values >= 0): keeps the loads alive
}

// non-atomic read-modify-write through LDS for comparison (races ignored: timing only)
template <typename T>
__global__ void k_rmw(T* out, int iters)
{
	__shared__ T lds[LDS_ELEMS];
	for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) lds[i] = T(0);
	__syncthreads();
	unsigned idx = threadIdx.x;
	for (int it = 0; it < iters; ++it)
	{
		unsigned a = (idx + it * 64) & (LDS_ELEMS - 1);
		volatile T* p = &lds[a];
		*p = *p + T(1);
	}
	__syncthreads();
	T s = T(0);
	for (int i = threadIdx.x; i < LDS_ELEMS; i += blockDim.x) s += lds[i];
	if (s == T(12345)) out[blockIdx.x] = s;
}

template <typename T>
T* argOf(void (*)(T*, int));

template <typename K>
void run(const char* name, K kern, int blocks, int threads, void* out)
{
	static void* sink = nullptr;  // a real buffer: a kernel's guard store must never see a null pointer
	if (!sink) CHECK(hipMalloc(&sink, 1 << 20));
	(void)out;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<decltype(argOf(kern))>(sink), 16);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<decltype(argOf(kern))>(sink), ITERS);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	const double ops = double(blocks) * threads * ITERS;
	const double perClkCu = ops / (ms * 1e-3) / 256.0 / 2.4e9;
	printf("%-34s blocks %5d x %4d : %8.3f ms  %8.1f Gatomics/s  %6.2f lanes/clk/CU (at 2.4 GHz)\n", name, blocks, threads, ms, ops / ms / 1e6, perClkCu);
}

#define RUNALL(T, name) \
	run(name " linear", k_atomic<T, 0>, blocks, threads, nullptr); \
	run(name " random", k_atomic<T, 1>, blocks, threads, nullptr); \
	run(name " same-address", k_atomic<T, 2>, blocks, threads, nullptr); \
	run(name " 8-lanes-share", k_atomic<T, 3>, blocks, threads, nullptr); \
	run(name " stride2", k_atomic<T, 4>, blocks, threads, nullptr);

int main(int argc, char** argv)
{
	int threads = 256;
	for (int wgPerCu : {1, 4})
	{
		int blocks = 256 * wgPerCu;
		printf("---- %d workgroup(s) per CU ----\n", wgPerCu);
		RUNALL(double, "ds_add_f64")
		RUNALL(unsigned long long, "ds_add_u64")
		RUNALL(float, "ds_add_f32")
		RUNALL(unsigned int, "ds_add_u32")
		run("rmw f64 (read+add+write) linear", k_rmw<double>, blocks, threads, nullptr);
		run("ds_add_u64 7-lane groups", k_atomic<unsigned long long, 5>, blocks, threads, nullptr);
		run("ds_add_u64 random base + tap", k_atomic<unsigned long long, 6>, blocks, threads, nullptr);
		run("ds_add_u32 random base + tap", k_atomic<unsigned int, 6>, blocks, threads, nullptr);
		run("ds_add_u32 7-lane groups", k_atomic<unsigned int, 5>, blocks, threads, nullptr);
		run("ds_read_b64 linear", k_read<0>, blocks, threads, nullptr);
		run("ds_read_b64 random", k_read<1>, blocks, threads, nullptr);
		run("ds_read_b64 7-lane groups", k_read<5>, blocks, threads, nullptr);
		run("ds_read_b64 random base + tap", k_read<6>, blocks, threads, nullptr);
		run("7x7 taps ds_add_u64, any base", k_taps<unsigned long long, 0, true>, blocks, threads, nullptr);
		run("7x7 taps ds_add_u64, distinct mod 16", k_taps<unsigned long long, 1, true>, blocks, threads, nullptr);
		run("7x7 taps ds_add_u64, distinct mod 32", k_taps<unsigned long long, 2, true>, blocks, threads, nullptr);
		run("7x7 taps ds_read_b64, any base", k_taps<unsigned long long, 0, false>, blocks, threads, nullptr);
		run("7x7 taps ds_read_b64, distinct mod 16", k_taps<unsigned long long, 1, false>, blocks, threads, nullptr);
		run("7x7 taps ds_read_b64, distinct mod 32", k_taps<unsigned long long, 2, false>, blocks, threads, nullptr);
		// (patterns 7 / 8: bound by their own index arithmetic, kept for reference)
		run("ds_add_u64 base+tap, distinct mod 16", k_atomic<unsigned long long, 7>, blocks, threads, nullptr);
		run("ds_add_u64 base+tap, distinct mod 32", k_atomic<unsigned long long, 8>, blocks, threads, nullptr);
		run("ds_read_b64 base+tap, distinct mod 16", k_read<7>, blocks, threads, nullptr);
		run("ds_read_b64 base+tap, distinct mod 32", k_read<8>, blocks, threads, nullptr);
	}
	return 0;
}
