// roundtrip.hip — how long does the host wait for a small kernel's results?  (round 4: the lock-step solve of ONE
// reference-default window is 15 rounds of launch -> 28 us kernel -> completion -> 9 us host LM step.)
// Three ways to learn that a launch has finished, timed per round over 2000 rounds with a kernel of ~20 us:
//   (a) hipEventRecord + hipEventSynchronize            (what eval_finish does)
//   (b) hipStreamSynchronize
//   (c) hipStreamWriteValue32 into pinned memory behind the kernel + a host spin on that word
//   (d) the kernel's last workgroup writes the word itself (system-scope release) + a host spin
//   hipcc --offload-arch=gfx950 -O3 roundtrip.hip -o roundtrip && ./roundtrip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void k_work(double* out, int iters, unsigned int* counter, volatile unsigned int* flag, unsigned int seq, int nBlocks)
{
	double a = threadIdx.x * 1e-3 + blockIdx.x;
	for (int i = 0; i < iters; ++i)
	{
		a = a * 1.0000001 + 1e-9;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a;
	if (flag)
	{
		__syncthreads();
		if (threadIdx.x == 0)
		{
			__threadfence_system();
			if (atomicAdd(counter, 1u) == static_cast<unsigned int>(nBlocks) - 1)
			{
				*counter = 0;
				__threadfence_system();
				*flag = seq;
			}
		}
	}
}

int main()
{
	const int blocks = 108, threads = 256, iters = 4000, rounds = 2000;
	double* d_out;
	unsigned int* d_counter;
	volatile unsigned int* h_flag;
	hipMalloc(&d_out, blocks * threads * sizeof(double));
	hipMalloc(&d_counter, 4);
	hipMemset(d_counter, 0, 4);
	hipHostMalloc(const_cast<unsigned int**>(&h_flag), 64, hipHostMallocDefault);
	*h_flag = 0;
	hipStream_t s;
	hipStreamCreate(&s);
	hipEvent_t e;
	hipEventCreateWithFlags(&e, hipEventDisableTiming);
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
	for (int mode = 0; mode < 4; ++mode)
	{
		for (int warm = 0; warm < 2; ++warm)
		{
			const auto t0 = now();
			for (int r = 1; r <= rounds; ++r)
			{
				const unsigned int seq = static_cast<unsigned int>(mode * 100000 + warm * 10000 + r);
				if (mode == 3)
				{
					hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s, d_out, iters, d_counter, h_flag, seq, blocks);
					while (*h_flag != seq)
					{
						__builtin_ia32_pause();
					}
					continue;
				}
				hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s, d_out, iters, d_counter, nullptr, 0u, blocks);
				if (mode == 0)
				{
					hipEventRecord(e, s);
					hipEventSynchronize(e);
				}
				else if (mode == 1)
				{
					hipStreamSynchronize(s);
				}
				else
				{
					if (hipStreamWriteValue32(s, const_cast<unsigned int*>(h_flag), seq, 0) != hipSuccess)
					{
						std::printf("hipStreamWriteValue32 not available\n");
						return 1;
					}
					while (*h_flag != seq)
					{
						__builtin_ia32_pause();
					}
				}
			}
			const auto t1 = now();
			if (warm)
			{
				const char* names[] = {"event record + synchronize", "stream synchronize", "stream write + host spin", "kernel writes flag + host spin"};
				std::printf("%-32s %.2f us per round (launch + ~kernel + completion)\n", names[mode], us(t0, t1) / rounds);
			}
		}
	}
	// the kernel alone (events on the stream)
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipEventRecord(a, s);
	for (int r = 0; r < 200; ++r)
	{
		hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s, d_out, iters, d_counter, nullptr, 0u, blocks);
	}
	hipEventRecord(b, s);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	std::printf("kernel back to back: %.2f us per launch\n", ms * 1000.0f / 200);
	return 0;
}
