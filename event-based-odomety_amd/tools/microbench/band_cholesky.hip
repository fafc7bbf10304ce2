// band_cholesky.hip — what would the LM step of the reference's default problem cost ON the device?
// (round 4, review item 6: a one-launch solve of edge + TV needs the 216-unknown banded Cholesky of csrc/host_lm.cpp
// between two evaluations; on a host core of the GPU box the whole LM step is 9.5 us per round.)
// One workgroup factors A = L L' (n = 216, band = 25: 108 patches x 2 flows, TV couples a patch with its right and
// lower neighbour) and solves L L' x = b, from LDS, in the RIGHT-looking order: once column k is final every entry
// (i, j) of the 25 x 25 triangle behind it takes its term -L(i,k) L(j,k) — per entry the terms arrive in ascending
// k with one rounding per operation, i.e. the bits of host_lm.cpp's chains (compared below with a host factor).
// Timed with the shader clock inside the kernel (factor, forward + backward substitution) and with HIP events
// around 200 launches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off band_cholesky.hip -o band_cholesky && ./band_cholesky
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int N = 216;
constexpr int BAND = 25;
constexpr int PITCH = BAND + 1;  // column j: entries (j .. j+BAND, j) at [j * PITCH + (i - j)]

template <int LANES>
__global__ __launch_bounds__(LANES) void k_band_solve(const double* __restrict__ A0, const double* __restrict__ b0,
													  double* __restrict__ Lout, double* __restrict__ xout,
													  unsigned long long* __restrict__ clocks)
{
	__shared__ double A[(N + BAND + 1) * PITCH];
	__shared__ double x[N];
	const int t = threadIdx.x;
	for (int i = t; i < (N + BAND + 1) * PITCH; i += LANES)
	{
		A[i] = i < N * PITCH ? A0[i] : 0.0;
	}
	for (int i = t; i < N; i += LANES)
	{
		x[i] = b0[i];
	}
	__syncthreads();
	const unsigned long long c0 = clock64();
	// the triangle behind a pivot column: entry e -> (r, c), 1 <= c <= r <= BAND, fixed per lane
	constexpr int TRI = BAND * (BAND + 1) / 2;
	constexpr int PER = (TRI + LANES - 1) / LANES;
	int er[PER], ec[PER];
#pragma unroll
	for (int q = 0; q < PER; ++q)
	{
		const int e = t + q * LANES;
		int r = 1, base = 0;
		while (e < TRI && base + r <= e)
		{
			base += r;
			++r;
		}
		er[q] = e < TRI ? r : 0;
		ec[q] = e < TRI ? e - base + 1 : 0;
	}
	for (int k = 0; k < N; ++k)
	{
		double* const Ck = A + k * PITCH;
		const double d = Ck[0];
		const double l = sqrt(d);
		if (t >= 1 && t <= BAND)
		{
			Ck[t] = Ck[t] / l;
		}
		if (t == 0)
		{
			Ck[0] = l;
		}
		__syncthreads();
#pragma unroll
		for (int q = 0; q < PER; ++q)
		{
			if (er[q])
			{
				// entry (k + r, k + c) lives in column k + c at offset r - c
				double* const p = A + (k + ec[q]) * PITCH + (er[q] - ec[q]);
				const double prod = Ck[er[q]] * Ck[ec[q]];
				*p = *p - prod;
			}
		}
		__syncthreads();
	}
	const unsigned long long c1 = clock64();
	// forward substitution by columns, backward by rows (host_lm.cpp's order), one wave's worth of lanes at most
	for (int k = 0; k < N; ++k)
	{
		const double xk = x[k] / A[k * PITCH];
		__syncthreads();
		if (t == 0)
		{
			x[k] = xk;
		}
		if (t >= 1 && t <= BAND && k + t < N)
		{
			const double prod = A[k * PITCH + t] * xk;
			x[k + t] = x[k + t] - prod;
		}
		__syncthreads();
	}
	for (int i = N - 1; i >= 0; --i)
	{
		if (t == 0)
		{
			double s = x[i];
			for (int c = 1; c <= BAND && i + c < N; ++c)
			{
				const double prod = A[i * PITCH + c] * x[i + c];
				s = s - prod;
			}
			x[i] = s / A[i * PITCH];
		}
		__syncthreads();
	}
	const unsigned long long c2 = clock64();
	for (int i = t; i < N * PITCH; i += LANES)
	{
		Lout[i] = A[i];
	}
	for (int i = t; i < N; i += LANES)
	{
		xout[i] = x[i];
	}
	if (t == 0)
	{
		clocks[0] = c1 - c0;
		clocks[1] = c2 - c1;
	}
}

// The factor's critical path with nothing around it: pivot -> sqrt -> divide -> multiply -> subtract -> next pivot,
// N times, in registers on one lane (no LDS, no barrier): no arrangement of the factorisation is shorter than this.
__global__ void k_chain(double a, double b, double c, double* out, unsigned long long* clocks)
{
	const unsigned long long c0 = clock64();
	double x = a;
	for (int k = 0; k < N; ++k)
	{
		const double l = sqrt(x);
		const double y = b / l;
		const double prod = y * y;
		x = c - prod;
	}
	const unsigned long long c1 = clock64();
	out[threadIdx.x] = x;
	clocks[0] = c1 - c0;
}

static void host_factor(std::vector<double>& A)
{
	// the left-looking chains of csrc/host_lm.cpp (scalar form), entries beyond the matrix are zero
	for (int j = 0; j < N; ++j)
	{
		for (int c = 0; c <= BAND && j + c < N; ++c)
		{
			double v = A[j * PITCH + c];
			for (int k = std::max(0, j + c - BAND); k < j; ++k)
			{
				const double prod = A[k * PITCH + (j + c - k)] * A[k * PITCH + (j - k)];
				v = v - prod;
			}
			A[j * PITCH + c] = v;
		}
		const double l = std::sqrt(A[j * PITCH]);
		A[j * PITCH] = l;
		for (int c = 1; c <= BAND && j + c < N; ++c)
		{
			A[j * PITCH + c] = A[j * PITCH + c] / l;
		}
	}
}

template <int LANES>
static void run(const std::vector<double>& hA, const std::vector<double>& hb, const std::vector<double>& ref)
{
	double *dA, *db, *dL, *dx;
	unsigned long long* dclk;
	hipMalloc(&dA, hA.size() * 8);
	hipMalloc(&db, hb.size() * 8);
	hipMalloc(&dL, hA.size() * 8);
	hipMalloc(&dx, hb.size() * 8);
	hipMalloc(&dclk, 16);
	hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
	hipMemcpy(db, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 20; ++i)
	{
		hipLaunchKernelGGL(k_band_solve<LANES>, dim3(1), dim3(LANES), 0, 0, dA, db, dL, dx, dclk);
	}
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	const int reps = 200;
	for (int i = 0; i < reps; ++i)
	{
		hipLaunchKernelGGL(k_band_solve<LANES>, dim3(1), dim3(LANES), 0, 0, dA, db, dL, dx, dclk);
	}
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	std::vector<double> L(hA.size());
	unsigned long long clk[2];
	hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost);
	hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
	size_t diff = 0;
	for (int j = 0; j < N; ++j)
	{
		for (int c = 0; c <= BAND && j + c < N; ++c)
		{
			diff += std::memcmp(&L[j * PITCH + c], &ref[j * PITCH + c], 8) != 0;
		}
	}
	const double usPerClk = 1e3 * ms / reps / static_cast<double>(clk[0] + clk[1]);  // (load + store are a few us of the launch)
	std::printf("%4d lanes: %.1f us per launch (events); factor %llu shader clocks (~%.0f us), two substitutions %llu "
				"(~%.0f us); factor entries that differ from the host's bits: %zu\n",
				LANES, 1e3 * ms / reps, clk[0], clk[0] * usPerClk, clk[1], clk[1] * usPerClk, diff);
	hipFree(dA);
	hipFree(db);
	hipFree(dL);
	hipFree(dx);
	hipFree(dclk);
}

int main()
{
	// J'J + D'D of the reference default: 2 x 2 data blocks on the diagonal, TV terms to the right and the lower
	// neighbour (12 x 9 patches), a damping diagonal
	std::vector<double> A(N * PITCH, 0.0), b(N);
	uint64_t s = 0x9E3779B97F4A7C15ull;
	auto rnd = [&] {
		s ^= s << 13;
		s ^= s >> 7;
		s ^= s << 17;
		return (s >> 11) * (1.0 / 9007199254740992.0);
	};
	const int npx = 12, npy = 9;
	for (int p = 0; p < npx * npy; ++p)
	{
		const double a = 50 + 100 * rnd(), c = 50 + 100 * rnd(), o = 20 * (rnd() - 0.5);
		A[(2 * p) * PITCH] += a + 1.0;
		A[(2 * p + 1) * PITCH] += c + 1.0;
		A[(2 * p) * PITCH + 1] += o;
		for (int nb : {p % npx + 1 < npx ? p + 1 : -1, p / npx + 1 < npy ? p + npx : -1})
		{
			if (nb < 0)
			{
				continue;
			}
			const double w = 1e3 * (0.2 + rnd());
			for (int k = 0; k < 2; ++k)
			{
				A[(2 * p + k) * PITCH] += w;
				A[(2 * nb + k) * PITCH] += w;
				A[(2 * p + k) * PITCH + 2 * (nb - p)] -= w;
			}
		}
	}
	for (auto& v : b)
	{
		v = rnd() - 0.5;
	}
	std::vector<double> ref = A;
	host_factor(ref);
	{
		double* dout;
		unsigned long long *dclk, clk = 0;
		hipMalloc(&dout, 64 * 8);
		hipMalloc(&dclk, 8);
		for (int i = 0; i < 3; ++i)
		{
			hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, 4.0, 1.0, 4.5, dout, dclk);
		}
		hipDeviceSynchronize();
		hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
		std::printf("critical path alone (216 x sqrt -> divide -> multiply -> subtract in registers): %llu shader clocks = %.0f per column\n",
					clk, static_cast<double>(clk) / N);
	}
	run<64>(A, b, ref);
	run<128>(A, b, ref);
	run<256>(A, b, ref);
	run<512>(A, b, ref);
	return 0;
}
