// tensor_mfma.hip — can the matrix pipe take the edge loss's 7x7 structure-tensor filter?  (round 4, item 4)
//
// The filter is separable: S_c(y, x) = N sum_i e(i) sum_j e(j) p_c(y + i, x + j) for the three products p_c of the
// forward differences.  As banded-Toeplitz products on v_mfma_f64_16x16x4_f64, for a tile of 16 columns x 10 rows:
//   pass 1 (horizontal)  H[16 rows][16 cols] = P[16 rows][24 cols] . T[24][16]      6 MFMA per channel
//   pass 2 (vertical)    S[16 rows][16 cols] = V[16][16 rows] . H[16 rows][16 cols]  4 MFMA per channel
// and pass 2 needs NO lane movement: the f64 C/D layout (col = lane & 15, row = (lane >> 4) + 4 reg) puts row 4 s + g
// of H in register s of lane group g, which is exactly the B operand of k-step s.
// This program (a) checks that formulation against a plain double loop, bit-tolerance 1e-13 relative, (b) times it:
// cycles per MFMA back to back, cycles per tile with operand preparation from an LDS image, per workgroup of 4 waves,
// for the 41 x 41 eigenvalue region of a reference-default unit (3 x 5 tiles), next to a VALU-only kernel doing the
// same region the way tensor_runs does (8-row register runs).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tensor_mfma.hip -o tensor_mfma && ./tensor_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kN = 64;       // canvas 64 x 64 (the reference's is 60 x 60)
constexpr int kPitch = 65;   // odd pitch, as the product
constexpr int kR0 = 8, kRegion = 41;  // eigenvalue region [8, 49) x [8, 49)

__device__ __forceinline__ double tapw(int d, double hs) { return (d >= -3 && d <= 3) ? exp(hs * d * d) : 0.0; }

// one 16 x 10 tile at (ty0, tx0): returns the lane's four (row = g + 4 r, col = n) tensor sums per channel
__device__ __forceinline__ void tile_mfma(const double* __restrict__ I, int ty0, int tx0, const double (&T)[6], const double (&V)[4],
										  double4_t (&S)[3])
{
	const int lane = threadIdx.x & 63;
	const int n = lane & 15, g = lane >> 4;
	double4_t H[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	// pass 1: A = P (m = lane & 15 -> image row ty0 - 3 + m; k = 4 s + g -> image column tx0 - 3 + 4 s + g), B = T
	const double* row = I + (ty0 - 3 + n) * kPitch + (tx0 - 3 + g);
#pragma unroll
	for (int s = 0; s < 6; ++s)
	{
		const double c0 = row[4 * s], cx = row[4 * s + 1], cy = row[4 * s + kPitch];
		const double gx = cx - c0, gy = cy - c0;
		H[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(gx * gx, T[s], H[0], 0, 0, 0);
		H[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(gx * gy, T[s], H[1], 0, 0, 0);
		H[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(gy * gy, T[s], H[2], 0, 0, 0);
	}
	// pass 2: A = V (m = output row, k = 4 s + g = H row), B = register s of H: row 4 s + g, column n -- in place
#pragma unroll
	for (int c = 0; c < 3; ++c)
	{
		double4_t acc = {0, 0, 0, 0};
#pragma unroll
		for (int s = 0; s < 4; ++s)
		{
			acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V[s], H[c][s], acc, 0, 0, 0);
		}
		S[c] = acc;
	}
}

__global__ void __launch_bounds__(256) k_tensor_mfma(const double* __restrict__ image, double* __restrict__ E, long long* __restrict__ clk,
													   int reps, double hs)
{
	__shared__ double I[kN * kPitch];
	for (int p = threadIdx.x; p < kN * kN; p += blockDim.x)
	{
		I[(p / kN) * kPitch + p % kN] = image[p];
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int n = lane & 15, g = lane >> 4;
	double T[6], V[4];
#pragma unroll
	for (int s = 0; s < 6; ++s)
	{
		T[s] = tapw(4 * s + g - 3 - n, hs);  // T[k = 4 s + g][n]: input column k - 3 against output column n
	}
#pragma unroll
	for (int s = 0; s < 4; ++s)
	{
		V[s] = tapw(4 * s + g - 3 - n, hs);  // V[m = n][k = 4 s + g]: H row k (image row ty0 - 3 + k) against output row m
	}
	__syncthreads();
	const long long t0 = clock64();
	for (int rep = 0; rep < reps; ++rep)
	{
		// 3 x 5 tiles of 16 columns x 10 rows over the 41 x 41 region, dealt to the four waves
		for (int t = wave; t < 15; t += 4)
		{
			const int tx0 = kR0 + 16 * (t % 3), ty0 = kR0 + 10 * (t / 3);
			double4_t S[3];
			tile_mfma(I, ty0, tx0, T, V, S);
#pragma unroll
			for (int r = 0; r < 4; ++r)
			{
				const int m = g + 4 * r, y = ty0 + m, x = tx0 + n;
				if (m < 10 && y < kR0 + kRegion && x < kR0 + kRegion)
				{
					const double s00 = S[0][r], s01 = S[1][r], s11 = S[2][r];
					const double tr = s00 + s11, det = s00 * s11 - s01 * s01;
					E[y * kN + x] = 0.5 * (tr + sqrt(tr * tr - 4.0 * det));
				}
			}
		}
	}
	const long long t1 = clock64();
	if (threadIdx.x == 0)
	{
		clk[blockIdx.x] = t1 - t0;
	}
}

// the matrix pipe alone: `n` dependent-free MFMAs per wave (four accumulators)
__global__ void __launch_bounds__(256) k_mfma_rate(double* __restrict__ sink, long long* __restrict__ clk, int n)
{
	double4_t a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	const double x = 1.0 + threadIdx.x * 1e-3, y = 0.5;
	const long long t0 = clock64();
	for (int i = 0; i < n; ++i)
	{
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			a[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[k], 0, 0, 0);
		}
	}
	const long long t1 = clock64();
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a[0][0] + a[1][1] + a[2][2] + a[3][3];
	if (threadIdx.x == 0)
	{
		clk[blockIdx.x] = t1 - t0;
	}
}

// the same region the way the product's tensor_runs does it: a lane owns a vertical run of 8 output rows of one column,
// slides down 14 rows keeping seven horizontally filtered rows in registers (general structure, not the tuned code)
__global__ void __launch_bounds__(256) k_tensor_valu(const double* __restrict__ image, double* __restrict__ E, long long* __restrict__ clk,
													   int reps, double hs)
{
	__shared__ double I[kN * kPitch];
	for (int p = threadIdx.x; p < kN * kN; p += blockDim.x)
	{
		I[(p / kN) * kPitch + p % kN] = image[p];
	}
	double e[4];
	for (int k = 0; k < 4; ++k)
	{
		e[k] = exp(hs * k * k);
	}
	__syncthreads();
	const long long t0 = clock64();
	for (int rep = 0; rep < reps; ++rep)
	{
		const int runsPerCol = (kRegion + 7) / 8;
		for (int task = threadIdx.x; task < kRegion * runsPerCol; task += blockDim.x)
		{
			const int x = kR0 + task % kRegion, y0 = kR0 + 8 * (task / kRegion);
			double h[7][3];
			for (int step = 0; step < 14; ++step)
			{
				const int yy = y0 - 3 + step;
				const double* row = I + yy * kPitch + x - 3;
				double h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
				for (int j = 0; j < 7; ++j)
				{
					const double c0 = row[j], gx = row[j + 1] - c0, gy = row[j + kPitch] - c0;
					const double w = e[j < 3 ? 3 - j : j - 3];
					h0 = fma(w, gx * gx, h0);
					h1 = fma(w, gx * gy, h1);
					h2 = fma(w, gy * gy, h2);
				}
#pragma unroll
				for (int k = 0; k < 6; ++k)
				{
					h[k][0] = h[k + 1][0];
					h[k][1] = h[k + 1][1];
					h[k][2] = h[k + 1][2];
				}
				h[6][0] = h0;
				h[6][1] = h1;
				h[6][2] = h2;
				if (step >= 6)
				{
					double s00 = 0, s01 = 0, s11 = 0;
#pragma unroll
					for (int k = 0; k < 7; ++k)
					{
						const double w = e[k < 3 ? 3 - k : k - 3];
						s00 = fma(w, h[k][0], s00);
						s01 = fma(w, h[k][1], s01);
						s11 = fma(w, h[k][2], s11);
					}
					const int y = y0 + step - 6;
					if (y < kR0 + kRegion)
					{
						const double tr = s00 + s11, det = s00 * s11 - s01 * s01;
						E[y * kN + x] = 0.5 * (tr + sqrt(tr * tr - 4.0 * det));
					}
				}
			}
		}
	}
	const long long t1 = clock64();
	if (threadIdx.x == 0)
	{
		clk[blockIdx.x] = t1 - t0;
	}
}

// (c) what sharing through LDS would buy if the LDS were there: products of the differences once per pixel, horizontally
// filtered rows once per pixel (both staged in LDS: 24 + 24 B per pixel of the region and its halo), then the vertical pass.
// 133 KB of LDS for one 64 x 64 canvas: NOT available to k_eval_edge (its two-per-CU layout has 77 KB per unit in all).
__global__ void __launch_bounds__(256) k_tensor_shared(const double* __restrict__ image, double* __restrict__ E, long long* __restrict__ clk,
														 int reps, double hs)
{
	extern __shared__ double sh[];
	double* I = sh;                         // kN * kPitch
	constexpr int kH = kRegion + 6;         // rows / columns of the halo region
	double* P = I + kN * kPitch;            // 3 x kH x (kH + 1)
	double* Hh = P + 3 * kH * (kH + 1);     // 3 x kH x kRegion
	for (int p = threadIdx.x; p < kN * kN; p += blockDim.x)
	{
		I[(p / kN) * kPitch + p % kN] = image[p];
	}
	double e[4];
	for (int k = 0; k < 4; ++k)
	{
		e[k] = exp(hs * k * k);
	}
	__syncthreads();
	const long long t0 = clock64();
	for (int rep = 0; rep < reps; ++rep)
	{
		for (int k = threadIdx.x; k < kH * kH; k += blockDim.x)
		{
			const int yy = kR0 - 3 + k / kH, xx = kR0 - 3 + k % kH;
			const double c0 = I[yy * kPitch + xx], gx = I[yy * kPitch + xx + 1] - c0, gy = I[(yy + 1) * kPitch + xx] - c0;
			double* q = P + (k / kH) * (kH + 1) + k % kH;
			q[0] = gx * gx;
			q[kH * (kH + 1)] = gx * gy;
			q[2 * kH * (kH + 1)] = gy * gy;
		}
		__syncthreads();
		for (int k = threadIdx.x; k < kH * kRegion; k += blockDim.x)
		{
			const int r = k / kRegion, x = k % kRegion;
#pragma unroll
			for (int c = 0; c < 3; ++c)
			{
				const double* q = P + c * kH * (kH + 1) + r * (kH + 1) + x;
				double h = e[0] * q[3];
				h = fma(e[1], q[2] + q[4], h);
				h = fma(e[2], q[1] + q[5], h);
				h = fma(e[3], q[0] + q[6], h);
				Hh[c * kH * kRegion + k] = h;
			}
		}
		__syncthreads();
		for (int k = threadIdx.x; k < kRegion * kRegion; k += blockDim.x)
		{
			const int y = k / kRegion, x = k % kRegion;
			double s[3];
#pragma unroll
			for (int c = 0; c < 3; ++c)
			{
				const double* q = Hh + c * kH * kRegion + y * kRegion + x;
				double v = e[0] * q[3 * kRegion];
				v = fma(e[1], q[2 * kRegion] + q[4 * kRegion], v);
				v = fma(e[2], q[1 * kRegion] + q[5 * kRegion], v);
				v = fma(e[3], q[0] + q[6 * kRegion], v);
				s[c] = v;
			}
			const double tr = s[0] + s[2], det = s[0] * s[2] - s[1] * s[1];
			E[(kR0 + y) * kN + kR0 + x] = 0.5 * (tr + sqrt(tr * tr - 4.0 * det));
		}
		__syncthreads();
	}
	const long long t1 = clock64();
	if (threadIdx.x == 0)
	{
		clk[blockIdx.x] = t1 - t0;
	}
}

int main()
{
	const double hs = -0.5 / (1.5 * 1.5);
	std::vector<double> img(kN * kN);
	unsigned s = 12345;
	for (double& v : img)
	{
		s = s * 1664525u + 1013904223u;
		v = (s >> 8) * (1.0 / 16777216.0);
	}
	// host reference of the region
	std::vector<double> ref(kN * kN, 0.0);
	for (int y = kR0; y < kR0 + kRegion; ++y)
	{
		for (int x = kR0; x < kR0 + kRegion; ++x)
		{
			double s00 = 0, s01 = 0, s11 = 0;
			for (int i = -3; i <= 3; ++i)
			{
				for (int j = -3; j <= 3; ++j)
				{
					const double c0 = img[(y + i) * kN + x + j], gx = img[(y + i) * kN + x + j + 1] - c0, gy = img[(y + i + 1) * kN + x + j] - c0;
					const double w = std::exp(hs * i * i) * std::exp(hs * j * j);
					s00 += w * gx * gx;
					s01 += w * gx * gy;
					s11 += w * gy * gy;
				}
			}
			const double tr = s00 + s11, det = s00 * s11 - s01 * s01;
			ref[y * kN + x] = 0.5 * (tr + std::sqrt(tr * tr - 4.0 * det));
		}
	}
	double *d_img, *d_E, *d_sink;
	long long* d_clk;
	const int blocks = 512;  // two per CU, as the product's 256-lane layout
	hipMalloc(&d_img, img.size() * 8);
	hipMalloc(&d_E, img.size() * 8);
	hipMalloc(&d_sink, blocks * 256 * 8);
	hipMalloc(&d_clk, blocks * 8);
	hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice);
	std::vector<long long> clk(blocks);
	std::vector<double> got(kN * kN);
	auto report = [&](const char* name, int reps) {
		hipMemcpy(clk.data(), d_clk, blocks * 8, hipMemcpyDeviceToHost);
		double mean = 0;
		for (long long c : clk) mean += static_cast<double>(c);
		mean /= blocks * reps;
		hipMemcpy(got.data(), d_E, got.size() * 8, hipMemcpyDeviceToHost);
		double worst = 0;
		for (int y = kR0; y < kR0 + kRegion; ++y)
			for (int x = kR0; x < kR0 + kRegion; ++x)
				worst = std::fmax(worst, std::fabs(got[y * kN + x] - ref[y * kN + x]) / std::fabs(ref[y * kN + x]));
		std::printf("%-14s %8.0f cycles per unit (41 x 41 region, 512 workgroups of 256 lanes), max rel err %.2e\n", name, mean, worst);
	};
	for (int pass = 0; pass < 2; ++pass)
	{
		hipMemset(d_E, 0, img.size() * 8);
		hipLaunchKernelGGL(k_tensor_mfma, dim3(blocks), dim3(256), 0, 0, d_img, d_E, d_clk, 20, hs);
		hipDeviceSynchronize();
		if (pass) report("MFMA Toeplitz", 20);
		hipMemset(d_E, 0, img.size() * 8);
		hipLaunchKernelGGL(k_tensor_valu, dim3(blocks), dim3(256), 0, 0, d_img, d_E, d_clk, 20, hs);
		hipDeviceSynchronize();
		if (pass) report("VALU 8-row runs", 20);
		{
			constexpr int kH = kRegion + 6;
			const size_t lds = (kN * kPitch + 3 * kH * (kH + 1) + 3 * kH * kRegion) * sizeof(double);
			hipFuncSetAttribute(reinterpret_cast<const void*>(k_tensor_shared), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
			hipMemset(d_E, 0, img.size() * 8);
			hipLaunchKernelGGL(k_tensor_shared, dim3(blocks), dim3(256), lds, 0, d_img, d_E, d_clk, 20, hs);
			hipDeviceSynchronize();
			if (pass) report("VALU, LDS-shared", 20);
		}
	}
	hipLaunchKernelGGL(k_mfma_rate, dim3(blocks), dim3(256), 0, 0, d_sink, d_clk, 1000);
	hipDeviceSynchronize();
	hipMemcpy(clk.data(), d_clk, blocks * 8, hipMemcpyDeviceToHost);
	double mean = 0;
	for (long long c : clk) mean += static_cast<double>(c);
	std::printf("v_mfma_f64_16x16x4_f64 back to back: %.1f cycles per instruction per wave (one wave per SIMD, every CU busy)\n",
				mean / blocks / 4000.0);
	return 0;
}
