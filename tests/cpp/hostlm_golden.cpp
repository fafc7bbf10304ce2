// hostlm_golden.cpp — the bits of HostLm's trajectories, pinned.
//
// 200 solves of a 12 x 9 grid (the reference's default geometry) on smooth synthetic data terms; the flows of
// every solve are folded into one checksum.  The value below was produced by the row-by-row scalar banded
// Cholesky of rounds 1-2; round 3's vectorised column form (four chains per vector register, AVX2 or SSE2) is
// meant to perform the same operations in the same order per entry, and this test holds it to that: the same
// checksum with either instruction set (EBO_LM_NO_AVX2=1 selects the plain build at run time).
// Test infrastructure; the data terms need exp / sin / cos of the C library this image ships.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../event-based-odomety_amd/csrc/host_lm.h"

int main()
{
	ebo_solver_opts o;
	std::memset(&o, 0, sizeof(o));
	o.max_num_iterations = 50;
	o.use_nonmonotonic = 1;
	o.function_tolerance = 1e-6;
	o.gradient_tolerance = 1e-10;
	o.parameter_tolerance = 1e-8;
	o.initial_radius = 1e4;
	o.max_radius = 1e16;
	o.min_radius = 1e-32;
	o.min_relative_decrease = 1e-3;
	o.min_lm_diagonal = 1e-6;
	o.max_lm_diagonal = 1e32;
	o.max_consecutive_nonmonotonic = 5;
	o.max_consecutive_invalid = 5;
	o.jacobi_scaling = 1;
	o.mode = EBO_SOLVE_GLOBAL;
	const int npx = 12, npy = 9, P = npx * npy;
	unsigned long long sum = 0;
	long supplies = 0;
	for (int rep = 0; rep < 200; ++rep)
	{
		std::vector<uint8_t> active(P, 1);
		for (int k = 0; k < 11; ++k)
		{
			active[(k * 17 + rep) % P] = 0;
		}
		ebo::HostLm lm(npx, npy, active, 1e3, 10.0, o);
		std::vector<double> x(2 * P), r(P), J(2 * P);
		for (;;)
		{
			const ebo::HostLm::Request q = lm.request(x.data());
			if (q == ebo::HostLm::DONE)
			{
				break;
			}
			for (int p = 0; p < P; ++p)
			{
				const double t0 = 0.4 * std::sin(0.37 * p + 0.2 + rep), t1 = 0.3 * std::cos(0.91 * p);
				const double d0 = x[2 * p] - t0, d1 = x[2 * p + 1] - t1;
				const double e = (40.0 + 10.0 * std::sin(1.3 * p)) * std::exp(-0.5 * (d0 * d0 + d1 * d1));
				r[p] = 1000.0 - e;
				J[2 * p] = e * d0;
				J[2 * p + 1] = e * d1;
			}
			lm.supply(r.data(), q == ebo::HostLm::NEED_JACOBIAN ? J.data() : nullptr);
			++supplies;
		}
		lm.result(x.data());
		for (double v : x)
		{
			unsigned long long b;
			std::memcpy(&b, &v, 8);
			sum = sum * 1315423911ull + b;
		}
	}
	const unsigned long long golden = 0xff189a6ad5aea9b2ull;
	std::printf("hostlm_golden: %ld supplies, checksum %016llx (%s)\n", supplies, sum, sum == golden ? "as pinned" : "DIFFERENT");
	return sum == golden ? 0 : 1;
}
