// solve_edge_outlined.hip — CPU-only probe (hipcc -S, never linked, never run): the SHAPE of the
// first k_solve_edge that faulted on the device in round 2: the edge objective
// reached from three sites of the per-unit LM, so that the compiler keeps ONE out-of-line copy of it
// and every site passes the workgroup's LDS arrays as generic pointers.  `make check-kernels` must
// reject this file's kernel; tools/check_kernels.py --expect-fail builds it and asserts that.
#include "../../csrc/ebo_kernels.hip"

namespace ebo
{
namespace
{
__device__ __noinline__ void edge_eval_outlined(const uint64_t* ev, const Unit& u, double m0, double m1, bool wantJac,
												 int capPx, char* slice, double* lds, const EvalConsts& c,
												 const EdgeConsts& ec, double2* csUnit, double& r, double& j0, double& j1)
{
	edge_eval_point<true, true>(ev, u, m0, m1, wantJac, capPx, slice, lds, c, ec, csUnit, r, j0, j1);
}

__global__ void __launch_bounds__(1024) k_solve_edge_outlined(const uint64_t* __restrict__ events,
															   const Unit* __restrict__ units, int capPx,
															   char* __restrict__ scratch, size_t scratchStride,
															   double* __restrict__ flowsOut, EvalConsts c, EdgeConsts ec,
															   SolveConsts o)
{
	extern __shared__ double lds[];
	const Unit u = units[blockIdx.x];
	const uint64_t* ev = events + u.ev_off;
	char* slice = scratch + static_cast<size_t>(blockIdx.x) * scratchStride;
	double2* csUnit = nullptr;
	double x0 = 0.0, x1 = 0.0, r, a, b;
	// site 1: the first evaluation
	edge_eval_outlined(ev, u, x0, x1, true, capPx, slice, lds, c, ec, csUnit, r, a, b);
	double cost = 0.5 * r * r, radius = o.initial_radius;
	for (int it = 0; it < o.max_num_iterations; ++it)
	{
		const double d0 = a * a + 1.0 / radius, d1 = b * b + 1.0 / radius;
		const double c0 = x0 - a * r / d0, c1 = x1 - b * r / d1;
		double rc, ac, bc;
		__syncthreads();
		// site 2: the cost at the candidate
		edge_eval_outlined(ev, u, c0, c1, false, capPx, slice, lds, c, ec, csUnit, rc, ac, bc);
		if (0.5 * rc * rc < cost)
		{
			x0 = c0;
			x1 = c1;
			__syncthreads();
			// site 3: value and Jacobian at the accepted point
			edge_eval_outlined(ev, u, x0, x1, true, capPx, slice, lds, c, ec, csUnit, r, a, b);
			cost = 0.5 * r * r;
			radius *= 3.0;
		}
		else
		{
			radius *= 0.5;
		}
	}
	if (threadIdx.x == 0)
	{
		flowsOut[2 * u.flow_idx] = x0;
		flowsOut[2 * u.flow_idx + 1] = x1;
	}
}
}  // namespace
}  // namespace ebo
