// field_tv.cpp — see field_tv.h.  The iteration is Ceres 2.0's TrustRegionMinimizer with
// LevenbergMarquardtStrategy and TrustRegionStepEvaluator as published, specialised to the
// problem of feature_detector.cpp:154-214; the linear solve (SPARSE_NORMAL_CHOLESKY in the
// reference, :221) is a multigrid-preconditioned conjugate-gradient run to a relative residual
// of 1e-13 on the device, i.e. to the accuracy a direct factorisation delivers.
#include "field_tv.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace ebo
{
namespace
{
struct Scal
{
	double v[32];
};

int read_scal(const TvfArgs& A, hipStream_t s, Scal& out, std::string* err)
{
	hipError_t e = hipMemcpyAsync(out.v, A.scal, sizeof(out.v), hipMemcpyDeviceToHost, s);
	if (e == hipSuccess)
	{
		e = hipStreamSynchronize(s);
	}
	if (e != hipSuccess)
	{
		*err = std::string("field TV: ") + hipGetErrorString(e);
		return EBO_ERR_HIP;
	}
	return EBO_OK;
}

// indices into TvfArgs::scal (ebo_fieldtv.inc)
enum
{
	kRr = 6,
	kBb = 8,
	kCost = 10,
	kXnorm2 = 11,
	kGmax = 12,
	kYg = 13,
	kYLy = 14,
	kYnorm2 = 15,
	kNorm = 16,
	kCandCost = 20  // cost of the candidate (cost-only linearisation)
};
}  // namespace

int field_tv_solve(int w, int h, float* d_field, const int* d_fixed, int n_fixed, bool use_l1,
				   const ebo_solver_opts& o, void* workspace, void* stream, FieldTvStats* stats,
				   std::string* err)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	FieldTvStats st;
	TvfArgs A;
	TvfMg M;
	std::memset(&A, 0, sizeof(A));
	std::memset(&M, 0, sizeof(M));
	double2* xbest = nullptr;
	tvf_carve(A, M, w, h, workspace, &xbest);
	{
		// EBO_TVF_PRECOND=jacobi: the diagonal preconditioner (A/B, ~30x more CG iterations)
		const char* pre = ab_env("EBO_TVF_PRECOND");
		if (pre && std::strcmp(pre, "jacobi") == 0)
		{
			M.levels = 0;
		}
	}
	// iterations between two looks at the residual: multigrid converges in a few tens
	const char* chunkEnv = ab_env("EBO_TVF_CG_CHUNK");
	const int cgChunk = chunkEnv ? std::max(2, std::atoi(chunkEnv) & ~1) : (M.levels >= 2 ? 8 : 32);
	A.huber_a = use_l1 ? 1e-5 : 0.0;  // feature_detector.cpp:182,187
	A.lm_lo = o.min_lm_diagonal;
	A.lm_hi = o.max_lm_diagonal;
	const size_t vecBytes = static_cast<size_t>(A.n) * sizeof(double2);
	const char* tolEnv = ab_env("EBO_TVF_CG_TOL");
	const double cgTol = tolEnv ? std::atof(tolEnv) : 1e-13;
	const char* capEnv = ab_env("EBO_TVF_CG_MAX");
	const int cgMax = capEnv ? std::atoi(capEnv) : 200000;

	auto fail = [&](const char* what) {
		*err = std::string("field TV: ") + what;
		return EBO_ERR_HIP;
	};
	Scal sc;
	if (launch_tvf_prepare(A, d_field, d_fixed, n_fixed, s))
	{
		return fail("prepare launch");
	}
	int rc = read_scal(A, s, sc, err);
	if (rc)
	{
		return rc;
	}
	if (!(sc.v[kNorm] > 0.0))  // feature_detector.cpp:152
	{
		*stats = st;
		return EBO_OK;
	}
	st.smoothed = 1;

	auto linearize = [&](const double2* X, int first, int costOnly) -> int {
		if (launch_tvf_linearize(A, X, first, costOnly, s))
		{
			return fail("linearize launch");
		}
		if (costOnly)
		{
			st.evals_cost++;
		}
		else
		{
			st.evals_jac++;
		}
		return read_scal(A, s, sc, err);
	};

	// One chunk of CG iterations as a graph: iterations after the first chunk differ only
	// in the parity of their index (which direction buffer is old), and a chunk is even.
	// The kernels take every pointer from A, which is fixed for this call.
	struct GraphHolder
	{
		hipGraph_t g = nullptr;
		hipGraphExec_t e = nullptr;
		~GraphHolder()
		{
			if (e)
			{
				hipGraphExecDestroy(e);
			}
			if (g)
			{
				hipGraphDestroy(g);
			}
		}
	} gh;
	hipGraphExec_t chunkGraph = nullptr;
	const char* graphEnv = ab_env("EBO_TVF_GRAPH");
	if (!graphEnv || std::atoi(graphEnv) != 0)
	{
		if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess)
		{
			const int lrc = launch_tvf_cg_iters(A, M, cgChunk, cgChunk, s);
			const hipError_t ce = hipStreamEndCapture(s, &gh.g);
			if (lrc == 0 && ce == hipSuccess && gh.g &&
				hipGraphInstantiate(&gh.e, gh.g, nullptr, nullptr, 0) == hipSuccess)
			{
				chunkGraph = gh.e;
			}
		}
		(void)hipGetLastError();  // a failed capture falls back to plain launches
	}

	// Iteration zero.
	rc = linearize(A.x, o.jacobi_scaling ? 1 : 2, 0);
	if (rc)
	{
		return rc;
	}
	double xCost = sc.v[kCost];
	if (!std::isfinite(xCost))
	{
		st.termination = 2;
		*stats = st;
		return EBO_OK;
	}
	st.initial_cost = xCost;
	double xNorm = std::sqrt(sc.v[kXnorm2]);
	double gradMax = sc.v[kGmax];
	double minimumCost = xCost;
	if (hipMemcpyAsync(xbest, A.x, vecBytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
	{
		return fail("copy");
	}

	const int maxNonmono = o.use_nonmonotonic ? o.max_consecutive_nonmonotonic : 0;
	double seMinimum = xCost, seCurrent = xCost, seReference = xCost, seCandidate = xCost;
	double seAccRef = 0.0, seAccCand = 0.0;
	int seNumNonmono = 0;
	double radius = o.initial_radius;
	double decreaseFactor = 2.0;
	int iteration = 0, numInvalid = 0;
	bool lastSuccessful = true;
	int termination = 1;

	for (;;)
	{
		if (lastSuccessful && xCost < minimumCost)
		{
			minimumCost = xCost;
			if (hipMemcpyAsync(xbest, A.x, vecBytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
			{
				return fail("copy");
			}
		}
		if (iteration >= o.max_num_iterations)
		{
			termination = 1;
			break;
		}
		if (lastSuccessful && gradMax <= o.gradient_tolerance)
		{
			termination = 0;
			break;
		}
		if (radius < o.min_radius)
		{
			termination = 0;
			break;
		}
		iteration++;
		lastSuccessful = false;

		// LevenbergMarquardtStrategy::ComputeStep: (L_w + C) y = -g by CG.
		if (launch_tvf_cg_init(A, M, radius, s))
		{
			return fail("cg init launch");
		}
		bool valid = false;
		int done = 0;
		while (done < cgMax)
		{
			if (done > 0 && chunkGraph)
			{
				if (hipGraphLaunch(chunkGraph, s) != hipSuccess)
				{
					return fail("cg graph launch");
				}
			}
			else if (launch_tvf_cg_iters(A, M, done, cgChunk, s))
			{
				return fail("cg launch");
			}
			done += cgChunk;
			rc = read_scal(A, s, sc, err);
			if (rc)
			{
				return rc;
			}
			const double lim0 = cgTol * cgTol * sc.v[kBb], lim1 = cgTol * cgTol * sc.v[kBb + 1];
			if (!std::isfinite(sc.v[kRr]) || !std::isfinite(sc.v[kRr + 1]))
			{
				break;
			}
			if (sc.v[kRr] <= lim0 && sc.v[kRr + 1] <= lim1)
			{
				valid = true;
				break;
			}
		}
		st.cg_iterations += done;
		double modelCostChange = 0.0, stepNorm = 0.0;
		if (valid)
		{
			if (launch_tvf_model(A, s))
			{
				return fail("model launch");
			}
			rc = read_scal(A, s, sc, err);
			if (rc)
			{
				return rc;
			}
			modelCostChange = -sc.v[kYg] - 0.5 * sc.v[kYLy];
			stepNorm = std::sqrt(sc.v[kYnorm2]);
			valid = std::isfinite(modelCostChange) && std::isfinite(stepNorm) && modelCostChange > 0.0;
		}
		if (!valid)
		{
			numInvalid++;
			if (numInvalid >= o.max_consecutive_invalid)
			{
				termination = 2;
				break;
			}
			radius *= 0.5;
			continue;
		}
		numInvalid = 0;
		rc = linearize(A.xc, 0, 1);
		if (rc)
		{
			return rc;
		}
		double candCost = sc.v[kCandCost];
		if (!std::isfinite(candCost))
		{
			candCost = std::numeric_limits<double>::max();
		}
		if (stepNorm <= o.parameter_tolerance * (xNorm + o.parameter_tolerance))
		{
			termination = 0;
			break;
		}
		const double costChange = xCost - candCost;
		if (std::fabs(costChange) <= o.function_tolerance * xCost)
		{
			termination = 0;
			break;
		}
		const double relDec = (seCurrent - candCost) / modelCostChange;
		const double histDec = (seReference - candCost) / (seAccRef + modelCostChange);
		const double quality = std::max(relDec, histDec);
		if (quality > o.min_relative_decrease)
		{
			std::swap(A.x, A.xc);
			rc = linearize(A.x, 0, 0);
			if (rc)
			{
				return rc;
			}
			xCost = sc.v[kCost];
			if (!std::isfinite(xCost))
			{
				termination = 2;
				break;
			}
			xNorm = std::sqrt(sc.v[kXnorm2]);
			gradMax = sc.v[kGmax];
			lastSuccessful = true;
			radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * quality - 1.0, 3));
			radius = std::min(o.max_radius, radius);
			decreaseFactor = 2.0;
			seCurrent = candCost;
			seAccCand += modelCostChange;
			seAccRef += modelCostChange;
			if (seCurrent < seMinimum)
			{
				seMinimum = seCurrent;
				seNumNonmono = 0;
				seCandidate = seCurrent;
				seAccCand = 0.0;
			}
			else
			{
				++seNumNonmono;
				if (seCurrent > seCandidate)
				{
					seCandidate = seCurrent;
					seAccCand = 0.0;
				}
			}
			if (seNumNonmono == maxNonmono)
			{
				seReference = seCandidate;
				seAccRef = seAccCand;
			}
		}
		else
		{
			radius = radius / decreaseFactor;
			decreaseFactor *= 2.0;
		}
	}

	if (launch_tvf_store(A, xbest, d_field, s))
	{
		return fail("store launch");
	}
	if (hipStreamSynchronize(s) != hipSuccess)
	{
		return fail("sync");
	}
	st.iterations = iteration;
	st.termination = termination;
	st.final_cost = minimumCost;
	*stats = st;
	return EBO_OK;
}
}  // namespace ebo
