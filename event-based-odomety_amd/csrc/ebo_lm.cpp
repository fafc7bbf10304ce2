// ebo_lm.cpp — the host solver of EBO_SOLVE_GLOBAL (csrc/host_lm.cpp, the trust-region LM the reference
// delegates to ceres::Solve, feature_detector.cpp:401-414) behind the C ABI as the resumable state machine it
// is: request -> (the caller evaluates the data terms wherever they live) -> supply.  ebo_solve drives it
// internally with batched device evaluations; a caller that must put something BETWEEN "evaluate" and "step"
// -- SURVEY 8(e)'s reference-faithful TV mode across GPUs: every rank evaluates its patch rows, ONE all-gather
// of (r, J0, J1) per evaluation, the same solver replicated on every rank -- drives it through these entry
// points.  Host only; no device call here.
#include <new>
#include <vector>

#include "../../include/ebo.h"
#include "host_lm.h"

struct ebo_lm
{
	ebo::HostLm lm;
	int P;
	ebo_lm(int npx, int npy, const std::vector<uint8_t>& active, double w, double h, const ebo_solver_opts& o)
		: lm(npx, npy, active, w, h, o), P(npx * npy)
	{
	}
};

extern "C" {

int ebo_lm_create(int npx, int npy, const uint8_t* active, double tv_weight, double tv_huber, const ebo_solver_opts* o,
				  ebo_lm** out)
{
	if (!out || !active || !o || npx <= 0 || npy <= 0 || static_cast<long long>(npx) * npy > (1 << 24))
	{
		return EBO_ERR_ARG;
	}
	if (o->max_num_iterations < 0 || !(o->initial_radius > 0) || o->max_consecutive_invalid < 1)
	{
		return EBO_ERR_ARG;
	}
	const std::vector<uint8_t> a(active, active + static_cast<size_t>(npx) * npy);
	ebo_lm* h = new (std::nothrow) ebo_lm(npx, npy, a, tv_weight, tv_huber, *o);
	if (!h)
	{
		return EBO_ERR_ARG;
	}
	*out = h;
	return EBO_OK;
}

int ebo_lm_request(ebo_lm* h, double* flows)
{
	if (!h || !flows)
	{
		return EBO_ERR_ARG;
	}
	return static_cast<int>(h->lm.request(flows));  // 0 done, 1 value + Jacobian, 2 value
}

int ebo_lm_supply(ebo_lm* h, const double* r, const double* jac)
{
	if (!h || !r)
	{
		return EBO_ERR_ARG;
	}
	std::vector<double> dummy(static_cast<size_t>(h->P) * 2);
	const ebo::HostLm::Request pending = h->lm.request(dummy.data());
	if (pending == ebo::HostLm::DONE)
	{
		return EBO_ERR_STATE;
	}
	if (pending == ebo::HostLm::NEED_JACOBIAN && !jac)
	{
		return EBO_ERR_ARG;
	}
	h->lm.supply(r, jac);
	return EBO_OK;
}

int ebo_lm_result(const ebo_lm* h, double* flows, ebo_summary* summary)
{
	if (!h || !flows)
	{
		return EBO_ERR_ARG;
	}
	h->lm.result(flows);
	if (summary)
	{
		const ebo::HostLm::Stats& s = h->lm.stats();
		summary->iterations = s.iterations;
		summary->num_evals_cost = s.evals_cost;  // rounds of the whole problem (ebo_solve reports them per data term)
		summary->num_evals_jac = s.evals_jac;
		summary->termination = s.termination;
		summary->initial_cost = s.initial_cost;
		summary->final_cost = s.final_cost;
	}
	return EBO_OK;
}

void ebo_lm_destroy(ebo_lm* h) { delete h; }

}  // extern "C"
