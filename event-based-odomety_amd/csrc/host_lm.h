// host_lm.h — host side of EBO_SOLVE_GLOBAL: the trust-region
// Levenberg-Marquardt iteration that the reference delegates to ceres::Solve
// (feature_detector.cpp:401-414) over ONE problem per window: a contrast data
// term per active patch (evaluated on the device, batched) plus total-variation
// terms between grid neighbours with a Huber loss (feature_detector.cpp:369-396,
// total_variance.h:14-20), which are evaluated here (they are 4 flops each).
//
// Written as a resumable state machine: request() says which point the data
// terms are needed at and whether Jacobians are needed; supply() feeds them back.
// That lets the caller advance many windows in lock step with a single batched
// kernel launch per round.  The linear algebra is a banded Cholesky on the
// normal equations (the grid couples patch p with p+1 and p+npx only).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/ebo.h"

namespace ebo
{
class HostLm
{
   public:
	enum Request
	{
		DONE = 0,
		NEED_JACOBIAN = 1,
		NEED_COST = 2
	};
	struct Stats
	{
		int iterations = 0;
		int evals_cost = 0;
		int evals_jac = 0;
		int termination = 1;
		double initial_cost = 0.0;
		double final_cost = 0.0;
	};

	HostLm(int npx, int npy, const std::vector<uint8_t>& active, double tvWeight, double tvHuber,
		   const ebo_solver_opts& opts);

	// Writes the point [P][2] the next evaluation is wanted at (when not DONE).
	Request request(double* flows) const;
	// r [P], J [P][2] (J may be null only when the pending request is NEED_COST).
	void supply(const double* r, const double* J);
	void result(double* flows) const;
	const Stats& stats() const { return stats_; }

   private:
	struct TvBlock
	{
		int p, q;
	};
	enum Phase
	{
		PH_ZERO,
		PH_CANDIDATE,
		PH_ACCEPTED,
		PH_DONE
	};

	void evaluateAt(const std::vector<double>& x, const double* r, const double* J, bool wantJac,
					double& cost);
	void afterJacobian();
	bool computeStep();
	void advance();
	void finish(int termination);

	int npx_, npy_, P_;
	std::vector<uint8_t> active_;
	double tvW_, tvH_;
	ebo_solver_opts o_;

	std::vector<int> col_;     // param -> column or -1
	std::vector<int> paramOf_; // column -> param
	std::vector<TvBlock> tv_;
	int n_ = 0, band_ = 0;

	// residual rows: data rows first (one per active patch), then 2 per TV block
	std::vector<int> dataPatch_;
	std::vector<double> f_;            // corrected residuals
	std::vector<double> jd_;           // data rows: 2 entries each (scaled)
	std::vector<double> jt_;           // TV rows: 4 entries each (scaled)
	std::vector<double> grad_, scale_, diag_, step_, band_store_;

	std::vector<double> x_, cand_, best_;
	double xCost_ = 0.0, candCost_ = 0.0, xNorm_ = 0.0, gradMax_ = 0.0, minimumCost_ = 0.0;
	double modelCostChange_ = 0.0;
	// step evaluator
	double seMin_ = 0, seCur_ = 0, seRef_ = 0, seCand_ = 0, seAccRef_ = 0, seAccCand_ = 0;
	int seNonmono_ = 0, maxNonmono_ = 0;
	// LM strategy
	double radius_ = 0, decrease_ = 2.0;
	bool reuseDiag_ = false;
	bool lastSuccessful_ = false;
	int numInvalid_ = 0;
	Phase phase_ = PH_ZERO;
	Stats stats_;
};
}  // namespace ebo
