// host_lm.cpp — see host_lm.h.  Algorithm: Ceres 2.0's documented trust-region
// minimizer with the Levenberg-Marquardt strategy and the non-monotonic step
// evaluator, the options the reference sets at feature_detector.cpp:401-410.
#include "host_lm.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace ebo
{
namespace
{
// ceres::HuberLoss(a) evaluated at s = |r|^2: value, first derivative.
inline void huber_rho(double a, double s, double& rho0, double& rho1)
{
	const double b = a * a;
	if (s > b)
	{
		const double root = std::sqrt(s);
		rho0 = 2.0 * a * root - b;
		rho1 = std::max(std::numeric_limits<double>::min(), a / root);
	}
	else
	{
		rho0 = s;
		rho1 = 1.0;
	}
}

// pitch of a column of the band: band + 1 entries, then at least three ZEROS that the four-wide loads below
// may read (and that stay zero: 0 - 0 * l), rounded up to a multiple of four
inline int band_pitch(int band)
{
	return (band + 1 + 3 + 3) & ~3;
}

// Banded Cholesky A = L L' in place on columns (entry (i, j), j <= i, at A[j * pitch + (i - j)]); false if a
// pivot is not positive and finite.  With 2 P = 216 unknowns and a band of 25 this is most of an LM step,
// and the LM steps are what bounds a lock-step round once the windows of a batch have thinned out
// (DESIGN 4.3).  Every entry is one chain
//     s = A(i,j) - L(i,k0) L(j,k0) - L(i,k0+1) L(j,k0+1) - ...   (k ascending from i - band),   L(i,j) = s / L(j,j),
// one rounding per operation (no contraction).  The chains of FOUR consecutive rows of a column advance
// together in one vector register -- each in its own order, so the factor has the bits of the scalar
// row-by-row loop this replaces (checked: same trajectories bit for bit); a row whose chain starts later
// meets the zero padding behind the pivot column's band until then (s - 0 = s).
typedef double lm_v4d __attribute__((vector_size(32), aligned(8), may_alias));
__attribute__((always_inline)) inline bool band_cholesky_body(double* A, int n, int band, int pitch)
{
	for (int j = 0; j < n; ++j)
	{
		double* const Cj = A + static_cast<size_t>(j) * pitch;
		const int mj = std::min(n - 1, j + band) - j;  // rows of the band below the diagonal
		for (int c = 0; c <= mj; c += 4)
		{
			lm_v4d v = *reinterpret_cast<const lm_v4d*>(Cj + c);
			for (int k = std::max(0, j + c - band); k < j; ++k)
			{
				const double* const Ck = A + static_cast<size_t>(k) * pitch + (j - k);
				const double ljk = Ck[0];
				const lm_v4d lk = *reinterpret_cast<const lm_v4d*>(Ck + c);
				v -= lk * ljk;
			}
			*reinterpret_cast<lm_v4d*>(Cj + c) = v;
		}
		const double d = Cj[0];
		if (!(d > 0.0) || !std::isfinite(d))
		{
			return false;
		}
		const double ljj = std::sqrt(d);
		Cj[0] = ljj;
		for (int t = 1; t <= mj; ++t)
		{
			Cj[t] = Cj[t] / ljj;
		}
		// what the vector stores left behind the band (rows beyond the band or the matrix collect only
		// 0 - 0 * l = 0, unless l is not finite -- then the pivot test above has already returned)
	}
	return true;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) bool band_cholesky_avx2(double* A, int n, int band, int pitch)
{
	return band_cholesky_body(A, n, band, pitch);
}
#endif
bool band_cholesky_plain(double* A, int n, int band, int pitch)
{
	return band_cholesky_body(A, n, band, pitch);
}
inline bool band_cholesky(double* A, int n, int band, int pitch)
{
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
	static const bool avx2 = __builtin_cpu_supports("avx2") && !std::getenv("EBO_LM_NO_AVX2");  // (the knob: tests run both)
	if (avx2)
	{
		return band_cholesky_avx2(A, n, band, pitch);
	}
#endif
	return band_cholesky_plain(A, n, band, pitch);
}
}  // namespace

HostLm::HostLm(int npx, int npy, const std::vector<uint8_t>& active, double tvWeight,
			   double tvHuber, const ebo_solver_opts& opts)
	: npx_(npx), npy_(npy), P_(npx * npy), active_(active), tvW_(tvWeight), tvH_(tvHuber), o_(opts)
{
	col_.assign(2 * P_, -1);
	std::vector<uint8_t> used(P_, 0);
	for (int p = 0; p < P_; ++p)
	{
		if (active_[p])
		{
			dataPatch_.push_back(p);
			used[p] = 1;
		}
	}
	if (tvW_ != 0.0)
	{
		// feature_detector.cpp:369-396: right neighbour, then bottom neighbour
		for (int y = 0; y < npy_; ++y)
		{
			for (int x = 0; x < npx_; ++x)
			{
				const int p = y * npx_ + x;
				if (x < npx_ - 1)
				{
					tv_.push_back({p, p + 1});
					used[p] = used[p + 1] = 1;
				}
				if (y < npy_ - 1)
				{
					tv_.push_back({p, p + npx_});
					used[p] = used[p + npx_] = 1;
				}
			}
		}
	}
	for (int p = 0; p < P_; ++p)
	{
		if (used[p])
		{
			col_[2 * p] = n_++;
			col_[2 * p + 1] = n_++;
			paramOf_.push_back(2 * p);
			paramOf_.push_back(2 * p + 1);
		}
	}
	band_ = 1;
	for (const TvBlock& b : tv_)
	{
		band_ = std::max(band_, std::abs(col_[2 * b.q] - col_[2 * b.p]) + 1);
	}
	x_.assign(2 * P_, 0.0);  // feature_detector.cpp:318-326
	cand_ = x_;
	best_ = x_;
	const size_t rows = dataPatch_.size() + 2 * tv_.size();
	f_.assign(rows, 0.0);
	jd_.assign(2 * dataPatch_.size(), 0.0);
	jt_.assign(4 * tv_.size(), 0.0);
	grad_.assign(n_, 0.0);
	scale_.assign(n_, 1.0);
	diag_.assign(n_, 0.0);
	step_.assign(n_, 0.0);
	band_store_.assign(static_cast<size_t>(n_) * band_pitch(band_), 0.0);
	maxNonmono_ = o_.use_nonmonotonic ? o_.max_consecutive_nonmonotonic : 0;
	radius_ = o_.initial_radius;
	if (n_ == 0 || rows == 0)
	{
		phase_ = PH_DONE;
		stats_.termination = 0;
	}
}

HostLm::Request HostLm::request(double* flows) const
{
	if (phase_ == PH_DONE)
	{
		return DONE;
	}
	const std::vector<double>& src = (phase_ == PH_CANDIDATE) ? cand_ : x_;
	std::copy(src.begin(), src.end(), flows);
	return (phase_ == PH_CANDIDATE) ? NEED_COST : NEED_JACOBIAN;
}

void HostLm::result(double* flows) const
{
	std::copy(best_.begin(), best_.end(), flows);
}

// cost = 1/2 sum rho_i(|f_i|^2); with wantJac also the loss-corrected residuals,
// the (unscaled) Jacobian entries and the gradient J'f.
void HostLm::evaluateAt(const std::vector<double>& x, const double* r, const double* J,
						bool wantJac, double& cost)
{
	double c = 0.0;
	if (wantJac)
	{
		std::fill(grad_.begin(), grad_.end(), 0.0);
	}
	const size_t nd = dataPatch_.size();
	for (size_t i = 0; i < nd; ++i)
	{
		const int p = dataPatch_[i];
		c += 0.5 * r[p] * r[p];
		if (wantJac)
		{
			f_[i] = r[p];
			jd_[2 * i] = J[2 * p];
			jd_[2 * i + 1] = J[2 * p + 1];
			grad_[col_[2 * p]] += J[2 * p] * r[p];
			grad_[col_[2 * p + 1]] += J[2 * p + 1] * r[p];
		}
	}
	for (size_t b = 0; b < tv_.size(); ++b)
	{
		const int p = tv_[b].p, q = tv_[b].q;
		double res[2], sgn[2];
		for (int k = 0; k < 2; ++k)
		{
			const double d = x[2 * p + k] - x[2 * q + k];
			sgn[k] = (d < 0.0) ? -1.0 : 1.0;  // ceres::abs on a Jet: f.a < 0 ? -f : f
			res[k] = tvW_ * (sgn[k] * d);
		}
		double rho0, rho1;
		huber_rho(tvH_, res[0] * res[0] + res[1] * res[1], rho0, rho1);
		c += 0.5 * rho0;
		if (wantJac)
		{
			const double sr = std::sqrt(rho1);  // Corrector with rho'' <= 0
			for (int k = 0; k < 2; ++k)
			{
				const size_t row = nd + 2 * b + k;
				f_[row] = res[k] * sr;
				const double jp = (tvW_ * sgn[k]) * sr;
				const double jq = -(tvW_ * sgn[k]) * sr;
				jt_[4 * b + 2 * k] = jp;
				jt_[4 * b + 2 * k + 1] = jq;
				grad_[col_[2 * p + k]] += jp * f_[row];
				grad_[col_[2 * q + k]] += jq * f_[row];
			}
		}
	}
	cost = c;
	if (wantJac)
	{
		stats_.evals_jac++;
	}
	else
	{
		stats_.evals_cost++;
	}
}

// Column scaling of the fresh Jacobian and the gradient max-norm.
void HostLm::afterJacobian()
{
	const size_t nd = dataPatch_.size();
	for (size_t i = 0; i < nd; ++i)
	{
		const int p = dataPatch_[i];
		jd_[2 * i] *= scale_[col_[2 * p]];
		jd_[2 * i + 1] *= scale_[col_[2 * p + 1]];
	}
	for (size_t b = 0; b < tv_.size(); ++b)
	{
		for (int k = 0; k < 2; ++k)
		{
			jt_[4 * b + 2 * k] *= scale_[col_[2 * tv_[b].p + k]];
			jt_[4 * b + 2 * k + 1] *= scale_[col_[2 * tv_[b].q + k]];
		}
	}
	gradMax_ = 0.0;
	for (double g : grad_)
	{
		gradMax_ = std::max(gradMax_, std::fabs(g));
	}
}

// LevenbergMarquardtStrategy::ComputeStep + the model cost change.  Returns
// false for an invalid step.
bool HostLm::computeStep()
{
	const size_t nd = dataPatch_.size();
	const int bw = band_pitch(band_);
	if (!reuseDiag_)
	{
		std::fill(diag_.begin(), diag_.end(), 0.0);
		for (size_t i = 0; i < nd; ++i)
		{
			const int p = dataPatch_[i];
			diag_[col_[2 * p]] += jd_[2 * i] * jd_[2 * i];
			diag_[col_[2 * p + 1]] += jd_[2 * i + 1] * jd_[2 * i + 1];
		}
		for (size_t b = 0; b < tv_.size(); ++b)
		{
			for (int k = 0; k < 2; ++k)
			{
				diag_[col_[2 * tv_[b].p + k]] += jt_[4 * b + 2 * k] * jt_[4 * b + 2 * k];
				diag_[col_[2 * tv_[b].q + k]] += jt_[4 * b + 2 * k + 1] * jt_[4 * b + 2 * k + 1];
			}
		}
		for (int c = 0; c < n_; ++c)
		{
			diag_[c] = std::min(std::max(diag_[c], o_.min_lm_diagonal), o_.max_lm_diagonal);
		}
	}
	reuseDiag_ = true;

	// Lower band of J'J + D'D, column by column: entry (i, j), j <= i, at [j*bw + (i-j)].
	std::fill(band_store_.begin(), band_store_.end(), 0.0);
	std::fill(step_.begin(), step_.end(), 0.0);
	auto addPair = [&](int ca, double va, int cb, double vb) {
		const int i = std::max(ca, cb), j = std::min(ca, cb);
		band_store_[static_cast<size_t>(j) * bw + (i - j)] += va * vb;
	};
	for (size_t i = 0; i < nd; ++i)
	{
		const int c0 = col_[2 * dataPatch_[i]], c1 = c0 + 1;
		const double a = jd_[2 * i], b = jd_[2 * i + 1];
		addPair(c0, a, c0, a);
		addPair(c1, b, c0, a);
		addPair(c1, b, c1, b);
		step_[c0] += a * f_[i];
		step_[c1] += b * f_[i];
	}
	for (size_t b = 0; b < tv_.size(); ++b)
	{
		for (int k = 0; k < 2; ++k)
		{
			const int cp = col_[2 * tv_[b].p + k], cq = col_[2 * tv_[b].q + k];
			const double vp = jt_[4 * b + 2 * k], vq = jt_[4 * b + 2 * k + 1];
			const double fr = f_[nd + 2 * b + k];
			addPair(cp, vp, cp, vp);
			addPair(cq, vq, cp, vp);
			addPair(cq, vq, cq, vq);
			step_[cp] += vp * fr;
			step_[cq] += vq * fr;
		}
	}
	for (int c = 0; c < n_; ++c)
	{
		const double l = std::sqrt(diag_[c] / radius_);
		band_store_[static_cast<size_t>(c) * bw] += l * l;
	}
	double* const A = band_store_.data();
	if (!band_cholesky(A, n_, band_, bw))
	{
		return false;
	}
	// L y = b by columns (y_k, then its multiples off the rows below: row i again collects its terms in
	// ascending k), L' x = y by rows of L' = columns of L (a dot product in ascending k, as before)
	for (int k = 0; k < n_; ++k)
	{
		const double* __restrict const Ck = A + static_cast<size_t>(k) * bw;
		const double y = step_[k] / Ck[0];
		step_[k] = y;
		const int m = std::min(n_ - 1, k + band_) - k;
		double* __restrict const sk = step_.data() + k;
		for (int t = 1; t <= m; ++t)
		{
			sk[t] -= Ck[t] * y;
		}
	}
	for (int i = n_ - 1; i >= 0; --i)
	{
		const double* const Ci = A + static_cast<size_t>(i) * bw;
		double sv = step_[i];
		const int m = std::min(n_ - 1, i + band_) - i;
		for (int t = 1; t <= m; ++t)
		{
			sv -= Ci[t] * step_[i + t];
		}
		step_[i] = sv / Ci[0];
	}
	for (int c = 0; c < n_; ++c)
	{
		if (!std::isfinite(step_[c]))
		{
			return false;
		}
		step_[c] = -step_[c];
	}
	// model_cost_change = -(J s)'(f + J s / 2)
	double mcc = 0.0;
	for (size_t i = 0; i < nd; ++i)
	{
		const int c0 = col_[2 * dataPatch_[i]];
		const double mr = jd_[2 * i] * step_[c0] + jd_[2 * i + 1] * step_[c0 + 1];
		mcc -= mr * (f_[i] + mr / 2.0);
	}
	for (size_t b = 0; b < tv_.size(); ++b)
	{
		for (int k = 0; k < 2; ++k)
		{
			const double mr = jt_[4 * b + 2 * k] * step_[col_[2 * tv_[b].p + k]] +
							  jt_[4 * b + 2 * k + 1] * step_[col_[2 * tv_[b].q + k]];
			mcc -= mr * (f_[nd + 2 * b + k] + mr / 2.0);
		}
	}
	modelCostChange_ = mcc;
	return mcc > 0.0;
}

void HostLm::finish(int termination)
{
	phase_ = PH_DONE;
	stats_.termination = termination;
	stats_.final_cost = minimumCost_;
}

// Runs the outer loop until an evaluation is needed (or the solve ends).
void HostLm::advance()
{
	for (;;)
	{
		if (lastSuccessful_ && xCost_ < minimumCost_)
		{
			minimumCost_ = xCost_;
			best_ = x_;
		}
		if (stats_.iterations >= o_.max_num_iterations)
		{
			finish(1);
			return;
		}
		if (lastSuccessful_ && gradMax_ <= o_.gradient_tolerance)
		{
			finish(0);
			return;
		}
		if (radius_ < o_.min_radius)
		{
			finish(0);
			return;
		}
		stats_.iterations++;
		lastSuccessful_ = false;
		if (!computeStep())
		{
			numInvalid_++;
			if (numInvalid_ >= o_.max_consecutive_invalid)
			{
				finish(2);
				return;
			}
			radius_ *= 0.5;  // LevenbergMarquardtStrategy::StepIsInvalid
			reuseDiag_ = true;
			continue;
		}
		numInvalid_ = 0;
		cand_ = x_;
		for (int c = 0; c < n_; ++c)
		{
			cand_[paramOf_[c]] = x_[paramOf_[c]] + step_[c] * scale_[c];
		}
		phase_ = PH_CANDIDATE;
		return;
	}
}

void HostLm::supply(const double* r, const double* J)
{
	if (phase_ == PH_DONE)
	{
		return;
	}
	if (phase_ == PH_ZERO)
	{
		evaluateAt(x_, r, J, true, xCost_);
		stats_.initial_cost = xCost_;
		minimumCost_ = xCost_;
		if (!std::isfinite(xCost_))
		{
			finish(2);
			return;
		}
		if (o_.jacobi_scaling)
		{
			std::vector<double> cn(n_, 0.0);
			const size_t nd = dataPatch_.size();
			for (size_t i = 0; i < nd; ++i)
			{
				const int p = dataPatch_[i];
				cn[col_[2 * p]] += jd_[2 * i] * jd_[2 * i];
				cn[col_[2 * p + 1]] += jd_[2 * i + 1] * jd_[2 * i + 1];
			}
			for (size_t b = 0; b < tv_.size(); ++b)
			{
				for (int k = 0; k < 2; ++k)
				{
					cn[col_[2 * tv_[b].p + k]] += jt_[4 * b + 2 * k] * jt_[4 * b + 2 * k];
					cn[col_[2 * tv_[b].q + k]] += jt_[4 * b + 2 * k + 1] * jt_[4 * b + 2 * k + 1];
				}
			}
			for (int c = 0; c < n_; ++c)
			{
				scale_[c] = 1.0 / (1.0 + std::sqrt(cn[c]));
			}
		}
		afterJacobian();
		xNorm_ = 0.0;
		seMin_ = seCur_ = seRef_ = seCand_ = xCost_;
		// iteration zero counts as successful: a start already within the gradient
		// tolerance converges immediately ("Gradient tolerance reached")
		lastSuccessful_ = true;
		advance();
		return;
	}
	if (phase_ == PH_CANDIDATE)
	{
		evaluateAt(cand_, r, nullptr, false, candCost_);
		if (!std::isfinite(candCost_))
		{
			candCost_ = std::numeric_limits<double>::max();
		}
		double stepNorm = 0.0;
		for (int c = 0; c < n_; ++c)
		{
			const double d = x_[paramOf_[c]] - cand_[paramOf_[c]];
			stepNorm += d * d;
		}
		stepNorm = std::sqrt(stepNorm);
		if (stepNorm <= o_.parameter_tolerance * (xNorm_ + o_.parameter_tolerance))
		{
			finish(0);
			return;
		}
		if (std::fabs(xCost_ - candCost_) <= o_.function_tolerance * xCost_)
		{
			finish(0);
			return;
		}
		// TrustRegionStepEvaluator::StepQuality
		const double rel = (seCur_ - candCost_) / modelCostChange_;
		const double hist = (seRef_ - candCost_) / (seAccRef_ + modelCostChange_);
		const double quality = std::max(rel, hist);
		if (quality > o_.min_relative_decrease)
		{
			x_ = cand_;
			double s = 0.0;
			for (int c = 0; c < n_; ++c)
			{
				s += x_[paramOf_[c]] * x_[paramOf_[c]];
			}
			xNorm_ = std::sqrt(s);
			// strategy / evaluator updates need nothing from the new Jacobian
			const double q = 2.0 * quality - 1.0;
			radius_ = radius_ / std::max(1.0 / 3.0, 1.0 - q * q * q);
			radius_ = std::min(o_.max_radius, radius_);
			decrease_ = 2.0;
			reuseDiag_ = false;
			seCur_ = candCost_;
			seAccCand_ += modelCostChange_;
			seAccRef_ += modelCostChange_;
			if (seCur_ < seMin_)
			{
				seMin_ = seCur_;
				seNonmono_ = 0;
				seCand_ = seCur_;
				seAccCand_ = 0.0;
			}
			else
			{
				++seNonmono_;
				if (seCur_ > seCand_)
				{
					seCand_ = seCur_;
					seAccCand_ = 0.0;
				}
			}
			if (seNonmono_ == maxNonmono_)
			{
				seRef_ = seCand_;
				seAccRef_ = seAccCand_;
			}
			phase_ = PH_ACCEPTED;
			return;
		}
		radius_ = radius_ / decrease_;  // LevenbergMarquardtStrategy::StepRejected
		decrease_ *= 2.0;
		reuseDiag_ = true;
		advance();
		return;
	}
	// PH_ACCEPTED: gradient and Jacobian at the new point.
	evaluateAt(x_, r, J, true, xCost_);
	if (!std::isfinite(xCost_))
	{
		finish(2);
		return;
	}
	afterJacobian();
	lastSuccessful_ = true;
	advance();
}

}  // namespace ebo
