// feature_tracker/total_variance.h — tracker::totalVarianceFunctor
// (implementation/feature_tracker/include/feature_tracker/total_variance.h:10-22): the
// regulariser between the flows of neighbouring patches, r[k] = weight * |x[k] - y[k]|.
// It is 4 flops per block and stays on the host (inside Ceres when Ceres drives the
// solve, inside csrc/host_lm.cpp when EBO_SOLVE_GLOBAL does).  Templated on the scalar
// so that AutoDiffCostFunction<totalVarianceFunctor,2,2,2> instantiates it with Jets.
#pragma once

#include <cmath>

namespace tracker
{
struct totalVarianceFunctor
{
	totalVarianceFunctor() : weight_(1.0) {}
	explicit totalVarianceFunctor(double weight) : weight_(weight) {}

	template <typename T>
	bool operator()(const T* x, const T* y, T* residual) const
	{
		using std::abs;  // ceres::abs for Jets is found by ADL
		for (int k = 0; k < 2; ++k)
		{
			residual[k] = T(weight_) * abs(x[k] - y[k]);
		}
		return true;
	}

	double weight_;
};
}  // namespace tracker
