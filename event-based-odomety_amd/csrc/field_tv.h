// field_tv.h — host side of FeatureDetector::interpolateMotionField
// (feature_detector.cpp:144-241): the trust-region Levenberg-Marquardt iteration the
// reference delegates to ceres::Solve (:216-226) over the per-pixel TV problem, with every
// per-pixel operation (linearisation, the linear solve, the step test sums) on the device
// (ebo_fieldtv.inc).  The host sees a handful of scalars per iteration.
#pragma once

#include <string>

#include "../../include/ebo.h"
#include "ebo_internal.h"

namespace ebo
{
struct FieldTvStats
{
	int iterations = 0;
	int evals_cost = 0;
	int evals_jac = 0;
	int termination = 0;  // 0 convergence, 1 no convergence, 2 failure
	double initial_cost = 0.0;
	double final_cost = 0.0;
	int cg_iterations = 0;  // total over the LM iterations
	int smoothed = 0;       // 0: cv::norm(field) == 0, the field was left as it is (:152)
};

// d_field: float32 [h][w][2] on the device, read as initMotionField left it and overwritten
// with the smoothed field.  d_fixed: [n_fixed][2] (x, y) on the device.  workspace:
// tvf_workspace_bytes(w, h) bytes on the device.  Synchronises the stream.
// Returns EBO_OK, EBO_ERR_HIP, or EBO_ERR_NUMERIC when the linear solver does not converge.
int field_tv_solve(int w, int h, float* d_field, const int* d_fixed, int n_fixed, bool use_l1,
				   const ebo_solver_opts& opts, void* workspace, void* stream, FieldTvStats* stats,
				   std::string* err);
}  // namespace ebo
