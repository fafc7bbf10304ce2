// feature_tracker/optimizer_cost.h — tracker::OptimizerCostFunctor, Grid and Interpolator with the
// reference's names and constructors (implementation/feature_tracker/include/feature_tracker/
// optimizer_cost.h:8-96), evaluated on the MI355X through the C ABI: the second Ceres CostFunction of
// feature_tracker.
//
// The reference builds, per optimisation (optimizer.cpp:27-30, 86-97),
//     gradGrid_.reset(new Grid(grad_.data(), 0, imageSize_.height, 0, imageSize_.width));
//     gradInterpolator_.reset(new Interpolator(*(gradGrid_.get())));
//     auto* c = new tracker::OptimizerCostFunctor(normalizedIntegratedNabla, gradInterpolator_.get(), currentRect, imageSize_);
//     new ceres::AutoDiffCostFunction<tracker::OptimizerCostFunctor, ceres::DYNAMIC, Sophus::SE2d::num_parameters, 1>(c, size);
// Those statements compile against this header unchanged (tests/cpp/optimizer_cost_lines_test.cpp):
//   * Grid (ceres::Grid2D<double, 2> there): the interleaved (gradX, gradY) image.  Here it owns a device
//     context of the grid's size and uploads the two planes once (ebo_optimizer_set_grad); like Grid2D it
//     is built from `data` as it is at construction (the reference makes a new Grid per setGrad).
//   * Interpolator (ceres::BiCubicInterpolator<Grid>): a handle on that context; the Catmull-Rom
//     interpolation itself runs inside the evaluation kernel.
//   * OptimizerCostFunctor::operator()(const T* sPose2D, const T* sFlowDir, T* sResiduals): T = double is
//     the functor's double path (what drawCostMap calls); T = ceres::Jet<double, 5> (any dual-number type
//     with a scalar part `.a`) evaluates residuals and both Jacobian blocks at the scalar parts in ONE
//     launch (ebo_optimizer_eval) and forms the partials by the chain rule AutoDiff expects, in Jet algebra
//     only -- so AutoDiffCostFunction<..., DYNAMIC, 4, 1>::Evaluate returns exactly ebo_optimizer_eval's
//     numbers.  operator() returns false when the device call fails (the reference returns true always).
// tracker::Optimizer does not go through here: it solves all ready patches in one launch
// (ebo_optimizer_solve).  This header is for callers that keep Ceres in the loop.
#pragma once

#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../common/data_types.h"
#include "types.h"

namespace tracker
{
class Grid
{
   public:
	// ceres::Grid2D<double, 2>(data, row_begin, row_end, col_begin, col_end): row-major, interleaved
	Grid(const double* data, int row_begin, int row_end, int col_begin, int col_end, int device = 0)
		: rows_(row_end - row_begin), cols_(col_end - col_begin)
	{
		if (!data || row_begin != 0 || col_begin != 0 || rows_ < 1 || cols_ < 1)
		{
			throw std::invalid_argument("tracker::Grid: data, row_begin = col_begin = 0 and a non-empty range (as Optimizer::setGrad builds it)");
		}
		ebo_params p;
		ebo_default_params(&p);
		p.device = device;
		p.image_w = cols_;
		p.image_h = rows_;
		p.patch_w = std::min(p.patch_w, cols_);  // (the compensation grid is not used on this path)
		p.patch_h = std::min(p.patch_h, rows_);
		if (ebo_create(&p, &ctx_) != EBO_OK)
		{
			throw std::runtime_error(std::string("tracker::Grid: ") + ebo_last_error(nullptr));
		}
		const size_t n = static_cast<size_t>(rows_) * cols_;
		std::vector<double> gx(n), gy(n);
		for (size_t i = 0; i < n; ++i)
		{
			gx[i] = data[2 * i];
			gy[i] = data[2 * i + 1];
		}
		if (ebo_optimizer_set_grad(ctx_, gx.data(), gy.data()) != EBO_OK)
		{
			const std::string why = ebo_last_error(ctx_);
			ebo_destroy(ctx_);
			throw std::runtime_error("tracker::Grid: " + why);
		}
	}
	~Grid() { ebo_destroy(ctx_); }
	Grid(const Grid&) = delete;
	Grid& operator=(const Grid&) = delete;
	int rows() const { return rows_; }
	int cols() const { return cols_; }
	ebo_ctx* handle() const { return ctx_; }

   private:
	int rows_, cols_;
	ebo_ctx* ctx_ = nullptr;
};
using GridPtr = std::unique_ptr<Grid>;

class Interpolator
{
   public:
	explicit Interpolator(const Grid& grid) : grid_(grid) {}
	ebo_ctx* handle() const { return grid_.handle(); }
	const Grid& grid() const { return grid_; }

   private:
	const Grid& grid_;
};
using InterpolatorPtr = std::unique_ptr<Interpolator>;

struct OptimizerCostFunctor
{
	OptimizerCostFunctor() {}

	// optimizer_cost.h:19-28
	OptimizerCostFunctor(const Mat64 normalizedIntegratedNabla, Interpolator* interpolator, const Rect2d& patch,
						 const Size& imageSize)
		: normalizedIntegratedNabla_(normalizedIntegratedNabla), patch_(patch), imageSize_(imageSize)
	{
		gradInterpolator_ = interpolator;
	}

	// T = double (optimizer_cost.h:30-47 with T = double): residuals only
	bool operator()(const double* sPose2D, const double* sFlowDir, double* sResiduals) const
	{
		const double rect[4] = {patch_.x, patch_.y, patch_.width, patch_.height};
		return usable() && ebo_optimizer_eval(gradInterpolator_->handle(), 1, rect, normalizedIntegratedNabla_.ptr(), sPose2D, sFlowDir,
											  sResiduals, nullptr, nullptr) == EBO_OK;
	}

	// T = ceres::Jet<double, N>: residual k = r_k + sum_j dr_k/dpose_j * pose_j.v + dr_k/dflow * flow.v
	template <typename T>
	bool operator()(const T* sPose2D, const T* sFlowDir, T* sResiduals) const
	{
		if (!usable())
		{
			return false;
		}
		const int m = static_cast<int>(patch_.width) * static_cast<int>(patch_.height);
		const double rect[4] = {patch_.x, patch_.y, patch_.width, patch_.height};
		const double pose[4] = {static_cast<double>(sPose2D[0].a), static_cast<double>(sPose2D[1].a),
								static_cast<double>(sPose2D[2].a), static_cast<double>(sPose2D[3].a)};
		const double flow = static_cast<double>(sFlowDir[0].a);
		res_.resize(static_cast<size_t>(m));
		jacPose_.resize(4 * static_cast<size_t>(m));
		jacFlow_.resize(static_cast<size_t>(m));
		if (ebo_optimizer_eval(gradInterpolator_->handle(), 1, rect, normalizedIntegratedNabla_.ptr(), pose, &flow, res_.data(),
							   jacPose_.data(), jacFlow_.data()) != EBO_OK)
		{
			return false;
		}
		for (int k = 0; k < m; ++k)
		{
			const double* jp = &jacPose_[4 * static_cast<size_t>(k)];
			T out = sPose2D[0] * jp[0] + sPose2D[1] * jp[1] + sPose2D[2] * jp[2] + sPose2D[3] * jp[3] +
					sFlowDir[0] * jacFlow_[static_cast<size_t>(k)];
			out.a = res_[static_cast<size_t>(k)];
			sResiduals[k] = out;
		}
		return true;
	}

   private:
	bool usable() const
	{
		return gradInterpolator_ != nullptr && gradInterpolator_->grid().cols() == imageSize_.width &&
			   gradInterpolator_->grid().rows() == imageSize_.height &&
			   normalizedIntegratedNabla_.rows == static_cast<int>(patch_.height) &&
			   normalizedIntegratedNabla_.cols == static_cast<int>(patch_.width);
	}

	Mat64 normalizedIntegratedNabla_;
	Interpolator* gradInterpolator_ = nullptr;
	Rect2d patch_;
	Size imageSize_;
	mutable std::vector<double> res_, jacPose_, jacFlow_;
};

}  // namespace tracker
