// feature_tracker/contrast_functor.h — tracker::contrastFunctor with the reference's
// constructor and Ceres CostFunction surface
// (implementation/feature_tracker/include/feature_tracker/contrast_functor.h:10-36), evaluated on
// the MI355X through the C ABI.
//
// Three ways in, from "unchanged caller" to "fast":
//  1. tracker::contrastFunctor(events, patchRect, compensateScale) — same constructor and the
//     reference's templated operator()(const T* motion, T* residual): T = double AND T = ceres::Jet,
//     so ceres::AutoDiffCostFunction<tracker::contrastFunctor,1,2> is built by the reference's own
//     statement (feature_detector.cpp:359-363) with no edit.  Evaluate(parameters, residuals,
//     jacobians) is the same call without Jets (1 parameter block of 2, 1 residual, Jacobian 1x2
//     row-major, jacobians == nullptr or jacobians[0] == nullptr => value only).  One device launch
//     per call and block (tests/cpp/ceres_reference_lines_test.cpp).
//  2. tracker::ContrastBatch — all patches of a problem in one context; evaluate(flows) is ONE
//     launch for every patch; HipContrastCost blocks read the cached results.
//  3. With Ceres present: HipContrastCost is a ceres::SizedCostFunction<1,2> and
//     ContrastBatch a ceres::EvaluationCallback (Problem::Options::evaluation_callback), so the
//     per-block CostFunction surface stays while every iteration is one batched launch.
// operator() "returns true always" in the reference (:35); here it returns false when the device
// call fails (Ceres then treats the step as failed) — never throws.
#pragma once

#include <list>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "types.h"

#if defined(__has_include)
#if __has_include(<ceres/ceres.h>)
#include <ceres/ceres.h>
#define EBO_HAVE_CERES 1
#endif
#endif

namespace tracker
{
// All data terms of one problem on the device: patch i = (events_i, rect_i).
class ContrastBatch
#ifdef EBO_HAVE_CERES
	: public ceres::EvaluationCallback
#endif
{
   public:
	ContrastBatch(double compensateScale, int loss = EBO_LOSS_EDGE, int device = 0,
				  unsigned int minEvents = 0)
		: scale_(compensateScale), loss_(loss), device_(device), minEvents_(minEvents)
	{
	}
	~ContrastBatch()
	{
		if (ctx_)
		{
			ebo_destroy(ctx_);
		}
	}
	ContrastBatch(const ContrastBatch&) = delete;
	ContrastBatch& operator=(const ContrastBatch&) = delete;

	// Returns the index of the new patch.  `params` (2 doubles, optional) is where
	// PrepareForEvaluation reads this patch's current flow from.
	int addPatch(const std::list<common::EventSample>& events, const Rect2i& rect,
				 const double* params = nullptr)
	{
		offsets_.push_back(events_.size());
		const std::vector<ebo_event> ev = common::toEboEvents(events);
		events_.insert(events_.end(), ev.begin(), ev.end());
		rects_.push_back(rect.x);
		rects_.push_back(rect.y);
		rects_.push_back(rect.width);
		rects_.push_back(rect.height);
		paramPtrs_.push_back(params);
		dirty_ = true;
		return static_cast<int>(paramPtrs_.size()) - 1;
	}
	int size() const { return static_cast<int>(paramPtrs_.size()); }

	// One launch: residual (and Jacobian) of every patch at flows[n][2].
	bool evaluate(const double* flows, bool wantJacobian)
	{
		if (!upload())
		{
			return false;
		}
		const int n = size();
		r_.resize(n);
		J_.resize(2 * static_cast<size_t>(n));
		return ebo_eval(ctx_, flows, r_.data(), wantJacobian ? J_.data() : nullptr) == EBO_OK;
	}
	double residual(int i) const { return r_[i]; }
	const double* jacobian(int i) const { return &J_[2 * static_cast<size_t>(i)]; }
	const char* lastError() const { return ctx_ ? ebo_last_error(ctx_) : ebo_last_error(nullptr); }

	// ceres::EvaluationCallback: gather every block's current parameters, one launch.
	void PrepareForEvaluation(bool evaluate_jacobians, bool /*new_evaluation_point*/)
#ifdef EBO_HAVE_CERES
		override
#endif
	{
		std::vector<double> flows(2 * static_cast<size_t>(size()), 0.0);
		for (int i = 0; i < size(); ++i)
		{
			if (paramPtrs_[i])
			{
				flows[2 * i] = paramPtrs_[i][0];
				flows[2 * i + 1] = paramPtrs_[i][1];
			}
		}
		ok_ = evaluate(flows.data(), evaluate_jacobians);
	}
	bool ok() const { return ok_; }

   private:
	bool upload()
	{
		if (!dirty_ && ctx_)
		{
			return true;
		}
		if (ctx_)
		{
			ebo_destroy(ctx_);
			ctx_ = nullptr;
		}
		ebo_params p;
		ebo_default_params(&p);
		p.device = device_;
		p.scale = scale_;
		p.loss = loss_;
		p.min_events = minEvents_;
		p.image_w = 16;  // the grid is unused by ebo_set_patches; capacity = max_windows * 1
		p.image_h = 16;
		p.patch_w = 16;
		p.patch_h = 16;
		p.max_windows = size();
		p.max_events = events_.empty() ? 1 : events_.size();
		if (ebo_create(&p, &ctx_) != EBO_OK)
		{
			return false;
		}
		std::vector<size_t> off(offsets_);
		off.push_back(events_.size());
		if (ebo_set_patches(ctx_, events_.data(), off.data(), rects_.data(), size()) != EBO_OK)
		{
			return false;
		}
		dirty_ = false;
		return true;
	}

	double scale_;
	int loss_, device_;
	unsigned int minEvents_;
	ebo_ctx* ctx_ = nullptr;
	bool dirty_ = true, ok_ = false;
	std::vector<ebo_event> events_;
	std::vector<size_t> offsets_;
	std::vector<int32_t> rects_;
	std::vector<const double*> paramPtrs_;
	std::vector<double> r_, J_;
};

// Same constructor as the reference's functor (contrast_functor.h:12-21).
struct contrastFunctor
{
	contrastFunctor(const std::list<common::EventSample>& events, const Rect2i patchRect,
					const double compensateScale, int loss = EBO_LOSS_EDGE, int device = 0)
		: batch_(std::make_shared<ContrastBatch>(compensateScale, loss, device))
	{
		batch_->addPatch(events, patchRect);
	}

	// T = double instantiation of the reference's templated operator() (:23-36).
	bool operator()(const double* motion, double* residual) const
	{
		if (!batch_->evaluate(motion, false))
		{
			return false;
		}
		residual[0] = batch_->residual(0);
		return true;
	}

	// T = ceres::Jet<double, N> instantiation (any type with a scalar part `.a` and Jet algebra):
	// what `new ceres::AutoDiffCostFunction<tracker::contrastFunctor, 1, 2>(new
	// tracker::contrastFunctor(patchEvents, patchRect, compensateScale))` -- the reference's own line,
	// feature_detector.cpp:359-363 -- instantiates, so that line compiles and runs UNCHANGED.  The
	// device evaluates the residual and its 1x2 Jacobian at (motion[0].a, motion[1].a) in one launch
	// (the same arithmetic the reference's Jet<double,2> carries through compensateEvents and the
	// loss); the partials follow by the chain rule AutoDiff expects,
	//     residual.a = r,   residual.v = J0 * motion[0].v + J1 * motion[1].v,
	// written in Jet algebra only (Jet * double, Jet + Jet), so Ceres' Eigen-backed Jet and any
	// other dual-number type work alike.
	template <typename T>
	bool operator()(const T* motion, T* residual) const
	{
		const double point[2] = {static_cast<double>(motion[0].a), static_cast<double>(motion[1].a)};
		if (!batch_->evaluate(point, true))
		{
			return false;
		}
		T out = motion[0] * batch_->jacobian(0)[0] + motion[1] * batch_->jacobian(0)[1];
		out.a = batch_->residual(0);
		residual[0] = out;
		return true;
	}

	// ceres::CostFunction::Evaluate as AutoDiffCostFunction<contrastFunctor,1,2> implements it.
	bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const
	{
		const bool wantJ = jacobians != nullptr && jacobians[0] != nullptr;
		if (!batch_->evaluate(parameters[0], wantJ))
		{
			return false;
		}
		residuals[0] = batch_->residual(0);
		if (wantJ)
		{
			jacobians[0][0] = batch_->jacobian(0)[0];
			jacobians[0][1] = batch_->jacobian(0)[1];
		}
		return true;
	}

   private:
	std::shared_ptr<ContrastBatch> batch_;
};

#ifdef EBO_HAVE_CERES
// One residual block of a ContrastBatch.  Add ContrastBatch as
// Problem::Options::evaluation_callback; Evaluate only copies the batched results.
class HipContrastCost : public ceres::SizedCostFunction<1, 2>
{
   public:
	HipContrastCost(ContrastBatch* batch, int index) : batch_(batch), index_(index) {}
	bool Evaluate(double const* const* /*parameters*/, double* residuals,
				  double** jacobians) const override
	{
		if (!batch_->ok())
		{
			return false;
		}
		residuals[0] = batch_->residual(index_);
		if (jacobians != nullptr && jacobians[0] != nullptr)
		{
			jacobians[0][0] = batch_->jacobian(index_)[0];
			jacobians[0][1] = batch_->jacobian(index_)[1];
		}
		return true;
	}

   private:
	ContrastBatch* batch_;
	int index_;
};
#endif

}  // namespace tracker
