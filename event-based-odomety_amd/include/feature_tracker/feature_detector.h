// feature_tracker/feature_detector.h — tracker::FeatureDetector and
// tracker::DetectorParams with the reference's names and call order
// (implementation/feature_tracker/include/feature_tracker/feature_detector.h:10-120,
// src/feature_detector.cpp:243-482,621-628), for the event-warping path only.  Every
// method forwards to the C ABI (include/ebo.h); nothing is computed on the CPU here.
//
// Drop-in use under the unchanged front end (tools/evaluator/src/evaluator.cpp:32-45):
//     tracker_->addEvent(sample);
//     if (window due) {
//         tracker_->compensateEventsContrast(tracker_->getEvents());
//         tracker_->integrateEvents(tracker_->getEvents());
//         tracker_->clearEvents();
//     }
// Result images are CV_64F-like (tracker::Mat64, row-major doubles, rows x cols =
// imageSize), valid until the next call — as the cv::Mat const& of the reference.
// With OpenCV available, wrap without a copy:
//     cv::Mat view(m.rows, m.cols, CV_64F, const_cast<double*>(m.ptr()));
#pragma once

#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "../common/data_types.h"

namespace tracker
{
struct Size
{
	int width = 0;
	int height = 0;
	Size() = default;
	Size(int w, int h) : width(w), height(h) {}
};

// cv::Rect2i stand-in; contains() is half-open like cv::Rect_::contains.
struct Rect2i
{
	int x = 0, y = 0, width = 0, height = 0;
	Rect2i() = default;
	Rect2i(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
	bool contains(const common::Point2i& p) const
	{
		return x <= p.x && p.x < x + width && y <= p.y && p.y < y + height;
	}
};

// CV_64F single-channel image stand-in.
class Mat64
{
   public:
	int rows = 0;
	int cols = 0;
	Mat64() = default;
	Mat64(int r, int c) : rows(r), cols(c), data_(static_cast<size_t>(r) * c, 0.0) {}
	template <typename T = double>
	T& at(int r, int c)
	{
		return data_[static_cast<size_t>(r) * cols + c];
	}
	template <typename T = double>
	const T& at(int r, int c) const
	{
		return data_[static_cast<size_t>(r) * cols + c];
	}
	double* ptr() { return data_.data(); }
	const double* ptr() const { return data_.data(); }

   private:
	std::vector<double> data_;
};

// The DetectorParams fields this path reads (feature_detector.h:17,21-30), same
// names and defaults; the last block selects what the north star adds.
struct DetectorParams
{
	Size imageSize = {240, 180};
	unsigned long maxNumEventsToStore = 15000;
	bool useAverageFlow = true;
	bool optimizeFlowTV = true;
	bool useL1 = false;
	Size patchCompensateSize = {20, 20};
	double compensateTVweight = 1e3;
	double compensateTVHuberLoss = 10;
	double compensateScale = 1e-3;
	unsigned int compensateMinNumEvents = 100;

	int device = 0;                 // HIP device ordinal
	int loss = EBO_LOSS_EDGE;       // reference default; EBO_LOSS_VARIANCE = north-star objective
	int grad = EBO_GRAD_JET;        // reference default (ceres::Jet)
	int solveMode = EBO_SOLVE_GLOBAL;  // reference default: one problem incl. TV terms
	// The reference never throws on this path.  ERRORS_THROW (default: a missing GPU or a failed
	// launch must not go unnoticed) raises std::runtime_error; ERRORS_STATUS never throws: a failed
	// call leaves its outputs as they were and is reported by status() / lastError() / ok().
	enum ErrorPolicy
	{
		ERRORS_THROW = 0,
		ERRORS_STATUS = 1
	};
	int errorPolicy = ERRORS_THROW;
};

class FeatureDetector
{
   public:
	explicit FeatureDetector(const DetectorParams& params) : params_(params)
	{
		ebo_params p;
		ebo_default_params(&p);
		p.device = params.device;
		p.image_w = params.imageSize.width;
		p.image_h = params.imageSize.height;
		p.patch_w = params.patchCompensateSize.width;
		p.patch_h = params.patchCompensateSize.height;
		p.tv_weight = params.compensateTVweight;
		p.tv_huber = params.compensateTVHuberLoss;
		p.scale = params.compensateScale;
		p.min_events = params.compensateMinNumEvents;
		p.loss = params.loss;
		p.grad = params.grad;
		p.max_events = params.maxNumEventsToStore > 0 ? params.maxNumEventsToStore : 1;
		p.max_windows = 1;
		const int rc = ebo_create(&p, &ctx_);
		if (rc != EBO_OK)
		{
			ctx_ = nullptr;
			fail(rc, ebo_last_error(nullptr));
			numPatchesX_ = p.patch_w > 0 ? p.image_w / p.patch_w : 0;
			numPatchesY_ = p.patch_h > 0 ? p.image_h / p.patch_h : 0;
		}
		else
		{
			ebo_grid(ctx_, &numPatchesX_, &numPatchesY_);
		}
		compensatedEventImage_ = Mat64(p.image_h, p.image_w);
		integratedEventImage_ = Mat64(p.image_h, p.image_w);
		motionField_.assign(static_cast<size_t>(p.image_h) * p.image_w * 2, 0.0f);
		patchFlows_.assign(static_cast<size_t>(numPatchesX_) * numPatchesY_ * 2, 0.0);
		lastCompensation = common::timestamp_t(0);
	}
	~FeatureDetector()
	{
		if (ctx_)
		{
			ebo_destroy(ctx_);
		}
	}
	FeatureDetector(const FeatureDetector&) = delete;
	FeatureDetector& operator=(const FeatureDetector&) = delete;

	// feature_detector.cpp:621-628
	void addEvent(const common::EventSample& event)
	{
		lastEvents_.push_back(event);
		while (lastEvents_.size() > params_.maxNumEventsToStore)
		{
			lastEvents_.pop_front();
		}
	}

	void clearEvents() { lastEvents_.clear(); }
	std::list<common::EventSample> const& getEvents() { return lastEvents_; }

	// feature_detector.cpp:298-464: solve the per-patch flows, build the final image.
	void compensateEventsContrast(const std::list<common::EventSample>& events)
	{
		if (events.empty())
		{
			return;
		}
		lastCompensation = events.back().timestamp;  // :307
		const std::vector<ebo_event> ev = common::toEboEvents(events);
		ebo_solver_opts o;
		ebo_default_solver(&o);
		o.mode = params_.solveMode;
		if (!check(ebo_compensate_events_contrast(ctx_, ev.data(), ev.size(), &o, patchFlows_.data(),
												  compensatedEventImage_.ptr(), &lastSummary_)))
		{
			return;
		}
		// :418-431 the reference stores the flows at the patch corners of its motion field
		for (int y = 0; y < numPatchesY_; ++y)
		{
			for (int x = 0; x < numPatchesX_; ++x)
			{
				const size_t px = static_cast<size_t>(y) * params_.patchCompensateSize.height *
									  params_.imageSize.width +
								  static_cast<size_t>(x) * params_.patchCompensateSize.width;
				motionField_[2 * px] = static_cast<float>(patchFlows_[2 * (y * numPatchesX_ + x)]);
				motionField_[2 * px + 1] = static_cast<float>(patchFlows_[2 * (y * numPatchesX_ + x) + 1]);
			}
		}
	}

	// feature_detector.cpp:466-482
	void integrateEvents(const std::list<common::EventSample>& events)
	{
		const std::vector<ebo_event> ev = common::toEboEvents(events);
		if (ev.empty())
		{
			integratedEventImage_ = Mat64(params_.imageSize.height, params_.imageSize.width);
			return;
		}
		if (check(ebo_set_window(ctx_, ev.data(), ev.size())))
		{
			check(ebo_count_image(ctx_, EBO_COUNT_INTEGRATED, nullptr, integratedEventImage_.ptr()));
		}
	}

	// feature_detector.cpp:243-296.  With patch trajectories installed (setPatchTrajectories)
	// this is the reference's sequence: mid timestamp (:247-248), interpolateMotionField or
	// initMotionField by optimizeFlowTV (:253-260), warp loop (:268-295).  Without
	// trajectories the field installed by setMotionField() is used as it is.
	void compensateEvents(const std::list<common::EventSample>& events)
	{
		if (events.empty())
		{
			return;
		}
		lastCompensation = events.back().timestamp;  // :250
		if (!trajectories_.empty())
		{
			const auto timestamp = common::timestamp_t(static_cast<int32_t>(
				(events.front().timestamp + events.back().timestamp).count() * 0.5));
			if (params_.optimizeFlowTV)
			{
				interpolateMotionField(timestamp);
			}
			else
			{
				initMotionField(timestamp);
			}
		}
		const std::vector<ebo_event> ev = common::toEboEvents(events);
		if (check(ebo_set_window(ctx_, ev.data(), ev.size())))
		{
			check(ebo_count_image(ctx_, EBO_COUNT_FIELD, motionField_.data(), compensatedEventImage_.ptr()));
		}
	}

	// feature_detector.cpp:53-142.  The reference reads the trajectories of its tracked
	// Patches (Patch::getTrajectory(), frame-based tracker, outside this path); hand them in with
	// setPatchTrajectories().  useAverageFlow selects average vs nearest fill.
	using Trajectory = std::vector<common::Sample<common::Point2d>>;
	void setPatchTrajectories(const std::vector<Trajectory>& trajectories) { trajectories_ = trajectories; }
	void initMotionField(const common::timestamp_t timestamp)
	{
		std::vector<size_t> off(1, 0);
		std::vector<double> xy;
		std::vector<int64_t> tt;
		for (const auto& tr : trajectories_)
		{
			for (const auto& s : tr)
			{
				xy.push_back(s.value.x);
				xy.push_back(s.value.y);
				tt.push_back(s.timestamp.count());
			}
			off.push_back(tt.size());
		}
		check(ebo_init_motion_field(ctx_, timestamp.count(), params_.useAverageFlow ? 1 : 0,
									static_cast<int>(trajectories_.size()), off.data(), xy.data(), tt.data(),
									motionField_.data(), nullptr, nullptr));
	}

	// feature_detector.cpp:144-241: initMotionField, then the per-pixel TV problem (useL1
	// selects HuberLoss(1e-5)) solved on the device; getLastFieldSummary() is what the
	// reference logs as summary.BriefReport() (:228).
	void interpolateMotionField(const common::timestamp_t timestamp)
	{
		initMotionField(timestamp);
		if (status_ != EBO_OK)
		{
			return;
		}
		check(ebo_interpolate_motion_field(ctx_, params_.useL1 ? 1 : 0, nullptr, motionField_.data(),
										   &lastFieldSummary_, nullptr));
	}
	const ebo_summary& getLastFieldSummary() const { return lastFieldSummary_; }

	// float32 [height][width][2], the at<cv::Vec2f> view of the reference's motionField_
	void setMotionField(const std::vector<float>& field)
	{
		if (field.size() != motionField_.size())
		{
			if (params_.errorPolicy == DetectorParams::ERRORS_THROW)
			{
				throw std::invalid_argument("motion field must be height*width*2 floats");
			}
			fail(EBO_ERR_ARG, "motion field must be height*width*2 floats");
			return;
		}
		status_ = EBO_OK;
		motionField_ = field;
	}
	const std::vector<float>& getMotionField() const { return motionField_; }

	Mat64 const& getCompensatedEventImage() { return compensatedEventImage_; }
	Mat64 const& getIntegratedEventImage() { return integratedEventImage_; }
	common::timestamp_t const& getLastCompensation() { return lastCompensation; }

	// The solved per-patch flows mf[P][2] (leaked by the reference, :318) and the
	// solver's summary (the reference logs summary.BriefReport(), :416).
	const std::vector<double>& getPatchFlows() const { return patchFlows_; }
	const ebo_summary& getLastSummary() const { return lastSummary_; }
	int numPatchesX() const { return numPatchesX_; }
	int numPatchesY() const { return numPatchesY_; }
	void setParams(const DetectorParams& params) { params_.maxNumEventsToStore = params.maxNumEventsToStore; }
	ebo_ctx* handle() { return ctx_; }

	// DetectorParams::ERRORS_STATUS: the EBO_* code and message of the LAST call (EBO_OK / "" after
	// a call that succeeded).
	int status() const { return status_; }
	bool ok() const { return status_ == EBO_OK; }
	const std::string& lastError() const { return lastError_; }

   private:
	bool check(int rc)
	{
		if (ctx_ == nullptr)
		{
			fail(EBO_ERR_NO_DEVICE, "no device context (construction failed)");
			return false;
		}
		if (rc != EBO_OK)
		{
			fail(rc, ebo_last_error(ctx_));
			return false;
		}
		status_ = EBO_OK;
		lastError_.clear();
		return true;
	}
	void fail(int rc, const char* what)
	{
		status_ = rc;
		lastError_ = std::string("tracker::FeatureDetector: ") + (what ? what : "");
		if (params_.errorPolicy == DetectorParams::ERRORS_THROW)
		{
			throw std::runtime_error(lastError_);
		}
	}
	int status_ = EBO_OK;
	std::string lastError_;

	DetectorParams params_;
	ebo_summary lastFieldSummary_{};
	ebo_ctx* ctx_ = nullptr;
	int numPatchesX_ = 0, numPatchesY_ = 0;
	Mat64 compensatedEventImage_;
	Mat64 integratedEventImage_;
	std::vector<float> motionField_;
	std::vector<Trajectory> trajectories_;
	std::vector<double> patchFlows_;
	std::list<common::EventSample> lastEvents_;
	common::timestamp_t lastCompensation;
	ebo_summary lastSummary_ = {};
};

}  // namespace tracker
