// feature_tracker/patch.h — tracker::Patch with the reference's names
// (implementation/feature_tracker/include/feature_tracker/patch.h:15-160,
// src/patch.cpp:8-63,156-330), for the per-feature tracker path: the state
// a tracked feature carries between two Optimizer::optimize calls.  Bookkeeping — the event
// window, the rect, the warp and the trajectory — on the host; the per-pixel work runs on the
// device through the C ABI: for all patches of a round at once when tracker::Optimizer /
// FeatureDetector drive it (their batched calls are what the product uses), and for ONE patch
// through the reference's own per-patch members (integrateEvents, integrateMotionCompensatedEvents,
// warpImage, setGrad: patch.h:24-26,46,80), which need a device context -- the one the patch was
// bound to (bind) or the process-wide default (Patch::setDefaultContext) -- and throw without one:
// there is no host implementation.  FeatureDetector does not bind the patches it holds (a copy handed
// to the front end would carry a pointer into the detector): it never needs the per-patch form.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <list>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../common/data_types.h"
#include "types.h"

namespace tracker
{
class Patch
{
   public:
	// patch.cpp:8-16
	Patch(const Corner& corner, int extent, const common::timestamp_t& timestamp) : currentTimestamp_(timestamp)
	{
		patch_ = Rect2d(corner.x - extent, corner.y - extent, 2 * extent + 1, 2 * extent + 1);
		init();
	}

	// patch.cpp:18-36
	void init()
	{
		init_ = false;
		lost_ = false;
		trackId_ = -1;
		numOfEvents_ = 75;
		initPoint_ = toCorner();
		counter_ = 0;
		integratedNabla_ = Mat64(static_cast<int>(patch_.height), static_cast<int>(patch_.width));
		predictedNabla_ = Mat64(static_cast<int>(patch_.height), static_cast<int>(patch_.width));  // patch.cpp:28: zeros
		motionCompensatedIntegratedNabla_ = Mat64(static_cast<int>(patch_.height), static_cast<int>(patch_.width));
		timeWithoutUpdate_ = common::timestamp_t(static_cast<int64_t>(1e7));
		initTime_ = currentTimestamp_;
		addTrajectoryPosition();
	}

	// patch.cpp:38-47: newest event at the front, window capped at numOfEvents_
	void addEvent(const common::EventSample& event)
	{
		events_.push_front(event);
		while (events_.size() > numOfEvents_)
		{
			events_.pop_back();
		}
		counter_++;
	}

	// patch.cpp:49-63
	void updatePatchRect()
	{
		const auto warpInv = warp_.inverse().matrix2x3();
		const double newCenterX = warpInv(0, 0) * initPoint_.x + warpInv(0, 1) * initPoint_.y + warpInv(0, 2);
		const double newCenterY = warpInv(1, 0) * initPoint_.x + warpInv(1, 1) * initPoint_.y + warpInv(1, 2);
		const int extentX = static_cast<int>((patch_.width - 1) / 2);
		const int extentY = static_cast<int>((patch_.height - 1) / 2);
		patch_ = Rect2d(newCenterX - extentX, newCenterY - extentY, 2 * extentX + 1, 2 * extentY + 1);
	}

	// patch.cpp:255-266
	void setCorner(const Corner& corner, const common::timestamp_t& timestamp)
	{
		patch_ = Rect2d(corner.x - (patch_.width - 1) / 2., corner.y - (patch_.height - 1) / 2., patch_.width, patch_.height);
		initPoint_ = toCorner();
		init_ = false;
		currentTimestamp_ = timestamp;
		addTrajectoryPosition();
		resetBatch();
	}

	// patch.cpp:132-154, called per patch at feature_detector.cpp:508,615: the predicted gradient patch
	//     predictedNabla_ = -warpedGradX(patch_) cos(flowDir_) - warpedGradY(patch_) sin(flowDir_),
	// the gradient images warped by cv::warpAffine(..., warp_.matrix2x3(), cv::WARP_INVERSE_MAP).  The
	// reference's Patch holds gradX_ / gradY_ (setGrad); here they live on the device in `ctx`
	// (ebo_optimizer_set_grad, what tracker::Optimizer::setGrad / TrackedPatches::setGradients install)
	// and the warp runs there (ebo_patch_warp_image).  A rect that touches the image border leaves
	// predictedNabla_ as it was (:145-150) -- with the reference's own test set-up (patch_test.cpp:62-108:
	// an 11x11 image and a patch of extent 5 at its centre) that is every call, and the image stays 0.
	// false: the device call failed (ebo_last_error(ctx)).
	bool warpImage(ebo_ctx* ctx)
	{
		const double rect[4] = {patch_.x, patch_.y, patch_.width, patch_.height};
		const size_t off = 0;
		Mat64 out(static_cast<int>(std::nearbyint(patch_.height)), static_cast<int>(std::nearbyint(patch_.width)));
		int32_t updated = 0;
		if (ebo_patch_warp_image(ctx, 1, rect, warp_.data(), &flowDir_, &off, out.ptr(), &updated) != EBO_OK)
		{
			return false;
		}
		if (updated)
		{
			predictedNabla_ = out;
		}
		return true;
	}
	Mat64 const& getPredictedNabla() const { return predictedNabla_; }
	void setPredictedNabla(const Mat64& m) { predictedNabla_ = m; }

	// ---- the device context of the per-patch calls below --------------------------------------
	void bind(ebo_ctx* ctx) { ctx_ = ctx; }
	static void setDefaultContext(ebo_ctx* ctx) { defaultContext() = ctx; }
	ebo_ctx* context() const { return ctx_ ? ctx_ : defaultContext(); }

	// patch.cpp:65-85: the signed count image of the patch's events, currentTimestamp_ = int32 mid time of
	// the window, timeLastUpdate_ = its oldest event (one launch for this one patch: ebo_patch_integrate)
	void integrateEvents()
	{
		if (events_.empty())
		{
			throw std::invalid_argument("tracker::Patch::integrateEvents: no events (the reference reads events_.front())");
		}
		const std::vector<ebo_event> ev = common::toEboEvents(events_);
		const size_t evOff[2] = {0, ev.size()}, nablaOff = 0;
		const double rect[4] = {patch_.x, patch_.y, patch_.width, patch_.height};
		Mat64 m(static_cast<int>(patch_.height), static_cast<int>(patch_.width));
		int64_t cur = 0, last = 0;
		check(ebo_patch_integrate(need("integrateEvents"), ev.data(), evOff, 1, rect, &nablaOff, m.ptr(), &cur, &last), "integrateEvents");
		integratedNabla_ = m;
		currentTimestamp_ = common::timestamp_t(cur);
		timeLastUpdate_ = common::timestamp_t(last);
	}

	// patch.cpp:87-130: the count image of the events moved along the last trajectory segment to the mid time;
	// untouched when the trajectory is shorter than two points, there are no events or the time test fails
	void integrateMotionCompensatedEvents()
	{
		if (trajectory_.size() < 2 || events_.empty())
		{
			return;
		}
		const std::vector<ebo_event> ev = common::toEboEvents(events_);
		const size_t evOff[2] = {0, ev.size()}, nablaOff = 0;
		const double rect[4] = {patch_.x, patch_.y, patch_.width, patch_.height};
		const auto& pre = trajectory_[trajectory_.size() - 2];
		const auto& lastPt = trajectory_[trajectory_.size() - 1];
		const double traj[6] = {pre.value.x,    pre.value.y,    static_cast<double>(pre.timestamp.count()),
								lastPt.value.x, lastPt.value.y, static_cast<double>(lastPt.timestamp.count())};
		const int64_t mid = currentTimestamp_.count();
		Mat64 m(static_cast<int>(patch_.height), static_cast<int>(patch_.width));
		int32_t updated = 0;
		check(ebo_patch_integrate_mc(need("integrateMotionCompensatedEvents"), ev.data(), evOff, 1, rect, traj, &mid, &nablaOff,
									 m.ptr(), &updated),
			  "integrateMotionCompensatedEvents");
		if (updated)
		{
			motionCompensatedIntegratedNabla_ = m;
		}
	}

	// patch.cpp:132-154 with the reference's signature: the gradient pair of setGrad (or of whoever installed one on
	// the context: Optimizer::setGrad, TrackedPatches::setGradients), warped on the patch's context
	void warpImage()
	{
		if (!warpImage(need("warpImage")))
		{
			throw std::runtime_error(std::string("tracker::Patch::warpImage: ") + ebo_last_error(context()));
		}
	}

	// patch.cpp:280-294.  The reference's Patch keeps two cv::Mat headers on the frame's gradient images; here
	// the pair lives ONCE per device context (ebo_optimizer_set_grad), so setGrad installs it there for every
	// patch of that context, and keeps a shared copy for getGradX / getGradY.  FeatureDetector does not go
	// through here per patch (it installs a frame's pair once: setGradients).
	void setGrad(const Mat64& gradX, const Mat64& gradY)
	{
		if (gradX.rows != gradY.rows || gradX.cols != gradY.cols)
		{
			throw std::invalid_argument("tracker::Patch::setGrad: gradX and gradY differ in size");
		}
		check(ebo_optimizer_set_grad(need("setGrad"), gradX.ptr(), gradY.ptr()), "setGrad");
		gradX_ = std::make_shared<const Mat64>(gradX);
		gradY_ = std::make_shared<const Mat64>(gradY);
	}
	void setGrad(std::shared_ptr<const Mat64> gradX, std::shared_ptr<const Mat64> gradY)  // already installed: no copy, no upload
	{
		gradX_ = std::move(gradX);
		gradY_ = std::move(gradY);
	}
	Mat64 const& getGradX() const { return gradX_ ? *gradX_ : emptyMat(); }
	Mat64 const& getGradY() const { return gradY_ ? *gradY_ : emptyMat(); }

	// patch.cpp:156-159: integratedNabla_ / cv::norm(integratedNabla_) (L2; an all-zero patch gives NaN as there).
	// A getter for callers that want the image; Optimizer::optimize normalises on the device (ebo_optimizer_solve).
	Mat64 getNormalizedIntegratedNabla() const
	{
		Mat64 out(integratedNabla_.rows, integratedNabla_.cols);
		const size_t n = static_cast<size_t>(integratedNabla_.rows) * integratedNabla_.cols;
		const double* a = integratedNabla_.ptr();
		double ss = 0.0;
		for (size_t i = 0; i < n; ++i)
		{
			ss += a[i] * a[i];
		}
		const double inv = 1.0 / std::sqrt(ss);  // `Mat / double` is OpenCV's MatExpr a * (1 / s)
		for (size_t i = 0; i < n; ++i)
		{
			out.ptr()[i] = a[i] * inv;
		}
		return out;
	}

	// patch.h:57,78 / patch.cpp:268-273
	const Mat64& getCostMap() const { return costMap_; }
	void setCostMap(const Mat64& costMap) { costMap_ = costMap; }
	Rect2d getInitPatch() const
	{
		return Rect2d(initPoint_.x - (patch_.width - 1) / 2., initPoint_.y - (patch_.height - 1) / 2., patch_.width, patch_.height);
	}

	void resetBatch() { counter_ = 0; }
	void addTrajectoryPosition() { trajectory_.push_back({toCorner(), currentTimestamp_}); }
	void addFinalCost(double finalCost) { finalCosts_.emplace_back(finalCost); }
	Corner toCorner() const { return Corner(patch_.x + (patch_.width - 1) / 2., patch_.y + (patch_.height - 1) / 2.); }
	bool isInPatch(const common::Point2i& point) const { return patch_.contains(point); }
	bool isReady() const { return counter_ >= 30 && events_.size() >= numOfEvents_; }
	// how many more addEvent calls until isReady() (0: ready now); what a batched caller asks
	// the device router for
	size_t eventsUntilReady() const
	{
		const size_t byCounter = counter_ >= 30 ? 0 : 30 - counter_;
		const size_t bySize = events_.size() >= numOfEvents_ ? 0 : numOfEvents_ - events_.size();
		return std::max(byCounter, bySize);
	}
	bool isLost() const { return lost_; }
	bool isInit() const { return init_; }

	common::EventSequence const& getEvents() const { return events_; }
	Mat64 const& getIntegratedNabla() const { return integratedNabla_; }
	Mat64 const& getCompenatedIntegratedNabla() const { return motionCompensatedIntegratedNabla_; }
	Rect2d const& getPatch() const { return patch_; }
	TrackId getTrackId() const { return trackId_; }
	const common::Pose2d& getWarp() const { return warp_; }
	float getFlow() const { return static_cast<float>(flowDir_); }  // float, as the reference (patch.h:56)
	double getFlowDir() const { return flowDir_; }                 // the member itself (what warpImage reads)
	std::vector<common::Sample<common::Point2d>> const& getTrajectory() const { return trajectory_; }
	size_t getNumOfEvents() const { return numOfEvents_; }
	common::timestamp_t getCurrentTimestamp() const { return currentTimestamp_; }
	common::timestamp_t getTimeWithoutUpdate() const { return timeWithoutUpdate_; }
	common::timestamp_t getTimeLastUpdate() const { return timeLastUpdate_; }
	common::timestamp_t getInitTime() const { return initTime_; }
	const std::vector<double>& getFinalCosts() const { return finalCosts_; }

	void setLost() { lost_ = true; }
	void setNumOfEvents(size_t numOfEvents)
	{
		numOfEvents_ = std::max(minNumOfEvents_, numOfEvents);
		numOfEvents_ = std::min(numOfEvents_, maxNumOfEvents_);
	}
	void setTrackId(TrackId trackId) { trackId_ = trackId; }
	void setFlowDir(const double flowDir)
	{
		flowDir_ = flowDir;
		init_ = true;
	}
	void setWarp(const common::Pose2d& warp)
	{
		warp_ = warp;
		init_ = true;
	}
	void setTs(const common::timestamp_t& ts) { currentTimestamp_ = ts; }
	void setTimeWithoutUpdate(const common::timestamp_t& t) { timeWithoutUpdate_ = t; }
	void setIntegratedNabla(const Mat64& m) { integratedNabla_ = m; }
	void setMotionCompensatedIntegratedNabla(const Mat64& m) { motionCompensatedIntegratedNabla_ = m; }
	void setTimestamps(common::timestamp_t current, common::timestamp_t lastUpdate)
	{
		currentTimestamp_ = current;  // what integrateEvents leaves (patch.cpp:78-84)
		timeLastUpdate_ = lastUpdate;
	}

   private:
	static ebo_ctx*& defaultContext()
	{
		static ebo_ctx* c = nullptr;
		return c;
	}
	static const Mat64& emptyMat()
	{
		static const Mat64 m;
		return m;
	}
	ebo_ctx* need(const char* what) const
	{
		ebo_ctx* c = context();
		if (!c)
		{
			throw std::runtime_error(std::string("tracker::Patch::") + what +
									 ": no device context (Patch::bind / Patch::setDefaultContext; there is no host implementation)");
		}
		return c;
	}
	void check(int rc, const char* what) const
	{
		if (rc != EBO_OK)
		{
			throw std::runtime_error(std::string("tracker::Patch::") + what + ": " + ebo_last_error(context()));
		}
	}

	ebo_ctx* ctx_ = nullptr;
	std::shared_ptr<const Mat64> gradX_, gradY_;
	Mat64 costMap_;
	bool init_ = false;
	bool lost_ = false;
	Rect2d patch_;
	common::Point2d initPoint_;
	TrackId trackId_ = -1;
	common::timestamp_t currentTimestamp_{0};
	common::timestamp_t timeLastUpdate_{0};
	common::timestamp_t timeWithoutUpdate_{0};
	common::timestamp_t initTime_{0};
	common::EventSequence events_;
	size_t numOfEvents_ = 75;
	size_t minNumOfEvents_ = 100;
	size_t maxNumOfEvents_ = 300;
	size_t counter_ = 0;
	Mat64 integratedNabla_;
	Mat64 predictedNabla_;
	Mat64 motionCompensatedIntegratedNabla_;
	double flowDir_ = 0.0;
	common::Pose2d warp_;
	std::vector<common::Sample<common::Point2d>> trajectory_;
	std::vector<double> finalCosts_;
};

using Patches = std::list<Patch>;  // patch.h:132 (a list: element addresses stay valid while patches are erased)

}  // namespace tracker
