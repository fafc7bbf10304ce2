// feature_tracker/feature_detector.h — tracker::FeatureDetector and tracker::DetectorParams: the
// class the reference's unchanged front end holds
// (implementation/feature_tracker/include/feature_tracker/feature_detector.h:10-120,
// src/feature_detector.cpp).  Every member the front end calls is here under the reference's name:
//
//   tools::Evaluator::eventCallback (tools/evaluator/src/evaluator.cpp:32-45)
//       tracker_->addEvent(sample);
//       tracker_->updatePatches(sample);
//       if (window due) {
//           tracker_->compensateEventsContrast(tracker_->getEvents());
//           tracker_->integrateEvents(tracker_->getEvents());
//           tracker_->clearEvents();
//       }
//   ~Evaluator (:15-21)    preExit, getArchivedPatches, getOptimizedFinalCosts
//   getPatches (:23-30), reset (:101-118: DetectorParams::drawImages / imageSize), setTrackerParams (:120-123)
//   visual_odometry::Keyframe(tracker_->getPatches(), ts) (visual_odometry/src/keyframe.cpp:5-14)
//
// Everything per-event / per-pixel forwards to the C ABI (include/ebo.h); the host side here is the
// reference's bookkeeping (patch association, archive, optimizer user counts).  The three OpenCV-only
// pieces of newImage — cv::goodFeaturesToTrack (:568-583), log image + cv::Sobel (:713-731) and
// cv::calcOpticalFlowPyrLK (flow_estimator.cpp:86-108) — are outside the event-warping path: they are
// taken as FrontEndHooks (three lambdas holding the reference's own OpenCV calls); without them
// newImage / extractPatches / detectFeatures report EBO_ERR_UNSUPPORTED through the error policy.
//
// Result images are CV_64F-like (tracker::Mat64, row-major doubles, rows x cols = imageSize), valid
// until the next call — as the cv::Mat const& of the reference.  With OpenCV available, wrap
// without a copy:  cv::Mat view(m.rows, m.cols, CV_64F, const_cast<double*>(m.ptr()));
#pragma once

#include <cmath>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../common/data_types.h"
#include "optimizer.h"
#include "patch.h"
#include "tracked_patches.h"
#include "types.h"

namespace tracker
{
// Every field of the reference's DetectorParams (feature_detector.h:10-31), same names, order and
// defaults.  qualityLevel / minDistance / blockSize are read only by the detectFeatures hook's owner
// (cv::goodFeaturesToTrack); the last block selects what the north star adds.
struct DetectorParams
{
	double qualityLevel = 0.01;
	double minDistance = 10;
	double associationDistance = 5;
	int32_t patchExtent = 12;
	int32_t blockSize = 3;
	Size imageSize = {240, 180};
	bool drawImages = false;
	OptimizerParams optimizerParams = {};
	int initNumEvents = 75;
	unsigned long maxNumEventsToStore = 15000;
	bool useAverageFlow = true;
	bool optimizeFlowTV = true;
	bool useL1 = false;
	Size patchCompensateSize = {20, 20};
	double compensateTVweight = 1e3;
	double compensateTVHuberLoss = 10;
	double compensateScale = 1e-3;
	unsigned int compensateMinNumEvents = 100;
	size_t maxPatches = 100;

	// updatePatches(event) is the front end's per-event call; one optimisation launch per ready patch and event is
	// what the device is worst at (100 patches x 6000 events: 762 ms per event against 19 ms as one chunk).  Patches do
	// not interact and nothing outside reads them between two events, so the detector keeps the events and brings the
	// patches up to date -- all of them in lock-step rounds, bit-equal to the per-event sequence -- when `eventBatch`
	// events have gathered or when anything is read or changed (getPatches, getArchivedPatches, preExit, newImage,
	// setPatches, setParams, getOptimizedFinalCosts, associatePatches, updateNumOfEvents).  A reference obtained from
	// getPatches() is current at the time of that call.  0 or 1: every event at once, as the reference.
	size_t eventBatch = 4096;

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

// The OpenCV-side pieces of FeatureDetector::newImage, which are not on the event-warping path.  With
// OpenCV present an integrator fills them with the reference's own calls.
struct FrontEndHooks
{
	// detectFeatures (:568-583): cv::goodFeaturesToTrack under mask_ (a border of patchExtent)
	std::function<Corners(const common::Image8&)> detectFeatures;
	// getLogImage + getGradients (:713-731): log(I/255 + 0.1), cv::Sobel(image / 8, CV_64F, dx, dy, 3)
	std::function<void(const common::Image8&, Mat64& gradX, Mat64& gradY)> gradients;
	// FlowEstimator::getFlow (flow_estimator.cpp:86-108): cv::calcOpticalFlowPyrLK from the previous image to
	// this one for ONE point (cv::Point2f in and out); false = status 0.  Called from the second image on.
	std::function<bool(float x, float y, float& nextX, float& nextY)> flow;
	double patchTimeWithoutUpdateScale = 1e6;  // FlowEstimatorParams (flow_estimator.h:16)
};

class FeatureDetector
{
   public:
	explicit FeatureDetector(const DetectorParams& params) : params_(params)
	{
		createContext();
		tracked_.reset(new TrackedPatches(ctx_, params_.imageSize, params_.initNumEvents));
		reset();
	}
	~FeatureDetector()
	{
		pending_.clear();  // events nobody asked the result of
		tracked_.reset();
		if (ctx_)
		{
			ebo_destroy(ctx_);
		}
	}
	FeatureDetector(const FeatureDetector&) = delete;
	FeatureDetector& operator=(const FeatureDetector&) = delete;

	// feature_detector.cpp:484-491
	void preExit()
	{
		flushPatches();
		for (const Patch& patch : tracked_->getPatches())
		{
			archivedPatches_.push_back(patch);
		}
	}

	void setFrontEndHooks(const FrontEndHooks& hooks) { hooks_ = hooks; }

	// feature_detector.cpp:493-541.  Needs the detectFeatures and gradients hooks (EBO_ERR_UNSUPPORTED
	// without them); without the flow hook new patches stay un-initialised, as before the reference's
	// second image (flow_estimator.cpp:29-32).
	void newImage(const common::ImageSample& image)
	{
		if (!hooks_.detectFeatures || !hooks_.gradients)
		{
			fail(EBO_ERR_UNSUPPORTED,
				 "newImage needs the OpenCV front end (goodFeaturesToTrack, Sobel, calcOpticalFlowPyrLK): "
				 "install FrontEndHooks");
			return;
		}
		flushPatches();
		guarded([&] {
			extractPatchesImpl(image);
			// flowEstimator_->addImage / getFlowPatches (flow_estimator.cpp:16-85)
			if (imageCounter_ < 2)
			{
				imageCounter_++;
			}
			if (imageCounter_ == 2 && hooks_.flow)
			{
				flowPatches();
			}
			std::vector<Patch*> all;
			for (Patch& patch : tracked_->getPatches())
			{
				all.push_back(&patch);
			}
			tracked_->updateNumOfEvents(all);
			if (params_.drawImages)
			{
				warpByFrame(all);
			}
			auto& optimizers = tracked_->optimizers();
			Patches& patches = tracked_->getPatches();
			for (auto patchIt = patches.begin(); patchIt != patches.end();)
			{
				if (patchIt->isLost())
				{
					archivedPatches_.push_back(*patchIt);
					auto opt = optimizers.find(patchIt->getInitTime().count());
					if (opt != optimizers.end() && opt->second)
					{
						opt->second->deleteUser();
					}
					patchIt = patches.erase(patchIt);
					continue;
				}
				++patchIt;
			}
			for (auto optIt = optimizers.begin(); optIt != optimizers.end();)
			{
				if (!optIt->second || !optIt->second->isUsed())
				{
					optIt = optimizers.erase(optIt);
				}
				else
				{
					++optIt;
				}
			}
		});
	}

	// feature_detector.cpp:543-566
	void extractPatches(const common::ImageSample& image)
	{
		if (!hooks_.detectFeatures || !hooks_.gradients)
		{
			fail(EBO_ERR_UNSUPPORTED, "extractPatches needs the detectFeatures and gradients hooks");
			return;
		}
		flushPatches();
		guarded([&] { extractPatchesImpl(image); });
	}

	// feature_detector.cpp:568-583
	Corners detectFeatures(const common::Image8& image)
	{
		if (!hooks_.detectFeatures)
		{
			fail(EBO_ERR_UNSUPPORTED, "detectFeatures is cv::goodFeaturesToTrack: install FrontEndHooks::detectFeatures");
			return Corners();
		}
		status_ = EBO_OK;
		return hooks_.detectFeatures(image);
	}

	// feature_detector.cpp:585-619: the reference's per-event call (DetectorParams::eventBatch: the event is kept
	// and the patches are brought up to date in chunks; every reader sees them current)
	void updatePatches(const common::EventSample& event)
	{
		if (params_.eventBatch <= 1)
		{
			guarded([&] { tracked_->updatePatches(event); });
			return;
		}
		pending_.push_back(event);
		if (pending_.size() >= params_.eventBatch)
		{
			flushPatches();
		}
	}
	// the same for a chunk of the stream: all patches advance in lock-step rounds, one launch per stage
	// (tracked_patches.h); per patch the sequence of addEvent / optimize calls is the per-event one
	void updatePatches(const std::vector<common::EventSample>& chunk)
	{
		flushPatches();
		guarded([&] { tracked_->updatePatches(chunk); });
	}
	// brings the tracked patches up to the last event handed to updatePatches(event)
	void flushPatches()
	{
		if (pending_.empty())
		{
			return;
		}
		std::vector<common::EventSample> chunk;
		chunk.swap(pending_);
		guarded([&] { tracked_->updatePatches(chunk); });
	}

	// feature_detector.cpp:630-664.  The reference dereferences optimizers_[timestamp] (created by
	// extractPatches); called without one (its own associatedPatchesTest does) the user count is skipped.
	void associatePatches(Patches& newPatches, const common::timestamp_t& timestamp)
	{
		flushPatches();
		Patches& patches = tracked_->getPatches();
		for (auto& patch : patches)
		{
			const auto corner = patch.toCorner();
			for (auto& newPatch : newPatches)
			{
				const auto newCorner = newPatch.toCorner();
				const double dx = corner.x - newCorner.x, dy = corner.y - newCorner.y;
				if (newPatch.getTrackId() == -1 && std::sqrt(dx * dx + dy * dy) < params_.associationDistance)
				{
					newPatch.setTrackId(patch.getTrackId());
					break;
				}
			}
			patch.setTs(timestamp);
			patch.addTrajectoryPosition();
		}
		auto& optimizers = tracked_->optimizers();
		for (auto& newPatch : newPatches)
		{
			if (newPatch.getTrackId() == -1 && patches.size() < params_.maxPatches)
			{
				newPatch.setTrackId(static_cast<TrackId>(nextTrackId_));
				patches.push_back(newPatch);  // newPatch.setGrad(gradX_, gradY_): the frame's Optimizer holds them
				auto opt = optimizers.find(timestamp.count());
				if (opt != optimizers.end() && opt->second)
				{
					opt->second->addUser();
				}
				nextTrackId_++;
			}
		}
	}

	// feature_detector.cpp:666-711 (the estimate on the device against the latest frame's gradients)
	void updateNumOfEvents(Patch& patch)
	{
		flushPatches();
		guarded([&] { tracked_->updateNumOfEvents(patch); });
	}

	void setPatches(const Patches& patches)  // feature_detector.h:67
	{
		flushPatches();
		tracked_->setPatches(patches);
	}
	void setTrackId(TrackId trackId) { nextTrackId_ = static_cast<size_t>(trackId); }
	Patches const& getPatches() const
	{
		const_cast<FeatureDetector*>(this)->flushPatches();  // (logically const: the events were handed over before)
		return tracked_->getPatches();
	}
	Patches& getPatches()
	{
		flushPatches();
		return tracked_->getPatches();
	}
	Corners const& getFeatures() const { return corners_; }
	Patches const& getArchivedPatches() const { return archivedPatches_; }
	// feature_detector.h:80-83: optimizers_.begin()->second->getFinalCosts() (undefined there when no
	// optimizer exists; empty here).  optimizers_ is ordered by frame time here: begin() = oldest frame in use.
	std::vector<tracker::OptimizerFinalLoss> getOptimizedFinalCosts() const
	{
		const_cast<FeatureDetector*>(this)->flushPatches();
		const auto& optimizers = tracked_->optimizers();
		if (optimizers.empty() || !optimizers.begin()->second)
		{
			return {};
		}
		return optimizers.begin()->second->getFinalCosts();
	}
	// optimizers_[image.timestamp.count()] for callers that bring patches in through setPatches
	void setOptimizer(const common::timestamp_t& initTime, std::shared_ptr<Optimizer> optimizer)
	{
		flushPatches();
		tracked_->setOptimizer(initTime, std::move(optimizer));
	}
	// gradX_ / gradY_ (:554-555) for callers that do not go through newImage
	void setGradients(const Mat64& gradX, const Mat64& gradY)
	{
		guarded([&] { tracked_->setGradients(gradX, gradY); });
	}
	TrackedPatches& tracked()
	{
		flushPatches();
		return *tracked_;
	}

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
	// feature_detector.cpp:733-741: params_ = params; every optimizer takes optimizerParams; reset().
	// The device context is re-created when a field it was built from changed.
	void setParams(const tracker::DetectorParams& params)
	{
		flushPatches();
		const bool rebuild = params.imageSize.width != params_.imageSize.width ||
							 params.imageSize.height != params_.imageSize.height ||
							 params.patchCompensateSize.width != params_.patchCompensateSize.width ||
							 params.patchCompensateSize.height != params_.patchCompensateSize.height ||
							 params.compensateTVweight != params_.compensateTVweight ||
							 params.compensateTVHuberLoss != params_.compensateTVHuberLoss ||
							 params.compensateScale != params_.compensateScale ||
							 params.compensateMinNumEvents != params_.compensateMinNumEvents ||
							 params.maxNumEventsToStore != params_.maxNumEventsToStore || params.loss != params_.loss ||
							 params.grad != params_.grad || params.device != params_.device;
		params_ = params;
		for (const auto& opt : tracked_->optimizers())
		{
			if (opt.second)
			{
				opt.second->setParams(params_.optimizerParams);
			}
		}
		tracked_->setInitNumEvents(params_.initNumEvents);
		if (rebuild || ctx_ == nullptr)
		{
			if (ctx_)
			{
				ebo_destroy(ctx_);
				ctx_ = nullptr;
			}
			createContext();
			tracked_->rebind(ctx_, params_.imageSize);
		}
		reset();
	}
	ebo_ctx* handle() { return ctx_; }

	// DetectorParams::ERRORS_STATUS: the EBO_* code and message of the LAST call (EBO_OK / "" after
	// a call that succeeded).
	int status() const { return status_; }
	bool ok() const { return status_ == EBO_OK; }
	const std::string& lastError() const { return lastError_; }

   private:
	void createContext()
	{
		ebo_params p;
		ebo_default_params(&p);
		p.device = params_.device;
		p.image_w = params_.imageSize.width;
		p.image_h = params_.imageSize.height;
		p.patch_w = params_.patchCompensateSize.width;
		p.patch_h = params_.patchCompensateSize.height;
		p.tv_weight = params_.compensateTVweight;
		p.tv_huber = params_.compensateTVHuberLoss;
		p.scale = params_.compensateScale;
		p.min_events = params_.compensateMinNumEvents;
		p.loss = params_.loss;
		p.grad = params_.grad;
		p.max_events = params_.maxNumEventsToStore > 0 ? params_.maxNumEventsToStore : 1;
		p.max_windows = 1;
		const int rc = ebo_create(&p, &ctx_);
		if (rc != EBO_OK)
		{
			ctx_ = nullptr;
			numPatchesX_ = p.patch_w > 0 ? p.image_w / p.patch_w : 0;
			numPatchesY_ = p.patch_h > 0 ? p.image_h / p.patch_h : 0;
			fail(rc, ebo_last_error(nullptr));
		}
		else
		{
			ebo_grid(ctx_, &numPatchesX_, &numPatchesY_);
		}
		patchFlows_.assign(static_cast<size_t>(numPatchesX_) * numPatchesY_ * 2, 0.0);
	}

	// feature_detector.cpp:32-51 (mask_ belongs to the detectFeatures hook's owner)
	void reset()
	{
		motionField_.assign(static_cast<size_t>(params_.imageSize.height) * params_.imageSize.width * 2, 0.0f);
		compensatedEventImage_ = Mat64(params_.imageSize.height, params_.imageSize.width);
		integratedEventImage_ = Mat64(params_.imageSize.height, params_.imageSize.width);
		lastCompensation = common::timestamp_t(0);
		lastEvents_.clear();
	}

	// feature_detector.cpp:543-566
	void extractPatchesImpl(const common::ImageSample& image)
	{
		corners_ = hooks_.detectFeatures(image.value);
		Patches newPatches;
		for (const auto& corner : corners_)
		{
			newPatches.emplace_back(Patch(corner, params_.patchExtent, image.timestamp));
		}
		Mat64 gradX, gradY;
		hooks_.gradients(image.value, gradX, gradY);
		auto optimizer = std::make_shared<Optimizer>(params_.optimizerParams, params_.imageSize);
		optimizer->setGrad(gradX, gradY);
		tracked_->setOptimizer(image.timestamp, optimizer);
		tracked_->setGradients(gradX, gradY);  // gradX_ / gradY_: what updateNumOfEvents warps
		associatePatches(newPatches, image.timestamp);
	}

	// FlowEstimator::getFlowPatches (flow_estimator.cpp:27-85), the point flow through the hook
	void flowPatches()
	{
		for (Patch& patch : tracked_->getPatches())
		{
			if (patch.isInit())
			{
				continue;
			}
			const Corner corner = patch.toCorner();
			float nx = 0.f, ny = 0.f;
			if (!hooks_.flow(static_cast<float>(corner.x), static_cast<float>(corner.y), nx, ny))
			{
				patch.setLost();
				continue;
			}
			const double dirX = nx - corner.x;
			const double dirY = ny - corner.y;
			const double flowDir = std::atan2(dirY, dirX);
			common::Pose2d warp;
			warp.data()[2] = -dirX;
			warp.data()[3] = -dirY;
			patch.setWarp(warp);
			patch.setFlowDir(flowDir);
			patch.setTimeWithoutUpdate(common::timestamp_t(static_cast<int64_t>(
				hooks_.patchTimeWithoutUpdateScale / std::fmax(1e-1, std::sqrt(dirX * dirX + dirY * dirY)))));
			const Corner newCorner = patch.toCorner();
			if (newCorner.x <= 5 || newCorner.y <= 5 || newCorner.x >= params_.imageSize.width - 5 ||
				newCorner.y >= params_.imageSize.height - 5)
			{
				patch.setLost();
			}
		}
	}

	// patch.warpImage() for every patch (:505-511), one launch per frame the patches came from
	void warpByFrame(const std::vector<Patch*>& all)
	{
		std::map<int64_t, std::vector<Patch*>> byFrame;
		for (Patch* p : all)
		{
			byFrame[p->getInitTime().count()].push_back(p);
		}
		for (auto& group : byFrame)
		{
			auto opt = tracked_->optimizers().find(group.first);
			if (opt != tracked_->optimizers().end() && opt->second)
			{
				tracked_->warpImages(group.second, opt->second->handle());
			}
		}
	}

	// runs the tracked-patch calls (which throw) under the detector's error policy
	template <class F>
	void guarded(F&& f)
	{
		if (params_.errorPolicy == DetectorParams::ERRORS_THROW)
		{
			f();
			status_ = EBO_OK;
			lastError_.clear();
			return;
		}
		try
		{
			f();
			status_ = EBO_OK;
			lastError_.clear();
		}
		catch (const std::exception& e)
		{
			status_ = ctx_ ? EBO_ERR_HIP : EBO_ERR_NO_DEVICE;
			lastError_ = e.what();
		}
	}

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
	FrontEndHooks hooks_;
	std::unique_ptr<TrackedPatches> tracked_;
	std::vector<common::EventSample> pending_;  // handed to updatePatches(event), not yet routed (eventBatch)
	Corners corners_;
	size_t nextTrackId_ = 0;
	Patches archivedPatches_;
	size_t imageCounter_ = 0;
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
