// feature_tracker/tracked_patches.h — the tracked-patch half of tracker::FeatureDetector with
// the reference's names: `patches_`, `optimizers_`, `updatePatches`, `updateNumOfEvents`
// (implementation/feature_tracker/src/feature_detector.cpp:585-619,666-711; members
// feature_detector.h:100-118).  tracker::FeatureDetector owns one (sharing its device context) and
// forwards updatePatches / getPatches / setPatches / updateNumOfEvents to it; the class is usable
// on its own too (it then owns a context).
//
//   updatePatches(const common::EventSample&)        the reference's per-event call, verbatim order:
//        for every patch that is not lost: addEvent if the event is inside its rect; then, if the
//        patch isReady() && isInit(): optimizers_[initTime]->optimize(patch); updateNumOfEvents.
//   updatePatches(const std::vector<EventSample>&)   the same for a whole chunk of the stream, the
//        way the device wants it.  Patches do not interact (a patch's events, readiness and
//        optimisation depend on its own state only), so all patches advance in lock-step ROUNDS:
//        one ebo_route_events launch finds, for every patch, the events inside its current rect up
//        to the one that makes it ready (events -> patches, all at once, instead of one host test
//        per (event, patch)); the patches that became ready are optimised together (one launch
//        per stage, Optimizer::optimize(std::vector<Patch*>)); their rects have moved, so the next
//        round routes the rest of the chunk from where each patch stopped.  Per patch the sequence
//        of addEvent / optimize calls is exactly the per-event one.
//
// updateNumOfEvents (:666-711) is complete: the two border branches on the host, the event-count
// estimate (:689-707: cv::warpAffine of gradX_ / gradY_ with flags = WARP_INVERSE_MAP, i.e. nearest
// neighbour through OpenCV's 10-bit fixed-point map, and the L1 norm over the rect) on the device
// (ebo_estimate_num_events) from the gradient images given to setGradients -- gradX_ / gradY_ of the
// latest frame in the reference (:516-517).  setNumOfEventsEstimator overrides it; without gradients
// and without an estimator a patch keeps its event count.  Patch::warpImage (:615, after every
// optimisation: the predicted gradient patch, a visualisation) runs on the device too
// (ebo_patch_warp_image, one launch for the patches of a round that share an optimiser) against the
// gradient images of the patch's OWN frame -- the reference's Patch keeps the cv::Mat headers it was
// given when it was extracted (setGrad, :508 region), which are the images its frame's Optimizer holds;
// setWarpImages(false) skips it.
#pragma once

#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "optimizer.h"
#include "patch.h"

namespace tracker
{
class TrackedPatches
{
   public:
	// imageSize = gradX_.size() of the reference; initNumEvents = DetectorParams::initNumEvents
	TrackedPatches(const Size& imageSize, int initNumEvents = 75, int device = 0)
		: imageSize_(imageSize), initNumEvents_(initNumEvents)
	{
		ebo_params p;
		ebo_default_params(&p);
		p.device = device;
		p.image_w = imageSize.width;
		p.image_h = imageSize.height;
		// the compensation grid is not used by the tracker path: keep it valid for images smaller than a patch
		p.patch_w = std::min(p.patch_w, std::max(imageSize.width, 1));
		p.patch_h = std::min(p.patch_h, std::max(imageSize.height, 1));
		if (ebo_create(&p, &ctx_) != EBO_OK)
		{
			throw std::runtime_error(std::string("tracker::TrackedPatches: ") + ebo_last_error(nullptr));
		}
	}
	// the same on a context somebody else owns (tracker::FeatureDetector's): routing and the event-count
	// estimate keep their state apart from the compensation window, so one context serves both
	TrackedPatches(ebo_ctx* shared, const Size& imageSize, int initNumEvents)
		: imageSize_(imageSize), initNumEvents_(initNumEvents), ctx_(shared), ownsCtx_(false)
	{
	}
	~TrackedPatches()
	{
		if (ownsCtx_ && ctx_)
		{
			ebo_destroy(ctx_);
		}
	}
	TrackedPatches(const TrackedPatches&) = delete;
	TrackedPatches& operator=(const TrackedPatches&) = delete;

	Patches& getPatches() { return patches_; }
	Patches const& getPatches() const { return patches_; }
	void setPatches(const Patches& patches) { patches_ = patches; }  // feature_detector.h:72
	void addPatch(const Patch& patch) { patches_.push_back(patch); }
	void setInitNumEvents(int initNumEvents) { initNumEvents_ = initNumEvents; }
	// a FeatureDetector that re-creates its context (setParams with another image size) hands the new one over
	void rebind(ebo_ctx* shared, const Size& imageSize)
	{
		if (!ownsCtx_)
		{
			ctx_ = shared;
			imageSize_ = imageSize;
			haveGradients_ = false;
		}
	}
	std::map<int64_t, std::shared_ptr<Optimizer>>& optimizers() { return optimizers_; }
	const std::map<int64_t, std::shared_ptr<Optimizer>>& optimizers() const { return optimizers_; }

	// optimizers_[image.timestamp.count()] of the reference (:560-563): the optimizer holding the
	// gradient grid of the image a patch was extracted from, keyed by the patch's init time
	void setOptimizer(const common::timestamp_t& initTime, std::shared_ptr<Optimizer> optimizer)
	{
		optimizers_[initTime.count()] = std::move(optimizer);
	}
	void setNumOfEventsEstimator(std::function<size_t(const Patch&)> estimator) { estimator_ = std::move(estimator); }
	// gradX_ / gradY_ of FeatureDetector (set per frame in newImage, :516-517): what updateNumOfEvents warps
	void setGradients(const Mat64& gradX, const Mat64& gradY)
	{
		if (gradX.rows != imageSize_.height || gradX.cols != imageSize_.width || gradY.rows != gradX.rows ||
			gradY.cols != gradX.cols)
		{
			throw std::invalid_argument("gradient images must be imageSize");
		}
		check(ebo_optimizer_set_grad(ctx_, gradX.ptr(), gradY.ptr()));
		haveGradients_ = true;
	}

	// feature_detector.cpp:666-711, one patch
	void updateNumOfEvents(Patch& patch)
	{
		std::vector<Patch*> one(1, &patch);
		updateNumOfEvents(one);
	}

	// the same for several patches: the border branches per patch, ONE launch for the estimates
	void updateNumOfEvents(const std::vector<Patch*>& patches)
	{
		std::vector<Patch*> need;
		for (Patch* pp : patches)
		{
			Patch& patch = *pp;
			const Rect2d& rect = patch.getPatch();
			// (rect.tl() + rect.br()) * 0.5
			const double cx = (rect.x + (rect.x + rect.width)) * 0.5, cy = (rect.y + (rect.y + rect.height)) * 0.5;
			if (cx <= 5 || cy <= 5 || cx >= imageSize_.width - 5 || cy >= imageSize_.height - 5)
			{
				patch.setLost();
				continue;
			}
			if (rect.x < 0 || rect.y < 0 || rect.x + rect.width >= imageSize_.width ||
				rect.y + rect.height >= imageSize_.height)
			{
				patch.setNumOfEvents(static_cast<size_t>(initNumEvents_));
				continue;
			}
			if (estimator_)
			{
				patch.setNumOfEvents(estimator_(patch));
			}
			else if (haveGradients_)
			{
				need.push_back(pp);
			}
		}
		if (need.empty())
		{
			return;
		}
		const size_t n = need.size();
		std::vector<double> rects(4 * n), poses(4 * n), flows(n);
		std::vector<uint64_t> est(n);
		for (size_t i = 0; i < n; ++i)
		{
			const Rect2d& r = need[i]->getPatch();
			rects[4 * i + 0] = r.x;
			rects[4 * i + 1] = r.y;
			rects[4 * i + 2] = r.width;
			rects[4 * i + 3] = r.height;
			const double* w = need[i]->getWarp().data();
			std::copy(w, w + 4, &poses[4 * i]);
			flows[i] = need[i]->getFlow();  // a float in the reference (patch.h:56)
		}
		check(ebo_estimate_num_events(ctx_, static_cast<int>(n), rects.data(), poses.data(), flows.data(), est.data()));
		for (size_t i = 0; i < n; ++i)
		{
			need[i]->setNumOfEvents(static_cast<size_t>(est[i]));
		}
	}

	// Patch::warpImage (patch.cpp:132-154) for patches that share an optimiser, i.e. a frame's gradient
	// images: one launch.  A rect that touches the image border keeps its old predictedNabla_ (:145-150).
	void setWarpImages(bool on) { warpImages_ = on; }
	void warpImages(const std::vector<Patch*>& patches, ebo_ctx* gradCtx)
	{
		if (!warpImages_ || patches.empty())
		{
			return;
		}
		const size_t n = patches.size();
		std::vector<double> rects(4 * n), poses(4 * n), flows(n);
		std::vector<size_t> off(n, 0);
		std::vector<int32_t> updated(n, 0);
		size_t total = 0;
		for (size_t i = 0; i < n; ++i)
		{
			const Rect2d& r = patches[i]->getPatch();
			rects[4 * i + 0] = r.x;
			rects[4 * i + 1] = r.y;
			rects[4 * i + 2] = r.width;
			rects[4 * i + 3] = r.height;
			const double* w = patches[i]->getWarp().data();
			std::copy(w, w + 4, &poses[4 * i]);
			flows[i] = patches[i]->getFlowDir();  // the double member, as patch.cpp:152
			off[i] = total;
			total += static_cast<size_t>(std::nearbyint(r.height)) * static_cast<size_t>(std::nearbyint(r.width));
		}
		std::vector<double> out(total > 0 ? total : 1, 0.0);
		check(ebo_patch_warp_image(gradCtx, static_cast<int>(n), rects.data(), poses.data(), flows.data(), off.data(), out.data(),
								   updated.data()));
		for (size_t i = 0; i < n; ++i)
		{
			if (updated[i])
			{
				const Rect2d& r = patches[i]->getPatch();
				Mat64 m(static_cast<int>(std::nearbyint(r.height)), static_cast<int>(std::nearbyint(r.width)));
				std::copy(out.begin() + static_cast<std::ptrdiff_t>(off[i]),
						  out.begin() + static_cast<std::ptrdiff_t>(off[i] + static_cast<size_t>(m.rows) * m.cols), m.ptr());
				patches[i]->setPredictedNabla(m);
			}
		}
	}

	// feature_detector.cpp:585-619, one event
	void updatePatches(const common::EventSample& event)
	{
		for (auto& patch : patches_)
		{
			if (!patch.isLost())
			{
				if (patch.isInPatch(event.value.point))
				{
					patch.addEvent(event);
				}
				if (patch.isReady() && patch.isInit())
				{
					Optimizer& opt = optimizerOf(patch);
					opt.optimize(patch);
					updateNumOfEvents(patch);
					warpImages(std::vector<Patch*>(1, &patch), opt.handle());  // :615
				}
			}
		}
	}

	// the same for a chunk of the stream, all patches in lock-step rounds (see the header comment)
	void updatePatches(const std::vector<common::EventSample>& chunk)
	{
		const size_t n = chunk.size();
		if (n == 0 || patches_.empty())
		{
			return;
		}
		const size_t np = patches_.size();
		std::vector<Patch*> at;  // patches_ is a list (the reference's type): index it once
		at.reserve(np);
		for (Patch& p : patches_)
		{
			at.push_back(&p);
		}
		std::vector<uint32_t> cursor(np, 0);
		std::vector<int> live;  // patches still consuming the chunk
		for (size_t i = 0; i < np; ++i)
		{
			Patch& p = *at[i];
			if (p.isLost())
			{
				continue;
			}
			if (!p.isInit())
			{
				// never optimised (:608): it only collects events; no rect change, no round needed
				for (const auto& e : chunk)
				{
					if (p.isInPatch(e.value.point))
					{
						p.addEvent(e);
					}
				}
				continue;
			}
			live.push_back(static_cast<int>(i));
		}
		rounds_ = 0;
		if (live.empty())
		{
			return;  // nothing is being tracked yet: the patches only collected events, the device was not needed
		}
		const std::vector<ebo_event> ev = common::toEboEvents(chunk);
		check(ebo_route_set_events(ctx_, ev.data(), n));
		std::vector<double> rects;
		std::vector<uint32_t> start, take, index, count, next;
		while (!live.empty())
		{
			++rounds_;
			const int k = static_cast<int>(live.size());
			rects.resize(4 * static_cast<size_t>(k));
			start.resize(k);
			take.resize(k);
			count.assign(k, 0);
			next.assign(k, 0);
			uint32_t cap = 1;
			for (int j = 0; j < k; ++j)
			{
				const Patch& p = *at[live[j]];
				const Rect2d& r = p.getPatch();
				rects[4 * j + 0] = r.x;
				rects[4 * j + 1] = r.y;
				rects[4 * j + 2] = r.width;
				rects[4 * j + 3] = r.height;
				start[j] = cursor[live[j]];
				// events still missing for isReady(): counter_ >= 30 && events_.size() >= numOfEvents_.
				// Already ready (its state was changed from outside): the reference optimises it at the
				// very next event of the stream, after adding that event if it is inside.
				take[j] = static_cast<uint32_t>(p.eventsUntilReady());
				cap = std::max(cap, take[j]);
			}
			index.assign(static_cast<size_t>(k) * cap, 0);
			check(ebo_route_events(ctx_, k, rects.data(), start.data(), take.data(), cap, index.data(), count.data(),
								   next.data()));
			std::vector<int> still;
			std::map<int64_t, std::vector<Patch*>> ready;
			for (int j = 0; j < k; ++j)
			{
				const int i = live[j];
				Patch& p = *at[i];
				if (take[j] == 0)
				{
					const common::EventSample& e = chunk[cursor[i]];
					if (p.isInPatch(e.value.point))
					{
						p.addEvent(e);
					}
					cursor[i] += 1;
					ready[p.getInitTime().count()].push_back(&p);
					still.push_back(i);
					continue;
				}
				for (uint32_t q = 0; q < count[j]; ++q)
				{
					p.addEvent(chunk[index[static_cast<size_t>(j) * cap + q]]);
				}
				cursor[i] = next[j];
				if (count[j] == take[j])
				{
					ready[p.getInitTime().count()].push_back(&p);
					still.push_back(i);
				}
				// else: the chunk ran out before the patch became ready
			}
			for (auto& group : ready)
			{
				auto it = optimizers_.find(group.first);
				if (it == optimizers_.end() || !it->second)
				{
					throw std::runtime_error("tracker::TrackedPatches: no optimizer for a patch's init time");
				}
				it->second->optimize(group.second);
				updateNumOfEvents(group.second);  // also after a patch was lost in optimize, as :613-614
				warpImages(group.second, it->second->handle());  // :615
			}
			live.clear();
			for (int i : still)
			{
				if (!at[i]->isLost() && cursor[i] < n)
				{
					live.push_back(i);
				}
			}
		}
	}

	int lastRounds() const { return rounds_; }
	ebo_ctx* handle() { return ctx_; }

   private:
	Optimizer& optimizerOf(const Patch& patch)
	{
		auto it = optimizers_.find(patch.getInitTime().count());
		if (it == optimizers_.end() || !it->second)
		{
			throw std::runtime_error("tracker::TrackedPatches: no optimizer for a patch's init time");
		}
		return *it->second;
	}
	void check(int rc)
	{
		if (rc != EBO_OK)
		{
			throw std::runtime_error(std::string("tracker::TrackedPatches: ") + ebo_last_error(ctx_));
		}
	}

	Size imageSize_;
	int initNumEvents_;
	ebo_ctx* ctx_ = nullptr;
	bool ownsCtx_ = true;
	Patches patches_;
	std::map<int64_t, std::shared_ptr<Optimizer>> optimizers_;
	std::function<size_t(const Patch&)> estimator_;
	bool haveGradients_ = false;
	bool warpImages_ = true;
	int rounds_ = 0;
};

}  // namespace tracker
