// feature_tracker/optimizer.h — tracker::Optimizer / tracker::OptimizerParams with the
// reference's names (implementation/feature_tracker/include/feature_tracker/optimizer.h:14-73,
// src/optimizer.cpp:5-31,62-206), for the per-feature tracker path.
//
// optimize(Patch&) keeps the reference's signature and sequence:
//   integrateEvents -> Ceres problem {SE2 warp with LocalParameterizationSE2, flow direction;
//   OptimizerCostFunctor over the patch pixels; HuberLoss} -> Solve -> cost filter -> update of
//   flow / warp / rect / trajectory -> integrateMotionCompensatedEvents -> resetBatch.
// optimize(std::vector<Patch*>) is the same for many patches with ONE launch per stage — what
// the device is for: the reference spends 1-3 ms per patch on the CPU (report §5), the batched
// solve takes ~4 us per patch.  Every per-pixel stage forwards to the C ABI (include/ebo.h):
// ebo_patch_integrate, ebo_optimizer_solve (normalisation of the integrated nabla included),
// ebo_patch_integrate_mc.  Nothing but the per-patch bookkeeping runs on the CPU.
//
// OptimizerParams::drawCostMap (optimizer.cpp:33-60, off by default, "it is too slow" there): the cost
// maps of all surviving patches in one more launch (ebo_optimizer_cost_map), into Patch::setCostMap.
#pragma once

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../common/data_types.h"
#include "patch.h"
#include "types.h"

namespace tracker
{
struct OptimizerParams
{
	bool drawCostMap = false;
	int maxNumIterations = 10;
	int numThreads = 1;
	double optimizerThreshold = 0.6;
	double huberLoss = 0.3;
	int costMapWidth = 11;
	int costMapHeight = 11;
	// seconds in double to microseconds
	double patchTimeWithoutUpdateScale = 1e6;
};

struct OptimizerFinalLoss
{
	tracker::TrackId trackId;
	double lossValue;
	int64_t timeStampMicrosecond;
};

class Optimizer
{
   public:
	Optimizer(const OptimizerParams& params, const Size& imageSize) : params_(params), imageSize_(imageSize)
	{
		ebo_params p;
		ebo_default_params(&p);
		p.image_w = imageSize.width;
		p.image_h = imageSize.height;
		// the compensation grid is not used by the tracker path: keep it valid for images smaller than a patch
		p.patch_w = std::min(p.patch_w, std::max(imageSize.width, 1));
		p.patch_h = std::min(p.patch_h, std::max(imageSize.height, 1));
		if (ebo_create(&p, &ctx_) != EBO_OK)
		{
			throw std::runtime_error(std::string("tracker::Optimizer: ") + ebo_last_error(nullptr));
		}
	}
	~Optimizer()
	{
		if (ctx_)
		{
			ebo_destroy(ctx_);
		}
	}
	Optimizer(const Optimizer&) = delete;
	Optimizer& operator=(const Optimizer&) = delete;

	// optimizer.cpp:15-31
	void setGrad(const Mat64& gradX, const Mat64& gradY)
	{
		if (gradX.rows != imageSize_.height || gradX.cols != imageSize_.width || gradY.rows != gradX.rows ||
			gradY.cols != gradX.cols)
		{
			throw std::invalid_argument("gradient images must be imageSize");
		}
		check(ebo_optimizer_set_grad(ctx_, gradX.ptr(), gradY.ptr()));
	}

	// optimizer.cpp:62-206
	void optimize(Patch& patch)
	{
		std::vector<Patch*> one(1, &patch);
		optimize(one);
	}

	void optimize(const std::vector<Patch*>& patches)
	{
		const int n = static_cast<int>(patches.size());
		if (n == 0)
		{
			return;
		}
		// ---- patch.integrateEvents() for every patch (patch.cpp:65-85) -----------------------
		std::vector<ebo_event> ev;
		std::vector<size_t> evOff(1, 0), nablaOff;
		std::vector<double> rects(4 * static_cast<size_t>(n));
		size_t nablaTotal = 0;
		for (int i = 0; i < n; ++i)
		{
			const Patch& p = *patches[i];
			if (p.getEvents().empty())
			{
				throw std::invalid_argument("tracker::Optimizer: a patch without events (the reference reads events_.front())");
			}
			const std::vector<ebo_event> pe = common::toEboEvents(p.getEvents());
			ev.insert(ev.end(), pe.begin(), pe.end());
			evOff.push_back(ev.size());
			const Rect2d& r = p.getPatch();
			rects[4 * i + 0] = r.x;
			rects[4 * i + 1] = r.y;
			rects[4 * i + 2] = r.width;
			rects[4 * i + 3] = r.height;
			nablaOff.push_back(nablaTotal);
			nablaTotal += static_cast<size_t>(static_cast<int>(r.height)) * static_cast<int>(r.width);
		}
		std::vector<double> nabla(nablaTotal, 0.0);
		std::vector<int64_t> curTs(n), lastUpd(n);
		check(ebo_patch_integrate(ctx_, ev.data(), evOff.data(), n, rects.data(), nablaOff.data(), nabla.data(),
								  curTs.data(), lastUpd.data()));
		// ---- the Ceres problem of optimizer.cpp:72-119, all patches in one launch -------------
		std::vector<double> poses(4 * static_cast<size_t>(n)), flows(n);
		for (int i = 0; i < n; ++i)
		{
			Patch& p = *patches[i];
			const Rect2d& r = p.getPatch();
			Mat64 m(static_cast<int>(r.height), static_cast<int>(r.width));
			std::copy(nabla.begin() + nablaOff[i], nabla.begin() + nablaOff[i] + static_cast<size_t>(m.rows) * m.cols,
					  m.ptr());
			p.setIntegratedNabla(m);
			p.setTimestamps(common::timestamp_t(curTs[i]), common::timestamp_t(lastUpd[i]));
			std::copy(p.getWarp().data(), p.getWarp().data() + 4, poses.begin() + 4 * i);
			flows[i] = p.getFlow();  // through float, optimizer.cpp:81 / patch.h:56
		}
		ebo_solver_opts o;
		ebo_optimizer_default_solver(&o);
		o.max_num_iterations = params_.maxNumIterations;
		std::vector<ebo_summary> sums(n);
		check(ebo_optimizer_solve(ctx_, n, rects.data(), nabla.data(), 1, params_.huberLoss, &o, poses.data(),
								  flows.data(), sums.data()));
		lastSummaries_ = sums;
		// ---- per-patch bookkeeping (optimizer.cpp:131-178) -----------------------------------
		std::vector<int> alive;
		for (int i = 0; i < n; ++i)
		{
			Patch& p = *patches[i];
			p.addFinalCost(sums[i].final_cost);
			vectorFinalCost_.push_back({p.getTrackId(), sums[i].final_cost, p.getCurrentTimestamp().count()});
			const auto& costs = p.getFinalCosts();
			// :141-160 reads costs[size - 6 + i], i = 0..4, once size >= 5: with exactly five
			// costs that is index -1 (undefined); the filter is applied from six costs on
			if (costs.size() >= 6)
			{
				std::vector<double> lastFive;
				for (int k = 0; k < 5; k++)
				{
					lastFive.push_back(costs[costs.size() - 6 + k]);
				}
				std::sort(lastFive.begin(), lastFive.end());
				if (lastFive[2] > params_.optimizerThreshold)
				{
					p.setLost();
					continue;
				}
			}
			double flowDir = std::fmod(flows[i], 2 * M_PI);
			p.setFlowDir(flowDir);
			common::Pose2d warp;
			std::copy(poses.begin() + 4 * i, poses.begin() + 4 * i + 4, warp.data());
			p.setWarp(warp);
			const Corner oldCenter = p.toCorner();
			p.updatePatchRect();
			const Corner newCenter = p.toCorner();
			const double moved = std::sqrt((newCenter.x - oldCenter.x) * (newCenter.x - oldCenter.x) +
										   (newCenter.y - oldCenter.y) * (newCenter.y - oldCenter.y));
			p.setTimeWithoutUpdate(common::timestamp_t(
				static_cast<int64_t>(params_.patchTimeWithoutUpdateScale / std::fmax(1e-1, moved))));
			p.addTrajectoryPosition();
			alive.push_back(i);
		}
		// ---- patch.integrateMotionCompensatedEvents() (patch.cpp:87-130), one launch ----------
		std::vector<int> mc;
		for (int i : alive)
		{
			if (patches[i]->getTrajectory().size() >= 2)
			{
				mc.push_back(i);
			}
		}
		if (!mc.empty())
		{
			const int k = static_cast<int>(mc.size());
			std::vector<ebo_event> ev2;
			std::vector<size_t> off2(1, 0), noff2;
			std::vector<double> rects2(4 * static_cast<size_t>(k)), traj(6 * static_cast<size_t>(k));
			std::vector<int64_t> mid(k);
			size_t total2 = 0;
			for (int j = 0; j < k; ++j)
			{
				const Patch& p = *patches[mc[j]];
				const std::vector<ebo_event> pe = common::toEboEvents(p.getEvents());
				ev2.insert(ev2.end(), pe.begin(), pe.end());
				off2.push_back(ev2.size());
				const Rect2d& r = p.getPatch();
				rects2[4 * j + 0] = r.x;
				rects2[4 * j + 1] = r.y;
				rects2[4 * j + 2] = r.width;
				rects2[4 * j + 3] = r.height;
				const auto& tr = p.getTrajectory();
				const auto& pre = tr[tr.size() - 2];
				const auto& last = tr[tr.size() - 1];
				traj[6 * j + 0] = pre.value.x;
				traj[6 * j + 1] = pre.value.y;
				traj[6 * j + 2] = static_cast<double>(pre.timestamp.count());
				traj[6 * j + 3] = last.value.x;
				traj[6 * j + 4] = last.value.y;
				traj[6 * j + 5] = static_cast<double>(last.timestamp.count());
				mid[j] = p.getCurrentTimestamp().count();
				noff2.push_back(total2);
				total2 += static_cast<size_t>(static_cast<int>(r.height)) * static_cast<int>(r.width);
			}
			std::vector<double> nabla2(total2, 0.0);
			std::vector<int32_t> updated(k, 0);
			check(ebo_patch_integrate_mc(ctx_, ev2.data(), off2.data(), k, rects2.data(), traj.data(), mid.data(),
										 noff2.data(), nabla2.data(), updated.data()));
			for (int j = 0; j < k; ++j)
			{
				if (updated[j])
				{
					Patch& p = *patches[mc[j]];
					const Rect2d& r = p.getPatch();
					Mat64 m(static_cast<int>(r.height), static_cast<int>(r.width));
					std::copy(nabla2.begin() + noff2[j], nabla2.begin() + noff2[j] + static_cast<size_t>(m.rows) * m.cols,
							  m.ptr());
					p.setMotionCompensatedIntegratedNabla(m);
				}
			}
		}
		for (int i : alive)
		{
			patches[i]->resetBatch();
		}
		// ---- drawCostMap (optimizer.cpp:33-60,191-204), all surviving patches in one launch: the functor of the
		// optimisation -- the rect and the integrated nabla from BEFORE the solve -- at the solved pose and flow
		if (params_.drawCostMap && !alive.empty())
		{
			const int k = static_cast<int>(alive.size()), mw = params_.costMapWidth, mh = params_.costMapHeight;
			std::vector<double> rects3(4 * static_cast<size_t>(k)), poses3(4 * static_cast<size_t>(k)), flows3(k), nabla3;
			for (int j = 0; j < k; ++j)
			{
				const int i = alive[j];
				std::copy(rects.begin() + 4 * i, rects.begin() + 4 * i + 4, rects3.begin() + 4 * j);
				std::copy(patches[i]->getWarp().data(), patches[i]->getWarp().data() + 4, poses3.begin() + 4 * j);
				flows3[j] = patches[i]->getFlow();  // double flowDir = patch.getFlow(): through float (:37)
				const size_t m = static_cast<size_t>(static_cast<int>(rects[4 * i + 3])) * static_cast<int>(rects[4 * i + 2]);
				nabla3.insert(nabla3.end(), nabla.begin() + nablaOff[i], nabla.begin() + nablaOff[i] + m);
			}
			std::vector<double> maps(static_cast<size_t>(k) * mw * mh, 0.0);
			check(ebo_optimizer_cost_map(ctx_, k, rects3.data(), nabla3.data(), 1, poses3.data(), flows3.data(), mw, mh, maps.data()));
			for (int j = 0; j < k; ++j)
			{
				Mat64 m(mh, mw);
				std::copy(maps.begin() + static_cast<size_t>(j) * mw * mh, maps.begin() + static_cast<size_t>(j + 1) * mw * mh, m.ptr());
				patches[alive[j]]->setCostMap(m);
			}
		}
	}

	OptimizerParams* getParams() { return &params_; }
	std::vector<OptimizerFinalLoss> getFinalCosts() { return vectorFinalCost_; }
	void setParams(const OptimizerParams& params) { params_ = params; }
	void addUser() { used_++; }
	void deleteUser() { used_--; }
	bool isUsed() const { return used_ != 0; }
	// ceres::Solver::Summary of the last optimize call, one per patch (the reference logs
	// summary.BriefReport(), optimizer.cpp:119)
	const std::vector<ebo_summary>& getLastSummaries() const { return lastSummaries_; }
	ebo_ctx* handle() { return ctx_; }

   private:
	void check(int rc)
	{
		if (rc != EBO_OK)
		{
			throw std::runtime_error(std::string("tracker::Optimizer: ") + ebo_last_error(ctx_));
		}
	}

	OptimizerParams params_;
	Size imageSize_;
	size_t used_ = 0;
	ebo_ctx* ctx_ = nullptr;
	std::vector<OptimizerFinalLoss> vectorFinalCost_;
	std::vector<ebo_summary> lastSummaries_;
};

}  // namespace tracker
