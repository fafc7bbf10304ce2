// tools/evaluator.h — the track side of tools::Evaluator that the multi-GPU layout needs
// (BASELINE config 5: independent sequences, one per GPU, gather of per-sequence tracks):
//   * tools::saveFeaturesTrajectory   = tools::Evaluator::saveFeaturesTrajectory
//     (tools/evaluator/src/evaluator.cpp:125-150): trajectory.txt, "feature_id timestamp x y";
//   * tools::trackPoints              flattens tracker::Patches into the 32-byte records the
//     exchange carries (same order as the file: patch by patch, trajectory order);
//   * tools::gatherFeaturesTrajectories  every rank's records on every rank, rank 0's first
//     (ebo_allgather_tracks: one RCCL all-gather of max-padded records + the counts).
// Nothing here computes; it is plumbing over include/ebo.h.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../feature_tracker/tracked_patches.h"

namespace tools
{
inline std::vector<ebo_track_point> trackPoints(const tracker::Patches& patches)
{
	std::vector<ebo_track_point> out;
	for (const auto& patch : patches)
	{
		for (const auto& pos : patch.getTrajectory())
		{
			ebo_track_point t;
			t.id = patch.getTrackId();
			t.t_us = pos.timestamp.count();
			t.x = pos.value.x;
			t.y = pos.value.y;
			out.push_back(t);
		}
	}
	return out;
}

// tools::Evaluator::saveFeaturesTrajectory with the output path spelled out
// (the reference writes params_.outputDir + "/trajectory.txt")
inline void saveFeaturesTrajectory(const tracker::Patches& patches, const std::string& outputFilename)
{
	const std::vector<ebo_track_point> pts = trackPoints(patches);
	if (ebo_write_tracks_txt(outputFilename.c_str(), pts.data(), pts.size()) != EBO_OK)
	{
		throw std::runtime_error("tools::saveFeaturesTrajectory: cannot write " + outputFilename);
	}
}

// all ranks' tracks on every rank; `counts` (optional) receives the per-rank record counts.
// ctx must carry a communicator (ebo_comm_init); every rank of it must make this call.
inline std::vector<ebo_track_point> gatherFeaturesTrajectories(ebo_ctx* ctx, const tracker::Patches& patches,
															   std::vector<size_t>* counts = nullptr)
{
	const std::vector<ebo_track_point> mine = trackPoints(patches);
	int nranks = 1;
	if (ebo_comm_size(ctx, nullptr, &nranks) != EBO_OK || nranks < 1)
	{
		throw std::runtime_error("tools::gatherFeaturesTrajectories: no context");
	}
	std::vector<size_t> cnt(static_cast<size_t>(nranks), 0);  // sized from the communicator, never from the caller
	size_t total = 0;
	// the counts collective sizes the result, the gather fills it (both on every rank)
	if (ebo_allgather_track_counts(ctx, mine.size(), &total, cnt.data()) != EBO_OK)
	{
		throw std::runtime_error(std::string("tools::gatherFeaturesTrajectories: ") + ebo_last_error(ctx));
	}
	std::vector<ebo_track_point> all(total);
	if (ebo_allgather_tracks(ctx, mine.data(), mine.size(), all.data(), all.size(), &total, cnt.data()) != EBO_OK)
	{
		throw std::runtime_error(std::string("tools::gatherFeaturesTrajectories: ") + ebo_last_error(ctx));
	}
	if (counts)
	{
		*counts = cnt;
	}
	return all;
}

}  // namespace tools
