// dataset_reader/davis240c_reader.h — tools::Davis240cReader for the EVENT side of a DAVIS240C recording, with the
// reference's names (tools/dataset_reader/include/dataset_reader/davis240c_reader.h:7-29, dataset_reader.h:17-31,
// src/davis240c_reader.cpp:60-92,153-212,279-299): SURVEY §8(f) #3.
//
//   tools::Davis240cReader reader(path);            // a directory holding events.txt
//   while (auto events = reader.getEvents()) ...    // EVENT_LENGTH = 1 000 000 lines per call, the next call continues
//   reader.getEventSample(line)                     // one "<seconds> <x> <y> <0|1>" line
//   reader.getTrajectory() / getTrajectoryLine      // trajectory.txt: "<id> <seconds> <x> <y>", one Patch per line
//
// Parsing runs in the library (ebo_read_events_txt_at: strtod / strtol on a streamed buffer instead of a substr per
// field and a thread pool over std::strings; the same values: seconds through a double, truncated to microseconds),
// or, when `events.bin` — the packed sidecar ebo_write_events_bin makes — lies next to events.txt, from that at memory
// speed.  The frame, ground-truth and calibration files are not on the event path (images need an image decoder): their
// getters throw std::runtime_error naming that.
#pragma once

#include <algorithm>
#include <fstream>
#include <optional>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "../common/data_types.h"
#include "../feature_tracker/patch.h"

namespace tools
{
class Davis240cReader
{
   public:
	static constexpr size_t EVENT_LENGTH = 1000000;  // davis240c_reader.cpp:14

	explicit Davis240cReader(const std::string& path) : path_(path) {}

	// davis240c_reader.cpp:186-212: the next (at most) EVENT_LENGTH events; empty optional at the end of the file
	std::optional<common::EventSequence> getEvents()
	{
		std::vector<ebo_event> buf(EVENT_LENGTH);
		size_t n = 0;
		if (useSidecar())
		{
			if (!sidecarLoaded_)
			{
				loadSidecar();
			}
			n = std::min(EVENT_LENGTH, sidecar_.size() - eventStart_);
			std::copy(sidecar_.begin() + static_cast<std::ptrdiff_t>(eventStart_),
					  sidecar_.begin() + static_cast<std::ptrdiff_t>(eventStart_ + n), buf.begin());
		}
		else
		{
			const int rc = ebo_read_events_txt_at((path_ + "/events.txt").c_str(), &offset_, buf.data(), buf.size(), &n);
			if (rc == EBO_ERR_RANGE)
			{
				throw std::runtime_error("Sign is not equal to 0/1");  // (or a malformed line; :85-88 throws this)
			}
			if (rc != EBO_OK)
			{
				throw std::runtime_error("tools::Davis240cReader: cannot read " + path_ + "/events.txt");
			}
		}
		if (n == 0)
		{
			return {};
		}
		eventStart_ += n;
		common::EventSequence events;
		for (size_t i = 0; i < n; ++i)
		{
			events.push_back(toSample(buf[i]));
		}
		return std::make_optional(std::move(events));
	}

	// davis240c_reader.cpp:60-92 (the line is consumed, as there)
	common::EventSample getEventSample(std::string& line) const
	{
		size_t pos = line.find(' ');
		const double seconds = std::stod(line.substr(0, pos));
		const common::timestamp_t timestamp(static_cast<int64_t>(seconds * 1000000.0));  // duration_cast of duration<double>: truncation
		line = line.substr(pos + 1);
		common::Point2i point;
		pos = line.find(' ');
		point.x = std::stoi(line.substr(0, pos));
		line = line.substr(pos + 1);
		pos = line.find(' ');
		point.y = std::stoi(line.substr(0, pos));
		line = line.substr(pos + 1);
		const int32_t sign = std::stoi(line.substr(0, pos));
		common::EventPolarity polarity = common::POSITIVE;
		if (sign == 0)
		{
			polarity = common::NEGATIVE;
		}
		else if (sign != 1)
		{
			throw std::runtime_error("Sign is not equal to 0/1");
		}
		common::EventSample s;
		s.value.point = point;
		s.value.sign = polarity;
		s.timestamp = timestamp;
		return s;
	}

	// davis240c_reader.cpp:153-176: one "<id> <seconds> <x> <y>" line of a trajectory.txt (what
	// tools::Evaluator::saveFeaturesTrajectory writes) -> a Patch of extent 1 at that point with that track id
	tracker::Patch getTrajectoryLine(std::string& line) const
	{
		size_t pos = line.find(' ');
		const int32_t id = std::stoi(line.substr(0, pos));
		line = line.substr(pos + 1);
		pos = line.find(' ');
		const common::timestamp_t timestamp(static_cast<int64_t>(std::stod(line.substr(0, pos)) * 1000000.0));
		line = line.substr(pos + 1);
		pos = line.find(' ');
		const double x = std::stod(line.substr(0, pos));
		line = line.substr(pos + 1);
		const double y = std::stod(line);
		tracker::Patch patch({x, y}, 1, timestamp);
		patch.setTrackId(id);
		return patch;
	}

	// davis240c_reader.cpp:279-299: trajectory.txt line by line (at most EVENT_LENGTH lines, as there), one Patch per line
	tracker::Patches getTrajectory() const
	{
		std::ifstream in(path_ + "/trajectory.txt");
		if (!in)
		{
			throw std::runtime_error("tools::Davis240cReader: cannot read " + path_ + "/trajectory.txt");
		}
		tracker::Patches out;
		std::string line;
		size_t lines = 0;
		while (lines < EVENT_LENGTH && std::getline(in, line))
		{
			++lines;
			out.push_back(getTrajectoryLine(line));
		}
		return out;
	}

	[[noreturn]] void getImages() const { notOnThePath("getImages (images.txt + the frames: an image decoder)"); }
	[[noreturn]] void getGroundTruth() const { notOnThePath("getGroundTruth (groundtruth.txt)"); }
	[[noreturn]] void getCalibration() const { notOnThePath("getCalibration (calib.txt)"); }

	size_t eventsRead() const { return eventStart_; }

   private:
	static common::EventSample toSample(const ebo_event& e)
	{
		common::EventSample s;
		s.value.point = common::Point2i(e.x, e.y);
		s.value.sign = e.sign < 0 ? common::NEGATIVE : common::POSITIVE;
		s.timestamp = common::timestamp_t(e.t_us);
		return s;
	}
	bool useSidecar() const
	{
		struct stat a, b;
		const bool haveBin = ::stat((path_ + "/events.bin").c_str(), &b) == 0;
		const bool haveTxt = ::stat((path_ + "/events.txt").c_str(), &a) == 0;
		return haveBin && (!haveTxt || b.st_mtime >= a.st_mtime);  // a sidecar older than its text file is stale
	}
	void loadSidecar()
	{
		struct stat st;
		size_t n = 0;
		if (::stat((path_ + "/events.bin").c_str(), &st) == 0 && st.st_size >= 32)
		{
			n = (static_cast<size_t>(st.st_size) - 32) / 16;  // 32-byte header + 16 bytes per event (include/ebo.h)
		}
		sidecar_.resize(n);
		if (ebo_read_events_bin((path_ + "/events.bin").c_str(), sidecar_.data(), sidecar_.size(), &n) != EBO_OK || n != sidecar_.size())
		{
			throw std::runtime_error("tools::Davis240cReader: cannot read " + path_ + "/events.bin");
		}
		sidecarLoaded_ = true;
	}
	[[noreturn]] static void notOnThePath(const char* what)
	{
		throw std::runtime_error(std::string("tools::Davis240cReader::") + what + " is not on the event path and not built");
	}

	std::string path_;
	size_t eventStart_ = 0;
	uint64_t offset_ = 0;
	bool sidecarLoaded_ = false;
	std::vector<ebo_event> sidecar_;
};

}  // namespace tools
