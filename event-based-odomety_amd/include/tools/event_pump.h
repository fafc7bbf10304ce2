// tools/event_pump.h — the event side of tools::Evaluator and tools::Replayer, reduced to
// what drives the motion-compensation path:
//   * tools::EvaluatorParams::{compensationFrequencyTime 300000 us, compensationFrequencyEvents
//     15000} (tools/evaluator/include/evaluator/evaluator.h:21-22);
//   * tools::Evaluator::eventCallback (tools/evaluator/src/evaluator.cpp:32-45):
//       addEvent -> [updatePatches: frame-based tracker, out of scope] ->
//       if (ts - lastCompensation >= time || events >= count)
//           compensateEventsContrast(getEvents()); integrateEvents(getEvents()); clearEvents();
//   * reading events.txt (Davis240cReader) and pumping them in timestamp order
//     (Replayer::next without the image stream).
// This is the plumbing of BASELINE config 1; it computes nothing itself.
#pragma once

#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../feature_tracker/feature_detector.h"

namespace tools
{
struct EvaluatorParams
{
	tracker::Size imageSize = {240, 180};
	// compensate whole image each k microseconds
	uint32_t compensationFrequencyTime = 300000;
	uint32_t compensationFrequencyEvents = 15000;
};

class EventPump
{
   public:
	// called after every compensation with the detector holding that window's results
	using WindowCallback = std::function<void(tracker::FeatureDetector&, size_t /*events in window*/)>;

	EventPump(tracker::FeatureDetector& tracker, const EvaluatorParams& params = EvaluatorParams())
		: tracker_(tracker), params_(params)
	{
	}

	void onWindow(WindowCallback cb) { onWindow_ = std::move(cb); }

	// tools::Evaluator::eventCallback
	void eventCallback(const common::EventSample& sample)
	{
		tracker_.addEvent(sample);
		if ((sample.timestamp - tracker_.getLastCompensation()).count() >=
				static_cast<long long>(params_.compensationFrequencyTime) ||
			tracker_.getEvents().size() >= params_.compensationFrequencyEvents)
		{
			const size_t n = tracker_.getEvents().size();
			tracker_.compensateEventsContrast(tracker_.getEvents());
			tracker_.integrateEvents(tracker_.getEvents());
			if (onWindow_)
			{
				onWindow_(tracker_, n);
			}
			tracker_.clearEvents();
			++windows_;
		}
	}

	// Replayer: deliver every event of a recording in order.
	void replay(const std::vector<common::EventSample>& events)
	{
		for (const auto& e : events)
		{
			eventCallback(e);
		}
	}

	size_t windows() const { return windows_; }

	// Davis240cReader::getEvents on <dir>/events.txt
	static std::vector<common::EventSample> readEvents(const std::string& path, size_t maxEvents = 1u << 24)
	{
		std::vector<ebo_event> raw(maxEvents);
		size_t n = 0;
		const int rc = ebo_read_events_txt(path.c_str(), raw.data(), raw.size(), &n);
		if (rc != EBO_OK)
		{
			throw std::runtime_error("cannot parse " + path);  // the reference throws on a bad sign
		}
		std::vector<common::EventSample> out(n);
		for (size_t i = 0; i < n; ++i)
		{
			out[i].value.point = {raw[i].x, raw[i].y};
			out[i].value.sign = raw[i].sign > 0 ? common::POSITIVE : common::NEGATIVE;
			out[i].timestamp = common::timestamp_t(raw[i].t_us);
		}
		return out;
	}

	// the packed binary sidecar of the same recording (ebo_write_events_bin): no parsing
	static std::vector<common::EventSample> readEventsBin(const std::string& path, size_t maxEvents = 1u << 24)
	{
		std::vector<ebo_event> raw(maxEvents);
		size_t n = 0;
		if (ebo_read_events_bin(path.c_str(), raw.data(), raw.size(), &n) != EBO_OK)
		{
			throw std::runtime_error("cannot read " + path);
		}
		std::vector<common::EventSample> out(n);
		for (size_t i = 0; i < n; ++i)
		{
			out[i].value.point = {raw[i].x, raw[i].y};
			out[i].value.sign = raw[i].sign > 0 ? common::POSITIVE : common::NEGATIVE;
			out[i].timestamp = common::timestamp_t(raw[i].t_us);
		}
		return out;
	}
	static void writeEventsBin(const std::string& path, const std::vector<common::EventSample>& events)
	{
		const std::vector<ebo_event> raw = common::toEboEvents(events);
		if (ebo_write_events_bin(path.c_str(), raw.data(), raw.size()) != EBO_OK)
		{
			throw std::runtime_error("cannot write " + path);
		}
	}

   private:
	tracker::FeatureDetector& tracker_;
	EvaluatorParams params_;
	WindowCallback onWindow_;
	size_t windows_ = 0;
};

}  // namespace tools
