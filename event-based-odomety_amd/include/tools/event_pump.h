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

#include <cstdio>
#include <cstdlib>
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

// tools::Replayer (tools/replayer/src/replayer.cpp:42-131) for the two streams that decide WHEN the
// tracker is called: events and image TIMESTAMPS (the frames themselves belong to the frame-based
// detector, out of scope; an ImageSample here is a timestamp and the file name of images.txt).
// next() delivers the earlier of the next event and the next image (an image wins a tie, :73);
// nextInterval(d) plays until d has passed since the first sample delivered; nextImage() plays up to
// and including the next image; finished() = the events ran out (a next() found none left) or no
// image is left, as in the reference.
enum class EventType
{
	EVENT,
	IMAGE
};
struct ImageStamp
{
	common::timestamp_t timestamp;
	std::string file;
};

class StreamPump
{
   public:
	StreamPump(std::vector<common::EventSample> events, std::vector<ImageStamp> images)
		: events_(std::move(events)), images_(std::move(images))
	{
		hasEvents_ = !events_.empty();
		reset();
	}

	// <dir>/events.txt and <dir>/images.txt ("<seconds> <file>" per line, Davis240cReader::getImages)
	static StreamPump fromDirectory(const std::string& dir)
	{
		std::vector<ImageStamp> images;
		FILE* fp = std::fopen((dir + "/images.txt").c_str(), "rb");
		if (!fp)
		{
			throw std::runtime_error("cannot open " + dir + "/images.txt");
		}
		char line[1024];
		while (std::fgets(line, sizeof(line), fp))
		{
			char* end = nullptr;
			const double sec = std::strtod(line, &end);
			if (end == line)
			{
				continue;
			}
			std::string file(end);
			while (!file.empty() && (file.back() == '\n' || file.back() == '\r' || file.back() == ' '))
			{
				file.pop_back();
			}
			while (!file.empty() && file.front() == ' ')
			{
				file.erase(file.begin());
			}
			images.push_back({common::timestamp_t(static_cast<int64_t>(sec * 1000000.0)), file});
		}
		std::fclose(fp);
		return StreamPump(EventPump::readEvents(dir + "/events.txt"), std::move(images));
	}

	void addEventCallback(std::function<void(const common::EventSample&)> cb) { eventCallbacks_.push_back(std::move(cb)); }
	void addImageCallback(std::function<void(const ImageStamp&)> cb) { imageCallbacks_.push_back(std::move(cb)); }

	bool finished() const { return !hasEvents_ || imageIt_ == images_.size(); }

	void reset()
	{
		eventIt_ = 0;
		imageIt_ = 0;
		hasEvents_ = !events_.empty();
		lastTimestamp_ = common::timestamp_t(0);
		imageArrived_ = false;
	}

	void next()
	{
		if (hasEvents_ && eventIt_ == events_.size())
		{
			hasEvents_ = false;  // the reader has no further chunk (replayer.cpp:58-71)
		}
		const bool haveEvent = eventIt_ < events_.size(), haveImage = imageIt_ < images_.size();
		if (haveEvent && (!haveImage || events_[eventIt_].timestamp < images_[imageIt_].timestamp))
		{
			lastTimestamp_ = events_[eventIt_].timestamp;
			for (auto& cb : eventCallbacks_)
			{
				cb(events_[eventIt_]);
			}
			++eventIt_;
		}
		else if (haveImage)
		{
			lastTimestamp_ = images_[imageIt_].timestamp;
			imageArrived_ = true;
			for (auto& cb : imageCallbacks_)
			{
				cb(images_[imageIt_]);
			}
			++imageIt_;
		}
	}

	void nextInterval(const common::timestamp_t& interval)
	{
		if (finished())
		{
			return;
		}
		next();
		const auto firstTime = lastTimestamp_;
		do
		{
			next();
		} while ((lastTimestamp_ - firstTime) < interval && !finished());
	}

	void nextImage()
	{
		if (finished())
		{
			return;
		}
		imageArrived_ = false;
		while (!imageArrived_ && (eventIt_ < events_.size() || imageIt_ < images_.size()))
		{
			next();
		}
	}

   private:
	std::vector<common::EventSample> events_;
	std::vector<ImageStamp> images_;
	size_t eventIt_ = 0, imageIt_ = 0;
	bool hasEvents_ = false, imageArrived_ = false;
	common::timestamp_t lastTimestamp_{0};
	std::vector<std::function<void(const common::EventSample&)>> eventCallbacks_;
	std::vector<std::function<void(const ImageStamp&)>> imageCallbacks_;
};

}  // namespace tools
