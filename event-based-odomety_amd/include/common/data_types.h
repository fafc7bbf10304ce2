// common/data_types.h — the event types of the reference
// (common/include/common/data_types.h:10-38, common/include/common/geometry.h:10-14)
// without the OpenCV/Sophus dependency: only what the event-warping path touches.
// Field names, order and layout are the reference's, so code written against
// common::EventSample compiles unchanged and an array of EventSample can be handed to
// the C ABI as an array of ebo_event.
#pragma once

#include <chrono>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <list>
#include <vector>

#include "../../../include/ebo.h"

namespace common
{
// cv::Point2i stand-in (two ints, x then y).
struct Point2i
{
	int x = 0;
	int y = 0;
	Point2i() = default;
	Point2i(int x_, int y_) : x(x_), y(y_) {}
};

struct Point2d
{
	double x = 0.0;
	double y = 0.0;
	Point2d() = default;
	Point2d(double x_, double y_) : x(x_), y(y_) {}
};

using timestamp_t = std::chrono::microseconds;

template <typename T>
struct Sample
{
	Sample(const T& value_, const timestamp_t timestamp_) : value(value_), timestamp(timestamp_) {}
	Sample() {}

	T value;
	timestamp_t timestamp;
};

enum EventPolarity
{
	NEGATIVE = -1,
	POSITIVE = 1
};

struct Event
{
	Point2i point;
	EventPolarity sign;
};

using EventSample = Sample<Event>;
using EventSequence = std::deque<EventSample>;

static_assert(sizeof(EventSample) == sizeof(ebo_event), "EventSample must match ebo_event");
static_assert(offsetof(EventSample, timestamp) == offsetof(ebo_event, t_us), "timestamp offset");
static_assert(offsetof(Event, sign) == offsetof(ebo_event, sign), "sign offset");

// list / deque of EventSample -> contiguous ebo_event array for the C ABI.
template <class Container>
inline std::vector<ebo_event> toEboEvents(const Container& events)
{
	std::vector<ebo_event> out;
	out.reserve(events.size());
	for (const auto& e : events)
	{
		ebo_event r;
		r.x = e.value.point.x;
		r.y = e.value.point.y;
		r.sign = static_cast<int32_t>(e.value.sign);
		r.reserved = 0;
		r.t_us = e.timestamp.count();
		out.push_back(r);
	}
	return out;
}

}  // namespace common
