// common/data_types.h — the event types of the reference
// (common/include/common/data_types.h:10-38, common/include/common/geometry.h:10-14)
// without the OpenCV/Sophus dependency: only what the event-warping path touches.
// Field names, order and layout are the reference's, so code written against
// common::EventSample compiles unchanged and an array of EventSample can be handed to
// the C ABI as an array of ebo_event.
#pragma once

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <list>
#include <vector>

#include "../../../include/ebo.h"

// With OpenCV on the include path the OpenCV value types ARE the reference's types (common/include/common/geometry.h,
// data_types.h:39): the stand-ins below give way to them, so that code written against the reference -- cv::Point2i in
// an Event, cv::Mat in an ImageSample -- compiles without a rename.
#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define EBO_HAVE_OPENCV 1
#endif
#endif

namespace common
{
#ifdef EBO_HAVE_OPENCV
using Point2i = cv::Point2i;
using Point2d = cv::Point2d;
#else
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
#endif

using timestamp_t = std::chrono::microseconds;

// Sophus::SE2d stand-in for common::Pose2d (common/include/common/geometry.h): the storage
// the tracker hands to Ceres — unit complex (cos, sin), then the translation — and the few
// operations the tracker path calls (matrix2x3, inverse, data).
struct Pose2d
{
	double d[4] = {1.0, 0.0, 0.0, 0.0};
	Pose2d() = default;
	Pose2d(double theta, const Point2d& t);
	double* data() { return d; }
	const double* data() const { return d; }
	struct Matrix2x3
	{
		double m[2][3];
		double operator()(int r, int c) const { return m[r][c]; }
	};
	Matrix2x3 matrix2x3() const
	{
		return Matrix2x3{{{d[0], -d[1], d[2]}, {d[1], d[0], d[3]}}};
	}
	Pose2d inverse() const
	{
		Pose2d r;
		const double c = d[0], s = -d[1];
		r.d[0] = c;
		r.d[1] = s;
		r.d[2] = -(c * d[2] - s * d[3]);
		r.d[3] = -(s * d[2] + c * d[3]);
		return r;
	}
};

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

// CV_8U single-channel cv::Mat stand-in: what ImageSample carries to FeatureDetector::newImage
// (data_types.h:39).  The event-warping path never reads pixels; the front-end hooks do.
#ifdef EBO_HAVE_OPENCV
using Image8 = cv::Mat;
#else
struct Image8
{
	int rows = 0;
	int cols = 0;
	std::vector<uint8_t> data;  // row-major
	Image8() = default;
	Image8(int r, int c) : rows(r), cols(c), data(static_cast<size_t>(r) * c, 0) {}
};
#endif

using EventSample = Sample<Event>;
using ImageSample = Sample<Image8>;
using EventSequence = std::deque<EventSample>;
using ImageSequence = std::vector<ImageSample>;

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

inline Pose2d::Pose2d(double theta, const Point2d& t)
{
	d[0] = std::cos(theta);
	d[1] = std::sin(theta);
	d[2] = t.x;
	d[3] = t.y;
}

}  // namespace common
