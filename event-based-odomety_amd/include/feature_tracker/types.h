// feature_tracker/types.h — the OpenCV value types the reference's tracker headers use
// (cv::Size, cv::Rect2i, cv::Rect2d, cv::Mat of CV_64F, cv::Point2d as tracker::Corner;
// implementation/feature_tracker/include/feature_tracker/patch.h:10-13,
// feature_detector.h:17,24), as plain structs so that the façade builds without OpenCV.
// Member names are OpenCV's, so code written against the reference's types reads the same.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../common/data_types.h"

namespace tracker
{
#ifdef EBO_HAVE_OPENCV
// OpenCV is there (common/data_types.h found <opencv2/core.hpp>): the reference's own types, no stand-ins.  Mat64 is a
// cv::Mat of CV_64F with the two conveniences the facade's own code uses, so a `Mat64 const&` IS a `cv::Mat const&`
// (tools::Evaluator::getCompensatedEventImage returns one unchanged).
using Size = cv::Size;
using Rect2i = cv::Rect2i;
using Rect2d = cv::Rect2d;
class Mat64 : public cv::Mat
{
   public:
	Mat64() = default;
	Mat64(int r, int c) : cv::Mat(cv::Mat::zeros(r, c, CV_64F)) {}
	Mat64(const cv::Mat& m) : cv::Mat(m) {}  // a CV_64F image of the caller's (cv::Mat copies are shallow)
	double* ptr() { return cv::Mat::ptr<double>(0); }
	const double* ptr() const { return cv::Mat::ptr<double>(0); }
};
#else
struct Size
{
	int width = 0;
	int height = 0;
	Size() = default;
	Size(int w, int h) : width(w), height(h) {}
};

// cv::Rect2i stand-in; contains() is half-open like cv::Rect_::contains.
struct Rect2i
{
	int x = 0, y = 0, width = 0, height = 0;
	Rect2i() = default;
	Rect2i(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
	bool contains(const common::Point2i& p) const
	{
		return x <= p.x && p.x < x + width && y <= p.y && p.y < y + height;
	}
};

// cv::Rect2d stand-in; contains() is half-open on the integer point, as cv::Rect_::contains.
struct Rect2d
{
	double x = 0, y = 0, width = 0, height = 0;
	Rect2d() = default;
	Rect2d(double x_, double y_, double w_, double h_) : x(x_), y(y_), width(w_), height(h_) {}
	bool contains(const common::Point2i& p) const
	{
		return x <= p.x && p.x < x + width && y <= p.y && p.y < y + height;
	}
};

// CV_64F single-channel image stand-in.
class Mat64
{
   public:
	int rows = 0;
	int cols = 0;
	Mat64() = default;
	Mat64(int r, int c) : rows(r), cols(c), data_(static_cast<size_t>(r) * c, 0.0) {}
	template <typename T = double>
	T& at(int r, int c)
	{
		return data_[static_cast<size_t>(r) * cols + c];
	}
	template <typename T = double>
	const T& at(int r, int c) const
	{
		return data_[static_cast<size_t>(r) * cols + c];
	}
	double* ptr() { return data_.data(); }
	const double* ptr() const { return data_.data(); }
	bool empty() const { return data_.empty(); }

   private:
	std::vector<double> data_;
};

#endif

using Corner = common::Point2d;        // patch.h:10
using Corners = std::vector<Corner>;   // patch.h:11
using TrackId = int32_t;               // patch.h:12

}  // namespace tracker
