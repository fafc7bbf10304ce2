// TEST-ONLY declarations of the handful of OpenCV types the reference's tracker headers use (cv::Point_, cv::Size_,
// cv::Rect_, cv::Mat of CV_8U / CV_64F), written from OpenCV's published API, so that the EBO_HAVE_OPENCV branch of the
// facade headers (feature_tracker/types.h, common/data_types.h) meets a compiler in an image without OpenCV.  Never on
// the product's include path; with the real <opencv2/core.hpp> installed this directory is not used.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_64F 6

namespace cv
{
template <typename T>
struct Point_
{
	T x, y;
	Point_() : x(0), y(0) {}
	Point_(T x_, T y_) : x(x_), y(y_) {}
	template <typename U>
	operator Point_<U>() const
	{
		return Point_<U>(static_cast<U>(x), static_cast<U>(y));  // (saturate_cast in OpenCV: round-half-even for double -> int)
	}
};
using Point2i = Point_<int>;
using Point2f = Point_<float>;
using Point2d = Point_<double>;
using Point = Point2i;

template <typename T>
struct Size_
{
	T width, height;
	Size_() : width(0), height(0) {}
	Size_(T w, T h) : width(w), height(h) {}
};
using Size2i = Size_<int>;
using Size = Size2i;

template <typename T>
struct Rect_
{
	T x, y, width, height;
	Rect_() : x(0), y(0), width(0), height(0) {}
	Rect_(T x_, T y_, T w_, T h_) : x(x_), y(y_), width(w_), height(h_) {}
	Point_<T> tl() const { return Point_<T>(x, y); }
	Point_<T> br() const { return Point_<T>(x + width, y + height); }
	bool contains(const Point_<T>& p) const { return x <= p.x && p.x < x + width && y <= p.y && p.y < y + height; }
};
using Rect2i = Rect_<int>;
using Rect2d = Rect_<double>;
using Rect = Rect2i;

// cv::Mat: a reference-counted header over a buffer (copies share the data, clone() does not)
class Mat
{
   public:
	int rows = 0, cols = 0;
	Mat() = default;
	Mat(int r, int c, int type) : rows(r), cols(c), type_(type), buf_(std::make_shared<std::vector<unsigned char>>(static_cast<size_t>(r) * c * elem(type)))
	{
		data = buf_->data();
	}
	Mat(int r, int c, int type, void* external) : rows(r), cols(c), data(static_cast<unsigned char*>(external)), type_(type) {}
	static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }  // (the vector is value-initialised)
	static Mat zeros(Size s, int type) { return Mat(s.height, s.width, type); }
	int type() const { return type_; }
	bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
	Size size() const { return Size(cols, rows); }
	template <typename T>
	T& at(int r, int c)
	{
		return reinterpret_cast<T*>(data)[static_cast<size_t>(r) * cols + c];
	}
	template <typename T>
	const T& at(int r, int c) const
	{
		return reinterpret_cast<const T*>(data)[static_cast<size_t>(r) * cols + c];
	}
	template <typename T = unsigned char>
	T* ptr(int r = 0)
	{
		return reinterpret_cast<T*>(data) + static_cast<size_t>(r) * cols;
	}
	template <typename T = unsigned char>
	const T* ptr(int r = 0) const
	{
		return reinterpret_cast<const T*>(data) + static_cast<size_t>(r) * cols;
	}
	Mat clone() const
	{
		Mat m(rows, cols, type_);
		if (!empty())
		{
			std::copy(data, data + static_cast<size_t>(rows) * cols * elem(type_), m.data);
		}
		return m;
	}
	unsigned char* data = nullptr;

   private:
	static size_t elem(int type) { return type == CV_64F ? 8 : 1; }
	int type_ = CV_8U;
	std::shared_ptr<std::vector<unsigned char>> buf_;
};
}  // namespace cv
