// optimizer_cost_lines_test.cpp — the tracker's Ceres cost function built through the facade's
// <feature_tracker/optimizer_cost.h> the way the reference's Optimizer builds it.
//
// The reference constructs, per patch (optimizer.cpp:9,15-31,72-97): an interleaved gradient grid, `Grid` and
// `Interpolator` over it, `OptimizerCostFunctor(nabla, interpolator, rect, imageSize)` and
// `ceres::AutoDiffCostFunction<tracker::OptimizerCostFunctor, ceres::DYNAMIC, Sophus::SE2d::num_parameters, 1>`.
// This file does the same with its own statements (GradientHolder) and pins the constructor signatures with
// static_asserts; cv:: types come from the test-only OpenCV declarations in stubs_opencv/, ceres:: from the test-only
// declarations in stubs/ (neither library is in this image).  Compiled with -Wall -Wextra.
// Checked on the GPU: cost_function->Evaluate (residuals, both Jacobian blocks; one block; none) returns, bit for bit,
// what ebo_optimizer_eval returns for the same patch on another context; the functor's double path likewise.
// `--cpu`: compiles, and a functor without an interpolator reports failure instead of evaluating anywhere else.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include <ceres/ceres.h>
#include <common/data_types.h>
#include <feature_tracker/optimizer.h>
#include <feature_tracker/optimizer_cost.h>
#include <feature_tracker/patch.h>

static int g_fail = 0;
#define EXPECT_TRUE(c)                                                 \
	do                                                                 \
	{                                                                  \
		if (!(c))                                                      \
		{                                                              \
			std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
			++g_fail;                                                  \
		}                                                              \
	} while (0)

namespace Sophus  // test-only: the one constant of Sophus the construction line names
{
struct SE2d
{
	static constexpr int num_parameters = 4;
};
}  // namespace Sophus

// What tracker::Optimizer holds around its Ceres cost function (optimizer.h:60-70), as this test's own holder: the
// interleaved gradient grid, the Grid / Interpolator pair over it, and the construction of the cost function.  The
// INTERFACE it exercises is the reference's -- Grid(data, row_begin, row_end, col_begin, col_end), Interpolator(grid),
// OptimizerCostFunctor(nabla, interpolator, rect, imageSize), AutoDiffCostFunction<Functor, DYNAMIC, 4, 1>(functor, n)
// (optimizer.cpp:9,15-31,72-97) -- the statements and names are this file's.
using SE2Parameters = std::integral_constant<int, Sophus::SE2d::num_parameters>;
static_assert(SE2Parameters::value == 4, "an SE2 block has four parameters");
static_assert(std::is_constructible<tracker::Grid, const double*, int, int, int, int>::value, "Grid(data, r0, r1, c0, c1)");
static_assert(std::is_constructible<tracker::Interpolator, const tracker::Grid&>::value, "Interpolator(grid)");
static_assert(std::is_constructible<tracker::OptimizerCostFunctor, const cv::Mat&, tracker::Interpolator*, const cv::Rect2d&,
									 const cv::Size2i&>::value,
			  "OptimizerCostFunctor(nabla, interpolator, rect, imageSize)");
static_assert(std::is_same<tracker::GridPtr::element_type, tracker::Grid>::value &&
				  std::is_same<tracker::InterpolatorPtr::element_type, tracker::Interpolator>::value,
			  "GridPtr / InterpolatorPtr own a Grid / an Interpolator");

class GradientHolder
{
   public:
	explicit GradientHolder(const cv::Size2i& sensor)
		: sensor_(sensor), samples_(2u * static_cast<size_t>(sensor.height + 25) * static_cast<size_t>(sensor.width + 25), 0.0)
	{  // (the reference sizes its buffer with a 25-pixel margin on both axes)
	}

	// two values per pixel, x-gradient first, rows of `sensor.width` pixels
	void install(const cv::Mat& gx, const cv::Mat& gy)
	{
		const size_t pitch = 2u * static_cast<size_t>(sensor_.width);
		for (int r = 0; r < gx.rows; ++r)
		{
			double* line = samples_.data() + pitch * static_cast<size_t>(r);
			for (int c = 0; c < gx.cols; ++c)
			{
				line[2 * c] = gx.at<double>(r, c);
				line[2 * c + 1] = gy.at<double>(r, c);
			}
		}
		grid_ = tracker::GridPtr(new tracker::Grid(samples_.data(), 0, sensor_.height, 0, sensor_.width));
		lookup_ = tracker::InterpolatorPtr(new tracker::Interpolator(*grid_));
	}

	// the per-patch cost function Optimizer::optimize hands to Ceres: one residual per patch pixel, an SE2 block and a
	// one-parameter flow-direction block
	std::unique_ptr<ceres::CostFunction> costOf(const tracker::Patch& patch) const
	{
		const cv::Rect2d rect = patch.getPatch();
		const int residuals = static_cast<int>(rect.height * rect.width);
		const cv::Mat unitNabla = patch.getNormalizedIntegratedNabla();
		tracker::OptimizerCostFunctor* functor = new tracker::OptimizerCostFunctor(unitNabla, lookup_.get(), rect, sensor_);
		using Cost = ceres::AutoDiffCostFunction<tracker::OptimizerCostFunctor, ceres::DYNAMIC, SE2Parameters::value, 1>;
		return std::unique_ptr<ceres::CostFunction>(new Cost(functor, residuals));
	}

	tracker::Interpolator* lookup() const { return lookup_.get(); }

   private:
	cv::Size2i sensor_;
	std::vector<double> samples_;
	tracker::GridPtr grid_;
	tracker::InterpolatorPtr lookup_;
};

static bool sameBits(const std::vector<double>& a, const std::vector<double>& b)
{
	return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0;
}

static void deviceTest()
{
	const int W = 96, H = 72;
	cv::Mat gradX = cv::Mat::zeros(H, W, CV_64F), gradY = cv::Mat::zeros(H, W, CV_64F);
	for (int y = 0; y < H; ++y)
	{
		for (int x = 0; x < W; ++x)
		{
			gradX.at<double>(y, x) = std::sin(0.21 * x) * std::cos(0.13 * y);
			gradY.at<double>(y, x) = std::cos(0.17 * x + 0.3) * std::sin(0.19 * y);
		}
	}
	GradientHolder shell(cv::Size2i(W, H));
	shell.install(gradX, gradY);

	tracker::Patch patch({40.3, 30.6}, 12, common::timestamp_t(0));  // a fractional rect
	tracker::Mat64 nabla(25, 25);
	for (int i = 0; i < 625; ++i)
	{
		nabla.ptr()[i] = std::sin(0.37 * i) + 0.2 * std::cos(1.3 * i);
	}
	patch.setIntegratedNabla(nabla);
	patch.setFlowDir(0.7);
	patch.setWarp(common::Pose2d(0.03, common::Point2d(0.4, -0.3)));

	std::unique_ptr<ceres::CostFunction> cost = shell.costOf(patch);
	EXPECT_TRUE(cost->num_residuals() == 625);
	EXPECT_TRUE(cost->parameter_block_sizes().size() == 2 && cost->parameter_block_sizes()[0] == 4 &&
				cost->parameter_block_sizes()[1] == 1);

	// the same patch through the batched ABI on ANOTHER context with the same gradients
	tracker::Optimizer direct(tracker::OptimizerParams(), tracker::Size(W, H));
	direct.setGrad(gradX, gradY);
	const tracker::Mat64 nn = patch.getNormalizedIntegratedNabla();
	const tracker::Rect2d r = patch.getPatch();
	const double rect[4] = {r.x, r.y, r.width, r.height};
	const double flow = patch.getFlow();
	std::vector<double> res0(625), jp0(2500), jf0(625), resV(625);
	EXPECT_TRUE(ebo_optimizer_eval(direct.handle(), 1, rect, nn.ptr(), patch.getWarp().data(), &flow, res0.data(), jp0.data(),
								   jf0.data()) == EBO_OK);
	EXPECT_TRUE(ebo_optimizer_eval(direct.handle(), 1, rect, nn.ptr(), patch.getWarp().data(), &flow, resV.data(), nullptr,
								   nullptr) == EBO_OK);

	const double* params[2] = {patch.getWarp().data(), &flow};
	std::vector<double> res(625), jp(2500), jf(625);
	double* jac[2] = {jp.data(), jf.data()};
	EXPECT_TRUE(cost->Evaluate(params, res.data(), jac));
	EXPECT_TRUE(sameBits(res, res0) && sameBits(jp, jp0) && sameBits(jf, jf0));
	// one block only, then none (the double path)
	std::vector<double> res1(625), jf1(625);
	double* jacFlowOnly[2] = {nullptr, jf1.data()};
	EXPECT_TRUE(cost->Evaluate(params, res1.data(), jacFlowOnly));
	EXPECT_TRUE(sameBits(res1, res0) && sameBits(jf1, jf0));
	std::vector<double> res2(625);
	EXPECT_TRUE(cost->Evaluate(params, res2.data(), nullptr));
	EXPECT_TRUE(sameBits(res2, resV));
	double norm = 0, jn = 0;
	for (int i = 0; i < 625; ++i)
	{
		norm += res[static_cast<size_t>(i)] * res[static_cast<size_t>(i)];
		jn += std::fabs(jf[static_cast<size_t>(i)]) + std::fabs(jp[4 * static_cast<size_t>(i) + 2]);
	}
	EXPECT_TRUE(norm > 0.1 && std::isfinite(norm) && jn > 0.1 && std::isfinite(jn));
	// a functor whose nabla does not fit its rect refuses
	tracker::OptimizerCostFunctor bad(tracker::Mat64(7, 7), shell.lookup(), r, tracker::Size(W, H));
	EXPECT_TRUE(!bad(patch.getWarp().data(), &flow, res.data()));
}

int main(int argc, char** argv)
{
	const bool cpuOnly = argc > 1 && std::string(argv[1]) == "--cpu";
	{
		tracker::OptimizerCostFunctor none(tracker::Mat64(3, 3), nullptr, tracker::Rect2d(0, 0, 3, 3), tracker::Size(10, 10));
		const double pose[4] = {1, 0, 0, 0}, flow = 0.0;
		double out[9];
		EXPECT_TRUE(!none(pose, &flow, out));  // no interpolator = no device context: failure, nothing evaluated elsewhere
	}
	if (!cpuOnly)
	{
		deviceTest();
	}
	std::printf("optimizer_cost_lines_test%s: %s (%d failure%s)\n", cpuOnly ? " --cpu" : "", g_fail ? "FAILED" : "OK", g_fail,
				g_fail == 1 ? "" : "s");
	return g_fail ? 1 : 0;
}
