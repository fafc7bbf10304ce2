// optimizer_cost_lines_test.cpp — the reference's OWN statements that build and use the tracker's Ceres cost
// function, against the facade's <feature_tracker/optimizer_cost.h>.
//
// Verbatim blocks (compiled with -Wall -Wextra; cv:: types from the test-only OpenCV declarations in stubs_opencv/,
// ceres:: from the test-only declarations in stubs/ -- neither library is in this image):
//   * optimizer.cpp:5-13,15-31   Optimizer::Optimizer's grad_ sizing and the whole body of Optimizer::setGrad: the
//                                interleaved grid, `new Grid(grad_.data(), 0, h, 0, w)`, `new Interpolator(*grid)`;
//   * optimizer.cpp:72-79,86-97  currentRect / size / normalizedIntegratedNabla / warp / flowDir, `new
//                                tracker::OptimizerCostFunctor(...)` and `new ceres::AutoDiffCostFunction<
//                                tracker::OptimizerCostFunctor, ceres::DYNAMIC, Sophus::SE2d::num_parameters, 1>(c, size)`.
// Checked on the GPU: cost_function->Evaluate (residuals, both Jacobian blocks; one block; none) returns, bit for bit,
// what ebo_optimizer_eval returns for the same patch on another context; the functor's double path likewise.
// `--cpu`: compiles, and a functor without an interpolator reports failure instead of evaluating anywhere else.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
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

// A shell with the members of tracker::Optimizer that the verbatim lines name (optimizer.h:60-70)
struct OptimizerShell
{
	cv::Size2i imageSize_;
	std::vector<double> grad_;
	tracker::GridPtr gradGrid_;
	tracker::InterpolatorPtr gradInterpolator_;

	explicit OptimizerShell(const cv::Size2i& imageSize) : imageSize_(imageSize)
	{
		// ---- optimizer.cpp:9, verbatim ----
		grad_.resize((imageSize.height + 25) * (imageSize.width + 25) * 2);
	}

	// ---- optimizer.cpp:15-31, verbatim ----
	void setGrad(const cv::Mat& gradX, const cv::Mat& gradY)
	{
		for (int row = 0; row < gradX.rows; row++)
		{
			for (int col = 0; col < gradX.cols; col++)
			{
				grad_[2 * row * imageSize_.width + 2 * col] =
					gradX.at<double>(row, col);
				grad_[2 * row * imageSize_.width + 2 * col + 1] =
					gradY.at<double>(row, col);
			}
		}
		gradGrid_.reset(
			new Grid(grad_.data(), 0, imageSize_.height, 0, imageSize_.width));
		gradInterpolator_.reset(new Interpolator(*(gradGrid_.get())));
	}
	using Grid = tracker::Grid;                  // (the reference's file is inside namespace tracker)
	using Interpolator = tracker::Interpolator;

	std::unique_ptr<ceres::CostFunction> costFunctionOf(tracker::Patch& patch)
	{
		// ---- optimizer.cpp:72-79, verbatim ----
		const cv::Rect2d currentRect = patch.getPatch();
		int size = currentRect.height * currentRect.width;

		const cv::Mat normalizedIntegratedNabla =
			patch.getNormalizedIntegratedNabla();

		auto warp = patch.getWarp();
		double flowDir = patch.getFlow();
		(void)warp;
		(void)flowDir;
		// ---- optimizer.cpp:86-97, verbatim ----
		auto* c = new tracker::OptimizerCostFunctor(normalizedIntegratedNabla,
													gradInterpolator_.get(),
													currentRect, imageSize_);

		ceres::CostFunction* cost_function =
			new ceres::AutoDiffCostFunction<tracker::OptimizerCostFunctor,
											ceres::DYNAMIC,
											Sophus::SE2d::num_parameters, 1>(c,
																			size);
		return std::unique_ptr<ceres::CostFunction>(cost_function);
	}
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
	OptimizerShell shell(cv::Size2i(W, H));
	shell.setGrad(gradX, gradY);

	tracker::Patch patch({40.3, 30.6}, 12, common::timestamp_t(0));  // a fractional rect
	tracker::Mat64 nabla(25, 25);
	for (int i = 0; i < 625; ++i)
	{
		nabla.ptr()[i] = std::sin(0.37 * i) + 0.2 * std::cos(1.3 * i);
	}
	patch.setIntegratedNabla(nabla);
	patch.setFlowDir(0.7);
	patch.setWarp(common::Pose2d(0.03, common::Point2d(0.4, -0.3)));

	std::unique_ptr<ceres::CostFunction> cost = shell.costFunctionOf(patch);
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
	tracker::OptimizerCostFunctor bad(tracker::Mat64(7, 7), shell.gradInterpolator_.get(), r, tracker::Size(W, H));
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
