// patch_lines_test.cpp — the facade's tracker::Patch under the scenarios of the reference's patch tests.
//
// implementation/feature_tracker/test/patch_test.cpp holds three tests (addEventsTest :7-33, integrateEventsTest :35-60,
// warpImageTest :62-91).  Their SCENARIOS and known answers are restated here with this file's own statements (the
// integrateEvents known answer is a data table), and a member-pointer table pins every public member of the reference's
// Patch (patch.h:15-160) by name, argument and result type: source compatibility is shown by conformance, not by
// carrying the reference's test text.  The per-patch members the reference's Patch has besides the bookkeeping --
// integrateEvents, integrateMotionCompensatedEvents, warpImage(), setGrad / getGradX / getGradY,
// getNormalizedIntegratedNabla, getCostMap / setCostMap, getInitPatch -- run on the device through the context the patch
// is bound to (or Patch::setDefaultContext) and throw without one.
//
// Checked on the GPU: the two device scenarios of the reference; integrateMotionCompensatedEvents against a restatement
// of patch.cpp:87-130 written out in this file; warpImage() on a patch away from the border = the batched ABI call;
// OptimizerParams::drawCostMap through tracker::Optimizer::optimize = ebo_optimizer_cost_map called directly with the
// functor's rect / nabla from before the solve and the solved pose.
// `--cpu`: the host-only subset (the event window, the getters, and that every device member throws without a context).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <list>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include <common/data_types.h>
#include <feature_tracker/optimizer.h>
#include <feature_tracker/patch.h>

static int g_fail = 0;
#define EXPECT_TRUE(c)                                                 \
	do                                                                 \
	{                                                                  \
		const bool ok_ = static_cast<bool>(c);                         \
		if (!ok_)                                                      \
		{                                                              \
			std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
			++g_fail;                                                  \
		}                                                              \
	} while (0)
#define EXPECT_FALSE(c) EXPECT_TRUE(!(c))
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == static_cast<decltype(a)>(b))
#define ASSERT_EQ(a, b) EXPECT_TRUE((a) == static_cast<decltype(a)>(b))
#define EXPECT_LE(a, b) EXPECT_TRUE((a) <= (b))
#define EXPECT_FLOAT_EQ(a, b) EXPECT_TRUE(static_cast<float>(a) == static_cast<float>(b))

static common::EventSample eventAt(int x, int y, int64_t t, bool positive)
{
	common::EventSample e;
	e.timestamp = common::timestamp_t(t);
	e.value.point = {x, y};
	e.value.sign = positive ? common::EventPolarity::POSITIVE : common::EventPolarity::NEGATIVE;
	return e;
}

// ---- signature conformance with implementation/feature_tracker/include/feature_tracker/patch.h:15-160 ----------
// Every member below is named by the reference's class with these argument / result types (images: the facade's image
// type, which IS cv::Mat when OpenCV is on the include path).  A member-pointer cast only compiles for an exact match.
namespace conformance
{
using P = tracker::Patch;
using Img = tracker::Mat64;
static_assert(std::is_constructible<P, const tracker::Corner&, int, const common::timestamp_t&>::value, "Patch(corner, extent, t)");
[[maybe_unused]] static const auto kMembers = std::make_tuple(
	static_cast<void (P::*)()>(&P::init), static_cast<void (P::*)(const common::EventSample&)>(&P::addEvent),
	static_cast<void (P::*)()>(&P::integrateEvents), static_cast<void (P::*)()>(&P::integrateMotionCompensatedEvents),
	static_cast<void (P::*)()>(&P::resetBatch), static_cast<void (P::*)()>(&P::addTrajectoryPosition),
	static_cast<void (P::*)(double)>(&P::addFinalCost), static_cast<void (P::*)()>(&P::updatePatchRect),
	static_cast<tracker::Corner (P::*)() const>(&P::toCorner), static_cast<bool (P::*)(const common::Point2i&) const>(&P::isInPatch),
	static_cast<bool (P::*)() const>(&P::isReady), static_cast<bool (P::*)() const>(&P::isLost),
	static_cast<void (P::*)()>(&P::warpImage), static_cast<common::EventSequence const& (P::*)() const>(&P::getEvents),
	static_cast<Img const& (P::*)() const>(&P::getIntegratedNabla), static_cast<Img const& (P::*)() const>(&P::getPredictedNabla),
	static_cast<tracker::Rect2d const& (P::*)() const>(&P::getPatch), static_cast<tracker::TrackId (P::*)() const>(&P::getTrackId),
	static_cast<const common::Pose2d& (P::*)() const>(&P::getWarp), static_cast<float (P::*)() const>(&P::getFlow),
	static_cast<Img (P::*)() const>(&P::getNormalizedIntegratedNabla), static_cast<const Img& (P::*)() const>(&P::getCostMap),
	static_cast<std::vector<common::Sample<common::Point2d>> const& (P::*)() const>(&P::getTrajectory),
	static_cast<size_t (P::*)() const>(&P::getNumOfEvents), static_cast<tracker::Rect2d (P::*)() const>(&P::getInitPatch),
	static_cast<common::timestamp_t (P::*)() const>(&P::getCurrentTimestamp),
	static_cast<common::timestamp_t (P::*)() const>(&P::getTimeWithoutUpdate),
	static_cast<const std::vector<double>& (P::*)() const>(&P::getFinalCosts),
	static_cast<common::timestamp_t (P::*)() const>(&P::getTimeLastUpdate),
	static_cast<Img const& (P::*)() const>(&P::getCompenatedIntegratedNabla),
	static_cast<common::timestamp_t (P::*)() const>(&P::getInitTime), static_cast<Img const& (P::*)() const>(&P::getGradX),
	static_cast<Img const& (P::*)() const>(&P::getGradY), static_cast<void (P::*)()>(&P::setLost),
	static_cast<void (P::*)(size_t)>(&P::setNumOfEvents), static_cast<void (P::*)(tracker::TrackId)>(&P::setTrackId),
	static_cast<void (P::*)(const double)>(&P::setFlowDir), static_cast<void (P::*)(const common::Pose2d&)>(&P::setWarp),
	static_cast<void (P::*)(const Img&)>(&P::setCostMap), static_cast<void (P::*)(const Img&)>(&P::setIntegratedNabla),
	static_cast<void (P::*)(const tracker::Corner&, const common::timestamp_t&)>(&P::setCorner),
	static_cast<void (P::*)(const Img&, const Img&)>(&P::setGrad), static_cast<void (P::*)(const common::timestamp_t&)>(&P::setTs),
	static_cast<void (P::*)(const Img&)>(&P::setMotionCompensatedIntegratedNabla),
	static_cast<void (P::*)(const common::timestamp_t&)>(&P::setTimeWithoutUpdate));
static_assert(std::is_same<tracker::Patches, std::list<tracker::Patch>>::value, "Patches is a std::list<Patch>");
}  // namespace conformance

// ---- the event window of a patch (the members patch_test.cpp:7-33 names; own scenario) -------------------------
// What patch.cpp implies, asserted directly: setNumOfEvents clamps into [100, 300] (patch.cpp:208-212), so thirty
// events do not make a patch ready; addEvent pushes to the FRONT (patch.cpp:37-47), so front() is the newest event.
// (The reference's own test expects the opposite on three counts and cannot pass against the reference's patch.cpp.)
static void eventWindowScenario()
{
	const int kEvents = 30;
	tracker::Patch p(tracker::Corner(10, 10), 5, common::timestamp_t(0));
	p.setNumOfEvents(kEvents);
	EXPECT_TRUE(p.getNumOfEvents() == 100);
	uint32_t lcg = 2463534242u;
	for (int k = 0; k < kEvents; ++k)
	{
		lcg = lcg * 1664525u + 1013904223u;
		p.addEvent(eventAt(5 + static_cast<int>((lcg >> 8) % 10), 5 + static_cast<int>((lcg >> 16) % 10), k, (lcg >> 24) & 1u));
	}
	EXPECT_FALSE(p.isReady());
	const common::EventSequence& window = p.getEvents();
	EXPECT_TRUE(window.size() == static_cast<size_t>(kEvents));
	EXPECT_TRUE(window.front().timestamp == common::timestamp_t(kEvents - 1));
	EXPECT_TRUE(window.back().timestamp == common::timestamp_t(0));
	// 100 events: ready (counter >= 30 and the window holds numOfEvents); resetBatch takes the readiness away
	for (int k = kEvents; k < 100; ++k)
	{
		p.addEvent(eventAt(10, 10, k, true));
	}
	EXPECT_TRUE(p.isReady());
	p.resetBatch();
	EXPECT_FALSE(p.isReady());
}

// ---- the reference-held known answer for Patch::integrateEvents (patch_test.cpp:35-60), as DATA ------------------
// Scenario: a 7 x 7 patch around (10, 10); event k of 30 sits at column 7 + k / 7, row 7 + k % 7 with alternating
// polarity starting positive.  Known answer: the signed count image holds +1 / -1 at (row k % 7, column k / 7) -- the
// same table tests/test_oracle_golden.py holds for the oracle -- and, beyond what the reference checks, 0 elsewhere.
static void signedCountKnownAnswer()
{
	const int side = 7, nEvents = 30;
	double expected[side][side] = {};
	tracker::Patch p(tracker::Corner(10, 10), 3, common::timestamp_t(0));
	p.setNumOfEvents(nEvents);
	for (int k = 0; k < nEvents; ++k)
	{
		const int col = k / side, row = k % side;
		const bool positive = (k & 1) == 0;
		expected[row][col] = positive ? 1.0 : -1.0;
		p.addEvent(eventAt(7 + col, 7 + row, k, positive));
	}
	p.integrateEvents();
	const tracker::Mat64& image = p.getIntegratedNabla();
	EXPECT_TRUE(image.rows == side && image.cols == side);
	for (int row = 0; row < side; ++row)
	{
		for (int col = 0; col < side; ++col)
		{
			EXPECT_TRUE(image.at<double>(row, col) == expected[row][col]);
		}
	}
	// patch.cpp:65-85 also sets the two times: the int32 mid time of newest / oldest, and the OLDEST event's time
	EXPECT_TRUE(p.getCurrentTimestamp() == common::timestamp_t(14));
	EXPECT_TRUE(p.getTimeLastUpdate() == common::timestamp_t(0));
}

// ---- the reference's warpImage scenario (patch_test.cpp:62-91), own statements ----------------------------------
// An 11 x 11 gradient pair (a vertical line of ones in gradX, a horizontal one in gradY, both through the centre), a
// patch of extent 5 at (5, 5), flow direction and warp rotation both pi / 4.  The reference's expectation: the predicted
// image is <= 0 on both diagonals.  With an 11 x 11 image a patch of extent 5 touches the border, patch.cpp:145-150
// returns early and the image is the zeros of init, which is what satisfies it -- there and here.
static void warpScenarioOfTheReference()
{
	const int n = 11, c = 5;
	tracker::Mat64 gx(n, n), gy(n, n);
	for (int k = 0; k < n; ++k)
	{
		gx.at<double>(k, c) = 1.0;
		gy.at<double>(c, k) = 1.0;
	}
	tracker::Patch p(tracker::Corner(c, c), c, common::timestamp_t(0));
	p.setFlowDir(static_cast<float>(M_PI / 4));
	p.setWarp(common::Pose2d(M_PI / 4, common::Point2d(0, 0)));  // a pure rotation (the reference builds it with Sophus)
	p.setGrad(gx, gy);
	p.warpImage();
	const tracker::Mat64 predicted = p.getPredictedNabla();
	EXPECT_TRUE(predicted.rows == n && predicted.cols == n);
	for (int d = 1; d < n - 1; ++d)
	{
		EXPECT_LE(predicted.at<double>(d, d), 0);
		EXPECT_LE(predicted.at<double>(d, n - 1 - d), 0);
	}
	EXPECT_TRUE(p.getGradX().rows == n && p.getGradY().at<double>(c, 3) == 1.0);
}

// warpImage() away from the border: the same numbers as the batched ABI call, and the image is not trivial
static void warpImageInteriorTest(ebo_ctx* ctx)
{
	tracker::Mat64 gradX(41, 41), gradY(41, 41);
	for (int k = 0; k <= 40; ++k)
	{
		gradX.at<double>(k, 20) = 1;
		gradY.at<double>(20, k) = 1;
	}
	tracker::Patch patch({20, 20}, 5, common::timestamp_t(0));
	patch.bind(ctx);
	patch.setFlowDir(static_cast<float>(M_PI / 4));
	patch.setWarp(common::Pose2d(0.05, common::Point2d(1.0, -1.0)));  // (a rotation about the image origin by 45 degrees, as in
																	 // the reference's test, would carry the patch off the two lines)
	patch.setGrad(gradX, gradY);
	patch.warpImage();
	const auto image = patch.getPredictedNabla();
	const tracker::Rect2d r = patch.getPatch();
	const double rect[4] = {r.x, r.y, r.width, r.height};
	const size_t off = 0;
	const double flow = patch.getFlowDir();
	std::vector<double> direct(121, 7.0);
	int32_t updated = 0;
	EXPECT_TRUE(ebo_patch_warp_image(ctx, 1, rect, patch.getWarp().data(), &flow, &off, direct.data(), &updated) == EBO_OK);
	EXPECT_TRUE(updated == 1);
	int negative = 0;
	for (int i = 0; i < 11; ++i)
	{
		for (int j = 0; j < 11; ++j)
		{
			EXPECT_TRUE(image.at<double>(i, j) == direct[static_cast<size_t>(i) * 11 + j]);
			EXPECT_LE(image.at<double>(i, j), 0);
			negative += image.at<double>(i, j) < 0;
		}
	}
	EXPECT_TRUE(negative >= 10);
}

// integrateMotionCompensatedEvents against patch.cpp:87-130 written out here
static void motionCompensatedTest(ebo_ctx* ctx)
{
	tracker::Patch patch({20, 20}, 6, common::timestamp_t(1000));
	patch.bind(ctx);
	patch.setNumOfEvents(100);  // clamped to [100, 300]
	std::vector<common::EventSample> evs;
	for (int i = 0; i < 90; ++i)
	{
		evs.push_back(eventAt(14 + (i * 7) % 13, 14 + (i * 5) % 13, 1000 + 40 * i, i % 3 != 0));
	}
	for (const auto& e : evs)
	{
		patch.addEvent(e);
	}
	patch.integrateEvents();  // currentTimestamp_ = mid time of the window
	patch.integrateMotionCompensatedEvents();  // one trajectory point: nothing happens
	const auto untouched = patch.getCompenatedIntegratedNabla();
	double sumAbs = 0;
	for (int i = 0; i < 13 * 13; ++i)
	{
		sumAbs += std::fabs(untouched.ptr()[i]);
	}
	EXPECT_TRUE(sumAbs == 0.0);
	// a second trajectory point 2.5 px to the right, 1 px up, at the mid time
	patch.setWarp(common::Pose2d(0.0, common::Point2d(-2.5, 1.0)));
	patch.updatePatchRect();
	patch.addTrajectoryPosition();
	patch.integrateMotionCompensatedEvents();
	const auto& traj = patch.getTrajectory();
	ASSERT_EQ(traj.size(), 2);
	const auto& pre = traj[0];
	const auto& last = traj[1];
	const tracker::Rect2d rect = patch.getPatch();
	std::vector<double> want(13 * 13, 0.0);
	const double dirx = last.value.x - pre.value.x, diry = last.value.y - pre.value.y;
	const double tDif = static_cast<double>((last.timestamp - pre.timestamp).count());
	const double t = static_cast<double>(patch.getCurrentTimestamp().count());
	const auto half = common::timestamp_t(static_cast<int32_t>((last.timestamp - pre.timestamp).count() * 0.5));
	const bool timeTest = last.timestamp + half >= patch.getCurrentTimestamp() && pre.timestamp < patch.getCurrentTimestamp();
	EXPECT_TRUE(timeTest);
	for (const auto& e : patch.getEvents())
	{
		const double f = (t - static_cast<double>(e.timestamp.count())) / tDif;
		// static_cast<common::Point2i>(Point2d) is OpenCV's saturate_cast: round half to even (:116-117);
		// frameToPatchCoords forms int - double and truncates it into a Point2i (patch.cpp:181-186)
		const int cx = static_cast<int>(std::nearbyint(static_cast<double>(e.value.point.x) + f * dirx));
		const int cy = static_cast<int>(std::nearbyint(static_cast<double>(e.value.point.y) + f * diry));
		if (rect.contains(common::Point2i(cx, cy)))
		{
			const int px = static_cast<int>(cx - rect.x), py = static_cast<int>(cy - rect.y);
			want[static_cast<size_t>(py) * 13 + px] += static_cast<double>(e.value.sign);
		}
	}
	const auto& got = patch.getCompenatedIntegratedNabla();
	double diff = 0, mass = 0;
	for (int i = 0; i < 13 * 13; ++i)
	{
		diff += std::fabs(got.ptr()[i] - want[i]);
		mass += std::fabs(want[i]);
	}
	EXPECT_TRUE(diff == 0.0);
	EXPECT_TRUE(mass >= 20.0);
}

// OptimizerParams::drawCostMap through Optimizer::optimize = the ABI call with the functor's inputs
static void costMapTest()
{
	const int W = 96, H = 72;
	tracker::Mat64 gx(H, W), gy(H, W);
	for (int y = 0; y < H; ++y)
	{
		for (int x = 0; x < W; ++x)
		{
			gx.at<double>(y, x) = std::sin(0.21 * x) * std::cos(0.13 * y);
			gy.at<double>(y, x) = std::cos(0.17 * x + 0.3) * std::sin(0.19 * y);
		}
	}
	tracker::OptimizerParams op;
	op.drawCostMap = true;
	op.costMapWidth = 7;
	op.costMapHeight = 5;
	tracker::Optimizer opt(op, tracker::Size(W, H));
	opt.setGrad(gx, gy);
	tracker::Patch patch({40, 30}, 12, common::timestamp_t(0));
	patch.setNumOfEvents(100);
	patch.setFlowDir(0.7);
	patch.setWarp(common::Pose2d(0.02, common::Point2d(0.4, -0.3)));
	for (int i = 0; i < 120; ++i)
	{
		patch.addEvent(eventAt(29 + (i * 11) % 23, 19 + (i * 7) % 23, 10 * i, (i / 3) % 2 == 0));
	}
	EXPECT_TRUE(patch.getCostMap().rows == 0);
	const tracker::Rect2d before = patch.getPatch();
	opt.optimize(patch);
	EXPECT_FALSE(patch.isLost());
	const auto& cm = patch.getCostMap();
	EXPECT_TRUE(cm.rows == 5 && cm.cols == 7);
	// the functor of the optimisation: rect and integrated nabla from before the solve; pose and flow after it
	const double rect[4] = {before.x, before.y, before.width, before.height};
	const double flow = patch.getFlow();
	std::vector<double> direct(35, -1.0);
	EXPECT_TRUE(ebo_optimizer_cost_map(opt.handle(), 1, rect, patch.getIntegratedNabla().ptr(), 1, patch.getWarp().data(), &flow, 7,
									   5, direct.data()) == EBO_OK);
	double lo = 1e300, hi = 0;
	for (int i = 0; i < 35; ++i)
	{
		EXPECT_TRUE(cm.ptr()[i] == direct[static_cast<size_t>(i)]);
		lo = std::fmin(lo, direct[static_cast<size_t>(i)]);
		hi = std::fmax(hi, direct[static_cast<size_t>(i)]);
	}
	EXPECT_TRUE(lo > 0.0 && hi > lo && std::isfinite(hi));
	// getNormalizedIntegratedNabla: unit L2 norm
	const auto nn = patch.getNormalizedIntegratedNabla();
	double ss = 0;
	for (int i = 0; i < nn.rows * nn.cols; ++i)
	{
		ss += nn.ptr()[i] * nn.ptr()[i];
	}
	EXPECT_TRUE(std::fabs(ss - 1.0) < 1e-12);
}

template <typename F>
static bool throwsWithoutContext(F&& f)
{
	try
	{
		f();
	}
	catch (const std::runtime_error& e)
	{
		return std::string(e.what()).find("no device context") != std::string::npos;
	}
	return false;
}

static void hostOnlyTests()
{
	eventWindowScenario();
	tracker::Patch patch({10, 12}, 5, common::timestamp_t(0));
	patch.addEvent(eventAt(10, 12, 5, true));
	// no context: every device member fails loudly (there is no host implementation)
	EXPECT_TRUE(throwsWithoutContext([&] { patch.integrateEvents(); }));
	EXPECT_TRUE(throwsWithoutContext([&] { patch.warpImage(); }));
	EXPECT_TRUE(throwsWithoutContext([&] { patch.setGrad(tracker::Mat64(4, 4), tracker::Mat64(4, 4)); }));
	patch.setWarp(common::Pose2d(0.0, common::Point2d(1.0, 0.0)));
	patch.updatePatchRect();
	patch.addTrajectoryPosition();
	EXPECT_TRUE(throwsWithoutContext([&] { patch.integrateMotionCompensatedEvents(); }));
	// getInitPatch (patch.cpp:268-273): the rect around the point the patch was created at
	const tracker::Rect2d init = patch.getInitPatch();
	EXPECT_TRUE(init.x == 5.0 && init.y == 7.0 && init.width == 11.0 && init.height == 11.0);
	EXPECT_TRUE(patch.getPatch().x == 4.0);  // moved by the inverse warp
	tracker::Mat64 cm(3, 2);
	cm.at<double>(2, 1) = 4.5;
	patch.setCostMap(cm);
	EXPECT_TRUE(patch.getCostMap().rows == 3 && patch.getCostMap().at<double>(2, 1) == 4.5);
	EXPECT_TRUE(patch.getGradX().rows == 0);
}

int main(int argc, char** argv)
{
	const bool cpuOnly = argc > 1 && std::string(argv[1]) == "--cpu";
	hostOnlyTests();
	if (!cpuOnly)
	{
		// the stand-alone patches of the reference's tests: one context for the process, as large as their images
		tracker::Optimizer small(tracker::OptimizerParams(), tracker::Size(11, 11));
		tracker::Patch::setDefaultContext(small.handle());
		signedCountKnownAnswer();
		warpScenarioOfTheReference();
		tracker::Patch::setDefaultContext(nullptr);
		tracker::Optimizer mid(tracker::OptimizerParams(), tracker::Size(41, 41));
		warpImageInteriorTest(mid.handle());
		motionCompensatedTest(mid.handle());
		costMapTest();
	}
	std::printf("patch_lines_test%s: %s (%d failure%s)\n", cpuOnly ? " --cpu" : "", g_fail ? "FAILED" : "OK", g_fail,
				g_fail == 1 ? "" : "s");
	return g_fail ? 1 : 0;
}
