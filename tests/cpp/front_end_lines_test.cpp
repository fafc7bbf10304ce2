// front_end_lines_test.cpp — the facade's ONE tracker::FeatureDetector under the front end's call pattern.
//
// The unchanged visual_odometry front end holds ONE tracker::FeatureDetector and calls, per event,
// addEvent / updatePatches and, when a window is full, compensateEventsContrast / integrateEvents / clearEvents
// (tools/evaluator/src/evaluator.cpp:32-45); around that getPatches (:23-30), preExit / getArchivedPatches /
// getOptimizedFinalCosts (:15-21), DetectorParams::drawImages / imageSize (:106-109), setParams (:120-123), and a
// keyframe reads toCorner() / getTrackId() of getPatches() (visual_odometry/src/keyframe.cpp:5-14).
// Source compatibility is shown by (i) a member-pointer table that pins each of those members by name, argument and
// result type and (ii) this file's OWN driver (FrontEndDriver) that calls them in the evaluator's order — not by
// carrying the reference's statements.  Compiled with -Wall -Wextra against <feature_tracker/feature_detector.h>.
//
// Checked on the GPU:
//   * the scenarios of the reference's updatePatchTest / associatedPatchesTest (feature_detector_test.cpp:43-125);
//   * the front-end loop over a stream: the ONE FeatureDetector gives, bit for bit, the patches / flows /
//     images of the former pair (a stand-alone tracker::TrackedPatches + a FeatureDetector used for the
//     compensation only), although tracker and compensation now share one device context;
//   * the newImage life cycle through FrontEndHooks (associate, archive lost, optimizer user counts);
//   * newImage without hooks reports EBO_ERR_UNSUPPORTED through the error policy.
// `--cpu` runs the host-only subset (no device): the two scenarios' bookkeeping and the error policy.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <memory>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include <common/data_types.h>
#include <feature_tracker/feature_detector.h>
#include <tools/evaluator.h>

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
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))
#define ASSERT_EQ(a, b) EXPECT_TRUE((a) == (b))

// Two flavours of this program: without OpenCV on the include path the driver below spells the two OpenCV types of the
// front end's interface with the facade's stand-ins; built with -Istubs_opencv (test-only declarations of cv::Size_,
// cv::Point_, cv::Rect_, cv::Mat) the facade's EBO_HAVE_OPENCV branch is live and the driver holds the reference's own
// `cv::Size2i` and receives `cv::Mat const&` -- no rename at all.
#ifdef EBO_HAVE_OPENCV
using FrameSize = cv::Size2i;
using FrameImage = cv::Mat;
static common::Image8 makeImage8(int rows, int cols) { return cv::Mat(rows, cols, CV_8U); }
#else
using FrameSize = tracker::Size;
using FrameImage = tracker::Mat64;
static common::Image8 makeImage8(int rows, int cols) { return common::Image8(rows, cols); }
#endif

// ---- signature conformance with implementation/feature_tracker/include/feature_tracker/feature_detector.h:33-88 ----
// The members the front end (tools/evaluator/src/evaluator.cpp:15-45,106-109,120-123; visual_odometry/src/keyframe.cpp:
// 5-14) calls on its ONE tracker::FeatureDetector, by name, argument and result type.  A member-pointer cast only
// compiles for an exact match; this table replaces carrying the front end's statements as text.
namespace conformance
{
using D = tracker::FeatureDetector;
static_assert(std::is_constructible<D, const tracker::DetectorParams&>::value, "FeatureDetector(params)");
[[maybe_unused]] static const auto kMembers = std::make_tuple(
	static_cast<void (D::*)()>(&D::preExit), static_cast<void (D::*)(const common::ImageSample&)>(&D::newImage),
	static_cast<void (D::*)(const common::ImageSample&)>(&D::extractPatches),
	static_cast<tracker::Corners (D::*)(const common::Image8&)>(&D::detectFeatures),
	static_cast<void (D::*)(const common::EventSample&)>(&D::updatePatches),
	static_cast<void (D::*)(const common::EventSample&)>(&D::addEvent),
	static_cast<void (D::*)(const common::timestamp_t)>(&D::initMotionField),
	static_cast<void (D::*)(const common::timestamp_t)>(&D::interpolateMotionField),
	static_cast<void (D::*)(const std::list<common::EventSample>&)>(&D::compensateEvents),
	static_cast<void (D::*)(const std::list<common::EventSample>&)>(&D::compensateEventsContrast),
	static_cast<void (D::*)()>(&D::clearEvents),
	static_cast<void (D::*)(const std::list<common::EventSample>&)>(&D::integrateEvents),
	static_cast<void (D::*)(tracker::Patches&, const common::timestamp_t&)>(&D::associatePatches),
	static_cast<void (D::*)(tracker::Patch&)>(&D::updateNumOfEvents),
	static_cast<void (D::*)(const tracker::Patches&)>(&D::setPatches),
	static_cast<void (D::*)(const tracker::DetectorParams&)>(&D::setParams),
	static_cast<void (D::*)(tracker::TrackId)>(&D::setTrackId),
	static_cast<tracker::Patches const& (D::*)() const>(&D::getPatches), static_cast<tracker::Patches& (D::*)()>(&D::getPatches),
	static_cast<tracker::Corners const& (D::*)() const>(&D::getFeatures),
	static_cast<tracker::Patches const& (D::*)() const>(&D::getArchivedPatches),
	static_cast<std::list<common::EventSample> const& (D::*)()>(&D::getEvents),
	static_cast<tracker::Mat64 const& (D::*)()>(&D::getCompensatedEventImage),
	static_cast<tracker::Mat64 const& (D::*)()>(&D::getIntegratedEventImage),
	static_cast<common::timestamp_t const& (D::*)()>(&D::getLastCompensation),
	static_cast<std::vector<tracker::OptimizerFinalLoss> (D::*)() const>(&D::getOptimizedFinalCosts));
// the two images bind to the front end's `cv::Mat const&` (Mat64 IS a cv::Mat with OpenCV on the include path)
static_assert(std::is_convertible<tracker::Mat64 const&, FrameImage const&>::value, "Mat64 const& -> cv::Mat const&");
static_assert(std::is_same<decltype(tracker::DetectorParams::imageSize), FrameSize>::value, "DetectorParams::imageSize");
static_assert(std::is_same<decltype(tracker::DetectorParams::drawImages), bool>::value, "DetectorParams::drawImages");
}  // namespace conformance

// ---- what a keyframe takes from the tracker: one landmark per track id ---------------------------------------------
// (visual_odometry/src/keyframe.cpp:5-14 walks getPatches() and keys toCorner() by getTrackId(); own statements)
struct Landmark
{
	double x = 0.0, y = 0.0;
};
static std::unordered_map<tracker::TrackId, Landmark> landmarksOf(const tracker::Patches& tracked)
{
	std::unordered_map<tracker::TrackId, Landmark> table;
	for (auto it = tracked.begin(); it != tracked.end(); ++it)
	{
		const tracker::Corner at = it->toCorner();
		Landmark& slot = table[it->getTrackId()];
		slot.x = at.x;
		slot.y = at.y;
	}
	return table;
}

// ---- the front end's side of the tracker, as this test's own driver ------------------------------------------------
// Holds ONE tracker::FeatureDetector the way tools::Evaluator does and drives it in the evaluator's call order:
// per event addEvent -> updatePatches -> [window full: compensateEventsContrast -> integrateEvents -> clearEvents]
// (evaluator.cpp:32-45: full = time since the last compensation >= a period, or the event count reached a cap); at the
// end preExit and the two result files (:15-21).  Names and control flow are this file's.
struct WindowRule
{
	FrameSize sensor = {240, 180};
	std::string resultsDir = "/tmp";
	bool draw = false;
	uint32_t periodMicros = 300000;  // the evaluator's compensationFrequencyTime default
	uint32_t eventCap = 15000;       // compensationFrequencyEvents
};

class FrontEndDriver
{
   public:
	explicit FrontEndDriver(const WindowRule& rule) : rule_(rule)
	{
		tracker::DetectorParams dp;
		dp.imageSize = rule_.sensor;
		dp.drawImages = rule_.draw;
		detector_ = std::make_unique<tracker::FeatureDetector>(dp);
	}
	~FrontEndDriver()
	{
		detector_->preExit();
		tools::saveFeaturesTrajectory(detector_->getArchivedPatches(), rule_.resultsDir + "/trajectory.txt");
		std::ofstream out(rule_.resultsDir + "/final_cost.txt");
		out << std::fixed << std::setprecision(8);
		for (const tracker::OptimizerFinalLoss& c : detector_->getOptimizedFinalCosts())
		{
			out << c.trackId << ' ' << c.lossValue << ' ' << c.timeStampMicrosecond << '\n';  // the evaluator's line format (:208-213)
		}
	}
	// one event of the stream; returns whether it closed a window
	bool feed(const common::EventSample& e)
	{
		tracker::FeatureDetector& d = *detector_;
		d.addEvent(e);
		d.updatePatches(e);
		const bool periodOver = (e.timestamp - d.getLastCompensation()).count() >= rule_.periodMicros;
		const bool capReached = d.getEvents().size() >= rule_.eventCap;
		if (!periodOver && !capReached)
		{
			return false;
		}
		const std::list<common::EventSample>& window = d.getEvents();
		d.compensateEventsContrast(window);
		d.integrateEvents(window);
		d.clearEvents();
		++windowsClosed;
		return true;
	}
	void reconfigure(const tracker::DetectorParams& dp) { detector_->setParams(dp); }
	tracker::Patches const& patches() const { return detector_->getPatches(); }
	FrameImage const& warpedImage() { return detector_->getCompensatedEventImage(); }
	FrameImage const& plainImage() { return detector_->getIntegratedEventImage(); }
	tracker::FeatureDetector& detector() { return *detector_; }
	int windowsClosed = 0;

   private:
	WindowRule rule_;
	std::unique_ptr<tracker::FeatureDetector> detector_;
};

// ---- the scenarios of the reference's two detector tests (feature_detector_test.cpp:43-125), own statements --------
// Routing: three 11-extent patches at (0,0), (5,5), (20,20) on a detector with patchExtent 5; five events in [0,30)^2;
// every patch of the detector must hold exactly the events a stand-alone copy of the patch accepted through isInPatch,
// in the same order.  `policy` selects the error policy (without a device the default one throws from the constructor).
static void routingScenario(tracker::DetectorParams::ErrorPolicy policy, bool setPolicy)
{
	tracker::DetectorParams dp;
	dp.patchExtent = 5;
	if (setPolicy)
	{
		dp.errorPolicy = policy;
	}
	tracker::FeatureDetector d(dp);
	const common::timestamp_t t0(0);
	const int centres[3] = {0, 5, 20};
	tracker::Patches mirror;
	for (int c : centres)
	{
		mirror.emplace_back(tracker::Corner(c, c), 11, t0);
	}
	d.setPatches(mirror);
	uint32_t lcg = 90210u;
	for (int k = 0; k < 5; ++k)
	{
		lcg = lcg * 1664525u + 1013904223u;
		common::EventSample e;
		e.timestamp = common::timestamp_t(k);
		e.value.point = {static_cast<int>((lcg >> 8) % 30), static_cast<int>((lcg >> 16) % 30)};
		e.value.sign = (lcg >> 24) & 1u ? common::EventPolarity::POSITIVE : common::EventPolarity::NEGATIVE;
		d.updatePatches(e);
		for (tracker::Patch& m : mirror)
		{
			if (m.isInPatch(e.value.point))
			{
				m.addEvent(e);
			}
		}
	}
	const tracker::Patches routed = d.getPatches();
	EXPECT_TRUE(routed.size() == mirror.size());
	auto r = routed.begin();
	size_t held = 0;
	for (auto m = mirror.begin(); m != mirror.end() && r != routed.end(); ++m, ++r)
	{
		const common::EventSequence& want = m->getEvents();
		const common::EventSequence& got = r->getEvents();
		EXPECT_TRUE(got.size() == want.size());
		for (size_t k = 0; k < want.size() && k < got.size(); ++k)
		{
			EXPECT_TRUE(got[k].timestamp == want[k].timestamp);
		}
		held += got.size();
	}
	EXPECT_TRUE(held >= 5);  // all five land in the 23 x 23 patch around (5,5) at least
}

// Association: the same three patches with track ids 0..2; three new patches at (3,0), (0,1), (18,18) with
// associationDistance 5: two are within reach of an old patch, one is not -> four patches afterwards (the reference's
// expectation); beyond it: the unmatched one takes the next track id, a matched one gains a trajectory point.
static void associationScenario(tracker::DetectorParams::ErrorPolicy policy, bool setPolicy)
{
	tracker::DetectorParams dp;
	dp.patchExtent = 5;
	dp.associationDistance = 5;
	if (setPolicy)
	{
		dp.errorPolicy = policy;
	}
	tracker::FeatureDetector d(dp);
	const common::timestamp_t t0(0);
	tracker::Patches old;
	tracker::TrackId next = 0;
	for (int c : {0, 5, 20})
	{
		old.emplace_back(tracker::Corner(c, c), 11, t0);
		old.back().setTrackId(next++);
	}
	d.setPatches(old);
	d.setTrackId(next);
	tracker::Patches fresh;
	fresh.emplace_back(tracker::Corner(3, 0), 11, t0);
	fresh.emplace_back(tracker::Corner(0, 1), 11, t0);
	fresh.emplace_back(tracker::Corner(18, 18), 11, t0);
	d.associatePatches(fresh, t0);
	const tracker::Patches after = d.getPatches();
	EXPECT_TRUE(after.size() == 4u);
	EXPECT_TRUE(after.back().getTrackId() == 3);
	EXPECT_TRUE(after.front().getTrajectory().size() == 2u);
}

template <class L>
static auto nth(L& l, size_t i) -> decltype(*l.begin())
{
	return *std::next(l.begin(), static_cast<std::ptrdiff_t>(i));
}

static const double* pixels(const tracker::Mat64& m) { return m.ptr(); }
#ifdef EBO_HAVE_OPENCV
static const double* pixels(const cv::Mat& m) { return m.ptr<double>(0); }
#endif
template <class A, class B>
static bool sameImage(const A& a, const B& b)
{
	return a.rows == b.rows && a.cols == b.cols &&
		   std::memcmp(pixels(a), pixels(b), sizeof(double) * static_cast<size_t>(a.rows) * a.cols) == 0;
}

static bool samePatch(const tracker::Patch& p, const tracker::Patch& q)
{
	bool same = p.isLost() == q.isLost() && p.isInit() == q.isInit() && p.getTrackId() == q.getTrackId() &&
				p.getNumOfEvents() == q.getNumOfEvents() && p.getEvents().size() == q.getEvents().size() &&
				p.getFinalCosts().size() == q.getFinalCosts().size() && p.getTrajectory().size() == q.getTrajectory().size() &&
				p.getPatch().x == q.getPatch().x && p.getPatch().y == q.getPatch().y && p.getFlowDir() == q.getFlowDir();
	for (int k = 0; same && k < 4; ++k)
	{
		same = p.getWarp().data()[k] == q.getWarp().data()[k];
	}
	for (size_t k = 0; same && k < p.getFinalCosts().size(); ++k)
	{
		same = p.getFinalCosts()[k] == q.getFinalCosts()[k];
	}
	for (size_t k = 0; same && k < p.getEvents().size(); ++k)
	{
		same = p.getEvents()[k].timestamp == q.getEvents()[k].timestamp;
	}
	for (size_t k = 0; same && k < p.getTrajectory().size(); ++k)
	{
		same = p.getTrajectory()[k].value.x == q.getTrajectory()[k].value.x &&
			   p.getTrajectory()[k].value.y == q.getTrajectory()[k].value.y &&
			   p.getTrajectory()[k].timestamp == q.getTrajectory()[k].timestamp;
	}
	same = same && sameImage(p.getIntegratedNabla(), q.getIntegratedNabla()) &&
		   sameImage(p.getCompenatedIntegratedNabla(), q.getCompenatedIntegratedNabla()) &&
		   sameImage(p.getPredictedNabla(), q.getPredictedNabla());
	return same;
}

// gradient images of five Gaussian blobs (what a log-intensity frame with five dots would give)
static void blobGradients(int W, int H, double shift, tracker::Mat64& gx, tracker::Mat64& gy)
{
	gx = tracker::Mat64(H, W);
	gy = tracker::Mat64(H, W);
	for (int y = 0; y < H; ++y)
	{
		for (int x = 0; x < W; ++x)
		{
			double vx = 0, vy = 0;
			for (int k = 0; k < 5; ++k)
			{
				const double cx = 40 + 38 * k + shift, cy = 40 + 25 * k, sg = 6 + k;
				const double e = std::exp(-((x - cx) * (x - cx) + (y - cy) * (y - cy)) / (2 * sg * sg));
				vx += -(x - cx) / (sg * sg) * e;
				vy += -(y - cy) / (sg * sg) * e;
			}
			gx.at<double>(y, x) = 4.0 * vx + 0.25 + 0.001 * x;
			gy.at<double>(y, x) = 4.0 * vy - 0.2 + 0.0015 * y;
		}
	}
}

static std::vector<common::EventSample> makeStream(int n, int W, int H)
{
	std::vector<common::EventSample> stream;
	uint64_t st = 424242;
	auto rnd = [&]() {
		st = st * 6364136223846793005ull + 1442695040888963407ull;
		return static_cast<uint32_t>(st >> 33);
	};
	for (int i = 0; i < n; ++i)
	{
		common::EventSample e;
		const int k = static_cast<int>(rnd() % 8);
		const double drift = 2e-4 * i;
		int x, y;
		if (k < 4)
		{
			x = static_cast<int>(42.0 + 38 * k + drift + static_cast<int>(rnd() % 21) - 10);
			y = static_cast<int>(38.0 + 25 * k + static_cast<int>(rnd() % 21) - 10);
		}
		else
		{
			// moving edges everywhere, so that the compensation grid has patches above compensateMinNumEvents
			const int col = 3 * static_cast<int>(rnd() % 4), row = 3 * static_cast<int>(rnd() % 3);
			x = static_cast<int>(col * 20 + 4 + 1e-3 * i * 0.3 + rnd() % 3);
			y = static_cast<int>(row * 20 + rnd() % 20);
		}
		x = std::min(std::max(x, 0), W - 1);
		y = std::min(std::max(y, 0), H - 1);
		e.value.point = {x, y};
		e.value.sign = (rnd() & 1) ? common::POSITIVE : common::NEGATIVE;
		e.timestamp = common::timestamp_t(2000 + 11 * i);
		stream.push_back(e);
	}
	return stream;
}

template <class Holder>
static std::shared_ptr<tracker::Optimizer> installTracked(Holder& h, int W, int H)
{
	tracker::Mat64 gx, gy, gxBig, gyBig;
	blobGradients(W, H, 0.0, gxBig, gyBig);
	gx = tracker::Mat64(H, W);
	gy = tracker::Mat64(H, W);
	for (int i = 0; i < W * H; ++i)
	{
		gx.ptr()[i] = (gxBig.ptr()[i] - 0.25 - 0.001 * (i % W)) / 4.0;
		gy.ptr()[i] = (gyBig.ptr()[i] + 0.2 - 0.0015 * (i / W)) / 4.0;
	}
	tracker::OptimizerParams op;
	auto opt = std::make_shared<tracker::Optimizer>(op, tracker::Size(W, H));
	opt->setGrad(gx, gy);
	h.setOptimizer(common::timestamp_t(1000), opt);
	h.setGradients(gxBig, gyBig);
	tracker::Patches patches;
	for (int k = 0; k < 4; ++k)
	{
		tracker::Patch p(tracker::Corner(42.0 + 38 * k, 38.0 + 25 * k), 12, common::timestamp_t(1000));
		p.setTrackId(k);
		p.setFlowDir(0.4 + 0.3 * k);
		patches.push_back(p);
	}
	patches.push_back(tracker::Patch(tracker::Corner(120.0, 90.0), 12, common::timestamp_t(1000)));  // never initialised
	tracker::Patch edge(tracker::Corner(4.0, 100.0), 12, common::timestamp_t(1000));                   // lost after its first optimisation
	edge.setFlowDir(0.2);
	patches.push_back(edge);
	h.setPatches(patches);
	return opt;
}

static int hostOnly()
{
	routingScenario(tracker::DetectorParams::ERRORS_STATUS, true);
	associationScenario(tracker::DetectorParams::ERRORS_STATUS, true);
	{
		tracker::DetectorParams params;
		params.errorPolicy = tracker::DetectorParams::ERRORS_STATUS;
		tracker::FeatureDetector detector(params);
		common::ImageSample image(makeImage8(180, 240), common::timestamp_t(5));
		detector.newImage(image);
		EXPECT_TRUE(detector.status() == EBO_ERR_UNSUPPORTED);
		EXPECT_TRUE(detector.detectFeatures(image.value).empty() && detector.status() == EBO_ERR_UNSUPPORTED);
		detector.preExit();
		EXPECT_TRUE(detector.getArchivedPatches().empty() && detector.getOptimizedFinalCosts().empty());
		// every reference field is there, with the reference's default (feature_detector.h:10-31)
		EXPECT_TRUE(params.qualityLevel == 0.01 && params.minDistance == 10 && params.associationDistance == 5 &&
					params.patchExtent == 12 && params.blockSize == 3 && params.imageSize.width == 240 &&
					params.imageSize.height == 180 && !params.drawImages && params.optimizerParams.maxNumIterations == 10 &&
					params.initNumEvents == 75 && params.maxNumEventsToStore == 15000 && params.useAverageFlow &&
					params.optimizeFlowTV && !params.useL1 && params.patchCompensateSize.width == 20 &&
					params.compensateTVweight == 1e3 && params.compensateTVHuberLoss == 10 && params.compensateScale == 1e-3 &&
					params.compensateMinNumEvents == 100 && params.maxPatches == 100);
	}
	std::printf(g_fail ? "front_end_lines_test (host only): %d FAILED\n" : "front_end_lines_test (host only): all passed\n", g_fail);
	return g_fail ? 1 : 0;
}

int main(int argc, char** argv)
{
	if (argc > 1 && std::strcmp(argv[1], "--cpu") == 0)
	{
		return hostOnly();
	}
	const int W = 240, H = 180;

	// ---- the scenarios of the reference's two detector tests, default error policy --------------------
	routingScenario(tracker::DetectorParams::ERRORS_STATUS, false);
	associationScenario(tracker::DetectorParams::ERRORS_STATUS, false);

	// ---- the evaluator loop: ONE FeatureDetector against the former pair ----------------------------
	{
		WindowRule ep;
		ep.resultsDir = "/tmp";
		ep.eventCap = 6000;
		ep.periodMicros = 4000000000u;  // windows by event count only
		FrontEndDriver evaluator(ep);
		auto optOne = installTracked(evaluator.detector(), W, H);

		tracker::TrackedPatches pairTracked(tracker::Size(W, H));
		auto optPair = installTracked(pairTracked, W, H);
		tracker::DetectorParams dp;
		tracker::FeatureDetector pairDetector(dp);

		const std::vector<common::EventSample> stream = makeStream(15000, W, H);
		int windows = 0, sameWindows = 0;
		for (const auto& sample : stream)
		{
			evaluator.feed(sample);
			// the pair, in the same call order
			pairDetector.addEvent(sample);
			pairTracked.updatePatches(sample);
			if (pairDetector.getEvents().size() >= ep.eventCap)
			{
				pairDetector.compensateEventsContrast(pairDetector.getEvents());
				pairDetector.integrateEvents(pairDetector.getEvents());
				pairDetector.clearEvents();
				++windows;
				const bool same = sameImage(evaluator.warpedImage(), pairDetector.getCompensatedEventImage()) &&
								  sameImage(evaluator.plainImage(), pairDetector.getIntegratedEventImage()) &&
								  evaluator.detector().getPatchFlows() == pairDetector.getPatchFlows() &&
								  evaluator.detector().getLastCompensation() == pairDetector.getLastCompensation();
				sameWindows += same ? 1 : 0;
				double maxFlow = 0;
				for (double f : evaluator.detector().getPatchFlows())
				{
					maxFlow = std::fmax(maxFlow, std::fabs(f));
				}
				std::printf("window %d: %d iterations, max |flow| %.4f, same as the pair: %d\n", windows,
							evaluator.detector().getLastSummary().iterations, maxFlow, int(same));
				EXPECT_TRUE(maxFlow > 1e-3);  // the solve moved
			}
		}
		EXPECT_TRUE(windows == 2 && evaluator.windowsClosed == windows && sameWindows == windows);
		const tracker::Patches& one = evaluator.patches();
		EXPECT_TRUE(one.size() == 6 && pairTracked.getPatches().size() == 6);
		size_t optimisations = 0;
		for (size_t i = 0; i < one.size(); ++i)
		{
			EXPECT_TRUE(samePatch(nth(one, i), nth(pairTracked.getPatches(), i)));
			optimisations += nth(one, i).getFinalCosts().size();
		}
		EXPECT_TRUE(optimisations >= 8 && optOne->getFinalCosts().size() == optimisations &&
					optPair->getFinalCosts().size() == optimisations);
		EXPECT_TRUE(nth(one, 4).getFinalCosts().empty() && !nth(one, 4).isInit());  // nobody initialised it: it only collects events
		const auto costs = evaluator.detector().getOptimizedFinalCosts();
		EXPECT_TRUE(costs.size() == optimisations);
		std::printf("evaluator loop: %zu events, %d windows, %zu optimisations, one detector == the pair\n", stream.size(),
					windows, optimisations);

		// the chunked call of the same class: same patches as the per-event loop
		FrontEndDriver chunked(ep);
		installTracked(chunked.detector(), W, H);
		chunked.detector().updatePatches(stream);
		for (size_t i = 0; i < one.size(); ++i)
		{
			EXPECT_TRUE(samePatch(nth(one, i), nth(chunked.patches(), i)));
		}

		// DetectorParams::eventBatch: the per-event call keeps the events and routes them in chunks (the loop above ran
		// that way: the evaluator's DetectorParams are the defaults); with eventBatch = 1 every event is processed at
		// once, as the reference does -- the same patches either way, and the time it takes
		{
			tracker::DetectorParams immediate, deferred;
			immediate.eventBatch = 1;
			tracker::FeatureDetector a(immediate), b(deferred);
			installTracked(a, W, H);
			installTracked(b, W, H);
			const auto t0 = std::chrono::steady_clock::now();
			for (size_t i = 0; i < 5000; ++i)
			{
				a.updatePatches(stream[i]);
			}
			const auto t1 = std::chrono::steady_clock::now();
			for (size_t i = 0; i < 5000; ++i)
			{
				b.updatePatches(stream[i]);
			}
			const tracker::Patches& pb = b.getPatches();  // brings them up to the last event
			const auto t2 = std::chrono::steady_clock::now();
			for (size_t i = 0; i < pb.size(); ++i)
			{
				EXPECT_TRUE(samePatch(nth(a.getPatches(), i), nth(pb, i)));
			}
			std::printf("updatePatches(event) x 5000, 6 patches: every event at once %.1f ms, kept and routed in chunks %.1f ms\n",
						std::chrono::duration<double, std::milli>(t1 - t0).count(),
						std::chrono::duration<double, std::milli>(t2 - t1).count());
		}

		// what a keyframe takes from the detector's patches (evaluator.cpp:85-87 -> keyframe.cpp:5-14)
		const auto landmarks = landmarksOf(evaluator.detector().getPatches());
		EXPECT_TRUE(landmarks.size() == 5);  // track ids 0..3 and the two -1 share one key
		const Landmark& lm = landmarks.at(2);
		EXPECT_TRUE(lm.x == nth(one, 2).toCorner().x && lm.y == nth(one, 2).toCorner().y);

		// setParams: params_ = params, optimizers take optimizerParams, reset() (:733-741)
		tracker::DetectorParams np;
		np.optimizerParams.maxNumIterations = 3;
		np.maxNumEventsToStore = 5000;
		evaluator.feed(stream[0]);
		evaluator.reconfigure(np);
		EXPECT_TRUE(evaluator.detector().getEvents().empty() && evaluator.detector().getLastCompensation().count() == 0);
		EXPECT_TRUE(optOne->getParams()->maxNumIterations == 3);
		EXPECT_TRUE(evaluator.warpedImage().template at<double>(90, 120) == 0.0);
		for (int i = 0; i < 5200; ++i)
		{
			evaluator.detector().addEvent(stream[i]);
		}
		EXPECT_TRUE(evaluator.detector().getEvents().size() == 5000);
		// ~FrontEndDriver: preExit, trajectory.txt, final_cost.txt
	}
	{
		std::ifstream costs("/tmp/final_cost.txt"), traj("/tmp/trajectory.txt");
		std::string line;
		size_t nCosts = 0, nTraj = 0;
		while (std::getline(costs, line))
		{
			++nCosts;
		}
		while (std::getline(traj, line))
		{
			++nTraj;
		}
		EXPECT_TRUE(nCosts >= 8 && nTraj >= 6 + 8);  // every archived patch's trajectory: its start + one point per optimisation
	}

	// ---- newImage through the front-end hooks: the reference's life cycle (:493-541) ------------------
	{
		tracker::DetectorParams dp;
		dp.drawImages = true;
		dp.maxPatches = 6;
		tracker::FeatureDetector detector(dp);
		double shift = 0.0;
		tracker::FrontEndHooks hooks;
		int detectCalls = 0;
		hooks.detectFeatures = [&](const common::Image8& image) {
			++detectCalls;
			EXPECT_TRUE(image.rows == H && image.cols == W);
			tracker::Corners c;
			for (int k = 0; k < 4; ++k)
			{
				c.emplace_back(42.0 + 38 * k + shift, 38.0 + 25 * k);
			}
			c.emplace_back(7.0, 100.0);  // LK loses it on the second image (status 0)
			if (shift > 0)
			{
				c.emplace_back(200.0, 30.0);  // appears on the second image
			}
			return c;
		};
		hooks.gradients = [&](const common::Image8&, tracker::Mat64& gx, tracker::Mat64& gy) { blobGradients(W, H, shift, gx, gy); };
		hooks.flow = [&](float x, float y, float& nx, float& ny) {
			nx = x - 3.0f;
			ny = y + 1.0f;
			return !(x < 10.f);
		};
		detector.setFrontEndHooks(hooks);
		common::ImageSample first(makeImage8(H, W), common::timestamp_t(1000));
		detector.newImage(first);
		EXPECT_TRUE(detector.getFeatures().size() == 5 && detector.getPatches().size() == 5);
		EXPECT_TRUE(detector.tracked().optimizers().size() == 1);
		for (const auto& p : detector.getPatches())
		{
			EXPECT_TRUE(!p.isInit() && !p.isLost() && p.getTrajectory().size() == 1);  // one image: no flow yet
		}
		shift = 1.0;
		common::ImageSample second(makeImage8(H, W), common::timestamp_t(41000));
		detector.newImage(second);
		// the four blobs are re-detected 1 px away (associated, no new track); (7,100) has no flow: lost and
		// archived, its frame's optimizer loses a user; (200,30) is new: track id 5 on the second frame's optimizer
		EXPECT_TRUE(detector.getPatches().size() == 5 && detector.getArchivedPatches().size() == 1);
		EXPECT_TRUE(detector.getArchivedPatches().front().getTrackId() == 4);
		EXPECT_TRUE(detector.tracked().optimizers().size() == 2);
		size_t k = 0;
		for (const auto& p : detector.getPatches())
		{
			if (k < 4)
			{
				EXPECT_TRUE(p.isInit() && p.getTrackId() == static_cast<tracker::TrackId>(k) && p.getTrajectory().size() == 2);
				EXPECT_TRUE(p.getFlowDir() == std::atan2(1.0, -3.0) && p.getWarp().data()[2] == 3.0 && p.getWarp().data()[3] == -1.0);
				EXPECT_TRUE(p.getNumOfEvents() >= 100 && p.getNumOfEvents() <= 300);
				// drawImages: warpImage ran against the patch's own frame (a rect clear of the border)
				double s = 0;
				for (int i = 0; i < 25 * 25; ++i)
				{
					s += std::fabs(p.getPredictedNabla().ptr()[i]);
				}
				EXPECT_TRUE(s > 0);
			}
			else
			{
				EXPECT_TRUE(p.getTrackId() == 5 && p.getInitTime().count() == 41000 && p.getTrajectory().size() == 1 && p.isInit());
			}
			++k;
		}
		// events now move the four initialised patches
		for (const auto& e : makeStream(6000, W, H))
		{
			common::EventSample s = e;
			s.timestamp += common::timestamp_t(50000);
			detector.addEvent(s);
			detector.updatePatches(s);
		}
		EXPECT_TRUE(detector.getOptimizedFinalCosts().size() >= 4);
		EXPECT_TRUE(detector.getPatches().front().getTrajectory().size() > 2);
		detector.preExit();
		EXPECT_TRUE(detector.getArchivedPatches().size() == 6);
		EXPECT_TRUE(detectCalls == 2);
	}

	// ---- without hooks newImage is reported, never silently skipped --------------------------------------
	{
		tracker::DetectorParams dp;
		tracker::FeatureDetector detector(dp);
		common::ImageSample image(makeImage8(H, W), common::timestamp_t(5));
		bool threw = false;
		try
		{
			detector.newImage(image);
		}
		catch (const std::runtime_error& e)
		{
			threw = std::string(e.what()).find("FrontEndHooks") != std::string::npos;
		}
		EXPECT_TRUE(threw && detector.status() == EBO_ERR_UNSUPPORTED);
	}

	std::printf(g_fail ? "front_end_lines_test: %d FAILED\n" : "front_end_lines_test: all passed\n", g_fail);
	return g_fail ? 1 : 0;
}
