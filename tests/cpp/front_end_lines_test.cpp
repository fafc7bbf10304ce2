// front_end_lines_test.cpp — the reference front end's OWN statements against the facade class.
//
// The unchanged visual_odometry front end holds ONE tracker::FeatureDetector and calls, per event,
//     tracker_->addEvent(sample); tracker_->updatePatches(sample); ... compensateEventsContrast /
//     integrateEvents / clearEvents                       (tools/evaluator/src/evaluator.cpp:32-45)
// and, around that, getPatches (:23-30), preExit / getArchivedPatches / getOptimizedFinalCosts
// (:15-21), DetectorParams::drawImages / imageSize (:106-109), setParams (:120-123) and
// visual_odometry::Keyframe(tracker_->getPatches(), ts) (visual_odometry/src/keyframe.cpp:5-14).
// The blocks marked "verbatim" below are those statements, unchanged, compiled with -Wall -Wextra
// against <feature_tracker/feature_detector.h> of the facade.  What the front end has besides the
// tracker (spdlog loggers, VisualOdometryFrontEnd, Eigen) is not on the path: tools::Evaluator here is
// a shell with the reference's member names, Eigen::Vector2d a two-double stand-in (test-only).
//
// Checked on the GPU:
//   * the reference's updatePatchTest / associatedPatchesTest (feature_detector_test.cpp:43-125), verbatim;
//   * the evaluator loop over a stream: the ONE FeatureDetector gives, bit for bit, the patches / flows /
//     images of the former pair (a stand-alone tracker::TrackedPatches + a FeatureDetector used for the
//     compensation only), although tracker and compensation now share one device context;
//   * the newImage life cycle through FrontEndHooks (associate, archive lost, optimizer user counts);
//   * newImage without hooks reports EBO_ERR_UNSUPPORTED through the error policy.
// `--cpu` runs the host-only subset (no device): the two reference tests' bookkeeping and the error policy.
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
#include <unordered_map>
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

// ---- stand-ins for what the front end has besides the tracker (test-only) --------------------------
namespace Eigen
{
struct Vector2d
{
	double v[2] = {0.0, 0.0};
	Vector2d() = default;
	Vector2d(double x, double y) : v{x, y} {}
	double x() const { return v[0]; }
	double y() const { return v[1]; }
};
}  // namespace Eigen

namespace visual_odometry
{
using Landmarks = std::unordered_map<tracker::TrackId, Eigen::Vector2d>;  // keyframe.h:10

class Keyframe  // keyframe.h:24-43 without the pose
{
   public:
	Keyframe() {}
	Keyframe(const tracker::Patches& patches, const common::timestamp_t& timestamp);
	const Landmarks& getLandmarks() const { return landmarks_; }

   public:
	common::timestamp_t timestamp;

   private:
	Landmarks landmarks_;
};

// ---- visual_odometry/src/keyframe.cpp:5-14, verbatim ---------------------------------------------
Keyframe::Keyframe(const tracker::Patches& patches,
				   const common::timestamp_t& timestamp)
	: timestamp(timestamp)
{
	for (const auto& patch : patches)
	{
		const auto corner = patch.toCorner();
		landmarks_[patch.getTrackId()] = Eigen::Vector2d(corner.x, corner.y);
	}
}
}  // namespace visual_odometry

// Two flavours of this program: without OpenCV on the include path the shell below spells the two OpenCV types of the
// evaluator's interface with the facade's stand-ins; built with -Istubs_opencv (test-only declarations of cv::Size_,
// cv::Point_, cv::Rect_, cv::Mat) the facade's EBO_HAVE_OPENCV branch is live and the shell keeps the reference's own
// `cv::Size2i imageSize` and `cv::Mat const& getCompensatedEventImage()` -- no rename at all.
#ifdef EBO_HAVE_OPENCV
using EvaluatorSize = cv::Size2i;
using EvaluatorImage = cv::Mat;
static common::Image8 makeImage8(int rows, int cols) { return cv::Mat(rows, cols, CV_8U); }
#else
using EvaluatorSize = tracker::Size;
using EvaluatorImage = tracker::Mat64;
static common::Image8 makeImage8(int rows, int cols) { return common::Image8(rows, cols); }
#endif

namespace tools
{
struct EvaluatorParams  // tools/evaluator/include/evaluator/evaluator.h:14-26
{
	EvaluatorSize imageSize = {240, 180};
	std::string outputDir = "/tmp";
	bool drawImages = false;
	// compensate whole image each k microseconds
	uint32_t compensationFrequencyTime = 300000;
	uint32_t compensationFrequencyEvents = 15000;
	bool trackerExperiment = false;
	bool visOdometryExperiment = false;
};

class Evaluator  // evaluator.h:28-88, the members the tracker side touches
{
   public:
	Evaluator(const EvaluatorParams& params);
	~Evaluator();
	void eventCallback(const common::EventSample& sample);
	void reset();
	void setTrackerParams(const tracker::DetectorParams& params);
	void saveFeaturesTrajectory(const tracker::Patches& patches);
	void saveFinalCosts(const std::vector<tracker::OptimizerFinalLoss>& vectorFinalCosts);
	tracker::Patches const& getPatches() const;
	EvaluatorImage const& getCompensatedEventImage();
	EvaluatorImage const& getIntegratedEventImage();
	tracker::FeatureDetector& tracker() { return *tracker_; }  // test access
	int windows = 0;                                           // test instrumentation

   private:
	EvaluatorParams params_;
	std::unique_ptr<tracker::FeatureDetector> tracker_;
	tracker::Patches patches_;
};

Evaluator::Evaluator(const EvaluatorParams& params) : params_(params)
{
	reset();
}

Evaluator::~Evaluator()
{
	// ---- evaluator.cpp:17,18,20, verbatim (:19 savePoses belongs to the VO front end) ------------
	tracker_->preExit();
	saveFeaturesTrajectory(tracker_->getArchivedPatches());
	saveFinalCosts(tracker_->getOptimizedFinalCosts());
}

// ---- evaluator.cpp:23-30, verbatim ---------------------------------------------------------------
tracker::Patches const& Evaluator::getPatches() const
{
	if (params_.visOdometryExperiment)
	{
		return patches_;
	}
	return tracker_->getPatches();
}

// ---- evaluator.cpp:32-45, verbatim (plus the window counter) -------------------------------------
void Evaluator::eventCallback(const common::EventSample& sample)
{
	tracker_->addEvent(sample);
	tracker_->updatePatches(sample);
	if ((sample.timestamp - tracker_->getLastCompensation()).count() >=
			params_.compensationFrequencyTime or
		tracker_->getEvents().size() >= params_.compensationFrequencyEvents)
	{
		//		tracker_->compensateEvents(tracker_->getEvents());
		tracker_->compensateEventsContrast(tracker_->getEvents());
		tracker_->integrateEvents(tracker_->getEvents());
		tracker_->clearEvents();
		++windows;
	}
}

void Evaluator::reset()
{
	// ---- evaluator.cpp:106-109, verbatim ---------------------------------------------------------
	tracker::DetectorParams params;
	params.drawImages = params_.drawImages;
	params.imageSize = params_.imageSize;
	tracker_.reset(new tracker::FeatureDetector(params));
}

// ---- evaluator.cpp:120-123, verbatim -------------------------------------------------------------
void Evaluator::setTrackerParams(const tracker::DetectorParams& params)
{
	tracker_->setParams(params);
}

void Evaluator::saveFeaturesTrajectory(const tracker::Patches& patches)
{
	const std::string outputFilename = params_.outputDir + "/trajectory.txt";  // evaluator.cpp:129
	tools::saveFeaturesTrajectory(patches, outputFilename);
}

void Evaluator::saveFinalCosts(const std::vector<tracker::OptimizerFinalLoss>& vectorFinalCosts)
{
	const std::string outputFilename = params_.outputDir + "/final_cost.txt";
	std::ofstream costFile;
	costFile.open(outputFilename);
	// ---- evaluator.cpp:208-213, verbatim ---------------------------------------------------------
	for (const auto& v : vectorFinalCosts)
	{
		costFile << v.trackId << " " << std::fixed << std::setprecision(8)
				 << v.lossValue << " " << v.timeStampMicrosecond << std::endl;
	}

	costFile.close();
}

// ---- evaluator.cpp:219-227, verbatim (EvaluatorImage = cv::Mat with OpenCV) -------------------------
EvaluatorImage const& Evaluator::getCompensatedEventImage()
{
	return tracker_->getCompensatedEventImage();
}

EvaluatorImage const& Evaluator::getIntegratedEventImage()
{
	return tracker_->getIntegratedEventImage();
}
}  // namespace tools

// ---- the reference's own two detector tests (feature_detector_test.cpp:43-125) -----------------------
// POLICY is empty for the verbatim run; the host-only run adds one statement that selects ERRORS_STATUS
// (without a device the default policy throws from the constructor, as it should).
#define REFERENCE_UPDATE_PATCH_TEST(POLICY)                                                           \
	{                                                                                                 \
		tracker::DetectorParams params;                                                               \
		params.patchExtent = 5;                                                                       \
		POLICY;                                                                                       \
		tracker::FeatureDetector detector(params);                                                    \
		const common::timestamp_t timestamp(0);                                                       \
                                                                                                      \
		tracker::Patches patches = {tracker::Patch({0, 0}, 11, timestamp),                            \
									tracker::Patch({5, 5}, 11, timestamp),                            \
									tracker::Patch({20, 20}, 11, timestamp)};                         \
                                                                                                      \
		detector.setPatches(patches);                                                                 \
                                                                                                      \
		common::EventSequence events;                                                                 \
                                                                                                      \
		for (size_t i = 0; i < 5; ++i)                                                                \
		{                                                                                             \
			common::EventSample event;                                                                \
			event.timestamp = common::timestamp_t(i);                                                 \
			event.value.point = {std::rand() % 30, std::rand() % 30};                                 \
			event.value.sign = std::rand() % 2 == 1                                                   \
								   ? common::EventPolarity::POSITIVE                                  \
								   : common::EventPolarity::NEGATIVE;                                 \
			detector.updatePatches(event);                                                            \
                                                                                                      \
			for (auto& patch : patches)                                                               \
			{                                                                                         \
				if (patch.isInPatch(event.value.point))                                               \
				{                                                                                     \
					patch.addEvent(event);                                                            \
				}                                                                                     \
			}                                                                                         \
		}                                                                                             \
                                                                                                      \
		const auto detectorPatches = detector.getPatches();                                           \
                                                                                                      \
		ASSERT_EQ(detectorPatches.size(), patches.size());                                            \
                                                                                                      \
		auto detectorPatchesIt = detectorPatches.begin();                                             \
		for (auto patchIt = patches.begin(); patchIt != patches.end();                                \
			 ++patchIt, ++detectorPatchesIt)                                                          \
		{                                                                                             \
			const auto detectorEvents = detectorPatchesIt->getEvents();                               \
			const auto gtEvents = patchIt->getEvents();                                               \
			ASSERT_EQ(detectorEvents.size(), gtEvents.size());                                        \
                                                                                                      \
			auto detectorIt = detectorEvents.begin();                                                 \
			for (auto gtIt = gtEvents.begin();                                                        \
				 gtIt != gtEvents.end() && detectorIt != detectorEvents.end();                        \
				 ++gtIt, ++detectorIt)                                                                \
			{                                                                                         \
				EXPECT_EQ(detectorIt->timestamp, gtIt->timestamp);                                    \
			}                                                                                         \
		}                                                                                             \
	}

#define REFERENCE_ASSOCIATED_PATCHES_TEST(POLICY)                                                     \
	{                                                                                                 \
		tracker::DetectorParams params;                                                               \
		params.patchExtent = 5;                                                                       \
		params.associationDistance = 5;                                                               \
		POLICY;                                                                                       \
		tracker::FeatureDetector detector(params);                                                    \
		const common::timestamp_t timestamp(0);                                                       \
                                                                                                      \
		tracker::Patches patches = {tracker::Patch({0, 0}, 11, timestamp),                            \
									tracker::Patch({5, 5}, 11, timestamp),                            \
									tracker::Patch({20, 20}, 11, timestamp)};                         \
                                                                                                      \
		tracker::TrackId trackId = 0;                                                                 \
		for (auto& patch : patches)                                                                   \
		{                                                                                             \
			patch.setTrackId(trackId++);                                                              \
		}                                                                                             \
		detector.setPatches(patches);                                                                 \
		detector.setTrackId(trackId);                                                                 \
                                                                                                      \
		tracker::Patches newPatches = {tracker::Patch({3, 0}, 11, timestamp),                         \
									   tracker::Patch({0, 1}, 11, timestamp),                         \
									   tracker::Patch({18, 18}, 11, timestamp)};                      \
                                                                                                      \
		detector.associatePatches(newPatches, common::timestamp_t(0));                                \
		const auto updatedPatches = detector.getPatches();                                            \
		EXPECT_EQ(updatedPatches.size(), 4u);                                                         \
		/* beyond the reference's check: the one unmatched patch takes the next track id */           \
		EXPECT_EQ(updatedPatches.back().getTrackId(), 3);                                             \
		EXPECT_EQ(updatedPatches.front().getTrajectory().size(), 2u);                                 \
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
	REFERENCE_UPDATE_PATCH_TEST(params.errorPolicy = tracker::DetectorParams::ERRORS_STATUS)
	REFERENCE_ASSOCIATED_PATCHES_TEST(params.errorPolicy = tracker::DetectorParams::ERRORS_STATUS)
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

	// ---- the reference's two detector tests, unchanged ---------------------------------------------
	REFERENCE_UPDATE_PATCH_TEST((void)0)
	REFERENCE_ASSOCIATED_PATCHES_TEST((void)0)

	// ---- the evaluator loop: ONE FeatureDetector against the former pair ----------------------------
	{
		tools::EvaluatorParams ep;
		ep.outputDir = "/tmp";
		ep.compensationFrequencyEvents = 6000;
		ep.compensationFrequencyTime = 4000000000u;  // windows by event count only
		tools::Evaluator evaluator(ep);
		auto optOne = installTracked(evaluator.tracker(), W, H);

		tracker::TrackedPatches pairTracked(tracker::Size(W, H));
		auto optPair = installTracked(pairTracked, W, H);
		tracker::DetectorParams dp;
		tracker::FeatureDetector pairDetector(dp);

		const std::vector<common::EventSample> stream = makeStream(15000, W, H);
		int windows = 0, sameWindows = 0;
		for (const auto& sample : stream)
		{
			evaluator.eventCallback(sample);
			// the pair, in the same call order
			pairDetector.addEvent(sample);
			pairTracked.updatePatches(sample);
			if (pairDetector.getEvents().size() >= ep.compensationFrequencyEvents)
			{
				pairDetector.compensateEventsContrast(pairDetector.getEvents());
				pairDetector.integrateEvents(pairDetector.getEvents());
				pairDetector.clearEvents();
				++windows;
				const bool same = sameImage(evaluator.getCompensatedEventImage(), pairDetector.getCompensatedEventImage()) &&
								  sameImage(evaluator.getIntegratedEventImage(), pairDetector.getIntegratedEventImage()) &&
								  evaluator.tracker().getPatchFlows() == pairDetector.getPatchFlows() &&
								  evaluator.tracker().getLastCompensation() == pairDetector.getLastCompensation();
				sameWindows += same ? 1 : 0;
				double maxFlow = 0;
				for (double f : evaluator.tracker().getPatchFlows())
				{
					maxFlow = std::fmax(maxFlow, std::fabs(f));
				}
				std::printf("window %d: %d iterations, max |flow| %.4f, same as the pair: %d\n", windows,
							evaluator.tracker().getLastSummary().iterations, maxFlow, int(same));
				EXPECT_TRUE(maxFlow > 1e-3);  // the solve moved
			}
		}
		EXPECT_TRUE(windows == 2 && evaluator.windows == windows && sameWindows == windows);
		const tracker::Patches& one = evaluator.getPatches();
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
		const auto costs = evaluator.tracker().getOptimizedFinalCosts();
		EXPECT_TRUE(costs.size() == optimisations);
		std::printf("evaluator loop: %zu events, %d windows, %zu optimisations, one detector == the pair\n", stream.size(),
					windows, optimisations);

		// the chunked call of the same class: same patches as the per-event loop
		tools::Evaluator chunked(ep);
		installTracked(chunked.tracker(), W, H);
		chunked.tracker().updatePatches(stream);
		for (size_t i = 0; i < one.size(); ++i)
		{
			EXPECT_TRUE(samePatch(nth(one, i), nth(chunked.getPatches(), i)));
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

		// visual_odometry::Keyframe over the detector's patches (evaluator.cpp:85-87)
		auto keyframe = visual_odometry::Keyframe(evaluator.tracker().getPatches(), common::timestamp_t(99));
		EXPECT_TRUE(keyframe.getLandmarks().size() == 5);  // track ids 0..3 and the two -1 share one key
		const auto& lm = keyframe.getLandmarks().at(2);
		EXPECT_TRUE(lm.x() == nth(one, 2).toCorner().x && lm.y() == nth(one, 2).toCorner().y);

		// setTrackerParams: params_ = params, optimizers take optimizerParams, reset() (:733-741)
		tracker::DetectorParams np;
		np.optimizerParams.maxNumIterations = 3;
		np.maxNumEventsToStore = 5000;
		evaluator.eventCallback(stream[0]);
		evaluator.setTrackerParams(np);
		EXPECT_TRUE(evaluator.tracker().getEvents().empty() && evaluator.tracker().getLastCompensation().count() == 0);
		EXPECT_TRUE(optOne->getParams()->maxNumIterations == 3);
		EXPECT_TRUE(evaluator.getCompensatedEventImage().template at<double>(90, 120) == 0.0);
		for (int i = 0; i < 5200; ++i)
		{
			evaluator.tracker().addEvent(stream[i]);
		}
		EXPECT_TRUE(evaluator.tracker().getEvents().size() == 5000);
		// ~Evaluator: preExit, trajectory.txt, final_cost.txt
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
