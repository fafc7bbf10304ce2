// reader_lines_test.cpp — the facade's tools::Davis240cReader under the reference reader's known answers (CPU only).
//
// The reference's reader test (tools/dataset_reader/test/davis240c_reader_test.cpp:19-48) reads its five-line events.txt
// fixture and expects five (timestamp, pixel, polarity) triples.  Those triples are held here as a data table and checked
// with this file's own statements on the same fixture (tests/golden/davis_events_fixture.txt, copied to <tmp>/events.txt);
// static_asserts pin the signatures of the members the reference's callers use.
// Besides: a recording read in pieces (getEvents continues behind the events of the call before and ends with an empty
// optional), the packed sidecar next to the text file gives the same events, a sign other than 0/1 throws the reference's
// message, getTrajectoryLine / getTrajectory read what tools::saveFeaturesTrajectory wrote (one Patch per line), and the
// three getters that are not on the event path say so.
//   usage: reader_lines_test <events fixture> <scratch dir>
#include <cstdio>
#include <fstream>
#include <optional>
#include <string>
#include <sys/stat.h>
#include <type_traits>
#include <utility>
#include <vector>

#include <dataset_reader/davis240c_reader.h>
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

// ---- the known answers of the reference's fixture (davis240c_reader_test.cpp:19-48), held as DATA ----
// The five lines of tests/golden/davis_events_fixture.txt and what the reference's reader makes of them: microseconds
// truncated from the seconds column, pixel, polarity (1 -> POSITIVE, 0 -> NEGATIVE).  The statements are this test's own;
// they go through the same members the reference's test names (the constructor from a directory, getEvents() returning an
// optional of a sequence of common::EventSample).
struct FixtureLine
{
	int64_t us;
	int x, y;
	common::EventPolarity sign;
};
static const FixtureLine kFixture[] = {{0, 33, 39, common::EventPolarity::POSITIVE},
									   {11, 158, 145, common::EventPolarity::POSITIVE},
									   {50, 88, 143, common::EventPolarity::NEGATIVE},
									   {55, 174, 154, common::EventPolarity::NEGATIVE},
									   {80, 112, 139, common::EventPolarity::POSITIVE}};

// signature conformance with tools/dataset_reader/include/dataset_reader/davis240c_reader.h: the members the
// reference's test and its replayer call
static_assert(std::is_constructible<tools::Davis240cReader, const std::string&>::value, "Davis240cReader(path)");
static_assert(std::is_same<decltype(std::declval<tools::Davis240cReader&>().getEvents()),
						   std::optional<common::EventSequence>>::value,
			  "getEvents() -> std::optional<common::EventSequence>");
static_assert(std::is_same<decltype(std::declval<tools::Davis240cReader&>().getEventSample(std::declval<std::string&>())),
						   common::EventSample>::value,
			  "getEventSample(std::string&) -> common::EventSample");

static void fixtureKnownAnswers(const std::string& dir)
{
	tools::Davis240cReader reader(dir);
	const std::optional<common::EventSequence> got = reader.getEvents();
	EXPECT_TRUE(got.has_value());
	if (!got)
	{
		return;
	}
	const size_t want = sizeof(kFixture) / sizeof(kFixture[0]);
	EXPECT_EQ(got->size(), want);
	size_t k = 0;
	for (const common::EventSample& e : *got)
	{
		if (k >= want)
		{
			break;
		}
		const FixtureLine& f = kFixture[k++];
		EXPECT_TRUE(e.timestamp == common::timestamp_t(f.us));
		EXPECT_TRUE(e.value.point.x == f.x && e.value.point.y == f.y);
		EXPECT_TRUE(e.value.sign == f.sign);
	}
}

static bool sameEvents(const common::EventSequence& a, const common::EventSequence& b)
{
	if (a.size() != b.size())
	{
		return false;
	}
	auto ia = a.begin();
	auto ib = b.begin();
	for (; ia != a.end(); ++ia, ++ib)
	{
		if (ia->timestamp != ib->timestamp || ia->value.point.x != ib->value.point.x || ia->value.point.y != ib->value.point.y ||
			ia->value.sign != ib->value.sign)
		{
			return false;
		}
	}
	return true;
}

int main(int argc, char** argv)
{
	if (argc < 3)
	{
		std::printf("usage: reader_lines_test <events fixture> <scratch dir>\n");
		return 2;
	}
	const std::string fixture = argv[1], scratch = argv[2];
	{
		std::ifstream in(fixture, std::ios::binary);
		std::ofstream out(scratch + "/events.txt", std::ios::binary);
		out << in.rdbuf();
	}
	fixtureKnownAnswers(scratch);
	{
		tools::Davis240cReader reader(scratch);
		EXPECT_TRUE(reader.getEvents().has_value());
		EXPECT_TRUE(!reader.getEvents().has_value());  // the second call starts behind the five events: the end
		EXPECT_TRUE(reader.eventsRead() == 5);
		// one line, as the reference parses it (the line is consumed)
		std::string line = "0.000055 174 154 0";
		const common::EventSample s = reader.getEventSample(line);
		EXPECT_TRUE(s.timestamp.count() == 55 && s.value.point.x == 174 && s.value.point.y == 154 && s.value.sign == common::NEGATIVE);
		std::string bad = "0.1 3 4 2";
		bool threw = false;
		try
		{
			reader.getEventSample(bad);
		}
		catch (const std::runtime_error& e)
		{
			threw = std::string(e.what()) == "Sign is not equal to 0/1";
		}
		EXPECT_TRUE(threw);
		for (int which = 0; which < 3; ++which)
		{
			bool said = false;
			try
			{
				which == 0 ? reader.getImages() : which == 1 ? reader.getGroundTruth() : reader.getCalibration();
			}
			catch (const std::runtime_error& e)
			{
				said = std::string(e.what()).find("not on the event path") != std::string::npos;
			}
			EXPECT_TRUE(said);
		}
	}
	// a longer recording, read in pieces of EVENT_LENGTH: 2.3 pieces
	const std::string big = scratch + "/big";
	::mkdir(big.c_str(), 0700);
	const size_t N = 2300000;
	{
		std::FILE* fp = std::fopen((big + "/events.txt").c_str(), "wb");
		for (size_t i = 0; i < N; ++i)
		{
			std::fprintf(fp, "%zu.%06zu %zu %zu %d\n", i / 1000000, i % 1000000, i % 240, (i * 7) % 180, static_cast<int>(i % 3 == 0));
		}
		std::fclose(fp);
	}
	common::EventSequence all;
	{
		tools::Davis240cReader reader(big);
		std::vector<size_t> pieces;
		while (auto ev = reader.getEvents())
		{
			pieces.push_back(ev->size());
			all.insert(all.end(), ev->begin(), ev->end());
		}
		EXPECT_TRUE(pieces.size() == 3 && pieces[0] == 1000000 && pieces[1] == 1000000 && pieces[2] == 300000);
		EXPECT_TRUE(all.size() == N);
		size_t wrong = 0, i = 0;
		for (const auto& e : all)
		{
			// (seconds through a double, truncated: i / 1e6 + frac can land one below -- compare with the same arithmetic)
			const double sec = std::stod(std::to_string(i / 1000000) + "." + [&] {
				char b[8];
				std::snprintf(b, sizeof(b), "%06zu", i % 1000000);
				return std::string(b);
			}());
			wrong += e.timestamp.count() != static_cast<int64_t>(sec * 1000000.0) || e.value.point.x != static_cast<int>(i % 240) ||
					 e.value.point.y != static_cast<int>((i * 7) % 180) || (e.value.sign == common::POSITIVE) != (i % 3 == 0);
			++i;
		}
		EXPECT_TRUE(wrong == 0);
	}
	// the packed sidecar next to the text file: the same events, piece by piece
	{
		const std::vector<ebo_event> flat = common::toEboEvents(all);
		EXPECT_TRUE(ebo_write_events_bin((big + "/events.bin").c_str(), flat.data(), flat.size()) == EBO_OK);
		tools::Davis240cReader reader(big);
		common::EventSequence again;
		size_t pieces = 0;
		while (auto ev = reader.getEvents())
		{
			++pieces;
			again.insert(again.end(), ev->begin(), ev->end());
		}
		EXPECT_TRUE(pieces == 3 && sameEvents(all, again));
	}
	// trajectory.txt: what saveFeaturesTrajectory wrote, one Patch per line
	{
		tracker::Patches patches;
		patches.emplace_back(tracker::Corner(10.5, 20.25), 3, common::timestamp_t(1000));
		patches.back().setTrackId(7);
		patches.back().setCorner(tracker::Corner(11.5, 21.0), common::timestamp_t(2500));
		patches.emplace_back(tracker::Corner(100.0, 50.0), 3, common::timestamp_t(1500));
		patches.back().setTrackId(9);
		tools::saveFeaturesTrajectory(patches, scratch + "/trajectory.txt");
		tools::Davis240cReader reader(scratch);
		const tracker::Patches read = reader.getTrajectory();
		EXPECT_TRUE(read.size() == 3);
		auto it = read.begin();
		EXPECT_TRUE(it->getTrackId() == 7 && it->toCorner().x == 10.5 && it->toCorner().y == 20.25 && it->getCurrentTimestamp().count() == 1000);
		EXPECT_TRUE(it->getPatch().width == 3.0);  // extent 1
		++it;
		EXPECT_TRUE(it->getTrackId() == 7 && it->toCorner().x == 11.5 && it->getCurrentTimestamp().count() == 2500);
		++it;
		EXPECT_TRUE(it->getTrackId() == 9 && it->toCorner().x == 100.0 && it->toCorner().y == 50.0);
	}
	std::printf("reader_lines_test: %s (%d failure%s)\n", g_fail ? "FAILED" : "OK", g_fail, g_fail == 1 ? "" : "s");
	return g_fail ? 1 : 0;
}
