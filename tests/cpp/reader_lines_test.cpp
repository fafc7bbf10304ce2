// reader_lines_test.cpp — the reference's own reader test against the facade's tools::Davis240cReader (CPU only).
//
// tools/dataset_reader/test/davis240c_reader_test.cpp:19-48 (Davis240cReader.eventsTest) is the block marked "verbatim",
// run on the reference's own events.txt fixture (tests/golden/davis_events_fixture.txt, copied to <tmp>/events.txt).
// Besides: a recording read in pieces (getEvents continues behind the events of the call before and ends with an empty
// optional), the packed sidecar next to the text file gives the same events, a sign other than 0/1 throws the reference's
// message, getTrajectoryLine / getTrajectory read what tools::saveFeaturesTrajectory wrote (one Patch per line), and the
// three getters that are not on the event path say so.
//   usage: reader_lines_test <events fixture> <scratch dir>
#include <cstdio>
#include <fstream>
#include <string>
#include <sys/stat.h>
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

std::string TEST_DATA_PATH = "test/test_data";

// ---- davis240c_reader_test.cpp:19-48, verbatim ------------------------------------------------------
static void eventsTest()
{
	tools::Davis240cReader reader(TEST_DATA_PATH);
	const auto events = reader.getEvents();

	EXPECT_TRUE(events.has_value());

	std::vector<common::Point2i> points = {
		{33, 39}, {158, 145}, {88, 143}, {174, 154}, {112, 139}};

	std::vector<common::EventPolarity> signs = {
		common::EventPolarity::POSITIVE, common::EventPolarity::POSITIVE,
		common::EventPolarity::NEGATIVE, common::EventPolarity::NEGATIVE,
		common::EventPolarity::POSITIVE};

	std::vector<common::timestamp_t> timestamps = {
		common::timestamp_t(0), common::timestamp_t(11),
		common::timestamp_t(50), common::timestamp_t(55),
		common::timestamp_t(80)};

	ASSERT_EQ(timestamps.size(), events.value().size());

	for (size_t i = 0; i < events.value().size(); ++i)
	{
		EXPECT_EQ(events.value()[i].timestamp.count(), timestamps[i].count());
		EXPECT_EQ(events.value()[i].value.point.x, points[i].x);
		EXPECT_EQ(events.value()[i].value.point.y, points[i].y);
		EXPECT_EQ(events.value()[i].value.sign, signs[i]);
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
	TEST_DATA_PATH = scratch;
	eventsTest();
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
