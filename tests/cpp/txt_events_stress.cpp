// txt_events_stress.cpp -- the parallel events.txt reader (csrc/txt_events.h) against the single-thread reader of
// rounds 1-4, on the CPU (plain and under ThreadSanitizer: tests/test_host_sanitizers.py).
//
// The comparator below IS the reader of rounds 1-4 (fread into a buffer, one std::string per line, strtod / strtol),
// kept here as test infrastructure.  For every file, thread count, cap and chain of offsets the product's reader must
// return the same events, the same count, the same byte offset and the same status.
//   usage: txt_events_stress <scratch dir> [lines of the large file, default 10000000]
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../event-based-odomety_amd/csrc/txt_events.h"

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

// ---- the reader of rounds 1-4 -----------------------------------------------------------------------
static int old_reader(const char* path, uint64_t* offset, ebo_event* out, size_t cap, size_t* n)
{
	*n = 0;
	FILE* fp = std::fopen(path, "rb");
	if (!fp)
	{
		return EBO_ERR_ARG;
	}
	uint64_t pos = offset ? *offset : 0;
	if (pos && fseeko(fp, static_cast<off_t>(pos), SEEK_SET) != 0)
	{
		std::fclose(fp);
		return EBO_ERR_ARG;
	}
	std::vector<char> buf(1 << 20);
	std::string line;
	size_t count = 0;
	int rc = EBO_OK;
	auto take = [&](const std::string& ln) -> bool {
		const char* s = ln.c_str();
		char* end = nullptr;
		const double sec = std::strtod(s, &end);
		if (end == s)
		{
			return true;
		}
		const char* p = end;
		const long x = std::strtol(p, &end, 10);
		if (end == p)
		{
			rc = EBO_ERR_RANGE;
			return false;
		}
		p = end;
		const long y = std::strtol(p, &end, 10);
		if (end == p)
		{
			rc = EBO_ERR_RANGE;
			return false;
		}
		p = end;
		const long sign = std::strtol(p, &end, 10);
		if (end == p || (sign != 0 && sign != 1))
		{
			rc = EBO_ERR_RANGE;
			return false;
		}
		if (count >= cap)
		{
			return false;
		}
		ebo_event& e = out[count++];
		e.x = static_cast<int32_t>(x);
		e.y = static_cast<int32_t>(y);
		e.sign = sign == 0 ? -1 : 1;
		e.reserved = 0;
		e.t_us = static_cast<int64_t>(sec * 1000000.0);
		return true;
	};
	bool go = true;
	while (go)
	{
		const size_t got = std::fread(buf.data(), 1, buf.size(), fp);
		if (got == 0)
		{
			break;
		}
		size_t start = 0;
		for (size_t i = 0; i < got && go; ++i)
		{
			if (buf[i] == '\n')
			{
				line.append(buf.data() + start, i - start);
				go = take(line);
				if (go)
				{
					pos += line.size() + 1;
				}
				line.clear();
				start = i + 1;
			}
		}
		if (go)
		{
			line.append(buf.data() + start, got - start);
		}
	}
	if (go && !line.empty() && take(line))
	{
		pos += line.size();
	}
	std::fclose(fp);
	*n = count;
	if (offset)
	{
		*offset = pos;
	}
	return rc;
}

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
	g_rng ^= g_rng << 13;
	g_rng ^= g_rng >> 7;
	g_rng ^= g_rng << 17;
	return g_rng;
}

// one call of each reader with the same arguments: same everything
static bool same_call(const std::string& path, uint64_t offset, size_t cap, unsigned threads, size_t room, uint64_t* nextOffset,
					  size_t* got, int* status)
{
	std::vector<ebo_event> a(room + 1), b(room + 1);
	std::memset(a.data(), 0x5a, a.size() * sizeof(ebo_event));
	std::memset(b.data(), 0x5a, b.size() * sizeof(ebo_event));
	uint64_t oa = offset, ob = offset;
	size_t na = 0, nb = 0;
	const int ra = old_reader(path.c_str(), &oa, a.data(), cap, &na);
	const int rb = ebo::txt::read_events_file(path.c_str(), &ob, b.data(), cap, &nb, threads);
	// the events [0, n) and nothing at or behind slot cap (slots in [n, cap) are the caller's scratch: the parallel reader
	// writes chunks at their line-counted slots first and closes the gaps afterwards)
	const bool same = ra == rb && na == nb && oa == ob && std::memcmp(a.data(), b.data(), na * sizeof(ebo_event)) == 0 &&
					  std::memcmp(a.data() + cap, b.data() + cap, (a.size() - cap) * sizeof(ebo_event)) == 0;
	if (!same)
	{
		std::printf("  differs: %s offset %" PRIu64 " cap %zu threads %u: old rc %d n %zu off %" PRIu64 " | new rc %d n %zu off %" PRIu64 "\n",
					path.c_str(), offset, cap, threads, ra, na, oa, rb, nb, ob);
	}
	if (nextOffset)
	{
		*nextOffset = ob;
	}
	if (got)
	{
		*got = nb;
	}
	if (status)
	{
		*status = rb;
	}
	return same;
}

// a file of `lines` lines: mostly canonical, with the shapes a fast path must hand to strtod / strtol mixed in
static size_t write_mixed(const std::string& path, size_t lines, bool crlf, bool lastNewline, long malformedAt)
{
	FILE* fp = std::fopen(path.c_str(), "wb");
	size_t events = 0;
	double t = 1468941032.0;
	for (size_t i = 0; i < lines; ++i)
	{
		t += static_cast<double>(rnd() % 97) * 1e-6;
		const unsigned x = static_cast<unsigned>(rnd() % 240), y = static_cast<unsigned>(rnd() % 180), s = static_cast<unsigned>(rnd() & 1);
		const unsigned shape = static_cast<unsigned>(rnd() % 64);
		const char* eol = (crlf ? "\r\n" : "\n");
		if (i + 1 == lines && !lastNewline)
		{
			eol = "";
		}
		if (static_cast<long>(i) == malformedAt)
		{
			std::fprintf(fp, "%.6f %u %u 2%s", t, x, y, eol);  // a sign that is neither 0 nor 1
			continue;
		}
		switch (shape)
		{
		case 0: std::fprintf(fp, "%s", eol); break;                                          // blank
		case 1: std::fprintf(fp, "   %s", eol); break;                                       // blanks only
		case 2: std::fprintf(fp, "  %.6f  %u\t%u   %u  %s", t, x, y, s, eol); ++events; break;  // several blanks, a tab
		case 3: std::fprintf(fp, "%.9e %u %u %u%s", t, x, y, s, eol); ++events; break;         // an exponent
		case 4: std::fprintf(fp, "+%.6f +%u %u %u%s", t, x, y, s, eol); ++events; break;       // signs
		case 5: std::fprintf(fp, "%.6f %u %u %u trailing text%s", t, x, y, s, eol); ++events; break;
		case 6: std::fprintf(fp, "%.12f %u %u %u%s", t, x, y, s, eol); ++events; break;        // 22 digits: beyond the fast path
		case 7: std::fprintf(fp, "9007199254740993 %u %u %u%s", x, y, s, eol); ++events; break;  // mantissa above 2^53
		case 8: std::fprintf(fp, "# a comment line%s", eol); break;                          // strtod takes nothing: skipped
		case 9: std::fprintf(fp, "0.000011 %u %u %u%s", x, y, s, eol); ++events; break;        // leading zeros
		case 10: std::fprintf(fp, "%u %u %u %u%s", static_cast<unsigned>(i), x, y, s, eol); ++events; break;  // whole seconds
		case 11: std::fprintf(fp, "%.6f %u %u 0%u%s", t, x, y, s, eol); ++events; break;       // "00" / "01"
		default: std::fprintf(fp, "%.6f %u %u %u%s", t, x, y, s, eol); ++events; break;
		}
	}
	std::fclose(fp);
	return events;
}

int main(int argc, char** argv)
{
	if (argc < 2)
	{
		std::printf("usage: txt_events_stress <scratch dir> [lines]\n");
		return 2;
	}
	const std::string dir = argv[1];
	const size_t bigLines = argc > 2 ? static_cast<size_t>(std::strtoull(argv[2], nullptr, 10)) : 10000000;
	const bool quick = argc > 3 && std::string(argv[3]) == "quick";  // (under ThreadSanitizer: fewer sizes and thread counts)

	// 1. small and mid-size mixed files: every combination of thread count, cap, offset chain
	{
		const std::vector<size_t> sizes = quick ? std::vector<size_t>{0, 5, 4097, 60000} : std::vector<size_t>{0, 1, 5, 4097, 60000, 250000};
		const std::vector<unsigned> threadCounts = quick ? std::vector<unsigned>{1u, 3u, 8u} : std::vector<unsigned>{1u, 2u, 3u, 8u, 16u};
		int k = 0;
		for (size_t lines : sizes)
		{
			for (int variant = 0; variant < 4; ++variant, ++k)
			{
				const std::string path = dir + "/mixed_" + std::to_string(k) + ".txt";
				const bool crlf = variant == 1, lastNl = variant != 2;
				const long bad = variant == 3 && lines > 10 ? static_cast<long>(lines * 2 / 3) : -1;
				const size_t events = write_mixed(path, lines, crlf, lastNl, bad);
				for (unsigned threads : threadCounts)
				{
					for (size_t cap : {static_cast<size_t>(0), static_cast<size_t>(1), static_cast<size_t>(4095), static_cast<size_t>(4096),
									   static_cast<size_t>(4097), events / 3, events, events + 5})
					{
						EXPECT_TRUE(same_call(path, 0, cap, threads, std::max<size_t>(cap, 1), nullptr, nullptr, nullptr));
					}
					// the reference's way: pieces of a fixed number of lines, each call continuing behind the one before
					uint64_t off = 0;
					const size_t piece = std::max<size_t>(1, events / 7 + 3);
					for (int calls = 0; calls < 12; ++calls)
					{
						uint64_t next = 0;
						size_t got = 0;
						int st = 0;
						EXPECT_TRUE(same_call(path, off, piece, threads, piece, &next, &got, &st));
						if (st != EBO_OK || got == 0)
						{
							break;
						}
						off = next;
					}
				}
			}
		}
		std::printf("mixed files: %d files x %zu thread counts x 8 caps + offset chains compared\n", k, threadCounts.size());
	}

	// 1b. compact 8-byte records straight from the text (Sink8): the records ebo_pack_events8 makes of the 24-byte reader's
	// events with the first event's time as the base; an event a compact record cannot hold ends the result like a
	// malformed line does
	{
		const std::string path = dir + "/compact.txt";
		std::vector<size_t> lineStart;
		const size_t lines = quick ? 60000 : 400000;
		{
			FILE* fp = std::fopen(path.c_str(), "wb");
			double t = 17.25;
			size_t at = 0;
			for (size_t i = 0; i < lines; ++i)
			{
				t += static_cast<double>(rnd() % 61) * 1e-6;
				lineStart.push_back(at);
				int w;
				if (i % 97 == 5)
				{
					w = std::fprintf(fp, "\r\n");
				}
				else
				{
					w = std::fprintf(fp, "%.6f %u %u %u%s", t, static_cast<unsigned>(rnd() % 1280), static_cast<unsigned>(rnd() % 720),
									 static_cast<unsigned>(rnd() & 1), (i % 5 == 0) ? "\r\n" : "\n");
				}
				at += static_cast<size_t>(w);
			}
			std::fclose(fp);
		}
		std::vector<ebo_event> a(lines);
		size_t na = 0;
		uint64_t oa = 0;
		EXPECT_TRUE(ebo::txt::read_events_file(path.c_str(), &oa, a.data(), a.size(), &na, 1) == EBO_OK && na > lines * 9 / 10);
		for (unsigned threads : {1u, 3u, 8u})
		{
			for (size_t cap : {na, na / 2 + 7, static_cast<size_t>(4096)})
			{
				std::vector<ebo_event8> b(cap + 1);
				size_t nb = 0;
				uint64_t ob = 0;
				int64_t base = -1;
				const int rc = ebo::txt::read_events_file<ebo::txt::Sink8>(path.c_str(), &ob, b.data(), cap, &nb, threads, nullptr, &base);
				std::vector<ebo_event> ref(cap + 1);
				size_t nr = 0;
				uint64_t orf = 0;
				const int rr = ebo::txt::read_events_file(path.c_str(), &orf, ref.data(), cap, &nr, 1);
				bool same = rc == rr && nb == nr && ob == orf && base == a[0].t_us;
				for (size_t i = 0; same && i < nb; ++i)
				{
					const ebo_event& e = ref[i];
					const uint32_t xy = (static_cast<uint32_t>(e.x) & 0x7FFFu) | (static_cast<uint32_t>(e.sign > 0) << 15) |
										((static_cast<uint32_t>(e.y) & 0x7FFFu) << 16);
					same = b[i].xy == xy && b[i].t_rel_us == static_cast<int32_t>(e.t_us - base);
				}
				EXPECT_TRUE(same);
			}
		}
		// line 1234 gets a coordinate no compact record holds: the events before it, EBO_ERR_RANGE, the offset in front of it
		{
			const std::string bad = dir + "/compact_bad.txt";
			FILE* fp = std::fopen(bad.c_str(), "wb");
			size_t at = 0, badAt = 0, before = 0;
			for (size_t i = 0; i < 3000; ++i)
			{
				if (i == 1234)
				{
					badAt = at;
					at += static_cast<size_t>(std::fprintf(fp, "0.500000 20000 3 1\n"));
				}
				else
				{
					at += static_cast<size_t>(std::fprintf(fp, "0.%06zu %zu %zu 0\n", i, i % 640, i % 480));
					before += i < 1234;
				}
			}
			std::fclose(fp);
			for (unsigned threads : {1u, 8u})
			{
				std::vector<ebo_event8> b(4000);
				size_t nb = 0;
				uint64_t ob = 0;
				int64_t base = -1;
				const int rc = ebo::txt::read_events_file<ebo::txt::Sink8>(bad.c_str(), &ob, b.data(), b.size(), &nb, threads, nullptr, &base);
				EXPECT_TRUE(rc == EBO_ERR_RANGE && nb == before && ob == badAt && base == 0);
			}
		}
		std::printf("compact records: %zu lines, 3 thread counts x 3 caps, and the out-of-range line\n", lines);
	}

	// 2. the large canonical file: bit-identical, and the rates
	{
		const std::string path = dir + "/large.txt";
		{
			FILE* fp = std::fopen(path.c_str(), "wb");
			double t = 1468941032.0;
			for (size_t i = 0; i < bigLines; ++i)
			{
				t += static_cast<double>(rnd() % 53) * 1e-6;
				std::fprintf(fp, "%.6f %u %u %u\n", t, static_cast<unsigned>(rnd() % 240), static_cast<unsigned>(rnd() % 180),
							 static_cast<unsigned>(rnd() & 1));
			}
			std::fclose(fp);
		}
		std::vector<ebo_event> a(bigLines), b(bigLines);
		size_t na = 0;
		auto t0 = std::chrono::steady_clock::now();
		EXPECT_TRUE(old_reader(path.c_str(), nullptr, a.data(), a.size(), &na) == EBO_OK && na == bigLines);
		const double oldMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		std::printf("large file, %zu lines: rounds 1-4 reader %.0f ms = %.2f Mlines/s\n", bigLines, oldMs, bigLines / oldMs * 1e-3);
		for (unsigned threads : {1u, 2u, 4u, 8u, 16u})
		{
			size_t nb = 0;
			unsigned used = 0;
			std::memset(b.data(), 0, b.size() * sizeof(ebo_event));
			t0 = std::chrono::steady_clock::now();
			const int rc = ebo::txt::read_events_file(path.c_str(), nullptr, b.data(), b.size(), &nb, threads, &used);
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			EXPECT_TRUE(rc == EBO_OK && nb == bigLines && std::memcmp(a.data(), b.data(), bigLines * sizeof(ebo_event)) == 0);
			std::printf("   %2u thread(s) asked, %2u used: %.0f ms = %.2f Mlines/s (x %.1f)\n", threads, used, ms, bigLines / ms * 1e-3, oldMs / ms);
		}
		// the reference's pieces of 1 000 000 lines (Davis240cReader::getEvents)
		uint64_t off = 0;
		size_t total = 0;
		for (;;)
		{
			size_t nb = 0;
			const int rc = ebo::txt::read_events_file(path.c_str(), &off, b.data(), 1000000, &nb, 0);
			EXPECT_TRUE(rc == EBO_OK);
			if (nb == 0)
			{
				break;
			}
			EXPECT_TRUE(std::memcmp(a.data() + total, b.data(), nb * sizeof(ebo_event)) == 0);
			total += nb;
		}
		EXPECT_TRUE(total == bigLines);
	}
	std::printf("txt_events_stress: %s (%d failure%s)\n", g_fail ? "FAILED" : "OK", g_fail, g_fail == 1 ? "" : "s");
	return g_fail ? 1 : 0;
}
