// txt_events.h -- the DAVIS events.txt reader, parallel since round 5.  Header-only and free of
// HIP, so that it is built and run on the CPU under ThreadSanitizer exactly as it ships inside libebo_hip.so
// (tests/cpp/txt_events_stress.cpp, tests/test_host_sanitizers.py).
//
// What it reads: one event per line, "<seconds> <x> <y> <0|1>" (Davis240cReader::getEventSample,
// tools/dataset_reader/src/davis240c_reader.cpp:60-92).  Seconds go through a double and are truncated to
// microseconds exactly as std::stod + duration_cast do there; a sign other than 0 / 1 is an error (the reference
// throws "Sign is not equal to 0/1").
//
// How the reference does it (tools/dataset_reader/include/dataset_reader/dataset_reader.h:33-97): the file is mapped,
// cut into std::strings line by line on one thread, and hardware_concurrency() threads parse equal shares of the lines.
// Here: the file is mapped, the byte range that can hold the events still wanted is cut at line breaks into one chunk
// per thread, every thread walks and parses its own chunk (no per-line allocation, no shared state), and the chunks'
// events are copied behind each other in file order.  The result -- events, their number, the byte offset behind the
// last line taken, the error -- is the single-thread walk's (read_serial below), whatever the thread count.
//
// A line is parsed by a fast path when it has the canonical shape  digits[.digits] SP digits SP digits SP (0|1)
// [CR]: the seconds as an integer mantissa m < 2^53 over an exact power of ten <= 10^22, ONE IEEE division -- by
// Clinger's theorem the correctly rounded value, i.e. what strtod returns -- and three small integers.  Any other line
// (signs, exponents, tabs, several blanks, inf / nan, garbage) takes the strtod / strtol path of rounds 1-4 on a copy
// of the line, so malformed input is judged exactly as before.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/ebo.h"

namespace ebo
{
namespace txt
{
// what one line is
enum LineKind
{
	kBlank = 0,     // nothing a number starts with: consumed, no event (strtod takes nothing)
	kEvent = 1,     // an event (written to `out`)
	kMalformed = 2  // a number is missing, or the sign is neither 0 nor 1
};

// The strtod / strtol path of rounds 1-4 (the reference's stod / stoi calls on the line's pieces).
inline LineKind parse_line_slow(const char* s, const char* e, ebo_event& out)
{
	const std::string line(s, e);  // NUL-terminated copy: strtod must not run into the next line
	const char* b = line.c_str();
	char* end = nullptr;
	const double sec = std::strtod(b, &end);
	if (end == b)
	{
		return kBlank;
	}
	const char* p = end;
	const long x = std::strtol(p, &end, 10);
	if (end == p)
	{
		return kMalformed;
	}
	p = end;
	const long y = std::strtol(p, &end, 10);
	if (end == p)
	{
		return kMalformed;
	}
	p = end;
	const long sign = std::strtol(p, &end, 10);
	if (end == p || (sign != 0 && sign != 1))
	{
		return kMalformed;  // "Sign is not equal to 0/1" (davis240c_reader.cpp:85-88)
	}
	out.x = static_cast<int32_t>(x);
	out.y = static_cast<int32_t>(y);
	out.sign = sign == 0 ? -1 : 1;
	out.reserved = 0;
	out.t_us = static_cast<int64_t>(sec * 1000000.0);
	return kEvent;
}

// line = [s, e), without its '\n'
inline LineKind parse_line(const char* s, const char* e, ebo_event& out)
{
	static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
									  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
	const char* p = s;
	uint64_t m = 0;
	int digits = 0, frac = 0;
	while (p < e && static_cast<unsigned>(*p - '0') < 10u && digits < 19)
	{
		m = m * 10 + static_cast<unsigned>(*p - '0');
		++digits;
		++p;
	}
	if (digits == 0 || digits == 19)
	{
		return parse_line_slow(s, e, out);
	}
	if (p < e && *p == '.')
	{
		++p;
		while (p < e && static_cast<unsigned>(*p - '0') < 10u && digits < 19)
		{
			m = m * 10 + static_cast<unsigned>(*p - '0');
			++digits;
			++frac;
			++p;
		}
		if (digits == 19)
		{
			return parse_line_slow(s, e, out);
		}
	}
	if (m >= (static_cast<uint64_t>(1) << 53) || frac > 22 || p >= e || *p != ' ')
	{
		return parse_line_slow(s, e, out);
	}
	long v[2];
	for (int k = 0; k < 2; ++k)
	{
		++p;  // the blank
		long a = 0;
		int n = 0;
		while (p < e && static_cast<unsigned>(*p - '0') < 10u && n < 9)
		{
			a = a * 10 + (*p - '0');
			++n;
			++p;
		}
		if (n == 0 || n == 9 || p >= e || *p != ' ')
		{
			return parse_line_slow(s, e, out);
		}
		v[k] = a;
	}
	++p;
	if (p >= e || (*p != '0' && *p != '1'))
	{
		return parse_line_slow(s, e, out);
	}
	const bool positive = *p == '1';
	++p;
	if (!(p == e || (p + 1 == e && *p == '\r')))
	{
		return parse_line_slow(s, e, out);  // more digits, trailing text: strtol's business
	}
	const double sec = static_cast<double>(m) / kPow10[frac];
	out.x = static_cast<int32_t>(v[0]);
	out.y = static_cast<int32_t>(v[1]);
	out.sign = positive ? 1 : -1;
	out.reserved = 0;
	out.t_us = static_cast<int64_t>(sec * 1000000.0);
	return kEvent;
}

// end of the line that starts at p (the '\n', or `end` for a last line without one)
inline const char* line_end(const char* p, const char* end)
{
	const void* nl = std::memchr(p, '\n', static_cast<size_t>(end - p));
	return nl ? static_cast<const char*>(nl) : end;
}

// Where a parsed event goes: as it is (24 bytes, common::EventSample's layout), or as the compact 8-byte record of
// include/ebo.h (ebo_event8: x:15 | polarity:1 | y:15 | 0:1, microseconds relative to a base time) that
// ebo_set_windows8 takes -- text to device input with no 24-byte array in between.  A compact record that cannot hold
// the event (a coordinate beyond +-16384, a time further than 2^31 us from the base) makes its line a malformed one.
struct Sink24
{
	using Rec = ebo_event;
	int64_t base = 0;
	bool put(Rec& dst, const ebo_event& e) const
	{
		dst = e;
		return true;
	}
};
struct Sink8
{
	using Rec = ebo_event8;
	int64_t base = 0;
	bool put(Rec& dst, const ebo_event& e) const
	{
		const int64_t dt = e.t_us - base;
		if (e.x < -16384 || e.x > 16383 || e.y < -16384 || e.y > 16383 || dt < INT32_MIN || dt > INT32_MAX)
		{
			return false;
		}
		dst.xy = (static_cast<uint32_t>(e.x) & 0x7FFFu) | (static_cast<uint32_t>(e.sign > 0 ? 1 : 0) << 15) |
				 ((static_cast<uint32_t>(e.y) & 0x7FFFu) << 16);  // (ebo_internal.h: pack_lo)
		dst.t_rel_us = static_cast<int32_t>(dt);
		return true;
	}
};

// The single-thread walk: at most cap events from `pos` on.  Returns the position behind the last line taken; a line
// that holds an event beyond cap, or a malformed line (rc = EBO_ERR_RANGE), stays in front of it.
template <class Sink>
inline const char* read_serial(const char* pos, const char* end, typename Sink::Rec* out, size_t cap, size_t& count, int& rc,
							   const Sink& sink)
{
	rc = EBO_OK;
	while (pos < end)
	{
		const char* le = line_end(pos, end);
		ebo_event ev;
		LineKind k = parse_line(pos, le, ev);
		typename Sink::Rec rec;
		if (k == kEvent && !sink.put(rec, ev))
		{
			k = kMalformed;
		}
		if (k == kMalformed)
		{
			rc = EBO_ERR_RANGE;
			return pos;
		}
		if (k == kEvent)
		{
			if (count >= cap)
			{
				return pos;
			}
			out[count++] = rec;
		}
		pos = le < end ? le + 1 : end;
	}
	return pos;
}

// One thread's share of a window of the file.  The events go STRAIGHT into the caller's array: a first pass counts the
// chunks' lines, so that chunk t may assume that every line before it holds an event and starts writing at the slot
// that many places in; where lines held none (blank lines, comments) the merge closes the gap with one memmove.
struct Chunk
{
	const char* begin = nullptr;
	const char* end = nullptr;        // chunks end behind a '\n' (or at the end of the file)
	size_t lines = 0;                 // lines that start inside the chunk
	size_t slot0 = 0;                 // the chunk's first slot (lines before it in the window), relative to the window's first
	size_t events = 0;                // events written to slots [slot0, slot0 + events)
	const char* stop = nullptr;       // null: parsed to its end; else the line it stopped in front of
	bool malformed = false;           // ... because that line is malformed (else: no room for its event)
};

inline void count_lines(Chunk& c)
{
	size_t n = 0;
	const char* q = c.begin;
	while (q < c.end)
	{
		const void* nl = std::memchr(q, '\n', static_cast<size_t>(c.end - q));
		++n;  // a line starts at q (the last one of a file may have no '\n')
		if (!nl)
		{
			break;
		}
		q = static_cast<const char*>(nl) + 1;
	}
	c.lines = n;
}

// out: the window's first slot; room: slots from there to the end of the caller's array
template <class Sink>
inline void parse_chunk(Chunk& c, const char* fileEnd, typename Sink::Rec* out, size_t room, const Sink& sink)
{
	size_t slot = c.slot0;
	const char* pos = c.begin;
	while (pos < c.end)
	{
		const char* le = line_end(pos, fileEnd);
		ebo_event ev;
		LineKind k = parse_line(pos, le, ev);
		typename Sink::Rec rec;
		if (k == kEvent && !sink.put(rec, ev))
		{
			k = kMalformed;
		}
		if (k == kMalformed)
		{
			c.stop = pos;
			c.malformed = true;
			break;
		}
		if (k == kEvent)
		{
			if (slot >= room)
			{
				c.stop = pos;
				break;
			}
			out[slot++] = rec;
		}
		pos = le < fileEnd ? le + 1 : fileEnd;
	}
	c.events = slot - c.slot0;
}

inline unsigned thread_budget()
{
	unsigned hw = std::thread::hardware_concurrency();
	const char* v = std::getenv("EBO_HOST_THREADS");
	const unsigned want = v ? static_cast<unsigned>(std::max(1, std::atoi(v))) : (hw ? hw : 1u);
	return std::max(1u, std::min(want, 64u));
}

template <class F>
inline void on_threads(unsigned T, F&& fn)
{
	std::vector<std::thread> workers;
	workers.reserve(T - 1);
	for (unsigned t = 1; t < T; ++t)
	{
		workers.emplace_back([&fn, t] { fn(t); });
	}
	fn(0u);
	for (auto& w : workers)
	{
		w.join();
	}
}

// At most cap events from `pos` on with up to `threads` threads; same results as read_serial.
template <class Sink>
inline const char* read_parallel(const char* pos, const char* end, typename Sink::Rec* out, size_t cap, size_t& count, int& rc,
								 unsigned threads, const Sink& sink, unsigned* threadsUsed = nullptr)
{
	using Rec = typename Sink::Rec;
	rc = EBO_OK;
	const size_t kMinChunk = static_cast<size_t>(256) << 10;  // below this a thread is not worth starting
	if (threadsUsed)
	{
		*threadsUsed = 1;
	}
	while (count < cap && pos < end)
	{
		// the bytes that can hold the events still wanted: the mean line length of a sample, 2 % and 64 KiB of slack
		const size_t left = static_cast<size_t>(end - pos);
		const size_t sample = std::min<size_t>(left, static_cast<size_t>(64) << 10);
		size_t lines = 0;
		for (const char* q = pos; (q = static_cast<const char*>(std::memchr(q, '\n', static_cast<size_t>(pos + sample - q)))) != nullptr; ++q)
		{
			++lines;
		}
		const double perLine = static_cast<double>(sample) / static_cast<double>(std::max<size_t>(lines, 1));
		const double wantBytes = static_cast<double>(cap - count) * perLine * 1.02 + 65536.0;
		const size_t window = wantBytes >= static_cast<double>(left) ? left : static_cast<size_t>(wantBytes);
		const char* wEnd = window == left ? end : line_end(pos + window, end);
		wEnd = wEnd < end ? wEnd + 1 : end;  // the window ends behind a '\n' (or with the file)
		const unsigned T = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(threads, window / kMinChunk)));
		if (T <= 1)
		{
			// small input (or one thread): the serial walk over the window, then on if it was too short
			const char* next = read_serial(pos, wEnd, out, cap, count, rc, sink);
			if (rc != EBO_OK || next < wEnd)
			{
				return next;  // malformed line, or cap reached in front of an event
			}
			pos = next;
			continue;
		}
		if (threadsUsed)
		{
			*threadsUsed = std::max(*threadsUsed, T);
		}
		std::vector<Chunk> chunks(T);
		const char* cut = pos;
		for (unsigned t = 0; t < T; ++t)
		{
			chunks[t].begin = cut;
			if (t + 1 == T)
			{
				cut = wEnd;
			}
			else
			{
				const char* target = pos + (static_cast<size_t>(wEnd - pos) / T) * (t + 1);
				const char* le = line_end(std::max(target, cut), wEnd);
				cut = le < wEnd ? le + 1 : wEnd;
			}
			chunks[t].end = cut;
		}
		on_threads(T, [&chunks](unsigned t) { count_lines(chunks[t]); });
		size_t before = 0;
		for (unsigned t = 0; t < T; ++t)
		{
			chunks[t].slot0 = before;
			before += chunks[t].lines;
		}
		Rec* const first = out + count;
		const size_t room = cap - count;
		on_threads(T, [&chunks, end, first, room, &sink](unsigned t) { parse_chunk(chunks[t], end, first, room, sink); });
		// in file order: close the gaps that lines without an event left, stop at the first chunk that stopped
		size_t have = 0;  // events of the window so far
		for (unsigned t = 0; t < T; ++t)
		{
			const Chunk& c = chunks[t];
			if (c.events && c.slot0 != have)
			{
				std::memmove(first + have, first + c.slot0, c.events * sizeof(Rec));
			}
			have += c.events;
			if (c.stop)
			{
				count += have;
				if (c.malformed)
				{
					rc = EBO_ERR_RANGE;
					return c.stop;
				}
				// no room at the slot the chunk had counted up to: with lines that held no event before it there may be
				// room after all -- the serial walk goes on from the line the chunk stopped at (at most as many events
				// as there were such lines), and it is also what consumes blank lines behind the last event taken
				return read_serial(c.stop, end, out, cap, count, rc, sink);
			}
		}
		count += have;
		pos = wEnd;
	}
	if (count >= cap && pos < end)
	{
		// cap == 0, or cap reached exactly at a window's end: blank lines behind it are still consumed
		return read_serial(pos, end, out, cap, count, rc, sink);
	}
	return pos;
}

// A read-only mapping of a file (empty files map to nothing).
struct Mapping
{
	const char* data = nullptr;
	size_t size = 0;
	bool ok = false;
	explicit Mapping(const char* path)
	{
		const int fd = ::open(path, O_RDONLY);
		if (fd < 0)
		{
			return;
		}
		struct stat st;
		if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode))
		{
			::close(fd);
			return;
		}
		size = static_cast<size_t>(st.st_size);
		ok = true;
		if (size > 0)
		{
			void* p = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
			if (p == MAP_FAILED)
			{
				ok = false;
				size = 0;
			}
			else
			{
				data = static_cast<const char*>(p);
				(void)::madvise(p, size, MADV_SEQUENTIAL);
			}
		}
		::close(fd);
	}
	~Mapping()
	{
		if (data)
		{
			::munmap(const_cast<char*>(data), size);
		}
	}
	Mapping(const Mapping&) = delete;
	Mapping& operator=(const Mapping&) = delete;
};

// at most cap events from byte *offset of the file on (null: from the start); *offset moves behind the last line
// taken; threads == 0: EBO_HOST_THREADS or the machine's hardware threads
template <class Sink = Sink24>
inline int read_events_file(const char* path, uint64_t* offset, typename Sink::Rec* out, size_t cap, size_t* n, unsigned threads,
							unsigned* threadsUsed = nullptr, int64_t* baseOut = nullptr)
{
	if (!path || !n || (cap && !out))
	{
		return EBO_ERR_ARG;
	}
	*n = 0;
	Mapping map(path);
	if (!map.ok)
	{
		return EBO_ERR_ARG;
	}
	const uint64_t start = offset ? *offset : 0;
	if (start > map.size)
	{
		return EBO_ERR_ARG;
	}
	const char* begin = map.data ? map.data : "";
	const char* end = begin + map.size;
	size_t count = 0;
	int rc = EBO_OK;
	Sink sink;
	if (baseOut)
	{
		// the base time of compact records: the first event's time stamp of this call (a short serial look-ahead; blank
		// lines in front of it are skipped, a malformed line there is found again -- and reported -- by the walk itself)
		sink.base = 0;
		for (const char* q = begin + start; q < end;)
		{
			const char* le = line_end(q, end);
			ebo_event ev;
			const LineKind k = parse_line(q, le, ev);
			if (k == kEvent)
			{
				sink.base = ev.t_us;
			}
			if (k != kBlank)
			{
				break;
			}
			q = le < end ? le + 1 : end;
		}
		*baseOut = sink.base;
	}
	const char* pos = read_parallel(begin + start, end, out, cap, count, rc, threads ? threads : thread_budget(), sink, threadsUsed);
	*n = count;
	if (offset)
	{
		*offset = static_cast<uint64_t>(pos - begin);
	}
	return rc;
}
}  // namespace txt
}  // namespace ebo
