// ebo_io.cpp — DAVIS events.txt reader and the packed binary sidecar (include/ebo.h).
#include "ebo_ctx.h"
#include "txt_events.h"

using namespace ebo;

extern "C" {

// ---- packed binary sidecar of an events.txt -------------------------------
// Parsing the text format costs tens of nanoseconds per event and host thread (csrc/txt_events.h): at device rates
// the recording, not the GPU, is the bottleneck.  The sidecar stores what Davis240cReader::getEventSample produces
// from each line -- the microsecond timestamp AFTER the double -> int64 truncation, x, y, sign --
// so reading it back gives bit-identical events with no parsing.
//   header (32 B): magic "EBOEVT1\0", uint64 n_events, uint32 record_bytes (= 16), uint32 flags (0),
//                  uint64 reserved
//   record (16 B): int64 t_us, int16 x, int16 y, int8 sign (-1 / +1), 3 bytes 0       little endian
namespace
{
const char kBinMagic[8] = {'E', 'B', 'O', 'E', 'V', 'T', '1', '\0'};
struct BinHeader
{
	char magic[8];
	uint64_t n;
	uint32_t record_bytes;
	uint32_t flags;
	uint64_t reserved;
};
struct BinRecord
{
	int64_t t_us;
	int16_t x, y;
	int8_t sign;
	uint8_t pad[3];
};
static_assert(sizeof(BinHeader) == 32 && sizeof(BinRecord) == 16, "sidecar layout");
}  // namespace

int ebo_write_events_bin(const char* path, const ebo_event* ev, size_t n)
{
	if (!path || (n && !ev))
	{
		return EBO_ERR_ARG;
	}
	for (size_t i = 0; i < n; ++i)
	{
		if (ev[i].x < -32768 || ev[i].x > 32767 || ev[i].y < -32768 || ev[i].y > 32767 ||
			(ev[i].sign != 1 && ev[i].sign != -1))
		{
			return EBO_ERR_RANGE;
		}
	}
	FILE* fp = std::fopen(path, "wb");
	if (!fp)
	{
		return EBO_ERR_ARG;
	}
	BinHeader h;
	std::memset(&h, 0, sizeof(h));
	std::memcpy(h.magic, kBinMagic, 8);
	h.n = n;
	h.record_bytes = sizeof(BinRecord);
	bool ok = std::fwrite(&h, sizeof(h), 1, fp) == 1;
	std::vector<BinRecord> buf(1 << 16);
	for (size_t i = 0; i < n && ok; i += buf.size())
	{
		const size_t m = std::min(buf.size(), n - i);
		for (size_t k = 0; k < m; ++k)
		{
			BinRecord& r = buf[k];
			r.t_us = ev[i + k].t_us;
			r.x = static_cast<int16_t>(ev[i + k].x);
			r.y = static_cast<int16_t>(ev[i + k].y);
			r.sign = static_cast<int8_t>(ev[i + k].sign);
			r.pad[0] = r.pad[1] = r.pad[2] = 0;
		}
		ok = std::fwrite(buf.data(), sizeof(BinRecord), m, fp) == m;
	}
	ok = (std::fclose(fp) == 0) && ok;
	return ok ? EBO_OK : EBO_ERR_ARG;
}

int ebo_read_events_bin(const char* path, ebo_event* out, size_t cap, size_t* n)
{
	if (!path || !n || (cap && !out))
	{
		return EBO_ERR_ARG;
	}
	*n = 0;
	FILE* fp = std::fopen(path, "rb");
	if (!fp)
	{
		return EBO_ERR_ARG;
	}
	BinHeader h;
	if (std::fread(&h, sizeof(h), 1, fp) != 1 || std::memcmp(h.magic, kBinMagic, 8) != 0 ||
		h.record_bytes != sizeof(BinRecord) || h.flags != 0)
	{
		std::fclose(fp);
		return EBO_ERR_RANGE;
	}
	const size_t want = static_cast<size_t>(std::min<uint64_t>(h.n, cap));
	std::vector<BinRecord> buf(1 << 16);
	size_t count = 0;
	int rc = EBO_OK;
	while (count < want)
	{
		const size_t m = std::min(buf.size(), want - count);
		if (std::fread(buf.data(), sizeof(BinRecord), m, fp) != m)
		{
			rc = EBO_ERR_RANGE;  // shorter than its header says
			break;
		}
		for (size_t k = 0; k < m; ++k)
		{
			const BinRecord& r = buf[k];
			if (r.sign != 1 && r.sign != -1)
			{
				rc = EBO_ERR_RANGE;
				break;
			}
			ebo_event& e = out[count++];
			e.x = r.x;
			e.y = r.y;
			e.sign = r.sign;
			e.reserved = 0;
			e.t_us = r.t_us;
		}
		if (rc)
		{
			break;
		}
	}
	std::fclose(fp);
	*n = count;
	return rc;
}

// DAVIS240C events.txt (tools/dataset_reader/src/davis240c_reader.cpp:60-92): one event per line
// "<seconds> <x> <y> <0|1>".  The reader itself -- mapped file, one chunk per host thread, the fast and the strtod path
// of a line -- is csrc/txt_events.h (HIP-free, tested on the CPU under ThreadSanitizer).
int ebo_read_events_txt(const char* path, ebo_event* out, size_t cap, size_t* n)
{
	return ebo::txt::read_events_file(path, nullptr, out, cap, n, 0);
}

int ebo_read_events_txt_at(const char* path, uint64_t* offset, ebo_event* out, size_t cap, size_t* n)
{
	if (!offset)
	{
		return EBO_ERR_ARG;
	}
	return ebo::txt::read_events_file(path, offset, out, cap, n, 0);
}

int ebo_read_events_txt8(const char* path, uint64_t* offset, ebo_event8* out, size_t cap, size_t* n, int64_t* t_base, int threads)
{
	if (!t_base || threads < 0)
	{
		return EBO_ERR_ARG;
	}
	return ebo::txt::read_events_file<ebo::txt::Sink8>(path, offset, out, cap, n, static_cast<unsigned>(threads), nullptr, t_base);
}

int ebo_read_events_txt_threads(const char* path, uint64_t* offset, ebo_event* out, size_t cap, size_t* n, int threads,
								int* threads_used)
{
	if (threads < 0)
	{
		return EBO_ERR_ARG;
	}
	unsigned used = 1;
	const int rc = ebo::txt::read_events_file(path, offset, out, cap, n, static_cast<unsigned>(threads), &used);
	if (threads_used)
	{
		*threads_used = static_cast<int>(used);
	}
	return rc;
}

// trajectory.txt of tools::Evaluator::saveFeaturesTrajectory (tools/evaluator/src/evaluator.cpp:125-150):
// `trajFile << std::fixed << std::setprecision(8) << id << " " << duration<double>(ts).count()
//  << " " << x << " " << y << std::endl` per trajectory point, patch by patch.  duration<double>
// of microseconds is count / 1e6 (one division); "%.8f" is what std::fixed + setprecision(8) prints.
int ebo_write_tracks_txt(const char* path, const ebo_track_point* pts, size_t n)
{
	if (!path || (n && !pts))
	{
		return EBO_ERR_ARG;
	}
	FILE* fp = std::fopen(path, "wb");
	if (!fp)
	{
		return EBO_ERR_ARG;
	}
	bool ok = true;
	for (size_t i = 0; i < n && ok; ++i)
	{
		const double sec = static_cast<double>(pts[i].t_us) / 1000000.0;
		ok = std::fprintf(fp, "%lld %.8f %.8f %.8f\n", static_cast<long long>(pts[i].id), sec, pts[i].x, pts[i].y) > 0;
	}
	ok = (std::fclose(fp) == 0) && ok;
	return ok ? EBO_OK : EBO_ERR_ARG;
}

int ebo_read_tracks_txt(const char* path, ebo_track_point* out, size_t cap, size_t* n)
{
	if (!path || !n || (cap && !out))
	{
		return EBO_ERR_ARG;
	}
	*n = 0;
	FILE* fp = std::fopen(path, "rb");
	if (!fp)
	{
		return EBO_ERR_ARG;
	}
	char line[512];
	size_t count = 0;
	int rc = EBO_OK;
	size_t total = 0;  // records in the file, also those beyond cap
	while (std::fgets(line, sizeof(line), fp))
	{
		char* end = nullptr;
		const long long id = std::strtoll(line, &end, 10);
		if (end == line)
		{
			const char* s = line;
			while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n')
			{
				++s;
			}
			if (*s == '\0')
			{
				continue;  // blank line
			}
			rc = EBO_ERR_RANGE;
			break;
		}
		double v[3];
		bool bad = false;
		for (int k = 0; k < 3; ++k)
		{
			const char* p = end;
			v[k] = std::strtod(p, &end);
			if (end == p)
			{
				bad = true;
				break;
			}
		}
		if (bad)
		{
			rc = EBO_ERR_RANGE;
			break;
		}
		++total;
		if (count >= cap)
		{
			continue;  // counted, not stored: the caller learns how large a buffer the file needs
		}
		ebo_track_point& t = out[count++];
		t.id = id;
		t.t_us = static_cast<int64_t>(std::llround(v[0] * 1000000.0));
		t.x = v[1];
		t.y = v[2];
	}
	std::fclose(fp);
	if (rc == EBO_OK && total > cap)
	{
		*n = total;  // never a silently truncated list: out holds the first cap records, *n says what is needed
		return EBO_ERR_ARG;
	}
	*n = count;
	return rc;
}

}  // extern "C"
