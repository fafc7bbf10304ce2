// ebo_windows.cpp — loading events into a context (include/ebo.h: ebo_set_window(s), _device,
// ebo_set_patches) and the per-window / per-patch queries: device bucketing (ebo_bucket.inc) by
// default, the host counting sort for A/B and for grids too fine for the device histogram.
#include "ebo_ctx.h"

using namespace ebo;

namespace
{
// the canonical order of a unit (ebo_internal.h: ranks ascending, dealt out with order_stride), host side
void canonical_order(uint64_t* p, uint32_t n)
{
	std::sort(p, p + n);
	const uint32_t st = order_stride(n);
	if (st == 1)
	{
		return;
	}
	std::vector<uint64_t> tmp(p, p + n);
	uint32_t r = 0;
	for (uint32_t q = 0; q < n; ++q)  // position q takes rank (q * st) mod n
	{
		p[q] = tmp[r];
		r += st;
		r = r >= n ? r - n : r;
	}
}
}  // namespace

extern "C" {

// Bucketing + packing on the device (ebo_bucket.inc).  d_raw: ebo_event[] on the device,
// offsets: host, absolute indices into d_raw.
// h_src != nullptr: the records are still on the host; they are uploaded in a few groups of
// windows on a second stream while the groups before them are bucketed (the upload is most of
// the set-up time: PCIe moves ~50 GB/s from page-locked memory, the bucketing kernels 6-10 G events/s).
// compact: 8-byte ebo_event8 records with per-window base times (host array t_base).
static int set_windows_on_device(ebo_ctx* c, const void* d_raw, const size_t* offsets, int n_windows, int compact = 0,
								 const int64_t* t_base = nullptr, const void* h_src = nullptr)
{
	const auto tStart = std::chrono::steady_clock::now();
	const size_t total = offsets[n_windows] - offsets[0];
	const int P = c->P;
	const size_t nUnits = static_cast<size_t>(n_windows) * (P + 1);
	size_t maxWin = 0;
	std::vector<unsigned long long> off64(n_windows + 1);
	for (int w = 0; w <= n_windows; ++w)
	{
		off64[w] = offsets[w];
		if (w && offsets[w] < offsets[w - 1])
		{
			return c->fail(EBO_ERR_ARG, "offsets must be non-decreasing");
		}
		if (w)
		{
			maxWin = std::max(maxWin, offsets[w] - offsets[w - 1]);
		}
	}
	(void)hipSetDevice(c->prm.device);
	// one scratch block: offsets | cnt | tmin | tmax | unit tref | window tref | flag | base times
	auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
	const size_t bOff = al((n_windows + 1) * 8), bCnt = al(nUnits * 4), bT = al(nUnits * 8), bW = al(n_windows * 8);
	const size_t need = bOff + bCnt + 3 * bT + bW + 256 + bW;
	if (need > c->bucket_cap)
	{
		if (c->d_bucket)
		{
			hipFree(c->d_bucket);
			c->d_bucket = nullptr;
			c->bucket_cap = 0;
		}
		int rc = c->hip(hipMalloc(&c->d_bucket, need), "hipMalloc bucket scratch");
		if (rc)
		{
			return rc;
		}
		c->bucket_cap = need;
	}
	char* base = static_cast<char*>(c->d_bucket);
	BucketLaunch L;
	L.d_raw = d_raw;
	L.d_offsets = reinterpret_cast<unsigned long long*>(base);
	L.d_cnt = reinterpret_cast<int*>(base + bOff);
	L.d_tmin = reinterpret_cast<long long*>(base + bOff + bCnt);
	L.d_tmax = reinterpret_cast<long long*>(base + bOff + bCnt + bT);
	L.d_unit_tref = reinterpret_cast<long long*>(base + bOff + bCnt + 2 * bT);
	L.d_win_tref = reinterpret_cast<long long*>(base + bOff + bCnt + 3 * bT);
	L.d_flag = reinterpret_cast<int*>(base + bOff + bCnt + 3 * bT + bW);
	L.compact = compact;
	L.d_tbase = reinterpret_cast<long long*>(base + bOff + bCnt + 3 * bT + bW + 256);
	L.n_windows = n_windows;
	L.P = P;
	// small inputs (one window of the reference configuration is 15 k events) get 256-event chunks: the
	// scatter is one WAVE per chunk, and eight waves are not a launch
	L.chunk_events = (offsets[n_windows] - offsets[0]) <= (static_cast<size_t>(1) << 19) ? 256 : 2048;
	L.max_chunks = static_cast<int>((maxWin + L.chunk_events - 1) / L.chunk_events);
	{
		// [windows][chunks][P + 1] counters of the stable scatter: events x (P + 1) / 512 bytes
		const size_t needHist = static_cast<size_t>(n_windows) * std::max(L.max_chunks, 1) * (P + 1);
		if (needHist > c->chunk_hist_cap)
		{
			if (c->d_chunk_hist)
			{
				hipFree(c->d_chunk_hist);
				c->d_chunk_hist = nullptr;
				c->chunk_hist_cap = 0;
			}
			int rch = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_chunk_hist), needHist * sizeof(unsigned int)), "hipMalloc chunk histograms");
			if (rch)
			{
				return rch;
			}
			c->chunk_hist_cap = needHist;
		}
		L.d_chunk_hist = c->d_chunk_hist;
	}
	L.min_events = c->prm.min_events;
	L.d_units = c->d_units;
	L.d_unit_maxdt = c->d_unit_maxdt;
	L.d_packed = c->d_events;
	L.c = make_consts(c);
	// pinned mirror: offsets go up and (units | unit tref, window tref, flag) come back as three
	// truly asynchronous copies and ONE synchronisation (copies from/to pageable memory are
	// staged one by one by the runtime: 0.176 -> 0.131 ms for a 15 k-event window)
	const size_t tail = bT + bW + 256;  // unit tref | window tref | flag, contiguous in the scratch block
	const size_t pUnits = al(nUnits * sizeof(Unit));
	const size_t pinNeed = bOff + pUnits + tail + bW;
	if (pinNeed > c->pin_bucket_cap)
	{
		if (c->pin_bucket)
		{
			(void)hipHostFree(c->pin_bucket);
			c->pin_bucket = nullptr;
			c->pin_bucket_cap = 0;
		}
		int rcp = c->hip(hipHostMalloc(&c->pin_bucket, pinNeed, hipHostMallocDefault), "hipHostMalloc bucket mirror");
		if (rcp)
		{
			return rcp;
		}
		c->pin_bucket_cap = pinNeed;
	}
	char* pin = static_cast<char*>(c->pin_bucket);
	std::memcpy(pin, off64.data(), off64.size() * 8);
	int rc = c->hip(hipMemcpyAsync(const_cast<unsigned long long*>(L.d_offsets), pin, off64.size() * 8,
								   hipMemcpyHostToDevice, c->stream),
					"H2D offsets");
	if (rc)
	{
		return rc;
	}
	if (compact)
	{
		char* pinBase = pin + bOff + pUnits + tail;
		std::memcpy(pinBase, t_base, static_cast<size_t>(n_windows) * 8);
		rc = c->hip(hipMemcpyAsync(const_cast<long long*>(L.d_tbase), pinBase, static_cast<size_t>(n_windows) * 8,
								   hipMemcpyHostToDevice, c->stream),
					"H2D base times");
		if (rc)
		{
			return rc;
		}
	}
	const size_t recBytes = compact ? 8 : sizeof(ebo_event);
	if (!h_src || total == 0)
	{
		if (launch_bucket(L, c->stream))
		{
			return c->hip(hipGetLastError(), "bucket launch");
		}
	}
	else
	{
		// groups of whole windows, each at least ~8 MB
		if (!c->copy_stream)
		{
			rc = c->hip(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking), "copy stream");
			for (int g = 0; g < 8 && rc == EBO_OK; ++g)
			{
				rc = c->hip(hipEventCreateWithFlags(&c->copy_done[g], hipEventDisableTiming), "copy event");
			}
			if (rc)
			{
				return rc;
			}
		}
		// three groups: 12.8 M compact events take 2.63 / 2.40 / 2.34 / 2.37 / 2.88 ms in 1 / 2 / 3 / 4 / 8 groups
		// (every group is five launches and two event hand-offs; a plain copy of the bytes 1.79 ms)
		const size_t wantGroups = std::min<size_t>(std::max<size_t>(ab_size("EBO_INGEST_GROUPS", 3), 1), 7);
		const size_t perGroup = std::max<size_t>((total * recBytes + wantGroups - 1) / wantGroups, static_cast<size_t>(8) << 20);
		int w0 = 0, g = 0;
		// the upload may not overtake earlier work of the context's stream that still reads d_raw
		hipError_t he = hipEventRecord(c->copy_done[7], c->stream);
		if (he == hipSuccess) he = hipStreamWaitEvent(c->copy_stream, c->copy_done[7], 0);
		while (w0 < n_windows && he == hipSuccess)
		{
			int w1 = w0 + 1;
			while (w1 < n_windows && (g == 6 ? false : (offsets[w1] - offsets[w0]) * recBytes < perGroup))
			{
				++w1;
			}
			if (g == 6)
			{
				w1 = n_windows;  // the last group takes the rest
			}
			const size_t b0 = (offsets[w0] - offsets[0]) * recBytes, b1 = (offsets[w1] - offsets[0]) * recBytes;
			if (b1 > b0)
			{
				he = hipMemcpyAsync(static_cast<char*>(const_cast<void*>(d_raw)) + b0, static_cast<const char*>(h_src) + b0, b1 - b0,
									hipMemcpyHostToDevice, c->copy_stream);
			}
			if (he == hipSuccess) he = hipEventRecord(c->copy_done[g], c->copy_stream);
			if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, c->copy_done[g], 0);
			if (he != hipSuccess)
			{
				break;
			}
			L.w0 = w0;
			L.w1 = w1;
			if (launch_bucket(L, c->stream))
			{
				return c->hip(hipGetLastError(), "bucket launch");
			}
			w0 = w1;
			++g;
		}
		if (he != hipSuccess)
		{
			c->n_windows = 0;
			return c->hip(he, "pipelined upload");
		}
	}
	const bool trace = std::getenv("EBO_INGEST_TRACE") != nullptr;
	const auto tIssued = std::chrono::steady_clock::now();
	hipError_t e = hipMemcpyAsync(pin + bOff, c->d_units, nUnits * sizeof(Unit), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(pin + bOff + pUnits, L.d_unit_tref, tail, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (trace)
	{
		const auto tDone = std::chrono::steady_clock::now();
		std::fprintf(stderr, "[ebo ingest] issue %.3f ms, wait %.3f ms\n",
					 std::chrono::duration<double, std::milli>(tIssued - tStart).count(),
					 std::chrono::duration<double, std::milli>(tDone - tIssued).count());
	}
	if (e != hipSuccess)
	{
		c->n_windows = 0;
		return c->hip(e, "bucket results");
	}
	std::vector<Unit> units(nUnits);
	std::vector<int64_t> utref(nUnits);
	std::vector<long long> wtref(n_windows);
	int flag = 0;
	std::memcpy(units.data(), pin + bOff, nUnits * sizeof(Unit));
	std::memcpy(utref.data(), pin + bOff + pUnits, nUnits * 8);
	std::memcpy(wtref.data(), pin + bOff + pUnits + bT, n_windows * 8);
	std::memcpy(&flag, pin + bOff + pUnits + bT + bW, sizeof(int));
	if (flag)
	{
		c->n_windows = 0;
		c->custom_n = 0;
		return c->fail(EBO_ERR_RANGE,
					   (flag & 1)	? "event coordinate outside [-16384,16383]"
					   : (flag & 2) ? "mid-time outside int32 microseconds (undefined in the reference)"
									: "event time further than 2^31 us from the reference time");
	}
	std::vector<WindowInfo> wins(n_windows);
	for (int w = 0; w < n_windows; ++w)
	{
		wins[w].t_ref = wtref[w];
		wins[w].n_events = offsets[w + 1] - offsets[w];
	}
	(void)total;
	c->units.swap(units);
	++c->units_gen;
	c->unit_tref.swap(utref);
	c->windows.swap(wins);
	c->n_windows = n_windows;
	c->custom_n = 0;
	c->max_rw = c->grid_max_rw;
	c->max_rh = c->grid_max_rh;
	c->reg_rw = c->prm.patch_w;
	c->reg_rh = c->prm.patch_h;
	return EBO_OK;
}

// k_bucket_count's per-chunk histogram (count + min/max time per bucket) has to fit a
// workgroup's LDS; finer grids are bucketed on the host.
static bool device_bucketing_fits(const ebo_ctx* c)
{
	return static_cast<size_t>(c->P + 1) * (2 * sizeof(long long) + sizeof(int)) + 8 <= kLdsBudget - 1024;
}

int ebo_set_windows_device(ebo_ctx* c, const ebo_event* d_ev, const size_t* offsets, int n_windows)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!d_ev || !offsets || n_windows <= 0)
	{
		return c->fail(EBO_ERR_ARG, "null events/offsets or no window");
	}
	if (n_windows > c->cap_windows)
	{
		return c->fail(EBO_ERR_ARG, "more windows than max_windows");
	}
	if (offsets[n_windows] - offsets[0] > c->cap_events)
	{
		return c->fail(EBO_ERR_ARG, "more events than max_events");
	}
	if (!device_bucketing_fits(c))
	{
		return c->fail(EBO_ERR_UNSUPPORTED,
					   "device bucketing keeps one histogram slot per patch in LDS (about 8000 patches); "
					   "pass host events to ebo_set_windows for finer grids");
	}
	return set_windows_on_device(c, d_ev, offsets, n_windows);
}

static int set_windows_host(ebo_ctx* c, const ebo_event* ev, const size_t* offsets, int n_windows);

int ebo_set_windows(ebo_ctx* c, const ebo_event* ev, const size_t* offsets, int n_windows)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!offsets || n_windows <= 0 || (!ev && offsets[n_windows] > offsets[0]))
	{
		return c->fail(EBO_ERR_ARG, "null events/offsets or no window");
	}
	if (n_windows > c->cap_windows)
	{
		return c->fail(EBO_ERR_ARG, "more windows than max_windows");
	}
	const size_t total = offsets[n_windows] - offsets[0];
	if (total > c->cap_events)
	{
		return c->fail(EBO_ERR_ARG, "more events than max_events");
	}
	const char* mode = ab_env("EBO_BUCKET");
	if ((mode && std::strcmp(mode, "host") == 0) || !device_bucketing_fits(c))
	{
		return set_windows_host(c, ev, offsets, n_windows);
	}
	// default: raw events go to the device once; bucketing and packing happen there
	(void)hipSetDevice(c->prm.device);
	if (!c->d_raw)
	{
		int rc = c->hip(hipMalloc(&c->d_raw, c->cap_events * sizeof(ebo_event)), "hipMalloc raw events");
		if (rc)
		{
			return rc;
		}
	}
	std::vector<size_t> rel(n_windows + 1);
	for (int w = 0; w <= n_windows; ++w)
	{
		rel[w] = offsets[w] - offsets[0];
	}
	return set_windows_on_device(c, c->d_raw, rel.data(), n_windows, 0, nullptr, total > 0 ? ev + offsets[0] : nullptr);
}

// ---- compact 8-byte input (ebo_event8) -----------------------------------------------------
int ebo_pack_events8(const ebo_event* ev, size_t n, int64_t t_base, ebo_event8* out)
{
	if ((n && !ev) || (n && !out))
	{
		return EBO_ERR_ARG;
	}
	for (size_t i = 0; i < n; ++i)
	{
		const int64_t dt = ev[i].t_us - t_base;
		if (ev[i].x < kCoordMin || ev[i].x > kCoordMax || ev[i].y < kCoordMin || ev[i].y > kCoordMax ||
			dt < INT32_MIN || dt > INT32_MAX)
		{
			return EBO_ERR_RANGE;
		}
		out[i].xy = pack_lo(ev[i].x, ev[i].y, ev[i].sign > 0);
		out[i].t_rel_us = static_cast<int32_t>(dt);
	}
	return EBO_OK;
}

static int check_windows8(ebo_ctx* c, const void* ev, const int64_t* t_base, const size_t* offsets, int n_windows)
{
	if (!offsets || !t_base || n_windows <= 0 || (!ev && offsets[n_windows] > offsets[0]))
	{
		return c->fail(EBO_ERR_ARG, "null events/base times/offsets or no window");
	}
	if (n_windows > c->cap_windows)
	{
		return c->fail(EBO_ERR_ARG, "more windows than max_windows");
	}
	if (offsets[n_windows] - offsets[0] > c->cap_events)
	{
		return c->fail(EBO_ERR_ARG, "more events than max_events");
	}
	if (!device_bucketing_fits(c))
	{
		return c->fail(EBO_ERR_UNSUPPORTED,
					   "device bucketing keeps one histogram slot per patch in LDS (about 8000 patches); "
					   "pass 24-byte host events to ebo_set_windows for finer grids");
	}
	return EBO_OK;
}

int ebo_set_windows8_device(ebo_ctx* c, const ebo_event8* d_ev, const int64_t* t_base, const size_t* offsets, int n_windows)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	int rc = check_windows8(c, d_ev, t_base, offsets, n_windows);
	if (rc)
	{
		return rc;
	}
	return set_windows_on_device(c, d_ev, offsets, n_windows, 1, t_base, nullptr);
}

int ebo_set_windows8(ebo_ctx* c, const ebo_event8* ev, const int64_t* t_base, const size_t* offsets, int n_windows)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	int rc = check_windows8(c, ev, t_base, offsets, n_windows);
	if (rc)
	{
		return rc;
	}
	(void)hipSetDevice(c->prm.device);
	if (!c->d_raw)
	{
		rc = c->hip(hipMalloc(&c->d_raw, c->cap_events * sizeof(ebo_event)), "hipMalloc raw events");
		if (rc)
		{
			return rc;
		}
	}
	const size_t total = offsets[n_windows] - offsets[0];
	std::vector<size_t> rel(n_windows + 1);
	for (int w = 0; w <= n_windows; ++w)
	{
		rel[w] = offsets[w] - offsets[0];
	}
	return set_windows_on_device(c, c->d_raw, rel.data(), n_windows, 1, t_base, total > 0 ? ev + offsets[0] : nullptr);
}

static int set_windows_host(ebo_ctx* c, const ebo_event* ev, const size_t* offsets, int n_windows)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (!offsets || n_windows <= 0 || (!ev && offsets[n_windows] > offsets[0]))
	{
		return c->fail(EBO_ERR_ARG, "null events/offsets or no window");
	}
	if (n_windows > c->cap_windows)
	{
		return c->fail(EBO_ERR_ARG, "more windows than max_windows");
	}
	const size_t total = offsets[n_windows] - offsets[0];
	if (total > c->cap_events)
	{
		return c->fail(EBO_ERR_ARG, "more events than max_events");
	}
	const int P = c->P;
	const int pw = c->prm.patch_w, ph = c->prm.patch_h;
	std::vector<Unit> units(static_cast<size_t>(n_windows) * (P + 1));
	std::vector<int64_t> utref(units.size(), 0);
	std::vector<WindowInfo> wins(n_windows);
	c->h_packed.resize(total);
	std::vector<uint32_t> cnt(P + 1), cur(P + 1);
	std::vector<int64_t> first(P + 1), last(P + 1), tlo(P + 1), thi(P + 1);
	std::vector<int32_t> maxdt(units.size(), 0);
	size_t base = 0;
	for (int w = 0; w < n_windows; ++w)
	{
		if (offsets[w + 1] < offsets[w])
		{
			return c->fail(EBO_ERR_ARG, "offsets must be non-decreasing");
		}
		const ebo_event* we = ev + offsets[w];
		const size_t n = offsets[w + 1] - offsets[w];
		int64_t tw = 0;
		if (n > 0 && !mid_timestamp(we[0].t_us, we[n - 1].t_us, tw))
		{
			return c->fail(EBO_ERR_RANGE, "window mid-time outside int32 microseconds (undefined in the reference)");
		}
		wins[w].t_ref = tw;
		wins[w].n_events = n;
		std::fill(cnt.begin(), cnt.end(), 0u);
		// patch of an event == the grid rect that contains it (feature_detector.cpp:332-355)
		auto bucket_of = [&](const ebo_event& e) -> int {
			if (e.x < 0 || e.x >= c->prm.image_w || e.y < 0 || e.y >= c->prm.image_h)
			{
				return P;
			}
			const int bx = std::min(e.x / pw, c->npx - 1);
			const int by = std::min(e.y / ph, c->npy - 1);
			return by * c->npx + bx;
		};
		for (size_t i = 0; i < n; ++i)
		{
			if (we[i].x < kCoordMin || we[i].x > kCoordMax || we[i].y < kCoordMin || we[i].y > kCoordMax)
			{
				return c->fail(EBO_ERR_RANGE, "event coordinate outside [-16384,16383]");
			}
			const int b = bucket_of(we[i]);
			if (cnt[b] == 0)
			{
				first[b] = we[i].t_us;
				tlo[b] = thi[b] = we[i].t_us;
			}
			last[b] = we[i].t_us;
			tlo[b] = std::min(tlo[b], we[i].t_us);
			thi[b] = std::max(thi[b], we[i].t_us);
			cnt[b]++;
		}
		size_t off = base;
		for (int b = 0; b <= P; ++b)
		{
			Unit& u = units[static_cast<size_t>(w) * (P + 1) + b];
			u.ev_off = static_cast<uint32_t>(off);
			u.n_ev = cnt[b];
			u.flags = 0;
			u.flow_idx = static_cast<uint32_t>(static_cast<size_t>(w) * P + std::min(b, P - 1));
			int64_t tu = tw;
			if (b < P)
			{
				int x, y, rw, rh;
				rect_of(c, b % c->npx, b / c->npx, x, y, rw, rh);
				u.rx = static_cast<int16_t>(x);
				u.ry = static_cast<int16_t>(y);
				u.rw = static_cast<int16_t>(rw);
				u.rh = static_cast<int16_t>(rh);
				if (cnt[b] > 0 && !mid_timestamp(first[b], last[b], tu))
				{
					return c->fail(EBO_ERR_RANGE, "patch mid-time outside int32 microseconds");
				}
				if (cnt[b] > c->prm.min_events)  // feature_detector.cpp:357, strictly greater
				{
					u.flags |= kUnitActive;
				}
			}
			else
			{
				u.rx = u.ry = 0;
				u.rw = u.rh = 1;
				u.flags |= kUnitStray;
			}
			utref[static_cast<size_t>(w) * (P + 1) + b] = tu;
			if (cnt[b] > 0)
			{
				const int64_t a = std::llabs(tw - tlo[b]), z = std::llabs(tw - thi[b]);
				maxdt[static_cast<size_t>(w) * (P + 1) + b] = static_cast<int32_t>(std::min<int64_t>(std::max(a, z), INT32_MAX));
			}
			const int64_t dwin = tw - tu;
			if (dwin < INT32_MIN || dwin > INT32_MAX)
			{
				return c->fail(EBO_ERR_RANGE, "time span exceeds int32 microseconds");
			}
			u.dt_win = static_cast<int32_t>(dwin);
			cur[b] = static_cast<uint32_t>(off - base);
			off += cnt[b];
		}
		for (size_t i = 0; i < n; ++i)
		{
			const int b = bucket_of(we[i]);
			const int64_t tu = utref[static_cast<size_t>(w) * (P + 1) + b];
			const int64_t dt = tu - we[i].t_us;
			const int64_t dtw = tw - we[i].t_us;
			if (dt < INT32_MIN || dt > INT32_MAX || dtw < INT32_MIN || dtw > INT32_MAX)
			{
				return c->fail(EBO_ERR_RANGE, "event time further than 2^31 us from the reference time");
			}
			const uint64_t rec = static_cast<uint64_t>(pack_lo(we[i].x, we[i].y, we[i].sign > 0)) |
								 (static_cast<uint64_t>(static_cast<uint32_t>(static_cast<int32_t>(dt))) << 32);
			c->h_packed[base + cur[b]++] = rec;
		}
		// canonical order inside a unit (as k_bucket_sort): both bucketing paths then
		// hand identical arrays to the kernels
		for (int b = 0; b <= P; ++b)
		{
			const Unit& u = units[static_cast<size_t>(w) * (P + 1) + b];
			if (u.n_ev >= 2 && u.n_ev <= 8192 && !ab_env("EBO_KEEP_ORDER"))  // (A/B build: list order, for the reference-order diagnostic)
			{
				canonical_order(c->h_packed.data() + u.ev_off, u.n_ev);
			}
		}
		base += n;
	}
	(void)hipSetDevice(c->prm.device);
	int rc = EBO_OK;
	if (total > 0)
	{
		rc = c->hip(hipMemcpyAsync(c->d_events, c->h_packed.data(), total * sizeof(uint64_t),
								   hipMemcpyHostToDevice, c->stream),
					"H2D events");
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipMemcpyAsync(c->d_units, units.data(), units.size() * sizeof(Unit),
								   hipMemcpyHostToDevice, c->stream),
					"H2D units");
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipMemcpyAsync(c->d_unit_maxdt, maxdt.data(), maxdt.size() * sizeof(int32_t),
								   hipMemcpyHostToDevice, c->stream),
					"H2D unit time spans");
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipStreamSynchronize(c->stream), "sync after upload");
	}
	if (rc != EBO_OK)
	{
		c->n_windows = 0;
		return rc;
	}
	c->units.swap(units);
	++c->units_gen;
	c->unit_tref.swap(utref);
	c->windows.swap(wins);
	c->n_windows = n_windows;
	c->custom_n = 0;
	c->max_rw = c->grid_max_rw;
	c->max_rh = c->grid_max_rh;
	c->reg_rw = c->prm.patch_w;
	c->reg_rh = c->prm.patch_h;
	return EBO_OK;
}

int ebo_set_patches(ebo_ctx* c, const ebo_event* ev, const size_t* offsets, const int32_t* rects,
					int n_patches)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!ev || !offsets || !rects || n_patches <= 0)
	{
		return c->fail(EBO_ERR_ARG, "null argument or no patch");
	}
	if (static_cast<size_t>(n_patches) > static_cast<size_t>(c->cap_windows) * c->P)
	{
		return c->fail(EBO_ERR_ARG, "more patches than max_windows * grid patches");
	}
	const size_t total = offsets[n_patches] - offsets[0];
	if (total > c->cap_events)
	{
		return c->fail(EBO_ERR_ARG, "more events than max_events");
	}
	std::vector<Unit> units(n_patches);
	std::vector<int64_t> utref(n_patches, 0), utmin(n_patches, 0), utmax(n_patches, 0);
	std::vector<int16_t> ubox(4 * static_cast<size_t>(n_patches), 0);
	c->h_packed.resize(total);
	int mrw = 0, mrh = 0, lrw = 0x7fffffff, lrh = 0x7fffffff;
	size_t base = 0;
	for (int p = 0; p < n_patches; ++p)
	{
		if (offsets[p + 1] < offsets[p])
		{
			return c->fail(EBO_ERR_ARG, "offsets must be non-decreasing");
		}
		const ebo_event* pe = ev + offsets[p];
		const size_t n = offsets[p + 1] - offsets[p];
		const int32_t* r = rects + 4 * p;
		if (r[2] <= 0 || r[3] <= 0 || r[2] > 10922 || r[3] > 10922 || r[0] < kCoordMin ||
			r[0] > kCoordMax || r[1] < kCoordMin || r[1] > kCoordMax)
		{
			return c->fail(EBO_ERR_RANGE, "patch rect outside the packed range");
		}
		Unit& u = units[p];
		u.ev_off = static_cast<uint32_t>(base);
		u.n_ev = static_cast<uint32_t>(n);
		u.rx = static_cast<int16_t>(r[0]);
		u.ry = static_cast<int16_t>(r[1]);
		u.rw = static_cast<int16_t>(r[2]);
		u.rh = static_cast<int16_t>(r[3]);
		u.dt_win = 0;
		u.flags = (n > c->prm.min_events) ? kUnitActive : 0u;
		u.flow_idx = static_cast<uint32_t>(p);
		mrw = std::max(mrw, r[2]);
		mrh = std::max(mrh, r[3]);
		lrw = std::min(lrw, r[2]);
		lrh = std::min(lrh, r[3]);
		int64_t tu = 0;
		if (n > 0 && !mid_timestamp(pe[0].t_us, pe[n - 1].t_us, tu))  // contrast_functor.h:18-20
		{
			return c->fail(EBO_ERR_RANGE, "patch mid-time outside int32 microseconds");
		}
		utref[p] = tu;
		int64_t tlo = n ? pe[0].t_us : 0, thi = tlo;
		int32_t bx0 = n ? pe[0].x : 0, bx1 = bx0, by0 = n ? pe[0].y : 0, by1 = by0;
		for (size_t i = 0; i < n; ++i)
		{
			tlo = std::min<int64_t>(tlo, pe[i].t_us);
			thi = std::max<int64_t>(thi, pe[i].t_us);
			bx0 = std::min(bx0, pe[i].x);
			bx1 = std::max(bx1, pe[i].x);
			by0 = std::min(by0, pe[i].y);
			by1 = std::max(by1, pe[i].y);
			if (pe[i].x < kCoordMin || pe[i].x > kCoordMax || pe[i].y < kCoordMin || pe[i].y > kCoordMax)
			{
				return c->fail(EBO_ERR_RANGE, "event coordinate outside [-16384,16383]");
			}
			const int64_t dt = tu - pe[i].t_us;
			if (dt < INT32_MIN || dt > INT32_MAX)
			{
				return c->fail(EBO_ERR_RANGE, "event time further than 2^31 us from the reference time");
			}
			c->h_packed[base + i] =
				static_cast<uint64_t>(pack_lo(pe[i].x, pe[i].y, pe[i].sign > 0)) |
				(static_cast<uint64_t>(static_cast<uint32_t>(static_cast<int32_t>(dt))) << 32);
		}
		// the canonical order of a unit (ascending packed record; units above 8192 events stay in
		// list order), as the window paths leave it: a patch loaded here evaluates to the same bits as
		// the same patch inside a window
		if (n >= 2 && n <= 8192 && !ab_env("EBO_KEEP_ORDER"))  // (A/B build: the list order, for the reference-order diagnostic)
		{
			canonical_order(c->h_packed.data() + base, static_cast<uint32_t>(n));
		}
		utmin[p] = tlo;
		utmax[p] = thi;
		// (coordinates were range-checked above: they fit int16)
		ubox[4 * static_cast<size_t>(p) + 0] = static_cast<int16_t>(bx0);
		ubox[4 * static_cast<size_t>(p) + 1] = static_cast<int16_t>(bx1);
		ubox[4 * static_cast<size_t>(p) + 2] = static_cast<int16_t>(by0);
		ubox[4 * static_cast<size_t>(p) + 3] = static_cast<int16_t>(by1);
		base += n;
	}
	(void)hipSetDevice(c->prm.device);
	int rc = EBO_OK;
	if (total > 0)
	{
		rc = c->hip(hipMemcpyAsync(c->d_events, c->h_packed.data(), total * sizeof(uint64_t),
								   hipMemcpyHostToDevice, c->stream),
					"H2D events");
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipMemcpyAsync(c->d_units, units.data(), units.size() * sizeof(Unit),
								   hipMemcpyHostToDevice, c->stream),
					"H2D units");
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipStreamSynchronize(c->stream), "sync after upload");
	}
	if (rc != EBO_OK)
	{
		c->n_windows = 0;
		c->custom_n = 0;
		return rc;
	}
	c->units.swap(units);
	c->unit_tref.swap(utref);
	c->unit_tmin.swap(utmin);
	c->unit_tmax.swap(utmax);
	c->unit_box.swap(ubox);
	++c->units_gen;
	c->windows.assign(1, WindowInfo{0, total});
	c->n_windows = 1;
	c->custom_n = n_patches;
	c->max_rw = mrw;
	c->max_rh = mrh;
	c->reg_rw = std::min(lrw, mrw);
	c->reg_rh = std::min(lrh, mrh);
	return EBO_OK;
}

int ebo_set_window(ebo_ctx* c, const ebo_event* ev, size_t n)
{
	const size_t offsets[2] = {0, n};
	return ebo_set_windows(c, ev, offsets, 1);
}

int ebo_num_windows(const ebo_ctx* c, int* n)
{
	if (!c || !n)
	{
		return EBO_ERR_ARG;
	}
	*n = c->n_windows;
	return EBO_OK;
}

int ebo_window_info(const ebo_ctx* c, int window, int64_t* t_ref_us, uint64_t* n_events)
{
	if (!c || window < 0 || window >= c->n_windows)
	{
		return EBO_ERR_ARG;
	}
	if (t_ref_us) *t_ref_us = c->windows[window].t_ref;
	if (n_events) *n_events = c->windows[window].n_events;
	return EBO_OK;
}

int ebo_patch_info(const ebo_ctx* c, int window, int patch, int32_t* n_events, int32_t* active,
				   int64_t* t_ref_us)
{
	if (!c || window < 0 || window >= c->n_windows || patch < 0 || patch >= c->cur_patches())
	{
		return EBO_ERR_ARG;
	}
	const size_t i = c->unit_index(window, patch);
	if (n_events) *n_events = static_cast<int32_t>(c->units[i].n_ev);
	if (active) *active = (c->units[i].flags & kUnitActive) ? 1 : 0;
	if (t_ref_us) *t_ref_us = c->unit_tref[i];
	return EBO_OK;
}

}  // extern "C"
