// ebo_band.cpp — host side of the band-limited final image of row-sharded windows (include/ebo.h: ebo_band_plan,
// ebo_count_image_band_device, ebo_band_finish_device; the RCCL side, ebo_band_exchange_device and
// ebo_band_gather_device, is in ebo_comm.cpp) and the per-unit table the shard images share.
#include "ebo_ctx.h"

namespace ebo_host
{
// The table k_count_band / k_count_shard read per unit of a row shard, built for the windows' reference times and
// kept on the device until the units or those times change (one upload per set of windows, not one per step).
int shard_table(ebo_ctx* c, int n_windows, const int64_t* window_t_ref_us, const BandUnit** out, bool* uniformFlows)
{
	if (!c->custom_n)
	{
		return c->fail(EBO_ERR_STATE, "the shard images need the units of a shard (ebo_set_patches)");
	}
	if (n_windows <= 0 || c->custom_n % n_windows != 0)
	{
		return c->fail(EBO_ERR_ARG, "the loaded units are not n_windows equal groups");
	}
	const size_t n = static_cast<size_t>(c->custom_n);
	const bool cached = c->shard_tbl_gen == c->units_gen && c->shard_tbl_tref.size() == static_cast<size_t>(n_windows) &&
						std::equal(c->shard_tbl_tref.begin(), c->shard_tbl_tref.end(), window_t_ref_us);
	if (!cached)
	{
		const int per = c->custom_n / n_windows;
		std::vector<BandUnit> tbl(n);
		bool uniform = true;
		for (size_t k = 0; k < n; ++k)
		{
			BandUnit& b = tbl[k];
			b = BandUnit{};
			if (c->units[k].n_ev == 0)
			{
				continue;
			}
			// dt of an event against the WINDOW's reference time = dt against the unit's + dt_win: both int32
			const int64_t tw = window_t_ref_us[k / static_cast<size_t>(per)];
			const int64_t d = tw - c->unit_tref[k];
			const int64_t early = tw - c->unit_tmin[k], late = tw - c->unit_tmax[k];
			if (d < INT32_MIN || d > INT32_MAX || early > INT32_MAX || early < INT32_MIN || late < INT32_MIN || late > INT32_MAX)
			{
				return c->fail(EBO_ERR_RANGE, "window reference time further than 2^31 us from a unit's events");
			}
			b.dt_win = static_cast<int32_t>(d);
			b.max_dt = static_cast<int32_t>(std::max(std::llabs(early), std::llabs(late)));
			const int16_t* box = &c->unit_box[4 * k];
			b.x0 = box[0];
			b.x1 = box[1];
			b.y0 = box[2];
			b.y1 = box[3];
			// the patch of the final loop (feature_detector.cpp:436-441): min(int(x / pw), npx - 1), clamped at 0
			auto cell = [&](int x, int y) {
				const int px = std::max(std::min(x / c->prm.patch_w, c->npx - 1), 0);
				const int py = std::max(std::min(y / c->prm.patch_h, c->npy - 1), 0);
				return py * c->npx + px;
			};
			const int a = cell(b.x0, b.y0), z = cell(b.x1, b.y1);
			b.flow = a == z && cell(b.x0, b.y1) == a ? a : -1;  // monotone in x and y: the corners decide
			uniform = uniform && b.flow >= 0;
		}
		(void)hipSetDevice(c->prm.device);
		if (n * sizeof(BandUnit) > c->shard_tbl_cap)
		{
			hipFree(c->d_shard_tbl);
			c->d_shard_tbl = nullptr;
			c->shard_tbl_cap = 0;
			const int rc = c->hip(hipMalloc(&c->d_shard_tbl, n * sizeof(BandUnit)), "hipMalloc shard table");
			if (rc)
			{
				return rc;
			}
			c->shard_tbl_cap = n * sizeof(BandUnit);
		}
		c->shard_tbl_gen = 0;
		int rc = c->hip(hipMemcpyAsync(c->d_shard_tbl, tbl.data(), n * sizeof(BandUnit), hipMemcpyHostToDevice, c->stream),
						"H2D shard table");
		if (rc == EBO_OK)
		{
			rc = c->hip(hipStreamSynchronize(c->stream), "sync");  // tbl is a local; once per set of windows
		}
		if (rc)
		{
			return rc;
		}
		c->shard_tbl_gen = c->units_gen;
		c->shard_tbl_tref.assign(window_t_ref_us, window_t_ref_us + n_windows);
		c->shard_tbl_kind = uniform ? 2 : 1;
	}
	*out = static_cast<const BandUnit*>(c->d_shard_tbl);
	if (uniformFlows)
	{
		*uniformFlows = c->shard_tbl_kind == 2;
	}
	return EBO_OK;
}

bool band_ok(const ebo_ctx* c, const ebo_band* b)
{
	return b && 0 <= b->band_row0 && b->band_row0 <= b->own_row0 && b->own_row0 <= b->own_row1 && b->own_row1 <= b->band_row1 &&
		   b->band_row1 <= c->prm.image_h && b->recv_above >= 0 && b->recv_below >= 0 &&
		   b->recv_above <= b->own_row1 - b->own_row0 && b->recv_below <= b->own_row1 - b->own_row0;
}
}  // namespace ebo_host

extern "C" {

int ebo_band_plan(int image_h, const int* row_bounds, int nranks, int rank, int halo, ebo_band* out)
{
	if (!row_bounds || !out || nranks <= 0 || rank < 0 || rank >= nranks || halo < 0 || image_h <= 0)
	{
		return EBO_ERR_ARG;
	}
	if (row_bounds[0] != 0 || row_bounds[nranks] != image_h)
	{
		return EBO_ERR_ARG;
	}
	for (int q = 0; q < nranks; ++q)
	{
		if (row_bounds[q + 1] < row_bounds[q])
		{
			return EBO_ERR_ARG;
		}
	}
	// Every rank checks EVERY rank's halo (the same inputs everywhere: the same verdict everywhere): what a rank sends
	// up / down must fit inside its neighbour's own rows, or the halo would have to travel two ranks.
	for (int q = 0; q < nranks; ++q)
	{
		const int own0 = row_bounds[q], own1 = row_bounds[q + 1];
		if (own1 == own0)
		{
			continue;  // no rows, no units, nothing to send
		}
		const int up = std::min(halo, own0), down = std::min(halo, image_h - own1);
		if ((up > 0 && (q == 0 || up > row_bounds[q] - row_bounds[q - 1])) ||
			(down > 0 && (q == nranks - 1 || down > row_bounds[q + 2] - row_bounds[q + 1])))
		{
			return EBO_ERR_UNSUPPORTED;
		}
	}
	const int own0 = row_bounds[rank], own1 = row_bounds[rank + 1];
	out->own_row0 = own0;
	out->own_row1 = own1;
	out->band_row0 = own1 > own0 ? own0 - std::min(halo, own0) : own0;
	out->band_row1 = own1 > own0 ? own1 + std::min(halo, image_h - own1) : own1;
	// what the neighbours send: the rank above sends its `bottom` (rows [own0, own0 + n)), the one below its `top`
	const bool aboveHasRows = rank > 0 && row_bounds[rank] > row_bounds[rank - 1];
	const bool belowHasRows = rank < nranks - 1 && row_bounds[rank + 2] > row_bounds[rank + 1];
	out->recv_above = aboveHasRows ? std::min(halo, image_h - own0) : 0;
	out->recv_below = belowHasRows ? std::min(halo, own1) : 0;
	out->recv_above = std::min(out->recv_above, own1 - own0);
	out->recv_below = std::min(out->recv_below, own1 - own0);
	return EBO_OK;
}

int ebo_count_image_band_device(ebo_ctx* c, int n_windows, const int64_t* window_t_ref_us, const double* d_flows_grid,
								const ebo_band* band, uint32_t* d_top, uint32_t* d_own, uint32_t* d_bottom, int32_t* d_escaped)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!window_t_ref_us || !d_flows_grid || !d_escaped || !band_ok(c, band) ||
		(band->own_row0 > band->band_row0 && !d_top) || (band->band_row1 > band->own_row1 && !d_bottom) ||
		(band->own_row1 > band->own_row0 && !d_own))
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_count_image_band_device");
	}
	const BandUnit* tbl = nullptr;
	bool uniform = false;
	int rc = shard_table(c, n_windows, window_t_ref_us, &tbl, &uniform);
	if (rc)
	{
		return rc;
	}
	if (!uniform)
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "band image: a unit's events select more than one grid patch (use ebo_count_image_shard)");
	}
	(void)hipSetDevice(c->prm.device);
	rc = c->hip(hipMemsetAsync(d_escaped, 0, sizeof(int32_t), c->stream), "zero flag");
	if (rc)
	{
		return rc;
	}
	BandLaunch L;
	L.d_events = c->d_events;
	L.d_units = c->d_units;
	L.d_band_units = tbl;
	L.per = c->custom_n / n_windows;
	L.n_windows = n_windows;
	L.d_flows = d_flows_grid;
	L.band0 = band->band_row0;
	L.own0 = band->own_row0;
	L.own1 = band->own_row1;
	L.band1 = band->band_row1;
	L.d_top = d_top;
	L.d_own = d_own;
	L.d_bottom = d_bottom;
	L.d_escaped = d_escaped;
	L.c = make_consts(c);
	if (launch_count_band(L, c->stream))
	{
		return c->fail(EBO_ERR_HIP, "k_count_band launch failed");
	}
	return EBO_OK;
}

int ebo_band_finish_device(ebo_ctx* c, int n_windows, const ebo_band* band, const uint32_t* d_own, const uint32_t* d_from_above,
						   const uint32_t* d_from_below, double* d_image_own)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (n_windows <= 0 || !band_ok(c, band) || (band->own_row1 > band->own_row0 && (!d_own || !d_image_own)) ||
		(band->recv_above > 0 && !d_from_above) || (band->recv_below > 0 && !d_from_below))
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_band_finish_device");
	}
	(void)hipSetDevice(c->prm.device);
	if (launch_band_finish(d_own, band->recv_above > 0 ? d_from_above : nullptr, band->recv_below > 0 ? d_from_below : nullptr,
						   band->own_row1 - band->own_row0, band->recv_above, band->recv_below, c->prm.image_w, n_windows, d_image_own,
						   c->stream))
	{
		return c->fail(EBO_ERR_HIP, "k_band_finish launch failed");
	}
	return EBO_OK;
}

}  // extern "C"
