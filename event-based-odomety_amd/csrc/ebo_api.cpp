// ebo_api.cpp — host side of libebo_hip.so: context, window bucketing/packing,
// kernel launches and the extern "C" entry points declared in include/ebo.h.
// There is no CPU compute path here: every objective value, Jacobian, solved flow
// and count image comes from the HIP kernels in ebo_kernels.hip.
#include "ebo_ctx.h"
#include "lockstep.h"

#include <atomic>
#include <chrono>

namespace ebo_host
{
EvalConsts make_consts(const ebo_ctx* c)
{
	EvalConsts k;
	const double sig = c->prm.k.sigma_compensate;
	k.scale = c->prm.scale;
	k.max_res = c->prm.k.max_possible_residual;
	k.norm = 1.0 / ((2 * M_PI) * (sig * sig));
	k.hs = -0.5 / (sig * sig);
	k.inv_sigsq = 1.0 / (sig * sig);
	k.ck1 = std::exp(k.hs * 1.0);
	k.ck2 = std::exp(k.hs * 4.0);
	k.ck3 = std::exp(k.hs * 9.0);
	// fixed-point grid of the value image: a tap is < norm; choose k with norm < 2^(k-1)
	int kexp = 0;
	while (k.norm >= std::ldexp(0.5, kexp))
	{
		++kexp;
	}
	k.fix_bias = std::ldexp(1.5, kexp);
	k.fix_scale = std::ldexp(1.0, kexp - 52);
	k.image_w = c->prm.image_w;
	k.image_h = c->prm.image_h;
	k.patch_w = c->prm.patch_w;
	k.patch_h = c->prm.patch_h;
	k.npx = c->npx;
	k.npy = c->npy;
	k.inv_pw = static_cast<uint32_t>((uint64_t(1) << 32) / static_cast<uint64_t>(std::max(k.patch_w, 2)) + 1);
	k.inv_ph = static_cast<uint32_t>((uint64_t(1) << 32) / static_cast<uint64_t>(std::max(k.patch_h, 2)) + 1);
	return k;
}

void rect_of(const ebo_ctx* c, int px, int py, int& x, int& y, int& w, int& h)
{
	// feature_detector.cpp:332-346
	x = px * c->prm.patch_w;
	y = py * c->prm.patch_h;
	w = c->prm.patch_w;
	h = c->prm.patch_h;
	if (px == c->npx - 1)
	{
		w = c->prm.image_w - px * c->prm.patch_w;
	}
	if (py == c->npy - 1)
	{
		h = c->prm.image_h - py * c->prm.patch_h;
	}
}

// int32 truncation of the mean of two timestamps (contrast_functor.h:18-20,
// feature_detector.cpp:305-306).  Outside int32 the reference's cast is undefined.
bool mid_timestamp(int64_t a, int64_t b, int64_t& out)
{
	const double half = static_cast<double>(a + b) * 0.5;
	if (!(half > -2147483648.0 && half < 2147483648.0))
	{
		return false;
	}
	out = static_cast<int64_t>(static_cast<int32_t>(half));
	return true;
}

extern const size_t kLdsBudget = 160 * 1024;
// host memory that kernels read and write directly: mapped into the device's address space and
// coherent (fine-grained), whatever HIP_HOST_COHERENT says
extern const unsigned int kZeroCopyFlags = hipHostMallocMapped | hipHostMallocCoherent;
const size_t kRedBytes = 16 * 8 * sizeof(double);

size_t lds_for(int channels, int tiles, int rw, int rh)
{
	const int rows = (3 * rh + tiles - 1) / tiles;
	return static_cast<size_t>(channels) * rows * 3 * rw * sizeof(double) + kRedBytes;
}

int min_tiles(int channels, int rw, int rh, size_t budget)
{
	for (int t = 1; t <= 3 * rh; ++t)
	{
		if (lds_for(channels, t, rw, rh) <= budget)
		{
			return t;
		}
	}
	return -1;
}

int ensure_partials(ebo_ctx* c, size_t n)
{
	if (n <= c->partials_cap)
	{
		return EBO_OK;
	}
	if (c->d_partials)
	{
		hipFree(c->d_partials);
		c->d_partials = nullptr;
		c->partials_cap = 0;
	}
	int rc = c->hip(hipMalloc(&c->d_partials, n * sizeof(double)), "hipMalloc partials");
	if (rc == EBO_OK)
	{
		c->partials_cap = n;
	}
	return rc;
}

int ensure_aux(ebo_ctx* c, size_t bytes)
{
	if (bytes <= c->aux_cap)
	{
		return EBO_OK;
	}
	if (c->d_aux)
	{
		hipFree(c->d_aux);
		c->d_aux = nullptr;
		c->aux_cap = 0;
	}
	int rc = c->hip(hipMalloc(&c->d_aux, bytes), "hipMalloc aux");
	if (rc == EBO_OK)
	{
		c->aux_cap = bytes;
	}
	return rc;
}

int ensure_scratch(ebo_ctx* c, size_t bytes)
{
	if (bytes <= c->scratch_cap)
	{
		return EBO_OK;
	}
	if (c->d_scratch)
	{
		hipFree(c->d_scratch);
		c->d_scratch = nullptr;
		c->scratch_cap = 0;
	}
	int rc = c->hip(hipMalloc(&c->d_scratch, bytes), "hipMalloc scratch");
	if (rc == EBO_OK)
	{
		c->scratch_cap = bytes;
	}
	return rc;
}

// Evaluation geometry.  tiles: row tiles per unit (parallel workgroups).  More
// tiles = more workgroups and less LDS each; the events of a unit are re-read
// (from L2) by each of its tiles.  EBO_EVAL_TILES / EBO_EVAL_BLOCK override.
// impl 1/2: the image is the bounding box of the warped events; one workgroup owns
// `cap` pixels of LDS (default 32 KiB => 5 workgroups per CU) and walks larger boxes
// in sequential sub-bands.  A full canvas row must fit.
int image_capacity(ebo_ctx* c, int& capDoubles, size_t& lds, size_t defaultKb = 40)
{
	const size_t headerBytes = 160 * sizeof(double);
	// 40 KB at three waves per SIMD (four workgroups per CU; the device-resident solve): as fast as 32 at small
	// flows, +5-8 % near convergence (fewer sub-bands); 48 and more cost occupancy.  The batched evaluation
	// runs four waves per SIMD and passes 31 KB: five 192-lane workgroups per CU.
	size_t kb = ab_size("EBO_LDS_KB", defaultKb);
	size_t bytes = std::min<size_t>(std::max<size_t>(kb, 4) * 1024, kLdsBudget);
	// keep at least 24 rows of the widest canvas where that fits
	const size_t want = static_cast<size_t>(24) * 3 * c->max_rw * sizeof(double) + headerBytes;
	bytes = std::min(std::max(bytes, want), kLdsBudget);
	if (bytes < headerBytes + static_cast<size_t>(3) * c->max_rw * sizeof(double))
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "patch too wide: one canvas row does not fit LDS");
	}
	capDoubles = static_cast<int>((bytes - headerBytes) / sizeof(double));
	lds = bytes;
	return EBO_OK;
}

int eval_impl()
{
	const int v = static_cast<int>(ab_size("EBO_EVAL_IMPL", 3));
	return (v < 0 || v > 3) ? 3 : v;
}

int eval_geometry(ebo_ctx* c, int channels, int& tiles, int& block, size_t& lds)
{
	const size_t budget = ab_size("EBO_LDS_BUDGET", kLdsBudget);
	const int fit = min_tiles(channels, c->max_rw, c->max_rh, budget);
	if (fit < 0)
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "patch too wide for LDS row tiling");
	}
	int t = static_cast<int>(ab_size("EBO_EVAL_TILES", 0));
	if (t <= 0)
	{
		t = fit;
		const int nUnits = static_cast<int>(c->n_flows());
		// fill the chip (256 CUs) when there are few units, but keep >= 16 rows per tile
		while (nUnits * t < 512 && (3 * c->max_rh) / (t + 1) >= 16)
		{
			++t;
		}
	}
	t = std::max(t, fit);
	tiles = t;
	block = static_cast<int>(ab_size("EBO_EVAL_BLOCK", 256));
	if (block < 64 || block > 1024 || (block & 63))
	{
		return c->fail(EBO_ERR_ARG, "EBO_EVAL_BLOCK must be a multiple of 64 in [64,1024]");
	}
	lds = lds_for(channels, t, c->max_rw, c->max_rh);
	return EBO_OK;
}

// Edge loss (contrast_functor.h:152-277): one workgroup per unit, arrays in LDS, a
// per-unit global slice as fallback for boxes that do not fit.
// Geometry of an edge-loss launch (LDS layout, workgroup size, global fallback slices, tensor
// weights): shared by the batched evaluation (k_eval_edge) and the device-resident solve (k_solve_edge).
int edge_launch_setup(ebo_ctx* c, EdgeLaunch& L)
{
	L.d_events = c->d_events;
	L.d_units = c->d_units;
	L.n_units = static_cast<int>(c->units.size());
	// one workgroup per CU (LDS-limited): a big workgroup is the only source of waves
	// (measured, C2 x 64 windows: 256 threads 2.1, 512: 2.8, 1024: 3.3 Gevents/s)
	L.block = static_cast<int>(ab_size("EBO_EDGE_BLOCK", 0));
	if (L.block != 0 && (L.block < 64 || L.block > 768 || (L.block & 63)))
	{
		return c->fail(EBO_ERR_ARG, "EBO_EDGE_BLOCK must be a multiple of 64 in [64,768]");
	}
	if (L.for_solve && L.block > 512)
	{
		L.block = 512;  // k_solve_edge is built for 256 and 512 lanes
	}
	const size_t headerBytes = (168 + 1024) * sizeof(double);  // kEdgeHeader
	const size_t canvasPx = static_cast<size_t>(9) * c->max_rw * c->max_rh;
	// Two LDS layouts (ebo_edge.inc): I, E, A (f64) + cnt (i32) = 28 B per pixel with the
	// separable tensor filter, one 1024-lane workgroup per CU; or, when the whole canvas then
	// fits TWICE into a CU's LDS (a 60x60 canvas: 77 KB), I, E/A + cnt = 20 B per pixel with the
	// direct filter and two 512-lane workgroups per CU (128 VGPRs each: together the CU's
	// register file) that overlap each other's barriers.  Measured, reference-default
	// configuration x 256 windows: 4.27 -> 2.64 ms; C2 (canvas 90x66) stays on the first layout
	// (0.93 ms vs 0.98 / 1.27 ms for the second with the direct / short-band separable filter).
	const size_t bytesPerPx = 3 * sizeof(double) + sizeof(int32_t);
	const size_t aliasPerPx = 2 * sizeof(double) + sizeof(int32_t);
	const char* layoutEnv = ab_env("EBO_EDGE_LAYOUT");  // 0 / 1 force a layout (A/B)
	const bool fitsTwice = headerBytes + canvasPx * aliasPerPx + 64 <= 80 * 1024 - 256;
	// The grid's REGULAR patches decide (the last row / column of a grid absorbs the remainder of
	// the sensor and can be almost twice as large): when their canvas fits twice, the 20 B layout
	// runs two workgroups per CU and the few larger border units take the global-memory slice
	// (C3, 21x16 patches with a 31x20 corner: 1.47 -> 0.95 ms).  And when the 28 B layout cannot
	// hold the regular canvas in LDS at all but the 20 B one can (C4, 40x22 patches: 120x66 canvas),
	// the 20 B layout spares most units the global slice (1.92 -> 1.65 ms).
	const size_t regularPx = c->custom_n ? canvasPx : static_cast<size_t>(9) * c->prm.patch_w * c->prm.patch_h;
	const bool regularFitsTwice = headerBytes + regularPx * aliasPerPx + 64 <= 80 * 1024 - 256;
	// (boxes are smaller than the canvas: holding 90 % of it is as good as all of it)
	const bool onlyAliasFits = headerBytes + regularPx * bytesPerPx + 64 > kLdsBudget &&
							   headerBytes + (regularPx * 9 / 10) * aliasPerPx + 64 <= kLdsBudget && regularPx <= 8 * 1024;
	L.alias_lds = layoutEnv ? (std::atoi(layoutEnv) != 0) : ((fitsTwice || regularFitsTwice || onlyAliasFits) ? 1 : 0);
	const size_t ldsPerPx = L.alias_lds ? aliasPerPx : bytesPerPx;
	const bool ldsEnv = ab_env("EBO_EDGE_LDS_KB") != nullptr;
	size_t ldsBytes = std::min<size_t>(ab_size("EBO_EDGE_LDS_KB", 160) * 1024, kLdsBudget);
	if (L.alias_lds && !layoutEnv && !ldsEnv && !fitsTwice && regularFitsTwice)
	{
		ldsBytes = 80 * 1024 - 256;  // two workgroups per CU; larger border units: global slice
	}
	// no point in reserving more LDS than the whole canvas needs
	ldsBytes = std::min(ldsBytes, headerBytes + canvasPx * ldsPerPx + 64);
	L.cap_px = static_cast<int>((ldsBytes - headerBytes) / ldsPerPx);
	if (L.block == 0)
	{
		// two workgroups per CU: 256 lanes each (256 VGPRs); one per CU: 768 lanes (168 VGPRs) for the
		// batched evaluation, 512 (256 VGPRs) for the device-resident solve -- ebo_edge.inc, MAXT
		const bool twoPerCu = L.alias_lds && headerBytes + static_cast<size_t>(L.cap_px) * ldsPerPx <= 80 * 1024 - 256;
		L.block = twoPerCu ? 256 : (L.for_solve ? 512 : 768);
	}
	L.lds_bytes = headerBytes + static_cast<size_t>(L.cap_px) * ldsPerPx;
	// Workgroup slots.  The batched evaluation runs PERSISTENT workgroups -- as many as the chip holds at once (two
	// 256-lane ones per CU, else one), each walking the launch's units -- and keys its per-evaluation temporaries (the
	// global fallback slice, the direction table) by workgroup; the device-resident solve runs one workgroup per unit.
	if (c->n_cus == 0)
	{
		hipDeviceProp_t prop;
		c->n_cus = (hipGetDeviceProperties(&prop, c->prm.device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
	}
	{
		const size_t items = static_cast<size_t>(L.n_units) * L.flow_sets;
		// (the 256-lane instantiations of k_eval_edge only: the 768-lane ones keep one workgroup per item, ebo_edge.inc)
		L.wide_kernel = ab_size("EBO_EDGE_WIDE", 0) != 0;
		const bool persistent = !L.for_solve && L.block <= 256 && !L.wide_kernel;
		const size_t resident = static_cast<size_t>(c->n_cus) * (2 * L.lds_bytes <= 160 * 1024 ? 2 : 1);
		const size_t forced = ab_size("EBO_EDGE_PERSIST", 1);  // A/B: 0 = one workgroup per item (rounds 1-4), n > 1 = n workgroups
		const size_t slots = !persistent ? items : (forced == 0 ? items : (forced > 1 ? forced : resident));
		L.wg_slots = static_cast<int>(std::max<size_t>(1, std::min(items, slots)));
	}
	// The compact layout (round 5): 16.5 B per pixel and a header sized by the launch let THREE 256-lane workgroups share
	// a CU (measured at zero flow, where every box fits: 1.32 -> 0.98 ms for 256 reference-default windows).  Only
	// where the 20 B layout runs two per CU, for batches larger than the chip holds at once, one flow set.  Units whose
	// box does not fit the compact arrays are deferred to a second launch on the 20 B layout.
	L.compact = EdgeCompact();
	{
		const size_t items = static_cast<size_t>(L.n_units);
		const char* force = ab_env("EBO_EDGE_COMPACT");  // A/B: 0 never, 1 whenever the layout allows
		const bool eligible = L.alias_lds && L.block == 256 && !L.for_solve && L.flow_sets == 1 && !L.wide_kernel &&
							  2 * L.lds_bytes <= 160 * 1024;
		const size_t itemsNow = L.live.n > 0 ? static_cast<size_t>(L.live.n) * L.live.upw : items;  // a thinned-out lock-step round
		// (from eight units per CU up: measured on the reference-default lock-step call, the two launches of the compact
		// path cost a 16-window solve +15 %, break even at 64 windows and save 9 % at 256 -- profiles/r05_edge_levers.txt)
		const size_t atLeast = static_cast<size_t>(8) * c->n_cus;
		const bool want = force ? std::atoi(force) != 0 : itemsNow > atLeast;
		const bool wantFull = force ? std::atoi(force) != 0 : items > atLeast;  // ... some launch of this context
		L.compact_table_px = 0;
		if (eligible && wantFull)
		{
			const size_t budget = ab_size("EBO_EDGE_COMPACT_KB", 52) * 1024;
			// red[128 doubles] | 80 ints | list (one int per 4 pixels + slack) | I (8 B per STORED pixel: the rows with
			// taps) | E (8 B per pixel of the box) | 4-bit counters.  A box fits when its arrays fit the bytes of `cap`
			// untrimmed pixels (16.5 B each) and it has at most maxPx pixels -- what the list and the direction table
			// are laid out for: no box of more than budget / 13 pixels can fit (I stores at least half of its rows).
			const size_t maxPx = std::min((canvasPx + 7) & ~static_cast<size_t>(7), (budget / 13) & ~static_cast<size_t>(7));
			const size_t listCap = ((maxPx / 4 + 64) + 1) & ~static_cast<size_t>(1);
			const size_t hdrBytes = (168 + listCap / 2) * sizeof(double);
			size_t cap = budget > hdrBytes ? ((budget - hdrBytes) * 2 / 33) & ~static_cast<size_t>(7) : 0;
			cap = std::min(cap, maxPx);
			const size_t bytes = hdrBytes + cap * 16 + cap / 2;
			if (cap >= 1024 && bytes <= budget)
			{
				L.compact_table_px = static_cast<int>(maxPx);
			}
			if (cap >= 1024 && bytes <= budget && want)
			{
				const size_t listInts = (items + 3) & ~static_cast<size_t>(3);
				if (c->edge_defer_cap < items)
				{
					if (c->d_edge_defer)
					{
						c->hip(hipStreamSynchronize(c->stream), "sync");
						hipFree(c->d_edge_defer);
						c->d_edge_defer = nullptr;
						c->edge_defer_cap = 0;
					}
					int rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_edge_defer), (4 + listInts) * sizeof(int) + items * 4 * sizeof(int)),
									"hipMalloc edge deferred list");
					if (rc)
					{
						return rc;
					}
					c->edge_defer_cap = items;
				}
				L.compact.hdr_doubles = static_cast<int>(168 + listCap / 2);
				L.compact.list_cap = static_cast<int>(listCap);
				L.compact.defer_count = c->d_edge_defer;
				L.compact.defer_list = c->d_edge_defer + 4;
				// (behind the list as laid out for THIS context's items: the allocation may be larger, the offsets are this launch's)
				L.compact.bbox = ab_size("EBO_EDGE_CLASSIFY", 1) ? static_cast<void*>(c->d_edge_defer + 4 + listInts) : nullptr;
				L.compact_cap_px = static_cast<int>(cap);
				L.compact_lds_bytes = bytes;
			}
		}
	}
	L.scratch_stride = (canvasPx * (bytesPerPx + 1) + 256 + 255) & ~static_cast<size_t>(255);  // I, E, A, cnt + the argmax list (canvasPx / 4 + 64 ints)
	L.d_scratch = nullptr;
	// the most pixels an LDS-resident box of the aliased layouts may have: what the argmax list (2048 ints, the last 128
	// of them the device-resident solver's state; a window per four pixels) and a slot of the direction table hold
	const size_t maxLdsPx = std::min<size_t>(canvasPx, 4 * (2048 - 128 - 64));
	if (static_cast<size_t>(L.cap_px) < canvasPx || maxLdsPx < canvasPx)
	{
		// some box could exceed LDS: keep a global slice per (set, unit)
		const size_t need = L.scratch_stride * static_cast<size_t>(L.wg_slots);
		if (need > (static_cast<size_t>(16) << 30))
		{
			return c->fail(EBO_ERR_UNSUPPORTED, "edge loss fallback scratch would exceed 16 GiB; use fewer windows per batch");
		}
		if (need > c->edge_scratch_cap)
		{
			if (c->d_edge_scratch)
			{
				hipFree(c->d_edge_scratch);
				c->d_edge_scratch = nullptr;
				c->edge_scratch_cap = 0;
			}
			int rc = c->hip(hipMalloc(&c->d_edge_scratch, need), "hipMalloc edge scratch");
			if (rc)
			{
				return rc;
			}
			c->edge_scratch_cap = need;
		}
		L.d_scratch = static_cast<char*>(c->d_edge_scratch);
	}
	int rc = ensure_partials(c, static_cast<size_t>(5) * L.n_units * 3);
	if (rc)
	{
		return rc;
	}
	L.d_sets = c->d_partials;
	L.c = make_consts(c);
	// weights exactly as the reference builds them (:193-202): gaussian(0, 0, j, i, sigmaST)
	const double sig = c->prm.k.sigma_st;
	const double sigmaSq = sig * sig;
	const double normCoef = 1.0 / ((2 * M_PI) * sigmaSq);
	if (!c->d_edge_w || c->edge_w_sigma != sig)
	{
		double w[49];
		for (int i = -3; i <= 3; ++i)
		{
			for (int j = -3; j <= 3; ++j)
			{
				const double x = j, y = i;
				w[(i + 3) * 7 + (j + 3)] = normCoef * std::exp(-0.5 / sigmaSq * (x * x + y * y));
			}
		}
		if (!c->d_edge_w)
		{
			rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_edge_w), sizeof(w)), "hipMalloc edge weights");
			if (rc)
			{
				return rc;
			}
		}
		// synchronous: the table must not change under a launch still in flight
		rc = c->hip(hipStreamSynchronize(c->stream), "sync");
		if (rc == EBO_OK) rc = c->hip(hipMemcpy(c->d_edge_w, w, sizeof(w), hipMemcpyHostToDevice), "H2D edge weights");
		if (rc)
		{
			return rc;
		}
		c->edge_w_sigma = sig;
	}
	L.ec.w = c->d_edge_w;
	L.ec.w_max = normCoef;
	for (int k = -3; k <= 3; ++k)
	{
		L.ec.g[k + 3] = std::exp(-0.5 / sigmaSq * static_cast<double>(k * k));
	}
	L.ec.norm_st = normCoef;
	// Eigenvector directions per pixel for the reverse pass (16 B per LDS-resident pixel and WORKGROUP SLOT, in
	// L2 / Infinity Cache): spares every argmax entry the re-derivation of its tensor sums.  Beyond 4 GiB
	// (EBO_EDGE_CS_MB) the reverse pass re-derives them instead.
	L.ec.cs = nullptr;
	L.ec.cs_stride = static_cast<int>(L.alias_lds ? maxLdsPx : static_cast<size_t>(L.cap_px));
	{
		// (the compact launch keys its slots by unit, one workgroup each: the larger of the two launches' tables)
		// (sized by the context's units, not by this launch's window list: the rounds of a lock-step solve must not
		// flip between two sizes -- a reallocation synchronises the device)
		const size_t need = std::max(static_cast<size_t>(L.wg_slots) * L.ec.cs_stride,
									 L.compact_table_px > 0 ? static_cast<size_t>(L.n_units) * L.compact_table_px : 0) * 2 * sizeof(double);
		// The table is an optimisation nobody asked for by name, so it must not surprise: at most
		// EBO_EDGE_CS_MB (default 4096) AND at most a quarter of the memory that is free right now
		// (several contexts share a GPU: one per FeatureDetector / Optimizer / TrackedPatches of the
		// facade, the ranks of a rehearsal); a failed allocation is not an error (the reverse pass
		// re-derives what the table would have held); and a context that kept a large table for one
		// big batch gives it back when the batches that follow need less than a quarter of it.
		const size_t limit = env_size("EBO_EDGE_CS_MB", 4096) << 20;
		if (L.want_jac && L.flow_sets == 1 && need <= limit && need > 0)
		{
			// (the free-memory probe and the allocation happen only when the table has to change size -- once per
			// batch shape, never per launch: hipFree synchronises the whole device)
			if (need > c->edge_cs_cap || need < c->edge_cs_cap / 4)
			{
				size_t freeB = 0, totalB = 0;
				const bool probed = hipMemGetInfo(&freeB, &totalB) == hipSuccess;
				if (c->d_edge_cs)
				{
					c->hip(hipStreamSynchronize(c->stream), "sync");
					hipFree(c->d_edge_cs);
					c->d_edge_cs = nullptr;
					freeB += c->edge_cs_cap;
					c->edge_cs_cap = 0;
				}
				if ((!probed || need <= freeB / 4) && hipMalloc(reinterpret_cast<void**>(&c->d_edge_cs), need) == hipSuccess)
				{
					c->edge_cs_cap = need;
				}
				else
				{
					(void)hipGetLastError();  // not an error: evaluate without the table
					c->d_edge_cs = nullptr;
				}
			}
			L.ec.cs = c->d_edge_cs;
		}
	}
	L.ec.stats = nullptr;
	L.ec.mean_threshold = 0.0001;
	L.ec.ablate = static_cast<int>(ab_size("EBO_EDGE_ABLATE", 0));
	L.ec.reserved = static_cast<int>(ab_size("EBO_EDGE_SEPARABLE", 7));  // forms of the separable tensor filter (see EdgeConsts / ebo_edge.inc)
	return EBO_OK;
}

int run_eval_edge(ebo_ctx* c, const double* d_flows, int want_jac, double* d_out)
{
	const bool central = want_jac && c->prm.grad == EBO_GRAD_CENTRAL;
	EdgeLaunch L;
	L.d_modes = c->modes_active;
	L.live = c->live_active;
	L.d_flows = d_flows;
	L.want_jac = (want_jac && !central) ? 1 : 0;
	L.flow_sets = central ? 5 : 1;
	L.fd_step = central ? c->prm.fd_step : 0.0;
	L.d_out = d_out;
	int rc = edge_launch_setup(c, L);
	if (rc)
	{
		return rc;
	}
	L.ec.stats = c->edge_stats_dev;  // non-null only inside ebo_edge_work_stats
	if (launch_eval_edge(L, c->stream))
	{
		return c->hip(hipGetLastError(), "edge eval launch");
	}
	return EBO_OK;
}

int run_eval_device(ebo_ctx* c, const double* d_flows, int want_jac, double* d_out)
{
	if (c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	if (c->prm.loss == EBO_LOSS_EDGE)
	{
		return run_eval_edge(c, d_flows, want_jac, d_out);
	}
	const bool central = want_jac && c->prm.grad == EBO_GRAD_CENTRAL;
	EvalLaunch L;
	L.d_modes = c->modes_active;
	L.live = c->live_active;
	L.d_events = c->d_events;
	L.d_units = c->d_units;
	L.n_units = static_cast<int>(c->units.size());
	L.d_flows = d_flows;
	L.n_flow = static_cast<int>(c->n_flows());
	L.flow_sets = central ? 5 : 1;
	L.channels = (want_jac && !central) ? 3 : 1;
	L.fd_step = central ? c->prm.fd_step : 0.0;
	L.impl = eval_impl();
	L.cap_doubles = 0;
	L.rotate = ab_size("EBO_EVAL_ROT", 1) ? 1 : 0;
	L.deal = ab_size("EBO_EVAL_DEAL", 0) ? 1 : 0;
	int rc;
	if (L.impl == 0)
	{
		rc = eval_geometry(c, L.channels, L.tiles, L.block, L.lds_bytes);
	}
	else
	{
		// Many units: k_eval3 runs four waves per SIMD (128 VGPRs), i.e. 16 waves per CU, and what a launch
		// chooses is how to cut them into workgroups -- more, smaller workgroups overlap their barrier-separated
		// passes better on the CU's one LDS, but each gets less of it, and a box that does not fit is counted in
		// sequential sub-bands (every event warped again per sub-band).  By the canvas of the REGULAR patch (the
		// geometry only, never the data: a window evaluates bit-identically alone and inside any batch):
		//    canvas <= 32 KB (the reference's 20x20 patches, C3's 21x16): 7 workgroups of 128 lanes, 22 KB
		//    larger (C2 30x22, C4 40x22):                                  4 workgroups of 256 lanes, 39 KB
		// Measured against three waves per SIMD with four 192-lane workgroups of 40 KB (round 2), evaluation with
		// Jacobian at flows 0 / 0.5 / 1.0 x ground truth: 20x20 patches of 139 events -23 / -19 / -13 % time,
		// C3 -8.5 / -5.3 / +0.8 %, C2 -6.3 / -2.5 / -2.3 %, C4 -3.9 / -2.9 / -4.4 % (profiles/r03_eval3_waves4_ab.txt;
		// 128 lanes x 19 KB: better still at small flows, +6 % at the ground truth of C3; 192 lanes x 31 KB:
		// -2 ... -4 % everywhere at C3, but only -8 ... -11 % on the 20x20 patches).
		const int nUnitsAll = std::max(1, static_cast<int>(c->n_flows()));
		const int regW = c->reg_rw, regH = c->reg_rh;
		const bool smallCanvas = static_cast<size_t>(9) * regW * regH * sizeof(double) <= 32 * 1024;
		const int manyBlock = smallCanvas ? 128 : 256;
		rc = image_capacity(c, L.cap_doubles, L.lds_bytes, nUnitsAll < 1024 ? 40 : (smallCanvas ? 22 : 39));
		// Measured (tools/sweep_impl.py): one row band per unit is best at every batch size
		// (each band workgroup pays the bounding-box pass over all events); with few units
		// a 512-thread workgroup shortens the per-unit critical path (29 vs 46 us for one
		// 64-patch window); with many units the smaller workgroups chosen above pack the CU better and
		// waste fewer lanes in a unit's last round of events.  (The sums of a unit are reduced over the
		// workgroup, so the two regimes differ in the last bits: a window is bit-identical alone
		// and inside a batch as long as both are on the same side of 1024 units.)
		const int nUnits = std::max(1, static_cast<int>(c->n_flows()));
		L.block = static_cast<int>(ab_size("EBO_EVAL_BLOCK", nUnits < 1024 ? 512 : manyBlock));
		if (L.block < 64 || L.block > 512 || (L.block & 63))
		{
			return c->fail(EBO_ERR_ARG, "EBO_EVAL_BLOCK must be a multiple of 64 in [64,512]");
		}
		int t = static_cast<int>(ab_size("EBO_EVAL_TILES", 0));
		if (t <= 0)
		{
			// One workgroup per unit, unless a unit's image needs many sequential sub-bands (a
			// patch as large as the frame, configs[0]: a 720x540 canvas): then its rows are split
			// over a few parallel workgroups (measured there: 4-8 tiles best for few windows, 2 for
			// many).  The choice depends on the patch geometry only, NOT on the batch size, so that
			// a window evaluates bit-identically alone and inside any batch.
			const long canvas = 9L * c->max_rw * c->max_rh;
			const long subBands = (canvas + L.cap_doubles - 1) / std::max(L.cap_doubles, 1);
			t = static_cast<int>(std::min<long>(std::max<long>(subBands / 6, 1), 4));
			while (t > 1 && (3 * c->max_rh) / t < 16)
			{
				--t;
			}
		}
		L.tiles = std::max(1, std::min(t, 64));
	}
	if (rc)
	{
		return rc;
	}
	rc = ensure_partials(c, static_cast<size_t>(L.flow_sets) * L.n_units * L.tiles * kPartialStride);
	if (rc)
	{
		return rc;
	}
	L.d_partials = c->d_partials;
	L.d_out = d_out;
	L.c = make_consts(c);
	if (launch_eval_variance(L, c->stream))
	{
		return c->hip(hipGetLastError(), "eval launch");
	}
	return EBO_OK;
}

// modes (optional, [n_flows]): 0 = slot not wanted this round (r / jac left as they are),
// 1 = value only, 2 = value + Jacobian; lets a lock-step solve skip finished problems and
// Jacobians nobody asked for.
// pinned staging of evaluation rounds: flows [nf][2], results [nf][3], and mode tables [4][nf]
// (the pipelined lock-step solve has up to four rounds in flight)
int ensure_eval_staging(ebo_ctx* c, size_t nf)
{
	int rc = EBO_OK;
	if (nf > c->pin_cap)
	{
		(void)hipSetDevice(c->prm.device);  // the staging is mapped for the context's device
		if (c->pin_flows)
		{
			(void)hipHostFree(c->pin_flows);
			(void)hipHostFree(c->pin_out);
			(void)hipHostFree(c->pin_modes);
			c->pin_flows = c->pin_out = nullptr;
			c->pin_modes = nullptr;
			c->pin_cap = 0;
		}
		hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->pin_flows), nf * 2 * sizeof(double), kZeroCopyFlags);
		if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->pin_out), nf * 3 * sizeof(double), kZeroCopyFlags);
		if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->pin_modes), 4 * nf, kZeroCopyFlags);  // one table per round in flight (up to 4)
		rc = c->hip(e, "hipHostMalloc evaluation staging");
		if (rc)
		{
			return rc;
		}
		std::memset(c->pin_out, 0, nf * 3 * sizeof(double));
		c->pin_cap = nf;
	}
	return rc;
}

int ensure_device_modes(ebo_ctx* c, size_t nf)
{
	if (nf > c->modes_cap)
	{
		if (c->d_modes)
		{
			hipFree(c->d_modes);
			c->d_modes = nullptr;
			c->modes_cap = 0;
		}
		int rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_modes), nf), "hipMalloc modes");
		if (rc)
		{
			return rc;
		}
		c->modes_cap = nf;
	}
	return EBO_OK;
}

// One half of a pipelined lock-step round: slots [s0, s1) of the flows go up, the launch covers
// every unit with the mode table `modes` (zero outside the half: those workgroups exit at once),
// the half's results come back, and `done` marks the end; nothing here waits.
// The running windows of slots [s0, s1) as a kernel-argument list (LiveWindows), if the context is a grid of
// windows, the caller vouches for one mode per window (windowSlots == P) and at most kLiveMax are running.
bool live_windows_of(const ebo_ctx* c, const unsigned char* modes, size_t s0, size_t s1, int windowSlots, bool wantJac,
					 LiveWindows& live)
{
	const bool noCompact = ab_env("EBO_SOLVE_NO_COMPACT") != nullptr;  // (A/B and tests: read per round)
	live.n = 0;
	// A central-difference Jacobian round evaluates five flow sets and combines them (k_combine_variance /
	// k_edge_central): those launches cover every unit and know neither the list nor the modes, so such a
	// round takes the full path, which copies back only its own slots [s0, s1).
	const bool central = wantJac && c->prm.grad == EBO_GRAD_CENTRAL;
	if (c->custom_n || c->P <= 0 || windowSlots != c->P || s0 % c->P != 0 || s1 % c->P != 0 || noCompact || central)
	{
		return false;
	}
	const size_t P = static_cast<size_t>(c->P);
	live.upw = c->P + 1;
	for (size_t w = s0 / P; w < s1 / P; ++w)
	{
		const unsigned char m = modes[w * P];
		if (m != 0)
		{
			if (live.n == kLiveMax)
			{
				live.n = 0;
				return false;
			}
			live.ent[live.n++] = static_cast<int>(w << 2) | m;
		}
	}
	return live.n > 0;
}

int eval_begin(ebo_ctx* c, const double* flows, const unsigned char* modes, int which, size_t s0, size_t s1, bool wantJac,
			   hipEvent_t done, int windowSlots)
{
	const size_t nf = c->n_flows();
	// A half that has thinned out to a few windows (the late rounds of a batch: most of a solve's ROUNDS, little of
	// its work): a launch over every unit with a mode table, the half's flows up and its results down were
	// ~0.6 MB and 27 k mostly empty workgroups per round for two live windows.  Up to kLiveMax live windows go
	// as a list in the kernel arguments instead: workgroups for their units only, their mode in the list entry,
	// flows read from and results written to the pinned buffers by the kernel itself -- one launch, no copy.
	// (Windows of a grid context: P flow slots and P + 1 units each, one mode per window.)
	{
		LiveWindows live;
		if (live_windows_of(c, modes, s0, s1, windowSlots, wantJac, live))
		{
			const size_t P = static_cast<size_t>(c->P);
			for (int k = 0; k < live.n; ++k)
			{
				const size_t w = static_cast<size_t>(live.ent[k] >> 2);
				std::memcpy(c->pin_flows + 2 * w * P, flows + 2 * w * P, P * 2 * sizeof(double));
			}
			c->live_active = live;
			const int rc = run_eval_device(c, c->pin_flows, wantJac, c->pin_out);
			c->live_active.n = 0;
			if (rc)
			{
				return rc;
			}
			return c->hip(hipEventRecord(done, c->stream), "compact round");
		}
	}
	std::memcpy(c->pin_flows + 2 * s0, flows + 2 * s0, (s1 - s0) * 2 * sizeof(double));
	hipError_t e = hipMemcpyAsync(c->d_flows + 2 * s0, c->pin_flows + 2 * s0, (s1 - s0) * 2 * sizeof(double),
								  hipMemcpyHostToDevice, c->stream);
	unsigned char* pm = c->pin_modes + static_cast<size_t>(which) * nf;
	std::memcpy(pm, modes, nf);
	if (e == hipSuccess) e = hipMemcpyAsync(c->d_modes, pm, nf, hipMemcpyHostToDevice, c->stream);
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D round");
	}
	c->modes_active = c->d_modes;
	int rc = run_eval_device(c, c->d_flows, wantJac, c->d_out);
	c->modes_active = nullptr;
	if (rc)
	{
		return rc;
	}
	e = hipMemcpyAsync(c->pin_out + 3 * s0, c->d_out + 3 * s0, (s1 - s0) * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipEventRecord(done, c->stream);
	return c->hip(e, "D2H round");
}

int eval_finish(ebo_ctx* c, const unsigned char* modes, size_t s0, size_t s1, bool wantJac, double* r, double* jac,
				hipEvent_t done, int windowSlots)
{
	int rc = c->hip(hipEventSynchronize(done), "round");
	if (rc)
	{
		return rc;
	}
	const double* h_out = c->pin_out;
	const size_t run = windowSlots > 0 ? static_cast<size_t>(windowSlots) : 1;  // one mode per aligned run of slots
	for (size_t i = s0; i < s1; ++i)
	{
		if (modes[i] == 0)
		{
			if (run > 1 && i % run == 0)
			{
				i += run - 1;  // a finished window: none of its slots is wanted
			}
			continue;
		}
		r[i] = h_out[3 * i];
		if (wantJac && modes[i] == 2)
		{
			jac[2 * i] = h_out[3 * i + 1];
			jac[2 * i + 1] = h_out[3 * i + 2];
		}
	}
	return EBO_OK;
}

int eval_host(ebo_ctx* c, const double* flows, double* r, double* jac, const unsigned char* modes = nullptr, int windowSlots = 0)
{
	const size_t nf = c->n_flows();
	int rc = ensure_eval_staging(c, nf);
	if (rc)
	{
		return rc;
	}
	// a batch of windows of which some have finished: the running ones as a list in the kernel arguments, flows and
	// results through the pinned buffers (as in eval_begin)
	LiveWindows live;
	if (modes && live_windows_of(c, modes, 0, nf, windowSlots, jac != nullptr, live))
	{
		const size_t P = static_cast<size_t>(c->P);
		for (int k = 0; k < live.n; ++k)
		{
			const size_t w = static_cast<size_t>(live.ent[k] >> 2);
			std::memcpy(c->pin_flows + 2 * w * P, flows + 2 * w * P, P * 2 * sizeof(double));
		}
		c->live_active = live;
		rc = run_eval_device(c, c->pin_flows, jac != nullptr, c->pin_out);
		c->live_active.n = 0;
		if (rc == EBO_OK)
		{
			rc = c->hip(hipStreamSynchronize(c->stream), "evaluation");
		}
		if (rc)
		{
			return rc;
		}
		const double* h = c->pin_out;
		for (int k = 0; k < live.n; ++k)
		{
			const size_t w = static_cast<size_t>(live.ent[k] >> 2);
			const bool wj = jac && (live.ent[k] & 3) == 2;
			for (size_t i = w * P; i < (w + 1) * P; ++i)
			{
				r[i] = h[3 * i];
				if (wj)
				{
					jac[2 * i] = h[3 * i + 1];
					jac[2 * i + 1] = h[3 * i + 2];
				}
			}
		}
		return EBO_OK;
	}
	// Small rounds (a single window's LM round is 108 flows): the kernels read the flows from
	// and write the results to the pinned buffers themselves; the round is one launch + one
	// sync.  Larger rounds move the data with async copies (which really are async from pinned).
	const size_t zeroCopyMax = ab_size("EBO_ZERO_COPY_MAX", 4096);
	const bool zeroCopy = nf <= zeroCopyMax;
	std::memcpy(c->pin_flows, flows, nf * 2 * sizeof(double));
	const double* dFlows = c->pin_flows;
	double* dOut = c->pin_out;
	if (!zeroCopy)
	{
		rc = c->hip(hipMemcpyAsync(c->d_flows, c->pin_flows, nf * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream),
					"H2D flows");
		if (rc)
		{
			return rc;
		}
		dFlows = c->d_flows;
		dOut = c->d_out;
	}
	if (modes)
	{
		std::memcpy(c->pin_modes, modes, nf);
		c->modes_active = c->pin_modes;
		if (!zeroCopy)
		{
			rc = ensure_device_modes(c, nf);
			if (rc)
			{
				return rc;
			}
			rc = c->hip(hipMemcpyAsync(c->d_modes, c->pin_modes, nf, hipMemcpyHostToDevice, c->stream), "H2D modes");
			if (rc)
			{
				return rc;
			}
			c->modes_active = c->d_modes;
		}
	}
	rc = run_eval_device(c, dFlows, jac != nullptr, dOut);
	c->modes_active = nullptr;
	if (rc)
	{
		return rc;
	}
	if (!zeroCopy)
	{
		rc = c->hip(hipMemcpyAsync(c->pin_out, c->d_out, nf * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream),
					"D2H out");
		if (rc)
		{
			return rc;
		}
	}
	rc = c->hip(hipStreamSynchronize(c->stream), "sync");
	if (rc)
	{
		return rc;
	}
	const double* h_out = c->pin_out;
	for (size_t i = 0; i < nf; ++i)
	{
		if (modes && modes[i] == 0)
		{
			continue;
		}
		r[i] = h_out[3 * i];
		if (jac)
		{
			jac[2 * i] = h_out[3 * i + 1];
			jac[2 * i + 1] = h_out[3 * i + 2];
		}
	}
	return EBO_OK;
}

// The device behind the lock-step drivers of lockstep.h (which are free of HIP and run under the
// sanitizers on the CPU with another backend).
struct CtxLockstepBackend
{
	explicit CtxLockstepBackend(ebo_ctx* ctx) : c(ctx) {}
	int eval(const double* flows, double* r, double* J, const unsigned char* modes, int windowSlots = 0)
	{
		return eval_host(c, flows, r, J, modes, windowSlots);
	}
	bool pipelined(int windows, size_t slots) const
	{
		// (rounds of up to kLiveMax windows per group are kernel-argument lists over the pinned buffers: no copies to
		// amortise, so two groups in flight pay from four windows up -- 4 windows 2.3 -> 1.85 ms, 16 windows 4.1 -> 3.3 ms, 32 windows
		// 9.2 -> 6.6 ms; until round 3 the threshold was 16 windows AND more flow slots than the zero-copy limit)
		(void)slots;
		return windows >= static_cast<int>(ab_size("EBO_SOLVE_PIPELINE_MIN", 4)) && !ab_env("EBO_SOLVE_NO_PIPELINE");
	}
	int groups() const { return static_cast<int>(ab_size("EBO_SOLVE_GROUPS", 2)); }
	int pipeline_begin(size_t slots, int G)
	{
		int rc = ensure_eval_staging(c, slots);
		if (rc == EBO_OK) rc = ensure_device_modes(c, slots);
		if (rc)
		{
			return rc;
		}
		for (int g = 0; g < G; ++g)
		{
			if (hipEventCreateWithFlags(&done[g], hipEventDisableTiming) != hipSuccess)
			{
				for (int k = 0; k < g; ++k)
				{
					(void)hipEventDestroy(done[k]);
					done[k] = nullptr;
				}
				return c->fail(EBO_ERR_HIP, "hipEventCreate");
			}
		}
		return EBO_OK;
	}
	void pipeline_end(int G)
	{
		(void)hipStreamSynchronize(c->stream);
		for (int g = 0; g < G; ++g)
		{
			if (done[g])
			{
				(void)hipEventDestroy(done[g]);
				done[g] = nullptr;
			}
		}
	}
	int eval_begin(const double* flows, const unsigned char* modes, int g, size_t s0, size_t s1, bool wantJac, int windowSlots)
	{
		return ebo_host::eval_begin(c, flows, modes, g, s0, s1, wantJac, done[g], windowSlots);
	}
	int eval_finish(const unsigned char* modes, int g, size_t s0, size_t s1, bool wantJac, double* r, double* J, int windowSlots)
	{
		return ebo_host::eval_finish(c, modes, s0, s1, wantJac, r, J, done[g], windowSlots);
	}
	ebo_ctx* c;
	hipEvent_t done[4] = {nullptr, nullptr, nullptr, nullptr};
};

SolveConsts make_solve_consts(const ebo_solver_opts* o)
{
	SolveConsts s;
	s.max_num_iterations = o->max_num_iterations;
	s.max_nonmono = o->use_nonmonotonic ? o->max_consecutive_nonmonotonic : 0;
	s.max_invalid = o->max_consecutive_invalid;
	s.jacobi_scaling = o->jacobi_scaling;
	s.function_tolerance = o->function_tolerance;
	s.gradient_tolerance = o->gradient_tolerance;
	s.parameter_tolerance = o->parameter_tolerance;
	s.initial_radius = o->initial_radius;
	s.max_radius = o->max_radius;
	s.min_radius = o->min_radius;
	s.min_relative_decrease = o->min_relative_decrease;
	s.min_lm_diagonal = o->min_lm_diagonal;
	s.max_lm_diagonal = o->max_lm_diagonal;
	return s;
}

int run_solve_device(ebo_ctx* c, const ebo_solver_opts* o, double* d_flows_out, int32_t* d_stats)
{
	if (c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	if (c->prm.grad != EBO_GRAD_JET)
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "device solver is built for EBO_GRAD_JET");
	}
	if (c->prm.loss == EBO_LOSS_EDGE)
	{
		// the reference's own objective (calculateEdgeLoss): the same launch geometry as its
		// batched evaluation, the whole per-patch LM inside the workgroup
		EdgeLaunch E;
		E.d_modes = nullptr;
		E.d_flows = nullptr;
		E.want_jac = 1;
		E.flow_sets = 1;
		E.fd_step = 0.0;
		E.d_out = nullptr;
		E.for_solve = true;
		int rce = edge_launch_setup(c, E);
		if (rce)
		{
			return rce;
		}
		if (launch_solve_edge(E, make_solve_consts(o), d_flows_out, d_stats, c->stream))
		{
			return c->hip(hipGetLastError(), "edge solve launch");
		}
		return EBO_OK;
	}
	SolveLaunch L;
	L.d_events = c->d_events;
	L.d_units = c->d_units;
	L.n_units = static_cast<int>(c->units.size());
	L.impl = std::max(1, eval_impl());
	// k_solve_independent runs three waves per SIMD (152-166 VGPRs since round 3: one next_step() site in the
	// solver; 205 and two waves before), i.e. 12 waves per CU.  By the canvas of the regular patch, as the
	// batched evaluation does: up to 32 KB six workgroups of 128 lanes with 26 KB each, above that four of 192
	// lanes with 39 KB (fewer sub-bands near convergence).  Measured against 128 lanes x 40 KB at two waves
	// (tools/ab/solve_waves.sh): C3 x 64 windows 25.1 -> 21.8 ms, 256 windows of 20x20 patches 12.5 -> 9.5 ms,
	// C2 x 64 8.4 -> 7.7 ms, C4 x 4 10.1 -> 9.1 ms.
	const int regW = c->reg_rw, regH = c->reg_rh;
	const bool smallCanvas = static_cast<size_t>(9) * regW * regH * sizeof(double) <= 32 * 1024;
	int rc = image_capacity(c, L.cap_doubles, L.lds_bytes, smallCanvas ? 26 : 39);
	if (rc)
	{
		return rc;
	}
	L.block = static_cast<int>(ab_size("EBO_SOLVE_BLOCK", smallCanvas ? 128 : 192));
	if (L.block < 64 || L.block > 512 || (L.block & 63))
	{
		return c->fail(EBO_ERR_ARG, "EBO_SOLVE_BLOCK must be a multiple of 64 in [64,512]");
	}
	L.d_flows_out = d_flows_out;
	L.d_stats = d_stats;
	L.c = make_consts(c);
	L.s = make_solve_consts(o);
	if (launch_solve_independent(L, c->stream))
	{
		return c->hip(hipGetLastError(), "solve launch");
	}
	return EBO_OK;
}

int count_device(ebo_ctx* c, int mode, const void* d_aux, double* d_image)
{
	if (c->custom_n)
	{
		return c->fail(EBO_ERR_STATE, "count images need a window (ebo_set_window), not ebo_set_patches");
	}
	CountLaunch L;
	L.d_events = c->d_events;
	L.d_units = c->d_units;
	L.n_units_total = static_cast<int>(c->units.size());
	L.n_windows = c->n_windows;
	L.units_per_window = c->P + 1;
	L.mode = mode;
	{
		const char* v = ab_env("EBO_COUNT_IMPL");
		L.impl = (v && *v) ? std::atoi(v) : -1;
	}
	L.lds_kb = static_cast<int>(std::min<size_t>(ab_size("EBO_COUNT_LDS_KB", 0), 160));
	L.max_window_events = 0;
	for (const WindowInfo& wi : c->windows)
	{
		L.max_window_events = std::max<uint64_t>(L.max_window_events, wi.n_events);
	}
	for (int w = 0; w < c->n_windows && !L.any_stray; ++w)
	{
		L.any_stray = c->units[static_cast<size_t>(w) * (c->P + 1) + c->P].n_ev > 0;
	}
	L.d_aux = d_aux;
	L.d_counts = c->d_counts;
	L.d_image = d_image;
	L.c = make_consts(c);
	L.d_overflow = nullptr;
	L.d_sort_bins = nullptr;
	L.sort_bins_cap = 0;
	L.d_sorted = nullptr;
	L.sorted_cap = 0;
	L.d_unit_maxdt = c->d_unit_maxdt;
	if (L.impl < 0 || L.impl == 2 || L.impl == 3)
	{
		size_t total = 0;
		for (const WindowInfo& wi : c->windows)
		{
			total += wi.n_events;
		}
		const size_t need = (total + 1) * sizeof(unsigned long long);
		if (need > c->count_ovf_cap)
		{
			if (c->d_count_ovf)
			{
				hipFree(c->d_count_ovf);
				c->d_count_ovf = nullptr;
				c->count_ovf_cap = 0;
			}
			const size_t cap = std::max(need, (static_cast<size_t>(c->prm.max_events) + 1) * sizeof(unsigned long long));
			int rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_count_ovf), cap), "hipMalloc count overflow list");
			if (rc)
			{
				return rc;
			}
			c->count_ovf_cap = cap;
		}
		L.d_overflow = c->d_count_ovf;
		if (mode != EBO_COUNT_INTEGRATED)
		{
			// sorted bands: the same buffer holds the 4-byte destination list; bins: at most one
			// band per image row
			const size_t bins = static_cast<size_t>(c->n_windows) * c->prm.image_h;
			if (bins > c->count_bins_cap)
			{
				if (c->d_count_bins)
				{
					hipFree(c->d_count_bins);
					c->d_count_bins = nullptr;
					c->count_bins_cap = 0;
				}
				int rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_count_bins), (3 * bins + 2) * sizeof(unsigned int)),
								"hipMalloc count bins");
				if (rc)
				{
					return rc;
				}
				c->count_bins_cap = bins;
			}
			L.d_sort_bins = c->d_count_bins;
			L.sort_bins_cap = static_cast<int>(std::min<size_t>(c->count_bins_cap, 1u << 30));
			L.d_sorted = reinterpret_cast<unsigned int*>(c->d_count_ovf);
			L.sorted_cap = c->count_ovf_cap / sizeof(unsigned long long);  // the buffer holds 8 B per event: two 4 B lists
		}
	}
	if (launch_count_image(L, c->stream))
	{
		return c->hip(hipGetLastError(), "count launch");
	}
	return EBO_OK;
}

int check_solver_opts(ebo_ctx* c, const ebo_solver_opts* o)
{
	if (!o)
	{
		return c->fail(EBO_ERR_ARG, "solver options are null");
	}
	if (o->max_num_iterations < 0 || !(o->initial_radius > 0) || o->max_consecutive_invalid < 1)
	{
		return c->fail(EBO_ERR_ARG, "bad solver options");
	}
	if (o->mode != EBO_SOLVE_GLOBAL && o->mode != EBO_SOLVE_INDEPENDENT)
	{
		return c->fail(EBO_ERR_ARG, "unknown solver mode");
	}
	return EBO_OK;
}

// EBO_SOLVE_GLOBAL: one trust-region LM per window on the host (as the
// reference: one ceres::Problem incl. TV blocks), windows advanced in lock step
// so that every round is ONE batched device evaluation of all data terms.
int solve_global(ebo_ctx* c, const ebo_solver_opts* o, double* flows_out, ebo_summary* summary)
{
	const int Wn = c->n_windows;
	const int P = c->P;
	std::vector<HostLm> lm;
	lm.reserve(Wn);
	for (int w = 0; w < Wn; ++w)
	{
		std::vector<uint8_t> active(P);
		for (int p = 0; p < P; ++p)
		{
			active[p] = (c->units[static_cast<size_t>(w) * (P + 1) + p].flags & kUnitActive) ? 1 : 0;
		}
		lm.emplace_back(c->npx, c->npy, active, c->prm.tv_weight, c->prm.tv_huber, *o);
	}
	std::vector<double> flows(static_cast<size_t>(Wn) * P * 2, 0.0);
	CtxLockstepBackend backend(c);
	const int rcl = lockstep_global(backend, Wn, P, lm, flows, std::getenv("EBO_SOLVE_TRACE") != nullptr);
	if (rcl)
	{
		return rcl;
	}
	int worst = 0;
	for (int w = 0; w < Wn; ++w)
	{
		lm[w].result(flows_out + static_cast<size_t>(w) * P * 2);
		const HostLm::Stats& s = lm[w].stats();
		worst = std::max(worst, s.termination);
		if (summary)
		{
			// counts are per data term (one contrastFunctor evaluation each), as the
			// per-patch solver reports them
			int nActive = 0;
			for (int p = 0; p < P; ++p)
			{
				nActive += (c->units[static_cast<size_t>(w) * (P + 1) + p].flags & kUnitActive) ? 1 : 0;
			}
			summary[w].iterations = s.iterations;
			summary[w].num_evals_cost = s.evals_cost * nActive;
			summary[w].num_evals_jac = s.evals_jac * nActive;
			summary[w].termination = s.termination;
			summary[w].initial_cost = s.initial_cost;
			summary[w].final_cost = s.final_cost;
		}
	}
	(void)worst;
	return EBO_OK;
}

// EBO_SOLVE_INDEPENDENT when the device-resident solver does not cover the loss (edge
// loss): one 2-parameter LM per active patch on the host, all of them advanced in lock
// step so that each round is ONE batched device evaluation.
int solve_independent_lockstep(ebo_ctx* c, const ebo_solver_opts* o, double* flows_out,
							   ebo_summary* summary)
{
	const size_t nf = c->n_flows();
	const int np = c->cur_patches();
	std::vector<HostLm> lms;
	std::vector<size_t> slot;  // flow index of each LM
	const std::vector<uint8_t> one(1, 1);
	for (int w = 0; w < c->n_windows; ++w)
	{
		for (int p = 0; p < np; ++p)
		{
			if (c->units[c->unit_index(w, p)].flags & kUnitActive)
			{
				lms.emplace_back(1, 1, one, 0.0, c->prm.tv_huber, *o);
				slot.push_back(static_cast<size_t>(w) * np + p);
			}
		}
	}
	std::vector<double> flows(nf * 2, 0.0);
	CtxLockstepBackend backend(c);
	const int rcl = lockstep_independent(backend, lms, slot, nf, flows);
	if (rcl)
	{
		return rcl;
	}
	std::fill(flows_out, flows_out + nf * 2, 0.0);
	if (summary)
	{
		std::memset(summary, 0, sizeof(ebo_summary) * c->n_windows);
	}
	for (size_t k = 0; k < lms.size(); ++k)
	{
		lms[k].result(flows_out + 2 * slot[k]);
		if (summary)
		{
			const HostLm::Stats& s = lms[k].stats();
			ebo_summary& d = summary[slot[k] / np];
			d.iterations = std::max(d.iterations, s.iterations);
			d.num_evals_cost += s.evals_cost;
			d.num_evals_jac += s.evals_jac;
			d.termination = std::max(d.termination, s.termination);
			d.initial_cost += s.initial_cost;
			d.final_cost += s.final_cost;
		}
	}
	return EBO_OK;
}

int solve_independent_host(ebo_ctx* c, const ebo_solver_opts* o, double* flows_out, ebo_summary* summary)
{
	// device-resident per-patch LM for both losses (forward-mode-equivalent Jacobians);
	// central differences and EBO_SOLVE_EDGE=lockstep (A/B) run host LMs over batched evaluations
	// Edge loss, measured (tools/time_edge_solve.py, round 3: solver state in LDS, the LM on one lane,
	// spill-free instantiations): one reference-default window 2.1 ms on the device against 4.3 ms in
	// lock step (67 evaluations = 67 round trips); 256 windows 78 against 161-172 ms; C2 x 64 windows 33
	// against 61-65 ms; C3 x 16 windows 21 against 44 ms.  The device solve at every size (round 2 lost
	// to lock step from 2048 units up: its solver ran replicated in every wave and cost the workgroup
	// as many vector instructions as the evaluations themselves).  EBO_SOLVE_EDGE=lockstep forces the
	// host LMs (A/B); ebo_solve_device is always the device solve.
	const char* edgeMode = ab_env("EBO_SOLVE_EDGE");
	bool lockstepEdge = false;
	if (c->prm.loss == EBO_LOSS_EDGE && edgeMode && *edgeMode)
	{
		lockstepEdge = std::strcmp(edgeMode, "lockstep") == 0;
	}
	if (c->prm.grad != EBO_GRAD_JET || lockstepEdge)
	{
		return solve_independent_lockstep(c, o, flows_out, summary);
	}
	const size_t nf = c->n_flows();
	int rc = run_solve_device(c, o, c->d_flows, c->d_stats);
	if (rc)
	{
		return rc;
	}
	std::vector<int32_t> st(nf * 4);
	rc = c->hip(hipMemcpyAsync(flows_out, c->d_flows, nf * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H flows");
	if (rc)
	{
		return rc;
	}
	rc = c->hip(hipMemcpyAsync(st.data(), c->d_stats, nf * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream), "D2H stats");
	if (rc)
	{
		return rc;
	}
	rc = c->hip(hipStreamSynchronize(c->stream), "sync");
	if (rc)
	{
		return rc;
	}
	if (summary)
	{
		for (int w = 0; w < c->n_windows; ++w)
		{
			ebo_summary s;
			std::memset(&s, 0, sizeof(s));
			const int np = c->cur_patches();
			for (int p = 0; p < np; ++p)
			{
				const int32_t* q = &st[(static_cast<size_t>(w) * np + p) * 4];
				s.iterations = std::max(s.iterations, q[0]);
				s.num_evals_cost += q[1];
				s.num_evals_jac += q[2];
				s.termination = std::max(s.termination, q[3]);
			}
			summary[w] = s;
		}
	}
	return EBO_OK;
}

}  // namespace ebo_host

namespace ebo_host
{
thread_local std::string g_create_error;  // what ebo_last_error(NULL) reports
}

extern "C" {

const char* ebo_version(void)
{
	return "ebo_hip 0.1 (gfx950)";
}

int ebo_device_count(int* n)
{
	if (!n)
	{
		return EBO_ERR_ARG;
	}
	int cnt = 0;
	if (hipGetDeviceCount(&cnt) != hipSuccess)
	{
		cnt = 0;
		(void)hipGetLastError();
	}
	*n = cnt;
	return EBO_OK;
}

void ebo_default_params(ebo_params* p)
{
	if (!p)
	{
		return;
	}
	std::memset(p, 0, sizeof(*p));
	p->device = 0;
	p->image_w = 240;
	p->image_h = 180;
	p->patch_w = 20;
	p->patch_h = 20;
	p->tv_weight = 1e3;
	p->tv_huber = 10;
	p->scale = 1e-3;
	p->min_events = 100;
	p->loss = EBO_LOSS_EDGE;
	p->grad = EBO_GRAD_JET;
	p->fd_step = 1e-6;
	p->k.max_possible_residual = 1e3;
	p->k.sigma_compensate = 1.0;
	p->k.kernel_compensate = 3;
	p->k.kernel_st = 3;
	p->k.sigma_st = 1.5;
	p->k.kernel_nms = 2;
	p->max_events = 1u << 20;
	p->max_windows = 1;
}

void ebo_default_solver(ebo_solver_opts* o)
{
	if (!o)
	{
		return;
	}
	o->max_num_iterations = 50;
	o->use_nonmonotonic = 1;
	o->function_tolerance = 1e-12;
	o->gradient_tolerance = 1e-12;
	o->parameter_tolerance = 1e-12;
	o->initial_radius = 1e4;
	o->max_radius = 1e16;
	o->min_radius = 1e-32;
	o->min_relative_decrease = 1e-3;
	o->min_lm_diagonal = 1e-6;
	o->max_lm_diagonal = 1e32;
	o->max_consecutive_nonmonotonic = 5;
	o->max_consecutive_invalid = 5;
	o->jacobi_scaling = 1;
	o->mode = EBO_SOLVE_GLOBAL;
}

const char* ebo_last_error(const ebo_ctx* ctx)
{
	return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int ebo_create(const ebo_params* p, ebo_ctx** out)
{
	if (!p || !out)
	{
		g_create_error = "null argument";
		return EBO_ERR_ARG;
	}
	*out = nullptr;
	if (p->image_w <= 0 || p->image_h <= 0 || p->patch_w <= 0 || p->patch_h <= 0 ||
		p->patch_w > p->image_w || p->patch_h > p->image_h || p->image_w > kCoordMax ||
		p->image_h > kCoordMax || p->max_events == 0 || p->max_windows <= 0 ||
		!(p->k.sigma_compensate > 0))
	{
		g_create_error = "bad image/patch geometry or capacity";
		return EBO_ERR_ARG;
	}
	if (p->k.kernel_compensate != 3 || p->k.kernel_st != 3 || p->k.kernel_nms != 2)
	{
		g_create_error = "only kernel sizes 3/3/2 (the reference's constants) are built";
		return EBO_ERR_UNSUPPORTED;
	}
	if ((p->loss != EBO_LOSS_EDGE && p->loss != EBO_LOSS_VARIANCE) ||
		(p->grad != EBO_GRAD_JET && p->grad != EBO_GRAD_CENTRAL))
	{
		g_create_error = "unknown loss or gradient mode";
		return EBO_ERR_ARG;
	}
	if (p->max_events >= (1ull << 32) - (1ull << 20))  // 32-bit event indices; kernels step in chunks of up to 2^14
	{
		g_create_error = "max_events must be below 2^32 - 2^20";
		return EBO_ERR_ARG;
	}
	int cnt = 0;
	if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
	{
		(void)hipGetLastError();
		g_create_error = "no HIP device (libebo_hip has no CPU path)";
		return EBO_ERR_NO_DEVICE;
	}
	if (p->device < 0 || p->device >= cnt)
	{
		g_create_error = "device ordinal out of range";
		return EBO_ERR_NO_DEVICE;
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, p->device) != hipSuccess)
	{
		g_create_error = "hipGetDeviceProperties failed";
		return EBO_ERR_HIP;
	}
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
	{
		g_create_error = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
		return EBO_ERR_NO_DEVICE;
	}
	if (hipSetDevice(p->device) != hipSuccess)
	{
		g_create_error = "hipSetDevice failed";
		return EBO_ERR_HIP;
	}
	ebo_ctx* c = new (std::nothrow) ebo_ctx();
	if (!c)
	{
		g_create_error = "out of host memory";
		return EBO_ERR_ARG;
	}
	c->prm = *p;
	c->npx = p->image_w / p->patch_w;  // feature_detector.cpp:301-304
	c->npy = p->image_h / p->patch_h;
	c->P = c->npx * c->npy;
	c->cap_events = p->max_events;
	c->cap_windows = p->max_windows;
	for (int py = 0; py < c->npy; ++py)
	{
		for (int px = 0; px < c->npx; ++px)
		{
			int x, y, w, h;
			rect_of(c, px, py, x, y, w, h);
			c->max_rw = std::max(c->max_rw, w);
			c->max_rh = std::max(c->max_rh, h);
		}
	}
	c->grid_max_rw = c->max_rw;
	c->grid_max_rh = c->max_rh;
	c->reg_rw = p->patch_w;
	c->reg_rh = p->patch_h;
	const size_t nf = static_cast<size_t>(c->cap_windows) * c->P;
	const size_t npix = static_cast<size_t>(c->cap_windows) * p->image_w * p->image_h;
	hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	c->own_stream = (e == hipSuccess);
	if (e == hipSuccess) e = hipMalloc(&c->d_events, c->cap_events * sizeof(uint64_t));
	if (e == hipSuccess) e = hipMalloc(&c->d_units, static_cast<size_t>(c->cap_windows) * (c->P + 1) * sizeof(Unit));
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_unit_maxdt), static_cast<size_t>(c->cap_windows) * (c->P + 1) * sizeof(int32_t));
	if (e == hipSuccess) e = hipMalloc(&c->d_flows, nf * 2 * sizeof(double));
	if (e == hipSuccess) e = hipMalloc(&c->d_out, nf * 3 * sizeof(double));
	if (e == hipSuccess) e = hipMalloc(&c->d_stats, nf * 4 * sizeof(int32_t));
	if (e == hipSuccess) e = hipMalloc(&c->d_counts, npix * sizeof(int32_t));
	if (e == hipSuccess) e = hipMalloc(&c->d_image, npix * sizeof(double));
	if (e == hipSuccess) e = hipMemset(c->d_counts, 0, npix * sizeof(int32_t));
	if (e == hipSuccess) e = hipEventCreate(&c->ev0);
	if (e == hipSuccess) e = hipEventCreate(&c->ev1);
	if (e != hipSuccess)
	{
		g_create_error = std::string("device allocation failed: ") + hipGetErrorString(e);
		ebo_destroy(c);
		return EBO_ERR_HIP;
	}
	*out = c;
	return EBO_OK;
}

void ebo_destroy(ebo_ctx* c)
{
	if (c && c->copy_stream)
	{
		(void)hipSetDevice(c->prm.device);
		(void)hipStreamSynchronize(c->copy_stream);
		for (hipEvent_t& e : c->copy_done)
		{
			if (e)
			{
				(void)hipEventDestroy(e);
			}
		}
		(void)hipStreamDestroy(c->copy_stream);
		c->copy_stream = nullptr;
	}
	if (!c)
	{
		return;
	}
	(void)hipSetDevice(c->prm.device);
	if (c->capturing)
	{
		// a context destroyed with a recording still open: end it first (a synchronisation inside a recording
		// would leave the stream unusable for the rest of the process)
		hipGraph_t g = nullptr;
		if (hipStreamEndCapture(c->stream, &g) == hipSuccess && g)
		{
			(void)hipGraphDestroy(g);
		}
		(void)hipGetLastError();
		c->capturing = false;
	}
	if (c->stream)
	{
		(void)hipStreamSynchronize(c->stream);
	}
	(void)ebo_comm_destroy(c);
	hipFree(c->d_events);
	hipFree(c->d_shard_tbl);
	hipFree(c->d_units);
	hipFree(c->d_unit_maxdt);
	hipFree(c->d_flows);
	hipFree(c->d_out);
	if (c->pin_flows)
	{
		(void)hipHostFree(c->pin_flows);
		(void)hipHostFree(c->pin_out);
		(void)hipHostFree(c->pin_modes);
	}
	if (c->pin_bucket)
	{
		(void)hipHostFree(c->pin_bucket);
	}
	if (c->pin_route)
	{
		(void)hipHostFree(c->pin_route);
	}
	hipFree(c->d_route_xy);
	hipFree(c->d_partials);
	hipFree(c->d_counts);
	hipFree(c->d_count_ovf);
	hipFree(c->d_count_bins);
	hipFree(c->d_opt_grid);
	hipFree(c->d_modes);
	hipFree(c->d_opt);
	hipFree(c->d_image);
	hipFree(c->d_aux);
	hipFree(c->d_stats);
	hipFree(c->d_scratch);
	hipFree(c->d_edge_scratch);
	hipFree(c->d_edge_w);
	hipFree(c->d_edge_cs);
	hipFree(c->d_edge_defer);
	hipFree(c->d_raw);
	hipFree(c->d_bucket);
	hipFree(c->d_chunk_hist);
	hipFree(c->d_field);
	hipFree(c->d_tvf);
	if (c->ev0) hipEventDestroy(c->ev0);
	if (c->ev1) hipEventDestroy(c->ev1);
	if (c->own_stream && c->stream)
	{
		hipStreamDestroy(c->stream);
	}
	delete c;
}

int ebo_set_stream(ebo_ctx* c, void* hip_stream)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (c->own_stream && c->stream)
	{
		(void)hipStreamSynchronize(c->stream);
		hipStreamDestroy(c->stream);
	}
	c->stream = static_cast<hipStream_t>(hip_stream);
	c->own_stream = false;
	return EBO_OK;
}

int ebo_synchronize(ebo_ctx* c)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	return c->hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
}

int ebo_grid(const ebo_ctx* c, int* npx, int* npy)
{
	if (!c || !npx || !npy)
	{
		return EBO_ERR_ARG;
	}
	*npx = c->npx;
	*npy = c->npy;
	return EBO_OK;
}

int ebo_patch_rect(const ebo_ctx* c, int px, int py, int* x, int* y, int* w, int* h)
{
	if (!c || !x || !y || !w || !h || px < 0 || py < 0 || px >= c->npx || py >= c->npy)
	{
		return EBO_ERR_ARG;
	}
	rect_of(c, px, py, *x, *y, *w, *h);
	return EBO_OK;
}

int ebo_eval(ebo_ctx* c, const double* flows, double* r, double* jac)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!flows || !r)
	{
		return c->fail(EBO_ERR_ARG, "null flows or residual pointer");
	}
	if (c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	(void)hipSetDevice(c->prm.device);
	return eval_host(c, flows, r, jac);
}

int ebo_eval_device(ebo_ctx* c, const double* d_flows, int want_jac, double* d_out)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (!d_flows || !d_out)
	{
		return c->fail(EBO_ERR_ARG, "null device pointer");
	}
	return run_eval_device(c, d_flows, want_jac, d_out);
}

int ebo_contrast_image(ebo_ctx* c, int window, int patch, const double* flow, int channels,
					   double* image)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!flow || !image || (channels != 1 && channels != 3) || window < 0 ||
		window >= c->n_windows || patch < 0 || patch >= c->cur_patches())
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_contrast_image");
	}
	(void)hipSetDevice(c->prm.device);
	const size_t ui = c->unit_index(window, patch);
	const Unit& u = c->units[ui];
	const size_t npx = static_cast<size_t>(9) * u.rw * u.rh;
	int rc = ensure_aux(c, npx * channels * sizeof(double) + 2 * sizeof(double));
	if (rc)
	{
		return rc;
	}
	double* d_img = static_cast<double*>(c->d_aux);
	double* d_flow = d_img + npx * channels;
	rc = c->hip(hipMemcpyAsync(d_flow, flow, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream), "H2D flow");
	if (rc)
	{
		return rc;
	}
	EvalLaunch L;
	std::memset(&L, 0, sizeof(L));
	L.d_events = c->d_events;
	L.d_units = c->d_units;
	L.d_flows = d_flow;
	L.channels = channels;
	L.tiles = min_tiles(channels, u.rw, u.rh, kLdsBudget);
	if (L.tiles < 0)
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "patch too wide for LDS row tiling");
	}
	L.block = 256;
	L.lds_bytes = lds_for(channels, L.tiles, u.rw, u.rh);
	L.c = make_consts(c);
	if (launch_dump_image(L, static_cast<int>(ui), d_img, c->stream))
	{
		return c->hip(hipGetLastError(), "dump launch");
	}
	rc = c->hip(hipMemcpyAsync(image, d_img, npx * channels * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H image");
	if (rc)
	{
		return rc;
	}
	return c->hip(hipStreamSynchronize(c->stream), "sync");
}

int ebo_solve(ebo_ctx* c, const ebo_solver_opts* o, double* flows_out, ebo_summary* summary)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!flows_out)
	{
		return c->fail(EBO_ERR_ARG, "null flows_out");
	}
	int rc = check_solver_opts(c, o);
	if (rc)
	{
		return rc;
	}
	if (c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	(void)hipSetDevice(c->prm.device);
	if (o->mode == EBO_SOLVE_INDEPENDENT)
	{
		return solve_independent_host(c, o, flows_out, summary);
	}
	if (c->custom_n)
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "EBO_SOLVE_GLOBAL needs the patch grid of a window");
	}
	return solve_global(c, o, flows_out, summary);
}

int ebo_solve_device(ebo_ctx* c, const ebo_solver_opts* o, double* d_flows_out, int32_t* d_stats)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (!d_flows_out)
	{
		return c->fail(EBO_ERR_ARG, "null device pointer");
	}
	int rc = check_solver_opts(c, o);
	if (rc)
	{
		return rc;
	}
	if (o->mode != EBO_SOLVE_INDEPENDENT)
	{
		return c->fail(EBO_ERR_UNSUPPORTED, "ebo_solve_device runs EBO_SOLVE_INDEPENDENT only");
	}
	return run_solve_device(c, o, d_flows_out, d_stats);
}

int ebo_count_image(ebo_ctx* c, int mode, const void* aux, double* image)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	// EBO_COUNT_FIELD with aux == NULL: the field left on the device by
	// ebo_init_motion_field (one window only: the field belongs to the context)
	const bool residentField = mode == EBO_COUNT_FIELD && !aux && c->field_valid && c->n_windows == 1;
	if (!image || mode < 0 || mode > 2 || (mode != EBO_COUNT_INTEGRATED && !aux && !residentField))
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_count_image");
	}
	if (c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	(void)hipSetDevice(c->prm.device);
	const size_t npix = static_cast<size_t>(c->n_windows) * c->prm.image_w * c->prm.image_h;
	const void* d_aux = nullptr;
	int rc = EBO_OK;
	if (mode == EBO_COUNT_WARPED)
	{
		const size_t bytes = static_cast<size_t>(c->n_windows) * c->P * 2 * sizeof(double);
		rc = c->hip(hipMemcpyAsync(c->d_flows, aux, bytes, hipMemcpyHostToDevice, c->stream), "H2D flows");
		d_aux = c->d_flows;
	}
	else if (residentField)
	{
		d_aux = c->d_field;
	}
	else if (mode == EBO_COUNT_FIELD)
	{
		const size_t bytes = npix * 2 * sizeof(float);
		rc = ensure_aux(c, bytes);
		if (rc == EBO_OK)
		{
			rc = c->hip(hipMemcpyAsync(c->d_aux, aux, bytes, hipMemcpyHostToDevice, c->stream), "H2D field");
		}
		d_aux = c->d_aux;
	}
	if (rc)
	{
		return rc;
	}
	rc = count_device(c, mode, d_aux, c->d_image);
	if (rc)
	{
		return rc;
	}
	rc = c->hip(hipMemcpyAsync(image, c->d_image, npix * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H image");
	if (rc)
	{
		return rc;
	}
	return c->hip(hipStreamSynchronize(c->stream), "sync");
}

int ebo_count_image_device(ebo_ctx* c, int mode, const void* d_aux, double* d_image)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (!d_image || mode < 0 || mode > 2 || (mode != EBO_COUNT_INTEGRATED && !d_aux))
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_count_image_device");
	}
	if (c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	return count_device(c, mode, d_aux, d_image);
}

int ebo_edge_work_stats(ebo_ctx* c, const double* d_flows, int want_jac, uint64_t* out)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!d_flows || !out)
	{
		return c->fail(EBO_ERR_ARG, "null pointer");
	}
	if (c->prm.loss != EBO_LOSS_EDGE || c->n_windows == 0)
	{
		return c->fail(EBO_ERR_STATE, "needs the edge loss and a loaded window");
	}
	(void)hipSetDevice(c->prm.device);
	int rc = ensure_aux(c, 6 * sizeof(unsigned long long));
	if (rc)
	{
		return rc;
	}
	rc = c->hip(hipMemsetAsync(c->d_aux, 0, 6 * sizeof(unsigned long long), c->stream), "zero edge stats");
	if (rc)
	{
		return rc;
	}
	c->edge_stats_dev = static_cast<unsigned long long*>(c->d_aux);
	rc = run_eval_device(c, d_flows, want_jac, c->d_out);
	c->edge_stats_dev = nullptr;
	if (rc)
	{
		return rc;
	}
	unsigned long long h[6];
	rc = c->hip(hipMemcpyAsync(h, c->d_aux, sizeof(h), hipMemcpyDeviceToHost, c->stream), "D2H edge stats");
	if (rc == EBO_OK)
	{
		rc = c->hip(hipStreamSynchronize(c->stream), "sync");
	}
	for (int k = 0; k < 6 && rc == EBO_OK; ++k)
	{
		out[k] = h[k];
	}
	return rc;
}

int ebo_lds_rates(ebo_ctx* c, double* gops)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!gops)
	{
		return c->fail(EBO_ERR_ARG, "null output");
	}
	(void)hipSetDevice(c->prm.device);
	int cus = 256;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, c->prm.device) == hipSuccess && prop.multiProcessorCount > 0)
	{
		cus = prop.multiProcessorCount;
	}
	const int blocks = 4 * cus, iters = 49 * 42;  // 42 footprints per lane
	int rc = ensure_aux(c, static_cast<size_t>(blocks) * sizeof(double));
	if (rc)
	{
		return rc;
	}
	for (int kind = 0; kind < 2 && rc == EBO_OK; ++kind)
	{
		if (launch_lds_rate(kind == 0, blocks, 49, static_cast<double*>(c->d_aux), c->stream))  // warm-up
		{
			return c->fail(EBO_ERR_HIP, "k_lds_rate launch failed");
		}
		double best = 0.0;
		for (int rep = 0; rep < 3 && rc == EBO_OK; ++rep)
		{
			rc = c->hip(hipEventRecord(c->ev0, c->stream), "hipEventRecord");
			if (rc == EBO_OK && launch_lds_rate(kind == 0, blocks, iters, static_cast<double*>(c->d_aux), c->stream))
			{
				return c->fail(EBO_ERR_HIP, "k_lds_rate launch failed");
			}
			if (rc == EBO_OK) rc = c->hip(hipEventRecord(c->ev1, c->stream), "hipEventRecord");
			if (rc == EBO_OK) rc = c->hip(hipEventSynchronize(c->ev1), "hipEventSynchronize");
			float ms = 0.0f;
			if (rc == EBO_OK) rc = c->hip(hipEventElapsedTime(&ms, c->ev0, c->ev1), "hipEventElapsedTime");
			if (rc == EBO_OK && ms > 0.0f)
			{
				best = std::max(best, static_cast<double>(blocks) * 256.0 * iters / (ms * 1e-3) / 1e9);
			}
		}
		gops[kind] = best;
	}
	return rc;
}

int ebo_stream_yardstick_device(ebo_ctx* c, double* d_image, uint64_t* bytes)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!d_image)
	{
		return c->fail(EBO_ERR_ARG, "null image");
	}
	if (c->n_windows == 0 || c->custom_n)
	{
		return c->fail(EBO_ERR_STATE, "no window loaded");
	}
	size_t nEv = 0;
	for (const WindowInfo& wi : c->windows)
	{
		nEv += wi.n_events;
	}
	const size_t nPx = static_cast<size_t>(c->n_windows) * c->prm.image_w * c->prm.image_h;
	if ((reinterpret_cast<uintptr_t>(d_image) & 15) != 0)
	{
		return c->fail(EBO_ERR_ARG, "image must be 16-byte aligned");
	}
	if (bytes)
	{
		*bytes = (nEv / 2) * 16 + (nPx / 2) * 16;
	}
	if (launch_stream_yardstick(c->d_events, nEv, d_image, nPx, c->stream))
	{
		return c->fail(EBO_ERR_HIP, "k_stream_yardstick launch failed");
	}
	return EBO_OK;
}

int ebo_window_ref_time(int64_t t_first_us, int64_t t_last_us, int64_t* t_ref_us)
{
	if (!t_ref_us)
	{
		return EBO_ERR_ARG;
	}
	return mid_timestamp(t_first_us, t_last_us, *t_ref_us) ? EBO_OK : EBO_ERR_RANGE;
}

int ebo_count_image_shard_device(ebo_ctx* c, int n_windows, const int64_t* window_t_ref_us, const double* d_flows_grid,
								 double* d_image)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n_windows <= 0 || !window_t_ref_us || !d_flows_grid || !d_image)
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_count_image_shard");
	}
	(void)hipSetDevice(c->prm.device);
	// the per-unit times against the windows' reference times: built once per set of windows, kept on the device
	const BandUnit* tbl = nullptr;
	int rc = shard_table(c, n_windows, window_t_ref_us, &tbl, nullptr);
	if (rc)
	{
		return rc;
	}
	const int per = c->custom_n / n_windows;
	rc = c->hip(hipMemsetAsync(d_image, 0, static_cast<size_t>(n_windows) * c->prm.image_w * c->prm.image_h * sizeof(double),
							   c->stream),
				"zero shard image");
	if (rc)
	{
		return rc;
	}
	if (launch_count_shard(c->d_events, c->d_units, c->custom_n, per, tbl, d_flows_grid, d_image, make_consts(c), c->stream))
	{
		return c->fail(EBO_ERR_HIP, "k_count_shard launch failed");
	}
	return EBO_OK;
}

int ebo_count_image_shard(ebo_ctx* c, int n_windows, const int64_t* window_t_ref_us, const double* flows_grid, double* image)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n_windows <= 0 || !window_t_ref_us || !flows_grid || !image)
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_count_image_shard");
	}
	if (static_cast<size_t>(n_windows) > static_cast<size_t>(c->cap_windows))
	{
		return c->fail(EBO_ERR_ARG, "more windows than max_windows");
	}
	(void)hipSetDevice(c->prm.device);
	const size_t fbytes = static_cast<size_t>(n_windows) * c->P * 2 * sizeof(double);
	const size_t ibytes = static_cast<size_t>(n_windows) * c->prm.image_w * c->prm.image_h * sizeof(double);
	int rc = c->hip(hipMemcpyAsync(c->d_flows, flows_grid, fbytes, hipMemcpyHostToDevice, c->stream), "H2D flows");
	if (rc == EBO_OK)
	{
		rc = ebo_count_image_shard_device(c, n_windows, window_t_ref_us, c->d_flows, c->d_image);
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipMemcpyAsync(image, c->d_image, ibytes, hipMemcpyDeviceToHost, c->stream), "D2H image");
	}
	if (rc == EBO_OK)
	{
		rc = c->hip(hipStreamSynchronize(c->stream), "sync");
	}
	return rc;
}

int ebo_compensate_events_contrast(ebo_ctx* c, const ebo_event* ev, size_t n,
								   const ebo_solver_opts* o, double* flows_out,
								   double* image_out, ebo_summary* summary)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!ev || n == 0 || !flows_out)
	{
		return c->fail(EBO_ERR_ARG, "empty window or null output");
	}
	int rc = ebo_set_window(c, ev, n);
	if (rc)
	{
		return rc;
	}
	rc = ebo_solve(c, o, flows_out, summary);
	if (rc)
	{
		return rc;
	}
	if (image_out)
	{
		rc = ebo_count_image(c, EBO_COUNT_WARPED, flows_out, image_out);
	}
	return rc;
}

int ebo_timer_begin(ebo_ctx* c)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	return c->hip(hipEventRecord(c->ev0, c->stream), "hipEventRecord");
}

int ebo_timer_end(ebo_ctx* c, float* ms)
{
	if (!c || !ms)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	int rc = c->hip(hipEventRecord(c->ev1, c->stream), "hipEventRecord");
	if (rc) return rc;
	rc = c->hip(hipEventSynchronize(c->ev1), "hipEventSynchronize");
	if (rc) return rc;
	return c->hip(hipEventElapsedTime(ms, c->ev0, c->ev1), "hipEventElapsedTime");
}

}  // extern "C"
