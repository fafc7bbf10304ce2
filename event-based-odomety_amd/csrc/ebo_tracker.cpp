// ebo_tracker.cpp — the per-feature tracker entry points of include/ebo.h: Patch::integrateEvents /
// integrateMotionCompensatedEvents, the updatePatches routing, and the Optimizer (set_grad, eval, solve).
#include "ebo_ctx.h"

using namespace ebo;

extern "C" {

static int patch_integrate_common(ebo_ctx* c, const ebo_event* ev, const size_t* offsets,
								  int n_patches, const double* rects, const double* traj,
								  const int64_t* mid_time, const size_t* nabla_offsets,
								  double* nabla, int64_t* current_ts, int64_t* time_last_update,
								  int32_t* updated)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!ev || !offsets || n_patches <= 0 || !rects || !nabla_offsets || !nabla)
	{
		return c->fail(EBO_ERR_ARG, "null argument to patch integrate");
	}
	const bool mc = traj != nullptr;
	if (mc && (!mid_time || !updated))
	{
		return c->fail(EBO_ERR_ARG, "null mid_time/updated");
	}
	(void)hipSetDevice(c->prm.device);
	const size_t e0 = offsets[0];
	const size_t total = offsets[n_patches] - e0;
	if (total >= (1ull << 32))
	{
		return c->fail(EBO_ERR_ARG, "too many events");
	}
	std::vector<uint64_t> packed(total);
	std::vector<uint32_t> off32(n_patches + 1);
	std::vector<double> tr(mc ? static_cast<size_t>(n_patches) * 4 : 0);
	std::vector<uint64_t> noff(n_patches);
	size_t nablaEnd = 0;
	for (int p = 0; p < n_patches; ++p)
	{
		const size_t a = offsets[p], b = offsets[p + 1];
		if (b < a)
		{
			return c->fail(EBO_ERR_ARG, "offsets must be non-decreasing");
		}
		off32[p] = static_cast<uint32_t>(a - e0);
		const double rw = rects[4 * p + 2], rh = rects[4 * p + 3];
		const int cols = static_cast<int>(rw), rows = static_cast<int>(rh);
		if (cols <= 0 || rows <= 0 || static_cast<size_t>(cols) * rows > 16384)
		{
			return c->fail(EBO_ERR_UNSUPPORTED, "patch image must have 1..16384 pixels");
		}
		noff[p] = nabla_offsets[p];
		nablaEnd = std::max(nablaEnd, nabla_offsets[p] + static_cast<size_t>(cols) * rows);
		int64_t tref = 0;
		bool pass = true;
		if (mc)
		{
			// patch.cpp:94-100
			const int64_t preT = static_cast<int64_t>(traj[6 * p + 2]);
			const int64_t lastT = static_cast<int64_t>(traj[6 * p + 5]);
			const double halfD = static_cast<double>(lastT - preT) * 0.5;
			if (!(halfD > -2147483648.0 && halfD < 2147483648.0))
			{
				return c->fail(EBO_ERR_RANGE, "trajectory time step outside int32 microseconds");
			}
			const int64_t half = static_cast<int64_t>(static_cast<int32_t>(halfD));
			tref = mid_time[p];
			pass = (b > a) && (lastT + half >= tref) && (preT < tref);
			tr[4 * p + 0] = traj[6 * p + 3] - traj[6 * p + 0];
			tr[4 * p + 1] = traj[6 * p + 4] - traj[6 * p + 1];
			tr[4 * p + 2] = static_cast<double>(lastT - preT);
			tr[4 * p + 3] = pass ? 1.0 : 0.0;
			updated[p] = pass ? 1 : 0;
		}
		else if (b > a)
		{
			// patch.cpp:78-83
			int64_t mid;
			if (!mid_timestamp(ev[a].t_us, ev[b - 1].t_us, mid))
			{
				return c->fail(EBO_ERR_RANGE, "patch mid-time outside int32 microseconds");
			}
			if (current_ts) current_ts[p] = mid;
			if (time_last_update)
			{
				time_last_update[p] = static_cast<int64_t>(static_cast<int32_t>(ev[b - 1].t_us));
			}
		}
		for (size_t i = a; i < b; ++i)
		{
			if (ev[i].x < kCoordMin || ev[i].x > kCoordMax || ev[i].y < kCoordMin || ev[i].y > kCoordMax)
			{
				return c->fail(EBO_ERR_RANGE, "event coordinate outside [-16384,16383]");
			}
			const int64_t dt = mc ? (tref - ev[i].t_us) : 0;
			if (dt < INT32_MIN || dt > INT32_MAX)
			{
				return c->fail(EBO_ERR_RANGE, "event time further than 2^31 us from mid time");
			}
			packed[i - e0] = static_cast<uint64_t>(pack_lo(ev[i].x, ev[i].y, ev[i].sign > 0)) |
							 (static_cast<uint64_t>(static_cast<uint32_t>(static_cast<int32_t>(dt))) << 32);
		}
	}
	off32[n_patches] = static_cast<uint32_t>(total);
	// staging layout: events | offsets | rects | traj | nabla offsets | nabla
	auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
	const size_t bEv = align(total * 8), bOff = align(off32.size() * 4), bRect = align(static_cast<size_t>(n_patches) * 32);
	const size_t bTraj = align(tr.size() * 8), bNoff = align(noff.size() * 8), bNabla = align(nablaEnd * 8);
	int rc = ensure_scratch(c, bEv + bOff + bRect + bTraj + bNoff + bNabla);
	if (rc)
	{
		return rc;
	}
	char* base = static_cast<char*>(c->d_scratch);
	char* dEv = base;
	char* dOff = dEv + bEv;
	char* dRect = dOff + bOff;
	char* dTraj = dRect + bRect;
	char* dNoff = dTraj + bTraj;
	char* dNabla = dNoff + bNoff;
	hipError_t e = hipSuccess;
	if (total) e = hipMemcpyAsync(dEv, packed.data(), total * 8, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dOff, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dRect, rects, static_cast<size_t>(n_patches) * 32, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && mc) e = hipMemcpyAsync(dTraj, tr.data(), tr.size() * 8, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dNoff, noff.data(), noff.size() * 8, hipMemcpyHostToDevice, c->stream);
	// R6 leaves images of patches that fail the time test untouched: start from the caller's data
	if (e == hipSuccess && mc) e = hipMemcpyAsync(dNabla, nabla, nablaEnd * 8, hipMemcpyHostToDevice, c->stream);
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D patch data");
	}
	PatchIntLaunch L;
	L.d_events = reinterpret_cast<const uint64_t*>(dEv);
	L.d_offsets = reinterpret_cast<const uint32_t*>(dOff);
	L.d_rects = reinterpret_cast<const double*>(dRect);
	L.d_traj = mc ? reinterpret_cast<const double*>(dTraj) : nullptr;
	L.d_nabla_off = reinterpret_cast<const uint64_t*>(dNoff);
	L.d_nabla = reinterpret_cast<double*>(dNabla);
	L.n_patches = n_patches;
	if (launch_patch_integrate(L, c->stream))
	{
		return c->hip(hipGetLastError(), "patch integrate launch");
	}
	rc = c->hip(hipMemcpyAsync(nabla, dNabla, nablaEnd * 8, hipMemcpyDeviceToHost, c->stream), "D2H nabla");
	if (rc)
	{
		return rc;
	}
	return c->hip(hipStreamSynchronize(c->stream), "sync");
}

int ebo_route_set_events(ebo_ctx* c, const ebo_event* ev, size_t n)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if ((!ev && n > 0) || n > 0x7FFFFFFFu)  // k_route's 32-bit indices step by 256: no wrap-around
	{
		return c->fail(EBO_ERR_ARG, "null events or more than 2^31 events in a chunk");
	}
	(void)hipSetDevice(c->prm.device);
	c->route_n = 0;
	if (n > c->route_cap)
	{
		hipFree(c->d_route_xy);
		c->d_route_xy = nullptr;
		c->route_cap = 0;
		int rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_route_xy), n * sizeof(uint32_t)), "hipMalloc route events");
		if (rc)
		{
			return rc;
		}
		c->route_cap = n;
	}
	std::vector<uint32_t> xy(n);
	for (size_t i = 0; i < n; ++i)
	{
		if (ev[i].x < kCoordMin || ev[i].x > kCoordMax || ev[i].y < kCoordMin || ev[i].y > kCoordMax)
		{
			return c->fail(EBO_ERR_RANGE, "event coordinate outside [-16384,16383]");
		}
		xy[i] = (static_cast<uint32_t>(ev[i].x) & 0xFFFFu) | (static_cast<uint32_t>(ev[i].y) << 16);
	}
	if (n > 0)
	{
		int rc = c->hip(hipMemcpy(c->d_route_xy, xy.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice), "H2D route events");
		if (rc)
		{
			return rc;
		}
	}
	c->route_n = n;
	return EBO_OK;
}

int ebo_route_events(ebo_ctx* c, int n_patches, const double* rects, const uint32_t* start, const uint32_t* max_take,
					 uint32_t cap, uint32_t* out_index, uint32_t* out_count, uint32_t* out_next)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n_patches < 0 || (n_patches > 0 && (!rects || !start || !max_take || !out_count || !out_next)) ||
		(cap > 0 && n_patches > 0 && !out_index))
	{
		return c->fail(EBO_ERR_ARG, "null argument");
	}
	if (n_patches == 0)
	{
		return EBO_OK;
	}
	(void)hipSetDevice(c->prm.device);
	// one pinned block the kernel reads its arguments from and writes its results to (all small:
	// no copy packets, one launch + one sync): rects | start | take | count | next | index
	const size_t np = static_cast<size_t>(n_patches);
	auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
	const size_t bR = al(np * 32), bU = al(np * 4), bI = al(np * cap * 4 + 4);
	const size_t need = bR + 4 * bU + bI;
	if (need > c->pin_route_cap)
	{
		if (c->pin_route)
		{
			(void)hipHostFree(c->pin_route);
			c->pin_route = nullptr;
			c->pin_route_cap = 0;
		}
		int rc = c->hip(hipHostMalloc(&c->pin_route, need, kZeroCopyFlags), "hipHostMalloc route staging");
		if (rc)
		{
			return rc;
		}
		c->pin_route_cap = need;
	}
	char* pin = static_cast<char*>(c->pin_route);
	std::memcpy(pin, rects, np * 32);
	std::memcpy(pin + bR, start, np * 4);
	std::memcpy(pin + bR + bU, max_take, np * 4);
	RouteLaunch L;
	L.d_xy = c->d_route_xy;
	L.n_events = static_cast<uint32_t>(c->route_n);
	L.n_patches = n_patches;
	L.d_rects = reinterpret_cast<const double*>(pin);
	L.d_start = reinterpret_cast<const uint32_t*>(pin + bR);
	L.d_take = reinterpret_cast<const uint32_t*>(pin + bR + bU);
	L.d_count = reinterpret_cast<uint32_t*>(pin + bR + 2 * bU);
	L.d_next = reinterpret_cast<uint32_t*>(pin + bR + 3 * bU);
	L.d_index = reinterpret_cast<uint32_t*>(pin + bR + 4 * bU);
	L.cap = cap;
	if (launch_route(L, c->stream))
	{
		return c->hip(hipGetLastError(), "route launch");
	}
	int rc = c->hip(hipStreamSynchronize(c->stream), "sync");
	if (rc)
	{
		return rc;
	}
	std::memcpy(out_count, L.d_count, np * 4);
	std::memcpy(out_next, L.d_next, np * 4);
	for (size_t p = 0; p < np; ++p)
	{
		std::memcpy(out_index + p * cap, L.d_index + p * cap, static_cast<size_t>(out_count[p]) * 4);
	}
	return EBO_OK;
}

int ebo_patch_integrate(ebo_ctx* c, const ebo_event* ev, const size_t* offsets, int n_patches,
						const double* rects, const size_t* nabla_offsets, double* nabla,
						int64_t* current_ts, int64_t* time_last_update)
{
	return patch_integrate_common(c, ev, offsets, n_patches, rects, nullptr, nullptr,
								  nabla_offsets, nabla, current_ts, time_last_update, nullptr);
}

int ebo_patch_integrate_mc(ebo_ctx* c, const ebo_event* ev, const size_t* offsets, int n_patches,
						   const double* rects, const double* traj, const int64_t* mid_time,
						   const size_t* nabla_offsets, double* nabla, int32_t* updated)
{
	if (c && !traj)
	{
		return c->fail(EBO_ERR_ARG, "null trajectory");
	}
	return patch_integrate_common(c, ev, offsets, n_patches, rects, traj, mid_time, nabla_offsets,
								  nabla, nullptr, nullptr, updated);
}

// ---- per-feature tracker objective ---------------------------------------
void ebo_optimizer_default_solver(ebo_solver_opts* o)
{
	if (!o)
	{
		return;
	}
	ebo_default_solver(o);
	o->max_num_iterations = 10;  // OptimizerParams::maxNumIterations
	o->use_nonmonotonic = 1;     // optimizer.cpp:110
	o->function_tolerance = 1e-6;
	o->gradient_tolerance = 1e-10;
	o->parameter_tolerance = 1e-8;
	o->mode = EBO_SOLVE_INDEPENDENT;
}

int ebo_optimizer_set_grad(ebo_ctx* c, const double* grad_x, const double* grad_y)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!grad_x || !grad_y)
	{
		return c->fail(EBO_ERR_ARG, "null gradient image");
	}
	(void)hipSetDevice(c->prm.device);
	const size_t n = static_cast<size_t>(c->prm.image_w) * c->prm.image_h;
	int rc = EBO_OK;
	if (!c->d_opt_grid)
	{
		rc = c->hip(hipMalloc(reinterpret_cast<void**>(&c->d_opt_grid), n * sizeof(double2)), "hipMalloc gradient grid");
		if (rc)
		{
			return rc;
		}
	}
	rc = ensure_aux(c, 2 * n * sizeof(double));
	if (rc)
	{
		return rc;
	}
	double* stage = static_cast<double*>(c->d_aux);
	hipError_t e = hipMemcpyAsync(stage, grad_x, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess)
	{
		e = hipMemcpyAsync(stage + n, grad_y, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
	}
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D gradient images");
	}
	if (launch_optimizer_interleave(stage, stage + n, n, c->d_opt_grid, c->stream))
	{
		return c->hip(hipGetLastError(), "gradient grid launch");
	}
	rc = c->hip(hipStreamSynchronize(c->stream), "sync");
	c->opt_grid_valid = rc == EBO_OK;
	return rc;
}

// FeatureDetector::updateNumOfEvents' estimate (feature_detector.cpp:689-707) for n patches, one launch.
int ebo_estimate_num_events(ebo_ctx* c, int n, const double* rects, const double* poses, const double* flow_dirs,
							uint64_t* out)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n < 0 || (n && (!rects || !poses || !flow_dirs || !out)))
	{
		return c->fail(EBO_ERR_ARG, "null argument");
	}
	if (!c->opt_grid_valid)
	{
		return c->fail(EBO_ERR_STATE, "no gradient grid: call ebo_optimizer_set_grad first");
	}
	if (n == 0)
	{
		return EBO_OK;
	}
	(void)hipSetDevice(c->prm.device);
	const size_t nn = static_cast<size_t>(n);
	int rc = ensure_scratch(c, nn * 10 * sizeof(double));
	if (rc)
	{
		return rc;
	}
	double* d = static_cast<double*>(c->d_scratch);
	double* dRects = d;
	double* dPoses = d + 4 * nn;
	double* dFlows = d + 8 * nn;
	double* dSums = d + 9 * nn;
	hipError_t e = hipMemcpyAsync(dRects, rects, 4 * nn * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dPoses, poses, 4 * nn * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dFlows, flow_dirs, nn * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D estimate arguments");
	}
	if (launch_estimate_num_events(c->d_opt_grid, c->prm.image_w, c->prm.image_h, n, dRects, dPoses, dFlows, dSums, c->stream))
	{
		return c->hip(hipGetLastError(), "estimate launch");
	}
	std::vector<double> sums(nn);
	e = hipMemcpyAsync(sums.data(), dSums, nn * sizeof(double), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess)
	{
		return c->hip(e, "D2H estimates");
	}
	for (size_t i = 0; i < nn; ++i)
	{
		out[i] = static_cast<uint64_t>(sums[i]);  // size_t sumPatch = cv::norm(...)
	}
	return EBO_OK;
}

int ebo_patch_warp_image(ebo_ctx* c, int n, const double* rects, const double* poses, const double* flow_dirs,
						 const size_t* nabla_offsets, double* predicted, int32_t* updated)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n < 0 || (n && (!rects || !poses || !flow_dirs || !nabla_offsets || !predicted || !updated)))
	{
		return c->fail(EBO_ERR_ARG, "null argument");
	}
	if (!c->opt_grid_valid)
	{
		return c->fail(EBO_ERR_STATE, "no gradient grid: call ebo_optimizer_set_grad first");
	}
	if (n == 0)
	{
		return EBO_OK;
	}
	(void)hipSetDevice(c->prm.device);
	const size_t nn = static_cast<size_t>(n);
	// patch.cpp:145-150: a rect that touches the image border leaves predictedNabla_ as it is
	std::vector<int> skip(nn, 0);
	std::vector<size_t> off(nn, 0), len(nn, 0);
	size_t total = 0;
	for (size_t i = 0; i < nn; ++i)
	{
		const double* r = rects + 4 * i;
		const bool border = r[0] < 0 || r[1] < 0 || r[0] + r[2] >= c->prm.image_w || r[1] + r[3] >= c->prm.image_h;
		const double w = std::nearbyint(r[2]), h = std::nearbyint(r[3]);
		if (!(w >= 0 && h >= 0 && w <= 32767 && h <= 32767))  // also NaN
		{
			return c->fail(EBO_ERR_RANGE, "patch rect size out of range");
		}
		skip[i] = border ? 1 : 0;
		updated[i] = border ? 0 : 1;
		off[i] = total;
		len[i] = border ? 0 : static_cast<size_t>(w) * static_cast<size_t>(h);
		total += len[i];
	}
	if (total == 0)
	{
		return EBO_OK;
	}
	const size_t head = nn * 9 * sizeof(double) + nn * sizeof(int) + nn * sizeof(size_t);
	const size_t headAligned = (head + 15) & ~static_cast<size_t>(15);
	int rc = ensure_scratch(c, headAligned + total * sizeof(double));
	if (rc)
	{
		return rc;
	}
	char* base = static_cast<char*>(c->d_scratch);
	double* dRects = reinterpret_cast<double*>(base);
	double* dPoses = dRects + 4 * nn;
	double* dFlows = dPoses + 4 * nn;
	size_t* dOff = reinterpret_cast<size_t*>(dFlows + nn);
	int* dSkip = reinterpret_cast<int*>(dOff + nn);
	double* dOut = reinterpret_cast<double*>(base + headAligned);
	hipError_t e = hipMemcpyAsync(dRects, rects, 4 * nn * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dPoses, poses, 4 * nn * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dFlows, flow_dirs, nn * sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dOff, off.data(), nn * sizeof(size_t), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(dSkip, skip.data(), nn * sizeof(int), hipMemcpyHostToDevice, c->stream);
	if (e != hipSuccess)
	{
		(void)hipStreamSynchronize(c->stream);
		return c->hip(e, "H2D warpImage arguments");
	}
	if (launch_patch_warp_image(c->d_opt_grid, c->prm.image_w, c->prm.image_h, n, dRects, dPoses, dFlows, dSkip, dOff, dOut,
								c->stream))
	{
		(void)hipStreamSynchronize(c->stream);
		return c->hip(hipGetLastError(), "warpImage launch");
	}
	for (size_t i = 0; i < nn && e == hipSuccess; ++i)
	{
		if (len[i])
		{
			e = hipMemcpyAsync(predicted + nabla_offsets[i], dOut + off[i], len[i] * sizeof(double), hipMemcpyDeviceToHost,
							   c->stream);
		}
	}
	const hipError_t es = hipStreamSynchronize(c->stream);  // also covers the host vectors of the H2D copies
	return c->hip(e != hipSuccess ? e : es, "D2H predicted nabla");
}

namespace
{
struct OptBuffers
{
	OptPatch* patches;
	double* nabla_in;
	double* nabla;
	double* x;
	double* stats;
	double* res;
	double* jac_pose;
	double* jac_flow;
	size_t total;
	int max_pixels;
};

// Validates the rects, uploads patches / nabla / parameters; normalize: nabla is the raw
// integrated nabla and is normalised on the device (Patch::getNormalizedIntegratedNabla).
int optimizer_stage(ebo_ctx* c, int n, const double* rects, const double* nabla, int normalize,
					const double* poses, const double* flow_dirs, bool wantRes, bool wantJac, OptBuffers& B)
{
	if (!c->opt_grid_valid)
	{
		return c->fail(EBO_ERR_STATE, "ebo_optimizer_set_grad has not been called");
	}
	if (n < 0 || (n > 0 && (!rects || !nabla || !poses || !flow_dirs)))
	{
		return c->fail(EBO_ERR_ARG, "bad argument to the optimizer");
	}
	std::vector<OptPatch> hp(n);
	size_t total = 0;
	int maxPx = 1;
	for (int i = 0; i < n; ++i)
	{
		const double w = rects[4 * i + 2], h = rects[4 * i + 3];
		if (!(w >= 1.0) || !(h >= 1.0) || w > 4096.0 || h > 4096.0 || !std::isfinite(rects[4 * i]) ||
			!std::isfinite(rects[4 * i + 1]))
		{
			return c->fail(EBO_ERR_ARG, "bad patch rect");
		}
		hp[i].rx = rects[4 * i];
		hp[i].ry = rects[4 * i + 1];
		hp[i].pw = static_cast<int>(w);
		hp[i].ph = static_cast<int>(h);
		hp[i].off = total;
		const int px = hp[i].pw * hp[i].ph;
		if (px > 3000)
		{
			return c->fail(EBO_ERR_UNSUPPORTED, "tracked patch larger than 3000 pixels");
		}
		maxPx = std::max(maxPx, px);
		total += static_cast<size_t>(px);
	}
	auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
	const size_t bPatch = al(static_cast<size_t>(n) * sizeof(OptPatch)), bVec = al(total * 8);
	const size_t bX = al(static_cast<size_t>(n) * 5 * 8), bStats = al(static_cast<size_t>(n) * 8 * 8);
	const size_t need = bPatch + 2 * bVec + bX + bStats + (wantRes ? bVec : 0) + (wantJac ? 5 * bVec : 0) + 256;
	(void)hipSetDevice(c->prm.device);
	if (need > c->opt_cap)
	{
		if (c->d_opt)
		{
			hipFree(c->d_opt);
			c->d_opt = nullptr;
			c->opt_cap = 0;
		}
		int rc = c->hip(hipMalloc(&c->d_opt, need), "hipMalloc optimizer scratch");
		if (rc)
		{
			return rc;
		}
		c->opt_cap = need;
	}
	char* b = static_cast<char*>(c->d_opt);
	B.patches = reinterpret_cast<OptPatch*>(b);
	b += bPatch;
	B.nabla_in = reinterpret_cast<double*>(b);
	b += bVec;
	B.nabla = reinterpret_cast<double*>(b);
	b += bVec;
	B.x = reinterpret_cast<double*>(b);
	b += bX;
	B.stats = reinterpret_cast<double*>(b);
	b += bStats;
	B.res = wantRes ? reinterpret_cast<double*>(b) : nullptr;
	b += wantRes ? bVec : 0;
	B.jac_pose = wantJac ? reinterpret_cast<double*>(b) : nullptr;
	b += wantJac ? 4 * bVec : 0;
	B.jac_flow = wantJac ? reinterpret_cast<double*>(b) : nullptr;
	B.total = total;
	B.max_pixels = maxPx;
	if (n == 0)
	{
		return EBO_OK;
	}
	std::vector<double> hx(static_cast<size_t>(n) * 5);
	for (int i = 0; i < n; ++i)
	{
		for (int k = 0; k < 4; ++k)
		{
			hx[5 * i + k] = poses[4 * i + k];
		}
		hx[5 * i + 4] = flow_dirs[i];
	}
	hipError_t e = hipMemcpyAsync(B.patches, hp.data(), static_cast<size_t>(n) * sizeof(OptPatch), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess)
	{
		e = hipMemcpyAsync(normalize ? B.nabla_in : B.nabla, nabla, total * 8, hipMemcpyHostToDevice, c->stream);
	}
	if (e == hipSuccess)
	{
		e = hipMemcpyAsync(B.x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice, c->stream);
	}
	if (e == hipSuccess)
	{
		e = hipStreamSynchronize(c->stream);  // hp / hx are locals
	}
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D optimizer inputs");
	}
	if (normalize && launch_optimizer_normalize(B.patches, n, B.nabla_in, B.nabla, c->stream))
	{
		return c->hip(hipGetLastError(), "nabla normalisation launch");
	}
	return EBO_OK;
}

OptLaunch optimizer_launch(const ebo_ctx* c, int n, const OptBuffers& B)
{
	OptLaunch L;
	std::memset(&L, 0, sizeof(L));
	L.d_grid = c->d_opt_grid;
	L.img_w = c->prm.image_w;
	L.img_h = c->prm.image_h;
	L.d_patches = B.patches;
	L.n_patches = n;
	L.max_pixels = B.max_pixels;
	L.d_nabla = B.nabla;
	L.d_x = B.x;
	L.d_res = B.res;
	L.d_jac_pose = B.jac_pose;
	L.d_jac_flow = B.jac_flow;
	L.d_stats = B.stats;
	return L;
}
}  // namespace

int ebo_optimizer_eval(ebo_ctx* c, int n, const double* rects, const double* nabla, const double* poses,
					   const double* flow_dirs, double* residuals, double* jac_pose, double* jac_flow)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!residuals || ((jac_pose == nullptr) != (jac_flow == nullptr)))
	{
		return c->fail(EBO_ERR_ARG, "residuals are required; the two Jacobians come together");
	}
	OptBuffers B;
	int rc = optimizer_stage(c, n, rects, nabla, 0, poses, flow_dirs, true, jac_pose != nullptr, B);
	if (rc || n == 0)
	{
		return rc;
	}
	OptLaunch L = optimizer_launch(c, n, B);
	if (launch_optimizer_eval(L, c->stream))
	{
		return c->hip(hipGetLastError(), "optimizer evaluation launch");
	}
	hipError_t e = hipMemcpyAsync(residuals, B.res, B.total * 8, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess && jac_pose)
	{
		e = hipMemcpyAsync(jac_pose, B.jac_pose, B.total * 4 * 8, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess && jac_flow)
	{
		e = hipMemcpyAsync(jac_flow, B.jac_flow, B.total * 8, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess)
	{
		e = hipStreamSynchronize(c->stream);
	}
	return c->hip(e, "D2H optimizer results");
}

int ebo_optimizer_cost_map(ebo_ctx* c, int n, const double* rects, const double* nabla, int normalize, const double* poses,
						   const double* flow_dirs, int map_w, int map_h, double* cost_maps)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (map_w < 1 || map_h < 1 || map_w > 255 || map_h > 255 || (n > 0 && !cost_maps))
	{
		return c->fail(EBO_ERR_ARG, "cost map: 1 <= width, height <= 255 and an output buffer");
	}
	OptBuffers B;
	int rc = optimizer_stage(c, n, rects, nabla, normalize, poses, flow_dirs, false, false, B);
	if (rc || n == 0)
	{
		return rc;
	}
	// poseNew of optimizer.cpp:46-51, on the host as the reference forms it: SE2(pose.log().z(), (float(x) + tx, float(y) + ty))
	// -- the angle through atan2 and back through cos / sin -- for x (outer) and y (inner); cell (y + hy, x + hx) of the map
	const int cells = map_w * map_h, hx = (map_w - 1) / 2, hy = (map_h - 1) / 2;
	std::vector<double> hxc(static_cast<size_t>(n) * cells * 5, 0.0);
	std::vector<uint8_t> filled(static_cast<size_t>(cells), 0);
	for (int i = 0; i < n; ++i)
	{
		const double* ps = poses + 4 * static_cast<size_t>(i);
		const double theta = std::atan2(ps[1], ps[0]);
		const double ct = std::cos(theta), st = std::sin(theta);
		for (int x = -hx; x <= hx; ++x)
		{
			for (int y = -hy; y <= hy; ++y)
			{
				double* o = &hxc[(static_cast<size_t>(i) * cells + static_cast<size_t>(y + hy) * map_w + (x + hx)) * 5];
				o[0] = ct;
				o[1] = st;
				o[2] = static_cast<double>(static_cast<float>(x)) + ps[2];
				o[3] = static_cast<double>(static_cast<float>(y)) + ps[3];
				o[4] = flow_dirs[i];
				filled[static_cast<size_t>(y + hy) * map_w + (x + hx)] = 1;
			}
		}
	}
	const size_t bx = hxc.size() * 8, bo = static_cast<size_t>(n) * cells * 8;
	rc = ensure_scratch(c, bx + bo + 256);
	if (rc)
	{
		return rc;
	}
	double* dX = static_cast<double*>(c->d_scratch);
	double* dOut = reinterpret_cast<double*>(static_cast<char*>(c->d_scratch) + ((bx + 255) & ~static_cast<size_t>(255)));
	hipError_t e = hipMemcpyAsync(dX, hxc.data(), bx, hipMemcpyHostToDevice, c->stream);
	if (e != hipSuccess)
	{
		(void)hipStreamSynchronize(c->stream);
		return c->hip(e, "H2D cost-map poses");
	}
	OptLaunch L = optimizer_launch(c, n, B);
	if (launch_optimizer_cost_map(L, dX, cells, dOut, c->stream))
	{
		(void)hipStreamSynchronize(c->stream);
		return c->hip(hipGetLastError(), "cost-map launch");
	}
	e = hipMemcpyAsync(cost_maps, dOut, bo, hipMemcpyDeviceToHost, c->stream);
	const hipError_t es = hipStreamSynchronize(c->stream);
	rc = c->hip(e != hipSuccess ? e : es, "D2H cost maps");
	if (rc)
	{
		return rc;
	}
	// an even width / height leaves the last column / row of cv::Mat::zeros untouched (:43-45 visit 2 * half + 1 cells)
	for (int i = 0; i < n; ++i)
	{
		for (int q = 0; q < cells; ++q)
		{
			if (!filled[q])
			{
				cost_maps[static_cast<size_t>(i) * cells + q] = 0.0;
			}
		}
	}
	return EBO_OK;
}

int ebo_optimizer_solve(ebo_ctx* c, int n, const double* rects, const double* nabla, int normalize, double huber,
						const ebo_solver_opts* opts, double* poses, double* flow_dirs, ebo_summary* summaries)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	ebo_solver_opts o;
	if (opts)
	{
		o = *opts;
		o.mode = EBO_SOLVE_INDEPENDENT;
		int rc = check_solver_opts(c, &o);
		if (rc)
		{
			return rc;
		}
	}
	else
	{
		ebo_optimizer_default_solver(&o);
	}
	if (!(huber > 0.0))
	{
		return c->fail(EBO_ERR_ARG, "the Huber parameter must be positive");
	}
	OptBuffers B;
	int rc = optimizer_stage(c, n, rects, nabla, normalize, poses, flow_dirs, false, false, B);
	if (rc || n == 0)
	{
		return rc;
	}
	OptLaunch L = optimizer_launch(c, n, B);
	L.huber = huber;
	L.s = make_solve_consts(&o);
	if (launch_optimizer_solve(L, c->stream))
	{
		return c->hip(hipGetLastError(), "optimizer solve launch");
	}
	std::vector<double> hx(static_cast<size_t>(n) * 5), hs(static_cast<size_t>(n) * 8);
	hipError_t e = hipMemcpyAsync(hx.data(), B.x, hx.size() * 8, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess)
	{
		e = hipMemcpyAsync(hs.data(), B.stats, hs.size() * 8, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess)
	{
		e = hipStreamSynchronize(c->stream);
	}
	if (e != hipSuccess)
	{
		return c->hip(e, "D2H optimizer results");
	}
	for (int i = 0; i < n; ++i)
	{
		for (int k = 0; k < 4; ++k)
		{
			poses[4 * i + k] = hx[5 * i + k];
		}
		flow_dirs[i] = hx[5 * i + 4];
		if (summaries)
		{
			summaries[i].iterations = static_cast<int32_t>(hs[8 * i + 0]);
			summaries[i].num_evals_cost = static_cast<int32_t>(hs[8 * i + 1]);
			summaries[i].num_evals_jac = static_cast<int32_t>(hs[8 * i + 2]);
			summaries[i].termination = static_cast<int32_t>(hs[8 * i + 3]);
			summaries[i].initial_cost = hs[8 * i + 4];
			summaries[i].final_cost = hs[8 * i + 5];
		}
	}
	return EBO_OK;
}

}  // extern "C"
