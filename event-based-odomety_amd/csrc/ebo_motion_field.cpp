// ebo_motion_field.cpp — initMotionField / interpolateMotionField entry points of include/ebo.h.
#include "ebo_ctx.h"
#include "field_tv.h"

using namespace ebo;

extern "C" {

// FeatureDetector::initMotionField (feature_detector.cpp:53-142).
int ebo_init_motion_field(ebo_ctx* c, int64_t timestamp, int use_average, int n_patches,
						  const size_t* traj_offsets, const double* traj_xy, const int64_t* traj_t,
						  float* field_out, int32_t* n_fixed, int32_t* fixed_xy)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n_patches < 0 || (n_patches > 0 && (!traj_offsets || !traj_xy || !traj_t)))
	{
		return c->fail(EBO_ERR_ARG, "null trajectory arrays");
	}
	(void)hipSetDevice(c->prm.device);
	const int w = c->prm.image_w, h = c->prm.image_h;
	const size_t npx = static_cast<size_t>(w) * h;
	const size_t nSamples = n_patches > 0 ? traj_offsets[n_patches] : 0;
	auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
	const size_t bField = al(npx * 2 * sizeof(float));
	const size_t bOff = al((static_cast<size_t>(n_patches) + 1) * 8), bXY = al(nSamples * 16), bT = al(nSamples * 8);
	const size_t bFix = al(static_cast<size_t>(std::max(n_patches, 1)) * 8);
	const size_t need = bField + bOff + bXY + bT + bFix + 512;
	if (need > c->field_cap)
	{
		if (c->d_field)
		{
			hipFree(c->d_field);
			c->d_field = nullptr;
			c->field_cap = 0;
		}
		int rc = c->hip(hipMalloc(&c->d_field, need), "hipMalloc motion field");
		if (rc)
		{
			return rc;
		}
		c->field_cap = need;
	}
	char* base = static_cast<char*>(c->d_field);
	FieldLaunch L;
	L.w = w;
	L.h = h;
	L.scale = c->prm.scale;
	L.use_average = use_average ? 1 : 0;
	L.n_patches = n_patches;
	L.d_field = reinterpret_cast<float*>(base);
	L.d_off = reinterpret_cast<unsigned long long*>(base + bField);
	L.d_xy = reinterpret_cast<double*>(base + bField + bOff);
	L.d_t = reinterpret_cast<long long*>(base + bField + bOff + bXY);
	L.d_fixed = reinterpret_cast<int*>(base + bField + bOff + bXY + bT);
	L.d_avg = reinterpret_cast<double*>(base + bField + bOff + bXY + bT + bFix);
	L.d_nfixed = reinterpret_cast<int*>(base + bField + bOff + bXY + bT + bFix + 256);
	L.timestamp = timestamp;
	std::vector<unsigned long long> off64(static_cast<size_t>(n_patches) + 1, 0ull);
	for (int k = 0; k <= n_patches && n_patches > 0; ++k)
	{
		off64[k] = traj_offsets[k];
	}
	hipError_t e = hipMemcpyAsync(const_cast<unsigned long long*>(L.d_off), off64.data(), off64.size() * 8,
								  hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && nSamples)
	{
		e = hipMemcpyAsync(const_cast<double*>(L.d_xy), traj_xy, nSamples * 16, hipMemcpyHostToDevice, c->stream);
	}
	if (e == hipSuccess && nSamples)
	{
		e = hipMemcpyAsync(const_cast<long long*>(L.d_t), traj_t, nSamples * 8, hipMemcpyHostToDevice, c->stream);
	}
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D trajectories");
	}
	if (launch_init_field(L, c->stream))
	{
		return c->hip(hipGetLastError(), "motion field launch");
	}
	int nf = 0;
	if (field_out)
	{
		e = hipMemcpyAsync(field_out, L.d_field, npx * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess)
	{
		e = hipMemcpyAsync(&nf, L.d_nfixed, sizeof(int), hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess)
	{
		e = hipStreamSynchronize(c->stream);
	}
	if (e == hipSuccess && fixed_xy && nf > 0)
	{
		e = hipMemcpy(fixed_xy, L.d_fixed, static_cast<size_t>(nf) * 8, hipMemcpyDeviceToHost);
	}
	if (e != hipSuccess)
	{
		return c->hip(e, "D2H motion field");
	}
	if (n_fixed)
	{
		*n_fixed = nf;
	}
	c->field_valid = true;
	c->d_field_fixed = L.d_fixed;
	c->field_nfixed = nf;
	return EBO_OK;
}

int ebo_interpolate_motion_field(ebo_ctx* c, int use_l1, const ebo_solver_opts* opts, float* field_out,
								 ebo_summary* summary, int32_t* cg_iterations)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!c->field_valid)
	{
		return c->fail(EBO_ERR_STATE, "ebo_interpolate_motion_field needs ebo_init_motion_field first");
	}
	(void)hipSetDevice(c->prm.device);
	const int w = c->prm.image_w, h = c->prm.image_h;
	if (w < 2 || h < 2)
	{
		return c->fail(EBO_ERR_ARG, "image too small for the TV problem");
	}
	// a fixed point at pixel (w-1, h-1) is no parameter block of the reference's problem
	// (feature_detector.cpp:170-204): Ceres aborts in IsParameterBlockConstant (:208)
	std::vector<int32_t> fixed(static_cast<size_t>(c->field_nfixed) * 2);
	if (c->field_nfixed > 0)
	{
		int rc = c->hip(hipMemcpyAsync(fixed.data(), c->d_field_fixed, fixed.size() * sizeof(int32_t),
									   hipMemcpyDeviceToHost, c->stream), "D2H fixed points");
		if (rc == EBO_OK)
		{
			rc = c->hip(hipStreamSynchronize(c->stream), "sync");
		}
		if (rc)
		{
			return rc;
		}
		for (int i = 0; i < c->field_nfixed; ++i)
		{
			if (fixed[2 * i] == w - 1 && fixed[2 * i + 1] == h - 1)
			{
				return c->fail(EBO_ERR_RANGE, "fixed point at the last pixel: not a parameter of the TV problem");
			}
		}
	}
	const size_t need = tvf_workspace_bytes(w, h);
	if (need > c->tvf_cap)
	{
		if (c->d_tvf)
		{
			hipFree(c->d_tvf);
			c->d_tvf = nullptr;
			c->tvf_cap = 0;
		}
		int rc = c->hip(hipMalloc(&c->d_tvf, need), "hipMalloc TV workspace");
		if (rc)
		{
			return rc;
		}
		c->tvf_cap = need;
	}
	ebo_solver_opts o;
	if (opts)
	{
		o = *opts;
	}
	else
	{
		// ceres::Solver::Options defaults with feature_detector.cpp:216-222 applied
		ebo_default_solver(&o);
		o.use_nonmonotonic = 0;
		o.function_tolerance = 1e-6;
		o.gradient_tolerance = 1e-10;
		o.parameter_tolerance = 1e-8;
	}
	FieldTvStats st;
	int rc = field_tv_solve(w, h, static_cast<float*>(c->d_field), c->d_field_fixed, c->field_nfixed,
							use_l1 != 0, o, c->d_tvf, c->stream, &st, &c->err);
	if (rc)
	{
		return rc;
	}
	if (summary)
	{
		summary->iterations = st.iterations;
		summary->num_evals_cost = st.evals_cost;
		summary->num_evals_jac = st.evals_jac;
		summary->termination = st.termination;
		summary->initial_cost = st.initial_cost;
		summary->final_cost = st.final_cost;
	}
	if (cg_iterations)
	{
		*cg_iterations = st.cg_iterations;
	}
	if (field_out)
	{
		rc = c->hip(hipMemcpy(field_out, c->d_field, static_cast<size_t>(w) * h * 2 * sizeof(float),
							  hipMemcpyDeviceToHost), "D2H motion field");
		if (rc)
		{
			return rc;
		}
	}
	return st.termination == 2 ? c->fail(EBO_ERR_SOLVER, "field TV solve failed") : EBO_OK;
}

}  // extern "C"
