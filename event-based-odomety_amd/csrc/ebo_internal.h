// ebo_internal.h — types shared by the host side (ebo_api.cpp) and the device
// side (ebo_kernels.hip) of libebo_hip.so.  Not part of the ABI.
#pragma once

#include "ab_env.h"

#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace ebo
{
// One event in HBM: 8 bytes (the algorithmic bytes per
// event-evaluation).
//   lo bits  0..14  x   (15-bit two's complement)
//      bit   15     polarity (1 = POSITIVE)
//      bits 16..30  y   (15-bit two's complement)
//      bit   31     unused
//   hi             dt = t_ref(unit) - t_event   in microseconds (int32)
// Events are stored bucketed by unit (window, patch), time order kept inside a
// unit, so that one workgroup streams one contiguous, coalesced range.
static const int kCoordMin = -16384;
static const int kCoordMax = 16383;

inline uint32_t pack_lo(int x, int y, int positive)
{
	return (static_cast<uint32_t>(x) & 0x7FFFu) |
		   (static_cast<uint32_t>(positive ? 1 : 0) << 15) |
		   ((static_cast<uint32_t>(y) & 0x7FFFu) << 16);
}

// A unit = one patch of one window (a contrastFunctor instance), or the stray
// bucket of a window (events outside the sensor; flag bit 1).
struct Unit
{
	uint32_t ev_off;   // first packed event
	uint32_t n_ev;     // number of events
	int16_t rx, ry;    // patch rect (cv::Rect2i)
	int16_t rw, rh;
	int32_t dt_win;    // t_ref(window) - t_ref(unit): dt_window = dt + dt_win
	uint32_t flags;    // bit0: active (n_ev > min_events); bit1: stray bucket
	uint32_t flow_idx; // index of this unit's flow in the [Wn][P] arrays
};
static_assert(sizeof(Unit) == 28, "Unit layout");

// Canonical order inside a unit (units of 2 .. 8192 events; larger ones stay in list order): the records ranked
// ascending as u64 -- by (uint32) dt = t_ref - t, then by the coordinate word -- and then DEALT OUT with a fixed
// stride: the record of rank (q * stride) mod n sits at position q.  Rank order is time order, and on a moving edge
// events that follow each other in time are neighbours in space: 64 consecutive records then splat onto overlapping
// 7 x 7 windows and their LDS atomics hit the same words (k_eval3: 64 % of LDS cycles were bank conflicts).  With the
// stride, consecutive positions are far apart in time, i.e. scattered over the patch: k_eval3 C3 x 64 windows
// 0.543 -> 0.500 ms, C2 0.160 -> 0.150, C4 0.352 -> 0.330 (profiles/r04_event_order.txt; strides 101 .. 151 are
// within 1 % of each other, row-sorted orders are 16-27 % SLOWER).  The order depends on the unit's events only
// (never on the batch), is unique, and is the same in every loading path (host sort, device bucketing, patches).
constexpr uint32_t kOrderStride = 127;
inline __host__ __device__ uint32_t order_stride(uint32_t n)  // coprime to n, 1 <= stride < max(n, 2)
{
	if (n < 3)
	{
		return 1;
	}
	uint32_t st = kOrderStride % n;
	st = st == 0 ? 1 : st;
	for (;; ++st)
	{
		uint32_t a = st, b = n;
		while (b)
		{
			const uint32_t t = a % b;
			a = b;
			b = t;
		}
		if (a == 1)
		{
			return st;
		}
	}
}
// position of the record of canonical rank r: q with (q * stride) mod n == r, i.e. q = r * stride^-1 mod n
inline __host__ __device__ uint32_t order_inverse(uint32_t st, uint32_t n)
{
	// extended Euclid on (st, n); n <= 8192
	int t0 = 0, t1 = 1;
	int r0 = static_cast<int>(n), r1 = static_cast<int>(st);
	while (r1 != 0)
	{
		const int q = r0 / r1;
		const int r2 = r0 - q * r1, t2 = t0 - q * t1;
		r0 = r1;
		r1 = r2;
		t0 = t1;
		t1 = t2;
	}
	return static_cast<uint32_t>(t0 < 0 ? t0 + static_cast<int>(n) : t0) % (n ? n : 1);
}

static const uint32_t kUnitActive = 1u;
static const uint32_t kUnitStray = 2u;

// Constants of one evaluation, passed by value to kernels.
struct EvalConsts
{
	double scale;         // compensateScale
	double max_res;       // maxPossibleResidual_
	double norm;          // 1 / (2 pi sigma^2)
	double hs;            // -0.5 / sigma^2
	double inv_sigsq;     // 1 / sigma^2
	double ck1, ck2, ck3; // exp(hs * k^2), k = 1..3 (factored 1-D Gaussian taps)
	double fix_bias;      // 1.5 * 2^k: adding it aligns a tap value to the fixed-point grid
	double fix_scale;     // 2^(k-52): value of one fixed-point unit
	int32_t image_w, image_h;
	int32_t patch_w, patch_h;
	int32_t npx, npy;
	uint32_t inv_pw, inv_ph; // floor(2^32 / patch_w) + 1 (resp. patch_h): x / patch_w == umulhi(x, inv_pw) for 0 <= x < 2^15, patch_w >= 2
};

// ceres-style solver options for the device solver (mirror of ebo_solver_opts).
struct SolveConsts
{
	int32_t max_num_iterations;
	int32_t max_nonmono;  // 0 when use_nonmonotonic is off
	int32_t max_invalid;
	int32_t jacobi_scaling;
	double function_tolerance, gradient_tolerance, parameter_tolerance;
	double initial_radius, max_radius, min_radius;
	double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
};

// The windows of a batch that a launch covers, for lock-step solves whose batch has thinned out (late rounds
// of EBO_SOLVE_GLOBAL: a few stragglers of 256 windows): workgroups are created for these windows' units
// only, and the mode of a window (1 value, 2 value + Jacobian) rides in its entry.  Passed BY VALUE as a
// kernel argument -- no copy, no extra command on the stream; n = 0: the launch covers every unit.
constexpr int kLiveMax = 64;
struct LiveWindows
{
	int n = 0;
	int upw = 1;          // units per window (P + 1: the stray unit)
	int ent[kLiveMax];    // window << 2 | mode
};

// Per-block partial sums of the variance objective: S1, S2, n, D1[2], D2[2], pad.
static const int kPartialStride = 8;

// ---- launch wrappers implemented in ebo_kernels.hip (hipStream_t as void*) ----
struct EvalLaunch
{
	const uint64_t* d_events;
	const Unit* d_units;
	int n_units;           // patch units only (strays excluded)
	const double* d_flows; // [flow sets][n_flow][2]
	int n_flow;            // flows per flow set (= Wn * P)
	int flow_sets;         // 1, or 5 for central differences
	int channels;          // 1 or 3
	int tiles;             // row tiles per unit
	int block;             // threads per workgroup
	int impl;              // 0: 3-channel f64 atomics on the full canvas (first version)
	                       // 1: scatter value / gather derivatives, f64 atomics
	                       // 2: same with exact fixed-point (u64) accumulation
	int cap_doubles;       // impl 1/2: image capacity of one workgroup's LDS, in pixels
	int rotate;            // impl 1/2: per-lane tap rotation in the scatter pass
	int deal;              // impl 3: re-deal a wave's events over its lanes by LDS bank residue (EBO_EVAL_DEAL)
	size_t lds_bytes;
	double* d_partials;    // [flow sets][n_units][tiles][kPartialStride]
	double* d_out;         // [n_flow][3]
	double fd_step;        // > 0: combine as central differences
	const unsigned char* d_modes = nullptr;  // per flow slot: 0 skip, 1 value, 2 value + Jacobian (impl 3, fused path)
	LiveWindows live;      // n > 0: only these windows (impl 3, fused path)
	EvalConsts c;
};
int launch_eval_variance(const EvalLaunch& L, void* stream);
int launch_dump_image(const EvalLaunch& L, int unit, double* d_image, void* stream);

// Edge (structure-tensor) loss, contrast_functor.h:152-277.
struct EdgeConsts
{
	const double* w;       // DEVICE table [49]: tensor weights [i+3][j+3] = gaussian(0,0,j,i,sigmaST) (:193-202); read by scalar loads where used (49 kernel-argument doubles pinned 98 SGPRs and the reverse pass spilled hundreds)
	double w_max;          // the largest of them, w[24]
	double g[7];           // 1-D factors exp(hs_st k^2), k = -3..3:  w(i,j) = norm_st g[i] g[j]
	double norm_st;        // 1 / (2 pi sigmaST^2)
	double mean_threshold; // 1e-4 (:159)
	int ablate;            // timing-only ablation mask (EBO_EDGE_ABLATE): 1 eigen, 2 NMS, 4 reverse, 8 gather, 16 scatter, 32 no register runs, 64 reference-order image, 128 bank-spread fake entries in the reverse sweep
	double* cs;            // DEVICE table [workgroup slot][cap_px][2] or null: (s00 - s11)/d and 2 s01/d of every pixel with a positive eigenvalue, written by the eigenvalue pass of a Jacobian evaluation, read by its reverse pass for the argmax pixels (null: the reverse pass re-derives the tensor sums)
	int cs_stride;         // pixels per slot of cs = the most pixels an LDS-resident box may have (aliased layouts)
	int reserved;          // tensor filter forms (EBO_EDGE_SEPARABLE): 1 band buffers on the 28 B layout, 2 register runs on the 20 B layout, 4 register runs on the 28 B layout
	unsigned long long* stats;  // null, or DEVICE counters [6] of ebo_edge_work_stats: units past the penalty test, their events, box pixels, eigenvalue-region pixels, NMS windows, argmax entries
};

// The compact LDS layout of k_eval_edge_wg (round 5): 16.5 B per pixel (I, E / A, 4-bit claim counters) and a header
// sized by the launch, so that THREE 256-lane workgroups share a CU; a unit whose box does not fit is not evaluated
// on its global slice but appended to defer_list for the launch that follows with the 20 B layout (two per CU).
// list_cap == 0: the layouts of rounds 1-4.
struct EdgeCompact
{
	int hdr_doubles = 0;       // doubles in front of the arrays: red[128] | 80 ints | list_cap ints
	int list_cap = 0;          // ints of the list / cell region
	int* defer_list = nullptr; // DEVICE [items]
	int* defer_count = nullptr;// DEVICE, zeroed before the launch
	void* bbox = nullptr;      // DEVICE int4[items]: the tap bounding boxes of this launch's units (k_edge_classify), or null
};

struct EdgeLaunch
{
	const uint64_t* d_events;
	const Unit* d_units;
	int n_units;
	const double* d_flows;
	int want_jac;
	int flow_sets;        // 1, or 5 (central differences, value only)
	double fd_step;
	int block;
	int cap_px;           // pixels per array that fit LDS
	int alias_lds;        // 1: 20 B/pixel LDS layout (A in E's storage, direct tensor form), 2 workgroups per CU
	size_t lds_bytes;
	EdgeCompact compact;  // list_cap > 0: this launch uses the compact layout (edge_launch_setup)
	size_t compact_lds_bytes = 0;
	int compact_cap_px = 0;
	int compact_table_px = 0;  // pixels per unit of the direction table whenever SOME launch of the context takes the compact layout
	bool wide_kernel = false;  // A/B (EBO_EDGE_WIDE): the 168-VGPR instantiation whatever the workgroup size (three 256-lane workgroups per CU)
	int wg_slots = 1;     // workgroups of a k_eval_edge launch (persistent: what the chip holds at once); k_solve_edge: one per unit
	char* d_scratch;      // global fallback, [workgroup slot][stride]
	size_t scratch_stride;
	double* d_sets;       // staging [5][n_units][3] for central differences
	double* d_out;        // [n_flow][3]
	const unsigned char* d_modes = nullptr;  // per flow slot: 0 skip, 1 value, 2 value + Jacobian
	bool for_solve = false;  // the launch is k_solve_edge (picks the workgroup size of its instantiations)
	LiveWindows live;        // n > 0: only these windows (one flow set)
	EvalConsts c;
	EdgeConsts ec;
};
int launch_lds_rate(int atomic, int blocks, int iters, double* d_sink, void* stream);
int launch_stream_yardstick(const uint64_t* d_events, size_t n_events, double* d_image, size_t n_pixels, void* stream);
// what k_count_band needs of a unit of a row shard besides its Unit: times against the WINDOW's reference time,
// the bounding box of its events (they may lie outside the rect: strays take the clamped patch) and its grid patch
struct BandUnit
{
	int32_t dt_win;  // t_ref(window) - t_ref(unit)
	int32_t max_dt;  // max |t_ref(window) - t| over the unit's events
	int16_t x0, x1, y0, y1;  // inclusive bounding box of the unit's events
	int32_t flow;    // grid patch whose flow the unit's events take (feature_detector.cpp:436-441)
	int32_t reserved;
};
static_assert(sizeof(BandUnit) == 24, "BandUnit layout");

struct BandLaunch
{
	const uint64_t* d_events;
	const Unit* d_units;
	const BandUnit* d_band_units;
	int per;        // units per window
	int n_windows;
	const double* d_flows;  // [n_windows][P][2], all grid patches
	int band0, own0, own1, band1;
	unsigned int* d_top;
	unsigned int* d_own;
	unsigned int* d_bottom;
	int* d_escaped;
	EvalConsts c;
};
int launch_count_band(const BandLaunch& L, void* stream);
int launch_count_shard(const uint64_t* d_events, const Unit* d_units, int n_units, int units_per_window,
					   const BandUnit* d_table, const double* d_flows, double* d_image, const EvalConsts& c, void* stream);
int launch_band_finish(const unsigned int* d_own, const unsigned int* d_from_above, const unsigned int* d_from_below,
					   int own_rows, int recv_above, int recv_below, int W, int n_windows, double* d_image, void* stream);
int launch_eval_edge(const EdgeLaunch& L, void* stream);
int launch_solve_edge(const EdgeLaunch& L, const SolveConsts& o, double* d_flows_out, int32_t* d_stats, void* stream);

struct SolveLaunch
{
	const uint64_t* d_events;
	const Unit* d_units;
	int n_units;
	int impl;            // 1 or 2 (see EvalLaunch)
	int cap_doubles;
	int block;
	size_t lds_bytes;
	double* d_flows_out; // [n_flow][2]
	int32_t* d_stats;    // [n_flow][4] or null
	EvalConsts c;
	SolveConsts s;
};
int launch_solve_independent(const SolveLaunch& L, void* stream);

struct CountLaunch
{
	const uint64_t* d_events;
	const Unit* d_units;
	int n_units_total;    // patch units + stray units
	int n_windows;
	int units_per_window; // P + 1
	int mode;             // EBO_COUNT_*
	int impl;             // -1 auto, 0 global int atomics + convert, 1 whole-window LDS bands, 2 patch-row bands, 3 sorted bands
	int lds_kb;           // LDS per band workgroup (0 = default of the implementation)
	unsigned long long* d_overflow;  // impl 2: [1 + total events] count + pixel indices (may be null)
	unsigned int* d_sort_bins;       // impl 3: [3 * sort_bins_cap + 2] counts, starts, cursors (may be null)
	int sort_bins_cap;               // bins (windows x bands) the buffer holds
	unsigned int* d_sorted;          // impl 3: [sorted_cap] destination pixels sorted by band, then [sorted_cap] in event order
	size_t sorted_cap;               // events either half holds
	const int32_t* d_unit_maxdt;     // [units] max |t_ref(window) - t| per unit (impl 4's displacement bound)
	uint64_t max_window_events;
	int any_stray = 0;    // some window has events outside the sensor (impl 6 counts them in a pass of their own)
	const void* d_aux;    // flows f64 [Wn][P][2] or field f32 [Wn][H][W][2]
	int32_t* d_counts;    // [Wn][H][W] scratch, zero on entry, zero on exit
	double* d_image;      // [Wn][H][W]
	EvalConsts c;
};
int launch_count_image(const CountLaunch& L, void* stream);

// Device-side bucketing of raw 24-byte events (ebo_bucket.inc).
struct BucketLaunch
{
	const void* d_raw;                 // ebo_event[] (24 B) or ebo_event8[] (8 B) on the device
	int compact = 0;                   // 1: d_raw holds ebo_event8 records, times relative to d_tbase[window]
	const long long* d_tbase = nullptr; // [n_windows] base times of the compact records
	int w0 = 0, w1 = -1;               // window range of this call ([0, n_windows) when w1 < 0); init runs with w0 == 0
	const unsigned long long* d_offsets; // [n_windows + 1], absolute indices into d_raw
	int n_windows;
	int P;
	int chunk_events;                  // events per chunk of the count / scatter passes: 2048, or 256 for small inputs
	int max_chunks;                    // ceil(max events per window / chunk_events)
	unsigned int min_events;
	int* d_cnt;                        // [n_windows][P+1] scratch (counts)
	unsigned int* d_chunk_hist;        // [n_windows][max_chunks][P+1]: per-chunk histograms, then the chunks' first ranks
	long long* d_tmin;                 // [n_windows][P+1]
	long long* d_tmax;
	Unit* d_units;                     // out [n_windows][P+1]
	long long* d_unit_tref;            // out [n_windows][P+1]
	int32_t* d_unit_maxdt;             // out [n_windows][P+1]: max |t_ref(window) - t| per unit
	long long* d_win_tref;             // out [n_windows]
	uint64_t* d_packed;                // out packed events
	int* d_flag;                       // out error bits
	EvalConsts c;
};
int launch_bucket(const BucketLaunch& L, void* stream);

// FeatureDetector::initMotionField (ebo_field.inc).
struct FieldLaunch
{
	int w, h;
	double scale;
	int use_average;
	int n_patches;
	const unsigned long long* d_off; // [n_patches + 1]
	const double* d_xy;              // [samples][2]
	const long long* d_t;            // [samples]
	long long timestamp;
	float* d_field;                  // out [h][w][2]
	int* d_fixed;                    // out [n_patches][2]
	double* d_avg;                   // scratch [3]
	int* d_nfixed;                   // out
};
int launch_init_field(const FieldLaunch& L, void* stream);

// interpolateMotionField's per-pixel TV problem (ebo_fieldtv.inc / field_tv.cpp).
// n = w * h pixels; "2" vectors hold both flow components of a pixel side by side.
struct TvfArgs
{
	int w, h, n;
	unsigned char* mask;  // 0 free, 1 fixed point, 2 not in the problem (pixel (w-1, h-1))
	double* wh;           // weight of edge p-(p+1), 0 where the block does not exist
	double* wv;           // weight of edge p-(p+w)
	double* deg;          // sum of the incident weights = diag(J'J)
	double* s2;           // Jacobi scaling squared, from iteration 0
	double* diag;         // deg + damping (the CG preconditioner)
	double2* x;           // current point
	double2* xc;          // candidate
	double2* g;           // gradient
	double2* y;           // LM step (unscaled)
	double2* r;
	double2* z;
	double2* p0;
	double2* p1;
	double2* q;
	double* partials;     // [4 * 1024] per-workgroup sums of the last kernel
	double* partials_rz;  // [4 * 1024] same, for r'z / r'r (read while `partials` is rewritten)
	double* scal;         // [32] reduction results, see ebo_fieldtv.inc
	double huber_a;       // 0: no loss (useL1 false); 1e-5: HuberLoss(1e-5)
	double lm_lo, lm_hi;  // min/max_lm_diagonal
};
// One level of the multigrid preconditioner (ebo_fieldtv.inc): a 5-point operator
// (A x)_i = diag_i x_i - sum_e w_e x_j on a w x h grid, wh / wv = weight of the edge to the right /
// lower neighbour (0 where there is none).
struct TvfLevel
{
	int w, h;
	const double* wh;
	const double* wv;
	const double* diag;
	double2* b;   // right-hand side
	double2* x;   // after pre-smoothing
	double2* xo;  // after the coarse correction and post-smoothing
};
struct TvfMg
{
	int levels;          // 0: multigrid off (diagonal preconditioner)
	TvfLevel lv[12];
	double* wh_m[12];    // writable views of lv[l].wh / wv / diag (level 0: masked weights only)
	double* wv_m[12];
	double* diag_m[12];
	double omega, kappa;
	int coarse_sweeps;
};
size_t tvf_workspace_bytes(int w, int h);
void tvf_carve(TvfArgs& A, TvfMg& M, int w, int h, void* base, double2** xbest);
int launch_tvf_mg_build(const TvfArgs& A, const TvfMg& M, void* stream);
int launch_tvf_prepare(const TvfArgs& A, const float* d_field, const int* d_fixed, int n_fixed, void* stream);
int launch_tvf_linearize(const TvfArgs& A, const double2* X, int first, int cost_only, void* stream);
int launch_tvf_cg_init(const TvfArgs& A, const TvfMg& M, double radius, void* stream);
// CG iterations first_iter .. first_iter + iters - 1 (iteration k reads direction buffer k & 1).
int launch_tvf_cg_iters(const TvfArgs& A, const TvfMg& M, int first_iter, int iters, void* stream);
int launch_tvf_model(const TvfArgs& A, void* stream);
int launch_tvf_store(const TvfArgs& A, const double2* X, float* d_field, void* stream);

// The per-feature tracker objective (ebo_optimizer.inc): one tracked patch.
struct OptPatch
{
	double rx, ry;           // patch_.tl()
	int pw, ph;              // int(patch_.width), int(patch_.height)
	unsigned long long off;  // first residual of this patch in the concatenated arrays
};
struct OptLaunch
{
	const double2* d_grid;   // [img_h][img_w] (gradX, gradY)
	int img_w, img_h;
	const OptPatch* d_patches;
	int n_patches;
	int max_pixels;          // max pw * ph over the patches (LDS sizing)
	const double* d_nabla;   // normalizedIntegratedNabla, concatenated
	double* d_x;             // [n][5] = pose (cos, sin, tx, ty), flow direction
	double* d_res;           // eval: residuals, concatenated
	double* d_jac_pose;      // eval: [..][4] or null
	double* d_jac_flow;      // eval: [..] or null
	double* d_stats;         // solve: [n][8]
	double huber;
	SolveConsts s;
};
int launch_optimizer_eval(const OptLaunch& L, void* stream);
int launch_optimizer_solve(const OptLaunch& L, void* stream);
int launch_optimizer_cost_map(const OptLaunch& L, const double* d_xcells, int cells, double* d_out, void* stream);
int launch_optimizer_normalize(const OptPatch* d_patches, int n, const double* d_in, double* d_out, void* stream);
int launch_optimizer_interleave(const double* d_gx, const double* d_gy, size_t n, double2* d_grid, void* stream);
int launch_estimate_num_events(const double2* d_grid, int w, int h, int n, const double* d_rects, const double* d_poses,
								const double* d_flows, double* d_sums, void* stream);

int launch_patch_warp_image(const double2* d_grid, int w, int h, int n, const double* d_rects, const double* d_poses,
							const double* d_flows, const int* d_skip, const size_t* d_offsets, double* d_out, void* stream);

struct PatchIntLaunch
{
	const uint64_t* d_events;  // packed, dt = mid_time - t
	const uint32_t* d_offsets; // [n+1]
	const double* d_rects;     // [n][4]
	const double* d_traj;      // [n][4] = dirX, dirY, tDif, pass(0/1); null for R5
	const uint64_t* d_nabla_off; // [n]
	double* d_nabla;
	int n_patches;
};
int launch_patch_integrate(const PatchIntLaunch& L, void* stream);

// Event -> tracked-patch routing (FeatureDetector::updatePatches): one wave per patch.
struct RouteLaunch
{
	const uint32_t* d_xy;      // [n_events] x:16 | y:16 (two's complement halves)
	uint32_t n_events;
	int n_patches;
	const double* d_rects;     // [n][4]
	const uint32_t* d_start;   // [n]
	const uint32_t* d_take;    // [n]
	uint32_t cap;
	uint32_t* d_index;         // [n][cap]
	uint32_t* d_count;         // [n]
	uint32_t* d_next;          // [n]
};
int launch_route(const RouteLaunch& L, void* stream);

}  // namespace ebo
