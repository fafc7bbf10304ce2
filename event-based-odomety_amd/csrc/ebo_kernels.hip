// ebo_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// Hot path of nurlanov-zh/event-based-odomety's motion compensation:
//   warp each event by its patch's candidate flow over its dt      (contrast_functor.h:47-54)
//   7x7 Gaussian splat into the 3W x 3H image of warped events     (contrast_functor.h:56-86)
//   variance objective + forward-mode Jacobian                     (contrast_functor.h:101-150)
//   per-patch Levenberg-Marquardt on the device                    (feature_detector.cpp:401-414, TV = 0)
//   integer event-count images                                     (feature_detector.cpp:433-482, :270-295; patch.cpp:65-130)
//
// Design (DESIGN.md has the numbers):
//   * events are 8-byte packed records, bucketed by patch, streamed from HBM in
//     coalesced 512-B wave reads; one lane owns one event.
//   * the image of warped events lives only in LDS (planar f64 channels: value,
//     d/dm0, d/dm1); taps are scatter-added with native ds_add_f64; nothing but
//     7 partial sums per workgroup ever goes back to HBM.
//   * reductions: wave64 shuffles, then one LDS hop across waves, fixed order.
//   * no MFMA: this is scatter + reduce, not a contraction.
// Built with -ffp-contract=off: the warped coordinate must round exactly like
// the CPU path (one rounding per operation) because it is truncated to a bin.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "ebo_internal.h"

namespace ebo
{
namespace
{
__device__ __forceinline__ void unpack(uint64_t rec, int& x, int& y, int& pos, int& dt)
{
	const uint32_t lo = static_cast<uint32_t>(rec);
	dt = static_cast<int>(static_cast<uint32_t>(rec >> 32));
	x = static_cast<int>(lo << 17) >> 17;
	y = static_cast<int>(lo << 1) >> 17;
	pos = (lo >> 15) & 1u;
}

// Phase clocks of the edge and variance kernels (build with -DEBO_EDGE_TIMING,
// tools/edge_phase_clock.py, tools/eval_phase_clock.py): every wave keeps the shader-clock cycles
// between two barrier-separated points in scalar accumulators; thread 0 of the workgroup adds
// them to a device table when the unit is done (64 rows by workgroup index: same-address atomics
// from every workgroup at every phase would dominate what is being measured).  Never compiled
// into the shipped library.
#ifdef EBO_EDGE_TIMING
__device__ unsigned long long g_edge_clk[64 * 32];
#define EDGE_TICK_DECL                       \
	unsigned long long edgeClk_ = clock64(); \
	unsigned long long edgeAcc_[24] = {0}
#define EDGE_TICK(k)                                  \
	do                                                \
	{                                                 \
		const unsigned long long now_ = clock64();    \
		edgeAcc_[k] += now_ - edgeClk_;               \
		edgeClk_ = now_;                              \
	} while (0)
#define EDGE_COUNT(k, v) edgeAcc_[k] += static_cast<unsigned long long>(v)
#define EDGE_TICK_FLUSH                                                                       \
	do                                                                                        \
	{                                                                                         \
		if (threadIdx.x == 0)                                                                 \
		{                                                                                     \
			for (int k_ = 0; k_ < 24; ++k_)                                                   \
			{                                                                                 \
				if (edgeAcc_[k_])                                                             \
				{                                                                             \
					atomicAdd(&g_edge_clk[(blockIdx.x & 63) * 32 + k_], edgeAcc_[k_]);        \
				}                                                                             \
			}                                                                                 \
		}                                                                                     \
	} while (0)
#define EDGE_TICK_ARG , unsigned long long &edgeClk_, unsigned long long(&edgeAcc_)[24]
#define EDGE_TICK_PASS , edgeClk_, edgeAcc_
#else
#define EDGE_TICK(k) do { } while (0)
#define EDGE_TICK_DECL do { } while (0)
#define EDGE_COUNT(k, v) do { } while (0)
#define EDGE_TICK_FLUSH do { } while (0)
#define EDGE_TICK_ARG
#define EDGE_TICK_PASS
#endif

__device__ __forceinline__ bool convertible(double c)
{
	return fabs(c) < 1073741824.0;  // int(double) is defined; same guard as the CPU path
}

// Wave-wide reductions on the DPP path (no LDS round trip: __shfl_down is a ds_bpermute, ~100
// cycles of latency per step on the queue the LDS atomics use).  The source lanes of one step:
// quad neighbours, the other pair of the quad, 4 and 8 lanes down the row of 16, then lane 15 of
// the previous row into rows 1 and 3 and lane 31 into rows 2 and 3.  A lane with no source gets
// `idle` (the operation's identity).  The wave's result is in LANE 63 (kWaveResultLane).
// Whole waves only: every launch site of a kernel that reduces uses a workgroup size that is a
// multiple of 64 (the environment overrides are validated), so lane 63 of every wave exists.
constexpr int kWaveResultLane = 63;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_take(int v, int idle)
{
	return __builtin_amdgcn_update_dpp(idle, v, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v, double idle)
{
	const int lo = dpp_take<CTRL, ROW_MASK>(__double2loint(v), __double2loint(idle));
	const int hi = dpp_take<CTRL, ROW_MASK>(__double2hiint(v), __double2hiint(idle));
	return __hiloint2double(hi, lo);
}

// v <- op(v, source lane's v) over the six steps; Op(a, b) must be commutative and associative
// up to what the caller tolerates (sums: a fixed order, so results are reproducible run to run).
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, T idle, Op op)
{
	v = op(v, dpp_take<0xb1, 0xf>(v, idle));   // quad_perm [1,0,3,2]
	v = op(v, dpp_take<0x4e, 0xf>(v, idle));   // quad_perm [2,3,0,1]
	v = op(v, dpp_take<0x114, 0xf>(v, idle));  // row_shr 4
	v = op(v, dpp_take<0x118, 0xf>(v, idle));  // row_shr 8
	v = op(v, dpp_take<0x142, 0xa>(v, idle));  // row_bcast 15 into rows 1, 3
	v = op(v, dpp_take<0x143, 0xc>(v, idle));  // row_bcast 31 into rows 2, 3
	return v;
}

// The fixed-point form of a tap is the double (value + bias) with the bias's high dword taken off its high dword: ONE
// 32-bit subtract, the low dword goes as it is.  Left alone, the compiler re-forms the 64-bit difference (the double as
// an integer minus biasHi << 32) and spends v_subrev_co_u32 (low dword minus 0, for a carry that is always 0) + s_nop +
// v_subb_co_u32 per tap -- 49 times per event in the scatter, 64 times per entry in the edge loss's reverse pass.  An
// empty asm on the high dword keeps the halves apart.
__device__ __forceinline__ void keep32(unsigned int& v)
{
#ifndef EBO_FIX_UNPINNED
	asm volatile("" : "+v"(v));
#endif
}

// One 8-byte LDS read that stays ONE ds_read_b64.  The compiler pairs neighbouring 8-byte reads of one base into
// ds_read2_b64, which the LDS serves as two accesses in four groups of 16 lanes (128 B/clk/CU, banks mod 32) where a
// ds_read_b64 goes in two groups of 32 lanes (256 B/clk/CU, banks mod 64): with per-lane random bases -- every lane
// reads at its own event's footprint -- the paired form is the slower one.  A relaxed wavefront-scope atomic load is
// the same instruction, and the load / store optimiser leaves ordered accesses alone.
__device__ __forceinline__ double lds_ld(const double* p)
{
#ifdef EBO_LDS_PAIRED_READS
	return *p;
#else
	return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
														__HIP_MEMORY_SCOPE_WAVEFRONT));
#endif
}

__device__ __forceinline__ double wave_sum(double v)  // total in lane kWaveResultLane
{
	return wave_reduce(v, 0.0, [](double a, double b) { return a + b; });
}

__device__ __forceinline__ int wave_min(int v)
{
	return wave_reduce(v, 0x7fffffff, [](int a, int b) { return min(a, b); });
}

__device__ __forceinline__ int wave_max(int v)
{
	return wave_reduce(v, static_cast<int>(0x80000000u), [](int a, int b) { return max(a, b); });
}

// Sums NV per-thread values over the workgroup; every thread gets the totals.
// Order is fixed (lanes by the DPP tree of wave_reduce, then waves 0..nw-1) => deterministic.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red)
{
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
	for (int k = 0; k < NV; ++k)
	{
		v[k] = wave_sum(v[k]);
	}
	__syncthreads();  // red may still be read by a previous call
	if (lane == kWaveResultLane)
	{
#pragma unroll
		for (int k = 0; k < NV; ++k)
		{
			red[wave * 8 + k] = v[k];
		}
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < NV; ++k)
	{
		double s = red[k];
		for (int w = 1; w < nw; ++w)
		{
			s += red[w * 8 + k];
		}
		v[k] = s;
	}
}

// Up to 16 values per thread in ONE pass (two barriers) for workgroups of at most eight waves: the same wave
// reduction and the same order over the waves per value as block_sum, so a value has the bits block_sum gives it.
template <int NV>
__device__ __forceinline__ void block_sum_wide(double (&v)[NV], double* red)
{
	static_assert(NV <= 16, "block_sum_wide: at most 16 values");
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int nw = (blockDim.x + 63) >> 6;  // <= 8: red holds 128 doubles
#pragma unroll
	for (int k = 0; k < NV; ++k)
	{
		v[k] = wave_sum(v[k]);
	}
	__syncthreads();  // red may still be read by a previous call
	if (lane == kWaveResultLane)
	{
#pragma unroll
		for (int k = 0; k < NV; ++k)
		{
			red[wave * 16 + k] = v[k];
		}
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < NV; ++k)
	{
		double s = red[k];
		for (int w = 1; w < nw; ++w)
		{
			s += red[w * 16 + k];
		}
		v[k] = s;
	}
}

// ---------------------------------------------------------------------------
// Warp + 7x7 Gaussian splat of one unit's events into rows [r0, r0+rows) of its
// 3W x 3H image (contrast_functor.h:38-88).  C = 1: value only (the T=double
// instantiation); C = 3: value, d/dm0, d/dm1 (the Jet<double,2> instantiation).
//   c = p + (t_ref - t) * scale * m;  b = int(c)  (truncation)
//   tap (i,j): w = N * exp(hs * ((b.x+i-c.x)^2 + (b.y+j-c.y)^2))
//   dw/dm0 = w * (b.x+i-c.x) * tau / sigma^2,   dw/dm1 likewise in y
// The Gaussian is evaluated as a product of two 1-D factors (7+7 exps instead
// of 49): same real number, last-bit differences only.
// ---------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void splat_rows(const uint64_t* __restrict__ ev, uint32_t nEv,
											int rx, int ry, int rw, int rh, int r0, int rows,
											double m0, double m1, const EvalConsts& c,
											double* __restrict__ img, int plane)
{
	const int W3 = 3 * rw;
	for (uint32_t e = threadIdx.x; e < nEv; e += blockDim.x)
	{
		int x, y, pos, dt;
		unpack(ev[e], x, y, pos, dt);
		const double tau = static_cast<double>(dt) * c.scale;
		const double cx = static_cast<double>(x) + tau * m0;
		const double cy = static_cast<double>(y) + tau * m1;
		if (!convertible(cx) || !convertible(cy))
		{
			continue;
		}
		const int bx = static_cast<int>(cx);
		const int by = static_cast<int>(cy);
		const int pxc = bx - rx + rw;       // column of tap i = 0
		const int pyc = by - ry + rh - r0;  // tile-local row of tap j = 0
		if (pxc + 3 < 0 || pxc - 3 >= W3 || pyc + 3 < 0 || pyc - 3 >= rows)
		{
			continue;
		}
		const double fx = cx - static_cast<double>(bx);  // exact
		const double fy = cy - static_cast<double>(by);
		const double g = tau * c.inv_sigsq;

		double wx[7], wy[7], ax[7], ay[7];
#pragma unroll
		for (int k = 0; k < 7; ++k)
		{
			const double dx = static_cast<double>(k - 3) - fx;  // == (b.x+i) - c.x
			const double dy = static_cast<double>(k - 3) - fy;
			wx[k] = c.norm * exp(c.hs * (dx * dx));
			wy[k] = exp(c.hs * (dy * dy));
			if (C == 3)
			{
				ax[k] = wx[k] * (g * dx);
				ay[k] = wy[k] * (g * dy);
			}
		}
#pragma unroll
		for (int j = 0; j < 7; ++j)
		{
			const int row = pyc + j - 3;
			if (row < 0 || row >= rows)
			{
				continue;
			}
			double* rowp = img + row * W3 + (pxc - 3);
#pragma unroll
			for (int i = 0; i < 7; ++i)
			{
				const int col = pxc + i - 3;
				if (col < 0 || col >= W3)
				{
					continue;
				}
				atomicAdd(rowp + i, wx[i] * wy[j]);
				if (C == 3)
				{
					atomicAdd(rowp + plane + i, ax[i] * wy[j]);
					atomicAdd(rowp + 2 * plane + i, wx[i] * ay[j]);
				}
			}
		}
	}
}

// Sums over the pixels with I > 0 of one tile (contrast_functor.h:111-121,
// :129-139 folded into one pass): S1 = sum I, S2 = sum I^2, n, D1k = sum dIk,
// D2k = sum I dIk.
template <int C>
__device__ __forceinline__ void tile_sums(const double* __restrict__ img, int plane,
										   double* red, double (&out)[7])
{
	double v[7] = {0, 0, 0, 0, 0, 0, 0};
	for (int p = threadIdx.x; p < plane; p += blockDim.x)
	{
		const double I = img[p];
		if (I > 0.0)
		{
			v[0] += I;
			v[1] += I * I;
			v[2] += 1.0;
			if (C == 3)
			{
				const double a = img[plane + p];
				const double b = img[2 * plane + p];
				v[3] += a;
				v[4] += b;
				v[5] += I * a;
				v[6] += I * b;
			}
		}
	}
	block_sum<7>(v, red);
#pragma unroll
	for (int k = 0; k < 7; ++k)
	{
		out[k] = v[k];
	}
}

// contrast_functor.h:122-149 from the sums.  counterNonZero starts at 1 (:110).
//   mean = S1/cnt;  var = sum_{I>0}(I-mean)^2 / cnt = (S2 - 2 mean S1 + n mean^2)/cnt
//   r = maxRes - var, or the out-of-window penalty maxRes (1 + m0^2 + m1^2) if mean <= 0.
__device__ __forceinline__ void variance_from_sums(const double* S, bool wantJac, double m0,
													double m1, double maxRes, double& r,
													double& j0, double& j1)
{
	const double n = S[2];
	const double cntInv = 1.0 / (n + 1.0);
	const double mean = S[0] * cntInv;
	if (mean > 0.0)
	{
		const double var = (S[1] - 2.0 * mean * S[0] + n * mean * mean) * cntInv;
		r = maxRes - var;
		if (wantJac)
		{
			const double dm0 = S[3] * cntInv;
			const double dm1 = S[4] * cntInv;
			j0 = -(2.0 * (S[5] - mean * S[3] - dm0 * S[0] + n * mean * dm0) * cntInv);
			j1 = -(2.0 * (S[6] - mean * S[4] - dm1 * S[0] + n * mean * dm1) * cntInv);
		}
	}
	else
	{
		r = maxRes * (1.0 + m0 * m0 + m1 * m1);
		j0 = maxRes * (m0 + m0);
		j1 = maxRes * (m1 + m1);
	}
}

__device__ __forceinline__ void fd_offset(int set, double h, double& m0, double& m1)
{
	if (set == 1) m0 += h;
	if (set == 2) m0 -= h;
	if (set == 3) m1 += h;
	if (set == 4) m1 -= h;
}

// ===========================================================================
// Second-generation evaluation (impl 1 / 2): scatter the VALUE, gather the
// DERIVATIVES.
//
// The Jacobian of the variance objective needs only  D1k = sum_px dI_k  and
// D2k = sum_px I dI_k  over the touched pixels (every touched pixel has I > 0:
// Gaussian taps are strictly positive).  With dI_k(px) = sum_e d_k(e, px):
//     D1k = sum_e sum_taps d_k(e,tap)                    -- no image at all
//     D2k = sum_e sum_taps I(px(e,tap)) d_k(e,tap)       -- a READ of the value image
// i.e. forward-mode Jets are re-associated into "value image, then one gather
// pass": 49 LDS atomics + 49 LDS reads per event instead of 147 atomics, and a
// third of the LDS.  Mathematically identical to the Jet result.
//
// Further: (a) the image covers only the bounding box of the warped events inside
// the 3W x 3H canvas (the canvas is mostly empty), split in `tiles` row bands
// (parallel workgroups) and, if a band exceeds the workgroup's LDS, in sequential
// sub-bands; (b) impl 2 accumulates taps as exact 64-bit fixed point
// (ds_add_u64): tap values are < 0.5, so (v + 1.5) has ulp 2^-52 and its mantissa
// IS the fixed-point number -- one v_add_f64 + a 64-bit integer subtract.  Integer
// adds commute: the image, and with the fixed reduction order the whole result,
// is bit-reproducible from run to run; it is also faster than ds_add_f64 under
// same-address conflicts (tools/microbench/lds_atomics.hip); (c) the 7 taps of
// an axis come from 3 exps:  exp(hs (k-f)^2) = exp(hs k^2) exp(hs f^2) exp(f/s^2)^k.
// ===========================================================================
constexpr int kRedDoubles = 128;  // 16 waves x 8
constexpr int kLdsHeader = 160;   // red[128] + 32 doubles of int scratch

__device__ __forceinline__ bool warp_event(uint64_t rec, int rx, int ry, int rw, int rh,
											double m0, double m1, const EvalConsts& c, int& pxc,
											int& pyc, double& fx, double& fy, double& tau)
{
	int x, y, pos, dt;
	unpack(rec, x, y, pos, dt);
	tau = static_cast<double>(dt) * c.scale;
	const double cx = static_cast<double>(x) + tau * m0;  // not fused: -ffp-contract=off
	const double cy = static_cast<double>(y) + tau * m1;
	if (!convertible(cx) || !convertible(cy))
	{
		return false;
	}
	const int bx = static_cast<int>(cx);  // truncation (contrast_functor.h:59,63)
	const int by = static_cast<int>(cy);
	pxc = bx - rx + rw;  // canvas column / row of the centre tap
	pyc = by - ry + rh;
	if (pxc + 3 < 0 || pxc - 3 >= 3 * rw || pyc + 3 < 0 || pyc - 3 >= 3 * rh)
	{
		return false;
	}
	fx = cx - static_cast<double>(bx);  // exact
	fy = cy - static_cast<double>(by);
	return true;
}

// w[k] = pre * exp(hs (k-3-f)^2), k = 0..6, from three exps.
__device__ __forceinline__ void axis_taps(double f, double pre, const EvalConsts& c, double (&w)[7])
{
	const double e0 = pre * exp(c.hs * (f * f));
	const double a = f * c.inv_sigsq;
	const double p = exp(a);
	const double q = exp(-a);
	const double p2 = p * p, q2 = q * q;
	const double e1 = c.ck1 * e0, e2 = c.ck2 * e0, e3 = c.ck3 * e0;
	w[3] = e0;
	w[4] = e1 * p;
	w[5] = e2 * p2;
	w[6] = e3 * (p2 * p);
	w[2] = e1 * q;
	w[1] = e2 * q2;
	w[0] = e3 * (q2 * q);
}

__device__ __forceinline__ void block_minmax(int& xmin, int& xmax, int& ymin, int& ymax, int* ired)
{
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int nw = (blockDim.x + 63) >> 6;
	xmin = wave_min(xmin);
	xmax = wave_max(xmax);
	ymin = wave_min(ymin);
	ymax = wave_max(ymax);
	__syncthreads();
	if (lane == kWaveResultLane)
	{
		ired[wave * 4 + 0] = xmin;
		ired[wave * 4 + 1] = xmax;
		ired[wave * 4 + 2] = ymin;
		ired[wave * 4 + 3] = ymax;
	}
	__syncthreads();
	xmin = ired[0];
	xmax = ired[1];
	ymin = ired[2];
	ymax = ired[3];
	for (int w = 1; w < nw; ++w)
	{
		xmin = min(xmin, ired[w * 4 + 0]);
		xmax = max(xmax, ired[w * 4 + 1]);
		ymin = min(ymin, ired[w * 4 + 2]);
		ymax = max(ymax, ired[w * 4 + 3]);
	}
}

// w'[k] = w[(k + r) mod 7], r in 0..6: a 3-stage barrel rotation on registers
// (v_cndmask only; a runtime-indexed register array would go to scratch).
__device__ __forceinline__ void rotate7(double (&w)[7], int r)
{
	if (r & 1)
	{
		const double t = w[0];
		w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; w[5] = w[6]; w[6] = t;
	}
	if (r & 2)
	{
		const double t0 = w[0], t1 = w[1];
		w[0] = w[2]; w[1] = w[3]; w[2] = w[4]; w[3] = w[5]; w[4] = w[6]; w[5] = t0; w[6] = t1;
	}
	if (r & 4)
	{
		const double t0 = w[0], t1 = w[1], t2 = w[2], t3 = w[3];
		w[0] = w[4]; w[1] = w[5]; w[2] = w[6]; w[3] = t0; w[4] = t1; w[5] = t2; w[6] = t3;
	}
}

// The seven sums of one (unit, row band) at flow (m0, m1); every thread returns
// the same totals.  lds: [red 128][int scratch 32][image capDoubles].
template <bool FIXED, bool ROT>
__device__ __forceinline__ void eval_unit2(const uint64_t* __restrict__ ev, const Unit& u, double m0,
											double m1, bool wantJac, int tile, int tiles,
											int capDoubles, const EvalConsts& c, double* lds,
											double (&S)[7])
{
	double* red = lds;
	int* ired = reinterpret_cast<int*>(lds + kRedDoubles);
	double* img = lds + kLdsHeader;
	unsigned long long* imgq = reinterpret_cast<unsigned long long*>(img);
	const int rx = u.rx, ry = u.ry, rw = u.rw, rh = u.rh;
	const int W3 = 3 * rw, H3 = 3 * rh;
	const uint32_t nEv = u.n_ev;
#pragma unroll
	for (int k = 0; k < 7; ++k)
	{
		S[k] = 0.0;
	}

	// ---- pass A: bounding box of the centre taps that touch the canvas ----
	int xmin = 0x7fffffff, xmax = -0x7fffffff, ymin = 0x7fffffff, ymax = -0x7fffffff;
	for (uint32_t e = threadIdx.x; e < nEv; e += blockDim.x)
	{
		int pxc, pyc;
		double fx, fy, tau;
		if (warp_event(ev[e], rx, ry, rw, rh, m0, m1, c, pxc, pyc, fx, fy, tau))
		{
			xmin = min(xmin, pxc);
			xmax = max(xmax, pxc);
			ymin = min(ymin, pyc);
			ymax = max(ymax, pyc);
		}
	}
	block_minmax(xmin, xmax, ymin, ymax, ired);
	if (xmin > xmax)
	{
		return;  // nothing lands in the window: all sums 0 => penalty branch
	}
	const int x0 = max(xmin - 3, 0), x1 = min(xmax + 3, W3 - 1);
	const int y0 = max(ymin - 3, 0), y1 = min(ymax + 3, H3 - 1);
	const int cols = x1 - x0 + 1;
	const int rowsAll = y1 - y0 + 1;
	const int R = (rowsAll + tiles - 1) / tiles;
	const int ty0 = y0 + tile * R;
	const int ty1 = min(ty0 + R, y1 + 1);
	const int maxRows = max(capDoubles / cols, 1);
	const unsigned long long biasBits = static_cast<unsigned long long>(__double_as_longlong(c.fix_bias));
	const int lane = threadIdx.x & 63;
	const int rotRow = ROT ? (lane % 7) : 0;
	const int rotCol = ROT ? ((lane / 7) % 7) : 0;

	for (int sy0 = ty0; sy0 < ty1; sy0 += maxRows)
	{
		const int srows = min(maxRows, ty1 - sy0);
		const int npx = srows * cols;
		__syncthreads();
		for (int i = threadIdx.x; i < npx; i += blockDim.x)
		{
			img[i] = 0.0;  // also the all-zero bit pattern of the fixed-point image
		}
		__syncthreads();

		// ---- pass B: scatter the value taps ----
		for (uint32_t e = threadIdx.x; e < nEv; e += blockDim.x)
		{
			int pxc, pyc;
			double fx, fy, tau;
			if (!warp_event(ev[e], rx, ry, rw, rh, m0, m1, c, pxc, pyc, fx, fy, tau))
			{
				continue;
			}
			const int rowLo = pyc - 3 - sy0;  // band-local row of tap j = 0
			if (rowLo + 6 < 0 || rowLo >= srows)
			{
				continue;
			}
			double wx[7], wy[7];
			axis_taps(fx, c.norm, c, wx);
			axis_taps(fy, 1.0, c, wy);
			const int colLo = pxc - 3 - x0;
			// Tap rotation: lane l walks the 7x7 taps starting at (l%7, (l/7)%7).  Two
			// lanes whose events share a centre pixel then never address the same pixel
			// in the same wave instruction (same-address LDS atomics serialise).
			rotate7(wy, rotRow);
			rotate7(wx, rotCol);
#pragma unroll
			for (int jr = 0; jr < 7; ++jr)
			{
				int j = jr + rotRow;
				j -= (j >= 7) ? 7 : 0;
				const int row = rowLo + j;
				if (row < 0 || row >= srows)
				{
					continue;
				}
				const int rowBase = row * cols + colLo;
#pragma unroll
				for (int ir = 0; ir < 7; ++ir)
				{
					int i = ir + rotCol;
					i -= (i >= 7) ? 7 : 0;
					const int col = colLo + i;
					if (col < 0 || col >= cols)
					{
						continue;
					}
					const double v = wx[ir] * wy[jr];
					if (FIXED)
					{
						const unsigned long long q =
							static_cast<unsigned long long>(__double_as_longlong(v + c.fix_bias)) - biasBits;
						atomicAdd(&imgq[rowBase + i], q);
					}
					else
					{
						atomicAdd(&img[rowBase + i], v);
					}
				}
			}
		}
		__syncthreads();

		// ---- pass C: pixel sums (contrast_functor.h:111-121, :129-139) ----
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			double I;
			if (FIXED)
			{
				const unsigned long long q = imgq[p];
				I = static_cast<double>(q) * c.fix_scale;
				img[p] = I;
			}
			else
			{
				I = img[p];
			}
			if (I > 0.0)
			{
				S[0] += I;
				S[1] = fma(I, I, S[1]);
				S[2] += 1.0;
			}
		}
		if (!wantJac)
		{
			continue;
		}
		__syncthreads();

		// ---- pass D: gather the derivative sums ----
		for (uint32_t e = threadIdx.x; e < nEv; e += blockDim.x)
		{
			int pxc, pyc;
			double fx, fy, tau;
			if (!warp_event(ev[e], rx, ry, rw, rh, m0, m1, c, pxc, pyc, fx, fy, tau))
			{
				continue;
			}
			const int rowLo = pyc - 3 - sy0;
			if (rowLo + 6 < 0 || rowLo >= srows)
			{
				continue;
			}
			double wx[7], wy[7];
			axis_taps(fx, c.norm, c, wx);
			axis_taps(fy, 1.0, c, wy);
			const int colLo = pxc - 3 - x0;
			const double g = tau * c.inv_sigsq;
			// columns outside the canvas carry no tap: zero weight, clamped address
			double ax[7];
			int ci[7];
			double sumW = 0.0, sumA = 0.0;
#pragma unroll
			for (int i = 0; i < 7; ++i)
			{
				const int col = colLo + i;
				const bool ok = col >= 0 && col < cols;
				wx[i] = ok ? wx[i] : 0.0;
				ax[i] = wx[i] * (g * (static_cast<double>(i - 3) - fx));
				ci[i] = min(max(col, 0), cols - 1);
				sumW += wx[i];
				sumA += ax[i];
			}
			double sumWy = 0.0, sumAy = 0.0, d2a = 0.0, d2b = 0.0;
#pragma unroll
			for (int j = 0; j < 7; ++j)
			{
				const int row = rowLo + j;
				if (row < 0 || row >= srows)
				{
					continue;
				}
				const double ay = wy[j] * (g * (static_cast<double>(j - 3) - fy));
				sumWy += wy[j];
				sumAy += ay;
				const double* rowp = img + row * cols;
				double ra = 0.0, rb = 0.0;
#pragma unroll
				for (int i = 0; i < 7; ++i)
				{
					const double I = rowp[ci[i]];
					ra = fma(ax[i], I, ra);
					rb = fma(wx[i], I, rb);
				}
				d2a = fma(wy[j], ra, d2a);
				d2b = fma(ay, rb, d2b);
			}
			S[3] = fma(sumA, sumWy, S[3]);
			S[4] = fma(sumW, sumAy, S[4]);
			S[5] += d2a;
			S[6] += d2b;
		}
	}
	block_sum<7>(S, red);
}

#ifdef EBO_AB  // second-generation evaluation (impl 1 / 2): kept for A/B against k_eval3, not shipped
template <bool FIXED, bool ROT>
__global__ void __launch_bounds__(512) k_eval2(const uint64_t* __restrict__ events, const Unit* __restrict__ units,
						const double* __restrict__ flows, int tiles, int wantJac, int capDoubles,
						double fdStep, double* __restrict__ partials, double* __restrict__ out,
						EvalConsts c)
{
	extern __shared__ double lds[];
	const int unit = blockIdx.x / tiles;
	const int tile = blockIdx.x - unit * tiles;
	const int set = blockIdx.y;
	const Unit u = units[unit];
	double* part = partials + ((static_cast<size_t>(set) * gridDim.x) + blockIdx.x) * kPartialStride;
	const bool fused = (tiles == 1 && gridDim.y == 1);
	if (!(u.flags & kUnitActive))
	{
		if (!fused && threadIdx.x < 7)
		{
			part[threadIdx.x] = 0.0;
		}
		if (fused && !(u.flags & kUnitStray) && threadIdx.x < 3)
		{
			out[3 * u.flow_idx + threadIdx.x] = 0.0;
		}
		return;
	}
	double m0 = flows[2 * u.flow_idx];
	double m1 = flows[2 * u.flow_idx + 1];
	fd_offset(set, fdStep, m0, m1);
	double S[7];
	eval_unit2<FIXED, ROT>(events + u.ev_off, u, m0, m1, wantJac != 0, tile, tiles, capDoubles, c, lds,
						   S);
	if (threadIdx.x == 0)
	{
		if (fused)
		{
			double r, j0 = 0.0, j1 = 0.0;
			variance_from_sums(S, wantJac != 0, m0, m1, c.max_res, r, j0, j1);
			out[3 * u.flow_idx + 0] = r;
			out[3 * u.flow_idx + 1] = j0;
			out[3 * u.flow_idx + 2] = j1;
		}
		else
		{
#pragma unroll
			for (int k = 0; k < 7; ++k)
			{
				part[k] = S[k];
			}
		}
	}
}
#endif  // EBO_AB

#ifdef EBO_AB
// ---------------------------------------------------------------------------
// First-generation batched evaluation (impl 0, kept for A/B): full 3W x 3H canvas,
// one f64 atomic per tap and channel.  Workgroup = (flow set, unit, row tile).
// ---------------------------------------------------------------------------
template <int C>
__global__ void k_eval_variance(const uint64_t* __restrict__ events,
								const Unit* __restrict__ units, const double* __restrict__ flows,
								int tiles, double fdStep, double* __restrict__ partials,
								double* __restrict__ out, EvalConsts c)
{
	extern __shared__ double lds[];
	const int unit = blockIdx.x / tiles;
	const int tile = blockIdx.x - unit * tiles;
	const int set = blockIdx.y;
	const Unit u = units[unit];
	double* part = partials + ((static_cast<size_t>(set) * gridDim.x) + blockIdx.x) * kPartialStride;
	const bool fused = (tiles == 1 && gridDim.y == 1);
	if (!(u.flags & kUnitActive))
	{
		if (threadIdx.x < 7)
		{
			part[threadIdx.x] = 0.0;
		}
		if (fused && !(u.flags & kUnitStray) && threadIdx.x < 3)
		{
			out[3 * u.flow_idx + threadIdx.x] = 0.0;
		}
		return;
	}
	const int W3 = 3 * u.rw;
	const int H3 = 3 * u.rh;
	const int R = (H3 + tiles - 1) / tiles;
	const int r0 = tile * R;
	const int rows = min(R, H3 - r0);
	double S[7] = {0, 0, 0, 0, 0, 0, 0};
	double m0 = flows[2 * u.flow_idx];
	double m1 = flows[2 * u.flow_idx + 1];
	fd_offset(set, fdStep, m0, m1);
	if (rows > 0)
	{
		const int plane = rows * W3;
		double* red = lds + C * plane;
		for (int i = threadIdx.x; i < C * plane; i += blockDim.x)
		{
			lds[i] = 0.0;
		}
		__syncthreads();
		splat_rows<C>(events + u.ev_off, u.n_ev, u.rx, u.ry, u.rw, u.rh, r0, rows, m0, m1, c,
					  lds, plane);
		__syncthreads();
		tile_sums<C>(lds, plane, red, S);
	}
	if (threadIdx.x == 0)
	{
#pragma unroll
		for (int k = 0; k < 7; ++k)
		{
			part[k] = S[k];
		}
		if (fused)
		{
			double r, j0 = 0.0, j1 = 0.0;
			variance_from_sums(S, C == 3, m0, m1, c.max_res, r, j0, j1);
			out[3 * u.flow_idx + 0] = r;
			out[3 * u.flow_idx + 1] = j0;
			out[3 * u.flow_idx + 2] = j1;
		}
	}
}
#endif  // EBO_AB

// Adds the row tiles of each unit in tile order and finishes the objective.
// flow sets: 1 (C = 1 or 3), or 5 value-only sets for central differences.
__global__ void k_combine_variance(const Unit* __restrict__ units, int nUnits,
								   const double* __restrict__ flows, int tiles, int sets,
								   int channels, double fdStep,
								   const double* __restrict__ partials, double* __restrict__ out,
								   EvalConsts c)
{
	const int unit = blockIdx.x * blockDim.x + threadIdx.x;
	if (unit >= nUnits)
	{
		return;
	}
	const Unit u = units[unit];
	if (u.flags & kUnitStray)
	{
		return;  // stray buckets carry no objective and own no output slot
	}
	double res[5] = {0, 0, 0, 0, 0};
	double j0 = 0.0, j1 = 0.0;
	if (u.flags & kUnitActive)
	{
		for (int s = 0; s < sets; ++s)
		{
			double S[7] = {0, 0, 0, 0, 0, 0, 0};
			const double* p =
				partials + ((static_cast<size_t>(s) * nUnits + unit) * tiles) * kPartialStride;
			for (int t = 0; t < tiles; ++t)
			{
				for (int k = 0; k < 7; ++k)
				{
					S[k] += p[t * kPartialStride + k];
				}
			}
			double m0 = flows[2 * u.flow_idx];
			double m1 = flows[2 * u.flow_idx + 1];
			fd_offset(s, fdStep, m0, m1);
			double a = 0.0, b = 0.0;
			variance_from_sums(S, channels == 3, m0, m1, c.max_res, res[s], a, b);
			if (s == 0)
			{
				j0 = a;
				j1 = b;
			}
		}
		if (sets == 5)
		{
			j0 = (res[1] - res[2]) / (2.0 * fdStep);
			j1 = (res[3] - res[4]) / (2.0 * fdStep);
		}
	}
	out[3 * u.flow_idx + 0] = res[0];
	out[3 * u.flow_idx + 1] = j0;
	out[3 * u.flow_idx + 2] = j1;
}

// Diagnostic: the image of warped events of one unit, tile by tile, to HBM.
template <int C>
__global__ void k_dump_image(const uint64_t* __restrict__ events, const Unit* __restrict__ units,
							 int unit, const double* __restrict__ flow, int tiles,
							 double* __restrict__ image, EvalConsts c)
{
	extern __shared__ double lds[];
	const Unit u = units[unit];
	const int W3 = 3 * u.rw;
	const int H3 = 3 * u.rh;
	const int R = (H3 + tiles - 1) / tiles;
	const size_t full = static_cast<size_t>(W3) * H3;
	for (int tile = 0; tile < tiles; ++tile)
	{
		const int r0 = tile * R;
		const int rows = min(R, H3 - r0);
		if (rows <= 0)
		{
			break;
		}
		const int plane = rows * W3;
		for (int i = threadIdx.x; i < C * plane; i += blockDim.x)
		{
			lds[i] = 0.0;
		}
		__syncthreads();
		splat_rows<C>(events + u.ev_off, u.n_ev, u.rx, u.ry, u.rw, u.rh, r0, rows, flow[0],
					  flow[1], c, lds, plane);
		__syncthreads();
		for (int ch = 0; ch < C; ++ch)
		{
			for (int i = threadIdx.x; i < plane; i += blockDim.x)
			{
				image[ch * full + static_cast<size_t>(r0) * W3 + i] = lds[ch * plane + i];
			}
		}
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------
// Device-resident solve, one workgroup per patch (EBO_SOLVE_INDEPENDENT).
// Trust-region Levenberg-Marquardt exactly as ebo_solver_opts describes it
// (Ceres 2.0: TrustRegionMinimizer, LevenbergMarquardtStrategy,
// TrustRegionStepEvaluator) on a 1-residual / 2-parameter problem, options of
// feature_detector.cpp:401-410.  Every thread carries the (uniform) solver state;
// the objective is evaluated cooperatively.  No host round trips.
// ---------------------------------------------------------------------------
#include "ebo_eval3.inc"

// FIXED: 1 = impl 1 (f64 atomics), 2 = impl 2, 3 = impl 3 with exp_small, 4 = impl 3
// with the library exp.
template <int FIXED>
__device__ __forceinline__ void eval_unit(const uint64_t* __restrict__ ev, const Unit& u,
										   double m0, double m1, bool wantJac, int capDoubles,
										   const EvalConsts& c, double* lds, double& r, double& j0,
										   double& j1, EvalReuse* ru = nullptr)
{
	double S[7];
	if (FIXED == 3)
	{
		eval_unit3<true>(ev, u, m0, m1, wantJac, 0, 1, capDoubles, c, lds, S, ru);
	}
	else if (FIXED == 4)
	{
		eval_unit3<false>(ev, u, m0, m1, wantJac, 0, 1, capDoubles, c, lds, S, ru);
	}
	else
	{
		eval_unit2<FIXED == 2, false>(ev, u, m0, m1, wantJac, 0, 1, capDoubles, c, lds, S);
	}
	j0 = 0.0;
	j1 = 0.0;
	variance_from_sums(S, wantJac, m0, m1, c.max_res, r, j0, j1);
}

// The whole trust-region LM of ONE 2-parameter, 1-residual problem (the per-patch problem of
// EBO_SOLVE_INDEPENDENT; Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy with the options
// of feature_detector.cpp:401-410) as a resumable state machine, replicated in every thread of the
// workgroup (uniform control flow).  The kernel owns ONE evaluation site:
//     lm.begin();  do { evaluate(lm.q0, lm.q1, lm.qJac) -> r, J0, J1 } while (lm.advance(r, J0, J1, o));
// (three inlined copies of an objective as large as the edge loss made the register allocator
// spill several hundred VGPRs).  Value-only for the cost at a candidate, value + Jacobian after an
// accepted step -- the sequence Ceres follows.
struct LmUnit
{
	// the evaluation wanted next
	double q0, q1;
	bool qJac;
	// results
	double best0, best1;
	int iteration, evalsCost, evalsJac, termination;
	// solver state
	int phase;  // 0: first evaluation, 1: cost at a candidate, 2: Jacobian at an accepted point
	double x0, x1, f, J0, J1, xCost, g0, g1, sc0, sc1, j0, j1, xNorm, gradMax, minimumCost;
	double seMinimum, seCurrent, seReference, seCandidate, seAccRef, seAccCand;
	int seNumNonmono;
	double radius, decreaseFactor, d0, d1;
	bool reuseDiagonal, lastSuccessful;
	int numInvalid;
	double c0, c1, modelCostChange, candCost, quality;

	// Where the objective needs the registers (the edge loss), the state lives in LDS and ONE lane
	// runs the solver between two evaluations (k_solve_edge): replicated in every lane it is ~100
	// VGPRs that stay live across the objective (176-185 spilled VGPRs, ~0.5 KB of scratch per lane in
	// round 2), and replicated in every WAVE the solver's serial f64 code (divisions, square roots)
	// cost the workgroup as many vector instructions as a value-only evaluation itself.
	int more;  // k_solve_edge: advance()'s answer, for the other waves

	__device__ __forceinline__ void begin()
	{
		best0 = best1 = 0.0;
		iteration = evalsCost = evalsJac = 0;
		termination = 1;
		phase = 0;
		more = 1;
		x0 = x1 = 0.0;  // feature_detector.cpp:318-326
		q0 = q1 = 0.0;
		qJac = true;
	}

	// the loop of TrustRegionMinimizer from its top to the next evaluation; false = finished
	__device__ __forceinline__ bool next_step(const SolveConsts& o)
	{
		for (;;)
		{
			if (lastSuccessful && xCost < minimumCost)
			{
				minimumCost = xCost;
				best0 = x0;
				best1 = x1;
			}
			if (iteration >= o.max_num_iterations)
			{
				termination = 1;
				return false;
			}
			if (lastSuccessful && gradMax <= o.gradient_tolerance)
			{
				termination = 0;
				return false;
			}
			if (radius < o.min_radius)
			{
				termination = 0;
				return false;
			}
			iteration++;
			lastSuccessful = false;

			if (!reuseDiagonal)
			{
				d0 = fmin(fmax(j0 * j0, o.min_lm_diagonal), o.max_lm_diagonal);
				d1 = fmin(fmax(j1 * j1, o.min_lm_diagonal), o.max_lm_diagonal);
			}
			const double l0 = sqrt(d0 / radius);
			const double l1 = sqrt(d1 / radius);
			reuseDiagonal = true;
			// (J'J + D'D) y = J'f by Cholesky; step = -y.
			const double h00 = j0 * j0 + l0 * l0;
			const double h10 = j1 * j0;
			const double h11 = j1 * j1 + l1 * l1;
			bool valid = (h00 > 0.0) && isfinite(h00);
			double s0 = 0.0, s1 = 0.0;
			if (valid)
			{
				const double L00 = sqrt(h00);
				const double L10 = h10 / L00;
				const double dd = h11 - L10 * L10;
				valid = (dd > 0.0) && isfinite(dd);
				if (valid)
				{
					const double L11 = sqrt(dd);
					double b0 = (j0 * f) / L00;
					double b1 = ((j1 * f) - L10 * b0) / L11;
					b1 = b1 / L11;
					b0 = (b0 - L10 * b1) / L00;
					valid = isfinite(b0) && isfinite(b1);
					s0 = -b0;
					s1 = -b1;
				}
			}
			modelCostChange = 0.0;
			if (valid)
			{
				const double mr = j0 * s0 + j1 * s1;
				modelCostChange = 0.0 - mr * (f + mr / 2.0);
				valid = modelCostChange > 0.0;
			}
			if (!valid)
			{
				numInvalid++;
				if (numInvalid >= o.max_invalid)
				{
					termination = 2;
					return false;
				}
				radius *= 0.5;
				reuseDiagonal = true;
				continue;
			}
			numInvalid = 0;
			c0 = x0 + s0 * sc0;
			c1 = x1 + s1 * sc1;
			q0 = c0;
			q1 = c1;
			qJac = false;
			phase = 1;
			return true;
		}
	}

	// takes the result of the evaluation asked for; true = another evaluation is wanted
	__device__ __forceinline__ bool advance(double r, double a, double b, const SolveConsts& o)
	{
		if (phase == 0)
		{
			f = r;
			J0 = a;
			J1 = b;
			evalsJac++;
			xCost = 0.5 * f * f;
			termination = 1;
			if (!isfinite(xCost))
			{
				termination = 2;
				return false;
			}
			g0 = J0 * f;
			g1 = J1 * f;
			sc0 = 1.0;
			sc1 = 1.0;
			if (o.jacobi_scaling)
			{
				sc0 = 1.0 / (1.0 + sqrt(J0 * J0));
				sc1 = 1.0 / (1.0 + sqrt(J1 * J1));
			}
			j0 = J0 * sc0;
			j1 = J1 * sc1;
			xNorm = sqrt(x0 * x0 + x1 * x1);
			gradMax = fmax(fabs(g0), fabs(g1));
			minimumCost = xCost;
			seMinimum = seCurrent = seReference = seCandidate = xCost;
			seAccRef = seAccCand = 0.0;
			seNumNonmono = 0;
			radius = o.initial_radius;
			decreaseFactor = 2.0;
			// iteration zero counts as successful: a start already within the gradient
			// tolerance converges immediately
			reuseDiagonal = false;
			lastSuccessful = true;
			d0 = d1 = 0.0;
			numInvalid = 0;
		}
		else if (phase == 1)
		{
			evalsCost++;
			candCost = 0.5 * r * r;
			if (!isfinite(candCost))
			{
				candCost = 1.7976931348623157e308;
			}
			const double e0 = x0 - c0, e1 = x1 - c1;
			const double stepNorm = sqrt(e0 * e0 + e1 * e1);
			if (stepNorm <= o.parameter_tolerance * (xNorm + o.parameter_tolerance))
			{
				termination = 0;
				return false;
			}
			const double costChange = xCost - candCost;
			if (fabs(costChange) <= o.function_tolerance * xCost)
			{
				termination = 0;
				return false;
			}
			const double relDec = (seCurrent - candCost) / modelCostChange;
			const double histDec = (seReference - candCost) / (seAccRef + modelCostChange);
			quality = fmax(relDec, histDec);
			if (quality > o.min_relative_decrease)
			{
				x0 = c0;
				x1 = c1;
				xNorm = sqrt(x0 * x0 + x1 * x1);
				q0 = x0;
				q1 = x1;
				qJac = true;
				phase = 2;
				return true;
			}
			radius = radius / decreaseFactor;
			decreaseFactor *= 2.0;
			reuseDiagonal = true;
		}
		else
		{
		// phase 2: the Jacobian at the accepted point
		f = r;
		J0 = a;
		J1 = b;
		evalsJac++;
		xCost = 0.5 * f * f;
		if (!isfinite(xCost))
		{
			termination = 2;
			return false;
		}
		g0 = J0 * f;
		g1 = J1 * f;
		j0 = J0 * sc0;
		j1 = J1 * sc1;
		gradMax = fmax(fabs(g0), fabs(g1));
		lastSuccessful = true;
		const double q = 2.0 * quality - 1.0;
		radius = radius / fmax(1.0 / 3.0, 1.0 - q * q * q);
		radius = fmin(o.max_radius, radius);
		decreaseFactor = 2.0;
		reuseDiagonal = false;
		seCurrent = candCost;
		seAccCand += modelCostChange;
		seAccRef += modelCostChange;
		if (seCurrent < seMinimum)
		{
			seMinimum = seCurrent;
			seNumNonmono = 0;
			seCandidate = seCurrent;
			seAccCand = 0.0;
		}
		else
		{
			++seNumNonmono;
			if (seCurrent > seCandidate)
			{
				seCandidate = seCurrent;
				seAccCand = 0.0;
			}
		}
		if (seNumNonmono == o.max_nonmono)
		{
			seReference = seCandidate;
			seAccRef = seAccCand;
		}
		}
		// ONE copy of the loop's top (three inlined copies cost the kernels that hold this solver ~70 VGPRs)
		return next_step(o);
	}
};

template <int FIXED>
__global__ void __launch_bounds__(512) k_solve_independent(const uint64_t* __restrict__ events,
									const Unit* __restrict__ units, int capDoubles,
									double* __restrict__ flowsOut, int32_t* __restrict__ stats,
									EvalConsts c, SolveConsts o, int noReuse)
{
	extern __shared__ double lds[];
	const Unit u = units[blockIdx.x];
	const uint64_t* ev = events + u.ev_off;
	int iteration = 0, evalsCost = 0, evalsJac = 0, termination = 0;
	double best0 = 0.0, best1 = 0.0;

	if (u.flags & kUnitActive)
	{
		// The solver state lives in LDS (the reduction scratch rows of waves 8..15, which a workgroup of
		// at most 512 lanes never uses) and thread 0 runs the solver between two evaluations: replicated
		// in every wave its serial f64 code (divisions, square roots) was a fifth of the workgroup's
		// vector instructions, replicated in every lane ~100 VGPRs live across the objective.
		static_assert(sizeof(LmUnit) <= 64 * sizeof(double), "LmUnit must fit red[64..127]");
		LmUnit& lm = *reinterpret_cast<LmUnit*>(lds + 64);
		// the record of the image in LDS (ebo_eval3.inc, EvalReuse): doubles 148..157 of the header, behind the 32
		// ints the bounding-box reduction uses (at most eight waves)
		EvalReuse* ru = (FIXED >= 3 && !noReuse) ? reinterpret_cast<EvalReuse*>(lds + kRedDoubles + 20) : nullptr;
		static_assert(sizeof(EvalReuse) <= (kLdsHeader - kRedDoubles - 20) * sizeof(double), "EvalReuse must fit the header");
		if (threadIdx.x == 0)
		{
			lm.begin();
			if (ru)
			{
				ru->valid = 0.0;
			}
		}
		for (;;)
		{
			__syncthreads();  // the point to evaluate (or the end) is published; every thread is past the previous evaluation
			if (!lm.more)
			{
				break;
			}
			const double q0 = lm.q0, q1 = lm.q1;
			const bool qJac = lm.qJac;
			double r, a, b;
			eval_unit<FIXED>(ev, u, q0, q1, qJac, capDoubles, c, lds, r, a, b, ru);
			if (threadIdx.x == 0)
			{
				lm.more = lm.advance(r, a, b, o) ? 1 : 0;
			}
		}
		if (threadIdx.x == 0)
		{
			iteration = lm.iteration;
			evalsCost = lm.evalsCost;
			evalsJac = lm.evalsJac;
			termination = lm.termination;
			best0 = lm.best0;
			best1 = lm.best1;
		}
	}
	if (threadIdx.x == 0 && !(u.flags & kUnitStray))
	{
		flowsOut[2 * u.flow_idx] = best0;
		flowsOut[2 * u.flow_idx + 1] = best1;
		if (stats)
		{
			stats[4 * u.flow_idx + 0] = iteration;
			stats[4 * u.flow_idx + 1] = evalsCost;
			stats[4 * u.flow_idx + 2] = evalsJac;
			stats[4 * u.flow_idx + 3] = termination;
		}
	}
}

// ---------------------------------------------------------------------------
// Integer event-count images (bit-exact by construction: int32 adds commute).
//   EBO_COUNT_INTEGRATED feature_detector.cpp:466-482
//   EBO_COUNT_WARPED     feature_detector.cpp:433-463  round() = half away from zero
//   EBO_COUNT_FIELD      feature_detector.cpp:270-295  float32 field (at<Vec2f>)
// Workgroup = unit (its events are one contiguous range and share one flow).
// ---------------------------------------------------------------------------
__global__ void k_count_scatter(const uint64_t* __restrict__ events, const Unit* __restrict__ units,
								int unitsPerWindow, int mode, const void* __restrict__ aux,
								int32_t* __restrict__ counts, EvalConsts c)
{
	const Unit u = units[blockIdx.x];
	const int w = blockIdx.x / unitsPerWindow;
	const int P = c.npx * c.npy;
	const size_t imgSize = static_cast<size_t>(c.image_w) * c.image_h;
	int32_t* img = counts + static_cast<size_t>(w) * imgSize;
	const uint64_t* ev = events + u.ev_off;
	const bool stray = (u.flags & kUnitStray) != 0;
	double m0 = 0.0, m1 = 0.0;
	if (mode == 1 && !stray)
	{
		const double* flows = static_cast<const double*>(aux);
		m0 = flows[2 * u.flow_idx];
		m1 = flows[2 * u.flow_idx + 1];
	}
	for (uint32_t e = threadIdx.x; e < u.n_ev; e += blockDim.x)
	{
		int x, y, pos, dt;
		unpack(ev[e], x, y, pos, dt);
		int nx = x, ny = y;
		if (mode != 0)
		{
			if (mode == 1 && stray)
			{
				// :436-441 with the index clamped at 0 (negative indices are undefined there)
				const int px = max(min(x / c.patch_w, c.npx - 1), 0);
				const int py = max(min(y / c.patch_h, c.npy - 1), 0);
				const double* flows = static_cast<const double*>(aux);
				m0 = flows[2 * (static_cast<size_t>(w) * P + py * c.npx + px)];
				m1 = flows[2 * (static_cast<size_t>(w) * P + py * c.npx + px) + 1];
			}
			if (mode == 2)
			{
				if (x < 0 || x >= c.image_w || y < 0 || y >= c.image_h)
				{
					continue;
				}
				const float* field = static_cast<const float*>(aux) +
									 2 * (static_cast<size_t>(w) * imgSize +
										  static_cast<size_t>(y) * c.image_w + x);
				m0 = static_cast<double>(field[0]);
				m1 = static_cast<double>(field[1]);
			}
			const double dtw = static_cast<double>(dt + u.dt_win);
			const double fx = static_cast<double>(x) + dtw * c.scale * m0;
			const double fy = static_cast<double>(y) + dtw * c.scale * m1;
			if (!convertible(fx) || !convertible(fy))
			{
				continue;
			}
			nx = static_cast<int>(round(fx));
			ny = static_cast<int>(round(fy));
		}
		if (nx >= 0 && nx < c.image_w && ny >= 0 && ny < c.image_h)
		{
			atomicAdd(&img[static_cast<size_t>(ny) * c.image_w + nx], 1);
		}
	}
}

constexpr float kSureBase = 0.499999f;  // 0.5 - 1e-6: slack for the f64 roundings of the reference's own expression

// Destination pixel of one event for the count images; MODE = EBO_COUNT_*.  Returns false
// when the event contributes nothing (outside the image, or a coordinate the reference's
// int conversion does not define).  Straight-line code: every lane of a wave runs it.
//   MODE 1: newP = round(p + (t_ref - t) * scale * mf[patch])      feature_detector.cpp:443-451
//   MODE 2: same with the float32 field at the event's own pixel   :276-285
template <int MODE>
__device__ __forceinline__ bool count_target(uint64_t rec, bool live, int dtWin, double m0, double m1,
											 const float* __restrict__ field /* window's field */,
											 const EvalConsts& c, int& nx, int& ny)
{
	int x, y, pos, dt;
	unpack(rec, x, y, pos, dt);
	nx = x;
	ny = y;
	if (MODE != 0)
	{
		if (MODE == 2)
		{
			live = live && x >= 0 && x < c.image_w && y >= 0 && y < c.image_h;
			const size_t at = live ? 2 * (static_cast<size_t>(y) * c.image_w + x) : 0;
			const float2 f = *reinterpret_cast<const float2*>(field + at);
			m0 = static_cast<double>(f.x);
			m1 = static_cast<double>(f.y);
		}
		// Float first.  The reference's position is fl64(x + fl64(fl64(dtw * scale) * m)); the same
		// expression in float differs from it by at most 4e-7 |displacement| + 6e-8 |position| (five
		// roundings of 2^-24 on the product, one on the sum, inputs rounded to float).  If the float
		// value is farther than that (+ 1e-6) from every half-integer, both round to the same pixel,
		// and the double arithmetic (half-rate, conversions, truncations and compares) is not needed.
		// Anything else -- a value near a rounding boundary (0.04 % of the events at sensor
		// coordinates: a wave of 64 takes the exact branch 2-3 % of the time; with the position term
		// bounded by its worst case 2^15 instead, 0.8 % and 40 %), huge, infinite or NaN (comparisons
		// false) -- takes the exact path below.
		const float prod = static_cast<float>(dt + dtWin) * static_cast<float>(c.scale);
		const float px = prod * static_cast<float>(m0), py = prod * static_cast<float>(m1);
		const float vx = static_cast<float>(x) + px, vy = static_cast<float>(y) + py;
		const float rx = rintf(vx), ry = rintf(vy);
		const bool sure = fabsf(vx - rx) < kSureBase - 4e-7f * fabsf(px) - 6e-8f * fabsf(vx) &&
						  fabsf(vy - ry) < kSureBase - 4e-7f * fabsf(py) - 6e-8f * fabsf(vy);
		if (sure)
		{
			nx = static_cast<int>(rx);
			ny = static_cast<int>(ry);
		}
		else
		{
			const double dtw = static_cast<double>(dt + dtWin);
			const double fx = static_cast<double>(x) + dtw * c.scale * m0;
			const double fy = static_cast<double>(y) + dtw * c.scale * m1;
			live = live && convertible(fx) && convertible(fy);
			nx = static_cast<int>(round(live ? fx : 0.0));
			ny = static_cast<int>(round(live ? fy : 0.0));
		}
	}
	return live && nx >= 0 && nx < c.image_w && ny >= 0 && ny < c.image_h;
}

// The unit of an event from its own coordinates, without walking the unit table: the grid patch
// that contains it (feature_detector.cpp:332-355: index min(x / pw, npx - 1)), or the stray unit P
// for an event outside the sensor, whose flow is that of the clamped patch (:436-441).  Division
// by the (uniform) patch size is one mul_hi: exact for 0 <= x < 2^15 (coordinates are 15-bit).
__device__ __forceinline__ void event_unit(uint64_t rec, const EvalConsts& c, int& patch, int& unit)
{
	int x, y, pos, dt;
	unpack(rec, x, y, pos, dt);
	const bool inSensor = static_cast<unsigned>(x) < static_cast<unsigned>(c.image_w) &&
						  static_cast<unsigned>(y) < static_cast<unsigned>(c.image_h);
	const unsigned xc = static_cast<unsigned>(max(x, 0)), yc = static_cast<unsigned>(max(y, 0));
	const int bx = min(static_cast<int>(c.patch_w == 1 ? xc : __umulhi(xc, c.inv_pw)), c.npx - 1);
	const int by = min(static_cast<int>(c.patch_h == 1 ? yc : __umulhi(yc, c.inv_ph)), c.npy - 1);
	patch = by * c.npx + bx;
	unit = inSensor ? patch : c.npx * c.npy;
}

// The flow of the patch a stray event (outside the sensor) is attributed to in the final
// loop (:436-441, index clamped at 0: negative indices are undefined there).
__device__ __forceinline__ void stray_flow(uint64_t rec, const double* __restrict__ windowFlows,
										   const EvalConsts& c, double& m0, double& m1)
{
	int x, y, pos, dt;
	unpack(rec, x, y, pos, dt);
	const int px = max(min(x / c.patch_w, c.npx - 1), 0);
	const int py = max(min(y / c.patch_h, c.npy - 1), 0);
	m0 = windowFlows[2 * (py * c.npx + px)];
	m1 = windowFlows[2 * (py * c.npx + px) + 1];
}

// LDS-privatised count image: workgroup = (row band, window).  The band's counters
// live in LDS (16-bit counters packed two per dword when the window has < 65536
// events, else 32-bit); the workgroup streams ALL events of its window once
// (8 B/lane coalesced; with more than one band the re-reads are L2 hits), counts
// those that land in its band with ds_add_u32, and writes the finished f64 rows
// with plain coalesced stores.  HBM traffic = events once + image once: no global
// atomics, no int32 intermediate image.  Bit-exact (integer adds commute).
template <bool U16, int MODE, bool MULTI>
__global__ void __launch_bounds__(1024) k_count_window_lds(
	const uint64_t* __restrict__ events, const Unit* __restrict__ units, int unitsPerWindow,
	const void* __restrict__ aux, int rowsPerBand, int nWindows, double* __restrict__ image,
	EvalConsts c)
{
	extern __shared__ unsigned int cnt[];
	// 1-D grid; consecutive workgroup ids go round-robin over the 8 XCDs, so the bands of one
	// window take consecutive slots of ONE XCD: the second band's event reads hit that L2.
	const int nBands = (c.image_h + rowsPerBand - 1) / rowsPerBand;
	const int slot = blockIdx.x >> 3;
	const int w = (slot / nBands) * 8 + (blockIdx.x & 7);
	if (w >= nWindows)
	{
		return;
	}
	const int row0 = (slot % nBands) * rowsPerBand;
	const int rows = min(rowsPerBand, c.image_h - row0);
	const int W = c.image_w;
	const int npx = rows * W;
	const int nWords = U16 ? (npx + 1) >> 1 : npx;
	for (int i = threadIdx.x; i < nWords; i += blockDim.x)
	{
		cnt[i] = 0u;
	}
	const Unit* wu = units + static_cast<size_t>(w) * unitsPerWindow;
	const int P = c.npx * c.npy;
	const size_t imgSize = static_cast<size_t>(W) * c.image_h;
	const double* windowFlows = static_cast<const double*>(aux) + (MODE == 1 ? 2 * static_cast<size_t>(w) * P : 0);
	const float* windowField = static_cast<const float*>(aux) + (MODE == 2 ? 2 * static_cast<size_t>(w) * imgSize : 0);
	// The window's unit table -- reference-time offset per unit, flow per patch -- behind the
	// counters in LDS: every event looks its unit up (event_unit), and three gathers per event
	// through the vector memory path cost more than the warp arithmetic (an LDS read of an
	// address shared by most of a wave is a broadcast).
	const int tblBase = (rowsPerBand * W * (U16 ? 2 : 4) + 15) & ~15;  // bytes; the band size of the launch
	double* tblFlow = reinterpret_cast<double*>(reinterpret_cast<char*>(cnt) + tblBase);
	int* tblDt = reinterpret_cast<int*>(tblFlow + (MODE == 1 ? 2 * P : 0));
	if (MODE != 0)
	{
		for (int i = threadIdx.x; i <= P; i += blockDim.x)
		{
			tblDt[i] = wu[i].dt_win;
		}
		if (MODE == 1)
		{
			for (int i = threadIdx.x; i < 2 * P; i += blockDim.x)
			{
				tblFlow[i] = windowFlows[i];
			}
		}
	}
	__syncthreads();
	// kInFlight independent 8-byte loads per lane are issued before the first is
	// consumed: ~64 KiB in flight per CU, enough to cover HBM latency (Little's law).
	constexpr int kInFlight = 8;
	// Every event finds its unit (reference time, flow) from its own coordinates: no walk along
	// the unit table, no branch per event, and the 8 in-flight events of a lane are independent
	// (the first version tracked the current unit per lane: two dependent loads and a divergent
	// loop per event).  The window's units, the stray one included, are contiguous in `events`.
	{
		const uint32_t evBegin = wu[0].ev_off;
		const uint32_t evEnd = wu[P].ev_off + wu[P].n_ev;
		for (uint32_t eb = evBegin + threadIdx.x; eb < evEnd; eb += kInFlight * blockDim.x)
		{
			uint64_t recs[kInFlight];
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const uint32_t ek = eb + k * blockDim.x;
				recs[k] = (ek < evEnd) ? events[ek] : 0ull;
			}
			int dtWin[kInFlight];
			double m0[kInFlight], m1[kInFlight];
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				dtWin[k] = 0;
				m0[k] = 0.0;
				m1[k] = 0.0;
				if (MODE != 0)
				{
					int patch, unit;
					event_unit(recs[k], c, patch, unit);
					dtWin[k] = tblDt[unit];
					if (MODE == 1)
					{
						m0[k] = tblFlow[2 * patch];
						m1[k] = tblFlow[2 * patch + 1];
					}
				}
			}
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const bool live = eb + k * blockDim.x < evEnd;
				int nx = 0, ny = -1;
				bool hit = false;
				bool maybe = true;
				if (MODE == 1 && MULTI)
				{
					// Several bands (MULTI): every band workgroup sees every event of the window, but only the
					// events that land in its rows need the exact (f64, reference-order) warp.  A float
					// estimate of the destination row decides: its error is below 2e-7 of the
					// displacement plus 6e-8 of the row, so an estimate more than a row outside the band
					// cannot round into it (a displacement large enough to break that bound leaves the
					// image anyway); NaN compares false and takes the exact path.  Events of a wave
					// belong to one or two patches, so whole waves skip.
					int x, y, pos, dt;
					unpack(recs[k], x, y, pos, dt);
					const float fy = static_cast<float>(y) + static_cast<float>(dt + dtWin[k]) * static_cast<float>(c.scale) *
															   static_cast<float>(m1[k]);
					maybe = !(fy < static_cast<float>(row0) - 1.5f || fy > static_cast<float>(row0 + rows) + 0.5f);
				}
				if (maybe)
				{
					hit = count_target<MODE>(recs[k], live, dtWin[k], m0[k], m1[k], windowField, c, nx, ny);
				}
				const int ry = ny - row0;
				if (hit && ry >= 0 && ry < rows)
				{
					const int p = ry * W + nx;
					if (U16)
					{
						atomicAdd(&cnt[p >> 1], 1u << ((p & 1) * 16));
					}
					else
					{
						atomicAdd(&cnt[p], 1u);
					}
				}
			}
		}
	}
	__syncthreads();
	double* out = image + static_cast<size_t>(w) * imgSize + static_cast<size_t>(row0) * W;
	if (U16 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
	{
		// one packed dword = two pixels = one 16-byte store
		const int pairs = npx >> 1;
		double2* out2 = reinterpret_cast<double2*>(out);
		for (int i = threadIdx.x; i < pairs; i += blockDim.x)
		{
			const unsigned int v = cnt[i];
			out2[i] = make_double2(static_cast<double>(v & 0xFFFFu), static_cast<double>(v >> 16));
		}
		if ((npx & 1) && threadIdx.x == 0)
		{
			out[npx - 1] = static_cast<double>(cnt[pairs] & 0xFFFFu);
		}
	}
	else
	{
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			const unsigned int v = U16 ? ((cnt[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu) : cnt[p];
			out[p] = static_cast<double>(v);
		}
	}
}

// Unit waves (impl 4): workgroup = (row band, window) with the band's counters in LDS, as impl 1;
// but the events are taken UNIT BY UNIT, one wave per unit at a time (a work counter in LDS hands
// the units out).  A unit's reference-time offset and flow are then wave-uniform scalars -- no
// per-event patch lookup, no table reads -- and a band only takes the units whose events can
// reach it: a unit's rows grown by its largest possible displacement,
// max|t_ref - t| (kept per unit by the bucketing) x scale x |flow_y|, rounded up.  With several
// bands a window's events are then warped ~1.2-2 times instead of once per band.
template <bool U16, int MODE>
__global__ void __launch_bounds__(1024) k_count_units(
	const uint64_t* __restrict__ events, const Unit* __restrict__ units, const int32_t* __restrict__ unitMaxDt,
	int unitsPerWindow, const void* __restrict__ aux, int rowsPerBand, int nWindows, double* __restrict__ image,
	EvalConsts c)
{
	extern __shared__ unsigned int cnt[];
	const int nBands = (c.image_h + rowsPerBand - 1) / rowsPerBand;
	const int slot = blockIdx.x >> 3;
	const int w = (slot / nBands) * 8 + (blockIdx.x & 7);
	if (w >= nWindows)
	{
		return;
	}
	const int row0 = (slot % nBands) * rowsPerBand;
	const int rows = min(rowsPerBand, c.image_h - row0);
	const int W = c.image_w;
	const int npx = rows * W;
	const int nWords = U16 ? (npx + 1) >> 1 : npx;
	const int P = c.npx * c.npy;
	const int hdr = (rowsPerBand * W * (U16 ? 2 : 4) + 15) & ~15;  // bytes; the band size of the launch
	int* ctl = reinterpret_cast<int*>(reinterpret_cast<char*>(cnt) + hdr);  // [0] next, [1] nSel, [2..] list
	int* list = ctl + 2;
	for (int i = threadIdx.x; i < nWords; i += blockDim.x)
	{
		cnt[i] = 0u;
	}
	if (threadIdx.x < 2)
	{
		ctl[threadIdx.x] = 0;
	}
	__syncthreads();
	const Unit* wu = units + static_cast<size_t>(w) * unitsPerWindow;
	const int32_t* wmax = unitMaxDt + static_cast<size_t>(w) * unitsPerWindow;
	const size_t imgSize = static_cast<size_t>(W) * c.image_h;
	const double* windowFlows = static_cast<const double*>(aux) + (MODE == 1 ? 2 * static_cast<size_t>(w) * P : 0);
	const float* windowField = static_cast<const float*>(aux) + (MODE == 2 ? 2 * static_cast<size_t>(w) * imgSize : 0);
	// which units can reach this band
	for (int u = threadIdx.x; u <= P; u += blockDim.x)
	{
		const Unit un = wu[u];
		bool take = un.n_ev > 0;
		if (take && u < P && nBands > 1 && MODE != 2)
		{
			double reach = 0.0;
			if (MODE == 1)
			{
				// |fl(fl(dtw * scale) * m1)| <= fl(fl(maxdt * |scale|) * |m1|): rounding is monotonic
				reach = static_cast<double>(wmax[u]) * fabs(c.scale) * fabs(windowFlows[2 * u + 1]) + 1.0;
			}
			const double lo = static_cast<double>(un.ry) - reach, hi = static_cast<double>(un.ry + un.rh - 1) + reach;
			// NaN / inf reach: comparisons false -> taken
			take = !(hi < static_cast<double>(row0) - 0.5 || lo > static_cast<double>(row0 + rows) - 0.5);
		}
		if (take)
		{
			list[atomicAdd(&ctl[1], 1)] = u;
		}
	}
	__syncthreads();
	const int nSel = ctl[1];
	const int lane = threadIdx.x & 63;
	constexpr int kInFlight = 4;
	for (;;)
	{
		int pick = 0;
		if (lane == 0)
		{
			pick = atomicAdd(&ctl[0], 1);
		}
		pick = __shfl(pick, 0, 64);
		if (pick >= nSel)
		{
			break;
		}
		const int u = list[pick];
		const Unit un = wu[u];
		const bool stray = u == P;
		double m0 = 0.0, m1 = 0.0;
		if (MODE == 1 && !stray)
		{
			m0 = windowFlows[2 * u];
			m1 = windowFlows[2 * u + 1];
		}
		const int dtWin = un.dt_win;
		const uint32_t evEnd = un.ev_off + un.n_ev;
		for (uint32_t eb = un.ev_off + lane; eb < evEnd; eb += kInFlight * 64)
		{
			uint64_t recs[kInFlight];
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const uint32_t ek = eb + k * 64;
				recs[k] = (ek < evEnd) ? events[ek] : 0ull;
			}
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const bool live = eb + k * 64 < evEnd;
				if (MODE == 1 && stray)
				{
					stray_flow(recs[k], windowFlows, c, m0, m1);
				}
				int nx, ny;
				const bool hit = count_target<MODE>(recs[k], live, dtWin, m0, m1, windowField, c, nx, ny);
				const int ry = ny - row0;
				if (hit && ry >= 0 && ry < rows)
				{
					const int p = ry * W + nx;
					if (U16)
					{
						atomicAdd(&cnt[p >> 1], 1u << ((p & 1) * 16));
					}
					else
					{
						atomicAdd(&cnt[p], 1u);
					}
				}
			}
		}
	}
	__syncthreads();
	double* out = image + static_cast<size_t>(w) * imgSize + static_cast<size_t>(row0) * W;
	if (U16 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
	{
		const int pairs = npx >> 1;
		double2* out2 = reinterpret_cast<double2*>(out);
		for (int i = threadIdx.x; i < pairs; i += blockDim.x)
		{
			const unsigned int v = cnt[i];
			out2[i] = make_double2(static_cast<double>(v & 0xFFFFu), static_cast<double>(v >> 16));
		}
		if ((npx & 1) && threadIdx.x == 0)
		{
			out[npx - 1] = static_cast<double>(cnt[pairs] & 0xFFFFu);
		}
	}
	else
	{
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			const unsigned int v = U16 ? ((cnt[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu) : cnt[p];
			out[p] = static_cast<double>(v);
		}
	}
}

// One event of a unit whose flow is wave-uniform, added to the tile [x0, x0 + tw) x [row0, row0 + th)
// of the warped count image (feature_detector.cpp:443-455): straight-line code, ONE divergent
// branch (the exact path) and one predicated LDS add.  The first versions of the count kernels
// spent more scalar instructions on exec-mask bookkeeping (nested && tests, per-event patch
// lookups) than vector instructions on the events: ~85 SALU per 64 events against the one scalar
// unit a CU has.  Float first: the float position differs from the reference's f64 one,
// fl64(x + fl64(fl64(dtw * scale) * m)), by at most 4e-7 |displacement| + 6e-8 |x|; farther than
// that from every half-integer both round to the same pixel; the rest (~0.4 % of the events, and
// anything huge or NaN: comparisons false) takes the f64 expression in the reference's order.
template <bool U16>
__device__ __forceinline__ void count_hit_uniform(uint64_t rec, bool live, int dtWin, double m0, double m1, float m0f,
												   float m1f, float thrX, float thrY, float scalef, double scale, int x0,
												   int row0, int tw, int th, unsigned int* cnt)
{
	int x, y, pos, dt;
	unpack(rec, x, y, pos, dt);
	const int dtw = dt + dtWin;
	const float prod = static_cast<float>(dtw) * scalef;
	const float vx = static_cast<float>(x) + prod * m0f, vy = static_cast<float>(y) + prod * m1f;
	const float rx = rintf(vx), ry = rintf(vy);
	// thrX / thrY: the unit's tolerance (count_unit_tolerance), wave-uniform -- the per-event form
	// kSureBase - 4e-7 |px| - 6e-8 |vx| cost eight vector instructions of the ~30 an event takes
	// (bitwise, not &&: no short-circuit branches)
	const bool sure = (static_cast<int>(fabsf(vx - rx) < thrX) & static_cast<int>(fabsf(vy - ry) < thrY)) != 0;
	int nx = static_cast<int>(rx), ny = static_cast<int>(ry);
	if (!sure)
	{
		const double d = static_cast<double>(dtw);
		const double fx = static_cast<double>(x) + d * scale * m0;
		const double fy = static_cast<double>(y) + d * scale * m1;
		live = (static_cast<int>(live) & static_cast<int>(convertible(fx)) & static_cast<int>(convertible(fy))) != 0;
		nx = static_cast<int>(round(live ? fx : 0.0));
		ny = static_cast<int>(round(live ? fy : 0.0));
	}
	const int cx = nx - x0, cy = ny - row0;
	if (static_cast<int>(live) & static_cast<int>(static_cast<unsigned>(cx) < static_cast<unsigned>(tw)) &
		static_cast<int>(static_cast<unsigned>(cy) < static_cast<unsigned>(th)))
	{
		// cy < th, tw < 2^15: the 24-bit multiply-add (full rate; v_mul_lo_u32 is a quarter-rate instruction)
		const unsigned int p = __umul24(static_cast<unsigned int>(cy), static_cast<unsigned int>(tw)) + static_cast<unsigned int>(cx);
		if (U16)
		{
			atomicAdd(&cnt[p >> 1], 1u << ((p & 1u) * 16u));
		}
		else
		{
			atomicAdd(&cnt[p], 1u);
		}
	}
}

// The float pre-test's tolerance for every event of a unit, one number per axis: the float
// position differs from the reference's f64 one by at most 4e-7 |displacement| + 6e-8 |position|
// (count_hit_uniform), |displacement| <= max|t_ref - t| |scale| |flow| (rounding is monotonic; 1e-6
// relative covers the float conversions of scale and flow and the float products), |position| <= the
// sensor's extent + the displacement.  NaN / inf flow: the tolerance is NaN or -inf, no event is
// "sure", all take the exact path.
__device__ __forceinline__ float count_unit_tolerance(int maxDt, double scale, double m, int extent)
{
	const float reach = static_cast<float>(static_cast<double>(maxDt) * fabs(scale) * fabs(m)) * 1.000001f;
	return kSureBase - 4e-7f * reach - 6e-8f * (static_cast<float>(extent) + reach + 1.0f);
}

// Unit waves over 2-D tiles (impl 5; the warped image of R2's final loop): workgroup = (tile,
// window) with the tile's counters in LDS; the events are taken unit by unit, one wave per unit at
// a time (an LDS work counter hands the units out), so a unit's reference-time offset and flow
// are wave-uniform scalars; a tile only takes the units whose events can reach it: the unit's rect
// grown by max|t_ref - t| x |scale| x |flow| (+1) on each axis.  Full-width row bands (impl 4) make
// a large sensor's units visit 5-6 bands each (C4: 15-row bands against a reach of +-25 rows);
// tiles a few hundred pixels on a side cut that to ~1.7 visits.  Tiles of a window take
// consecutive slots of one XCD (blockIdx % 8) so that re-reads of a unit's events hit that L2.
// One pass of a tile workgroup: rows [row0, row0 + th) x columns [x0, x0 + tw) counted in LDS from the
// selected units (headers in hdr[0, nSel)), then stored.  U16: two 16-bit counters per dword.
template <bool U16>
__device__ __forceinline__ void tile_pass(const uint64_t* __restrict__ events, const double* __restrict__ windowFlows,
										   const int32_t* __restrict__ wmax, const int4* hdr, int nSel, int* ctl, unsigned int* cnt, int x0, int row0, int tw, int th,
										   double* __restrict__ out /* image(row0, x0) */, int W, bool alignedImage,
										   const EvalConsts& c, bool preZeroed EDGE_TICK_ARG)
{
	const int npx = tw * th;
	const int nWords = U16 ? (npx + 1) >> 1 : npx;
	if (!preZeroed)  // (the caller zeroed the counters and ctl[0] under the latency of its unit-table loads)
	{
		for (int i = threadIdx.x; i < nWords; i += blockDim.x)
		{
			cnt[i] = 0u;
		}
		if (threadIdx.x == 0)
		{
			ctl[0] = 0;
		}
		__syncthreads();
	}
	EDGE_TICK(17);
	const int lane = threadIdx.x & 63;
	const float scalef = static_cast<float>(c.scale);
	// six 8-byte loads per lane and batch (round 4, end: 4 -> 6 is +1 point of HBM fraction at every size, 7 is mixed, 8 takes
	// the kernel over 64 VGPRs = one workgroup per CU; 16-byte loads are 3-5 points SLOWER: profiles/r04_count_*_ab.txt)
	constexpr int kInFlight = 6;
	// Software pipeline: the loads of the NEXT batch of 384 events (the same unit's, or the first
	// of the next unit the wave picks) are in flight while the current batch is counted.
	// what a wave needs of a unit, fetched when the unit is PICKED (one batch ahead of its use):
	// header, flow (f64 and float) and the float pre-test's tolerances -- all wave-uniform
	struct Picked
	{
		int4 h;
		double m0, m1;
		float m0f, m1f, thrX, thrY;
	};
	auto pick_unit = [&](Picked& q) -> bool {
		int pick = 0;
		if (lane == 0)
		{
			pick = atomicAdd(&ctl[0], 1);
		}
		pick = __builtin_amdgcn_readfirstlane(pick);
		if (pick >= nSel)
		{
			return false;
		}
		q.h = hdr[pick];
		const int u = q.h.w;
		q.m0 = windowFlows[2 * u];  // 16 KB per window: L1 / L2
		q.m1 = windowFlows[2 * u + 1];
		q.m0f = static_cast<float>(q.m0);
		q.m1f = static_cast<float>(q.m1);
		const int maxDt = wmax[u];
		q.thrX = count_unit_tolerance(maxDt, c.scale, q.m0, c.image_w);
		q.thrY = count_unit_tolerance(maxDt, c.scale, q.m1, c.image_h);
		return true;
	};
	Picked A;
	A.h = make_int4(0, 0, 0, 0);
	uint32_t posA = 0;
	bool haveA = pick_unit(A);
	if (haveA)
	{
		posA = static_cast<uint32_t>(A.h.x);
	}
	uint64_t recsA[kInFlight];
	if (haveA)
	{
#pragma unroll
		for (int k = 0; k < kInFlight; ++k)
		{
			const uint32_t ek = posA + lane + k * 64;
			recsA[k] = (ek < static_cast<uint32_t>(A.h.y)) ? events[ek] : 0ull;
		}
	}
	while (haveA)
	{
		// the batch after this one
		Picked B = A;
		uint32_t posB = posA + kInFlight * 64;
		bool haveB = true;
		if (posB >= static_cast<uint32_t>(A.h.y))
		{
			haveB = pick_unit(B);
			if (haveB)
			{
				posB = static_cast<uint32_t>(B.h.x);
			}
		}
		uint64_t recsB[kInFlight];
		if (haveB)
		{
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const uint32_t ek = posB + lane + k * 64;
				recsB[k] = (ek < static_cast<uint32_t>(B.h.y)) ? events[ek] : 0ull;
			}
		}
		// count batch A
#pragma unroll
		for (int k = 0; k < kInFlight; ++k)
		{
			count_hit_uniform<U16>(recsA[k], posA + lane + k * 64 < static_cast<uint32_t>(A.h.y), A.h.z, A.m0, A.m1, A.m0f,
								   A.m1f, A.thrX, A.thrY, scalef, c.scale, x0, row0, tw, th, cnt);
		}
		haveA = haveB;
		A = B;
		posA = posB;
#pragma unroll
		for (int k = 0; k < kInFlight; ++k)
		{
			recsA[k] = recsB[k];
		}
	}
	EDGE_TICK(18);
	__syncthreads();
	EDGE_TICK(19);
	if (U16 && !(tw & 1) && !(W & 1) && !(x0 & 1) && alignedImage)
	{
		// one packed dword = two pixels of one row = one 16-byte store
		const int pairsPerRow = tw >> 1;
		for (int i = threadIdx.x; i < (npx >> 1); i += blockDim.x)
		{
			const int r = i / pairsPerRow, q = i - r * pairsPerRow;
			const unsigned int v = cnt[i];
			*reinterpret_cast<double2*>(out + static_cast<size_t>(r) * W + 2 * q) =
				make_double2(static_cast<double>(v & 0xFFFFu), static_cast<double>(v >> 16));
		}
	}
	else
	{
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			const int r = p / tw, q = p - r * tw;
			const unsigned int v = U16 ? ((cnt[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu) : cnt[p];
			out[static_cast<size_t>(r) * W + q] = static_cast<double>(v);
		}
	}
	EDGE_TICK(20);
	__syncthreads();
	EDGE_TICK(21);
}

__global__ void __launch_bounds__(1024) k_count_tiles(
	const uint64_t* __restrict__ events, const Unit* __restrict__ units, const int32_t* __restrict__ unitMaxDt,
	int unitsPerWindow, const double* __restrict__ flows, int tileW, int tileH, int tilesX, int tilesY, int cntBytes,
	int nWindows, double* __restrict__ image, EvalConsts c)
{
	// tileW x tileH: the pitch of the tile grid (the last tile of a row / column takes what is left
	// of the image); cntBytes: 2 bytes per pixel of a full tile
	extern __shared__ unsigned int cnt[];
	const int nTiles = tilesX * tilesY;
	const int slot = blockIdx.x >> 3;
	const int w = (slot / nTiles) * 8 + (blockIdx.x & 7);
	if (w >= nWindows)
	{
		return;
	}
	const int tile = slot % nTiles;
	const int tix = tile % tilesX, tiy = tile / tilesX;
	const int x0 = tix * tileW, row0 = tiy * tileH;
	const int tw = (tix == tilesX - 1) ? c.image_w - x0 : tileW, th = (tiy == tilesY - 1) ? c.image_h - row0 : tileH;
	const int W = c.image_w;
	const int P = c.npx * c.npy;
	// behind the counters: [0] next, [1] nSel, [2] most events of a selected unit, [3] largest reach
	// (ceil, both axes), then one 4-int header per selected unit {first event, end, dt_win, unit}: the
	// waves walk the units from LDS, not through dependent global loads
	EDGE_TICK_DECL;
	int* ctl = reinterpret_cast<int*>(reinterpret_cast<char*>(cnt) + cntBytes);
	int4* hdr = reinterpret_cast<int4*>(ctl + 4);
	const Unit* wu = units + static_cast<size_t>(w) * unitsPerWindow;
	const int32_t* wmax = unitMaxDt + static_cast<size_t>(w) * unitsPerWindow;
	const size_t imgSize = static_cast<size_t>(W) * c.image_h;
	const double* windowFlows = flows + 2 * static_cast<size_t>(w) * P;
	// which units can reach this tile (the stray unit is left to k_count_stray).  The first round of
	// unit-table loads is issued, the tile's counters are zeroed while they are in flight (a loaded
	// memory system answers in microseconds), then the units are sorted out.
	auto consider = [&](int u, const Unit& un, int maxDt, double f0, double f1) {
		bool take = un.n_ev > 0;
		// |fl(fl(dtw * scale) * m)| <= fl(fl(maxdt * |scale|) * |m|): rounding is monotonic
		const double t = static_cast<double>(maxDt) * fabs(c.scale);
		const double reachX = t * fabs(f0) + 1.0, reachY = t * fabs(f1) + 1.0;
		if (take && nTiles > 1)
		{
			const double lox = static_cast<double>(un.rx) - reachX, hix = static_cast<double>(un.rx + un.rw - 1) + reachX;
			const double loy = static_cast<double>(un.ry) - reachY, hiy = static_cast<double>(un.ry + un.rh - 1) + reachY;
			// NaN / inf reach: comparisons false -> taken
			take = !(hix < static_cast<double>(x0) - 0.5 || lox > static_cast<double>(x0 + tw) - 0.5 ||
					 hiy < static_cast<double>(row0) - 0.5 || loy > static_cast<double>(row0 + th) - 0.5);
		}
		if (take)
		{
			hdr[atomicAdd(&ctl[1], 1)] = make_int4(static_cast<int>(un.ev_off), static_cast<int>(un.ev_off + un.n_ev), un.dt_win, u);
			atomicMax(&ctl[2], static_cast<int>(min(un.n_ev, 0x7fffffffu)));
			const double rr = fmax(reachX, reachY);
			atomicMax(&ctl[3], (rr < 1e6) ? static_cast<int>(ceil(rr)) : 1000000);  // NaN -> 1000000
		}
	};
	{
		const int u0 = threadIdx.x;
		const bool have0 = u0 < P;
		Unit un0 = {};
		int maxDt0 = 0;
		double f00 = 0.0, f01 = 0.0;
		if (have0)
		{
			un0 = wu[u0];
			maxDt0 = wmax[u0];
			f00 = windowFlows[2 * u0];
			f01 = windowFlows[2 * u0 + 1];
		}
		const int nWords16 = (tw * th + 1) >> 1;
		for (int i = threadIdx.x; i < nWords16; i += blockDim.x)
		{
			cnt[i] = 0u;
		}
		if (threadIdx.x < 4)
		{
			ctl[threadIdx.x] = 0;
		}
		__syncthreads();
		if (have0)
		{
			consider(u0, un0, maxDt0, f00, f01);
		}
		for (int u = threadIdx.x + blockDim.x; u < P; u += blockDim.x)
		{
			consider(u, wu[u], wmax[u], windowFlows[2 * u], windowFlows[2 * u + 1]);
		}
	}
	__syncthreads();
	EDGE_TICK(16);
	const int nSel = ctl[1];
	EDGE_COUNT(22, nSel);
	EDGE_COUNT(23, 1);
	// 16-bit counters are safe when no pixel can collect 65536 events: a pixel is within reach of at
	// most (1 + 2 ceil(R / pw)) (1 + 2 ceil(R / ph)) patches (the grid's last patches are larger: fewer),
	// each with at most ctl[2] events.  Otherwise (wild flows, one patch holding most of a window)
	// the tile is counted in slices of half its rows with 32-bit counters: any input is handled, that one slowly.
	const long nx = 1 + 2 * ((ctl[3] + c.patch_w - 1) / c.patch_w), ny = 1 + 2 * ((ctl[3] + c.patch_h - 1) / c.patch_h);
	const bool safe16 = nx * ny * static_cast<long>(ctl[2]) < 65536;
	double* out = image + static_cast<size_t>(w) * imgSize + static_cast<size_t>(row0) * W + x0;
	const bool alignedImage = (reinterpret_cast<uintptr_t>(image) & 15) == 0;
	if (safe16)
	{
		tile_pass<true>(events, windowFlows, wmax, hdr, nSel, ctl, cnt, x0, row0, tw, th, out, W, alignedImage, c, true EDGE_TICK_PASS);
	}
	else
	{
		// the LDS holds 2 bytes per pixel of a full tile: 32-bit counters for tileH / 2 rows at a time
		const int hMax = max(tileH / 2, 1);
		for (int r = 0; r < th; r += hMax)
		{
			tile_pass<false>(events, windowFlows, wmax, hdr, nSel, ctl, cnt, x0, row0 + r, tw, min(hMax, th - r),
							 out + static_cast<size_t>(r) * W, W, alignedImage, c, false EDGE_TICK_PASS);
		}
	}
	EDGE_TICK_FLUSH;
}

// The stray unit of every window (events outside the sensor; none in a real recording) for the
// kernels that leave it out: warped by the flow of the clamped patch (:436-441), added with f64
// atomics after the image has been stored.
__global__ void k_count_stray(const uint64_t* __restrict__ events, const Unit* __restrict__ units, int unitsPerWindow,
							  const double* __restrict__ flows, double* __restrict__ image, EvalConsts c)
{
	const int w = blockIdx.x;
	const int P = c.npx * c.npy;
	const Unit un = units[static_cast<size_t>(w) * unitsPerWindow + P];
	const double* windowFlows = flows + 2 * static_cast<size_t>(w) * P;
	double* img = image + static_cast<size_t>(w) * c.image_w * c.image_h;
	for (uint32_t e = threadIdx.x; e < un.n_ev; e += blockDim.x)
	{
		const uint64_t rec = events[un.ev_off + e];
		double m0, m1;
		stray_flow(rec, windowFlows, c, m0, m1);
		int nx, ny;
		if (count_target<1>(rec, true, un.dt_win, m0, m1, nullptr, c, nx, ny))
		{
			unsafeAtomicAdd(&img[static_cast<size_t>(ny) * c.image_w + nx], 1.0);  // exact on integer counts
		}
	}
}

// LDS rates in the evaluation kernels' own access shape, measured in the run that quotes them (bench.py's roofline.lds;
// the same loop as tools/microbench/lds_atomics.hip, "7x7 taps ..., any base"): per "event" a pseudo-random base slot per
// lane, then the 49 taps of a 7 x 7 footprint at immediate offsets (row pitch 41 slots) as 64-bit LDS atomic adds (ATOMIC)
// or single 64-bit LDS reads -- what k_eval3's scatter and gather issue.  `iters` operations per lane (a multiple of 49),
// 4 workgroups of 256 lanes per CU.  (Rounds 1-3 probed ONE random operation per loop trip with the generator in between,
// a latency-bound figure 15-25 % below these.)
template <bool ATOMIC>
__global__ void __launch_bounds__(256) k_lds_rate(double* __restrict__ sink, int iters)
{
	constexpr int kElems = 4096, kPitch = 41;
	__shared__ unsigned long long cell[kElems + 7 * kPitch];
	for (int i = threadIdx.x; i < kElems + 7 * kPitch; i += blockDim.x)
	{
		cell[i] = static_cast<unsigned long long>(i);
	}
	__syncthreads();
	unsigned rnd = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	unsigned long long acc = 0;
	for (int ev = 0; ev < iters / 49; ++ev)
	{
		rnd = rnd * 1664525u + 1013904223u;
		unsigned long long* p = cell + ((rnd >> 10) & (kElems - 1));
#pragma unroll
		for (int j = 0; j < 7; ++j)
		{
#pragma unroll
			for (int i = 0; i < 7; ++i)
			{
				if (ATOMIC)
				{
					atomicAdd(p + j * kPitch + i, 1ull);
				}
				else
				{
					acc += __hip_atomic_load(p + j * kPitch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // one ds_read_b64 (lds_ld)
				}
			}
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < kElems; i += blockDim.x)
	{
		acc += cell[i];
	}
	if (acc == 0x0123456789abcdefull)  // never: keeps the loads and the adds alive
	{
		sink[blockIdx.x] = 1.0;
	}
}

// Yardstick of the count-image kernels (bench.py, diagnostic): the same bytes with no work -- every packed
// event read once with 16-byte loads, every image pixel written once with 16-byte stores -- by 2048
// workgroups that each take a contiguous slice of both.  What this reaches on the chip is what "100 %"
// means for a kernel of that traffic (the loads are kept alive by an XOR the compiler must produce).
__global__ void __launch_bounds__(256) k_stream_yardstick(const uint4* __restrict__ events16, size_t nEv16,
														   double2* __restrict__ image16, size_t nPx16)
{
	// grid-stride over both streams at once, four 16-byte loads and four 16-byte stores in flight per lane:
	// reads and writes are interleaved at instruction level everywhere on the chip
	const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
	const size_t first = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	const size_t n = nEv16 > nPx16 ? nEv16 : nPx16;
	unsigned int x = 0;
	const double2 zero = make_double2(0.0, 0.0);
	for (size_t i = first; i < n; i += 4 * stride)
	{
		const size_t i1 = i + stride, i2 = i + 2 * stride, i3 = i + 3 * stride;
		const uint4 a = (i < nEv16) ? events16[i] : make_uint4(0, 0, 0, 0);
		const uint4 b = (i1 < nEv16) ? events16[i1] : make_uint4(0, 0, 0, 0);
		const uint4 c = (i2 < nEv16) ? events16[i2] : make_uint4(0, 0, 0, 0);
		const uint4 d = (i3 < nEv16) ? events16[i3] : make_uint4(0, 0, 0, 0);
		if (i < nPx16) image16[i] = zero;
		if (i1 < nPx16) image16[i1] = zero;
		if (i2 < nPx16) image16[i2] = zero;
		if (i3 < nPx16) image16[i3] = zero;
		x ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
	}
	asm volatile("" ::"v"(x));  // the loads are used
}

// The final image of warped events (feature_detector.cpp:433-463) of windows whose patches are SHARDED
// over ranks (BASELINE config 4): the context holds, per window, the units of this rank's
// patch rows (ebo_set_patches), `flows` are those of ALL patches of the grid (after the all-gather of
// the solved flows) and dtWin[unit] = t_ref(window) - t_ref(unit) with the WINDOW's reference time,
// which a shard cannot derive from its own events.  Workgroup = unit; every event takes the flow of
// the grid patch its own coordinates select (:436-441) and adds 1.0 at its rounded warped position.
// The partial images of the ranks are integer-valued doubles: their sum is exact in any order.
__global__ void k_count_shard(const uint64_t* __restrict__ events, const Unit* __restrict__ units, int unitsPerWindow,
							  const BandUnit* __restrict__ table, const double* __restrict__ flows,
							  double* __restrict__ image, EvalConsts c)
{
	const Unit un = units[blockIdx.x];
	const int w = blockIdx.x / unitsPerWindow;
	const int P = c.npx * c.npy;
	const double* windowFlows = flows + 2 * static_cast<size_t>(w) * P;
	double* img = image + static_cast<size_t>(w) * c.image_w * c.image_h;
	const int dtw = table[blockIdx.x].dt_win;
	for (uint32_t e = threadIdx.x; e < un.n_ev; e += blockDim.x)
	{
		const uint64_t rec = events[un.ev_off + e];
		double m0, m1;
		stray_flow(rec, windowFlows, c, m0, m1);
		int nx, ny;
		if (count_target<1>(rec, true, dtw, m0, m1, nullptr, c, nx, ny))
		{
			unsafeAtomicAdd(&img[static_cast<size_t>(ny) * c.image_w + nx], 1.0);  // exact on integer counts
		}
	}
}

// Patch-row bands (impl 2): workgroup = (band of whole patch rows, window).  Events are
// stored unit by unit in patch order, so the events that START in a band are one contiguous
// range: the workgroup streams only those (no re-reads by other bands, any image size),
// counts the ones whose warped position stays inside the band in LDS and appends the rare
// ones that leave the band (but stay inside the image) to an overflow list, which
// k_count_overflow adds with f64 atomics after every band has stored its rows.  The stray
// unit of the window (events outside the sensor) is streamed by band 0.  HBM traffic = events
// once + image once, for any flow; integer adds commute => bit-exact.
template <bool U16, int MODE>
__global__ void __launch_bounds__(512) k_count_bands(
	const uint64_t* __restrict__ events, const Unit* __restrict__ units, int unitsPerWindow,
	const void* __restrict__ aux, int patchRowsPerBand, int nRegular, int colTiles, double* __restrict__ image,
	unsigned long long* __restrict__ ovf /* [0] = count, then entries */, EvalConsts c)
{
	extern __shared__ unsigned int cnt[];
	const int w = blockIdx.y;
	// colTiles > 1 (large sensors, one patch row per band): the band is cut into column tiles of
	// whole patches -- the units of a tile are still one contiguous event range -- so that the
	// counters of a workgroup stay small enough for several workgroups per CU
	const int band = blockIdx.x / colTiles;
	const int tile = blockIdx.x - band * colTiles;
	const int unitsPerTile = (c.npx + colTiles - 1) / colTiles;
	const int uLo = min(tile * unitsPerTile, c.npx), uHi = min(uLo + unitsPerTile, c.npx);
	const int x0 = uLo * c.patch_w;
	const int x1 = (uHi == c.npx) ? c.image_w : uHi * c.patch_w;
	if (uLo >= uHi)
	{
		return;
	}
	// bands 0..nRegular-1: patchRowsPerBand whole patch rows each, over patch rows [0, npy-1);
	// then the last patch row (which absorbs the remainder of the image height and can be
	// almost twice as tall) in sub-bands of at most patchRowsPerBand * patch_h rows.
	int pr0, pr1, row0, row1;
	bool reportsOutside = true;  // this workgroup appends the hits outside [regionRow0, regionRow1)
	if (band < nRegular)
	{
		pr0 = band * patchRowsPerBand;
		pr1 = min(pr0 + patchRowsPerBand, c.npy - 1);
		row0 = pr0 * c.patch_h;
		row1 = pr1 * c.patch_h;
	}
	else
	{
		pr0 = c.npy - 1;
		pr1 = c.npy;
		const int sub = band - nRegular;
		row0 = pr0 * c.patch_h + sub * patchRowsPerBand * c.patch_h;
		row1 = min(row0 + patchRowsPerBand * c.patch_h, c.image_h);
		reportsOutside = sub == 0;
	}
	// rows another workgroup streaming the same events counts in its own LDS
	const int regionRow0 = pr0 * c.patch_h;
	const int regionRow1 = (pr1 == c.npy) ? c.image_h : pr1 * c.patch_h;
	const int rows = row1 - row0;
	const int W = c.image_w;
	const int tw = x1 - x0;  // == W without column tiles
	const int npx = rows * tw;
	const int nWords = U16 ? (npx + 1) >> 1 : npx;
	for (int i = threadIdx.x; i < nWords; i += blockDim.x)
	{
		cnt[i] = 0u;
	}
	__syncthreads();
	const Unit* wu = units + static_cast<size_t>(w) * unitsPerWindow;
	const int P = c.npx * c.npy;
	const size_t imgSize = static_cast<size_t>(W) * c.image_h;
	const double* windowFlows = static_cast<const double*>(aux) + (MODE == 1 ? 2 * static_cast<size_t>(w) * P : 0);
	const float* windowField = static_cast<const float*>(aux) + (MODE == 2 ? 2 * static_cast<size_t>(w) * imgSize : 0);
	constexpr int kInFlight = 8;
	// pass 0: the band's own patch units; pass 1 (band 0, tile 0 only): the stray unit
	for (int pass = 0; pass < ((band == 0 && tile == 0) ? 2 : 1); ++pass)
	{
		int ui = pass == 0 ? pr0 * c.npx + uLo : P;
		const int uLast = pass == 0 ? (pr1 - 1) * c.npx + uHi - 1 : P;
		const uint32_t evBegin = wu[ui].ev_off;
		const uint32_t evEnd = wu[uLast].ev_off + wu[uLast].n_ev;
		uint32_t uEnd = wu[ui].ev_off + wu[ui].n_ev;
		double m0 = 0.0, m1 = 0.0;
		int dtWin = wu[ui].dt_win;
		const bool stray = pass == 1;
		if (MODE == 1 && !stray)
		{
			m0 = windowFlows[2 * ui];
			m1 = windowFlows[2 * ui + 1];
		}
		for (uint32_t eb = evBegin + threadIdx.x; eb < evEnd; eb += kInFlight * blockDim.x)
		{
			uint64_t recs[kInFlight];
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const uint32_t ek = eb + k * blockDim.x;
				recs[k] = (ek < evEnd) ? events[ek] : 0ull;
			}
#pragma unroll
			for (int k = 0; k < kInFlight; ++k)
			{
				const uint32_t e = eb + k * blockDim.x;
				// every lane of the wave walks the same k: the overflow append below is one
				// atomic per wave (ballot + prefix count), so the tail lanes stay in the loop
				const bool inRange = e < evEnd;
				if (MODE != 0 && inRange && e >= uEnd)
				{
					do
					{
						++ui;
						uEnd = wu[ui].ev_off + wu[ui].n_ev;
					} while (e >= uEnd);
					dtWin = wu[ui].dt_win;
					if (MODE == 1)
					{
						m0 = windowFlows[2 * ui];
						m1 = windowFlows[2 * ui + 1];
					}
				}
				if (MODE == 1 && stray)
				{
					stray_flow(recs[k], windowFlows, c, m0, m1);
				}
				int nx, ny;
				const bool live = count_target<MODE>(recs[k], inRange, dtWin, m0, m1, windowField, c, nx, ny);
				const int ry = ny - row0;
				const bool inBand = live && ry >= 0 && ry < rows && nx >= x0 && nx < x1;
				if (inBand)
				{
					const int p = ry * tw + (nx - x0);
					if (U16)
					{
						atomicAdd(&cnt[p >> 1], 1u << ((p & 1) * 16));
					}
					else
					{
						atomicAdd(&cnt[p], 1u);
					}
				}
				// (a stray event is streamed by this workgroup only; an own event that lands in
				// another sub-band of the same patch row is counted there)
				const bool spill = MODE != 0 && live && !inBand &&
								   (stray || (reportsOutside && (ny < regionRow0 || ny >= regionRow1 || nx < x0 || nx >= x1)));
				const unsigned long long spillMask = __ballot(spill);
				if (spillMask != 0ull)
				{
					const int lane = threadIdx.x & 63;
					const int leader = __ffsll(static_cast<long long>(spillMask)) - 1;
					unsigned int baseLo = 0u, baseHi = 0u;
					if (lane == leader)
					{
						const unsigned long long b =
							atomicAdd(ovf, static_cast<unsigned long long>(__popcll(spillMask)));
						baseLo = static_cast<unsigned int>(b);
						baseHi = static_cast<unsigned int>(b >> 32);
					}
					baseLo = __shfl(baseLo, leader, 64);
					baseHi = __shfl(baseHi, leader, 64);
					if (spill)
					{
						const unsigned long long base = (static_cast<unsigned long long>(baseHi) << 32) | baseLo;
						const unsigned long long before = spillMask & ((1ull << lane) - 1ull);
						ovf[1 + base + __popcll(before)] =
							static_cast<unsigned long long>(w) * imgSize + static_cast<size_t>(ny) * W + nx;
					}
				}
			}
		}
	}
	__syncthreads();
	double* out = image + static_cast<size_t>(w) * imgSize + static_cast<size_t>(row0) * W;
	if (tw != W)
	{
		// column tile: row segments of tw pixels
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			const int r = p / tw, lx = p - r * tw;
			const unsigned int v = U16 ? ((cnt[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu) : cnt[p];
			out[static_cast<size_t>(r) * W + x0 + lx] = static_cast<double>(v);
		}
	}
	else if (U16 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
	{
		const int pairs = npx >> 1;
		double2* out2 = reinterpret_cast<double2*>(out);
		for (int i = threadIdx.x; i < pairs; i += blockDim.x)
		{
			const unsigned int v = cnt[i];
			out2[i] = make_double2(static_cast<double>(v & 0xFFFFu), static_cast<double>(v >> 16));
		}
		if ((npx & 1) && threadIdx.x == 0)
		{
			out[npx - 1] = static_cast<double>(cnt[pairs] & 0xFFFFu);
		}
	}
	else
	{
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			const unsigned int v = U16 ? ((cnt[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu) : cnt[p];
			out[p] = static_cast<double>(v);
		}
	}
}

// Adds the events that left their band (k_count_bands) and re-arms the list.
__global__ void k_count_overflow(unsigned long long* __restrict__ ovf, double* __restrict__ image)
{
	const unsigned long long n = ovf[0];
	const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
	for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
		 i += stride)
	{
		unsafeAtomicAdd(&image[ovf[1 + i]], 1.0);  // global_atomic_add_f64; exact on integer counts
	}
}

// ---------------------------------------------------------------------------------------
// Sorted bands (impl 3): warped count images of sensors too large for a whole-window LDS image.
// Every event's DESTINATION decides which row band counts it, so the events are first sorted
// by destination band -- histogram, scan, scatter of 4-byte destination pixels -- and then each
// (band, window) workgroup counts its own list in LDS and writes finished f64 rows.  No global
// atomics on the image, no per-event random HBM access, the same cost for any flow magnitude.
// Traffic per event: 8 B (histogram) + 8 B (scatter) + 4 B written + 4 B read, + the image
// once: ~2x the algorithmic bytes, all of it streaming.
//   sortBins: [0, nBins) counts, [nBins, 2 nBins] exclusive starts, [2 nBins + 1, 3 nBins + 1)
//   cursors; bin = window * bandsPerWindow + band.
// ---------------------------------------------------------------------------------------
constexpr int kSortChunk = 4096;  // events per workgroup step (256 lanes x 16)

// The chunk [e0, e1) of window w: destination pixel (ny * W + nx) of each of the lane's 16
// events, or 0xFFFFFFFF when the event contributes nothing.
template <int MODE>
__device__ __forceinline__ void sort_targets(const uint64_t* __restrict__ events, const Unit* __restrict__ wu,
											 int unitsPerWindow, const void* __restrict__ aux, int w, uint32_t e0,
											 uint32_t e1, const EvalConsts& c, unsigned int (&dst)[16])
{
	const int P = c.npx * c.npy;
	const size_t imgSize = static_cast<size_t>(c.image_w) * c.image_h;
	const double* windowFlows = static_cast<const double*>(aux) + (MODE == 1 ? 2 * static_cast<size_t>(w) * P : 0);
	const float* windowField = static_cast<const float*>(aux) + (MODE == 2 ? 2 * static_cast<size_t>(w) * imgSize : 0);
	uint64_t recs[16];
#pragma unroll
	for (int k = 0; k < 16; ++k)
	{
		const uint32_t e = e0 + threadIdx.x + k * 256;
		recs[k] = e < e1 ? events[e] : 0ull;
	}
#pragma unroll
	for (int k = 0; k < 16; ++k)
	{
		const uint32_t e = e0 + threadIdx.x + k * 256;
		int patch, unit;
		event_unit(recs[k], c, patch, unit);
		double m0 = 0.0, m1 = 0.0;
		if (MODE == 1)
		{
			m0 = windowFlows[2 * patch];
			m1 = windowFlows[2 * patch + 1];
		}
		int nx, ny;
		const bool hit = count_target<MODE>(recs[k], e < e1, wu[unit].dt_win, m0, m1, windowField, c, nx, ny);
		dst[k] = hit ? static_cast<unsigned int>(ny * c.image_w + nx) : 0xFFFFFFFFu;
	}
}

template <int MODE>
__global__ void __launch_bounds__(256) k_csort_hist(const uint64_t* __restrict__ events, const Unit* __restrict__ units,
													 int unitsPerWindow, const void* __restrict__ aux, int rowsPerBand,
													 int bandsPerWindow, unsigned int* __restrict__ sortBins,
													 unsigned int* __restrict__ dstList, EvalConsts c)
{
	extern __shared__ unsigned int hist[];
	const int w = blockIdx.y;
	const Unit* wu = units + static_cast<size_t>(w) * unitsPerWindow;
	const uint32_t evBegin = wu[0].ev_off;
	const uint32_t evEnd = wu[unitsPerWindow - 1].ev_off + wu[unitsPerWindow - 1].n_ev;
	const uint32_t e0 = evBegin + blockIdx.x * kSortChunk;
	if (e0 >= evEnd)
	{
		return;
	}
	const uint32_t e1 = min(e0 + kSortChunk, evEnd);
	for (int b = threadIdx.x; b < bandsPerWindow; b += blockDim.x)
	{
		hist[b] = 0u;
	}
	__syncthreads();
	unsigned int dst[16];
	sort_targets<MODE>(events, wu, unitsPerWindow, aux, w, e0, e1, c, dst);
	const unsigned int bandPx = static_cast<unsigned int>(rowsPerBand) * c.image_w;
#pragma unroll
	for (int k = 0; k < 16; ++k)
	{
		// the destinations are kept (4 B/event, in event order) so that the scatter pass does
		// not warp again: the f64 warp arithmetic, not the traffic, is what these passes cost
		const uint32_t e = e0 + threadIdx.x + k * 256;
		if (e < e1)
		{
			dstList[e] = dst[k];
		}
		if (dst[k] != 0xFFFFFFFFu)
		{
			atomicAdd(&hist[dst[k] / bandPx], 1u);
		}
	}
	__syncthreads();
	for (int b = threadIdx.x; b < bandsPerWindow; b += blockDim.x)
	{
		if (hist[b])
		{
			atomicAdd(&sortBins[static_cast<size_t>(w) * bandsPerWindow + b], hist[b]);
		}
	}
}

// Exclusive prefix sum of one value per thread over the workgroup (wave scans by shuffles, then
// the wave totals); tmp: at least blockDim / 64 + 1 words of LDS.  Returns the exclusive prefix,
// total = sum over the workgroup.  Ends with the values in tmp still needed: callers barrier before
// reusing tmp.
__device__ __forceinline__ unsigned int block_exclusive_scan(unsigned int v, unsigned int* tmp, unsigned int& total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
	unsigned int incl = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1)
	{
		const unsigned int up = __shfl_up(incl, d, 64);
		if (lane >= d)
		{
			incl += up;
		}
	}
	if (lane == 63)
	{
		tmp[wave] = incl;
	}
	__syncthreads();
	unsigned int before = 0;
	total = 0;
	for (int k = 0; k < nWaves; ++k)
	{
		const unsigned int t = tmp[k];
		before += k < wave ? t : 0u;
		total += t;
	}
	return before + incl - v;
}

// Exclusive scan of the bin counts (a few thousand bins: one workgroup), cursors = starts.
__global__ void __launch_bounds__(1024) k_csort_scan(unsigned int* __restrict__ sortBins, int nBins)
{
	__shared__ unsigned int part[1024];
	unsigned int* counts = sortBins;
	unsigned int* starts = sortBins + nBins;
	unsigned int* cursors = sortBins + 2 * nBins + 1;
	const int per = (nBins + 1023) / 1024;
	const int b0 = threadIdx.x * per, b1 = min(b0 + per, nBins);
	unsigned int s = 0;
	for (int b = b0; b < b1; ++b)
	{
		s += counts[b];
	}
	unsigned int total;
	unsigned int run = block_exclusive_scan(s, part, total);
	if (threadIdx.x == 0)
	{
		starts[nBins] = total;
	}
	for (int b = b0; b < b1; ++b)
	{
		starts[b] = run;
		cursors[b] = run;
		run += counts[b];
	}
}

__global__ void __launch_bounds__(256) k_csort_scatter(const Unit* __restrict__ units, int unitsPerWindow,
														const unsigned int* __restrict__ dstList, int rowsPerBand,
														int bandsPerWindow, unsigned int* __restrict__ sortBins, int nBins,
														unsigned int* __restrict__ sorted, EvalConsts c)
{
	extern __shared__ unsigned int sortLds[];
	unsigned int* hist = sortLds;                       // counts of this chunk per band
	unsigned int* base = sortLds + bandsPerWindow;      // start of the chunk's range in the band's list
	unsigned int* lstart = base + bandsPerWindow;       // start of the band inside the staged chunk
	unsigned int* lcur = lstart + bandsPerWindow;       // cursor inside the staged chunk
	unsigned int* staged = lcur + bandsPerWindow;       // [kSortChunk] destinations grouped by band
	unsigned short* bandOf = reinterpret_cast<unsigned short*>(staged + kSortChunk);  // [kSortChunk]
	__shared__ unsigned int part[256];
	const int w = blockIdx.y;
	const Unit* wu = units + static_cast<size_t>(w) * unitsPerWindow;
	const uint32_t evBegin = wu[0].ev_off;
	const uint32_t evEnd = wu[unitsPerWindow - 1].ev_off + wu[unitsPerWindow - 1].n_ev;
	const uint32_t e0 = evBegin + blockIdx.x * kSortChunk;
	if (e0 >= evEnd)
	{
		return;
	}
	const uint32_t e1 = min(e0 + kSortChunk, evEnd);
	for (int b = threadIdx.x; b < bandsPerWindow; b += blockDim.x)
	{
		hist[b] = 0u;
	}
	__syncthreads();
	unsigned int dst[16];
#pragma unroll
	for (int k = 0; k < 16; ++k)
	{
		const uint32_t e = e0 + threadIdx.x + k * 256;
		dst[k] = e < e1 ? dstList[e] : 0xFFFFFFFFu;
	}
	const unsigned int bandPx = static_cast<unsigned int>(rowsPerBand) * c.image_w;
#pragma unroll
	for (int k = 0; k < 16; ++k)
	{
		if (dst[k] != 0xFFFFFFFFu)
		{
			atomicAdd(&hist[dst[k] / bandPx], 1u);
		}
	}
	__syncthreads();
	// exclusive scan of hist over the bands (segments per thread, then 256 partials), and one
	// global atomic per (chunk, band) reserves the chunk's range in the band's list
	unsigned int* cursors = sortBins + 2 * nBins + 1;
	const int per = (bandsPerWindow + 255) / 256;
	const int s0 = min(static_cast<int>(threadIdx.x) * per, bandsPerWindow), s1 = min(s0 + per, bandsPerWindow);
	unsigned int sum = 0;
	for (int bnd = s0; bnd < s1; ++bnd)
	{
		sum += hist[bnd];
	}
	unsigned int chunkTotal;
	unsigned int run = block_exclusive_scan(sum, part, chunkTotal);  // (a serial scan of the 256 partials by one thread was 45 % of this kernel)
	for (int bnd = s0; bnd < s1; ++bnd)
	{
		const unsigned int h = hist[bnd];
		lstart[bnd] = run;
		lcur[bnd] = run;
		base[bnd] = h ? atomicAdd(&cursors[static_cast<size_t>(w) * bandsPerWindow + bnd], h) : 0u;
		run += h;
	}
	__syncthreads();
	// group the chunk's destinations by band in LDS, then stream them out: consecutive lanes
	// write consecutive entries of a band's list (4-byte stores scattered per lane were the
	// bottleneck of the first version)
#pragma unroll
	for (int k = 0; k < 16; ++k)
	{
		if (dst[k] != 0xFFFFFFFFu)
		{
			// (one LDS atomic per distinct band of the wave -- ballot, leader add, prefix count -- was
			// tried for these ranks and for the histograms: C3 0.34 -> 0.23 ms, but C4 0.32 -> 0.49 ms,
			// where a wave's events spread over ~12 of the 180 four-row bands; C3 runs impl 1 anyway)
			const unsigned int bnd = dst[k] / bandPx;
			const unsigned int at = atomicAdd(&lcur[bnd], 1u);
			staged[at] = dst[k];
			bandOf[at] = static_cast<unsigned short>(bnd);
		}
	}
	__syncthreads();
	const unsigned int nValid = lstart[bandsPerWindow - 1] + hist[bandsPerWindow - 1];
	for (unsigned int j = threadIdx.x; j < nValid; j += blockDim.x)
	{
		const unsigned int bnd = bandOf[j];
		sorted[base[bnd] + (j - lstart[bnd])] = staged[j];
	}
}

template <bool U16>
__global__ void __launch_bounds__(512) k_csort_count(const unsigned int* __restrict__ sortBins, int nBins,
													  const unsigned int* __restrict__ sorted, int rowsPerBand,
													  int bandsPerWindow, double* __restrict__ image, EvalConsts c)
{
	extern __shared__ unsigned int cnt[];
	const int w = blockIdx.y, band = blockIdx.x;
	const int row0 = band * rowsPerBand;
	const int rows = min(rowsPerBand, c.image_h - row0);
	const int W = c.image_w;
	const int npx = rows * W;
	const int nWords = U16 ? (npx + 1) >> 1 : npx;
	for (int i = threadIdx.x; i < nWords; i += blockDim.x)
	{
		cnt[i] = 0u;
	}
	__syncthreads();
	const unsigned int* starts = sortBins + nBins;
	const size_t bin = static_cast<size_t>(w) * bandsPerWindow + band;
	const unsigned int b0 = starts[bin], b1 = starts[bin + 1];
	const unsigned int origin = static_cast<unsigned int>(row0) * W;
	for (unsigned int i = b0 + threadIdx.x; i < b1; i += blockDim.x)
	{
		const int p = static_cast<int>(sorted[i] - origin);
		if (U16)
		{
			atomicAdd(&cnt[p >> 1], 1u << ((p & 1) * 16));
		}
		else
		{
			atomicAdd(&cnt[p], 1u);
		}
	}
	__syncthreads();
	double* out = image + static_cast<size_t>(w) * W * c.image_h + static_cast<size_t>(row0) * W;
	if (U16 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
	{
		const int pairs = npx >> 1;
		double2* out2 = reinterpret_cast<double2*>(out);
		for (int i = threadIdx.x; i < pairs; i += blockDim.x)
		{
			const unsigned int v = cnt[i];
			out2[i] = make_double2(static_cast<double>(v & 0xFFFFu), static_cast<double>(v >> 16));
		}
		if ((npx & 1) && threadIdx.x == 0)
		{
			out[npx - 1] = static_cast<double>(cnt[pairs] & 0xFFFFu);
		}
	}
	else
	{
		for (int p = threadIdx.x; p < npx; p += blockDim.x)
		{
			const unsigned int v = U16 ? ((cnt[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu) : cnt[p];
			out[p] = static_cast<double>(v);
		}
	}
}

// int32 counts -> f64 image (the reference's CV_64F); re-zeroes the scratch.
__global__ void k_counts_to_f64(int32_t* __restrict__ counts, double* __restrict__ image, size_t n)
{
	const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
	for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
	{
		image[i] = static_cast<double>(counts[i]);
		counts[i] = 0;
	}
}

// Patch::integrateEvents (patch.cpp:65-85) / integrateMotionCompensatedEvents
// (patch.cpp:87-130): signed counts in an LDS tile, one workgroup per patch.
__global__ void k_patch_integrate(const uint64_t* __restrict__ events,
								  const uint32_t* __restrict__ offsets,
								  const double* __restrict__ rects, const double* __restrict__ traj,
								  const uint64_t* __restrict__ nablaOff, double* __restrict__ nabla)
{
	extern __shared__ int tile[];
	const int p = blockIdx.x;
	const double rx = rects[4 * p + 0], ry = rects[4 * p + 1];
	const double rw = rects[4 * p + 2], rh = rects[4 * p + 3];
	const int cols = static_cast<int>(rw);
	const int rows = static_cast<int>(rh);
	double* out = nabla + nablaOff[p];
	double dirX = 0.0, dirY = 0.0, tDif = 1.0;
	bool mc = traj != nullptr;
	if (mc)
	{
		if (traj[4 * p + 3] == 0.0)
		{
			return;  // patch.cpp:99-100 time test failed: image left untouched
		}
		dirX = traj[4 * p + 0];
		dirY = traj[4 * p + 1];
		tDif = traj[4 * p + 2];
	}
	for (int i = threadIdx.x; i < rows * cols; i += blockDim.x)
	{
		tile[i] = 0;
	}
	__syncthreads();
	const uint32_t e0 = offsets[p], e1 = offsets[p + 1];
	for (uint32_t e = e0 + threadIdx.x; e < e1; e += blockDim.x)
	{
		int x, y, pos, dt;
		unpack(events[e], x, y, pos, dt);
		int ix = x, iy = y;
		if (mc)
		{
			const double f = static_cast<double>(dt) / tDif;
			const double cx = static_cast<double>(x) + f * dirX;
			const double cy = static_cast<double>(y) + f * dirY;
			if (!convertible(cx) || !convertible(cy))
			{
				continue;
			}
			ix = static_cast<int>(rint(cx));  // cv::saturate_cast<int>: half to even
			iy = static_cast<int>(rint(cy));
		}
		const double dx = static_cast<double>(ix);
		const double dy = static_cast<double>(iy);
		if (rx <= dx && dx < rx + rw && ry <= dy && dy < ry + rh)
		{
			const int px = static_cast<int>(dx - rx);
			const int py = static_cast<int>(dy - ry);
			atomicAdd(&tile[py * cols + px], pos ? 1 : -1);
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < rows * cols; i += blockDim.x)
	{
		out[i] = static_cast<double>(tile[i]);
	}
}

// FeatureDetector::updatePatches' routing test for a chunk of the stream (feature_detector.cpp:
// 589-596, cv::Rect2d::contains on the integer point): one wave per tracked patch walks the
// chunk from start[p], 256 events per step (four coalesced 256-byte loads), compacts the indices
// of the events inside the rect with ballots -- no barriers, no atomics, stream order kept -- and
// stops at the event that fills the patch's quota.
__global__ void __launch_bounds__(64) k_route(const uint32_t* __restrict__ xy, uint32_t nEvents,
											  const double* __restrict__ rects, const uint32_t* __restrict__ start,
											  const uint32_t* __restrict__ take, uint32_t cap,
											  uint32_t* __restrict__ outIndex, uint32_t* __restrict__ outCount,
											  uint32_t* __restrict__ outNext)
{
	const int p = blockIdx.x;
	const double rx = rects[4 * p + 0], ry = rects[4 * p + 1];
	const double rx1 = rx + rects[4 * p + 2], ry1 = ry + rects[4 * p + 3];
	const uint32_t quota = min(take[p], cap);
	uint32_t* out = outIndex + static_cast<size_t>(p) * cap;
	const int lane = threadIdx.x;
	uint32_t taken = 0;
	uint32_t next = nEvents;
	for (uint32_t base = start[p]; base < nEvents && taken < quota; base += 256)
	{
		uint32_t v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			const uint32_t e = base + k * 64 + lane;
			v[k] = e < nEvents ? xy[e] : 0u;
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			const uint32_t e = base + k * 64 + lane;
			const double px = static_cast<double>(static_cast<int>(static_cast<int16_t>(v[k] & 0xFFFFu)));
			const double py = static_cast<double>(static_cast<int>(static_cast<int16_t>(v[k] >> 16)));
			const bool in = e < nEvents && rx <= px && px < rx1 && ry <= py && py < ry1;
			const unsigned long long m = __ballot(in);
			const uint32_t rank = taken + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
			if (in && rank < quota)
			{
				out[rank] = e;
				if (rank + 1 == quota)
				{
					next = e + 1;  // exactly one lane of the whole walk
				}
			}
			taken += static_cast<uint32_t>(__popcll(m));
		}
	}
	// `next` lives in the lane that took the last event: publish it to the wave
	const unsigned long long who = __ballot(next != nEvents);
	if (who)
	{
		next = __shfl(next, __ffsll(static_cast<long long>(who)) - 1);
	}
	if (lane == 0)
	{
		outCount[p] = min(taken, quota);
		outNext[p] = (quota == 0) ? start[p] : next;
	}
}

#include "ebo_edge.inc"
#include "ebo_bucket.inc"
#include "ebo_field.inc"
#include "ebo_fieldtv.inc"
#include "ebo_optimizer.inc"
#include "ebo_band.inc"

int check_launch()
{
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename K>
int allow_big_lds(K kernel, size_t bytes)
{
	if (bytes <= 64 * 1024)
	{
		return 0;
	}
	return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
							   hipFuncAttributeMaxDynamicSharedMemorySize,
							   static_cast<int>(bytes)) == hipSuccess
			   ? 0
			   : -2;
}

}  // namespace

int launch_eval_variance(const EvalLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (L.n_units == 0)
	{
		return 0;
	}
	const dim3 grid(L.n_units * L.tiles, L.flow_sets);
#ifdef EBO_AB
	if (L.impl == 0)
	{
		auto kern = (L.channels == 3) ? k_eval_variance<3> : k_eval_variance<1>;
		if (allow_big_lds(kern, L.lds_bytes))
		{
			return -2;
		}
		hipLaunchKernelGGL(kern, grid, dim3(L.block), L.lds_bytes, s, L.d_events, L.d_units,
						   L.d_flows, L.tiles, L.fd_step, L.d_partials, L.d_out, L.c);
	}
	else if (L.impl < 3)
	{
		auto kern = (L.impl == 2) ? (L.rotate ? k_eval2<true, true> : k_eval2<true, false>)
								  : (L.rotate ? k_eval2<false, true> : k_eval2<false, false>);
		if (allow_big_lds(kern, L.lds_bytes))
		{
			return -2;
		}
		hipLaunchKernelGGL(kern, grid, dim3(L.block), L.lds_bytes, s, L.d_events, L.d_units,
						   L.d_flows, L.tiles, L.channels == 3 ? 1 : 0, L.cap_doubles, L.fd_step,
						   L.d_partials, L.d_out, L.c);
	}
	else
#endif  // EBO_AB
	{
#ifdef EBO_AB
		auto kern = (L.c.inv_sigsq <= 1.0) ? (L.deal ? k_eval3<true, true> : k_eval3<true, false>)
										   : (L.deal ? k_eval3<false, true> : k_eval3<false, false>);
#else
		auto kern = (L.c.inv_sigsq <= 1.0) ? k_eval3<true, false> : k_eval3<false, false>;
#endif
		if (allow_big_lds(kern, L.lds_bytes))
		{
			return -2;
		}
		const bool fusedPath = L.tiles == 1 && L.flow_sets == 1;
		LiveWindows live = L.live;
		if (!fusedPath)
		{
			live.n = 0;
		}
		hipLaunchKernelGGL(kern, live.n > 0 ? dim3(live.n * live.upw) : grid, dim3(L.block), L.lds_bytes, s, L.d_events, L.d_units,
						   L.d_flows, L.tiles, L.channels == 3 ? 1 : 0, L.cap_doubles, L.fd_step,
						   L.d_partials, L.d_out, L.c, fusedPath ? L.d_modes : nullptr, live);
	}
	if (check_launch())
	{
		return -2;
	}
	if (!(L.tiles == 1 && L.flow_sets == 1))
	{
		hipLaunchKernelGGL(k_combine_variance, dim3((L.n_units + 127) / 128), dim3(128), 0, s,
						   L.d_units, L.n_units, L.d_flows, L.tiles, L.flow_sets, L.channels,
						   L.fd_step, L.d_partials, L.d_out, L.c);
	}
	return check_launch();
}

int launch_init_field(const FieldLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const size_t npx = static_cast<size_t>(L.w) * L.h;
	if (hipMemsetAsync(L.d_field, 0, npx * 2 * sizeof(float), s) != hipSuccess)
	{
		return -2;
	}
	hipLaunchKernelGGL(k_field_fixed, dim3(1), dim3(64), 0, s, L.w, L.h, L.scale, L.n_patches, L.d_off,
					   L.d_xy, L.d_t, L.timestamp, L.d_field, L.d_fixed, L.d_avg, L.d_nfixed);
	if (check_launch())
	{
		return -2;
	}
	hipLaunchKernelGGL(k_field_fill, dim3(static_cast<unsigned>((npx + 255) / 256)), dim3(256), 0, s, L.w,
					   L.h, L.use_average, L.d_field, L.d_fixed, L.d_avg, L.d_nfixed);
	return check_launch();
}

namespace
{
// grid sizes of the multigrid levels: halve until <= 256 nodes
int tvf_mg_dims(int w, int h, int (&lw)[12], int (&lh)[12])
{
	int n = 0;
	lw[0] = w;
	lh[0] = h;
	n = 1;
	while (lw[n - 1] * lh[n - 1] > 256 && n < 12)
	{
		lw[n] = (lw[n - 1] + 1) / 2;
		lh[n] = (lh[n - 1] + 1) / 2;
		++n;
	}
	return n;
}
size_t al256(size_t v)
{
	return (v + 255) & ~static_cast<size_t>(255);
}
}  // namespace

size_t tvf_workspace_bytes(int w, int h)
{
	const size_t n = static_cast<size_t>(w) * h;
	const size_t nAl = al256(n);
	size_t bytes = nAl /*mask*/ + 5 * nAl * 8 + 10 * nAl * 16 + 2 * (4 * 1024 * 8) + 256 /*scal*/;
	// multigrid: level 0 masked weights + x; levels >= 1 three weight arrays + three vectors
	int lw[12], lh[12];
	const int L = tvf_mg_dims(w, h, lw, lh);
	bytes += 2 * al256(n * 8) + al256(n * 16);
	for (int l = 1; l < L; ++l)
	{
		const size_t nl = static_cast<size_t>(lw[l]) * lh[l];
		bytes += 3 * al256(nl * 8) + 3 * al256(nl * 16);
	}
	return bytes;
}

void tvf_carve(TvfArgs& A, TvfMg& M, int w, int h, void* base, double2** xbest)
{
	const size_t n = static_cast<size_t>(w) * h;
	const size_t nAl = al256(n);
	char* b = static_cast<char*>(base);
	auto take = [&](size_t bytes) {
		char* r = b;
		b += al256(bytes);
		return r;
	};
	A.w = w;
	A.h = h;
	A.n = static_cast<int>(n);
	double2** v2[] = {&A.x, &A.xc, xbest, &A.g, &A.y, &A.r, &A.z, &A.p0, &A.p1, &A.q};
	for (double2** v : v2)
	{
		*v = reinterpret_cast<double2*>(take(nAl * 16));
	}
	double** v1[] = {&A.wh, &A.wv, &A.deg, &A.s2, &A.diag};
	for (double** v : v1)
	{
		*v = reinterpret_cast<double*>(take(nAl * 8));
	}
	A.partials = reinterpret_cast<double*>(take(4 * 1024 * 8));
	A.partials_rz = reinterpret_cast<double*>(take(4 * 1024 * 8));
	A.scal = reinterpret_cast<double*>(take(256));
	A.mask = reinterpret_cast<unsigned char*>(take(nAl));
	// multigrid hierarchy
	int lw[12], lh[12];
	M.levels = tvf_mg_dims(w, h, lw, lh);
	for (int l = 0; l < M.levels; ++l)
	{
		const size_t nl = static_cast<size_t>(lw[l]) * lh[l];
		TvfLevel& L = M.lv[l];
		L.w = lw[l];
		L.h = lh[l];
		M.wh_m[l] = reinterpret_cast<double*>(take(nl * 8));
		M.wv_m[l] = reinterpret_cast<double*>(take(nl * 8));
		if (l == 0)
		{
			M.diag_m[0] = A.diag;  // deg + damping, 1 on the pixels that are not free (k_tvf_cg_init)
			L.b = A.r;             // the CG residual
			L.x = reinterpret_cast<double2*>(take(nl * 16));
			L.xo = A.z;            // z = M^-1 r
		}
		else
		{
			M.diag_m[l] = reinterpret_cast<double*>(take(nl * 8));
			L.b = reinterpret_cast<double2*>(take(nl * 16));
			L.x = reinterpret_cast<double2*>(take(nl * 16));
			L.xo = reinterpret_cast<double2*>(take(nl * 16));
		}
		L.wh = M.wh_m[l];
		L.wv = M.wv_m[l];
		L.diag = M.diag_m[l];
	}
	M.omega = 0.8;
	M.kappa = 1.8;
	M.coarse_sweeps = 8;
}

namespace
{
// Workgroups of the chunk-walking kernels: a multiple of 8 (one band per XCD), <= 1024.
unsigned tvf_grid(int n)
{
	const int chunks = (n + 255) / 256;
	const int perBand = (chunks + 7) / 8;
	return 8u * static_cast<unsigned>(std::min(perBand, 128));
}
}  // namespace

int launch_tvf_prepare(const TvfArgs& A, const float* d_field, const int* d_fixed, int n_fixed, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (hipMemsetAsync(A.scal, 0, 256, s) != hipSuccess)
	{
		return -2;
	}
	const unsigned G = tvf_grid(A.n);
	hipLaunchKernelGGL(k_tvf_prepare, dim3(G), dim3(256), 0, s, A, reinterpret_cast<const float2*>(d_field));
	hipLaunchKernelGGL((k_tvf_reduce<1, 0>), dim3(1), dim3(256), 0, s, A.partials, static_cast<int>(G),
					   A.scal + kTvfNorm, 1);
	if (n_fixed > 0)
	{
		hipLaunchKernelGGL(k_tvf_mark_fixed, dim3((n_fixed + 255) / 256), dim3(256), 0, s, A, d_fixed, n_fixed);
	}
	return check_launch();
}

int launch_tvf_linearize(const TvfArgs& A, const double2* X, int first, int cost_only, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const unsigned G = tvf_grid(A.n);
	hipLaunchKernelGGL(k_tvf_linearize, dim3(G), dim3(256), 0, s, A, X, first, cost_only);
	// cost only: the |x|^2 and max |g| slots of the current point stay as they are
	if (cost_only)
	{
		hipLaunchKernelGGL((k_tvf_reduce<3, 1>), dim3(1), dim3(256), 0, s, A.partials, static_cast<int>(G),
						   A.scal + 20, 0);
	}
	else
	{
		hipLaunchKernelGGL((k_tvf_reduce<3, 1>), dim3(1), dim3(256), 0, s, A.partials, static_cast<int>(G),
						   A.scal + kTvfCost, 0);
	}
	return check_launch();
}

namespace
{
// z = M^-1 r by one V(1,1) cycle; the last kernel leaves the r'z partials for the CG.
void tvf_mg_vcycle(const TvfArgs& A, const TvfMg& M, hipStream_t s)
{
	const int L = M.levels;
	for (int l = 0; l + 1 < L; ++l)
	{
		const TvfLevel& f = M.lv[l];
		const TvfLevel& c = M.lv[l + 1];
		const int nc = c.w * c.h;
		hipLaunchKernelGGL(k_mg_down, dim3((nc + 255) / 256), dim3(256), 0, s, f, c.w, c.h, c.b, M.omega);
	}
	hipLaunchKernelGGL(k_mg_coarse, dim3(1), dim3(256), 0, s, M.lv[L - 1], M.omega, M.coarse_sweeps);
	for (int l = L - 2; l >= 0; --l)
	{
		const TvfLevel& f = M.lv[l];
		const TvfLevel& c = M.lv[l + 1];
		hipLaunchKernelGGL(k_mg_up, dim3(tvf_grid(f.w * f.h)), dim3(256), 0, s, f, c.xo, c.w, M.omega, M.kappa,
						   l == 0 ? A.partials_rz : nullptr, l == 0 ? A.mask : nullptr);
	}
}
}  // namespace

// The hierarchy of the operator cg_init just defined (weights of this linearisation, damping of
// this radius), then the first V-cycle (z and r'z of CG iteration 0).
int launch_tvf_mg_build(const TvfArgs& A, const TvfMg& M, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (M.levels < 2)
	{
		return 0;
	}
	hipLaunchKernelGGL(k_mg_mask0, dim3(tvf_grid(A.n)), dim3(256), 0, s, A, M.wh_m[0], M.wv_m[0]);
	for (int l = 0; l + 1 < M.levels; ++l)
	{
		const TvfLevel& c = M.lv[l + 1];
		const int nc = c.w * c.h;
		hipLaunchKernelGGL(k_mg_coarsen, dim3((nc + 255) / 256), dim3(256), 0, s, M.lv[l], c, M.wh_m[l + 1],
						   M.wv_m[l + 1], M.diag_m[l + 1]);
	}
	tvf_mg_vcycle(A, M, s);
	return check_launch();
}

int launch_tvf_cg_init(const TvfArgs& A, const TvfMg& M, double radius, void* stream)
{
	hipLaunchKernelGGL(k_tvf_cg_init, dim3(tvf_grid(A.n)), dim3(256), 0, static_cast<hipStream_t>(stream), A, radius,
					   M.levels >= 2 ? 1 : 0);
	if (check_launch())
	{
		return -2;
	}
	return launch_tvf_mg_build(A, M, stream);
}

int launch_tvf_cg_iters(const TvfArgs& A, const TvfMg& M, int first_iter, int iters, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const unsigned G = tvf_grid(A.n);
	const int mg = M.levels >= 2 ? 1 : 0;
	for (int k = first_iter; k < first_iter + iters; ++k)
	{
		const double2* pOld = (k & 1) ? A.p1 : A.p0;
		double2* pNew = (k & 1) ? A.p0 : A.p1;
		hipLaunchKernelGGL(k_tvf_cg_apply, dim3(G), dim3(256), 0, s, A, pOld, pNew, k);
		hipLaunchKernelGGL(k_tvf_cg_update, dim3(G), dim3(256), 0, s, A, pNew, k, mg);
		if (mg)
		{
			tvf_mg_vcycle(A, M, s);
		}
	}
	return check_launch();
}

int launch_tvf_model(const TvfArgs& A, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const unsigned G = tvf_grid(A.n);
	hipLaunchKernelGGL(k_tvf_model, dim3(G), dim3(256), 0, s, A);
	hipLaunchKernelGGL((k_tvf_reduce<3, 0>), dim3(1), dim3(256), 0, s, A.partials, static_cast<int>(G),
					   A.scal + kTvfYg, 0);
	return check_launch();
}

int launch_tvf_store(const TvfArgs& A, const double2* X, float* d_field, void* stream)
{
	const unsigned blocks = static_cast<unsigned>((A.n + 255) / 256);
	hipLaunchKernelGGL(k_tvf_store, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), A.n, X,
					   reinterpret_cast<float2*>(d_field));
	return check_launch();
}

namespace
{
size_t optimizer_lds_bytes(int maxPixels)
{
	return (128 + static_cast<size_t>(maxPixels) * 6) * sizeof(double);
}
}  // namespace

int launch_optimizer_eval(const OptLaunch& L, void* stream)
{
	if (L.n_patches == 0)
	{
		return 0;
	}
	const size_t lds = optimizer_lds_bytes(L.max_pixels);
	if (lds > 160 * 1024 - 512 || allow_big_lds(k_optimizer_eval, lds))
	{
		return -2;
	}
	hipLaunchKernelGGL(k_optimizer_eval, dim3(L.n_patches), dim3(256), lds, static_cast<hipStream_t>(stream),
					   L.d_grid, L.img_w, L.img_h, L.d_patches, L.d_nabla, L.d_x, L.d_res, L.d_jac_pose,
					   L.d_jac_flow);
	return check_launch();
}

int launch_optimizer_cost_map(const OptLaunch& L, const double* d_xcells, int cells, double* d_out, void* stream)
{
	if (L.n_patches == 0 || cells == 0)
	{
		return 0;
	}
	const size_t lds = optimizer_lds_bytes(L.max_pixels);
	if (lds > 160 * 1024 - 512 || allow_big_lds(k_optimizer_cost_map, lds))
	{
		return -2;
	}
	hipLaunchKernelGGL(k_optimizer_cost_map, dim3(cells, L.n_patches), dim3(256), lds, static_cast<hipStream_t>(stream),
					   L.d_grid, L.img_w, L.img_h, L.d_patches, L.d_nabla, d_xcells, d_out);
	return check_launch();
}

int launch_optimizer_solve(const OptLaunch& L, void* stream)
{
	if (L.n_patches == 0)
	{
		return 0;
	}
	const size_t lds = optimizer_lds_bytes(L.max_pixels);
	if (lds > 160 * 1024 - 512 || allow_big_lds(k_optimizer_solve, lds))
	{
		return -2;
	}
	// measured (25x25 patches): 256 lanes best up to ~100 patches (0.27 / 0.38 ms for 1 / 100), 128
	// lanes at 1000 (1.13 vs 1.32 ms); 384+ lanes slower everywhere.  EBO_OPT_BLOCK for A/B.
	int block = L.n_patches >= 512 ? 128 : 256;
	if (const char* v = ab_env("EBO_OPT_BLOCK"))
	{
		block = std::min(std::max((std::atoi(v) / 64) * 64, 64), 256);
	}
	hipLaunchKernelGGL(k_optimizer_solve, dim3(L.n_patches), dim3(block), lds, static_cast<hipStream_t>(stream),
					   L.d_grid, L.img_w, L.img_h, L.d_patches, L.d_nabla, L.d_x, L.d_stats, L.huber, L.s,
					   ab_env("EBO_OPT_NO_SPECULATE") ? 0 : 1);
	return check_launch();
}

int launch_optimizer_normalize(const OptPatch* d_patches, int n, const double* d_in, double* d_out, void* stream)
{
	if (n == 0)
	{
		return 0;
	}
	hipLaunchKernelGGL(k_optimizer_normalize, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), d_patches,
					   d_in, d_out);
	return check_launch();
}

int launch_optimizer_interleave(const double* d_gx, const double* d_gy, size_t n, double2* d_grid, void* stream)
{
	hipLaunchKernelGGL(k_optimizer_interleave, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
					   static_cast<hipStream_t>(stream), d_gx, d_gy, n, d_grid);
	return check_launch();
}

int launch_estimate_num_events(const double2* d_grid, int w, int h, int n, const double* d_rects, const double* d_poses,
								const double* d_flows, double* d_sums, void* stream)
{
	if (n <= 0)
	{
		return 0;
	}
	hipLaunchKernelGGL(k_estimate_num_events, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), d_grid, w, h, n,
					   d_rects, d_poses, d_flows, d_sums);
	return check_launch();
}

int launch_patch_warp_image(const double2* d_grid, int w, int h, int n, const double* d_rects, const double* d_poses,
							const double* d_flows, const int* d_skip, const size_t* d_offsets, double* d_out, void* stream)
{
	if (n <= 0)
	{
		return 0;
	}
	hipLaunchKernelGGL(k_patch_warp_image, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), d_grid, w, h, n, d_rects,
					   d_poses, d_flows, d_skip, d_offsets, d_out);
	return check_launch();
}

template <class Rec>
static int launch_bucket_t(const BucketLaunch& L, Rec raw, hipStream_t s)
{
	const int nUnits = L.n_windows * (L.P + 1);
	const int w0 = L.w0, w1 = L.w1 < 0 ? L.n_windows : L.w1;
	const int nw = w1 - w0;
	if (w0 == 0)
	{
		hipLaunchKernelGGL(k_bucket_init, dim3((nUnits + 255) / 256), dim3(256), 0, s, L.d_cnt, L.d_tmin,
						   L.d_tmax, nUnits, L.d_flag);
		if (check_launch())
		{
			return -2;
		}
	}
	if (nw <= 0)
	{
		return 0;
	}
	const size_t lds = static_cast<size_t>(L.P + 1) * (2 * sizeof(long long) + sizeof(int)) + 8;
	if (L.max_chunks > 0)
	{
		if (allow_big_lds(k_bucket_count<Rec>, lds))
		{
			return -2;
		}
		hipLaunchKernelGGL(k_bucket_count<Rec>, dim3(L.max_chunks, nw), dim3(256), lds, s, raw,
						   L.d_offsets, w0, L.P, L.d_cnt, L.d_tmin, L.d_tmax, L.d_flag, L.d_chunk_hist, L.chunk_events, L.c);
		if (check_launch())
		{
			return -2;
		}
	}
	hipLaunchKernelGGL(k_bucket_scan<Rec>, dim3(nw), dim3(256), 0, s, raw, L.d_offsets, w0,
					   L.n_windows, L.P, L.d_cnt, L.d_tmin, L.d_tmax, L.min_events, L.d_units,
					   L.d_unit_tref, L.d_unit_maxdt, L.d_win_tref, L.d_flag, L.c);
	if (check_launch())
	{
		return -2;
	}
	if (L.max_chunks > 0)
	{
		hipLaunchKernelGGL(k_bucket_chunk_scan, dim3((L.P + 1 + 255) / 256, nw), dim3(256), 0, s, L.d_offsets, w0, L.P,
						   L.max_chunks, L.chunk_events, L.d_chunk_hist);
		if (check_launch())
		{
			return -2;
		}
		const size_t curLds = static_cast<size_t>(L.P + 1) * sizeof(unsigned int);
		if (allow_big_lds(k_bucket_scatter<Rec>, curLds))
		{
			return -2;
		}
		hipLaunchKernelGGL(k_bucket_scatter<Rec>, dim3(L.max_chunks, nw), dim3(64), curLds, s, raw,
						   L.d_offsets, w0, L.P, L.d_chunk_hist, L.d_units, L.d_unit_tref, L.d_win_tref, L.d_packed,
						   L.d_flag, L.chunk_events, L.c);
		if (check_launch())
		{
			return -2;
		}
		const size_t sortLds = static_cast<size_t>(kSortMax) * sizeof(unsigned long long);
		if (allow_big_lds(k_bucket_canon, sortLds))
		{
			return -2;
		}
		hipLaunchKernelGGL(k_bucket_canon, dim3(nw * (L.P + 1)), dim3(256), sortLds, s,
						   L.d_units + static_cast<size_t>(w0) * (L.P + 1), L.d_packed);
	}
	return check_launch();
}

int launch_bucket(const BucketLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (L.compact)
	{
		Rec8 r;
		r.p = static_cast<const uint2*>(L.d_raw);
		r.tbase = L.d_tbase;
		return launch_bucket_t(L, r, s);
	}
	Rec24 r;
	r.p = static_cast<const RawEvent*>(L.d_raw);
	return launch_bucket_t(L, r, s);
}

int launch_lds_rate(int atomic, int blocks, int iters, double* d_sink, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (atomic)
	{
		hipLaunchKernelGGL(k_lds_rate<true>, dim3(blocks), dim3(256), 0, s, d_sink, iters);
	}
	else
	{
		hipLaunchKernelGGL(k_lds_rate<false>, dim3(blocks), dim3(256), 0, s, d_sink, iters);
	}
	return check_launch();
}

int launch_stream_yardstick(const uint64_t* d_events, size_t n_events, double* d_image, size_t n_pixels, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	hipLaunchKernelGGL(k_stream_yardstick, dim3(2048), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_events), n_events / 2,
					   reinterpret_cast<double2*>(d_image), n_pixels / 2);
	return check_launch();
}

int launch_count_shard(const uint64_t* d_events, const Unit* d_units, int n_units, int units_per_window,
					   const BandUnit* d_table, const double* d_flows, double* d_image, const EvalConsts& c, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (n_units == 0)
	{
		return 0;
	}
	hipLaunchKernelGGL(k_count_shard, dim3(n_units), dim3(256), 0, s, d_events, d_units, units_per_window, d_table,
					   d_flows, d_image, c);
	return check_launch();
}

int launch_count_band(const BandLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const int bandRows = L.band1 - L.band0;
	if (L.n_windows <= 0 || L.per <= 0 || bandRows <= 0)
	{
		return 0;
	}
	// tiles of <= 15360 counters (60 KB) + the selection list: two workgroups per CU; about 160 columns wide, so
	// that a tile's units are few (events are re-read once per tile their reach touches)
	const int W = L.c.image_w;
	const int tilesX = std::max(1, (W + 159) / 160);
	int tw = (W + tilesX - 1) / tilesX;
	const int thMax = std::max(1, 15360 / tw);
	const int tilesY = (bandRows + thMax - 1) / thMax;
	const int th = (bandRows + tilesY - 1) / tilesY;
	BandGeom g;
	g.band0 = L.band0;
	g.own0 = L.own0;
	g.own1 = L.own1;
	g.band1 = L.band1;
	g.tileW = tw;
	g.tileH = th;
	g.tilesX = tilesX;
	g.tilesY = tilesY;
	const size_t lds = (static_cast<size_t>(tw) * th + 1 + static_cast<size_t>(L.per)) * sizeof(int);
	if (lds > 160 * 1024 || allow_big_lds(k_count_band, lds))
	{
		return -2;
	}
	hipLaunchKernelGGL(k_count_band, dim3(tilesX * tilesY, L.n_windows), dim3(512), lds, s, L.d_events, L.d_units, L.d_band_units,
					   L.per, L.d_flows, g, L.d_top, L.d_own, L.d_bottom, L.d_escaped, L.c);
	return check_launch();
}

int launch_band_finish(const unsigned int* d_own, const unsigned int* d_from_above, const unsigned int* d_from_below,
					   int own_rows, int recv_above, int recv_below, int W, int n_windows, double* d_image, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const size_t n = static_cast<size_t>(n_windows) * own_rows * W;
	if (n == 0)
	{
		return 0;
	}
	const int blocks = static_cast<int>(std::min<size_t>((n + 255) / 256, 4096));
	hipLaunchKernelGGL(k_band_finish, dim3(blocks), dim3(256), 0, s, d_own, d_from_above, d_from_below, own_rows, recv_above,
					   recv_below, W, n, d_image);
	return check_launch();
}

int launch_eval_edge(const EdgeLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (L.n_units == 0)
	{
		return 0;
	}
	const bool narrow = L.block <= 256 && !L.wide_kernel;  // MAXT 256 (two workgroups per CU, persistent) or 768 (one workgroup per item)
	LiveWindows live = L.live;
	if (L.flow_sets != 1)
	{
		live.n = 0;
	}
	const int nItemsX = live.n > 0 ? live.n * live.upw : L.n_units;
	if (narrow)
	{
		using EdgeKern = decltype(&k_eval_edge<true, true, 256>);
		EdgeKern edgeKern = L.alias_lds ? (L.want_jac ? k_eval_edge<true, true, 256> : k_eval_edge<true, false, 256>)
										: (L.want_jac ? k_eval_edge<false, true, 256> : k_eval_edge<false, false, 256>);
		if (allow_big_lds(edgeKern, L.lds_bytes))
		{
			return -2;
		}
		EdgeKArgs ka;
		ka.events = L.d_events;
		ka.units = L.d_units;
		ka.flows = L.d_flows;
		ka.wantJac = L.want_jac;
		ka.capPx = L.cap_px;
		ka.fdStep = L.fd_step;
		ka.scratch = L.d_scratch;
		ka.scratchStride = L.scratch_stride;
		ka.sets = L.d_sets;
		ka.out = L.d_out;
		ka.c = L.c;
		ka.ec = L.ec;
		ka.modes = L.flow_sets == 1 ? L.d_modes : nullptr;
		ka.live = live;
		ka.nItemsX = nItemsX;
		ka.nSets = L.flow_sets;
		ka.itemList = nullptr;
		ka.itemCount = nullptr;
		ka.bbox = nullptr;
		if (L.compact.list_cap > 0 && L.alias_lds && L.flow_sets == 1)
		{
			// Two launches (round 5).  First every unit on the COMPACT layout, three 256-lane workgroups per CU (the
			// 168-VGPR instantiation); a unit whose box does not fit goes on the deferred list.  Then the deferred
			// units on the 20 B layout, two per CU, persistent workgroups that read the list's length on the device.
			using WideKern = decltype(&k_eval_edge_wg<true, true, 768>);
			int4* const bbox = static_cast<int4*>(L.compact.bbox);
			WideKern wide = bbox ? (L.want_jac ? k_eval_edge_wg<true, true, 768, true> : k_eval_edge_wg<true, false, 768, true>)
								 : (L.want_jac ? k_eval_edge_wg<true, true, 768> : k_eval_edge_wg<true, false, 768>);
			if (allow_big_lds(wide, L.compact_lds_bytes))
			{
				return -2;
			}
			if (hipMemsetAsync(L.compact.defer_count, 0, sizeof(int), s) != hipSuccess)
			{
				return -2;
			}
			EdgeConsts ecCompact = L.ec;
			ecCompact.cs_stride = L.compact_table_px;  // one slot per unit, as in rounds 1-4; the most pixels a box of this launch may have
			if (bbox)
			{
				// the bounding-box pass of every unit, and the list of the second launch, in one small kernel up front
				hipLaunchKernelGGL(k_edge_classify, dim3((nItemsX + 15) / 16), dim3(1024), 0, s, L.d_events, L.d_units, L.d_flows, L.d_modes,
								   live, nItemsX, L.compact_cap_px, L.compact_table_px, L.c, bbox, L.compact.defer_list, L.compact.defer_count);
				if (check_launch())
				{
					return -2;
				}
			}
			hipLaunchKernelGGL(wide, dim3(nItemsX, 1), dim3(256), L.compact_lds_bytes, s, L.d_events, L.d_units, L.d_flows,
							   L.want_jac, L.compact_cap_px, L.fd_step, L.d_scratch, L.scratch_stride, L.d_sets, L.d_out, L.c, ecCompact,
							   L.d_modes, live, L.compact, bbox);
			if (check_launch())
			{
				return -2;
			}
			ka.itemList = L.compact.defer_list;
			ka.itemCount = L.compact.defer_count;
			ka.bbox = bbox;
		}
		const int grid = std::min(nItemsX * L.flow_sets, std::max(L.wg_slots, 1));  // persistent workgroups
		hipLaunchKernelGGL(edgeKern, dim3(grid), dim3(L.block), L.lds_bytes, s, ka);
	}
	else
	{
		using EdgeKern = decltype(&k_eval_edge_wg<true, true, 768>);
		EdgeKern edgeKern = L.alias_lds ? (L.want_jac ? k_eval_edge_wg<true, true, 768> : k_eval_edge_wg<true, false, 768>)
										: (L.want_jac ? k_eval_edge_wg<false, true, 768> : k_eval_edge_wg<false, false, 768>);
		if (allow_big_lds(edgeKern, L.lds_bytes))
		{
			return -2;
		}
		hipLaunchKernelGGL(edgeKern, dim3(nItemsX, L.flow_sets), dim3(L.block), L.lds_bytes, s, L.d_events, L.d_units, L.d_flows,
						   L.want_jac, L.cap_px, L.fd_step, L.d_scratch, L.scratch_stride, L.d_sets, L.d_out, L.c, L.ec,
						   L.flow_sets == 1 ? L.d_modes : nullptr, live, EdgeCompact(), static_cast<const int4*>(nullptr));
	}
	if (check_launch())
	{
		return -2;
	}
	if (L.flow_sets == 5)
	{
		hipLaunchKernelGGL(k_edge_central, dim3((L.n_units + 127) / 128), dim3(128), 0, s, L.d_units,
						   L.n_units, L.fd_step, L.d_sets, L.d_out);
	}
	return check_launch();
}

int launch_solve_edge(const EdgeLaunch& L, const SolveConsts& o, double* d_flows_out, int32_t* d_stats, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (L.n_units == 0)
	{
		return 0;
	}
	using SolveKern = decltype(&k_solve_edge<true, 256>);
	if (L.block > 512)
	{
		return -2;  // the solve's instantiations are 256 and 512 lanes (edge_launch_setup clamps EBO_EDGE_BLOCK)
	}
	const bool narrow = L.block <= 256;
	SolveKern kern = L.alias_lds ? (narrow ? k_solve_edge<true, 256> : k_solve_edge<true, 512>)
								 : (narrow ? k_solve_edge<false, 256> : k_solve_edge<false, 512>);
	if (allow_big_lds(kern, L.lds_bytes))
	{
		return -2;
	}
	hipLaunchKernelGGL(kern, dim3(L.n_units), dim3(L.block), L.lds_bytes, s, L.d_events, L.d_units, L.cap_px,
					   L.d_scratch, L.scratch_stride, d_flows_out, d_stats, L.c, L.ec, o, ab_env("EBO_SOLVE_NO_REUSE") ? 1 : 0);
	return check_launch();
}

int launch_dump_image(const EvalLaunch& L, int unit, double* d_image, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	auto kern = (L.channels == 3) ? k_dump_image<3> : k_dump_image<1>;
	if (allow_big_lds(kern, L.lds_bytes))
	{
		return -2;
	}
	hipLaunchKernelGGL(kern, dim3(1), dim3(L.block), L.lds_bytes, s, L.d_events, L.d_units, unit,
					   L.d_flows, L.tiles, d_image, L.c);
	return check_launch();
}

int launch_solve_independent(const SolveLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (L.n_units == 0)
	{
		return 0;
	}
	const bool smallExp = L.c.inv_sigsq <= 1.0;
#ifdef EBO_AB
	auto kern = (L.impl == 1)   ? k_solve_independent<1>
				: (L.impl == 2) ? k_solve_independent<2>
				: smallExp		? k_solve_independent<3>
								: k_solve_independent<4>;
#else
	auto kern = smallExp ? k_solve_independent<3> : k_solve_independent<4>;
#endif
	if (allow_big_lds(kern, L.lds_bytes))
	{
		return -2;
	}
	hipLaunchKernelGGL(kern, dim3(L.n_units), dim3(L.block), L.lds_bytes, s, L.d_events,
					   L.d_units, L.cap_doubles, L.d_flows_out, L.d_stats, L.c, L.s, ab_env("EBO_SOLVE_NO_REUSE") ? 1 : 0);
	return check_launch();
}

int launch_count_image(const CountLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	const size_t n = static_cast<size_t>(L.n_windows) * L.c.image_w * L.c.image_h;
	// Unit waves over 2-D tiles (impl 5, k_count_tiles): the default for images warped by per-patch
	// flows (mode 1) that do not fit one workgroup's counters, in launches with enough workgroups.
	// Tile grid: the split (counters + unit headers <= 76 KB: two workgroups per CU, so that one's
	// store phase overlaps the other's event phase) that minimises the expected number of tiles a
	// unit visits.
	if ((L.impl == 5 || (L.impl < 0 && L.mode == 1)) && L.mode == 1 && L.d_unit_maxdt && L.n_units_total > 0)
	{
		const int b = 2;  // 16-bit counters; a workgroup that cannot prove them safe counts its tile in two 32-bit halves
		const int Pn = L.c.npx * L.c.npy;
		const size_t ctlBytes = static_cast<size_t>(Pn + 1) * 16 + 16;  // one 16-byte header per unit the tile may select
		// 76 KB per workgroup (two 1024-lane workgroups per CU) -- except for small sensors, whose whole counter
		// image is little more than that: there four 512-lane workgroups of <= 38 KB per CU overlap their
		// select / wait / store phases better than two large ones (C2, 240x180: 58.8 -> 60.7 % of 8 TB/s; the
		// same split costs C3 and C4 7-10 points: their units straddle the smaller tiles' borders)
		const bool smallSensor = static_cast<size_t>(L.c.image_w) * L.c.image_h * b <= 100 * 1024;
		const size_t budgetAll = static_cast<size_t>(L.lds_kb > 0 ? L.lds_kb : (smallSensor ? 38 : 76)) * 1024;
		const size_t budget = budgetAll > ctlBytes + 4096 ? budgetAll - ctlBytes : 4096;
		const int W = L.c.image_w, H = L.c.image_h;
		int bestX = 0, bestY = 0, bestW = 0, bestH = 0;
		size_t bestBytes = 0;
		double bestCost = 1e300;
		// EQUAL tiles (balance beats alignment to the patch grid: whole-patch tiles with a larger last
		// tile measured 3-4 points of HBM fraction worse at C3 and C4), even width (two 16-bit
		// counters of a dword never straddle rows), cost = expected tiles a unit visits
		const int pw = L.c.patch_w, ph = L.c.patch_h;
		int coarseTiles = 1 << 30;
		for (int tx = 1; tx <= 16 && ctlBytes <= 48 * 1024; ++tx)
		{
			int tw = (W + tx - 1) / tx;
			tw += tw & 1;
			if (tx > 1 && tw * (tx - 1) >= W)
			{
				continue;  // a coarser split covers the image with the same tile width
			}
			const int thMax = static_cast<int>(std::min<size_t>(budget / (static_cast<size_t>(tw) * b), static_cast<size_t>(H)));
			if (thMax < 8)
			{
				continue;
			}
			const int tyMin = (H + thMax - 1) / thMax;
			coarseTiles = std::min(coarseTiles, tx * tyMin);
			const double Rx = 0.5 * pw + 12.0, Ry = 0.5 * ph + 12.0;
			// more, smaller tiles than the LDS asks for when the launch would not fill the chip (two
			// 1024-lane workgroups per CU = 512): the visits a finer split adds against the CUs it wakes
			for (int ty = tyMin; ty <= H / 8; ty = (ty < 4 ? ty + 1 : ty * 2))
			{
				const int th = (H + ty - 1) / ty;
				const double visits = (tx > 1 ? (tw + 2 * Rx) / tw : 1.0) * (ty > 1 ? (th + 2 * Ry) / th : 1.0);
				const double idle = std::max(1.0, 512.0 / (static_cast<double>(L.n_windows) * tx * ty));
				const double cost = visits * idle;
				if (cost < bestCost - 1e-9)
				{
					bestCost = cost;
					bestX = tx;
					bestY = ty;
					bestW = tw;
					bestH = th;
					bestBytes = (static_cast<size_t>(tw) * th * b + 15) & ~size_t(15);
				}
				if (idle <= 1.0)
				{
					break;  // the chip is full: finer only costs visits
				}
			}
		}
		// eligibility as before the finer splits existed: the coarsest split the LDS allows must already
		// give the launch 64 workgroups (single windows and tiny batches stay with impl 0 / 1)
		// (A/B build: force the tile pitch, e.g. whole patch rows / columns)
		if (const size_t fw = ab_size("EBO_COUNT_TILE_W", 0), fh = ab_size("EBO_COUNT_TILE_H", 0); fw > 0 && fh > 0)
		{
			bestW = static_cast<int>(fw);
			bestH = static_cast<int>(fh);
			bestX = (W + bestW - 1) / bestW;
			bestY = (H + bestH - 1) / bestH;
			// the last tile of a row / column takes what is left: size the counters for the largest tile
			const int lastW = W - (bestX - 1) * bestW, lastH = H - (bestY - 1) * bestH;
			bestBytes = (static_cast<size_t>(std::max(bestW, lastW) + 1) * std::max(bestH, lastH) * b + 15) & ~size_t(15);
		}
		const long coarse = static_cast<long>(L.n_windows) * coarseTiles;
		if (bestX > 0 && (L.impl == 5 || (coarseTiles > 1 && coarse >= 64)))
		{
			auto kern = k_count_tiles;
			const size_t lds = bestBytes + ctlBytes;
			if (lds <= 160 * 1024 && allow_big_lds(kern, lds) == 0)
			{
				const int groups = (L.n_windows + 7) / 8;
				const char* be = ab_env("EBO_COUNT_BLOCK");
				const int tileBlock = be && *be ? std::max(64, std::min(1024, (std::atoi(be) / 64) * 64))
												: (budgetAll <= 38 * 1024 ? 512 : 1024);
				hipLaunchKernelGGL(kern, dim3(groups * bestX * bestY * 8), dim3(tileBlock), lds, s, L.d_events, L.d_units,
								   L.d_unit_maxdt, L.units_per_window, static_cast<const double*>(L.d_aux), bestW, bestH, bestX,
								   bestY, static_cast<int>(bestBytes), L.n_windows, L.d_image, L.c);
				if (check_launch())
				{
					return -2;
				}
				if (L.any_stray)
				{
					hipLaunchKernelGGL(k_count_stray, dim3(L.n_windows), dim3(256), 0, s, L.d_events, L.d_units, L.units_per_window,
									   static_cast<const double*>(L.d_aux), L.d_image, L.c);
				}
				return check_launch();
			}
		}
	}
	// Unit waves (impl 4, k_count_units): the default for images warped by per-patch flows that
	// need SEVERAL bands, in launches with enough (band, window) workgroups -- C3 x 128 windows
	// 0.179 -> 0.122 ms against impl 1, C4 x 32 0.311 -> 0.231 ms against impl 3; with one band
	// (C2) impl 1 is as fast, small launches are better off with global atomics.
	if ((L.impl == 4 || (L.impl < 0 && L.mode == 1)) && L.d_unit_maxdt && L.n_units_total > 0)
	{
		const bool u16 = L.max_window_events < 65536;
		const int Pn = L.c.npx * L.c.npy;
		const size_t ctlBytes = static_cast<size_t>(Pn + 3) * 4 + 16;
		// 76 KB: two workgroups per CU, so that one's store phase overlaps the other's event phase
		// (C3: 0.122 -> 0.101 ms against 128 KB bands, C4: 0.228 -> 0.211 ms)
		const size_t ldsWant = static_cast<size_t>(L.lds_kb > 0 ? L.lds_kb : 76) * 1024;
		const size_t ldsBytes = std::min(ldsWant, static_cast<size_t>(160 * 1024 - 1024) - std::min(ctlBytes, static_cast<size_t>(64 * 1024)));
		const size_t pxPerBand = u16 ? ldsBytes / 2 : ldsBytes / 4;
		const int rowsMax = static_cast<int>(std::min<size_t>(pxPerBand / L.c.image_w, L.c.image_h));
		if (rowsMax > 0 && ctlBytes <= 64 * 1024)
		{
			// bands of EQUAL height (C2: 86 KB of counters are two bands of 90 rows, not 158 + 22:
			// 0.235 -> 0.176 ms, and better than the single 86 KB band of impl 1, 0.194 ms, which
			// leaves room for one workgroup per CU only)
			const int bands = (L.c.image_h + rowsMax - 1) / rowsMax;
			const int rowsPerBand = (L.c.image_h + bands - 1) / bands;
			const bool want4 = L.impl == 4 || (bands > 1 && bands <= 64 && static_cast<long>(L.n_windows) * bands >= 64);
			auto kern = u16 ? (L.mode == 0	 ? k_count_units<true, 0>
							   : L.mode == 1 ? k_count_units<true, 1>
											 : k_count_units<true, 2>)
							: (L.mode == 0	 ? k_count_units<false, 0>
							   : L.mode == 1 ? k_count_units<false, 1>
											 : k_count_units<false, 2>);
			const size_t lds = ((static_cast<size_t>(rowsPerBand) * L.c.image_w * (u16 ? 2 : 4) + 15) & ~size_t(15)) + ctlBytes;
			if (want4 && lds <= 160 * 1024 && allow_big_lds(kern, lds) == 0)
			{
				const int groups = (L.n_windows + 7) / 8;
				hipLaunchKernelGGL(kern, dim3(groups * bands * 8), dim3(1024), lds, s, L.d_events, L.d_units, L.d_unit_maxdt,
								   L.units_per_window, L.d_aux, rowsPerBand, L.n_windows, L.d_image, L.c);
				return check_launch();
			}
		}
	}
	// LDS-privatised path when the image splits into few row bands and there are
	// enough (band, window) workgroups to occupy the chip; else global int atomics.
	{
		const bool u16 = L.max_window_events < 65536;
		const int Pn = L.c.npx * L.c.npy;
		// unit table behind the counters: flows [P][2] f64 (mode 1) + dt_win [P + 1] i32
		const size_t tblBytes = L.mode == 0 ? 0 : (L.mode == 1 ? static_cast<size_t>(Pn) * 16 : 0) + static_cast<size_t>(Pn + 1) * 4 + 16;
		const size_t ldsWant = static_cast<size_t>(L.lds_kb > 0 ? L.lds_kb : 128) * 1024;
		const size_t ldsBytes = std::min(ldsWant, static_cast<size_t>(160 * 1024 - 1024) - std::min(tblBytes, static_cast<size_t>(96 * 1024)));
		const size_t pxPerBand = u16 ? ldsBytes / 2 : ldsBytes / 4;
		const int rowsPerBand = static_cast<int>(std::min<size_t>(pxPerBand / L.c.image_w, L.c.image_h));
		const int bands = rowsPerBand > 0 ? (L.c.image_h + rowsPerBand - 1) / rowsPerBand : 1 << 30;
		const bool tableFits = tblBytes <= 64 * 1024;  // finer grids: the other implementations
		const bool want = tableFits && (L.impl == 1 || (L.impl < 0 && L.mode != 0 && bands <= 4 && L.n_windows * bands >= 64));
		if (want && rowsPerBand > 0 && L.n_units_total > 0)
		{
			const bool multi = bands > 1 && L.mode == 1;  // float pre-test of the destination row
			auto kern = u16 ? (L.mode == 0	 ? k_count_window_lds<true, 0, false>
							   : L.mode == 1 ? (multi ? k_count_window_lds<true, 1, true> : k_count_window_lds<true, 1, false>)
											 : k_count_window_lds<true, 2, false>)
							: (L.mode == 0	 ? k_count_window_lds<false, 0, false>
							   : L.mode == 1 ? (multi ? k_count_window_lds<false, 1, true> : k_count_window_lds<false, 1, false>)
											 : k_count_window_lds<false, 2, false>);
			const size_t lds = ((static_cast<size_t>(rowsPerBand) * L.c.image_w * (u16 ? 2 : 4) + 15) & ~size_t(15)) + tblBytes;
			if (lds > 160 * 1024)
			{
				return -2;
			}
			if (allow_big_lds(kern, lds))
			{
				return -2;
			}
			const int groups = (L.n_windows + 7) / 8;  // 8 windows (one per XCD) x bands slots each
			hipLaunchKernelGGL(kern, dim3(groups * bands * 8), dim3(1024), lds, s, L.d_events, L.d_units,
							   L.units_per_window, L.d_aux, rowsPerBand, L.n_windows, L.d_image, L.c);
			return check_launch();
		}
	}
	// Sorted bands (impl 3): warped images of sensors too large for the whole-window LDS image.
	// Three passes over the events pay off once the launch holds several million of them
	// (C4 x 32 windows: 0.53 ms against 0.83 ms of global atomics; C4 x 2: 0.064 against 0.058).
	const bool manyEvents = static_cast<size_t>(L.n_windows) * L.max_window_events >= (size_t(8) << 20);
	if ((L.impl == 3 || (L.impl < 0 && manyEvents)) && L.d_sort_bins && L.n_units_total > 0 && L.mode != 0)
	{
		const bool u16 = L.max_window_events < 65536;
		const size_t ldsBytes = static_cast<size_t>(L.lds_kb > 0 ? L.lds_kb : 24) * 1024;
		const size_t rowBytes = static_cast<size_t>(L.c.image_w) * (u16 ? 2 : 4);
		const int rowsPerBand = static_cast<int>(std::min<size_t>(std::max<size_t>(ldsBytes / rowBytes, 1), L.c.image_h));
		const int bands = (L.c.image_h + rowsPerBand - 1) / rowsPerBand;
		const size_t lds = (static_cast<size_t>(rowsPerBand) * rowBytes + 3) & ~size_t(3);
		const int nBins = bands * L.n_windows;
		auto count = u16 ? k_csort_count<true> : k_csort_count<false>;
		if (lds <= 160 * 1024 - 512 && bands <= 4096 && nBins <= L.sort_bins_cap && allow_big_lds(count, lds) == 0)
		{
			if (hipMemsetAsync(L.d_sort_bins, 0, static_cast<size_t>(nBins) * sizeof(unsigned int), s) != hipSuccess)
			{
				return -2;
			}
			const unsigned chunks = static_cast<unsigned>((L.max_window_events + kSortChunk - 1) / kSortChunk);
			const dim3 grid(std::max(chunks, 1u), L.n_windows);
			const size_t ldsHist = static_cast<size_t>(bands) * sizeof(unsigned int);
			unsigned int* dstList = L.d_sorted + L.sorted_cap;  // second half of the list buffer
			const size_t ldsScatter = 4 * ldsHist + kSortChunk * (sizeof(unsigned int) + sizeof(unsigned short));
			if (L.mode == 1)
			{
				hipLaunchKernelGGL(k_csort_hist<1>, grid, dim3(256), ldsHist, s, L.d_events, L.d_units, L.units_per_window,
								   L.d_aux, rowsPerBand, bands, L.d_sort_bins, dstList, L.c);
			}
			else
			{
				hipLaunchKernelGGL(k_csort_hist<2>, grid, dim3(256), ldsHist, s, L.d_events, L.d_units, L.units_per_window,
								   L.d_aux, rowsPerBand, bands, L.d_sort_bins, dstList, L.c);
			}
			hipLaunchKernelGGL(k_csort_scan, dim3(1), dim3(1024), 0, s, L.d_sort_bins, nBins);
			hipLaunchKernelGGL(k_csort_scatter, grid, dim3(256), ldsScatter, s, L.d_units, L.units_per_window, dstList,
							   rowsPerBand, bands, L.d_sort_bins, nBins, L.d_sorted, L.c);
			hipLaunchKernelGGL(count, dim3(bands, L.n_windows), dim3(512), lds, s, L.d_sort_bins, nBins, L.d_sorted,
							   rowsPerBand, bands, L.d_image, L.c);
			return check_launch();
		}
	}
	// Patch-row bands (impl 2): any image size, no event read twice.  The default for the
	// un-warped image (events never leave their band: 4.3 TB/s at C2, 3.9 at C3, 2.7 at C4);
	// with warping the events that leave a band cost a random HBM access each, which loses
	// against the paths above unless the flows are small (selectable, EBO_COUNT_IMPL=2).
	if ((L.impl == 2 || (L.impl < 0 && L.mode == 0)) && L.d_overflow && L.n_units_total > 0)
	{
		const bool u16 = L.max_window_events < 65536;
		const size_t rowBytes = static_cast<size_t>(L.c.image_w) * (u16 ? 2 : 4);
		size_t ldsBytes = static_cast<size_t>(L.lds_kb > 0 ? L.lds_kb : (L.mode == 0 ? 24 : 76)) * 1024;
		if (L.lds_kb <= 0 && L.mode == 0)
		{
			// un-warped image, measured: ONE patch row per band when it is up to 24 KB of counters
			// (C2 10.5 KB: 63 -> 67 % of HBM against two rows; C3 22 KB: 71 %), column tiles of
			// ~16 KB above that (C4: 64 -> 68 % against 24 KB tiles); several rows only when a
			// patch row is tiny
			const size_t patchRowBytes = static_cast<size_t>(L.c.patch_h) * rowBytes;
			ldsBytes = patchRowBytes > 24 * 1024 ? 16 * 1024 : std::max<size_t>(patchRowBytes, 12 * 1024);
		}
		const int prb = std::max(1, static_cast<int>(ldsBytes / rowBytes) / L.c.patch_h);
		const int bandRows = prb * L.c.patch_h;
		// One patch row already above the target (large sensors; C4: 22 rows x 1280 x 4 B = 112 KB,
		// one workgroup per CU): cut it into column tiles of whole patches.  Un-warped image only --
		// with warping more events would leave a tile than a band.
		int colTiles = 1;
		size_t tileRowBytes = rowBytes;
		if (L.mode == 0 && static_cast<size_t>(L.c.patch_h) * rowBytes > ldsBytes && L.c.npx > 1)
		{
			const size_t pxBytes = u16 ? 2 : 4;
			const int unitsPerTile = std::max<int>(1, static_cast<int>(ldsBytes / (static_cast<size_t>(bandRows) * pxBytes)) / L.c.patch_w);
			colTiles = (L.c.npx + unitsPerTile - 1) / unitsPerTile;
			const int perTile = (L.c.npx + colTiles - 1) / colTiles;  // as the kernel divides them
			const int lastLo = std::min((colTiles - 1) * perTile, L.c.npx - 1);
			const int widest = std::max(perTile * L.c.patch_w, L.c.image_w - lastLo * L.c.patch_w);
			tileRowBytes = static_cast<size_t>(widest) * pxBytes;
		}
		if (const char* v = ab_env("EBO_COUNT_COLTILES"))  // A/B: 1 = off
		{
			if (std::atoi(v) == 1)
			{
				colTiles = 1;
				tileRowBytes = rowBytes;
			}
		}
		const size_t lds = (static_cast<size_t>(bandRows) * tileRowBytes + 3) & ~size_t(3);
		auto kern = u16 ? (L.mode == 0	 ? k_count_bands<true, 0>
						   : L.mode == 1 ? k_count_bands<true, 1>
										 : k_count_bands<true, 2>)
						: (L.mode == 0	 ? k_count_bands<false, 0>
						   : L.mode == 1 ? k_count_bands<false, 1>
										 : k_count_bands<false, 2>);
		if (lds <= 160 * 1024 - 512 && allow_big_lds(kern, lds) == 0)
		{
			const int nRegular = (L.c.npy - 1 + prb - 1) / prb;
			const int tallest = L.c.image_h - (L.c.npy - 1) * L.c.patch_h;
			const int nSub = (tallest + bandRows - 1) / bandRows;
			if (L.mode != 0 && hipMemsetAsync(L.d_overflow, 0, 8, s) != hipSuccess)
			{
				return -2;
			}
			hipLaunchKernelGGL(kern, dim3((nRegular + nSub) * colTiles, L.n_windows), dim3(512), lds, s, L.d_events,
							   L.d_units, L.units_per_window, L.d_aux, prb, nRegular, colTiles, L.d_image, L.d_overflow, L.c);
			if (L.mode != 0)
			{
				hipLaunchKernelGGL(k_count_overflow, dim3(512), dim3(256), 0, s, L.d_overflow, L.d_image);
			}
			return check_launch();
		}
	}
	if (L.n_units_total > 0)
	{
		hipLaunchKernelGGL(k_count_scatter, dim3(L.n_units_total), dim3(256), 0, s, L.d_events,
						   L.d_units, L.units_per_window, L.mode, L.d_aux, L.d_counts, L.c);
		if (check_launch())
		{
			return -2;
		}
	}
	const int blocks = static_cast<int>(std::min<size_t>((n + 255) / 256, 2048));
	hipLaunchKernelGGL(k_counts_to_f64, dim3(blocks), dim3(256), 0, s, L.d_counts, L.d_image, n);
	return check_launch();
}

int launch_route(const RouteLaunch& L, void* stream)
{
	if (L.n_patches == 0)
	{
		return 0;
	}
	hipLaunchKernelGGL(k_route, dim3(L.n_patches), dim3(64), 0, static_cast<hipStream_t>(stream), L.d_xy, L.n_events,
					   L.d_rects, L.d_start, L.d_take, L.cap, L.d_index, L.d_count, L.d_next);
	return check_launch();
}

int launch_patch_integrate(const PatchIntLaunch& L, void* stream)
{
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (L.n_patches == 0)
	{
		return 0;
	}
	const size_t lds = 64 * 1024;  // tiles up to 16384 pixels (default patch is 25x25)
	hipLaunchKernelGGL(k_patch_integrate, dim3(L.n_patches), dim3(256), lds, s, L.d_events,
					   L.d_offsets, L.d_rects, L.d_traj, L.d_nabla_off, L.d_nabla);
	return check_launch();
}

}  // namespace ebo

#ifdef EBO_EDGE_TIMING
// Phase clocks (instrumented build only): the 64 rows of the device table summed into out32.
extern "C" int ebo_debug_edge_clocks(unsigned long long* out32, int reset)
{
	static unsigned long long rows[64 * 32];
	if (hipDeviceSynchronize() != hipSuccess)
	{
		return -1;
	}
	if (out32)
	{
		if (hipMemcpyFromSymbol(rows, HIP_SYMBOL(ebo::g_edge_clk), sizeof(rows)) != hipSuccess)
		{
			return -1;
		}
		for (int k = 0; k < 32; ++k)
		{
			out32[k] = 0;
			for (int r = 0; r < 64; ++r)
			{
				out32[k] += rows[r * 32 + k];
			}
		}
	}
	if (reset)
	{
		for (auto& v : rows)
		{
			v = 0;
		}
		if (hipMemcpyToSymbol(HIP_SYMBOL(ebo::g_edge_clk), rows, sizeof(rows)) != hipSuccess)
		{
			return -1;
		}
	}
	return 0;
}
#endif
