// oracle.cpp — CPU ORACLE (test infrastructure only; see ebo_oracle.h header).
//
// Single-threaded restatement of the reference's motion-compensation path.  The
// reference is single-threaded too.  All arithmetic is IEEE
// double with one rounding per source-level operation: build with
// -ffp-contract=off so that no multiply-add is fused behind the source's back.
//
// Reference paths are relative to the reference checkout root.
#include "ebo_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace
{
// ---------------------------------------------------------------------------
// Dual number with two derivative slots.  Follows the algebra that
// ceres::Jet<double,2> publishes (ceres/jet.h, v2.0): products/quotients by the
// product/quotient rule, quotient computed through the reciprocal of the
// denominator's scalar part, exp/sqrt by the chain rule, comparisons on the
// scalar part only.  This is what AutoDiffCostFunction<contrastFunctor,1,2>
// instantiates the functor with (feature_detector.cpp:359-363).
// ---------------------------------------------------------------------------
struct Dual2
{
	double a;
	double d0;
	double d1;
	Dual2() : a(0), d0(0), d1(0) {}
	explicit Dual2(double s) : a(s), d0(0), d1(0) {}
	Dual2(double s, double e0, double e1) : a(s), d0(e0), d1(e1) {}
};

inline Dual2 operator+(const Dual2& f, const Dual2& g)
{
	return Dual2(f.a + g.a, f.d0 + g.d0, f.d1 + g.d1);
}
inline Dual2 operator-(const Dual2& f, const Dual2& g)
{
	return Dual2(f.a - g.a, f.d0 - g.d0, f.d1 - g.d1);
}
inline Dual2 operator-(const Dual2& f)
{
	return Dual2(-f.a, -f.d0, -f.d1);
}
inline Dual2 operator*(const Dual2& f, const Dual2& g)
{
	return Dual2(f.a * g.a, f.a * g.d0 + f.d0 * g.a, f.a * g.d1 + f.d1 * g.a);
}
inline Dual2 operator/(const Dual2& f, const Dual2& g)
{
	const double gInv = 1.0 / g.a;
	const double q = f.a * gInv;
	return Dual2(q, (f.d0 - q * g.d0) * gInv, (f.d1 - q * g.d1) * gInv);
}
inline Dual2& operator+=(Dual2& f, const Dual2& g)
{
	f = f + g;
	return f;
}
inline Dual2& operator-=(Dual2& f, const Dual2& g)
{
	f = f - g;
	return f;
}
inline bool operator>(const Dual2& f, const Dual2& g) { return f.a > g.a; }
inline bool operator<=(const Dual2& f, const Dual2& g) { return f.a <= g.a; }

inline double expT(double x) { return std::exp(x); }
inline Dual2 expT(const Dual2& f)
{
	const double e = std::exp(f.a);
	return Dual2(e, e * f.d0, e * f.d1);
}
inline double sqrtT(double x) { return std::sqrt(x); }
inline Dual2 sqrtT(const Dual2& f)
{
	const double s = std::sqrt(f.a);
	const double twoInv = 1.0 / (2.0 * s);
	return Dual2(s, f.d0 * twoInv, f.d1 * twoInv);
}
inline double scalarOf(double x) { return x; }
inline double scalarOf(const Dual2& f) { return f.a; }

struct RectI
{
	int x, y, w, h;
};

// contrast_functor.h:18-20 — int32 truncation of the mean of the first and last
// event time.  Values outside int32 are undefined behaviour in
// the reference; recordings it targets are < 2^31 us long.
int64_t midTimestamp(int64_t frontUs, int64_t backUs)
{
	const double half = static_cast<double>(frontUs + backUs) * 0.5;
	return static_cast<int64_t>(static_cast<int32_t>(half));
}

// contrast_functor.h:90-99, same association order.
template <class T>
T gaussianT(const T& meanX, const T& meanY, const T& x, const T& y, double sigma)
{
	const T sigmaSq = T(sigma * sigma);
	const T normCoef = T(1.0) / (T(2 * M_PI) * sigmaSq);
	return normCoef * expT(T(-0.5) / sigmaSq *
						   ((x - meanX) * (x - meanX) + (y - meanY) * (y - meanY)));
}

// The reference converts the warped coordinate with int(double).  Beyond the int
// range that conversion is undefined; every such event is far outside the
// 3W x 3H window anyway, so it is skipped.  2^30 keeps all later int sums exact.
inline bool coordConvertible(double c) { return std::fabs(c) < 1073741824.0; }

// contrast_functor.h:38-88.  img is row-major [3h][3w] (the reference's
// Eigen::Array is indexed (row = y, col = x); storage order is irrelevant here).
template <class T>
void splatEvents(const orc_event* ev, size_t n, const RectI& r, double scale,
				 int64_t tRef, const orc_functor_consts& k, const T* motion,
				 std::vector<T>& img)
{
	const int W3 = 3 * r.w;
	const int H3 = 3 * r.h;
	const int K = k.kernel_compensate;
	for (size_t e = 0; e < n; ++e)
	{
		const double tau = static_cast<double>(tRef - ev[e].t_us) * scale;
		const T cx = T(static_cast<double>(ev[e].x)) + T(tau) * motion[0];
		const T cy = T(static_cast<double>(ev[e].y)) + T(tau) * motion[1];
		if (!coordConvertible(scalarOf(cx)) || !coordConvertible(scalarOf(cy)))
		{
			continue;
		}
		const int bx = static_cast<int>(scalarOf(cx));  // truncation, F7
		const int by = static_cast<int>(scalarOf(cy));
		for (int i = -K; i <= K; ++i)
		{
			for (int j = -K; j <= K; ++j)
			{
				const int px = bx + i - r.x + r.w;
				const int py = by + j - r.y + r.h;
				if (px >= 0 && px < W3 && py >= 0 && py < H3)
				{
					img[static_cast<size_t>(py) * W3 + px] +=
						gaussianT(cx, cy, T(static_cast<double>(bx + i)),
								  T(static_cast<double>(by + j)),
								  k.sigma_compensate);
				}
			}
		}
	}
}

// contrast_functor.h:101-150.
template <class T>
T varianceLoss(const T* motion, const std::vector<T>& img, int W3, int H3,
			   const orc_functor_consts& k)
{
	T mean = T(0.0);
	int counterNonZero = 1;
	for (int i = 0; i < H3; ++i)
	{
		for (int j = 0; j < W3; ++j)
		{
			const T& v = img[static_cast<size_t>(i) * W3 + j];
			if (v > T(0.0))
			{
				mean += v;
				counterNonZero++;
			}
		}
	}
	mean = mean / T(static_cast<double>(counterNonZero));

	if (mean > T(0.0))
	{
		T acc = T(0.0);
		for (int i = 0; i < H3; ++i)
		{
			for (int j = 0; j < W3; ++j)
			{
				const T& v = img[static_cast<size_t>(i) * W3 + j];
				if (v > T(0.0))
				{
					acc += (v - mean) * (v - mean);
				}
			}
		}
		acc = acc / T(static_cast<double>(counterNonZero));
		return T(k.max_possible_residual) - acc;
	}
	return T(k.max_possible_residual) *
		   (T(1.0) + (motion[0] * motion[0]) + (motion[1] * motion[1]));
}

// contrast_functor.h:152-277.  NaN handling is deliberately identical to the
// reference: every comparison is on the scalar part and "NaN > 0" is false.
template <class T>
T edgeLoss(const T* motion, const std::vector<T>& img, int W3, int H3,
		   const orc_functor_consts& k)
{
	// :159 Eigen's mean() = sum / size.  Summed here in column-major storage
	// order (Eigen's default layout); Eigen's vectorised reduction order for
	// doubles is not reproduced (affects only the last bits of a 1e-4 threshold).
	T sum = T(0.0);
	for (int x = 0; x < W3; ++x)
	{
		for (int y = 0; y < H3; ++y)
		{
			sum += img[static_cast<size_t>(y) * W3 + x];
		}
	}
	const T meanAll = sum / T(static_cast<double>(static_cast<long>(W3) * H3));
	if (meanAll <= T(0.0001))
	{
		return T(k.max_possible_residual) *
			   (T(1.0) + (motion[0] * motion[0]) + (motion[1] * motion[1]));
	}

	T residual = T(k.max_possible_residual);
	const size_t npx = static_cast<size_t>(W3) * H3;
	std::vector<T> gradX(npx, T(0.0)), gradY(npx, T(0.0)), firstEig(npx, T(0.0));
	for (int y = 0; y < H3 - 1; ++y)
	{
		for (int x = 0; x < W3 - 1; ++x)
		{
			const size_t p = static_cast<size_t>(y) * W3 + x;
			gradX[p] = img[p + 1] - img[p];
			gradY[p] = img[p + W3] - img[p];
		}
	}

	const int KS = k.kernel_st;
	const int KW = 2 * KS + 1;
	std::vector<T> weight(static_cast<size_t>(KW) * KW, T(0.0));
	for (int i = -KS; i <= KS; ++i)
	{
		for (int j = -KS; j <= KS; ++j)
		{
			weight[static_cast<size_t>(i + KS) * KW + (j + KS)] =
				gaussianT(T(0.0), T(0.0), T(static_cast<double>(j)),
						  T(static_cast<double>(i)), k.sigma_st);
		}
	}

	for (int y = 0; y < H3 - 1; ++y)
	{
		for (int x = 0; x < W3 - 1; ++x)
		{
			T s00 = T(0.0), s01 = T(0.0), s11 = T(0.0), s10 = T(0.0);
			for (int i = -KS; i <= KS; ++i)
			{
				for (int j = -KS; j <= KS; ++j)
				{
					if (x + j >= 0 && x + j < W3 - 1 && y + i >= 0 &&
						y + i < H3 - 1)
					{
						const T& w = weight[static_cast<size_t>(i + KS) * KW + (j + KS)];
						const size_t q = static_cast<size_t>(y + i) * W3 + (x + j);
						s00 += w * gradX[q] * gradX[q];
						s01 += w * gradX[q] * gradY[q];
						s11 += w * gradY[q] * gradY[q];
						s10 += w * gradX[q] * gradY[q];
					}
				}
			}
			const T trace = s00 + s11;
			const T det = s00 * s11 - s01 * s10;
			const T diffEig = sqrtT(trace * trace - T(4.0) * det);
			const T first = T(0.5) * (trace + diffEig);
			if (first > T(0.0))
			{
				firstEig[static_cast<size_t>(y) * W3 + x] = first;
			}
		}
	}

	const int KN = k.kernel_nms;
	for (int y = KN; y < H3 - 1 - KN; y += KN)
	{
		for (int x = KN; x < W3 - 1 - KN; x += KN)
		{
			T maxVal = T(0.0);
			for (int i = -KN; i <= KN; ++i)
			{
				for (int j = -KN; j <= KN; ++j)
				{
					const T& v = firstEig[static_cast<size_t>(y + i) * W3 + (x + j)];
					if (v > maxVal)
					{
						maxVal = v;
					}
				}
			}
			if (maxVal > T(0.0))
			{
				residual -= maxVal / T(k.max_possible_residual);
			}
		}
	}
	return residual;
}

// contrast_functor.h:23-36 with the loss selectable (reference calls the edge
// loss; the variance call is commented out at :33).
template <class T>
T functorEval(const orc_event* ev, size_t n, const RectI& r, double scale,
			  const orc_functor_consts& k, int loss, const T* motion)
{
	const int W3 = 3 * r.w;
	const int H3 = 3 * r.h;
	std::vector<T> img(static_cast<size_t>(W3) * H3, T(0.0));
	const int64_t tRef = midTimestamp(ev[0].t_us, ev[n - 1].t_us);
	splatEvents(ev, n, r, scale, tRef, k, motion, img);
	if (loss == 1)
	{
		return varianceLoss(motion, img, W3, H3, k);
	}
	return edgeLoss(motion, img, W3, H3, k);
}

void evalPatch(const orc_event* ev, size_t n, const RectI& r, double scale,
			   const orc_functor_consts& k, int loss, const double* motion,
			   double* residual, double* jac)
{
	if (jac)
	{
		const Dual2 m[2] = {Dual2(motion[0], 1.0, 0.0), Dual2(motion[1], 0.0, 1.0)};
		const Dual2 out = functorEval<Dual2>(ev, n, r, scale, k, loss, m);
		*residual = out.a;
		jac[0] = out.d0;
		jac[1] = out.d1;
	}
	else
	{
		*residual = functorEval<double>(ev, n, r, scale, k, loss, motion);
	}
}

// --------------------------- grid + bucketing ------------------------------
struct Grid
{
	int npx, npy;
};

Grid gridOf(const orc_params& p)  // feature_detector.cpp:301-304
{
	Grid g;
	g.npx = p.image_w / p.patch_w;
	g.npy = p.image_h / p.patch_h;
	return g;
}

RectI rectOf(const orc_params& p, const Grid& g, int x, int y)  // :332-346
{
	RectI r;
	r.x = x * p.patch_w;
	r.y = y * p.patch_h;
	r.w = p.patch_w;
	r.h = p.patch_h;
	if (x == g.npx - 1)
	{
		r.w = p.image_w - x * p.patch_w;
	}
	if (y == g.npy - 1)
	{
		r.h = p.image_h - y * p.patch_h;
	}
	return r;
}

inline bool rectContains(const RectI& r, int x, int y)  // cv::Rect_::contains
{
	return r.x <= x && x < r.x + r.w && r.y <= y && y < r.y + r.h;
}

struct Window
{
	orc_params prm;
	Grid g;
	std::vector<RectI> rects;
	std::vector<std::vector<orc_event>> bucket;  // per patch, list order
};

// feature_detector.cpp:348-355: every patch scans ALL events (O(P*N)) and keeps
// those its rect contains, in list order.  Restated literally.
void buildWindow(const orc_event* ev, size_t n, const orc_params& p, Window& w)
{
	w.prm = p;
	w.g = gridOf(p);
	const int P = w.g.npx * w.g.npy;
	w.rects.resize(P);
	w.bucket.assign(P, {});
	for (int y = 0; y < w.g.npy; ++y)
	{
		for (int x = 0; x < w.g.npx; ++x)
		{
			const int idx = y * w.g.npx + x;
			w.rects[idx] = rectOf(p, w.g, x, y);
			for (size_t e = 0; e < n; ++e)
			{
				if (rectContains(w.rects[idx], ev[e].x, ev[e].y))
				{
					w.bucket[idx].push_back(ev[e]);
				}
			}
		}
	}
}

inline bool patchActive(const Window& w, int p)  // :357 strictly greater
{
	return w.bucket[p].size() > w.prm.min_events;
}

// ------------------------------ least squares -------------------------------
// A residual block as ceres::Problem holds it (feature_detector.cpp:365-396).
struct Block
{
	int type;  // 0 = contrast data term on patch p; 1 = TV between p and q
	int p, q;
};

// ceres::HuberLoss(a): rho(s) = s (s <= a^2) else 2a*sqrt(s) - a^2.
inline void huber(double a, double s, double rho[3])
{
	const double b = a * a;
	if (s > b)
	{
		const double r = std::sqrt(s);
		rho[0] = 2.0 * a * r - b;
		rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
		rho[2] = -rho[1] / (2.0 * s);
	}
	else
	{
		rho[0] = s;
		rho[1] = 1.0;
		rho[2] = 0.0;
	}
}

// total_variance.h:14-20 with Jet semantics: abs(f) = f.a < 0 ? -f : f.
void tvEval(double w, const double* x, const double* y, double* r, double* jx,
			double* jy)
{
	for (int c = 0; c < 2; ++c)
	{
		const double d = x[c] - y[c];
		const double sgn = (d < 0.0) ? -1.0 : 1.0;
		r[c] = w * (sgn * d);
		if (jx)
		{
			jx[c * 2 + 0] = 0.0;
			jx[c * 2 + 1] = 0.0;
			jy[c * 2 + 0] = 0.0;
			jy[c * 2 + 1] = 0.0;
			jx[c * 2 + c] = w * sgn;
			jy[c * 2 + c] = -(w * sgn);
		}
	}
}

// One row of a block-sparse Jacobian: at most 4 (column, value) entries.
struct JRow
{
	int c[4];
	double v[4];
	int nnz;
};

struct Problem
{
	const Window* win = nullptr;
	std::vector<Block> blocks;
	std::vector<int> col;  // param index (0..2P) -> column in reduced problem or -1
	int ncols = 0;
	int nrows = 0;
	std::vector<int> rowStart;  // per block
	// statistics
	int evalsCost = 0;
	int evalsJac = 0;

	void finalize(int P)
	{
		col.assign(2 * P, -1);
		for (const Block& b : blocks)
		{
			col[2 * b.p] = 0;
			col[2 * b.p + 1] = 0;
			if (b.type == 1)
			{
				col[2 * b.q] = 0;
				col[2 * b.q + 1] = 0;
			}
		}
		ncols = 0;
		for (int i = 0; i < 2 * P; ++i)
		{
			if (col[i] == 0)
			{
				col[i] = ncols++;
			}
		}
		nrows = 0;
		rowStart.clear();
		for (const Block& b : blocks)
		{
			rowStart.push_back(nrows);
			nrows += (b.type == 0) ? 1 : 2;
		}
	}

	// Evaluates cost = 1/2 sum rho(|f_i|^2), loss-corrected residuals and (when
	// J != nullptr) the loss-corrected Jacobian as dense rows x ncols... kept
	// block-sparse: each row stores 4 (col,val) slots.
	using Row = JRow;

	void evaluate(const double* xFull, double* cost, std::vector<double>* res,
				  std::vector<Row>* jac)
	{
		const orc_params& prm = win->prm;
		double c = 0.0;
		if (res)
		{
			res->assign(nrows, 0.0);
		}
		if (jac)
		{
			jac->assign(nrows, Row());
		}
		for (size_t bi = 0; bi < blocks.size(); ++bi)
		{
			const Block& b = blocks[bi];
			const int r0 = rowStart[bi];
			if (b.type == 0)
			{
				double r, j[2];
				const auto& evs = win->bucket[b.p];
				evalPatch(evs.data(), evs.size(), win->rects[b.p], prm.scale,
						  prm.k, prm.loss, xFull + 2 * b.p, &r, jac ? j : nullptr);
				if (jac)
				{
					evalsJac++;
				}
				else
				{
					evalsCost++;
				}
				c += 0.5 * r * r;
				if (res)
				{
					(*res)[r0] = r;
				}
				if (jac)
				{
					Row& row = (*jac)[r0];
					row.nnz = 2;
					row.c[0] = col[2 * b.p];
					row.c[1] = col[2 * b.p + 1];
					row.v[0] = j[0];
					row.v[1] = j[1];
				}
			}
			else
			{
				double r[2], jx[4], jy[4];
				tvEval(prm.tv_weight, xFull + 2 * b.p, xFull + 2 * b.q, r,
					   jac ? jx : nullptr, jac ? jy : nullptr);
				const double s = r[0] * r[0] + r[1] * r[1];
				double rho[3];
				huber(prm.tv_huber, s, rho);
				c += 0.5 * rho[0];
				// ceres Corrector with rho'' <= 0: scale by sqrt(rho').
				const double sr = std::sqrt(rho[1]);
				if (res)
				{
					(*res)[r0] = r[0] * sr;
					(*res)[r0 + 1] = r[1] * sr;
				}
				if (jac)
				{
					for (int k = 0; k < 2; ++k)
					{
						Row& row = (*jac)[r0 + k];
						row.nnz = 4;
						row.c[0] = col[2 * b.p];
						row.c[1] = col[2 * b.p + 1];
						row.c[2] = col[2 * b.q];
						row.c[3] = col[2 * b.q + 1];
						row.v[0] = jx[k * 2 + 0] * sr;
						row.v[1] = jx[k * 2 + 1] * sr;
						row.v[2] = jy[k * 2 + 0] * sr;
						row.v[3] = jy[k * 2 + 1] * sr;
					}
				}
			}
		}
		*cost = c;
	}
};

// Symmetric positive-definite solve by Cholesky, lower band storage:
// entry (i, j), i - band <= j <= i, lives at A[i * (band + 1) + (j - i + band)].
// Returns false when a pivot is not positive.
inline double& bandAt(std::vector<double>& A, int band, int i, int j)
{
	return A[static_cast<size_t>(i) * (band + 1) + (j - i + band)];
}

bool choleskySolve(std::vector<double>& A, int n, int band, std::vector<double>& b)
{
	for (int j = 0; j < n; ++j)
	{
		double d = bandAt(A, band, j, j);
		const int k0 = std::max(0, j - band);
		for (int k = k0; k < j; ++k)
		{
			const double l = bandAt(A, band, j, k);
			d -= l * l;
		}
		if (!(d > 0.0) || !std::isfinite(d))
		{
			return false;
		}
		const double ljj = std::sqrt(d);
		bandAt(A, band, j, j) = ljj;
		const int i1 = std::min(n - 1, j + band);
		for (int i = j + 1; i <= i1; ++i)
		{
			double s = bandAt(A, band, i, j);
			const int kk0 = std::max(0, i - band);
			for (int k = std::max(k0, kk0); k < j; ++k)
			{
				s -= bandAt(A, band, i, k) * bandAt(A, band, j, k);
			}
			bandAt(A, band, i, j) = s / ljj;
		}
	}
	for (int i = 0; i < n; ++i)
	{
		double s = b[i];
		for (int k = std::max(0, i - band); k < i; ++k)
		{
			s -= bandAt(A, band, i, k) * b[k];
		}
		b[i] = s / bandAt(A, band, i, i);
	}
	for (int i = n - 1; i >= 0; --i)
	{
		double s = b[i];
		for (int k = i + 1; k <= std::min(n - 1, i + band); ++k)
		{
			s -= bandAt(A, band, k, i) * b[k];
		}
		b[i] = s / bandAt(A, band, i, i);
	}
	return true;
}

// Trust-region Levenberg-Marquardt as Ceres 2.0 publishes it
// (TrustRegionMinimizer::Minimize, LevenbergMarquardtStrategy,
// TrustRegionStepEvaluator); options as feature_detector.cpp:401-410.
// x is the full [2P] vector; only columns of the reduced problem move.
// Optional per-iteration log in the columns of ceres::Solver's progress table (cost,
// cost_change, |gradient| max norm, |step|, tr_ratio, tr_radius): set by orc_lm_powell to check
// this restatement against the table Ceres' own tutorial publishes.
thread_local std::vector<double>* g_lmTrace = nullptr;

template <class PB>
int minimize(PB& pb, const orc_solver_opts& o, double* x, orc_summary* sum)
{
	const int n = pb.ncols;
	const int nFull = static_cast<int>(pb.col.size());
	orc_summary local;
	std::memset(&local, 0, sizeof(local));
	if (n == 0 || pb.nrows == 0)
	{
		if (sum)
		{
			*sum = local;
		}
		return 0;
	}
	std::vector<int> fullOf(n);
	for (int i = 0; i < nFull; ++i)
	{
		if (pb.col[i] >= 0)
		{
			fullOf[pb.col[i]] = i;
		}
	}

	std::vector<double> xcur(x, x + nFull), xcand(nFull), xbest(x, x + nFull);
	std::vector<double> f, scale(n, 1.0), grad(n), diag(n), lmDiag(n), step(n), delta(n);
	std::vector<JRow> J;
	double xCost = 0.0;

	auto evalJac = [&](void) {
		pb.evaluate(xcur.data(), &xCost, &f, &J);
		std::fill(grad.begin(), grad.end(), 0.0);
		for (int r = 0; r < pb.nrows; ++r)
		{
			for (int k = 0; k < J[r].nnz; ++k)
			{
				grad[J[r].c[k]] += J[r].v[k] * f[r];
			}
		}
	};
	auto scaleJac = [&](void) {
		for (int r = 0; r < pb.nrows; ++r)
		{
			for (int k = 0; k < J[r].nnz; ++k)
			{
				J[r].v[k] *= scale[J[r].c[k]];
			}
		}
	};
	auto normOfActive = [&](const std::vector<double>& v) {
		double s = 0.0;
		for (int c = 0; c < n; ++c)
		{
			s += v[fullOf[c]] * v[fullOf[c]];
		}
		return std::sqrt(s);
	};
	auto maxAbs = [&](const std::vector<double>& v) {
		double m = 0.0;
		for (double e : v)
		{
			m = std::max(m, std::fabs(e));
		}
		return m;
	};

	// Iteration zero.
	evalJac();
	if (!std::isfinite(xCost))
	{
		local.termination = 2;
		if (sum)
		{
			*sum = local;
		}
		return 0;
	}
	local.initial_cost = xCost;
	if (o.jacobi_scaling)
	{
		std::vector<double> cn(n, 0.0);
		for (int r = 0; r < pb.nrows; ++r)
		{
			for (int k = 0; k < J[r].nnz; ++k)
			{
				cn[J[r].c[k]] += J[r].v[k] * J[r].v[k];
			}
		}
		for (int c = 0; c < n; ++c)
		{
			scale[c] = 1.0 / (1.0 + std::sqrt(cn[c]));
		}
	}
	scaleJac();
	double xNorm = normOfActive(xcur);
	double gradMax = maxAbs(grad);
	double minimumCost = xCost;

	// TrustRegionStepEvaluator state.
	const int maxNonmono = o.use_nonmonotonic ? o.max_consecutive_nonmonotonic : 0;
	double seMinimum = xCost, seCurrent = xCost, seReference = xCost, seCandidate = xCost;
	double seAccRef = 0.0, seAccCand = 0.0;
	int seNumNonmono = 0;

	// LevenbergMarquardtStrategy state.
	double radius = o.initial_radius;
	double decreaseFactor = 2.0;
	bool reuseDiagonal = false;

	int iteration = 0;
	int numInvalid = 0;
	// Iteration zero counts as a successful evaluation point: a start that already
	// satisfies the gradient tolerance terminates with CONVERGENCE (Ceres: "Gradient
	// tolerance reached"), e.g. TV terms only at the all-zero start.
	bool lastSuccessful = true;
	int termination = 1;

	// Band of the normal equations.
	int band = 0;
	for (int r = 0; r < pb.nrows; ++r)
	{
		for (int a = 0; a < J[r].nnz; ++a)
		{
			for (int b = 0; b < J[r].nnz; ++b)
			{
				band = std::max(band, std::abs(J[r].c[a] - J[r].c[b]));
			}
		}
	}
	std::vector<double> H(static_cast<size_t>(n) * (band + 1));

	for (;;)
	{
		// FinalizeIterationAndCheckIfMinimizerCanContinue.
		if (lastSuccessful && xCost < minimumCost)
		{
			minimumCost = xCost;
			xbest = xcur;
		}
		if (iteration >= o.max_num_iterations)
		{
			termination = 1;
			break;
		}
		if (lastSuccessful && gradMax <= o.gradient_tolerance)
		{
			termination = 0;
			break;
		}
		if (radius < o.min_radius)
		{
			termination = 0;
			break;
		}
		iteration++;
		lastSuccessful = false;

		// LevenbergMarquardtStrategy::ComputeStep.
		if (!reuseDiagonal)
		{
			std::fill(diag.begin(), diag.end(), 0.0);
			for (int r = 0; r < pb.nrows; ++r)
			{
				for (int k = 0; k < J[r].nnz; ++k)
				{
					diag[J[r].c[k]] += J[r].v[k] * J[r].v[k];
				}
			}
			for (int c = 0; c < n; ++c)
			{
				diag[c] = std::min(std::max(diag[c], o.min_lm_diagonal), o.max_lm_diagonal);
			}
		}
		for (int c = 0; c < n; ++c)
		{
			lmDiag[c] = std::sqrt(diag[c] / radius);
		}
		reuseDiagonal = true;
		std::fill(H.begin(), H.end(), 0.0);
		std::fill(step.begin(), step.end(), 0.0);
		for (int r = 0; r < pb.nrows; ++r)
		{
			for (int a = 0; a < J[r].nnz; ++a)
			{
				step[J[r].c[a]] += J[r].v[a] * f[r];
				for (int b = 0; b < J[r].nnz; ++b)
				{
					if (J[r].c[b] <= J[r].c[a])
					{
						bandAt(H, band, J[r].c[a], J[r].c[b]) += J[r].v[a] * J[r].v[b];
					}
				}
			}
		}
		for (int c = 0; c < n; ++c)
		{
			bandAt(H, band, c, c) += lmDiag[c] * lmDiag[c];
		}
		bool valid = choleskySolve(H, n, band, step);
		if (valid)
		{
			for (int c = 0; c < n; ++c)
			{
				if (!std::isfinite(step[c]))
				{
					valid = false;
				}
				step[c] = -step[c];
			}
		}
		double modelCostChange = 0.0;
		if (valid)
		{
			// -(J s)'(f + J s / 2)
			for (int r = 0; r < pb.nrows; ++r)
			{
				double mr = 0.0;
				for (int k = 0; k < J[r].nnz; ++k)
				{
					mr += J[r].v[k] * step[J[r].c[k]];
				}
				modelCostChange -= mr * (f[r] + mr / 2.0);
			}
			valid = modelCostChange > 0.0;
		}
		if (!valid)
		{
			// HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid.
			numInvalid++;
			if (numInvalid >= o.max_consecutive_invalid)
			{
				termination = 2;
				break;
			}
			radius *= 0.5;
			reuseDiagonal = true;
			continue;
		}
		numInvalid = 0;
		xcand = xcur;
		for (int c = 0; c < n; ++c)
		{
			delta[c] = step[c] * scale[c];
			xcand[fullOf[c]] = xcur[fullOf[c]] + delta[c];
		}
		double candCost = 0.0;
		pb.evaluate(xcand.data(), &candCost, nullptr, nullptr);
		if (!std::isfinite(candCost))
		{
			candCost = std::numeric_limits<double>::max();
		}

		// ParameterToleranceReached.
		double stepNorm = 0.0;
		for (int c = 0; c < n; ++c)
		{
			const double d = xcur[fullOf[c]] - xcand[fullOf[c]];
			stepNorm += d * d;
		}
		stepNorm = std::sqrt(stepNorm);
		if (stepNorm <= o.parameter_tolerance * (xNorm + o.parameter_tolerance))
		{
			termination = 0;
			break;
		}
		// FunctionToleranceReached.
		const double costChange = xCost - candCost;
		if (std::fabs(costChange) <= o.function_tolerance * xCost)
		{
			termination = 0;
			break;
		}
		// TrustRegionStepEvaluator::StepQuality.
		const double relDec = (seCurrent - candCost) / modelCostChange;
		const double histDec = (seReference - candCost) / (seAccRef + modelCostChange);
		const double quality = std::max(relDec, histDec);

		if (quality > o.min_relative_decrease)
		{
			// HandleSuccessfulStep.
			xcur = xcand;
			xNorm = normOfActive(xcur);
			evalJac();
			if (!std::isfinite(xCost))
			{
				termination = 2;
				break;
			}
			scaleJac();
			gradMax = maxAbs(grad);
			lastSuccessful = true;
			// LevenbergMarquardtStrategy::StepAccepted.
			radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * quality - 1.0, 3));
			radius = std::min(o.max_radius, radius);
			if (g_lmTrace)
			{
				const double row[6] = {xCost, costChange, gradMax, stepNorm, quality, radius};
				g_lmTrace->insert(g_lmTrace->end(), row, row + 6);
			}
			decreaseFactor = 2.0;
			reuseDiagonal = false;
			// TrustRegionStepEvaluator::StepAccepted(candidate_cost, model_cost_change).
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
			if (seNumNonmono == maxNonmono)
			{
				seReference = seCandidate;
				seAccRef = seAccCand;
			}
		}
		else
		{
			// LevenbergMarquardtStrategy::StepRejected.
			radius = radius / decreaseFactor;
			decreaseFactor *= 2.0;
			reuseDiagonal = true;
		}
	}

	// The user's parameter array holds the lowest-cost point visited.
	for (int c = 0; c < n; ++c)
	{
		x[fullOf[c]] = xbest[fullOf[c]];
	}
	local.iterations = iteration;
	local.termination = termination;
	local.final_cost = minimumCost;
	local.num_evals_cost = pb.evalsCost;
	local.num_evals_jac = pb.evalsJac;
	if (sum)
	{
		*sum = local;
	}
	return 0;
}

// Powell's singular function as Ceres' tutorial sets it up (docs "Non-linear Least Squares",
// examples/powell.cc): four parameter blocks of size 1, four residual blocks
//   f1 = x1 + 10 x2, f2 = sqrt(5) (x3 - x4), f3 = (x2 - 2 x3)^2, f4 = sqrt(10) (x1 - x4)^2,
// automatic differentiation, no loss.  Used only to pin minimize() against the iteration table
// that tutorial publishes (tests/golden/ceres_tutorial_powell.json).
struct PowellProblem
{
	std::vector<int> col{0, 1, 2, 3};
	int ncols = 4, nrows = 4;
	int evalsCost = 0, evalsJac = 0;
	void evaluate(const double* x, double* cost, std::vector<double>* res, std::vector<JRow>* jac)
	{
		const double d3 = x[1] - 2.0 * x[2], d4 = x[0] - x[3];
		const double s5 = std::sqrt(5.0), s10 = std::sqrt(10.0);
		const double f[4] = {x[0] + 10.0 * x[1], s5 * (x[2] - x[3]), d3 * d3, s10 * d4 * d4};
		*cost = 0.5 * (f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + f[3] * f[3]);
		if (res)
		{
			res->assign(f, f + 4);
		}
		if (jac)
		{
			evalsJac++;
			jac->assign(4, JRow());
			auto set = [&](int r, int c0, double v0, int c1, double v1) {
				JRow& row = (*jac)[r];
				row.nnz = 2;
				row.c[0] = c0;
				row.v[0] = v0;
				row.c[1] = c1;
				row.v[1] = v1;
			};
			set(0, 0, 1.0, 1, 10.0);
			set(1, 2, s5, 3, -s5);
			set(2, 1, d3 + d3, 2, -2.0 * d3 - 2.0 * d3);  // Jet product rule: a v + v a
			set(3, 0, s10 * d4 + s10 * d4, 3, -(s10 * d4) - s10 * d4);
		}
		else
		{
			evalsCost++;
		}
	}
};

// feature_detector.cpp:316-414.
void solveWindow(const Window& w, const orc_solver_opts& o, double* flows,
				 orc_summary* sum)
{
	const int P = w.g.npx * w.g.npy;
	for (int i = 0; i < 2 * P; ++i)
	{
		flows[i] = 0.0;  // :318-326
	}
	orc_summary total;
	std::memset(&total, 0, sizeof(total));
	if (o.mode == 0)
	{
		Problem pb;
		pb.win = &w;
		for (int y = 0; y < w.g.npy; ++y)
		{
			for (int x = 0; x < w.g.npx; ++x)
			{
				const int p = y * w.g.npx + x;
				if (patchActive(w, p))
				{
					pb.blocks.push_back({0, p, -1});
				}
				if (w.prm.tv_weight != 0.0)
				{
					if (x < w.g.npx - 1)
					{
						pb.blocks.push_back({1, p, p + 1});
					}
					if (y < w.g.npy - 1)
					{
						pb.blocks.push_back({1, p, p + w.g.npx});
					}
				}
			}
		}
		pb.finalize(P);
		minimize(pb, o, flows, &total);
	}
	else
	{
		total.termination = 0;
		for (int p = 0; p < P; ++p)
		{
			if (!patchActive(w, p))
			{
				continue;
			}
			Problem pb;
			pb.win = &w;
			pb.blocks.push_back({0, p, -1});
			pb.finalize(P);
			orc_summary s;
			minimize(pb, o, flows, &s);
			total.iterations = std::max(total.iterations, s.iterations);
			total.num_evals_cost += s.num_evals_cost;
			total.num_evals_jac += s.num_evals_jac;
			total.initial_cost += s.initial_cost;
			total.final_cost += s.final_cost;
			total.termination = std::max(total.termination, s.termination);
		}
	}
	if (sum)
	{
		*sum = total;
	}
}

// The per-pixel smoothing problem of FeatureDetector::interpolateMotionField
// (feature_detector.cpp:154-214): one 2-parameter block per pixel, blocks
// totalVarianceFunctor(weight 1) between (y,x)-(y,x+1) and (y,x)-(y+1,x) for y < H-1,
// x < W-1 (:170-204), loss none or HuberLoss(1e-5) (useL1), fixed points constant
// (:206-214).  As ceres::Problem reduces it: constant blocks leave the parameter vector,
// residual blocks between two constants leave the cost, pixel (H-1, W-1) is in no block.
// Columns are component-major over the free pixels in raster order (a block-diagonal
// normal matrix of band <= W); Ceres' own fill-reducing order is not reproducible here,
// so the last bits are unpinned exactly as for the patch solve.
struct FieldProblem
{
	int W = 0, H = 0;
	bool l1 = false;
	std::vector<char> fixed;  // per pixel
	std::vector<int> col;	  // 2 * pixel + c -> column or -1
	std::vector<int> eP, eQ;  // residual blocks kept, creation order
	int ncols = 0, nrows = 0;
	int evalsCost = 0, evalsJac = 0;

	void build(int w, int h, bool useL1, int nFixed, const int32_t* fixedXy)
	{
		W = w;
		H = h;
		l1 = useL1;
		fixed.assign(static_cast<size_t>(W) * H, 0);
		for (int i = 0; i < nFixed; ++i)
		{
			fixed[static_cast<size_t>(fixedXy[2 * i + 1]) * W + fixedXy[2 * i]] = 1;
		}
		std::vector<char> used(static_cast<size_t>(W) * H, 0);
		for (int y = 0; y < H - 1; ++y)
		{
			for (int x = 0; x < W - 1; ++x)
			{
				const int p = y * W + x;
				used[p] = used[p + 1] = used[p + W] = 1;
				if (!(fixed[p] && fixed[p + 1]))
				{
					eP.push_back(p);
					eQ.push_back(p + 1);
				}
				if (!(fixed[p] && fixed[p + W]))
				{
					eP.push_back(p);
					eQ.push_back(p + W);
				}
			}
		}
		int nFree = 0;
		std::vector<int> freeIdx(static_cast<size_t>(W) * H, -1);
		for (int p = 0; p < W * H; ++p)
		{
			if (used[p] && !fixed[p])
			{
				freeIdx[p] = nFree++;
			}
		}
		col.assign(static_cast<size_t>(W) * H * 2, -1);
		for (int p = 0; p < W * H; ++p)
		{
			if (freeIdx[p] >= 0)
			{
				col[2 * p] = freeIdx[p];
				col[2 * p + 1] = freeIdx[p] + nFree;
			}
		}
		ncols = 2 * nFree;
		nrows = 2 * static_cast<int>(eP.size());
	}

	void evaluate(const double* xFull, double* cost, std::vector<double>* res,
				  std::vector<JRow>* jac)
	{
		double c = 0.0;
		if (res)
		{
			res->assign(nrows, 0.0);
		}
		if (jac)
		{
			jac->assign(nrows, JRow());
			evalsJac++;
		}
		else
		{
			evalsCost++;
		}
		for (size_t e = 0; e < eP.size(); ++e)
		{
			const int p = eP[e], q = eQ[e];
			double r[2], jx[4], jy[4];
			tvEval(1.0, xFull + 2 * p, xFull + 2 * q, r, jac ? jx : nullptr, jac ? jy : nullptr);
			const double s = r[0] * r[0] + r[1] * r[1];
			double sr = 1.0;
			if (l1)
			{
				double rho[3];
				huber(1e-5, s, rho);  // :182,187
				c += 0.5 * rho[0];
				sr = std::sqrt(rho[1]);
			}
			else
			{
				c += 0.5 * s;
			}
			for (int k = 0; k < 2; ++k)
			{
				if (res)
				{
					(*res)[2 * e + k] = r[k] * sr;
				}
				if (jac)
				{
					JRow& row = (*jac)[2 * e + k];
					row.nnz = 0;
					if (col[2 * p + k] >= 0)
					{
						row.c[row.nnz] = col[2 * p + k];
						row.v[row.nnz++] = jx[k * 2 + k] * sr;
					}
					if (col[2 * q + k] >= 0)
					{
						row.c[row.nnz] = col[2 * q + k];
						row.v[row.nnz++] = jy[k * 2 + k] * sr;
					}
				}
			}
		}
		*cost = c;
	}
};

// cv::norm(motionField_) > 0 (:152) on the CV_64FC2 matrix whose first half of every row
// holds the float pairs written through at<cv::Vec2f>: each pixel's two floats are read
// back as one double and squared (a pair with a zero second float is a subnormal double
// whose square underflows to zero).
bool fieldNormPositive(const float* field, size_t pixels)
{
	double acc = 0.0;
	for (size_t i = 0; i < pixels; ++i)
	{
		double v;
		std::memcpy(&v, field + 2 * i, sizeof(double));
		acc += v * v;
	}
	return std::sqrt(acc) > 0.0;
}

// feature_detector.cpp:433-463.  round() = half away from zero (F7).
void finalCountImage(const orc_event* ev, size_t n, const orc_params& p,
					 const double* flows, double* image)
{
	const Grid g = gridOf(p);
	std::fill(image, image + static_cast<size_t>(p.image_w) * p.image_h, 0.0);
	if (n == 0)
	{
		return;
	}
	const int64_t tRef = midTimestamp(ev[0].t_us, ev[n - 1].t_us);  // :305-306
	for (size_t e = 0; e < n; ++e)
	{
		int px = std::min(static_cast<int>(ev[e].x / p.patch_w), g.npx - 1);
		int py = std::min(static_cast<int>(ev[e].y / p.patch_h), g.npy - 1);
		// Coordinates at or below -patch size index before mf[0] in the reference
		// (undefined); such events do not occur on a sensor.  Clamped here.
		px = std::max(px, 0);
		py = std::max(py, 0);
		const double* mf = flows + 2 * (py * g.npx + px);
		const double dt = static_cast<double>(tRef - ev[e].t_us);
		const double fx = ev[e].x + dt * p.scale * mf[0];
		const double fy = ev[e].y + dt * p.scale * mf[1];
		if (!coordConvertible(fx) || !coordConvertible(fy))
		{
			continue;
		}
		const int nx = static_cast<int>(std::round(fx));
		const int ny = static_cast<int>(std::round(fy));
		if (nx >= 0 && nx < p.image_w && ny >= 0 && ny < p.image_h)
		{
			image[static_cast<size_t>(ny) * p.image_w + nx] += 1.0;
		}
	}
}

}  // namespace

// ------------------------------- C API --------------------------------------
extern "C" {

void orc_default_consts(orc_functor_consts* k)
{
	k->max_possible_residual = 1e3;
	k->sigma_compensate = 1.0;
	k->kernel_compensate = 3;
	k->kernel_st = 3;
	k->sigma_st = 1.5;
	k->kernel_nms = 2;
	k->reserved = 0;
}

void orc_default_params(orc_params* p)
{
	p->image_w = 240;
	p->image_h = 180;
	p->patch_w = 20;
	p->patch_h = 20;
	p->tv_weight = 1e3;
	p->tv_huber = 10;
	p->scale = 1e-3;
	p->min_events = 100;
	p->loss = 0;
	orc_default_consts(&p->k);
}

void orc_default_solver(orc_solver_opts* o)
{
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
	o->mode = 0;
}

int64_t orc_mid_timestamp(int64_t front_us, int64_t back_us)
{
	return midTimestamp(front_us, back_us);
}

int orc_contrast_eval(const orc_event* ev, size_t n, int rx, int ry, int rw,
					  int rh, double scale, const orc_functor_consts* k,
					  int loss, const double* motion, double* residual,
					  double* jac)
{
	if (!ev || n == 0 || rw <= 0 || rh <= 0 || !k || !motion || !residual)
	{
		return -1;
	}
	const RectI r = {rx, ry, rw, rh};
	evalPatch(ev, n, r, scale, *k, loss, motion, residual, jac);
	return 0;
}

int orc_contrast_image(const orc_event* ev, size_t n, int rx, int ry, int rw,
					   int rh, double scale, const orc_functor_consts* k,
					   const double* motion, int channels, double* img)
{
	if (!ev || n == 0 || rw <= 0 || rh <= 0 || !k || !motion || !img ||
		(channels != 1 && channels != 3))
	{
		return -1;
	}
	const RectI r = {rx, ry, rw, rh};
	const size_t npx = static_cast<size_t>(9) * rw * rh;
	const int64_t tRef = midTimestamp(ev[0].t_us, ev[n - 1].t_us);
	if (channels == 1)
	{
		std::vector<double> im(npx, 0.0);
		splatEvents<double>(ev, n, r, scale, tRef, *k, motion, im);
		std::copy(im.begin(), im.end(), img);
	}
	else
	{
		std::vector<Dual2> im(npx);
		const Dual2 m[2] = {Dual2(motion[0], 1.0, 0.0), Dual2(motion[1], 0.0, 1.0)};
		splatEvents<Dual2>(ev, n, r, scale, tRef, *k, m, im);
		for (size_t i = 0; i < npx; ++i)
		{
			img[i] = im[i].a;
			img[npx + i] = im[i].d0;
			img[2 * npx + i] = im[i].d1;
		}
	}
	return 0;
}

int orc_tv_eval(double weight, const double* x, const double* y, double* r,
				double* jx, double* jy)
{
	if (!x || !y || !r || ((jx == nullptr) != (jy == nullptr)))
	{
		return -1;
	}
	tvEval(weight, x, y, r, jx, jy);
	return 0;
}

int orc_grid(const orc_params* p, int* npx, int* npy)
{
	if (!p || p->patch_w <= 0 || p->patch_h <= 0)
	{
		return -1;
	}
	const Grid g = gridOf(*p);
	*npx = g.npx;
	*npy = g.npy;
	return 0;
}

int orc_patch_rect(const orc_params* p, int px, int py, int* rx, int* ry,
				   int* rw, int* rh)
{
	const Grid g = gridOf(*p);
	if (px < 0 || py < 0 || px >= g.npx || py >= g.npy)
	{
		return -1;
	}
	const RectI r = rectOf(*p, g, px, py);
	*rx = r.x;
	*ry = r.y;
	*rw = r.w;
	*rh = r.h;
	return 0;
}

int orc_window_eval(const orc_event* ev, size_t n, const orc_params* p,
					const double* flows, double* r, double* jac,
					int32_t* active, int32_t* counts)
{
	if (!ev || !p || !flows || !r)
	{
		return -1;
	}
	Window w;
	buildWindow(ev, n, *p, w);
	const int P = w.g.npx * w.g.npy;
	for (int i = 0; i < P; ++i)
	{
		const bool act = patchActive(w, i);
		if (active)
		{
			active[i] = act ? 1 : 0;
		}
		if (counts)
		{
			counts[i] = static_cast<int32_t>(w.bucket[i].size());
		}
		r[i] = 0.0;
		if (jac)
		{
			jac[2 * i] = 0.0;
			jac[2 * i + 1] = 0.0;
		}
		if (act)
		{
			evalPatch(w.bucket[i].data(), w.bucket[i].size(), w.rects[i], p->scale,
					  p->k, p->loss, flows + 2 * i, &r[i], jac ? jac + 2 * i : nullptr);
		}
	}
	return 0;
}

// Timing helper for bench.py's cpu_baseline: buckets once (as one solve does),
// then times `reps` batched evaluations (every active patch, value + Jacobian when
// want_jac) with a steady clock.  Single thread, like the reference.
int orc_window_eval_timed(const orc_event* ev, size_t n, const orc_params* p,
						  const double* flows, int want_jac, int reps,
						  double* seconds, uint64_t* event_evals)
{
	if (!ev || !p || !flows || !seconds || !event_evals || reps <= 0)
	{
		return -1;
	}
	Window w;
	buildWindow(ev, n, *p, w);
	const int P = w.g.npx * w.g.npy;
	uint64_t cnt = 0;
	volatile double sink = 0.0;
	const auto t0 = std::chrono::steady_clock::now();
	for (int rep = 0; rep < reps; ++rep)
	{
		for (int i = 0; i < P; ++i)
		{
			if (!patchActive(w, i))
			{
				continue;
			}
			double r, j[2];
			evalPatch(w.bucket[i].data(), w.bucket[i].size(), w.rects[i], p->scale, p->k,
					  p->loss, flows + 2 * i, &r, want_jac ? j : nullptr);
			sink = sink + r;
			cnt += w.bucket[i].size();
		}
	}
	const auto t1 = std::chrono::steady_clock::now();
	*seconds = std::chrono::duration<double>(t1 - t0).count();
	*event_evals = cnt;
	return 0;
}

int orc_compensate_events_contrast(const orc_event* ev, size_t n,
								   const orc_params* p,
								   const orc_solver_opts* o, double* flows,
								   double* image, orc_summary* summary)
{
	if (!ev || n == 0 || !p || !o || !flows)
	{
		return -1;
	}
	Window w;
	buildWindow(ev, n, *p, w);
	solveWindow(w, *o, flows, summary);
	if (image)
	{
		finalCountImage(ev, n, *p, flows, image);
	}
	return 0;
}

int orc_final_count_image(const orc_event* ev, size_t n, const orc_params* p,
						  const double* flows, double* image)
{
	if (!p || !flows || !image || (n && !ev))
	{
		return -1;
	}
	finalCountImage(ev, n, *p, flows, image);
	return 0;
}

// feature_detector.cpp:466-482.
int orc_integrate_events(const orc_event* ev, size_t n, int w, int h,
						 double* image)
{
	if (!image || (n && !ev) || w <= 0 || h <= 0)
	{
		return -1;
	}
	std::fill(image, image + static_cast<size_t>(w) * h, 0.0);
	for (size_t e = 0; e < n; ++e)
	{
		const int nx = ev[e].x;
		const int ny = ev[e].y;
		if (nx >= 0 && nx < w && ny >= 0 && ny < h)
		{
			image[static_cast<size_t>(ny) * w + nx] += 1.0;
		}
	}
	return 0;
}

// feature_detector.cpp:246-295 (warp loop only; the field is an input).
int orc_compensate_events_field(const orc_event* ev, size_t n, int w, int h,
								double scale, const float* field,
								double* image)
{
	if (!image || !field || (n && !ev) || w <= 0 || h <= 0)
	{
		return -1;
	}
	std::fill(image, image + static_cast<size_t>(w) * h, 0.0);
	if (n == 0)
	{
		return 0;
	}
	const int64_t tRef = midTimestamp(ev[0].t_us, ev[n - 1].t_us);
	for (size_t e = 0; e < n; ++e)
	{
		if (ev[e].x < 0 || ev[e].x >= w || ev[e].y < 0 || ev[e].y >= h)
		{
			continue;  // the reference reads the field out of bounds here (undefined)
		}
		const float* fl = field + 2 * (static_cast<size_t>(ev[e].y) * w + ev[e].x);
		const double dt = static_cast<double>(tRef - ev[e].t_us);
		const double fx = ev[e].x + dt * scale * fl[0];
		const double fy = ev[e].y + dt * scale * fl[1];
		if (!coordConvertible(fx) || !coordConvertible(fy))
		{
			continue;
		}
		const int nx = static_cast<int>(std::round(fx));
		const int ny = static_cast<int>(std::round(fy));
		if (nx >= 0 && nx < w && ny >= 0 && ny < h)
		{
			image[static_cast<size_t>(ny) * w + nx] += 1.0;
		}
	}
	return 0;
}

// FeatureDetector::initMotionField (feature_detector.cpp:53-142).  The field is the
// at<cv::Vec2f> view the reference uses, i.e. float32 [h][w][2].  Patch k
// contributes its trajectory samples traj[off[k]..off[k+1]) = (x, y, t_us).
// A trajectory point that rounds outside the image is written out of bounds by the
// reference (undefined); such a patch is skipped here.
int orc_init_motion_field(int w, int h, double scale, int use_average, int n_patches,
						  const size_t* traj_offsets, const double* traj_xy, const int64_t* traj_t,
						  int64_t timestamp, float* field, int32_t* n_fixed, int32_t* fixed_xy)
{
	if (!field || w <= 0 || h <= 0 || (n_patches > 0 && (!traj_offsets || !traj_xy || !traj_t)))
	{
		return -1;
	}
	std::fill(field, field + static_cast<size_t>(w) * h * 2, 0.0f);
	std::vector<int> fx, fy;
	double avgX = 0.0, avgY = 0.0, count = 0.0;
	if (timestamp > 0 && n_patches > 0)  // :60
	{
		for (int k = 0; k < n_patches; ++k)
		{
			const size_t a = traj_offsets[k], b = traj_offsets[k + 1];
			// std::lower_bound by lhs.timestamp < target (:66-72)
			size_t low = a;
			while (low < b && traj_t[low] < timestamp)
			{
				++low;
			}
			if (low != b && low + 1 != b)  // :73-74
			{
				const int px = static_cast<int>(std::round(traj_xy[2 * low]));
				const int py = static_cast<int>(std::round(traj_xy[2 * low + 1]));
				if (px < 0 || px >= w || py < 0 || py >= h)
				{
					continue;
				}
				float* f = field + 2 * (static_cast<size_t>(py) * w + px);
				const double dt = static_cast<double>(traj_t[low + 1] - traj_t[low]);
				f[0] = static_cast<float>((1 / scale) * (traj_xy[2 * (low + 1)] - traj_xy[2 * low]) / dt);
				f[1] = static_cast<float>((1 / scale) * (traj_xy[2 * (low + 1) + 1] - traj_xy[2 * low + 1]) / dt);
				fx.push_back(px);
				fy.push_back(py);
				avgX += f[0];
				avgY += f[1];
				count += 1.0;
			}
		}
		if (!fx.empty())  // :100-140
		{
			for (int y = 0; y < h; ++y)
			{
				for (int x = 0; x < w; ++x)
				{
					float* f = field + 2 * (static_cast<size_t>(y) * w + x);
					if (f[0] == 0 && f[1] == 0)
					{
						if (use_average)
						{
							f[0] = static_cast<float>(avgX / count);
							f[1] = static_cast<float>(avgY / count);
						}
						else
						{
							int bestId = 0;
							double bestDist = 1e16;
							for (int id = 0; id < static_cast<int>(fx.size()); ++id)
							{
								const double dist = (x - fx[id]) * (x - fx[id]) + (y - fy[id]) * (y - fy[id]);
								if (dist < bestDist)
								{
									bestDist = dist;
									bestId = id;
								}
							}
							const float* g = field + 2 * (static_cast<size_t>(fy[bestId]) * w + fx[bestId]);
							f[0] = g[0];
							f[1] = g[1];
						}
					}
				}
			}
		}
	}
	if (n_fixed)
	{
		*n_fixed = static_cast<int32_t>(fx.size());
	}
	if (fixed_xy)
	{
		for (size_t i = 0; i < fx.size(); ++i)
		{
			fixed_xy[2 * i] = fx[i];
			fixed_xy[2 * i + 1] = fy[i];
		}
	}
	return 0;
}

// FeatureDetector::interpolateMotionField after its initMotionField call
// (feature_detector.cpp:149-240).  field: float32 [h][w][2] as initMotionField left it,
// overwritten with the smoothed field (:230-239).  opts == NULL: ceres::Solver::Options
// defaults with :216-222 applied (50 iterations, tolerances 1e-6 / 1e-10 / 1e-8, monotonic).
int orc_interpolate_motion_field(int w, int h, int use_l1, float* field, int n_fixed,
								 const int32_t* fixed_xy, const orc_solver_opts* opts,
								 orc_summary* sum)
{
	if (!field || w < 2 || h < 2 || n_fixed < 0 || (n_fixed > 0 && !fixed_xy))
	{
		return -1;
	}
	for (int i = 0; i < n_fixed; ++i)
	{
		const int x = fixed_xy[2 * i], y = fixed_xy[2 * i + 1];
		if (x < 0 || x >= w || y < 0 || y >= h)
		{
			return -1;
		}
		if (x == w - 1 && y == h - 1)
		{
			return -2;  // not a parameter block of the problem: Ceres aborts (:208)
		}
	}
	orc_summary local;
	std::memset(&local, 0, sizeof(local));
	const size_t pixels = static_cast<size_t>(w) * h;
	if (fieldNormPositive(field, pixels))
	{
		orc_solver_opts o;
		if (opts)
		{
			o = *opts;
		}
		else
		{
			orc_default_solver(&o);
			o.use_nonmonotonic = 0;
			o.function_tolerance = 1e-6;
			o.gradient_tolerance = 1e-10;
			o.parameter_tolerance = 1e-8;
		}
		std::vector<double> mf(pixels * 2);
		for (size_t i = 0; i < pixels * 2; ++i)
		{
			mf[i] = field[i];  // :159-168
		}
		FieldProblem pb;
		pb.build(w, h, use_l1 != 0, n_fixed, fixed_xy);
		minimize(pb, o, mf.data(), &local);
		for (size_t i = 0; i < pixels * 2; ++i)
		{
			field[i] = static_cast<float>(mf[i]);  // :230-239
		}
	}
	if (sum)
	{
		*sum = local;
	}
	return 0;
}

// minimize() on Powell's function (see PowellProblem).  opts == NULL: ceres::Solver::Options
// defaults with max_num_iterations = 100 as in the tutorial.  trace: up to cap rows of
// {cost, cost_change, gradient max norm, step norm, tr_ratio, tr_radius}, one per accepted step.
int orc_lm_powell(const orc_solver_opts* opts, double* x, double* trace, int cap, int* n_rows,
				  orc_summary* sum)
{
	if (!x)
	{
		return -1;
	}
	orc_solver_opts o;
	if (opts)
	{
		o = *opts;
	}
	else
	{
		orc_default_solver(&o);
		o.max_num_iterations = 100;
		o.use_nonmonotonic = 0;
		o.function_tolerance = 1e-6;
		o.gradient_tolerance = 1e-10;
		o.parameter_tolerance = 1e-8;
	}
	PowellProblem pb;
	std::vector<double> rows;
	g_lmTrace = &rows;
	minimize(pb, o, x, sum);
	g_lmTrace = nullptr;
	const int n = static_cast<int>(rows.size() / 6);
	if (n_rows)
	{
		*n_rows = n;
	}
	if (trace)
	{
		std::copy(rows.begin(), rows.begin() + 6 * std::min(n, cap), trace);
	}
	return 0;
}

// patch.cpp:65-85.  cv::Rect2d::contains on the int point; frameToPatchCoords
// (patch.cpp:184-189) converts (int - double) back to int by truncation.
int orc_patch_integrate(const orc_event* ev, size_t n, double rx, double ry,
						double rw, double rh, double* nabla,
						int64_t* current_ts, int64_t* time_last_update)
{
	if (!ev || n == 0 || !nabla)
	{
		return -1;
	}
	const int cols = static_cast<int>(rw);
	const int rows = static_cast<int>(rh);
	std::fill(nabla, nabla + static_cast<size_t>(cols) * rows, 0.0);
	for (size_t e = 0; e < n; ++e)
	{
		const double x = ev[e].x;
		const double y = ev[e].y;
		if (rx <= x && x < rx + rw && ry <= y && y < ry + rh)
		{
			const int px = static_cast<int>(ev[e].x - rx);
			const int py = static_cast<int>(ev[e].y - ry);
			nabla[static_cast<size_t>(py) * cols + px] += static_cast<double>(ev[e].sign);
		}
	}
	if (current_ts)
	{
		*current_ts = midTimestamp(ev[0].t_us, ev[n - 1].t_us);
	}
	if (time_last_update)
	{
		*time_last_update = static_cast<int64_t>(static_cast<int32_t>(ev[n - 1].t_us));
	}
	return 0;
}

// FeatureDetector::updatePatches' routing (feature_detector.cpp:585-596): events in stream
// order (outer loop), patches inner, `if (patch.isInPatch(event.value.point)) patch.addEvent`;
// isInPatch = cv::Rect2d::contains(Point2i) (patch.cpp:172-175).  A patch starts listening at
// start[p] and stops after max_take[p] events (the caller optimises it then and its rect moves).
int orc_route_events(const orc_event* ev, size_t n, int n_patches, const double* rects,
					 const uint32_t* start, const uint32_t* max_take, uint32_t cap, uint32_t* out_index,
					 uint32_t* out_count, uint32_t* out_next)
{
	if ((!ev && n) || n_patches < 0 || !rects || !start || !max_take || !out_count || !out_next)
	{
		return -1;
	}
	for (int p = 0; p < n_patches; ++p)
	{
		out_count[p] = 0;
		out_next[p] = (std::min(max_take[p], cap) == 0) ? start[p] : static_cast<uint32_t>(n);
	}
	for (size_t e = 0; e < n; ++e)
	{
		for (int p = 0; p < n_patches; ++p)
		{
			const uint32_t quota = std::min(max_take[p], cap);
			if (e < start[p] || out_count[p] >= quota)
			{
				continue;
			}
			const double rx = rects[4 * p], ry = rects[4 * p + 1], rw = rects[4 * p + 2], rh = rects[4 * p + 3];
			const double x = ev[e].x, y = ev[e].y;
			if (rx <= x && x < rx + rw && ry <= y && y < ry + rh)
			{
				out_index[static_cast<size_t>(p) * cap + out_count[p]] = static_cast<uint32_t>(e);
				if (++out_count[p] == quota)
				{
					out_next[p] = static_cast<uint32_t>(e + 1);
				}
			}
		}
	}
	return 0;
}

// patch.cpp:87-130.  Point2d -> Point2i is cv::saturate_cast<int>(double) =
// cvRound = round half to even under the default rounding mode (F7).
int orc_patch_integrate_mc(const orc_event* ev, size_t n, double rx, double ry,
						   double rw, double rh, const double* prelast_xy,
						   int64_t prelast_t, const double* last_xy,
						   int64_t last_t, int64_t mid_time, double* nabla,
						   int32_t* updated)
{
	if (!nabla || !prelast_xy || !last_xy || !updated)
	{
		return -1;
	}
	*updated = 0;
	if (n == 0 || !ev)
	{
		return 0;
	}
	const int64_t half = static_cast<int64_t>(
		static_cast<int32_t>(static_cast<double>(last_t - prelast_t) * 0.5));
	if (!(last_t + half >= mid_time && prelast_t < mid_time))
	{
		return 0;
	}
	const int cols = static_cast<int>(rw);
	const int rows = static_cast<int>(rh);
	std::fill(nabla, nabla + static_cast<size_t>(cols) * rows, 0.0);
	const double dirX = last_xy[0] - prelast_xy[0];
	const double dirY = last_xy[1] - prelast_xy[1];
	const double tDif = static_cast<double>(last_t - prelast_t);
	const double t = static_cast<double>(mid_time);
	for (size_t e = 0; e < n; ++e)
	{
		const double f = (t - static_cast<double>(ev[e].t_us)) / tDif;
		const double cx = static_cast<double>(ev[e].x) + f * dirX;
		const double cy = static_cast<double>(ev[e].y) + f * dirY;
		if (!coordConvertible(cx) || !coordConvertible(cy))
		{
			continue;
		}
		const int ix = static_cast<int>(std::nearbyint(cx));
		const int iy = static_cast<int>(std::nearbyint(cy));
		const double x = ix;
		const double y = iy;
		if (rx <= x && x < rx + rw && ry <= y && y < ry + rh)
		{
			const int px = static_cast<int>(ix - rx);
			const int py = static_cast<int>(iy - ry);
			nabla[static_cast<size_t>(py) * cols + px] += static_cast<double>(ev[e].sign);
		}
	}
	*updated = 1;
	return 0;
}

// davis240c_reader.cpp:60-92: "<seconds> <x> <y> <0|1>" per line; seconds are
// parsed as double and duration_cast to microseconds (multiply by 1e6, truncate).
int orc_parse_events_txt(const char* path, orc_event* out, size_t cap,
						 size_t* n)
{
	if (!path || !out || !n)
	{
		return -1;
	}
	FILE* fp = std::fopen(path, "r");
	if (!fp)
	{
		return -2;
	}
	char line[256];
	size_t cnt = 0;
	int rc = 0;
	while (cnt < cap && std::fgets(line, sizeof(line), fp))
	{
		char* s = line;
		char* end = nullptr;
		const double sec = std::strtod(s, &end);
		if (end == s)
		{
			continue;  // blank line
		}
		s = end;
		const long x = std::strtol(s, &end, 10);
		s = end;
		const long y = std::strtol(s, &end, 10);
		s = end;
		const long sign = std::strtol(s, &end, 10);
		if (end == s || (sign != 0 && sign != 1))
		{
			rc = -3;  // "Sign is not equal to 0/1"
			break;
		}
		out[cnt].t_us = static_cast<int64_t>(sec * 1000000.0);
		out[cnt].x = static_cast<int32_t>(x);
		out[cnt].y = static_cast<int32_t>(y);
		out[cnt].sign = sign == 0 ? -1 : 1;
		out[cnt].reserved = 0;
		cnt++;
	}
	std::fclose(fp);
	*n = cnt;
	return rc;
}

}  // extern "C"
