// optimizer_oracle.cpp — CPU ORACLE of the per-feature tracker objective.
//
// TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ebo_oracle.h).  Single thread, plain C++17,
// built with -ffp-contract=off.  Restates, line by line,
//   implementation/feature_tracker/include/feature_tracker/optimizer_cost.h:15-96
//       tracker::OptimizerCostFunctor (operator(), warp)
//   implementation/feature_tracker/src/optimizer.cpp:15-31,62-119,140-160
//       Optimizer::setGrad (the interleaved gradient grid), the Ceres problem of
//       Optimizer::optimize, the parameter update after the solve
//   implementation/feature_tracker/include/feature_tracker/local_parameterization_se2.hpp
//   implementation/feature_tracker/src/patch.cpp:49-63,156-165 (updatePatchRect, toCorner,
//       getNormalizedIntegratedNabla)
// and the third-party arithmetic those lines call, none of which is under /root/reference
// (empty thirdparty/ submodules, commits unknown) — restated from the published algorithms:
//   ceres-solver (>=2.0,<2.2): ceres/jet.h (Jet algebra: product rule, quotient through the
//       reciprocal, sqrt, cos, sin, pow(Jet,double)); ceres/cubic_interpolation.h (Grid2D with
//       clamped indices, CubicHermiteSpline = Catmull-Rom, BiCubicInterpolator incl. its Jet
//       overload f.v = dfdr r.v + dfdc c.v); ResidualBlock (local-parameterisation product,
//       Corrector for HuberLoss); TrustRegionMinimizer + LevenbergMarquardtStrategy with
//       DENSE_QR (QR of the Jacobian stacked on the LM diagonal).
//   Sophus (SE2): storage [cos, sin, tx, ty]; matrix2x3; group product with the one-step
//       renormalisation of the unit complex; exp with the small-angle series below 1e-10;
//       inverse; Dx_this_mul_exp_x_at_0.
// PARITY UNPINNED: no test of the reference exercises the optimizer; Ceres and Sophus are
// absent.  tests/test_optimizer.py checks this file against central differences of its own
// double-precision path and against group identities, not against the reference.
#include "ebo_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace
{
constexpr int NP = 5;  // Sophus::SE2d::num_parameters (4) + the flow direction (1)
constexpr int NT = 4;  // tangent size: SE2d::DoF (3) + 1

// ------------------------------ ceres::Jet<double, 5> -----------------------------
struct Jet5
{
	double a;
	double v[NP];
	Jet5() : a(0.0) { std::memset(v, 0, sizeof(v)); }
	explicit Jet5(double s) : a(s) { std::memset(v, 0, sizeof(v)); }
};
inline Jet5 operator+(const Jet5& f, const Jet5& g)
{
	Jet5 r;
	r.a = f.a + g.a;
	for (int k = 0; k < NP; ++k) r.v[k] = f.v[k] + g.v[k];
	return r;
}
inline Jet5 operator*(const Jet5& f, const Jet5& g)
{
	Jet5 r;
	r.a = f.a * g.a;
	for (int k = 0; k < NP; ++k) r.v[k] = f.a * g.v[k] + f.v[k] * g.a;
	return r;
}
inline Jet5 operator/(const Jet5& f, const Jet5& g)  // jet.h: through the reciprocal
{
	const double gInv = 1.0 / g.a;
	const double q = f.a * gInv;
	Jet5 r;
	r.a = q;
	for (int k = 0; k < NP; ++k) r.v[k] = (f.v[k] - q * g.v[k]) * gInv;
	return r;
}
inline Jet5& operator+=(Jet5& f, const Jet5& g)
{
	f = f + g;
	return f;
}
inline Jet5 sqrtT(const Jet5& f)
{
	const double t = std::sqrt(f.a);
	const double twoInv = 1.0 / (2.0 * t);
	Jet5 r;
	r.a = t;
	for (int k = 0; k < NP; ++k) r.v[k] = f.v[k] * twoInv;
	return r;
}
inline Jet5 cosT(const Jet5& f)
{
	Jet5 r;
	r.a = std::cos(f.a);
	const double s = -std::sin(f.a);
	for (int k = 0; k < NP; ++k) r.v[k] = s * f.v[k];
	return r;
}
inline Jet5 sinT(const Jet5& f)
{
	Jet5 r;
	r.a = std::sin(f.a);
	const double c = std::cos(f.a);
	for (int k = 0; k < NP; ++k) r.v[k] = c * f.v[k];
	return r;
}
inline Jet5 powT(const Jet5& f, double g)  // pow(Jet, double)
{
	const double t = g * std::pow(f.a, g - 1.0);
	Jet5 r;
	r.a = std::pow(f.a, g);
	for (int k = 0; k < NP; ++k) r.v[k] = t * f.v[k];
	return r;
}
inline double sqrtT(double x) { return std::sqrt(x); }
inline double cosT(double x) { return std::cos(x); }
inline double sinT(double x) { return std::sin(x); }
inline double powT(double x, double g) { return std::pow(x, g); }
inline double scalarOf(double x) { return x; }
inline double scalarOf(const Jet5& x) { return x.a; }
template <class T>
inline T constT(double s);
template <>
inline double constT<double>(double s)
{
	return s;
}
template <>
inline Jet5 constT<Jet5>(double s)
{
	return Jet5(s);
}
inline double neg(double x) { return -x; }
inline Jet5 neg(const Jet5& f)
{
	Jet5 r;
	r.a = -f.a;
	for (int k = 0; k < NP; ++k) r.v[k] = -f.v[k];
	return r;
}

// -------------------- ceres::Grid2D<double, 2> + BiCubicInterpolator ----------------
// Optimizer::setGrad (optimizer.cpp:15-31): grad_[2 * (row * width + col) + {0,1}] =
// {gradX, gradY}(row, col); Grid(grad_.data(), 0, height, 0, width), row-major, interleaved.
struct Grid
{
	const double* data;
	int rows, cols;
	void get(int r, int c, double* f) const  // Grid2D::GetValue: indices clamped
	{
		const int ri = std::min(std::max(0, r), rows - 1);
		const int ci = std::min(std::max(0, c), cols - 1);
		const size_t n = static_cast<size_t>(cols) * ri + ci;
		f[0] = data[2 * n];
		f[1] = data[2 * n + 1];
	}
};

// CubicHermiteSpline<2>: a, b, c, d of the Catmull-Rom segment, Horner evaluation.
void cubicHermite(const double* p0, const double* p1, const double* p2, const double* p3, double x,
				  double* f, double* dfdx)
{
	for (int k = 0; k < 2; ++k)
	{
		const double a = 0.5 * (-p0[k] + 3.0 * p1[k] - 3.0 * p2[k] + p3[k]);
		const double b = 0.5 * (2.0 * p0[k] - 5.0 * p1[k] + 4.0 * p2[k] - p3[k]);
		const double c = 0.5 * (-p0[k] + p2[k]);
		const double d = p1[k];
		if (f)
		{
			f[k] = d + x * (c + x * (b + x * a));
		}
		if (dfdx)
		{
			dfdx[k] = c + x * (2.0 * b + 3.0 * a * x);
		}
	}
}

// BiCubicInterpolator::Evaluate(r, c, f, dfdr, dfdc).
void bicubic(const Grid& g, double r, double c, double* f, double* dfdr, double* dfdc)
{
	const int row = static_cast<int>(std::floor(r));
	const int col = static_cast<int>(std::floor(c));
	double fr[4][2], dfr[4][2];
	for (int k = 0; k < 4; ++k)
	{
		double p0[2], p1[2], p2[2], p3[2];
		g.get(row - 1 + k, col - 1, p0);
		g.get(row - 1 + k, col, p1);
		g.get(row - 1 + k, col + 1, p2);
		g.get(row - 1 + k, col + 2, p3);
		cubicHermite(p0, p1, p2, p3, c - col, fr[k], dfr[k]);
	}
	cubicHermite(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
	if (dfdc)
	{
		cubicHermite(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, nullptr);
	}
}
inline void interpolate(const Grid& g, double r, double c, double* f) { bicubic(g, r, c, f, nullptr, nullptr); }
inline void interpolate(const Grid& g, const Jet5& r, const Jet5& c, Jet5* f)  // the Jet overload
{
	double frc[2], dfdr[2], dfdc[2];
	bicubic(g, r.a, c.a, frc, dfdr, dfdc);
	for (int i = 0; i < 2; ++i)
	{
		f[i].a = frc[i];
		for (int k = 0; k < NP; ++k)
		{
			f[i].v[k] = dfdr[i] * r.v[k] + dfdc[i] * c.v[k];
		}
	}
}

// ----------------------------- OptimizerCostFunctor --------------------------------
struct Functor
{
	Grid grid;
	double tlx, tly;  // patch_.tl()
	int pw, ph;		  // static_cast<int>(patch_.width / height)
	int imgW, imgH;	  // imageSize_
	const double* nabla;  // normalizedIntegratedNabla_ [ph][pw]

	// optimizer_cost.h:48-90
	template <class T>
	void warp(const T* pose, const T* flowDir, T* res, T& normPred) const
	{
		const T vx = cosT(flowDir[0]);
		const T vy = sinT(flowDir[0]);
		// Sophus::SE2::matrix2x3(): [c -s tx; s c ty] from [c, s, tx, ty]
		const T t00 = pose[0], t01 = neg(pose[1]), t02 = pose[2];
		const T t10 = pose[1], t11 = pose[0], t12 = pose[3];
		for (int y = 0; y < ph; ++y)
		{
			for (int x = 0; x < pw; ++x)
			{
				const T X = constT<T>(x + tlx), Y = constT<T>(y + tly);
				const T wx = t00 * X + t01 * Y + t02;
				const T wy = t10 * X + t11 * Y + t12;
				const double wxa = scalarOf(wx), wya = scalarOf(wy);
				if (wxa >= imgW || wya >= imgH || wxa < 0.0 || wya < 0.0)
				{
					res[x + pw * y] = constT<T>(0.0);
				}
				else
				{
					T grads[2];
					interpolate(grid, wy, wx, grads);
					res[x + pw * y] = grads[0] * vx + grads[1] * vy;
					normPred += powT(res[x + pw * y], 2);
				}
			}
		}
	}

	// optimizer_cost.h:30-46
	template <class T>
	void operator()(const T* pose, const T* flowDir, T* res) const
	{
		T normPred = constT<T>(1e-5);
		warp(pose, flowDir, res, normPred);
		for (int y = 0; y < ph; ++y)
		{
			for (int x = 0; x < pw; ++x)
			{
				res[x + y * pw] = res[x + y * pw] / sqrtT(normPred) + constT<T>(nabla[y * pw + x]);
			}
		}
	}
};

// ceres::AutoDiffCostFunction<OptimizerCostFunctor, DYNAMIC, 4, 1>::Evaluate.
// jacPose [n][4], jacFlow [n] row-major; both null: the double path.
void evaluateFunctor(const Functor& fn, const double* pose, double flowDir, double* res, double* jacPose,
					 double* jacFlow)
{
	const int n = fn.pw * fn.ph;
	if (!jacPose && !jacFlow)
	{
		fn(pose, &flowDir, res);
		return;
	}
	Jet5 p[4], fd;
	for (int k = 0; k < 4; ++k)
	{
		p[k].a = pose[k];
		p[k].v[k] = 1.0;
	}
	fd.a = flowDir;
	fd.v[4] = 1.0;
	std::vector<Jet5> r(n);
	fn(p, &fd, r.data());
	for (int i = 0; i < n; ++i)
	{
		res[i] = r[i].a;
		if (jacPose)
		{
			for (int k = 0; k < 4; ++k)
			{
				jacPose[4 * i + k] = r[i].v[k];
			}
		}
		if (jacFlow)
		{
			jacFlow[i] = r[i].v[4];
		}
	}
}

// ------------------------------------ Sophus::SE2d ---------------------------------
// storage: [cos, sin, tx, ty]
void se2Exp(const double* a, double* T)  // tangent (ux, uy, theta)
{
	const double theta = a[2];
	const double c = std::cos(theta), s = std::sin(theta);
	double sinByTheta, oneMinusCosByTheta;
	if (std::fabs(theta) < 1e-10)  // Constants<double>::epsilon()
	{
		const double thetaSq = theta * theta;
		sinByTheta = 1.0 - (1.0 / 6.0) * thetaSq;
		oneMinusCosByTheta = 0.5 * theta - (1.0 / 24.0) * theta * thetaSq;
	}
	else
	{
		sinByTheta = s / theta;
		oneMinusCosByTheta = (1.0 - c) / theta;
	}
	T[0] = c;
	T[1] = s;
	T[2] = sinByTheta * a[0] - oneMinusCosByTheta * a[1];
	T[3] = oneMinusCosByTheta * a[0] + sinByTheta * a[1];
}
void se2Mul(const double* A, const double* B, double* C)
{
	// SO2 product with the renormalisation step of SO2Base::operator*
	double re = A[0] * B[0] - A[1] * B[1];
	double im = A[0] * B[1] + A[1] * B[0];
	const double sq = re * re + im * im;
	if (sq != 1.0)
	{
		const double scale = 2.0 / (1.0 + sq);
		re *= scale;
		im *= scale;
	}
	const double tx = A[2] + (A[0] * B[2] - A[1] * B[3]);
	const double ty = A[3] + (A[1] * B[2] + A[0] * B[3]);
	C[0] = re;
	C[1] = im;
	C[2] = tx;
	C[3] = ty;
}
void se2Inverse(const double* A, double* C)
{
	const double c = A[0], s = -A[1];  // conjugate
	C[0] = c;
	C[1] = s;
	// -(R^-1 t)
	const double rx = c * A[2] - s * A[3];
	const double ry = s * A[2] + c * A[3];
	C[2] = -rx;
	C[3] = -ry;
}
// LocalParameterizationSE2::Plus and the scalar block's plus: x [5], delta [4].
void plusAll(const double* x, const double* delta, double* out)
{
	double e[4];
	se2Exp(delta, e);
	se2Mul(x, e, out);
	out[4] = x[4] + delta[3];
}
// Dx_this_mul_exp_x_at_0: 4 x 3 row-major.
void se2PlusJacobian(const double* T, double* J)
{
	const double c = T[0], s = T[1];
	const double M[12] = {0, 0, -s, 0, 0, c, c, -s, 0, s, c, 0};
	std::memcpy(J, M, sizeof(M));
}

// ceres::HuberLoss(a)
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

// One residual block as ceres::ResidualBlock::Evaluate leaves it: cost = rho(|r|^2)/2,
// residuals and the LOCAL Jacobian (global Jacobian times the plus-Jacobian) scaled by
// sqrt(rho') (Corrector with rho'' <= 0).  jac [n][4] or null.
struct Block
{
	Functor fn;
	double huberA;
	int evalsCost = 0, evalsJac = 0;

	bool evaluate(const double* x, double* cost, std::vector<double>* res, std::vector<double>* jac)
	{
		const int n = fn.pw * fn.ph;
		std::vector<double> r(n), jp, jf;
		if (jac)
		{
			jp.resize(static_cast<size_t>(n) * 4);
			jf.resize(n);
			evaluateFunctor(fn, x, x[4], r.data(), jp.data(), jf.data());
			evalsJac++;
		}
		else
		{
			evaluateFunctor(fn, x, x[4], r.data(), nullptr, nullptr);
			evalsCost++;
		}
		double s = 0.0;
		for (int i = 0; i < n; ++i)
		{
			s += r[i] * r[i];
		}
		double rho[3];
		huber(huberA, s, rho);
		*cost = 0.5 * rho[0];
		const double sr = std::sqrt(rho[1]);
		if (jac)
		{
			double P[12];
			se2PlusJacobian(x, P);
			jac->assign(static_cast<size_t>(n) * NT, 0.0);
			for (int i = 0; i < n; ++i)
			{
				for (int c = 0; c < 3; ++c)
				{
					double acc = 0.0;
					for (int k = 0; k < 4; ++k)
					{
						acc += jp[4 * i + k] * P[3 * k + c];
					}
					(*jac)[NT * i + c] = acc * sr;
				}
				(*jac)[NT * i + 3] = jf[i] * sr;
			}
		}
		if (res)
		{
			res->resize(n);
			for (int i = 0; i < n; ++i)
			{
				(*res)[i] = r[i] * sr;
			}
		}
		return std::isfinite(*cost);
	}
};

// Least squares min |A x - b| by Householder QR, A [m][n] row-major (DenseQRSolver on the
// Jacobian stacked on the LM diagonal).  Returns false on a zero column.
bool qrSolve(std::vector<double>& A, int m, int n, std::vector<double>& b, double* x)
{
	for (int k = 0; k < n; ++k)
	{
		double norm = 0.0;
		for (int i = k; i < m; ++i)
		{
			norm += A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(i) * n + k];
		}
		norm = std::sqrt(norm);
		if (!(norm > 0.0) || !std::isfinite(norm))
		{
			return false;
		}
		const double akk = A[static_cast<size_t>(k) * n + k];
		const double alpha = akk > 0.0 ? -norm : norm;
		// v = a_k - alpha e_k (stored in place), H = I - 2 v v' / (v'v)
		A[static_cast<size_t>(k) * n + k] = akk - alpha;
		double vv = 0.0;
		for (int i = k; i < m; ++i)
		{
			vv += A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(i) * n + k];
		}
		for (int j = k + 1; j < n; ++j)
		{
			double dot = 0.0;
			for (int i = k; i < m; ++i)
			{
				dot += A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(i) * n + j];
			}
			const double f = 2.0 * dot / vv;
			for (int i = k; i < m; ++i)
			{
				A[static_cast<size_t>(i) * n + j] -= f * A[static_cast<size_t>(i) * n + k];
			}
		}
		double dot = 0.0;
		for (int i = k; i < m; ++i)
		{
			dot += A[static_cast<size_t>(i) * n + k] * b[i];
		}
		const double f = 2.0 * dot / vv;
		for (int i = k; i < m; ++i)
		{
			b[i] -= f * A[static_cast<size_t>(i) * n + k];
		}
		A[static_cast<size_t>(k) * n + k] = alpha;  // R(k,k); the rest of the column is v
	}
	for (int k = n - 1; k >= 0; --k)
	{
		double s = b[k];
		for (int j = k + 1; j < n; ++j)
		{
			s -= A[static_cast<size_t>(k) * n + j] * x[j];
		}
		x[k] = s / A[static_cast<size_t>(k) * n + k];
	}
	return true;
}

// Ceres 2.0 TrustRegionMinimizer / LevenbergMarquardtStrategy / TrustRegionStepEvaluator with
// the options of optimizer.cpp:103-112 (DENSE_QR, non-monotonic steps, max_num_iterations;
// everything else at its default).  x [5] in, lowest-cost point visited out.
void minimizeBlock(Block& blk, const orc_solver_opts& o, double* x, orc_summary* sum)
{
	const int n = NT;
	const int m = blk.fn.pw * blk.fn.ph;
	orc_summary local;
	std::memset(&local, 0, sizeof(local));
	std::vector<double> f, J;
	double xcur[NP], xcand[NP], xbest[NP];
	std::memcpy(xcur, x, sizeof(xcur));
	std::memcpy(xbest, x, sizeof(xbest));
	double xCost = 0.0, scale[NT] = {1, 1, 1, 1}, grad[NT], diag[NT], lmDiag[NT], step[NT], delta[NT];
	double gradMax = 0.0;

	auto vecNorm = [](const double* v, int k) {
		double s = 0.0;
		for (int i = 0; i < k; ++i)
		{
			s += v[i] * v[i];
		}
		return std::sqrt(s);
	};
	auto evalJac = [&]() -> bool {
		if (!blk.evaluate(xcur, &xCost, &f, &J))
		{
			return false;
		}
		for (int c = 0; c < n; ++c)
		{
			grad[c] = 0.0;
		}
		for (int r = 0; r < m; ++r)
		{
			for (int c = 0; c < n; ++c)
			{
				grad[c] += J[static_cast<size_t>(r) * n + c] * f[r];
			}
		}
		return true;
	};
	auto scaleJac = [&]() {
		for (int r = 0; r < m; ++r)
		{
			for (int c = 0; c < n; ++c)
			{
				J[static_cast<size_t>(r) * n + c] *= scale[c];
			}
		}
	};
	// gradient_max_norm = |x - Plus(x, -g)|_inf (the gradient lives in the tangent space)
	auto projectedGradientMax = [&]() {
		double ng[NT], xp[NP];
		for (int c = 0; c < n; ++c)
		{
			ng[c] = -grad[c];
		}
		plusAll(xcur, ng, xp);
		double mx = 0.0;
		for (int i = 0; i < NP; ++i)
		{
			mx = std::max(mx, std::fabs(xcur[i] - xp[i]));
		}
		return mx;
	};

	if (!evalJac())
	{
		local.termination = 2;
		if (sum)
		{
			*sum = local;
		}
		return;
	}
	local.initial_cost = xCost;
	if (o.jacobi_scaling)
	{
		for (int c = 0; c < n; ++c)
		{
			double cn = 0.0;
			for (int r = 0; r < m; ++r)
			{
				cn += J[static_cast<size_t>(r) * n + c] * J[static_cast<size_t>(r) * n + c];
			}
			scale[c] = 1.0 / (1.0 + std::sqrt(cn));
		}
	}
	scaleJac();
	double xNorm = vecNorm(xcur, NP);
	gradMax = projectedGradientMax();
	double minimumCost = xCost;

	const int maxNonmono = o.use_nonmonotonic ? o.max_consecutive_nonmonotonic : 0;
	double seMinimum = xCost, seCurrent = xCost, seReference = xCost, seCandidate = xCost;
	double seAccRef = 0.0, seAccCand = 0.0;
	int seNumNonmono = 0;
	double radius = o.initial_radius, decreaseFactor = 2.0;
	bool reuseDiagonal = false;
	int iteration = 0, numInvalid = 0, termination = 1;
	bool lastSuccessful = true;

	for (;;)
	{
		if (lastSuccessful && xCost < minimumCost)
		{
			minimumCost = xCost;
			std::memcpy(xbest, xcur, sizeof(xbest));
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

		if (!reuseDiagonal)
		{
			for (int c = 0; c < n; ++c)
			{
				double d = 0.0;
				for (int r = 0; r < m; ++r)
				{
					d += J[static_cast<size_t>(r) * n + c] * J[static_cast<size_t>(r) * n + c];
				}
				diag[c] = std::min(std::max(d, o.min_lm_diagonal), o.max_lm_diagonal);
			}
		}
		for (int c = 0; c < n; ++c)
		{
			lmDiag[c] = std::sqrt(diag[c] / radius);
		}
		reuseDiagonal = true;
		// DenseQRSolver: [J; diag(D)] x = [f; 0], step = -x
		std::vector<double> A(static_cast<size_t>(m + n) * n, 0.0), b(m + n, 0.0);
		std::copy(J.begin(), J.end(), A.begin());
		for (int c = 0; c < n; ++c)
		{
			A[static_cast<size_t>(m + c) * n + c] = lmDiag[c];
		}
		std::copy(f.begin(), f.end(), b.begin());
		bool valid = true;
		if (o.mode == 1)
		{
			// diagnostic only (tests/diag_optimizer.py): the same step from the normal equations
			// (J'J + D'D) x = J'f by Gaussian elimination, to tell the conditioning of the
			// problem from the linear solver used
			double Hn[NT][NT + 1];
			for (int a = 0; a < n; ++a)
			{
				for (int c = 0; c <= n; ++c)
				{
					Hn[a][c] = 0.0;
				}
			}
			for (int r = 0; r < m + n; ++r)
			{
				for (int a = 0; a < n; ++a)
				{
					for (int c = 0; c < n; ++c)
					{
						Hn[a][c] += A[static_cast<size_t>(r) * n + a] * A[static_cast<size_t>(r) * n + c];
					}
					Hn[a][n] += A[static_cast<size_t>(r) * n + a] * b[r];
				}
			}
			for (int k = 0; k < n && valid; ++k)
			{
				valid = Hn[k][k] > 0.0;
				for (int r = k + 1; r < n && valid; ++r)
				{
					const double fct = Hn[r][k] / Hn[k][k];
					for (int c = k; c <= n; ++c)
					{
						Hn[r][c] -= fct * Hn[k][c];
					}
				}
			}
			for (int k = n - 1; k >= 0 && valid; --k)
			{
				double s = Hn[k][n];
				for (int c = k + 1; c < n; ++c)
				{
					s -= Hn[k][c] * step[c];
				}
				step[k] = s / Hn[k][k];
			}
		}
		else
		{
			valid = qrSolve(A, m + n, n, b, step);
		}
		if (valid)
		{
			for (int c = 0; c < n; ++c)
			{
				valid = valid && std::isfinite(step[c]);
				step[c] = -step[c];
			}
		}
		double modelCostChange = 0.0;
		if (valid)
		{
			for (int r = 0; r < m; ++r)
			{
				double mr = 0.0;
				for (int c = 0; c < n; ++c)
				{
					mr += J[static_cast<size_t>(r) * n + c] * step[c];
				}
				modelCostChange -= mr * (f[r] + mr / 2.0);
			}
			valid = modelCostChange > 0.0;
		}
		if (!valid)
		{
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
		for (int c = 0; c < n; ++c)
		{
			delta[c] = step[c] * scale[c];
		}
		plusAll(xcur, delta, xcand);
		double candCost = 0.0;
		if (!blk.evaluate(xcand, &candCost, nullptr, nullptr))
		{
			candCost = std::numeric_limits<double>::max();
		}
		double d5[NP];
		for (int i = 0; i < NP; ++i)
		{
			d5[i] = xcur[i] - xcand[i];
		}
		const double stepNorm = vecNorm(d5, NP);
		if (stepNorm <= o.parameter_tolerance * (xNorm + o.parameter_tolerance))
		{
			termination = 0;
			break;
		}
		const double costChange = xCost - candCost;
		if (std::fabs(costChange) <= o.function_tolerance * xCost)
		{
			termination = 0;
			break;
		}
		const double relDec = (seCurrent - candCost) / modelCostChange;
		const double histDec = (seReference - candCost) / (seAccRef + modelCostChange);
		const double quality = std::max(relDec, histDec);
		if (quality > o.min_relative_decrease)
		{
			std::memcpy(xcur, xcand, sizeof(xcur));
			xNorm = vecNorm(xcur, NP);
			if (!evalJac())
			{
				termination = 2;
				break;
			}
			scaleJac();
			gradMax = projectedGradientMax();
			lastSuccessful = true;
			radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * quality - 1.0, 3));
			radius = std::min(o.max_radius, radius);
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
			if (seNumNonmono == maxNonmono)
			{
				seReference = seCandidate;
				seAccRef = seAccCand;
			}
		}
		else
		{
			radius = radius / decreaseFactor;
			decreaseFactor *= 2.0;
			reuseDiagonal = true;
		}
	}
	std::memcpy(x, xbest, sizeof(xbest));
	local.iterations = iteration;
	local.termination = termination;
	local.final_cost = minimumCost;
	local.num_evals_cost = blk.evalsCost;
	local.num_evals_jac = blk.evalsJac;
	if (sum)
	{
		*sum = local;
	}
}

bool makeFunctor(const double* grad, int imgW, int imgH, double rx, double ry, double rw, double rh,
				 const double* nabla, Functor& fn)
{
	if (!grad || !nabla || imgW <= 0 || imgH <= 0 || !(rw >= 1.0) || !(rh >= 1.0) || rw > 4096 || rh > 4096)
	{
		return false;
	}
	fn.grid = Grid{grad, imgH, imgW};
	fn.tlx = rx;
	fn.tly = ry;
	fn.pw = static_cast<int>(rw);
	fn.ph = static_cast<int>(rh);
	fn.imgW = imgW;
	fn.imgH = imgH;
	fn.nabla = nabla;
	return true;
}
}  // namespace

extern "C" {

int orc_optimizer_cost(const double* grad, int img_w, int img_h, double rx, double ry, double rw, double rh,
					   const double* nabla, const double* pose, double flow_dir, double* residuals,
					   double* jac_pose, double* jac_flow)
{
	Functor fn;
	if (!pose || !residuals || !makeFunctor(grad, img_w, img_h, rx, ry, rw, rh, nabla, fn))
	{
		return -1;
	}
	evaluateFunctor(fn, pose, flow_dir, residuals, jac_pose, jac_flow);
	return 0;
}

// Optimizer::drawCostMap (optimizer.cpp:33-60): costMap(y + (H-1)/2, x + (W-1)/2) = cv::norm(image, NORM_L2) of the
// functor's residual image at poseNew = SE2(pose.log().z(), (float(x) + tx, float(y) + ty)); pose.log().z() is
// Sophus' SO2::log = atan2(sin, cos), and SE2(theta, t) stores (cos theta, sin theta).  flow_dir: the caller passes
// patch.getFlow(), i.e. the double member rounded through float (patch.h:56).  out [map_h][map_w], zeros first
// (cv::Mat::zeros: an even size leaves its last row / column unvisited).
int orc_optimizer_cost_map(const double* grad, int img_w, int img_h, double rx, double ry, double rw, double rh,
						   const double* nabla, const double* pose, double flow_dir, int map_w, int map_h, double* out)
{
	Functor fn;
	if (!pose || !out || map_w < 1 || map_h < 1 || !makeFunctor(grad, img_w, img_h, rx, ry, rw, rh, nabla, fn))
	{
		return -1;
	}
	const int n = fn.pw * fn.ph;
	std::vector<double> image(static_cast<size_t>(n));
	for (int i = 0; i < map_w * map_h; ++i)
	{
		out[i] = 0.0;
	}
	const double theta = std::atan2(pose[1], pose[0]);
	for (int x = -(map_w - 1) / 2; x <= (map_w - 1) / 2; ++x)
	{
		for (int y = -(map_h - 1) / 2; y <= (map_h - 1) / 2; ++y)
		{
			const double poseNew[4] = {std::cos(theta), std::sin(theta), static_cast<float>(x) + pose[2],
									   static_cast<float>(y) + pose[3]};
			evaluateFunctor(fn, poseNew, flow_dir, image.data(), nullptr, nullptr);
			double ss = 0.0;
			for (int i = 0; i < n; ++i)
			{
				ss += image[i] * image[i];
			}
			out[(y + (map_h - 1) / 2) * map_w + (x + (map_w - 1) / 2)] = std::sqrt(ss);
		}
	}
	return 0;
}

void orc_optimizer_default_solver(orc_solver_opts* o)
{
	orc_default_solver(o);
	o->max_num_iterations = 10;  // OptimizerParams::maxNumIterations
	o->use_nonmonotonic = 1;	 // optimizer.cpp:110
	o->function_tolerance = 1e-6;
	o->gradient_tolerance = 1e-10;
	o->parameter_tolerance = 1e-8;
}

int orc_optimizer_solve(const double* grad, int img_w, int img_h, double rx, double ry, double rw, double rh,
						const double* nabla, double huber_a, const orc_solver_opts* opts, double* pose,
						double* flow_dir, orc_summary* summary)
{
	Block blk;
	if (!pose || !flow_dir || !makeFunctor(grad, img_w, img_h, rx, ry, rw, rh, nabla, blk.fn))
	{
		return -1;
	}
	blk.huberA = huber_a;
	orc_solver_opts o;
	if (opts)
	{
		o = *opts;
	}
	else
	{
		orc_optimizer_default_solver(&o);
	}
	double x[NP] = {pose[0], pose[1], pose[2], pose[3], *flow_dir};
	minimizeBlock(blk, o, x, summary);
	std::memcpy(pose, x, 4 * sizeof(double));
	*flow_dir = x[4];
	return 0;
}

int orc_se2_plus(const double* pose, const double* delta3, double* out)
{
	if (!pose || !delta3 || !out)
	{
		return -1;
	}
	double e[4];
	se2Exp(delta3, e);
	se2Mul(pose, e, out);
	return 0;
}

// cv::warpAffine(src, dst, M, size, cv::WARP_INVERSE_MAP) as FeatureDetector::updateNumOfEvents
// calls it (feature_detector.cpp:697-700): no interpolation bits in the flags = INTER_NEAREST;
// WARP_INVERSE_MAP = M maps DESTINATION pixels to source positions as given.  OpenCV evaluates the
// map in fixed point (imgwarp.cpp): AB_BITS = 10, adelta[x] = cvRound(M[0] x 1024), bdelta[x] =
// cvRound(M[3] x 1024), per row X0 = cvRound((M[1] y + M[2]) 1024) + 512, Y0 likewise, source pixel
// (X0 + adelta[x]) >> 10, (Y0 + bdelta[x]) >> 10, BORDER_CONSTANT 0 outside.  cvRound = lrint
// (round half to even).  Then warped(rect): cv::Rect2d -> cv::Rect by saturate_cast = cvRound of
// x, y, width, height; cv::norm(0.6 gradX cos + 0.6 gradY sin, NORM_L1): the MatExpr scales are
// folded, (0.6 * cos) and (0.6 * sin), one a*x + b*y per pixel, summed row-major; size_t truncates.
int orc_estimate_num_events(const double* grad, int img_w, int img_h, double rx, double ry, double rw, double rh,
							const double* warp, double flow_dir, uint64_t* out)
{
	if (!grad || !warp || !out || img_w <= 0 || img_h <= 0)
	{
		return -1;
	}
	// matrix2x3 of SE2 (c, s, tx, ty): [c -s tx; s c ty]
	const double M[6] = {warp[0], -warp[1], warp[2], warp[1], warp[0], warp[3]};
	const int x0 = static_cast<int>(std::lrint(rx)), y0 = static_cast<int>(std::lrint(ry));
	const int w = static_cast<int>(std::lrint(rw)), h = static_cast<int>(std::lrint(rh));
	// `const auto flow = patch.getFlow();` is a float (patch.h:56): float cos / sin, widened
	const float ff = static_cast<float>(flow_dir);
	const double a = 0.6 * static_cast<double>(std::cos(ff)), b = 0.6 * static_cast<double>(std::sin(ff));
	double sum = 0.0;
	for (int y = y0; y < y0 + h; ++y)
	{
		const int X0 = static_cast<int>(std::lrint((M[1] * y + M[2]) * 1024.0)) + 512;
		const int Y0 = static_cast<int>(std::lrint((M[4] * y + M[5]) * 1024.0)) + 512;
		for (int x = x0; x < x0 + w; ++x)
		{
			const int X = (X0 + static_cast<int>(std::lrint(M[0] * x * 1024.0))) >> 10;
			const int Y = (Y0 + static_cast<int>(std::lrint(M[3] * x * 1024.0))) >> 10;
			double gx = 0.0, gy = 0.0;
			if (x >= 0 && x < img_w && y >= 0 && y < img_h && X >= 0 && X < img_w && Y >= 0 && Y < img_h)
			{
				gx = grad[2 * (static_cast<size_t>(Y) * img_w + X)];
				gy = grad[2 * (static_cast<size_t>(Y) * img_w + X) + 1];
			}
			sum += std::fabs(gx * a + gy * b);
		}
	}
	*out = static_cast<uint64_t>(sum);
	return 0;
}

// Patch::warpImage (patch.cpp:132-154): the two gradient images through cv::warpAffine(..., warp_.matrix2x3(),
// cv::WARP_INVERSE_MAP) -- the same nearest-neighbour fixed-point map as above --, then, unless the patch
// rect touches the image border (:145-150: `patch_.x < 0 || patch_.y < 0 || patch_.x + patch_.width >= cols
// || patch_.y + patch_.height >= rows` -> return, predictedNabla_ keeps its value),
//     predictedNabla_ = -warpedGradX(patch_) * cos(flowDir_) - warpedGradY(patch_) * sin(flowDir_)
// over the rect (cv::Rect2d -> cv::Rect by cvRound).  flowDir_ is the double member here (not the float
// getFlow() returns).  The MatExpr folds the signs into the scales: one gx * (-cos) + gy * (-sin) per pixel
// (cv::addWeighted).  out: [h][w] row-major, h = cvRound(rh), w = cvRound(rw); *updated = 0 on the early
// return (out untouched).
int orc_patch_warp_image(const double* grad, int img_w, int img_h, double rx, double ry, double rw, double rh,
						 const double* warp, double flow_dir, double* out, int* updated)
{
	if (!grad || !warp || !out || !updated || img_w <= 0 || img_h <= 0)
	{
		return -1;
	}
	*updated = 0;
	if (rx < 0 || ry < 0 || rx + rw >= img_w || ry + rh >= img_h)
	{
		return 0;
	}
	const double M[6] = {warp[0], -warp[1], warp[2], warp[1], warp[0], warp[3]};
	const int x0 = static_cast<int>(std::lrint(rx)), y0 = static_cast<int>(std::lrint(ry));
	const int w = static_cast<int>(std::lrint(rw)), h = static_cast<int>(std::lrint(rh));
	const double a = -std::cos(flow_dir), b = -std::sin(flow_dir);
	for (int y = y0; y < y0 + h; ++y)
	{
		const int X0 = static_cast<int>(std::lrint((M[1] * y + M[2]) * 1024.0)) + 512;
		const int Y0 = static_cast<int>(std::lrint((M[4] * y + M[5]) * 1024.0)) + 512;
		for (int x = x0; x < x0 + w; ++x)
		{
			const int X = (X0 + static_cast<int>(std::lrint(M[0] * x * 1024.0))) >> 10;
			const int Y = (Y0 + static_cast<int>(std::lrint(M[3] * x * 1024.0))) >> 10;
			double gx = 0.0, gy = 0.0;
			if (x >= 0 && x < img_w && y >= 0 && y < img_h && X >= 0 && X < img_w && Y >= 0 && Y < img_h)
			{
				gx = grad[2 * (static_cast<size_t>(Y) * img_w + X)];
				gy = grad[2 * (static_cast<size_t>(Y) * img_w + X) + 1];
			}
			out[static_cast<size_t>(y - y0) * w + (x - x0)] = gx * a + gy * b;
		}
	}
	*updated = 1;
	return 0;
}

int orc_patch_update_rect(const double* warp, double init_x, double init_y, double rw, double rh, double* rect)
{
	if (!warp || !rect)
	{
		return -1;
	}
	double inv[4];
	se2Inverse(warp, inv);  // patch.cpp:51
	const double cx = inv[0] * init_x + (-inv[1]) * init_y + inv[2];
	const double cy = inv[1] * init_x + inv[0] * init_y + inv[3];
	const int ex = static_cast<int>((rw - 1) / 2);  // patch.cpp:58-59
	const int ey = static_cast<int>((rh - 1) / 2);
	rect[0] = cx - ex;
	rect[1] = cy - ey;
	rect[2] = 2 * ex + 1;
	rect[3] = 2 * ey + 1;
	return 0;
}

int orc_normalize_nabla(const double* nabla, int n, double* out)
{
	if (!nabla || !out || n <= 0)
	{
		return -1;
	}
	double s = 0.0;
	for (int i = 0; i < n; ++i)
	{
		s += nabla[i] * nabla[i];
	}
	// patch.cpp:158: cv::norm(NORM_L2); `Mat / double` is OpenCV's MatExpr a * (1 / s)
	const double inv = 1.0 / std::sqrt(s);
	for (int i = 0; i < n; ++i)
	{
		out[i] = nabla[i] * inv;
	}
	return 0;
}
}
