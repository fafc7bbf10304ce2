// ceres_reference_lines_test.cpp — the reference's Ceres construction expressions against the facade.
//
// FeatureDetector::compensateEventsContrast builds its problem with
//     new ceres::AutoDiffCostFunction<tracker::contrastFunctor, 1, 2>(new tracker::contrastFunctor(
//         patchEvents, patchRect, params_.compensateScale))                (feature_detector.cpp:359-363)
//     new ceres::AutoDiffCostFunction<tracker::totalVarianceFunctor, 2, 2, 2>(
//         new tracker::totalVarianceFunctor(params_.compensateTVweight))   (:371-375, :384-388)
//     problem.AddResidualBlock(cost_function, new ceres::HuberLoss(params_.compensateTVHuberLoss), ...)
//     Solve(options, &problem, &summary)                                   (:401-414)
// This program uses those expressions (inside its own assembly of the patch grid) against the facade headers (tracker::contrastFunctor with
// its Jet-capable operator(), tracker::totalVarianceFunctor) and the test-only Ceres declarations of
// stubs/ceres/ceres.h, so the AutoDiff instantiation of the functor meets a compiler and runs.
// ceres::Solve is DEFINED here: it recovers the patch grid from the blocks, checks that every TV
// block is the block the product's host LM models (value, both Jacobian blocks and the Huber loss at
// a probe point, through AutoDiffCostFunction::Evaluate), and runs the product's trust-region LM
// (csrc/host_lm.cpp) with every data-term evaluation going through
// AutoDiffCostFunction<contrastFunctor,1,2>::Evaluate -> contrastFunctor::operator()<Jet<double,2>>
// -> the device.  The flows must equal ebo_solve(EBO_SOLVE_GLOBAL) of the same window bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <list>
#include <vector>

#include <feature_tracker/contrast_functor.h>
#include <feature_tracker/feature_detector.h>
#include <feature_tracker/total_variance.h>

#include "../../event-based-odomety_amd/csrc/host_lm.h"

#ifndef EBO_HAVE_CERES
#error "build with -Itests/cpp/stubs"
#endif

static int failures = 0;
#define EXPECT_TRUE(c)                                                    \
	do                                                                    \
	{                                                                     \
		if (!(c))                                                         \
		{                                                                 \
			std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
			++failures;                                                   \
		}                                                                 \
	} while (0)

namespace ceres
{
// The product's host LM behind the Ceres call.  Blocks with one parameter block are data terms;
// blocks with two are the TV terms, whose structure (right / lower neighbour on an npx-wide grid,
// weight, Huber parameter) HostLm carries itself.
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary)
{
	const std::vector<Problem::Block>& blocks = problem->blocks();
	double* base = nullptr;
	double* top = nullptr;
	for (const Problem::Block& b : blocks)
	{
		for (double* p : b.params)
		{
			base = (base == nullptr || p < base) ? p : base;
			top = (top == nullptr || p > top) ? p : top;
		}
	}
	const int P = static_cast<int>((top - base) / 2) + 1;
	int npx = P;  // a lower-neighbour block tells the row length
	double tvWeight = 0.0, tvHuber = 0.0;
	int tvBlocks = 0;
	for (const Problem::Block& b : blocks)
	{
		if (b.params.size() != 2)
		{
			continue;
		}
		++tvBlocks;
		const int p = static_cast<int>((b.params[0] - base) / 2), q = static_cast<int>((b.params[1] - base) / 2);
		EXPECT_TRUE(q > p);
		if (q - p > 1)
		{
			npx = std::min(npx, q - p);
		}
		// the block at a probe point: r = w |x - y| per component, d/dx = w sgn, d/dy = -w sgn
		const double x[2] = {0.75, -0.5}, y[2] = {0.25, 1.5};
		double const* params[2] = {x, y};
		double r[2], jx[4], jy[4];
		double* jac[2] = {jx, jy};
		EXPECT_TRUE(b.cost->num_residuals() == 2 && b.cost->parameter_block_sizes().size() == 2);
		EXPECT_TRUE(b.cost->Evaluate(params, r, jac));
		const double w = r[0] / 0.5;
		EXPECT_TRUE(r[0] == w * 0.5 && r[1] == w * 2.0);
		EXPECT_TRUE(jx[0] == w && jx[1] == 0.0 && jx[2] == 0.0 && jx[3] == -w);
		EXPECT_TRUE(jy[0] == -w && jy[1] == 0.0 && jy[2] == 0.0 && jy[3] == w);
		EXPECT_TRUE(tvWeight == 0.0 || tvWeight == w);
		tvWeight = w;
		const HuberLoss* huber = dynamic_cast<const HuberLoss*>(b.loss.get());
		EXPECT_TRUE(huber != nullptr);
		if (huber)
		{
			EXPECT_TRUE(tvHuber == 0.0 || tvHuber == huber->a());
			tvHuber = huber->a();
			double rho[3];
			huber->Evaluate(4.0 * tvHuber * tvHuber, rho);  // outside the inlier region: 2 a sqrt(s) - a^2
			EXPECT_TRUE(std::fabs(rho[0] - 3.0 * tvHuber * tvHuber) < 1e-9 && std::fabs(rho[1] - 0.5) < 1e-12);
		}
	}
	const int npy = P / npx;
	EXPECT_TRUE(npx * npy == P);
	EXPECT_TRUE(tvBlocks == (npx - 1) * npy + npx * (npy - 1));  // :369-396: right and lower neighbours

	std::vector<uint8_t> active(P, 0);
	for (const Problem::Block& b : blocks)
	{
		if (b.params.size() == 1)
		{
			EXPECT_TRUE(b.cost->num_residuals() == 1 && b.cost->parameter_block_sizes()[0] == 2 && !b.loss);
			active[(b.params[0] - base) / 2] = 1;
		}
	}

	ebo_solver_opts o;
	ebo_default_solver(&o);
	o.max_num_iterations = options.max_num_iterations;
	o.use_nonmonotonic = options.use_nonmonotonic_steps ? 1 : 0;
	o.function_tolerance = options.function_tolerance;
	o.gradient_tolerance = options.gradient_tolerance;
	o.parameter_tolerance = options.parameter_tolerance;
	ebo::HostLm lm(npx, npy, active, tvWeight, tvHuber, o);
	std::vector<double> x(2 * static_cast<size_t>(P)), r(P), J(2 * static_cast<size_t>(P));
	for (;;)
	{
		const ebo::HostLm::Request req = lm.request(x.data());
		if (req == ebo::HostLm::DONE)
		{
			break;
		}
		const bool wantJ = req == ebo::HostLm::NEED_JACOBIAN;
		std::fill(r.begin(), r.end(), 0.0);
		std::fill(J.begin(), J.end(), 0.0);
		for (const Problem::Block& b : blocks)
		{
			if (b.params.size() != 1)
			{
				continue;
			}
			const size_t p = (b.params[0] - base) / 2;
			double const* params[1] = {&x[2 * p]};
			double jac[2] = {0.0, 0.0};
			double* jacs[1] = {jac};
			EXPECT_TRUE(b.cost->Evaluate(params, &r[p], wantJ ? jacs : nullptr));
			J[2 * p] = jac[0];
			J[2 * p + 1] = jac[1];
		}
		lm.supply(r.data(), wantJ ? J.data() : nullptr);
	}
	lm.result(x.data());
	std::memcpy(base, x.data(), sizeof(double) * 2 * P);  // Ceres leaves the solution in the user's blocks
	const ebo::HostLm::Stats& st = lm.stats();
	summary->num_successful_steps = st.iterations;
	summary->num_residual_evaluations = st.evals_cost + st.evals_jac;
	summary->num_jacobian_evaluations = st.evals_jac;
	summary->initial_cost = st.initial_cost;
	summary->final_cost = st.final_cost;
	summary->termination_type = st.termination;
}
}  // namespace ceres

namespace tracker
{
// The facade's functor takes the loss as a defaulted fourth constructor argument; the reference's
// three-argument construction gets the reference's loss (edge).  For the variance case of this test
// the same statements are compiled a second time with this alias.
struct contrastFunctorVariance : contrastFunctor
{
	contrastFunctorVariance(const std::list<common::EventSample>& events, const Rect2i patchRect, double scale)
		: contrastFunctor(events, patchRect, scale, EBO_LOSS_VARIANCE)
	{
	}
};
}  // namespace tracker

static std::vector<common::EventSample> makeEvents(int n, uint64_t seed)
{
	std::vector<common::EventSample> out;
	uint64_t s = seed;
	auto rnd = [&]() {
		s += 0x9E3779B97F4A7C15ull;
		uint64_t z = s;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	};
	auto unit = [&]() { return static_cast<double>(rnd() >> 11) * (1.0 / 9007199254740992.0); };
	for (int i = 0; i < n; ++i)
	{
		const int64_t t = 1000000 + static_cast<int64_t>(50000.0 * i / n);
		const int px = static_cast<int>(unit() * 12), py = static_cast<int>(unit() * 9);
		const double vx = ((px * 7 + py * 3) % 11) / 5.5 - 1.0, vy = ((px * 5 + py * 9) % 13) / 6.5 - 1.0;
		const double dt = (static_cast<double>(t) - 1025000.0) * 1e-3;
		double x = px * 20 + 10 + (unit() * 2 - 1) * 6 * ((px + py) % 2 ? 1.0 : 0.2) + vx * dt + std::floor(unit() * 3) - 1;
		double y = py * 20 + 10 + (unit() * 2 - 1) * 6 * ((px + py) % 2 ? 0.2 : 1.0) + vy * dt + std::floor(unit() * 3) - 1;
		common::EventSample e;
		e.value.point = {static_cast<int>(std::fmin(std::fmax(std::floor(x), 0), 239)),
						 static_cast<int>(std::fmin(std::fmax(std::floor(y), 0), 179))};
		e.value.sign = (rnd() & 1) ? common::EventPolarity::POSITIVE : common::EventPolarity::NEGATIVE;
		e.timestamp = common::timestamp_t(t);
		out.push_back(e);
	}
	return out;
}

// The problem FeatureDetector::compensateEventsContrast hands to Ceres (feature_detector.cpp:301-414), assembled by
// this file's own statements.  What is the reference's are the CONSTRUCTION EXPRESSIONS -- the interface under test:
//     new ceres::AutoDiffCostFunction<tracker::contrastFunctor, 1, 2>(new tracker::contrastFunctor(events, rect, scale))
//     new ceres::AutoDiffCostFunction<tracker::totalVarianceFunctor, 2, 2, 2>(new tracker::totalVarianceFunctor(weight))
//     problem.AddResidualBlock(cost, new ceres::HuberLoss(delta), a, b);   Solve(options, &problem, &summary);
// and the option values of :401-410.  Functor = tracker::contrastFunctor is the reference's type; the variance alias only
// changes which loss the device evaluates.
struct GridOfPatches
{
	int nx, ny, sensorW, sensorH, pw, ph;
	explicit GridOfPatches(const tracker::DetectorParams& dp)
		: nx(dp.imageSize.width / dp.patchCompensateSize.width), ny(dp.imageSize.height / dp.patchCompensateSize.height),
		  sensorW(dp.imageSize.width), sensorH(dp.imageSize.height), pw(dp.patchCompensateSize.width),
		  ph(dp.patchCompensateSize.height)
	{
	}
	int count() const { return nx * ny; }
	int index(int gx, int gy) const { return gy * nx + gx; }
	// regular cells; the last column / row takes what is left of the sensor (:332-346)
	tracker::Rect2i cell(int gx, int gy) const
	{
		tracker::Rect2i r;
		r.x = gx * pw;
		r.y = gy * ph;
		r.width = gx + 1 == nx ? sensorW - r.x : pw;
		r.height = gy + 1 == ny ? sensorH - r.y : ph;
		return r;
	}
};

template <typename Functor>
static std::vector<double> solveThroughCeresSurface(const tracker::DetectorParams& dp, const std::list<common::EventSample>& events,
													ceres::Solver::Summary& summary, int& numBlocks)
{
	const GridOfPatches grid(dp);
	std::vector<double> flow(2 * static_cast<size_t>(grid.count()), 0.0);  // every solve starts at zero flow (:318-326)
	auto block = [&](int gx, int gy) { return flow.data() + 2 * grid.index(gx, gy); };
	auto coupling = [&](double* a, double* b, ceres::Problem& problem) {
		ceres::CostFunction* tv =
			new ceres::AutoDiffCostFunction<tracker::totalVarianceFunctor, 2, 2, 2>(new tracker::totalVarianceFunctor(dp.compensateTVweight));
		problem.AddResidualBlock(tv, new ceres::HuberLoss(dp.compensateTVHuberLoss), a, b);
	};

	ceres::Problem problem;
	for (int gy = 0; gy < grid.ny; ++gy)
	{
		for (int gx = 0; gx < grid.nx; ++gx)
		{
			const tracker::Rect2i rect = grid.cell(gx, gy);
			std::list<common::EventSample> inside;
			std::copy_if(events.begin(), events.end(), std::back_inserter(inside),
						 [&](const common::EventSample& e) { return rect.contains(e.value.point); });
			if (inside.size() > dp.compensateMinNumEvents)  // strictly more (:357)
			{
				ceres::CostFunction* data = new ceres::AutoDiffCostFunction<Functor, 1, 2>(new Functor(inside, rect, dp.compensateScale));
				problem.AddResidualBlock(data, nullptr, block(gx, gy));
			}
			if (gx + 1 < grid.nx)
			{
				coupling(block(gx, gy), block(gx + 1, gy), problem);  // right neighbour first, then the one below (:369-396)
			}
			if (gy + 1 < grid.ny)
			{
				coupling(block(gx, gy), block(gx, gy + 1), problem);
			}
		}
	}

	ceres::Solver::Options options;  // feature_detector.cpp:401-410
	options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
	options.use_nonmonotonic_steps = true;
	options.max_num_iterations = 50;
	options.function_tolerance = options.gradient_tolerance = options.parameter_tolerance = 1e-12;
	options.num_threads = 1;
	options.minimizer_progress_to_stdout = false;
	options.logging_type = ceres::SILENT;
	Solve(options, &problem, &summary);
	numBlocks = problem.NumResidualBlocks();
	return flow;
}

static void runCase(int loss, const char* name)
{
	const tracker::DetectorParams self;  // reference defaults: 240x180, 20x20, TV 1e3, Huber 10, scale 1e-3, > 100 events
	const std::vector<common::EventSample> samples = makeEvents(15000, 20200701);
	const std::list<common::EventSample> events(samples.begin(), samples.end());

	ceres::Solver::Summary summary;
	int numBlocks = 0;
	const std::vector<double> mf = loss == EBO_LOSS_EDGE
									   ? solveThroughCeresSurface<tracker::contrastFunctor>(self, events, summary, numBlocks)
									   : solveThroughCeresSurface<tracker::contrastFunctorVariance>(self, events, summary, numBlocks);

	// the same window through the C ABI's own global solve
	ebo_params prm;
	ebo_default_params(&prm);
	prm.loss = loss;
	prm.max_events = samples.size();
	ebo_ctx* ctx = nullptr;
	EXPECT_TRUE(ebo_create(&prm, &ctx) == EBO_OK);
	const std::vector<ebo_event> ev = common::toEboEvents(events);
	EXPECT_TRUE(ebo_set_window(ctx, ev.data(), ev.size()) == EBO_OK);
	std::vector<double> ref(mf.size(), 0.0);
	ebo_solver_opts opts;
	ebo_default_solver(&opts);
	ebo_summary sum;
	EXPECT_TRUE(ebo_solve(ctx, &opts, ref.data(), &sum) == EBO_OK);
	ebo_destroy(ctx);

	double maxd = 0.0;
	int differing = 0;
	for (size_t i = 0; i < mf.size(); ++i)
	{
		maxd = std::fmax(maxd, std::fabs(ref[i] - mf[i]));
		differing += std::memcmp(&ref[i], &mf[i], sizeof(double)) != 0;
	}
	std::printf("%s: %d residual blocks, %s; ebo_solve: %d iterations; max |flow - ebo_solve| = %.3e, %d of %zu flow "
				"components differ in any bit\n",
				name, numBlocks, summary.BriefReport().c_str(), sum.iterations, maxd, differing, mf.size());
	EXPECT_TRUE(numBlocks > 50 + 2 * 12 * 9 - 12 - 9);
	EXPECT_TRUE(summary.num_successful_steps == sum.iterations);
	EXPECT_TRUE(differing == 0);
}

// The Jet path of the functor against its double path and against finite differences of itself.
static void jetPathAgreesWithEvaluate()
{
	const std::vector<common::EventSample> samples = makeEvents(15000, 7);
	std::list<common::EventSample> patchEvents;
	const tracker::Rect2i rect(100, 80, 20, 20);
	for (const auto& e : samples)
	{
		if (rect.contains(e.value.point))
		{
			patchEvents.push_back(e);
		}
	}
	const tracker::contrastFunctor functor(patchEvents, rect, 1e-3);
	const double m[2] = {0.31, -0.17};
	double const* params[1] = {m};
	double r0 = 0.0, jac[2] = {0.0, 0.0};
	double* jacs[1] = {jac};
	EXPECT_TRUE(functor.Evaluate(params, &r0, jacs));
	// Jets with a general seed (3 partials): v_out = J0 v0 + J1 v1
	using J3 = ceres::Jet<double, 3>;
	J3 x[2] = {J3(m[0]), J3(m[1])};
	x[0].v[0] = 1.0;
	x[0].v[2] = 0.5;
	x[1].v[1] = 1.0;
	x[1].v[2] = -2.0;
	J3 out;
	EXPECT_TRUE(functor(x, &out));
	EXPECT_TRUE(out.a == r0);
	EXPECT_TRUE(out.v[0] == jac[0] && out.v[1] == jac[1]);
	EXPECT_TRUE(std::fabs(out.v[2] - (0.5 * jac[0] - 2.0 * jac[1])) <= 1e-15 * (std::fabs(jac[0]) + std::fabs(jac[1])));
	double rd = 0.0;
	EXPECT_TRUE(functor(m, &rd));  // the T = double overload: value only
	EXPECT_TRUE(std::fabs(rd - r0) <= 1e-12 * std::fabs(r0));
	std::printf("Jet path: r = %.15g, J = (%.6e, %.6e)\n", r0, jac[0], jac[1]);
}

int main()
{
	jetPathAgreesWithEvaluate();
	runCase(EBO_LOSS_EDGE, "reference statements, edge loss (AutoDiffCostFunction<tracker::contrastFunctor,1,2>)");
	runCase(EBO_LOSS_VARIANCE, "reference statements, variance loss");
	if (failures == 0)
	{
		std::printf("all passed\n");
		return 0;
	}
	std::printf("%d failures\n", failures);
	return 1;
}
