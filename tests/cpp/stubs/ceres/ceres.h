// TEST-ONLY stand-in for <ceres/ceres.h>: the declarations of Ceres Solver's PUBLIC interface that
// event-based-odomety_amd/include/feature_tracker/contrast_functor.h compiles against when Ceres
// is present (its EBO_HAVE_CERES branch), written from the published API documentation
// (ceres-solver.org: "Modeling Non-linear Least Squares" -- CostFunction, SizedCostFunction,
// EvaluationCallback, Problem::Options::evaluation_callback, Problem::AddResidualBlock).  Ceres is
// not installed in this image, so without this file that branch never meets a compiler.  Nothing
// here solves anything: tests/cpp/ceres_adaptor_test.cpp drives the PRODUCT's own host LM through
// these interfaces.  Not used to build the reference, not part of the product.
//
// Round 3: also what the reference's OWN construction lines need (feature_detector.cpp:316-414) --
// ceres::Jet (ceres/jet.h: a scalar part `a` and an N-vector of partials `v`, the published
// operator formulas), ceres::AutoDiffCostFunction (ceres/autodiff_cost_function.h: seeds one Jet
// per parameter, calls the functor's templated operator(), unpacks the partials into the row-major
// Jacobian blocks), ceres::LossFunction / HuberLoss (ceres/loss_function.h), Problem::AddResidualBlock
// with two parameter blocks, Solver::Options / Summary and the declaration of ceres::Solve (DEFINED
// by the test program, which drives the product's host LM; this header solves nothing).
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ceres
{
// ceres/evaluation_callback.h
class EvaluationCallback
{
   public:
	virtual ~EvaluationCallback() {}
	// Called before Ceres requests residuals or Jacobians for a given setting of the parameters.
	virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

// ceres/cost_function.h
class CostFunction
{
   public:
	CostFunction() : num_residuals_(0) {}
	CostFunction(const CostFunction&) = delete;
	void operator=(const CostFunction&) = delete;
	virtual ~CostFunction() {}
	virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
	const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
	int num_residuals() const { return num_residuals_; }

   protected:
	std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
	void set_num_residuals(int num_residuals) { num_residuals_ = num_residuals; }

   private:
	std::vector<int32_t> parameter_block_sizes_;
	int num_residuals_;
};

// ceres/sized_cost_function.h
template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction
{
   public:
	SizedCostFunction()
	{
		set_num_residuals(kNumResiduals);
		*mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
	}
	virtual ~SizedCostFunction() {}
};

// ceres/types.h
enum Ownership
{
	DO_NOT_TAKE_OWNERSHIP,
	TAKE_OWNERSHIP
};
enum LoggingType
{
	SILENT,
	PER_MINIMIZER_ITERATION
};
enum LinearSolverType
{
	DENSE_NORMAL_CHOLESKY,
	DENSE_QR,
	SPARSE_NORMAL_CHOLESKY
};

// ceres/jet.h: f = a + sum_i v[i] * e_i, e_i * e_j = 0.
template <typename T, int N>
struct Jet
{
	struct Partials  // Eigen::Matrix<T, N, 1> in Ceres; only operator[] / size() are relied upon
	{
		T d[N];
		T& operator[](int i) { return d[i]; }
		const T& operator[](int i) const { return d[i]; }
		static constexpr int size() { return N; }
	};
	enum
	{
		DIMENSION = N
	};
	T a;
	Partials v;

	Jet() : a()
	{
		for (int i = 0; i < N; ++i) v[i] = T();
	}
	Jet(const T& value) : a(value)  // NOLINT: implicit, as in Ceres
	{
		for (int i = 0; i < N; ++i) v[i] = T();
	}
	Jet(const T& value, int k) : a(value)
	{
		for (int i = 0; i < N; ++i) v[i] = T();
		v[k] = T(1.0);
	}
	Jet& operator+=(const Jet& y)
	{
		*this = *this + y;
		return *this;
	}
	Jet& operator-=(const Jet& y)
	{
		*this = *this - y;
		return *this;
	}
};
template <typename T, int N>
inline Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g)
{
	Jet<T, N> h(f.a + g.a);
	for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i];
	return h;
}
template <typename T, int N>
inline Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g)
{
	Jet<T, N> h(f.a - g.a);
	for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i];
	return h;
}
template <typename T, int N>
inline Jet<T, N> operator-(const Jet<T, N>& f)
{
	Jet<T, N> h(-f.a);
	for (int i = 0; i < N; ++i) h.v[i] = -f.v[i];
	return h;
}
template <typename T, int N>
inline Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g)
{
	Jet<T, N> h(f.a * g.a);
	for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
	return h;
}
template <typename T, int N>
inline Jet<T, N> operator*(const Jet<T, N>& f, T s)
{
	Jet<T, N> h(f.a * s);
	for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s;
	return h;
}
template <typename T, int N>
inline Jet<T, N> operator*(T s, const Jet<T, N>& f)
{
	Jet<T, N> h(s * f.a);
	for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i];
	return h;
}
template <typename T, int N>
inline Jet<T, N> operator+(const Jet<T, N>& f, T s)
{
	Jet<T, N> h(f);
	h.a = f.a + s;
	return h;
}
template <typename T, int N>
inline Jet<T, N> abs(const Jet<T, N>& f)  // jet.h: abs(a + h) ~= abs(a) + sgn(a) h
{
	Jet<T, N> h(std::abs(f.a));
	const T sgn = std::copysign(T(1.0), f.a);
	for (int i = 0; i < N; ++i) h.v[i] = sgn * f.v[i];
	return h;
}
inline double abs(double x) { return std::abs(x); }

// ceres/types.h: the residual count of a cost function whose size is given at run time
enum
{
	DYNAMIC = -1
};

// ceres/autodiff_cost_function.h (static parameter-block sizes; kNumResiduals static or DYNAMIC with the count given to
// the constructor, as `AutoDiffCostFunction<F, ceres::DYNAMIC, 4, 1>(functor, size)` of optimizer.cpp:91-97).  Evaluate without Jacobians calls the functor
// on doubles; with Jacobians it evaluates the functor once on Jet<double, N0 + N1 + ...>, parameter
// j of block i seeded with the unit partial offset_i + j, and copies output[k].v[offset_i + j] into
// jacobians[i][k * Ni + j] for every block whose Jacobian was asked for.
template <typename CostFunctor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...>
{
	static constexpr int kNumBlocks = sizeof...(Ns);
	static constexpr int kTotal = (Ns + ...);
	using JetT = Jet<double, kTotal>;

   public:
	explicit AutoDiffCostFunction(CostFunctor* functor, Ownership ownership = TAKE_OWNERSHIP)
		: functor_(functor), ownership_(ownership)
	{
		static_assert(kNumResiduals != DYNAMIC, "a DYNAMIC cost function takes its residual count in the constructor");
	}
	AutoDiffCostFunction(CostFunctor* functor, int num_residuals, Ownership ownership = TAKE_OWNERSHIP)
		: functor_(functor), ownership_(ownership)
	{
		static_assert(kNumResiduals == DYNAMIC, "only a DYNAMIC cost function takes a residual count");
		this->set_num_residuals(num_residuals);
	}
	~AutoDiffCostFunction() override
	{
		if (ownership_ == DO_NOT_TAKE_OWNERSHIP)
		{
			functor_.release();
		}
	}
	bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override
	{
		if (jacobians == nullptr)
		{
			return call(parameters, residuals, std::make_index_sequence<kNumBlocks>());
		}
		const int sizes[kNumBlocks] = {Ns...};
		JetT x[kTotal];
		const JetT* blocks[kNumBlocks];
		int offset = 0;
		for (int i = 0; i < kNumBlocks; ++i)
		{
			blocks[i] = &x[offset];
			for (int j = 0; j < sizes[i]; ++j)
			{
				x[offset + j] = JetT(parameters[i][j], offset + j);
			}
			offset += sizes[i];
		}
		const int nres = this->num_residuals();
		std::vector<JetT> outv(static_cast<size_t>(nres));
		JetT* out = outv.data();
		if (!call(blocks, out, std::make_index_sequence<kNumBlocks>()))
		{
			return false;
		}
		for (int k = 0; k < nres; ++k)
		{
			residuals[k] = out[k].a;
		}
		offset = 0;
		for (int i = 0; i < kNumBlocks; ++i)
		{
			if (jacobians[i] != nullptr)
			{
				for (int k = 0; k < nres; ++k)
				{
					for (int j = 0; j < sizes[i]; ++j)
					{
						jacobians[i][k * sizes[i] + j] = out[k].v[offset + j];
					}
				}
			}
			offset += sizes[i];
		}
		return true;
	}
	const CostFunctor& functor() const { return *functor_; }

   private:
	template <typename T, size_t... I>
	bool call(T const* const* blocks, T* out, std::index_sequence<I...>) const
	{
		return (*functor_)(blocks[I]..., out);
	}
	std::unique_ptr<CostFunctor> functor_;
	Ownership ownership_;
};

// ceres/loss_function.h: out = (rho(s), rho'(s), rho''(s)), s = squared norm of the residual block
class LossFunction
{
   public:
	virtual ~LossFunction() {}
	virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunction
{
   public:
	explicit HuberLoss(double a) : a_(a), b_(a * a) {}
	void Evaluate(double s, double rho[3]) const override
	{
		if (s > b_)
		{
			const double r = std::sqrt(s);  // rho(s) = 2 a sqrt(s) - a^2 outside the inlier region
			rho[0] = 2.0 * a_ * r - b_;
			rho[1] = std::fmax(std::numeric_limits<double>::min(), a_ / r);
			rho[2] = -rho[1] / (2.0 * s);
		}
		else
		{
			rho[0] = s;
			rho[1] = 1.0;
			rho[2] = 0.0;
		}
	}
	double a() const { return a_; }  // test driver only

   private:
	const double a_, b_;
};

// ceres/problem.h -- only what builds the problem of feature_detector.cpp:316-396
using ResidualBlockId = const void*;
class Problem
{
   public:
	struct Options
	{
		EvaluationCallback* evaluation_callback = nullptr;
	};
	Problem() {}
	explicit Problem(const Options& options) : options_(options) {}
	template <typename... Ts>
	ResidualBlockId AddResidualBlock(CostFunction* cost_function, LossFunction* loss_function, double* x0, Ts*... xs)
	{
		Block b;  // TAKE_OWNERSHIP of cost and loss, the default
		b.cost.reset(cost_function);
		b.loss.reset(loss_function);
		b.x = x0;
		b.params = std::vector<double*>{x0, xs...};
		blocks_.push_back(std::move(b));
		return blocks_.back().cost.get();
	}
	int NumResidualBlocks() const { return static_cast<int>(blocks_.size()); }

	// what a solver does with the blocks (test driver only)
	struct Block
	{
		std::unique_ptr<CostFunction> cost;
		std::unique_ptr<LossFunction> loss;
		double* x = nullptr;
		std::vector<double*> params;
	};
	const std::vector<Block>& blocks() const { return blocks_; }
	const Options& options() const { return options_; }

   private:
	Options options_;
	std::vector<Block> blocks_;
};

// ceres/solver.h: the options feature_detector.cpp:401-410 sets, the summary :412-416 reads
class Solver
{
   public:
	struct Options
	{
		bool minimizer_progress_to_stdout = false;
		int num_threads = 1;
		LoggingType logging_type = PER_MINIMIZER_ITERATION;
		LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
		bool use_nonmonotonic_steps = false;
		int max_num_iterations = 50;
		double function_tolerance = 1e-6;
		double gradient_tolerance = 1e-10;
		double parameter_tolerance = 1e-8;
	};
	struct Summary
	{
		int num_successful_steps = 0, num_unsuccessful_steps = 0;
		int num_residual_evaluations = 0, num_jacobian_evaluations = 0;
		double initial_cost = 0.0, final_cost = 0.0;
		int termination_type = 1;
		std::string BriefReport() const
		{
			return "iterations " + std::to_string(num_successful_steps + num_unsuccessful_steps) + ", initial cost " +
				   std::to_string(initial_cost) + ", final cost " + std::to_string(final_cost);
		}
	};
};
// Declared only: the test program defines it (the product's host LM behind the Ceres call).
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);
}  // namespace ceres
