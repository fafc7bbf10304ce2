// TEST-ONLY stand-in for <ceres/ceres.h>: the declarations of Ceres Solver's PUBLIC interface that
// event-based-odomety_amd/include/feature_tracker/contrast_functor.h compiles against when Ceres
// is present (its EBO_HAVE_CERES branch), written from the published API documentation
// (ceres-solver.org: "Modeling Non-linear Least Squares" -- CostFunction, SizedCostFunction,
// EvaluationCallback, Problem::Options::evaluation_callback, Problem::AddResidualBlock).  Ceres is
// not installed in this image, so without this file that branch never meets a compiler.  Nothing
// here solves anything: tests/cpp/ceres_adaptor_test.cpp drives the PRODUCT's own host LM through
// these interfaces.  Not used to build the reference, not part of the product.
#pragma once

#include <cstdint>
#include <memory>
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

class LossFunction;

// ceres/problem.h -- only what builds the problem of feature_detector.cpp:357-367
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
	ResidualBlockId AddResidualBlock(CostFunction* cost_function, LossFunction* /*loss_function*/, double* x0)
	{
		blocks_.push_back(Block{std::unique_ptr<CostFunction>(cost_function), x0});  // TAKE_OWNERSHIP, the default
		return blocks_.back().cost.get();
	}
	int NumResidualBlocks() const { return static_cast<int>(blocks_.size()); }

	// what a solver does with the blocks (test driver only)
	struct Block
	{
		std::unique_ptr<CostFunction> cost;
		double* x;
	};
	const std::vector<Block>& blocks() const { return blocks_; }
	const Options& options() const { return options_; }

   private:
	Options options_;
	std::vector<Block> blocks_;
};
}  // namespace ceres
