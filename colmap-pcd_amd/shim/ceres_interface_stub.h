// ceres_interface_stub.h -- the two abstract interfaces of Ceres Solver that shim/ceres_adapter.h implements,
// declared here ONLY so that the adapter compiles and is exercised where Ceres is not installed (this image has no
// Ceres headers).  Declarations of ceres/cost_function.h and ceres/evaluation_callback.h (Ceres 2.1, the version
// the reference's README names), nothing else: no solver, no autodiff, no arithmetic.  With Ceres present,
// ceres_adapter.h includes <ceres/ceres.h> instead and this file is not used.
#pragma once
#include <cstdint>
#include <vector>

namespace ceres {

class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  // parameters[i]: block i; residuals: num_residuals(); jacobians: NULL, or per block NULL (constant block) or a
  // row-major num_residuals x parameter_block_sizes()[i] array
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

class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  // called once before each batch of CostFunction::Evaluate calls; the user's parameter blocks hold the evaluation
  // point at that moment (Problem::Options::evaluation_callback)
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

}  // namespace ceres
