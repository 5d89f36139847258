// Operator<VectorType>: mirror of include/mfmg/common/operator.hpp:19-52.
// Two non-pure extensions let the concrete HIP operators fuse the vector updates
// that the reference performs as separate passes.
#pragma once

#include <memory>

#include "../common.hpp"

namespace mfmg
{
enum class OperatorMode
{
  NO_TRANS,
  TRANS
};

template <typename VectorType>
class Operator
{
public:
  using operator_type = Operator<VectorType>;
  using vector_type = VectorType;

  virtual ~Operator() = default;

  virtual void apply(vector_type const &x, vector_type &y, OperatorMode mode = OperatorMode::NO_TRANS) const = 0;

  virtual std::shared_ptr<operator_type> transpose() const = 0;

  virtual std::shared_ptr<operator_type> multiply(std::shared_ptr<operator_type const> b) const = 0;

  virtual std::shared_ptr<operator_type> multiply_transpose(std::shared_ptr<operator_type const> b) const = 0;

  virtual std::shared_ptr<vector_type> build_domain_vector() const = 0;

  virtual std::shared_ptr<vector_type> build_range_vector() const = 0;

  virtual size_t grid_complexity() const = 0;

  virtual size_t operator_complexity() const = 0;

  // res = A x - b : `a->apply(x, *res); res->add(-1., b);` of hierarchy.hpp:284-286
  virtual void residual(vector_type const &x, vector_type const &b, vector_type &res) const
  {
    apply(x, res);
    res.add(-1., b);
  }
  // Called on a RESTRICTOR: b_coarse = R (A x - b), the residual and its restriction (hierarchy.hpp:281-290) in one
  // pass where the restrictor knows the rows of R A (structured_restrictor.hpp).  false: not available, nothing done.
  virtual bool restrict_residual(operator_type const & /*a*/, vector_type const & /*x*/, vector_type const & /*b*/,
                                 vector_type & /*b_coarse*/) const
  {
    return false;
  }
  // Called on a RESTRICTOR at the start of a cycle, with the right-hand side the cycle will restrict later: a distributed run
  // may refresh the ghost entries of b meanwhile (restrict_residual finds them in place).  Default: nothing.
  virtual void prefetch_rhs(vector_type const & /*b*/) const {}
  // ... and the end of that cycle: nothing of it may still be in flight, nothing is known about b any more
  virtual void release_rhs() const {}
  // y -= op(A) x : `restrictor->apply(*x_coarse, *x_correction, TRANS); x.add(-1., *x_correction);`
  // of hierarchy.hpp:297-302
  virtual void apply_subtract(vector_type const &x, vector_type &y, OperatorMode mode) const
  {
    auto tmp = (mode == OperatorMode::NO_TRANS) ? build_range_vector() : build_domain_vector();
    apply(x, *tmp, mode);
    y.add(-1., *tmp);
  }
};
} // namespace mfmg
