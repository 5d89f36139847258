// Smoother<VectorType>: mirror of include/mfmg/common/smoother.hpp:23-42.
#pragma once

#include "operator.hpp"
#include "ptree.hpp"

namespace mfmg
{
template <typename VectorType>
class Smoother
{
public:
  using vector_type = VectorType;
  using operator_type = Operator<vector_type>;

  Smoother(std::shared_ptr<operator_type const> op, std::shared_ptr<ptree const> params)
      : _operator(op), _params(params)
  {
  }
  virtual ~Smoother() = default;

  virtual void apply(vector_type const &b, vector_type &x) const = 0;

  // Extension: the same update from x_in into a DIFFERENT vector x_out.  A smoother that runs its whole polynomial in one sweep
  // over the mesh cannot write into the vector it reads; Hierarchy::apply alternates between x and a workspace vector when
  // prefers_out_of_place() says so (an even number of applications per cycle: the result lands in x without a copy).
  virtual void apply_to(vector_type const &b, vector_type const &x_in, vector_type &x_out) const
  {
    x_out = x_in;
    apply(b, x_out);
  }
  virtual bool prefers_out_of_place() const { return false; }
  // Extension: x_out = the update from x = 0 WITHOUT a zeroed input vector (the pre-smoother of a preconditioner application,
  // hierarchy.hpp:253-259).  False (nothing done): the caller zeroes x and applies as usual.
  virtual bool apply_from_zero(vector_type const & /*b*/, vector_type & /*x_out*/) const { return false; }

protected:
  std::shared_ptr<operator_type const> _operator;
  std::shared_ptr<ptree const> _params;
};
} // namespace mfmg
