// Solver<VectorType>: mirror of include/mfmg/common/solver.hpp:23-42.
#pragma once

#include "operator.hpp"
#include "ptree.hpp"

namespace mfmg
{
template <typename VectorType>
class Solver
{
public:
  using vector_type = VectorType;
  using operator_type = Operator<vector_type>;

  Solver(std::shared_ptr<operator_type const> op, std::shared_ptr<ptree const> params)
      : _operator(op), _params(params)
  {
  }
  virtual ~Solver() = default;

  virtual void apply(vector_type const &b, vector_type &x) const = 0;
  // true when apply() overwrites x without reading it: the hierarchy then skips its x = 0 (hierarchy.hpp:253-259)
  // in front of the coarsest-level solve
  virtual bool ignores_initial_guess() const { return false; }

protected:
  std::shared_ptr<operator_type const> _operator;
  std::shared_ptr<ptree const> _params;
};
} // namespace mfmg
