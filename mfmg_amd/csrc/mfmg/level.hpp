// Level<VectorType>: mirror of include/mfmg/common/level.hpp:22-76.
#pragma once

#include "operator.hpp"
#include "smoother.hpp"
#include "solver.hpp"

namespace mfmg
{
template <typename VectorType>
class Level
{
public:
  using vector_type = VectorType;
  using operator_type = Operator<VectorType>;

  std::shared_ptr<operator_type const> get_operator() const { return _operator; }

  std::shared_ptr<operator_type const> get_restrictor() const { return _restrictor; }

  std::shared_ptr<Smoother<vector_type> const> get_smoother() const { return _smoother; }

  std::shared_ptr<Solver<vector_type> const> get_solver() const { return _solver; }

  void set_operator(std::shared_ptr<operator_type const> op)
  {
    _operator = op;
    _workspace.clear();
  }

  void set_restrictor(std::shared_ptr<operator_type const> r) { _restrictor = r; }

  void set_smoother(std::shared_ptr<Smoother<vector_type> const> s) { _smoother = s; }

  void set_solver(std::shared_ptr<Solver<vector_type> const> s) { _solver = s; }

  std::shared_ptr<vector_type> build_vector() const
  {
    auto a = get_operator();
    ASSERT_THROW(a != nullptr, "build_vector() can only be called after the level operator was set.");
    return a->build_domain_vector();
  }

  // The reference heap-allocates its four temporaries per level per cycle
  // (level.hpp:63-70 via hierarchy.hpp:284-297); here they are built once and reused.
  std::shared_ptr<vector_type> workspace_vector(unsigned int slot) const
  {
    if (_workspace.size() <= slot)
      _workspace.resize(slot + 1);
    if (!_workspace[slot])
      _workspace[slot] = build_vector();
    return _workspace[slot];
  }

private:
  std::shared_ptr<operator_type const> _operator, _restrictor;
  std::shared_ptr<Smoother<vector_type> const> _smoother;
  std::shared_ptr<Solver<vector_type> const> _solver;
  mutable std::vector<std::shared_ptr<vector_type>> _workspace;
};
} // namespace mfmg
