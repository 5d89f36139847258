// HierarchyHelpers<VectorType>: mirror of include/mfmg/common/hierarchy_helpers.hpp:29-58.
// The MPI_Comm argument of build_restrictor becomes an opaque communicator handle:
// one process per GPU, partition described by the evaluator.
#pragma once

#include "mesh_evaluator.hpp"
#include "operator.hpp"
#include "ptree.hpp"
#include "smoother.hpp"
#include "solver.hpp"

namespace mfmg
{
using Communicator = void *; // stands in for MPI_Comm (hierarchy.hpp:159)

template <typename VectorType>
class HierarchyHelpers
{
public:
  using vector_type = VectorType;

  virtual ~HierarchyHelpers() = default;

  virtual std::shared_ptr<Operator<vector_type>>
  get_global_operator(std::shared_ptr<MeshEvaluator> mesh_evaluator) = 0;

  virtual std::shared_ptr<Operator<vector_type>> build_restrictor(Communicator comm,
                                                                   std::shared_ptr<MeshEvaluator> mesh_evaluator,
                                                                   std::shared_ptr<ptree const> params) = 0;

  virtual std::shared_ptr<Operator<vector_type>> fast_multiply_transpose()
  {
    ASSERT_THROW_NOT_IMPLEMENTED();
    return nullptr;
  }

  virtual std::shared_ptr<Smoother<vector_type>> build_smoother(std::shared_ptr<Operator<vector_type> const> op,
                                                                 std::shared_ptr<ptree const> params) = 0;

  virtual std::shared_ptr<Solver<vector_type>>
  build_coarse_solver(std::shared_ptr<Operator<vector_type> const> op, std::shared_ptr<ptree const> params) = 0;

  // Extension: the restrictor the coarse operator was built with, so that a multilevel coarse solver can
  // derive its near-null-space vector (the coarse image of the constant) from it.  Default: ignored.
  virtual void set_coarse_space_hint(std::shared_ptr<Operator<vector_type> const> /*restrictor*/) {}
  // Extension: lets a restrictor prepare `restrict_residual` for the operator of its fine level.  Default: nothing.
  virtual void prepare_residual_restriction(std::shared_ptr<Operator<vector_type> const> /*a*/,
                                            std::shared_ptr<Operator<vector_type>> /*restrictor*/,
                                            std::shared_ptr<ptree const> /*params*/)
  {
  }
};
} // namespace mfmg
