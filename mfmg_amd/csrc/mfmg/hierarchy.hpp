// Hierarchy<VectorType>: mirror of include/mfmg/common/hierarchy.hpp:155-373.
// The constructor follows the reference line by line in *structure* (operator ->
// smoother -> restrictor -> A R^T -> R (A R^T) -> coarse solver, hierarchy.hpp:159-236)
// and apply() is the same recursive V-cycle with the negative-residual convention
// (hierarchy.hpp:246-309); what changes is what runs underneath: fused HIP kernels,
// temporaries built once per level, no host round trips.
#pragma once

#include "hierarchy_helpers.hpp"
#include "hip_hierarchy_helpers.hpp"
#include "level.hpp"
#include "timer.hpp"

namespace mfmg
{
// String switch on the evaluator tag (include/mfmg/common/hierarchy.hpp:49-153).
template <typename VectorType>
std::unique_ptr<HierarchyHelpers<VectorType>> create_hierarchy_helpers(std::shared_ptr<MeshEvaluator> evaluator)
{
  std::unique_ptr<HierarchyHelpers<VectorType>> hierarchy_helpers;
  std::string evaluator_type = evaluator->get_mesh_evaluator_type();
  if (evaluator_type == "HipMeshEvaluator" || evaluator_type == "HipMatrixFreeMeshEvaluator")
  {
    auto hip_evaluator = std::dynamic_pointer_cast<HipMeshEvaluator>(evaluator);
    ASSERT_THROW(hip_evaluator != nullptr, "downcasting failed");
    hierarchy_helpers.reset(new HipHierarchyHelpers<VectorType>(hip_evaluator->get_hip_handle()));
  }
  else
  {
    // "DealIIMeshEvaluator", "DealIIMatrixFreeMeshEvaluator", "CudaMeshEvaluator" belong to
    // back-ends that are not part of this build.
    ASSERT_THROW_NOT_IMPLEMENTED("mesh evaluator type \"" + evaluator_type + "\" is not available in the HIP build");
  }
  return hierarchy_helpers;
}

template <typename VectorType>
class Hierarchy
{
public:
  Hierarchy(Communicator comm, std::shared_ptr<MeshEvaluator> evaluator, std::shared_ptr<ptree> params = nullptr,
            std::shared_ptr<TimerOutput> timer = nullptr)
      : _timer(timer)
  {
    timer_enter_subsection(_timer, "Setup");
    if (!params)
      params = std::make_shared<ptree>();
    _helpers = create_hierarchy_helpers<VectorType>(evaluator);
    auto &hierarchy_helpers = _helpers;

    _is_preconditioner = params->get("is preconditioner", true);
    _n_smoothing_steps = params->get("smoother.n_smoothing_steps", 1);

    int const num_levels = params->get("max levels", 2);
    ASSERT_THROW(num_levels > 0, "number of levels specified by \"max levels\" parameter must be positive");
    _levels.resize(num_levels);

    _levels[0].set_operator(hierarchy_helpers->get_global_operator(evaluator));
    for (int level_index = 0; level_index < num_levels; level_index++)
    {
      auto &level_fine = _levels[level_index];
      auto a = level_fine.get_operator();

      if (level_index == num_levels - 1)
      {
        if (level_index == 0)
          params->put("coarse.params.zero starting solution", _is_preconditioner);

        timer_enter_subsection(_timer, "Setup: build coarse solver");
        auto coarse_solver = hierarchy_helpers->build_coarse_solver(a, params);
        level_fine.set_solver(coarse_solver);
        timer_leave_subsection(_timer);
        break;
      }

      auto &level_coarse = _levels[level_index + 1];

      timer_enter_subsection(_timer, "Setup: build smoother");
      auto smoother = hierarchy_helpers->build_smoother(a, params);
      level_fine.set_smoother(smoother);
      timer_leave_subsection(_timer);

      timer_enter_subsection(_timer, "Setup: build restrictor");
      auto restrictor = hierarchy_helpers->build_restrictor(comm, evaluator, params);
      level_coarse.set_restrictor(restrictor);
      hierarchy_helpers->set_coarse_space_hint(restrictor);
      hierarchy_helpers->prepare_residual_restriction(a, restrictor, params);
      timer_leave_subsection(_timer);

      std::shared_ptr<Operator<VectorType>> ap;
      bool fast_ap = params->get("fast_ap", false);
      if (fast_ap)
      {
        timer_enter_subsection(_timer, "Setup: fast_ap");
        ap = hierarchy_helpers->fast_multiply_transpose();
        timer_leave_subsection(_timer);
      }
      else
      {
        timer_enter_subsection(_timer, "Setup: ap");
        ap = a->multiply_transpose(restrictor);
        timer_leave_subsection(_timer);
      }

      timer_enter_subsection(_timer, "Setup: build coarse matrix");
      auto a_coarse = restrictor->multiply(ap);
      timer_leave_subsection(_timer);
      // Extension ("keep_ap", default false): A R^T of every level stays reachable (ap_operators()), so that a test
      // can compare it with another way of forming it, as tests/test_hierarchy.cc:507-642 does.
      if (params->get("keep_ap", false))
        _ap_operators.push_back(ap);

      level_coarse.set_operator(a_coarse);
    }
    _params = params;
    timer_leave_subsection(_timer);
  }

  void vmult(VectorType &x, VectorType const &b) const
  {
    timer_enter_subsection(_timer, "Apply");
    apply(b, x, 0);
    timer_leave_subsection(_timer);
  }

  void apply(VectorType const &b, VectorType &x, int level_index = 0) const
  {
    auto const num_levels = _levels.size();

    auto &level_fine = _levels[level_index];
    auto a = level_fine.get_operator();

    const bool coarsest = level_index == static_cast<int>(num_levels) - 1;
    const bool from_zero = level_index > 0 || _is_preconditioner;
    // Zero out any garbage in x (hierarchy.hpp:253-259); a solver that does not read x needs no pass over it, and neither does a
    // smoother that can start from zero by itself (Smoother::apply_from_zero, tried below).
    bool zero_pending = from_zero && !(coarsest && level_fine.get_solver()->ignores_initial_guess());
    if (zero_pending && (coarsest || _n_smoothing_steps == 0))
    {
      x = 0.;
      zero_pending = false;
    }

    if (coarsest)
    {
      timer_enter_subsection(_timer, "Apply: coarsest level");
      auto coarse_solver = level_fine.get_solver();
      coarse_solver->apply(b, x);
      timer_leave_subsection(_timer);
    }
    else
    {
      timer_enter_subsection(_timer, "Apply: fine levels");
      auto &level_coarse = _levels[level_index + 1];
      auto restrictor = level_coarse.get_restrictor();

      // (distributed runs: the ghost entries of b, which only the restriction of the residual reads, travel beside the
      // pre-smoother instead of in front of the restriction)
      restrictor->prefetch_rhs(b);

      // pre-smoother.  A smoother that works out of place (the multi-term sweep: it cannot write the vector it reads) alternates
      // between x and a workspace vector; pre- and post-smoothing together switch an even number of times.
      auto smoother = level_fine.get_smoother();
      VectorType *cur = &x, *other = nullptr;
      unsigned int first_step = 0;
      if (zero_pending)
      {
        // the first pre-smoothing step from x = 0: the smoother writes its result into the workspace vector without reading x
        bool done = false;
        if (smoother->prefers_out_of_place())
        {
          other = level_fine.workspace_vector(3).get();
          done = smoother->apply_from_zero(b, *other);
        }
        if (done)
        {
          std::swap(cur, other);
          first_step = 1;
        }
        else
          x = 0.;
      }
      auto smooth = [&] {
        for (unsigned int i = first_step; i < _n_smoothing_steps; ++i)
        {
          if (smoother->prefers_out_of_place())
          {
            if (other == nullptr)
              other = level_fine.workspace_vector(3).get();
            smoother->apply_to(b, *cur, *other);
            std::swap(cur, other);
          }
          else
            smoother->apply(b, *cur);
        }
      };
      smooth();
      first_step = 0;

      // negative residual -r = A x - b and its restriction: one pass where the restrictor holds the rows of R A,
      // otherwise one fused kernel for the residual and the restriction after it
      auto b_coarse = level_coarse.workspace_vector(1);
      if (!restrictor->restrict_residual(*a, *cur, b, *b_coarse))
      {
        auto res = level_fine.workspace_vector(0);
        a->residual(*cur, b, *res);
        restrictor->apply(*res, *b_coarse);
      }

      // coarse grid correction
      auto x_coarse = level_coarse.workspace_vector(2);
      apply(*b_coarse, *x_coarse, level_index + 1);

      // x -= R^T x_c (prolongation fused with the update)
      restrictor->apply_subtract(*x_coarse, *cur, OperatorMode::TRANS);

      // post-smoother
      smooth();
      if (cur != &x)
        x = *cur;
      restrictor->release_rhs();
      timer_leave_subsection(_timer);
    }
  }

  // Replace R on level 1 and re-derive A_c and the coarse solver (same sequence as
  // the constructor, hierarchy.hpp:209-233,193-196): lets CPU and GPU runs share one R.
  void set_restrictor(std::shared_ptr<Operator<VectorType>> restrictor)
  {
    ASSERT_THROW(_levels.size() == 2, "set_restrictor supports the two-level hierarchy only");
    auto a = _levels[0].get_operator();
    _levels[1].set_restrictor(restrictor);
    _helpers->prepare_residual_restriction(a, restrictor, _params);
    auto ap = a->multiply_transpose(restrictor);
    auto a_coarse = restrictor->multiply(ap);
    _levels[1].set_operator(a_coarse);
    _helpers->set_coarse_space_hint(restrictor);
    _levels[1].set_solver(_helpers->build_coarse_solver(a_coarse, _params));
  }

  // stubbed to 0 upstream (hierarchy.hpp:311-366)
  double grid_complexity() const { return 0; }
  double operator_complexity() const { return 0; }

  std::vector<Level<VectorType>> const &levels() const { return _levels; }
  bool is_preconditioner() const { return _is_preconditioner; }
  unsigned int n_smoothing_steps() const { return _n_smoothing_steps; }
  std::shared_ptr<TimerOutput> timer() const { return _timer; }
  std::vector<std::shared_ptr<Operator<VectorType>>> const &ap_operators() const { return _ap_operators; }

private:
  std::shared_ptr<TimerOutput> _timer;
  std::unique_ptr<HierarchyHelpers<VectorType>> _helpers;
  std::shared_ptr<ptree> _params;
  std::vector<Level<VectorType>> _levels;
  std::vector<std::shared_ptr<Operator<VectorType>>> _ap_operators; // only with "keep_ap"
  bool _is_preconditioner = true;
  unsigned int _n_smoothing_steps;
};
} // namespace mfmg
