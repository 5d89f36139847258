// The HIP back-end behind the mfmg operator API: twins of the classes of
// include/mfmg/cuda/ (CudaMeshEvaluator, CudaMatrixFreeMeshEvaluator,
// CudaMatrixOperator, CudaMatrixFreeOperator, CudaSmoother, CudaSolver,
// CudaHierarchyHelpers) for VectorType = mfmg::Vector<double>.
#pragma once

#include "../amg_setup.hpp"
#include "../amge_structured.hpp"
#include "../mf_laplace.hpp"
#include "../sparse_matrix_device.hpp"
#include "../structured_restrictor.hpp"
#include "hierarchy_helpers.hpp"
#include "vector.hpp"

namespace mfmg
{
using DVector = Vector<double>;

// ---- evaluators ------------------------------------------------------------
// Twin of CudaMeshEvaluator<dim> (include/mfmg/cuda/cuda_mesh_evaluator.cuh:30-72): the
// user-facing object that knows the mesh.  deal.II's DoFHandler/AffineConstraints are
// replaced by the plain mesh description of mfmg_hip_mesh_desc.
class HipMeshEvaluator : public MeshEvaluator
{
public:
  HipMeshEvaluator(HipHandle &handle, mfmg_hip_mesh_desc const &mesh);
  ~HipMeshEvaluator() override = default;

  int get_dim() const override { return _mesh.dim; }
  std::string get_mesh_evaluator_type() const override { return "HipMeshEvaluator"; }

  HipHandle &get_hip_handle() const { return _handle; }
  StructuredMesh const &get_mesh() const { return _mesh; }
  mfmg_hip_mesh_desc const &get_mesh_desc() const { return _desc; }

  // evaluate_global (cuda_mesh_evaluator.cuh:39-44): the assembled system matrix on the device.
  // Default: the Q1 Laplace of tests/laplace.hpp:154-204 assembled from the coefficient table.
  virtual std::shared_ptr<SparseMatrixDevice<double>> evaluate_global() const;
  // get_locally_relevant_diag (cuda_mesh_evaluator.cuh:57-60)
  virtual std::vector<double> get_locally_relevant_diag() const;
  // how the agglomerate eigenproblems are posed (evaluate_agglomerate, cuda_mesh_evaluator.cuh:46-55)
  virtual RestrictorOptions agglomerate_options(ptree const &params) const;
  virtual ConstraintSemantics constraint_semantics() const { return ConstraintSemantics::assembled; }

protected:
  HipHandle &_handle;
  mfmg_hip_mesh_desc _desc;
  StructuredMesh _mesh;
};

// Twin of CudaMatrixFreeMeshEvaluator<dim> (include/mfmg/cuda/cuda_matrix_free_mesh_evaluator.cuh:25-97),
// whose hooks throw upstream; here they run the HIP Laplace kernels.
class HipMatrixFreeMeshEvaluator : public HipMeshEvaluator
{
public:
  HipMatrixFreeMeshEvaluator(HipHandle &handle, mfmg_hip_mesh_desc const &mesh);

  std::string get_mesh_evaluator_type() const override { return "HipMatrixFreeMeshEvaluator"; }

  std::shared_ptr<DVector> build_range_vector() const;
  virtual void matrix_free_evaluate_global(DVector const &src, DVector &dst) const;
  virtual double const *matrix_free_get_diagonal_inverse() const; // device pointer
  virtual std::vector<double> get_diagonal() const;               // host copy, constrained entries 1
  std::vector<double> get_locally_relevant_diag() const override { return get_diagonal(); }
  RestrictorOptions agglomerate_options(ptree const &params) const override;
  ConstraintSemantics constraint_semantics() const override { return ConstraintSemantics::matrix_free; }

  std::shared_ptr<MatrixFreeLaplaceDevice<double>> get_device_operator() const { return _op; }

private:
  std::shared_ptr<MatrixFreeLaplaceDevice<double>> _op;
};

template <>
struct is_matrix_free<HipMatrixFreeMeshEvaluator> : std::true_type
{
};

// ---- operators ---------------------------------------------------------------
// Common interface of operators a HipSmoother can drive with one fused kernel per
// polynomial term.
class HipOperator : public Operator<DVector>
{
public:
  // out = x + alpha (x - x_prev) - beta D^{-1} (A x - b); x_prev may be null when alpha == 0
  virtual void smoother_step(DVector const &b, DVector const &x, DVector const *x_prev, double alpha, double beta,
                             DVector &out) const = 0;
  // the first n_terms (2 or 3) polynomial terms in ONE sweep: out = x_{n_terms}, out_prev (may be null) = x_{n_terms - 1};
  // alpha[0] = 0.  False (nothing done) where the operator has no such kernel: the smoother then goes term by term.
  virtual bool smoother_sweep(int /*n_terms*/, double const * /*alpha*/, double const * /*beta*/, DVector const & /*b*/, DVector const & /*x*/,
                              DVector & /*out*/, DVector * /*out_prev*/) const
  {
    return false;
  }
  virtual bool sweep_available(int /*n_terms*/) const { return false; }
  // the same sweep from x_0 = 0, which is not read (hierarchy.hpp:253-259: a preconditioner application starts from zero)
  virtual bool smoother_sweep_from_zero(int /*n_terms*/, double const * /*alpha*/, double const * /*beta*/, DVector const & /*b*/,
                                        DVector & /*out*/) const
  {
    return false;
  }
  virtual double const *get_diagonal_inverse() const = 0; // device pointer
  virtual HipHandle &get_hip_handle() const = 0;
  // distributed vector space of the domain / range (0: rank-local, 1: fine DoFs, 2: first coarse level)
  virtual int domain_space() const { return 0; }
  virtual int range_space() const { return 0; }
  // y = A x on the rows this rank computes, from the local x as it stands: no exchange, no collective (setup probes)
  virtual void apply_local(DVector const &x, DVector &y) const { apply(x, y); }
};

// dot product over the owned entries of a distributed space, summed over the ranks
double distributed_dot(HipHandle &handle, int space, DVector const &x, DVector const &y);

// Twin of CudaMatrixOperator (include/mfmg/cuda/cuda_matrix_operator.cuh, source/cuda/cuda_matrix_operator.cu)
class HipMatrixOperator : public HipOperator
{
public:
  explicit HipMatrixOperator(std::shared_ptr<SparseMatrixDevice<double>> sparse_matrix);

  void apply(DVector const &x, DVector &y, OperatorMode mode = OperatorMode::NO_TRANS) const override;
  std::shared_ptr<Operator<DVector>> transpose() const override;
  std::shared_ptr<Operator<DVector>> multiply(std::shared_ptr<Operator<DVector> const> b) const override;
  std::shared_ptr<Operator<DVector>> multiply_transpose(std::shared_ptr<Operator<DVector> const> b) const override;
  std::shared_ptr<DVector> build_domain_vector() const override;
  std::shared_ptr<DVector> build_range_vector() const override;
  size_t grid_complexity() const override;
  size_t operator_complexity() const override;
  void residual(DVector const &x, DVector const &b, DVector &res) const override;
  void apply_subtract(DVector const &x, DVector &y, OperatorMode mode) const override;
  void apply_local(DVector const &x, DVector &y) const override { _matrix->vmult(y.get_values(), x.get_values()); }

  void smoother_step(DVector const &b, DVector const &x, DVector const *x_prev, double alpha, double beta,
                     DVector &out) const override;
  double const *get_diagonal_inverse() const override;
  HipHandle &get_hip_handle() const override { return _matrix->handle(); }

  std::shared_ptr<SparseMatrixDevice<double>> get_matrix() const { return _matrix; }
  std::shared_ptr<SparseMatrixDevice<double>> get_transposed_matrix() const;
  int domain_space() const override { return _domain_space; }
  int range_space() const override { return _range_space; }
  void set_spaces(int domain, int range)
  {
    _domain_space = domain;
    _range_space = range;
  }
  // after y = M x (NO_TRANS): return the partial sums in the ghost layers of y to their owners (a transposed
  // prolongator of a distributed level: its rows of ghost aggregates hold contributions of owned fine rows)
  void set_reverse_range_space(int space) { _reverse_range_space = space; }
  int reverse_range_space() const { return _reverse_range_space; }
  // agglomerate-wise evaluation of a restrictor and its transpose (structured_restrictor.hpp); the CSR copy
  // stays for the setup algebra and for get_restrictor
  void set_structured(std::shared_ptr<StructuredRestrictorDevice> s) { _structured = std::move(s); }
  bool has_structured() const { return _structured != nullptr; }
  // y = R x on this rank's arrays, no exchange: through the agglomerate-wise form where there is one (the CSR arrays of such a
  // restrictor stay on the host until somebody asks for them)
  void apply_local(double const *x, double *y) const
  {
    if (_structured)
      _structured->restrict_to_coarse(x, y);
    else
      _matrix->vmult(y, x);
  }
  bool structured_float_planes() const { return _structured != nullptr && _structured->float_planes(); }
  // b_c = R (A x - b) in one pass (structured_restrictor.hpp): probes the rows of R A for `a`, checks the result against
  // residual + restriction on a random pair of vectors and keeps it only if the two agree to rounding
  bool prepare_residual_restriction(std::shared_ptr<Operator<DVector> const> a);
  bool has_residual_restriction() const { return _rr_operator != nullptr; }
  int residual_restriction_classes() const { return _structured ? _structured->residual_restriction_classes() : 0; }
  bool restrict_residual(Operator<DVector> const &a, DVector const &x, DVector const &b, DVector &b_coarse) const override;
  void prefetch_rhs(DVector const &b) const override;
  void release_rhs() const override;
  // y = A x + beta D^-1 b (x in the domain space: its ghost entries are refreshed first)
  void apply_plus_scaled(DVector const &x, double const *dinv, DVector const &b, double beta, DVector &y) const;
  // the same from the FP32 vectors of the fine level of apply_f32 (one rank; sums and result in FP64)
  bool restrict_residual_f32(Operator<DVector> const &a, float const *x, float const *b, DVector &b_coarse) const;

private:
  int _domain_space = 0, _range_space = 0, _reverse_range_space = 0;
  std::shared_ptr<SparseMatrixDevice<double>> _matrix;
  mutable std::shared_ptr<SparseMatrixDevice<double>> _transposed_matrix; // built lazily (cuda_matrix_operator.cu:93-130)
  std::shared_ptr<StructuredRestrictorDevice> _structured;
  std::shared_ptr<Operator<DVector> const> _rr_operator; // the fine operator the rows of R A were probed for
  int _rr_space = 0; // distributed runs: the fine space with two ghost layers refreshed per side (the 5 layers of R A)
  mutable double const *_prefetched_rhs = nullptr; // the b whose ghost entries are travelling on the exchange stream
  mutable DeviceBuffer<double> _dinv;
};

// Twin of CudaMatrixFreeOperator (source/cuda/cuda_matrix_free_operator.cu)
class HipMatrixFreeOperator : public HipOperator
{
public:
  explicit HipMatrixFreeOperator(std::shared_ptr<HipMatrixFreeMeshEvaluator> matrix_free_mesh_evaluator);

  void vmult(DVector &dst, DVector const &src) const;
  void apply(DVector const &x, DVector &y, OperatorMode mode = OperatorMode::NO_TRANS) const override;
  std::shared_ptr<Operator<DVector>> transpose() const override;
  std::shared_ptr<Operator<DVector>> multiply(std::shared_ptr<Operator<DVector> const> b) const override;
  std::shared_ptr<Operator<DVector>> multiply_transpose(std::shared_ptr<Operator<DVector> const> b) const override;
  std::shared_ptr<DVector> build_domain_vector() const override;
  std::shared_ptr<DVector> build_range_vector() const override;
  size_t grid_complexity() const override;
  size_t operator_complexity() const override;
  void residual(DVector const &x, DVector const &b, DVector &res) const override;

  void smoother_step(DVector const &b, DVector const &x, DVector const *x_prev, double alpha, double beta,
                     DVector &out) const override;
  bool smoother_sweep(int n_terms, double const *alpha, double const *beta, DVector const &b, DVector const &x, DVector &out,
                      DVector *out_prev) const override;
  bool sweep_available(int n_terms) const override;
  bool smoother_sweep_from_zero(int n_terms, double const *alpha, double const *beta, DVector const &b, DVector &out) const override;
  double const *get_diagonal_inverse() const override;
  HipHandle &get_hip_handle() const override { return _mesh_evaluator->get_hip_handle(); }
  std::shared_ptr<HipMatrixFreeMeshEvaluator> get_mesh_evaluator() const { return _mesh_evaluator; }
  void apply_mode(MfMode mode, double const *x, double const *b, double const *x_prev, double alpha, double beta,
                  double *out) const;
  int domain_space() const override { return 1; }
  int range_space() const override { return 1; }
  void apply_local(DVector const &x, DVector &y) const override;

private:
  std::shared_ptr<HipMatrixFreeMeshEvaluator> _mesh_evaluator;
};

// A R^T kept symbolic: R->multiply(ap) turns it into A_c = R A R^T by probing on the device (27 n_eig applications of
// R^T, A and R over colour classes of agglomerates) -- for the matrix-free operator (the reference builds A R^T with n_coarse
// operator applies, include/mfmg/dealii/dealii_utils.hpp:32-81) and, since round 3, for an ASSEMBLED A whose restrictor has
// the agglomerate-wise form (the explicit products A R^T and R (A R^T) of 516 M and 223 M entries took 6 of the 12 s of
// that setup at 257^3 DoFs).
class HipGalerkinHalfProduct : public Operator<DVector>
{
public:
  HipGalerkinHalfProduct(std::shared_ptr<HipMatrixFreeOperator const> a, std::shared_ptr<HipMatrixOperator const> r)
      : _a(a), _a_op(a), _r(r)
  {
  }
  HipGalerkinHalfProduct(std::shared_ptr<HipMatrixOperator const> a, std::shared_ptr<HipMatrixOperator const> r)
      : _a_matrix(a), _a_op(a), _r(r)
  {
  }
  void apply(DVector const &x, DVector &y, OperatorMode mode = OperatorMode::NO_TRANS) const override;
  std::shared_ptr<Operator<DVector>> transpose() const override
  {
    ASSERT_THROW_NOT_IMPLEMENTED();
    return nullptr;
  }
  std::shared_ptr<Operator<DVector>> multiply(std::shared_ptr<Operator<DVector> const>) const override
  {
    ASSERT_THROW_NOT_IMPLEMENTED();
    return nullptr;
  }
  std::shared_ptr<Operator<DVector>> multiply_transpose(std::shared_ptr<Operator<DVector> const>) const override
  {
    ASSERT_THROW_NOT_IMPLEMENTED();
    return nullptr;
  }
  std::shared_ptr<DVector> build_domain_vector() const override { return _r->build_range_vector(); }
  std::shared_ptr<DVector> build_range_vector() const override { return _a_op->build_range_vector(); }
  size_t grid_complexity() const override { return 0; }
  size_t operator_complexity() const override { return 0; }

  std::shared_ptr<HipMatrixFreeOperator const> get_a() const { return _a; }               // null for an assembled A
  std::shared_ptr<HipMatrixOperator const> get_a_matrix() const { return _a_matrix; }     // null for a matrix-free A
  std::shared_ptr<HipMatrixOperator const> get_r() const { return _r; }

private:
  std::shared_ptr<HipMatrixFreeOperator const> _a;
  std::shared_ptr<HipMatrixOperator const> _a_matrix;
  std::shared_ptr<HipOperator const> _a_op; // whichever of the two
  std::shared_ptr<HipMatrixOperator const> _r;
};

// ---- smoother ------------------------------------------------------------------
// Twin of CudaSmoother (source/cuda/cuda_smoother.cu:99-173, Jacobi) and of
// DealIIMatrixFreeSmoother (source/dealii/dealii_matrix_free_smoother.cc:19-76, Chebyshev).
class HipSmoother : public Smoother<DVector>
{
public:
  HipSmoother(std::shared_ptr<Operator<DVector> const> op, std::shared_ptr<ptree const> params);

  void apply(DVector const &b, DVector &x) const override;
  // same update when x is known to be zero on entry (content of x ignored); saves one operator application
  void apply_zero_guess(DVector const &b, DVector &x) const;
  // same update from x_in into a different vector x_out: the whole polynomial in one sweep where the operator offers it
  // (smoother.fused_terms, default 3), no copy back from a scratch vector
  void apply_to(DVector const &b, DVector const &x_in, DVector &x_out) const override;
  bool apply_from_zero(DVector const &b, DVector &x_out) const override;
  bool prefers_out_of_place() const override;
  int fused_terms() const { return _fused_terms; }
  // polynomial terms the fine smoother runs as ONE sweep: by apply() (in place: the last term is a launch of its own) and by
  // apply_to() (0: one launch per term)
  void sweep_terms(int &in_place, int &out_of_place) const;

  int degree() const { return (int)_coefficients.size(); }
  // (alpha_k, beta_k) of the polynomial terms: x_{k+1} = x_k + alpha_k (x_k - x_{k-1}) - beta_k D^-1 (A x_k - b)
  std::vector<std::pair<double, double>> const &coefficients() const { return _coefficients; }
  double lambda_min() const { return _lambda_min; }
  double lambda_max() const { return _lambda_max; }
  std::string const &type() const { return _type; }

private:
  void estimate_eigenvalues(int n_iterations, double residual, double &min_est, double &max_est) const;
  void run_terms(int k0, DVector const &b, DVector const *cur, DVector const *prev, DVector &x_out) const;
  bool run_sweep(int K, DVector const &b, DVector const &x_in, DVector &out, DVector *out_prev) const;

  std::shared_ptr<HipOperator const> _hip_operator;
  std::string _type;
  std::string _eig_start = "hashed";
  double _lambda_min = 1., _lambda_max = 1.;
  std::vector<std::pair<double, double>> _coefficients; // (alpha_k, beta_k)
  int _fused_terms = 3; // polynomial terms per sweep (1: one launch per term)
  mutable std::shared_ptr<DVector> _scratch_a, _scratch_b;
};

// ---- coarse solver -------------------------------------------------------------
// Twin of CudaSolver (source/cuda/cuda_solver.cu:204-515).
class HipSolver : public Solver<DVector>
{
public:
  // `near_null` (optional): near-null-space vector of the operator for solver.type amg
  // (`grid`: agglomerate grid of the rows, see AmgGridHint; in a distributed run the hierarchy below
  // the first coarse level is built on, and stays local to, the owned block of the operator)
  HipSolver(HipHandle &handle, std::shared_ptr<Operator<DVector> const> op, std::shared_ptr<ptree const> params,
            std::vector<double> const *near_null = nullptr, AmgGridHint const *grid = nullptr);

  void apply(DVector const &b, DVector &x) const override;
  bool ignores_initial_guess() const override { return true; } // every variant starts from zero by itself
  std::string const &type() const { return _solver; }
  int n_iterations() const { return _n_iterations; }

  // multilevel coarse solver (solver.type amg): one V-cycle per apply over an aggregation hierarchy
  struct AmgLevel
  {
    std::shared_ptr<HipMatrixOperator> a;
    std::shared_ptr<HipMatrixOperator> restrictor; // P^T as a matrix; its (lazy) transpose is P
    std::shared_ptr<HipMatrixOperator> prolongator;
    std::shared_ptr<HipSmoother> smoother;
    // V(0,1) levels with a damped-Jacobi post-smoother: x = P x_c, then x' = x - beta D^-1 (A x - b), is ONE operator applied
    // to x_c, x' = P~ x_c + beta D^-1 b with P~ = (I - beta D^-1 A) P formed at setup (amg_device_setup.hip): one launch and
    // -- in a distributed run -- one exchange instead of two of each, and A is not applied in the cycle at all
    std::shared_ptr<HipMatrixOperator> smoothed_prolongator;
    double smoothed_beta = 0.;
    mutable std::shared_ptr<DVector> res, b_coarse, x_coarse, x_work;
  };
  std::vector<AmgLevel> const &amg_levels() const { return _amg; }
  // "release setup matrices": the CSR arrays of the table-driven operators of the aggregation hierarchy (SparseMatrixDevice::
  // release_csr); returns how many matrices let go of theirs
  int release_setup_matrices()
  {
    int n = 0;
    for (auto &L : _amg)
      for (auto const &op : {L.a, L.restrictor, L.prolongator, L.smoothed_prolongator})
        if (op && op->get_matrix() && op->get_matrix()->release_csr())
          ++n;
    return n;
  }
  // index of the first level that is gathered and solved redundantly on every rank (-1: one rank / none)
  int amg_gather_level() const { return _amg_gather_level; }

private:
  void amg_cycle(size_t level, DVector const &b, DVector &x) const;
  // b, x in the local layout of the gathered level's space: all-gather of the owned parts, replicated cycle from
  // `level` down, the local run of layers copied back (ghost layers included)
  void amg_cycle_gathered(size_t level, DVector const &b_local, DVector &x_local) const;
  // aggregation hierarchy by probing on the device, coupled across the ranks (amg_device_setup.hip)
  void setup_amg_on_device(std::shared_ptr<SparseMatrixDevice<double>> matrix, std::vector<double> const &near_null,
                           AmgGridHint const &grid, AmgOptions const &opts, std::shared_ptr<ptree> smoother_params);
  // (`geom`: the level as a halo space -- local box, owned box, global box; on one rank all three coincide)
  void finish_amg_replicated(std::shared_ptr<HipMatrixOperator> a_op, HostCsr A, std::vector<double> B, int space,
                             HaloSpace const &geom, AmgOptions const &opts, std::shared_ptr<ptree> smoother_params);
  // Dense LU with partial pivoting, factored ONCE at setup (the reference re-factorises in every apply,
  // source/cuda/dealii_operator_device_helpers.cu:169-228).  Up to kTriangularInverseLimit rows the factors are
  // stored inverted as two dense triangular matrices and the solve is two SpMV launches over the whole chip
  // (x = U^-1 (L^-1 P) b); above, one workgroup sweeps the packed factors column by column.
  struct DenseLu
  {
    int64_t n = 0;
    std::shared_ptr<SparseMatrixDevice<double>> l_inv_p, u_inv; // n <= kTriangularInverseLimit
    mutable DeviceBuffer<double> tmp;
    DeviceBuffer<double> lu;                                       // column-major L\\U otherwise
    DeviceBuffer<int32_t> perm;
  };
  void setup_direct(std::shared_ptr<SparseMatrixDevice<double>> matrix, DenseLu &f) const;
  void solve_direct(DenseLu const &f, double const *b, double *x) const;
  std::vector<AmgLevel> _amg;
  int _amg_gather_level = -1;
  int _gather_space = 0;
  mutable DeviceBuffer<double> _gather_in;
  std::shared_ptr<DVector> _gather_b, _gather_x;
  // boxes: the gathered blocks arrive in rank order; [global entry] -> position among them, [local entry] -> global entry
  mutable DeviceBuffer<double> _gather_ranked;
  DeviceBuffer<int32_t> _gather_from_ranked, _gather_local_ids;
  int _amg_cycles = 1;
  int _amg_pre_smoothing_levels = 1 << 20;
  DenseLu _amg_bottom;

  HipHandle &_handle;
  std::string _solver;
  std::shared_ptr<HipMatrixOperator const> _matrix_operator;
  DenseLu _dense; // solver.type lu_dense | cholesky | lu_sparse_host
  // pcg
  int _n_iterations = 0;
  mutable DeviceBuffer<double> _scal;
  mutable std::shared_ptr<DVector> _r, _z, _p, _ap;
  DeviceBuffer<double> _dinv;
};

// ---- helpers -------------------------------------------------------------------
// Twin of CudaHierarchyHelpers (source/cuda/cuda_hierarchy_helpers.cu:25-105).
template <typename VectorType>
class HipHierarchyHelpers : public HierarchyHelpers<VectorType>
{
public:
  explicit HipHierarchyHelpers(HipHandle &handle) : _handle(handle) {}
  ~HipHierarchyHelpers() override
  {
    // the halo spaces of this hierarchy's levels go with it (common.hpp: HaloCommunicator::spaces_owner)
    if (_handle.comm.spaces_owner == this)
    {
      _handle.comm.spaces_owner = nullptr;
      _handle.comm.spaces.resize(3);
      _handle.comm.spaces[2] = HaloSpace();
    }
  }

  std::shared_ptr<Operator<VectorType>> get_global_operator(std::shared_ptr<MeshEvaluator> mesh_evaluator) override;

  std::shared_ptr<Operator<VectorType>> build_restrictor(Communicator comm,
                                                         std::shared_ptr<MeshEvaluator> mesh_evaluator,
                                                         std::shared_ptr<ptree const> params) override;

  // `fast_ap = true` (the reference's driver forces it, tests/hierarchy_driver.cc:270): A R^T as build_restrictor
  // prepared it (include/mfmg/common/hierarchy.hpp:214-221, source/dealii/dealii_matrix_free_hierarchy_helpers.cc:326-329).
  // Matrix-free operator: the symbolic half product whose `R->multiply` forms R A R^T from 27 n_eig applications of
  // R^T, A and R over colour classes of agglomerates -- the idea of the reference's fast path (agglomerate-local
  // applications plus the correction of the agglomerate borders, :77-288) carried out with global operator
  // applications, which need no correction.  Assembled operator: the device product A R^T.
  std::shared_ptr<Operator<VectorType>> fast_multiply_transpose() override;

  std::shared_ptr<Smoother<VectorType>> build_smoother(std::shared_ptr<Operator<VectorType> const> op,
                                                       std::shared_ptr<ptree const> params) override;

  std::shared_ptr<Solver<VectorType>> build_coarse_solver(std::shared_ptr<Operator<VectorType> const> op,
                                                          std::shared_ptr<ptree const> params) override;

  void set_coarse_space_hint(std::shared_ptr<Operator<VectorType> const> restrictor) override
  {
    _restrictor_hint = restrictor;
  }

  // restrictor.fused_residual (default true): b_c = R (A x - b) in one pass where R A repeats itself
  void prepare_residual_restriction(std::shared_ptr<Operator<VectorType> const> a, std::shared_ptr<Operator<VectorType>> restrictor,
                                    std::shared_ptr<ptree const> params) override;

private:
  HipHandle &_handle;
  std::shared_ptr<Operator<VectorType>> _operator;
  std::shared_ptr<Operator<VectorType> const> _restrictor_hint;
  std::shared_ptr<Operator<VectorType> const> _own_restrictor; // the one build_restrictor made (grid known)
  std::shared_ptr<Operator<VectorType>> _ap_operator;            // A R^T, prepared by build_restrictor when fast_ap = true
  std::weak_ptr<Operator<VectorType>> _ap_weak;                  // ... after it was handed over
  bool _fast_ap_prepared = false;
  AmgGridHint _grid_hint;
};

// x = U^{-1} L^{-1} P b with the packed column-major LU of dense_lu_factor (getrs)
void dense_lu_solve(HipHandle &handle, int n, double const *lu, int32_t const *perm, double const *b, double *x);
} // namespace mfmg
