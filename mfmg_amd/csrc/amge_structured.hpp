// Host-side (setup, off the hot path) restatement of the spectral-AMGe restrictor
// and the Galerkin coarse operator on a logically structured Q1 mesh.
//
// Reference: block agglomerates include/mfmg/common/amge.templates.hpp:412-499;
// local eigenproblems include/mfmg/dealii/amge_host.templates.hpp:278-483 and
// include/mfmg/cuda/amge_device.templates.cuh:217-310; weights
// include/mfmg/common/amge.templates.hpp:271-325; A_c = R (A R^T)
// include/mfmg/common/hierarchy.hpp:214-233.
// SURVEY.md 8f ranks this setup "next"; it runs on the host cores with OpenMP
// so that the apply path can be exercised end to end.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "common.hpp"

namespace mfmg
{
struct HostCsr
{
  int64_t n_rows = 0, n_cols = 0;
  std::vector<int32_t> row_ptr, col;
  std::vector<double> val;
  int64_t nnz() const { return row_ptr.empty() ? 0 : row_ptr.back(); }
};

// Host copy of what the driver hands over (mfmg_hip_mesh_desc), plus the node map.
struct StructuredMesh
{
  int dim = 3;
  int n[3] = {1, 1, 1}; // cells
  int N[3] = {2, 2, 1}; // nodes
  double h[3] = {1, 1, 1};
  int64_t n_dofs = 0, n_cells = 0;
  std::vector<int32_t> cell_dofs;    // [n_cells][2^dim]
  std::vector<double> coefficient;   // [n_cells][2^dim]
  std::vector<uint8_t> constrained;  // [n_dofs]
  std::vector<int32_t> node_dof;     // DoF id of node (i,j,k), lexicographic nodes

  int nc() const { return 1 << dim; }
  int64_t node_index(int i, int j, int k) const { return i + (int64_t)N[0] * (j + (int64_t)N[1] * k); }
  int64_t cell_index(int i, int j, int k) const { return i + (int64_t)n[0] * (j + (int64_t)n[1] * k); }

  static StructuredMesh from_desc(mfmg_hip_mesh_desc const &desc, hipStream_t stream);
  void build_node_map(); // validates the logical structure, throws std::runtime_error otherwise
};

// How Dirichlet rows appear in the operator
enum class ConstraintSemantics
{
  assembled,  // AffineConstraints::distribute_local_to_global: off-diagonals dropped, summed local
              // diagonal kept (tests/laplace.hpp:198-199)
  matrix_free // MatrixFreeOperators::Base::vmult: identity rows (tests/laplace_matrix_free.hpp:121-156)
};

// K[q][i][j] = sum_d JxW/h_d^2 dphi_i/dxi_d dphi_j/dxi_d at Gauss point q  (A_e = sum_q c_q K[q])
std::vector<double> reference_cell_tables(int dim, double const h[3]);

// one operator row (DoF ids, unsorted) of the Q1 Laplace operator at node (i,j,k)
void operator_row(StructuredMesh const &mesh, std::vector<double> const &Kq, ConstraintSemantics sem, int i,
                  int j, int k, std::vector<int32_t> &cols, std::vector<double> &vals);

// assembled CSR in the caller's DoF numbering, columns sorted (tests/laplace.hpp:154-204)
HostCsr assemble_global_matrix(StructuredMesh const &mesh, ConstraintSemantics sem);
std::vector<double> operator_diagonal(StructuredMesh const &mesh, ConstraintSemantics sem);

struct RestrictorOptions
{
  int agglomerate[3] = {2, 2, 2};
  int n_eigenvectors = 1;
  // 'device': unshifted dense eigenproblem, B = I (amge_device.templates.cuh:256-310)
  // 'host'  : shifted by the mean diagonal, constrained diagonals := 200 (amge_host.templates.hpp:378-394)
  // 'mf'    : matrix-free agglomerate operator [A_ff 0; 0 I] (amge_host.templates.hpp:278-350)
  std::string variant = "device";
  // 'lapack': first n columns of the dense solver; 'krylov': one vector per distinct eigenvalue, the
  // projection of the start vector of DealIIMeshEvaluator::set_initial_guess
  // (source/dealii/dealii_mesh_evaluator.cc:44-56) onto the eigenspace -- what ARPACK / Lanczos return.
  std::string selection = "lapack";
  bool use_coefficient = true; // tests/test_hierarchy_device.cu:239-244 ignores it on agglomerates
};

// R as CSR (rows = coarse DoFs: agglomerates x-fastest, eigenvectors inside)
// `row_agglomerate` (optional): agglomerate index (x fastest) of every row; `agglomerate_counts`: grid
// `device` (optional): solve the agglomerate eigenproblems on the GPU of that handle (amge_device.hip) instead of
// on the host cores; same rules, results equal to rounding
HostCsr build_restrictor_structured(StructuredMesh const &mesh, std::vector<double> const &global_diag,
                                    RestrictorOptions const &opts, std::vector<int32_t> *row_agglomerate = nullptr,
                                    int *agglomerate_counts = nullptr, HipHandle *device = nullptr);

// A_c = R A R^T without storing A or A R^T: operator rows generated on the fly from the
// coefficient table.  `Rt` must be the transpose of `R`.
HostCsr galerkin_triple_product(StructuredMesh const &mesh, ConstraintSemantics sem, HostCsr const &R,
                                HostCsr const &Rt);

// symmetric dense eigen-decomposition (cyclic Jacobi), ascending eigenvalues, V column-major n x n
void symmetric_eigen(int n, std::vector<double> &A, std::vector<double> &w, std::vector<double> &V);

// dense LU with partial pivoting (getrf): on exit `A` (row-major in) holds L\\U in COLUMN-major
// order, `perm[i]` = source row of row i of P A; throws if a pivot is exactly zero
void dense_lu_factor(int n, std::vector<double> &A, std::vector<int32_t> &perm);
// in place: the column-major packed L\\U of dense_lu_factor -> L^{-1} below the diagonal (unit diagonal implied),
// U^{-1} on and above it; used for small systems, where two dense products beat the substitution sweeps
void dense_triangular_inverses(int n, std::vector<double> &lu);
constexpr int kTriangularInverseLimit = 2048;

// Host threads the setup may use: min(affinity mask, cgroup cpu.max quota); applied to OpenMP once
// (an over-subscribed quota-limited container is what makes a 256-thread default pathological).
int effective_cpu_count();
void configure_host_threads();

// libstdc++ std::default_random_engine + uniform_real_distribution<double>(0,1)
struct MinstdUniform
{
  uint64_t state = 1;
  double next();
};
} // namespace mfmg
