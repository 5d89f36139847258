// Host-side setup of the algebraic multilevel coarse solver ("solver.type amg").
//
// The reference hands the coarse problem to third parties: ML smoothed aggregation
// (source/dealii/dealii_solver.cc:48-66, ML_Epetra::SetDefaults("SA")) or AMGx
// (source/cuda/cuda_solver.cu:204-445).  An AmgX shim is ruled out and Trilinos is absent, so
// the same idea -- smoothed aggregation driven by a near-null-space vector -- is restated
// here (Vanek/Mandel/Brezina aggregation, damped-Jacobi prolongator smoothing, Galerkin
// products).  Setup runs on the host cores; the apply runs on the GPU with the CSR kernels.
#pragma once

#include "amge_structured.hpp"

namespace mfmg
{
struct AmgOptions
{
  int max_levels = 10;
  int64_t coarsest_size = 1100; // dense solve below this many rows (two SpMV launches over inverted triangular factors)
  double strength = 0.08;       // |a_ij| > strength * sqrt(a_ii a_jj) is a strong connection
  bool smooth_prolongator = true;
  double omega = 4. / 3.;
  // geometric aggregates: from level `deep_level` on (0 = the operator handed in) blocks of `deep_block` nodes per
  // direction instead of the hint's block (the small levels are bound by launch latency, not by their size)
  int deep_level = 1 << 30;
  int deep_block = 2;
};

// Optional geometric information: row i of the operator lives on node `node_of_row[i]` of a structured
// grid `dims` (x fastest); aggregates are then blocks of `block` nodes (all rows of a node together)
// instead of the greedy strength-of-connection aggregates.  For the coarse level of the AMGe hierarchy
// the nodes are the agglomerates.
// Rows of one node are told apart by `component_of_row` (for the AMGe coarse level: which eigenvector
// of the agglomerate); components are never mixed in one aggregate, so the coarse levels keep
// `n_components` unknowns per node -- piecewise constants per component, then smoothed.
struct AmgGridHint
{
  int dims[3] = {0, 0, 0};
  int block[3] = {2, 2, 2};
  int n_components = 1;
  std::vector<int32_t> node_of_row;
  std::vector<int32_t> component_of_row; // empty: all rows are component 0
  bool valid(int64_t n_rows) const { return dims[0] > 0 && (int64_t)node_of_row.size() == n_rows; }
};

struct AmgLevelHost
{
  HostCsr A; // operator of this level
  HostCsr P; // prolongator from the next (coarser) level; empty on the last level
  std::vector<double> near_null;
};

// `A0` and its near-null-space vector `b0` (e.g. the coarse representation of the constant).
std::vector<AmgLevelHost> build_aggregation_hierarchy(HostCsr A0, std::vector<double> b0, AmgOptions const &opts,
                                                       AmgGridHint const *grid = nullptr);

// greedy aggregation on the strength graph; returns aggregate id per row (-1 never) and the count
int64_t aggregate_rows(HostCsr const &A, double strength, std::vector<int32_t> &aggregate_of);
} // namespace mfmg
