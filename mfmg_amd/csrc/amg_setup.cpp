#include "amg_setup.hpp"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "sparse_matrix_device.hpp"

namespace mfmg
{
namespace
{
double wall_seconds()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
} // namespace

int64_t aggregate_rows(HostCsr const &A, double strength, std::vector<int32_t> &agg)
{
  const int64_t n = A.n_rows;
  std::vector<double> diag(n, 0.);
  for (int64_t i = 0; i < n; ++i)
    for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1]; ++p)
      if (A.col[p] == i)
        diag[i] = std::abs(A.val[p]);
  auto strong = [&](int64_t i, int p) {
    const int j = A.col[p];
    return j != i && std::abs(A.val[p]) > strength * std::sqrt(diag[i] * diag[j]);
  };
  agg.assign(n, -1);
  int64_t n_agg = 0;
  // pass 1: a root whose strong neighbourhood is entirely free starts an aggregate
  for (int64_t i = 0; i < n; ++i)
  {
    if (agg[i] >= 0)
      continue;
    bool free_nbh = true, has_strong = false;
    for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1] && free_nbh; ++p)
      if (strong(i, p))
      {
        has_strong = true;
        if (agg[A.col[p]] >= 0)
          free_nbh = false;
      }
    if (!free_nbh || !has_strong)
      continue;
    agg[i] = (int32_t)n_agg;
    for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1]; ++p)
      if (strong(i, p))
        agg[A.col[p]] = (int32_t)n_agg;
    ++n_agg;
  }
  // pass 2: leftovers join the aggregate they are most strongly connected to
  std::vector<int32_t> joined(agg);
  for (int64_t i = 0; i < n; ++i)
  {
    if (agg[i] >= 0)
      continue;
    double best = 0.;
    int32_t best_agg = -1;
    for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1]; ++p)
      if (strong(i, p) && agg[A.col[p]] >= 0 && std::abs(A.val[p]) > best)
      {
        best = std::abs(A.val[p]);
        best_agg = agg[A.col[p]];
      }
    joined[i] = best_agg;
  }
  agg.swap(joined);
  // pass 3: what is still free (isolated rows, e.g. constrained DoFs) forms aggregates with its free
  // strong neighbours, or stays alone
  for (int64_t i = 0; i < n; ++i)
  {
    if (agg[i] >= 0)
      continue;
    agg[i] = (int32_t)n_agg;
    for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1]; ++p)
      if (strong(i, p) && agg[A.col[p]] < 0)
        agg[A.col[p]] = (int32_t)n_agg;
    ++n_agg;
  }
  return n_agg;
}

namespace
{
HostCsr multiply(HostCsr const &a, HostCsr const &b)
{
  HostCsr c;
  c.n_rows = a.n_rows;
  c.n_cols = b.n_cols;
  csr_multiply_host<double>(a.n_rows, a.n_cols, a.row_ptr, a.col, a.val, b.n_cols, b.row_ptr, b.col, b.val, c.row_ptr,
                            c.col, c.val);
  return c;
}
HostCsr transpose(HostCsr const &a)
{
  HostCsr t;
  t.n_rows = a.n_cols;
  t.n_cols = a.n_rows;
  csr_transpose_host<double>(a.n_rows, a.n_cols, a.row_ptr, a.col, a.val, t.row_ptr, t.col, t.val);
  return t;
}
} // namespace

std::vector<AmgLevelHost> build_aggregation_hierarchy(HostCsr A0, std::vector<double> b0, AmgOptions const &opts,
                                                       AmgGridHint const *grid_in)
{
  configure_host_threads();
  AmgGridHint grid;
  if (grid_in)
    grid = *grid_in;
  std::vector<AmgLevelHost> levels;
  levels.emplace_back();
  levels.back().A = std::move(A0);
  levels.back().near_null = std::move(b0);
  while ((int)levels.size() < opts.max_levels && levels.back().A.n_rows > opts.coarsest_size)
  {
    HostCsr const &A = levels.back().A;
    std::vector<double> const &B = levels.back().near_null;
    const int64_t n = A.n_rows;
    const bool verbose = std::getenv("MFMG_HIP_VERBOSE") != nullptr;
    const double t_level = wall_seconds();
    ASSERT_THROW((int64_t)B.size() == n, "near-null-space vector has the wrong size");
    std::vector<int32_t> agg;
    int64_t n_agg = 0;
    int cdims[3] = {0, 0, 0};
    const bool geometric = grid.valid(n);
    if (geometric)
    {
      for (int d = 0; d < 3; ++d)
        cdims[d] = (std::max(grid.dims[d], 1) + grid.block[d] - 1) / grid.block[d];
      const int ncomp = std::max(grid.n_components, 1);
      n_agg = (int64_t)cdims[0] * cdims[1] * cdims[2] * ncomp;
      agg.resize(n);
      for (int64_t i = 0; i < n; ++i)
      {
        const int32_t nd = grid.node_of_row[i];
        const int ni = nd % grid.dims[0], nj = (nd / grid.dims[0]) % std::max(grid.dims[1], 1),
                  nk = nd / (grid.dims[0] * std::max(grid.dims[1], 1));
        const int64_t blk = (ni / grid.block[0]) + (int64_t)cdims[0] * ((nj / grid.block[1]) + (int64_t)cdims[1] * (nk / grid.block[2]));
        const int comp = grid.component_of_row.empty() ? 0 : grid.component_of_row[i];
        agg[i] = (int32_t)(blk * ncomp + comp);
      }
    }
    else
      n_agg = aggregate_rows(A, opts.strength, agg);
    // tentative prolongator: one column per aggregate, the normalised restriction of B
    std::vector<double> norm2(n_agg, 0.);
    for (int64_t i = 0; i < n; ++i)
      norm2[agg[i]] += B[i] * B[i];
    // aggregates on which B vanishes carry no coarse function: renumber the others
    std::vector<int32_t> new_id(n_agg, -1);
    int64_t n_coarse = 0;
    for (int64_t a = 0; a < n_agg; ++a)
      if (norm2[a] > 0.)
        new_id[a] = (int32_t)n_coarse++;
    if (n_coarse == 0 || n_coarse * 3 > n * 2) // no useful coarsening
      break;
    HostCsr Pt; // tentative
    Pt.n_rows = n;
    Pt.n_cols = n_coarse;
    Pt.row_ptr.assign(n + 1, 0);
    for (int64_t i = 0; i < n; ++i)
      Pt.row_ptr[i + 1] = Pt.row_ptr[i] + ((new_id[agg[i]] >= 0 && B[i] != 0.) ? 1 : 0);
    Pt.col.resize(Pt.row_ptr[n]);
    Pt.val.resize(Pt.row_ptr[n]);
    std::vector<double> Bc(n_coarse);
    for (int64_t a = 0; a < n_agg; ++a)
      if (new_id[a] >= 0)
        Bc[new_id[a]] = std::sqrt(norm2[a]);
    for (int64_t i = 0; i < n; ++i)
      if (Pt.row_ptr[i + 1] > Pt.row_ptr[i])
      {
        Pt.col[Pt.row_ptr[i]] = new_id[agg[i]];
        Pt.val[Pt.row_ptr[i]] = B[i] / std::sqrt(norm2[agg[i]]);
      }
    HostCsr P;
    if (opts.smooth_prolongator)
    {
      // P = (I - omega/rho D^-1 A) P_tent with rho = max_i sum_j |a_ij| / a_ii >= lambda_max(D^-1 A)
      std::vector<double> dinv(n, 0.);
      double rho = 0.;
      for (int64_t i = 0; i < n; ++i)
      {
        double d = 0., s = 0.;
        for (int p = A.row_ptr[i]; p < A.row_ptr[i + 1]; ++p)
        {
          s += std::abs(A.val[p]);
          if (A.col[p] == i)
            d = A.val[p];
        }
        ASSERT_THROW(d != 0., "zero diagonal in the multilevel coarse solver setup");
        dinv[i] = 1. / d;
        rho = std::max(rho, s / std::abs(d));
      }
      HostCsr S = A; // S = I - omega/rho D^-1 A
      const double w = opts.omega / rho;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i)
        for (int p = S.row_ptr[i]; p < S.row_ptr[i + 1]; ++p)
          S.val[p] = ((S.col[p] == i) ? 1. : 0.) - w * dinv[i] * S.val[p];
      const double t0 = wall_seconds();
      P = multiply(S, Pt);
      if (verbose)
        std::fprintf(stderr, "[mfmg_hip] amg level %d (%lld rows): smoothing matrix %.2f s, P = S P_tent %.2f s\n",
                     (int)levels.size() - 1, (long long)n, t0 - t_level, wall_seconds() - t0);
    }
    else
      P = Pt;
    const double t1 = wall_seconds();
    HostCsr AP = multiply(A, P);
    const double t2 = wall_seconds();
    HostCsr PT = transpose(P);
    const double t3 = wall_seconds();
    HostCsr Ac = multiply(PT, AP);
    if (verbose)
      std::fprintf(stderr, "[mfmg_hip] amg level %d: A P %.2f s, transpose %.2f s, P^T (A P) %.2f s\n", (int)levels.size() - 1,
                   t2 - t1, t3 - t2, wall_seconds() - t3);
    levels.back().P = std::move(P);
    levels.emplace_back();
    levels.back().A = std::move(Ac);
    levels.back().near_null = std::move(Bc);
    if (geometric)
    {
      // the next level has one row per kept aggregate, living on the coarsened grid
      AmgGridHint next;
      for (int d = 0; d < 3; ++d)
      {
        next.dims[d] = cdims[d];
        next.block[d] = ((int)levels.size() >= opts.deep_level) ? opts.deep_block : grid.block[d];
      }
      const int ncomp = std::max(grid.n_components, 1);
      next.n_components = ncomp;
      next.node_of_row.resize(n_coarse);
      next.component_of_row.resize(n_coarse);
      for (int64_t a = 0; a < n_agg; ++a)
        if (new_id[a] >= 0)
        {
          next.node_of_row[new_id[a]] = (int32_t)(a / ncomp);
          next.component_of_row[new_id[a]] = (int32_t)(a % ncomp);
        }
      grid = std::move(next);
    }
    else
      grid = AmgGridHint();
  }
  return levels;
}
} // namespace mfmg
