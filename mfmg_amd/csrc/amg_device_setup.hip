// Setup of the smoothed-aggregation hierarchy below the first coarse level ON THE DEVICE, by probing, for runs
// where the levels are coupled across the ranks of a slab decomposition (and, on request, for one rank).
//
// The reference delegates the coarse problem to ML / AMGx (source/dealii/dealii_solver.cc:48-66,
// source/cuda/cuda_solver.cu:204-445) and forms its Galerkin products with host / cuSPARSE SpGEMM
// (source/cuda/cuda_matrix_operator.cu:132-225, include/mfmg/cuda/sparse_matrix_device.templates.cuh:373-434).  Here the
// aggregates are 2x2x2 blocks of nodes of a structured grid, so the sparsity of every matrix of the hierarchy is a
// box stencil of known reach, and a matrix can be read off from a few applications of operators that already exist
// on the device:
//   P_l     = (I - omega/rho D^-1 A_l) P_tent     columns of aggregates (1 + r_l) apart never overlap:
//                                                  (1 + r_l)^3 n_comp applications of A_l;
//   A_{l+1} = P_l^T A_l P_l                        columns 2 r_{l+1} + 1 apart never meet in a row:
//                                                  (2 r_{l+1} + 1)^3 n_comp applications of P_l, A_l, P_l^T
// with r_l the reach of A_l in nodes (r_0 = 1 for the AMGe coarse operator, r_{l+1} = floor((1 + 3 r_l) / 2); aggregates of
// b nodes per direction: r_{l+1} = floor((b - 1 + 3 r_l) / b), so b = 3 keeps a reach of 1).
// Nothing but vectors crosses between ranks: the probing vectors are defined on global coordinates, A_l reads its
// ghost layers after a forward halo exchange and P_l^T returns the partial sums of ghost aggregates to their owners
// by a reverse (adding) exchange -- exactly the exchanges of the cycle itself.  No sparse rows are communicated.
// Below `solver.amg.replicate_rows` global rows (or where a slab can no longer be halved) the level is gathered and
// the rest of the hierarchy is built and applied redundantly on every rank by the host code path of one rank.
#include "mfmg/hip_hierarchy_helpers.hpp"
#include "probe_assembly.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>

namespace mfmg
{
namespace
{
double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct LevelGeom
{
  int dims[3] = {0, 0, 0}; // local nodes (x, y, z incl. ghost layers)
  int n_comp = 1;
  int space = 0;             // halo space of the level's vectors (0: one rank)
  int64_t owned_begin = 0, owned_count = 0; // owned z layers of the local box
  int64_t global_begin = 0, global_layers = 0;
  int reach = 1;             // stencil reach of the level's operator in nodes
  int64_t layer_elems() const { return (int64_t)dims[0] * dims[1] * n_comp; }
  int64_t n_rows() const { return layer_elems() * dims[2]; }
  int64_t owned_row_begin() const { return owned_begin * layer_elems(); }
  int64_t owned_rows() const { return owned_count * layer_elems(); }
};

std::shared_ptr<SparseMatrixDevice<double>> upload_csr(HipHandle &handle, HostCsr &&m, bool keep_host = true)
{
  return std::make_shared<SparseMatrixDevice<double>>(handle, m.n_rows, m.n_cols, std::move(m.row_ptr), std::move(m.col),
                                                      std::move(m.val), keep_host);
}
} // namespace

// Builds `_amg` for an operator whose rows live on a structured node grid (node-major, component-minor).
void HipSolver::setup_amg_on_device(std::shared_ptr<SparseMatrixDevice<double>> matrix, std::vector<double> const &near_null,
                                    AmgGridHint const &grid, AmgOptions const &opts, std::shared_ptr<ptree> smoother_params)
{
  HipHandle &h = _handle;
  HaloCommunicator &comm = h.comm;
  const int op_space = _matrix_operator->domain_space();
  const bool distributed = comm.enabled() && op_space > 0;
  const bool verbose = std::getenv("MFMG_HIP_VERBOSE") != nullptr;
  // (one rank: the small levels too are read off from operator applications -- 0.7 s less than their host products at
  // 257^3 DoFs; many ranks: below 200000 global rows a level is gathered and the rest replicated)
  const int64_t replicate_rows = this->_params->get("solver.amg.replicate_rows", distributed ? 200000 : 4000);
  const int C = std::max(grid.n_components, 1);
  // aggregates: cubes of `blk` nodes (solver.amg.aggregate_block), aligned globally.  With 2 the box stencil of the
  // operators grows from level to level (reach 1, 2, 3, 5: 53, 236, 582, 1520 entries per row); with 3 -- the classic
  // coarsening of smoothed aggregation -- a reach of 1 stays 1.  General rule: r_{l+1} = floor((blk - 1 + 3 r_l) / blk).
  const int blk = grid.block[0];
  ASSERT_THROW(blk >= 2 && grid.block[1] == blk && grid.block[2] == blk, "the device setup needs cubic aggregates");

  LevelGeom g;
  for (int d = 0; d < 3; ++d)
    g.dims[d] = std::max(grid.dims[d], 1);
  g.n_comp = C;
  g.reach = 1;
  if (distributed)
  {
    HaloSpace const &s = comm.spaces[op_space];
    ASSERT_THROW(s.layer_elems == g.layer_elems() && s.n_layers == g.dims[2], "internal: coarse space does not match the agglomerate grid");
    g.space = op_space;
    g.owned_begin = s.owned_begin;
    g.owned_count = s.owned_count;
    g.global_begin = s.global_begin;
    g.global_layers = s.global_layers;
  }
  else
  {
    g.owned_begin = 0;
    g.owned_count = g.dims[2];
    g.global_begin = 0;
    g.global_layers = g.dims[2];
  }
  ASSERT_THROW(matrix->m() == g.n_rows(), "internal: operator rows do not match the agglomerate grid");
  for (int64_t r = 0; r < (int64_t)grid.node_of_row.size(); r += std::max<int64_t>(1, (int64_t)grid.node_of_row.size() / 8191))
    ASSERT_THROW(grid.node_of_row[r] == r / C && (grid.component_of_row.empty() ? 0 : grid.component_of_row[r]) == r % C,
                 "the device setup of the aggregation hierarchy needs node-major rows with a fixed number of components");

  // host copy of the current level: owned rows of A (local column ids), near-null vector (local, owned part valid)
  // (the level's matrix stays on the device: its diagonal and row sums come from a kernel; only the level that is gathered at
  // the end is downloaded -- the first level alone is 2.7 GB)
  HostCsr A;
  std::vector<double> B = near_null;
  ASSERT_THROW((int64_t)B.size() == g.n_rows(), "near-null-space vector has the wrong size");
  std::shared_ptr<HipMatrixOperator> a_op = std::const_pointer_cast<HipMatrixOperator>(_matrix_operator);

  _amg.clear();
  _amg_gather_level = -1;
  for (;;)
  {
    const int level = (int)_amg.size();
    const int64_t global_rows = g.layer_elems() * g.global_layers;
    // ---- can the next level be built distributed?
    LevelGeom c;
    for (int d = 0; d < 2; ++d)
      c.dims[d] = (g.dims[d] + blk - 1) / blk;
    c.n_comp = C;
    c.reach = (blk - 1 + 3 * g.reach) / blk;
    bool coarsen_here = global_rows > std::max<int64_t>(replicate_rows, opts.coarsest_size) && level + 1 < opts.max_levels;
    if (coarsen_here)
    {
      if (distributed)
      {
        HaloSpace const &s = comm.spaces[g.space];
        // whole aggregates per rank, aligned globally; the ghost layers of the coarse level must come from the
        // immediate neighbour
        double ok = (g.owned_count % blk == 0 && (g.global_begin + g.owned_begin) % blk == 0 && g.owned_count / blk >= c.reach) ? 1. : 0.;
        ok = -h.allreduce_max(-ok); // min over the ranks
        coarsen_here = ok > 0.5;
        c.owned_count = g.owned_count / blk;
        c.owned_begin = s.has_low ? c.reach : 0;
        c.dims[2] = (int)(c.owned_begin + c.owned_count + (s.has_high ? c.reach : 0));
        c.global_begin = (g.global_begin + g.owned_begin) / blk - c.owned_begin;
        c.global_layers = g.global_layers / blk;
        ASSERT_THROW(!coarsen_here || g.global_layers % blk == 0, "internal: layers of a distributed level not a multiple of the aggregate size");
      }
      else
      {
        c.dims[2] = (g.dims[2] + blk - 1) / blk;
        c.owned_begin = 0;
        c.owned_count = c.dims[2];
        c.global_begin = 0;
        c.global_layers = c.dims[2];
      }
    }
    if (!coarsen_here)
    {
      // ---- gather this level; the rest of the hierarchy is replicated (host setup of one rank)
      A.n_rows = A.n_cols = g.n_rows();
      a_op->get_matrix()->download(A.row_ptr, A.col, A.val);
      finish_amg_replicated(a_op, std::move(A), std::move(B), g.space, g.owned_begin, g.owned_count, g.global_begin, g.global_layers,
                            g.dims, C, opts, smoother_params);
      break;
    }
    const double t0 = wall_now();
    if (distributed)
    {
      HaloSpace s;
      s.layer_elems = c.layer_elems();
      s.n_layers = c.dims[2];
      s.owned_begin = c.owned_begin;
      s.owned_count = c.owned_count;
      s.global_begin = c.global_begin;
      s.global_layers = c.global_layers;
      s.width = c.reach;
      s.has_low = comm.spaces[g.space].has_low;
      s.has_high = comm.spaces[g.space].has_high;
      c.space = comm.add_space(s);
      // the fine level of this pair exchanges as many layers as its operator reaches
      comm.spaces[g.space].width = std::max(comm.spaces[g.space].width, g.reach);
      ASSERT_THROW(comm.spaces[g.space].width <= std::min<int64_t>(s.has_low ? comm.spaces[g.space].ghost_low() : 1 << 30,
                                                                    s.has_high ? comm.spaces[g.space].ghost_high() : 1 << 30),
                   "internal: not enough ghost layers for the stencil of this level");
    }
    const int64_t n_f = g.n_rows(), n_c = c.n_rows();
    const int64_t row0 = g.owned_row_begin(), n_own = g.owned_rows();

    // ---- diagonal, rho = max_i sum_j |a_ij| / |a_ii| over the owned rows of all ranks
    DeviceBuffer<double> d_dinv((size_t)n_f);
    double rho = 0.;
    {
      DeviceBuffer<double> d_ratio((size_t)n_f);
      a_op->get_matrix()->row_ratios(d_dinv.data(), d_ratio.data());
      const std::vector<double> ratio = d_ratio.download(h.stream);
#pragma omp parallel for schedule(static) reduction(max : rho)
      for (int64_t i = row0; i < row0 + n_own; ++i)
        rho = std::max(rho, ratio[i]);
    }
    rho = h.allreduce_max(rho); // (first the collective, then the check: every rank throws or none does)
    ASSERT_THROW(rho < HUGE_VAL, "zero diagonal in the multilevel coarse solver setup");
    const double w = opts.omega / rho;

    // ---- tentative prolongator: t = B / |B|_aggregate on the owned nodes, exchanged to the ghosts; B_c = |B|_aggregate
    auto agg_of_row = [&](int64_t row, int &I, int &J, int &K, int &comp) {
      comp = (int)(row % C);
      const int64_t nd = row / C;
      const int i = (int)(nd % g.dims[0]), j = (int)((nd / g.dims[0]) % g.dims[1]), k = (int)(nd / ((int64_t)g.dims[0] * g.dims[1]));
      I = i / blk;
      J = j / blk;
      K = (int)((k + g.global_begin) / blk - c.global_begin); // local coarse layer
    };
    std::vector<double> norm2((size_t)n_c, 0.), t((size_t)n_f, 0.), Bc((size_t)n_c, 0.);
    for (int64_t i = row0; i < row0 + n_own; ++i)
    {
      int I, J, K, comp;
      agg_of_row(i, I, J, K, comp);
      norm2[(((int64_t)K * c.dims[1] + J) * c.dims[0] + I) * C + comp] += B[i] * B[i];
    }
    for (int64_t i = row0; i < row0 + n_own; ++i)
    {
      int I, J, K, comp;
      agg_of_row(i, I, J, K, comp);
      const double nn = norm2[(((int64_t)K * c.dims[1] + J) * c.dims[0] + I) * C + comp];
      ASSERT_THROW(nn > 0., "the device setup of the aggregation hierarchy needs a near-null-space vector without zero aggregates");
      t[i] = B[i] / std::sqrt(nn);
    }
    for (int64_t r = c.owned_row_begin(); r < c.owned_row_begin() + c.owned_rows(); ++r)
      Bc[r] = std::sqrt(norm2[r]);
    DVector t_dev(h, n_f), y_f(h, n_f), z_f(h, n_f), u_c(h, n_c), y_c(h, n_c);
    MFMG_HIP_CHECK(hipMemcpyAsync(t_dev.get_values(), t.data(), (size_t)n_f * sizeof(double), hipMemcpyHostToDevice, h.stream));
    h.exchange(g.space, t_dev.get_values());

    // ---- P = (I - w D^-1 A) P_tent by probing: the columns of aggregates floor((blk - 1 + 2 reach) / blk) + 1 apart
    //      (1 + reach for blk = 2) are disjoint
    const int gdims_c[3] = {c.dims[0], c.dims[1], (int)c.global_layers};
    int period_p[3];
    for (int d = 0; d < 3; ++d)
      period_p[d] = std::max(1, std::min((blk - 1 + 2 * g.reach) / blk + 1, gdims_c[d]));
    const int n_col_p = period_p[0] * period_p[1] * period_p[2] * C;
    // (the probes stay on the device and the rows are assembled there: probe_assembly.hip)
    DeviceBuffer<double> Z((size_t)n_col_p * (size_t)n_own);
    for (int col = 0; col < n_col_p; ++col)
    {
      const int comp = col % C, oc = col / C;
      const int phase[3] = {oc % period_p[0], (oc / period_p[0]) % period_p[1], oc / (period_p[0] * period_p[1])};
      vec::select_rows(h, g.dims, C, blk, (int)g.global_begin, period_p, phase, comp, t_dev.get_values(), y_f.get_values());
      a_op->get_matrix()->vmult(z_f.get_values(), y_f.get_values()); // ghosts of y are set locally: no exchange
      MFMG_HIP_CHECK(hipMemcpyAsync(Z.data() + (size_t)col * n_own, z_f.get_values() + row0, (size_t)n_own * sizeof(double),
                                    hipMemcpyDeviceToDevice, h.stream));
    }
    MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
    const double t_probed = wall_now();
    // candidates of owned row i: aggregates whose nodes lie within `reach` of its node; v = t_i [own aggregate] - w d_i^-1 (A y)_i
    auto p_mat = prolongator_from_probes(h, g.dims, c.dims, gdims_c, C, blk, g.reach, period_p, g.global_begin, c.global_begin, row0, n_own,
                                         w, Z.data(), t_dev.get_values(), d_dinv.data());
    Z.release();
    const double t_assembled = wall_now();
    const double t_uploaded = wall_now();
    auto pt_mat = p_mat->transpose();
    if (verbose)
      std::fprintf(stderr, "[mfmg_hip] amg level %d: P probes %.2f s, assembly %.2f s, upload + layouts %.2f s, transpose %.2f s\n", level,
                   t_probed - t0, t_assembled - t_probed, t_uploaded - t_assembled, wall_now() - t_uploaded);
    const double t1 = wall_now();

    // ---- A_c = P^T A P by probing: coarse nodes (2 reach_c + 1) apart never meet in a row
    int period_a[3];
    for (int d = 0; d < 3; ++d)
      period_a[d] = std::max(1, std::min(2 * c.reach + 1, gdims_c[d]));
    const int n_col_a = period_a[0] * period_a[1] * period_a[2] * C;
    const int64_t crow0 = c.owned_row_begin(), cn_own = c.owned_rows();
    DeviceBuffer<double> Y((size_t)n_col_a * (size_t)cn_own);
    for (int col = 0; col < n_col_a; ++col)
    {
      const int comp = col % C, oc = col / C;
      const int phase[3] = {oc % period_a[0], (oc / period_a[0]) % period_a[1], oc / (period_a[0] * period_a[1])};
      vec::select_rows(h, c.dims, C, 1, (int)c.global_begin, period_a, phase, comp, nullptr, u_c.get_values());
      p_mat->vmult(y_f.get_values(), u_c.get_values());
      h.exchange(g.space, y_f.get_values());
      a_op->get_matrix()->vmult(z_f.get_values(), y_f.get_values());
      pt_mat->vmult(y_c.get_values(), z_f.get_values());
      h.exchange_reverse_add(c.space, y_c.get_values());
      MFMG_HIP_CHECK(hipMemcpyAsync(Y.data() + (size_t)col * cn_own, y_c.get_values() + crow0, (size_t)cn_own * sizeof(double),
                                    hipMemcpyDeviceToDevice, h.stream));
    }
    MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
    auto ac_mat = coarse_operator_from_probes(h, c.dims, gdims_c, C, c.reach, period_a, c.global_begin, crow0, cn_own, Y.data());
    Y.release();
    if (verbose)
      std::fprintf(stderr, "[mfmg_hip] amg level %d on the device (%lld local rows, reach %d): P %d probes %.2f s, A_c %d probes %.2f s\n",
                   level, (long long)n_f, g.reach, n_col_p, t1 - t0, n_col_a, wall_now() - t1);

    // ---- the level's operators
    AmgLevel L;
    L.a = a_op;
    L.prolongator = std::make_shared<HipMatrixOperator>(p_mat);
    L.prolongator->set_spaces(c.space, 0); // x -= P x_c reads the ghost aggregates of x_c
    L.restrictor = std::make_shared<HipMatrixOperator>(pt_mat);
    L.restrictor->set_spaces(0, 0);        // P^T has entries in owned fine rows only ...
    L.restrictor->set_reverse_range_space(c.space); // ... and returns the sums of ghost aggregates to their owners
    L.smoother = std::make_shared<HipSmoother>(L.a, smoother_params);
    _amg.push_back(std::move(L));

    // ---- next level
    a_op = std::make_shared<HipMatrixOperator>(ac_mat);
    a_op->set_spaces(c.space, c.space);
    B = std::move(Bc);
    g = c;
  }
}

// The level `a_op` (owned rows in `A`, local numbering) becomes the first replicated level: its operator and near-null
// vector are gathered, the remaining hierarchy is built by the host code of one rank, identically on every rank.
void HipSolver::finish_amg_replicated(std::shared_ptr<HipMatrixOperator> a_op, HostCsr A, std::vector<double> B, int space,
                                      int64_t owned_begin, int64_t owned_count, int64_t global_begin, int64_t global_layers,
                                      int const dims[3], int n_comp, AmgOptions const &opts, std::shared_ptr<ptree> smoother_params)
{
  HipHandle &h = _handle;
  const bool distributed = h.comm.enabled() && space > 0;
  const int64_t le = (int64_t)dims[0] * dims[1] * n_comp;
  HostCsr Ag;
  std::vector<double> Bg;
  if (!distributed)
  {
    Ag = std::move(A);
    Bg = std::move(B);
  }
  else
  {
    const int n_ranks = h.comm.n_ranks;
    const int64_t row0 = owned_begin * le, n_own = owned_count * le, n_glob = global_layers * le;
    ASSERT_THROW(n_own * n_ranks == n_glob, "internal: the slabs of a gathered level must have equal size");
    const int64_t shift = global_begin * le; // local column -> global column
    double nnz_max = (double)(A.row_ptr[row0 + n_own] - A.row_ptr[row0]);
    nnz_max = h.allreduce_max(nnz_max);
    const int64_t pad = (int64_t)nnz_max;
    // [row lengths | near-null | columns (as doubles, exact) | values], padded to the same length on every rank
    const int64_t each = 2 * n_own + 2 * pad;
    std::vector<double> send((size_t)each, 0.);
    for (int64_t r = 0; r < n_own; ++r)
    {
      send[r] = (double)(A.row_ptr[row0 + r + 1] - A.row_ptr[row0 + r]);
      send[n_own + r] = B[row0 + r];
    }
    const int p0 = A.row_ptr[row0];
    for (int p = p0; p < A.row_ptr[row0 + n_own]; ++p)
    {
      send[2 * n_own + (p - p0)] = (double)(A.col[p] + shift);
      send[2 * n_own + pad + (p - p0)] = A.val[p];
    }
    DeviceBuffer<double> d_in((size_t)each), d_out((size_t)each * n_ranks);
    MFMG_HIP_CHECK(hipMemcpyAsync(d_in.data(), send.data(), (size_t)each * sizeof(double), hipMemcpyHostToDevice, h.stream));
    h.comm.transport->allgather(d_in.data(), each, d_out.data(), h.stream);
    std::vector<double> all = d_out.download(h.stream);
    Ag.n_rows = Ag.n_cols = n_glob;
    Ag.row_ptr.assign(n_glob + 1, 0);
    Bg.assign((size_t)n_glob, 0.);
    for (int rk = 0; rk < n_ranks; ++rk)
      for (int64_t r = 0; r < n_own; ++r)
      {
        Ag.row_ptr[rk * n_own + r + 1] = (int32_t)all[(size_t)rk * each + r];
        Bg[rk * n_own + r] = all[(size_t)rk * each + n_own + r];
      }
    for (int64_t r = 0; r < n_glob; ++r)
    {
      ASSERT_THROW((int64_t)Ag.row_ptr[r] + Ag.row_ptr[r + 1] < (int64_t(1) << 31), "gathered operator exceeds int32 entries");
      Ag.row_ptr[r + 1] += Ag.row_ptr[r];
    }
    Ag.col.resize(Ag.row_ptr[n_glob]);
    Ag.val.resize(Ag.row_ptr[n_glob]);
    for (int rk = 0; rk < n_ranks; ++rk)
    {
      const int64_t base = Ag.row_ptr[rk * n_own];
      const int64_t cnt = Ag.row_ptr[(rk + 1) * n_own] - base;
      for (int64_t q = 0; q < cnt; ++q)
      {
        Ag.col[base + q] = (int32_t)all[(size_t)rk * each + 2 * n_own + q];
        Ag.val[base + q] = all[(size_t)rk * each + 2 * n_own + pad + q];
      }
    }
    _amg_gather_level = (int)_amg.size();
    _gather_space = space;
    _gather_in.resize((size_t)n_own);
    _gather_b = std::make_shared<DVector>(h, n_glob);
    _gather_x = std::make_shared<DVector>(h, n_glob);
  }
  // geometric hint of the (global) level: the same blocks of nodes the distributed levels use
  AmgGridHint hint;
  hint.dims[0] = dims[0];
  hint.dims[1] = dims[1];
  hint.dims[2] = (int)global_layers;
  hint.n_components = n_comp;
  hint.node_of_row.resize((size_t)Ag.n_rows);
  hint.component_of_row.resize((size_t)Ag.n_rows);
  for (int64_t r = 0; r < Ag.n_rows; ++r)
  {
    hint.node_of_row[r] = (int32_t)(r / n_comp);
    hint.component_of_row[r] = (int32_t)(r % n_comp);
  }
  const int blk = this->_params->get("solver.amg.aggregate_block", 2);
  for (int d = 0; d < 3; ++d)
    hint.block[d] = blk;
  auto host_levels = build_aggregation_hierarchy(std::move(Ag), std::move(Bg), opts, &hint);
  const size_t first = _amg.size();
  _amg.resize(first + host_levels.size());
  for (size_t l = 0; l < host_levels.size(); ++l)
  {
    AmgLevel &L = _amg[first + l];
    if (l == 0 && !distributed)
      L.a = a_op; // the operator the caller handed in (local = global)
    else
      L.a = std::make_shared<HipMatrixOperator>(upload_csr(h, std::move(host_levels[l].A)));
    if (l + 1 < host_levels.size())
    {
      L.prolongator = std::make_shared<HipMatrixOperator>(upload_csr(h, std::move(host_levels[l].P)));
      L.restrictor = std::dynamic_pointer_cast<HipMatrixOperator>(L.prolongator->transpose());
      L.smoother = std::make_shared<HipSmoother>(L.a, smoother_params);
    }
  }
  auto last = _amg.back().a->get_matrix();
  ASSERT_THROW(last->m() <= 16384, "the coarsest level of the multilevel solver is too large for the dense LU (" +
                                       std::to_string(last->m()) + " rows)");
  setup_direct(last, _amg_bottom);
}
} // namespace mfmg
