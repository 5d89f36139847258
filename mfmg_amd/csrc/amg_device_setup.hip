// Setup of the smoothed-aggregation hierarchy below the first coarse level ON THE DEVICE, by probing, for runs
// where the levels are coupled across the ranks of a slab or box decomposition (and, on request, for one rank).
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

// a level: its nodes as a halo space (local box with ghost layers, owned box, global position and size per axis;
// comps entries per node) + the reach of its operator
struct LevelGeom
{
  HaloSpace s;
  int space = 0; // index of the halo space of the level's vectors (0: one rank)
  int reach = 1; // stencil reach of the level's operator in nodes
  int64_t n_rows() const { return s.n_local(); }
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
  // (distributed default: 20 000 global rows.  With 200 000 -- the default until the end of round 3 -- a rank of the 2 x 2 x 2 run
  // of 512^3 cells replicated the level of 65 536 global rows, whose stencils have grown to 2000 entries per row at that depth:
  // 1.6 GB per application; measured with a wire that costs nothing (scratch/rank_cycle_on_one_gpu.py): 2.67 -> 2.40 ms per
  // cycle and rank, for three more exchanges)
  const int64_t replicate_rows = this->_params->get("solver.amg.replicate_rows", distributed ? 20000 : 4000);
  const int C = std::max(grid.n_components, 1);
  // aggregates: cubes of `blk` nodes (solver.amg.aggregate_block), aligned globally.  With 2 the box stencil of the
  // operators grows from level to level (reach 1, 2, 3, 5: 53, 236, 582, 1520 entries per row); with 3 -- the classic
  // coarsening of smoothed aggregation -- a reach of 1 stays 1.  General rule: r_{l+1} = floor((blk - 1 + 3 r_l) / blk).
  const int blk = grid.block[0];
  ASSERT_THROW(blk >= 2 && grid.block[1] == blk && grid.block[2] == blk, "the device setup needs cubic aggregates");

  LevelGeom g;
  g.reach = 1;
  if (distributed)
  {
    g.s = comm.spaces[op_space];
    g.s.check();
    for (int d = 0; d < 3; ++d)
      ASSERT_THROW(g.s.dim(d) == std::max(grid.dims[d], 1), "internal: coarse space does not match the agglomerate grid");
    ASSERT_THROW(g.s.comps == C, "internal: coarse space does not match the agglomerate grid");
    g.space = op_space;
  }
  else
  {
    g.s.set_whole_xy(std::max(grid.dims[0], 1), std::max(grid.dims[1], 1), C);
    g.s.layer_elems = (int64_t)C * g.s.n_xy[0] * g.s.n_xy[1];
    g.s.n_layers = g.s.owned_count = g.s.global_layers = std::max(grid.dims[2], 1);
  }
  ASSERT_THROW(matrix->m() == g.n_rows(), "internal: operator rows do not match the agglomerate grid");
  for (int64_t r = 0; r < (int64_t)grid.node_of_row.size(); r += std::max<int64_t>(1, (int64_t)grid.node_of_row.size() / 8191))
    ASSERT_THROW(grid.node_of_row[r] == r / C && (grid.component_of_row.empty() ? 0 : grid.component_of_row[r]) == r % C,
                 "the device setup of the aggregation hierarchy needs node-major rows with a fixed number of components");

  // host copy of the current level: near-null vector (local, owned part valid)
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
    const int64_t global_rows = g.s.n_global();
    MemoryPhase phase("Setup: build coarse solver, aggregation level " + std::to_string(level) + " (" + std::to_string(global_rows) + " rows)");
    // ---- can the next level be built distributed?
    LevelGeom c;
    c.reach = (blk - 1 + 3 * g.reach) / blk;
    bool coarsen_here = global_rows > std::max<int64_t>(replicate_rows, opts.coarsest_size) && level + 1 < opts.max_levels;
    if (coarsen_here)
    {
      // per axis: an axis the ranks split needs whole aggregates per rank, aligned globally, and the ghost layers of the
      // coarse level must come from the immediate neighbour; an axis one rank holds whole rounds up
      double ok = 1.;
      int64_t cdim[3], cown0[3], cown_n[3], cg0[3], cgn[3];
      for (int d = 0; d < 3; ++d)
      {
        const bool split = distributed && comm.grid[d] > 1;
        if (split)
        {
          if (!(g.s.own_n(d) % blk == 0 && (g.s.g0(d) + g.s.own0(d)) % blk == 0 && g.s.own_n(d) / blk >= c.reach && g.s.gn(d) % blk == 0))
            ok = 0.;
          cown_n[d] = g.s.own_n(d) / blk;
          cown0[d] = g.s.low(d) ? c.reach : 0;
          cdim[d] = cown0[d] + cown_n[d] + (g.s.high(d) ? c.reach : 0);
          cg0[d] = (g.s.g0(d) + g.s.own0(d)) / blk - cown0[d];
          cgn[d] = g.s.gn(d) / blk;
        }
        else
        {
          cdim[d] = cown_n[d] = cgn[d] = (g.s.dim(d) + blk - 1) / blk;
          cown0[d] = cg0[d] = 0;
        }
      }
      if (distributed)
      {
        ok = -h.allreduce_max(-ok); // min over the ranks
        coarsen_here = ok > 0.5;
      }
      c.s.comps = C;
      c.s.width = c.reach;
      for (int d = 0; d < 2; ++d)
      {
        c.s.n_xy[d] = cdim[d];
        c.s.own0_xy[d] = cown0[d];
        c.s.own_n_xy[d] = cown_n[d];
        c.s.g0_xy[d] = cg0[d];
        c.s.gn_xy[d] = cgn[d];
        c.s.low_xy[d] = g.s.low(d);
        c.s.high_xy[d] = g.s.high(d);
      }
      c.s.layer_elems = (int64_t)C * cdim[0] * cdim[1];
      c.s.n_layers = cdim[2];
      c.s.owned_begin = cown0[2];
      c.s.owned_count = cown_n[2];
      c.s.global_begin = cg0[2];
      c.s.global_layers = cgn[2];
      c.s.has_low = g.s.has_low;
      c.s.has_high = g.s.has_high;
    }
    if (!coarsen_here)
    {
      // ---- gather this level; the rest of the hierarchy is replicated (host setup of one rank)
      A.n_rows = A.n_cols = g.n_rows();
      a_op->get_matrix()->download(A.row_ptr, A.col, A.val);
      finish_amg_replicated(a_op, std::move(A), std::move(B), g.space, g.s, opts, smoother_params);
      break;
    }
    const double t0 = wall_now();
    if (distributed)
    {
      c.space = comm.add_space(c.s);
      // the fine level of this pair exchanges as many layers as its operator reaches
      HaloSpace &fs = comm.spaces[g.space];
      fs.width = std::max(fs.width, g.reach);
      g.s.width = fs.width;
      for (int d = 0; d < 3; ++d)
        ASSERT_THROW((!fs.low(d) || fs.width <= fs.own0(d)) && (!fs.high(d) || fs.width <= fs.dim(d) - fs.own0(d) - fs.own_n(d)),
                     "internal: not enough ghost layers for the stencil of this level");
    }
    const int64_t n_f = g.n_rows(), n_c = c.n_rows();

    // ---- diagonal, rho = max_i sum_j |a_ij| / |a_ii| over the owned rows of all ranks
    DeviceBuffer<double> d_dinv((size_t)n_f);
    double rho = 0.;
    {
      DeviceBuffer<double> d_ratio((size_t)n_f);
      a_op->get_matrix()->row_ratios(d_dinv.data(), d_ratio.data());
      const std::vector<double> ratio = d_ratio.download(h.stream);
#pragma omp parallel for schedule(static) reduction(max : rho)
      for (int64_t i = 0; i < n_f; ++i)
        if (g.s.owned(i))
          rho = std::max(rho, ratio[i]);
    }
    rho = h.allreduce_max(rho); // (first the collective, then the check: every rank throws or none does)
    ASSERT_THROW(rho < HUGE_VAL, "zero diagonal in the multilevel coarse solver setup");
    const double w = opts.omega / rho;

    // ---- tentative prolongator: t = B / |B|_aggregate on the owned nodes, exchanged to the ghosts; B_c = |B|_aggregate
    auto agg_row = [&](int64_t row) {
      int64_t nd[3];
      g.s.node_of(row, nd);
      int64_t a[3];
      for (int d = 0; d < 3; ++d)
        a[d] = (nd[d] + g.s.g0(d)) / blk - c.s.g0(d); // local coarse node
      return ((a[2] * c.s.n_xy[1] + a[1]) * c.s.n_xy[0] + a[0]) * C + row % C;
    };
    std::vector<double> norm2((size_t)n_c, 0.), t((size_t)n_f, 0.), Bc((size_t)n_c, 0.);
    for (int64_t i = 0; i < n_f; ++i)
      if (g.s.owned(i))
        norm2[agg_row(i)] += B[i] * B[i];
    for (int64_t i = 0; i < n_f; ++i)
      if (g.s.owned(i))
      {
        const double nn = norm2[agg_row(i)];
        ASSERT_THROW(nn > 0., "the device setup of the aggregation hierarchy needs a near-null-space vector without zero aggregates");
        t[i] = B[i] / std::sqrt(nn);
      }
    for (int64_t r = 0; r < n_c; ++r)
      if (c.s.owned(r))
        Bc[r] = std::sqrt(norm2[r]);
    DVector t_dev(h, n_f), y_f(h, n_f), z_f(h, n_f), u_c(h, n_c);
    MFMG_HIP_CHECK(hipMemcpyAsync(t_dev.get_values(), t.data(), (size_t)n_f * sizeof(double), hipMemcpyHostToDevice, h.stream));
    h.exchange(g.space, t_dev.get_values());

    // ---- P = (I - w D^-1 A) P_tent by probing: the columns of aggregates floor((blk - 1 + 2 reach) / blk) + 1 apart
    //      (1 + reach for blk = 2) are disjoint
    const int fdims[3] = {(int)g.s.dim(0), (int)g.s.dim(1), (int)g.s.dim(2)}, cdims[3] = {(int)c.s.dim(0), (int)c.s.dim(1), (int)c.s.dim(2)};
    const int f_off[3] = {(int)g.s.g0(0), (int)g.s.g0(1), (int)g.s.g0(2)}, c_off[3] = {(int)c.s.g0(0), (int)c.s.g0(1), (int)c.s.g0(2)};
    int period_p[3];
    for (int d = 0; d < 3; ++d)
      period_p[d] = (int)std::max<int64_t>(1, std::min<int64_t>((blk - 1 + 2 * g.reach) / blk + 1, c.s.gn(d)));
    const int n_col_p = period_p[0] * period_p[1] * period_p[2] * C;
    // (the probes stay on the device and the rows are assembled there: probe_assembly.hip)
    DeviceBuffer<double> Z;
    {
      MemoryKind probe_kind("probe vectors (freed after the level is assembled)");
      Z.resize((size_t)n_col_p * (size_t)n_f);
    }
    for (int col = 0; col < n_col_p; ++col)
    {
      const int comp = col % C, oc = col / C;
      const int phase[3] = {oc % period_p[0], (oc / period_p[0]) % period_p[1], oc / (period_p[0] * period_p[1])};
      vec::select_rows(h, fdims, C, blk, f_off, period_p, phase, comp, t_dev.get_values(), y_f.get_values());
      a_op->get_matrix()->vmult(Z.data() + (size_t)col * n_f, y_f.get_values()); // ghosts of y are set locally: no exchange
    }
    MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
    const double t_probed = wall_now();
    // candidates of owned row i: aggregates whose nodes lie within `reach` of its node; v = t_i [own aggregate] - w d_i^-1 (A y)_i
    auto p_mat = prolongator_from_probes(h, g.s, c.s, blk, g.reach, period_p, w, Z.data(), t_dev.get_values(), d_dinv.data());
    Z.release();
    const double t_assembled = wall_now();
    auto pt_mat = p_mat->transpose();
    if (verbose)
      std::fprintf(stderr, "[mfmg_hip] amg level %d: P probes %.2f s, assembly %.2f s, transpose %.2f s\n", level,
                   t_probed - t0, t_assembled - t_probed, wall_now() - t_assembled);
    const double t1 = wall_now();

    // ---- A_c = P^T A P by probing: coarse nodes (2 reach_c + 1) apart never meet in a row
    int period_a[3];
    for (int d = 0; d < 3; ++d)
      period_a[d] = (int)std::max<int64_t>(1, std::min<int64_t>(2 * c.reach + 1, c.s.gn(d)));
    const int n_col_a = period_a[0] * period_a[1] * period_a[2] * C;
    // Near the bottom of the hierarchy the periods reach the whole level -- every coarse column its own probe (8192 at 513^3
    // DoFs: 3.8 s) --: there P^T (A P) is formed by the CSR product on the device instead (csr_algebra.hip; rows of <= 8192
    // columns in a table addressed by the column).  One rank only; a hierarchy rounded to float keeps the probes, whose
    // assembly rounds.
    std::shared_ptr<SparseMatrixDevice<double>> ac_mat;
    const bool by_product = !distributed && !h.setup_values_float && n_col_a >= 1024 && n_c <= 8192 && n_col_a * 2 >= n_c;
    if (by_product)
    {
      auto ap = a_op->get_matrix()->mmult(*p_mat);
      ac_mat = pt_mat->mmult(*ap);
    }
    else
    {
      DeviceBuffer<double> Y;
      {
        MemoryKind probe_kind("probe vectors (freed after the level is assembled)");
        Y.resize((size_t)n_col_a * (size_t)n_c);
      }
      for (int col = 0; col < n_col_a; ++col)
      {
        const int comp = col % C, oc = col / C;
        const int phase[3] = {oc % period_a[0], (oc / period_a[0]) % period_a[1], oc / (period_a[0] * period_a[1])};
        double *y_c = Y.data() + (size_t)col * n_c;
        vec::select_rows(h, cdims, C, 1, c_off, period_a, phase, comp, nullptr, u_c.get_values());
        p_mat->vmult(y_f.get_values(), u_c.get_values());
        h.exchange(g.space, y_f.get_values());
        a_op->get_matrix()->vmult(z_f.get_values(), y_f.get_values());
        pt_mat->vmult(y_c, z_f.get_values());
        h.exchange_reverse_add(c.space, y_c);
      }
      MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
      ac_mat = coarse_operator_from_probes(h, c.s, c.reach, period_a, Y.data());
      Y.release();
    }
    if (verbose)
      std::fprintf(stderr, "[mfmg_hip] amg level %d on the device (%lld local rows, reach %d): P %d probes %.2f s, A_c %d %s %.2f s\n",
                   level, (long long)n_f, g.reach, n_col_p, t1 - t0, n_col_a, by_product ? "columns by the CSR product" : "probes", wall_now() - t1);

    // ---- the level's operators
    AmgLevel L;
    L.a = a_op;
    L.prolongator = std::make_shared<HipMatrixOperator>(p_mat);
    L.prolongator->set_spaces(c.space, 0); // x -= P x_c reads the ghost aggregates of x_c
    L.restrictor = std::make_shared<HipMatrixOperator>(pt_mat);
    L.restrictor->set_spaces(0, 0);        // P^T has entries in owned fine rows only ...
    L.restrictor->set_reverse_range_space(c.space); // ... and returns the sums of ghost aggregates to their owners
    L.smoother = std::make_shared<HipSmoother>(L.a, smoother_params);
    // ---- P~ = (I - beta D^-1 A) P for a level that runs V(0,1) with a damped-Jacobi post-smoother: the correction of such a
    //      level is x = P x_c followed by x' = x - beta D^-1 (A x - b) = P~ x_c + beta D^-1 b.  Probing with the columns of P
    //      themselves: the columns of coarse nodes floor((blk - 1 + 2 R) / blk) + 1 apart are disjoint, R = 2 reach the reach of
    //      P~ in fine nodes.  ("solver.amg.smoothed_prolongation false" keeps the two steps; a hierarchy whose matrices are
    //      rounded to float keeps them too: the rounded P~ would not be the product of the rounded matrices.)
    static const bool smoothed_env = !(std::getenv("MFMG_AMG_SMOOTHED_PROLONGATION") && std::string(std::getenv("MFMG_AMG_SMOOTHED_PROLONGATION")) == "0");
    // Where it pays (measured at 257^3 DoFs, one GPU, us per cycle, two steps -> one): levels whose operators are evaluated from
    // stencil tables (4.2 M rows: 33 + 67 -> 76; 524 k: 16 + 38 -> 52) and small levels, which are bound by their launches
    // (8192 rows: 10 + 40 -> 22); in between (65 536 rows) P~ has 488 stored entries per row where A comes from tables and P has
    // 105: 18 + 25 -> 70, so one rank keeps the two steps there.  A distributed run takes it wherever it fits: every level
    // saves a blocking exchange.
    // (one rank, measured end to end with all of them: 1.677-1.702 ms per cycle, 1.696-1.700 without, and 0.5-0.8 s more setup:
    // there the small levels take it by default -- 0.04 s of setup --, "solver.amg.smoothed_prolongation true" adds all large ones)
    const bool one_rank_on = this->_params->get_optional<bool>("solver.amg.smoothed_prolongation").value_or(false);
    // (... and the largest level, >= 2 M rows: 33 + 67 -> 76 us per cycle for 0.2 s of setup; measured end to end with the small
    // level: 1.681-1.703 -> 1.652-1.653 ms per cycle on the same box.  MFMG_AMG_SMOOTHED_LARGE=rows moves that threshold.)
    static const int64_t large_rows = std::getenv("MFMG_AMG_SMOOTHED_LARGE") ? std::atoll(std::getenv("MFMG_AMG_SMOOTHED_LARGE")) : 2000000;
    // (only the first level of the hierarchy, whose operator reaches one node, and up to 8 M rows: on the global problem of
    // BASELINE configs[3] on one GPU -- 33.5 M rows on that level -- the cycle went from 11.8 to 13.4 ms and the setup from 27 to 33 s)
    // ... and only where the level operator repeats its stencils (a variable coefficient stores P~ entry by entry: 1 % of the cycle
    // for 1.5 s of setup)
    const bool large = level == 0 && g.reach == 1 && n_f >= large_rows && n_f <= (int64_t(1) << 23) && a_op->get_matrix()->stencil_classes() > 0;
    // (MFMG_AMG_SMOOTHED_DISTRIBUTED=0: a distributed run decides like one rank -- measurement switch)
    static const bool distributed_all = !(std::getenv("MFMG_AMG_SMOOTHED_DISTRIBUTED") && std::string(std::getenv("MFMG_AMG_SMOOTHED_DISTRIBUTED")) == "0");
    const bool pays = (distributed && distributed_all) || n_f <= 16384 || large || (one_rank_on && n_f >= 262144);
    if (smoothed_env && pays && level >= _amg_pre_smoothing_levels && L.smoother->coefficients().size() == 1 && !h.setup_values_float &&
        this->_params->get("solver.amg.smoothed_prolongation", true))
    {
      const double t2 = wall_now();
      const double beta = L.smoother->coefficients()[0].second;
      const int R = 2 * g.reach;
      // the coarse nodes a row of P~ couples to -- up to R fine nodes beyond the owned box -- must lie inside the local coarse
      // box and inside the layers its exchange refreshes; decided collectively (a rank that skipped it would wait alone)
      double misfit = 0.;
      for (int d = 0; d < 3; ++d)
      {
        const int64_t f0 = g.s.g0(d) + g.s.own0(d), f1 = g.s.g0(d) + g.s.own0(d) + g.s.own_n(d) - 1; // owned fine nodes, global
        const int64_t lo = std::max<int64_t>(0, f0 - R) / blk, hi = std::min<int64_t>(c.s.gn(d) - 1, (f1 + R) / blk);
        const int64_t c_lo = c.s.g0(d), c_hi = c.s.g0(d) + c.s.dim(d) - 1;
        const int64_t own_lo = c.s.g0(d) + c.s.own0(d), own_hi = own_lo + c.s.own_n(d) - 1;
        if (lo < c_lo || hi > c_hi || own_lo - lo > (distributed ? c.s.width : own_lo) || hi - own_hi > (distributed ? c.s.width : hi))
          misfit = 1.;
      }
      if (distributed)
        misfit = h.allreduce_max(misfit);
      if (misfit == 0.)
      {
      int period_t[3];
      for (int d = 0; d < 3; ++d)
        period_t[d] = (int)std::max<int64_t>(1, std::min<int64_t>((blk - 1 + 2 * R) / blk + 1, c.s.gn(d)));
      const int n_col_t = period_t[0] * period_t[1] * period_t[2] * C;
      DeviceBuffer<double> Yp((size_t)n_col_t * (size_t)n_f), Zt((size_t)n_col_t * (size_t)n_f);
      for (int col = 0; col < n_col_t; ++col)
      {
        const int comp = col % C, oc = col / C;
        const int phase[3] = {oc % period_t[0], (oc / period_t[0]) % period_t[1], oc / (period_t[0] * period_t[1])};
        double *yp = Yp.data() + (size_t)col * n_f;
        vec::select_rows(h, cdims, C, 1, c_off, period_t, phase, comp, nullptr, u_c.get_values()); // (ghost aggregates set locally)
        p_mat->vmult(yp, u_c.get_values());
        h.exchange(g.space, yp);
        a_op->get_matrix()->vmult(Zt.data() + (size_t)col * n_f, yp);
      }
      MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
      auto pt_smoothed = prolongator_from_probes(h, g.s, c.s, blk, R, period_t, beta, Zt.data(), nullptr, L.a->get_diagonal_inverse(), Yp.data());
      Yp.release();
      Zt.release();
      L.smoothed_prolongator = std::make_shared<HipMatrixOperator>(pt_smoothed);
      L.smoothed_prolongator->set_spaces(c.space, 0);
      L.smoothed_beta = beta;
      if (verbose)
        std::fprintf(stderr, "[mfmg_hip] amg level %d: smoothed prolongator of the cycle, %d probes, %lld entries, %.2f s\n", level, n_col_t,
                     (long long)pt_smoothed->n_nonzero_elements(), wall_now() - t2);
      }
      else if (verbose)
        std::fprintf(stderr, "[mfmg_hip] amg level %d: the smoothed prolongator does not fit the ghost layers of the coarse level, two steps kept\n", level);
    }
    _amg.push_back(std::move(L));

    // ---- next level
    a_op = std::make_shared<HipMatrixOperator>(ac_mat);
    a_op->set_spaces(c.space, c.space);
    B = std::move(Bc);
    g = c;
  }
}

// The level `a_op` (owned rows in `A`, local numbering; `geom` its nodes) becomes the first replicated level: its operator and near-null
// vector are gathered, the remaining hierarchy is built by the host code of one rank, identically on every rank.
void HipSolver::finish_amg_replicated(std::shared_ptr<HipMatrixOperator> a_op, HostCsr A, std::vector<double> B, int space,
                                      HaloSpace const &geom, AmgOptions const &opts, std::shared_ptr<ptree> smoother_params)
{
  HipHandle &h = _handle;
  const bool distributed = h.comm.enabled() && space > 0;
  const int n_comp = geom.comps;
  HostCsr Ag;
  std::vector<double> Bg;
  if (!distributed)
  {
    Ag = std::move(A);
    Bg = std::move(B);
  }
  else
  {
    HaloCommunicator const &comm = h.comm;
    const int n_ranks = comm.n_ranks;
    const int64_t n_loc = geom.n_local(), n_own = geom.n_owned(), n_glob = geom.n_global();
    ASSERT_THROW(n_own * n_ranks == n_glob, "internal: the boxes of a gathered level must have equal size");
    // owned rows in local lexicographic order
    std::vector<int64_t> own_rows;
    own_rows.reserve((size_t)n_own);
    int64_t nnz_own = 0;
    for (int64_t i = 0; i < n_loc; ++i)
      if (geom.owned(i))
      {
        own_rows.push_back(i);
        nnz_own += A.row_ptr[i + 1] - A.row_ptr[i];
      }
    ASSERT_THROW((int64_t)own_rows.size() == n_own, "internal: owned rows of a gathered level");
    const int64_t pad = (int64_t)h.allreduce_max((double)nnz_own);
    // [row lengths | near-null | GLOBAL columns (as doubles, exact) | values], padded to the same length on every rank
    const int64_t each = 2 * n_own + 2 * pad;
    std::vector<double> send((size_t)each, 0.);
    {
      int64_t q0 = 0;
      for (int64_t q = 0; q < n_own; ++q)
      {
        const int64_t r = own_rows[q];
        send[q] = (double)(A.row_ptr[r + 1] - A.row_ptr[r]);
        send[n_own + q] = B[r];
        for (int p = A.row_ptr[r]; p < A.row_ptr[r + 1]; ++p, ++q0)
        {
          send[2 * n_own + q0] = (double)geom.global_id(A.col[p]);
          send[2 * n_own + pad + q0] = A.val[p];
        }
      }
    }
    DeviceBuffer<double> d_in((size_t)each), d_out((size_t)each * n_ranks);
    MFMG_HIP_CHECK(hipMemcpyAsync(d_in.data(), send.data(), (size_t)each * sizeof(double), hipMemcpyHostToDevice, h.stream));
    h.comm.transport->allgather(d_in.data(), each, d_out.data(), h.stream);
    std::vector<double> all = d_out.download(h.stream);
    // q-th owned row of rank rk -> its global row (equal boxes: the owned box of rank rk starts at coord * owned nodes)
    const int64_t on[3] = {geom.own_n(0), geom.own_n(1), geom.own_n(2)};
    auto global_row = [&](int rk, int64_t q) {
      const int cr[3] = {rk % comm.grid[0], (rk / comm.grid[0]) % comm.grid[1], rk / (comm.grid[0] * comm.grid[1])};
      const int64_t nd = q / n_comp;
      const int64_t o[3] = {nd % on[0], (nd / on[0]) % on[1], nd / (on[0] * on[1])};
      int64_t gc[3];
      for (int d = 0; d < 3; ++d)
        gc[d] = (comm.grid[d] > 1 ? (int64_t)cr[d] * on[d] : 0) + o[d];
      return ((gc[2] * geom.gn(1) + gc[1]) * geom.gn(0) + gc[0]) * n_comp + q % n_comp;
    };
    Ag.n_rows = Ag.n_cols = n_glob;
    Ag.row_ptr.assign(n_glob + 1, 0);
    Bg.assign((size_t)n_glob, 0.);
    std::vector<int32_t> from_ranked((size_t)n_glob, -1);
    for (int rk = 0; rk < n_ranks; ++rk)
      for (int64_t q = 0; q < n_own; ++q)
      {
        const int64_t gr = global_row(rk, q);
        ASSERT_THROW(gr >= 0 && gr < n_glob && from_ranked[gr] < 0, "internal: the boxes of a gathered level do not tile the global level");
        from_ranked[gr] = (int32_t)(rk * n_own + q);
        Ag.row_ptr[gr + 1] = (int32_t)all[(size_t)rk * each + q];
        Bg[gr] = all[(size_t)rk * each + n_own + q];
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
      int64_t q0 = 0;
      for (int64_t q = 0; q < n_own; ++q)
      {
        const int64_t gr = global_row(rk, q);
        const int64_t len = Ag.row_ptr[gr + 1] - Ag.row_ptr[gr];
        for (int64_t e = 0; e < len; ++e, ++q0)
        {
          Ag.col[Ag.row_ptr[gr] + e] = (int32_t)all[(size_t)rk * each + 2 * n_own + q0];
          Ag.val[Ag.row_ptr[gr] + e] = all[(size_t)rk * each + 2 * n_own + pad + q0];
        }
      }
    }
    // (the columns of a row arrive in local order; a box leaves them unsorted in the global numbering)
    if (geom.split_xy())
#pragma omp parallel for schedule(static)
      for (int64_t r = 0; r < n_glob; ++r)
      {
        const int p0 = Ag.row_ptr[r], p1 = Ag.row_ptr[r + 1];
        std::vector<std::pair<int32_t, double>> row((size_t)(p1 - p0));
        for (int p = p0; p < p1; ++p)
          row[p - p0] = {Ag.col[p], Ag.val[p]};
        std::sort(row.begin(), row.end());
        for (int p = p0; p < p1; ++p)
        {
          Ag.col[p] = row[p - p0].first;
          Ag.val[p] = row[p - p0].second;
        }
      }
    _amg_gather_level = (int)_amg.size();
    _gather_space = space;
    _gather_in.resize((size_t)n_own);
    _gather_b = std::make_shared<DVector>(h, n_glob);
    _gather_x = std::make_shared<DVector>(h, n_glob);
    if (geom.split_xy())
    {
      std::vector<int32_t> local_ids((size_t)n_loc);
      for (int64_t i = 0; i < n_loc; ++i)
        local_ids[i] = (int32_t)geom.global_id(i);
      _gather_ranked.resize((size_t)n_glob);
      _gather_from_ranked.upload(from_ranked.data(), from_ranked.size(), h.stream);
      _gather_local_ids.upload(local_ids.data(), local_ids.size(), h.stream);
      MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
    }
  }
  // geometric hint of the (global) level: the same blocks of nodes the distributed levels use
  AmgGridHint hint;
  hint.dims[0] = (int)geom.gn(0);
  hint.dims[1] = (int)geom.gn(1);
  hint.dims[2] = (int)geom.gn(2);
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
      // prolongation and post-smoothing as one operator (AmgLevel::smoothed_prolongator) on the small levels built here: P~ =
      // (I - beta D^-1 A) P column by column -- n_c applications of P and A on the device -- and kept as CSR without its zeros.
      // The stencils of these levels span most of the level (8192 rows of 3900 entries below a 512^3-cell mesh): P~ is hardly
      // larger than P and the level operator, the largest matrix of the replicated part, leaves the cycle (95 + 19 -> ~25 us).
      const int64_t nf = L.a->get_matrix()->m(), nc = L.prolongator->get_matrix()->n();
      static const bool smoothed_env = !(std::getenv("MFMG_AMG_SMOOTHED_PROLONGATION") && std::string(std::getenv("MFMG_AMG_SMOOTHED_PROLONGATION")) == "0");
      if (smoothed_env && (int)(first + l) >= _amg_pre_smoothing_levels && L.smoother->coefficients().size() == 1 && !h.setup_values_float &&
          this->_params->get("solver.amg.smoothed_prolongation", true) && nf <= 16384 && nf * nc <= (int64_t(1) << 25))
      {
        const double beta = L.smoother->coefficients()[0].second;
        DVector e(h, nc), y(h, nf), z(h, nf);
        DeviceBuffer<double> dense((size_t)nf * (size_t)nc); // column j at dense[j nf ...]
        double const *dinv = L.a->get_diagonal_inverse();
        e = 0.;
        const std::vector<double> unit = {1., 0.}; // (alive until the download below has synchronised the stream)
        double const &one = unit[0], &zero = unit[1];
        for (int64_t j = 0; j < nc; ++j)
        {
          MFMG_HIP_CHECK(hipMemcpyAsync(e.get_values() + j, &one, sizeof(double), hipMemcpyHostToDevice, h.stream));
          L.prolongator->get_matrix()->vmult(y.get_values(), e.get_values());
          L.a->get_matrix()->vmult(z.get_values(), y.get_values());
          // column = y - beta D^-1 z  (the fused first-term epilogue of the vector kernels: out = x - beta dinv (A x - b) with A x = z, b = 0)
          vec::scaled_pointwise<double>(h, nf, -beta, dinv, z.get_values(), z.get_values());
          vec::sadd<double>(h, nf, 1., 1., y.get_values(), z.get_values()); // z = z + y
          MFMG_HIP_CHECK(hipMemcpyAsync(dense.data() + (size_t)j * nf, z.get_values(), (size_t)nf * sizeof(double), hipMemcpyDeviceToDevice, h.stream));
          MFMG_HIP_CHECK(hipMemcpyAsync(e.get_values() + j, &zero, sizeof(double), hipMemcpyHostToDevice, h.stream));
        }
        const std::vector<double> cols = dense.download(h.stream);
        HostCsr pt;
        pt.n_rows = nf;
        pt.n_cols = nc;
        pt.row_ptr.assign((size_t)nf + 1, 0);
        for (int64_t i = 0; i < nf; ++i)
        {
          int32_t cnt = 0;
          for (int64_t j = 0; j < nc; ++j)
            cnt += cols[(size_t)j * nf + i] != 0. ? 1 : 0;
          pt.row_ptr[i + 1] = pt.row_ptr[i] + cnt;
        }
        pt.col.resize((size_t)pt.row_ptr[nf]);
        pt.val.resize((size_t)pt.row_ptr[nf]);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nf; ++i)
        {
          int32_t p = pt.row_ptr[i];
          for (int64_t j = 0; j < nc; ++j)
            if (cols[(size_t)j * nf + i] != 0.)
            {
              pt.col[p] = (int32_t)j;
              pt.val[p++] = cols[(size_t)j * nf + i];
            }
        }
        L.smoothed_prolongator = std::make_shared<HipMatrixOperator>(upload_csr(h, std::move(pt), false));
        L.smoothed_beta = beta;
      }
    }
  }
  auto last = _amg.back().a->get_matrix();
  ASSERT_THROW(last->m() <= 16384, "the coarsest level of the multilevel solver is too large for the dense LU (" +
                                       std::to_string(last->m()) + " rows)");
  setup_direct(last, _amg_bottom);
}
} // namespace mfmg
