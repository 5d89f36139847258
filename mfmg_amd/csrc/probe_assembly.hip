// CSR matrices of the setup assembled ON THE DEVICE from the results of colour probes.
//
// The Galerkin products of the hierarchy are read off from operator applications (hip_hierarchy.hip: R A R^T,
// amg_device_setup.hip: P = S P_tent and P^T A P): `Y[colour][row]` holds the entry of `row` towards the one column of
// that colour within reach.  Until round 3 the probes were copied to the host, the rows assembled there (0.8 s for the
// 223 M entries of the first coarse operator at 257^3 DoFs, 0.5 s for the next level) and uploaded again; here two
// kernels (count, fill) walk the same candidate boxes in the same order and write the same entries, the row pointer is
// an exclusive scan of the counts (done on the host: one integer per row).  Arithmetic that the host loop performed
// (the damped-Jacobi update of the prolongator) is repeated with contraction off: same bits.
#include "probe_assembly.hpp"

#include <algorithm>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace mfmg
{
namespace
{
dim3 grid_for(int64_t work) { return dim3(n_blocks_for(work, 256, 1 << 16)); }

// counts[r + 1] (counts[0] = 0) -> row_ptr, in place on the device; returns the number of entries
int64_t scan_counts(HipHandle &h, DeviceBuffer<int32_t> &row_ptr, int64_t n_rows, char const *what)
{
  std::vector<int32_t> rp = row_ptr.download(h.stream);
  // two-pass scan over blocks of rows
  const int nt =
#ifdef _OPENMP
      omp_get_max_threads();
#else
      1;
#endif
  std::vector<int64_t> part((size_t)nt + 1, 0);
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    const int64_t b0 = n_rows * t / nt + 1, b1 = n_rows * (t + 1) / nt + 1;
    int64_t s = 0;
    for (int64_t r = b0; r < b1; ++r)
      s += rp[r];
    part[t + 1] = s;
#pragma omp barrier
#pragma omp single
    for (int q = 0; q < nt; ++q)
      part[q + 1] += part[q];
    int64_t run = part[t];
    for (int64_t r = b0; r < b1; ++r)
    {
      run += rp[r];
      rp[r] = (int32_t)std::min<int64_t>(run, INT32_MAX);
    }
  }
  const int64_t total = part[nt];
  ASSERT_THROW(total < (int64_t(1) << 31), std::string(what) + " exceeds int32 entries");
  MFMG_HIP_CHECK(hipMemcpyAsync(row_ptr.data(), rp.data(), rp.size() * sizeof(int32_t), hipMemcpyHostToDevice, h.stream));
  MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
  return total;
}

// ---- R A R^T ----------------------------------------------------------------------------------------------------
struct GalerkinGeom
{
  int na[3], ne, k[3];
  int off[3], own0[3], own1[3]; // global index of local agglomerate 0; owned agglomerates [own0, own1) per axis
  int all_owned;
  int64_t nc;
  int round_to_float; // values rounded to float-representable doubles ("setup value precision" float)
};

template <bool FILL>
__global__ void galerkin_rows_kernel(GalerkinGeom g, double const *Y, int32_t *row_ptr, int32_t *col, double *val)
{
  const int64_t n_agg = (int64_t)g.na[0] * g.na[1] * g.na[2];
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_agg * g.ne; r += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t a = r / g.ne;
    const int ax = (int)(a % g.na[0]), ay = (int)((a / g.na[0]) % g.na[1]), az = (int)(a / ((int64_t)g.na[0] * g.na[1]));
    // (rows of the neighbours' agglomerates stay empty)
    const bool owned = ax >= g.own0[0] && ax < g.own1[0] && ay >= g.own0[1] && ay < g.own1[1] && az >= g.own0[2] && az < g.own1[2];
    int p = FILL ? row_ptr[r] : 0;
    if (owned)
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx)
          {
            const int bx = ax + dx, by = ay + dy, bz = az + dz;
            if (bx < 0 || bx >= g.na[0] || by < 0 || by >= g.na[1] || bz < 0 || bz >= g.na[2])
              continue;
            if (FILL)
            {
              const int64_t b = bx + (int64_t)g.na[0] * (by + (int64_t)g.na[1] * bz);
              const int oc = ((bx + g.off[0]) % g.k[0]) + g.k[0] * (((by + g.off[1]) % g.k[1]) + g.k[1] * ((bz + g.off[2]) % g.k[2]));
              for (int e2 = 0; e2 < g.ne; ++e2, ++p)
              {
                col[p] = (int32_t)(b * g.ne + e2);
                double v = Y[(size_t)(oc * g.ne + e2) * (size_t)g.nc + (size_t)r];
                if (g.round_to_float)
                {
                  // (the product is symmetric to rounding; an entry and its transposed partner must round to the SAME float,
                  // or the symmetric-half storage is lost: both are replaced by their mean first -- one rank only, where
                  // every row is computed here)
                  if (g.all_owned)
                  {
                    const int ocr = ((ax + g.off[0]) % g.k[0]) + g.k[0] * (((ay + g.off[1]) % g.k[1]) + g.k[1] * ((az + g.off[2]) % g.k[2]));
                    const double vt = Y[(size_t)(ocr * g.ne + (int)(r % g.ne)) * (size_t)g.nc + (size_t)(b * g.ne + e2)];
                    v = 0.5 * (v + vt);
                  }
                  v = (double)(float)v;
                }
                val[p] = v;
              }
            }
            else
              p += g.ne;
          }
    if (!FILL)
      row_ptr[r + 1] = p;
  }
}

// ---- P = (I - w D^-1 A) P_tent -----------------------------------------------------------------------------------
struct ProlongatorGeom
{
  int fdims[3], cdims[3], gdims_c[3], C, blk, reach, period[3];
  int f_g0[3], c_g0[3];   // global index of the local fine node / coarse node 0 per axis
  int own0[3], own1[3];   // owned fine nodes [own0, own1) per axis: the other rows stay empty
  int64_t n_f;            // local fine rows = the stride of the probes
  double w;
  int round_to_float;
};

// (Yp != nullptr: the smoothed prolongator of the cycle, P~ = (I - w D^-1 A) P: the probes are columns of P -- Yp = P e, Z = A P e
// -- instead of columns of the tentative prolongator, whose only entry in row i is t_i)
template <bool FILL>
__global__ void prolongator_rows_kernel(ProlongatorGeom g, double const *Z, double const *t, double const *dinv, int32_t *row_ptr,
                                        int32_t *col, double *val, double const *Yp = nullptr)
{
#pragma clang fp contract(off) // multiply, multiply, subtract: the rounding of the host loop this replaces
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < g.n_f; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t nd = i / g.C;
    const int xl = (int)(nd % g.fdims[0]), yl = (int)((nd / g.fdims[0]) % g.fdims[1]), zl = (int)(nd / ((int64_t)g.fdims[0] * g.fdims[1]));
    const bool owned = xl >= g.own0[0] && xl < g.own1[0] && yl >= g.own0[1] && yl < g.own1[1] && zl >= g.own0[2] && zl < g.own1[2];
    int p = FILL ? row_ptr[i] : 0;
    if (owned)
    {
      const int x = xl + g.f_g0[0], y = yl + g.f_g0[1], zg = zl + g.f_g0[2]; // global node
      const int lo[3] = {max(0, x - g.reach) / g.blk, max(0, y - g.reach) / g.blk, max(0, zg - g.reach) / g.blk};
      const int hi[3] = {min(g.gdims_c[0] - 1, (x + g.reach) / g.blk), min(g.gdims_c[1] - 1, (y + g.reach) / g.blk),
                         min(g.gdims_c[2] - 1, (zg + g.reach) / g.blk)};
      for (int K = lo[2]; K <= hi[2]; ++K)
        for (int J = lo[1]; J <= hi[1]; ++J)
          for (int I = lo[0]; I <= hi[0]; ++I)
          {
            const int oc = (I % g.period[0]) + g.period[0] * ((J % g.period[1]) + g.period[1] * (K % g.period[2]));
            const int Il = I - g.c_g0[0], Jl = J - g.c_g0[1], Kl = K - g.c_g0[2];
            const bool own_agg = (I == x / g.blk) && (J == y / g.blk) && (K == zg / g.blk);
            for (int comp = 0; comp < g.C; ++comp)
            {
              const double ay = Z[(size_t)(oc * g.C + comp) * (size_t)g.n_f + (size_t)i];
              const double yi = Yp != nullptr ? Yp[(size_t)(oc * g.C + comp) * (size_t)g.n_f + (size_t)i]
                                              : ((own_agg && comp == (int)(i % g.C)) ? t[i] : 0.);
              const double v0 = yi - g.w * dinv[i] * ay;
              const double v = g.round_to_float ? (double)(float)v0 : v0;
              if (v != 0.)
              {
                if (FILL)
                {
                  col[p] = (int32_t)((((int64_t)Kl * g.cdims[1] + Jl) * g.cdims[0] + Il) * g.C + comp);
                  val[p] = v;
                }
                ++p;
              }
            }
          }
    }
    if (!FILL)
      row_ptr[i + 1] = p;
  }
}

// ---- A_c = P^T A P -----------------------------------------------------------------------------------------------
struct CoarseGeom
{
  int cdims[3], gdims_c[3], C, reach, period[3];
  int c_g0[3], own0[3], own1[3]; // global index of local node 0; owned nodes [own0, own1) per axis
  int all_owned;
  int64_t n_c;                   // local rows = the stride of the probes
  int round_to_float;
};

template <bool FILL>
__global__ void coarse_rows_kernel(CoarseGeom g, double const *Y, int32_t *row_ptr, int32_t *col, double *val)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < g.n_c; r += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t nd = r / g.C;
    const int Xl = (int)(nd % g.cdims[0]), Yl = (int)((nd / g.cdims[0]) % g.cdims[1]), Zl = (int)(nd / ((int64_t)g.cdims[0] * g.cdims[1]));
    const bool owned = Xl >= g.own0[0] && Xl < g.own1[0] && Yl >= g.own0[1] && Yl < g.own1[1] && Zl >= g.own0[2] && Zl < g.own1[2];
    int p = FILL ? row_ptr[r] : 0;
    if (owned)
    {
      const int X = Xl + g.c_g0[0], Yc = Yl + g.c_g0[1], Zg = Zl + g.c_g0[2];
      for (int K = max(0, Zg - g.reach); K <= min(g.gdims_c[2] - 1, Zg + g.reach); ++K)
        for (int J = max(0, Yc - g.reach); J <= min(g.gdims_c[1] - 1, Yc + g.reach); ++J)
          for (int I = max(0, X - g.reach); I <= min(g.gdims_c[0] - 1, X + g.reach); ++I)
          {
            const int oc = (I % g.period[0]) + g.period[0] * ((J % g.period[1]) + g.period[1] * (K % g.period[2]));
            const int Il = I - g.c_g0[0], Jl = J - g.c_g0[1], Kl = K - g.c_g0[2];
            for (int comp = 0; comp < g.C; ++comp)
            {
              double v = Y[(size_t)(oc * g.C + comp) * (size_t)g.n_c + (size_t)r];
              if (g.round_to_float)
              {
                if (g.all_owned)
                {
                  // symmetric to rounding: an entry and its transposed partner are replaced by their mean (see above)
                  const int ocr = (X % g.period[0]) + g.period[0] * ((Yc % g.period[1]) + g.period[1] * (Zg % g.period[2]));
                  const int64_t cr = (((int64_t)Kl * g.cdims[1] + Jl) * g.cdims[0] + Il) * g.C + comp;
                  const double vt = Y[(size_t)(ocr * g.C + (int)(r % g.C)) * (size_t)g.n_c + (size_t)cr];
                  v = 0.5 * (v + vt);
                }
                v = (double)(float)v;
              }
              if (v != 0.)
              {
                if (FILL)
                {
                  col[p] = (int32_t)((((int64_t)Kl * g.cdims[1] + Jl) * g.cdims[0] + Il) * g.C + comp);
                  val[p] = v;
                }
                ++p;
              }
            }
          }
    }
    if (!FILL)
      row_ptr[r + 1] = p;
  }
}

// ---- the assembled fine operator (tests/laplace.hpp:154-204 + AffineConstraints::distribute_local_to_global) -----------------
// One thread per DoF repeats amge_structured.cpp: operator_row -- the 3^dim stencil of the node from the cell matrices
// A_e[m][n] = sum_q c(cell, q) K[q][m][n] of the cells around it, in the same order of additions (contraction off: the
// same bits), Dirichlet rows and columns eliminated -- and writes its entries sorted by column.
struct FineGeom
{
  int dim, nc, n[3], N[3], matrix_free_semantics;
  int64_t n_dofs;
};

template <bool FILL>
__global__ void fine_rows_kernel(FineGeom g, int32_t const *node_dof, int32_t const *dof_node, uint8_t const *constrained,
                                 double const *coefficient, double const *Kq, int32_t *row_ptr, int32_t *col, double *val)
{
#pragma clang fp contract(off)
  for (int64_t gd = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gd < g.n_dofs; gd += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t nd = dof_node[gd];
    const int i = (int)(nd % g.N[0]), j = (int)((nd / g.N[0]) % g.N[1]), k = (int)(nd / ((int64_t)g.N[0] * g.N[1]));
    const bool con = constrained[gd] == 1;
    int32_t cols[27];
    double vals[27];
    int cnt = 0;
    if (con && g.matrix_free_semantics)
    {
      cols[0] = (int32_t)gd;
      vals[0] = 1.;
      cnt = 1;
    }
    else
    {
      double st[27];
      const int ns = (g.dim == 3) ? 27 : 9;
      for (int t = 0; t < ns; ++t)
        st[t] = 0.;
      for (int m = 0; m < g.nc; ++m)
      {
        const int a = m & 1, b = (m >> 1) & 1, d = (m >> 2) & 1;
        const int ci = i - a, cj = j - b, ck = (g.dim == 3) ? k - d : 0;
        if (ci < 0 || cj < 0 || ck < 0 || ci >= g.n[0] || cj >= g.n[1] || (g.dim == 3 && ck >= g.n[2]))
          continue;
        const int64_t c = ci + (int64_t)g.n[0] * (cj + (int64_t)g.n[1] * ck);
        double const *coef = coefficient + c * g.nc;
        for (int mp = 0; mp < g.nc; ++mp)
        {
          if (con && mp != m)
            continue;
          double const *kq = Kq + (size_t)m * g.nc + mp;
          double v = 0.;
          for (int q = 0; q < g.nc; ++q)
            v += coef[q] * kq[(size_t)q * g.nc * g.nc];
          const int da = (mp & 1) - a + 1, db = ((mp >> 1) & 1) - b + 1, dd = (g.dim == 3) ? ((mp >> 2) & 1) - d + 1 : 0;
          st[da + 3 * db + 9 * dd] += v;
        }
      }
      if (con)
      {
        cols[0] = (int32_t)gd;
        vals[0] = st[(g.dim == 3) ? 13 : 4];
        cnt = 1;
      }
      else
        for (int t = 0; t < ns; ++t)
        {
          const int ni = i + (t % 3) - 1, nj = j + ((t / 3) % 3) - 1, nk = (g.dim == 3) ? k + (t / 9) - 1 : 0;
          if (ni < 0 || nj < 0 || nk < 0 || ni >= g.N[0] || nj >= g.N[1] || (g.dim == 3 && nk >= g.N[2]))
            continue;
          const int32_t gp = node_dof[ni + (int64_t)g.N[0] * (nj + (int64_t)g.N[1] * nk)];
          if (constrained[gp] == 1)
            continue;
          cols[cnt] = gp;
          vals[cnt] = st[t];
          ++cnt;
        }
    }
    if (!FILL)
    {
      row_ptr[gd + 1] = cnt;
      continue;
    }
    // insertion sort by column (<= 27 entries), then the row
    for (int t = 1; t < cnt; ++t)
    {
      const int32_t cc = cols[t];
      const double vv = vals[t];
      int u = t - 1;
      while (u >= 0 && cols[u] > cc)
      {
        cols[u + 1] = cols[u];
        vals[u + 1] = vals[u];
        --u;
      }
      cols[u + 1] = cc;
      vals[u + 1] = vv;
    }
    const int p0 = row_ptr[gd];
    for (int t = 0; t < cnt; ++t)
    {
      col[p0 + t] = cols[t];
      val[p0 + t] = vals[t];
    }
  }
}

template <typename Geom, typename CountLaunch, typename FillLaunch>
std::shared_ptr<SparseMatrixDevice<double>> assemble(HipHandle &h, int64_t n_rows, int64_t n_cols, char const *what, CountLaunch &&count,
                                                     FillLaunch &&fill)
{
  MemoryKind kind("CSR arrays (val, col, row_ptr)");
  DeviceBuffer<int32_t> row_ptr((size_t)n_rows + 1);
  MFMG_HIP_CHECK(hipMemsetAsync(row_ptr.data(), 0, ((size_t)n_rows + 1) * sizeof(int32_t), h.stream));
  count(row_ptr.data());
  MFMG_HIP_CHECK(hipGetLastError());
  const int64_t nnz = scan_counts(h, row_ptr, n_rows, what);
  DeviceBuffer<int32_t> col((size_t)nnz);
  DeviceBuffer<double> val((size_t)nnz);
  if (nnz > 0)
  {
    fill(row_ptr.data(), col.data(), val.data());
    MFMG_HIP_CHECK(hipGetLastError());
  }
  return std::make_shared<SparseMatrixDevice<double>>(h, n_rows, n_cols, std::move(row_ptr), std::move(col), std::move(val));
}
} // namespace

std::shared_ptr<SparseMatrixDevice<double>> galerkin_from_probes(HipHandle &h, int const na[3], int ne, int const k[3], int const off[3],
                                                                 int64_t const own0[3], int64_t const own1[3], double const *Y)
{
  GalerkinGeom g;
  g.round_to_float = h.setup_values_float ? 1 : 0;
  g.all_owned = 1;
  for (int d = 0; d < 3; ++d)
  {
    g.na[d] = na[d];
    g.k[d] = k[d];
    g.off[d] = off[d];
    g.own0[d] = (int)own0[d];
    g.own1[d] = (int)own1[d];
    if (own0[d] != 0 || own1[d] != na[d])
      g.all_owned = 0;
  }
  g.ne = ne;
  g.nc = (int64_t)na[0] * na[1] * na[2] * ne;
  return assemble<GalerkinGeom>(
      h, g.nc, g.nc, "coarse operator",
      [&](int32_t *rp) { hipLaunchKernelGGL(galerkin_rows_kernel<false>, grid_for(g.nc), dim3(256), 0, h.stream, g, Y, rp, nullptr, nullptr); },
      [&](int32_t *rp, int32_t *col, double *val) {
        hipLaunchKernelGGL(galerkin_rows_kernel<true>, grid_for(g.nc), dim3(256), 0, h.stream, g, Y, rp, col, val);
      });
}

std::shared_ptr<SparseMatrixDevice<double>> prolongator_from_probes(HipHandle &h, HaloSpace const &fine, HaloSpace const &coarse, int blk,
                                                                    int reach, int const period[3], double w, double const *Z,
                                                                    double const *t, double const *dinv, double const *Yp)
{
  ProlongatorGeom g;
  g.round_to_float = h.setup_values_float ? 1 : 0;
  for (int d = 0; d < 3; ++d)
  {
    g.fdims[d] = (int)fine.dim(d);
    g.cdims[d] = (int)coarse.dim(d);
    g.gdims_c[d] = (int)coarse.gn(d);
    g.period[d] = period[d];
    g.f_g0[d] = (int)fine.g0(d);
    g.c_g0[d] = (int)coarse.g0(d);
    g.own0[d] = (int)fine.own0(d);
    g.own1[d] = (int)(fine.own0(d) + fine.own_n(d));
  }
  g.C = fine.comps;
  g.blk = blk;
  g.reach = reach;
  g.w = w;
  const int64_t n_f = fine.n_local(), n_c = coarse.n_local();
  g.n_f = n_f;
  return assemble<ProlongatorGeom>(
      h, n_f, n_c, "prolongator",
      [&](int32_t *rp) {
        hipLaunchKernelGGL(prolongator_rows_kernel<false>, grid_for(n_f), dim3(256), 0, h.stream, g, Z, t, dinv, rp, nullptr, nullptr, Yp);
      },
      [&](int32_t *rp, int32_t *col, double *val) {
        hipLaunchKernelGGL(prolongator_rows_kernel<true>, grid_for(n_f), dim3(256), 0, h.stream, g, Z, t, dinv, rp, col, val, Yp);
      });
}

std::shared_ptr<SparseMatrixDevice<double>> coarse_operator_from_probes(HipHandle &h, HaloSpace const &coarse, int reach, int const period[3],
                                                                        double const *Y)
{
  CoarseGeom g;
  g.round_to_float = h.setup_values_float ? 1 : 0;
  g.all_owned = 1;
  for (int d = 0; d < 3; ++d)
  {
    g.cdims[d] = (int)coarse.dim(d);
    g.gdims_c[d] = (int)coarse.gn(d);
    g.period[d] = period[d];
    g.c_g0[d] = (int)coarse.g0(d);
    g.own0[d] = (int)coarse.own0(d);
    g.own1[d] = (int)(coarse.own0(d) + coarse.own_n(d));
    if (coarse.g0(d) != 0 || coarse.own0(d) != 0 || coarse.own_n(d) != coarse.dim(d))
      g.all_owned = 0;
  }
  g.C = coarse.comps;
  g.reach = reach;
  const int64_t n_c = coarse.n_local();
  g.n_c = n_c;
  return assemble<CoarseGeom>(
      h, n_c, n_c, "coarse operator",
      [&](int32_t *rp) { hipLaunchKernelGGL(coarse_rows_kernel<false>, grid_for(n_c), dim3(256), 0, h.stream, g, Y, rp, nullptr, nullptr); },
      [&](int32_t *rp, int32_t *col, double *val) {
        hipLaunchKernelGGL(coarse_rows_kernel<true>, grid_for(n_c), dim3(256), 0, h.stream, g, Y, rp, col, val);
      });
}
std::shared_ptr<SparseMatrixDevice<double>> fine_operator_on_device(HipHandle &h, StructuredMesh const &mesh, bool matrix_free_semantics)
{
  FineGeom g;
  g.dim = mesh.dim;
  g.nc = mesh.nc();
  for (int d = 0; d < 3; ++d)
  {
    g.n[d] = mesh.n[d];
    g.N[d] = mesh.N[d];
  }
  g.matrix_free_semantics = matrix_free_semantics ? 1 : 0;
  g.n_dofs = mesh.n_dofs;
  std::vector<int32_t> dof_node((size_t)mesh.n_dofs);
#pragma omp parallel for schedule(static)
  for (int64_t nd = 0; nd < (int64_t)mesh.node_dof.size(); ++nd)
    dof_node[mesh.node_dof[nd]] = (int32_t)nd;
  const std::vector<double> Kq = reference_cell_tables(mesh.dim, mesh.h);
  DeviceBuffer<int32_t> d_node_dof, d_dof_node;
  DeviceBuffer<uint8_t> d_con;
  DeviceBuffer<double> d_coef, d_kq;
  d_node_dof.upload(mesh.node_dof.data(), mesh.node_dof.size(), h.stream);
  d_dof_node.upload(dof_node.data(), dof_node.size(), h.stream);
  d_con.upload(mesh.constrained.data(), mesh.constrained.size(), h.stream);
  d_coef.upload(mesh.coefficient.data(), mesh.coefficient.size(), h.stream);
  d_kq.upload(Kq.data(), Kq.size(), h.stream);
  return assemble<FineGeom>(
      h, mesh.n_dofs, mesh.n_dofs, "assembled matrix",
      [&](int32_t *rp) {
        hipLaunchKernelGGL(fine_rows_kernel<false>, grid_for(mesh.n_dofs), dim3(256), 0, h.stream, g, d_node_dof.data(), d_dof_node.data(),
                           d_con.data(), d_coef.data(), d_kq.data(), rp, nullptr, nullptr);
      },
      [&](int32_t *rp, int32_t *col, double *val) {
        hipLaunchKernelGGL(fine_rows_kernel<true>, grid_for(mesh.n_dofs), dim3(256), 0, h.stream, g, d_node_dof.data(), d_dof_node.data(),
                           d_con.data(), d_coef.data(), d_kq.data(), rp, col, val);
      });
}
} // namespace mfmg
