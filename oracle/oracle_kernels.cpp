// CPU oracle, native part -- TEST INFRASTRUCTURE ONLY (see oracle/mfmg_oracle.py).
//
// C++17/OpenMP restatement of the V-cycle apply path of ORNL-CEES/mfmg on a structured Q1
// hyper-cube, used (a) to cross-check the numpy oracle at sizes numpy cannot reach and (b) as the
// `cpu_baseline` ("port") leg of bench.py: the reference's own CPU path (source/dealii + deal.II +
// Trilinos) cannot be built here.  Nothing in mfmg_amd/ links or loads this file.
//
// Follows: matrix-free operator tests/laplace_matrix_free.hpp:121-156 (+ MatrixFreeOperators::Base
// constrained rows), smoother wrapper source/dealii/dealii_matrix_free_smoother.cc:63-76 with the
// deal.II Chebyshev three-term recurrence, V-cycle include/mfmg/common/hierarchy.hpp:246-309,
// CSR products source/dealii/dealii_trilinos_matrix_operator.cc:28-35.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
constexpr double GA = 0.78867513459481288225, GB = 0.21132486540518711775;

struct Mesh
{
  int n[3], N[3];
  double f[3];
  const int32_t *cell_dofs;
  const double *coef;
  const uint8_t *con;
  int64_t n_dofs;
};

inline void interp(double i0, double i1, double &o0, double &o1)
{
  o0 = GA * i0 + GB * i1;
  o1 = GB * i0 + GA * i1;
}

inline void direction(double d00, double d10, double d01, double d11, double c00, double c10, double c01, double c11,
                      double f, double X[4])
{
  double t00, t10, t01, t11, g00, g01, g10, g11, w00, w01, w10, w11;
  interp(d00, d10, t00, t10);
  interp(d01, d11, t01, t11);
  interp(t00, t01, g00, g01);
  interp(t10, t11, g10, g11);
  const double s00 = g00 * c00, s10 = g10 * c10, s01 = g01 * c01, s11 = g11 * c11;
  interp(s00, s01, w00, w01);
  interp(s10, s11, w10, w11);
  interp(w00, w10, X[0], X[1]);
  interp(w01, w11, X[2], X[3]);
  for (int t = 0; t < 4; ++t)
    X[t] *= f;
}

// FEEvaluation::evaluate(gradients) / submit_gradient(coef * grad) / integrate(gradients), sum-factorised
inline void cell_apply(const double u[8], const double c[8], const double f[3], double v[8])
{
  double X[4];
  direction(u[1] - u[0], u[3] - u[2], u[5] - u[4], u[7] - u[6], c[0] + c[1], c[2] + c[3], c[4] + c[5], c[6] + c[7],
            f[0], X);
  v[0] = -X[0]; v[1] = X[0]; v[2] = -X[1]; v[3] = X[1]; v[4] = -X[2]; v[5] = X[2]; v[6] = -X[3]; v[7] = X[3];
  direction(u[2] - u[0], u[3] - u[1], u[6] - u[4], u[7] - u[5], c[0] + c[2], c[1] + c[3], c[4] + c[6], c[5] + c[7],
            f[1], X);
  v[0] -= X[0]; v[2] += X[0]; v[1] -= X[1]; v[3] += X[1]; v[4] -= X[2]; v[6] += X[2]; v[5] -= X[3]; v[7] += X[3];
  direction(u[4] - u[0], u[5] - u[1], u[6] - u[2], u[7] - u[3], c[0] + c[4], c[1] + c[5], c[2] + c[6], c[3] + c[7],
            f[2], X);
  v[0] -= X[0]; v[4] += X[0]; v[1] -= X[1]; v[5] += X[1]; v[2] -= X[2]; v[6] += X[2]; v[3] -= X[3]; v[7] += X[3];
}

// y = A x : cell loop of LaplaceOperator::local_apply; cell layers of equal parity run concurrently
// (a layer only scatters into its two DoF planes), layers themselves run in order -> deterministic.
void mf_apply(const Mesh &m, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < m.n_dofs; ++g)
    y[g] = 0.;
  for (int parity = 0; parity < 2; ++parity)
  {
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = parity; k < m.n[2]; k += 2)
      for (int j = 0; j < m.n[1]; ++j)
        for (int i = 0; i < m.n[0]; ++i)
        {
          const int64_t c = i + (int64_t)m.n[0] * (j + (int64_t)m.n[1] * k);
          const int32_t *cd = m.cell_dofs + c * 8;
          double u[8], v[8];
          for (int q = 0; q < 8; ++q)
            u[q] = m.con[cd[q]] ? 0. : x[cd[q]]; // read_dof_values: constrained -> 0
          cell_apply(u, m.coef + c * 8, m.f, v);
          for (int q = 0; q < 8; ++q)
            if (!m.con[cd[q]])
              y[cd[q]] += v[q]; // distribute_local_to_global
        }
  }
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < m.n_dofs; ++g)
    if (m.con[g])
      y[g] = x[g]; // MatrixFreeOperators::Base::vmult: dst_c = src_c
}

// compute_diagonal (tests/laplace_matrix_free.hpp:75-98, local_compute_diagonal :158-199): per cell, the operator
// applied to each unit vector, entry i kept; summed into the global vector; constrained entries set to one
// (set_constrained_entries_to_one).  Same layer schedule as mf_apply.
void mf_diagonal(const Mesh &m, double *d)
{
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < m.n_dofs; ++g)
    d[g] = 0.;
  for (int parity = 0; parity < 2; ++parity)
  {
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = parity; k < m.n[2]; k += 2)
      for (int j = 0; j < m.n[1]; ++j)
        for (int i = 0; i < m.n[0]; ++i)
        {
          const int64_t c = i + (int64_t)m.n[0] * (j + (int64_t)m.n[1] * k);
          const int32_t *cd = m.cell_dofs + c * 8;
          for (int q = 0; q < 8; ++q)
          {
            double u[8] = {0., 0., 0., 0., 0., 0., 0., 0.}, v[8];
            u[q] = 1.;
            cell_apply(u, m.coef + c * 8, m.f, v);
            d[cd[q]] += v[q];
          }
        }
  }
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < m.n_dofs; ++g)
    if (m.con[g])
      d[g] = 1.;
}

void csr_spmv(int64_t n_rows, const int32_t *rp, const int32_t *col, const double *val, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n_rows; ++r)
  {
    double s = 0.;
    for (int p = rp[r]; p < rp[r + 1]; ++p)
      s += val[p] * x[col[p]];
    y[r] = s;
  }
}

double dot(int64_t n, const double *a, const double *b)
{
  double s = 0.;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i)
    s += a[i] * b[i];
  return s;
}

struct Csr
{
  int64_t n_rows;
  const int32_t *rp, *col;
  const double *val;
};

// x <- x - B^{-1}(A x - b), B^{-1} = Chebyshev polynomial in D^{-1} A, deal.II update1/update2 form
void chebyshev_smoother(const Mesh &m, const double *dinv, int degree, double lmin, double lmax, const double *b,
                        double *x, std::vector<double> &r, std::vector<double> &dst, std::vector<double> &up1,
                        std::vector<double> &up2)
{
  const int64_t n = m.n_dofs;
  mf_apply(m, x, r.data());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    r[i] -= b[i]; // r = A x - b
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
  {
    up1[i] = dinv[i] * r[i] / theta;
    dst[i] = up1[i];
  }
  if (degree >= 2 && std::abs(delta) >= 1e-40)
  {
    double rhok = delta / theta;
    const double sigma = theta / delta;
    for (int k = 0; k < degree - 1; ++k)
    {
      mf_apply(m, dst.data(), up2.data());
      const double rhokp = 1. / (2. * sigma - rhok);
      const double f1 = rhokp * rhok, f2 = 2. * rhokp / delta;
      rhok = rhokp;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i)
      {
        up1[i] = f1 * up1[i] - f2 * dinv[i] * (up2[i] - r[i]);
        dst[i] += up1[i];
      }
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    x[i] -= dst[i];
}
// x <- x - B^{-1}(A x - b) on a CSR level, same polynomial as chebyshev_smoother
void csr_chebyshev_smoother(const Csr &A, const double *dinv, int degree, double lmin, double lmax, const double *b,
                            double *x, std::vector<double> &r, std::vector<double> &dst, std::vector<double> &up1,
                            std::vector<double> &up2)
{
  const int64_t n = A.n_rows;
  csr_spmv(n, A.rp, A.col, A.val, x, r.data());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    r[i] -= b[i];
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
  {
    up1[i] = dinv[i] * r[i] / theta;
    dst[i] = up1[i];
  }
  if (degree >= 2 && std::abs(delta) >= 1e-40)
  {
    double rhok = delta / theta;
    const double sigma = theta / delta;
    for (int k = 0; k < degree - 1; ++k)
    {
      csr_spmv(n, A.rp, A.col, A.val, dst.data(), up2.data());
      const double rhokp = 1. / (2. * sigma - rhok);
      const double f1 = rhokp * rhok, f2 = 2. * rhokp / delta;
      rhok = rhokp;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i)
      {
        up1[i] = f1 * up1[i] - f2 * dinv[i] * (up2[i] - r[i]);
        dst[i] += up1[i];
      }
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    x[i] -= dst[i];
}

struct AmgLevelData
{
  Csr A, P, Pt;
  int degree;
  double lmin, lmax;
  std::vector<double> dinv, r, dst, up1, up2, res, bc, xc;
  std::vector<double> lu; // dense, row-major (last level)
  std::vector<int> perm;
};

// (pre_levels: levels from that index on skip the pre-smoother: from the zero guess the residual is -b, so b is
// restricted and the correction added -- a V(0,1) cycle; the product's solver.amg.pre_smoothing_levels)
void amg_cycle(std::vector<AmgLevelData> &L, size_t l, const double *b, double *x, int pre_levels)
{
  AmgLevelData &lv = L[l];
  const int64_t n = lv.A.n_rows;
  if (l + 1 == L.size())
  {
    // dense LU solve (getrs)
    std::vector<double> y(n);
    for (int64_t i = 0; i < n; ++i)
      y[i] = b[lv.perm[i]];
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = 0; j < i; ++j)
        y[i] -= lv.lu[i * n + j] * y[j];
    for (int64_t i = n - 1; i >= 0; --i)
    {
      for (int64_t j = i + 1; j < n; ++j)
        y[i] -= lv.lu[i * n + j] * y[j];
      y[i] /= lv.lu[i * n + i];
    }
    std::copy(y.begin(), y.end(), x);
    return;
  }
  const int64_t nc = lv.Pt.n_rows;
  if ((int)l >= pre_levels)
  {
    csr_spmv(nc, lv.Pt.rp, lv.Pt.col, lv.Pt.val, b, lv.bc.data());
    std::fill(lv.xc.begin(), lv.xc.end(), 0.);
    amg_cycle(L, l + 1, lv.bc.data(), lv.xc.data(), pre_levels);
    csr_spmv(n, lv.P.rp, lv.P.col, lv.P.val, lv.xc.data(), x);
    csr_chebyshev_smoother(lv.A, lv.dinv.data(), lv.degree, lv.lmin, lv.lmax, b, x, lv.r, lv.dst, lv.up1, lv.up2);
    return;
  }
  csr_chebyshev_smoother(lv.A, lv.dinv.data(), lv.degree, lv.lmin, lv.lmax, b, x, lv.r, lv.dst, lv.up1, lv.up2);
  csr_spmv(n, lv.A.rp, lv.A.col, lv.A.val, x, lv.res.data());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    lv.res[i] -= b[i];
  csr_spmv(nc, lv.Pt.rp, lv.Pt.col, lv.Pt.val, lv.res.data(), lv.bc.data());
  std::fill(lv.xc.begin(), lv.xc.end(), 0.);
  amg_cycle(L, l + 1, lv.bc.data(), lv.xc.data(), pre_levels);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
  {
    double sum = 0.;
    for (int p = lv.P.rp[i]; p < lv.P.rp[i + 1]; ++p)
      sum += lv.P.val[p] * lv.xc[lv.P.col[p]];
    x[i] -= sum;
  }
  csr_chebyshev_smoother(lv.A, lv.dinv.data(), lv.degree, lv.lmin, lv.lmax, b, x, lv.r, lv.dst, lv.up1, lv.up2);
}
} // namespace

extern "C" {

// level descriptor passed from Python (ctypes.Structure with the same layout)
struct OracleCsr
{
  int64_t n_rows, n_cols;
  const int32_t *rp, *col;
  const double *val;
};
struct OracleAmgLevel
{
  OracleCsr A, P, Pt;
  int32_t degree;
  double lmin, lmax;
};

int oracle_num_threads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int t)
{
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}

static Mesh make_mesh(const int *n, const double *h, const int32_t *cell_dofs, const double *coef, const uint8_t *con)
{
  Mesh m;
  const double vol = h[0] * h[1] * h[2];
  m.n_dofs = 1;
  for (int d = 0; d < 3; ++d)
  {
    m.n[d] = n[d];
    m.N[d] = n[d] + 1;
    m.f[d] = vol / 8. / (h[d] * h[d]);
    m.n_dofs *= m.N[d];
  }
  m.cell_dofs = cell_dofs;
  m.coef = coef;
  m.con = con;
  return m;
}

void oracle_mf_apply(const int *n, const double *h, const int32_t *cell_dofs, const double *coef,
                     const uint8_t *con, const double *x, double *y)
{
  Mesh m = make_mesh(n, h, cell_dofs, coef, con);
  mf_apply(m, x, y);
}

void oracle_mf_diagonal(const int *n, const double *h, const int32_t *cell_dofs, const double *coef,
                        const uint8_t *con, double *d)
{
  Mesh m = make_mesh(n, h, cell_dofs, coef, con);
  mf_diagonal(m, d);
}

void oracle_csr_spmv(int64_t n_rows, const int32_t *rp, const int32_t *col, const double *val, const double *x,
                     double *y)
{
  csr_spmv(n_rows, rp, col, val, x, y);
}

// n_cycles V-cycles of Hierarchy::apply (is_preconditioner = false, one pre / post smoothing step):
// matrix-free fine operator, Chebyshev smoother, CSR restrictor (R and its explicit transpose), coarse
// "solve" = exactly `coarse_iters` Jacobi-PCG steps on the CSR coarse operator from a zero guess.
// Writes ||b - A x|| / ||b - A x0|| after every cycle into history[0..n_cycles].
void oracle_vcycles(const int *n, const double *h, const int32_t *cell_dofs, const double *coef, const uint8_t *con,
                    const double *dinv, int degree, double lmin, double lmax, int64_t n_coarse, const int32_t *r_rp,
                    const int32_t *r_col, const double *r_val, const int32_t *rt_rp, const int32_t *rt_col,
                    const double *rt_val, const int32_t *ac_rp, const int32_t *ac_col, const double *ac_val,
                    int coarse_iters, int n_amg_levels, const OracleAmgLevel *amg_levels, const double *b, double *x,
                    int n_cycles, double *history, int amg_pre_smoothing_levels)
{
  Mesh m = make_mesh(n, h, cell_dofs, coef, con);
  const int64_t nf = m.n_dofs, ncs = n_coarse;
  std::vector<double> r(nf), dst(nf), up1(nf), up2(nf), res(nf), bc(ncs), xc(ncs), cr(ncs), cz(ncs), cp(ncs),
      cap(ncs), cdinv(ncs);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < ncs; ++i)
  {
    double d = 1.;
    for (int p = ac_rp[i]; p < ac_rp[i + 1]; ++p)
      if (ac_col[p] == i)
        d = ac_val[p];
    cdinv[i] = 1. / d;
  }
  // multilevel coarse solver (n_amg_levels > 0): level data + dense LU of the last level
  std::vector<AmgLevelData> amg(n_amg_levels);
  for (int l = 0; l < n_amg_levels; ++l)
  {
    AmgLevelData &lv = amg[l];
    auto conv = [](OracleCsr const &c) { return Csr{c.n_rows, c.rp, c.col, c.val}; };
    lv.A = conv(amg_levels[l].A);
    lv.P = conv(amg_levels[l].P);
    lv.Pt = conv(amg_levels[l].Pt);
    lv.P.n_rows = amg_levels[l].P.n_rows;
    lv.degree = amg_levels[l].degree;
    lv.lmin = amg_levels[l].lmin;
    lv.lmax = amg_levels[l].lmax;
    const int64_t n = lv.A.n_rows;
    if (l + 1 < n_amg_levels)
    {
      lv.dinv.assign(n, 1.);
      for (int64_t i = 0; i < n; ++i)
        for (int p = lv.A.rp[i]; p < lv.A.rp[i + 1]; ++p)
          if (lv.A.col[p] == i)
            lv.dinv[i] = 1. / lv.A.val[p];
      for (auto *v : {&lv.r, &lv.dst, &lv.up1, &lv.up2, &lv.res})
        v->assign(n, 0.);
      lv.bc.assign(amg_levels[l].P.n_cols, 0.);
      lv.xc.assign(amg_levels[l].P.n_cols, 0.);
    }
    else
    {
      lv.lu.assign((size_t)n * n, 0.);
      for (int64_t i = 0; i < n; ++i)
        for (int p = lv.A.rp[i]; p < lv.A.rp[i + 1]; ++p)
          lv.lu[i * n + lv.A.col[p]] += lv.A.val[p];
      lv.perm.resize(n);
      for (int64_t i = 0; i < n; ++i)
        lv.perm[i] = (int)i;
      for (int64_t c = 0; c < n; ++c)
      {
        int64_t piv = c;
        for (int64_t r2 = c + 1; r2 < n; ++r2)
          if (std::abs(lv.lu[r2 * n + c]) > std::abs(lv.lu[piv * n + c]))
            piv = r2;
        if (piv != c)
        {
          for (int64_t cc = 0; cc < n; ++cc)
            std::swap(lv.lu[piv * n + cc], lv.lu[c * n + cc]);
          std::swap(lv.perm[piv], lv.perm[c]);
        }
        const double d = 1. / lv.lu[c * n + c];
#pragma omp parallel for schedule(static) if (n - c > 256)
        for (int64_t r2 = c + 1; r2 < n; ++r2)
        {
          const double f = lv.lu[r2 * n + c] * d;
          lv.lu[r2 * n + c] = f;
          if (f != 0.)
            for (int64_t cc = c + 1; cc < n; ++cc)
              lv.lu[r2 * n + cc] -= f * lv.lu[c * n + cc];
        }
      }
    }
  }
  auto resnorm = [&]() {
    mf_apply(m, x, res.data());
    double s = 0.;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < nf; ++i)
    {
      const double d = b[i] - res[i];
      s += d * d;
    }
    return std::sqrt(s);
  };
  double r0 = 1.;
  if (history)
  {
    r0 = resnorm();
    history[0] = 1.;
  }
  for (int cyc = 0; cyc < n_cycles; ++cyc)
  {
    chebyshev_smoother(m, dinv, degree, lmin, lmax, b, x, r, dst, up1, up2);
    mf_apply(m, x, res.data());
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nf; ++i)
      res[i] -= b[i]; // negative residual
    csr_spmv(ncs, r_rp, r_col, r_val, res.data(), bc.data());
    std::fill(xc.begin(), xc.end(), 0.);
    if (n_amg_levels > 0)
      amg_cycle(amg, 0, bc.data(), xc.data(), amg_pre_smoothing_levels);
    // coarse PCG from zero (coarse_iters = 0 when the multilevel solver is used)
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ncs; ++i)
    {
      cr[i] = bc[i];
      cz[i] = cdinv[i] * cr[i];
      cp[i] = cz[i];
    }
    double rz = dot(ncs, cr.data(), cz.data());
    for (int it = 0; it < coarse_iters; ++it)
    {
      csr_spmv(ncs, ac_rp, ac_col, ac_val, cp.data(), cap.data());
      const double pap = dot(ncs, cp.data(), cap.data());
      const double alpha = pap != 0. ? rz / pap : 0.;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < ncs; ++i)
      {
        xc[i] += alpha * cp[i];
        cr[i] -= alpha * cap[i];
        cz[i] = cdinv[i] * cr[i];
      }
      const double rz_new = dot(ncs, cr.data(), cz.data());
      const double beta = rz != 0. ? rz_new / rz : 0.;
      rz = rz_new;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < ncs; ++i)
        cp[i] = cz[i] + beta * cp[i];
    }
    // x -= R^T x_c
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < nf; ++g)
    {
      double s = 0.;
      for (int p = rt_rp[g]; p < rt_rp[g + 1]; ++p)
        s += rt_val[p] * xc[rt_col[p]];
      x[g] -= s;
    }
    chebyshev_smoother(m, dinv, degree, lmin, lmax, b, x, r, dst, up1, up2);
    if (history)
      history[cyc + 1] = resnorm() / r0;
  }
}
}
