#include "amge_structured.hpp"
#include "amge_device.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <numeric>
#include <sched.h>
#include <thread>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace mfmg
{
namespace
{
constexpr double kG0 = 0.21132486540518711775; // (1 - 1/sqrt(3))/2
constexpr double kG1 = 0.78867513459481288225;
} // namespace

int effective_cpu_count()
{
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0)
    n = CPU_COUNT(&set);
  // cgroup v2: "<quota> <period>" or "max <period>"
  std::ifstream f("/sys/fs/cgroup/cpu.max");
  std::string quota;
  double period = 0.;
  if (f && (f >> quota >> period) && quota != "max" && period > 0.)
  {
    const int q = (int)std::ceil(std::stod(quota) / period);
    if (q >= 1)
      n = std::min(n, q);
  }
  else
  {
    std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    double q = -1., p = 0.;
    if (fq && fp && (fq >> q) && (fp >> p) && q > 0. && p > 0.)
      n = std::min(n, (int)std::ceil(q / p));
  }
  return std::max(n, 1);
}

void configure_host_threads()
{
  static std::once_flag once;
  std::call_once(once, [] {
#ifdef _OPENMP
    // MFMG_HOST_THREADS wins; under torchrun (LOCAL_WORLD_SIZE set) the host cores are split between the
    // local ranks -- torchrun's blanket OMP_NUM_THREADS=1 would make the setup single-threaded; otherwise a
    // user-set OMP_NUM_THREADS is respected
    char const *forced = std::getenv("MFMG_HOST_THREADS");
    char const *lws = std::getenv("LOCAL_WORLD_SIZE");
    if (forced)
      omp_set_num_threads(std::max(1, std::atoi(forced)));
    else if (lws)
      omp_set_num_threads(std::max(1, effective_cpu_count() / std::max(1, std::atoi(lws))));
    else if (std::getenv("OMP_NUM_THREADS") == nullptr)
      omp_set_num_threads(effective_cpu_count());
#endif
  });
}

double MinstdUniform::next()
{
  // std::minstd_rand0 + generate_canonical<double,53> (two draws, range 2^31-2)
  auto draw = [&]() {
    state = (16807ull * state) % 2147483647ull;
    return (double)(state - 1);
  };
  const double R = 2147483646.0;
  double s = draw();
  s += draw() * R;
  double r = s / (R * R);
  if (r >= 1.0)
    r = std::nextafter(1.0, 0.0);
  return r;
}

StructuredMesh StructuredMesh::from_desc(mfmg_hip_mesh_desc const &desc, hipStream_t stream)
{
  StructuredMesh m;
  ASSERT_THROW(desc.dim == 2 || desc.dim == 3, "mesh dimension must be 2 or 3");
  m.dim = desc.dim;
  m.n_cells = 1;
  m.n_dofs = 1;
  for (int d = 0; d < 3; ++d)
  {
    if (d < m.dim)
    {
      ASSERT_THROW(desc.n_cells[d] >= 1, "n_cells must be positive");
      ASSERT_THROW(desc.cell_size[d] > 0., "cell_size must be positive");
      m.n[d] = desc.n_cells[d];
      m.N[d] = m.n[d] + 1;
      m.h[d] = desc.cell_size[d];
    }
    else
    {
      m.n[d] = 1; // a single "layer" so that loops stay uniform; N = 1
      m.N[d] = 1;
      m.h[d] = 1.;
    }
    if (d < m.dim)
    {
      m.n_cells *= m.n[d];
      m.n_dofs *= m.N[d];
    }
  }
  ASSERT_THROW(m.n_dofs == desc.n_dofs, "n_dofs does not match the cell grid (Q1: prod(n_cells+1))");
  ASSERT_THROW(desc.cell_dofs && desc.coefficient && desc.constrained, "mesh description arrays must not be null");
  const size_t nc = (size_t)1 << m.dim;
  m.cell_dofs.resize(m.n_cells * nc);
  m.coefficient.resize(m.n_cells * nc);
  m.constrained.resize(m.n_dofs);
  if (desc.arrays_on_device)
  {
    MFMG_HIP_CHECK(hipMemcpyAsync(m.cell_dofs.data(), desc.cell_dofs, m.cell_dofs.size() * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, stream));
    MFMG_HIP_CHECK(hipMemcpyAsync(m.coefficient.data(), desc.coefficient, m.coefficient.size() * sizeof(double),
                                  hipMemcpyDeviceToHost, stream));
    MFMG_HIP_CHECK(hipMemcpyAsync(m.constrained.data(), desc.constrained, m.constrained.size(),
                                  hipMemcpyDeviceToHost, stream));
    MFMG_HIP_CHECK(hipStreamSynchronize(stream));
  }
  else
  {
    std::memcpy(m.cell_dofs.data(), desc.cell_dofs, m.cell_dofs.size() * sizeof(int32_t));
    std::memcpy(m.coefficient.data(), desc.coefficient, m.coefficient.size() * sizeof(double));
    std::memcpy(m.constrained.data(), desc.constrained, m.constrained.size());
  }
  m.build_node_map();
  return m;
}

void StructuredMesh::build_node_map()
{
  const int ncorn = nc();
  const int nz_cells = (dim == 3) ? n[2] : 1;
  node_dof.assign((size_t)N[0] * N[1] * N[2], -1);
  int64_t n_bad = 0;
  // (cell layers of one parity touch disjoint node layers: two parallel sweeps instead of 134 M serial iterations)
  for (int parity = 0; parity < 2; ++parity)
  {
#pragma omp parallel for schedule(static) reduction(+ : n_bad)
    for (int k = parity; k < nz_cells; k += 2)
      for (int j = 0; j < n[1]; ++j)
        for (int i = 0; i < n[0]; ++i)
        {
          const int64_t c = cell_index(i, j, k);
          for (int m = 0; m < ncorn; ++m)
          {
            const int64_t nd = node_index(i + (m & 1), j + ((m >> 1) & 1), k + ((m >> 2) & 1));
            const int32_t g = cell_dofs[c * ncorn + m];
            if (g < 0 || g >= n_dofs)
              ++n_bad;
            else if (node_dof[nd] == -1)
              node_dof[nd] = g;
            else if (node_dof[nd] != g)
              ++n_bad;
          }
        }
  }
  std::vector<uint8_t> seen(n_dofs, 0);
  for (auto g : node_dof)
  {
    if (g < 0 || seen[g])
      ++n_bad;
    else
      seen[g] = 1;
  }
  ASSERT_THROW(n_bad == 0, "cell_dofs is not a logically structured Q1 mesh in lexicographic cell order (" +
                               std::to_string(n_bad) + " inconsistencies)");
}

std::vector<double> reference_cell_tables(int dim, double const h[3])
{
  const int nc = 1 << dim;
  double vol = 1.;
  for (int d = 0; d < dim; ++d)
    vol *= h[d];
  const double w = vol / nc;
  const double gp[2] = {kG0, kG1};
  std::vector<double> K((size_t)nc * nc * nc, 0.);
  std::vector<double> G((size_t)dim * nc);
  for (int q = 0; q < nc; ++q)
  {
    for (int m = 0; m < nc; ++m)
      for (int d = 0; d < dim; ++d)
      {
        double g = 1.;
        for (int e = 0; e < dim; ++e)
        {
          const int bit = (m >> e) & 1;
          const double xi = gp[(q >> e) & 1];
          if (e == d)
            g *= bit ? 1. : -1.;
          else
            g *= bit ? xi : (1. - xi);
        }
        G[d * nc + m] = g;
      }
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < nc; ++j)
      {
        double s = 0.;
        for (int d = 0; d < dim; ++d)
          s += w / (h[d] * h[d]) * G[d * nc + i] * G[d * nc + j];
        K[((size_t)q * nc + i) * nc + j] = s;
      }
  }
  return K;
}

void operator_row(StructuredMesh const &mesh, std::vector<double> const &Kq, ConstraintSemantics sem, int i,
                  int j, int k, std::vector<int32_t> &cols, std::vector<double> &vals)
{
  cols.clear();
  vals.clear();
  const int nc = mesh.nc();
  const int dim = mesh.dim;
  const int32_t g = mesh.node_dof[mesh.node_index(i, j, k)];
  const bool con = mesh.constrained[g] == 1;
  if (con && sem == ConstraintSemantics::matrix_free)
  {
    cols.push_back(g);
    vals.push_back(1.);
    return;
  }
  // contributions are gathered in a 3^dim stencil indexed by the offset of the neighbouring node
  double st[27];
  const int ns = (dim == 3) ? 27 : 9;
  for (int t = 0; t < ns; ++t)
    st[t] = 0.;
  for (int m = 0; m < nc; ++m)
  {
    const int a = m & 1, b = (m >> 1) & 1, d = (m >> 2) & 1;
    const int ci = i - a, cj = j - b, ck = (dim == 3) ? k - d : 0;
    if (ci < 0 || cj < 0 || ck < 0 || ci >= mesh.n[0] || cj >= mesh.n[1] || (dim == 3 && ck >= mesh.n[2]))
      continue;
    const int64_t c = mesh.cell_index(ci, cj, ck);
    double const *coef = &mesh.coefficient[c * nc];
    for (int mp = 0; mp < nc; ++mp)
    {
      if (con && mp != m)
        continue;
      double const *kq = &Kq[(size_t)m * nc + mp];
      double v = 0.;
      for (int q = 0; q < nc; ++q)
        v += coef[q] * kq[(size_t)q * nc * nc];
      const int da = (mp & 1) - a + 1, db = ((mp >> 1) & 1) - b + 1, dd = (dim == 3) ? ((mp >> 2) & 1) - d + 1 : 0;
      st[da + 3 * db + 9 * dd] += v;
    }
  }
  if (con)
  {
    cols.push_back(g);
    vals.push_back(st[(dim == 3) ? 13 : 4]);
    return;
  }
  for (int t = 0; t < ns; ++t)
  {
    const int ni = i + (t % 3) - 1, nj = j + ((t / 3) % 3) - 1, nk = (dim == 3) ? k + (t / 9) - 1 : 0;
    if (ni < 0 || nj < 0 || nk < 0 || ni >= mesh.N[0] || nj >= mesh.N[1] || (dim == 3 && nk >= mesh.N[2]))
      continue;
    const int32_t gp = mesh.node_dof[mesh.node_index(ni, nj, nk)];
    if (mesh.constrained[gp] == 1)
      continue;
    cols.push_back(gp);
    vals.push_back(st[t]);
  }
}

namespace
{
std::vector<int64_t> dof_to_node(StructuredMesh const &mesh)
{
  std::vector<int64_t> dn(mesh.n_dofs);
  for (size_t nd = 0; nd < mesh.node_dof.size(); ++nd)
    dn[mesh.node_dof[nd]] = (int64_t)nd;
  return dn;
}
inline void node_ijk(StructuredMesh const &mesh, int64_t nd, int &i, int &j, int &k)
{
  i = (int)(nd % mesh.N[0]);
  j = (int)((nd / mesh.N[0]) % mesh.N[1]);
  k = (int)(nd / ((int64_t)mesh.N[0] * mesh.N[1]));
}
} // namespace

HostCsr assemble_global_matrix(StructuredMesh const &mesh, ConstraintSemantics sem)
{
  configure_host_threads();
  const auto Kq = reference_cell_tables(mesh.dim, mesh.h);
  const auto dn = dof_to_node(mesh);
  HostCsr A;
  A.n_rows = A.n_cols = mesh.n_dofs;
  A.row_ptr.assign(mesh.n_dofs + 1, 0);
  std::vector<int32_t> counts(mesh.n_dofs);
#pragma omp parallel
  {
    std::vector<int32_t> cols;
    std::vector<double> vals;
#pragma omp for schedule(static)
    for (int64_t g = 0; g < mesh.n_dofs; ++g)
    {
      int i, j, k;
      node_ijk(mesh, dn[g], i, j, k);
      operator_row(mesh, Kq, sem, i, j, k, cols, vals);
      counts[g] = (int32_t)cols.size();
    }
  }
  int64_t total = 0;
  for (int64_t g = 0; g < mesh.n_dofs; ++g)
  {
    total += counts[g];
    ASSERT_THROW(total < (int64_t(1) << 31), "assembled matrix exceeds int32 nnz");
    A.row_ptr[g + 1] = (int32_t)total;
  }
  A.col.resize(total);
  A.val.resize(total);
#pragma omp parallel
  {
    std::vector<int32_t> cols;
    std::vector<double> vals;
    std::vector<int> perm;
#pragma omp for schedule(static)
    for (int64_t g = 0; g < mesh.n_dofs; ++g)
    {
      int i, j, k;
      node_ijk(mesh, dn[g], i, j, k);
      operator_row(mesh, Kq, sem, i, j, k, cols, vals);
      perm.resize(cols.size());
      std::iota(perm.begin(), perm.end(), 0);
      std::sort(perm.begin(), perm.end(), [&](int a, int b) { return cols[a] < cols[b]; });
      for (size_t t = 0; t < perm.size(); ++t)
      {
        A.col[A.row_ptr[g] + t] = cols[perm[t]];
        A.val[A.row_ptr[g] + t] = vals[perm[t]];
      }
    }
  }
  return A;
}

std::vector<double> operator_diagonal(StructuredMesh const &mesh, ConstraintSemantics sem)
{
  const auto Kq = reference_cell_tables(mesh.dim, mesh.h);
  const int nc = mesh.nc();
  std::vector<double> diag(mesh.n_dofs, 0.);
  const int nzc = (mesh.dim == 3) ? mesh.n[2] : 1;
  // every node gathers the contributions of its (up to 2^dim) cells in the order a serial scatter over the cells in
  // (k, j, i) order would add them: the same bits, on all threads
  const int Nz = (mesh.dim == 3) ? mesh.N[2] : 1;
#pragma omp parallel for schedule(static)
  for (int K = 0; K < Nz; ++K)
    for (int J = 0; J < mesh.N[1]; ++J)
      for (int I = 0; I < mesh.N[0]; ++I)
      {
        double sum = 0.;
        for (int k = std::max(K - 1, 0); k <= std::min(K, nzc - 1); ++k)
          for (int j = std::max(J - 1, 0); j <= std::min(J, mesh.n[1] - 1); ++j)
            for (int i = std::max(I - 1, 0); i <= std::min(I, mesh.n[0] - 1); ++i)
            {
              const int64_t c = mesh.cell_index(i, j, k);
              const int m = (I - i) + 2 * (J - j) + ((mesh.dim == 3) ? 4 * (K - k) : 0);
              double v = 0.;
              for (int q = 0; q < nc; ++q)
                v += mesh.coefficient[c * nc + q] * Kq[((size_t)q * nc + m) * nc + m];
              sum += v;
            }
        diag[mesh.node_dof[mesh.node_index(I, J, K)]] = sum;
      }
  if (sem == ConstraintSemantics::matrix_free)
    for (int64_t g = 0; g < mesh.n_dofs; ++g)
      if (mesh.constrained[g] == 1)
        diag[g] = 1.;
  return diag;
}

void symmetric_eigen(int n, std::vector<double> &A, std::vector<double> &w, std::vector<double> &V)
{
  V.assign((size_t)n * n, 0.);
  for (int i = 0; i < n; ++i)
    V[(size_t)i * n + i] = 1.; // V[col*n + row]
  auto a = [&](int r, int c) -> double & { return A[(size_t)r * n + c]; };
  for (int sweep = 0; sweep < 64; ++sweep)
  {
    double off = 0., dsum = 0.;
    for (int p = 0; p < n; ++p)
    {
      dsum += a(p, p) * a(p, p);
      for (int q = p + 1; q < n; ++q)
        off += a(p, q) * a(p, q);
    }
    if (off <= 1e-32 * (dsum + off) || off == 0.)
      break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        const double apq = a(p, q);
        if (std::abs(apq) < 1e-300)
          continue;
        const double theta = (a(q, q) - a(p, p)) / (2. * apq);
        const double t = (theta >= 0. ? 1. : -1.) / (std::abs(theta) + std::sqrt(theta * theta + 1.));
        const double c = 1. / std::sqrt(t * t + 1.);
        const double s = t * c;
        for (int r = 0; r < n; ++r)
        {
          const double arp = a(r, p), arq = a(r, q);
          a(r, p) = c * arp - s * arq;
          a(r, q) = s * arp + c * arq;
        }
        for (int r = 0; r < n; ++r)
        {
          const double apr = a(p, r), aqr = a(q, r);
          a(p, r) = c * apr - s * aqr;
          a(q, r) = s * apr + c * aqr;
        }
        for (int r = 0; r < n; ++r)
        {
          const double vrp = V[(size_t)p * n + r], vrq = V[(size_t)q * n + r];
          V[(size_t)p * n + r] = c * vrp - s * vrq;
          V[(size_t)q * n + r] = s * vrp + c * vrq;
        }
      }
  }
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return a(x, x) < a(y, y); });
  w.resize(n);
  std::vector<double> Vs((size_t)n * n);
  for (int c = 0; c < n; ++c)
  {
    w[c] = a(perm[c], perm[c]);
    std::copy(V.begin() + (size_t)perm[c] * n, V.begin() + (size_t)(perm[c] + 1) * n, Vs.begin() + (size_t)c * n);
  }
  V.swap(Vs);
}

void dense_lu_factor(int n, std::vector<double> &A, std::vector<int32_t> &perm)
{
  perm.resize(n);
  std::iota(perm.begin(), perm.end(), 0);
  // right-looking LU on the row-major copy
  for (int col = 0; col < n; ++col)
  {
    int piv = col;
    double best = std::abs(A[(size_t)col * n + col]);
    for (int r = col + 1; r < n; ++r)
      if (std::abs(A[(size_t)r * n + col]) > best)
      {
        best = std::abs(A[(size_t)r * n + col]);
        piv = r;
      }
    ASSERT_THROW(best > 0., "singular matrix in the dense coarse solver");
    if (piv != col)
    {
      for (int c = 0; c < n; ++c)
        std::swap(A[(size_t)piv * n + c], A[(size_t)col * n + c]);
      std::swap(perm[piv], perm[col]);
    }
    const double d = 1. / A[(size_t)col * n + col];
#pragma omp parallel for schedule(static) if (n - col > 256)
    for (int r = col + 1; r < n; ++r)
    {
      const double f = A[(size_t)r * n + col] * d;
      A[(size_t)r * n + col] = f;
      if (f == 0.)
        continue;
      for (int c = col + 1; c < n; ++c)
        A[(size_t)r * n + c] -= f * A[(size_t)col * n + c];
    }
  }
  // to column-major for coalesced column sweeps in the device triangular solves
  std::vector<double> cm((size_t)n * n);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c)
      cm[(size_t)c * n + r] = A[(size_t)r * n + c];
  A.swap(cm);
}

void dense_triangular_inverses(int n, std::vector<double> &lu)
{
  // column-major packed L\\U -> L^{-1} (unit diagonal implied) below, U^{-1} on and above the diagonal;
  // every column of an inverse is an independent substitution
  std::vector<double> out((size_t)n * n, 0.);
  auto at = [n](std::vector<double> const &m, int r, int c) { return m[(size_t)c * n + r]; };
#pragma omp parallel for schedule(dynamic, 8)
  for (int j = 0; j < n; ++j)
  {
    std::vector<double> x(n, 0.);
    // L x = e_j
    x[j] = 1.;
    for (int i = j + 1; i < n; ++i)
    {
      double s = 0.;
      for (int k = j; k < i; ++k)
        s += at(lu, i, k) * x[k];
      x[i] = -s;
      out[(size_t)j * n + i] = x[i];
    }
    // U x = e_j
    std::fill(x.begin(), x.end(), 0.);
    for (int i = j; i >= 0; --i)
    {
      double s = (i == j) ? 1. : 0.;
      for (int k = i + 1; k <= j; ++k)
        s -= at(lu, i, k) * x[k];
      x[i] = s / at(lu, i, i);
      out[(size_t)j * n + i] = x[i];
    }
  }
  lu.swap(out);
}

namespace
{
struct AggResult
{
  int n_vec = 0;
  std::vector<double> weights; // [n_vec][nloc] already multiplied by diag_loc, not yet by 1/diag_glob
};

struct AggKey
{
  std::vector<char> bytes;
  bool operator<(AggKey const &o) const { return bytes < o.bytes; }
};
} // namespace

HostCsr build_restrictor_structured(StructuredMesh const &mesh, std::vector<double> const &global_diag,
                                    RestrictorOptions const &opts, std::vector<int32_t> *row_agglomerate,
                                    int *agglomerate_counts, HipHandle *device)
{
  const int dim = mesh.dim;
  const int nc = mesh.nc();
  ASSERT_THROW(opts.variant == "device" || opts.variant == "host" || opts.variant == "mf",
               "unknown AMGe variant \"" + opts.variant + "\"");
  ASSERT_THROW(opts.selection == "lapack" || opts.selection == "krylov",
               "unknown eigenvector selection \"" + opts.selection + "\"");
  ASSERT_THROW(opts.n_eigenvectors >= 1, "number of eigenvectors must be positive");
  int ag[3] = {1, 1, 1}, cnt[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d)
  {
    ag[d] = opts.agglomerate[d];
    ASSERT_THROW(ag[d] >= 1, "agglomerate dimensions must be positive");
    cnt[d] = (mesh.n[d] + ag[d] - 1) / ag[d];
  }
  const int64_t n_agg = (int64_t)cnt[0] * cnt[1] * cnt[2];
  const auto Kq = reference_cell_tables(dim, mesh.h);
  double h1[3] = {mesh.h[0], mesh.h[1], mesh.h[2]};
  (void)h1;

  std::vector<std::shared_ptr<AggResult const>> result_of(n_agg);
  std::vector<std::vector<int32_t>> dofs_of(n_agg);
  // eigenproblems on the device (one wavefront per agglomerate, amge_device.hip) when a handle is given
  std::vector<double> dev_weights;
  std::vector<int32_t> dev_n_vec;
  int dev_nmax = 0;
  const bool on_device = device != nullptr && amge_device_supported(mesh, opts);
  const bool verbose_t = std::getenv("MFMG_HIP_VERBOSE") != nullptr;
  auto now_t = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tt0 = now_t();
  if (on_device)
    amge_device_eigen(*device, mesh, opts, cnt, dev_weights, dev_n_vec, dev_nmax);
  const double tt1 = now_t();
  // identical agglomerates (same shape, constraints and local matrix) share one eigen-solve;
  // the table is capped so that a spatially varying coefficient cannot blow up host memory
  // (one table per thread: no lock on the hot path; at most threads x classes eigen-solves)
  constexpr size_t kMemoCap = 1024;
  configure_host_threads();

#pragma omp parallel
  {
    std::vector<double> A, M, w, V, v0, proj;
    std::vector<char> lcon;
    std::map<AggKey, std::shared_ptr<AggResult const>> memo; // thread private
#pragma omp for schedule(dynamic, 64)
    for (int64_t a = 0; a < n_agg; ++a)
    {
      int ai[3] = {(int)(a % cnt[0]), (int)((a / cnt[0]) % cnt[1]), (int)(a / ((int64_t)cnt[0] * cnt[1]))};
      int lo[3] = {0, 0, 0}, ln[3] = {1, 1, 1}, lN[3] = {1, 1, 1};
      for (int d = 0; d < dim; ++d)
      {
        lo[d] = ai[d] * ag[d];
        ln[d] = std::min(ag[d], mesh.n[d] - lo[d]);
        lN[d] = ln[d] + 1;
      }
      if (dim == 2)
      {
        ln[2] = 1;
        lN[2] = 1;
      }
      const int nloc = lN[0] * lN[1] * lN[2];
      auto lidx = [&](int i, int j, int k) { return i + lN[0] * (j + lN[1] * k); };
      std::vector<int32_t> &gl = dofs_of[a];
      gl.resize(nloc);
      lcon.assign(nloc, 0);
      for (int k = 0; k < lN[2]; ++k)
        for (int j = 0; j < lN[1]; ++j)
          for (int i = 0; i < lN[0]; ++i)
          {
            const int32_t g = mesh.node_dof[mesh.node_index(lo[0] + i, lo[1] + j, (dim == 3) ? lo[2] + k : 0)];
            gl[lidx(i, j, k)] = g;
            lcon[lidx(i, j, k)] = (mesh.constrained[g] == 1);
          }
      if (on_device)
        continue;
      // local (Neumann) matrix from the cell matrices of the agglomerate
      A.assign((size_t)nloc * nloc, 0.);
      const int lkz = (dim == 3) ? ln[2] : 1;
      for (int k = 0; k < lkz; ++k)
        for (int j = 0; j < ln[1]; ++j)
          for (int i = 0; i < ln[0]; ++i)
          {
            const int64_t c = mesh.cell_index(lo[0] + i, lo[1] + j, (dim == 3) ? lo[2] + k : 0);
            int ld[8];
            for (int m = 0; m < nc; ++m)
              ld[m] = lidx(i + (m & 1), j + ((m >> 1) & 1), (dim == 3) ? k + ((m >> 2) & 1) : 0);
            for (int m = 0; m < nc; ++m)
              for (int mp = 0; mp < nc; ++mp)
              {
                double v = 0.;
                for (int q = 0; q < nc; ++q)
                  v += (opts.use_coefficient ? mesh.coefficient[c * nc + q] : 1.) *
                       Kq[((size_t)q * nc + m) * nc + mp];
                A[(size_t)ld[m] * nloc + ld[mp]] += v;
              }
          }
      // memo key: shape, constraint pattern and the matrix bytes
      AggKey key;
      key.bytes.resize(3 * sizeof(int) + nloc + A.size() * sizeof(double));
      std::memcpy(key.bytes.data(), lN, 3 * sizeof(int));
      std::memcpy(key.bytes.data() + 3 * sizeof(int), lcon.data(), nloc);
      std::memcpy(key.bytes.data() + 3 * sizeof(int) + nloc, A.data(), A.size() * sizeof(double));
      std::shared_ptr<AggResult const> found;
      {
        auto it = memo.find(key);
        if (it != memo.end())
          found = it->second;
      }
      if (!found)
      {
        AggResult res;
        std::vector<double> diag_loc(nloc);
        // eliminate constrained rows / columns
        std::vector<double> full_diag(nloc);
        for (int r = 0; r < nloc; ++r)
          full_diag[r] = A[(size_t)r * nloc + r];
        for (int r = 0; r < nloc; ++r)
          for (int c2 = 0; c2 < nloc; ++c2)
            if ((lcon[r] || lcon[c2]) && r != c2)
              A[(size_t)r * nloc + c2] = 0.;
        for (int r = 0; r < nloc; ++r)
        {
          if (lcon[r])
            A[(size_t)r * nloc + r] = (opts.variant == "mf") ? 1. : full_diag[r];
          diag_loc[r] = A[(size_t)r * nloc + r];
        }
        double avg = 0.;
        // free-DoF compaction for the matrix-free variant
        std::vector<int> act;
        if (opts.variant == "mf")
        {
          for (int r = 0; r < nloc; ++r)
            if (!lcon[r])
              act.push_back(r);
        }
        else
        {
          act.resize(nloc);
          std::iota(act.begin(), act.end(), 0);
        }
        const int na = (int)act.size();
        M.assign((size_t)na * na, 0.);
        for (int r = 0; r < na; ++r)
          for (int c2 = 0; c2 < na; ++c2)
            M[(size_t)r * na + c2] = A[(size_t)act[r] * nloc + act[c2]];
        if (opts.variant == "host")
        {
          for (int r = 0; r < nloc; ++r)
            avg += diag_loc[r];
          avg /= nloc;
          for (int r = 0; r < nloc; ++r)
            M[(size_t)r * nloc + r] = lcon[r] ? 200. : M[(size_t)r * nloc + r] + avg;
        }
        if (na > 0)
          symmetric_eigen(na, M, w, V);
        // selection
        std::vector<std::vector<double>> sel; // local vectors of length nloc
        if (opts.selection == "lapack")
        {
          for (int e = 0; e < std::min(opts.n_eigenvectors, na); ++e)
          {
            std::vector<double> vec(nloc, 0.);
            for (int r = 0; r < na; ++r)
              vec[act[r]] = V[(size_t)e * na + r];
            sel.push_back(std::move(vec));
          }
        }
        else
        {
          // start vector in deal.II's first-touch numbering of the patch
          std::vector<int> first_touch(nloc, -1);
          int nxt = 0;
          for (int k = 0; k < lkz; ++k)
            for (int j = 0; j < ln[1]; ++j)
              for (int i = 0; i < ln[0]; ++i)
                for (int m = 0; m < nc; ++m)
                {
                  const int l = lidx(i + (m & 1), j + ((m >> 1) & 1), (dim == 3) ? k + ((m >> 2) & 1) : 0);
                  if (first_touch[l] < 0)
                    first_touch[l] = nxt++;
                }
          std::vector<int> inv(nloc);
          for (int l = 0; l < nloc; ++l)
            inv[first_touch[l]] = l;
          MinstdUniform gen;
          std::vector<double> start(nloc, 0.);
          for (int t = 0; t < nloc; ++t)
          {
            const int l = inv[t];
            start[l] = lcon[l] ? 0. : gen.next();
          }
          v0.assign(na, 0.);
          double v0n = 0.;
          for (int r = 0; r < na; ++r)
          {
            v0[r] = start[act[r]];
            v0n += v0[r] * v0[r];
          }
          v0n = std::sqrt(v0n);
          const double scale = (na > 0) ? std::max(std::abs(w[na - 1]), 1e-300) : 1.;
          int i0 = 0;
          while (i0 < na && (int)sel.size() < opts.n_eigenvectors)
          {
            int i1 = i0 + 1;
            while (i1 < na && std::abs(w[i1] - w[i0]) <= 1e-9 * scale)
              ++i1;
            proj.assign(na, 0.);
            for (int e = i0; e < i1; ++e)
            {
              double dotp = 0.;
              for (int r = 0; r < na; ++r)
                dotp += V[(size_t)e * na + r] * v0[r];
              for (int r = 0; r < na; ++r)
                proj[r] += dotp * V[(size_t)e * na + r];
            }
            double pn = 0.;
            for (int r = 0; r < na; ++r)
              pn += proj[r] * proj[r];
            pn = std::sqrt(pn);
            if (pn > 1e-12 * v0n)
            {
              std::vector<double> vec(nloc, 0.);
              for (int r = 0; r < na; ++r)
                vec[act[r]] = proj[r] / pn;
              sel.push_back(std::move(vec));
            }
            i0 = i1;
          }
        }
        res.n_vec = (int)sel.size();
        res.weights.resize((size_t)res.n_vec * nloc);
        for (int e = 0; e < res.n_vec; ++e)
          for (int l = 0; l < nloc; ++l)
            res.weights[(size_t)e * nloc + l] = diag_loc[l] * sel[e][l];
        found = std::make_shared<AggResult const>(std::move(res));
        if (memo.size() < kMemoCap)
          memo.emplace(std::move(key), found);
      }
      result_of[a] = found;
    }
  }

  const double tt2 = now_t();
  // assemble R (rows: agglomerates in x-fastest order, eigenvectors inside; columns sorted)
  HostCsr R;
  R.n_cols = mesh.n_dofs;
  auto n_vec_of = [&](int64_t a) { return on_device ? (int)dev_n_vec[a] : result_of[a]->n_vec; };
  auto weight_of = [&](int64_t a, int e, int l, int nloc) {
    return on_device ? dev_weights[((size_t)a * opts.n_eigenvectors + e) * dev_nmax + l] : result_of[a]->weights[(size_t)e * nloc + l];
  };
  std::vector<int64_t> first_row(n_agg + 1, 0);
  for (int64_t a = 0; a < n_agg; ++a)
    first_row[a + 1] = first_row[a] + n_vec_of(a);
  R.n_rows = first_row[n_agg];
  if (row_agglomerate)
  {
    row_agglomerate->resize(R.n_rows);
    for (int64_t a = 0; a < n_agg; ++a)
      for (int64_t r = first_row[a]; r < first_row[a + 1]; ++r)
        (*row_agglomerate)[r] = (int32_t)a;
  }
  if (agglomerate_counts)
    for (int d = 0; d < 3; ++d)
      agglomerate_counts[d] = cnt[d];
  R.row_ptr.assign(R.n_rows + 1, 0);
  for (int64_t a = 0; a < n_agg; ++a)
    for (int e = 0; e < n_vec_of(a); ++e)
      R.row_ptr[first_row[a] + e + 1] = (int32_t)dofs_of[a].size();
  int64_t total = 0;
  for (int64_t r = 0; r < R.n_rows; ++r)
  {
    total += R.row_ptr[r + 1];
    ASSERT_THROW(total < (int64_t(1) << 31), "restriction matrix exceeds int32 nnz");
    R.row_ptr[r + 1] = (int32_t)total;
  }
  R.col.resize(total);
  R.val.resize(total);
#pragma omp parallel
  {
    std::vector<int> perm;
#pragma omp for schedule(static)
    for (int64_t a = 0; a < n_agg; ++a)
    {
      auto const &gl = dofs_of[a];
      const int nloc = (int)gl.size();
      perm.resize(nloc);
      std::iota(perm.begin(), perm.end(), 0);
      std::sort(perm.begin(), perm.end(), [&](int x, int y) { return gl[x] < gl[y]; });
      for (int e = 0; e < n_vec_of(a); ++e)
      {
        const int64_t base = R.row_ptr[first_row[a] + e];
        for (int t = 0; t < nloc; ++t)
        {
          const int l = perm[t];
          R.col[base + t] = gl[l];
          R.val[base + t] = weight_of(a, e, l, nloc) / global_diag[gl[l]];
        }
      }
    }
  }
  if (verbose_t)
    std::fprintf(stderr, "[mfmg_hip] restrictor rows: eigenproblems %.2f s, agglomerate DoF lists %.2f s, CSR assembly %.2f s\n", tt1 - tt0,
                 tt2 - tt1, now_t() - tt2);
  return R;
}

HostCsr galerkin_triple_product(StructuredMesh const &mesh, ConstraintSemantics sem, HostCsr const &R,
                                HostCsr const &Rt)
{
  configure_host_threads();
  ASSERT_THROW(R.n_cols == mesh.n_dofs && Rt.n_rows == mesh.n_dofs && Rt.n_cols == R.n_rows,
               "restrictor shape does not match the mesh");
  const auto Kq = reference_cell_tables(mesh.dim, mesh.h);
  const auto dn = dof_to_node(mesh);
  const int64_t nc_rows = R.n_rows;
  std::vector<std::vector<int32_t>> row_cols(nc_rows);
  std::vector<std::vector<double>> row_vals(nc_rows);
#pragma omp parallel
  {
    std::vector<double> facc(mesh.n_dofs, 0.);
    std::vector<char> fmark(mesh.n_dofs, 0);
    std::vector<int32_t> ftouched;
    std::vector<double> cacc(nc_rows, 0.);
    std::vector<char> cmark(nc_rows, 0);
    std::vector<int32_t> ctouched;
    // tiny direct-mapped cache of operator rows (the eigenvectors of one agglomerate share them)
    constexpr int kCache = 256; // keyed by the position inside the restrictor row
    std::vector<int32_t> cache_g(kCache, -1);
    std::vector<std::vector<int32_t>> cache_cols(kCache);
    std::vector<std::vector<double>> cache_vals(kCache);
#pragma omp for schedule(dynamic, 128)
    for (int64_t r = 0; r < nc_rows; ++r)
    {
      ftouched.clear();
      for (int p = R.row_ptr[r]; p < R.row_ptr[r + 1]; ++p)
      {
        const int32_t g = R.col[p];
        const double wgt = R.val[p];
        if (wgt == 0.)
          continue;
        const int slot = (p - R.row_ptr[r]) % kCache;
        if (cache_g[slot] != g)
        {
          int i, j, k;
          node_ijk(mesh, dn[g], i, j, k);
          operator_row(mesh, Kq, sem, i, j, k, cache_cols[slot], cache_vals[slot]);
          cache_g[slot] = g;
        }
        auto const &cols = cache_cols[slot];
        auto const &vals = cache_vals[slot];
        for (size_t t = 0; t < cols.size(); ++t)
        {
          const int32_t g2 = cols[t];
          if (!fmark[g2])
          {
            fmark[g2] = 1;
            ftouched.push_back(g2);
          }
          facc[g2] += wgt * vals[t];
        }
      }
      std::sort(ftouched.begin(), ftouched.end());
      ctouched.clear();
      for (int32_t g2 : ftouched)
      {
        const double v = facc[g2];
        facc[g2] = 0.;
        fmark[g2] = 0;
        for (int p = Rt.row_ptr[g2]; p < Rt.row_ptr[g2 + 1]; ++p)
        {
          const int32_t r2 = Rt.col[p];
          if (!cmark[r2])
          {
            cmark[r2] = 1;
            ctouched.push_back(r2);
          }
          cacc[r2] += v * Rt.val[p];
        }
      }
      std::sort(ctouched.begin(), ctouched.end());
      row_cols[r].assign(ctouched.begin(), ctouched.end());
      row_vals[r].resize(ctouched.size());
      for (size_t t = 0; t < ctouched.size(); ++t)
      {
        row_vals[r][t] = cacc[ctouched[t]];
        cacc[ctouched[t]] = 0.;
        cmark[ctouched[t]] = 0;
      }
    }
  }
  HostCsr Ac;
  Ac.n_rows = Ac.n_cols = nc_rows;
  Ac.row_ptr.assign(nc_rows + 1, 0);
  int64_t total = 0;
  for (int64_t r = 0; r < nc_rows; ++r)
  {
    total += (int64_t)row_cols[r].size();
    ASSERT_THROW(total < (int64_t(1) << 31), "coarse matrix exceeds int32 nnz");
    Ac.row_ptr[r + 1] = (int32_t)total;
  }
  Ac.col.resize(total);
  Ac.val.resize(total);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nc_rows; ++r)
  {
    std::copy(row_cols[r].begin(), row_cols[r].end(), Ac.col.begin() + Ac.row_ptr[r]);
    std::copy(row_vals[r].begin(), row_vals[r].end(), Ac.val.begin() + Ac.row_ptr[r]);
  }
  return Ac;
}
} // namespace mfmg
