// The agglomerate eigenproblems of the spectral AMGe restrictor on the device, batched: one wavefront per
// agglomerate, its local matrix and eigenvectors in LDS.
//
// Reference: AMGe_device solves them one after the other with cusolverDnDsygvd (include/mfmg/cuda/amge_device.templates.cuh:217-432,
// TODO at :391-392); the host path runs ARPACK / LAPACK / Lanczos per agglomerate under a TBB WorkStream
// (include/mfmg/dealii/amge_host.templates.hpp:356-483).  The rules restated here are those of
// build_restrictor_structured (amge_structured.cpp), which the oracle pins: local Neumann matrix from the cell
// matrices, constrained rows eliminated ("device": diagonal kept; "host": shifted by the mean diagonal, constrained
// diagonal 200; "mf": constrained DoFs dropped from the eigenproblem), cyclic Jacobi in the same rotation order,
// selection "lapack" (first columns) or "krylov" (one vector per distinct eigenvalue: the projection of the start
// vector of DealIIMeshEvaluator::set_initial_guess onto the eigenspace), weights diag_loc * vector
// (include/mfmg/common/amge.templates.hpp:300-321; the division by the global diagonal happens at assembly).
// Every floating-point operation is done in the order of the host code (sums over cells, over quadrature points,
// the Jacobi rotations, the projections), contraction off, so that host and device agree to rounding of the
// transcendental-free arithmetic; the lanes of the wavefront only split loops whose iterations are independent.
#include "amge_device.hpp"

#include <cstdlib>
#include <string>
#include <unordered_map>
#include <utility>

#include <algorithm>
#include <cmath>

namespace mfmg
{
namespace
{
struct AmgeArgs
{
  int dim, nc;
  int n[3], N[3];   // cells, nodes of the mesh
  int ag[3], cnt[3]; // cells per agglomerate, agglomerates per direction
  int variant;       // 0 device, 1 host, 2 mf
  int krylov;        // selection: 0 lapack, 1 krylov
  int n_eig;
  int use_coefficient;
  int32_t const *node_dof;
  uint8_t const *constrained;
  double const *coefficient; // [cells][nc]
  double const *Kq;          // [nc][nc][nc]
  double *weights;           // [agglomerates][n_eig][NMAX]
  int32_t *n_vec;            // [agglomerates]
  int64_t n_agg;
  int64_t const *list; // nullptr: every agglomerate; otherwise the n_agg agglomerates to solve (representatives)
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off);
  return v;
}

template <int NMAX>
__global__ __launch_bounds__(256) void amge_agglomerate_kernel(AmgeArgs a)
{
#pragma clang fp contract(off)
  extern __shared__ double smem[];
  constexpr int kPerWave = 2 * NMAX * NMAX + 5 * NMAX;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  double *M = smem + (size_t)wv * kPerWave; // [na][na] row-major, stride na
  double *V = M + NMAX * NMAX;              // V[col * na + row]
  double *w = V + NMAX * NMAX;              // eigenvalues (diagonal of M after the sweeps)
  double *v0 = w + NMAX;                    // start vector on the active DoFs
  double *proj = v0 + NMAX;
  double *dloc = proj + NMAX;               // diag_loc on all local DoFs
  double *startv = dloc + NMAX;             // start vector on all local DoFs
  const int dim = a.dim, nc = a.nc;

  for (int64_t slot = (int64_t)blockIdx.x * wpb + wv; slot < a.n_agg; slot += (int64_t)gridDim.x * wpb)
  {
    const int64_t agg = a.list ? a.list[slot] : slot;
    int ai[3] = {(int)(agg % a.cnt[0]), (int)((agg / a.cnt[0]) % a.cnt[1]), (int)(agg / ((int64_t)a.cnt[0] * a.cnt[1]))};
    int lo[3] = {0, 0, 0}, ln[3] = {1, 1, 1}, lN[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d)
    {
      lo[d] = ai[d] * a.ag[d];
      ln[d] = min(a.ag[d], a.n[d] - lo[d]);
      lN[d] = ln[d] + 1;
    }
    if (dim == 2)
    {
      ln[2] = 1;
      lN[2] = 1;
    }
    const int nloc = lN[0] * lN[1] * lN[2];
    const int lkz = (dim == 3) ? ln[2] : 1;
    // ---- this lane's local DoF
    const bool has = lane < nloc;
    const int li = lane % lN[0], lj = (lane / lN[0]) % lN[1], lk = lane / (lN[0] * lN[1]);
    bool con = false;
    if (has)
    {
      const int64_t node = (lo[0] + li) + (int64_t)a.N[0] * ((lo[1] + lj) + (int64_t)a.N[1] * ((dim == 3) ? lo[2] + lk : 0));
      con = a.constrained[a.node_dof[node]] == 1;
    }
    const unsigned long long con_mask = __ballot(con);
    const unsigned long long has_mask = __ballot(has);
    // active DoFs of the eigenproblem: all of them, or the unconstrained ones ("mf")
    const unsigned long long act_mask = (a.variant == 2) ? (has_mask & ~con_mask) : has_mask;
    const int na = __popcll(act_mask);
    auto pos_of = [&](int l) { return __popcll(act_mask & ((1ull << l) - 1ull)); }; // position of local DoF l among the active
    // entry (r, c) of the local Neumann matrix: contributions of the cells in (k, j, i) order
    auto entry = [&](int r, int c) {
      const int ri = r % lN[0], rj = (r / lN[0]) % lN[1], rk = r / (lN[0] * lN[1]);
      const int ci = c % lN[0], cj = (c / lN[0]) % lN[1], ck = c / (lN[0] * lN[1]);
      double sum = 0.;
      for (int k = 0; k < lkz; ++k)
        for (int j = 0; j < ln[1]; ++j)
          for (int i = 0; i < ln[0]; ++i)
          {
            const int mr0 = ri - i, mr1 = rj - j, mr2 = (dim == 3) ? rk - k : 0;
            const int mc0 = ci - i, mc1 = cj - j, mc2 = (dim == 3) ? ck - k : 0;
            if ((unsigned)mr0 > 1u || (unsigned)mr1 > 1u || (unsigned)mr2 > 1u || (unsigned)mc0 > 1u || (unsigned)mc1 > 1u ||
                (unsigned)mc2 > 1u)
              continue;
            const int m = mr0 + 2 * mr1 + 4 * mr2, mp = mc0 + 2 * mc1 + 4 * mc2;
            const int64_t cell = (lo[0] + i) + (int64_t)a.n[0] * ((lo[1] + j) + (int64_t)a.n[1] * ((dim == 3) ? lo[2] + k : 0));
            double v = 0.;
            for (int q = 0; q < nc; ++q)
              v += (a.use_coefficient ? a.coefficient[cell * nc + q] : 1.) * a.Kq[((size_t)q * nc + m) * nc + mp];
            sum += v;
          }
      return sum;
    };
    // ---- diag_loc: constrained rows keep 1 ("mf") or the summed local diagonal
    const double full_diag = has ? entry(lane, lane) : 0.;
    const double dl = con ? ((a.variant == 2) ? 1. : full_diag) : full_diag;
    if (has)
      dloc[lane] = dl;
    double avg = 0.;
    if (a.variant == 1)
    {
      // mean of diag_loc, summed in index order like the host loop
      for (int r = 0; r < nloc; ++r)
        avg += dloc[r];
      avg /= nloc;
    }
    // ---- the matrix of the eigenproblem: constrained rows / columns eliminated
    for (int e = lane; e < nloc * nloc; e += 64)
    {
      const int r = e / nloc, c = e % nloc;
      const bool rc = (con_mask >> r) & 1ull, cc = (con_mask >> c) & 1ull;
      if (a.variant == 2 && (rc || cc))
        continue; // not part of the eigenproblem
      double v;
      if (r == c)
      {
        const double d = dloc[r];
        v = (a.variant == 1) ? (rc ? 200. : d + avg) : d;
      }
      else
        v = (rc || cc) ? 0. : entry(r, c);
      M[pos_of(r) * na + pos_of(c)] = v;
    }
    for (int e = lane; e < na * na; e += 64)
      V[e] = (e / na == e % na) ? 1. : 0.;
    // ---- cyclic Jacobi, rotations in the order of symmetric_eigen (amge_structured.cpp)
    for (int sweep = 0; sweep < 64; ++sweep)
    {
      double off = 0., dsum = 0.;
      for (int e = lane; e < na * na; e += 64)
      {
        const int p = e / na, q = e % na;
        const double v = M[e];
        if (p == q)
          dsum += v * v;
        else if (q > p)
          off += v * v;
      }
      off = wave_sum(off);
      dsum = wave_sum(dsum);
      if (off <= 1e-32 * (dsum + off) || off == 0.)
        break;
      for (int p = 0; p < na - 1; ++p)
        for (int q = p + 1; q < na; ++q)
        {
          const double apq = M[p * na + q];
          if (fabs(apq) < 1e-300)
            continue;
          const double theta = (M[q * na + q] - M[p * na + p]) / (2. * apq);
          const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
          const double c = 1. / sqrt(t * t + 1.);
          const double s = t * c;
          const int r = lane;
          if (r < na)
          {
            const double arp = M[r * na + p], arq = M[r * na + q];
            M[r * na + p] = c * arp - s * arq;
            M[r * na + q] = s * arp + c * arq;
          }
          if (r < na)
          {
            const double apr = M[p * na + r], aqr = M[q * na + r];
            M[p * na + r] = c * apr - s * aqr;
            M[q * na + r] = s * apr + c * aqr;
          }
          if (r < na)
          {
            const double vrp = V[p * na + r], vrq = V[q * na + r];
            V[p * na + r] = c * vrp - s * vrq;
            V[q * na + r] = s * vrp + c * vrq;
          }
        }
    }
    // ---- ascending eigenvalues, ties in index order (std::stable_sort): rank of column x
    int rank = 0;
    double wx = 0.;
    if (lane < na)
    {
      wx = M[lane * na + lane];
      for (int y = 0; y < na; ++y)
      {
        const double wy = M[y * na + y];
        rank += (wy < wx || (wy == wx && y < lane)) ? 1 : 0;
      }
    }
    // column of rank e: perm[e]; kept in w / a small index array inside proj's space is not needed: every lane can find it
    if (lane < na)
      w[rank] = wx;
    // permutation as doubles in startv (reused below): perm[rank] = lane
    if (lane < na)
      startv[rank] = (double)lane;
    int perm_of_lane = (lane < na) ? (int)startv[lane] : 0; // perm[lane]
    // ---- selection
    int n_sel = 0;
    double *out = a.weights + (size_t)agg * a.n_eig * NMAX;
    if (!a.krylov)
    {
      const int ne = min(a.n_eig, na);
      for (int e = 0; e < ne; ++e)
      {
        const int col = __shfl(perm_of_lane, e);
        if (has)
        {
          const bool active = (act_mask >> lane) & 1ull;
          out[(size_t)e * NMAX + lane] = active ? dl * V[col * na + pos_of(lane)] : 0.;
        }
      }
      n_sel = ne;
    }
    else
    {
      // start vector: libstdc++ minstd_rand0 + uniform_real_distribution in deal.II's first-touch order of the patch,
      // zero on constrained DoFs (which consume no random number)
      if (lane == 0)
      {
        unsigned long long seen = 0ull, state = 1ull;
        for (int k = 0; k < lkz; ++k)
          for (int j = 0; j < ln[1]; ++j)
            for (int i = 0; i < ln[0]; ++i)
              for (int m = 0; m < nc; ++m)
              {
                const int l = (i + (m & 1)) + lN[0] * ((j + ((m >> 1) & 1)) + lN[1] * ((dim == 3) ? k + ((m >> 2) & 1) : 0));
                if ((seen >> l) & 1ull)
                  continue;
                seen |= 1ull << l;
                double val = 0.;
                if (!((con_mask >> l) & 1ull))
                {
                  const double R = 2147483646.0;
                  state = (16807ull * state) % 2147483647ull;
                  double s = (double)(state - 1);
                  state = (16807ull * state) % 2147483647ull;
                  s += (double)(state - 1) * R;
                  val = s / (R * R);
                  if (val >= 1.0)
                    val = 0.99999999999999988898; // nextafter(1, 0)
                }
                startv[l] = val;
              }
      }
      // (perm is in registers now; startv is rewritten by lane 0 above, LDS operations of a wavefront are in order)
      if (has && ((act_mask >> lane) & 1ull))
        v0[pos_of(lane)] = startv[lane];
      double v0n = 0.;
      for (int r = 0; r < na; ++r)
        v0n += v0[r] * v0[r];
      v0n = sqrt(v0n);
      const double scale = (na > 0) ? fmax(fabs(w[na - 1]), 1e-300) : 1.;
      int i0 = 0;
      while (i0 < na && n_sel < a.n_eig)
      {
        int i1 = i0 + 1;
        while (i1 < na && fabs(w[i1] - w[i0]) <= 1e-9 * scale)
          ++i1;
        double pr = 0.; // proj[lane]
        for (int e = i0; e < i1; ++e)
        {
          const int col = __shfl(perm_of_lane, e);
          double dotp = 0.;
          for (int r = 0; r < na; ++r)
            dotp += V[col * na + r] * v0[r];
          if (lane < na)
            pr += dotp * V[col * na + lane];
        }
        if (lane < na)
          proj[lane] = pr;
        double pn = 0.;
        for (int r = 0; r < na; ++r)
          pn += proj[r] * proj[r];
        pn = sqrt(pn);
        if (pn > 1e-12 * v0n)
        {
          if (has)
          {
            const bool active = (act_mask >> lane) & 1ull;
            out[(size_t)n_sel * NMAX + lane] = active ? dl * (proj[pos_of(lane)] / pn) : 0.;
          }
          ++n_sel;
        }
        i0 = i1;
      }
    }
    if (lane == 0)
      a.n_vec[agg] = n_sel;
  }
}
} // namespace

bool amge_device_supported(StructuredMesh const &mesh, RestrictorOptions const &opts)
{
  int nloc = 1;
  for (int d = 0; d < mesh.dim; ++d)
    nloc *= opts.agglomerate[d] + 1;
  return nloc <= 64 && (mesh.dim == 2 || mesh.dim == 3);
}

namespace
{
// What the eigenproblem of an agglomerate depends on: its shape, the constraint flags of its nodes and the coefficients of
// its cells.  Two independent 64-bit hashes of exactly that; agglomerates with equal keys get ONE solve (the kernel is
// deterministic: equal input, equal bits) -- for a coefficient that is the same in every cell 27 solves instead of 2.1 M.
__global__ void amge_key_kernel(AmgeArgs a, uint64_t *keys)
{
  for (int64_t agg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; agg < a.n_agg; agg += (int64_t)gridDim.x * blockDim.x)
  {
    const int dim = a.dim, nc = a.nc;
    int ai[3] = {(int)(agg % a.cnt[0]), (int)((agg / a.cnt[0]) % a.cnt[1]), (int)(agg / ((int64_t)a.cnt[0] * a.cnt[1]))};
    int lo[3] = {0, 0, 0}, ln[3] = {1, 1, 1}, lN[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d)
    {
      lo[d] = ai[d] * a.ag[d];
      ln[d] = min(a.ag[d], a.n[d] - lo[d]);
      lN[d] = ln[d] + 1;
    }
    if (dim == 2)
    {
      ln[2] = 1;
      lN[2] = 1;
    }
    uint64_t h1 = 1469598103934665603ull, h2 = 0x9e3779b97f4a7c15ull;
    auto mix = [&](uint64_t v) {
      h1 = (h1 ^ v) * 1099511628211ull;
      h1 ^= h1 >> 29;
      h2 += v + 0x9e3779b97f4a7c15ull;
      h2 = (h2 ^ (h2 >> 30)) * 0xbf58476d1ce4e5b9ull;
      h2 = (h2 ^ (h2 >> 27)) * 0x94d049bb133111ebull;
      h2 ^= h2 >> 31;
    };
    mix((uint64_t)ln[0] | ((uint64_t)ln[1] << 16) | ((uint64_t)ln[2] << 32));
    for (int k = 0; k < lN[2]; ++k)
      for (int j = 0; j < lN[1]; ++j)
        for (int i = 0; i < lN[0]; ++i)
        {
          const int64_t node = (lo[0] + i) + (int64_t)a.N[0] * ((lo[1] + j) + (int64_t)a.N[1] * ((dim == 3) ? lo[2] + k : 0));
          mix(a.constrained[a.node_dof[node]] == 1 ? 1u : 0u);
        }
    if (a.use_coefficient)
      for (int k = 0; k < ((dim == 3) ? ln[2] : 1); ++k)
        for (int j = 0; j < ln[1]; ++j)
          for (int i = 0; i < ln[0]; ++i)
          {
            const int64_t cell = (lo[0] + i) + (int64_t)a.n[0] * ((lo[1] + j) + (int64_t)a.n[1] * ((dim == 3) ? lo[2] + k : 0));
            for (int q = 0; q < nc; ++q)
              mix((uint64_t)__double_as_longlong(a.coefficient[cell * nc + q]));
          }
    keys[2 * agg] = h1;
    keys[2 * agg + 1] = h2;
  }
}

// results of the representatives to every member of their class
__global__ void amge_spread_kernel(int64_t n_agg, int per_agg, int64_t const *rep_of, double *weights, int32_t *n_vec)
{
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_agg * per_agg; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t agg = t / per_agg, r = rep_of[agg];
    if (r == agg)
      continue;
    const int64_t e = t - agg * per_agg;
    weights[agg * per_agg + e] = weights[r * per_agg + e];
    if (e == 0)
      n_vec[agg] = n_vec[r];
  }
}
} // namespace

void amge_device_eigen(HipHandle &handle, StructuredMesh const &mesh, RestrictorOptions const &opts, int const cnt[3],
                       std::vector<double> &weights, std::vector<int32_t> &n_vec, int &nmax)
{
  const int dim = mesh.dim, nc = mesh.nc();
  int nloc = 1;
  for (int d = 0; d < dim; ++d)
    nloc *= opts.agglomerate[d] + 1;
  ASSERT_THROW(nloc <= 64, "agglomerates of more than 64 nodes are solved on the host");
  nmax = nloc <= 27 ? 27 : 64;
  AmgeArgs a;
  a.dim = dim;
  a.nc = nc;
  for (int d = 0; d < 3; ++d)
  {
    a.n[d] = d < dim ? mesh.n[d] : 1;
    a.N[d] = d < dim ? mesh.N[d] : 1;
    a.ag[d] = d < dim ? opts.agglomerate[d] : 1;
    a.cnt[d] = cnt[d];
  }
  a.variant = opts.variant == "device" ? 0 : (opts.variant == "host" ? 1 : 2);
  a.krylov = opts.selection == "krylov" ? 1 : 0;
  a.n_eig = opts.n_eigenvectors;
  a.use_coefficient = opts.use_coefficient ? 1 : 0;
  a.n_agg = (int64_t)cnt[0] * cnt[1] * cnt[2];
  const auto Kq = reference_cell_tables(dim, mesh.h);
  hipStream_t st = handle.stream;
  DeviceBuffer<int32_t> d_node;
  DeviceBuffer<uint8_t> d_con;
  DeviceBuffer<double> d_coef, d_kq;
  d_node.upload(mesh.node_dof.data(), mesh.node_dof.size(), st);
  d_con.upload(mesh.constrained.data(), mesh.constrained.size(), st);
  d_coef.upload(mesh.coefficient.data(), mesh.coefficient.size(), st);
  d_kq.upload(Kq.data(), Kq.size(), st);
  DeviceBuffer<double> d_w((size_t)a.n_agg * a.n_eig * nmax);
  DeviceBuffer<int32_t> d_nv((size_t)a.n_agg);
  MFMG_HIP_CHECK(hipMemsetAsync(d_w.data(), 0, d_w.size() * sizeof(double), st));
  a.node_dof = d_node.data();
  a.constrained = d_con.data();
  a.coefficient = d_coef.data();
  a.Kq = d_kq.data();
  a.weights = d_w.data();
  a.n_vec = d_nv.data();
  a.list = nullptr;
  // identical agglomerates share one solve (the host path does the same with a table per thread)
  const int64_t n_all = a.n_agg;
  DeviceBuffer<int64_t> d_list, d_rep_of;
  bool shared_solves = false;
  if (!(std::getenv("MFMG_AMGE_MEMO") && std::string(std::getenv("MFMG_AMGE_MEMO")) == "0") && n_all >= 64)
  {
    DeviceBuffer<uint64_t> d_keys((size_t)2 * n_all);
    hipLaunchKernelGGL(amge_key_kernel, dim3(n_blocks_for(n_all, 256, 1 << 16)), dim3(256), 0, st, a, d_keys.data());
    MFMG_HIP_CHECK(hipGetLastError());
    const std::vector<uint64_t> keys = d_keys.download(st);
    struct KeyHash
    {
      size_t operator()(std::pair<uint64_t, uint64_t> const &k) const { return (size_t)(k.first ^ (k.second * 0x9e3779b97f4a7c15ull)); }
    };
    std::unordered_map<std::pair<uint64_t, uint64_t>, int64_t, KeyHash> first_of;
    std::vector<int64_t> rep_of((size_t)n_all), reps;
    for (int64_t agg = 0; agg < n_all; ++agg)
    {
      auto ins = first_of.emplace(std::make_pair(keys[2 * agg], keys[2 * agg + 1]), agg);
      rep_of[agg] = ins.first->second;
      if (ins.second)
        reps.push_back(agg);
      if ((int64_t)reps.size() * 2 > n_all && agg * 2 < n_all)
        break; // (a coefficient that differs from cell to cell: nothing to share)
    }
    if ((int64_t)reps.size() * 2 <= n_all)
    {
      shared_solves = true;
      d_list.upload(reps.data(), reps.size(), st);
      d_rep_of.upload(rep_of.data(), rep_of.size(), st);
      a.list = d_list.data();
      a.n_agg = (int64_t)reps.size();
    }
  }
  const int wpb = nmax == 27 ? 4 : 1;
  const size_t lds = (size_t)wpb * (2 * nmax * nmax + 5 * nmax) * sizeof(double);
  const unsigned int blocks = (unsigned int)std::min<int64_t>((a.n_agg + wpb - 1) / wpb, 256 * 16);
  if (nmax == 27)
  {
    MFMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(amge_agglomerate_kernel<27>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(amge_agglomerate_kernel<27>, dim3(blocks), dim3(64 * wpb), lds, st, a);
  }
  else
  {
    MFMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(amge_agglomerate_kernel<64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(amge_agglomerate_kernel<64>, dim3(blocks), dim3(64 * wpb), lds, st, a);
  }
  MFMG_HIP_CHECK(hipGetLastError());
  if (shared_solves)
  {
    hipLaunchKernelGGL(amge_spread_kernel, dim3(n_blocks_for(n_all * a.n_eig * nmax, 256, 1 << 16)), dim3(256), 0, st, n_all,
                       a.n_eig * nmax, d_rep_of.data(), d_w.data(), d_nv.data());
    MFMG_HIP_CHECK(hipGetLastError());
  }
  weights = d_w.download(st);
  n_vec = d_nv.download(st);
}
} // namespace mfmg
