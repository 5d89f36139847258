// gfx950 kernels of the agglomerate-wise restrictor (structured_restrictor.hpp).
//
// restrict : one thread per coarse row r = a E + e.  Consecutive lanes are consecutive rows, so every
//            plane is read as one contiguous run; the x entries of the patch are shared by the E rows of an
//            agglomerate and by neighbouring agglomerates (L1/L2).
// prolong  : one thread per fine node, owner computes (no atomics, fixed summation order).  A node lies
//            in at most two agglomerates per direction: the one it starts (position m = i mod a) and, on an
//            agglomerate boundary, the previous one (position m = a).  Lanes of even / odd nodes read two
//            planes at the rows of consecutive agglomerates; the two eigenvector passes use both halves of
//            every sector.
#include "structured_restrictor.hpp"

#include <omp.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <unordered_map>

namespace mfmg
{
namespace
{
struct SrArgs
{
  double const *planes;
  float const *planes_f; // the same planes kept in float (every value representable in it); then `planes` is null
  int32_t const *node_dof; // nullptr: identity
  int64_t n_coarse;
  int N[3], na[3], a[3];
  int n_eig, patch;
  // agglomerates whose n_eig x patch block repeats a reference block bit for bit (interior agglomerates of a
  // constant-coefficient problem) are evaluated from `table[m * n_eig + e]`; nullptr: none
  uint8_t const *exc;
  uint8_t const *exc_node; // per fine node: 0 = every agglomerate it lies in is regular
  double const *table;
  // the other agglomerates: class of the block they repeat (0xffff: a block of their own, read from the planes)
  uint16_t const *cls;
  double const *class_table; // [class][patch][n_eig]
};

// plane entry / pair of plane entries (the two eigenvectors of an agglomerate) from whichever storage the planes have
__device__ __forceinline__ double sr_plane(SrArgs const &s, size_t idx)
{
  return s.planes_f != nullptr ? (double)s.planes_f[idx] : s.planes[idx];
}
__device__ __forceinline__ double2 sr_plane_pair(SrArgs const &s, size_t pair_idx)
{
  if (s.planes_f != nullptr)
  {
    const float2 v = reinterpret_cast<float2 const *>(s.planes_f)[pair_idx];
    return make_double2((double)v.x, (double)v.y);
  }
  return reinterpret_cast<double2 const *>(s.planes)[pair_idx];
}

__global__ __launch_bounds__(256) void sr_restrict_kernel(SrArgs s, double const *x, double *y)
{
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= s.n_coarse)
    return;
  const int64_t ag = r / s.n_eig;
  const int ai = ag % s.na[0], aj = (ag / s.na[0]) % s.na[1], ak = ag / ((int64_t)s.na[0] * s.na[1]);
  const int64_t base = (int64_t)ai * s.a[0] + (int64_t)s.N[0] * ((int64_t)aj * s.a[1] + (int64_t)s.N[1] * ((int64_t)ak * s.a[2]));
  double sum = 0.;
  int m = 0;
  for (int mz = 0; mz <= s.a[2]; ++mz)
    for (int my = 0; my <= s.a[1]; ++my)
    {
      const int64_t row = base + (int64_t)s.N[0] * (my + (int64_t)s.N[1] * mz);
#pragma unroll 3
      for (int mx = 0; mx <= s.a[0]; ++mx, ++m)
      {
        const int64_t node = row + mx;
        const int64_t id = s.node_dof ? (int64_t)s.node_dof[node] : node;
        sum += sr_plane(s, (size_t)m * s.n_coarse + r) * x[id];
      }
    }
  y[r] = sum;
}

// two eigenvectors per agglomerate: one thread per agglomerate, both rows at once (the x values of the patch are
// fetched once, the two plane entries in one 16-byte request); the sums are formed as in the row kernel
// A = 2: 2 x 2 x 2 agglomerates with the loops unrolled, so that the 27 independent requests of a thread are in
// flight together (with run-time bounds they went out three at a time: latency, not bytes); A = 0: any size
template <int A>
__global__ __launch_bounds__(256) void sr_restrict_pair_kernel(SrArgs s, double const *x, double *y)
{
  const int a0 = A > 0 ? A : s.a[0], a1 = A > 0 ? A : s.a[1], a2 = A > 0 ? A : s.a[2];
  // workgroups are dealt to the 8 XCDs in turn: give every XCD a contiguous run of agglomerates, so that the node
  // planes two layers of agglomerates share stay in one L2 (the grid is a multiple of 8)
  const int64_t bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t ag = bid * (int64_t)blockDim.x + threadIdx.x;
  if (2 * ag >= s.n_coarse)
    return;
  const int ai = ag % s.na[0], aj = (ag / s.na[0]) % s.na[1], ak = ag / ((int64_t)s.na[0] * s.na[1]);
  const int64_t base = (int64_t)ai * a0 + (int64_t)s.N[0] * ((int64_t)aj * a1 + (int64_t)s.N[1] * ((int64_t)ak * a2));
  const size_t stride = (size_t)s.n_coarse / 2;
  const bool regular = s.exc != nullptr && s.exc[ag] == 0;
  // a block shared with other agglomerates comes from the class table (cached) instead of the planes (HBM)
  const unsigned int cl = (!regular && s.cls != nullptr) ? s.cls[ag] : 0xffffu;
  double2 const *ct = reinterpret_cast<double2 const *>(s.class_table) + (size_t)(cl == 0xffffu ? 0 : cl) * s.patch;
  double sum0 = 0., sum1 = 0.;
  if constexpr (A == 2)
  {
    // all 27 x requests first (they do not depend on where the weights come from), then one loop per source
    double xv[27];
#pragma unroll
    for (int m = 0; m < 27; ++m)
    {
      const int64_t node = base + (m % 3) + (int64_t)s.N[0] * ((m / 3) % 3 + (int64_t)s.N[1] * (m / 9));
      xv[m] = x[s.node_dof ? (int64_t)s.node_dof[node] : node];
    }
    if (regular)
    {
#pragma unroll
      for (int m = 0; m < 27; ++m)
      {
        sum0 += s.table[2 * m] * xv[m]; // wave-uniform addresses
        sum1 += s.table[2 * m + 1] * xv[m];
      }
    }
    else if (cl != 0xffffu)
    {
#pragma unroll
      for (int m = 0; m < 27; ++m)
      {
        const double2 pv = ct[m];
        sum0 += pv.x * xv[m];
        sum1 += pv.y * xv[m];
      }
    }
    else
    {
#pragma unroll
      for (int m = 0; m < 27; ++m)
      {
        const double2 pv = sr_plane_pair(s, (size_t)m * stride + ag);
        sum0 += pv.x * xv[m];
        sum1 += pv.y * xv[m];
      }
    }
  }
  else
  {
    int m = 0;
    for (int mz = 0; mz <= a2; ++mz)
      for (int my = 0; my <= a1; ++my)
      {
        const int64_t row = base + (int64_t)s.N[0] * (my + (int64_t)s.N[1] * mz);
#pragma unroll 3
        for (int mx = 0; mx <= a0; ++mx, ++m)
        {
          const int64_t node = row + mx;
          const int64_t id = s.node_dof ? (int64_t)s.node_dof[node] : node;
          double2 pv;
          if (regular)
            pv = make_double2(s.table[2 * m], s.table[2 * m + 1]); // wave-uniform address
          else if (cl != 0xffffu)
            pv = ct[m];
          else
            pv = sr_plane_pair(s, (size_t)m * stride + ag);
          const double xv = x[id];
          sum0 += pv.x * xv;
          sum1 += pv.y * xv;
        }
      }
  }
  reinterpret_cast<double2 *>(y)[ag] = make_double2(sum0, sum1);
}

// (R^T y) at fine node `node`: a node lies in at most two agglomerates per direction, the one it starts
// (position m = i mod a) and, on an agglomerate boundary, the previous one (position m = a); fixed order
// (z, y, x candidates, then eigenvectors)
__device__ __forceinline__ double sr_node_value(SrArgs const &s, double const *y, int64_t node)
{
  const int i = node % s.N[0], j = (node / s.N[0]) % s.N[1], k = node / ((int64_t)s.N[0] * s.N[1]);
  // candidate agglomerates per direction: c = 0 the one the node starts, c = 1 the previous one
  int ax[2], mx[2], ay[2], my[2], az[2], mz[2];
  auto candidates = [](int idx, int a, int na, int ag[2], int mm[2]) {
    ag[0] = idx / a;
    mm[0] = idx % a;
    if (ag[0] >= na)
      ag[0] = -1;
    ag[1] = (idx % a == 0 && idx / a >= 1) ? idx / a - 1 : -1;
    mm[1] = a;
  };
  candidates(i, s.a[0], s.na[0], ax, mx);
  candidates(j, s.a[1], s.na[1], ay, my);
  candidates(k, s.a[2], s.na[2], az, mz);
  const int px = s.a[0] + 1, py = s.a[1] + 1;
  const bool table = s.exc_node != nullptr && s.exc_node[node] == 0; // all agglomerates around regular (n_eig = 2)
  double sum = 0.;
  for (int cz = 0; cz < 2; ++cz)
  {
    if (az[cz] < 0)
      continue;
    for (int cy = 0; cy < 2; ++cy)
    {
      if (ay[cy] < 0)
        continue;
      for (int cx = 0; cx < 2; ++cx)
      {
        if (ax[cx] < 0)
          continue;
        const int64_t ag = ax[cx] + (int64_t)s.na[0] * (ay[cy] + (int64_t)s.na[1] * az[cz]);
        const int m = mx[cx] + px * (my[cy] + py * mz[cz]);
        const size_t p0 = (size_t)m * s.n_coarse + ag * s.n_eig;
        double const *yy = y + ag * s.n_eig;
        if (s.n_eig == 2)
        {
          // both eigenvectors of the agglomerate in one 16-byte request (rows 2 ag, 2 ag + 1 are adjacent)
          const double2 yv = *reinterpret_cast<double2 const *>(yy);
          double2 pv;
          if (table)
            pv = *reinterpret_cast<double2 const *>(s.table + 2 * m);
          else
          {
            const unsigned int cl = s.cls != nullptr ? s.cls[ag] : 0xffffu;
            pv = cl != 0xffffu ? reinterpret_cast<double2 const *>(s.class_table)[(size_t)cl * s.patch + m]
                               : sr_plane_pair(s, p0 / 2);
          }
          sum += pv.x * yv.x;
          sum += pv.y * yv.y;
        }
        else
          for (int e = 0; e < s.n_eig; ++e)
            sum += sr_plane(s, p0 + e) * yy[e];
      }
    }
  }
  return sum;
}

// one thread per fine node (any agglomerate size, any numbering)
__global__ __launch_bounds__(256) void sr_prolong_kernel(SrArgs s, double const *y, double *out, int subtract)
{
  const int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t n_nodes = (int64_t)s.N[0] * s.N[1] * s.N[2];
  if (node >= n_nodes)
    return;
  const double sum = sr_node_value(s, y, node);
  const int64_t id = s.node_dof ? (int64_t)s.node_dof[node] : node;
  out[id] = subtract ? out[id] - sum : sum;
}

// 2 x 2 x 2 agglomerates, two eigenvectors, lexicographic numbering, table-driven blocks: one thread per
// agglomerate position (na + 1 per direction) finishes the 2 x 2 x 2 nodes at the low corner of its agglomerate.
// The y pairs of the (at most) eight agglomerates around are fetched once for the eight nodes, the table
// entries are wave-uniform; every sum is formed in the order of sr_node_value (same bits).
struct __attribute__((aligned(8))) sr_pair
{
  double x, y;
};

// Both parts in ONE launch: the first `listed_blocks` workgroups take the nodes of the listed agglomerate positions (the
// blocks the table-driven part leaves out: a thread per node, latency-bound, so they start first and run beside the
// rest), the others one agglomerate position per thread.  The two nodes 2 i, 2 i + 1 of a row are read / written as one
// 16-byte access: consecutive lanes, consecutive addresses (with 8-byte accesses at a stride of 16 every request used
// half of what it touched).
// SUB (x -= R^T y, the cycle's form) is a template parameter: as a run-time flag every read of `out` sat in a block of its own
// behind a uniform branch with a wait for ALL outstanding loads behind it -- four round trips in a row -- and the 54 table entries
// were fetched by per-lane vector loads with three more waits (the ISA of round 4's kernel; a wavefront lived 11 us for 24 memory
// instructions).  The table is read through the constant address space: scalar loads, whatever the stores of the kernel are.
template <bool SUB>
__global__ __launch_bounds__(256) void sr_prolong_block222_kernel(SrArgs s, double const *y, double *out,
                                                                  uint8_t const *blk_exc, int32_t const *blocks, int64_t n_blocks,
                                                                  unsigned int listed_blocks)
{
  constexpr int subtract = SUB ? 1 : 0;
  typedef __attribute__((address_space(4))) const double ctable_t;
  ctable_t *table = reinterpret_cast<ctable_t *>(reinterpret_cast<uintptr_t>(s.table));
  if (blockIdx.x < listed_blocks)
  {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= 8 * n_blocks)
      return;
    const int64_t v = blocks[t >> 3];
    const int d = (int)(t & 7);
    const int vx = s.na[0] + 1, vy = s.na[1] + 1;
    const int i = 2 * (int)(v % vx) + (d & 1), j = 2 * (int)((v / vx) % vy) + ((d >> 1) & 1),
              k = 2 * (int)(v / ((int64_t)vx * vy)) + (d >> 2);
    if (i >= s.N[0] || j >= s.N[1] || k >= s.N[2])
      return;
    const int64_t node = i + (int64_t)s.N[0] * (j + (int64_t)s.N[1] * k);
    const double sum = sr_node_value(s, y, node);
    out[node] = subtract ? out[node] - sum : sum;
    return;
  }
  const int64_t t = (int64_t)(blockIdx.x - listed_blocks) * blockDim.x + threadIdx.x;
  const int vx = s.na[0] + 1, vy = s.na[1] + 1, vz = s.na[2] + 1;
  if (t >= (int64_t)vx * vy * vz || blk_exc[t] != 0)
    return;
  const int vi = t % vx, vj = (t / vx) % vy, vk = t / ((int64_t)vx * vy);
  // Every request of the thread is issued UNCONDITIONALLY, at a clamped position, and masked afterwards: with a branch around each
  // of them (`has ? y[..] : 0`, `if (live) ...`) every load sat in a basic block of its own with its wait behind it -- twelve
  // dependent round trips per wavefront (the ISA: 127 exec branches, 105 waits; the counters: 1.5 MB in flight, profiles/r04_n).
  double2 yv[8];
  bool has[8];
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx)
  {
    const int qi = vi - (sidx & 1), qj = vj - ((sidx >> 1) & 1), qk = vk - (sidx >> 2);
    has[sidx] = qi >= 0 && qi < s.na[0] && qj >= 0 && qj < s.na[1] && qk >= 0 && qk < s.na[2];
    const int ci = min(max(qi, 0), s.na[0] - 1), cj = min(max(qj, 0), s.na[1] - 1), ck = min(max(qk, 0), s.na[2] - 1);
    yv[sidx] = reinterpret_cast<double2 const *>(y)[ci + (int64_t)s.na[0] * (cj + (int64_t)s.na[1] * ck)];
  }
  const bool pair = 2 * vi + 1 < s.N[0];
  // the four rows this thread finishes: all four reads of `out` are requested up front, together with the y pairs above
  // (row by row each read waited behind the store of the row before: out may alias out, four dependent round trips).  The last
  // position of a row (one node, no pair) reads the 16 bytes that END at its node.
  bool live[4];
  double *orow[4];
  sr_pair ov[4];
#pragma unroll
  for (int dyz = 0; dyz < 4; ++dyz)
  {
    const int j = 2 * vj + (dyz & 1), k = 2 * vk + (dyz >> 1);
    live[dyz] = j < s.N[1] && k < s.N[2];
    orow[dyz] = out + 2 * vi + (int64_t)s.N[0] * (min(j, s.N[1] - 1) + (int64_t)s.N[1] * min(k, s.N[2] - 1));
    ov[dyz] = sr_pair{0., 0.};
    if (subtract)
    {
      const sr_pair t = *reinterpret_cast<sr_pair const *>(orow[dyz] - (pair ? 0 : 1));
      ov[dyz].x = pair ? t.x : t.y;
      ov[dyz].y = pair ? t.y : 0.;
    }
  }
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx)
    if (!has[sidx])
      yv[sidx] = make_double2(0., 0.); // (an agglomerate outside the mesh contributes exact zeros below)
#pragma unroll
  for (int dyz = 0; dyz < 4; ++dyz)
  {
    const int dy = dyz & 1, dz = dyz >> 1;
    if (!live[dyz])
      continue;
    double sums[2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
    {
      double sum = 0.; // (formed in the order of sr_node_value: same bits)
#pragma unroll
      for (int sz = 0; sz < 2; ++sz)
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
          {
            if ((sx && dx) || (sy && dy) || (sz && dz))
              continue; // the previous agglomerate holds the node only on the shared boundary
            const int sidx = sx + 2 * sy + 4 * sz;
            const int m = (sx ? 2 : dx) + 3 * ((sy ? 2 : dy) + 3 * (sz ? 2 : dz));
            sum += table[2 * m] * yv[sidx].x;
            sum += table[2 * m + 1] * yv[sidx].y;
          }
      sums[dx] = sum;
    }
    sr_pair v = ov[dyz];
    v.x = subtract ? v.x - sums[0] : sums[0];
    v.y = subtract ? v.y - sums[1] : sums[1];
    if (pair)
      *reinterpret_cast<sr_pair *>(orow[dyz]) = v;
    else
      orow[dyz][0] = v.x;
  }
}
} // namespace

std::shared_ptr<StructuredRestrictorDevice>
StructuredRestrictorDevice::create(HipHandle &handle, StructuredMesh const &mesh, int const agglomerate[3],
                                   int const agg_dims[3], std::vector<int32_t> const &row_agglomerate,
                                   HostCsr const &R)
{
  MemoryKind kind("restrictor: planes, block tables, node lists");
  if (mesh.dim != 3 || R.n_rows == 0 || (int64_t)row_agglomerate.size() != R.n_rows)
    return nullptr;
  int64_t n_agg = 1, n_nodes = 1;
  for (int d = 0; d < 3; ++d)
  {
    if (agglomerate[d] < 1 || mesh.n[d] % agglomerate[d] != 0 || agg_dims[d] != mesh.n[d] / agglomerate[d])
      return nullptr; // clipped agglomerates have smaller patches
    n_agg *= agg_dims[d];
    n_nodes *= mesh.N[d];
  }
  if (R.n_rows % n_agg != 0 || R.n_cols != mesh.n_dofs || n_nodes != mesh.n_dofs)
    return nullptr;
  const int n_eig = (int)(R.n_rows / n_agg);
  for (int64_t r = 0; r < R.n_rows; ++r)
    if (row_agglomerate[r] != r / n_eig)
      return nullptr;
  const int px = agglomerate[0] + 1, py = agglomerate[1] + 1, pz = agglomerate[2] + 1;
  const int patch = px * py * pz;
  if ((int64_t)mesh.node_dof.size() != n_nodes)
    return nullptr;
  // node of every DoF
  std::vector<int32_t> dof_node(mesh.n_dofs, -1);
  bool identity = true;
  for (int64_t nd = 0; nd < n_nodes; ++nd)
  {
    const int32_t g = mesh.node_dof[nd];
    if (g < 0 || g >= mesh.n_dofs)
      return nullptr;
    dof_node[g] = (int32_t)nd;
    identity = identity && (g == nd);
  }
  std::vector<double> planes((size_t)patch * R.n_rows, 0.);
  bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
  for (int64_t r = 0; r < R.n_rows; ++r)
  {
    const int64_t ag = r / n_eig;
    const int ai = (int)(ag % agg_dims[0]), aj = (int)((ag / agg_dims[0]) % agg_dims[1]),
              ak = (int)(ag / ((int64_t)agg_dims[0] * agg_dims[1]));
    for (int p = R.row_ptr[r]; p < R.row_ptr[r + 1]; ++p)
    {
      const int32_t nd = dof_node[R.col[p]];
      if (nd < 0)
      {
        ok = false;
        continue;
      }
      const int i = nd % mesh.N[0], j = (nd / mesh.N[0]) % mesh.N[1], k = nd / (mesh.N[0] * mesh.N[1]);
      const int mx = i - ai * agglomerate[0], my = j - aj * agglomerate[1], mz = k - ak * agglomerate[2];
      if (mx < 0 || mx >= px || my < 0 || my >= py || mz < 0 || mz >= pz)
      {
        ok = false;
        continue;
      }
      planes[(size_t)(mx + px * (my + py * mz)) * R.n_rows + r] += R.val[p];
    }
  }
  if (!ok)
    return nullptr;
  std::shared_ptr<StructuredRestrictorDevice> s(new StructuredRestrictorDevice(handle));
  for (int d = 0; d < 3; ++d)
  {
    s->_N[d] = mesh.N[d];
    s->_na[d] = agg_dims[d];
    s->_a[d] = agglomerate[d];
  }
  s->_n_eig = n_eig;
  s->_patch = patch;
  s->_n_coarse = R.n_rows;
  s->_n_fine = mesh.n_dofs;
  s->_nnz = R.row_ptr[R.n_rows];
  s->_identity_numbering = identity;
  {
    // planes whose values a float holds exactly ("setup value precision" float rounds R) are kept in float
    bool all_float = !planes.empty();
    // (a sample first: FP64 values fail at once, and the full pass over 0.9 GB is spent only on candidates)
    for (int64_t q = 0; q < (int64_t)planes.size() && all_float; q += std::max<int64_t>(1, (int64_t)planes.size() / 4099))
      all_float = (double)(float)planes[q] == planes[q];
    if (all_float)
    {
#pragma omp parallel for schedule(static) reduction(&& : all_float)
      for (int64_t q = 0; q < (int64_t)planes.size(); ++q)
        all_float = all_float && (double)(float)planes[q] == planes[q];
    }
    if (all_float && n_eig % 2 == 0)
    {
      std::vector<float> pf(planes.size());
#pragma omp parallel for schedule(static)
      for (int64_t q = 0; q < (int64_t)planes.size(); ++q)
        pf[q] = (float)planes[q];
      s->_planes_f32.upload(pf.data(), pf.size(), handle.stream);
    }
    else
      s->_planes.upload(planes.data(), planes.size(), handle.stream);
  }
  if (n_eig == 2 && identity && agglomerate[0] == 2 && agglomerate[1] == 2 && agglomerate[2] == 2)
  {
    // class of EVERY agglomerate by the bits of its block (for the residual restriction, residual_restriction.hip);
    // given up beyond 4096 classes (a coefficient that varies from cell to cell: no two blocks alike)
    std::vector<uint64_t> hash(n_agg);
#pragma omp parallel for schedule(static)
    for (int64_t ag = 0; ag < n_agg; ++ag)
    {
      uint64_t h = 1469598103934665603ull;
      for (int m = 0; m < patch; ++m)
        for (int e = 0; e < n_eig; ++e)
        {
          uint64_t bits;
          std::memcpy(&bits, &planes[(size_t)m * R.n_rows + ag * n_eig + e], 8);
          h = (h ^ bits) * 1099511628211ull;
          h ^= h >> 29;
        }
      hash[ag] = h;
    }
    auto same_block = [&](int64_t a1, int64_t a2) {
      for (int m = 0; m < patch; ++m)
        for (int e = 0; e < n_eig; ++e)
          if (planes[(size_t)m * R.n_rows + a1 * n_eig + e] != planes[(size_t)m * R.n_rows + a2 * n_eig + e])
            return false;
      return true;
    };
    std::vector<int64_t> first;
    std::vector<uint16_t> all(n_agg);
    bool fits = true;
    // Classes numbered in the order of their first agglomerate.  In parallel: the first agglomerate of every hash value (per
    // thread over its contiguous share, merged), classes in the order of those, then every agglomerate compared with the
    // representative of its hash; a hash shared by two different blocks sends the whole step to the serial loop below,
    // which compares block by block -- the same classes either way.
    bool parallel_done = false;
    {
      const int nt = std::max(1, omp_get_max_threads());
      std::vector<std::unordered_map<uint64_t, int64_t>> local(nt);
      bool too_many = false;
#pragma omp parallel num_threads(nt)
      {
        const int t = omp_get_thread_num();
        const int64_t lo = n_agg * t / nt, hi = n_agg * (t + 1) / nt;
        auto &m = local[t];
        for (int64_t ag = lo; ag < hi; ++ag)
        {
          m.emplace(hash[ag], ag); // (keeps the first)
          if (m.size() > 4096)
          {
#pragma omp atomic write
            too_many = true;
            break;
          }
        }
      }
      if (too_many)
        fits = false, parallel_done = true; // (more than 4096 different blocks: the serial loop would give up as well)
      else
      {
        std::unordered_map<uint64_t, int64_t> firsts;
        for (auto const &m : local)
          for (auto const &kv : m)
          {
            auto it = firsts.find(kv.first);
            if (it == firsts.end())
              firsts.emplace(kv.first, kv.second);
            else if (kv.second < it->second)
              it->second = kv.second;
          }
        if (firsts.size() > 4096)
          fits = false, parallel_done = true;
        else
        {
          std::vector<std::pair<int64_t, uint64_t>> order;
          for (auto const &kv : firsts)
            order.emplace_back(kv.second, kv.first);
          std::sort(order.begin(), order.end());
          std::unordered_map<uint64_t, int> class_of_hash;
          for (size_t c = 0; c < order.size(); ++c)
          {
            class_of_hash.emplace(order[c].second, (int)c);
            first.push_back(order[c].first);
          }
          bool exact = true;
#pragma omp parallel for schedule(static) reduction(&& : exact)
          for (int64_t ag = 0; ag < n_agg; ++ag)
          {
            const int c = class_of_hash.find(hash[ag])->second;
            all[ag] = (uint16_t)c;
            exact = exact && same_block(first[c], ag);
          }
          parallel_done = exact;
          if (!exact)
            first.clear();
        }
      }
    }
    if (!parallel_done)
    {
    std::unordered_map<uint64_t, std::vector<int>> by_hash; // hash -> classes with that hash
    for (int64_t ag = 0; ag < n_agg && fits; ++ag)
    {
      auto &cands = by_hash[hash[ag]];
      int found = -1;
      for (int c : cands)
        if (same_block(first[c], ag))
        {
          found = c;
          break;
        }
      if (found < 0)
      {
        if (first.size() >= 4096)
        {
          fits = false;
          break;
        }
        found = (int)first.size();
        first.push_back(ag);
        cands.push_back(found);
      }
      all[ag] = (uint16_t)found;
    }
    }
    if (fits)
      s->_cls_host = std::move(all);
  }
  if (n_eig == 2)
  {
    // reference agglomerate: of a few candidates the one a sample of agglomerates repeats most
    auto same = [&](int64_t a1, int64_t a2) {
      for (int m = 0; m < patch; ++m)
        for (int e = 0; e < n_eig; ++e)
          if (planes[(size_t)m * R.n_rows + a1 * n_eig + e] != planes[(size_t)m * R.n_rows + a2 * n_eig + e])
            return false;
      return true;
    };
    int64_t ref = n_agg / 2, best = -1;
    const double frac[] = {0.5, 0.377, 0.613, 0.431, 0.569, 0.289, 0.711, 0.457};
    for (double f : frac)
    {
      const int64_t cand = std::min<int64_t>(n_agg - 1, (int64_t)(f * n_agg) + 12345 % std::max<int64_t>(n_agg / 7, 1));
      int64_t hits = 0;
      for (int64_t t = 0; t < 2048; ++t)
        hits += same((t * 2654435761ll) % n_agg, cand) ? 1 : 0;
      if (hits > best)
      {
        best = hits;
        ref = cand;
      }
    }
    std::vector<uint8_t> exc(n_agg);
    int64_t n_regular = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_regular)
    for (int64_t ag = 0; ag < n_agg; ++ag)
    {
      exc[ag] = same(ag, ref) ? 0 : 1;
      n_regular += exc[ag] ? 0 : 1;
    }
    if (n_regular * 2 >= n_agg)
    {
      // classes among the others (hash of the block, exact comparison with the first agglomerate of the class)
      {
        std::vector<int64_t> others;
        for (int64_t ag = 0; ag < n_agg; ++ag)
          if (exc[ag])
            others.push_back(ag);
        std::vector<uint64_t> hash(others.size());
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < (int64_t)others.size(); ++q)
        {
          uint64_t h = 1469598103934665603ull;
          for (int m = 0; m < patch; ++m)
            for (int e = 0; e < n_eig; ++e)
            {
              uint64_t bits;
              std::memcpy(&bits, &planes[(size_t)m * R.n_rows + others[q] * n_eig + e], 8);
              h = (h ^ bits) * 1099511628211ull;
              h ^= h >> 29;
            }
          hash[q] = h;
        }
        std::vector<int64_t> order(others.size());
        std::iota(order.begin(), order.end(), (int64_t)0);
        std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return hash[x] != hash[y] ? hash[x] < hash[y] : x < y; });
        std::vector<uint16_t> cls(n_agg, kNoClass);
        std::vector<double> class_table;
        auto push_block = [&](int64_t ag) {
          for (int m = 0; m < patch; ++m)
            for (int e = 0; e < n_eig; ++e)
              class_table.push_back(planes[(size_t)m * R.n_rows + ag * n_eig + e]);
        };
        push_block(ref); // class 0: the reference block
        for (int64_t ag = 0; ag < n_agg; ++ag)
          if (!exc[ag])
            cls[ag] = 0;
        int n_classes = 1;
        for (size_t g0 = 0; g0 < order.size() && n_classes < 0xfff0;)
        {
          size_t g1 = g0;
          while (g1 < order.size() && hash[order[g1]] == hash[order[g0]])
            ++g1;
          if (g1 - g0 >= 4)
          {
            const int64_t rep = others[order[g0]];
            int64_t members = 0;
            for (size_t q = g0; q < g1; ++q)
              if (same(others[order[q]], rep))
              {
                cls[others[order[q]]] = (uint16_t)n_classes;
                ++members;
              }
            if (members > 0)
            {
              push_block(rep);
              ++n_classes;
            }
          }
          g0 = g1;
        }
        if (n_classes > 1)
        {
          s->_cls.upload(cls.data(), cls.size(), handle.stream);
          s->_class_table.upload(class_table.data(), class_table.size(), handle.stream);
          s->_n_classes = n_classes - 1;
        }
      }
      std::vector<double> table((size_t)patch * n_eig);
      for (int m = 0; m < patch; ++m)
        for (int e = 0; e < n_eig; ++e)
          table[(size_t)m * n_eig + e] = planes[(size_t)m * R.n_rows + ref * n_eig + e];
      // per fine node: is one of the (at most eight) agglomerates it lies in not regular?
      std::vector<uint8_t> exc_node(n_nodes, 0);
#pragma omp parallel for schedule(static)
      for (int64_t nd = 0; nd < n_nodes; ++nd)
      {
        const int ijk[3] = {(int)(nd % mesh.N[0]), (int)((nd / mesh.N[0]) % mesh.N[1]),
                            (int)(nd / ((int64_t)mesh.N[0] * mesh.N[1]))};
        int lo[3], hi[3];
        for (int d = 0; d < 3; ++d)
        {
          const int q = ijk[d] / agglomerate[d];
          hi[d] = std::min(q, agg_dims[d] - 1);
          lo[d] = (ijk[d] % agglomerate[d] == 0 && q >= 1) ? q - 1 : std::min(q, agg_dims[d] - 1);
        }
        uint8_t any = 0;
        for (int c2 = lo[2]; c2 <= hi[2]; ++c2)
          for (int c1 = lo[1]; c1 <= hi[1]; ++c1)
            for (int c0 = lo[0]; c0 <= hi[0]; ++c0)
              any |= exc[c0 + (int64_t)agg_dims[0] * (c1 + (int64_t)agg_dims[1] * c2)];
        exc_node[nd] = any;
      }
      if (identity && agglomerate[0] == 2 && agglomerate[1] == 2 && agglomerate[2] == 2)
      {
        // agglomerate positions all of whose (existing) agglomerates around are regular -> block kernel
        const int64_t vx = agg_dims[0] + 1, vy = agg_dims[1] + 1, vz = agg_dims[2] + 1;
        std::vector<uint8_t> blk_exc((size_t)(vx * vy * vz), 0);
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < vx * vy * vz; ++t)
        {
          const int vi = (int)(t % vx), vj = (int)((t / vx) % vy), vk = (int)(t / (vx * vy));
          uint8_t any = 0;
          for (int sidx = 0; sidx < 8; ++sidx)
          {
            const int qi = vi - (sidx & 1), qj = vj - ((sidx >> 1) & 1), qk = vk - (sidx >> 2);
            if (qi >= 0 && qi < agg_dims[0] && qj >= 0 && qj < agg_dims[1] && qk >= 0 && qk < agg_dims[2])
              any |= exc[qi + (int64_t)agg_dims[0] * (qj + (int64_t)agg_dims[1] * qk)];
          }
          blk_exc[t] = any;
        }
        std::vector<int32_t> exc_blocks;
        for (int64_t t = 0; t < vx * vy * vz; ++t)
          if (blk_exc[t])
            exc_blocks.push_back((int32_t)t);
        s->_blk_exc.upload(blk_exc.data(), blk_exc.size(), handle.stream);
        s->_exc_blocks.upload(exc_blocks.data(), exc_blocks.size(), handle.stream);
      }
      s->_exc.upload(exc.data(), exc.size(), handle.stream);
      s->_exc_node.upload(exc_node.data(), exc_node.size(), handle.stream);
      s->_table.upload(table.data(), table.size(), handle.stream);
    }
  }
  if (!identity)
    s->_node_dof.upload(mesh.node_dof.data(), mesh.node_dof.size(), handle.stream);
  MFMG_HIP_CHECK(hipStreamSynchronize(handle.stream));
  return s;
}

double StructuredRestrictorDevice::algorithmic_bytes() const
{
  return double(_nnz) * 12. + 4. * double(_n_coarse + 1) + 8. * double(_n_fine) + 8. * double(_n_coarse);
}

namespace
{
SrArgs make_args(double const *planes, float const *planes_f, int32_t const *node_dof, int64_t n_coarse, int const N[3], int const na[3],
                 int const a[3], int n_eig, int patch, uint8_t const *exc, uint8_t const *exc_node, double const *table,
                 uint16_t const *cls, double const *class_table)
{
  SrArgs s;
  s.planes = planes;
  s.planes_f = planes_f;
  s.node_dof = node_dof;
  s.n_coarse = n_coarse;
  for (int d = 0; d < 3; ++d)
  {
    s.N[d] = N[d];
    s.na[d] = na[d];
    s.a[d] = a[d];
  }
  s.n_eig = n_eig;
  s.patch = patch;
  s.exc = exc;
  s.exc_node = exc_node;
  s.table = table;
  s.cls = cls;
  s.class_table = class_table;
  return s;
}
} // namespace

void StructuredRestrictorDevice::restrict_to_coarse(double const *x, double *y) const
{
  ASSERT_THROW(x != nullptr && y != nullptr && x != y, "bad vectors");
  SrArgs s = make_args(_planes.data(), _planes_f32.size() ? _planes_f32.data() : nullptr, _identity_numbering ? nullptr : _node_dof.data(), _n_coarse, _N, _na, _a,
                       _n_eig, _patch, _exc.size() ? _exc.data() : nullptr, _exc_node.size() ? _exc_node.data() : nullptr, _table.data(),
                       _cls.size() ? _cls.data() : nullptr, _class_table.data());
  hipEvent_t stop = _handle.profiler.begin("csr_spmv_kernel", algorithmic_bytes(), _handle.stream);
  if (_n_eig == 2)
  {
    const dim3 grid((unsigned int)(((_n_coarse / 2 + 255) / 256 + 7) / 8 * 8));
    if (_a[0] == 2 && _a[1] == 2 && _a[2] == 2)
      hipLaunchKernelGGL(sr_restrict_pair_kernel<2>, grid, dim3(256), 0, _handle.stream, s, x, y);
    else
      hipLaunchKernelGGL(sr_restrict_pair_kernel<0>, grid, dim3(256), 0, _handle.stream, s, x, y);
  }
  else
    hipLaunchKernelGGL(sr_restrict_kernel, dim3((unsigned int)((_n_coarse + 255) / 256)), dim3(256), 0,
                       _handle.stream, s, x, y);
  KernelProfiler::end(stop, _handle.stream);
  MFMG_HIP_CHECK(hipGetLastError());
}

void StructuredRestrictorDevice::prolongate(double const *y, double *out, bool subtract) const
{
  ASSERT_THROW(y != nullptr && out != nullptr && y != out, "bad vectors");
  SrArgs s = make_args(_planes.data(), _planes_f32.size() ? _planes_f32.data() : nullptr, _identity_numbering ? nullptr : _node_dof.data(), _n_coarse, _N, _na, _a,
                       _n_eig, _patch, _exc.size() ? _exc.data() : nullptr, _exc_node.size() ? _exc_node.data() : nullptr, _table.data(),
                       _cls.size() ? _cls.data() : nullptr, _class_table.data());
  hipEvent_t stop = _handle.profiler.begin("csr_spmv_kernel", algorithmic_bytes() + (subtract ? 8. * double(_n_fine) : 0.),
                                           _handle.stream);
  if (_blk_exc.size() > 0)
  {
    const int64_t n_pos = (int64_t)_blk_exc.size(), n_listed = (int64_t)_exc_blocks.size();
    const unsigned int listed_blocks = (unsigned int)((8 * n_listed + 255) / 256);
    const dim3 grid(listed_blocks + (unsigned int)((n_pos + 255) / 256));
    if (subtract)
      hipLaunchKernelGGL(sr_prolong_block222_kernel<true>, grid, dim3(256), 0, _handle.stream, s, y, out, _blk_exc.data(), _exc_blocks.data(),
                         n_listed, listed_blocks);
    else
      hipLaunchKernelGGL(sr_prolong_block222_kernel<false>, grid, dim3(256), 0, _handle.stream, s, y, out, _blk_exc.data(), _exc_blocks.data(),
                         n_listed, listed_blocks);
  }
  else
    hipLaunchKernelGGL(sr_prolong_kernel, dim3((unsigned int)((_n_fine + 255) / 256)), dim3(256), 0, _handle.stream,
                       s, y, out, subtract ? 1 : 0);
  KernelProfiler::end(stop, _handle.stream);
  MFMG_HIP_CHECK(hipGetLastError());
}
} // namespace mfmg
