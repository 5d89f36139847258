// b_c = R (A x - b) in one pass (structured_restrictor.hpp): the residual and its restriction of the V-cycle
// (include/mfmg/common/hierarchy.hpp:281-290) without the fine residual ever being stored.
//
// A row of R lives on the 3 x 3 x 3 nodes of a 2 x 2 x 2-cell agglomerate, a row of a Q1 operator on the 3 x 3 x 3
// nodes around its node: a row of R A lives on the 5 x 5 x 5 nodes around the agglomerate.  Where the fine operator
// repeats itself from agglomerate to agglomerate these 125 x 2 weights (two eigenvectors) depend only on the CLASS of
// the agglomerate (the block of R it repeats and its distance from the faces of the box), so that
//   b_c[agglomerate] = sum_125 W[class][m] x[node m] - sum_27 R[class][m] b[node m]
// needs x and b once (16 B per fine DoF + 8 B per coarse row) where residual + restriction move 40 B per fine DoF plus
// the operator's own data.  The weights are probed: W[class] = A (R^T e_row) for one representative per class.
//
// Kernel: a wavefront takes 64 consecutive agglomerates of one row (same j, k); a lane loads its two own x values of a
// node row as one 16-byte request -- consecutive lanes, consecutive addresses -- and takes the three others of its five
// from the neighbouring lanes; the weights are wave-uniform (scalar loads).  The first and last agglomerate of a row
// and rows whose agglomerates do not share a class go to a thread-per-agglomerate part in the same launch.
#include "structured_restrictor.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

namespace mfmg
{
namespace
{
constexpr int kFoot = 125, kOwn = 27, kTab = kFoot + kOwn; // double2 entries of one class
constexpr int kRun = 62; // agglomerates of one wavefront of the row-wise part (lanes 1 .. 62)

struct RrArgs
{
  double2 const *table;
  uint16_t const *seg_class;
  uint16_t const *cls;
  int32_t const *listed;
  int64_t n_listed;
  int64_t n_main_waves;
  unsigned int main_blocks, listed_blocks;
  int main_last; // last agglomerate index i of a row the row-wise part takes
  int N[3], na[3];
  int segs;
  // tile form: workgroups of kTileRows agglomerate rows marching through `ka` agglomerate layers
  int tiles_j, ka;
  int64_t n_tiles;
};

// two consecutive entries of x or b (FP64: 16 bytes at an 8-byte boundary; FP32, the fine level of apply_f32: 8 at a 4-byte one)
template <typename TI>
struct __attribute__((aligned(sizeof(TI)))) pair_of
{
  TI x, y;
};

// value held by the previous / next lane of the wavefront (DPP wave shift: a VALU move, no LDS crossbar traffic)
__device__ __forceinline__ int rr_dpp_prev(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int rr_dpp_next(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ double rr_prev(double v)
{
  return __hiloint2double(rr_dpp_prev(__double2hiint(v)), rr_dpp_prev(__double2loint(v)));
}
__device__ __forceinline__ double rr_next(double v)
{
  return __hiloint2double(rr_dpp_next(__double2hiint(v)), rr_dpp_next(__double2loint(v)));
}

// ---- the listed agglomerates (first in the grid: they overlap with the rest): sixteen lanes each, the 125 + 27 nodes
// dealt round the lanes, every bound checked; the requests of a lane are in flight together, then a 16-lane reduction ----
template <typename TI>
__device__ __forceinline__ void rr_listed_part(RrArgs const &s, TI const *__restrict__ x, TI const *__restrict__ b, double *__restrict__ y)
{
  const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  const bool live = q < s.n_listed;
  const int64_t ag = s.listed[live ? q : 0];
  const int ai = (int)(ag % s.na[0]), aj = (int)((ag / s.na[0]) % s.na[1]), ak = (int)(ag / ((int64_t)s.na[0] * s.na[1]));
  double2 const *t = s.table + (size_t)s.cls[ag] * kTab;
  double s0 = 0., s1 = 0.;
#pragma unroll
  for (int it = 0; it < (kTab + 15) / 16; ++it)
  {
    const int m = min(it * 16 + sub, kTab - 1);
    const bool foot = m < kFoot;
    const int mm = foot ? m : m - kFoot, n = foot ? 5 : 3, o = foot ? 1 : 0;
    const int gx = 2 * ai - o + mm % n, gy = 2 * aj - o + (mm / n) % n, gz = 2 * ak - o + mm / (n * n);
    const bool in = it * 16 + sub < kTab && gx >= 0 && gx < s.N[0] && gy >= 0 && gy < s.N[1] && gz >= 0 && gz < s.N[2];
    const int64_t node = ((int64_t)min(max(gz, 0), s.N[2] - 1) * s.N[1] + min(max(gy, 0), s.N[1] - 1)) * s.N[0] + min(max(gx, 0), s.N[0] - 1);
    const double v = foot ? (double)x[node] : -(double)b[node];
    const double2 w = t[m];
    s0 += in ? w.x * v : 0.;
    s1 += in ? w.y * v : 0.;
  }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1)
  {
    s0 += __shfl_xor(s0, d);
    s1 += __shfl_xor(s1, d);
  }
  if (live && sub == 0)
    reinterpret_cast<double2 *>(y)[ag] = make_double2(s0, s1);
}

template <int WAVES, typename TI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void
residual_restriction_kernel(RrArgs s, TI const *__restrict__ x, TI const *__restrict__ b, double *__restrict__ y)
{
  if (blockIdx.x >= s.listed_blocks)
  {
    // workgroups are dealt to the 8 XCDs in turn: a contiguous run of rows per XCD keeps the node rows that
    // neighbouring agglomerate rows share in one L2 (main_blocks is a multiple of 8)
    const unsigned int mb = blockIdx.x - s.listed_blocks;
    const int64_t bid = (int64_t)(mb & 7) * (s.main_blocks >> 3) + (mb >> 3);
    const int64_t wave = bid * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= s.n_main_waves)
      return;
    const unsigned int cw = s.seg_class[wave];
    if (cw == 0xffffu)
      return; // the agglomerates of this run are in the list
    const int c = __builtin_amdgcn_readfirstlane((int)cw);
    // wavefront order (k, run, j), j fastest: the four wavefronts of a workgroup take four consecutive rows j of the same
    // run, whose 5 node rows per layer overlap (11 distinct of 20: the vector cache serves the rest)
    const int aj = (int)(wave % s.na[1]);
    const int seg = (int)((wave / s.na[1]) % s.segs);
    const int ak = (int)(wave / ((int64_t)s.na[1] * s.segs));
    const int lane = threadIdx.x & 63;
    // lanes 1 .. 62 own the agglomerates i0 .. i0 + 61; lane 0 and lane 63 fetch the pair to the left / right of them
    const int i0 = 1 + seg * kRun, last = min(s.main_last, i0 + kRun - 1);
    const int ai = i0 + lane - 1;
    const int aic = min(ai, s.na[0] - 1);
    // (constant address space: the weights stay scalar loads whatever the compiler assumes about the asm statements below)
    using const_weights = __attribute__((address_space(4))) const double;
    const_weights *t = reinterpret_cast<const_weights *>(reinterpret_cast<uintptr_t>(s.table + (size_t)c * kTab));
    const int N0 = s.N[0], N1 = s.N[1], N2 = s.N[2];
    // node rows outside the box carry zero weights: their index is clamped instead of branched around, so that the
    // requests of a layer go out together (a branch per row made every row a dependent round trip); the five rows of
    // the next layer are requested before the current one is used
    // (32-bit index arithmetic: the vectors have fewer than 2^31 entries, checked on the host)
    int zo[5], yo[5];
#pragma unroll
    for (int m = 0; m < 5; ++m)
    {
      zo[m] = min(max(2 * ak - 1 + m, 0), N2 - 1) * N1;
      yo[m] = min(max(2 * aj - 1 + m, 0), N1 - 1);
    }
    TI const *xc = x + 2 * aic, *bc = b + 2 * aic;
    using pair_in = pair_of<TI>;
    using pair8 = pair_of<double>;
    auto load_layer = [&](pair8(&v)[5], int mz) {
#pragma unroll
      for (int my = 0; my < 5; ++my)
      {
        const pair_in p = *reinterpret_cast<pair_in const *>(xc + (zo[mz] + yo[my]) * N0);
        v[my] = pair8{(double)p.x, (double)p.y};
      }
    };
    auto load_b = [&](pair8(&v)[3], int mz) {
#pragma unroll
      for (int my = 0; my < 3; ++my)
      {
        const pair_in p = *reinterpret_cast<pair_in const *>(bc + (zo[mz + 1] + yo[my + 1]) * N0); // (rows 2 ak + mz, 2 aj + my: never clamped)
        v[my] = pair8{(double)p.x, (double)p.y};
      }
    };
    double s0 = 0., s1 = 0., r0 = 0., r1 = 0.;
    pair8 cur[5], nxt[5], bcur[3];
    load_layer(cur, 0);
#pragma unroll
    for (int mz = 0; mz < 5; ++mz)
    {
      if (mz < 4)
        load_layer(nxt, mz + 1);
      if (mz >= 1 && mz <= 3)
        load_b(bcur, mz - 1);
      __builtin_amdgcn_sched_barrier(0); // (keeps the scheduler from hoisting every request and every weight to the top)
#pragma unroll
      for (int my = 0; my < 5; ++my)
      {
        const pair8 own = cur[my];
        const double xm1 = rr_prev(own.y), xp2 = rr_next(own.x), xp3 = rr_next(own.y);
        const int m = (mz * 5 + my) * 5;
        s0 += t[2 * m] * xm1 + t[2 * m + 2] * own.x + t[2 * m + 4] * own.y + t[2 * m + 6] * xp2 + t[2 * m + 8] * xp3;
        s1 += t[2 * m + 1] * xm1 + t[2 * m + 3] * own.x + t[2 * m + 5] * own.y + t[2 * m + 7] * xp2 + t[2 * m + 9] * xp3;
        // the sums are stored under a condition at the very end, and the compiler would sink ALL the arithmetic into
        // that block (every weight parked in a VGPR lane meanwhile): pin them here, one row of weights at a time
        asm volatile("" : "+v"(s0), "+v"(s1));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mz >= 1 && mz <= 3)
      {
#pragma unroll
        for (int my = 0; my < 3; ++my)
        {
          const pair8 own = bcur[my];
          const double bp2 = rr_next(own.x);
          const int m = kFoot + ((mz - 1) * 3 + my) * 3;
          r0 += t[2 * m] * own.x + t[2 * m + 2] * own.y + t[2 * m + 4] * bp2;
          r1 += t[2 * m + 1] * own.x + t[2 * m + 3] * own.y + t[2 * m + 5] * bp2;
          asm volatile("" : "+v"(r0), "+v"(r1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int my = 0; my < 5; ++my)
        cur[my] = nxt[my];
    }
    if (lane >= 1 && lane <= kRun && ai <= last)
    {
      const int64_t ag = ai + (int64_t)s.na[0] * (aj + (int64_t)s.na[1] * ak);
      reinterpret_cast<double2 *>(y)[ag] = make_double2(s0 - r0, s1 - r1);
    }
    return;
  }
  rr_listed_part<TI>(s, x, b, y);
}

// ---- tile form of the row-wise part -------------------------------------------------------------------------------------
// The row-wise kernel above asks for 25 + 9 node rows per agglomerate row: every node row is requested (5/2)^2 times, by the
// wavefronts of neighbouring agglomerate rows and layers.  Here a workgroup of eight wavefronts owns eight consecutive
// agglomerate rows j and marches through `ka` agglomerate layers k: the 19 node rows of x (17 of b) that the eight share in
// a node layer are requested once, one 16-byte request per lane and row, and staged in LDS; an agglomerate's five node layers
// arrive in order, so three agglomerate layers are in flight per wavefront (the one that starts, the middle one, the one
// that ends with this node layer) and a node layer is requested once per workgroup instead of 2.5 times.  The sums of an
// agglomerate are formed in the order of the row-wise kernel (node layers, then node rows, five weights per row; b likewise):
// the same bits.  Two LDS buffers: the rows of node layer g + 1 are written while g is used, one barrier per layer.
constexpr int kTileRows = 8;                                 // wavefronts = agglomerate rows of a workgroup
constexpr int kTileX = 2 * kTileRows + 3, kTileB = 2 * kTileRows + 1; // node rows of x and of b per node layer
constexpr int kTileLds = kTileX + kTileB;                    // rows of 64 x 16 bytes per buffer

template <typename TI>
__global__ __launch_bounds__(64 * kTileRows) __attribute__((amdgpu_waves_per_eu(4, 4))) void
residual_restriction_tile_kernel(RrArgs s, TI const *__restrict__ x, TI const *__restrict__ b, double *__restrict__ y)
{
  if (blockIdx.x < s.listed_blocks)
  {
    rr_listed_part<TI>(s, x, b, y);
    return;
  }
  extern __shared__ double2 rr_rows[]; // [2][kTileLds][64]
  // a contiguous run of tiles per XCD (main_blocks is a multiple of 8): neighbouring tiles share node rows and layers
  const unsigned int mb = blockIdx.x - s.listed_blocks;
  const int64_t bid = (int64_t)(mb & 7) * (s.main_blocks >> 3) + (mb >> 3);
  if (bid >= s.n_tiles)
    return; // (the whole workgroup: no barrier is left waiting)
  const int seg = (int)(bid % s.segs);
  const int tj = (int)((bid / s.segs) % s.tiles_j);
  const int tk = (int)(bid / ((int64_t)s.segs * s.tiles_j));
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int aj0 = tj * kTileRows, aj = aj0 + w;
  const bool row_live = aj < s.na[1];
  const int ak0 = tk * s.ka, ak1 = min(ak0 + s.ka, s.na[2]);
  const int i0 = 1 + seg * kRun, last = min(s.main_last, i0 + kRun - 1);
  const int ai = i0 + lane - 1;
  const int aic = min(ai, s.na[0] - 1);
  const int N0 = s.N[0], N1 = s.N[1], N2 = s.N[2];
  using const_weights = __attribute__((address_space(4))) const double;
  using pair_in = pair_of<TI>;
  using pair8 = pair_of<double>;
  // class of this wavefront's agglomerate row in layer ak: -1 where there is nothing to compute (outside the tile, run in the list)
  auto class_of = [&](int ak) -> int {
    if (!row_live || ak < ak0 || ak >= ak1)
      return -1;
    const unsigned int cw = s.seg_class[((int64_t)ak * s.segs + seg) * s.na[1] + aj];
    return cw == 0xffffu ? -1 : __builtin_amdgcn_readfirstlane((int)cw);
  };
  auto weights = [&](int c) { return reinterpret_cast<const_weights *>(reinterpret_cast<uintptr_t>(s.table + (size_t)max(c, 0) * kTab)); };
  // rows this wavefront fetches for the workgroup: r = w + 8 q < 36; r < 19 is row 2 aj0 - 1 + r of x, else row 2 aj0 + r - 19 of b
  // (rows outside the box carry zero weights: clamped, as in the row-wise kernel)
  constexpr int kFetch = (kTileLds + kTileRows - 1) / kTileRows;
  int row_off[kFetch];
#pragma unroll
  for (int q = 0; q < kFetch; ++q)
  {
    const int r = min(w + kTileRows * q, kTileLds - 1); // (the last round has rows for the first wavefronts only: the others repeat row 35)
    const int yy = r < kTileX ? 2 * aj0 - 1 + r : 2 * aj0 + r - kTileX;
    row_off[q] = min(max(yy, 0), N1 - 1) * N0 + 2 * aic;
  }
  const int g0 = 2 * ak0 - 1, n_layers = 2 * (ak1 - ak0) + 3;
  pair8 pre[kFetch];
  // the requests of a node layer leave together, without a branch between them: a row of b in a layer whose b nobody needs
  // (the first and the last two of a tile) is asked of x instead, where the same row is in the tile anyway
  auto fetch = [&](int li) {
    const int g = g0 + li;
    const int zoff = min(max(g, 0), N2 - 1) * N1 * N0;
    const bool with_b = g >= 2 * ak0 && g <= 2 * ak1;
    TI const *bsrc = with_b ? b : x;
#pragma unroll
    for (int q = 0; q < kFetch; ++q)
    {
      const int r = w + kTileRows * q; // (wave-uniform)
      const pair_in p = *reinterpret_cast<pair_in const *>((r < kTileX ? x : bsrc) + (zoff + row_off[q]));
      pre[q] = pair8{(double)p.x, (double)p.y};
    }
  };
  auto stage = [&](int li) {
    double2 *buf = rr_rows + (size_t)(li & 1) * kTileLds * 64;
#pragma unroll
    for (int q = 0; q < kFetch; ++q)
    {
      const int r = w + kTileRows * q;
      if (kTileRows * q + kTileRows <= kTileLds || r < kTileLds)
        buf[r * 64 + lane] = make_double2(pre[q].x, pre[q].y);
    }
  };
  // three agglomerate layers in flight: P ends with the current odd node layer, C is in its middle, N starts
  double Ps0 = 0., Ps1 = 0., Pr0 = 0., Pr1 = 0., Cs0 = 0., Cs1 = 0., Cr0 = 0., Cr1 = 0., Ns0 = 0., Ns1 = 0.;
  int cP = -1, cC = -1, cN = -1;
  // five weights of one node row of the footprint on the lane's five nodes
  auto row_x = [&](double &s0, double &s1, const_weights *t, int mz, int my, double xm1, pair8 own, double xp2, double xp3) {
    const int m = (mz * 5 + my) * 5;
    s0 += t[2 * m] * xm1 + t[2 * m + 2] * own.x + t[2 * m + 4] * own.y + t[2 * m + 6] * xp2 + t[2 * m + 8] * xp3;
    s1 += t[2 * m + 1] * xm1 + t[2 * m + 3] * own.x + t[2 * m + 5] * own.y + t[2 * m + 7] * xp2 + t[2 * m + 9] * xp3;
    asm volatile("" : "+v"(s0), "+v"(s1));
  };
  auto row_b = [&](double &r0, double &r1, const_weights *t, int mz, int my, pair8 own, double bp2) {
    const int m = kFoot + (mz * 3 + my) * 3;
    r0 += t[2 * m] * own.x + t[2 * m + 2] * own.y + t[2 * m + 4] * bp2;
    r1 += t[2 * m + 1] * own.x + t[2 * m + 3] * own.y + t[2 * m + 5] * bp2;
    asm volatile("" : "+v"(r0), "+v"(r1));
  };
  fetch(0);
  stage(0);
  if (n_layers > 1)
    fetch(1);
  __syncthreads();
  cN = class_of(ak0);
  for (int li = 0; li < n_layers; ++li)
  {
    // the rows of the next node layer go to the other buffer (last read a layer ago, before the barrier), the requests of
    // the one after leave before this layer's arithmetic
    if (li + 1 < n_layers)
      stage(li + 1);
    if (li + 2 < n_layers)
      fetch(li + 2);
    double2 const *buf = rr_rows + (size_t)(li & 1) * kTileLds * 64;
    const bool odd = (li & 1) == 0; // node layer g0 + li = 2 a + 1
    const bool any = odd ? (cP >= 0 || cC >= 0 || cN >= 0) : (cP >= 0 || cC >= 0);
    if (any)
    {
      const_weights *tP = weights(cP), *tC = weights(cC), *tN = weights(cN);
      // the rows of this wavefront leave the LDS together, before the first of them is used
      double2 xr[5], br[3];
#pragma unroll
      for (int my = 0; my < 5; ++my)
        xr[my] = buf[(2 * w + my) * 64 + lane];
#pragma unroll
      for (int my = 0; my < 3; ++my)
        br[my] = buf[(kTileX + 2 * w + my) * 64 + lane];
#pragma unroll
      for (int my = 0; my < 5; ++my)
      {
        const double2 v = xr[my];
        const pair8 own{v.x, v.y};
        const double xm1 = rr_prev(own.y), xp2 = rr_next(own.x), xp3 = rr_next(own.y);
        if (odd)
        {
          if (cN >= 0)
            row_x(Ns0, Ns1, tN, 0, my, xm1, own, xp2, xp3);
          if (cC >= 0)
            row_x(Cs0, Cs1, tC, 2, my, xm1, own, xp2, xp3);
          if (cP >= 0)
            row_x(Ps0, Ps1, tP, 4, my, xm1, own, xp2, xp3);
        }
        else
        {
          if (cC >= 0)
            row_x(Cs0, Cs1, tC, 1, my, xm1, own, xp2, xp3);
          if (cP >= 0)
            row_x(Ps0, Ps1, tP, 3, my, xm1, own, xp2, xp3);
        }
      }
      if (odd ? cC >= 0 : (cC >= 0 || cP >= 0))
      {
#pragma unroll
        for (int my = 0; my < 3; ++my)
        {
          const double2 v = br[my];
          const pair8 own{v.x, v.y};
          const double bp2 = rr_next(own.x);
          if (odd)
            row_b(Cr0, Cr1, tC, 1, my, own, bp2);
          else
          {
            if (cC >= 0)
              row_b(Cr0, Cr1, tC, 0, my, own, bp2);
            if (cP >= 0)
              row_b(Pr0, Pr1, tP, 2, my, own, bp2);
          }
        }
      }
    }
    if (odd)
    {
      // the oldest agglomerate layer is complete: (li - 3) / 2 layers above ak0
      const int akP = ak0 + (li >> 1) - 2;
      if (cP >= 0 && lane >= 1 && lane <= kRun && ai <= last)
      {
        const int64_t ag = ai + (int64_t)s.na[0] * (aj + (int64_t)s.na[1] * akP);
        reinterpret_cast<double2 *>(y)[ag] = make_double2(Ps0 - Pr0, Ps1 - Pr1);
      }
      Ps0 = Cs0, Ps1 = Cs1, Pr0 = Cr0, Pr1 = Cr1, cP = cC;
      Cs0 = Ns0, Cs1 = Ns1, Cr0 = 0., Cr1 = 0., cC = cN;
      Ns0 = 0., Ns1 = 0., cN = class_of(ak0 + (li >> 1) + 1);
    }
    __syncthreads();
  }
}

// table of one (class, eigenvector): the 125 values of w = A R^T e around the representative, the 27 of v = R^T e
__global__ void rr_gather_kernel(double const *v, double const *w, int N0, int N1, int N2, int ai, int aj, int ak, int e,
                                 double *table)
{
  const int m = threadIdx.x;
  if (m < kFoot)
  {
    const int gx = 2 * ai - 1 + m % 5, gy = 2 * aj - 1 + (m / 5) % 5, gz = 2 * ak - 1 + m / 25;
    const bool in = gx >= 0 && gx < N0 && gy >= 0 && gy < N1 && gz >= 0 && gz < N2;
    table[2 * m + e] = in ? w[((int64_t)gz * N1 + gy) * N0 + gx] : 0.;
  }
  else if (m < kTab)
  {
    const int mm = m - kFoot;
    const int gx = 2 * ai + mm % 3, gy = 2 * aj + (mm / 3) % 3, gz = 2 * ak + mm / 9;
    table[2 * m + e] = v[((int64_t)gz * N1 + gy) * N0 + gx];
  }
}

__global__ void rr_unit_kernel(double *y, int64_t row) { y[row] = 1.; }
} // namespace

void StructuredRestrictorDevice::drop_residual_restriction()
{
  _rr_table.release();
  _rr_cls.release();
  _rr_seg_class.release();
  _rr_listed.release();
  _rr_segs = _rr_classes = _rr_main_last = 0;
}

bool StructuredRestrictorDevice::build_residual_restriction(std::function<void(double const *, double *)> const &apply_a,
                                                            SlabInfo const &slab)
{
  drop_residual_restriction();
  const int64_t n_agg = (int64_t)_na[0] * _na[1] * _na[2];
  const bool say = std::getenv("MFMG_DEBUG_RR") != nullptr;
  if (_n_eig != 2 || _a[0] != 2 || _a[1] != 2 || _a[2] != 2 || !_identity_numbering || (int64_t)_cls_host.size() != n_agg)
  {
    if (say)
      fprintf(stderr, "[rr] not built: n_eig %d, agglomerate %d %d %d, lexicographic %d, blocks classified for %zu of %lld agglomerates\n", _n_eig,
              _a[0], _a[1], _a[2], (int)_identity_numbering, _cls_host.size(), (long long)n_agg);
    return false;
  }
  // class of an agglomerate for R A: the block of R it repeats and, per direction, which faces of the box its 5 nodes reach
  auto position = [&](int a, int d) { return (a == 0 ? 1 : 0) + (a == _na[d] - 1 ? 2 : 0); };
  // a representative must see its whole 5 x 5 x 5 neighbourhood computed by this rank
  auto can_represent = [&](int ai, int aj, int ak) {
    const int a[3] = {ai, aj, ak};
    for (int d = 0; d < 3; ++d)
    {
      const bool low = d == 2 ? slab.has_low : slab.has_low_xy[d], high = d == 2 ? slab.has_high : slab.has_high_xy[d];
      const int v0 = d == 2 ? slab.valid_begin : slab.valid_begin_xy[d], v1 = d == 2 ? slab.valid_end : slab.valid_end_xy[d];
      for (int m = 0; m < 5; ++m)
      {
        const int g = 2 * a[d] - 1 + m;
        const bool outside_box = (g < 0 && !low) || (g >= _N[d] && !high);
        if (!outside_box && !(g >= v0 && g < std::min(v1, _N[d])))
          return false;
      }
    }
    return true;
  };
  std::map<uint32_t, int> index;
  std::vector<int64_t> representative; // -1: none found yet
  std::vector<char> needed;
  std::vector<uint16_t> cls(n_agg);
  for (int64_t ag = 0; ag < n_agg; ++ag)
  {
    const int ai = (int)(ag % _na[0]), aj = (int)((ag / _na[0]) % _na[1]), ak = (int)(ag / ((int64_t)_na[0] * _na[1]));
    const uint32_t key = ((uint32_t)_cls_host[ag] << 6) | (position(ai, 0) | (position(aj, 1) << 2) | (position(ak, 2) << 4));
    auto it = index.find(key);
    if (it == index.end())
    {
      if (representative.size() >= 4096)
        return false;
      it = index.emplace(key, (int)representative.size()).first;
      representative.push_back(-1);
      needed.push_back(0);
    }
    cls[ag] = (uint16_t)it->second;
    if (representative[it->second] < 0 && can_represent(ai, aj, ak))
      representative[it->second] = ag;
    if (ak >= slab.owned_begin && ak < slab.owned_end && ai >= slab.owned_begin_xy[0] && ai < slab.owned_end_xy[0] &&
        aj >= slab.owned_begin_xy[1] && aj < slab.owned_end_xy[1])
      needed[it->second] = 1;
  }
  for (size_t c = 0; c < representative.size(); ++c)
    if (needed[c] && representative[c] < 0)
    {
      if (say)
        fprintf(stderr, "[rr] not built: a class of owned agglomerates has no representative with its neighbourhood on this rank\n");
      return false;
    }
  const int n_classes = (int)representative.size();
  // wavefronts of the row-wise part: runs of 62 agglomerates i = 1 .. main_last of one row, when they share a class; a
  // last run shorter than 24 is left to the list together with i = 0 and i = na0 - 1
  const int interior = std::max(_na[0] - 2, 0);
  const int segs = interior / kRun + (interior % kRun >= 24 ? 1 : 0);
  const int main_last = std::min(interior, segs * kRun);
  const int64_t n_rows = (int64_t)_na[1] * _na[2];
  std::vector<uint16_t> seg_class((size_t)(segs * n_rows), 0xffff);
  std::vector<int32_t> listed;
  for (int64_t row = 0; row < n_rows; ++row)
  {
    const int64_t base = row * _na[0];
    listed.push_back((int32_t)base);
    for (int sg = 0; sg < segs; ++sg)
    {
      const int i0 = 1 + sg * kRun, i1 = std::min(main_last, i0 + kRun - 1);
      bool uniform = true;
      for (int i = i0 + 1; i <= i1; ++i)
        uniform = uniform && cls[base + i] == cls[base + i0];
      if (uniform)
        seg_class[((row / _na[1]) * segs + sg) * _na[1] + row % _na[1]] = cls[base + i0];
      else
        for (int i = i0; i <= i1; ++i)
          listed.push_back((int32_t)(base + i));
    }
    for (int i = main_last + 1; i < _na[0]; ++i)
      listed.push_back((int32_t)(base + i));
  }
  // the thread-per-agglomerate part is meant for the faces of the box: when it would carry a quarter of a mesh with
  // full rows of agglomerates, the blocks do not repeat along the rows and the two-step path is the faster one
  if ((segs > 0 && (int64_t)listed.size() * 4 > n_agg) || (segs == 0 && n_classes > 64))
  {
    if (say)
      fprintf(stderr, "[rr] not built: %zu of %lld agglomerates outside the row-wise part\n", listed.size(), (long long)n_agg);
    return false;
  }
  // probe: W[class] = A (R^T e) on the 5^3 nodes around the representative
  hipStream_t st = _handle.stream;
  DeviceBuffer<double> unit((size_t)_n_coarse), v((size_t)_n_fine), w((size_t)_n_fine), table((size_t)n_classes * kTab * 2);
  MFMG_HIP_CHECK(hipMemsetAsync(table.data(), 0, table.size() * sizeof(double), st));
  for (int c = 0; c < n_classes; ++c)
  {
    const int64_t ag = representative[c];
    if (ag < 0)
      continue; // a class of the neighbours' agglomerates only: its rows are never used (the table stays zero)
    const int ai = (int)(ag % _na[0]), aj = (int)((ag / _na[0]) % _na[1]), ak = (int)(ag / ((int64_t)_na[0] * _na[1]));
    for (int e = 0; e < 2; ++e)
    {
      MFMG_HIP_CHECK(hipMemsetAsync(unit.data(), 0, (size_t)_n_coarse * sizeof(double), st));
      hipLaunchKernelGGL(rr_unit_kernel, dim3(1), dim3(1), 0, st, unit.data(), 2 * ag + e);
      prolongate(unit.data(), v.data(), false);
      apply_a(v.data(), w.data());
      hipLaunchKernelGGL(rr_gather_kernel, dim3(1), dim3(192), 0, st, v.data(), w.data(), _N[0], _N[1], _N[2], ai, aj, ak, e,
                         table.data() + (size_t)c * kTab * 2);
      MFMG_HIP_CHECK(hipGetLastError());
    }
  }
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
  _rr_table = std::move(table);
  _rr_cls.upload(cls.data(), cls.size(), st);
  if (!seg_class.empty())
    _rr_seg_class.upload(seg_class.data(), seg_class.size(), st);
  _rr_listed.upload(listed.data(), listed.size(), st);
  _rr_segs = segs;
  _rr_main_last = main_last;
  _rr_classes = n_classes;
  return true;
}

template <typename TI>
void StructuredRestrictorDevice::restrict_residual_any(TI const *x, TI const *b, double *y) const
{
  ASSERT_THROW(has_residual_restriction(), "the residual restriction has not been built");
  ASSERT_THROW(x != nullptr && b != nullptr && y != nullptr && (void const *)x != (void const *)y && (void const *)b != (void const *)y,
               "bad vectors");
  ASSERT_THROW(_n_fine < (int64_t(1) << 31), "the residual restriction indexes with 32 bits");
  RrArgs s;
  s.table = reinterpret_cast<double2 const *>(_rr_table.data());
  s.seg_class = _rr_seg_class.data();
  s.cls = _rr_cls.data();
  s.listed = _rr_listed.data();
  s.n_listed = (int64_t)_rr_listed.size();
  s.n_main_waves = (int64_t)_rr_segs * _na[1] * _na[2];
  s.main_blocks = (unsigned int)(((s.n_main_waves + 3) / 4 + 7) / 8 * 8);
  if (s.n_main_waves == 0)
    s.main_blocks = 0;
  for (int d = 0; d < 3; ++d)
  {
    s.N[d] = _N[d];
    s.na[d] = _na[d];
  }
  s.segs = _rr_segs;
  s.main_last = _rr_main_last;
  // MFMG_RR_KERNEL=rows: the row-wise kernel of rounds 2-3 (a wavefront per agglomerate row, no LDS) for comparisons
  static const bool tile_form = [] {
    char const *e = std::getenv("MFMG_RR_KERNEL");
    return !(e && std::string(e) == "rows");
  }();
  hipEvent_t stop = _handle.profiler.begin("residual_restriction", 2. * sizeof(TI) * double(_n_fine) + 8. * double(_n_coarse), _handle.stream);
  if (tile_form)
  {
    // height of a tile: whole rounds of two workgroups per CU; a workgroup of ka agglomerate layers passes 2 ka + 3 node layers
    static const int n_cus = [] {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        v = 256;
      return v > 0 ? v : 256;
    }();
    static const int ka_env = std::getenv("MFMG_RR_TILE_LAYERS") ? std::atoi(std::getenv("MFMG_RR_TILE_LAYERS")) : 0;
    s.tiles_j = (_na[1] + kTileRows - 1) / kTileRows;
    int ka = ka_env > 0 ? std::min(ka_env, _na[2]) : 0;
    if (ka == 0)
    {
      double best = 0.;
      for (int nk = 1; nk <= _na[2]; ++nk)
      {
        const int t = (_na[2] + nk - 1) / nk;
        if ((_na[2] + t - 1) / t != nk)
          continue;
        const int64_t tiles = (int64_t)_rr_segs * s.tiles_j * nk, slots = 2 * (int64_t)n_cus;
        const double cost = double((tiles + slots - 1) / slots) * (2. * t + 3.);
        if (ka == 0 || cost < best)
        {
          best = cost;
          ka = t;
        }
      }
    }
    s.ka = std::max(ka, 1);
    s.n_tiles = (int64_t)_rr_segs * s.tiles_j * ((_na[2] + s.ka - 1) / s.ka);
    s.main_blocks = s.n_main_waves == 0 ? 0u : (unsigned int)((s.n_tiles + 7) / 8 * 8);
    s.listed_blocks = (unsigned int)((s.n_listed + 16 * kTileRows / 4 - 1) / (16 * kTileRows / 4)); // sixteen lanes per agglomerate
    static const int dbg = std::getenv("MFMG_RR_DEBUG") ? std::atoi(std::getenv("MFMG_RR_DEBUG")) : 0; // (timing experiments: 1 no list, 2 no tiles)
    static int calls = 0; // (the check of the setup against the two steps sees the whole kernel)
    if (dbg == 1 && ++calls > 3)
      s.listed_blocks = 0, s.n_listed = 0;
    if (dbg == 2 && ++calls > 3)
      s.main_blocks = 0, s.n_tiles = 0;
    constexpr size_t lds = 2 * (size_t)kTileLds * 64 * sizeof(double2);
    static const bool attr = [] {
      MFMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<void const *>(&residual_restriction_tile_kernel<TI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      return true;
    }();
    (void)attr;
    hipLaunchKernelGGL((residual_restriction_tile_kernel<TI>), dim3(s.main_blocks + s.listed_blocks), dim3(64 * kTileRows), lds, _handle.stream, s, x, b, y);
    KernelProfiler::end(stop, _handle.stream);
    MFMG_HIP_CHECK(hipGetLastError());
    return;
  }
  s.listed_blocks = (unsigned int)((s.n_listed + 15) / 16);
  static const int waves = [] {
    char const *e = std::getenv("MFMG_RR_WAVES");
    return e ? std::atoi(e) : 4; // (3, 4 and 6 wavefronts per SIMD within 5 % of each other; 1, 2 and 8 slower)
  }();
  const dim3 grid(s.main_blocks + s.listed_blocks);
  if (waves == 1)
    hipLaunchKernelGGL((residual_restriction_kernel<1, TI>), grid, dim3(256), 0, _handle.stream, s, x, b, y);
  else if (waves == 2)
    hipLaunchKernelGGL((residual_restriction_kernel<2, TI>), grid, dim3(256), 0, _handle.stream, s, x, b, y);
  else if (waves == 3)
    hipLaunchKernelGGL((residual_restriction_kernel<3, TI>), grid, dim3(256), 0, _handle.stream, s, x, b, y);
  else if (waves == 6)
    hipLaunchKernelGGL((residual_restriction_kernel<6, TI>), grid, dim3(256), 0, _handle.stream, s, x, b, y);
  else if (waves == 8)
    hipLaunchKernelGGL((residual_restriction_kernel<8, TI>), grid, dim3(256), 0, _handle.stream, s, x, b, y);
  else
    hipLaunchKernelGGL((residual_restriction_kernel<4, TI>), grid, dim3(256), 0, _handle.stream, s, x, b, y);
  KernelProfiler::end(stop, _handle.stream);
  MFMG_HIP_CHECK(hipGetLastError());
}

void StructuredRestrictorDevice::restrict_residual(double const *x, double const *b, double *y) const
{
  restrict_residual_any<double>(x, b, y);
}

void StructuredRestrictorDevice::restrict_residual(float const *x, float const *b, double *y) const
{
  restrict_residual_any<float>(x, b, y);
}
} // namespace mfmg
