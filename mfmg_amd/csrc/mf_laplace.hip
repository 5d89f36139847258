// gfx950 kernels of the matrix-free Q1 Laplace operator.
//
// Data layout in HBM ("one cell slot per DoF", rows cut into aligned 64-slot chunks): the DoF
// columns are cut into runs of 63; chunk c of row (j,k) stores the 64 cell slots of the cells
// i = 63c-1 .. 63c+62 (the low halo cell is stored again, +1.6 % memory), so that every load of a
// wavefront is one aligned, contiguous KiB (no power-of-two stride between concurrent wavefronts):
//   S(c,k,j,lane) = ((k Ny + j) ncols + c) 64 + lane
//   fb0  int4 [S]      DoF ids of the b=0 face: corners (0,0,0) (1,0,0) (0,0,1) (1,0,1)
//   fb1  int4 [S]      DoF ids of the b=1 face: corners (0,1,0) (1,1,0) (0,1,1) (1,1,1)
//   coef 16 B [p][S]   the 8 quadrature coefficients, p = 0..NP-1
// Slot (i,j,k) holds the cell whose lowest corner is DoF (i,j,k); cells that stick out of the
// mesh on a high face are phantoms with zero coefficient.  The ids are the caller's global DoF
// ids (any numbering); bit 31 carries the Dirichlet flag, so the constrained-read-as-zero rule
// costs no extra load.
//
// Work decomposition (owner computes, no atomics, no inter-wave synchronisation, results
// independent of the tiling bit for bit): ONE WAVEFRONT per workgroup marches over a tile of
// 64 cell columns x (TY+1) cell rows x (TZ+1) cell layers and owns the 63 x TY x TZ DoFs whose
// eight cells all lie inside (one halo column / row / layer on the low side is recomputed).
// The 8 corner contributions of a cell are combined
//   in x : shift by one lane of the right-face values,
//   in y : a register carried from the previous cell row,
//   in z : a per-lane column in LDS carried from the previous cell layer,
// so every DoF value is complete exactly when its own slot is visited and the smoother
// epilogue (b, D^-1, x_prev) is fused there: A x is never stored.  x is read once per tile: the
// b=0 face of a cell is the b=1 face of the previous row (registers), the d=0 edge is the d=1 edge
// of the previous layer (a second per-lane LDS column), the a=1 corners are the a=0 corners of the
// next lane (DPP wave shift); the steady state loads one id vector and gathers ONE x value per cell.
#include "mf_laplace.hpp"

#include <algorithm>
#include <cmath>

namespace mfmg
{
namespace
{
constexpr unsigned int kFlag = 0x80000000u;  // bit 31: Dirichlet-constrained DoF (read as zero, row = identity)
constexpr unsigned int kGhost = 0x40000000u; // bit 30: DoF owned by another rank (read normally, never written)
constexpr unsigned int kIdMask = ~(kFlag | kGhost);

template <typename T>
struct MfArgs
{
  int4 const *fb0;
  int4 const *fb1;
  void const *coef;
  size_t n_slots;
  T const *x;
  T const *b;
  T const *dinv;
  T const *xprev;
  T *out;
  int Nx, Ny, Nz;
  int TY, TZ;
  unsigned int ncols, ntiles_y, ntiles_z;
  T fx, fy, fz;
  T alpha, beta;
  int mode;
};

// Gauss points of QGauss<1>(2) on [0,1]: interpolation weights S[p][i]
#define MFMG_GA 0.78867513459481288225 // 1 - g0
#define MFMG_GB 0.21132486540518711775 // g0

template <typename T>
__device__ __forceinline__ void load_coef(void const *base, size_t n_slots, size_t slot, T c[8]);

template <>
__device__ __forceinline__ void load_coef<double>(void const *base, size_t n_slots, size_t slot, double c[8])
{
  double2 const *p = reinterpret_cast<double2 const *>(base) + slot;
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    double2 v = p[q * n_slots];
    c[2 * q] = v.x;
    c[2 * q + 1] = v.y;
  }
}

template <>
__device__ __forceinline__ void load_coef<float>(void const *base, size_t n_slots, size_t slot, float c[8])
{
  float4 const *p = reinterpret_cast<float4 const *>(base) + slot;
#pragma unroll
  for (int q = 0; q < 2; ++q)
  {
    float4 v = p[q * n_slots];
    c[4 * q] = v.x;
    c[4 * q + 1] = v.y;
    c[4 * q + 2] = v.z;
    c[4 * q + 3] = v.w;
  }
}

// v = h-scaled G^T diag(c) G u for one Cartesian Q1 cell, sum-factorised.
// corner m = a + 2b + 4d ; quadrature point q = qa + 2qb + 4qc
// (FEEvaluation::evaluate / submit_gradient / integrate of
//  tests/laplace_matrix_free.hpp:145-155).
// Every multiply-add is written as an explicit fma and implicit contraction is off, so that a
// cell evaluates to the same bits in whichever (peeled / unrolled) copy of the loop body it is
// computed: a halo cell of one tile is an interior cell of its neighbour.
template <typename T>
__device__ __forceinline__ T fmadd(T a, T b, T c)
{
  return __builtin_fma(a, b, c);
}
template <>
__device__ __forceinline__ float fmadd<float>(float a, float b, float c)
{
  return __builtin_fmaf(a, b, c);
}

// value held by the previous / next lane of the wavefront (DPP wave shift: a VALU move, no LDS
// crossbar traffic); lanes without a source keep their own value
__device__ __forceinline__ int dpp_from_prev(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_from_next(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ double from_prev_lane(double v)
{
  return __hiloint2double(dpp_from_prev(__double2hiint(v)), dpp_from_prev(__double2loint(v)));
}
__device__ __forceinline__ float from_prev_lane(float v)
{
  return __int_as_float(dpp_from_prev(__float_as_int(v)));
}
__device__ __forceinline__ double from_next_lane(double v)
{
  return __hiloint2double(dpp_from_next(__double2hiint(v)), dpp_from_next(__double2loint(v)));
}
__device__ __forceinline__ float from_next_lane(float v)
{
  return __int_as_float(dpp_from_next(__float_as_int(v)));
}

// out[p][.] = S[p][0] in0 + S[p][1] in1 with S = [[A, B], [B, A]]
#define MFMG_INTERP(o0, o1, i0, i1)                                                                          \
  T o0 = fmadd<T>(A, i0, B * (i1));                                                                          \
  T o1 = fmadd<T>(B, i0, A * (i1));

template <typename T>
__device__ __forceinline__ void direction_apply(T d00, T d10, T d01, T d11, T c00, T c10, T c01, T c11, T f,
                                                T &X00, T &X10, T &X01, T &X11)
{
#pragma clang fp contract(off)
  const T A = T(MFMG_GA), B = T(MFMG_GB);
  // interpolate the one-sided differences d[p][r] to the 2x2 Gauss points of the two other directions
  MFMG_INTERP(t00, t10, d00, d10) // t[qp][r=0]
  MFMG_INTERP(t01, t11, d01, d11) // t[qp][r=1]
  MFMG_INTERP(g00, g01, t00, t01) // g[qp=0][qr]
  MFMG_INTERP(g10, g11, t10, t11) // g[qp=1][qr]
  // flux, already summed over the two Gauss points of the differentiated direction
  const T s00 = g00 * c00, s10 = g10 * c10, s01 = g01 * c01, s11 = g11 * c11;
  // transposed interpolation back to the corners
  MFMG_INTERP(w00, w01, s00, s01) // w[qp=0][r]
  MFMG_INTERP(w10, w11, s10, s11) // w[qp=1][r]
  MFMG_INTERP(x00, x10, w00, w10) // x[p][r=0]
  MFMG_INTERP(x01, x11, w01, w11) // x[p][r=1]
  X00 = f * x00;
  X10 = f * x10;
  X01 = f * x01;
  X11 = f * x11;
}

template <typename T>
__device__ __forceinline__ void cell_apply(T const u[8], T const c[8], T fx, T fy, T fz, T v[8])
{
#pragma clang fp contract(off)
  T X00, X10, X01, X11;
  // x: differences along a, (p, r) = (b, d); coefficient pairs summed over qa
  direction_apply<T>(u[1] - u[0], u[3] - u[2], u[5] - u[4], u[7] - u[6], c[0] + c[1], c[2] + c[3], c[4] + c[5],
                     c[6] + c[7], fx, X00, X10, X01, X11);
  v[0] = -X00;
  v[1] = X00;
  v[2] = -X10;
  v[3] = X10;
  v[4] = -X01;
  v[5] = X01;
  v[6] = -X11;
  v[7] = X11;
  // y: differences along b, (p, r) = (a, d); summed over qb
  direction_apply<T>(u[2] - u[0], u[3] - u[1], u[6] - u[4], u[7] - u[5], c[0] + c[2], c[1] + c[3], c[4] + c[6],
                     c[5] + c[7], fy, X00, X10, X01, X11);
  v[0] -= X00;
  v[2] += X00;
  v[1] -= X10;
  v[3] += X10;
  v[4] -= X01;
  v[6] += X01;
  v[5] -= X11;
  v[7] += X11;
  // z: differences along d, (p, r) = (a, b); summed over qc
  direction_apply<T>(u[4] - u[0], u[5] - u[1], u[6] - u[2], u[7] - u[3], c[0] + c[4], c[1] + c[5], c[2] + c[6],
                     c[3] + c[7], fz, X00, X10, X01, X11);
  v[0] -= X00;
  v[4] += X00;
  v[1] -= X10;
  v[5] += X10;
  v[2] -= X01;
  v[6] += X01;
  v[3] -= X11;
  v[7] += X11;
}

template <typename T>
__global__ __launch_bounds__(64) void mf_laplace_kernel(MfArgs<T> a)
{
#pragma clang fp contract(off)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *pt = reinterpret_cast<T *>(smem_raw); // [TY+1][64] z-carry of the partial sums, lane private
  T *xz = pt + (a.TY + 1) * 64;            // [TY+1][64] z-carry of x: x(ci, j+1, k) of cell row jj

  const int lane = threadIdx.x;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8,
  // observed, speed only); give every XCD a contiguous run of the tile list.
  const unsigned int n_tiles = a.ncols * a.ntiles_y * a.ntiles_z;
  unsigned int w = blockIdx.x;
  if (n_tiles >= 64)
  {
    const unsigned int per_xcd = (n_tiles + 7) / 8;
    w = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (w >= n_tiles)
      return; // (whole wavefront: no barrier is ever reached)
  }
  const int tc = w % a.ncols;
  const int tyi = (w / a.ncols) % a.ntiles_y;
  const int tzi = w / (a.ncols * a.ntiles_y);
  const int ci = tc * 63 - 1 + lane; // cell / DoF column of this lane
  const int Y0 = tyi * a.TY;
  const int Z0 = tzi * a.TZ;
  const bool col_ok = ci >= 0 && ci < a.Nx;
  const bool col_owned = lane >= 1 && ci < a.Nx;
  // the lane whose a=1 corners are not the a=0 corners of lane+1 inside this wavefront
  const bool no_next = (lane == 63) || (ci + 1 >= a.Nx);

  for (int kk = 0; kk <= a.TZ; ++kk)
  {
    const int k = Z0 - 1 + kk;
    if (k >= a.Nz)
      break;
    const bool layer_carry = (kk > 0) && (k >= 1); // xz holds x(., ., k) written by layer k-1
    T ry0 = T(0), ry1 = T(0);
    // b=0 face carried from the previous cell row: raw x values, ids (flag in bit 31)
    T cx[4] = {T(0), T(0), T(0), T(0)};
    int cid[4] = {0, 0, 0, 0};
    bool carried = false;
    for (int jj = 0; jj <= a.TY; ++jj)
    {
      const int j = Y0 - 1 + jj;
      if (j >= a.Ny)
        break;
      const bool slot = col_ok && j >= 0 && k >= 0;
      const bool cell = slot && (ci < a.Nx - 1) && (j < a.Ny - 1) && (k < a.Nz - 1);
      T v[8];
      T x0 = T(0);
      int id0 = 0;
      T n0 = T(0), n2 = T(0), n1x = T(0), n3x = T(0);
      int4 f1 = make_int4(0, 0, 0, 0);
      T c[8];
      const size_t s = (((size_t)max(k, 0) * a.Ny + (size_t)max(j, 0)) * a.ncols + (size_t)tc) * 64 + lane;
      if (slot)
      {
        if (!carried)
        {
          const int4 f0 = a.fb0[s];
          cid[0] = f0.x;
          cid[1] = f0.y;
          cid[2] = f0.z;
          cid[3] = f0.w;
          cx[0] = a.x[(unsigned int)f0.x & kIdMask];
          if (cell)
          {
            cx[1] = a.x[(unsigned int)f0.y & kIdMask];
            cx[2] = a.x[(unsigned int)f0.z & kIdMask];
            cx[3] = a.x[(unsigned int)f0.w & kIdMask];
          }
        }
        id0 = cid[0];
        x0 = cx[0];
        if (cell)
        {
          f1 = a.fb1[s];
          load_coef<T>(a.coef, a.n_slots, s, c);
          n2 = a.x[(unsigned int)f1.z & kIdMask];                                   // x(ci, j+1, k+1)
          n0 = layer_carry ? xz[jj * 64 + lane] : a.x[(unsigned int)f1.x & kIdMask]; // x(ci, j+1, k)
          if (no_next) // issued together with n2 so that the wavefront pays one memory round trip per row
          {
            n1x = a.x[(unsigned int)f1.y & kIdMask];
            n3x = a.x[(unsigned int)f1.w & kIdMask];
          }
        }
      }
      // a=1 corners: the a=0 corners of the next lane (every lane takes part in the shift)
      T n1 = from_next_lane(n0), n3 = from_next_lane(n2);
      if (cell)
      {
        if (no_next)
        {
          n1 = n1x;
          n3 = n3x;
        }
        xz[jj * 64 + lane] = n2;
        T u[8];
        // constrained DoFs read as zero
        u[0] = (cid[0] < 0) ? T(0) : cx[0];
        u[1] = (cid[1] < 0) ? T(0) : cx[1];
        u[4] = (cid[2] < 0) ? T(0) : cx[2];
        u[5] = (cid[3] < 0) ? T(0) : cx[3];
        u[2] = (f1.x < 0) ? T(0) : n0;
        u[3] = (f1.y < 0) ? T(0) : n1;
        u[6] = (f1.z < 0) ? T(0) : n2;
        u[7] = (f1.w < 0) ? T(0) : n3;
        cell_apply<T>(u, c, a.fx, a.fy, a.fz, v);
        cx[0] = n0;
        cx[1] = n1;
        cx[2] = n2;
        cx[3] = n3;
        cid[0] = f1.x;
        cid[1] = f1.y;
        cid[2] = f1.z;
        cid[3] = f1.w;
      }
      else
      {
#pragma unroll
        for (int m = 0; m < 8; ++m)
          v[m] = T(0);
      }
      carried = cell;

      // ---- x combine: DoF column ci gets the a=0 corners of its own cell and the a=1 corners of
      //      the cell of the lane to the left (lane 0 is the halo column: its sum is never used)
      const T s00 = v[0] + from_prev_lane(v[1]); // s[b][d]: b=0,d=0
      const T s10 = v[2] + from_prev_lane(v[3]); // b=1,d=0
      const T s01 = v[4] + from_prev_lane(v[5]); // b=0,d=1
      const T s11 = v[6] + from_prev_lane(v[7]); // b=1,d=1
      // ---- y combine (register carry), z combine (LDS column carry)
      const T t0 = s00 + ry0;
      const T t1 = s01 + ry1;
      ry0 = s10;
      ry1 = s11;
      T *ptj = pt + jj * 64 + lane;
      const T yv = (kk > 0) ? (t0 + *ptj) : T(0);
      *ptj = t1;

      if (slot && col_owned && jj > 0 && kk > 0 && !((unsigned int)id0 & kGhost))
      {
        const unsigned int g = (unsigned int)id0 & kIdMask;
        const T ax = (id0 < 0) ? x0 : yv; // constrained rows: dst_c = src_c
        T o;
        if (a.mode == 0)
          o = ax;
        else if (a.mode == 1)
          o = ax - a.b[g];
        else
        {
          const T wgt = -(a.beta * a.dinv[g]);
          const T r = ax - a.b[g];
          o = (a.mode == 2) ? fmadd<T>(wgt, r, x0) : fmadd<T>(wgt, r, fmadd<T>(a.alpha, x0 - a.xprev[g], x0));
        }
        a.out[g] = o;
      }
    }
  }
}

// ---- setup kernels -----------------------------------------------------------
template <typename T>
__global__ void mf_repack_kernel(int32_t const *cell_dofs, double const *coefficient,
                                 uint8_t const *constrained, int Nx, int Ny, int Nz, int ncols, int4 *fb0,
                                 int4 *fb1, T *coef)
{
  const int64_t n_slots = (int64_t)ncols * Nz * Ny * 64;
  const int nx = Nx - 1, ny = Ny - 1, nz = Nz - 1;
  constexpr int W = 16 / sizeof(T); // values per 16-byte vector
  constexpr int NP = 8 / W;
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n_slots;
       s += (int64_t)gridDim.x * blockDim.x)
  {
    const int lane = s & 63;
    const int64_t chunk = s >> 6;
    const int c = chunk % ncols;
    const int j = (chunk / ncols) % Ny;
    const int k = chunk / ((int64_t)ncols * Ny);
    const int i = c * 63 - 1 + lane;
    int id[8];
    T cf[8];
    if (i < 0 || i >= Nx)
    {
      for (int m = 0; m < 8; ++m)
      {
        id[m] = 0;
        cf[m] = T(0);
      }
    }
    else
    {
      // DoF id of node (ii,jj,kk) through any real cell that has it as a corner
      auto node = [&](int ii, int jj, int kk) {
        const int ic = min(ii, nx - 1), jc = min(jj, ny - 1), kc = min(kk, nz - 1);
        const int64_t cidx = ic + (int64_t)nx * (jc + (int64_t)ny * kc);
        const int g = cell_dofs[cidx * 8 + (ii - ic) + 2 * (jj - jc) + 4 * (kk - kc)];
        const int c = constrained[g];
        return g | (c == 1 ? (int)kFlag : (c == 2 ? (int)kGhost : 0));
      };
      const bool real = (i < nx) && (j < ny) && (k < nz);
      const int own = node(i, j, k);
      for (int m = 0; m < 8; ++m)
      {
        const int ii = i + (m & 1), jj = j + ((m >> 1) & 1), kk = k + (m >> 2);
        // corners of phantom cells that fall outside the mesh point at the slot's own DoF
        id[m] = (ii < Nx && jj < Ny && kk < Nz) ? node(ii, jj, kk) : own;
        cf[m] = real ? T(coefficient[(i + (int64_t)nx * (j + (int64_t)ny * k)) * 8 + m]) : T(0);
      }
    }
    fb0[s] = make_int4(id[0], id[1], id[4], id[5]);
    fb1[s] = make_int4(id[2], id[3], id[6], id[7]);
    for (int p = 0; p < NP; ++p)
      for (int w = 0; w < W; ++w)
        coef[((size_t)p * n_slots + s) * W + w] = cf[p * W + w];
  }
}

// slot of cell / DoF (i,j,k) in the tile-column-major layout (the copy owned by its column)
__device__ __forceinline__ size_t slot_of(int i, int j, int k, int Ny, int ncols)
{
  const int c = (i + 1) / 63;
  const int lane = i + 1 - 63 * c;
  return (((size_t)k * Ny + j) * ncols + c) * 64 + lane;
}

__global__ void mf_range_kernel(int32_t const *cell_dofs, int64_t n, int64_t n_dofs, int *n_bad)
{
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    if (cell_dofs[t] < 0 || cell_dofs[t] >= n_dofs)
      atomicAdd(n_bad, 1);
}

// Every corner (a,b,d) of real cell (i,j,k) as read from cell_dofs must be corner 0 of slot
// (i+a,j+b,k+d): the logical-structure precondition of the tiled kernel.  Also checks the id range.
__global__ void mf_validate_kernel(int32_t const *cell_dofs, int4 const *fb0, int Nx, int Ny, int Nz, int ncols,
                                   int64_t n_dofs, int *n_bad)
{
  const int nx = Nx - 1, ny = Ny - 1, nz = Nz - 1;
  const int64_t n = (int64_t)nx * ny * nz;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x)
  {
    const int i = c % nx;
    const int j = (c / nx) % ny;
    const int k = c / ((int64_t)nx * ny);
    bool bad = false;
    for (int m = 0; m < 8; ++m)
    {
      const int g = cell_dofs[c * 8 + m];
      if (g < 0 || g >= n_dofs)
      {
        bad = true;
        continue;
      }
      const size_t s = slot_of(i + (m & 1), j + ((m >> 1) & 1), k + (m >> 2), Ny, ncols);
      if ((int)((unsigned int)fb0[s].x & kIdMask) != g)
        bad = true;
    }
    if (bad)
      atomicAdd(n_bad, 1);
  }
}

struct DiagTable
{
  double K[8][8]; // K[q][m] = sum_d f_d G[q,d,m]^2
};

// compute_diagonal (tests/laplace_matrix_free.hpp:75-98,158-199): per-cell
// unit-vector applies summed per DoF; constrained entries set to one.
template <typename T>
__global__ void mf_diagonal_kernel(int4 const *fb0, void const *coef, size_t n_slots, int Nx, int Ny, int Nz,
                                   int ncols, DiagTable tab, T *diag, T *dinv)
{
  const int64_t n = (int64_t)Nx * Ny * Nz;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x)
  {
    const int i = t % Nx;
    const int j = (t / Nx) % Ny;
    const int k = t / ((int64_t)Nx * Ny);
    const int id0 = fb0[slot_of(i, j, k, Ny, ncols)].x;
    double sum = 0.;
    for (int m = 0; m < 8; ++m)
    {
      const int ci = i - (m & 1), cj = j - ((m >> 1) & 1), ck = k - (m >> 2);
      if (ci < 0 || cj < 0 || ck < 0 || ci >= Nx - 1 || cj >= Ny - 1 || ck >= Nz - 1)
        continue;
      T c[8];
      load_coef<T>(coef, n_slots, slot_of(ci, cj, ck, Ny, ncols), c);
      for (int q = 0; q < 8; ++q)
        sum += (double)c[q] * tab.K[q][m];
    }
    const unsigned int g = (unsigned int)id0 & kIdMask;
    const double d = (id0 < 0) ? 1. : sum;
    diag[g] = T(d);
    dinv[g] = T(1. / d);
  }
}
} // namespace

template <typename T>
MatrixFreeLaplaceDevice<T>::MatrixFreeLaplaceDevice(HipHandle &handle, mfmg_hip_mesh_desc const &mesh)
    : _handle(handle)
{
  if (mesh.dim != 3)
    ASSERT_THROW_NOT_IMPLEMENTED("the matrix-free HIP operator is implemented for dim = 3 only");
  ASSERT_THROW(mesh.cell_dofs && mesh.coefficient && mesh.constrained,
               "mesh description arrays must not be null");
  int64_t nd = 1, nc = 1;
  for (int d = 0; d < 3; ++d)
  {
    ASSERT_THROW(mesh.n_cells[d] >= 1, "n_cells must be positive");
    _n[d] = mesh.n_cells[d];
    _N[d] = _n[d] + 1;
    _h[d] = mesh.cell_size[d];
    ASSERT_THROW(_h[d] > 0., "cell_size must be positive");
    nd *= _N[d];
    nc *= _n[d];
  }
  ASSERT_THROW(nd == mesh.n_dofs, "n_dofs does not match the cell grid (Q1: (n+1)^dim)");
  ASSERT_THROW(nd < (int64_t(1) << 30), "DoF ids must fit 30 bits (bits 30/31 carry the ghost / constraint flags)");
  _n_dofs = nd;
  for (int d = 0; d < 3; ++d)
    ASSERT_THROW(_n[d] >= 1, "n_cells must be positive");

  hipStream_t st = _handle.stream;
  // stage the plain arrays on the device if they are host arrays
  DeviceBuffer<int32_t> cd_tmp;
  DeviceBuffer<double> co_tmp;
  DeviceBuffer<uint8_t> cn_tmp;
  int32_t const *cd = mesh.cell_dofs;
  double const *co = mesh.coefficient;
  uint8_t const *cn = mesh.constrained;
  if (!mesh.arrays_on_device)
  {
    cd_tmp.upload(mesh.cell_dofs, (size_t)nc * 8, st);
    co_tmp.upload(mesh.coefficient, (size_t)nc * 8, st);
    cn_tmp.upload(mesh.constrained, (size_t)nd, st);
    cd = cd_tmp.data();
    co = co_tmp.data();
    cn = cn_tmp.data();
  }
  _ncols = (_N[0] + 62) / 63;
  const size_t n_slots = (size_t)_ncols * _N[2] * _N[1] * 64;
  _n_slots = n_slots;
  _fb0.resize(n_slots);
  _fb1.resize(n_slots);
  _coef.resize(n_slots * 8);
  _diag.resize(nd);
  _dinv.resize(nd);

  // range check of the ids comes first: the repack kernel dereferences constrained[id]
  DeviceBuffer<int> bad(1);
  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_range_kernel, dim3(n_blocks_for(nc * 8, 256, 1 << 16)), dim3(256), 0, st, cd, nc * 8, nd,
                     bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  ASSERT_THROW(bad.download(st)[0] == 0, "cell_dofs is not a logically structured hex mesh in lexicographic cell "
                                         "order (DoF ids out of range)");
  hipLaunchKernelGGL(mf_repack_kernel<T>, dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st, cd,
                     co, cn, _N[0], _N[1], _N[2], _ncols, _fb0.data(), _fb1.data(), _coef.data());
  MFMG_HIP_CHECK(hipGetLastError());

  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_validate_kernel, dim3(n_blocks_for(nc, 256, 1 << 16)), dim3(256), 0, st, cd,
                     _fb0.data(), _N[0], _N[1], _N[2], _ncols, _n_dofs, bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  int n_bad = bad.download(st)[0];
  ASSERT_THROW(n_bad == 0, "cell_dofs is not a logically structured hex mesh in lexicographic cell order (" +
                               std::to_string(n_bad) + " inconsistent cells)");

  // diagonal: K[q][m] = sum_d f_d (dphi_m/dxi_d)^2 at Gauss point q
  DiagTable tab;
  const double vol = _h[0] * _h[1] * _h[2];
  const double f[3] = {vol / 8. / (_h[0] * _h[0]), vol / 8. / (_h[1] * _h[1]), vol / 8. / (_h[2] * _h[2])};
  const double gp[2] = {MFMG_GB, MFMG_GA};
  for (int q = 0; q < 8; ++q)
    for (int m = 0; m < 8; ++m)
    {
      double sum = 0.;
      for (int d = 0; d < 3; ++d)
      {
        double g = 1.;
        for (int e = 0; e < 3; ++e)
        {
          const int bit = (m >> e) & 1;
          const double xi = gp[(q >> e) & 1];
          if (e == d)
            g *= bit ? 1. : -1.;
          else
            g *= bit ? xi : (1. - xi);
        }
        sum += f[d] * g * g;
      }
      tab.K[q][m] = sum;
    }
  hipLaunchKernelGGL(mf_diagonal_kernel<T>, dim3(n_blocks_for(nd, 256, 1 << 16)), dim3(256), 0, st,
                     _fb0.data(), _coef.data(), _n_slots, _N[0], _N[1], _N[2], _ncols, tab, _diag.data(), _dinv.data());
  MFMG_HIP_CHECK(hipGetLastError());
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::choose_tile(int &ty, int &tz) const
{
  ty = _tile_y;
  tz = _tile_z;
  if (ty > 0 && tz > 0)
    return;
  // heuristic (measured on MI355X, profiles/): the kernel is latency-bound below ~4 rounds of resident
  // wavefronts (256 CUs x 20), so take the largest tile (least halo re-computation) that still yields that many
  const int64_t cols = (_N[0] + 62) / 63;
  const int64_t target_waves = 256 * 80;
  int best_ty = 1, best_tz = 1;
  double best_cost = 1e30;
  const int cand_y[] = {2, 4, 8, 16};
  const int cand_z[] = {4, 8, 16, 32, 64};
  for (int cy : cand_y)
    for (int cz : cand_z)
    {
      const int64_t waves = cols * ((_N[1] + cy - 1) / cy) * ((_N[2] + cz - 1) / cz);
      double cost = (1. + 1. / cy) * (1. + 1. / cz);
      if (waves < target_waves)
        cost *= double(target_waves) / double(std::max<int64_t>(waves, 1));
      if (cost < best_cost)
      {
        best_cost = cost;
        best_ty = cy;
        best_tz = cz;
      }
    }
  if (ty <= 0)
    ty = best_ty;
  if (tz <= 0)
    tz = best_tz;
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::launch(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha,
                                        T beta, T *out) const
{
  ASSERT_THROW(x != nullptr && out != nullptr, "null vector");
  ASSERT_THROW(x != out, "the operator kernel cannot run in place (out aliases x)");
  if (mode != MfMode::apply)
    ASSERT_THROW(b != nullptr, "null right-hand side");
  if (mode == MfMode::next)
    ASSERT_THROW(x_prev != nullptr, "null x_prev");
  int ty, tz;
  choose_tile(ty, tz);
  MfArgs<T> a;
  a.fb0 = _fb0.data();
  a.fb1 = _fb1.data();
  a.coef = _coef.data();
  a.n_slots = _n_slots;
  a.x = x;
  a.b = b;
  a.dinv = _dinv.data();
  a.xprev = x_prev;
  a.out = out;
  a.Nx = _N[0];
  a.Ny = _N[1];
  a.Nz = _N[2];
  a.TY = ty;
  a.TZ = tz;
  const double vol = _h[0] * _h[1] * _h[2];
  a.fx = T(vol / 8. / (_h[0] * _h[0]));
  a.fy = T(vol / 8. / (_h[1] * _h[1]));
  a.fz = T(vol / 8. / (_h[2] * _h[2]));
  a.alpha = alpha;
  a.beta = beta;
  a.mode = static_cast<int>(mode);
  const size_t lds = (size_t)2 * (ty + 1) * 64 * sizeof(T);
  a.ncols = _ncols;
  a.ntiles_y = (_N[1] + ty - 1) / ty;
  a.ntiles_z = (_N[2] + tz - 1) / tz;
  const uint64_t n_tiles = (uint64_t)a.ncols * a.ntiles_y * a.ntiles_z;
  ASSERT_THROW(n_tiles < (1ull << 31), "operator tile too small for this mesh (grid size limit)");
  // one wavefront per tile; rounded up to a multiple of 8 for the XCD-contiguous tile order
  dim3 grid((unsigned int)(n_tiles >= 64 ? ((n_tiles + 7) / 8) * 8 : n_tiles));
  // algorithmic bytes per launch: x + out + 8 idx + 8 coef, plus b / D^-1 / x_prev reads
  const double extra = (mode == MfMode::apply) ? 0. : (mode == MfMode::residual) ? 1. : (mode == MfMode::first) ? 2. : 3.;
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel", algorithmic_bytes_apply() + extra * sizeof(T) * double(_n_dofs),
                                           _handle.stream);
  hipLaunchKernelGGL(mf_laplace_kernel<T>, grid, dim3(64), lds, _handle.stream, a);
  KernelProfiler::end(stop, _handle.stream);
  MFMG_HIP_CHECK(hipGetLastError());
}

template class MatrixFreeLaplaceDevice<double>;
template class MatrixFreeLaplaceDevice<float>;
} // namespace mfmg
