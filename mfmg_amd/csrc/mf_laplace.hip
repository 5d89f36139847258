// gfx950 kernel of the matrix-free Q1 Laplace operator.
//
// Data layout in HBM ("one cell slot per DoF", rows cut into aligned 64-slot chunks): the DoF
// columns are cut into runs of 63; chunk c of row (j,k) holds the 64 cell slots of the cells
// i = 63c-1 .. 63c+62 (the low halo cell is stored again, +1.6 % memory).  Everything the steady
// state of the kernel needs for one chunk sits in ONE contiguous record, so that a wavefront streams
// a single run of memory per row instead of one run per array:
//   chunk r(c,j,k) = (k Ny + j) ncols + c
//   rec  [r]  : own   int  [64]      DoF id of the slot's own DoF (corner (0,0,0) of its cell)
//               coef  16 B [NP][64]  the 8 quadrature coefficients of the cell (NP = 8 sizeof(T) / 16)
//               dinv  T    [64]      1 / diagonal entry of the slot's own DoF
//   fb0  [r]  : int4 [64]            DoF ids of the b=0 face: corners (0,0,0) (1,0,0) (0,0,1) (1,0,1)
//                                    (read for the first row of a tile and by lanes without a cell only)
// Slot (i,j,k) holds the cell whose lowest corner is DoF (i,j,k); cells that stick out of the
// mesh on a high face are phantoms with zero coefficient.  The ids are the caller's global DoF
// ids (any numbering); bit 31 carries the Dirichlet flag, so the constrained-read-as-zero rule
// costs no extra load, bit 30 marks DoFs owned by another rank (read, never written).
// The seven other corner ids of a cell are the own ids of neighbouring slots: the b=1 face of row j is
// read as the own ids of row j+1 in layers k and k+1 (the lane+1 corners by a DPP shift, the layer-k pair
// from an LDS column written one layer earlier), so the steady state fetches 4 B of ids per cell, not 32.
//
// Work decomposition (owner computes, no atomics, results independent of the tiling bit for bit):
// a workgroup of NW wavefronts marches over a tile of 64 cell columns x NW TY cell rows x (TZ+1)
// cell layers and owns the 63 x (NW TY - 1) x TZ DoFs whose eight cells all lie inside (one halo
// column / row / layer on the low side is recomputed).  Wavefront w computes the TY cell rows
// [Yb + w TY, Yb + (w+1) TY) and hands the b=1 partial sums of its last row to wavefront w+1 through
// LDS (one barrier per layer) instead of letting w+1 recompute that row.
// The 8 corner contributions of a cell are combined
//   in x : shift by one lane of the right-face values (DPP),
//   in y : a register carried from the previous cell row (LDS hand-over between wavefronts),
//   in z : a per-lane column in LDS carried from the previous cell layer,
// so every DoF value is complete exactly when its own slot is visited and the smoother
// epilogue (b, D^-1, x_prev) is fused there: A x is never stored.  x is read once per tile: the
// b=0 face of a cell is the b=1 face of the previous row (registers), the d=0 edge is the d=1 edge
// of the previous layer (a second per-lane LDS column), the a=1 corners are the a=0 corners of the
// next lane; the steady state reads one record and gathers ONE x value per cell.
//
// Memory pipeline: a wavefront pays one memory round trip per row.  The ids of row j+1 are fetched
// while row j is computed, and the operands of the epilogue (which depend only on the id of the row's
// own DoF, known one row ahead) are requested together with the coefficients at the top of the row.
// Vector accesses use a uniform (SGPR) base and a 32-bit per-lane byte offset.
#include "mf_laplace.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mfmg
{
template <typename T>
struct MfArgs
{
  unsigned char const *rec;
  int4 const *fb0;
  T const *x;
  T const *b;
  T const *dinv; // by DoF id (the rows handed over between wavefronts)
  T const *xprev;
  T *out;
  int Nx, Ny, Nz;
  int TY, TZ;
  unsigned int ncols, ntiles_y, ntiles_z; // ntiles_z: z-tiles of THIS launch, the first one is tile z_tile0
  unsigned int ncols_active;              // chunk columns that get workgroups (the last one may go to the tail slab)
  unsigned int z_tile0;
  T fx, fy, fz;
  T alpha, beta;
  int mode;
};

namespace
{
constexpr unsigned int kFlag = 0x80000000u;  // bit 31: Dirichlet-constrained DoF (read as zero, row = identity)
constexpr unsigned int kGhost = 0x40000000u; // bit 30: DoF owned by another rank (read normally, never written)
constexpr unsigned int kIdMask = ~(kFlag | kGhost);

// geometry of one chunk record.  CC ("cell constant"): the eight quadrature coefficients of every cell are
// equal (a constant or cell-wise constant material: the reference's default `material_property constant`), and
// ONE value per cell is stored instead of eight -- 76 -> 20 bytes per cell in FP64.
template <typename T, bool CC>
struct Rec
{
  static constexpr int W = 16 / sizeof(T);                           // values per 16-byte vector
  static constexpr int NP = 8 / W;                                    // coefficient vectors per slot
  static constexpr size_t kCoefOff = 256;                             // after the own ids
  static constexpr size_t kCoefBytes = CC ? 64 * sizeof(T) : (size_t)NP * 1024;
  static constexpr size_t kDinvOff = kCoefOff + kCoefBytes;           // after the coefficients
  static constexpr size_t kBytes = kDinvOff + 64 * sizeof(T);         // FP64: 4864 general, 1280 cell-constant
};


// Gauss points of QGauss<1>(2) on [0,1]: interpolation weights S[p][i]
#define MFMG_GA 0.78867513459481288225 // 1 - g0
#define MFMG_GB 0.21132486540518711775 // g0

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

// The same with one coefficient per cell: the flux is a uniform scaling, so interpolation to the Gauss points
// and back collapses into the 1-D mass matrix M = S^T S = [[A^2 + B^2, 2AB], [2AB, A^2 + B^2]] (= [[2/3, 1/3],
// [1/3, 2/3]], the two-point rule is exact here) applied once per transverse direction: half the work.
template <typename T>
__device__ __forceinline__ void direction_apply_cc(T d00, T d10, T d01, T d11, T fw, T &X00, T &X10, T &X01, T &X11)
{
#pragma clang fp contract(off)
  const T A = T(MFMG_GA * MFMG_GA + MFMG_GB * MFMG_GB), B = T(2. * MFMG_GA * MFMG_GB);
  MFMG_INTERP(t00, t10, d00, d10) // mass matrix over p, r = 0
  MFMG_INTERP(t01, t11, d01, d11) // r = 1
  MFMG_INTERP(x00, x01, t00, t01) // mass matrix over r, p = 0
  MFMG_INTERP(x10, x11, t10, t11) // p = 1
  X00 = fw * x00;
  X10 = fw * x10;
  X01 = fw * x01;
  X11 = fw * x11;
}

template <typename T>
__device__ __forceinline__ void cell_apply_cc(T const u[8], T cv, T fx, T fy, T fz, T v[8])
{
#pragma clang fp contract(off)
  T X00, X10, X01, X11;
  const T w2 = cv + cv; // the two Gauss points of the differentiated direction
  direction_apply_cc<T>(u[1] - u[0], u[3] - u[2], u[5] - u[4], u[7] - u[6], fx * w2, X00, X10, X01, X11);
  v[0] = -X00;
  v[1] = X00;
  v[2] = -X10;
  v[3] = X10;
  v[4] = -X01;
  v[5] = X01;
  v[6] = -X11;
  v[7] = X11;
  direction_apply_cc<T>(u[2] - u[0], u[3] - u[1], u[6] - u[4], u[7] - u[5], fy * w2, X00, X10, X01, X11);
  v[0] -= X00;
  v[2] += X00;
  v[1] -= X10;
  v[3] += X10;
  v[4] -= X01;
  v[6] += X01;
  v[5] -= X11;
  v[7] += X11;
  direction_apply_cc<T>(u[4] - u[0], u[5] - u[1], u[6] - u[2], u[7] - u[3], fz * w2, X00, X10, X01, X11);
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

// 32-bit byte offsets from a uniform base: the global_load takes the base from SGPRs and one VGPR per
// lane instead of a 64-bit VGPR pair per request (vectors are limited to 2^32 bytes, checked on the host)
template <typename T>
__device__ __forceinline__ T ld_off(T const *base, unsigned int byte_off)
{
  return *reinterpret_cast<T const *>(reinterpret_cast<char const *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_off(T *base, unsigned int byte_off, T v)
{
  *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
template <typename T>
__device__ __forceinline__ unsigned int id_off(int id)
{
  return ((unsigned int)id & kIdMask) * (unsigned int)sizeof(T);
}

template <typename T, bool CC>
__device__ __forceinline__ void load_coef(unsigned char const *rec, int lane, T c[8])
{
  if constexpr (CC)
  {
    const T v = reinterpret_cast<T const *>(rec + Rec<T, true>::kCoefOff)[lane];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      c[q] = v;
  }
  else
  {
    constexpr int W = Rec<T, false>::W;
    using vec_t = typename std::conditional<sizeof(T) == 8, double2, float4>::type;
#pragma unroll
    for (int p = 0; p < Rec<T, false>::NP; ++p)
    {
      const vec_t v = reinterpret_cast<vec_t const *>(rec + Rec<T, false>::kCoefOff + p * 1024)[lane];
      T const *e = reinterpret_cast<T const *>(&v);
#pragma unroll
      for (int w = 0; w < W; ++w)
        c[p * W + w] = e[w];
    }
  }
}

// fused epilogue of one DoF: ax = (A x)_g, x0 = x_g
template <typename T>
__device__ __forceinline__ T mf_epilogue(MfArgs<T> const &a, int id0, T x0, T yv, T lb, T ld, T lxp)
{
#pragma clang fp contract(off)
  const T ax = (id0 < 0) ? x0 : yv; // constrained rows: dst_c = src_c
  if (a.mode == 0)
    return ax;
  if (a.mode == 1)
    return ax - lb;
  const T wgt = -(a.beta * ld);
  const T r = ax - lb;
  return (a.mode == 2) ? fmadd<T>(wgt, r, x0) : fmadd<T>(wgt, r, fmadd<T>(a.alpha, x0 - lxp, x0));
}

// TYC > 0: rows per wavefront known at compile time (the row loop is fully unrolled: no loop-carried register
// moves, constant LDS offsets); TYC = 0: taken from the arguments.
template <typename T, int TYC, bool CC>
__device__ __forceinline__ void mf_laplace_body(MfArgs<T> const &a, unsigned int bid)
{
#pragma clang fp contract(off)
  const int TY = TYC > 0 ? TYC : a.TY;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform, keep it scalar
  const int NW = blockDim.x >> 6;
  T *pt = reinterpret_cast<T *>(smem_raw) + (size_t)wv * 2 * TY * 64;   // [TY][64] z-carry of the partial sums
  T *xz = pt + TY * 64;                                                  // [TY][64] z-carry of x
  T *xport = reinterpret_cast<T *>(smem_raw) + (size_t)NW * 2 * TY * 64; // [2][NW][2][64] hand-over rows
  int2 *idz = reinterpret_cast<int2 *>(xport + (size_t)2 * NW * 2 * 64) + (size_t)wv * TY * 64; // [TY][64] z-carry of the ids

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8,
  // observed, speed only); give every XCD a contiguous run of the tile list.
  const unsigned int n_tiles = a.ncols_active * a.ntiles_y * a.ntiles_z;
  unsigned int w = bid;
  if (n_tiles >= 64)
  {
    const unsigned int per_xcd = (n_tiles + 7) / 8;
    w = (bid % 8) * per_xcd + bid / 8;
    if (w >= n_tiles)
      return; // the whole workgroup leaves: no barrier is left waiting
  }
  const int tc = w % a.ncols_active;
  const int tyi = (w / a.ncols_active) % a.ntiles_y;
  const int tzi = a.z_tile0 + w / (a.ncols_active * a.ntiles_y);
  const int ci = tc * 63 - 1 + lane;                     // cell / DoF column of this lane
  const int Yb = tyi * (NW * TY - 1) - 1 + wv * TY; // first cell row of this wavefront
  const int Z0 = tzi * a.TZ;
  const bool col_ok = ci >= 0 && ci < a.Nx;
  const bool col_cell = col_ok && ci < a.Nx - 1;
  const bool col_owned = lane >= 1 && ci < a.Nx;
  // the lane whose a=1 corners are not the a=0 corners of lane+1 inside this wavefront
  const bool no_next = (lane == 63) || (ci + 1 >= a.Nx);
  const int jj0 = (Yb < 0) ? 1 : 0; // cell row -1 does not exist (its sums are the zero initial carries)
  const size_t rec_row = (size_t)a.ncols * Rec<T, CC>::kBytes;
  const size_t rec_layer = (size_t)a.Ny * rec_row;
  const bool next_chunk = (lane == 63) && (ci + 1 < a.Nx); // its a=1 corners are lane 1 of the next chunk
  const size_t fb0_row = (size_t)a.ncols * 64;

  for (int kk = 0; kk <= a.TZ; ++kk)
  {
    const int k = Z0 - 1 + kk;
    if (k >= a.Nz)
      break; // uniform over the workgroup
    const bool kin = k >= 0;
    const bool kcell = kin && (k < a.Nz - 1);
    const bool layer_carry = (kk > 0) && (k >= 1); // xz holds x(., ., k) written by layer k-1
    T ry0 = T(0), ry1 = T(0);
    // b=0 face carried from the previous cell row: raw x values, ids (flag in bit 31)
    T cx[4] = {T(0), T(0), T(0), T(0)};
    int cid[4] = {0, 0, 0, 0};
    bool carried = false;
    // first DoF row of a wavefront w > 0, finished after the barrier
    T d00 = T(0), d01 = T(0), dx0 = T(0);
    int did0 = 0;
    bool dlive = false;

    // ids of the first row of the layer
    int j = Yb + jj0;
    const size_t r0 = ((size_t)max(k, 0) * a.Ny + (size_t)j) * a.ncols + (size_t)tc; // uniform
    unsigned char const *recp = a.rec + r0 * Rec<T, CC>::kBytes;
    int4 const *fb0p = a.fb0 + r0 * 64;
    bool slot = col_ok && kin && (j < a.Ny) && (jj0 < TY);
    bool cell = slot && col_cell && (j < a.Ny - 1) && kcell;
    int4 pf0 = make_int4(0, 0, 0, 0);
    // own ids of row j+1 in layer k (A) and k+1 (B); x: the same for column ci+1 where it lies in the next chunk
    int pfA = 0, pfB = 0, pfAx = 0, pfBx = 0;
    if (slot)
      pf0 = fb0p[lane];
    {
      const bool nodes = col_ok && kcell && (j + 1 < a.Ny) && (jj0 < TY);
      if (nodes)
      {
        pfB = reinterpret_cast<int const *>(recp + rec_row + rec_layer)[lane];
        if (!layer_carry)
          pfA = reinterpret_cast<int const *>(recp + rec_row)[lane];
        if (next_chunk)
        {
          pfBx = reinterpret_cast<int const *>(recp + rec_row + rec_layer + Rec<T, CC>::kBytes)[1];
          if (!layer_carry)
            pfAx = reinterpret_cast<int const *>(recp + rec_row + Rec<T, CC>::kBytes)[1];
        }
      }
    }

#pragma unroll
    for (int jj = jj0; jj < TY; ++jj, ++j)
    {
      if (j >= a.Ny)
        break;
      const bool rown = (jj + 1 < TY) && (j + 1 < a.Ny);
      const bool slotn = col_ok && kin && rown;
      const bool celln = slotn && col_cell && (j + 1 < a.Ny - 1) && kcell;
      unsigned char const *recn = recp + rec_row;
      int4 const *fb0n = fb0p + fb0_row;
      T v[8];
      T c[CC ? 1 : 8];
      T n0 = T(0), n2 = T(0), n1x = T(0), n3x = T(0);
      T lb, ld, lxp; // (only read where `stores` holds and the mode asks for them)
      int4 f1;
      {
        const int bn = dpp_from_next(pfB);
        f1.z = pfB;
        f1.w = next_chunk ? pfBx : bn;
        if (layer_carry)
        {
          const int2 c2 = cell ? idz[jj * 64 + lane] : make_int2(0, 0);
          f1.x = c2.x;
          f1.y = c2.y;
        }
        else
        {
          const int an = dpp_from_next(pfA);
          f1.x = pfA;
          f1.y = next_chunk ? pfAx : an;
        }
      }
      if (slot && !carried)
      {
        cid[0] = pf0.x;
        cid[1] = pf0.y;
        cid[2] = pf0.z;
        cid[3] = pf0.w;
      }
      const int id0 = cid[0];
      const bool stores = slot && col_owned && jj > 0 && kk > 0 && !((unsigned int)id0 & kGhost);
      // ---- every request of this row, and the ids of the next one
      if (slot)
      {
        if (!carried)
        {
          cx[0] = ld_off<T>(a.x, id_off<T>(cid[0]));
          if (cell)
          {
            cx[1] = ld_off<T>(a.x, id_off<T>(cid[1]));
            cx[2] = ld_off<T>(a.x, id_off<T>(cid[2]));
            cx[3] = ld_off<T>(a.x, id_off<T>(cid[3]));
          }
        }
        if (cell)
        {
          if constexpr (CC)
            c[0] = reinterpret_cast<T const *>(recp + Rec<T, true>::kCoefOff)[lane];
          else
            load_coef<T, false>(recp, lane, c);
          n2 = ld_off<T>(a.x, id_off<T>(f1.z));                                     // x(ci, j+1, k+1)
          n0 = layer_carry ? xz[jj * 64 + lane] : ld_off<T>(a.x, id_off<T>(f1.x)); // x(ci, j+1, k)
          if (no_next)
          {
            n1x = ld_off<T>(a.x, id_off<T>(f1.y));
            n3x = ld_off<T>(a.x, id_off<T>(f1.w));
          }
        }
        if (stores && a.mode != 0)
        {
          const unsigned int g = id_off<T>(id0);
          lb = ld_off<T>(a.b, g);
          if (a.mode >= 2)
            ld = reinterpret_cast<T const *>(recp + Rec<T, CC>::kDinvOff)[lane];
          if (a.mode == 3)
            lxp = ld_off<T>(a.xprev, g);
        }
      }
      int pfAn = 0, pfBn = 0, pfAxn = 0, pfBxn = 0;
      int pf0n = 0;
      if (col_ok && kcell && rown && (j + 2 < a.Ny))
      {
        pfBn = reinterpret_cast<int const *>(recn + rec_row + rec_layer)[lane];
        if (!layer_carry)
          pfAn = reinterpret_cast<int const *>(recn + rec_row)[lane];
        if (next_chunk)
        {
          pfBxn = reinterpret_cast<int const *>(recn + rec_row + rec_layer + Rec<T, CC>::kBytes)[1];
          if (!layer_carry)
            pfAxn = reinterpret_cast<int const *>(recn + rec_row + Rec<T, CC>::kBytes)[1];
        }
      }
      if (slotn && !celln)
        pf0n = reinterpret_cast<int const *>(fb0n)[4 * lane];

      const T x0 = cx[0];
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
        idz[jj * 64 + lane] = make_int2(f1.z, f1.w);
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
        if constexpr (CC)
          cell_apply_cc<T>(u, c[0], a.fx, a.fy, a.fz, v);
        else
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
      if (jj == 0 && wv > 0)
      {
        // the b=1 sums of the row below arrive from wavefront w-1 after the barrier
        d00 = s00;
        d01 = s01;
        dx0 = x0;
        did0 = id0;
        dlive = slot && col_owned && kk > 0 && !((unsigned int)id0 & kGhost);
      }
      else
      {
        // ---- y combine (register carry), z combine (LDS column carry)
        const T t0 = s00 + ry0;
        const T t1 = s01 + ry1;
        T *ptj = pt + jj * 64 + lane;
        const T yv = (kk > 0) ? (t0 + *ptj) : T(0);
        *ptj = t1;
        if (stores)
          st_off<T>(a.out, id_off<T>(id0), mf_epilogue<T>(a, id0, x0, yv, lb, ld, lxp));
      }
      ry0 = s10;
      ry1 = s11;
      pfA = pfAn;
      pfB = pfBn;
      pfAx = pfAxn;
      pfBx = pfBxn;
      pf0.x = pf0n;
      slot = slotn;
      cell = celln;
      recp = recn;
      fb0p = fb0n;
    }
    if (NW > 1)
    {
      // exports are double-buffered by layer parity: a slot written in layer k is read after barrier k
      // and rewritten in layer k+2, i.e. after barrier k+1, which the reader only passes once it has read
      T *xp = xport + ((size_t)((kk & 1) * NW + wv) * 2) * 64 + lane;
      xp[0] = ry0;
      xp[64] = ry1;
      __syncthreads();
      if (wv > 0)
      {
        T const *ip = xport + ((size_t)((kk & 1) * NW + wv - 1) * 2) * 64 + lane;
        const T t0 = d00 + ip[0];
        const T t1 = d01 + ip[64];
        T *ptj = pt + lane;
        const T yv = (kk > 0) ? (t0 + *ptj) : T(0);
        *ptj = t1;
        if (dlive)
        {
          const unsigned int g = id_off<T>(did0);
          T lb = T(0), ld = T(0), lxp = T(0);
          if (a.mode != 0)
            lb = ld_off<T>(a.b, g);
          if (a.mode >= 2)
            ld = ld_off<T>(a.dinv, g);
          if (a.mode == 3)
            lxp = ld_off<T>(a.xprev, g);
          st_off<T>(a.out, g, mf_epilogue<T>(a, did0, dx0, yv, lb, ld, lxp));
        }
      }
    }
  }
}

// One launch can carry two meshes: the first `n_tail_blocks` workgroups work on `at` (the rotated slab of the
// tail columns, see the constructor), the others on `am`.  Both share the tile shape (NW, TY, TZ).
//
// The cell-constant variant with three rows per wavefront needs 97 VGPRs: asking for five wavefronts per SIMD
// costs it one spilled register and buys a fifth resident wavefront (measured 513^3: 6.15 -> 5.65 ms per sweep;
// with four rows three registers spill and it loses).  The general variant (113-121 VGPRs) stays at four.
template <typename T, int TYC, bool CC>
__global__ __launch_bounds__(512) void mf_laplace_kernel(MfArgs<T> am, MfArgs<T> at, unsigned int n_tail_blocks)
{
  const bool tail = blockIdx.x < n_tail_blocks;
  mf_laplace_body<T, TYC, CC>(tail ? at : am, tail ? blockIdx.x : blockIdx.x - n_tail_blocks);
}
template <typename T, int TYC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5, 5))) void
mf_laplace_cc5_kernel(MfArgs<T> am, MfArgs<T> at, unsigned int n_tail_blocks)
{
  const bool tail = blockIdx.x < n_tail_blocks;
  mf_laplace_body<T, TYC, true>(tail ? at : am, tail ? blockIdx.x : blockIdx.x - n_tail_blocks);
}

// ---- setup kernels -----------------------------------------------------------
template <typename T, bool CC>
__global__ void mf_repack_kernel(int32_t const *cell_dofs, double const *coefficient,
                                 uint8_t const *constrained, int Nx, int Ny, int Nz, int ncols, int4 *fb0,
                                 unsigned char *rec)
{
  const int64_t n_slots = (int64_t)ncols * Nz * Ny * 64;
  const int nx = Nx - 1, ny = Ny - 1, nz = Nz - 1;
  constexpr int W = Rec<T, CC>::W;
  constexpr int NP = Rec<T, CC>::NP;
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
        return g | ((c & 1) ? (int)kFlag : 0) | ((c & 2) ? (int)kGhost : 0);
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
    unsigned char *r = rec + (size_t)chunk * Rec<T, CC>::kBytes;
    reinterpret_cast<int *>(r)[lane] = id[0];
    if constexpr (CC)
      reinterpret_cast<T *>(r + Rec<T, CC>::kCoefOff)[lane] = cf[0];
    else
      for (int p = 0; p < NP; ++p)
        for (int w = 0; w < W; ++w)
          reinterpret_cast<T *>(r + Rec<T, CC>::kCoefOff + p * 1024)[lane * W + w] = cf[p * W + w];
    reinterpret_cast<T *>(r + Rec<T, CC>::kDinvOff)[lane] = T(0);
  }
}

// chunk and lane of cell / DoF (i,j,k) (the copy owned by its column)
__device__ __forceinline__ size_t chunk_of(int i, int j, int k, int Ny, int ncols, int &lane)
{
  const int c = i / 63; // chunk c owns the columns 63c .. 63c+62 in its lanes 1 .. 63 (lane 0 repeats column 63c-1)
  lane = i + 1 - 63 * c;
  return ((size_t)k * Ny + j) * ncols + c;
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
      int lane;
      const size_t r = chunk_of(i + (m & 1), j + ((m >> 1) & 1), k + (m >> 2), Ny, ncols, lane);
      if ((int)((unsigned int)fb0[r * 64 + lane].x & kIdMask) != g)
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
template <typename T, bool CC>
__global__ void mf_diagonal_kernel(int4 const *fb0, unsigned char const *rec, int Nx, int Ny, int Nz, int ncols,
                                   DiagTable tab, T *diag, T *dinv)
{
  const int64_t n = (int64_t)Nx * Ny * Nz;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x)
  {
    const int i = t % Nx;
    const int j = (t / Nx) % Ny;
    const int k = t / ((int64_t)Nx * Ny);
    int lane;
    const size_t r = chunk_of(i, j, k, Ny, ncols, lane);
    const int id0 = fb0[r * 64 + lane].x;
    double sum = 0.;
    for (int m = 0; m < 8; ++m)
    {
      const int ci = i - (m & 1), cj = j - ((m >> 1) & 1), ck = k - (m >> 2);
      if (ci < 0 || cj < 0 || ck < 0 || ci >= Nx - 1 || cj >= Ny - 1 || ck >= Nz - 1)
        continue;
      T c[8];
      int cl;
      const size_t cr = chunk_of(ci, cj, ck, Ny, ncols, cl);
      load_coef<T, CC>(rec + cr * Rec<T, CC>::kBytes, cl, c);
      for (int q = 0; q < 8; ++q)
        sum += (double)c[q] * tab.K[q][m];
    }
    const unsigned int g = (unsigned int)id0 & kIdMask;
    const double d = (id0 < 0) ? 1. : sum;
    diag[g] = T(d);
    dinv[g] = T(1. / d);
  }
}

// are the eight quadrature coefficients of every cell equal?
__global__ void mf_cell_constant_kernel(double const *coefficient, int64_t n_cells, int *n_varying)
{
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_cells; c += (int64_t)gridDim.x * blockDim.x)
  {
    const double v = coefficient[c * 8];
    bool same = true;
    for (int q = 1; q < 8; ++q)
      same = same && (coefficient[c * 8 + q] == v);
    if (!same)
      atomicAdd(n_varying, 1);
  }
}

// copy of D^-1 in slot order inside the records (every slot of a real DoF, the duplicated halo slots too)
template <typename T, bool CC>
__global__ void mf_fill_dinv_kernel(int4 const *fb0, T const *dinv, int Nx, int ncols, int64_t n_slots,
                                    unsigned char *rec)
{
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n_slots;
       s += (int64_t)gridDim.x * blockDim.x)
  {
    const int lane = s & 63;
    const int64_t chunk = s >> 6;
    const int i = (int)(chunk % ncols) * 63 - 1 + lane;
    if (i < 0 || i >= Nx)
      continue;
    const unsigned int g = (unsigned int)fb0[s].x & kIdMask;
    reinterpret_cast<T *>(rec + (size_t)chunk * Rec<T, CC>::kBytes + Rec<T, CC>::kDinvOff)[lane] = dinv[g];
  }
}
} // namespace

namespace
{
// Mesh description of the slab of node columns i0 .. Nx-1 seen with x and y exchanged (x' = y, y' = x - i0):
// cells in the lexicographic order of the rotated frame, corners and quadrature points re-indexed
// (a <-> b), the DoFs of node column i0 flagged as not-to-be-written (they belong to the main launch).
__global__ void mf_slab_desc_kernel(int32_t const *cell_dofs, double const *coefficient, uint8_t const *constrained,
                                    int nx, int ny, int nz, int i0, int32_t *s_cell_dofs, double *s_coefficient,
                                    uint8_t *s_constrained)
{
  const int sc = nx - i0; // cell columns of the slab
  const int64_t n = (int64_t)ny * sc * nz;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
  {
    const int j = t % ny;
    const int ip = (t / ny) % sc;
    const int k = t / ((int64_t)ny * sc);
    const int64_t src = (i0 + ip) + (int64_t)nx * (j + (int64_t)ny * k);
    for (int m = 0; m < 8; ++m)
    {
      const int ms = ((m >> 1) & 1) | ((m & 1) << 1) | (m & 4); // a' = b, b' = a
      const int32_t g = cell_dofs[src * 8 + ms];
      s_cell_dofs[t * 8 + m] = g;
      s_coefficient[t * 8 + m] = coefficient[src * 8 + ms];
      if (ip == 0 && (ms & 1) == 0) // corner on node column i0
        s_constrained[g] = constrained[g] | 2;
    }
  }
}
} // namespace

template <typename T>
MatrixFreeLaplaceDevice<T>::MatrixFreeLaplaceDevice(HipHandle &handle, mfmg_hip_mesh_desc const &mesh, bool allow_compact,
                                                    bool sub_mesh)
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
  if (sub_mesh) // a sub-box of a larger numbering: the vectors are longer than the node count
    ASSERT_THROW(nd <= mesh.n_dofs, "sub-mesh larger than the numbering");
  else
    ASSERT_THROW(nd == mesh.n_dofs, "n_dofs does not match the cell grid (Q1: (n+1)^dim)");
  nd = mesh.n_dofs;
  ASSERT_THROW(nd < (int64_t(1) << 30), "DoF ids must fit 30 bits (bits 30/31 carry the ghost / constraint flags)");
  ASSERT_THROW((uint64_t)nd * sizeof(T) <= (uint64_t(1) << 32),
               "vectors are addressed with 32-bit byte offsets: at most 2^32 bytes per vector and rank");
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
  _diag.resize(nd);
  _dinv.resize(nd);

  // range check of the ids comes first: the repack kernel dereferences constrained[id]
  DeviceBuffer<int> bad(1);
  // cell-wise constant coefficient: one value per cell is kept (unless the caller asked for the general layout)
  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_cell_constant_kernel, dim3(n_blocks_for(nc, 256, 1 << 16)), dim3(256), 0, st, co, nc,
                     bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  _compact = allow_compact && bad.download(st)[0] == 0;
  _rec.resize((n_slots / 64) * (_compact ? Rec<T, true>::kBytes : Rec<T, false>::kBytes));
  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_range_kernel, dim3(n_blocks_for(nc * 8, 256, 1 << 16)), dim3(256), 0, st, cd, nc * 8, nd,
                     bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  ASSERT_THROW(bad.download(st)[0] == 0, "cell_dofs is not a logically structured hex mesh in lexicographic cell "
                                         "order (DoF ids out of range)");
  if (_compact)
    hipLaunchKernelGGL((mf_repack_kernel<T, true>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st, cd,
                       co, cn, _N[0], _N[1], _N[2], _ncols, _fb0.data(), _rec.data());
  else
    hipLaunchKernelGGL((mf_repack_kernel<T, false>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st, cd,
                       co, cn, _N[0], _N[1], _N[2], _ncols, _fb0.data(), _rec.data());
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
  if (_compact)
  {
    hipLaunchKernelGGL((mf_diagonal_kernel<T, true>), dim3(n_blocks_for(nd, 256, 1 << 16)), dim3(256), 0, st,
                       _fb0.data(), _rec.data(), _N[0], _N[1], _N[2], _ncols, tab, _diag.data(), _dinv.data());
    hipLaunchKernelGGL((mf_fill_dinv_kernel<T, true>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st,
                       _fb0.data(), _dinv.data(), _N[0], _ncols, (int64_t)n_slots, _rec.data());
  }
  else
  {
    hipLaunchKernelGGL((mf_diagonal_kernel<T, false>), dim3(n_blocks_for(nd, 256, 1 << 16)), dim3(256), 0, st,
                       _fb0.data(), _rec.data(), _N[0], _N[1], _N[2], _ncols, tab, _diag.data(), _dinv.data());
    hipLaunchKernelGGL((mf_fill_dinv_kernel<T, false>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st,
                       _fb0.data(), _dinv.data(), _N[0], _ncols, (int64_t)n_slots, _rec.data());
  }
  MFMG_HIP_CHECK(hipGetLastError());
  MFMG_HIP_CHECK(hipStreamSynchronize(st));

  // ---- nearly empty last chunk: hand its columns to a rotated slab operator
  // A row of 2^k + 1 DoFs needs one chunk more than 2^k columns fill (257 = 4 * 63 + 5); the wavefronts of that
  // chunk issue the full instruction stream for 5 of 63 columns, which costs where the kernel is bound by
  // instruction issue (the cell-constant variant).  Those columns (plus the last column of the previous chunk as
  // a halo that is read, not written) form a thin slab whose LONG direction is y: the same kernel runs on it
  // with x and y exchanged, lanes along y, inside the same launch, and the main part skips the last chunk.
  const int tail_cols = _N[0] - 63 * (_ncols - 1); // DoF columns owned by the last chunk
  if (!sub_mesh && _compact && _ncols >= 2 && tail_cols <= 16 && _N[1] >= 64)
  {
    const int i0 = 63 * (_ncols - 1) - 1; // halo column of the last chunk = last column of the chunk before
    const int sc = _n[0] - i0;            // cell columns of the slab
    const int64_t s_cells = (int64_t)_n[1] * sc * _n[2];
    DeviceBuffer<int32_t> s_cd((size_t)s_cells * 8);
    DeviceBuffer<double> s_co((size_t)s_cells * 8);
    DeviceBuffer<uint8_t> s_cn((size_t)nd);
    MFMG_HIP_CHECK(hipMemcpyAsync(s_cn.data(), cn, (size_t)nd, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(mf_slab_desc_kernel, dim3(n_blocks_for(s_cells, 256, 1 << 16)), dim3(256), 0, st, cd, co, cn,
                       _n[0], _n[1], _n[2], i0, s_cd.data(), s_co.data(), s_cn.data());
    MFMG_HIP_CHECK(hipGetLastError());
    mfmg_hip_mesh_desc sd = mesh;
    sd.n_cells[0] = _n[1];
    sd.n_cells[1] = sc;
    sd.n_cells[2] = _n[2];
    sd.cell_size[0] = _h[1];
    sd.cell_size[1] = _h[0];
    sd.cell_size[2] = _h[2];
    sd.cell_dofs = s_cd.data();
    sd.coefficient = s_co.data();
    sd.constrained = s_cn.data();
    sd.arrays_on_device = 1;
    _tail.reset(new MatrixFreeLaplaceDevice<T>(handle, sd, allow_compact, true));
    if (!_tail->cell_constant_layout())
      _tail.reset(); // (cannot happen: a sub-set of cell-constant cells) both parts must run the same kernel
  }
}

// ---- tile choice ---------------------------------------------------------------------------------
// The result does not depend on the tile (bit for bit), only the speed does.  Measured inside the V-cycle
// on MI355X (profiles/README.md): the largest tile wins as long as the launch still has about three
// workgroups of four wavefronts per CU and XCD round; below that the shorter tiles win (257^3 DoFs:
// (4,3,8) 0.450 ms, (4,4,8) 0.457, (4,4,16) 0.500; 513^3 DoFs: (4,4,16) 3.26 ms, (4,3,8) 3.43).  A timed
// choice at first use was tried and dropped: timings outside the cycle did not rank the tiles the way
// the cycle does, and the pick changed from run to run.
template <typename T>
void MatrixFreeLaplaceDevice<T>::choose_tile(int &nw, int &ty, int &tz) const
{
  nw = _tile_waves;
  ty = _tile_y;
  tz = _tile_z;
  if (nw > 0 && ty > 0 && tz > 0)
    return;
  // (the cell-constant variant is fastest with three rows per wavefront, where it fits five wavefronts per SIMD)
  static const int pref_general[][3] = {{4, 4, 16}, {4, 4, 8}, {4, 3, 8}, {4, 2, 8}, {2, 2, 8}, {2, 2, 4}, {1, 2, 4}};
  static const int pref_compact[][3] = {{4, 3, 16}, {4, 3, 8}, {4, 3, 8}, {4, 2, 8}, {2, 2, 8}, {2, 2, 4}, {1, 2, 4}};
  const int(*pref)[3] = _compact ? pref_compact : pref_general;
  constexpr int n_pref = 7;
  int pick = n_pref - 1;
  for (int c = 0; c < n_pref; ++c)
  {
    const int64_t wgs = (int64_t)_ncols * ((_N[1] + pref[c][0] * pref[c][1] - 2) / (pref[c][0] * pref[c][1] - 1)) *
                        ((_N[2] + pref[c][2] - 1) / pref[c][2]);
    if (wgs >= 3 * 1024)
    {
      pick = c;
      break;
    }
  }
  const bool tz_free = tz <= 0;
  if (nw <= 0)
    nw = pref[pick][0];
  if (ty <= 0)
    ty = pref[pick][1];
  if (tz <= 0)
    tz = pref[pick][2];
  if (nw * ty < 2)
    ty = 2;
  // Layers per tile of the cell-constant variant (bound by instruction issue, measured 257^3: 7 or 11 layers
  // 0.244 ms per launch, 8: 0.264, 10: 0.268, 16: 0.292, 21: 0.332): all workgroups cost the same, 5 wavefronts per SIMD
  // are resident, so the launch runs in ceil(workgroups / resident slots) rounds of (layers + 1) layer passes each --
  // choose the layer count that minimises rounds x (layers + 1) with at least two rounds (a single round pays its ramp
  // up and down in full).  The general variant is bound by bytes and does not follow this model (8 layers measured
  // best there).
  if (_compact && tz_free && nw == 4 && ty == 3)
  {
    static const int n_cus = [] {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        v = 256;
      return v > 0 ? v : 256;
    }();
    const double slots = 0.98 * n_cus * 5.; // 5 workgroups of 4 wavefronts per CU
    auto blocks = [&](MatrixFreeLaplaceDevice<T> const &op, int layers) {
      const int64_t cols = op._tail ? op._ncols - 1 : op._ncols;
      const int64_t t = cols * ((op._N[1] + nw * ty - 2) / (nw * ty - 1)) * ((op._N[2] + layers - 1) / layers);
      return t >= 64 ? ((t + 7) / 8) * 8 : t;
    };
    double best = 0.;
    int best_tz = 0;
    for (int layers = 6; layers <= 16; ++layers)
    {
      const int64_t w = blocks(*this, layers) + (_tail ? blocks(*_tail, layers) : 0);
      const double rounds = std::ceil((double)w / slots);
      if (rounds < 2.)
        continue;
      const double cost = rounds * (layers + 1);
      if (best_tz == 0 || cost < best)
      {
        best = cost;
        best_tz = layers;
      }
    }
    if (best_tz > 0)
      tz = best_tz;
  }
}

template <typename T>
bool MatrixFreeLaplaceDevice<T>::make_args(MfArgs<T> &a, unsigned int &n_blocks, MfMode mode, T const *x, T const *b,
                                           T const *x_prev, T alpha, T beta, T *out, int nw, int ty, int tz,
                                           int z_tile_begin, int z_tile_end) const
{
  a.rec = _rec.data();
  a.fb0 = _fb0.data();
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
  a.ncols = _ncols;
  a.ncols_active = _tail ? _ncols - 1 : _ncols;
  // ty cell rows per wavefront, nw ty - 1 owned DoF rows per workgroup
  a.ntiles_y = (_N[1] + nw * ty - 2) / (nw * ty - 1);
  const int all_z = (_N[2] + tz - 1) / tz;
  if (z_tile_end < 0)
    z_tile_end = all_z;
  ASSERT_THROW(z_tile_begin >= 0 && z_tile_end <= all_z, "z-tile range outside the tiling");
  n_blocks = 0;
  if (z_tile_begin >= z_tile_end)
    return false;
  a.z_tile0 = (unsigned int)z_tile_begin;
  a.ntiles_z = (unsigned int)(z_tile_end - z_tile_begin);
  const uint64_t n_tiles = (uint64_t)a.ncols_active * a.ntiles_y * a.ntiles_z;
  ASSERT_THROW(n_tiles < (1ull << 30), "operator tile too small for this mesh (grid size limit)");
  // rounded up to a multiple of 8 for the XCD-contiguous tile order
  n_blocks = (unsigned int)(n_tiles >= 64 ? ((n_tiles + 7) / 8) * 8 : n_tiles);
  return true;
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::run(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out,
                                     int nw, int ty, int tz, int z_tile_begin, int z_tile_end) const
{
  ASSERT_THROW(nw >= 1 && nw <= 8, "1..8 wavefronts per workgroup");
  ASSERT_THROW(ty >= 1 && tz >= 1 && nw * ty >= 2, "operator tile too small");
  MfArgs<T> am, at;
  unsigned int main_blocks = 0, tail_blocks = 0;
  if (!make_args(am, main_blocks, mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, z_tile_begin, z_tile_end))
    return;
  at = am;
  if (_tail) // the columns of the last chunk: same tile shape, same layers, first in the grid
    _tail->make_args(at, tail_blocks, mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, z_tile_begin, z_tile_end);
  const size_t lds = ((size_t)nw * 2 * ty + (size_t)2 * nw * 2) * 64 * sizeof(T) + (size_t)nw * ty * 64 * sizeof(int2);
  ASSERT_THROW(lds <= 160 * 1024, "operator tile too large for the LDS");
  static bool lds_attr_set = false; // (one flag per instantiation of this member)
  auto set_lds = [](const void *f) {
    MFMG_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  };
  if (!lds_attr_set)
  {
    set_lds(reinterpret_cast<const void *>(mf_laplace_kernel<T, 0, false>));
    set_lds(reinterpret_cast<const void *>(mf_laplace_kernel<T, 3, false>));
    set_lds(reinterpret_cast<const void *>(mf_laplace_kernel<T, 4, false>));
    set_lds(reinterpret_cast<const void *>(mf_laplace_kernel<T, 0, true>));
    set_lds(reinterpret_cast<const void *>(mf_laplace_cc5_kernel<T, 3>));
    set_lds(reinterpret_cast<const void *>(mf_laplace_kernel<T, 4, true>));
    lds_attr_set = true;
  }
  const dim3 grid(main_blocks + tail_blocks);
  const dim3 block(64 * nw);
  hipStream_t st = _handle.stream;
  if (_compact)
  {
    if (ty == 3)
      hipLaunchKernelGGL((mf_laplace_cc5_kernel<T, 3>), grid, block, lds, st, am, at, tail_blocks);
    else if (ty == 4)
      hipLaunchKernelGGL((mf_laplace_kernel<T, 4, true>), grid, block, lds, st, am, at, tail_blocks);
    else
      hipLaunchKernelGGL((mf_laplace_kernel<T, 0, true>), grid, block, lds, st, am, at, tail_blocks);
  }
  else
  {
    if (ty == 3)
      hipLaunchKernelGGL((mf_laplace_kernel<T, 3, false>), grid, block, lds, st, am, at, tail_blocks);
    else if (ty == 4)
      hipLaunchKernelGGL((mf_laplace_kernel<T, 4, false>), grid, block, lds, st, am, at, tail_blocks);
    else
      hipLaunchKernelGGL((mf_laplace_kernel<T, 0, false>), grid, block, lds, st, am, at, tail_blocks);
  }
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::check_vectors(MfMode mode, T const *x, T const *b, T const *x_prev, T const *out) const
{
  ASSERT_THROW(x != nullptr && out != nullptr, "null vector");
  ASSERT_THROW(x != out, "the operator kernel cannot run in place (out aliases x)");
  if (mode != MfMode::apply)
    ASSERT_THROW(b != nullptr, "null right-hand side");
  if (mode == MfMode::next)
    ASSERT_THROW(x_prev != nullptr, "null x_prev");
}

template <typename T>
int MatrixFreeLaplaceDevice<T>::tile_layers() const
{
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  return tz;
}

template <typename T>
int MatrixFreeLaplaceDevice<T>::n_z_tiles() const
{
  const int tz = tile_layers();
  return (_N[2] + tz - 1) / tz;
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::launch_z_range(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta,
                                                T *out, int z_tile_begin, int z_tile_end) const
{
  check_vectors(mode, x, b, x_prev, out);
  if (mode == MfMode::next && (x_prev == nullptr || alpha == T(0)))
    mode = MfMode::first;
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  const int all_z = (_N[2] + tz - 1) / tz;
  if (z_tile_begin >= z_tile_end)
    return;
  const double extra = (mode == MfMode::apply) ? 0. : (mode == MfMode::residual) ? 1. : (mode == MfMode::first) ? 2. : 3.;
  const double share = double(z_tile_end - z_tile_begin) / double(all_z);
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel",
                                           share * (algorithmic_bytes_apply() + extra * sizeof(T) * double(_n_dofs)),
                                           _handle.stream);
  run(mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, z_tile_begin, z_tile_end);
  KernelProfiler::end(stop, _handle.stream);
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::launch(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha,
                                        T beta, T *out) const
{
  check_vectors(mode, x, b, x_prev, out);
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  // algorithmic bytes per launch: x + out + 8 idx + 8 coef, plus b / D^-1 / x_prev reads
  const double extra = (mode == MfMode::apply) ? 0. : (mode == MfMode::residual) ? 1. : (mode == MfMode::first) ? 2. : 3.;
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel", algorithmic_bytes_apply() + extra * sizeof(T) * double(_n_dofs),
                                           _handle.stream);
  run(mode, x, b, x_prev, alpha, beta, out, nw, ty, tz);
  KernelProfiler::end(stop, _handle.stream);
}

template class MatrixFreeLaplaceDevice<double>;
template class MatrixFreeLaplaceDevice<float>;
} // namespace mfmg
