// Device-side pieces shared by the kernels of the matrix-free Q1 Laplace operator (mf_laplace.hip: one polynomial term per
// launch; mf_cheb_fused.hip: several terms of the Chebyshev smoother per sweep): the chunk-record geometry, the cell kernels, the
// lane shifts, the addressing helpers and the kernel argument blocks.  Included by those two translation units only.
#pragma once

#include "mf_laplace.hpp"

#include <type_traits>

namespace mfmg
{
template <typename T>
struct MfArgs
{
  unsigned char const *rec;
  T const *x;
  T const *b;
  T const *dinv; // by DoF id (the rows handed over between wavefronts)
  T const *xprev;
  T *out;
  int Nx, Ny, Nz;
  int TY, TZ;
  unsigned int ncols, ntiles_y, ntiles_z; // ntiles_y, ntiles_z: y- and z-tiles of THIS launch, the first ones are ty0 and z_tile0
  unsigned int ncols_active;              // chunk columns that get workgroups in this launch, the first one is col0 (the last
                                          // column of the mesh may go to the tail slab)
  unsigned int z_tile0, ty0, col0;
  int const *ztab; // z-tile t owns the DoF layers [ztab[t], ztab[t+1]) (device array; uniform TZ or graded, see z_tiling)
  T fx, fy, fz;
  T fax, fbx, fay, fby, faz, fbz; // one coefficient per cell: 2 f M00, 2 f M01 per direction (M = [[2/3, 1/3], [1/3, 2/3]])
  T kd;                           // ... and the diagonal entry of the reference cell matrix: diag = kd * sum of the 8 cell coefficients
  T alpha, beta;
  int mode;
  unsigned int rec_bytes;  // bytes of one chunk record
  int own, halo;           // chunk c holds the node columns own c - halo + lane; it owns its lanes [halo, halo + own)
  int dinv_in_record;      // D^-1 is part of the record (always for eight coefficients per cell)
  // structured numbering (AffineIds below): the kernel computes the ids instead of reading them from the records
  AffineIds aff;
};

// n_boxes > 0: the tiles of the main part of a launch are those of up to six boxes of (column, y, z) tiles, one after the other
// (box q: the tiles bx_end[q - 1] .. bx_end[q] - 1 of the list) -- the shell around the interior tiles of a distributed run as
// ONE launch (launch_outside).  Consecutive workgroups take consecutive tiles of the list: the shell is spread evenly over the
// XCDs (leaving the interior out of the full list instead gave two of the eight XCDs the whole x slab: 170 us for 30 % of the
// tiles).  An argument of its own, not part of MfArgs: the two argument blocks of a launch are selected field by field.
struct MfBoxes
{
  unsigned int n_boxes = 0;
  unsigned int bx_end[6] = {}, bx_c0[6] = {}, bx_nc[6] = {}, bx_y0[6] = {}, bx_ny[6] = {}, bx_z0[6] = {};
};

namespace
{
constexpr unsigned int kFlag = 0x80000000u;  // bit 31: Dirichlet-constrained DoF (read as zero, row = identity)
constexpr unsigned int kGhost = 0x40000000u; // bit 30: DoF owned by another rank (read normally, never written)
constexpr unsigned int kIdMask = ~(kFlag | kGhost);
// (DoF columns a chunk owns: MatrixFreeLaplaceDevice::_own, 62 with one halo lane on either side for the one-term kernels;
// the multi-term sweep needs as many halo lanes as it runs terms)

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
  static constexpr size_t kDinvOff = kCoefOff + kCoefBytes;           // after the coefficients (general layout only)
  // FP64: 4864 B general, 768 B cell-constant (1280 with D^-1).  By default the cell-constant record has no D^-1: with
  // one coefficient per cell the diagonal of a DoF is kd * (sum of the coefficients of the eight cells around it), and the
  // kernel forms that sum on the fly with the same lane / row / layer combines that assemble A x (8 bytes per DoF and
  // launch less to read); mfmg_hip_context_set_stored_diagonal keeps the stored form
  static constexpr size_t bytes(bool with_dinv) { return with_dinv ? kDinvOff + 64 * sizeof(T) : kDinvOff; }
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

// out[p][.] = S[p][0] in0 + S[p][1] in1 with S = [[A, B], [B, A]] and A + B = 1 (true for the Gauss
// interpolation weights and for the 1-D mass matrix S^T S):  out0 = in1 + A (in0 - in1),  out1 = in0 - A (in0 - in1)
// -- three instructions instead of four.
#define MFMG_INTERP(o0, o1, i0, i1, A)                                                                       \
  T o0, o1;                                                                                                  \
  {                                                                                                          \
    const T dlt = (i0) - (i1);                                                                               \
    o0 = fmadd<T>(A, dlt, i1);                                                                               \
    o1 = fmadd<T>(-(A), dlt, i0);                                                                            \
  }
// the same with both outputs scaled by a factor folded into the constants: fa = f A, fb = f B
#define MFMG_INTERP_SCALED(o0, o1, i0, i1, fa, fb)                                                           \
  const T o0 = fmadd<T>(fa, i0, (fb) * (i1));                                                                \
  const T o1 = fmadd<T>(fb, i0, (fa) * (i1));

template <typename T>
__device__ __forceinline__ void direction_apply(T d00, T d10, T d01, T d11, T c00, T c10, T c01, T c11, T f,
                                                T &X00, T &X10, T &X01, T &X11)
{
#pragma clang fp contract(off)
  const T A = T(MFMG_GA);
  // interpolate the one-sided differences d[p][r] to the 2x2 Gauss points of the two other directions
  MFMG_INTERP(t00, t10, d00, d10, A) // t[qp][r=0]
  MFMG_INTERP(t01, t11, d01, d11, A) // t[qp][r=1]
  MFMG_INTERP(g00, g01, t00, t01, A) // g[qp=0][qr]
  MFMG_INTERP(g10, g11, t10, t11, A) // g[qp=1][qr]
  // flux, already summed over the two Gauss points of the differentiated direction; the direction factor rides along
  const T s00 = g00 * (f * c00), s10 = g10 * (f * c10), s01 = g01 * (f * c01), s11 = g11 * (f * c11);
  // transposed interpolation back to the corners
  MFMG_INTERP(w00, w01, s00, s01, A) // w[qp=0][r]
  MFMG_INTERP(w10, w11, s10, s11, A) // w[qp=1][r]
  MFMG_INTERP(x00, x10, w00, w10, A) // x[p][r=0]
  MFMG_INTERP(x01, x11, w01, w11, A) // x[p][r=1]
  X00 = x00;
  X10 = x10;
  X01 = x01;
  X11 = x11;
}

// The same with one coefficient per cell: the flux is a uniform scaling, so interpolation to the Gauss points
// and back collapses into the 1-D mass matrix M = S^T S = [[A^2 + B^2, 2AB], [2AB, A^2 + B^2]] (= [[2/3, 1/3],
// [1/3, 2/3]], the two-point rule is exact here) applied once per transverse direction.  The direction factor
// (and the 2 of the two Gauss points of the differentiated direction) is folded into the constants of the second
// pass (fa = 2 f 2/3, fb = 2 f 1/3, computed on the host); the coefficient multiplies the eight corner sums at the end.
template <typename T>
__device__ __forceinline__ void direction_apply_cc(T d00, T d10, T d01, T d11, T fa, T fb, T &X00, T &X10, T &X01, T &X11)
{
#pragma clang fp contract(off)
  const T A = T(MFMG_GA * MFMG_GA + MFMG_GB * MFMG_GB);
  MFMG_INTERP(t00, t10, d00, d10, A)            // mass matrix over p, r = 0
  MFMG_INTERP(t01, t11, d01, d11, A)            // r = 1
  MFMG_INTERP_SCALED(x00, x01, t00, t01, fa, fb) // mass matrix over r, p = 0
  MFMG_INTERP_SCALED(x10, x11, t10, t11, fa, fb) // p = 1
  X00 = x00;
  X10 = x10;
  X01 = x01;
  X11 = x11;
}

template <typename T>
struct CellFactors
{
  T fx, fy, fz;             // h-scaling of the three directions (eight coefficients per cell)
  T fax, fbx, fay, fby, faz, fbz; // 2 f M00, 2 f M01 per direction (one coefficient per cell)
};

template <typename T>
__device__ __forceinline__ void cell_apply_cc(T const u[8], T cv, CellFactors<T> const &f, T v[8])
{
#pragma clang fp contract(off)
  T X00, X10, X01, X11, Y00, Y10, Y01, Y11, Z00, Z10, Z01, Z11;
  direction_apply_cc<T>(u[1] - u[0], u[3] - u[2], u[5] - u[4], u[7] - u[6], f.fax, f.fbx, X00, X10, X01, X11);
  direction_apply_cc<T>(u[2] - u[0], u[3] - u[1], u[6] - u[4], u[7] - u[5], f.fay, f.fby, Y00, Y10, Y01, Y11);
  direction_apply_cc<T>(u[4] - u[0], u[5] - u[1], u[6] - u[2], u[7] - u[3], f.faz, f.fbz, Z00, Z10, Z01, Z11);
  // corner m = a + 2b + 4d: x pairs (a): X[b][d]; y pairs (b): Y[a][d]; z pairs (d): Z[a][b]
  v[0] = cv * ((-X00 - Y00) - Z00);
  v[1] = cv * ((X00 - Y10) - Z10);
  v[2] = cv * ((Y00 - X10) - Z01);
  v[3] = cv * ((X10 + Y10) - Z11);
  v[4] = cv * ((Z00 - X01) - Y01);
  v[5] = cv * ((X01 - Y11) + Z10);
  v[6] = cv * ((Y01 - X11) + Z01);
  v[7] = cv * ((X11 + Y11) + Z11);
}

template <typename T>
__device__ __forceinline__ void cell_apply(T const u[8], T const c[8], CellFactors<T> const &f, T v[8])
{
#pragma clang fp contract(off)
#ifdef MFMG_MF_ABLATE_CELL
  // MEASUREMENT BUILD ONLY (scratch/r04_fp32_ablation.sh): the whole cell arithmetic replaced by one multiply per corner, every
  // load, lane shift, carry and store kept -- what the ~800 flops per cell cost a launch (BASELINE configs[4])
  (void)f;
#pragma unroll
  for (int m = 0; m < 8; ++m)
    v[m] = c[m] * u[m];
  return;
#endif
  T X00, X10, X01, X11, Y00, Y10, Y01, Y11, Z00, Z10, Z01, Z11;
  // x: differences along a, (p, r) = (b, d); coefficient pairs summed over qa
  direction_apply<T>(u[1] - u[0], u[3] - u[2], u[5] - u[4], u[7] - u[6], c[0] + c[1], c[2] + c[3], c[4] + c[5],
                     c[6] + c[7], f.fx, X00, X10, X01, X11);
  // y: differences along b, (p, r) = (a, d); summed over qb
  direction_apply<T>(u[2] - u[0], u[3] - u[1], u[6] - u[4], u[7] - u[5], c[0] + c[2], c[1] + c[3], c[4] + c[6],
                     c[5] + c[7], f.fy, Y00, Y10, Y01, Y11);
  // z: differences along d, (p, r) = (a, b); summed over qc
  direction_apply<T>(u[4] - u[0], u[5] - u[1], u[6] - u[2], u[7] - u[3], c[0] + c[4], c[1] + c[5], c[2] + c[6],
                     c[3] + c[7], f.fz, Z00, Z10, Z01, Z11);
  v[0] = (-X00 - Y00) - Z00;
  v[1] = (X00 - Y10) - Z10;
  v[2] = (Y00 - X10) - Z01;
  v[3] = (X10 + Y10) - Z11;
  v[4] = (Z00 - X01) - Y01;
  v[5] = (X01 - Y11) + Z10;
  v[6] = (Y01 - X11) + Z01;
  v[7] = (X11 + Y11) + Z11;
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
// constrained DoFs read as zero (bit 31 of the id)
template <typename T>
__device__ __forceinline__ T masked(T x, int id)
{
  return (id < 0) ? T(0) : x;
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

// fused epilogue of one DoF: yv = (A x)_g for an unconstrained row, x0 = x_g
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
} // namespace
} // namespace mfmg
