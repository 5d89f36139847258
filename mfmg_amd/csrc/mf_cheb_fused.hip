// Several terms of the Chebyshev / Jacobi smoother of the matrix-free Q1 Laplace operator in ONE sweep over the mesh
// (temporal blocking).
//
// What it computes, bit for bit, is what K launches of mf_laplace_kernel in its smoother modes compute
// (source/dealii/dealii_matrix_free_smoother.cc:63-76 with the operator of tests/laplace_matrix_free.hpp:129-156):
//   x_1 = x_0 - beta_1 D^-1 (A x_0 - b),      x_s = x_{s-1} + alpha_s (x_{s-1} - x_{s-2}) - beta_s D^-1 (A x_{s-1} - b),  s = 2 .. K
// with x_0, b, the coefficients and the ids read ONCE and only x_K (and x_{K-1} when a further term follows) written.
//
// Decomposition.  A workgroup of NW wavefronts owns (64 - 2 halo) columns x (NW TY - 2K + 1) rows x TZ layers and marches
// over the layers of its tile.  Term ("stage") s runs one cell layer behind term s - 1: in super-pass p stage 1 works on
// the cell layer c, stage 2 on c - 1, stage 3 on c - 2; a stage completes the DoF layer with the number of its cell layer
// (the cells below were the previous super-pass, their sums wait in registers).  A DoF of stage s is right only where its
// whole stencil of stage s - 1 was, so the valid region shrinks by one node per stage on every side of the tile: K halo
// lanes, rows and layers on either side are recomputed by the neighbouring tiles (owner computes, no atomics; results do
// not depend on the tiling, bit for bit, and equal those of the one-term kernel: same cell kernel, same order of the
// corner sums, explicit fma, contraction off).
//
// Where the state lives.  Everything a stage reads from the stage before is private to a LANE: its own node column
// (the neighbour column comes by DPP from the next lane), so the planes x_{s-1} of the two node layers a stage reads sit
// in an LDS ring of two slots per stage that only the lane itself writes and reads -- no barrier protects them.
// b, D^-1, the cell coefficient and the momentum term of a DoF ride in registers from the super-pass that loads /
// forms them to the later stages (shifted once per super-pass).  The only exchange between wavefronts is the one the
// one-term kernel has: the sums of a wavefront's first and last cell row go to the neighbours below and above (one
// barrier per stage pass), and BOTH complete the node row they share, with the same operands in the same order, so
// that each keeps the rows of its own cells to itself.
//
// Memory pipeline.  Only stage 1 reads global memory: the x_0 plane two node layers ahead, b and the coefficients
// one layer ahead are requested at the top of a super-pass and land in the ring / the registers at its end -- a whole
// super-pass of arithmetic hides the round trip, so two wavefronts per SIMD suffice (the per-lane state costs ~200
// VGPRs).  Dirichlet DoFs (identity rows: their recurrence involves no other DoF) are finished by stage 1 on the spot and
// read as zero by every stage.
//
// Preconditions (checked on the host; anything else takes the term-by-term kernels): cell-constant layout, ids the
// kernel can compute (AffineIds: lexicographic numbering, Dirichlet DoFs on whole faces), records with >= K halo lanes,
// one rank.
#include "mf_device.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
#include <utility>

namespace mfmg
{
template <typename T>
struct MfFusedArgs
{
  unsigned char const *rec;
  T const *x;  // x_0
  T const *b;
  T *out;      // x_K
  T *out_prev; // x_{K-1} (nullptr: not wanted)
  int Nx, Ny, Nz;
  unsigned int ncols, ntiles_y, ntiles_z;
  int own, halo;
  int TZ; // DoF layers a z-tile owns
  unsigned int rec_bytes;
  int dinv_in_record;
  T fax, fbx, fay, fby, faz, fbz, kd;
  T lam[7]; // the cell matrix in mode space (MODES): lambda of the modes dss, sds, ssd, dds, dsd, sdd, ddd
  T alpha[3], beta[3];
  AffineIds aff;
  unsigned int vec_bytes, rec_total_bytes; // extents of the vectors / of the record array (descriptors; at most 2^32 - 1)
  // narrow last chunk column (<= 32 - halo owned columns): its tiles come last in the tile list and take TWO y-tiles each, one
  // per half of the wavefront (wide_tiles = tiles of the other columns; ntiles_y2 = ceil(ntiles_y / 2); 0: no such tiles)
  unsigned int wide_tiles, ntiles_y2;
};

namespace
{
template <int V>
struct IntTag
{
  static constexpr int value = V;
};

// The cell kernel in MODE SPACE (one coefficient per cell).  Per direction a Q1 cell acts on (u0, u1) through s = u0 + u1 and
// d = u1 - u0 alone: the 1-D stiffness matrix is diag(0, 1), the mass matrices diag(1/2, 1/6) and f diag(1, 1/3) on (s, d), and the
// result is (A_s - A_d, A_s + A_d).  So the 8 x 8 cell matrix is DIAGONAL on the eight modes (alpha, beta, gamma) in {s, d}^3
// (lambda_sss = 0), and the transforms are butterflies that neighbouring cells share: the x-sums and differences belong to a NODE
// (two operations per node instead of eight per cell), the y-butterfly to a cell row and node layer.  43 FP64 operations per cell
// and 2 per node instead of 76 -- the sweep is bound by exactly these.  Different rounding from cell_apply_cc (same operator to
// 1e-15 per cell): the sweep keeps both, `MODES = false` is the bit-for-bit twin of the one-term kernel.
template <typename T>
struct ModeFactors
{
  T dss, sds, ssd, dds, dsd, sdd, ddd;
};

// cell row q of a stage pass: P / Q = x-sum / x-difference at the node rows q (0) and q + 1 (1) of the lower (l) and upper (u)
// node layer.  Out: what the row contributes to the node row below it (low: the b = 0 corners) and above it (up), as the
// z-modes s (A) and d (B), already summed over the two cells of the row that share a node column.
template <typename T>
__device__ __forceinline__ void cell_row_modes(T Pl0, T Pl1, T Ql0, T Ql1, T Pu0, T Pu1, T Qu0, T Qu1, T cv, ModeFactors<T> const &m,
                                               T &lowA, T &lowB, T &upA, T &upB)
{
#pragma clang fp contract(off)
  // y butterfly (per node layer)
  const T PPl = Pl0 + Pl1, PDl = Pl1 - Pl0, QPl = Ql0 + Ql1, QDl = Ql1 - Ql0;
  const T PPu = Pu0 + Pu1, PDu = Pu1 - Pu0, QPu = Qu0 + Qu1, QDu = Qu1 - Qu0;
  // z butterfly: the seven modes with a non-zero lambda, scaled by lambda and the cell's coefficient
  const T w_ssd = (m.ssd * (PPu - PPl)) * cv;
  const T w_sds = (m.sds * (PDl + PDu)) * cv;
  const T w_sdd = (m.sdd * (PDu - PDl)) * cv;
  const T w_dss = (m.dss * (QPl + QPu)) * cv;
  const T w_dsd = (m.dsd * (QPu - QPl)) * cv;
  const T w_dds = (m.dds * (QDl + QDu)) * cv;
  const T w_ddd = (m.ddd * (QDu - QDl)) * cv;
  // x back: corner a = 0 gets w_s - w_d, a = 1 gets w_s + w_d; the node column = its own cell's a = 0 + the left cell's a = 1
  const T G_ss = from_prev_lane(w_dss) - w_dss;                 // (w_sss = 0)
  const T G_sd = (w_ssd - w_dsd) + from_prev_lane(w_ssd + w_dsd); // (beta, gamma) = (s, d)
  const T G_ds = (w_sds - w_dds) + from_prev_lane(w_sds + w_dds);
  const T G_dd = (w_sdd - w_ddd) + from_prev_lane(w_sdd + w_ddd);
  // y back: node row q (b = 0) gets G_s - G_d, node row q + 1 gets G_s + G_d
  lowA = G_ss - G_ds;
  upA = G_ss + G_ds;
  lowB = G_sd - G_dd;
  upB = G_sd + G_dd;
}

// ring s holds x_s: three slots where a later stage still reads the own value of the layer two below (the momentum term of
// stage s + 2), two for the last one
__host__ __device__ constexpr int ring_depth(int s, int K) { return s <= K - 2 ? 3 : 2; }
__host__ __device__ constexpr int ring_planes(int K) { return K == 1 ? 2 : (K == 2 ? 5 : 8); }
__host__ __device__ constexpr int ring_base(int s, int K) { return s == 0 ? 0 : (s == 1 ? ring_depth(0, K) : ring_depth(0, K) + ring_depth(1, K)); }

// Global memory through buffer descriptors: a request is "descriptor (4 SGPRs, built once per array from the kernel
// arguments) + uniform 32-bit byte offset (an SGPR: row and layer) + per-lane 32-bit byte offset (one VGPR, loop invariant)".
// No 64-bit per-lane addresses: the loop optimiser otherwise keeps a 64-bit pointer per request stream in VGPRs.
template <typename T>
struct BufIO;
template <>
struct BufIO<double>
{
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ double ld(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff)
  {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  }
  static __device__ __forceinline__ void st(double v, __amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff)
  {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, 0);
  }
};
template <>
struct BufIO<float>
{
  static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff)
  {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  }
  static __device__ __forceinline__ void st(float v, __amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff)
  {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, soff, 0);
  }
};

// DBG (timing experiments only, wrong results): 1 = no barrier, 2 = no division, 3 = no global stores
// NARROW: the tile of a narrow last chunk column.  Lanes 0-31 work on y-tile 2 t, lanes 32-63 on y-tile 2 t + 1 of the same
// 32 columns: everything that depends on the y-tile (row numbers, row masks, the row part of an address) is per lane instead of
// per wavefront; the halves exchange nothing (lane 0 of a half is a halo lane; its last lane is at most the last column of the mesh,
// whose cell is a phantom with coefficient zero: what arrives there from the other half is multiplied by it), owner computes as
// everywhere.
// ZERO0: x_0 = 0 (the pre-smoother of a preconditioner application, hierarchy.hpp:253-259): stage 1 is x_1 = beta_1 D^-1 b -- no
// operator application, nothing read of x_0, no exchange between the wavefronts; the same bits as the sweep run on a zeroed vector.
template <typename T, int K, int TY, bool DREC, bool MODES, int DBG = 0, bool NARROW = false, bool ZERO0 = false>
__device__ __forceinline__ void mf_cheb_fused_body(MfFusedArgs<T> const &a, unsigned int w)
{
#pragma clang fp contract(off)
  constexpr int R = TY + 1; // node rows of a wavefront
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 6;
  // LDS: per wavefront the rings x_0 .. x_{K-1} ([planes][R rows][64 lanes]); then the exports [2][NW][4][64]
  T *ring = reinterpret_cast<T *>(smem_raw) + (size_t)wv * (ring_planes(K) * R * 64) + lane;
  // the same planes as the NEXT lane sees them (the last lane: its own, what a DPP shift with the source as old value returns): the
  // x_{s-1} values of the neighbour column are read from the ring, not shifted through the vector ALU, which bounds the sweep
  T const *ring_next = ring + (lane == 63 ? 0 : 1);
  T *xport = reinterpret_cast<T *>(smem_raw) + (size_t)NW * (ring_planes(K) * R * 64) + lane;

  // (tile w of the list: the kernel has dealt the list to the XCDs in contiguous runs)
  const int wcols = a.ntiles_y2 > 0 ? (int)a.ncols - 1 : (int)a.ncols; // chunk columns with tiles of their own width
  int tc, tyi, tzi, lx = lane;
  if constexpr (NARROW)
  {
    const unsigned int v = w - a.wide_tiles;
    tc = (int)a.ncols - 1;
    tyi = 2 * (int)(v % a.ntiles_y2) + (lane >> 5); // (per lane)
    tzi = (int)(v / a.ntiles_y2);
    lx = lane & 31;
  }
  else
  {
    tc = (int)(w % (unsigned int)wcols);
    tyi = (int)((w / (unsigned int)wcols) % a.ntiles_y);
    tzi = (int)(w / ((unsigned int)wcols * a.ntiles_y));
  }
  const bool tile_live = !NARROW || tyi < (int)a.ntiles_y; // (an odd number of y-tiles leaves the second half of the last pair idle)

  const int ci = tc * a.own - a.halo + lx; // node / cell column of this lane
  const bool lane_in = ci >= 0 && ci < a.Nx;
  const bool lane_face = ((a.aff.faces & 1) && ci == 0) || ((a.aff.faces & 2) && ci == a.Nx - 1);
  const bool lane_free = lane_in && !lane_face;
  const bool col_owned = tile_live && lx >= a.halo && lx < a.halo + a.own && ci < a.Nx - a.aff.ghost_hi[0] && ci >= a.aff.ghost_lo[0]; // (ghost DoFs: computed, never written)
  const int RY = NW * TY - 2 * K + 1;    // DoF rows a tile owns
  const int Yw = tyi * RY - K + wv * TY; // first node row of this wavefront: node rows Yw .. Yw + TY, cell rows Yw .. Yw + TY - 1
  const int own_y0 = tyi * RY, own_y1 = min(own_y0 + RY, a.Ny);
  const int Z0 = tzi * a.TZ, Z1 = min(Z0 + a.TZ, a.Nz);
  // per-lane part of an address: a 32-bit byte offset from a uniform base (lanes outside the mesh load at a clamped column;
  // what they load is never used)
  const unsigned int off_lane = (unsigned int)(a.aff.base + min(max(ci, 0), a.Nx - 1) * a.aff.s0) * (unsigned int)sizeof(T);
  const unsigned int rec_lane = (unsigned int)lx * (unsigned int)sizeof(T);

  // Arguments that only the epilogues and the requests need (alpha, beta, kd, the vector bases) are re-read from the kernel
  // argument segment where they are used (scalar loads through a pointer the compiler cannot see through): held in scalar
  // registers for the whole pass they pushed ~90 other scalars into VGPR lanes (a v_readlane / v_writelane per use).
  typedef const __attribute__((address_space(4))) char *karg_ptr_t;
  auto karg_T = [&](size_t off) -> T {
    karg_ptr_t p = (karg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(p + off);
  };
  auto karg_ptr = [&](size_t off) -> void * {
    karg_ptr_t p = (karg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<void *const __attribute__((address_space(4))) *>(p + off);
  };
  auto k_alpha = [&](int s) { return karg_T(offsetof(MfFusedArgs<T>, alpha) + (size_t)s * sizeof(T)); };
  auto k_beta = [&](int s) { return karg_T(offsetof(MfFusedArgs<T>, beta) + (size_t)s * sizeof(T)); };

  CellFactors<T> fac;
  fac.fx = fac.fy = fac.fz = T(0);
  fac.fax = a.fax;
  fac.fbx = a.fbx;
  fac.fay = a.fay;
  fac.fby = a.fby;
  fac.faz = a.faz;
  fac.fbz = a.fbz;
  ModeFactors<T> mf;
  mf.dss = a.lam[0];
  mf.sds = a.lam[1];
  mf.ssd = a.lam[2];
  mf.dds = a.lam[3];
  mf.dsd = a.lam[4];
  mf.sdd = a.lam[5];
  mf.ddd = a.lam[6];

  // wave-uniform description of the node rows, one bit per row
  unsigned int rows_free = 0, rows_own = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
  {
    const int j = Yw + r;
    const bool real = j >= 0 && j < a.Ny;
    if (real && !(((a.aff.faces & 4) && j == 0) || ((a.aff.faces & 8) && j == a.Ny - 1)))
      rows_free |= 1u << r;
    if (r < TY && j >= own_y0 && j < own_y1 && j >= a.aff.ghost_lo[1] && j < a.Ny - a.aff.ghost_hi[1]) // (the node row two wavefronts share is stored by the upper one)
      rows_own |= 1u << r;
  }
  // descriptors and uniform byte offsets: node row r / node layer n of a vector (clamped into the mesh), cell row q / layer n
  // of this chunk column's records
  auto rs_vec = [&](size_t off) { return __builtin_amdgcn_make_buffer_rsrc(karg_ptr(off), 0, a.vec_bytes, 0x00020000); };
  auto rs_rec_f = [&]() { return __builtin_amdgcn_make_buffer_rsrc(karg_ptr(offsetof(MfFusedArgs<T>, rec)), 0, a.rec_total_bytes, 0x00020000); };
  const bool want_prev = a.out_prev != nullptr;
  const unsigned int row_stride = (unsigned int)a.aff.s1 * (unsigned int)sizeof(T), layer_stride = (unsigned int)a.aff.s2 * (unsigned int)sizeof(T);
  const unsigned int rec_row = a.ncols * a.rec_bytes;
  const unsigned int rec_layer = (unsigned int)a.Ny * rec_row;
  const unsigned int rec_col = (unsigned int)tc * a.rec_bytes;
  auto layer_free = [&](int n) { return n >= 0 && n < a.Nz && !(((a.aff.faces & 16) && n == 0) || ((a.aff.faces & 32) && n == a.Nz - 1)); };
  auto layer_own = [&](int n) { return n >= Z0 && n < Z1 && n >= a.aff.ghost_lo[2] && n < a.Nz - a.aff.ghost_hi[2]; };
  // an address = per-lane part + wave-uniform part: the row belongs to the uniform part, except in a NARROW tile, whose halves
  // work on different rows
  struct Off
  {
    unsigned int v, s;
  };
  auto vec_off = [&](int r, int n) -> Off {
    const unsigned int rowb = (unsigned int)min(max(Yw + r, 0), a.Ny - 1) * row_stride, layb = (unsigned int)min(max(n, 0), a.Nz - 1) * layer_stride;
    if constexpr (NARROW)
      return Off{off_lane + rowb, layb};
    else
      return Off{off_lane, rowb + layb};
  };
  auto rec_off = [&](int q, int n, unsigned int field) -> Off {
    const unsigned int rowb = (unsigned int)min(max(Yw + q, 0), a.Ny - 1) * rec_row, layb = rec_col + (unsigned int)min(max(n, 0), a.Nz - 1) * rec_layer + field;
    if constexpr (NARROW)
      return Off{rec_lane + rowb, layb};
    else
      return Off{rec_lane, rowb + layb};
  };
  auto ld_vec = [&](__amdgpu_buffer_rsrc_t rs, int r, int n) -> T {
    const Off o = vec_off(r, n);
    return BufIO<T>::ld(rs, o.v, o.s);
  };
  auto st_vec = [&](T v, __amdgpu_buffer_rsrc_t rs, int r, int n) {
    const Off o = vec_off(r, n);
    BufIO<T>::st(v, rs, o.v, o.s);
  };
  auto ld_coef = [&](__amdgpu_buffer_rsrc_t rs_rec, int q, int n) -> T {
    const Off o = rec_off(q, n, (unsigned int)Rec<T, true>::kCoefOff);
    return BufIO<T>::ld(rs_rec, o.v, o.s);
  };
  auto ld_dinv = [&](__amdgpu_buffer_rsrc_t rs_rec, int r, int n) -> T {
    const Off o = rec_off(r, n, (unsigned int)Rec<T, true>::kDinvOff);
    return BufIO<T>::ld(rs_rec, o.v, o.s);
  };
  // slot of a node layer in a ring of depth 2 / 3 (layers are >= -1 here)
  auto slot2 = [](int n) { return n & 1; };
  auto slot3 = [](int n) { return (int)((unsigned int)(n + 3) % 3u); };
  auto ring_at = [&](int s, int n, int r) -> T * {
    const int sl = ring_depth(s, K) == 3 ? slot3(n) : slot2(n);
    return ring + ((ring_base(s, K) + sl) * R + r) * 64;
  };
  auto ring_next_at = [&](int s, int n, int r) -> T const * {
    const int sl = ring_depth(s, K) == 3 ? slot3(n) : slot2(n);
    return ring_next + ((ring_base(s, K) + sl) * R + r) * 64;
  };

  // ---- per-lane state carried from super-pass to super-pass (stage s + 1 uses entry s)
  T bq[K][R];  // b of the DoF (row r, layer of the stage)
  T dq[K][R];  // D^-1 likewise (entry 0 written by stage 1 / loaded from the records)
  T cq[K][TY]; // coefficient of the cell (row q, cell layer of the stage)
  T pt[K][R];  // z-carry of the partial sums
  T pcs[R];    // z-carry of the coefficient sums (D^-1 on the fly)
  T clo = T(0), chi = T(0); // coefficients of the cell rows below / above the wavefront's own, stage-1 layer (D^-1 of the shared rows)
#pragma unroll
  for (int s = 0; s < K; ++s)
  {
#pragma unroll
    for (int r = 0; r < R; ++r)
      bq[s][r] = dq[s][r] = pt[s][r] = T(0);
#pragma unroll
    for (int q = 0; q < TY; ++q)
      cq[s][q] = T(0);
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    pcs[r] = T(0);

  // stage s works on the cell layers start(s) .. end(s)
  auto stage_start = [&](int s) { return max(Z0 - K + (s - 1), 0); };
  auto stage_end = [&](int s) { return min(Z1 - 1 + (K - s), a.Nz - 1); };
  const int cb = stage_start(1);

  // ---- prologue: the two x_0 layers, b (D^-1) and the coefficients of the first super-pass
  {
    const __amdgpu_buffer_rsrc_t rs_x = rs_vec(offsetof(MfFusedArgs<T>, x)), rs_b = rs_vec(offsetof(MfFusedArgs<T>, b)), rs_rec = rs_rec_f();
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      if constexpr (!ZERO0)
      {
        *ring_at(0, cb, r) = ld_vec(rs_x, r, cb);
        *ring_at(0, cb + 1, r) = ld_vec(rs_x, r, cb + 1);
      }
      bq[0][r] = ld_vec(rs_b, r, cb);
      if constexpr (DREC)
        dq[0][r] = ld_dinv(rs_rec, r, cb);
    }
#pragma unroll
    for (int q = 0; q < TY; ++q)
    {
      const T cv = ld_coef(rs_rec, q, cb);
      cq[0][q] = (Yw + q >= 0) ? cv : T(0);
    }
    if constexpr (!DREC)
    {
      const T l = ld_coef(rs_rec, -1, cb);
      const T h = ld_coef(rs_rec, TY, cb);
      clo = (Yw - 1 >= 0) ? l : T(0);
      chi = (Yw + TY >= 0) ? h : T(0);
    }
  }

  unsigned int ex = 0; // stage passes so far (parity of the export buffers)

  // ---- one stage pass: stage S (1-based) on the cell layer c
  auto stage_pass = [&](auto tag, int c) {
    constexpr int S = decltype(tag)::value;
    const bool lo_free = layer_free(c), hi_free = layer_free(c + 1);
    const bool l_own = layer_own(c);
    T xl[R], xu[R], xown[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      if constexpr (S == 1 && ZERO0)
      {
        xown[r] = xl[r] = xu[r] = T(0);
        continue;
      }
      const T l = *ring_at(S - 1, c, r), u = *ring_at(S - 1, c + 1, r);
      xown[r] = l;
      if constexpr (S == 1)
      {
        // x_0 sits in the ring as it was read; Dirichlet and out-of-mesh values enter the cells as zero
        const bool rf = (rows_free >> r) & 1u;
        xl[r] = (lane_free && rf && lo_free) ? l : T(0);
        xu[r] = (lane_free && rf && hi_free) ? u : T(0);
      }
      else
      {
        xl[r] = l;
        xu[r] = u;
      }
    }
    T xln[R], xun[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      if constexpr (S == 1)
      {
        // (x_0 sits in its ring unmasked: the masked values are shifted)
        xln[r] = from_next_lane(xl[r]);
        xun[r] = from_next_lane(xu[r]);
      }
      else
      {
        // x_{S-1} sits in its ring as the cells read it: the neighbour column straight from there (an LDS read instead of two
        // DPP moves and their presets per value; the same numbers)
        xln[r] = *ring_next_at(S - 1, c, r);
        xun[r] = *ring_next_at(S - 1, c + 1, r);
      }
    }
    // per cell row: what it contributes to the node row below (low) and above (up), two values each -- the corner sums of the
    // d = 0 / d = 1 node layer (reference arithmetic) or the z-modes s / d (MODES)
    T lowA[TY], lowB[TY], upA[TY], upB[TY], sx[TY];
    if constexpr (S == 1 && ZERO0)
    {
      // A 0 = 0: no cell arithmetic; the coefficient sums of the diagonal remain
#pragma unroll
      for (int q = 0; q < TY; ++q)
      {
        lowA[q] = lowB[q] = upA[q] = upB[q] = sx[q] = T(0);
        if constexpr (!DREC)
        {
          const T cv = cq[0][q];
          sx[q] = cv + from_prev_lane(cv);
        }
      }
    }
    else if constexpr (MODES)
    {
      T Pl[R], Ql[R], Pu[R], Qu[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
      {
        Pl[r] = xl[r] + xln[r];
        Ql[r] = xln[r] - xl[r];
        Pu[r] = xu[r] + xun[r];
        Qu[r] = xun[r] - xu[r];
      }
#pragma unroll
      for (int q = 0; q < TY; ++q)
      {
        const T cv = cq[S - 1][q];
        cell_row_modes<T>(Pl[q], Pl[q + 1], Ql[q], Ql[q + 1], Pu[q], Pu[q + 1], Qu[q], Qu[q + 1], cv, mf, lowA[q], lowB[q], upA[q], upB[q]);
        sx[q] = T(0);
        if constexpr (S == 1 && !DREC)
          sx[q] = cv + from_prev_lane(cv);
      }
    }
    else
    {
#pragma unroll
      for (int q = 0; q < TY; ++q)
      {
        T u[8], v[8];
        u[0] = xl[q];
        u[1] = xln[q];
        u[2] = xl[q + 1];
        u[3] = xln[q + 1];
        u[4] = xu[q];
        u[5] = xun[q];
        u[6] = xu[q + 1];
        u[7] = xun[q + 1];
        const T cv = cq[S - 1][q];
        cell_apply_cc<T>(u, cv, fac, v);
        // x combine: DoF column ci gets the a=0 corners of its own cell and the a=1 corners of the cell of the lane to the left
        lowA[q] = v[0] + from_prev_lane(v[1]);
        upA[q] = v[2] + from_prev_lane(v[3]);
        lowB[q] = v[4] + from_prev_lane(v[5]);
        upB[q] = v[6] + from_prev_lane(v[7]);
        sx[q] = T(0);
        if constexpr (S == 1 && !DREC)
          sx[q] = cv + from_prev_lane(cv);
      }
    }
    // the sums of the last cell row go up, those of the first go down (double-buffered by the parity of the pass)
    constexpr bool kExchange = !(S == 1 && ZERO0); // (all sums are zero: nothing to hand over, no barrier)
    if constexpr (kExchange)
    {
      T *xp = xport + (size_t)((ex & 1) * NW + wv) * 4 * 64;
      xp[0] = upA[TY - 1];
      xp[64] = upB[TY - 1];
      xp[128] = lowA[0];
      xp[192] = lowB[0];
    }
    // a node row is complete: A, B = its two sums over the cell rows below and above; the value of the DoF layer c and the carry
    // for the layer above (reference: A belongs to layer c, B to c + 1; MODES: the z butterfly back)
    auto z_combine = [&](T A, T B, T &carry) -> T {
      if constexpr (MODES)
      {
        const T yv = (A - B) + carry;
        carry = A + B;
        return yv;
      }
      else
      {
        const T yv = A + carry;
        carry = B;
        return yv;
      }
    };
    // one DoF (row r, layer c) of this stage is complete: yv = (A x_{S-1}) there
    auto finish = [&](auto rtag, T yv, T tcs) {
      constexpr int r = decltype(rtag)::value;
      const bool fr = lane_free && ((rows_free >> r) & 1u) && lo_free;
      const bool st = col_owned && ((rows_own >> r) & 1u) && l_own;
      if constexpr (S == 1)
      {
        T d;
        if constexpr (!DREC)
        {
          // the eight cells of the DoF: two rows of this layer + the same of the layer below (carried)
          const T sum8 = tcs + pcs[r];
          pcs[r] = tcs;
          d = DBG == 2 ? karg_T(offsetof(MfFusedArgs<T>, kd)) * sum8 : T(1) / (karg_T(offsetof(MfFusedArgs<T>, kd)) * sum8);
          dq[0][r] = d;
        }
        else
          d = dq[0][r];
        const T x0 = xown[r], lb = bq[0][r];
        const T x1 = fmadd<T>(-(k_beta(0) * d), yv - lb, x0);
        if constexpr (K == 1)
        {
          if (st && fr)
            st_vec(x1, rs_vec(offsetof(MfFusedArgs<T>, out)), r, c);
        }
        else
        {
          *ring_at(1, c, r) = fr ? x1 : T(0);
          if constexpr (K == 2)
            if (st && fr && want_prev)
              st_vec(x1, rs_vec(offsetof(MfFusedArgs<T>, out_prev)), r, c);
        }
        // Dirichlet DoFs: identity rows with D^-1 = 1 (the stored diagonal says so too) -- the whole recurrence here
        if (st && !fr)
        {
          const T dc = DREC ? d : T(1);
          T xa = x0, xb = fmadd<T>(-(k_beta(0) * dc), x0 - lb, x0); // x_{s-1}, x_s
#pragma unroll
          for (int s = 1; s < K; ++s)
          {
            const T xn = fmadd<T>(-(k_beta(s) * dc), xb - lb, fmadd<T>(k_alpha(s), xb - xa, xb));
            xa = xb;
            xb = xn;
          }
          st_vec(xb, rs_vec(offsetof(MfFusedArgs<T>, out)), r, c);
          if (K > 1 && want_prev)
            st_vec(xa, rs_vec(offsetof(MfFusedArgs<T>, out_prev)), r, c);
        }
      }
      else
      {
        // x_{S-1} and x_{S-2} of the DoF (zero where it is not free: the result is dropped there)
        const T xo = xown[r];
        const T xoo = (S == 2 && ZERO0) ? T(0) : *ring_at(S - 2, c, r);
        const T xoo_m = (S == 2) ? ((lane_free && ((rows_free >> r) & 1u) && lo_free) ? xoo : T(0)) : xoo; // (x_0 sits in its ring unmasked)
        const T xs = fmadd<T>(-(k_beta(S - 1) * dq[S - 1][r]), yv - bq[S - 1][r], fmadd<T>(k_alpha(S - 1), xo - xoo_m, xo));
        if constexpr (S < K)
        {
          *ring_at(S, c, r) = fr ? xs : T(0);
          if constexpr (S == K - 1)
            if (st && fr && want_prev)
              st_vec(xs, rs_vec(offsetof(MfFusedArgs<T>, out_prev)), r, c);
        }
        else
        {
          if (st && fr)
            st_vec(xs, rs_vec(offsetof(MfFusedArgs<T>, out)), r, c);
        }
      }
    };
    // the rows between the wavefront's own cells
    auto inner = [&](auto rtag) {
      constexpr int r = decltype(rtag)::value;
      const T t0 = lowA[r] + upA[r - 1];
      const T t1 = lowB[r] + upB[r - 1];
      const T yv = z_combine(t0, t1, pt[S - 1][r]);
      finish(rtag, yv, sx[r] + sx[r - 1]);
    };
    if constexpr (TY >= 2)
      inner(IntTag<1>{});
    if constexpr (TY >= 3)
      inner(IntTag<2>{});
    if constexpr (TY >= 4)
      inner(IntTag<3>{});
    // (LDS only: the requests of the next super-pass stay in flight across the barrier -- __syncthreads() would drain them.
    // Tried instead: a word per wavefront that its two neighbours poll, no workgroup-wide rendezvous -- 7 % slower.)
    if constexpr (DBG != 1 && kExchange)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // the rows shared with the neighbours: both wavefronts complete them, with the same operands in the same order
    {
      T lo0 = T(0), lo1 = T(0), hi0 = T(0), hi1 = T(0);
      if (kExchange && wv > 0)
      {
        T const *ip = xport + (size_t)((ex & 1) * NW + wv - 1) * 4 * 64;
        lo0 = ip[0];
        lo1 = ip[64];
      }
      if (kExchange && wv + 1 < NW)
      {
        T const *ip = xport + (size_t)((ex & 1) * NW + wv + 1) * 4 * 64;
        hi0 = ip[128];
        hi1 = ip[192];
      }
      {
        const T t0 = lowA[0] + lo0;
        const T t1 = lowB[0] + lo1;
        const T yv = z_combine(t0, t1, pt[S - 1][0]);
        T tcs = T(0);
        if constexpr (S == 1 && !DREC)
          tcs = sx[0] + (clo + from_prev_lane(clo));
        finish(IntTag<0>{}, yv, tcs);
      }
      {
        const T t0 = hi0 + upA[TY - 1];
        const T t1 = hi1 + upB[TY - 1];
        const T yv = z_combine(t0, t1, pt[S - 1][TY]);
        T tcs = T(0);
        if constexpr (S == 1 && !DREC)
          tcs = (chi + from_prev_lane(chi)) + sx[TY - 1];
        finish(IntTag<TY>{}, yv, tcs);
      }
    }
    if constexpr (kExchange)
      ++ex;
  };

  // ---- the march
  const int c_last = stage_end(K) + (K - 1);
  for (int c1 = cb; c1 <= c_last; ++c1)
  {
    // requests for the next super-pass: x_0 two node layers ahead, b (D^-1) and the coefficients one
    T pfx[R], pfb[R], pfd[R], pfc[TY], pflo = T(0), pfhi = T(0);
    const __amdgpu_buffer_rsrc_t rs_x = rs_vec(offsetof(MfFusedArgs<T>, x)), rs_b = rs_vec(offsetof(MfFusedArgs<T>, b)), rs_rec = rs_rec_f();
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      pfx[r] = T(0);
      if constexpr (!ZERO0)
        pfx[r] = ld_vec(rs_x, r, c1 + 2);
      pfb[r] = ld_vec(rs_b, r, c1 + 1);
      pfd[r] = T(0);
      if constexpr (DREC)
        pfd[r] = ld_dinv(rs_rec, r, c1 + 1);
    }
#pragma unroll
    for (int q = 0; q < TY; ++q)
      pfc[q] = ld_coef(rs_rec, q, c1 + 1);
    if constexpr (!DREC)
    {
      pflo = ld_coef(rs_rec, -1, c1 + 1);
      pfhi = ld_coef(rs_rec, TY, c1 + 1);
    }

    if (c1 <= stage_end(1))
      stage_pass(IntTag<1>{}, c1);
    if constexpr (K >= 2)
      if (c1 - 1 >= stage_start(2) && c1 - 1 <= stage_end(2))
        stage_pass(IntTag<2>{}, c1 - 1);
    if constexpr (K >= 3)
      if (c1 - 2 >= stage_start(3) && c1 - 2 <= stage_end(3))
        stage_pass(IntTag<3>{}, c1 - 2);

    // shift the carried state by one stage, land the requests
#pragma unroll
    for (int s = K - 1; s >= 1; --s)
    {
#pragma unroll
      for (int r = 0; r < R; ++r)
      {
        bq[s][r] = bq[s - 1][r];
        dq[s][r] = dq[s - 1][r];
      }
#pragma unroll
      for (int q = 0; q < TY; ++q)
        cq[s][q] = cq[s - 1][q];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
      bq[0][r] = pfb[r];
      if constexpr (DREC)
        dq[0][r] = pfd[r];
      if constexpr (!ZERO0)
        *ring_at(0, c1 + 2, r) = pfx[r];
    }
#pragma unroll
    for (int q = 0; q < TY; ++q)
      cq[0][q] = (Yw + q >= 0) ? pfc[q] : T(0);
    if constexpr (!DREC)
    {
      clo = (Yw - 1 >= 0) ? pflo : T(0);
      chi = (Yw + TY >= 0) ? pfhi : T(0);
    }
  }
}

// NARROW_TOO: the kernel carries the body for the tiles of a narrow last chunk column as well (twice the code: only the tile
// shapes the sweeps use by default have it; any other shape runs that column with ordinary tiles, most of their lanes idle)
template <typename T, int K, int TY, bool DREC, bool MODES, int DBG = 0, bool NARROW_TOO = false, bool ZERO0 = false>
__global__ __launch_bounds__(512, 2) void mf_cheb_fused_kernel(MfFusedArgs<T> a)
{
  // XCD-aware tile order (as mf_laplace_body): every XCD takes a contiguous run of the tile list
  const unsigned int n_tiles = a.wide_tiles + a.ntiles_y2 * a.ntiles_z;
  unsigned int w = blockIdx.x;
  if (n_tiles >= 64)
  {
    const unsigned int per_xcd = (n_tiles + 7) / 8;
    w = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (w >= n_tiles)
      return; // the whole workgroup leaves
  }
  if constexpr (NARROW_TOO)
  {
    if (w >= a.wide_tiles)
    {
      mf_cheb_fused_body<T, K, TY, DREC, MODES, DBG, true, ZERO0>(a, w);
      return;
    }
  }
  mf_cheb_fused_body<T, K, TY, DREC, MODES, DBG, false, ZERO0>(a, w);
}
} // namespace

// ---- host side --------------------------------------------------------------------------------------------------
template <typename T>
bool MatrixFreeLaplaceDevice<T>::fused_sweep_available(int n_terms) const
{
  if (!(_dim == 3 && _compact && _affine_ids && !_tail && n_terms >= 1 && n_terms <= 3 && _halo >= n_terms && (uint64_t)_rec.size() <= 0xffffffffull))
    return false;
  // distributed: the ranks exchange x once per sweep, n_terms planes deep -- all of them or none (what every rank can do)
  if (_handle.comm.enabled() && n_terms > _handle.comm.sweep_terms())
    return false;
  // a side with ghost planes (a neighbouring rank) must hold n_terms of them: the sweep computes them redundantly
  for (int d = 0; d < 3; ++d)
    if ((_affine.ghost_lo[d] > 0 && _affine.ghost_lo[d] < n_terms) || (_affine.ghost_hi[d] > 0 && _affine.ghost_hi[d] < n_terms))
      return false;
  return true;
}

// The sweep from x_0 = 0 (smoother_sweep with x == nullptr): the three-term kernels with three rows per wavefront and the default
// arithmetic carry that variant.
template <typename T>
bool MatrixFreeLaplaceDevice<T>::fused_zero_guess_available(int n_terms) const
{
  if (n_terms != 3 || !fused_sweep_available(n_terms) || _fused_reference_arithmetic)
    return false;
  int nw, ty, tz;
  choose_fused_tile(n_terms, nw, ty, tz);
  return ty == 3;
}

// tile of the sweep: NW wavefronts of TY cell rows, TZ owned layers.  One workgroup of eight wavefronts per CU (two per
// SIMD: the per-lane state takes ~200 VGPRs) or two of four; the height is chosen so that the workgroups fill whole
// rounds of the chip -- a workgroup lives for K TZ + K^2 stage passes, a part-filled round costs a full one.
template <typename T>
void MatrixFreeLaplaceDevice<T>::choose_fused_tile(int n_terms, int &nw, int &ty, int &tz) const
{
  // (measured at 257^3 DoFs: three terms 8 x 3 rows -- four rows per wavefront spill --, two terms 4 x 4: two independent
  // workgroups per CU, 0.31 against 0.34 ms)
  nw = n_terms == 2 ? 4 : 8;
  ty = n_terms == 2 ? 4 : 3;
  tz = 0;
  static const std::string env = std::getenv("MFMG_MF_FUSED_TILE") ? std::getenv("MFMG_MF_FUSED_TILE") : "";
  if (_fused_tile[0] > 0)
  {
    nw = _fused_tile[0];
    ty = _fused_tile[1];
    tz = _fused_tile[2];
  }
  else if (!env.empty())
  {
    int v[3] = {0, 0, 0};
    if (std::sscanf(env.c_str(), "%d,%d,%d", &v[0], &v[1], &v[2]) == 3)
    {
      nw = v[0];
      ty = v[1];
      tz = v[2];
    }
  }
  ASSERT_THROW(nw >= 1 && nw <= 8 && (ty == 2 || ty == 3 || ty == 4), "tile of the multi-term sweep: 1..8 wavefronts of 2, 3 or 4 rows");
  const int ry = nw * ty - 2 * n_terms + 1;
  ASSERT_THROW(ry >= 1, "tile of the multi-term sweep too small for its halo rows");
  if (tz > 0)
    return;
  static const int n_cus = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      v = 256;
    return v > 0 ? v : 256;
  }();
  const int64_t slots = (int64_t)n_cus * (8 / nw);
  const int64_t nty = (_N[1] + ry - 1) / ry;
  const int64_t tiles_xy = (narrow_last_column() && fused_narrow_capable(n_terms, ty)) ? (int64_t)(_ncols - 1) * nty + (nty + 1) / 2 : (int64_t)_ncols * nty;
  double best = 0.;
  for (int nz = 1; nz <= _N[2]; ++nz)
  {
    const int t = (_N[2] + nz - 1) / nz;
    if ((_N[2] + t - 1) / t != nz)
      continue;
    const int64_t rounds = (tiles_xy * nz + slots - 1) / slots;
    const double cost = double(rounds) * (double(n_terms) * t + double(n_terms) * n_terms);
    if (tz == 0 || cost < best)
    {
      best = cost;
      tz = t;
    }
  }
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::smoother_sweep(int n_terms, T const *alpha, T const *beta, T const *b, T const *x, T *out, T *out_prev) const
{
  ASSERT_THROW(fused_sweep_available(n_terms), "the multi-term smoother sweep is not available for this operator");
  const bool zero_guess = x == nullptr; // x_0 = 0: nothing is read of it (fused_zero_guess_available)
  ASSERT_THROW(b != nullptr && out != nullptr, "null vector");
  ASSERT_THROW(!zero_guess || fused_zero_guess_available(n_terms), "the sweep from a zero guess is not available for this operator / tile");
  ASSERT_THROW((zero_guess || (x != out && x != out_prev)) && out != out_prev, "the multi-term sweep cannot run in place");
  ASSERT_THROW(alpha[0] == T(0), "the first term of a sweep takes no momentum");
  int nw, ty, tz;
  choose_fused_tile(n_terms, nw, ty, tz);
  MfFusedArgs<T> a{};
  a.rec = _rec.data();
  a.x = zero_guess ? b : x; // (never read from a zero guess; the descriptor wants an address)
  a.b = b;
  a.out = out;
  a.out_prev = out_prev;
  a.Nx = _N[0];
  a.Ny = _N[1];
  a.Nz = _N[2];
  a.ncols = (unsigned int)_ncols;
  const int ry = nw * ty - 2 * n_terms + 1;
  a.ntiles_y = (unsigned int)((_N[1] + ry - 1) / ry);
  a.ntiles_z = (unsigned int)((_N[2] + tz - 1) / tz);
  a.own = _own;
  a.halo = _halo;
  a.TZ = tz;
  a.rec_bytes = (unsigned int)_rec_bytes;
  a.dinv_in_record = _dinv_in_record ? 1 : 0;
  {
    const double vol = _h[0] * _h[1] * _h[2];
    const double m00 = MFMG_GA * MFMG_GA + MFMG_GB * MFMG_GB, m01 = 2. * MFMG_GA * MFMG_GB;
    const double f[3] = {vol / 8. / (_h[0] * _h[0]), vol / 8. / (_h[1] * _h[1]), vol / 8. / (_h[2] * _h[2])};
    a.fax = T(2. * f[0] * m00);
    a.fbx = T(2. * f[0] * m01);
    a.fay = T(2. * f[1] * m00);
    a.fby = T(2. * f[1] * m01);
    a.faz = T(2. * f[2] * m00);
    a.fbz = T(2. * f[2] * m01);
    a.kd = T(2. * m00 * m00 * (f[0] + f[1] + f[2]));
    // the cell matrix on the modes (cell_row_modes): stiffness diag(0, 1), masses diag(1/2, 1/6) and f diag(1, 1/3)
    a.lam[0] = T(f[0] / 2.);
    a.lam[1] = T(f[1] / 2.);
    a.lam[2] = T(f[2] / 2.);
    a.lam[3] = T((f[0] + f[1]) / 6.);
    a.lam[4] = T((f[0] + f[2]) / 6.);
    a.lam[5] = T((f[1] + f[2]) / 6.);
    a.lam[6] = T((f[0] + f[1] + f[2]) / 18.);
  }
  for (int s = 0; s < 3; ++s)
  {
    a.alpha[s] = s < n_terms ? alpha[s] : T(0);
    a.beta[s] = s < n_terms ? beta[s] : T(0);
  }
  a.aff = _affine;
  a.vec_bytes = (unsigned int)std::min<uint64_t>((uint64_t)_n_dofs * sizeof(T), 0xffffffffull);
  ASSERT_THROW((uint64_t)_rec.size() <= 0xffffffffull, "chunk records beyond 4 GiB: the multi-term sweep addresses them with 32-bit offsets");
  a.rec_total_bytes = (unsigned int)_rec.size();
  // a narrow last chunk column (mf_laplace.hip: at most 32 - 2 halo owned columns) is swept two y-tiles per workgroup by the
  // kernels that carry the body for it
  const bool narrow = narrow_last_column() && fused_narrow_capable(n_terms, ty);
  a.ntiles_y2 = narrow ? (a.ntiles_y + 1) / 2 : 0u;
  a.wide_tiles = (a.ncols - (narrow ? 1u : 0u)) * a.ntiles_y * a.ntiles_z;
  const uint64_t n_tiles = (uint64_t)a.wide_tiles + (uint64_t)a.ntiles_y2 * a.ntiles_z;
  ASSERT_THROW(n_tiles < (1ull << 30), "tile of the multi-term sweep too small for this mesh (grid size limit)");
  const unsigned int n_blocks = (unsigned int)(n_tiles >= 64 ? ((n_tiles + 7) / 8) * 8 : n_tiles);
  const size_t lds = ((size_t)nw * ring_planes(n_terms) * (ty + 1) + (size_t)2 * nw * 4) * 64 * sizeof(T);
  ASSERT_THROW(lds <= 160 * 1024, "tile of the multi-term sweep too large for the LDS");
  hipStream_t st = _handle.stream;
  // Algorithmic bytes of the launch: what its n_terms smoother terms require as launches of their own (x, out, one id, the
  // coefficient, b, x_prev: mf_laplace.hpp) -- the figure the one-term kernel is priced on, so that the two compare.  What the
  // sweep itself must move is fused_sweep_bytes(): x_0, b, the coefficient, x_K (+ x_{K-1}), the ids computed.
  double bytes = 0.;
  for (int k = 0; k < n_terms; ++k)
    bytes += required_bytes_apply() + epilogue_bytes(k == 0 ? 2 : 3);
  hipEvent_t stop = _handle.profiler.begin("mf_cheb_fused_kernel", bytes, st);
  auto go = [&](auto kernel) {
    static std::mutex attr_mutex;
    static std::set<std::pair<const void *, int>> attr_set;
    int dev = 0;
    MFMG_HIP_CHECK(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(attr_mutex);
      if (attr_set.insert({reinterpret_cast<const void *>(kernel), dev}).second)
        MFMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kernel, dim3(n_blocks), dim3(64 * nw), lds, st, a);
  };
  // the arithmetic of the cell kernel: mode space (default) or the bit-for-bit twin of the one-term kernel (set_fused_reference)
  const bool modes = !_fused_reference_arithmetic;
  auto pick = [&](auto kt, auto dt, auto mt) {
    constexpr int KK = decltype(kt)::value;
    constexpr bool DR = decltype(dt)::value != 0;
    constexpr bool MO = decltype(mt)::value != 0;
    // (fused_narrow_capable: these two shapes carry the second body)
    if constexpr (KK == 3)
      if (narrow && ty == 3)
      {
        go(mf_cheb_fused_kernel<T, 3, 3, DR, MO, 0, true>);
        return;
      }
    if constexpr (KK == 2)
      if (narrow && ty == 4)
      {
        go(mf_cheb_fused_kernel<T, 2, 4, DR, MO, 0, true>);
        return;
      }
    if (ty == 2)
      go(mf_cheb_fused_kernel<T, KK, 2, DR, MO>);
    else if (ty == 3)
      go(mf_cheb_fused_kernel<T, KK, 3, DR, MO>);
    else
      go(mf_cheb_fused_kernel<T, KK, 4, DR, MO>);
  };
  auto pick_k = [&](auto kt) {
    if (_dinv_in_record)
    {
      if (modes)
        pick(kt, IntTag<1>{}, IntTag<1>{});
      else
        pick(kt, IntTag<1>{}, IntTag<0>{});
    }
    else
    {
      if (modes)
        pick(kt, IntTag<0>{}, IntTag<1>{});
      else
        pick(kt, IntTag<0>{}, IntTag<0>{});
    }
  };
  static const int dbg = std::getenv("MFMG_MF_FUSED_DBG") ? std::atoi(std::getenv("MFMG_MF_FUSED_DBG")) : 0;
  if (zero_guess)
  {
    // (n_terms = 3, ty = 3, mode-space arithmetic: checked above)
    if (_dinv_in_record)
    {
      if (narrow)
        go(mf_cheb_fused_kernel<T, 3, 3, true, true, 0, true, true>);
      else
        go(mf_cheb_fused_kernel<T, 3, 3, true, true, 0, false, true>);
    }
    else
    {
      if (narrow)
        go(mf_cheb_fused_kernel<T, 3, 3, false, true, 0, true, true>);
      else
        go(mf_cheb_fused_kernel<T, 3, 3, false, true, 0, false, true>);
    }
  }
  else if (dbg > 0 && n_terms == 3 && ty == 3 && !_dinv_in_record && std::is_same<T, double>::value)
  {
    if (dbg == 1)
      go(mf_cheb_fused_kernel<T, 3, 3, false, true, 1>);
    else
      go(mf_cheb_fused_kernel<T, 3, 3, false, true, 2>);
  }
  else if (n_terms == 1)
    pick_k(IntTag<1>{});
  else if (n_terms == 2)
    pick_k(IntTag<2>{});
  else
    pick_k(IntTag<3>{});
  MFMG_HIP_CHECK(hipGetLastError());
  KernelProfiler::end(stop, st);
}

template bool MatrixFreeLaplaceDevice<double>::fused_zero_guess_available(int) const;
template bool MatrixFreeLaplaceDevice<float>::fused_zero_guess_available(int) const;
template bool MatrixFreeLaplaceDevice<double>::fused_sweep_available(int) const;
template bool MatrixFreeLaplaceDevice<float>::fused_sweep_available(int) const;
template void MatrixFreeLaplaceDevice<double>::choose_fused_tile(int, int &, int &, int &) const;
template void MatrixFreeLaplaceDevice<float>::choose_fused_tile(int, int &, int &, int &) const;
template void MatrixFreeLaplaceDevice<double>::smoother_sweep(int, double const *, double const *, double const *, double const *, double *,
                                                               double *) const;
template void MatrixFreeLaplaceDevice<float>::smoother_sweep(int, float const *, float const *, float const *, float const *, float *, float *) const;
} // namespace mfmg
