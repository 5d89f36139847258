// gfx950 kernel of the matrix-free Q1 Laplace operator.
//
// Data layout in HBM ("one cell slot per DoF", rows cut into aligned 64-slot chunks): the DoF
// columns are cut into runs of 62; chunk c of row (j,k) holds the 64 slots of the node columns
// i = 62c-1 .. 62c+62: lanes 1 .. 62 are the columns the chunk owns, lanes 0 and 63 repeat the last / first
// column of the neighbouring chunks (+3.2 % memory), so that a wavefront needs nothing from another chunk.
// Everything the kernel reads for one chunk sits in ONE contiguous record, so that a wavefront streams
// a single run of memory per row instead of one run per array:
//   chunk r(c,j,k) = (k Ny + j) ncols + c
//   rec  [r]  : own   int  [64]      DoF id of the slot's own DoF (corner (0,0,0) of its cell)
//               coef  16 B [NP][64]  the 8 quadrature coefficients of the cell (NP = 8 sizeof(T) / 16);
//                                    ONE value T[64] when the eight are equal in every cell
//               dinv  T    [64]      1 / diagonal entry of the slot's own DoF
// Slot (i,j,k) holds the cell whose lowest corner is DoF (i,j,k); cells that stick out of the
// mesh on a high face are phantoms with zero coefficient, slots outside the mesh carry id 0 and coefficient 0.
// The ids are the caller's global DoF ids (any numbering); bit 31 carries the Dirichlet flag, so the
// constrained-read-as-zero rule costs no extra load, bit 30 marks DoFs owned by another rank (read, never
// written).  The seven other corner ids of a cell are the own ids of neighbouring slots, so 4 B of ids are read
// per cell, not 32.
//
// Work decomposition (owner computes, no atomics, results independent of the tiling bit for bit):
// a workgroup of NW wavefronts marches over a tile of 64 node columns x NW TY cell rows x (TZ+1)
// cell layers and owns the 62 x (NW TY - 1) x TZ DoFs whose eight cells all lie inside (one halo
// column / row / layer on the low side is recomputed).  Wavefront w computes the TY cell rows
// [Yb + w TY, Yb + (w+1) TY) and hands the b=1 partial sums of its last row to wavefront w+1 through
// LDS (one barrier per layer) instead of letting w+1 recompute that row.
// The 8 corner contributions of a cell are combined
//   in x : shift by one lane of the right-face values (DPP),
//   in y : a register carried from the previous cell row (LDS hand-over between wavefronts),
//   in z : a per-lane column in LDS carried from the previous cell layer,
// so every DoF value is complete exactly when its own slot is visited and the smoother
// epilogue (b, D^-1, x_prev) is fused there: A x is never stored.  x is read once per wavefront and layer pass:
// the b=0 face of a cell is the b=1 face of the previous row (registers), the d=0 edge is the d=1 edge
// of the previous layer (a second per-lane LDS column), the a=1 corners are the a=0 corners of the
// next lane (DPP).
//
// Memory pipeline (see mf_laplace_body): one memory round trip per layer pass of a wavefront; vector accesses
// use a uniform (SGPR) base and a 32-bit per-lane byte offset.
#include "mf_laplace.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "mf_device.hpp"

namespace mfmg
{
namespace
{

// The tile body.  There is no per-lane control flow around the cell arithmetic: every lane of every row computes a
// cell.  What makes that legal is the data, not branches:
//   * slots without a real cell (the halo lanes past the last column, the last DoF row and layer) are stored with
//     coefficient zero and valid ids, so they contribute exact zeros; lane 63 is a halo NODE column only (its cell
//     would need column 64: it is computed on shifted garbage and nothing reads the result);
//   * the halo layer k = -1 of the first z-tile is evaluated at the clamped layer 0 with the coefficient forced to
//     zero (a wave-uniform select); the b = 1 face of the last DoF row is read at the clamped row Ny - 1 (its cells
//     are phantoms with coefficient zero);
//   * Dirichlet DoFs are zeroed ONCE, when a value enters the tile (one value per lane and node row), not at each
//     of the eight corners of each cell; the unmasked value of the row's own DoF rides along for the epilogue.
// Memory pipeline: ONE round trip per layer pass of a wavefront.  All requests of the pass -- the x values of the
// b=1 faces of its rows in layer k+1 (their ids were fetched during the previous pass), the coefficients, the
// epilogue operands of the rows it completes (their ids sit in LDS since the previous pass) and the ids of the
// next pass -- are issued before the first row is computed; the values of layer k come from LDS / registers.
// TYC > 0: rows per wavefront known at compile time (the row loop is fully unrolled: no loop-carried register
// moves, constant LDS offsets, the whole pass is one request batch for TYC <= 4); TYC = 0: rows from the
// arguments, one request batch per row.
template <typename T, int TYC, bool CC, int BATCH, bool AFF>
__device__ __forceinline__ void mf_laplace_body(MfArgs<T> const &a, unsigned int bid, MfBoxes const &bx, bool boxes)
{
#pragma clang fp contract(off)
  constexpr int B = TYC > 0 ? BATCH : 1; // cell rows per request batch (divides TYC)
  constexpr bool XP = TYC > 0;           // ids of the next pass are prefetched across the barrier
  static_assert(TYC == 0 || TYC % B == 0, "the batch must divide the rows per wavefront");
  const int TY = TYC > 0 ? TYC : a.TY;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform, keep it scalar
  const int NW = blockDim.x >> 6;
  // LDS: per wavefront pt[TY] (z-carry of the partial sums), xz[TY+1] (z-carry of x, one slot per node row),
  // then the hand-over rows of all wavefronts, then per wavefront idz[TY+1] (z-carry of the ids)
  T *pt = reinterpret_cast<T *>(smem_raw) + (size_t)wv * (3 * TY + 1) * 64;
  T *xz = pt + TY * 64;
  T *pc = xz + (TY + 1) * 64;                                                  // [TY][64] z-carry of the coefficient sums
  T *xport = reinterpret_cast<T *>(smem_raw) + (size_t)NW * (3 * TY + 1) * 64; // [2][NW][3][64]
  int *idz = reinterpret_cast<int *>(xport + (size_t)2 * NW * 3 * 64) + (size_t)wv * (TY + 1) * 64;
  // D^-1 on the fly (cell-constant layout, smoother modes): diag = kd * sum of the coefficients of the 8 cells of a DoF
  const bool make_dinv = CC && a.mode >= 2 && !a.dinv_in_record;

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8,
  // observed, speed only); give every XCD a contiguous run of the tile list.
  const unsigned int n_tiles = boxes ? bx.bx_end[5] : a.ncols_active * a.ntiles_y * a.ntiles_z;
  unsigned int w = bid;
  if (n_tiles >= 64)
  {
    const unsigned int per_xcd = (n_tiles + 7) / 8;
    if (!boxes)
      w = (bid % 8) * per_xcd + bid / 8;
    if (w >= n_tiles)
      return; // the whole workgroup leaves: no barrier is left waiting
  }
  // (the box of a tile list: wave-uniform selects over constant indices -- no dynamic indexing of the argument block)
  unsigned int c0 = a.col0, nc = a.ncols_active, y0 = a.ty0, ny = a.ntiles_y, z0 = a.z_tile0;
  if (boxes)
  {
    unsigned int first = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q)
    {
      const unsigned int lo = q > 0 ? bx.bx_end[q - 1] : 0u;
      if (w >= lo && w < bx.bx_end[q])
      {
        first = lo;
        c0 = bx.bx_c0[q];
        nc = bx.bx_nc[q];
        y0 = bx.bx_y0[q];
        ny = bx.bx_ny[q];
        z0 = bx.bx_z0[q];
      }
    }
    w -= first;
  }
  const int tc = c0 + w % nc;
  const int tyi = y0 + (w / nc) % ny;
  const int tzi = z0 + w / (nc * ny);
  const int ci = tc * a.own - a.halo + lane;        // cell / DoF column of this lane
  const int Yb = tyi * (NW * TY - 1) - 1 + wv * TY; // first cell row of this wavefront
  const int Z0 = a.ztab[tzi];            // (wave-uniform index: scalar loads)
  const int TZt = a.ztab[tzi + 1] - Z0;  // layers this tile owns
  const bool col_owned = lane >= a.halo && lane < a.halo + a.own && ci < a.Nx;
  const int jj0 = (Yb < 0) ? 1 : 0; // cell row -1 does not exist (its sums are the zero initial carries)
  const size_t rec_row = (size_t)a.ncols * a.rec_bytes;
  const size_t rec_layer = (size_t)a.Ny * rec_row;
  unsigned char const *rec_col = a.rec + (size_t)tc * a.rec_bytes;
  const bool have_rows = (jj0 < TY) && (Yb + jj0 < a.Ny);
  CellFactors<T> fac;
  fac.fx = a.fx;
  fac.fy = a.fy;
  fac.fz = a.fz;
  fac.fax = a.fax;
  fac.fbx = a.fbx;
  fac.fay = a.fay;
  fac.fby = a.fby;
  fac.faz = a.faz;
  fac.fbz = a.fbz;
  // own id of node row `jrow` (clamped into the mesh) in the layer whose records start at `layer`
  const int id_lane = a.aff.base + ci * a.aff.s0; // (AFF) the part of the id that belongs to the lane
  const bool lane_in = ci >= 0 && ci < a.Nx;
  const bool cell_lane = ci >= 0 && ci < a.Nx - 1; // the slot holds a real cell (the others carry coefficient zero)
  const bool lane_face = ((a.aff.faces & 1) && ci == 0) || ((a.aff.faces & 2) && ci == a.Nx - 1);
  // (one rank: no ghost planes at all -- the tests below would cost the eight-coefficient kernel a dozen scalar operations per row)
  const bool any_ghost = (a.aff.ghost_lo[0] | a.aff.ghost_hi[0] | a.aff.ghost_lo[1] | a.aff.ghost_hi[1] | a.aff.ghost_lo[2] | a.aff.ghost_hi[2]) != 0;
  const bool lane_ghost = any_ghost && (ci < a.aff.ghost_lo[0] || ci >= a.Nx - a.aff.ghost_hi[0]);
  auto own_id = [&](int kz, int jrow) {
    const int jr = min(max(jrow, 0), a.Ny - 1);
    if constexpr (AFF)
    {
      // wave-uniform: the row and layer part of the id, the faces j and k, the ghost layers
      const int id_row = jr * a.aff.s1 + kz * a.aff.s2;
      const bool row_face = ((a.aff.faces & 4) && jr == 0) || ((a.aff.faces & 8) && jr == a.Ny - 1) ||
                            ((a.aff.faces & 16) && kz == 0) || ((a.aff.faces & 32) && kz == a.Nz - 1);
      const bool row_ghost = any_ghost && (kz < a.aff.ghost_lo[2] || kz >= a.Nz - a.aff.ghost_hi[2] || jr < a.aff.ghost_lo[1] || jr >= a.Ny - a.aff.ghost_hi[1]);
      // (a node on a Dirichlet face carries the Dirichlet flag only, also in a ghost plane: mfmg_amd/distributed.py, local_problem)
      const bool face = lane_face || row_face;
      const int id = (id_lane + id_row) | (face ? (int)kFlag : 0) | (((lane_ghost || row_ghost) && !face) ? (int)kGhost : 0);
      return lane_in ? id : 0; // (lanes outside the mesh carry id 0 in the records too)
    }
    else
      return reinterpret_cast<int const *>(rec_col + (size_t)kz * rec_layer + (size_t)jr * rec_row)[lane];
  };
  int pf[XP ? TYC + 1 : 1]; // own ids of the node rows Yb .. Yb + TY in the layer after next
#pragma unroll
  for (int r = 0; r < (XP ? TYC + 1 : 1); ++r)
    pf[r] = 0;

  for (int kk = 0; kk <= TZt; ++kk)
  {
    const int k = Z0 - 1 + kk;
    if (k >= a.Nz)
      break; // uniform over the workgroup
    const int kc = max(k, 0);            // layer the records of "layer k" are read from
    const int kn = min(k + 1, a.Nz - 1); // ... and those of layer k + 1
    const bool no_cells = k < 0;         // the halo layer below the mesh: coefficient forced to zero
    const bool layer_carry = kk > 0;     // xz / idz hold x and the ids of layer k, written by the pass of layer k - 1
    unsigned char const *rec_k = rec_col + (size_t)kc * rec_layer;
    T ry0 = T(0), ry1 = T(0), rc = T(0); // (rc: coefficient sum of the previous cell row)
    // b=0 face carried from the previous cell row: Dirichlet-masked x values of the four corners, the raw value
    // and the id of the own DoF
    T cxm[4] = {T(0), T(0), T(0), T(0)};
    T cx0 = T(0);
    int id0 = 0;
    // first DoF row of a wavefront w > 0, finished after the barrier: sums, raw x, id, epilogue operands
    T d00 = T(0), d01 = T(0), dx0 = T(0), dlb = T(0), dld = T(0), dlx = T(0), dsc = T(0);
    int did0 = 0;

    // ---- first node row of the pass (the b=0 face of its first cell row)
    if (have_rows)
    {
      const int jf = Yb + jj0;
      int idBf, idAf;
      T xAf;
      if (layer_carry)
      {
        idAf = idz[jj0 * 64 + lane];
        xAf = xz[jj0 * 64 + lane];
        if constexpr (XP)
          idBf = jj0 ? pf[XP ? 1 : 0] : pf[0]; // (no run-time index into the register array)
        else
          idBf = own_id(kn, jf);
      }
      else
      {
        idAf = own_id(kc, jf);
        idBf = own_id(kn, jf);
        xAf = ld_off<T>(a.x, id_off<T>(idAf));
      }
      const T xBf = ld_off<T>(a.x, id_off<T>(idBf));
      if (wv > 0 && kk > 0 && a.mode != 0)
      {
        // epilogue operands of the deferred row (its id has been in LDS since the previous pass)
        const unsigned int g = id_off<T>(idAf);
        dlb = ld_off<T>(a.b, g);
        if (a.mode >= 2 && a.dinv_in_record)
          dld = reinterpret_cast<T const *>(rec_k + (size_t)jf * rec_row + Rec<T, CC>::kDinvOff)[lane];
        if (a.mode == 3)
          dlx = ld_off<T>(a.xprev, g);
      }
      xz[jj0 * 64 + lane] = xBf;
      idz[jj0 * 64 + lane] = idBf;
      id0 = idAf;
      cx0 = xAf;
      cxm[0] = masked<T>(xAf, idAf);
      cxm[2] = masked<T>(xBf, idBf);
      cxm[1] = from_next_lane(cxm[0]);
      cxm[3] = from_next_lane(cxm[2]);
    }

#pragma unroll
    for (int g = 0; g < TY; g += B)
    {
      // ---- every request of the batch: node rows g+1 .. g+B (the b=1 faces of the cell rows g .. g+B-1)
      bool ok[B];
      int idA[B], idB[B];
      T xA[B], xB[B], lb[B], ld[B], lxp[B];
      T c[B][CC ? 1 : 8];
#pragma unroll
      for (int b = 0; b < B; ++b)
      {
        const int jj = g + b, j = Yb + jj;
        ok[b] = jj >= jj0 && jj < TY && j < a.Ny;
        idA[b] = idB[b] = 0;
        xA[b] = xB[b] = lb[b] = ld[b] = lxp[b] = T(0);
        if (!ok[b])
          continue;
        if (layer_carry)
        {
          idA[b] = idz[(jj + 1) * 64 + lane];
          xA[b] = xz[(jj + 1) * 64 + lane];
          if constexpr (XP)
            idB[b] = pf[XP ? g + b + 1 : 0];
          else
            idB[b] = own_id(kn, j + 1);
        }
        else
        {
          idA[b] = own_id(kc, j + 1);
          idB[b] = own_id(kn, j + 1);
        }
      }
#pragma unroll
      for (int b = 0; b < B; ++b)
      {
        if (!ok[b])
          continue;
        const int jj = g + b, j = Yb + jj;
        unsigned char const *recp = rec_k + (size_t)j * rec_row;
        xB[b] = ld_off<T>(a.x, id_off<T>(idB[b]));
        if (!layer_carry)
          xA[b] = ld_off<T>(a.x, id_off<T>(idA[b]));
        if constexpr (CC)
          c[b][0] = reinterpret_cast<T const *>(recp + Rec<T, true>::kCoefOff)[lane];
        else
        {
          // slots without a real cell (the lanes past the last cell column: 54 of the 64 lanes of the last chunk at 257 DoFs per
          // row; the last DoF row and layer) are stored as zeros: do not fetch them (64 bytes per slot; the same bits)
#pragma unroll
          for (int q = 0; q < 8; ++q)
            c[b][q] = T(0);
          if (cell_lane && j < a.Ny - 1 && k >= 0 && k < a.Nz - 1)
            load_coef<T, false>(recp, lane, c[b]);
        }
        if (jj > 0 && kk > 0 && a.mode != 0)
        {
          // the DoF row this cell row completes: node row jj, id = the layer-k id of the previous node row
          const unsigned int gid = id_off<T>((b == 0 || !ok[b > 0 ? b - 1 : 0]) ? id0 : idA[b > 0 ? b - 1 : 0]);
          lb[b] = ld_off<T>(a.b, gid);
          if (a.mode >= 2 && a.dinv_in_record)
            ld[b] = reinterpret_cast<T const *>(recp + Rec<T, CC>::kDinvOff)[lane];
          if (a.mode == 3)
            lxp[b] = ld_off<T>(a.xprev, gid);
        }
      }
      if constexpr (XP)
      {
        // ids of the next pass (layer k + 2): the node rows whose current ids this batch has just consumed
        if (have_rows)
        {
          if (g == 0)
            pf[0] = own_id(min(k + 2, a.Nz - 1), Yb);
#pragma unroll
          for (int b = 0; b < B; ++b)
            pf[XP ? g + b + 1 : 0] = own_id(min(k + 2, a.Nz - 1), Yb + g + b + 1);
        }
      }

      // ---- the rows of the batch
#pragma unroll
      for (int b = 0; b < B; ++b)
      {
        if (!ok[b])
          continue;
        const int jj = g + b;
        xz[(jj + 1) * 64 + lane] = xB[b];
        idz[(jj + 1) * 64 + lane] = idB[b];
        const T n0m = masked<T>(xA[b], idA[b]), n2m = masked<T>(xB[b], idB[b]);
        const T n1m = from_next_lane(n0m), n3m = from_next_lane(n2m);
        T u[8], v[8];
        u[0] = cxm[0];
        u[1] = cxm[1];
        u[4] = cxm[2];
        u[5] = cxm[3];
        u[2] = n0m;
        u[3] = n1m;
        u[6] = n2m;
        u[7] = n3m;
        T sx = T(0); // coefficients of the two cells of this row and layer that touch DoF column ci
        if constexpr (CC)
        {
          const T cv = no_cells ? T(0) : c[b][0];
          cell_apply_cc<T>(u, cv, fac, v);
          if (make_dinv)
            sx = cv + from_prev_lane(cv);
        }
        else
        {
          if (no_cells)
          {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              c[b][q] = T(0);
          }
          cell_apply<T>(u, c[b], fac, v);
        }
        const T x0 = cx0;
        const int idr = id0;
        cxm[0] = n0m;
        cxm[1] = n1m;
        cxm[2] = n2m;
        cxm[3] = n3m;
        cx0 = xA[b];
        id0 = idA[b];

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
          did0 = idr;
          dsc = sx;
        }
        else
        {
          // ---- y combine (register carry), z combine (LDS column carry)
          const T t0 = s00 + ry0;
          const T t1 = s01 + ry1;
          T *ptj = pt + jj * 64 + lane;
          const T yv = t0 + *ptj; // (layer kk = 0 reads what an earlier tile left there: never stored)
          *ptj = t1;
          T dinv_here = ld[b];
          if (make_dinv)
          {
            // the eight cells of the DoF: two rows of this layer (register carry) + the same of the layer below (LDS)
            const T tc = sx + rc;
            T *pcj = pc + jj * 64 + lane;
            const T sum8 = tc + *pcj;
            *pcj = tc;
            dinv_here = (idr < 0) ? T(1) : T(1) / (a.kd * sum8);
          }
          if (jj > 0 && kk > 0 && col_owned && !((unsigned int)idr & kGhost))
            st_off<T>(a.out, id_off<T>(idr), mf_epilogue<T>(a, idr, x0, yv, lb[b], dinv_here, lxp[b]));
        }
        ry0 = s10;
        ry1 = s11;
        rc = sx;
      }
    }
    if (NW > 1)
    {
      // exports are double-buffered by layer parity: a slot written in layer k is read after barrier k
      // and rewritten in layer k+2, i.e. after barrier k+1, which the reader only passes once it has read
      T *xp = xport + ((size_t)((kk & 1) * NW + wv) * 3) * 64 + lane;
      xp[0] = ry0;
      xp[64] = ry1;
      if (make_dinv)
        xp[128] = rc;
      __syncthreads();
      if (wv > 0)
      {
        T const *ip = xport + ((size_t)((kk & 1) * NW + wv - 1) * 3) * 64 + lane;
        const T t0 = d00 + ip[0];
        const T t1 = d01 + ip[64];
        T *ptj = pt + lane;
        const T yv = t0 + *ptj;
        *ptj = t1;
        if (make_dinv)
        {
          const T tc = dsc + ip[128];
          const T sum8 = tc + pc[lane];
          pc[lane] = tc;
          dld = (did0 < 0) ? T(1) : T(1) / (a.kd * sum8);
        }
        if (kk > 0 && have_rows && col_owned && !((unsigned int)did0 & kGhost))
          st_off<T>(a.out, id_off<T>(did0), mf_epilogue<T>(a, did0, dx0, yv, dlb, dld, dlx));
      }
    }
  }
}

// One launch can carry two meshes: the first `n_tail_blocks` workgroups work on `at` (the rotated slab of the
// tail columns, see the constructor), the others on `am`.  Both share the tile shape (NW, TY, TZ).
// BATCH = cell rows whose requests are issued together (one memory round trip per batch): the whole pass for the
// cell-constant variant, fewer for eight coefficients per cell (16 VGPRs of coefficients per row in FP64).
// AFF: the ids are computed from the position (AffineIds) instead of read from the records.
template <typename T, int TYC, bool CC, int BATCH, bool AFF = false>
__global__ __launch_bounds__(512) void mf_laplace_kernel(MfArgs<T> am, MfArgs<T> at, unsigned int n_tail_blocks, MfBoxes boxes)
{
  const bool tail = blockIdx.x < n_tail_blocks;
  mf_laplace_body<T, TYC, CC, BATCH, AFF>(tail ? at : am, tail ? blockIdx.x : blockIdx.x - n_tail_blocks, boxes,
                                          !tail && boxes.n_boxes > 0);
}

// ---- setup kernels -----------------------------------------------------------
// *mismatch is raised when a slot of the records does not carry the id AffineIds gives it
__global__ void mf_affine_check_kernel(unsigned char const *rec, size_t rec_bytes, int64_t n_slots, int Nx, int Ny, int Nz, int ncols,
                                       int own, int halo, AffineIds f, int *mismatch)
{
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n_slots; s += (int64_t)gridDim.x * blockDim.x)
  {
    const int lane = s & 63;
    const int64_t chunk = s >> 6;
    const int c = chunk % ncols, j = (chunk / ncols) % Ny, k = chunk / ((int64_t)ncols * Ny);
    const int i = c * own - halo + lane;
    if (i < 0 || i >= Nx)
      continue;
    const bool face = ((f.faces & 1) && i == 0) || ((f.faces & 2) && i == Nx - 1) || ((f.faces & 4) && j == 0) ||
                      ((f.faces & 8) && j == Ny - 1) || ((f.faces & 16) && k == 0) || ((f.faces & 32) && k == Nz - 1);
    const bool ghost = i < f.ghost_lo[0] || i >= Nx - f.ghost_hi[0] || j < f.ghost_lo[1] || j >= Ny - f.ghost_hi[1] || k < f.ghost_lo[2] ||
                       k >= Nz - f.ghost_hi[2];
    const int want = (f.base + i * f.s0 + j * f.s1 + k * f.s2) | (face ? (int)kFlag : 0) | ((ghost && !face) ? (int)kGhost : 0);
    if (reinterpret_cast<int const *>(rec + (size_t)chunk * rec_bytes)[lane] != want)
      atomicOr(mismatch, 1);
  }
}

template <typename T, bool CC>
__global__ void mf_repack_kernel(int32_t const *cell_dofs, double const *coefficient,
                                 uint8_t const *constrained, int Nx, int Ny, int Nz, int ncols, int own, int halo,
                                 unsigned char *rec, size_t rec_bytes, int with_dinv)
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
    const int i = c * own - halo + lane;
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
    unsigned char *r = rec + (size_t)chunk * rec_bytes;
    reinterpret_cast<int *>(r)[lane] = id[0];
    if constexpr (CC)
      reinterpret_cast<T *>(r + Rec<T, CC>::kCoefOff)[lane] = cf[0];
    else
      for (int p = 0; p < NP; ++p)
        for (int w = 0; w < W; ++w)
          reinterpret_cast<T *>(r + Rec<T, CC>::kCoefOff + p * 1024)[lane * W + w] = cf[p * W + w];
    if (with_dinv)
      reinterpret_cast<T *>(r + Rec<T, CC>::kDinvOff)[lane] = T(0);
  }
}

// chunk and lane of cell / DoF (i,j,k) (the copy owned by its column)
__device__ __forceinline__ size_t chunk_of(int i, int j, int k, int Ny, int ncols, int own, int halo, int &lane)
{
  const int c = min(i / own, ncols - 1); // chunk c owns the columns own c .. own c + own - 1 in its lanes halo .. halo + own - 1
  lane = i + halo - own * c;
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
__global__ void mf_validate_kernel(int32_t const *cell_dofs, unsigned char const *rec, size_t rec_bytes, int Nx, int Ny, int Nz, int ncols,
                                   int own, int halo, int64_t n_dofs, int *n_bad)
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
      const size_t r = chunk_of(i + (m & 1), j + ((m >> 1) & 1), k + (m >> 2), Ny, ncols, own, halo, lane);
      if ((int)((unsigned int)reinterpret_cast<int const *>(rec + r * rec_bytes)[lane] & kIdMask) != g)
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
__global__ void mf_diagonal_kernel(unsigned char const *rec, size_t rec_bytes, int Nx, int Ny, int Nz, int ncols, int own, int halo,
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
    const size_t r = chunk_of(i, j, k, Ny, ncols, own, halo, lane);
    const int id0 = reinterpret_cast<int const *>(rec + r * rec_bytes)[lane];
    double sum = 0.;
    for (int m = 0; m < 8; ++m)
    {
      const int ci = i - (m & 1), cj = j - ((m >> 1) & 1), ck = k - (m >> 2);
      if (ci < 0 || cj < 0 || ck < 0 || ci >= Nx - 1 || cj >= Ny - 1 || ck >= Nz - 1)
        continue;
      T c[8];
      int cl;
      const size_t cr = chunk_of(ci, cj, ck, Ny, ncols, own, halo, cl);
      load_coef<T, CC>(rec + cr * rec_bytes, cl, c);
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
__global__ void mf_fill_dinv_kernel(T const *dinv, int Nx, int ncols, int own, int halo, int64_t n_slots,
                                    unsigned char *rec, size_t rec_bytes)
{
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n_slots;
       s += (int64_t)gridDim.x * blockDim.x)
  {
    const int lane = s & 63;
    const int64_t chunk = s >> 6;
    const int i = (int)(chunk % ncols) * own - halo + lane;
    if (i < 0 || i >= Nx)
      continue;
    unsigned char *r = rec + (size_t)chunk * rec_bytes;
    const unsigned int g = (unsigned int)reinterpret_cast<int const *>(r)[lane] & kIdMask;
    reinterpret_cast<T *>(r + Rec<T, CC>::kDinvOff)[lane] = dinv[g];
  }
}
} // namespace

// ---- dim = 2 --------------------------------------------------------------------------------------------------
// The reference exercises its matrix-free path in 2-D too (tests/test_hierarchy.cc:416-443, LaplaceMatrixFree<2>);
// those meshes are small (3 to 5 global refinements), so the 2-D operator is a plain owner-computes kernel: one
// thread per node gathers the 4 corners of its (up to) 4 cells, A_e[m][n] = sum_q c(cell, q) K[q][m][n] formed on the
// fly from the 64 reference constants, with the same fused epilogues and the same constrained-row rule
// (tests/laplace_matrix_free.hpp:121-156: constrained DoFs read as zero, their rows are identities).
struct Mf2dTable
{
  double K[4][4][4]; // [q][m][n] (amge_structured.cpp: reference_cell_tables)
};

template <typename T>
struct Mf2dArgs
{
  int32_t const *cell_dofs; // [n_cells][4]
  T const *coef;            // [n_cells][4]
  uint8_t const *constrained;
  int32_t const *node_dof;  // [Nx Ny]
  int nx, ny;
  T const *x, *b, *dinv, *xprev;
  T *out;
  T alpha, beta;
  int mode;
};

__global__ void mf2d_node_dof_kernel(int32_t const *cell_dofs, int nx, int ny, int32_t *node_dof)
{
  const int Nx = nx + 1, Ny = ny + 1;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Nx * Ny; t += gridDim.x * blockDim.x)
  {
    const int i = t % Nx, j = t / Nx;
    const int ic = min(i, nx - 1), jc = min(j, ny - 1);
    node_dof[t] = cell_dofs[(size_t)(ic + nx * jc) * 4 + (i - ic) + 2 * (j - jc)];
  }
}

// every corner of every cell must be the DoF its node carries (a logically structured mesh in lexicographic cell order)
__global__ void mf2d_validate_kernel(int32_t const *cell_dofs, int32_t const *node_dof, int nx, int ny, int64_t n_dofs, int *n_bad)
{
  const int Nx = nx + 1;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nx * ny; c += gridDim.x * blockDim.x)
  {
    const int i = c % nx, j = c / nx;
    for (int m = 0; m < 4; ++m)
    {
      const int g = cell_dofs[(size_t)c * 4 + m];
      if (g < 0 || g >= n_dofs || node_dof[(i + (m & 1)) + Nx * (j + (m >> 1))] != g)
        atomicAdd(n_bad, 1);
    }
  }
}

template <typename T>
__global__ void mf2d_diagonal_kernel(Mf2dArgs<T> a, Mf2dTable tab, T *diag, T *dinv)
{
  const int Nx = a.nx + 1, Ny = a.ny + 1;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Nx * Ny; t += gridDim.x * blockDim.x)
  {
    const int i = t % Nx, j = t / Nx;
    const int g = a.node_dof[t];
    double d = 0.;
    for (int m = 0; m < 4; ++m)
    {
      const int ci = i - (m & 1), cj = j - (m >> 1); // the cell that has this node as its corner m
      if (ci < 0 || ci >= a.nx || cj < 0 || cj >= a.ny)
        continue;
      const size_t c = (size_t)(ci + a.nx * cj);
      for (int q = 0; q < 4; ++q)
        d += (double)a.coef[c * 4 + q] * tab.K[q][m][m];
    }
    if (a.constrained[g] & 1)
      d = 1.;
    diag[g] = T(d);
    dinv[g] = T(1. / d);
  }
}

template <typename T>
__global__ void mf2d_apply_kernel(Mf2dArgs<T> a, Mf2dTable tab)
{
#pragma clang fp contract(off)
  const int Nx = a.nx + 1, Ny = a.ny + 1;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Nx * Ny; t += gridDim.x * blockDim.x)
  {
    const int i = t % Nx, j = t / Nx;
    const int g = a.node_dof[t];
    const uint8_t flag = a.constrained[g];
    if (flag & 2)
      continue; // owned by another rank: read, never written
    const T x0 = a.x[g];
    T ax = x0; // constrained rows: dst_c = src_c
    if (!(flag & 1))
    {
      T sum = T(0);
      for (int m = 0; m < 4; ++m)
      {
        const int ci = i - (m & 1), cj = j - (m >> 1);
        if (ci < 0 || ci >= a.nx || cj < 0 || cj >= a.ny)
          continue;
        const size_t c = (size_t)(ci + a.nx * cj);
        for (int n = 0; n < 4; ++n)
        {
          const int gn = a.cell_dofs[c * 4 + n];
          const T u = (a.constrained[gn] & 1) ? T(0) : a.x[gn];
          T e = T(0);
          for (int q = 0; q < 4; ++q)
            e += a.coef[c * 4 + q] * T(tab.K[q][m][n]);
          sum += e * u;
        }
      }
      ax = sum;
    }
    T o;
    if (a.mode == 0)
      o = ax;
    else if (a.mode == 1)
      o = ax - a.b[g];
    else
    {
      const T r = ax - a.b[g];
      const T wgt = -(a.beta * a.dinv[g]);
      o = (a.mode == 2) ? wgt * r + x0 : wgt * r + (a.alpha * (x0 - a.xprev[g]) + x0);
    }
    a.out[g] = o;
  }
}

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
  MemoryKind kind("matrix-free operator: chunk records, diagonal");
  if (mesh.dim != 3 && mesh.dim != 2)
    ASSERT_THROW_NOT_IMPLEMENTED("the matrix-free HIP operator is implemented for dim = 2 and dim = 3");
  ASSERT_THROW(mesh.cell_dofs && mesh.coefficient && mesh.constrained,
               "mesh description arrays must not be null");
  if (mesh.dim == 2)
  {
    init_2d(mesh);
    return;
  }
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
  // Chunk geometry.  One halo lane on either side (62 owned columns, the tail columns of a 2^k + 1 wide row as a rotated slab)
  // serves the kernels that run one polynomial term per launch; the multi-term sweep (mf_cheb_fused.hip) needs as many halo lanes
  // as it runs terms, and then every chunk owns the same number of columns (no tail slab: 257 columns = 5 chunks of 52).
  // The slab operator and eight coefficients per cell keep the one-lane geometry (their smoother runs term by term).
  // (a distributed rank holds two ghost planes of a lower neighbour, or four: at most two resp. three terms per sweep)
  _halo = (sub_mesh || !_compact) ? 1 : std::min(std::max(handle.mf_fused_terms, 1), handle.comm.enabled() ? handle.comm.sweep_terms() : 3);
  if (_halo == 1)
  {
    _own = 62;
    _ncols = (_N[0] + _own - 1) / _own;
  }
  else
  {
    _ncols = (_N[0] + (64 - 2 * _halo) - 1) / (64 - 2 * _halo);
    _own = (_N[0] + _ncols - 1) / _ncols;
    // ... unless full chunks leave a last one of at most 32 - halo columns (257 = 4 x 58 + 25; the 259 / 261 columns of a rank's
    // local mesh: + 27 / 29): the sweep runs that column two y-tiles per workgroup, one per half of the wavefront (NARROW in
    // mf_cheb_fused.hip), 4.5 columns of tiles instead of 5.  The column ends at the high face of the mesh: a half needs the halo
    // lanes below its owned columns only (the lane above the last one belongs to the other half: it enters through a phantom cell,
    // coefficient zero).
    const int full = 64 - 2 * _halo, rest = _N[0] - (_ncols - 1) * full;
    static const bool narrow_off = std::getenv("MFMG_MF_NARROW") && std::string(std::getenv("MFMG_MF_NARROW")) == "0";
    // (FP64 only: the FP32 sweep was slower with it, 1.08 against 1.02 ms per cycle at 257^3 -- one round of long tiles hides less)
    if (_ncols >= 2 && rest >= 1 && rest <= 32 - _halo && !narrow_off && sizeof(T) == 8)
    {
      _own = full;
      _narrow_last = true;
    }
  }
  const size_t n_slots = (size_t)_ncols * _N[2] * _N[1] * 64;
  _n_slots = n_slots;
  _dinv_in_record = !_compact || handle.stored_diagonal;
  _rec_bytes = _compact ? Rec<T, true>::bytes(_dinv_in_record) : Rec<T, false>::bytes(true);
  _rec.resize((n_slots / 64) * _rec_bytes);
  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_range_kernel, dim3(n_blocks_for(nc * 8, 256, 1 << 16)), dim3(256), 0, st, cd, nc * 8, nd,
                     bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  ASSERT_THROW(bad.download(st)[0] == 0, "cell_dofs is not a logically structured hex mesh in lexicographic cell "
                                         "order (DoF ids out of range)");
  if (_compact)
    hipLaunchKernelGGL((mf_repack_kernel<T, true>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st, cd,
                       co, cn, _N[0], _N[1], _N[2], _ncols, _own, _halo, _rec.data(), _rec_bytes, _dinv_in_record ? 1 : 0);
  else
    hipLaunchKernelGGL((mf_repack_kernel<T, false>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st, cd,
                       co, cn, _N[0], _N[1], _N[2], _ncols, _own, _halo, _rec.data(), _rec_bytes, _dinv_in_record ? 1 : 0);
  MFMG_HIP_CHECK(hipGetLastError());

  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_validate_kernel, dim3(n_blocks_for(nc, 256, 1 << 16)), dim3(256), 0, st, cd,
                     _rec.data(), _rec_bytes, _N[0], _N[1], _N[2], _ncols, _own, _halo, _n_dofs,
                     bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  int n_bad = bad.download(st)[0];
  ASSERT_THROW(n_bad == 0, "cell_dofs is not a logically structured hex mesh in lexicographic cell order (" +
                               std::to_string(n_bad) + " inconsistent cells)");

  // a numbering the kernel can compute (AffineIds): parameters from a few slots of the records, then every slot checked
  {
    char const *env = std::getenv("MFMG_MF_AFFINE_IDS");
    if (!(env && std::string(env) == "0") && _N[0] >= 3 && _N[1] >= 3 && _N[2] >= 3)
    {
      auto slot_id = [&](int i, int j, int k) {
        const int c = std::min(i / _own, _ncols - 1);
        const int64_t chunk = c + (int64_t)_ncols * (j + (int64_t)_N[1] * k);
        int v = 0;
        MFMG_HIP_CHECK(hipMemcpyAsync(&v, _rec.data() + (size_t)chunk * _rec_bytes + (size_t)(i + _halo - _own * c) * sizeof(int), sizeof(int),
                                      hipMemcpyDeviceToHost, st));
        MFMG_HIP_CHECK(hipStreamSynchronize(st));
        return (unsigned int)v;
      };
      AffineIds f;
      const int mi = _N[0] / 2, mj = _N[1] / 2, mk = _N[2] / 2;
      const int g0 = (int)(slot_id(mi, mj, mk) & kIdMask);
      f.s0 = (int)(slot_id(mi + 1, mj, mk) & kIdMask) - g0;
      f.s1 = (int)(slot_id(mi, mj + 1, mk) & kIdMask) - g0;
      f.s2 = (int)(slot_id(mi, mj, mk + 1) & kIdMask) - g0;
      f.base = g0 - mi * f.s0 - mj * f.s1 - mk * f.s2;
      f.faces = ((slot_id(0, mj, mk) & kFlag) ? 1 : 0) | ((slot_id(_N[0] - 1, mj, mk) & kFlag) ? 2 : 0) |
                ((slot_id(mi, 0, mk) & kFlag) ? 4 : 0) | ((slot_id(mi, _N[1] - 1, mk) & kFlag) ? 8 : 0) |
                ((slot_id(mi, mj, 0) & kFlag) ? 16 : 0) | ((slot_id(mi, mj, _N[2] - 1) & kFlag) ? 32 : 0);
      // ghost planes per axis, read along the three centre lines
      for (int d = 0; d < 3; ++d)
      {
        auto at = [&](int t) { return d == 0 ? slot_id(t, mj, mk) : (d == 1 ? slot_id(mi, t, mk) : slot_id(mi, mj, t)); };
        f.ghost_lo[d] = f.ghost_hi[d] = 0;
        while (f.ghost_lo[d] < std::min(_N[d] / 2, 4) && (at(f.ghost_lo[d]) & kGhost))
          ++f.ghost_lo[d];
        while (f.ghost_hi[d] < std::min(_N[d] / 2, 4) && (at(_N[d] - 1 - f.ghost_hi[d]) & kGhost))
          ++f.ghost_hi[d];
      }
      MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
      hipLaunchKernelGGL(mf_affine_check_kernel, dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st, _rec.data(), _rec_bytes,
                         (int64_t)n_slots, _N[0], _N[1], _N[2], _ncols, _own, _halo, f, bad.data());
      MFMG_HIP_CHECK(hipGetLastError());
      _affine_ids = bad.download(st)[0] == 0;
      _affine = f;
    }
  }

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
                       _rec.data(), _rec_bytes, _N[0], _N[1], _N[2], _ncols, _own, _halo, tab, _diag.data(), _dinv.data());
    if (_dinv_in_record)
      hipLaunchKernelGGL((mf_fill_dinv_kernel<T, true>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st,
                         _dinv.data(), _N[0], _ncols, _own, _halo, (int64_t)n_slots, _rec.data(), _rec_bytes);
  }
  else
  {
    hipLaunchKernelGGL((mf_diagonal_kernel<T, false>), dim3(n_blocks_for(nd, 256, 1 << 16)), dim3(256), 0, st,
                       _rec.data(), _rec_bytes, _N[0], _N[1], _N[2], _ncols, _own, _halo, tab, _diag.data(), _dinv.data());
    hipLaunchKernelGGL((mf_fill_dinv_kernel<T, false>), dim3(n_blocks_for(n_slots, 256, 1 << 16)), dim3(256), 0, st,
                       _dinv.data(), _N[0], _ncols, _own, _halo, (int64_t)n_slots, _rec.data(), _rec_bytes);
  }
  MFMG_HIP_CHECK(hipGetLastError());
  MFMG_HIP_CHECK(hipStreamSynchronize(st));

  // ---- nearly empty last chunk: hand its columns to a rotated slab operator
  // A row of 2^k + 1 DoFs needs one chunk more than 2^k columns fill (257 = 4 * 63 + 5); the wavefronts of that
  // chunk issue the full instruction stream for 5 of 63 columns, which costs where the kernel is bound by
  // instruction issue (the cell-constant variant).  Those columns (plus the last column of the previous chunk as
  // a halo that is read, not written) form a thin slab whose LONG direction is y: the same kernel runs on it
  // with x and y exchanged, lanes along y, inside the same launch, and the main part skips the last chunk.
  const int tail_cols = _N[0] - _own * (_ncols - 1); // DoF columns owned by the last chunk
  if (!sub_mesh && _halo == 1 && _compact && _ncols >= 2 && tail_cols <= 24 && _N[1] >= 64)
  {
    const int i0 = _own * (_ncols - 1) - 1; // halo column of the last chunk = last column of the chunk before
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

template <typename T>
void MatrixFreeLaplaceDevice<T>::init_2d(mfmg_hip_mesh_desc const &mesh)
{
  _dim = 2;
  int64_t nd = 1, nc = 1;
  for (int d = 0; d < 2; ++d)
  {
    ASSERT_THROW(mesh.n_cells[d] >= 1, "n_cells must be positive");
    _n[d] = mesh.n_cells[d];
    _N[d] = _n[d] + 1;
    _h[d] = mesh.cell_size[d];
    ASSERT_THROW(_h[d] > 0., "cell_size must be positive");
    nd *= _N[d];
    nc *= _n[d];
  }
  _n[2] = 0;
  _N[2] = 1;
  _h[2] = 1.;
  ASSERT_THROW(nd == mesh.n_dofs, "n_dofs does not match the cell grid (Q1: (n+1)^dim)");
  ASSERT_THROW(nd < (int64_t(1) << 30), "mesh too large");
  _n_dofs = nd;
  hipStream_t st = _handle.stream;
  _cd2.resize((size_t)nc * 4);
  _co2.resize((size_t)nc * 4);
  _cn2.resize((size_t)nd);
  const hipMemcpyKind kind = mesh.arrays_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  MFMG_HIP_CHECK(hipMemcpyAsync(_cd2.data(), mesh.cell_dofs, (size_t)nc * 4 * sizeof(int32_t), kind, st));
  MFMG_HIP_CHECK(hipMemcpyAsync(_cn2.data(), mesh.constrained, (size_t)nd, kind, st));
  {
    // the coefficient table in the operator's precision
    std::vector<double> co((size_t)nc * 4);
    MFMG_HIP_CHECK(hipMemcpyAsync(co.data(), mesh.coefficient, co.size() * sizeof(double),
                                  mesh.arrays_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, st));
    MFMG_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<T> cot(co.begin(), co.end());
    _co2.upload(cot.data(), cot.size(), st);
  }
  DeviceBuffer<int> bad(1);
  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf_range_kernel, dim3(n_blocks_for(nc * 4, 256, 1 << 16)), dim3(256), 0, st, _cd2.data(), nc * 4, nd, bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  ASSERT_THROW(bad.download(st)[0] == 0, "cell_dofs is not a logically structured quad mesh in lexicographic cell "
                                         "order (DoF ids out of range)");
  _node_dof2.resize((size_t)nd);
  hipLaunchKernelGGL(mf2d_node_dof_kernel, dim3(n_blocks_for(nd, 256, 1 << 16)), dim3(256), 0, st, _cd2.data(), _n[0], _n[1], _node_dof2.data());
  MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), st));
  hipLaunchKernelGGL(mf2d_validate_kernel, dim3(n_blocks_for(nc, 256, 1 << 16)), dim3(256), 0, st, _cd2.data(), _node_dof2.data(), _n[0], _n[1],
                     nd, bad.data());
  MFMG_HIP_CHECK(hipGetLastError());
  const int n_bad = bad.download(st)[0];
  ASSERT_THROW(n_bad == 0, "cell_dofs is not a logically structured quad mesh in lexicographic cell order (" + std::to_string(n_bad) +
                               " inconsistent corners)");
  // K[q][m][n] = sum_d JxW / h_d^2 dphi_m/dxi_d dphi_n/dxi_d at Gauss point q
  const double gp[2] = {MFMG_GB, MFMG_GA};
  const double w = _h[0] * _h[1] / 4.;
  for (int q = 0; q < 4; ++q)
  {
    double G[2][4];
    for (int m = 0; m < 4; ++m)
      for (int d = 0; d < 2; ++d)
      {
        double g = 1.;
        for (int e = 0; e < 2; ++e)
        {
          const int bit = (m >> e) & 1;
          const double xi = gp[(q >> e) & 1];
          g *= (e == d) ? (bit ? 1. : -1.) : (bit ? xi : (1. - xi));
        }
        G[d][m] = g;
      }
    for (int m = 0; m < 4; ++m)
      for (int n = 0; n < 4; ++n)
        _k2[(q * 4 + m) * 4 + n] = w / (_h[0] * _h[0]) * G[0][m] * G[0][n] + w / (_h[1] * _h[1]) * G[1][m] * G[1][n];
  }
  _diag.resize(nd);
  _dinv.resize(nd);
  Mf2dArgs<T> a{};
  a.cell_dofs = _cd2.data();
  a.coef = _co2.data();
  a.constrained = _cn2.data();
  a.node_dof = _node_dof2.data();
  a.nx = _n[0];
  a.ny = _n[1];
  Mf2dTable tab;
  for (int t = 0; t < 64; ++t)
    (&tab.K[0][0][0])[t] = _k2[t];
  hipLaunchKernelGGL((mf2d_diagonal_kernel<T>), dim3(n_blocks_for(nd, 256, 1 << 16)), dim3(256), 0, st, a, tab, _diag.data(), _dinv.data());
  MFMG_HIP_CHECK(hipGetLastError());
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
  _compact = false;
  _dinv_in_record = true;
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::launch_2d(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out) const
{
  Mf2dArgs<T> a{};
  a.cell_dofs = _cd2.data();
  a.coef = _co2.data();
  a.constrained = _cn2.data();
  a.node_dof = _node_dof2.data();
  a.nx = _n[0];
  a.ny = _n[1];
  a.x = x;
  a.b = b;
  a.dinv = _dinv.data();
  a.xprev = x_prev;
  a.out = out;
  a.alpha = alpha;
  a.beta = beta;
  a.mode = (int)mode;
  Mf2dTable tab;
  for (int t = 0; t < 64; ++t)
    (&tab.K[0][0][0])[t] = _k2[t];
  hipLaunchKernelGGL((mf2d_apply_kernel<T>), dim3(n_blocks_for(_n_dofs, 256, 1 << 16)), dim3(256), 0, _handle.stream, a, tab);
  MFMG_HIP_CHECK(hipGetLastError());
}

// ---- tile choice ---------------------------------------------------------------------------------
// The result does not depend on the tile (bit for bit), only the speed does.  With one memory round trip per
// layer pass the kernel runs at the rate the memory system sustains for its access mix (about 5 TB/s of HBM
// traffic on MI355X, profiles/README.md), and the tile only decides how many halo rows, columns and layers are
// read twice: measured at 257^3 and 512^3 DoFs all reasonable tiles lie within 3 % of each other.  The largest
// tile of the list that still gives every CU two rounds of wavefronts is taken (large tiles re-read less halo,
// but a launch of fewer than two rounds pays its ramp up and down in full).
template <typename T>
void MatrixFreeLaplaceDevice<T>::choose_tile(int &nw, int &ty, int &tz) const
{
  nw = _tile_waves;
  ty = _tile_y;
  tz = _tile_z;
  if (nw > 0 && ty > 0 && tz > 0)
    return;
  static const int pref_general[][3] = {{8, 2, 16}, {4, 3, 8}, {4, 2, 8}, {2, 2, 8}, {2, 2, 4}, {1, 2, 4}};
  static const int pref_compact[][3] = {{8, 3, 16}, {4, 3, 8}, {4, 3, 8}, {4, 2, 8}, {2, 2, 4}, {1, 2, 4}};
  const int(*pref)[3] = _compact ? pref_compact : pref_general;
  constexpr int n_pref = 6;
  static const int n_cus = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      v = 256;
    return v > 0 ? v : 256;
  }();
  const int64_t waves_wanted = (int64_t)2 * 16 * n_cus; // two rounds of 4 wavefronts per SIMD
  int pick = n_pref - 1;
  for (int c = 0; c < n_pref; ++c)
  {
    const int64_t wgs = (int64_t)_ncols * ((_N[1] + pref[c][0] * pref[c][1] - 2) / (pref[c][0] * pref[c][1] - 1)) *
                        ((_N[2] + pref[c][2] - 1) / pref[c][2]);
    // (the eight-wavefront tiles need four rounds: below that the shorter workgroups fill the chip better)
    if (wgs * pref[c][0] >= (pref[c][0] >= 8 ? 2 : 1) * waves_wanted)
    {
      pick = c;
      break;
    }
  }
  if (nw <= 0)
    nw = pref[pick][0];
  if (ty <= 0)
    ty = pref[pick][1];
  if (tz <= 0)
    tz = pref[pick][2];
  if (nw * ty < 2)
    ty = 2;
}

// Tiling along z.  A launch of a few rounds of workgroups pays for its last round: when the list of tiles runs out the
// chip drains for as long as the last tiles take (measured: 257^3 DoFs, three rounds of (4, 3, 8) tiles, 10 % of the launch
// against the same mesh four times as long in z).  So the tiles that are dispatched LAST are made SHORT: every XCD gets a
// contiguous run of layers (one eighth of the mesh, as before) cut into the same number of z-tiles whose heights fall
// off linearly -- 257 layers, tz = 8: 10, 8, 6, 5, 3 instead of 8, 8, 8, 8 -- at the price of one more halo layer per
// XCD.  Results do not depend on the tiling (bit for bit).  Ranges of z-tiles (the overlapped exchange of a distributed
// run) keep the uniform tiling: tile t owns the layers [t tz, (t + 1) tz).
template <typename T>
int const *MatrixFreeLaplaceDevice<T>::z_tiling(int tz, bool graded, int &n_tiles) const
{
  ZTiling &zt = graded ? _zt_graded : _zt_uniform;
  if (zt.tz != tz)
  {
    std::vector<int> tab;
    const int Nz = _N[2];
    static const int m_extra = std::getenv("MFMG_MF_GRADE_M") ? std::atoi(std::getenv("MFMG_MF_GRADE_M")) : 1;
    static const double w_off = std::getenv("MFMG_MF_GRADE_OFF") ? std::atof(std::getenv("MFMG_MF_GRADE_OFF")) : 0.5;
    const int m = (Nz / 8 + tz / 2) / tz + m_extra; // z-tiles per XCD
    if (graded && Nz / 8 >= 2 * m)
    {
      double total = 0.;
      for (int k = 0; k < m; ++k)
        total += m - k + w_off;
      for (int s = 0; s < 8; ++s)
      {
        const int l0 = (int)((int64_t)s * Nz / 8), l1 = (int)((int64_t)(s + 1) * Nz / 8);
        double cum = 0.;
        int prev = l0;
        for (int k = 0; k < m; ++k)
        {
          tab.push_back(prev);
          cum += m - k + w_off;
          int next = k + 1 == m ? l1 : l0 + (int)std::lround((l1 - l0) * cum / total);
          next = std::min(std::max(next, prev + 1), l1 - (m - 1 - k)); // every tile owns at least one layer
          prev = next;
        }
      }
      tab.push_back(Nz);
    }
    else
    {
      for (int l = 0; l < Nz; l += tz)
        tab.push_back(l);
      tab.push_back(Nz);
    }
    zt.n_tiles = (int)tab.size() - 1;
    zt.dev.upload(tab.data(), tab.size(), _handle.stream);
    MFMG_HIP_CHECK(hipStreamSynchronize(_handle.stream)); // (the host vector goes out of scope)
    zt.tz = tz;
  }
  n_tiles = zt.n_tiles;
  return zt.dev.data();
}

template <typename T>
bool MatrixFreeLaplaceDevice<T>::make_args(MfArgs<T> &a, unsigned int &n_blocks, MfMode mode, T const *x, T const *b,
                                           T const *x_prev, T alpha, T beta, T *out, int nw, int ty, int tz,
                                           int const *ztab, int z_tile_begin, int z_tile_end, int const *xy_range) const
{
  a.rec = _rec.data();
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
  {
    const double m00 = MFMG_GA * MFMG_GA + MFMG_GB * MFMG_GB, m01 = 2. * MFMG_GA * MFMG_GB;
    const double f[3] = {vol / 8. / (_h[0] * _h[0]), vol / 8. / (_h[1] * _h[1]), vol / 8. / (_h[2] * _h[2])};
    a.fax = T(2. * f[0] * m00);
    a.fbx = T(2. * f[0] * m01);
    a.fay = T(2. * f[1] * m00);
    a.fby = T(2. * f[1] * m01);
    a.faz = T(2. * f[2] * m00);
    a.fbz = T(2. * f[2] * m01);
    a.kd = T(2. * m00 * m00 * (f[0] + f[1] + f[2]));
  }
  a.alpha = alpha;
  a.beta = beta;
  a.mode = static_cast<int>(mode);
  a.rec_bytes = (unsigned int)_rec_bytes;
  a.own = _own;
  a.halo = _halo;
  a.dinv_in_record = _dinv_in_record ? 1 : 0;
  a.aff = _affine;
  a.ncols = _ncols;
  a.ncols_active = _tail ? _ncols - 1 : _ncols;
  // ty cell rows per wavefront, nw ty - 1 owned DoF rows per workgroup
  a.ntiles_y = (_N[1] + nw * ty - 2) / (nw * ty - 1);
  a.col0 = a.ty0 = 0;
  a.ztab = ztab;
  n_blocks = 0;
  if (xy_range)
  {
    // a sub-range of the column and y-tiles: {col begin, col end, y-tile begin, y-tile end}
    ASSERT_THROW(xy_range[0] >= 0 && xy_range[1] <= (int)a.ncols_active && xy_range[2] >= 0 && xy_range[3] <= (int)a.ntiles_y,
                 "tile range outside the tiling");
    if (xy_range[0] >= xy_range[1] || xy_range[2] >= xy_range[3])
      return false;
    a.col0 = (unsigned int)xy_range[0];
    a.ncols_active = (unsigned int)(xy_range[1] - xy_range[0]);
    a.ty0 = (unsigned int)xy_range[2];
    a.ntiles_y = (unsigned int)(xy_range[3] - xy_range[2]);
  }
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
                                     int nw, int ty, int tz, int z_tile_begin, int z_tile_end, int const *xy_range, bool with_main,
                                     bool with_tail, int const *exclude, hipStream_t on_stream) const
{
  ASSERT_THROW(nw >= 1 && nw <= 8, "1..8 wavefronts per workgroup");
  ASSERT_THROW(ty >= 1 && tz >= 1 && nw * ty >= 2, "operator tile too small");
  MfArgs<T> am, at;
  MfBoxes boxes;
  unsigned int main_blocks = 0, tail_blocks = 0;
  // the whole mesh: graded z-tiles; a range of z-tiles: the uniform tiling the caller counts in
  static const bool graded_env = !(std::getenv("MFMG_MF_GRADED_TILES") && std::string(std::getenv("MFMG_MF_GRADED_TILES")) == "0");
  const bool whole = z_tile_begin == 0 && z_tile_end < 0 && exclude == nullptr;
  int all_z = 0;
  int const *ztab = z_tiling(tz, whole && graded_env, all_z);
  if (z_tile_end < 0)
    z_tile_end = all_z;
  ASSERT_THROW(z_tile_begin >= 0 && z_tile_end <= all_z, "z-tile range outside the tiling");
  const bool have_main = make_args(am, main_blocks, mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, ztab, z_tile_begin, z_tile_end, xy_range);
  if (!have_main && !_tail)
    return;
  if (!with_main)
    main_blocks = 0;
  at = am;
  if (exclude)
  {
    // the tiles of the uniform tiling outside the box {lo[3], hi[3]}: z slabs over all columns and rows, y slabs between them,
    // x slabs between those (the tail columns are not column tiles: they are the `at` part of the launch as always)
    const int nt[3] = {(int)am.ncols_active, (int)am.ntiles_y, (int)am.ntiles_z};
    int lo[3], hi[3];
    for (int d = 0; d < 3; ++d)
    {
      lo[d] = std::min(std::max(exclude[d], 0), nt[d]);
      hi[d] = std::min(std::max(exclude[3 + d], lo[d]), nt[d]);
    }
    const int box[6][6] = {{0, nt[0], 0, nt[1], 0, lo[2]},         {0, nt[0], 0, nt[1], hi[2], nt[2]},
                           {0, nt[0], 0, lo[1], lo[2], hi[2]},     {0, nt[0], hi[1], nt[1], lo[2], hi[2]},
                           {0, lo[0], lo[1], hi[1], lo[2], hi[2]}, {hi[0], nt[0], lo[1], hi[1], lo[2], hi[2]}};
    unsigned int end = 0;
    for (int q = 0; q < 6; ++q)
    {
      const int ncq = box[q][1] - box[q][0], nyq = box[q][3] - box[q][2], nzq = box[q][5] - box[q][4];
      const bool empty = ncq <= 0 || nyq <= 0 || nzq <= 0;
      if (!empty)
        end += (unsigned int)ncq * nyq * nzq;
      boxes.bx_end[q] = end;
      boxes.bx_c0[q] = (unsigned int)box[q][0];
      boxes.bx_nc[q] = empty ? 1u : (unsigned int)ncq;
      boxes.bx_y0[q] = (unsigned int)box[q][2];
      boxes.bx_ny[q] = empty ? 1u : (unsigned int)nyq;
      boxes.bx_z0[q] = (unsigned int)box[q][4];
    }
    boxes.n_boxes = 6;
    main_blocks = with_main ? (end >= 64 ? ((end + 7) / 8) * 8 : end) : 0;
  }
  if (_tail && with_tail) // the columns of the last chunk: same tile shape, same layers (same table: Nz is the same), first in the grid
    _tail->make_args(at, tail_blocks, mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, ztab, z_tile_begin, z_tile_end, nullptr);
  if (main_blocks + tail_blocks == 0)
    return;
  const size_t lds = ((size_t)nw * (3 * ty + 1) + (size_t)2 * nw * 3) * 64 * sizeof(T) + (size_t)nw * (ty + 1) * 64 * sizeof(int);
  ASSERT_THROW(lds <= 160 * 1024, "operator tile too large for the LDS");
  const dim3 grid(main_blocks + tail_blocks);
  const dim3 block(64 * nw);
  hipStream_t st = on_stream ? on_stream : _handle.stream;
  auto go = [&](auto kernel) {
    // (the attribute is per kernel AND device; every instantiation decays to the same function-pointer type, so the
    // record of what has been set is keyed on the pointer -- a flag per lambda instantiation would be shared by all variants)
    static std::mutex attr_mutex;
    static std::set<std::pair<const void *, int>> attr_set;
    int dev = 0;
    MFMG_HIP_CHECK(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> lock(attr_mutex);
      if (attr_set.insert({reinterpret_cast<const void *>(kernel), dev}).second)
        MFMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024));
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, st, am, at, tail_blocks, boxes);
  };
  // both parts of the launch must read their ids the same way
  const bool affine = ids_computed();
  if (_compact)
  {
    if (ty == 2)
      go(mf_laplace_kernel<T, 2, true, 2>);
    else if (ty == 3)
      go(mf_laplace_kernel<T, 3, true, 3>);
    else if (ty == 4)
      go(mf_laplace_kernel<T, 4, true, 4>);
    else
      go(mf_laplace_kernel<T, 0, true, 1>);
  }
  else
  {
    // (experiments: MFMG_MF_BATCH=n asks for n rows per request batch where that instance exists)
    static const int batch_env = std::getenv("MFMG_MF_BATCH") ? std::atoi(std::getenv("MFMG_MF_BATCH")) : 0;
    if (ty == 2 && batch_env == 2)
      go(mf_laplace_kernel<T, 2, false, 2>);
    else if (ty == 3 && batch_env == 3)
      go(mf_laplace_kernel<T, 3, false, 3>);
    else if (ty == 4 && batch_env == 2)
      go(mf_laplace_kernel<T, 4, false, 2>);
    else if (affine && ty == 2)
      go(mf_laplace_kernel<T, 2, false, 1, true>);
    else if (affine && ty == 3)
      go(mf_laplace_kernel<T, 3, false, 1, true>);
    else if (affine && ty == 4)
      go(mf_laplace_kernel<T, 4, false, 1, true>);
    else if (affine)
      go(mf_laplace_kernel<T, 0, false, 1, true>);
    else if (ty == 2)
      go(mf_laplace_kernel<T, 2, false, 1>);
    else if (ty == 3)
      go(mf_laplace_kernel<T, 3, false, 1>);
    else if (ty == 4)
      go(mf_laplace_kernel<T, 4, false, 1>);
    else
      go(mf_laplace_kernel<T, 0, false, 1>);
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
  ASSERT_THROW(_dim == 3, "z-tile ranges belong to the 3-D operator");
  if (mode == MfMode::next && (x_prev == nullptr || alpha == T(0)))
    mode = MfMode::first;
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  const int all_z = (_N[2] + tz - 1) / tz;
  if (z_tile_begin >= z_tile_end)
    return;
  const double share = double(z_tile_end - z_tile_begin) / double(all_z);
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel", share * (required_bytes_apply() + epilogue_bytes((int)mode)),
                                           _handle.stream);
  run(mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, z_tile_begin, z_tile_end);
  KernelProfiler::end(stop, _handle.stream);
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::tiling(int n_tiles[3], int rows[3]) const
{
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  n_tiles[0] = _tail ? _ncols - 1 : _ncols;
  rows[0] = _own;
  rows[1] = nw * ty - 1;
  n_tiles[1] = (_N[1] + rows[1] - 1) / rows[1];
  rows[2] = tz;
  n_tiles[2] = (_N[2] + tz - 1) / tz;
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::launch_tiles(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out,
                                              int const begin[3], int const end[3], bool main_part, bool tail_part) const
{
  check_vectors(mode, x, b, x_prev, out);
  ASSERT_THROW(_dim == 3, "tile ranges belong to the 3-D operator");
  if (mode == MfMode::next && (x_prev == nullptr || alpha == T(0)))
    mode = MfMode::first;
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  int nt[3], rows[3];
  tiling(nt, rows);
  if (begin[2] >= end[2])
    return;
  const bool m = main_part && begin[0] < end[0] && begin[1] < end[1], t = tail_part && _tail != nullptr;
  if (!m && !t)
    return;
  // share of the DoFs this launch updates (for the profiler's bytes)
  const double zshare = double(end[2] - begin[2]) / double(nt[2]);
  const double tail_cols = _tail ? double(_N[0] - (_ncols - 1) * _own) : 0.;
  double share = 0.;
  if (m)
    share += zshare * (double(end[0] - begin[0]) * _own / double(_N[0])) * (double(end[1] - begin[1]) / double(nt[1]));
  if (t)
    share += zshare * tail_cols / double(_N[0]);
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel", std::min(1., share) * (required_bytes_apply() + epilogue_bytes((int)mode)),
                                           _handle.stream);
  const int xy[4] = {begin[0], end[0], begin[1], end[1]};
  run(mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, begin[2], end[2], xy, m, t);
  KernelProfiler::end(stop, _handle.stream);
}

// Everything launch_tiles(begin, end, main) leaves: the tiles of the uniform tiling OUTSIDE the box [begin, end) and the tail
// columns, as ONE launch (the workgroups of the excluded tiles leave at once).  Launched slab by slab -- up to six launches
// and the tail, each a fraction of a round of workgroups and as long as one workgroup lives -- the shell of a box rank cost
// more than the interior it surrounds.
template <typename T>
void MatrixFreeLaplaceDevice<T>::launch_outside(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out,
                                                int const begin[3], int const end[3], hipStream_t on_stream) const
{
  check_vectors(mode, x, b, x_prev, out);
  ASSERT_THROW(_dim == 3, "tile ranges belong to the 3-D operator");
  if (mode == MfMode::next && (x_prev == nullptr || alpha == T(0)))
    mode = MfMode::first;
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  int nt[3], rows[3];
  tiling(nt, rows);
  double inside = 1.;
  for (int d = 0; d < 3; ++d)
  {
    ASSERT_THROW(begin[d] >= 0 && begin[d] <= end[d] && end[d] <= nt[d], "tile range outside the tiling");
    inside *= double(end[d] - begin[d]) * (d == 0 ? double(rows[0]) / double(_N[0]) : 1. / double(nt[d]));
  }
  hipStream_t st = on_stream ? on_stream : _handle.stream;
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel", std::max(0., 1. - inside) * (required_bytes_apply() + epilogue_bytes((int)mode)), st);
  const int ex[6] = {begin[0], begin[1], begin[2], end[0], end[1], end[2]};
  run(mode, x, b, x_prev, alpha, beta, out, nw, ty, tz, 0, -1, nullptr, true, true, ex, st);
  KernelProfiler::end(stop, st);
}

template <typename T>
void MatrixFreeLaplaceDevice<T>::launch(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha,
                                        T beta, T *out) const
{
  check_vectors(mode, x, b, x_prev, out);
  if (_dim == 2)
  {
    if (mode == MfMode::next && (x_prev == nullptr || alpha == T(0)))
      mode = MfMode::first;
    launch_2d(mode, x, b, x_prev, alpha, beta, out);
    return;
  }
  int nw, ty, tz;
  choose_tile(nw, ty, tz);
  // bytes the layout requires per launch (mf_laplace.hpp), plus the b / D^-1 / x_prev reads of the epilogue
  hipEvent_t stop = _handle.profiler.begin("mf_laplace_kernel", required_bytes_apply() + epilogue_bytes((int)mode), _handle.stream);
  run(mode, x, b, x_prev, alpha, beta, out, nw, ty, tz);
  KernelProfiler::end(stop, _handle.stream);
}

template class MatrixFreeLaplaceDevice<double>;
template class MatrixFreeLaplaceDevice<float>;
} // namespace mfmg
