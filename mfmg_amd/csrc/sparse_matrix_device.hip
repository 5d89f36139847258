// gfx950 CSR SpMV with fused epilogues.  Replaces cusparseDcsrmv
// (include/mfmg/cuda/sparse_matrix_device.templates.cuh:53-70) for A, R, R^T and A_c,
// and folds the D^-1 "diagonal SpMV" + axpys of source/cuda/cuda_smoother.cu:48-59
// into the epilogue.
//
// A row is shared by LPR consecutive lanes of a wavefront (LPR = 1..64, chosen
// from the mean row length): consecutive lanes read consecutive (val, col) pairs,
// partial sums are combined with an xor-butterfly of __shfl_xor, so the summation
// order is fixed and the result is bit-reproducible.
#include "sparse_matrix_device.hpp"
#include "amge_structured.hpp"
#include "csr_algebra.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <limits>
#include <numeric>
#include <parallel/algorithm>
#include <type_traits>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace mfmg
{
namespace
{
template <typename T>
struct CsrArgs
{
  T const *val;
  int32_t const *col;
  int32_t const *row_ptr;
  int64_t n_rows;
  T const *x;
  T const *b;
  T const *dinv;
  T const *xprev;
  T *out;
  T alpha, beta;
  int mode;
  int pairs; // every vector the mode touches is 16-byte aligned: the two rows of a node are one access per operand (store_node)
  int32_t const *kept_ptr = nullptr; // after release_csr(): val / col hold the listed rows only, row q of the list at [kept_ptr[q], kept_ptr[q + 1])
};

// the fused epilogues of a row (the modes of CsrMode)
template <typename T>
__device__ __forceinline__ void store_row(CsrArgs<T> const &a, int64_t row, T sum)
{
  T o;
  switch (a.mode)
  {
  case 0:
    o = sum;
    break;
  case 1:
    o = sum - a.b[row];
    break;
  case 2:
    o = a.x[row] - a.beta * a.dinv[row] * (sum - a.b[row]);
    break;
  case 3:
  {
    const T xr = a.x[row];
    o = xr + a.alpha * (xr - a.xprev[row]) - a.beta * a.dinv[row] * (sum - a.b[row]);
    break;
  }
  case 4:
    o = a.out[row] - sum;
    break;
  case 6:
    o = sum + a.beta * a.dinv[row] * a.b[row];
    break;
  default:
    o = a.out[row] + sum;
    break;
  }
  a.out[row] = o;
}

// The epilogue of the C rows of one NODE (the node kernels below).  Row by row (store_row) the operands of the second row
// could only be requested after the store of the first -- nothing tells the compiler that out does not alias x, b or D^-1 --,
// one more dependent round trip in kernels that are bound by their latency.  Here every operand of the node is requested
// before the first store, and for C = 2 in FP64 the rows 2 n, 2 n + 1 are ONE 16-byte access per operand (a.pairs: the
// vectors are 16-byte aligned, checked at launch).  Same operations in the same order as store_row: same bits.
template <typename T, int C>
__device__ __forceinline__ void store_node(CsrArgs<T> const &a, int64_t node, T const (&sum)[C])
{
  const int64_t row0 = node * C;
  T xr[C], br[C], dr[C], pr[C], orr[C];
  const int mode = a.mode;
  const bool need_x = mode == 2 || mode == 3, need_b = (mode >= 1 && mode <= 3) || mode == 6, need_d = need_x || mode == 6,
             need_p = mode == 3, need_o = mode == 4 || mode == 5;
  if constexpr (C == 2 && sizeof(T) == 8)
  {
    if (a.pairs)
    {
      auto ld2 = [&](T const *v, T(&t)[C]) {
        const double2 q = *reinterpret_cast<double2 const *>(v + row0);
        t[0] = q.x;
        t[1] = q.y;
      };
      if (need_x)
        ld2(a.x, xr);
      if (need_b)
        ld2(a.b, br);
      if (need_d)
        ld2(a.dinv, dr);
      if (need_p)
        ld2(a.xprev, pr);
      if (need_o)
        ld2(a.out, orr);
      T o[C];
#pragma unroll
      for (int rc = 0; rc < C; ++rc)
      {
        switch (mode)
        {
        case 0:
          o[rc] = sum[rc];
          break;
        case 1:
          o[rc] = sum[rc] - br[rc];
          break;
        case 2:
          o[rc] = xr[rc] - a.beta * dr[rc] * (sum[rc] - br[rc]);
          break;
        case 3:
          o[rc] = xr[rc] + a.alpha * (xr[rc] - pr[rc]) - a.beta * dr[rc] * (sum[rc] - br[rc]);
          break;
        case 4:
          o[rc] = orr[rc] - sum[rc];
          break;
        case 6:
          o[rc] = sum[rc] + a.beta * dr[rc] * br[rc];
          break;
        default:
          o[rc] = orr[rc] + sum[rc];
          break;
        }
      }
      *reinterpret_cast<double2 *>(a.out + row0) = make_double2(o[0], o[1]);
      return;
    }
  }
#pragma unroll
  for (int rc = 0; rc < C; ++rc)
  {
    const int64_t row = row0 + rc;
    if (need_x)
      xr[rc] = a.x[row];
    if (need_b)
      br[rc] = a.b[row];
    if (need_d)
      dr[rc] = a.dinv[row];
    if (need_p)
      pr[rc] = a.xprev[row];
    if (need_o)
      orr[rc] = a.out[row];
  }
#pragma unroll
  for (int rc = 0; rc < C; ++rc)
  {
    T o;
    switch (mode)
    {
    case 0:
      o = sum[rc];
      break;
    case 1:
      o = sum[rc] - br[rc];
      break;
    case 2:
      o = xr[rc] - a.beta * dr[rc] * (sum[rc] - br[rc]);
      break;
    case 3:
      o = xr[rc] + a.alpha * (xr[rc] - pr[rc]) - a.beta * dr[rc] * (sum[rc] - br[rc]);
      break;
    case 4:
      o = orr[rc] - sum[rc];
      break;
    case 6:
      o = sum[rc] + a.beta * dr[rc] * br[rc];
      break;
    default:
      o = orr[rc] + sum[rc];
      break;
    }
    a.out[row0 + rc] = o;
  }
}

template <typename T, int LPR>
__global__ void csr_spmv_kernel(CsrArgs<T> a)
{
  const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t row = gtid / LPR;
  const int sub = threadIdx.x % LPR;
  T sum = T(0);
  if (row < a.n_rows)
  {
    const int s = a.row_ptr[row], e = a.row_ptr[row + 1];
    for (int p = s + sub; p < e; p += LPR)
      sum += a.val[p] * a.x[a.col[p]];
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1)
    sum += __shfl_xor(sum, off);
  if (row < a.n_rows && sub == 0)
  {
    T o;
    switch (a.mode)
    {
    case 0:
      o = sum;
      break;
    case 1:
      o = sum - a.b[row];
      break;
    case 2:
      o = a.x[row] - a.beta * a.dinv[row] * (sum - a.b[row]);
      break;
    case 3:
    {
      const T xr = a.x[row];
      o = xr + a.alpha * (xr - a.xprev[row]) - a.beta * a.dinv[row] * (sum - a.b[row]);
      break;
    }
    case 4:
      o = a.out[row] - sum;
      break;
    case 6:
      o = sum + a.beta * a.dinv[row] * a.b[row];
      break;
    default:
      o = a.out[row] + sum;
      break;
    }
    a.out[row] = o;
  }
}

// Few, long rows (the transfer operators and the triangular inverses at the bottom of the aggregation hierarchy:
// about a thousand rows of about a thousand entries): a workgroup of four wavefronts per row, so that the chip is not
// left with one wavefront per SIMD walking a long row; the four partial sums are added in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void csr_spmv_row_block_kernel(CsrArgs<T> a)
{
  __shared__ T part[4];
  const int64_t row = blockIdx.x;
  T sum = T(0);
  for (int p = a.row_ptr[row] + (int)threadIdx.x, e = a.row_ptr[row + 1]; p < e; p += 256)
    sum += a.val[p] * a.x[a.col[p]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    sum += __shfl_xor(sum, off);
  if ((threadIdx.x & 63) == 0)
    part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0)
    store_row(a, row, ((part[0] + part[1]) + part[2]) + part[3]);
}

// LDS-cached SpMV: the x entries a block of rows touches are gathered once (coalesced through the sorted
// list l2g) into LDS; rows then stream (val, 16-bit local column) pairs.  For the coarse operators of the
// AMGe hierarchy a 128-row block touches ~1200 distinct columns for ~7000 non-zeros, so x is fetched once
// instead of ~6 times through L1/L2, and the index stream halves.
template <typename T, int LPR>
__global__ void csr_spmv_lds_kernel(CsrArgs<T> a, int32_t const *blk_ptr, int32_t const *l2g, uint16_t const *lcol,
                                    int rows_per_block)
{
  extern __shared__ __align__(16) unsigned char csr_smem[];
  T *xs = reinterpret_cast<T *>(csr_smem);
  __shared__ int rp[256 + 1];
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row1 = min(row0 + rows_per_block, a.n_rows);
  const int l0 = blk_ptr[blockIdx.x], l1 = blk_ptr[blockIdx.x + 1];
  for (int i = threadIdx.x; i <= (int)(row1 - row0); i += blockDim.x)
    rp[i] = a.row_ptr[row0 + i];
  for (int i = threadIdx.x; i < l1 - l0; i += blockDim.x)
    xs[i] = a.x[l2g[l0 + i]];
  __syncthreads();
  const int group = threadIdx.x / LPR, sub = threadIdx.x % LPR, n_groups = blockDim.x / LPR;
  // every lane runs the same number of passes so that the shuffles are wave-uniform
  for (int64_t base = row0; base < row1; base += n_groups)
  {
    const int64_t row = base + group;
    T sum = T(0);
    if (row < row1)
    {
      const int s = rp[row - row0], e = rp[row - row0 + 1];
      for (int p = s + sub; p < e; p += LPR)
        sum += a.val[p] * xs[lcol[p]];
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1)
      sum += __shfl_xor(sum, off);
    if (row < row1 && sub == 0)
    {
      T o;
      switch (a.mode)
      {
      case 0:
        o = sum;
        break;
      case 1:
        o = sum - a.b[row];
        break;
      case 2:
        o = a.x[row] - a.beta * a.dinv[row] * (sum - a.b[row]);
        break;
      case 3:
      {
        const T xr = a.x[row];
        o = xr + a.alpha * (xr - a.xprev[row]) - a.beta * a.dinv[row] * (sum - a.b[row]);
        break;
      }
      case 4:
        o = a.out[row] - sum;
        break;
      case 6:
        o = sum + a.beta * a.dinv[row] * a.b[row];
        break;
      default:
        o = a.out[row] + sum;
        break;
      }
      a.out[row] = o;
    }
  }
}

template <typename T>
__global__ void csr_inv_diag_kernel(T const *val, int32_t const *col, int32_t const *row_ptr,
                                    int64_t n_rows, T *dinv)
{
  const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (row >= n_rows)
    return;
  T d = T(0);
  for (int p = row_ptr[row]; p < row_ptr[row + 1]; ++p)
    if (col[p] == row)
      d = val[p];
  dinv[row] = (d != T(0)) ? T(1) / d : T(0); // emptied (ghost) rows have no diagonal
}

// dinv[row] = 1 / a_rr (0 for a row without diagonal), ratio[row] = sum_j |a_rj| / |a_rr| (HUGE_VAL for a zero diagonal): what the setup of the aggregation hierarchy needs from a level -- in the order of the host loop
template <typename T>
__global__ void csr_row_ratio_kernel(T const *val, int32_t const *col, int32_t const *row_ptr, int64_t n_rows, T *dinv, T *ratio)
{
  const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (row >= n_rows)
    return;
  T d = T(0), sum = T(0);
  for (int p = row_ptr[row]; p < row_ptr[row + 1]; ++p)
  {
    sum += fabs(val[p]);
    if (col[p] == row)
      d = val[p];
  }
  dinv[row] = (d != T(0)) ? T(1) / d : T(0);
  ratio[row] = (d != T(0)) ? sum / fabs(d) : T(HUGE_VAL);
}

// Block-diagonal ("stencil") storage for square matrices whose rows come in nodes of C unknowns on a
// lexicographically numbered grid -- the coarse operators of the AMGe hierarchy on structured agglomerates:
// every C x C block sits on one of D block diagonals, node + offs[d].  Values are kept per diagonal and
// block column, val[(d C + cc) n_rows + r], so a wavefront reads 512 contiguous bytes per request and the
// matrix stream carries no column indices at all (8 B per stored entry instead of 10-12); x is read
// through L1/L2 (consecutive rows read consecutive entries).  One thread per row, fixed summation order.
// (V: storage type of the planes -- T, or float where every value is representable in it: half the bytes, sums in T)
template <typename T, int C, typename V = T>
__global__ __launch_bounds__(256) void bdia_spmv_kernel(CsrArgs<T> a, V const *val, int32_t const *offs, int D,
                                                        int32_t const *rows, int64_t n_listed)
{
  // rows != nullptr: only the listed rows (the others are regular, bdia_regular_node_kernel has them)
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (rows != nullptr ? n_listed : a.n_rows))
    return;
  const int64_t r = rows != nullptr ? (int64_t)rows[t] : t;
  const int64_t n_nodes = a.n_rows / C;
  const int64_t node = r / C;
  V const *vp = val + r;
  const size_t stride = (size_t)a.n_rows;
  T sum = T(0);
#pragma unroll 4
  for (int d = 0; d < D; ++d)
  {
    const int64_t nb = node + offs[d];
    if (nb >= 0 && nb < n_nodes)
    {
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        sum += T(vp[(size_t)(d * C + cc) * stride]) * a.x[nb * C + cc];
    }
  }
  const int64_t row = r;
  T o;
  switch (a.mode)
  {
  case 0:
    o = sum;
    break;
  case 1:
    o = sum - a.b[row];
    break;
  case 2:
    o = a.x[row] - a.beta * a.dinv[row] * (sum - a.b[row]);
    break;
  case 3:
  {
    const T xr = a.x[row];
    o = xr + a.alpha * (xr - a.xprev[row]) - a.beta * a.dinv[row] * (sum - a.b[row]);
    break;
  }
  case 4:
    o = a.out[row] - sum;
    break;
  case 6:
    o = sum + a.beta * a.dinv[row] * a.b[row];
    break;
  default:
    o = a.out[row] + sum;
    break;
  }
  a.out[row] = o;
}

// Row-base storage for rectangular stencil-like matrices whose values do not repeat (prolongators of a problem with a
// variable coefficient; with repeating values build_node_classes has them): every row couples to the same small box
// of unknowns, placed relative to a per-row base column: val[s n_rows + row] for the slots s of a shared offset
// list, column = base[row] + offs[s].  4 B of index per ROW instead of per entry, coalesced value planes, one
// thread per row, fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void rowbase_spmv_kernel(CsrArgs<T> a, T const *val, int32_t const *base,
                                                           int32_t const *offs, int S, int64_t n_cols)
{
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= a.n_rows)
    return;
  const int64_t b0 = base[r];
  T const *vp = val + r;
  const size_t stride = (size_t)a.n_rows;
  T sum = T(0);
#pragma unroll 4
  for (int sidx = 0; sidx < S; ++sidx)
  {
    const int64_t c = b0 + offs[sidx];
    if (c >= 0 && c < n_cols)
      sum += vp[(size_t)sidx * stride] * a.x[c];
  }
  store_row(a, r, sum);
}

// Symmetric matrices keep only the block diagonals with offset >= 0 (half the bytes from HBM): the entry
// A[(n,c)][(n-o,cc)] of a lower diagonal is read as its transpose A[(n-o,cc)][(n,c)] = val[o][c][(n-o) C + cc],
// i.e. the same plane a lower-numbered row streams as its upper part -- a second read of data that passed
// through the caches o rows earlier.  One thread per row, fixed summation order (upper part, then lower part).
// regular rows (see build_block_diagonals): the stencil comes from a table, only x is read
template <typename T>
struct BdiaRegular
{
  uint8_t const *exc; // nullptr: no regular rows
  int32_t const *exc_rows; // the rows that are NOT regular, ascending (what the row kernel works on)
  int64_t n_exc;
  T const *table;     // [C][Df][C]
  int32_t const *offs;
  int Df;
  int32_t const *base = nullptr; // rectangular matrices (node classes): first column node of a row node
  int64_t n_col_nodes = 0;
};

// Stencil tables and offset lists are read-only for the whole launch and addressed wave-uniformly: through the CONSTANT
// address space their loads are scalar loads whatever the compiler can prove about the stores of the kernel (with the plain
// pointer a change of the epilogue turned the 27 table loads of a node into per-lane 16-byte vector loads: 41 -> 60 us).
template <typename V>
using const_as = __attribute__((address_space(4))) const V;
template <typename V>
__device__ __forceinline__ const_as<V> *as_constant(V const *p)
{
  return reinterpret_cast<const_as<V> *>(reinterpret_cast<uintptr_t>(p));
}

// One thread per NODE whose C rows are all regular: the x values of a neighbour node are fetched once for the C
// rows (one 16-byte request for C = 2), the stencil constants are wave-uniform (scalar loads).
template <typename T, int C>
__global__ __launch_bounds__(256) void bdia_regular_node_kernel(CsrArgs<T> a, BdiaRegular<T> g)
{
  const int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (node * C >= a.n_rows || g.exc[node * C] != 0)
    return; // (the flag is set for every row of a node as soon as one of them is not regular)
  T sum[C];
#pragma unroll
  for (int rc = 0; rc < C; ++rc)
    sum[rc] = T(0);
  const_as<T> *table = as_constant(g.table);
  const_as<int32_t> *offs = as_constant(g.offs);
#pragma unroll 8
  for (int d = 0; d < g.Df; ++d)
  {
    const int64_t nb = node + offs[d];
    T xv[C];
    if constexpr (C == 2 && sizeof(T) == 8)
    {
      const double2 v = reinterpret_cast<double2 const *>(a.x)[nb];
      xv[0] = v.x;
      xv[1] = v.y;
    }
    else
    {
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        xv[cc] = a.x[nb * C + cc];
    }
#pragma unroll
    for (int rc = 0; rc < C; ++rc)
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        sum[rc] += table[((size_t)rc * g.Df + d) * C + cc] * xv[cc];
  }
  store_node<T, C>(a, node, sum);
}

// One wavefront over the CSR entries of one listed row (rows that are neither regular nor in a class): a thread per
// row would walk the diagonals one dependent load after the other.
template <typename T>
__device__ __forceinline__ void listed_row_wave(CsrArgs<T> const &a, int32_t const *rows, int64_t n_listed, int64_t w,
                                                int lane)
{
  if (w >= n_listed)
    return;
  const int64_t row = rows[w];
  T sum = T(0);
  const int p0 = a.kept_ptr ? a.kept_ptr[w] : a.row_ptr[row], e = a.kept_ptr ? a.kept_ptr[w + 1] : a.row_ptr[row + 1];
  for (int p = p0 + lane; p < e; p += 64)
    sum += a.val[p] * a.x[a.col[p]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    sum += __shfl_xor(sum, off);
  if (lane == 0)
    store_row(a, row, sum);
}

// Non-regular nodes that repeat one stencil among themselves (the nodes at the same distance from the faces of a
// box: boundary shells of a translation-invariant problem) are sorted by class, every class padded to whole
// wavefronts (node -1): the class of a wavefront is uniform, its stencil constants are scalar loads like the
// regular ones, and only the node list and x are read.  Neighbours outside the matrix carry the constant 0 and
// are read at a clamped position.
template <typename T, int C>
__global__ __launch_bounds__(256) void bdia_class_node_kernel(CsrArgs<T> a, BdiaRegular<T> g, int32_t const *nodes,
                                                              int32_t const *class_of_wave, T const *class_table,
                                                              int64_t n_slots, int32_t const *listed, int64_t n_listed)
{
  // the workgroups behind the class lists take the listed rows (a launch of their own would cost more than they do)
  const int64_t class_blocks = (n_slots + 255) / 256;
  if ((int64_t)blockIdx.x >= class_blocks)
  {
    listed_row_wave(a, listed, n_listed, ((int64_t)blockIdx.x - class_blocks) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
    return;
  }
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n_slots)
    return;
  const int cls = __builtin_amdgcn_readfirstlane(class_of_wave[t >> 6]);
  const int64_t node = nodes[t];
  if (node < 0)
    return;
  const int64_t last = g.n_col_nodes - 1;
  const int64_t first = g.base != nullptr ? (int64_t)g.base[node] : node;
  const_as<T> *tab = as_constant(class_table + (size_t)cls * (size_t)(C * g.Df * C));
  const_as<int32_t> *offs = as_constant(g.offs);
  T sum[C];
#pragma unroll
  for (int rc = 0; rc < C; ++rc)
    sum[rc] = T(0);
#pragma unroll 8
  for (int d = 0; d < g.Df; ++d)
  {
    // (node indices are int32 in every layout: the clamp is two 32-bit instructions instead of 64-bit selects)
    const int64_t nb = max(0, min((int)first + offs[d], (int)last));
    T xv[C];
    if constexpr (C == 2 && sizeof(T) == 8)
    {
      const double2 v = reinterpret_cast<double2 const *>(a.x)[nb];
      xv[0] = v.x;
      xv[1] = v.y;
    }
    else
    {
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        xv[cc] = a.x[nb * C + cc];
    }
#pragma unroll
    for (int rc = 0; rc < C; ++rc)
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        sum[rc] += tab[((size_t)rc * g.Df + d) * C + cc] * xv[cc];
  }
  store_node<T, C>(a, node, sum);
}

// The same two node kernels for wide stencils (the second level of the aggregation hierarchy couples 125 nodes):
// few nodes, long dependent gather chains -- the four wavefronts of a workgroup take a quarter of the stencil each
// for the same 64 nodes and wavefront 0 adds the parts in a fixed order.
template <typename T, int C, bool CLASSES, int P>
__global__ __launch_bounds__(1024) void bdia_node_split_kernel(CsrArgs<T> a, BdiaRegular<T> g, int32_t const *nodes,
                                                               int32_t const *class_of_wave, T const *class_table,
                                                               int64_t n_slots, int32_t const *listed, int64_t n_listed)
{
  if constexpr (CLASSES)
  {
    const int64_t class_blocks = (n_slots + 1024 / P - 1) / (1024 / P);
    if ((int64_t)blockIdx.x >= class_blocks) // (whole workgroups: nobody is left at the barrier below)
    {
      listed_row_wave(a, listed, n_listed, ((int64_t)blockIdx.x - class_blocks) * 16 + (threadIdx.x >> 6),
                      threadIdx.x & 63);
      return;
    }
  }
  // NB = 1024 / P consecutive nodes per workgroup (their stencils share cache lines), P parts of the stencil (4, or
  // 16 on the small levels where even that leaves most of the chip idle)
  constexpr int NB = 1024 / P;
  __shared__ T part[P - 1][C][NB];
  // (the part is the same for a whole wavefront: kept in a scalar register so that the stencil loads are scalar)
  const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NB)), lane = threadIdx.x % NB;
  const int64_t slot = (int64_t)blockIdx.x * NB + lane;
  int64_t node = -1;
  const_as<T> *tab = as_constant(g.table);
  const_as<int32_t> *offs = as_constant(g.offs);
  if constexpr (CLASSES)
  {
    // (n_slots is a multiple of 64: one class per wavefront, its table pointer stays in scalar registers)
    const int64_t ws = slot < n_slots ? slot : n_slots - 1;
    tab = as_constant(class_table + (size_t)__builtin_amdgcn_readfirstlane(class_of_wave[ws >> 6]) * (size_t)(C * g.Df * C));
    node = slot < n_slots ? (int64_t)nodes[ws] : -1;
  }
  else if (slot * C < a.n_rows && g.exc[slot * C] == 0)
    node = slot;
  const int64_t last = g.n_col_nodes - 1;
  const int64_t first = (CLASSES && g.base != nullptr && node >= 0) ? (int64_t)g.base[node] : node;
  const int d0 = (g.Df * q) / P, d1 = (g.Df * (q + 1)) / P;
  T sum[C];
#pragma unroll
  for (int rc = 0; rc < C; ++rc)
    sum[rc] = T(0);
  if (node >= 0)
  {
#pragma unroll 8
    for (int d = d0; d < d1; ++d)
    {
      int64_t nb = first + offs[d];
      if constexpr (CLASSES)
        nb = max(0, min((int)first + offs[d], (int)last)); // (int32 node indices: a 32-bit clamp)
      T xv[C];
      if constexpr (C == 2 && sizeof(T) == 8)
      {
        const double2 v = reinterpret_cast<double2 const *>(a.x)[nb];
        xv[0] = v.x;
        xv[1] = v.y;
      }
      else
      {
#pragma unroll
        for (int cc = 0; cc < C; ++cc)
          xv[cc] = a.x[nb * C + cc];
      }
#pragma unroll
      for (int rc = 0; rc < C; ++rc)
#pragma unroll
        for (int cc = 0; cc < C; ++cc)
          sum[rc] += tab[((size_t)rc * g.Df + d) * C + cc] * xv[cc];
    }
  }
  if (q > 0)
  {
#pragma unroll
    for (int rc = 0; rc < C; ++rc)
      part[q - 1][rc][lane] = sum[rc];
  }
  __syncthreads();
  if (q == 0 && node >= 0)
  {
    T total[C];
#pragma unroll
    for (int rc = 0; rc < C; ++rc)
    {
      total[rc] = sum[rc];
#pragma unroll
      for (int k = 0; k < P - 1; ++k)
        total[rc] += part[k][rc][lane];
    }
    store_node<T, C>(a, node, total);
  }
}

// the listed rows of a matrix without classes
template <typename T>
__global__ __launch_bounds__(256) void csr_listed_rows_kernel(CsrArgs<T> a, int32_t const *rows, int64_t n_listed)
{
  listed_row_wave(a, rows, n_listed, (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, threadIdx.x & 63);
}

// The same sums for a whole matrix with the block diagonals dealt to the four quarters of a 1024-thread workgroup
// (256 consecutive rows): four times the wavefronts, each with a quarter of the dependent loads; no branches
// around the loads (a neighbour outside the matrix is read at a clamped position and its entry replaced by 0), the
// quarters are added in a fixed order through LDS.
template <typename T, int C, typename V = T>
__global__ __launch_bounds__(1024) void bdia_sym_split_kernel(CsrArgs<T> a, V const *val, int32_t const *offs, int D)
{
  __shared__ T part[3][256];
  const int q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), lane = threadIdx.x & 255;
  const int64_t r = (int64_t)blockIdx.x * 256 + lane;
  const bool active = r < a.n_rows;
  const int64_t rr = active ? r : a.n_rows - 1;
  const int64_t last = a.n_rows / C - 1;
  const int64_t node = rr / C;
  const int c = (int)(rr - node * C);
  const size_t stride = (size_t)a.n_rows;
  V const *vp = val + rr;
  const int d0 = (D * q) / 4, d1 = (D * (q + 1)) / 4;
  T sum = T(0);
#pragma unroll 4
  for (int d = d0; d < d1; ++d) // offs[0] = 0 < offs[1] < ...; the entry of a neighbour behind the last node is 0
  {
    const int64_t nb = min(node + offs[d], last);
#pragma unroll
    for (int cc = 0; cc < C; ++cc)
      sum += T(vp[(size_t)(d * C + cc) * stride]) * a.x[nb * C + cc];
  }
#pragma unroll 4
  for (int d = max(d0, 1); d < d1; ++d)
  {
    const int64_t nbu = node - offs[d];
    const int64_t nb = max(nbu, (int64_t)0);
    V const *vq = val + (size_t)(d * C + c) * stride + nb * C;
#pragma unroll
    for (int cc = 0; cc < C; ++cc)
      sum += (nbu >= 0 ? T(vq[cc]) : T(0)) * a.x[nb * C + cc];
  }
  if (q > 0)
    part[q - 1][lane] = sum;
  __syncthreads();
  if (q == 0 && active)
    store_row(a, r, ((sum + part[0][lane]) + part[1][lane]) + part[2][lane]);
}

template <typename T, int C, typename V = T>
__global__ __launch_bounds__(256) void bdia_sym_spmv_kernel(CsrArgs<T> a, V const *val, int32_t const *offs, int D,
                                                            BdiaRegular<T> g)
{
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // with regular rows in the matrix (bdia_regular_node_kernel has them) only the listed rows are left
  if (t >= (g.exc != nullptr ? g.n_exc : a.n_rows))
    return;
  const int64_t r = g.exc != nullptr ? (int64_t)g.exc_rows[t] : t;
  const int64_t n_nodes = a.n_rows / C;
  const int64_t node = r / C;
  const int c = (int)(r - node * C);
  const size_t stride = (size_t)a.n_rows;
  V const *vp = val + r;
  T sum = T(0);
#pragma unroll 4
  for (int d = 0; d < D; ++d) // offs[0] = 0 < offs[1] < ...
  {
    const int64_t nb = node + offs[d];
    if (nb < n_nodes)
    {
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        sum += T(vp[(size_t)(d * C + cc) * stride]) * a.x[nb * C + cc];
    }
  }
#pragma unroll 4
  for (int d = 1; d < D; ++d)
  {
    const int64_t nb = node - offs[d];
    if (nb >= 0)
    {
      V const *vq = val + (size_t)(d * C + c) * stride + nb * C;
#pragma unroll
      for (int cc = 0; cc < C; ++cc)
        sum += T(vq[cc]) * a.x[nb * C + cc];
    }
  }
  const int64_t row = r;
  T o;
  switch (a.mode)
  {
  case 0:
    o = sum;
    break;
  case 1:
    o = sum - a.b[row];
    break;
  case 2:
    o = a.x[row] - a.beta * a.dinv[row] * (sum - a.b[row]);
    break;
  case 3:
  {
    const T xr = a.x[row];
    o = xr + a.alpha * (xr - a.xprev[row]) - a.beta * a.dinv[row] * (sum - a.b[row]);
    break;
  }
  case 4:
    o = a.out[row] - sum;
    break;
  case 6:
    o = sum + a.beta * a.dinv[row] * a.b[row];
    break;
  default:
    o = a.out[row] + sum;
    break;
  }
  a.out[row] = o;
}

// ---- layout analysis on the device -----------------------------------------------------------------------
// The analysis of an uploaded matrix (block-diagonal planes, regular rows, stencil classes, symmetry; node classes
// of the rectangular operators) walks planes of 1-15 GB several times; on the host cores these passes were a third
// of the setup of the hierarchy (2.7 s of 8.4 s at 257^3 DoFs).  They run as kernels on the CSR arrays that are on
// the device anyway; what comes back to the host is small: a flag per row, a hash per exceptional node, the tables.
// raise a flag once: the atomic is skipped when the flag is visibly up already (a failing analysis would otherwise issue
// one atomic per entry on the same address)
__device__ __forceinline__ void raise_flag(int *flag, int bits)
{
  if ((*reinterpret_cast<int const volatile *>(flag) & bits) != bits)
    atomicOr(flag, bits);
}

__device__ __forceinline__ int find_offset(int32_t const *offs, int D, int32_t o)
{
  int lo = 0, hi = D;
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (offs[mid] < o)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (lo < D && offs[lo] == o) ? lo : -1;
}

// planes dv[(d c + cc) n + r] of the block diagonals; *bad is raised by an entry on no listed diagonal
template <typename T>
__global__ void bdia_fill_kernel(int64_t n, int c, int D, int32_t const *offs, int32_t const *row_ptr, int32_t const *col,
                                 T const *val, T *dv, int *bad)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
    for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p)
    {
      const int d = find_offset(offs, D, (int32_t)(col[p] / c - r / c));
      if (d < 0)
      {
        raise_flag(bad, 1);
        continue;
      }
      dv[((size_t)d * c + (size_t)(col[p] % c)) * n + r] += val[p];
    }
}

// hits[q] = number of sampled nodes whose first row repeats the first row of candidate q
template <typename T>
__global__ void bdia_hits_kernel(int64_t n, int c, int D, T const *dv, int64_t n_nodes, int64_t const *cand, int n_cand, int n_sample,
                                 int *hits)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cand * n_sample)
    return;
  const int q = t / n_sample;
  const int64_t node = (int64_t)(((long long)(t % n_sample) * 2654435761ll) % n_nodes);
  const int64_t ref = cand[q];
  bool same = true;
  for (int d = 0; d < D && same; ++d)
    for (int cc = 0; cc < c; ++cc)
      if (dv[((size_t)d * c + cc) * n + node * c] != dv[((size_t)d * c + cc) * n + ref * c])
        same = false;
  if (same)
    atomicAdd(hits + q, 1);
}

// out[q][k] = the stencil tuple (k = (rc D + d) c + cc) of node nodes[q]
template <typename T>
__global__ void bdia_tuple_kernel(int64_t n, int c, int D, T const *dv, int64_t const *nodes, int64_t n_q, T *out)
{
  const size_t tuple = (size_t)c * D * c;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < (int64_t)(n_q * tuple); t += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t q = t / (int64_t)tuple;
    const size_t k = (size_t)(t % (int64_t)tuple);
    const int cc = (int)(k % c), d = (int)((k / c) % D), rc = (int)(k / ((size_t)c * D));
    out[t] = dv[((size_t)d * c + cc) * n + nodes[q] * c + rc];
  }
}

// exc[r] = 0 where row r repeats the reference stencil and has it wholly inside the matrix
template <typename T>
__global__ void bdia_exc_kernel(int64_t n, int c, int D, int32_t const *offs, T const *dv, T const *table, uint8_t *exc)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
  {
    const int rc = (int)(r % c);
    const int64_t node = r / c;
    bool same = true;
    for (int d = 0; d < D && same; ++d)
    {
      const int64_t nb = node + offs[d];
      if (nb < 0 || nb >= n / c)
        same = false;
      for (int cc = 0; cc < c && same; ++cc)
        same = dv[((size_t)d * c + cc) * n + r] == table[((size_t)rc * D + d) * c + cc];
    }
    exc[r] = same ? 0 : 1;
  }
}

// a node is regular only if all its rows are
__global__ void bdia_exc_node_kernel(int64_t n_nodes, int c, uint8_t *exc)
{
  for (int64_t nd = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; nd < n_nodes; nd += (int64_t)gridDim.x * blockDim.x)
  {
    uint8_t any = 0;
    for (int rc = 0; rc < c; ++rc)
      any |= exc[nd * c + rc];
    for (int rc = 0; rc < c; ++rc)
      exc[nd * c + rc] = any;
  }
}

__device__ __forceinline__ uint64_t hash_step(uint64_t h, double v)
{
  h = (h ^ (uint64_t)__double_as_longlong(v)) * 1099511628211ull;
  return h ^ (h >> 29);
}

// hash of the stencil tuple of every listed node (the order of the host loop it replaces: k = (rc D + d) c + cc)
template <typename T>
__global__ void bdia_hash_kernel(int64_t n, int c, int D, T const *dv, int64_t const *nodes, int64_t n_q, uint64_t *hash)
{
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n_q; q += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t nd = nodes[q];
    uint64_t h = 1469598103934665603ull;
    for (int rc = 0; rc < c; ++rc)
      for (int d = 0; d < D; ++d)
        for (int cc = 0; cc < c; ++cc)
          h = hash_step(h, (double)dv[((size_t)d * c + cc) * n + nd * c + rc]);
    hash[q] = h;
  }
}

// same[q] = 1 where the tuple of nodes[q] equals the tuple of rep[q] bit for bit (rep[q] < 0: not asked)
template <typename T>
__global__ void bdia_same_kernel(int64_t n, int c, int D, T const *dv, int64_t const *nodes, int64_t const *rep, int64_t n_q,
                                 uint8_t *same)
{
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n_q; q += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t nd = nodes[q], rp = rep[q];
    bool eq = rp >= 0;
    for (int rc = 0; rc < c && eq; ++rc)
      for (int d = 0; d < D && eq; ++d)
        for (int cc = 0; cc < c; ++cc)
          if (dv[((size_t)d * c + cc) * n + nd * c + rc] != dv[((size_t)d * c + cc) * n + rp * c + rc])
            eq = false;
    same[q] = eq ? 1 : 0;
  }
}

// *asym is raised by an entry that differs from its transposed partner by more than tol
template <typename T>
__global__ void bdia_symmetry_kernel(int64_t n, int c, int D, int32_t const *offs, T const *dv, double tol, int *asym)
{
  const int zero = D / 2;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t node = r / c;
    const int rc = (int)(r % c);
    for (int d = zero; d < D; ++d)
    {
      const int64_t nb = node + offs[d];
      if (nb >= n / c)
        continue;
      for (int cc = 0; cc < c; ++cc)
      {
        const double up = (double)dv[((size_t)d * c + cc) * n + r];
        const double lo = (double)dv[((size_t)(2 * zero - d) * c + rc) * n + (nb * c + cc)];
        if (fabs(up - lo) > tol)
          raise_flag(asym, 1);
      }
    }
  }
}

template <typename T>
__global__ void gather_strided_kernel(T const *src, int64_t const *index, int64_t n_q, T *out)
{
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n_q; q += (int64_t)gridDim.x * blockDim.x)
    out[q] = src[index[q]];
}

// node classes of a rectangular matrix: base[nd] = first column node of row node nd, tv[nd][(rc D + d) c + cc] its tuple
template <typename T>
__global__ void nodecls_fill_kernel(int64_t n_nodes, int c, int D, int32_t const *offs, int32_t const *row_ptr, int32_t const *col,
                                    T const *val, int32_t *base, T *tv, int *bad)
{
  const size_t tuple = (size_t)c * D * c;
  for (int64_t nd = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; nd < n_nodes; nd += (int64_t)gridDim.x * blockDim.x)
  {
    int64_t b = INT64_MAX;
    for (int64_t r = nd * c; r < (nd + 1) * c; ++r)
      for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p)
        b = min(b, (int64_t)(col[p] / c));
    if (b == INT64_MAX)
    {
      base[nd] = 0; // a node without entries: all zeros
      continue;
    }
    base[nd] = (int32_t)b;
    for (int rc = 0; rc < c; ++rc)
      for (int p = row_ptr[nd * c + rc]; p < row_ptr[nd * c + rc + 1]; ++p)
      {
        const int d = find_offset(offs, D, (int32_t)(col[p] / c - b));
        if (d < 0)
        {
          raise_flag(bad, 1);
          continue;
        }
        tv[(size_t)nd * tuple + ((size_t)rc * D + d) * c + (size_t)(col[p] % c)] += val[p];
      }
  }
}

template <typename T>
__global__ void nodecls_hash_kernel(int64_t n_nodes, size_t tuple, T const *tv, uint64_t *hash)
{
  for (int64_t nd = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; nd < n_nodes; nd += (int64_t)gridDim.x * blockDim.x)
  {
    uint64_t h = 1469598103934665603ull;
    for (size_t k = 0; k < tuple; ++k)
      h = hash_step(h, (double)tv[(size_t)nd * tuple + k]);
    hash[nd] = h;
  }
}

template <typename T>
__global__ void nodecls_same_kernel(int64_t n_nodes, size_t tuple, T const *tv, int64_t const *rep, uint8_t *same)
{
  for (int64_t nd = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; nd < n_nodes; nd += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t rp = rep[nd];
    bool eq = rp >= 0;
    for (size_t k = 0; k < tuple && eq; ++k)
      eq = tv[(size_t)nd * tuple + k] == tv[(size_t)rp * tuple + k];
    same[nd] = eq ? 1 : 0;
  }
}

// *flag is raised by a value that a float does not hold exactly
template <typename T>
__global__ void not_float_kernel(T const *v, int64_t n, int *flag)
{
  // (one atomic per wavefront at most, none once the flag is up: with FP64 values nearly every entry raises it, and 10^8
  // atomics on one address took 20 ms)
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n && !bad; i += (int64_t)gridDim.x * blockDim.x)
  {
    if (*reinterpret_cast<int const volatile *>(flag) != 0)
      return;
    bad = (T)(float)v[i] != v[i];
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == __builtin_ctzll(__ballot(bad)))
    atomicOr(flag, 1);
}
template <typename T>
__global__ void narrow_kernel(T const *v, int64_t n, float *out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (float)v[i];
}

constexpr size_t kMaxBlockDiagonals = 400; // (343: the third level of the aggregation hierarchy of a Q1 problem)
constexpr int kMaxStoredBlockDiagonals = 160;
constexpr int64_t kRegularAsClassNodes = 5000000; // nodes up to which regular nodes are evaluated as one more class (one launch instead of two: at 2.1 M nodes 54 against 38 + 23 us)
constexpr int kSplitStencil = 48;        // block diagonals from which a node's stencil is split over four wavefronts
constexpr int64_t kListedWaveRows = 32768; // listed rows up to which each gets a wavefront of its own

template <typename T, int LPR>
void launch_lds(CsrArgs<T> const &a, hipStream_t st, int32_t const *blk_ptr, int32_t const *l2g, uint16_t const *lcol,
                int rows_per_block, int max_cols)
{
  const int64_t nb = (a.n_rows + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((csr_spmv_lds_kernel<T, LPR>), dim3((unsigned int)nb), dim3(1024),
                     (size_t)max_cols * sizeof(T), st, a, blk_ptr, l2g, lcol, rows_per_block);
}

template <typename T, int LPR>
void launch_lpr(CsrArgs<T> const &a, hipStream_t st)
{
  const int64_t threads = a.n_rows * LPR;
  const int64_t nb = (threads + block_size - 1) / block_size;
  hipLaunchKernelGGL((csr_spmv_kernel<T, LPR>), dim3((unsigned int)std::max<int64_t>(nb, 1)),
                     dim3(block_size), 0, st, a);
}
} // namespace

template <typename T>
SparseMatrixDevice<T>::SparseMatrixDevice(HipHandle &handle, int64_t n_rows, int64_t n_cols,
                                          std::vector<int32_t> row_ptr, std::vector<int32_t> col,
                                          std::vector<T> val, bool keep_host, bool analyse)
    : _handle(handle), _n_rows(n_rows), _n_cols(n_cols)
{
  ASSERT_THROW((int64_t)row_ptr.size() == n_rows + 1, "row_ptr has the wrong size");
  _nnz = row_ptr.empty() ? 0 : row_ptr.back();
  ASSERT_THROW((int64_t)col.size() == _nnz && (int64_t)val.size() == _nnz,
               "column index / value arrays do not match row_ptr");
  ASSERT_THROW(n_rows * 64 < (int64_t(1) << 40), "matrix too large");
  {
    bool rows_ok = true, cols_ok = true;
#pragma omp parallel for schedule(static) reduction(&& : rows_ok)
    for (int64_t r = 0; r < n_rows; ++r)
      rows_ok = rows_ok && row_ptr[r] <= row_ptr[r + 1];
    ASSERT_THROW(rows_ok, "row_ptr must be non-decreasing");
#pragma omp parallel for schedule(static) reduction(&& : cols_ok)
    for (int64_t p = 0; p < _nnz; ++p)
      cols_ok = cols_ok && col[p] >= 0 && col[p] < n_cols;
    ASSERT_THROW(cols_ok, "column index out of range");
  }
  // (the CSR arrays go to the device first: the layout analysis runs on them.  Without an analysis and with the host arrays
  // kept -- the restrictor beside its agglomerate-wise form, 11 GB at 513^3 DoFs -- they go there when first asked for.)
  _device_csr_deferred = keep_host && !analyse && _nnz > 0;
  if (!_device_csr_deferred)
  {
    MemoryKind kind("CSR arrays (val, col, row_ptr)");
    _val.upload(val.data(), val.size(), handle.stream);
    _col.upload(col.data(), col.size(), handle.stream);
    _row_ptr.upload(row_ptr.data(), row_ptr.size(), handle.stream);
  }
  _row_ptr_host = std::move(row_ptr);
  _col_host = std::move(col);
  _val_host = std::move(val);
  choose_layouts(analyse);
  if (!keep_host)
  {
    std::vector<int32_t>().swap(_row_ptr_host);
    std::vector<int32_t>().swap(_col_host);
    std::vector<T>().swap(_val_host);
  }
}

namespace
{
__global__ void csr_validate_kernel(int64_t n_rows, int64_t n_cols, int32_t const *row_ptr, int32_t const *col, int *bad)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
  {
    if (row_ptr[r] > row_ptr[r + 1] || row_ptr[r] < 0)
    {
      raise_flag(bad, 1);
      continue;
    }
    for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p)
      if (col[p] < 0 || col[p] >= n_cols)
        raise_flag(bad, 2);
  }
}

// lengths of a list of rows, and their column indices packed one row after the other
__global__ void csr_row_lengths_kernel(int32_t const *row_ptr, int64_t const *rows, int64_t n_q, int32_t *len)
{
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n_q; q += (int64_t)gridDim.x * blockDim.x)
    len[q] = row_ptr[rows[q] + 1] - row_ptr[rows[q]];
}
__global__ void csr_pack_rows_kernel(int32_t const *row_ptr, int32_t const *col, int64_t const *rows, int64_t n_q, int32_t const *out_ptr,
                                     int32_t *out_col)
{
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n_q; q += (int64_t)gridDim.x * blockDim.x)
  {
    const int p0 = row_ptr[rows[q]], len = row_ptr[rows[q] + 1] - p0;
    for (int t = 0; t < len; ++t)
      out_col[out_ptr[q] + t] = col[p0 + t];
  }
}
} // namespace

// From arrays that are on the device already (the setup forms its matrices there): ownership is taken, the host copy
// is fetched only if something asks for it (download, the host fall-backs of transpose / mmult, the row-base and
// LDS-cached layouts).
template <typename T>
SparseMatrixDevice<T>::SparseMatrixDevice(HipHandle &handle, int64_t n_rows, int64_t n_cols, DeviceBuffer<int32_t> row_ptr,
                                          DeviceBuffer<int32_t> col, DeviceBuffer<T> val, bool analyse)
    : _handle(handle), _n_rows(n_rows), _n_cols(n_cols)
{
  ASSERT_THROW((int64_t)row_ptr.size() == n_rows + 1, "row_ptr has the wrong size");
  ASSERT_THROW(n_rows * 64 < (int64_t(1) << 40), "matrix too large");
  int32_t last = 0;
  MFMG_HIP_CHECK(hipMemcpyAsync(&last, row_ptr.data() + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, handle.stream));
  MFMG_HIP_CHECK(hipStreamSynchronize(handle.stream));
  _nnz = last;
  ASSERT_THROW((int64_t)col.size() == _nnz && (int64_t)val.size() == _nnz, "column index / value arrays do not match row_ptr");
  _row_ptr = std::move(row_ptr);
  _col = std::move(col);
  _val = std::move(val);
  {
    DeviceBuffer<int> bad(1);
    MFMG_HIP_CHECK(hipMemsetAsync(bad.data(), 0, sizeof(int), handle.stream));
    hipLaunchKernelGGL(csr_validate_kernel, dim3(n_blocks_for(n_rows, 256, 1 << 16)), dim3(256), 0, handle.stream, n_rows, n_cols,
                       _row_ptr.data(), _col.data(), bad.data());
    MFMG_HIP_CHECK(hipGetLastError());
    const int b = bad.download(handle.stream)[0];
    ASSERT_THROW((b & 1) == 0, "row_ptr must be non-decreasing");
    ASSERT_THROW((b & 2) == 0, "column index out of range");
  }
  choose_layouts(analyse);
}

template <typename T>
void SparseMatrixDevice<T>::ensure_host_copy() const
{
  if (!_row_ptr_host.empty() || _n_rows == 0)
    return;
  download(_row_ptr_host, _col_host, _val_host);
}

template <typename T>
void SparseMatrixDevice<T>::ensure_device_csr() const
{
  ASSERT_THROW(!_csr_released, "the CSR arrays of this matrix were released after the setup (\"release setup matrices\")");
  if (!_device_csr_deferred)
    return;
  MemoryKind kind("CSR arrays (val, col, row_ptr)");
  _val.upload(_val_host.data(), _val_host.size(), _handle.stream);
  _col.upload(_col_host.data(), _col_host.size(), _handle.stream);
  _row_ptr.upload(_row_ptr_host.data(), _row_ptr_host.size(), _handle.stream);
  _device_csr_deferred = false;
}

namespace
{
// rows `rows[q]` of a CSR matrix copied one after the other (a wavefront per row)
template <typename T>
__global__ void csr_keep_rows_kernel(int32_t const *row_ptr, int32_t const *col, T const *val, int32_t const *rows, int64_t n_q, int32_t const *kept_ptr,
                                     int32_t *kept_col, T *kept_val)
{
  const int64_t q = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (q >= n_q)
    return;
  const int s = row_ptr[rows[q]], e = row_ptr[rows[q] + 1], o = kept_ptr[q];
  for (int p = s + lane; p < e; p += 64)
  {
    kept_col[o + p - s] = col[p];
    kept_val[o + p - s] = val[p];
  }
}
__global__ void csr_listed_lengths_kernel(int32_t const *row_ptr, int32_t const *rows, int64_t n_q, int32_t *len)
{
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n_q; q += (int64_t)gridDim.x * blockDim.x)
    len[q] = row_ptr[rows[q] + 1] - row_ptr[rows[q]];
}
} // namespace

template <typename T>
bool SparseMatrixDevice<T>::release_csr()
{
  if (_csr_released)
    return true;
  const bool by_classes = _use_nodecls && _use_regular;
  const bool by_diagonals = _use_bdia && _bdia_regular && _use_regular && (int64_t)_bdia_exc_rows.size() <= kListedWaveRows;
  if (!by_classes && !by_diagonals)
    return false;
  ensure_device_csr();
  hipStream_t st = _handle.stream;
  DeviceBuffer<int32_t> const &rows = by_classes ? _nc_listed : _bdia_exc_rows;
  const int64_t n_q = (int64_t)rows.size();
  std::vector<int32_t> ptr((size_t)n_q + 1, 0);
  if (n_q > 0)
  {
    DeviceBuffer<int32_t> len((size_t)n_q);
    hipLaunchKernelGGL(csr_listed_lengths_kernel, dim3(n_blocks_for(n_q, 256, 1 << 16)), dim3(256), 0, st, _row_ptr.data(), rows.data(), n_q, len.data());
    MFMG_HIP_CHECK(hipGetLastError());
    const std::vector<int32_t> hl = len.download(st);
    for (int64_t q = 0; q < n_q; ++q)
      ptr[q + 1] = ptr[q] + hl[q];
  }
  MemoryKind kind("CSR arrays (val, col, row_ptr)");
  _kept_ptr.upload(ptr.data(), ptr.size(), st);
  _kept_col.resize((size_t)std::max<int32_t>(ptr[n_q], 1));
  _kept_val.resize((size_t)std::max<int32_t>(ptr[n_q], 1));
  if (n_q > 0)
  {
    hipLaunchKernelGGL(csr_keep_rows_kernel<T>, dim3((unsigned int)((n_q * 64 + 255) / 256)), dim3(256), 0, st, _row_ptr.data(), _col.data(), _val.data(),
                       rows.data(), n_q, _kept_ptr.data(), _kept_col.data(), _kept_val.data());
    MFMG_HIP_CHECK(hipGetLastError());
  }
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
  _val.release();
  _col.release();
  _row_ptr.release();
  std::vector<int32_t>().swap(_row_ptr_host);
  std::vector<int32_t>().swap(_col_host);
  std::vector<T>().swap(_val_host);
  _csr_released = true;
  return true;
}

// column indices of the rows `rows` (ascending), packed: row q occupies [ptr[q], ptr[q+1]) of cols
template <typename T>
void SparseMatrixDevice<T>::sample_rows(std::vector<int64_t> const &rows, std::vector<int32_t> &ptr, std::vector<int32_t> &cols) const
{
  const int64_t nq = (int64_t)rows.size();
  ptr.assign(nq + 1, 0);
  cols.clear();
  if (nq == 0)
    return;
  ASSERT_THROW(!_csr_released, "the CSR arrays of this matrix were released after the setup (\"release setup matrices\")");
  if (has_host_copy())
  {
    for (int64_t q = 0; q < nq; ++q)
      ptr[q + 1] = ptr[q] + (_row_ptr_host[rows[q] + 1] - _row_ptr_host[rows[q]]);
    cols.resize(ptr[nq]);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nq; ++q)
      std::copy(_col_host.begin() + _row_ptr_host[rows[q]], _col_host.begin() + _row_ptr_host[rows[q] + 1], cols.begin() + ptr[q]);
    return;
  }
  hipStream_t st = _handle.stream;
  DeviceBuffer<int64_t> d_rows;
  d_rows.upload(rows.data(), rows.size(), st);
  DeviceBuffer<int32_t> d_len((size_t)nq);
  hipLaunchKernelGGL(csr_row_lengths_kernel, dim3(n_blocks_for(nq, 256, 1 << 16)), dim3(256), 0, st, _row_ptr.data(), d_rows.data(), nq,
                     d_len.data());
  MFMG_HIP_CHECK(hipGetLastError());
  const std::vector<int32_t> len = d_len.download(st);
  for (int64_t q = 0; q < nq; ++q)
    ptr[q + 1] = ptr[q] + len[q];
  cols.resize(ptr[nq]);
  if (ptr[nq] == 0)
    return;
  DeviceBuffer<int32_t> d_ptr, d_cols((size_t)ptr[nq]);
  d_ptr.upload(ptr.data(), ptr.size(), st);
  hipLaunchKernelGGL(csr_pack_rows_kernel, dim3(n_blocks_for(nq, 256, 1 << 16)), dim3(256), 0, st, _row_ptr.data(), _col.data(), d_rows.data(),
                     nq, d_ptr.data(), d_cols.data());
  MFMG_HIP_CHECK(hipGetLastError());
  MFMG_HIP_CHECK(hipMemcpyAsync(cols.data(), d_cols.data(), cols.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
}

template <typename T>
void SparseMatrixDevice<T>::choose_layouts(bool analyse)
{
  MemoryKind kind("derived layouts (planes, tables, node lists)");
  HipHandle &handle = _handle;
  const int64_t n_rows = _n_rows, n_cols = _n_cols;
  const double avg = n_rows > 0 ? double(_nnz) / double(n_rows) : 0.;
  const auto t_begin = std::chrono::steady_clock::now();
  // lanes per row: about 3-6 entries per lane (measured on R, R^T, A_c and the prolongators, profiles/)
  int lpr = 4;
  while (lpr < 64 && lpr * 2 <= avg / 3.3)
    lpr *= 2;
  if (avg < 3.)
    lpr = avg < 1.5 ? 1 : 2;
  if ((n_rows <= 4096 && avg >= 256.) || (n_rows <= 16384 && avg >= 512.))
    lpr = 256; // a workgroup per row (csr_spmv_row_block_kernel)
  _lanes_per_row = lpr;
  // ---- block-diagonal storage (see bdia_spmv_kernel); when it applies the LDS lists are not needed
  if (analyse && n_rows == n_cols && n_rows >= 32768 && avg >= 8.)
    build_block_diagonals();
  // ---- row-base storage (see rowbase_spmv_kernel) for matrices the block diagonals do not fit
  // (one thread per row: below ~1000 workgroups the CSR kernels with several lanes per row fill the chip better)
  // ---- node classes (see build_node_classes) for the rectangular operators of a translation-invariant problem
  if (analyse && !_use_bdia && n_rows >= 32768 && avg >= 4.)
    build_node_classes();
  if (analyse && !_use_bdia && !_use_nodecls && n_rows >= 200000 && avg >= 4. && avg <= 160.)
  {
    ensure_host_copy();
    build_row_base(_row_ptr_host, _col_host, _val_host);
  }
  // ---- block-local column compression for the LDS-cached kernel
  // (a 128-row block per workgroup: below ~256 blocks the plain kernel fills the chip better)
  if (analyse && !_use_bdia && !_use_rowbase && !_use_nodecls && n_rows >= 256 * kRowsPerBlock && avg >= 4.)
  {
    ensure_host_copy();
    std::vector<int32_t> const &row_ptr = _row_ptr_host, &col = _col_host;
    const int64_t nb = (n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    std::vector<int32_t> blk_ptr(nb + 1, 0);
    std::vector<std::vector<int32_t>> uniq(nb);
    bool ok = true;
    int max_cols = 0;
#pragma omp parallel for schedule(static) reduction(max : max_cols)
    for (int64_t b = 0; b < nb; ++b)
    {
      const int64_t r0 = b * kRowsPerBlock, r1 = std::min<int64_t>(r0 + kRowsPerBlock, n_rows);
      std::vector<int32_t> &u = uniq[b];
      u.assign(col.begin() + row_ptr[r0], col.begin() + row_ptr[r1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      max_cols = std::max<int>(max_cols, (int)u.size());
    }
    int64_t total = 0;
    for (int64_t b = 0; b < nb; ++b)
    {
      total += (int64_t)uniq[b].size();
      blk_ptr[b + 1] = (int32_t)std::min<int64_t>(total, INT32_MAX);
    }
    // worth it when a column is reused at least ~3 times inside a block and the slice fits LDS comfortably
    ok = max_cols <= 6144 && total < INT32_MAX && double(_nnz) >= 3. * double(total);
    if (ok)
    {
      std::vector<int32_t> l2g(total);
      std::vector<uint16_t> lcol(_nnz);
#pragma omp parallel for schedule(static)
      for (int64_t b = 0; b < nb; ++b)
      {
        std::copy(uniq[b].begin(), uniq[b].end(), l2g.begin() + blk_ptr[b]);
        const int64_t r0 = b * kRowsPerBlock, r1 = std::min<int64_t>(r0 + kRowsPerBlock, n_rows);
        for (int64_t p = row_ptr[r0]; p < row_ptr[r1]; ++p)
          lcol[p] = (uint16_t)(std::lower_bound(uniq[b].begin(), uniq[b].end(), col[p]) - uniq[b].begin());
      }
      _blk_ptr.upload(blk_ptr.data(), blk_ptr.size(), handle.stream);
      _l2g.upload(l2g.data(), l2g.size(), handle.stream);
      _lcol.upload(lcol.data(), lcol.size(), handle.stream);
      _use_lds = true;
      _lds_max_cols = max_cols;
    }
  }
  if (std::getenv("MFMG_HIP_VERBOSE") != nullptr)
    std::fprintf(stderr,
                 "[mfmg_hip] matrix %lld x %lld, %.1f entries per row: kernel kind %d (lanes per row %d; block diagonals %d x "
                 "%d components%s, regular rows %d, stencil classes %d, listed rows %lld; row-base slots %d; node classes %d of %d x %d components, %lld rows listed), %.2f s\n",
                 (long long)_n_rows, (long long)_n_cols, avg, kernel_kind(), _lanes_per_row, _bdia_d, _bdia_c,
                 _bdia_sym ? " (symmetric half)" : "", (int)_bdia_regular, _bdia_n_classes,
                 (long long)_bdia_exc_rows.size(), _rb_slots, _nc_classes, _nc_d, _nc_c, (long long)_nc_listed.size(),
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
}

template <typename T>
void SparseMatrixDevice<T>::build_block_diagonals()
{
  ensure_device_csr();
  const int64_t n = _n_rows;
  int best_c = 0;
  std::vector<int32_t> best_offs;
  double best_fill = 0.;
  // block offsets met on a sample of the rows (their column indices fetched once); the fill pass below checks every entry
  int64_t step = std::max<int64_t>(1, n / 65536);
  while (step % 2 == 0 || step % 3 == 0) // the sample must meet every row position inside a node
    ++step;
  const int64_t n_sample = (n + step - 1) / step;
  std::vector<int64_t> sample(n_sample);
  for (int64_t q = 0; q < n_sample; ++q)
    sample[q] = q * step;
  std::vector<int32_t> row_ptr, col; // of the sample: row q = sample[q]
  sample_rows(sample, row_ptr, col);
  for (int c = 1; c <= 4; ++c)
  {
    if (n % c != 0)
      continue;
    std::vector<int32_t> offs;
    bool too_many = false;
    // (every thread collects the offsets of its share of the sample; serial, this loop was the second of the 1.1 s the
    // analysis of a 38 M-entry level took)
#pragma omp parallel
    {
      std::vector<int32_t> mine;
      bool mine_too_many = false;
#pragma omp for schedule(static)
      for (int64_t q = 0; q < n_sample; ++q)
      {
        const int64_t r = sample[q];
        if (mine_too_many)
          continue;
        int32_t last = INT32_MIN;
        for (int p = row_ptr[q]; p < row_ptr[q + 1]; ++p)
        {
          const int32_t o = (int32_t)(col[p] / c - r / c);
          if (o == last)
            continue;
          last = o;
          auto it = std::lower_bound(mine.begin(), mine.end(), o);
          if (it == mine.end() || *it != o)
          {
            mine.insert(it, o);
            if (mine.size() > kMaxBlockDiagonals)
            {
              mine_too_many = true;
              break;
            }
          }
        }
      }
#pragma omp critical
      {
        too_many = too_many || mine_too_many;
        std::vector<int32_t> merged;
        std::set_union(offs.begin(), offs.end(), mine.begin(), mine.end(), std::back_inserter(merged));
        offs.swap(merged);
      }
    }
    too_many = too_many || offs.size() > kMaxBlockDiagonals;
    if (too_many || offs.empty())
      continue;
    const double fill = double(_nnz) / (double(n) * double(offs.size()) * c);
    if (fill > best_fill)
    {
      best_fill = fill;
      best_c = c;
      best_offs = offs;
    }
  }
  if (best_c == 0 || best_fill < 0.8)
    return;
  const int c = best_c, D = (int)best_offs.size();
  if (double(n) * D * c * sizeof(T) > 48e9) // the planes are built in device memory
    return;
  hipStream_t st = _handle.stream;
  const int64_t n_nodes = n / c;
  const size_t tuple = (size_t)c * D * c;
  auto blocks = [](int64_t work) { return dim3(n_blocks_for(work, 256, 1 << 16)); };
  DeviceBuffer<int32_t> d_offs;
  d_offs.upload(best_offs.data(), best_offs.size(), st);
  DeviceBuffer<int> d_flag(1);
  MFMG_HIP_CHECK(hipMemsetAsync(d_flag.data(), 0, sizeof(int), st));
  DeviceBuffer<T> dv((size_t)n * D * c);
  MFMG_HIP_CHECK(hipMemsetAsync(dv.data(), 0, (size_t)n * D * c * sizeof(T), st));
  hipLaunchKernelGGL(bdia_fill_kernel<T>, blocks(n), dim3(256), 0, st, n, c, D, d_offs.data(), _row_ptr.data(), _col.data(), _val.data(),
                     dv.data(), d_flag.data());
  MFMG_HIP_CHECK(hipGetLastError());
  if (d_flag.download(st)[0] != 0)
    return;
  // tuples (k = (rc D + d) c + cc) of a list of nodes, on the host
  auto tuples_of = [&](std::vector<int64_t> const &nodes) {
    std::vector<T> out(nodes.size() * tuple);
    if (nodes.empty())
      return out;
    DeviceBuffer<int64_t> d_nodes;
    d_nodes.upload(nodes.data(), nodes.size(), st);
    DeviceBuffer<T> d_out(out.size());
    hipLaunchKernelGGL(bdia_tuple_kernel<T>, blocks((int64_t)out.size()), dim3(256), 0, st, n, c, D, dv.data(), d_nodes.data(),
                       (int64_t)nodes.size(), d_out.data());
    MFMG_HIP_CHECK(hipGetLastError());
    MFMG_HIP_CHECK(hipMemcpyAsync(out.data(), d_out.data(), out.size() * sizeof(T), hipMemcpyDeviceToHost, st));
    MFMG_HIP_CHECK(hipStreamSynchronize(st));
    return out;
  };
  // Translation invariance: on a uniform mesh with a constant coefficient every interior row of a coarse
  // operator repeats the same stencil, bit for bit.  Rows equal to a reference row (one per position inside a
  // node) are flagged regular and evaluated from a table of D C C constants -- no matrix values are read
  // for them; the others (boundary shells, variable coefficients: all rows) use the stored planes.
  {
    // reference node: of a few candidates spread over the matrix the one whose stencil a sample of rows repeats most
    int64_t ref_node = n_nodes / 2;
    {
      const double frac[] = {0.5, 0.377, 0.613, 0.431, 0.569, 0.289, 0.711, 0.457};
      std::vector<int64_t> cand;
      for (double f : frac)
        cand.push_back(std::min<int64_t>(n_nodes - 1, (int64_t)(f * n_nodes) + 12345 % std::max<int64_t>(n_nodes / 7, 1)));
      DeviceBuffer<int64_t> d_cand;
      d_cand.upload(cand.data(), cand.size(), st);
      DeviceBuffer<int> d_hits(cand.size());
      MFMG_HIP_CHECK(hipMemsetAsync(d_hits.data(), 0, cand.size() * sizeof(int), st));
      hipLaunchKernelGGL(bdia_hits_kernel<T>, dim3((unsigned)((cand.size() * 2048 + 255) / 256)), dim3(256), 0, st, n, c, D, dv.data(), n_nodes,
                         d_cand.data(), (int)cand.size(), 2048, d_hits.data());
      MFMG_HIP_CHECK(hipGetLastError());
      const std::vector<int> hits = d_hits.download(st);
      int best = -1;
      for (size_t q = 0; q < cand.size(); ++q)
        if (hits[q] > best)
        {
          best = hits[q];
          ref_node = cand[q];
        }
    }
    const std::vector<T> table = tuples_of({ref_node});
    DeviceBuffer<T> d_table;
    d_table.upload(table.data(), table.size(), st);
    DeviceBuffer<uint8_t> d_exc((size_t)n);
    hipLaunchKernelGGL(bdia_exc_kernel<T>, blocks(n), dim3(256), 0, st, n, c, D, d_offs.data(), dv.data(), d_table.data(), d_exc.data());
    hipLaunchKernelGGL(bdia_exc_node_kernel, blocks(n_nodes), dim3(256), 0, st, n_nodes, c, d_exc.data());
    MFMG_HIP_CHECK(hipGetLastError());
    std::vector<uint8_t> exc = d_exc.download(st);
    int64_t n_regular = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_regular)
    for (int64_t nd = 0; nd < n_nodes; ++nd)
      n_regular += exc[nd * c] ? 0 : c;
    bool regular_found = false;
    std::vector<int32_t> exc_rows;
    {
      // classes among the other nodes: same stencil bit for bit (hash, then an exact comparison with the first
      // node of the class); classes of a few nodes and nodes that match nothing stay with the stored values
      std::vector<int64_t> enodes;
      for (int64_t nd = 0; nd < n_nodes; ++nd)
        if (exc[nd * c])
          enodes.push_back(nd);
      std::vector<uint64_t> hash(enodes.size());
      DeviceBuffer<int64_t> d_enodes;
      if (!enodes.empty())
      {
        d_enodes.upload(enodes.data(), enodes.size(), st);
        DeviceBuffer<uint64_t> d_hash(enodes.size());
        hipLaunchKernelGGL(bdia_hash_kernel<T>, blocks((int64_t)enodes.size()), dim3(256), 0, st, n, c, D, dv.data(), d_enodes.data(),
                           (int64_t)enodes.size(), d_hash.data());
        MFMG_HIP_CHECK(hipGetLastError());
        hash = d_hash.download(st);
      }
      bool look_for_classes = true;
      if (enodes.size() > 65536)
      {
        // a sample first: with a variable coefficient every node has a stencil of its own
        std::vector<uint64_t> sample;
        for (size_t e = 0; e < enodes.size(); e += enodes.size() / 2048)
          sample.push_back(hash[e]);
        std::sort(sample.begin(), sample.end());
        const size_t distinct = std::unique(sample.begin(), sample.end()) - sample.begin();
        look_for_classes = distinct * 2 < sample.size();
      }
      if (!look_for_classes)
      {
        enodes.clear();
        hash.clear();
      }
      std::vector<int64_t> order(enodes.size());
      std::iota(order.begin(), order.end(), (int64_t)0);
      std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return hash[x] != hash[y] ? hash[x] < hash[y] : x < y; });
      constexpr int64_t kMinClass = 8;
      constexpr int kMaxClasses = 4096;
      // exact comparison of every member of a hash group with the group's first node, on the device
      std::vector<int64_t> rep(enodes.size(), -1);
      for (size_t g0 = 0; g0 < order.size();)
      {
        size_t g1 = g0;
        while (g1 < order.size() && hash[order[g1]] == hash[order[g0]])
          ++g1;
        if ((int64_t)(g1 - g0) >= kMinClass)
          for (size_t q = g0; q < g1; ++q)
            rep[order[q]] = enodes[order[g0]];
        g0 = g1;
      }
      std::vector<uint8_t> same(enodes.size(), 0);
      if (!enodes.empty())
      {
        DeviceBuffer<int64_t> d_rep;
        d_rep.upload(rep.data(), rep.size(), st);
        DeviceBuffer<uint8_t> d_same(enodes.size());
        hipLaunchKernelGGL(bdia_same_kernel<T>, blocks((int64_t)enodes.size()), dim3(256), 0, st, n, c, D, dv.data(), d_enodes.data(),
                           d_rep.data(), (int64_t)enodes.size(), d_same.data());
        MFMG_HIP_CHECK(hipGetLastError());
        same = d_same.download(st);
      }
      std::vector<int32_t> cls_nodes, cls_of_wave;
      std::vector<int64_t> cls_reps;
      std::vector<uint8_t> classed(n_nodes, 0);
      int n_classes = 0;
      for (size_t g0 = 0; g0 < order.size() && n_classes < kMaxClasses;)
      {
        size_t g1 = g0;
        while (g1 < order.size() && hash[order[g1]] == hash[order[g0]])
          ++g1;
        if ((int64_t)(g1 - g0) >= kMinClass)
        {
          std::vector<int64_t> members;
          for (size_t q = g0; q < g1; ++q)
            if (same[order[q]])
              members.push_back(enodes[order[q]]);
          if ((int64_t)members.size() >= kMinClass)
          {
            cls_reps.push_back(enodes[order[g0]]);
            for (size_t q = 0; q < members.size(); ++q)
            {
              if (q % 64 == 0)
                cls_of_wave.push_back(n_classes);
              cls_nodes.push_back((int32_t)members[q]);
              classed[members[q]] = 1;
            }
            while (cls_nodes.size() % 64 != 0)
              cls_nodes.push_back(-1);
            ++n_classes;
          }
        }
        g0 = g1;
      }
      std::vector<T> cls_table = tuples_of(cls_reps);
      int64_t n_classed = 0;
      for (int64_t nd = 0; nd < n_nodes; ++nd)
        n_classed += classed[nd] ? c : 0;
      if ((n_regular + n_classed) * 2 < n)
        n_classes = 0;
      // classes pay while their tables stay in the caches: a few hundred large ones (the boundary shells of a
      // translation-invariant operator: 60 to 240 classes of hundreds to thousands of nodes).  Thousands of small classes
      // -- values that merely coincide, e.g. a linear coefficient on a dyadic mesh after rounding to float -- run slower
      // than the stored planes (measured: 249 + 175 us against 249 us per application of the first coarse operator).
      if (n_classes >= 1024 || n_classed < (int64_t)32 * c * n_classes)
        n_classes = 0;
      // small levels are bound by the latency of a launch, not by their rows: there the regular nodes join the lists
      // as one more class (their table is the reference stencil; a regular node has its whole stencil inside the
      // matrix, so the clamping of the class kernel never acts) and the launch of their own is dropped
      static const int64_t regular_as_class_nodes = [] {
        char const *e = std::getenv("MFMG_REGULAR_AS_CLASS_NODES");
        return e ? std::atoll(e) : kRegularAsClassNodes;
      }();
      if (n_classes > 0 && n_classes < kMaxClasses && n_nodes <= regular_as_class_nodes && n_regular > 0)
      {
        cls_table.insert(cls_table.end(), table.begin(), table.end());
        int64_t q = 0;
        for (int64_t nd = 0; nd < n_nodes; ++nd)
          if (!exc[nd * c])
          {
            if (q++ % 64 == 0)
              cls_of_wave.push_back(n_classes);
            cls_nodes.push_back((int32_t)nd);
            classed[nd] = 1;
            for (int rc = 0; rc < c; ++rc)
              exc[nd * c + rc] = 1;
          }
        while (cls_nodes.size() % 64 != 0)
          cls_nodes.push_back(-1);
        ++n_classes;
        _bdia_all_in_classes = true;
      }
      if (n_classes > 0)
      {
        _bdia_cls_nodes.upload(cls_nodes.data(), cls_nodes.size(), st);
        _bdia_cls_of_wave.upload(cls_of_wave.data(), cls_of_wave.size(), st);
        _bdia_cls_table.upload(cls_table.data(), (size_t)n_classes * tuple, st);
        _bdia_n_classes = n_classes;
      }
      regular_found = n_regular * 2 >= n || n_classes > 0;
      for (int64_t r = 0; r < n && regular_found; ++r)
        if (exc[r] && !(n_classes > 0 && classed[r / c]))
          exc_rows.push_back((int32_t)r);
    }
    if (regular_found)
    {
      _bdia_table.upload(table.data(), table.size(), st);
      _bdia_exc_rows.upload(exc_rows.data(), exc_rows.size(), st);
      _bdia_exc.upload(exc.data(), exc.size(), st);
      _bdia_full_offs.upload(best_offs.data(), best_offs.size(), st);
      _bdia_full_d = D;
      _bdia_regular = true;
    }
    MFMG_HIP_CHECK(hipStreamSynchronize(st)); // (the host vectors of the uploads above go out of scope)
  }
  // stored planes pay up to ~160 block diagonals (one thread walks a row: with more the LDS-cached CSR kernel,
  // several lanes per row, is faster); wider stencils are kept only for their tables
  if (D > kMaxStoredBlockDiagonals && !_bdia_regular)
    return;
  // symmetric (to rounding)?  then the diagonals with a negative offset are the transposes of the positive ones
  bool symmetric = (D % 2 == 1);
  const int zero = D / 2;
  if (symmetric)
    for (int d = 0; d <= zero; ++d)
      symmetric = symmetric && (best_offs[zero + d] == -best_offs[zero - d]);
  if (symmetric)
  {
    // scale of the diagonal from a sample of rows
    std::vector<int64_t> idx;
    for (int64_t r = 0; r < n; r += std::max<int64_t>(1, n / 4096))
      idx.push_back((int64_t)(((size_t)zero * c + (size_t)(r % c)) * n + r));
    DeviceBuffer<int64_t> d_idx;
    d_idx.upload(idx.data(), idx.size(), st);
    DeviceBuffer<T> d_diag(idx.size());
    hipLaunchKernelGGL(gather_strided_kernel<T>, blocks((int64_t)idx.size()), dim3(256), 0, st, dv.data(), d_idx.data(), (int64_t)idx.size(),
                       d_diag.data());
    MFMG_HIP_CHECK(hipGetLastError());
    double scale = 0.;
    for (T v : d_diag.download(st))
      scale = std::max(scale, std::abs((double)v));
    const double tol = 1e-13 * std::max(scale, 1e-300);
    MFMG_HIP_CHECK(hipMemsetAsync(d_flag.data(), 0, sizeof(int), st));
    hipLaunchKernelGGL(bdia_symmetry_kernel<T>, blocks(n), dim3(256), 0, st, n, c, D, d_offs.data(), dv.data(), tol, d_flag.data());
    MFMG_HIP_CHECK(hipGetLastError());
    symmetric = d_flag.download(st)[0] == 0;
  }
  if (symmetric)
  {
    const int Dh = D - zero; // offsets 0 and the positive ones
    std::vector<int32_t> offs_h(best_offs.begin() + zero, best_offs.end());
    _bdia_val.resize((size_t)Dh * c * n);
    MFMG_HIP_CHECK(hipMemcpyAsync(_bdia_val.data(), dv.data() + (size_t)zero * c * n, (size_t)Dh * c * n * sizeof(T), hipMemcpyDeviceToDevice, st));
    _bdia_offs.upload(offs_h.data(), offs_h.size(), st);
    MFMG_HIP_CHECK(hipStreamSynchronize(st));
    _bdia_d = Dh;
    _bdia_sym = true;
  }
  else
  {
    _bdia_val = std::move(dv);
    _bdia_offs.upload(best_offs.data(), best_offs.size(), st);
    MFMG_HIP_CHECK(hipStreamSynchronize(st));
    _bdia_d = D;
  }
  _bdia_c = c;
  _use_bdia = true;
  // planes whose values are all representable in float (a setup that rounds its matrices: "setup value precision"
  // float) are kept in float: half the bytes per application, the same products and sums in T
  if constexpr (sizeof(T) == 8)
  {
    const int64_t n_val = (int64_t)_bdia_val.size();
    MFMG_HIP_CHECK(hipMemsetAsync(d_flag.data(), 0, sizeof(int), st));
    hipLaunchKernelGGL(not_float_kernel<T>, blocks(n_val), dim3(256), 0, st, _bdia_val.data(), n_val, d_flag.data());
    MFMG_HIP_CHECK(hipGetLastError());
    if (n_val > 0 && d_flag.download(st)[0] == 0)
    {
      _bdia_val_f32.resize((size_t)n_val);
      hipLaunchKernelGGL(narrow_kernel<T>, blocks(n_val), dim3(256), 0, st, _bdia_val.data(), n_val, _bdia_val_f32.data());
      MFMG_HIP_CHECK(hipGetLastError());
      MFMG_HIP_CHECK(hipStreamSynchronize(st));
      _bdia_val.release();
    }
  }
}

// Node classes for rectangular stencil-like matrices (the smoothed prolongators of the aggregation hierarchy and
// their transposes, the restrictor): a row node (C rows) couples to the column nodes base[node] + offs[d]; on a
// translation-invariant problem its C D C values repeat one of a few tuples (the position of the node inside its
// aggregate, the distance to the boundary).  Nodes are sorted by (tile of 8192 nodes, class), every run padded to
// whole wavefronts, and evaluated by bdia_class_node_kernel / bdia_node_split_kernel: 4 B of node id and 4 B of base
// per node instead of 12 B per entry.  Rows of nodes that match no class go through csr_listed_rows_kernel.  The
// format is used when at least 90 % of the rows are in classes.
template <typename T>
void SparseMatrixDevice<T>::build_node_classes()
{
  ensure_device_csr();
  const int64_t n = _n_rows, m = _n_cols;
  int best_c = 0;
  std::vector<int32_t> best_offs;
  double best_fill = 0.;
  for (int c = 1; c <= 4; ++c)
  {
    if (n % c != 0 || m % c != 0)
      continue;
    const int64_t n_nodes = n / c;
    std::vector<int32_t> offs;
    bool too_many = false;
    int64_t step = std::max<int64_t>(1, n_nodes / 65536);
    while (step % 2 == 0 || step % 3 == 0)
      ++step;
    const int64_t n_sample = (n_nodes + step - 1) / step;
    // the rows of the sampled nodes (c rows each), their column indices fetched once
    std::vector<int64_t> sample((size_t)n_sample * c);
    for (int64_t q = 0; q < n_sample; ++q)
      for (int rc = 0; rc < c; ++rc)
        sample[q * c + rc] = q * step * c + rc;
    std::vector<int32_t> row_ptr, col; // of the sample
    sample_rows(sample, row_ptr, col);
#pragma omp parallel
    {
      std::vector<int32_t> mine; // (the offsets of a thread's share of the sample, merged below)
      bool mine_too_many = false;
#pragma omp for schedule(static)
      for (int64_t q = 0; q < n_sample; ++q)
      {
        if (mine_too_many)
          continue;
        int64_t b = std::numeric_limits<int64_t>::max(); // first column node of the row node
        for (int p = row_ptr[q * c]; p < row_ptr[(q + 1) * c]; ++p)
          b = std::min<int64_t>(b, col[p] / c);
        for (int64_t r = q * c; r < (q + 1) * c && !mine_too_many; ++r)
        {
          int32_t last = INT32_MIN;
          for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p)
          {
            const int32_t o = (int32_t)(col[p] / c - b);
            if (o == last)
              continue;
            last = o;
            auto it = std::lower_bound(mine.begin(), mine.end(), o);
            if (it == mine.end() || *it != o)
            {
              mine.insert(it, o);
              if (mine.size() > kMaxBlockDiagonals)
              {
                mine_too_many = true;
                break;
              }
            }
          }
        }
      }
#pragma omp critical
      {
        too_many = too_many || mine_too_many;
        std::vector<int32_t> merged;
        std::set_union(offs.begin(), offs.end(), mine.begin(), mine.end(), std::back_inserter(merged));
        offs.swap(merged);
      }
    }
    too_many = too_many || offs.size() > kMaxBlockDiagonals;
    if (too_many || offs.empty())
      continue;
    const double fill = double(_nnz) / (double(n_nodes) * double(offs.size()) * c * c);
    if (fill > 0.98 * best_fill) // (several unknowns per node: one wide load instead of several narrow ones)
    {
      best_fill = fill;
      best_c = c;
      best_offs = offs;
    }
  }
  if (best_c == 0 || best_fill < 0.6)
    return;
  const int c = best_c, D = (int)best_offs.size();
  const int64_t n_nodes = n / c;
  const size_t tuple = (size_t)c * D * c;
  if (double(n_nodes) * tuple * sizeof(T) > 48e9)
    return;
  hipStream_t st = _handle.stream;
  auto blocks = [](int64_t work) { return dim3(n_blocks_for(work, 256, 1 << 16)); };
  DeviceBuffer<int32_t> d_offs;
  d_offs.upload(best_offs.data(), best_offs.size(), st);
  DeviceBuffer<int> d_flag(1);
  MFMG_HIP_CHECK(hipMemsetAsync(d_flag.data(), 0, sizeof(int), st));
  DeviceBuffer<T> tv((size_t)n_nodes * tuple); // [node][rc][d][cc]
  MFMG_HIP_CHECK(hipMemsetAsync(tv.data(), 0, (size_t)n_nodes * tuple * sizeof(T), st));
  DeviceBuffer<int32_t> d_base((size_t)n_nodes);
  hipLaunchKernelGGL(nodecls_fill_kernel<T>, blocks(n_nodes), dim3(256), 0, st, n_nodes, c, D, d_offs.data(), _row_ptr.data(), _col.data(),
                     _val.data(), d_base.data(), tv.data(), d_flag.data());
  MFMG_HIP_CHECK(hipGetLastError());
  if (d_flag.download(st)[0] != 0)
    return;
  std::vector<uint64_t> hash;
  {
    DeviceBuffer<uint64_t> d_hash((size_t)n_nodes);
    hipLaunchKernelGGL(nodecls_hash_kernel<T>, blocks(n_nodes), dim3(256), 0, st, n_nodes, tuple, tv.data(), d_hash.data());
    MFMG_HIP_CHECK(hipGetLastError());
    hash = d_hash.download(st);
  }
  std::vector<int64_t> order(n_nodes);
  std::iota(order.begin(), order.end(), (int64_t)0);
  // (all host threads: a serial sort of the 2.1 M nodes of the first prolongator took 0.25 s)
  __gnu_parallel::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return hash[x] != hash[y] ? hash[x] < hash[y] : x < y; });
  constexpr int64_t kMinClass = 8, kTile = 8192;
  constexpr int kMaxClasses = 4096;
  // exact comparison of every member of a hash group with the group's first node, on the device
  std::vector<uint8_t> same;
  {
    std::vector<int64_t> rep(n_nodes, -1);
    for (int64_t g0 = 0; g0 < n_nodes;)
    {
      int64_t g1 = g0;
      while (g1 < n_nodes && hash[order[g1]] == hash[order[g0]])
        ++g1;
      if (g1 - g0 >= kMinClass)
        for (int64_t q = g0; q < g1; ++q)
          rep[order[q]] = order[g0];
      g0 = g1;
    }
    DeviceBuffer<int64_t> d_rep;
    d_rep.upload(rep.data(), rep.size(), st);
    DeviceBuffer<uint8_t> d_same((size_t)n_nodes);
    hipLaunchKernelGGL(nodecls_same_kernel<T>, blocks(n_nodes), dim3(256), 0, st, n_nodes, tuple, tv.data(), d_rep.data(), d_same.data());
    MFMG_HIP_CHECK(hipGetLastError());
    same = d_same.download(st);
  }
  std::vector<int32_t> cls_of_node(n_nodes, -1);
  std::vector<int64_t> cls_reps;
  int n_classes = 0;
  int64_t n_classed = 0;
  for (int64_t g0 = 0; g0 < n_nodes && n_classes < kMaxClasses;)
  {
    int64_t g1 = g0;
    while (g1 < n_nodes && hash[order[g1]] == hash[order[g0]])
      ++g1;
    if (g1 - g0 >= kMinClass)
    {
      int64_t members = 0;
      for (int64_t q = g0; q < g1; ++q)
        if (same[order[q]])
        {
          cls_of_node[order[q]] = n_classes;
          ++members;
        }
      if (members >= kMinClass)
      {
        cls_reps.push_back(order[g0]);
        n_classed += members;
        ++n_classes;
      }
      else
        for (int64_t q = g0; q < g1; ++q)
          if (cls_of_node[order[q]] == n_classes)
            cls_of_node[order[q]] = -1;
    }
    g0 = g1;
  }
  if (n_classes == 0 || n_classed * 10 < n_nodes * 9)
    return;
  if (n_classes >= 1024 || n_classed < (int64_t)32 * n_classes)
    return; // thousands of small classes: tables that do not stay in the caches (see build_block_diagonals)
  // (tile, class, node) order, runs padded to whole wavefronts; tiles keep the rows of a wavefront close together
  // (partial cache lines of the output are completed by wavefronts of other classes of the same tile), the whole
  // matrix as one tile where small tiles would mostly hold padding
  std::vector<int32_t> nodes, class_of_wave, listed;
  for (int64_t nd = 0; nd < n_nodes; ++nd)
    if (cls_of_node[nd] < 0)
      for (int rc = 0; rc < c; ++rc)
        listed.push_back((int32_t)(nd * c + rc));
  std::vector<std::pair<int32_t, int32_t>> in_tile; // (class, node)
  bool done = false;
  for (int64_t tile : {kTile, 8 * kTile, n_nodes})
  {
    if (done)
      break;
    nodes.clear();
    class_of_wave.clear();
    for (int64_t t0 = 0; t0 < n_nodes; t0 += tile)
    {
      in_tile.clear();
      for (int64_t nd = t0; nd < std::min(n_nodes, t0 + tile); ++nd)
        if (cls_of_node[nd] >= 0)
          in_tile.emplace_back(cls_of_node[nd], (int32_t)nd);
      std::sort(in_tile.begin(), in_tile.end());
      for (size_t q = 0; q < in_tile.size(); ++q)
      {
        if (q > 0 && in_tile[q].first != in_tile[q - 1].first)
          while (nodes.size() % 64 != 0)
            nodes.push_back(-1);
        if (nodes.size() % 64 == 0)
          class_of_wave.push_back(in_tile[q].first);
        nodes.push_back(in_tile[q].second);
      }
      while (nodes.size() % 64 != 0)
        nodes.push_back(-1);
    }
    done = (double)nodes.size() <= 1.25 * (double)n_classed || tile >= n_nodes;
  }
  if ((double)nodes.size() > 2. * (double)n_classed)
    return; // mostly padding: the CSR kernels do better
  _nc_base = std::move(d_base);
  _nc_offs.upload(best_offs.data(), best_offs.size(), st);
  _nc_nodes.upload(nodes.data(), nodes.size(), st);
  _nc_class_of_wave.upload(class_of_wave.data(), class_of_wave.size(), st);
  _nc_listed.upload(listed.data(), listed.size(), st);
  // the class tables: the tuples of the representatives, copied inside the device
  _nc_table.resize((size_t)n_classes * tuple);
  for (int q = 0; q < n_classes; ++q)
    MFMG_HIP_CHECK(hipMemcpyAsync(_nc_table.data() + (size_t)q * tuple, tv.data() + (size_t)cls_reps[q] * tuple, tuple * sizeof(T),
                                  hipMemcpyDeviceToDevice, st));
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
  _nc_c = c;
  _nc_d = D;
  _nc_classes = n_classes;
  _use_nodecls = true;
}

template <typename T>
void SparseMatrixDevice<T>::build_row_base(std::vector<int32_t> const &row_ptr, std::vector<int32_t> const &col,
                                           std::vector<T> const &val)
{
  const int64_t n = _n_rows;
  // the offset list of the fullest rows (columns are sorted inside a row)
  int max_len = 0;
  for (int64_t r = 0; r < n; ++r)
    max_len = std::max(max_len, row_ptr[r + 1] - row_ptr[r]);
  if (max_len < 2 || max_len > 192)
    return;
  std::vector<int32_t> offs;
  for (int64_t r = 0; r < n; ++r)
    if (row_ptr[r + 1] - row_ptr[r] == max_len)
    {
      for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p)
        offs.push_back(col[p] - col[row_ptr[r]]);
      break;
    }
  const int S = (int)offs.size();
  if (double(_nnz) < 0.8 * double(n) * S)
    return;
  // every row: a base such that all its columns fall on slots (rows at a boundary miss the low slots)
  std::vector<int32_t> base(n, 0);
  ZeroedHostArray<T> dv((size_t)n * S);
  bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
  for (int64_t r = 0; r < n; ++r)
  {
    const int p0 = row_ptr[r], p1 = row_ptr[r + 1];
    if (p0 == p1)
    {
      base[r] = -(1 << 30); // no column in range
      continue;
    }
    bool placed = false;
    for (int s0 = 0; s0 < S && !placed; ++s0)
    {
      const int64_t b = (int64_t)col[p0] - offs[s0];
      int sidx = s0;
      bool fit = true;
      for (int p = p0; p < p1 && fit; ++p)
      {
        while (sidx < S && b + offs[sidx] < col[p])
          ++sidx;
        fit = sidx < S && b + offs[sidx] == col[p];
      }
      if (!fit || b < -(int64_t(1) << 30) || b > (int64_t(1) << 30))
        continue;
      base[r] = (int32_t)b;
      sidx = s0;
      for (int p = p0; p < p1; ++p)
      {
        while (b + offs[sidx] < col[p])
          ++sidx;
        dv[(size_t)sidx * n + r] += val[p];
      }
      placed = true;
    }
    if (!placed)
      ok = false;
  }
  if (!ok)
    return;
  _rb_val.upload(dv.data(), dv.size(), _handle.stream);
  _rb_base.upload(base.data(), base.size(), _handle.stream);
  _rb_offs.upload(offs.data(), offs.size(), _handle.stream);
  _rb_slots = S;
  _use_rowbase = true;
}

template <typename T>
void SparseMatrixDevice<T>::launch(CsrMode mode, T const *x, T const *b, T const *dinv, T const *x_prev,
                                   T alpha, T beta, T *out) const
{
  ASSERT_THROW(x != nullptr && out != nullptr, "null vector");
  ASSERT_THROW(x != out, "SpMV cannot run in place (out aliases x)");
  if (_n_rows == 0)
    return;
  if (!_csr_released)
    ensure_device_csr();
  CsrArgs<T> a;
  a.val = _val.data();
  a.col = _col.data();
  a.row_ptr = _row_ptr.data();
  if (_csr_released)
  {
    ASSERT_THROW((_use_nodecls && _use_regular) || (_use_bdia && _bdia_regular && _use_regular), "internal: released CSR arrays with a kernel that reads them");
    a.val = _kept_val.data();
    a.col = _kept_col.data();
    a.row_ptr = nullptr;
    a.kept_ptr = _kept_ptr.data();
  }
  a.n_rows = _n_rows;
  a.x = x;
  a.b = b;
  a.dinv = dinv;
  a.xprev = x_prev;
  a.out = out;
  a.alpha = alpha;
  a.beta = beta;
  a.mode = static_cast<int>(mode);
  {
    auto aligned = [](void const *p) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    a.pairs = (aligned(x) && aligned(b) && aligned(dinv) && aligned(x_prev) && aligned(out)) ? 1 : 0;
  }
  hipStream_t st = _handle.stream;
  const double extra = (mode == CsrMode::apply) ? 0. : (mode == CsrMode::first) ? 3. : (mode == CsrMode::next) ? 4. : 1.;
  hipEvent_t stop =
      _handle.profiler.begin("csr_spmv_kernel", algorithmic_bytes_apply() + extra * sizeof(T) * double(_n_rows), st);
  if (_use_nodecls && _use_regular)
  {
    BdiaRegular<T> g;
    g.exc = nullptr;
    g.exc_rows = nullptr;
    g.n_exc = 0;
    g.table = nullptr;
    g.offs = _nc_offs.data();
    g.Df = _nc_d;
    g.base = _nc_base.data();
    g.n_col_nodes = _n_cols / _nc_c;
    const int64_t n_slots = (int64_t)_nc_nodes.size();
    int32_t const *cn = _nc_nodes.data(), *cw = _nc_class_of_wave.data();
    T const *ct = _nc_table.data();
    const bool split = g.Df >= kSplitStencil;
    const bool many_parts = g.Df >= 4 * kSplitStencil && _n_rows / _nc_c <= 131072;
    int32_t const *ls = _nc_listed.data();
    const int64_t n_listed = (int64_t)_nc_listed.size();
    auto launch_nodes = [&](auto cc) {
      constexpr int C = decltype(cc)::value;
      if (split && many_parts)
        hipLaunchKernelGGL((bdia_node_split_kernel<T, C, true, 16>),
                           dim3((unsigned int)((n_slots + 63) / 64 + (n_listed + 15) / 16)), dim3(1024), 0, st, a, g, cn, cw,
                           ct, n_slots, ls, n_listed);
      else if (split)
        hipLaunchKernelGGL((bdia_node_split_kernel<T, C, true, 4>),
                           dim3((unsigned int)((n_slots + 255) / 256 + (n_listed + 15) / 16)), dim3(1024), 0, st, a, g, cn,
                           cw, ct, n_slots, ls, n_listed);
      else
        hipLaunchKernelGGL((bdia_class_node_kernel<T, C>), dim3((unsigned int)((n_slots + 255) / 256 + (n_listed + 3) / 4)),
                           dim3(256), 0, st, a, g, cn, cw, ct, n_slots, ls, n_listed);
    };
    switch (_nc_c)
    {
    case 1:
      launch_nodes(std::integral_constant<int, 1>());
      break;
    case 2:
      launch_nodes(std::integral_constant<int, 2>());
      break;
    case 3:
      launch_nodes(std::integral_constant<int, 3>());
      break;
    default:
      launch_nodes(std::integral_constant<int, 4>());
      break;
    }
    KernelProfiler::end(stop, st);
    MFMG_HIP_CHECK(hipGetLastError());
    return;
  }
  if (_use_rowbase)
  {
    hipLaunchKernelGGL(rowbase_spmv_kernel<T>, dim3((unsigned int)((_n_rows + 255) / 256)), dim3(256), 0, st, a,
                       _rb_val.data(), _rb_base.data(), _rb_offs.data(), _rb_slots, _n_cols);
    KernelProfiler::end(stop, st);
    MFMG_HIP_CHECK(hipGetLastError());
    return;
  }
  if (_use_bdia)
  {
    T const *dv = _bdia_val.data();
    int32_t const *of = _bdia_offs.data();
    BdiaRegular<T> g;
    g.exc = (_bdia_regular && _use_regular) ? _bdia_exc.data() : nullptr;
    g.table = _bdia_table.data();
    g.offs = _bdia_full_offs.data();
    g.Df = _bdia_full_d;
    g.exc_rows = _bdia_exc_rows.data();
    g.n_exc = (int64_t)_bdia_exc_rows.size();
    g.n_col_nodes = _n_rows / _bdia_c;
    if (g.exc != nullptr)
    {
      // regular nodes, then the classes of the others; wide stencils split over the wavefronts of a workgroup
      const bool split = g.Df >= kSplitStencil;
      const bool many_parts = g.Df >= 4 * kSplitStencil && _n_rows / _bdia_c <= 131072;
      const int64_t n_nodes = _n_rows / _bdia_c;
      const int64_t n_slots = (int64_t)_bdia_cls_nodes.size();
      int32_t const *cn = _bdia_cls_nodes.data(), *cw = _bdia_cls_of_wave.data();
      T const *ct = _bdia_cls_table.data();
      // few listed rows ride at the end of the class launch, one wavefront each
      const int64_t n_tail = (_bdia_n_classes > 0 && g.n_exc <= kListedWaveRows) ? g.n_exc : 0;
      auto launch_nodes = [&](auto cc) {
        constexpr int C = decltype(cc)::value;
        if (_bdia_all_in_classes)
          ; // (the regular nodes are one of the classes)
        else if (split && many_parts)
          hipLaunchKernelGGL((bdia_node_split_kernel<T, C, false, 16>), dim3((unsigned int)((n_nodes + 63) / 64)),
                             dim3(1024), 0, st, a, g, nullptr, nullptr, nullptr, 0, nullptr, 0);
        else if (split)
          hipLaunchKernelGGL((bdia_node_split_kernel<T, C, false, 4>), dim3((unsigned int)((n_nodes + 255) / 256)),
                             dim3(1024), 0, st, a, g, nullptr, nullptr, nullptr, 0, nullptr, 0);
        else
          hipLaunchKernelGGL((bdia_regular_node_kernel<T, C>), dim3((unsigned int)((n_nodes + 255) / 256)), dim3(256), 0,
                             st, a, g);
        if (_bdia_n_classes > 0 && split && many_parts)
          hipLaunchKernelGGL((bdia_node_split_kernel<T, C, true, 16>),
                             dim3((unsigned int)((n_slots + 63) / 64 + (n_tail + 15) / 16)), dim3(1024), 0, st, a, g, cn, cw,
                             ct, n_slots, g.exc_rows, n_tail);
        else if (_bdia_n_classes > 0 && split)
          hipLaunchKernelGGL((bdia_node_split_kernel<T, C, true, 4>),
                             dim3((unsigned int)((n_slots + 255) / 256 + (n_tail + 15) / 16)), dim3(1024), 0, st, a, g, cn,
                             cw, ct, n_slots, g.exc_rows, n_tail);
        else if (_bdia_n_classes > 0)
          hipLaunchKernelGGL((bdia_class_node_kernel<T, C>), dim3((unsigned int)((n_slots + 255) / 256 + (n_tail + 3) / 4)),
                             dim3(256), 0, st, a, g, cn, cw, ct, n_slots, g.exc_rows, n_tail);
      };
      switch (_bdia_c)
      {
      case 1:
        launch_nodes(std::integral_constant<int, 1>());
        break;
      case 2:
        launch_nodes(std::integral_constant<int, 2>());
        break;
      case 3:
        launch_nodes(std::integral_constant<int, 3>());
        break;
      default:
        launch_nodes(std::integral_constant<int, 4>());
        break;
      }
      if (g.n_exc <= kListedWaveRows)
      {
        if (g.n_exc > 0 && n_tail == 0)
          hipLaunchKernelGGL(csr_listed_rows_kernel<T>, dim3((unsigned int)((g.n_exc + 3) / 4)), dim3(256), 0, st, a,
                             g.exc_rows, g.n_exc);
        KernelProfiler::end(stop, st);
        MFMG_HIP_CHECK(hipGetLastError());
        return;
      }
    }
    const dim3 rgrid((unsigned int)(((g.exc != nullptr ? g.n_exc : _n_rows) + 255) / 256));
    // the stored planes: in T, or in float where every value is representable in it (half the bytes, sums in T)
    auto stored = [&](auto const *planes, auto cc) {
      using V = std::remove_cv_t<std::remove_pointer_t<decltype(planes)>>;
      constexpr int C = decltype(cc)::value;
      if (_bdia_sym && g.exc == nullptr)
        hipLaunchKernelGGL((bdia_sym_split_kernel<T, C, V>), dim3((unsigned int)((_n_rows + 255) / 256)), dim3(1024), 0, st, a, planes, of,
                           _bdia_d);
      else if (_bdia_sym)
      {
        if (rgrid.x > 0)
          hipLaunchKernelGGL((bdia_sym_spmv_kernel<T, C, V>), rgrid, dim3(256), 0, st, a, planes, of, _bdia_d, g);
      }
      else if (rgrid.x > 0)
        hipLaunchKernelGGL((bdia_spmv_kernel<T, C, V>), rgrid, dim3(256), 0, st, a, planes, of, _bdia_d,
                           g.exc != nullptr ? g.exc_rows : nullptr, g.n_exc);
    };
    auto stored_c = [&](auto const *planes) {
      switch (_bdia_c)
      {
      case 1:
        stored(planes, std::integral_constant<int, 1>());
        break;
      case 2:
        stored(planes, std::integral_constant<int, 2>());
        break;
      case 3:
        stored(planes, std::integral_constant<int, 3>());
        break;
      default:
        stored(planes, std::integral_constant<int, 4>());
        break;
      }
    };
    if (_bdia_val_f32.size() > 0)
      stored_c(_bdia_val_f32.data());
    else
      stored_c(dv);
    KernelProfiler::end(stop, st);
    MFMG_HIP_CHECK(hipGetLastError());
    return;
  }
  if (_use_lds)
  {
    int32_t const *bp = _blk_ptr.data(), *lg = _l2g.data();
    uint16_t const *lc = _lcol.data();
    switch (_lanes_per_row)
    {
    case 1:
    case 2:
    case 4:
      launch_lds<T, 4>(a, st, bp, lg, lc, kRowsPerBlock, _lds_max_cols);
      break;
    case 8:
      launch_lds<T, 8>(a, st, bp, lg, lc, kRowsPerBlock, _lds_max_cols);
      break;
    case 16:
      launch_lds<T, 16>(a, st, bp, lg, lc, kRowsPerBlock, _lds_max_cols);
      break;
    case 32:
      launch_lds<T, 32>(a, st, bp, lg, lc, kRowsPerBlock, _lds_max_cols);
      break;
    default:
      launch_lds<T, 64>(a, st, bp, lg, lc, kRowsPerBlock, _lds_max_cols);
      break;
    }
    KernelProfiler::end(stop, st);
    MFMG_HIP_CHECK(hipGetLastError());
    return;
  }
  switch (_lanes_per_row)
  {
  case 256:
    hipLaunchKernelGGL(csr_spmv_row_block_kernel<T>, dim3((unsigned int)_n_rows), dim3(256), 0, st, a);
    break;
  case 1:
    launch_lpr<T, 1>(a, st);
    break;
  case 2:
    launch_lpr<T, 2>(a, st);
    break;
  case 4:
    launch_lpr<T, 4>(a, st);
    break;
  case 8:
    launch_lpr<T, 8>(a, st);
    break;
  case 16:
    launch_lpr<T, 16>(a, st);
    break;
  case 32:
    launch_lpr<T, 32>(a, st);
    break;
  default:
    launch_lpr<T, 64>(a, st);
    break;
  }
  KernelProfiler::end(stop, st);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void SparseMatrixDevice<T>::inverse_diagonal(T *dinv) const
{
  ASSERT_THROW(_n_rows == _n_cols, "The matrix is not square. The matrix is a " + std::to_string(_n_rows) +
                                       " by " + std::to_string(_n_cols) + " .");
  if (_n_rows == 0)
    return;
  ensure_device_csr();
  hipLaunchKernelGGL(csr_inv_diag_kernel<T>, dim3(n_blocks_for(_n_rows)), dim3(block_size), 0,
                     _handle.stream, _val.data(), _col.data(), _row_ptr.data(), _n_rows, dinv);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void SparseMatrixDevice<T>::row_ratios(T *dinv, T *ratio) const
{
  ASSERT_THROW(_n_rows == _n_cols, "row_ratios needs a square matrix");
  ensure_device_csr();
  hipLaunchKernelGGL(csr_row_ratio_kernel<T>, dim3(n_blocks_for(_n_rows)), dim3(block_size), 0, _handle.stream, _val.data(), _col.data(),
                     _row_ptr.data(), _n_rows, dinv, ratio);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void SparseMatrixDevice<T>::download(std::vector<int32_t> &row_ptr, std::vector<int32_t> &col,
                                     std::vector<T> &val) const
{
  ASSERT_THROW(!_csr_released, "the CSR arrays of this matrix were released after the setup (\"release setup matrices\")");
  if (_device_csr_deferred)
  {
    row_ptr = _row_ptr_host;
    col = _col_host;
    val = _val_host;
    return;
  }
  row_ptr = _row_ptr.download(_handle.stream);
  col = _col.download(_handle.stream);
  val = _val.download(_handle.stream);
}

template <typename T>
void csr_transpose_host(int64_t n_rows, int64_t n_cols, std::vector<int32_t> const &row_ptr,
                        std::vector<int32_t> const &col, std::vector<T> const &val,
                        std::vector<int32_t> &t_row_ptr, std::vector<int32_t> &t_col, std::vector<T> &t_val)
{
  const int64_t nnz = row_ptr.empty() ? 0 : row_ptr[n_rows];
  t_row_ptr.assign(n_cols + 1, 0);
  for (int64_t p = 0; p < nnz; ++p)
    t_row_ptr[col[p] + 1]++;
  for (int64_t c = 0; c < n_cols; ++c)
    t_row_ptr[c + 1] += t_row_ptr[c];
  t_col.resize(nnz);
  t_val.resize(nnz);
  std::vector<int32_t> next(t_row_ptr.begin(), t_row_ptr.end() - 1);
  for (int64_t r = 0; r < n_rows; ++r)
    for (int p = row_ptr[r]; p < row_ptr[r + 1]; ++p)
    {
      const int q = next[col[p]]++;
      t_col[q] = (int32_t)r; // rows visited in order -> sorted columns
      t_val[q] = val[p];
    }
}

// Gustavson row-by-row SpGEMM with a dense accumulator per thread; columns sorted.
template <typename T>
void csr_multiply_host(int64_t a_rows, int64_t a_cols, std::vector<int32_t> const &a_ptr,
                       std::vector<int32_t> const &a_col, std::vector<T> const &a_val, int64_t b_cols,
                       std::vector<int32_t> const &b_ptr, std::vector<int32_t> const &b_col,
                       std::vector<T> const &b_val, std::vector<int32_t> &c_ptr, std::vector<int32_t> &c_col,
                       std::vector<T> &c_val)
{
  (void)a_cols;
  configure_host_threads();
  c_ptr.assign(a_rows + 1, 0);
  std::vector<std::vector<int32_t>> row_cols(a_rows);
  std::vector<std::vector<T>> row_vals(a_rows);
#pragma omp parallel
  {
    std::vector<T> acc(b_cols, T(0));
    std::vector<char> mark(b_cols, 0);
    std::vector<int32_t> touched;
#pragma omp for schedule(dynamic, 256)
    for (int64_t r = 0; r < a_rows; ++r)
    {
      touched.clear();
      for (int p = a_ptr[r]; p < a_ptr[r + 1]; ++p)
      {
        const int k = a_col[p];
        const T av = a_val[p];
        for (int q = b_ptr[k]; q < b_ptr[k + 1]; ++q)
        {
          const int c = b_col[q];
          if (!mark[c])
          {
            mark[c] = 1;
            touched.push_back(c);
          }
          acc[c] += av * b_val[q];
        }
      }
      std::sort(touched.begin(), touched.end());
      row_cols[r].assign(touched.begin(), touched.end());
      row_vals[r].resize(touched.size());
      for (size_t t = 0; t < touched.size(); ++t)
      {
        row_vals[r][t] = acc[touched[t]];
        acc[touched[t]] = T(0);
        mark[touched[t]] = 0;
      }
    }
  }
  int64_t total = 0;
  for (int64_t r = 0; r < a_rows; ++r)
  {
    total += (int64_t)row_cols[r].size();
    ASSERT_THROW(total < (int64_t(1) << 31), "SpGEMM result exceeds int32 nnz");
    c_ptr[r + 1] = (int32_t)total;
  }
  c_col.resize(total);
  c_val.resize(total);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < a_rows; ++r)
  {
    std::copy(row_cols[r].begin(), row_cols[r].end(), c_col.begin() + c_ptr[r]);
    std::copy(row_vals[r].begin(), row_vals[r].end(), c_val.begin() + c_ptr[r]);
  }
}

// MFMG_CSR_ALGEBRA=host forces the host algorithms (the tests compare the two paths bit for bit); =device_only
// turns the fall-back for rows beyond the LDS tables into an error (so that a test knows which path it measured)
static bool csr_algebra_on_device()
{
  char const *e = std::getenv("MFMG_CSR_ALGEBRA");
  return !(e && std::string(e) == "host");
}
static void csr_algebra_fell_back(char const *what)
{
  char const *e = std::getenv("MFMG_CSR_ALGEBRA");
  ASSERT_THROW(!(e && std::string(e) == "device_only"),
               std::string(what) + ": a row exceeds the LDS tables of the device algorithm (MFMG_CSR_ALGEBRA=device_only)");
}

template <typename T>
std::shared_ptr<SparseMatrixDevice<T>> SparseMatrixDevice<T>::transpose() const
{
  std::vector<int32_t> rp, cl, trp, tcl;
  std::vector<T> vl, tvl;
  ensure_device_csr();
  const int64_t nnz = (int64_t)_val.size();
  if (csr_algebra_on_device() && nnz > 0)
  {
    // (the transposed arrays stay on the device: the layout analysis of the new matrix runs there)
    DeviceBuffer<int32_t> d_ptr, d_col;
    DeviceBuffer<T> d_val;
    if (csr_transpose_device<T>(_handle, _n_rows, _n_cols, nnz, _row_ptr.data(), _col.data(), _val.data(), d_ptr, d_col, d_val))
      return std::make_shared<SparseMatrixDevice<T>>(_handle, _n_cols, _n_rows, std::move(d_ptr), std::move(d_col), std::move(d_val));
  }
  if (csr_algebra_on_device() && nnz > 0)
    csr_algebra_fell_back("transpose");
  if (has_host_copy())
    csr_transpose_host<T>(_n_rows, _n_cols, _row_ptr_host, _col_host, _val_host, trp, tcl, tvl);
  else
  {
    download(rp, cl, vl);
    csr_transpose_host<T>(_n_rows, _n_cols, rp, cl, vl, trp, tcl, tvl);
  }
  return std::make_shared<SparseMatrixDevice<T>>(_handle, _n_cols, _n_rows, std::move(trp), std::move(tcl),
                                                 std::move(tvl));
}

template <typename T>
std::shared_ptr<SparseMatrixDevice<T>> SparseMatrixDevice<T>::mmult(SparseMatrixDevice<T> const &b) const
{
  ASSERT_THROW(_n_cols == b.m(), "The matrices cannot be multiplied together because their sizes are "
                                 "incompatible.");
  ensure_device_csr();
  b.ensure_device_csr();
  std::vector<int32_t> arp, acl, brp, bcl, crp, ccl;
  std::vector<T> avl, bvl, cvl;
  if (csr_algebra_on_device() && _val.size() > 0 && b._val.size() > 0 &&
      csr_multiply_device<T>(_handle, _n_rows, _row_ptr.data(), _col.data(), _val.data(), b._row_ptr.data(), b._col.data(),
                             b._val.data(), crp, ccl, cvl, b.n()))
    return std::make_shared<SparseMatrixDevice<T>>(_handle, _n_rows, b.n(), std::move(crp), std::move(ccl), std::move(cvl));
  if (csr_algebra_on_device() && _val.size() > 0 && b._val.size() > 0)
    csr_algebra_fell_back("mmult");
  std::vector<int32_t> const *ap = &_row_ptr_host, *ac = &_col_host;
  std::vector<T> const *av = &_val_host;
  if (!has_host_copy())
  {
    download(arp, acl, avl);
    ap = &arp;
    ac = &acl;
    av = &avl;
  }
  std::vector<int32_t> const *bp = &b._row_ptr_host, *bc = &b._col_host;
  std::vector<T> const *bv = &b._val_host;
  if (!b.has_host_copy())
  {
    b.download(brp, bcl, bvl);
    bp = &brp;
    bc = &bcl;
    bv = &bvl;
  }
  csr_multiply_host<T>(_n_rows, _n_cols, *ap, *ac, *av, b.n(), *bp, *bc, *bv, crp, ccl, cvl);
  return std::make_shared<SparseMatrixDevice<T>>(_handle, _n_rows, b.n(), std::move(crp), std::move(ccl),
                                                 std::move(cvl));
}

template class SparseMatrixDevice<double>;
template class SparseMatrixDevice<float>;
template void csr_transpose_host<double>(int64_t, int64_t, std::vector<int32_t> const &,
                                         std::vector<int32_t> const &, std::vector<double> const &,
                                         std::vector<int32_t> &, std::vector<int32_t> &, std::vector<double> &);
template void csr_transpose_host<float>(int64_t, int64_t, std::vector<int32_t> const &,
                                        std::vector<int32_t> const &, std::vector<float> const &,
                                        std::vector<int32_t> &, std::vector<int32_t> &, std::vector<float> &);
template void csr_multiply_host<double>(int64_t, int64_t, std::vector<int32_t> const &,
                                        std::vector<int32_t> const &, std::vector<double> const &, int64_t,
                                        std::vector<int32_t> const &, std::vector<int32_t> const &,
                                        std::vector<double> const &, std::vector<int32_t> &,
                                        std::vector<int32_t> &, std::vector<double> &);
template void csr_multiply_host<float>(int64_t, int64_t, std::vector<int32_t> const &,
                                       std::vector<int32_t> const &, std::vector<float> const &, int64_t,
                                       std::vector<int32_t> const &, std::vector<int32_t> const &,
                                       std::vector<float> const &, std::vector<int32_t> &,
                                       std::vector<int32_t> &, std::vector<float> &);
} // namespace mfmg
