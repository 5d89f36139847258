// BASELINE.json configs[4]: "cell-local evaluation as batched dense contraction on MFMA".
//
// For a cell-wise constant coefficient the local operator of every cell is ONE reference matrix scaled by the cell's
// coefficient: v_e = c_e K_ref u_e (K_ref 8 x 8, Cartesian Q1 cell).  Over a batch of cells that is the GEMM
// V[8 x cells] = (K_ref U[8 x cells]) diag(c) -- the contraction the matrix cores can do.  This file holds that
// contraction on its own, on planar operands (U and V as eight planes of one value per cell, the layout in which the
// MFMA operands need no lane shuffles), in two variants:
//   valu : one lane per cell, the 8 x 8 product as 64 multiply-adds in registers;
//   mfma : v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64, A = K_ref (rows 8..15 of the 16 x 16 tile zero: the
//          8 x 8 operator fills half of it), B = the U values of 16 cells, two k-steps per 16 cells.
// It exists to MEASURE the MFMA formulation (bench.py: cell_contraction_*): both variants stream 68 B (FP32) or
// 136 B (FP64) per cell and sit on the same HBM roofline; inside the fused operator kernel the cell arithmetic is
// 1-4 % of the launch (profiles/r02_c_operator_kernel_ablation.txt), which is why the operator kernel keeps the
// vector ALU form.  The reference has no counterpart (its cell kernel is deal.II's FEEvaluation on the CPU,
// tests/laplace_matrix_free.hpp:138-154).
#include "cell_contraction.hpp"

#include "amge_structured.hpp"

namespace mfmg
{
namespace
{
template <typename T>
struct KRef
{
  T k[8][8];
};

template <typename T>
__global__ __launch_bounds__(256) void cell_contraction_valu_kernel(int64_t n, T const *u, T const *c, T *v, KRef<T> K)
{
#pragma clang fp contract(off)
  for (int64_t cell = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; cell < n; cell += (int64_t)gridDim.x * blockDim.x)
  {
    T uu[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      uu[k] = u[(size_t)k * n + cell];
    const T cv = c[cell];
    // the dense 8 x 8 product in registers: 64 multiply-adds per cell
#pragma unroll
    for (int m = 0; m < 8; ++m)
    {
      T s = T(0);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        s = __builtin_fma(K.k[m][k], uu[k], s);
      v[(size_t)m * n + cell] = cv * s;
    }
  }
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

// one wavefront per 64 consecutive cells and iteration: 4 groups of 16 cells, 2 k-steps each
__global__ __launch_bounds__(256) void cell_contraction_mfma_f32_kernel(int64_t n, float const *u, float const *c, float *v,
                                                                         KRef<float> K)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int col = lane & 15, kq = lane >> 4;
  // A operand: lane l holds A[i = l & 15][k = 4 s + (l >> 4)] = K_ref[i][k] for i < 8, zero rows below
  const float a0 = (col < 8) ? K.k[col][kq] : 0.f;
  const float a1 = (col < 8) ? K.k[col][4 + kq] : 0.f;
  for (int64_t base = wave * 64; base < n; base += n_waves * 64)
  {
#pragma unroll
    for (int g = 0; g < 4; ++g)
    {
      const int64_t cell = base + 16 * g + col;
      const bool ok = cell < n;
      // B operand: lane l holds B[k = 4 s + (l >> 4)][j = l & 15] = U[k][cell j of the group]
      const float b0 = ok ? u[(size_t)kq * n + cell] : 0.f;
      const float b1 = ok ? u[(size_t)(4 + kq) * n + cell] : 0.f;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
      // D: lane l holds rows 4 (l >> 4) + r of column l & 15: the corners m = 0..7 sit in lanes 0..31
      if (ok && kq < 2)
      {
        const float cv = c[cell];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[(size_t)(4 * kq + r) * n + cell] = cv * acc[r];
      }
    }
  }
}

__global__ __launch_bounds__(256) void cell_contraction_mfma_f64_kernel(int64_t n, double const *u, double const *c, double *v,
                                                                         KRef<double> K)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const double a0 = (col < 8) ? K.k[col][kq] : 0.;
  const double a1 = (col < 8) ? K.k[col][4 + kq] : 0.;
  for (int64_t base = wave * 64; base < n; base += n_waves * 64)
  {
#pragma unroll
    for (int g = 0; g < 4; ++g)
    {
      const int64_t cell = base + 16 * g + col;
      const bool ok = cell < n;
      const double b0 = ok ? u[(size_t)kq * n + cell] : 0.;
      const double b1 = ok ? u[(size_t)(4 + kq) * n + cell] : 0.;
      f64x4 acc = {0., 0., 0., 0.};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
      // D (f64 map): lane l holds rows (l >> 4) + 4 r of column l & 15: corners 0..7 are registers 0 and 1 of every lane
      if (ok)
      {
        const double cv = c[cell];
        v[(size_t)kq * n + cell] = cv * acc[0];
        v[(size_t)(4 + kq) * n + cell] = cv * acc[1];
      }
    }
  }
}

template <typename T>
KRef<T> reference_matrix(double const h[3])
{
  // K_ref = sum over the Gauss points of the tables the host setup uses (amge_structured.cpp)
  const auto Kq = reference_cell_tables(3, h);
  KRef<T> K;
  for (int m = 0; m < 8; ++m)
    for (int k = 0; k < 8; ++k)
    {
      double s = 0.;
      for (int q = 0; q < 8; ++q)
        s += Kq[((size_t)q * 8 + m) * 8 + k];
      K.k[m][k] = T(s);
    }
  return K;
}
} // namespace

template <typename T>
void cell_contraction(HipHandle &handle, int variant, int64_t n_cells, T const *u, T const *c, T *v, double const h[3])
{
  ASSERT_THROW(n_cells >= 0 && u && c && v, "bad argument");
  ASSERT_THROW(variant == 0 || variant == 1, "variant must be 0 (vector ALU) or 1 (MFMA)");
  if (n_cells == 0)
    return;
  const KRef<T> K = reference_matrix<T>(h);
  const unsigned int blocks = n_blocks_for(n_cells, 256, 256 * 32);
  const double bytes = double(n_cells) * 17. * sizeof(T);
  hipEvent_t stop = handle.profiler.begin(variant ? "cell_contraction_mfma" : "cell_contraction_valu", bytes, handle.stream);
  if (variant == 0)
    hipLaunchKernelGGL(cell_contraction_valu_kernel<T>, dim3(blocks), dim3(256), 0, handle.stream, n_cells, u, c, v, K);
  else if constexpr (sizeof(T) == 4)
    hipLaunchKernelGGL(cell_contraction_mfma_f32_kernel, dim3(blocks), dim3(256), 0, handle.stream, n_cells, u, c, v, K);
  else
    hipLaunchKernelGGL(cell_contraction_mfma_f64_kernel, dim3(blocks), dim3(256), 0, handle.stream, n_cells, u, c, v, K);
  MFMG_HIP_CHECK(hipGetLastError());
  KernelProfiler::end(stop, handle.stream);
}

template void cell_contraction<float>(HipHandle &, int, int64_t, float const *, float const *, float *, double const[3]);
template void cell_contraction<double>(HipHandle &, int, int64_t, double const *, double const *, double *, double const[3]);
} // namespace mfmg
