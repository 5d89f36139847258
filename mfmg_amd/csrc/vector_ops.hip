// Vector kernels (16-byte vectorised where alignment allows is left to the
// compiler: these are grid-stride streaming loops) and deterministic reductions:
// per-lane grid-stride partial -> wave __shfl_xor butterfly -> LDS across the 4
// waves of the block -> one partial per block -> a second single-block pass in a
// fixed order.  No atomics: the result does not depend on scheduling.
#include "vector_ops.hpp"

namespace mfmg
{
namespace vec
{
namespace
{
constexpr int kReduceBlocks = 1024;

template <typename T>
__global__ void set_kernel(int64_t n, T value, T *x)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = value;
}

template <typename T>
__global__ void copy_kernel(int64_t n, T const *src, T *dst)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

template <typename T>
__global__ void sadd_kernel(int64_t n, T s, T a, T const *v, T *x, int plain_add)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = plain_add ? (x[i] + a * v[i]) : (s * x[i] + a * v[i]);
}

template <typename T>
__global__ void scale_pointwise_kernel(int64_t n, T const *d, T const *v, T *out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = d[i] * v[i];
}

template <typename T>
__global__ void scaled_pointwise_kernel(int64_t n, T s, T const *d, T const *v, T *out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (s * d[i]) * v[i];
}

__global__ void probing_vector_kernel(int nx, int ny, int nz, int n_eig, int kx, int ky, int kz, int ox, int oy, int oz,
                                      int e0, double *u, int x_offset, int y_offset, int z_offset)
{
  const int64_t n = (int64_t)nx * ny * nz * n_eig;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int e = (int)(i % n_eig);
    const int64_t ag = i / n_eig;
    const int ax = (int)(ag % nx), ay = (int)((ag / nx) % ny), az = (int)(ag / ((int64_t)nx * ny));
    u[i] = (e == e0 && (ax + x_offset) % kx == ox && (ay + y_offset) % ky == oy && (az + z_offset) % kz == oz) ? 1. : 0.;
  }
}

// out[r] = in[r] (or 1 when `in` is null) where row r = node * n_comp + comp lives on a node whose block
// coordinates (GLOBAL node coordinates / block: local + offset) are congruent to
// `phase` modulo `period` and comp == comp0; 0 elsewhere.  Probing vectors of the aggregation-hierarchy setup.
__global__ void select_rows_kernel(int nx, int ny, int nz, int n_comp, int block, int x_offset, int y_offset, int z_offset, int px, int py, int pz,
                                   int ox, int oy, int oz, int comp0, double const *in, double *out)
{
  const int64_t n = (int64_t)nx * ny * nz * n_comp;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
  {
    const int c = (int)(r % n_comp);
    const int64_t nd = r / n_comp;
    const int i = (int)(nd % nx), j = (int)((nd / nx) % ny), k = (int)(nd / ((int64_t)nx * ny));
    const bool hit = c == comp0 && ((i + x_offset) / block) % px == ox && ((j + y_offset) / block) % py == oy && ((k + z_offset) / block) % pz == oz;
    out[r] = hit ? (in ? in[r] : 1.) : 0.;
  }
}

__global__ void add_layers_kernel(int64_t n, double const *src, double *dst)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}

// the regions of a box exchange (sub-boxes of the lexicographic array of nodes nx x ny x ., comps entries per node) against
// the packed buffer: mode 0 buf = v, 1 v = buf, 2 v += buf
__global__ void regions_copy_kernel(double *v, int comps, int64_t nx, int64_t ny, HaloRegions t, double *buf, int mode)
{
  const int64_t n = t.off[t.count];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    int r = 0;
    while (r + 1 < t.count && i >= t.off[r + 1])
      ++r;
    const int64_t q = i - t.off[r], rx = (int64_t)t.n[r][0] * comps;
    const int64_t x = q % rx, j = (q / rx) % t.n[r][1], k = q / (rx * t.n[r][1]);
    double *p = v + ((k + t.b[r][2]) * ny + (j + t.b[r][1])) * nx * comps + (int64_t)t.b[r][0] * comps + x;
    if (mode == 0)
      buf[i] = *p;
    else if (mode == 1)
      *p = buf[i];
    else
    {
      // the regions that receive sums overlap (an owned corner node belongs to three faces, three edges and the corner): the
      // thread of the FIRST region that holds the node adds the contributions of all of them, in the order of the regions
      const int c[3] = {(int)(x / comps) + t.b[r][0], (int)j + t.b[r][1], (int)k + t.b[r][2]};
      const int e = (int)(x % comps);
      auto inside = [&](int q2) {
        return c[0] >= t.b[q2][0] && c[0] < t.b[q2][0] + t.n[q2][0] && c[1] >= t.b[q2][1] && c[1] < t.b[q2][1] + t.n[q2][1] &&
               c[2] >= t.b[q2][2] && c[2] < t.b[q2][2] + t.n[q2][2];
      };
      bool first = true;
      for (int q2 = 0; q2 < r && first; ++q2)
        first = !inside(q2);
      if (!first)
        continue;
      double sum = *p;
      for (int q2 = r; q2 < t.count; ++q2)
        if (inside(q2))
          sum += buf[t.off[q2] + (((int64_t)(c[2] - t.b[q2][2]) * t.n[q2][1] + (c[1] - t.b[q2][1])) * t.n[q2][0] + (c[0] - t.b[q2][0])) * comps + e];
      *p = sum;
    }
  }
}

// the sub-box [b0, b0 + bn) of the lexicographic array of nodes `dims` (comps entries per node) against the contiguous buf
// (mode 0: buf = v, 1: v = buf)
__global__ void box_copy_kernel(double *v, int comps, int64_t nx, int64_t ny, int64_t bx0, int64_t by0, int64_t bz0, int64_t bnx,
                                int64_t bny, int64_t bnz, double *buf, int mode)
{
  const int64_t n = bnx * bny * bnz * comps, rx = bnx * comps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    const int64_t x = i % rx, j = (i / rx) % bny, k = i / (rx * bny);
    double *p = v + ((k + bz0) * ny + (j + by0)) * nx * comps + bx0 * comps + x;
    if (mode == 0)
      buf[i] = *p;
    else
      *p = buf[i];
  }
}

// out[i] = in[index[i]]
__global__ void gather_indexed_kernel(int64_t n, double const *in, int32_t const *index, double *out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[index[i]];
}

__global__ void widen_kernel(int64_t n, float const *in, double *out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (double)in[i];
}

__global__ void subtract_narrowed_kernel(int64_t n, double const *c, float *x)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = (float)((double)x[i] - c[i]);
}

__device__ __forceinline__ double block_reduce_sum(double v)
{
  __shared__ double wsum[16];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (lane == 0)
    wsum[wave] = v;
  __syncthreads();
  double total = 0.;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w)
      total += wsum[w];
  return total; // valid in thread 0
}

template <typename T>
__global__ void dot_stage1_kernel(int64_t n, T const *x, T const *y, double *partials)
{
  double acc = 0.;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc += (double)x[i] * (double)y[i];
  const double t = block_reduce_sum(acc);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = t;
}

__global__ void dot_stage2_kernel(int n_partials, double const *partials, double *result, int slot)
{
  double acc = 0.;
  for (int i = threadIdx.x; i < n_partials; i += blockDim.x)
    acc += partials[i];
  const double t = block_reduce_sum(acc);
  if (threadIdx.x == 0)
    result[slot] = t;
}

template <typename T>
__global__ void cg_update_kernel(int64_t n, T const *p, T const *Ap, T const *dinv, T *x, T *r, T *z,
                                 double const *scal, int slot_rz, int slot_pap)
{
  const double pap = scal[slot_pap];
  const T alpha = (pap != 0.) ? T(scal[slot_rz] / pap) : T(0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
  {
    x[i] += alpha * p[i];
    const T ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    if (dinv != nullptr) // (Jacobi-preconditioned CG of the coarse solver; the outer CG driver applies its preconditioner itself)
      z[i] = dinv[i] * ri;
  }
}

template <typename T>
__global__ void cg_direction_kernel(int64_t n, T const *z, T *p, double const *scal, int slot_new, int slot_old)
{
  const double old = scal[slot_old];
  const T beta = (old != 0.) ? T(scal[slot_new] / old) : T(0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = z[i] + beta * p[i];
}

inline unsigned int stream_blocks(int64_t n) { return n_blocks_for(n, block_size, 256 * 16); }
} // namespace

template <typename T>
void set(HipHandle &h, int64_t n, T value, T *x)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(set_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, value, x);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void copy(HipHandle &h, int64_t n, T const *src, T *dst)
{
  if (n <= 0 || src == dst)
    return;
  hipLaunchKernelGGL(copy_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, src, dst);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void add(HipHandle &h, int64_t n, T a, T const *v, T *x)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(sadd_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, T(1), a, v, x, 1);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void sadd(HipHandle &h, int64_t n, T s, T a, T const *v, T *x)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(sadd_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, s, a, v, x, 0);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void scale_pointwise(HipHandle &h, int64_t n, T const *d, T const *v, T *out)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(scale_pointwise_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, d, v,
                     out);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void scaled_pointwise(HipHandle &h, int64_t n, T s, T const *d, T const *v, T *out)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(scaled_pointwise_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, s, d,
                     v, out);
  MFMG_HIP_CHECK(hipGetLastError());
}

void probing_vector(HipHandle &h, int const na[3], int n_eig, int const k[3], int const o[3], int e0, double *u, int const offset[3])
{
  const int64_t n = (int64_t)na[0] * na[1] * na[2] * n_eig;
  if (n <= 0)
    return;
  hipLaunchKernelGGL(probing_vector_kernel, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, na[0], na[1], na[2],
                     n_eig, k[0], k[1], k[2], o[0], o[1], o[2], e0, u, offset[0], offset[1], offset[2]);
  MFMG_HIP_CHECK(hipGetLastError());
}

void select_rows(HipHandle &h, int const dims[3], int n_comp, int block, int const offset[3], int const period[3], int const phase[3],
                 int comp, double const *in, double *out)
{
  const int64_t n = (int64_t)dims[0] * dims[1] * dims[2] * n_comp;
  if (n <= 0)
    return;
  hipLaunchKernelGGL(select_rows_kernel, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, dims[0], dims[1], dims[2], n_comp,
                     block, offset[0], offset[1], offset[2], period[0], period[1], period[2], phase[0], phase[1], phase[2], comp, in, out);
  MFMG_HIP_CHECK(hipGetLastError());
}

void widen(HipHandle &h, int64_t n, float const *in, double *out)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(widen_kernel, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, in, out);
  MFMG_HIP_CHECK(hipGetLastError());
}

void subtract_narrowed(HipHandle &h, int64_t n, double const *correction, float *x)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(subtract_narrowed_kernel, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, correction, x);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void dot_async(HipHandle &h, int64_t n, T const *x, T const *y, double *result_dev, int slot)
{
  const unsigned int nb = n_blocks_for(n, block_size, kReduceBlocks);
  hipLaunchKernelGGL(dot_stage1_kernel<T>, dim3(nb), dim3(block_size), 0, h.stream, n, x, y,
                     h.reduce_partials.data());
  hipLaunchKernelGGL(dot_stage2_kernel, dim3(1), dim3(block_size), 0, h.stream, (int)nb,
                     h.reduce_partials.data(), result_dev, slot);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
double dot(HipHandle &h, int64_t n, T const *x, T const *y)
{
  dot_async<T>(h, n, x, y, h.reduce_result.data(), 0);
  MFMG_HIP_CHECK(hipMemcpyAsync(h.host_result, h.reduce_result.data(), sizeof(double), hipMemcpyDeviceToHost,
                                h.stream));
  MFMG_HIP_CHECK(hipStreamSynchronize(h.stream));
  return h.host_result[0];
}

template <typename T>
double l2_norm(HipHandle &h, int64_t n, T const *x)
{
  return std::sqrt(dot<T>(h, n, x, x));
}

template <typename T>
void cg_update(HipHandle &h, int64_t n, T const *p, T const *Ap, T const *dinv, T *x, T *r, T *z,
               double const *scal, int slot_rz, int slot_pap)
{
  hipLaunchKernelGGL(cg_update_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, p, Ap, dinv,
                     x, r, z, scal, slot_rz, slot_pap);
  MFMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void cg_direction(HipHandle &h, int64_t n, T const *z, T *p, double const *scal, int slot_rz_new,
                  int slot_rz_old)
{
  hipLaunchKernelGGL(cg_direction_kernel<T>, dim3(stream_blocks(n)), dim3(block_size), 0, h.stream, n, z, p,
                     scal, slot_rz_new, slot_rz_old);
  MFMG_HIP_CHECK(hipGetLastError());
}

#define MFMG_INSTANTIATE_VEC(T)                                                                              \
  template void set<T>(HipHandle &, int64_t, T, T *);                                                        \
  template void copy<T>(HipHandle &, int64_t, T const *, T *);                                               \
  template void add<T>(HipHandle &, int64_t, T, T const *, T *);                                             \
  template void sadd<T>(HipHandle &, int64_t, T, T, T const *, T *);                                         \
  template void scale_pointwise<T>(HipHandle &, int64_t, T const *, T const *, T *);                         \
  template void scaled_pointwise<T>(HipHandle &, int64_t, T, T const *, T const *, T *);                     \
  template void dot_async<T>(HipHandle &, int64_t, T const *, T const *, double *, int);                     \
  template double dot<T>(HipHandle &, int64_t, T const *, T const *);                                        \
  template double l2_norm<T>(HipHandle &, int64_t, T const *);                                               \
  template void cg_update<T>(HipHandle &, int64_t, T const *, T const *, T const *, T *, T *, T *,           \
                             double const *, int, int);                                                      \
  template void cg_direction<T>(HipHandle &, int64_t, T const *, T *, double const *, int, int);
MFMG_INSTANTIATE_VEC(double)
MFMG_INSTANTIATE_VEC(float)
} // namespace vec

void halo_regions_copy(double *v, HaloSpace const &s, HaloRegions const &regions, double *buf, int mode, hipStream_t stream)
{
  s.check();
  const int64_t n = regions.count > 0 ? regions.off[regions.count] : 0;
  if (n <= 0)
    return;
  hipLaunchKernelGGL(vec::regions_copy_kernel, dim3(n_blocks_for(n, block_size, 4096)), dim3(block_size), 0, stream, v, s.comps, s.n_xy[0],
                     s.n_xy[1], regions, buf, mode);
  MFMG_HIP_CHECK(hipGetLastError());
}

void halo_box_copy(double *v, HaloSpace const &s, bool owned_only, double *buf, int mode, hipStream_t stream)
{
  s.check();
  const int64_t b0[3] = {owned_only ? s.own0(0) : 0, owned_only ? s.own0(1) : 0, owned_only ? s.own0(2) : 0};
  const int64_t bn[3] = {owned_only ? s.own_n(0) : s.dim(0), owned_only ? s.own_n(1) : s.dim(1), owned_only ? s.own_n(2) : s.dim(2)};
  const int64_t n = bn[0] * bn[1] * bn[2] * s.comps;
  if (n <= 0)
    return;
  hipLaunchKernelGGL(vec::box_copy_kernel, dim3(n_blocks_for(n, block_size, 4096)), dim3(block_size), 0, stream, v, s.comps, s.n_xy[0],
                     s.n_xy[1], b0[0], b0[1], b0[2], bn[0], bn[1], bn[2], buf, mode);
  MFMG_HIP_CHECK(hipGetLastError());
}

void gather_indexed(int64_t n, double const *in, int32_t const *index, double *out, hipStream_t stream)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(vec::gather_indexed_kernel, dim3(n_blocks_for(n, block_size, 4096)), dim3(block_size), 0, stream, n, in, index, out);
  MFMG_HIP_CHECK(hipGetLastError());
}

namespace vec
{
// keeps the stream busy for `ticks` of the 100 MHz wall clock (s_memrealtime): the price of a wire in the one-GPU harness
__global__ void stream_delay_kernel(long long ticks)
{
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks)
    __builtin_amdgcn_s_sleep(8);
}
} // namespace vec

void stream_delay(double microseconds, hipStream_t stream)
{
  if (!(microseconds > 0.))
    return;
  hipLaunchKernelGGL(vec::stream_delay_kernel, dim3(1), dim3(1), 0, stream, (long long)(microseconds * 100.));
  MFMG_HIP_CHECK(hipGetLastError());
}

void halo_add_layers(double *dst, double const *src, int64_t n, hipStream_t stream)
{
  if (n <= 0)
    return;
  hipLaunchKernelGGL(vec::add_layers_kernel, dim3(n_blocks_for(n, block_size, 4096)), dim3(block_size), 0, stream, n, src, dst);
  MFMG_HIP_CHECK(hipGetLastError());
}
} // namespace mfmg
