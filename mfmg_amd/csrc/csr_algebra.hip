// Setup algebra of SparseMatrixDevice on the device: transpose and sparse matrix product.
//
// Reference: the transpose goes through the host (EpetraExt::RowMatrix_Transpose, source/cuda/cuda_matrix_operator.cu:93-130),
// the product through cusparseDcsrgemm (include/mfmg/cuda/sparse_matrix_device.templates.cuh:373-434).  Both are
// hand-written here and DETERMINISTIC: the result does not depend on the order in which wavefronts run --
//   transpose : column histogram, exclusive scan, scatter through per-column cursors (any order), then every output
//               row is sorted by its column index (= the input row index, unique): a wavefront bitonic sort in
//               registers for rows of <= 64 entries, a workgroup bitonic sort in LDS for longer ones;
//   product   : one workgroup per row of C = A B with a hash table in LDS (linear probing); the entries of the row of
//               A are visited in order with a barrier in between, within one step the workgroup walks one row of B,
//               whose columns are distinct, so every sum C_ik = sum_j A_ij B_jk is accumulated in the order of j --
//               the order of the host product (bitwise the same result); the row is then sorted by column.
// Rows longer than the LDS tables allow fall back to the host algorithms (sparse_matrix_device.hip).
#include "csr_algebra.hpp"

#include <algorithm>
#include <climits>

namespace mfmg
{
namespace
{
constexpr int kMaxSortRow = 4096;  // entries of one output row the workgroup sort holds in LDS
constexpr int kMaxHash = 4096;     // hash slots of one product row (load factor <= 1/2)

__global__ void count_columns_kernel(int64_t nnz, int32_t const *col, int32_t *count)
{
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&count[col[p]], 1);
}

template <typename T>
__global__ void scatter_transposed_kernel(int64_t n_rows, int32_t const *row_ptr, int32_t const *col, T const *val,
                                          int32_t const *t_row_ptr, int32_t *cursor, int32_t *t_col, T *t_val)
{
  // one wavefront per input row
  const int lane = threadIdx.x & 63;
  for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; r < n_rows; r += ((int64_t)gridDim.x * blockDim.x) >> 6)
    for (int p = row_ptr[r] + lane; p < row_ptr[r + 1]; p += 64)
    {
      const int c = col[p];
      const int q = t_row_ptr[c] + atomicAdd(&cursor[c], 1);
      t_col[q] = (int32_t)r;
      t_val[q] = val[p];
    }
}

// rows of <= 64 entries: one wavefront per row, bitonic network over the lanes
template <typename T>
__global__ void sort_short_rows_kernel(int64_t n_rows, int32_t const *row_ptr, int32_t *col, T *val)
{
  const int lane = threadIdx.x & 63;
  for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; r < n_rows; r += ((int64_t)gridDim.x * blockDim.x) >> 6)
  {
    const int begin = row_ptr[r], len = row_ptr[r + 1] - begin;
    if (len < 2 || len > 64)
      continue; // (uniform over the wavefront)
    int key = lane < len ? col[begin + lane] : INT_MAX;
    T v = lane < len ? val[begin + lane] : T(0);
    for (int k = 2; k <= 64; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1)
      {
        const int okey = __shfl_xor(key, j);
        const T ov = __shfl_xor(v, j);
        const bool up = (lane & k) == 0;      // ascending block
        const bool lower = (lane & j) == 0;   // this lane keeps the smaller of the pair in an ascending block
        const bool take = (key > okey) == (up == lower);
        if (take && key != okey)
        {
          key = okey;
          v = ov;
        }
      }
    if (lane < len)
    {
      col[begin + lane] = key;
      val[begin + lane] = v;
    }
  }
}

// rows of 65 .. kMaxSortRow entries, listed: one workgroup per row, bitonic sort in LDS
template <typename T>
__global__ __launch_bounds__(256) void sort_long_rows_kernel(int32_t const *rows, int32_t const *row_ptr, int32_t *col, T *val)
{
  extern __shared__ unsigned char smem[];
  int *keys = reinterpret_cast<int *>(smem);
  T *vals = reinterpret_cast<T *>(smem + kMaxSortRow * sizeof(int));
  const int r = rows[blockIdx.x];
  const int begin = row_ptr[r], len = row_ptr[r + 1] - begin;
  int n2 = 128;
  while (n2 < len)
    n2 <<= 1;
  for (int i = threadIdx.x; i < n2; i += blockDim.x)
  {
    keys[i] = i < len ? col[begin + i] : INT_MAX;
    vals[i] = i < len ? val[begin + i] : T(0);
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1)
    {
      for (int i = threadIdx.x; i < n2; i += blockDim.x)
      {
        const int partner = i ^ j;
        if (partner > i)
        {
          const bool up = (i & k) == 0;
          const int a = keys[i], b = keys[partner];
          if ((a > b) == up)
          {
            keys[i] = b;
            keys[partner] = a;
            const T t = vals[i];
            vals[i] = vals[partner];
            vals[partner] = t;
          }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < len; i += blockDim.x)
  {
    col[begin + i] = keys[i];
    val[begin + i] = vals[i];
  }
}

// ---- product ------------------------------------------------------------------------------------------------
__global__ void product_upper_bound_kernel(int64_t a_rows, int32_t const *a_ptr, int32_t const *a_col, int32_t const *b_ptr,
                                           int32_t *ub)
{
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < a_rows; r += (int64_t)gridDim.x * blockDim.x)
  {
    int64_t s = 0;
    for (int p = a_ptr[r]; p < a_ptr[r + 1]; ++p)
      s += b_ptr[a_col[p] + 1] - b_ptr[a_col[p]];
    ub[r] = (int32_t)min<int64_t>(s, INT_MAX);
  }
}

__device__ __forceinline__ unsigned int hash_of(int key, unsigned int mask) { return ((unsigned int)key * 2654435761u) & mask; }

// One workgroup per row of C.  NUMERIC = false: count the distinct columns (row lengths); true: accumulate the values,
// sort the row by column and write it at c_ptr[row].
template <typename T, bool NUMERIC>
__global__ __launch_bounds__(256) void product_row_kernel(int64_t a_rows, int32_t const *a_ptr, int32_t const *a_col, T const *a_val,
                                                           int32_t const *b_ptr, int32_t const *b_col, T const *b_val,
                                                           int32_t const *ub, int32_t *c_len, int32_t const *c_ptr, int32_t *c_col,
                                                           T *c_val)
{
#pragma clang fp contract(off) // multiply, then add: the rounding of the host product
  extern __shared__ unsigned char smem[];
  int *keys = reinterpret_cast<int *>(smem);
  T *vals = reinterpret_cast<T *>(smem + kMaxHash * sizeof(int));
  __shared__ int n_found;
  // (a grid of one workgroup per row would pass 2^32 threads at 17 M rows: the workgroups stride over the rows)
  for (int64_t r = blockIdx.x; r < a_rows; r += gridDim.x)
  {
    const int bound = ub[r];
    unsigned int size = 64;
    while (size < 2u * (unsigned int)bound && size < (unsigned int)kMaxHash)
      size <<= 1;
    const unsigned int mask = size - 1;
    for (unsigned int i = threadIdx.x; i < size; i += blockDim.x)
    {
      keys[i] = INT_MAX;
      if (NUMERIC)
        vals[i] = T(0);
    }
    if (threadIdx.x == 0)
      n_found = 0;
    __syncthreads();
    for (int p = a_ptr[r]; p < a_ptr[r + 1]; ++p)
    {
      const int k = a_col[p];
      const T av = NUMERIC ? a_val[p] : T(0);
      for (int q = b_ptr[k] + threadIdx.x; q < b_ptr[k + 1]; q += blockDim.x)
      {
        const int c = b_col[q];
        unsigned int h = hash_of(c, mask);
        for (;;)
        {
          const int old = atomicCAS(&keys[h], INT_MAX, c);
          if (old == INT_MAX || old == c)
            break;
          h = (h + 1) & mask;
        }
        if (NUMERIC)
          vals[h] += av * b_val[q]; // (the columns of one row of B are distinct: no two threads share a slot in this step)
      }
      __syncthreads(); // the next entry of the row of A accumulates after this one: sums in the order of the host product
    }
    if (!NUMERIC)
    {
      int mine = 0;
      for (unsigned int i = threadIdx.x; i < size; i += blockDim.x)
        mine += keys[i] != INT_MAX ? 1 : 0;
      atomicAdd(&n_found, mine);
      __syncthreads();
      if (threadIdx.x == 0)
        c_len[r] = n_found;
      __syncthreads(); // n_found and the table are reset for the next row only after everyone is here
      continue;
    }
    // sort the table by column (empty slots carry INT_MAX and end up behind the entries)
    for (unsigned int k2 = 2; k2 <= size; k2 <<= 1)
      for (unsigned int j = k2 >> 1; j > 0; j >>= 1)
      {
        for (unsigned int i = threadIdx.x; i < size; i += blockDim.x)
        {
          const unsigned int partner = i ^ j;
          if (partner > i)
          {
            const bool up = (i & k2) == 0;
            const int a = keys[i], b = keys[partner];
            if ((a > b) == up)
            {
              keys[i] = b;
              keys[partner] = a;
              const T t = vals[i];
              vals[i] = vals[partner];
              vals[partner] = t;
            }
          }
        }
        __syncthreads();
      }
    const int begin = c_ptr[r], len = c_ptr[r + 1] - begin;
    for (int i = threadIdx.x; i < len; i += blockDim.x)
    {
      c_col[begin + i] = keys[i];
      c_val[begin + i] = vals[i];
    }
    __syncthreads();
  }
}

template <typename K>
void set_lds_limit(K kernel, size_t bytes)
{
  MFMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}
} // namespace

template <typename T>
bool csr_transpose_device(HipHandle &h, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t const *row_ptr, int32_t const *col,
                          T const *val, DeviceBuffer<int32_t> &d_tptr, DeviceBuffer<int32_t> &d_tcol, DeviceBuffer<T> &d_tval)
{
  hipStream_t st = h.stream;
  std::vector<int32_t> t_row_ptr(n_cols + 1, 0);
  d_tptr.resize((size_t)n_cols + 1);
  d_tcol.resize((size_t)nnz);
  d_tval.resize((size_t)nnz);
  if (nnz == 0)
  {
    MFMG_HIP_CHECK(hipMemsetAsync(d_tptr.data(), 0, ((size_t)n_cols + 1) * sizeof(int32_t), st));
    return true;
  }
  MemoryKind kind("CSR arrays (val, col, row_ptr)");
  DeviceBuffer<int32_t> count((size_t)n_cols + 1), cursor((size_t)n_cols);
  MFMG_HIP_CHECK(hipMemsetAsync(count.data(), 0, ((size_t)n_cols + 1) * sizeof(int32_t), st));
  MFMG_HIP_CHECK(hipMemsetAsync(cursor.data(), 0, (size_t)n_cols * sizeof(int32_t), st));
  hipLaunchKernelGGL(count_columns_kernel, dim3(n_blocks_for(nnz, 256, 1 << 16)), dim3(256), 0, st, nnz, col, count.data());
  MFMG_HIP_CHECK(hipGetLastError());
  // exclusive scan of the histogram on the host (n_cols + 1 integers; the entries themselves stay on the device)
  std::vector<int32_t> cnt = count.download(st);
  int64_t run = 0;
  int max_len = 0;
  std::vector<int32_t> long_rows;
  for (int64_t c = 0; c < n_cols; ++c)
  {
    t_row_ptr[c] = (int32_t)run;
    run += cnt[c];
    max_len = std::max(max_len, cnt[c]);
    if (cnt[c] > 64)
      long_rows.push_back((int32_t)c);
  }
  t_row_ptr[n_cols] = (int32_t)run;
  ASSERT_THROW(run == nnz, "internal: column histogram does not add up");
  if (max_len > kMaxSortRow)
    return false; // a row of the transpose too long for the LDS sort: host path
  MFMG_HIP_CHECK(hipMemcpyAsync(d_tptr.data(), t_row_ptr.data(), ((size_t)n_cols + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL((scatter_transposed_kernel<T>), dim3(n_blocks_for(n_rows * 64, 256, 1 << 16)), dim3(256), 0, st, n_rows, row_ptr,
                     col, val, d_tptr.data(), cursor.data(), d_tcol.data(), d_tval.data());
  hipLaunchKernelGGL((sort_short_rows_kernel<T>), dim3(n_blocks_for(n_cols * 64, 256, 1 << 16)), dim3(256), 0, st, n_cols,
                     d_tptr.data(), d_tcol.data(), d_tval.data());
  MFMG_HIP_CHECK(hipGetLastError());
  if (!long_rows.empty())
  {
    DeviceBuffer<int32_t> d_rows;
    d_rows.upload(long_rows.data(), long_rows.size(), st);
    const size_t lds = (size_t)kMaxSortRow * (sizeof(int) + sizeof(T));
    set_lds_limit(sort_long_rows_kernel<T>, lds);
    hipLaunchKernelGGL((sort_long_rows_kernel<T>), dim3((unsigned int)long_rows.size()), dim3(256), lds, st, d_rows.data(), d_tptr.data(),
                       d_tcol.data(), d_tval.data());
    MFMG_HIP_CHECK(hipGetLastError());
  }
  MFMG_HIP_CHECK(hipStreamSynchronize(st)); // (t_row_ptr, the scratch buffers and long_rows go out of scope)
  return true;
}

template <typename T>
bool csr_transpose_device(HipHandle &h, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t const *row_ptr, int32_t const *col,
                          T const *val, std::vector<int32_t> &t_row_ptr, std::vector<int32_t> &t_col, std::vector<T> &t_val)
{
  DeviceBuffer<int32_t> d_tptr, d_tcol;
  DeviceBuffer<T> d_tval;
  t_row_ptr.assign(n_cols + 1, 0);
  t_col.resize(nnz);
  t_val.resize(nnz);
  if (nnz == 0)
    return true;
  if (!csr_transpose_device<T>(h, n_rows, n_cols, nnz, row_ptr, col, val, d_tptr, d_tcol, d_tval))
    return false;
  t_row_ptr = d_tptr.download(h.stream);
  t_col = d_tcol.download(h.stream);
  t_val = d_tval.download(h.stream);
  return true;
}

// The same for a B of few columns (the bottom levels of an aggregation hierarchy, whose rows are long and close to dense:
// 65 536 x 8192 with 2000 x 300 candidates per row): the table is addressed by the column itself -- no hashing, no sort --
// and compacted in column order.  Sums in the order of the host product, as above.
constexpr int kMaxDirect = 8192; // columns of B the direct table holds
template <typename T, bool NUMERIC>
__global__ __launch_bounds__(256) void product_row_direct_kernel(int64_t a_rows, int32_t const *a_ptr, int32_t const *a_col, T const *a_val,
                                                                  int32_t const *b_ptr, int32_t const *b_col, T const *b_val, int b_cols,
                                                                  int32_t *c_len, int32_t const *c_ptr, int32_t *c_col, T *c_val)
{
#pragma clang fp contract(off)
  extern __shared__ unsigned char smem[];
  unsigned char *hit = smem;                                   // [kMaxDirect]
  T *vals = reinterpret_cast<T *>(smem + kMaxDirect);          // [kMaxDirect]
  __shared__ int part[256];
  const int per = (b_cols + 255) / 256; // slots of one thread in the compaction: a contiguous run
  for (int64_t r = blockIdx.x; r < a_rows; r += gridDim.x)
  {
    for (int i = threadIdx.x; i < b_cols; i += blockDim.x)
    {
      hit[i] = 0;
      if (NUMERIC)
        vals[i] = T(0);
    }
    __syncthreads();
    for (int p = a_ptr[r]; p < a_ptr[r + 1]; ++p)
    {
      const int k = a_col[p];
      const T av = NUMERIC ? a_val[p] : T(0);
      for (int q = b_ptr[k] + threadIdx.x; q < b_ptr[k + 1]; q += blockDim.x)
      {
        const int c = b_col[q];
        hit[c] = 1;
        if (NUMERIC)
          vals[c] += av * b_val[q]; // (the columns of one row of B are distinct)
      }
      __syncthreads();
    }
    const int i0 = min((int)threadIdx.x * per, b_cols), i1 = min(i0 + per, b_cols);
    int mine = 0;
    for (int i = i0; i < i1; ++i)
      mine += hit[i];
    part[threadIdx.x] = mine;
    __syncthreads();
    if (!NUMERIC)
    {
      if (threadIdx.x == 0)
      {
        int total = 0;
        for (int t = 0; t < 256; ++t)
          total += part[t];
        c_len[r] = total;
      }
      __syncthreads();
      continue;
    }
    int before = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t)
      before += part[t];
    int o = c_ptr[r] + before;
    for (int i = i0; i < i1; ++i)
      if (hit[i])
      {
        c_col[o] = i;
        c_val[o] = vals[i];
        ++o;
      }
    __syncthreads();
  }
}

template <typename T>
bool csr_multiply_device(HipHandle &h, int64_t a_rows, int32_t const *a_ptr, int32_t const *a_col, T const *a_val, int32_t const *b_ptr,
                         int32_t const *b_col, T const *b_val, std::vector<int32_t> &c_ptr, std::vector<int32_t> &c_col,
                         std::vector<T> &c_val, int64_t b_cols)
{
  hipStream_t st = h.stream;
  c_ptr.assign(a_rows + 1, 0);
  c_col.clear();
  c_val.clear();
  if (a_rows == 0)
    return true;
  DeviceBuffer<int32_t> ub((size_t)a_rows), len((size_t)a_rows);
  hipLaunchKernelGGL(product_upper_bound_kernel, dim3(n_blocks_for(a_rows, 256, 1 << 16)), dim3(256), 0, st, a_rows, a_ptr, a_col, b_ptr,
                     ub.data());
  MFMG_HIP_CHECK(hipGetLastError());
  std::vector<int32_t> hub = ub.download(st);
  bool fits_hash = true;
  for (int32_t v : hub)
    if (2 * (int64_t)v > kMaxHash)
      fits_hash = false; // a row with more candidate columns than the LDS table holds at load factor 1/2
  const bool direct = !fits_hash && b_cols > 0 && b_cols <= kMaxDirect;
  if (!fits_hash && !direct)
    return false; // host path
  const size_t lds = direct ? (size_t)kMaxDirect * (1 + sizeof(T)) : (size_t)kMaxHash * (sizeof(int) + sizeof(T));
  const unsigned int product_grid = (unsigned int)std::min<int64_t>(a_rows, 1 << 20);
  if (direct)
  {
    set_lds_limit(product_row_direct_kernel<T, false>, lds);
    set_lds_limit(product_row_direct_kernel<T, true>, lds);
    hipLaunchKernelGGL((product_row_direct_kernel<T, false>), dim3(product_grid), dim3(256), lds, st, a_rows, a_ptr, a_col, a_val, b_ptr, b_col,
                       b_val, (int)b_cols, len.data(), (int32_t const *)nullptr, (int32_t *)nullptr, (T *)nullptr);
  }
  else
  {
    set_lds_limit(product_row_kernel<T, false>, lds);
    set_lds_limit(product_row_kernel<T, true>, lds);
    hipLaunchKernelGGL((product_row_kernel<T, false>), dim3(product_grid), dim3(256), lds, st, a_rows, a_ptr, a_col, a_val, b_ptr, b_col, b_val,
                       ub.data(), len.data(), (int32_t const *)nullptr, (int32_t *)nullptr, (T *)nullptr);
  }
  MFMG_HIP_CHECK(hipGetLastError());
  std::vector<int32_t> hlen = len.download(st);
  int64_t total = 0;
  for (int64_t r = 0; r < a_rows; ++r)
  {
    total += hlen[r];
    ASSERT_THROW(total < (int64_t(1) << 31), "SpGEMM result exceeds int32 nnz");
    c_ptr[r + 1] = (int32_t)total;
  }
  c_col.resize(total);
  c_val.resize(total);
  if (total == 0)
    return true;
  DeviceBuffer<int32_t> d_cptr, d_ccol((size_t)total);
  DeviceBuffer<T> d_cval((size_t)total);
  d_cptr.upload(c_ptr.data(), c_ptr.size(), st);
  if (direct)
    hipLaunchKernelGGL((product_row_direct_kernel<T, true>), dim3(product_grid), dim3(256), lds, st, a_rows, a_ptr, a_col, a_val, b_ptr, b_col,
                       b_val, (int)b_cols, (int32_t *)nullptr, d_cptr.data(), d_ccol.data(), d_cval.data());
  else
    hipLaunchKernelGGL((product_row_kernel<T, true>), dim3(product_grid), dim3(256), lds, st, a_rows, a_ptr, a_col, a_val, b_ptr, b_col, b_val,
                       ub.data(), (int32_t *)nullptr, d_cptr.data(), d_ccol.data(), d_cval.data());
  MFMG_HIP_CHECK(hipGetLastError());
  MFMG_HIP_CHECK(hipMemcpyAsync(c_col.data(), d_ccol.data(), (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MFMG_HIP_CHECK(hipMemcpyAsync(c_val.data(), d_cval.data(), (size_t)total * sizeof(T), hipMemcpyDeviceToHost, st));
  MFMG_HIP_CHECK(hipStreamSynchronize(st));
  return true;
}

template bool csr_transpose_device<double>(HipHandle &, int64_t, int64_t, int64_t, int32_t const *, int32_t const *, double const *,
                                           std::vector<int32_t> &, std::vector<int32_t> &, std::vector<double> &);
template bool csr_transpose_device<double>(HipHandle &, int64_t, int64_t, int64_t, int32_t const *, int32_t const *, double const *,
                                           DeviceBuffer<int32_t> &, DeviceBuffer<int32_t> &, DeviceBuffer<double> &);
template bool csr_transpose_device<float>(HipHandle &, int64_t, int64_t, int64_t, int32_t const *, int32_t const *, float const *,
                                          DeviceBuffer<int32_t> &, DeviceBuffer<int32_t> &, DeviceBuffer<float> &);
template bool csr_transpose_device<float>(HipHandle &, int64_t, int64_t, int64_t, int32_t const *, int32_t const *, float const *,
                                          std::vector<int32_t> &, std::vector<int32_t> &, std::vector<float> &);
template bool csr_multiply_device<double>(HipHandle &, int64_t, int32_t const *, int32_t const *, double const *, int32_t const *,
                                          int32_t const *, double const *, std::vector<int32_t> &, std::vector<int32_t> &,
                                          std::vector<double> &, int64_t);
template bool csr_multiply_device<float>(HipHandle &, int64_t, int32_t const *, int32_t const *, float const *, int32_t const *,
                                         int32_t const *, float const *, std::vector<int32_t> &, std::vector<int32_t> &,
                                         std::vector<float> &, int64_t);
} // namespace mfmg
