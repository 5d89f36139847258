// Device transpose and product of CSR matrices (csr_algebra.hip).  Inputs are DEVICE arrays, the results come back as
// host vectors (the SparseMatrixDevice constructor analyses the pattern on the host to choose its storage format).
// Both return false -- without having produced anything -- when a row exceeds what the LDS tables hold; the caller
// then takes the host algorithm.  (Product: a hash table of 4096 slots per row, or, when the number of columns of B is
// given and at most 8192, a table addressed by the column.)
#pragma once
#include "common.hpp"

#include <vector>

namespace mfmg
{
template <typename T>
bool csr_transpose_device(HipHandle &h, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t const *row_ptr, int32_t const *col,
                          T const *val, std::vector<int32_t> &t_row_ptr, std::vector<int32_t> &t_col, std::vector<T> &t_val);

// the same with the result left on the device
template <typename T>
bool csr_transpose_device(HipHandle &h, int64_t n_rows, int64_t n_cols, int64_t nnz, int32_t const *row_ptr, int32_t const *col,
                          T const *val, DeviceBuffer<int32_t> &t_row_ptr, DeviceBuffer<int32_t> &t_col, DeviceBuffer<T> &t_val);

template <typename T>
bool csr_multiply_device(HipHandle &h, int64_t a_rows, int32_t const *a_ptr, int32_t const *a_col, T const *a_val, int32_t const *b_ptr,
                         int32_t const *b_col, T const *b_val, std::vector<int32_t> &c_ptr, std::vector<int32_t> &c_col,
                         std::vector<T> &c_val, int64_t b_cols = -1);
} // namespace mfmg
