// CSR matrix resident in HBM with hand-written gfx950 SpMV kernels.
//
// Twin of SparseMatrixDevice<ScalarType> (include/mfmg/cuda/sparse_matrix_device.cuh:28-104):
// raw val/column_index/row_ptr device arrays owned by the object, 0-based, general.
// What is NOT reproduced: the all-gather of the whole source vector + reorder_vector
// kernel in front of every cusparseDcsrmv (…templates.cuh:104-138,351-371); column
// indices here are local (owned + ghost) ids and ghosts arrive by halo exchange.
#pragma once

#include "common.hpp"

namespace mfmg
{
enum class CsrMode : int
{
  apply = 0,    // out = A x
  residual = 1, // out = A x - b
  first = 2,    // out = x - beta dinv (A x - b)
  next = 3,     // out = x + alpha (x - x_prev) - beta dinv (A x - b)
  subtract = 4, // out -= A x            (x.add(-1, R^T x_c), hierarchy.hpp:297-302)
  add = 5,      // out += A x
  plus_scaled = 6 // out = A x + beta dinv b   (prolongation and damped-Jacobi post-smoothing of a correction in one step)
};

template <typename T>
class SparseMatrixDevice
{
public:
  using value_type = T;

  // convert_matrix (source/cuda/utils.cu:39-168): host CSR -> device
  SparseMatrixDevice(HipHandle &handle, int64_t n_rows, int64_t n_cols, std::vector<int32_t> row_ptr,
                     std::vector<int32_t> col, std::vector<T> val, bool keep_host = true, bool analyse = true);
  // (analyse = false: plain CSR only -- for a matrix kept for the setup algebra and inspection while another object
  // evaluates it, like the restrictor next to its agglomerate-wise form)

  // from arrays already on the device (ownership taken; the host copy is fetched on demand)
  SparseMatrixDevice(HipHandle &handle, int64_t n_rows, int64_t n_cols, DeviceBuffer<int32_t> row_ptr, DeviceBuffer<int32_t> col,
                     DeviceBuffer<T> val, bool analyse = true);

  int64_t m() const { return _n_rows; }
  int64_t n() const { return _n_cols; }
  int64_t n_nonzero_elements() const { return _nnz; }
  int64_t n_local_rows() const { return _n_rows; }

  // dst = A src   (SparseMatrixDevice::vmult, …templates.cuh:351-371)
  void vmult(T *dst, T const *src) const
  {
    launch(CsrMode::apply, src, nullptr, nullptr, nullptr, T(0), T(0), dst);
  }
  void residual(T const *x, T const *b, T *res) const
  {
    launch(CsrMode::residual, x, b, nullptr, nullptr, T(0), T(0), res);
  }
  void smoother_step(T const *dinv, T const *b, T const *x, T const *x_prev, T alpha, T beta, T *out) const
  {
    if (x_prev == nullptr || alpha == T(0))
      launch(CsrMode::first, x, b, dinv, nullptr, T(0), beta, out);
    else
      launch(CsrMode::next, x, b, dinv, x_prev, alpha, beta, out);
  }
  void vmult_subtract(T *inout, T const *src) const
  {
    launch(CsrMode::subtract, src, nullptr, nullptr, nullptr, T(0), T(0), inout);
  }
  void vmult_add(T *inout, T const *src) const
  {
    launch(CsrMode::add, src, nullptr, nullptr, nullptr, T(0), T(0), inout);
  }
  // dst = A src + beta dinv b (dinv, b: vectors of m() entries)
  void vmult_plus_scaled(T *dst, T const *src, T const *dinv, T const *b, T beta) const
  {
    launch(CsrMode::plus_scaled, src, b, dinv, nullptr, T(0), beta, dst);
  }
  // extract_inv_diag (source/cuda/cuda_smoother.cu:86-96)
  void inverse_diagonal(T *dinv) const;
  // dinv = 1 / diagonal and ratio = (sum of |entries| of the row) / |diagonal| per row (device arrays of m() entries)
  void row_ratios(T *dinv, T *ratio) const;

  // setup-time algebra on the host copy (the reference does the transpose on the
  // host through EpetraExt, cuda_matrix_operator.cu:93-130, and SpGEMM with
  // cusparseDcsrgemm, …templates.cuh:373-434)
  std::shared_ptr<SparseMatrixDevice<T>> transpose() const;
  std::shared_ptr<SparseMatrixDevice<T>> mmult(SparseMatrixDevice<T> const &b) const;

  std::vector<int32_t> const &host_row_ptr() const
  {
    ensure_host_copy();
    return _row_ptr_host;
  }
  std::vector<int32_t> const &host_col() const
  {
    ensure_host_copy();
    return _col_host;
  }
  std::vector<T> const &host_val() const
  {
    ensure_host_copy();
    return _val_host;
  }
  bool has_host_copy() const { return !_row_ptr_host.empty(); }
  void download(std::vector<int32_t> &row_ptr, std::vector<int32_t> &col, std::vector<T> &val) const;

  T const *val_dev() const
  {
    ensure_device_csr();
    return _val.data();
  }
  int32_t const *column_index_dev() const
  {
    ensure_device_csr();
    return _col.data();
  }
  int32_t const *row_ptr_dev() const
  {
    ensure_device_csr();
    return _row_ptr.data();
  }
  // Free the CSR arrays (device and host copies) of a matrix whose kernels read tables -- node classes, or block diagonals with
  // regular rows and stencil classes --, keeping the few rows the tables do not cover as a compact CSR of their own
  // ("release setup matrices": A_c and the operators of the aggregation levels hold 12 B per entry that only the setup
  // algebra and the exports read).  False (nothing done) for a matrix whose kernels read the CSR arrays themselves.
  // Afterwards download / transpose / mmult / the diagonal kernels and a change of kernel throw.
  bool release_csr();
  bool csr_released() const { return _csr_released; }
  // the CSR arrays on the device (a no-op unless the upload was deferred)
  void ensure_device_csr() const;
  bool device_csr_deferred() const { return _device_csr_deferred; }
  HipHandle &handle() const { return _handle; }

  bool uses_lds_path() const { return _use_lds; }
  int lanes_per_row() const { return _lanes_per_row; }
  // tuning knob: lanes of a wavefront that share a row (power of two, 1..64; 0 keeps the choice) and whether
  // the LDS-cached kernel is used (only where the block-local column lists were built)
  // use_lds: 0 plain, 1 LDS-cached, 2 / 3 block-diagonal storage (3 is reported when only the upper half of a
  // symmetric matrix is stored), 4 row-base storage, 5 node classes; each variant only where its data was built
  void set_kernel(int lanes_per_row, int use_lds)
  {
    ASSERT_THROW(!_csr_released || use_lds < 0, "the CSR arrays of this matrix were released after the setup: its kernel is fixed");
    if (lanes_per_row > 0)
      _lanes_per_row = lanes_per_row;
    if (use_lds >= 0)
    {
      _use_bdia = (use_lds == 2 || use_lds == 3) && (_bdia_val.size() > 0 || _bdia_val_f32.size() > 0);
      _use_rowbase = (use_lds == 4) && _rb_val.size() > 0;
      _use_nodecls = (use_lds == 5) && _nc_nodes.size() > 0;
      _use_lds = (use_lds == 1) && _lcol.size() > 0;
    }
  }
  int kernel_kind() const
  {
    return _use_nodecls ? 5 : _use_rowbase ? 4 : _use_bdia ? (_bdia_sym ? 3 : 2) : (_use_lds ? 1 : 0);
  }
  int block_diagonals() const { return _use_bdia ? _bdia_d : 0; }
  bool symmetric_storage() const { return _use_bdia && _bdia_sym; }
  bool float_storage() const { return _use_bdia && _bdia_val_f32.size() > 0; }
  bool regular_rows() const
  {
    return _use_regular && ((_use_bdia && _bdia_regular) || _use_nodecls);
  }
  void set_regular_rows(bool on)
  {
    ASSERT_THROW(!_csr_released || on, "the CSR arrays of this matrix were released after the setup: its kernel is fixed");
    _use_regular = on;
  }
  // rows evaluated from stored values although the matrix has regular rows, and the stencil classes next to the regular one
  int64_t listed_rows() const { return (int64_t)(_use_nodecls ? _nc_listed.size() : _bdia_exc_rows.size()); }
  int stencil_classes() const { return _use_nodecls ? _nc_classes : _bdia_n_classes; }
  // algorithmic bytes of one y = A x (12 B/nnz + 4 B/row ptr + x + y)
  double algorithmic_bytes_apply() const
  {
    return double(_nnz) * (sizeof(T) + 4) + 4. * double(_n_rows + 1) + sizeof(T) * double(_n_cols) +
           sizeof(T) * double(_n_rows);
  }

private:
  void launch(CsrMode mode, T const *x, T const *b, T const *dinv, T const *x_prev, T alpha, T beta,
              T *out) const;

  HipHandle &_handle;
  int64_t _n_rows, _n_cols, _nnz;
  int _lanes_per_row;
  // LDS-cached variant (chosen at construction when the columns of a block of rows are few): per
  // block of kRowsPerBlock rows the sorted unique columns (`l2g`) are gathered once into LDS and the
  // matrix stream carries 16-bit block-local column ids instead of 32-bit global ones
  static constexpr int kRowsPerBlock = 128;
  bool _use_lds = false;
  int _lds_max_cols = 0;
  DeviceBuffer<int32_t> _blk_ptr, _l2g;
  DeviceBuffer<uint16_t> _lcol;
  // block-diagonal storage (chosen at construction for stencil-like square matrices, see the .hip file)
  void build_block_diagonals();
  void choose_layouts(bool analyse);
  void ensure_host_copy() const;
  void sample_rows(std::vector<int64_t> const &rows, std::vector<int32_t> &ptr, std::vector<int32_t> &cols) const;
  // row-base storage (rectangular stencil-like matrices: the smoothed prolongators), see the .hip file
  void build_row_base(std::vector<int32_t> const &row_ptr, std::vector<int32_t> const &col, std::vector<T> const &val);
  bool _use_rowbase = false;
  int _rb_slots = 0;
  DeviceBuffer<T> _rb_val;
  DeviceBuffer<int32_t> _rb_base, _rb_offs;
  // node classes (rectangular stencil-like matrices of a translation-invariant problem: the prolongators and
  // their transposes), see the .hip file
  void build_node_classes();
  bool _use_nodecls = false;
  int _nc_c = 0, _nc_d = 0, _nc_classes = 0;
  DeviceBuffer<int32_t> _nc_base, _nc_offs, _nc_nodes, _nc_class_of_wave, _nc_listed;
  DeviceBuffer<T> _nc_table;
  bool _use_bdia = false;
  bool _bdia_sym = false;
  // regular rows of a translation-invariant operator: stencil table instead of stored values (see the .hip file)
  bool _bdia_regular = false, _use_regular = true;
  int _bdia_full_d = 0;
  DeviceBuffer<T> _bdia_table;
  DeviceBuffer<uint8_t> _bdia_exc;
  DeviceBuffer<int32_t> _bdia_full_offs, _bdia_exc_rows; // symmetric matrix: only the block diagonals with offset >= 0 are stored
  // classes of non-regular nodes that repeat one stencil among themselves (the shells next to the boundary)
  DeviceBuffer<int32_t> _bdia_cls_nodes, _bdia_cls_of_wave;
  DeviceBuffer<T> _bdia_cls_table; // [n_classes][C][Df][C]
  int _bdia_n_classes = 0;
  bool _bdia_all_in_classes = false; // small matrices: the regular nodes are one of the classes (no launch of their own)
  int _bdia_c = 0, _bdia_d = 0;
  DeviceBuffer<T> _bdia_val;
  DeviceBuffer<float> _bdia_val_f32; // ... or these, when every value is representable in float (see the .hip file)
  DeviceBuffer<int32_t> _bdia_offs;
  // (mutable: a matrix built without layouts from host arrays it keeps -- the restrictor beside its agglomerate-wise form --
  // uploads its CSR arrays only when somebody asks for them: ensure_device_csr)
  mutable DeviceBuffer<T> _val;
  mutable DeviceBuffer<int32_t> _col;
  mutable DeviceBuffer<int32_t> _row_ptr;
  mutable bool _device_csr_deferred = false;
  bool _csr_released = false;
  DeviceBuffer<int32_t> _kept_ptr, _kept_col; // the listed rows after release_csr(), in the order of the list
  DeviceBuffer<T> _kept_val;
  mutable std::vector<int32_t> _row_ptr_host, _col_host;
  mutable std::vector<T> _val_host;
};

// host CSR helpers shared by the setup code
template <typename T>
void csr_transpose_host(int64_t n_rows, int64_t n_cols, std::vector<int32_t> const &row_ptr,
                        std::vector<int32_t> const &col, std::vector<T> const &val,
                        std::vector<int32_t> &t_row_ptr, std::vector<int32_t> &t_col, std::vector<T> &t_val);
template <typename T>
void csr_multiply_host(int64_t a_rows, int64_t a_cols, std::vector<int32_t> const &a_ptr,
                       std::vector<int32_t> const &a_col, std::vector<T> const &a_val, int64_t b_cols,
                       std::vector<int32_t> const &b_ptr, std::vector<int32_t> const &b_col,
                       std::vector<T> const &b_val, std::vector<int32_t> &c_ptr, std::vector<int32_t> &c_col,
                       std::vector<T> &c_val);
} // namespace mfmg
