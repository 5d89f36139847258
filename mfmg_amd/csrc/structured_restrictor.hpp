// Restrictor of the spectral AMGe on block agglomerates of a structured mesh, stored by agglomerate.
//
// The reference keeps R as a CSR matrix and applies it (and its explicit transpose) with cusparseDcsrmv
// (source/cuda/cuda_matrix_operator.cu:58-91, include/mfmg/cuda/sparse_matrix_device.templates.cuh:351-371).
// On block agglomerates every row of R lives on the (ax+1)(ay+1)(az+1) nodes of one agglomerate and all
// rows of an agglomerate share these columns, so the column indices and row pointers carry no information:
// the values are kept as planes E[m][row] (m = node position inside the agglomerate) and both R x and
// R^T y are evaluated from the same planes -- 8 B per entry instead of 12 (+4 per row), no transposed copy.
#pragma once

#include "amge_structured.hpp"
#include "common.hpp"

namespace mfmg
{
class StructuredRestrictorDevice
{
public:
  // Returns nullptr when R does not have the agglomerate structure (rows of one agglomerate contiguous, the
  // same count on every agglomerate, every entry on a node of its agglomerate, full agglomerates only).
  static std::shared_ptr<StructuredRestrictorDevice> create(HipHandle &handle, StructuredMesh const &mesh,
                                                           int const agglomerate[3], int const agg_dims[3],
                                                           std::vector<int32_t> const &row_agglomerate,
                                                           HostCsr const &R);

  int64_t n_coarse() const { return _n_coarse; }
  int64_t n_fine() const { return _n_fine; }
  int agglomerates(int d) const { return _na[d]; }
  int n_eigenvectors() const { return _n_eig; }
  // y_c = R x
  void restrict_to_coarse(double const *x, double *y) const;
  // out = R^T y (subtract = false) or out -= R^T y (subtract = true)
  void prolongate(double const *y, double *out, bool subtract) const;
  double algorithmic_bytes() const; // the CSR figure of SURVEY.md 8d for one application

private:
  StructuredRestrictorDevice(HipHandle &handle) : _handle(handle) {}
  HipHandle &_handle;
  int _N[3] = {1, 1, 1};   // nodes
  int _na[3] = {1, 1, 1};  // agglomerates
  int _a[3] = {1, 1, 1};   // cells per agglomerate
  int _n_eig = 1;
  int _patch = 1;
  int64_t _n_coarse = 0, _n_fine = 0, _nnz = 0;
  bool _identity_numbering = false;
  DeviceBuffer<double> _planes;   // [patch][n_coarse]
  DeviceBuffer<uint8_t> _exc;     // per agglomerate: 0 = its block equals the reference block `_table`
  DeviceBuffer<uint8_t> _exc_node; // per fine node: 0 = all agglomerates around it are regular
  DeviceBuffer<uint8_t> _blk_exc;  // per agglomerate position (na + 1 per direction): 0 = table-driven block kernel
  DeviceBuffer<int32_t> _exc_blocks; // the other positions
  DeviceBuffer<double> _table;    // [patch][n_eig]
  // the other agglomerates that repeat a block among themselves (at the same distance from the faces of the box):
  // class per agglomerate (0 = the reference block, kNoClass = a block of its own -> planes) and the class blocks
  static constexpr uint16_t kNoClass = 0xffff;
  DeviceBuffer<uint16_t> _cls;
  DeviceBuffer<double> _class_table; // [class][patch][n_eig]
  int _n_classes = 0;

public:
  int block_classes() const { return _n_classes; }

private:
  DeviceBuffer<int32_t> _node_dof; // DoF id of lexicographic node (empty when the numbering is lexicographic)
};
} // namespace mfmg
