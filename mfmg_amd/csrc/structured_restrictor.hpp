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

#include <functional>

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

  // ---- b_c = R (A x - b) in ONE pass over x and b (residual_restriction.hip) ----------------------------------
  // For a fine operator that repeats its rows from agglomerate to agglomerate (constant coefficient: every
  // agglomerate at the same distance from the faces of the box sees the same 5 x 5 x 5 rows of R A), the product R A is
  // a table of 125 x 2 weights per agglomerate class; `hierarchy.hpp:284-290` (residual, then restriction) becomes one
  // kernel that reads x and b once and never stores the fine residual.  The tables are PROBED (A applied to R^T e for
  // one representative of every class: apply_a(v, w) must compute w = A v without touching other ranks); the caller
  // verifies the result against the two-step path before use (HipMatrixOperator::prepare_residual_restriction).
  // Returns false -- nothing built -- when the blocks of R fall into more than 4096 classes (no two alike), for
  // agglomerates other than 2 x 2 x 2 cells with two eigenvectors, or for a renumbered mesh.
  // Distributed runs (slabs along z, or boxes: the _xy members, [0] = x, [1] = y): rows of A are computed locally only on the
  // node layers [valid_begin, valid_end), and a layer outside the local mesh is outside the BOX only where no neighbour exists;
  // a class with an agglomerate in the layers [owned_begin, owned_end) of agglomerates needs a representative whose 5 layers
  // qualify (along every axis).
  struct SlabInfo
  {
    int valid_begin = 0, valid_end = 1 << 30; // node layers whose rows apply_a computes
    bool has_low = false, has_high = false;   // neighbours below / above
    int owned_begin = 0, owned_end = 1 << 30; // agglomerate layers this rank owns
    int valid_begin_xy[2] = {0, 0}, valid_end_xy[2] = {1 << 30, 1 << 30};
    bool has_low_xy[2] = {false, false}, has_high_xy[2] = {false, false};
    int owned_begin_xy[2] = {0, 0}, owned_end_xy[2] = {1 << 30, 1 << 30};
  };
  bool build_residual_restriction(std::function<void(double const *, double *)> const &apply_a, SlabInfo const &slab);
  bool build_residual_restriction(std::function<void(double const *, double *)> const &apply_a)
  {
    return build_residual_restriction(apply_a, SlabInfo());
  }
  bool has_residual_restriction() const { return _rr_table.size() > 0; }
  void drop_residual_restriction();
  int residual_restriction_classes() const { return _rr_classes; }
  void restrict_residual(double const *x, double const *b, double *y) const;
  // the same from FP32 vectors (the FP32 fine level of apply_f32): sums and result in FP64
  void restrict_residual(float const *x, float const *b, double *y) const;

private:
  template <typename TI>
  void restrict_residual_any(TI const *x, TI const *b, double *y) const;
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
  DeviceBuffer<float> _planes_f32; // ... or, when every value is representable in float, these instead
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
  bool float_planes() const { return _planes_f32.size() > 0; }

private:
  DeviceBuffer<int32_t> _node_dof; // DoF id of lexicographic node (empty when the numbering is lexicographic)
  // residual restriction
  std::vector<uint16_t> _cls_host;       // class of every agglomerate by the bits of its block (empty: > 4096 classes)
  DeviceBuffer<double> _rr_table;        // [class][125 + 27][2]: weights of R A on the 5^3 nodes around, of R on its 3^3
  DeviceBuffer<uint16_t> _rr_cls;        // per agglomerate
  DeviceBuffer<uint16_t> _rr_seg_class;  // per wavefront of the main part (64 agglomerates of one row): its class
  DeviceBuffer<int32_t> _rr_listed;      // agglomerates left to the thread-per-agglomerate part
  int _rr_segs = 0, _rr_classes = 0, _rr_main_last = 0;
};
} // namespace mfmg
