// Matrix-free Q1 Laplace operator on a logically structured hex mesh, device resident.
//
// Fills the hole of the reference's CUDA back-end: CudaMatrixFreeOperator::vmult
// only forwards to a user evaluator whose base class throws
// (source/cuda/cuda_matrix_free_operator.cu:32-37,
//  include/mfmg/cuda/cuda_matrix_free_mesh_evaluator.cuh:62-69); the arithmetic it
// has to reproduce is LaplaceOperator::local_apply / compute_diagonal of
// tests/laplace_matrix_free.hpp:75-98,129-156 with the constrained-row semantics of
// deal.II's MatrixFreeOperators::Base::vmult.
#pragma once

#include "common.hpp"

#include <memory>

namespace mfmg
{
// Fused epilogues of the operator kernel.
enum class MfMode : int
{
  apply = 0,    // out = A x
  residual = 1, // out = A x - b                       (hierarchy.hpp:284-286)
  first = 2,    // out = x - beta dinv (A x - b)       (Jacobi / first Chebyshev term)
  next = 3      // out = x + alpha (x - x_prev) - beta dinv (A x - b)
};

// A numbering the kernel can compute: id(i, j, k) = base + i s0 + j s1 + k s2 on the node grid (any lexicographic
// numbering; the rotated slab of the tail columns has s0 = row length, s1 = 1), Dirichlet flags on whole faces of the box
// (bit 0 / 1: i = 0 / Nx - 1, bits 2, 3: j, bits 4, 5: k) and ghost flags on whole planes along every axis: the ghost_lo[d] first
// and the ghost_hi[d] last node planes of axis d (the slabs and boxes of a distributed run; a node on a Dirichlet face carries
// the Dirichlet flag only, in a ghost plane too).  Verified slot by slot against the records at construction; the id loads -- a
// request per row whose answer the x requests wait for -- disappear from the kernel.
struct AffineIds
{
  int base, s0, s1, s2;
  int faces;
  int ghost_lo[3], ghost_hi[3];
};

template <typename T>
struct MfArgs; // kernel arguments (mf_laplace.hip)

template <typename T>
class MatrixFreeLaplaceDevice
{
public:
  // allow_compact: keep ONE coefficient per cell when the eight quadrature values of every cell are equal
  // (detected on the device); false forces the general eight-value layout
  // sub_mesh (internal, the tail slab): the cells are a sub-box of a larger numbering, n_dofs is the vector length
  MatrixFreeLaplaceDevice(HipHandle &handle, mfmg_hip_mesh_desc const &mesh, bool allow_compact = true,
                          bool sub_mesh = false);
  bool cell_constant_layout() const { return _compact; }
  bool diagonal_in_record() const { return _dinv_in_record; }

  int64_t n_dofs() const { return _n_dofs; }
  int dim() const { return _dim; }
  int dof_grid(int d) const { return _N[d]; }
  int n_cells(int d) const { return _n[d]; }
  double cell_size(int d) const { return _h[d]; }

  void vmult(T const *x, T *y) const { launch(MfMode::apply, x, nullptr, nullptr, T(0), T(0), y); }
  void residual(T const *x, T const *b, T *res) const
  {
    launch(MfMode::residual, x, b, nullptr, T(0), T(0), res);
  }
  // out = x + alpha (x - x_prev) - beta dinv (A x - b); out must not alias x
  void smoother_step(T const *b, T const *x, T const *x_prev, T alpha, T beta, T *out) const
  {
    if (x_prev == nullptr || alpha == T(0))
      launch(MfMode::first, x, b, nullptr, T(0), beta, out);
    else
      launch(MfMode::next, x, b, x_prev, alpha, beta, out);
  }

  // The same operations restricted to the z-tiles [z_tile_begin, z_tile_end) of the current tiling (every
  // DoF is owned by exactly one tile, so ranges that cover all tiles once give the full result): tiles
  // [0, n_z_tiles()) ; tile t owns the DoF layers [t tile_layers(), (t+1) tile_layers()) and reads the layers
  // one below and one above them.
  int n_z_tiles() const;
  int tile_layers() const;
  void launch_z_range(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out, int z_tile_begin,
                      int z_tile_end) const;
  // ... and to a box of tiles along all three axes (a box decomposition overlaps its exchange with the tiles that read
  // no ghost plane along any axis): n_tiles / rows per axis -- column tiles of `rows[0]` DoF columns, y-tiles of rows[1] DoF
  // rows, z-tiles of rows[2] layers; tile t of an axis owns [t rows, (t + 1) rows) and reads one plane below and above.
  // The columns of a nearly empty last chunk are not part of the column tiles: they are the `tail_part` (all y, the
  // z-tiles of the range), `main_part` the tiles [begin, end).
  void tiling(int n_tiles[3], int rows[3]) const;
  bool has_tail() const { return _tail != nullptr; }
  void launch_tiles(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out, int const begin[3],
                    int const end[3], bool main_part, bool tail_part) const;
  // ... and everything such a box leaves (the tiles outside [begin, end) and the tail columns) as ONE launch
  // (on_stream: another stream than the context's -- the shell of a distributed run is launched on the exchange stream, behind
  // the unpacking of the ghost planes, and runs BESIDE the interior tiles)
  void launch_outside(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out, int const begin[3],
                      int const end[3], hipStream_t on_stream = nullptr) const;

  // Several smoother terms in ONE sweep (mf_cheb_fused.hip): x_1 = x - beta[0] dinv (A x - b), then
  // x_s = x_{s-1} + alpha[s-1] (x_{s-1} - x_{s-2}) - beta[s-1] dinv (A x_{s-1} - b) up to s = n_terms (2 or 3); out = x_{n_terms},
  // out_prev (may be null) = x_{n_terms - 1}.  The same bits as n_terms calls of smoother_step.  alpha[0] must be zero; out,
  // out_prev and x must be three different vectors.  Available for the cell-constant layout with a numbering the kernel can
  // compute on one rank (fused_sweep_available); callers fall back to smoother_step otherwise.
  bool fused_sweep_available(int n_terms) const;
  // ... and from x_0 = 0 without reading it (smoother_sweep with x == nullptr: the pre-smoother of a preconditioner application)
  bool fused_zero_guess_available(int n_terms) const;
  // the last chunk column owns at most 32 - 2 halo DoF columns: the sweep kernels of the default tile shapes (three terms 8 x 3
  // rows, two terms 4 x 4) run it two y-tiles per workgroup, one per half of the wavefront
  bool narrow_last_column() const { return _narrow_last; }
  static bool fused_narrow_capable(int n_terms, int ty) { return (n_terms == 3 && ty == 3) || (n_terms == 2 && ty == 4); }
  void smoother_sweep(int n_terms, T const *alpha, T const *beta, T const *b, T const *x, T *out, T *out_prev) const;
  // tile of the sweep: nw wavefronts of ty cell rows, tz owned layers (0, 0, 0: chosen from the mesh)
  void set_fused_tile(int nw, int ty, int tz)
  {
    _fused_tile[0] = nw;
    _fused_tile[1] = ty;
    _fused_tile[2] = tz;
  }
  void get_fused_tile(int n_terms, int &nw, int &ty, int &tz) const { choose_fused_tile(n_terms, nw, ty, tz); }
  int halo_lanes() const { return _halo; }
  // the sweep's cell kernel: mode space (default: ~30 % fewer FP64 operations, its own rounding) or the arithmetic of the
  // one-term kernel (the sweep is then bit-identical to smoother_step after smoother_step)
  void set_fused_reference(bool on) { _fused_reference_arithmetic = on; }
  bool fused_reference() const { return _fused_reference_arithmetic; }
  // bytes one sweep of n_terms must move at least: x_0, b, one coefficient per cell (D^-1 where the records hold it), x_K
  double fused_sweep_bytes(bool with_prev) const { return double(_n_dofs) * sizeof(T) * (4. + (_dinv_in_record ? 1. : 0.) + (with_prev ? 1. : 0.)); }

  T const *diagonal() const { return _diag.data(); }
  T const *diagonal_inverse() const { return _dinv.data(); }

  // tile of one workgroup: n_waves wavefronts of ty cell rows each, tz layers (0 = chosen from the mesh size)
  void set_tile(int ty, int tz)
  {
    _tile_y = ty;
    _tile_z = tz;
  }
  void set_tile_waves(int nw) { _tile_waves = nw; }
  // the tile the next launch uses
  void get_tile(int &nw, int &ty, int &tz) const { choose_tile(nw, ty, tz); }
  HipHandle &handle() const { return _handle; }

  // Bytes of one operator application y = A x.
  // survey: the indexed form of SURVEY.md 8(d) -- x, y, 8 index ints, 8 coefficients (112 B/DoF in FP64; one
  //         coefficient per cell: 56);
  // required: what the chunk-record layout makes the kernel move at least -- x, y, ONE id (each slot stores its own
  //         DoF id, the other seven corners are neighbours' own ids) and the coefficients (84 / 28 B/DoF in FP64);
  //         halo re-reads of the tiling are not in this figure.
  double survey_bytes_apply() const
  {
    return double(_n_dofs) * (2.0 * sizeof(T) + 8 * 4 + (_compact ? 1 : 8) * sizeof(T));
  }
  // (ids: 4 bytes per DoF from the records; none where the kernel computes them, AffineIds)
  double required_bytes_apply() const
  {
    return double(_n_dofs) * (2.0 * sizeof(T) + (ids_computed() ? 0. : 4.) + (_compact ? 1 : 8) * sizeof(T));
  }
  // Used by the eight-coefficient kernels only (measured at 257^3 / 512^3 DoFs, Chebyshev(3) apply: 1.35 -> 1.27 ms and
  // 9.5 -> 9.0 ms; the one-coefficient kernels prefetch their ids a layer ahead and LOSE 3 % to the extra arithmetic).
  // Both parts of a launch (the mesh and the rotated slab of its tail columns) must have such a numbering.
  bool ids_computed() const { return !_compact && _affine_ids && (!_tail || _tail->_affine_ids); }
  // bytes of the epilogue operands of a fused mode on top of that: b, then D^-1 (NOT read in the cell-constant layout:
  // the kernel derives it from the cell coefficients), then x_prev
  double epilogue_bytes(int mode) const
  {
    const double w = sizeof(T) * double(_n_dofs);
    return mode == 0 ? 0. : mode == 1 ? w : (mode == 2 ? 1. : 2.) * w + (_dinv_in_record ? w : 0.);
  }

private:
  void launch(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out) const;
  void run(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out, int nw, int ty, int tz,
           int z_tile_begin = 0, int z_tile_end = -1, int const *xy_range = nullptr, bool with_main = true,
           bool with_tail = true, int const *exclude = nullptr, hipStream_t on_stream = nullptr) const;
  bool make_args(MfArgs<T> &a, unsigned int &n_blocks, MfMode mode, T const *x, T const *b, T const *x_prev, T alpha,
                 T beta, T *out, int nw, int ty, int tz, int const *ztab, int z_tile_begin, int z_tile_end,
                 int const *xy_range) const;
  // layers of the z-tiles (device table, built once per tz): graded = shorter tiles at the end of every XCD's run
  int const *z_tiling(int tz, bool graded, int &n_tiles) const;
  struct ZTiling
  {
    int tz = 0, n_tiles = 0;
    DeviceBuffer<int> dev;
  };
  mutable ZTiling _zt_uniform, _zt_graded;
  void check_vectors(MfMode mode, T const *x, T const *b, T const *x_prev, T const *out) const;
  // dim = 2 (mf_laplace.hip): a plain owner-computes kernel on the caller's arrays
  void init_2d(mfmg_hip_mesh_desc const &mesh);
  void launch_2d(MfMode mode, T const *x, T const *b, T const *x_prev, T alpha, T beta, T *out) const;
  int _dim = 3;
  DeviceBuffer<int32_t> _cd2, _node_dof2;
  DeviceBuffer<T> _co2;
  DeviceBuffer<uint8_t> _cn2;
  double _k2[64] = {};
  void choose_tile(int &nw, int &ty, int &tz) const;
  void choose_fused_tile(int n_terms, int &nw, int &ty, int &tz) const;
  int _fused_tile[3] = {0, 0, 0};
  bool _fused_reference_arithmetic = false;

  HipHandle &_handle;
  int _N[3]; // DoF grid
  int _n[3]; // cells
  double _h[3];
  int64_t _n_dofs;
  // internal layout (mf_laplace.hip): rows cut into aligned chunks of 64 cell slots (62 owned DoFs + the
  // halo cell); one record per chunk with the b=1 face ids, the coefficients and D^-1, plus the b=0 face ids
  int _ncols = 0;
  bool _narrow_last = false; // (see narrow_last_column)
  int _own = 62, _halo = 1; // chunk c holds the node columns _own c - _halo + lane and owns its lanes [_halo, _halo + _own)
  size_t _n_slots = 0;
  // columns of a nearly empty last chunk, as a slab operator with x and y exchanged (mf_laplace.hip)
  std::unique_ptr<MatrixFreeLaplaceDevice<T>> _tail;
  DeviceBuffer<unsigned char> _rec;
  DeviceBuffer<T> _diag, _dinv;
  int _tile_y = 0, _tile_z = 0, _tile_waves = 0;
  bool _compact = false;
  bool _dinv_in_record = true; // D^-1 is part of the chunk records (always for eight coefficients per cell)
  size_t _rec_bytes = 0;
  // a numbering the kernel computes instead of reading it from the records (mf_laplace.hip: AffineIds)
  AffineIds _affine = {0, 1, 0, 0, 0, {0, 0, 0}, {0, 0, 0}};
  bool _affine_ids = false;
};
} // namespace mfmg
