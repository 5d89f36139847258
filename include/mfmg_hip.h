/*
 * mfmg_hip.h -- C ABI of libmfmg_hip.so: the MI355X (gfx950) drop-in for the
 * V-cycle apply path of ORNL-CEES/mfmg.
 *
 * The reference exposes this path through C++ abstract classes
 * (include/mfmg/common/{operator,smoother,solver,hierarchy}.hpp), selected by the
 * string switch create_hierarchy_helpers (include/mfmg/common/hierarchy.hpp:49-153).
 * The same classes are mirrored, deal.II-free, under mfmg_amd/csrc/mfmg/ ; this
 * header is the flat boundary underneath them: plain pointers and sizes, device
 * pointers in, status code out, no C++/torch types.  Every entry point cites the
 * reference interface it replaces (path:line relative to the reference tree).
 *
 * Conventions
 *  - all vectors are raw DEVICE pointers to `double` (or `float` for the _f32
 *    entry points) of the length the operator reports; the caller owns them;
 *  - matrices / operators / hierarchies are opaque handles that OWN their
 *    device arrays (as SparseMatrixDevice does,
 *    include/mfmg/cuda/sparse_matrix_device.templates.cuh:244-272);
 *  - every call is asynchronous on the context's HIP stream unless stated;
 *  - return value: MFMG_HIP_SUCCESS or an error code; the message of the last
 *    error on the calling thread is available from mfmg_hip_last_error().
 *    MFMG_HIP_ERROR_RUNTIME      <-> ASSERT_THROW  (std::runtime_error,
 *                                    include/mfmg/common/exceptions.hpp:48-52)
 *    MFMG_HIP_ERROR_NOT_IMPLEMENTED <-> ASSERT_THROW_NOT_IMPLEMENTED (:54-73)
 */
#ifndef MFMG_HIP_H
#define MFMG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: bumped whenever an entry point changes its signature or disappears; additions leave it.
 *   1  rounds 1-2 up to the host-callback halo exchange (mfmg_hip_context_set_halo_buffers)
 *   2  round 2: mfmg_hip_context_set_communicator(ctx, rank, n_ranks, ghost_low, ghost_high) replaces the old signature,
 *      mfmg_hip_context_set_halo_buffers removed (transports: mfmg_hip_context_use_rccl / _use_host_transport)
 *      (round 3 added mfmg_hip_hierarchy_ap_apply, mfmg_hip_rccl_available, mfmg_hip_abi_version: no change of the version)
 *   3  round 3: box decomposition -- mfmg_hip_host_exchange_fn (any number of partner ranks per call) replaces
 *      mfmg_hip_host_sendrecv_fn (rank -+ 1); added mfmg_hip_context_set_communicator_box, mfmg_hip_context_halo_box,
 *      mfmg_hip_context_exchange_volume
 *      (later in round 3 added mfmg_hip_context_use_reflecting_transport: no change of the version)
 * mfmg_hip_abi_version() returns the value the loaded library was built with. */
#define MFMG_HIP_ABI_VERSION 3

#define MFMG_HIP_SUCCESS 0
#define MFMG_HIP_ERROR_RUNTIME 1
#define MFMG_HIP_ERROR_NOT_IMPLEMENTED 2
#define MFMG_HIP_ERROR_INVALID_ARGUMENT 3
#define MFMG_HIP_ERROR_DEVICE 4

/* OperatorMode (include/mfmg/common/operator.hpp:19-23) */
#define MFMG_HIP_NO_TRANS 0
#define MFMG_HIP_TRANS 1

typedef struct mfmg_hip_context_s *mfmg_hip_context_t;       /* CudaHandle, include/mfmg/cuda/cuda_handle.cuh:25-48 */
typedef struct mfmg_hip_csr_s *mfmg_hip_csr_t;               /* SparseMatrixDevice<double>, include/mfmg/cuda/sparse_matrix_device.cuh:28-104 */
typedef struct mfmg_hip_mf_laplace_s *mfmg_hip_mf_laplace_t; /* CudaMatrixFreeOperator + the user's LaplaceOperator, source/cuda/cuda_matrix_free_operator.cu:32-37, tests/laplace_matrix_free.hpp:121-156 */
typedef struct mfmg_hip_hierarchy_s *mfmg_hip_hierarchy_t;   /* Hierarchy<VectorType>, include/mfmg/common/hierarchy.hpp:155-373 */

const char *mfmg_hip_last_error(void);
const char *mfmg_hip_version(void);
int mfmg_hip_abi_version(void);
/* What the device memory of the library is spent on right now: one line per (setup section of the reference's TimerOutput |
 * kind of structure) with at least 1 MB live -- "Setup: build restrictor | CSR arrays (val, col, row_ptr)", ... -- and the total,
 * as text (GB).  Process-wide (all contexts). */
int mfmg_hip_memory_inventory(char *buffer, size_t buffer_size);

/* ---- context: stream + scratch (CudaHandle, source/cuda/cuda_handle.cu:17-56) ---- */
/* `hip_stream`: a hipStream_t borrowed from the caller; NULL = the legacy default stream (what the
 * reference runs on); MFMG_HIP_OWN_STREAM = the library creates and owns a non-blocking stream. */
#define MFMG_HIP_OWN_STREAM ((void *)(intptr_t)-1)
int mfmg_hip_context_create(void *hip_stream, mfmg_hip_context_t *ctx);
int mfmg_hip_context_destroy(mfmg_hip_context_t ctx);
int mfmg_hip_context_synchronize(mfmg_hip_context_t ctx);
void *mfmg_hip_context_stream(mfmg_hip_context_t ctx);

/* ---- distributed runs: one process per GPU, the mesh cut into slabs along z or into boxes (2 x 1 x 1, 2 x 2 x 1, 2 x 2 x 2 ...) ----
 * Replaces deal.II's distributed::Vector ghost exchange inside MatrixFree::cell_loop and the all-gather of
 * the whole vector in front of every SpMV on the CUDA path (source/cuda/utils.cu:363-482,
 * include/mfmg/cuda/sparse_matrix_device.templates.cuh:104-138).  The local mesh of a rank is its owned
 * cell slab (box) plus `ghost_cells_low/high` (0 or 2 = one agglomerate) cell layers of its neighbours (per split axis); local
 * DoF numbering must be lexicographic; ghost (not owned) DoFs carry the value 2 in mfmg_hip_mesh_desc.constrained.
 * Boxes (the reference's p4est partition, tests/laplace_matrix_free.hpp:222, with rank-local agglomerates,
 * include/mfmg/common/amge.templates.hpp:453-478,604-621): mfmg_hip_context_set_communicator_box places the rank at
 * (cx, cy, cz) = (rank % gx, (rank / gx) % gy, rank / (gx gy)) of a gx x gy x gz grid of equal boxes; interface planes belong
 * to the upper box (the last box of an axis also owns the top plane).  One exchange = one packing kernel, ONE grouped
 * send/recv with all neighbours (faces, edges, corners: 7 on a 2 x 2 x 2 grid, at most 26) and one unpacking kernel: per fine
 * exchange a rank of a 2 x 2 x 2 grid sends 3 faces of (N/2)^2 doubles (+ 3 edges + 1 corner) where a slab of 8 sends 2 planes
 * of N^2.
 * Every level of the V-cycle is coupled across the ranks: before an operator application the library refreshes
 * the ghost layers of its input (owner -> ghost), after a transposed prolongator it returns the partial sums in
 * the ghost layers to their owners (ghost -> owner, added); the levels of the aggregation hierarchy whose global
 * size is small are gathered and solved redundantly on every rank.  The distributed cycle is the same
 * preconditioner as the single-process one (same matrices to rounding).
 * Transport, one of:
 *   mfmg_hip_context_use_rccl            ncclSend / ncclRecv between slab neighbours on the library's HIP stream
 *                                        (RCCL over xGMI); the 128-byte id comes from mfmg_hip_rccl_unique_id on
 *                                        rank 0 and must reach every rank through the caller's own channel;
 *   mfmg_hip_context_use_host_transport  the library stages the layers through pinned host memory and calls the
 *                                        callbacks with HOST pointers (tests: gloo, several ranks on one card).
 * `exchange` sends count[i] doubles from send[i] to rank peers[i] and receives as many from it into recv[i], for i < n, all
 * at once (slabs: rank - 1 and / or rank + 1);
 * `allreduce` combines `n` doubles over all ranks in place (op 0: sum, 1: max); `allgather` collects `n` doubles of
 * every rank into `out` (n * n_ranks, rank order). */
typedef int (*mfmg_hip_host_exchange_fn)(void *user, int32_t n, const int32_t *peers, const double *const *send, double *const *recv,
                                         const int64_t *count);
typedef int (*mfmg_hip_host_allreduce_fn)(void *user, double *values, int n, int op);
typedef int (*mfmg_hip_host_allgather_fn)(void *user, const double *in, int64_t n, double *out);
int mfmg_hip_context_set_communicator(mfmg_hip_context_t ctx, int32_t rank, int32_t n_ranks, int32_t ghost_cells_low,
                                      int32_t ghost_cells_high);
/* grid[3]: ranks along x, y, z (their product = the number of ranks); ghost_low / ghost_high[3]: ghost cell layers of the local
 * mesh per axis (2 towards an existing neighbour -- or 4 below, see mfmg_hip_context_set_low_ghost_cells --, else 0).  set_communicator(rank, n, lo, hi) is the grid 1 x 1 x n. */
int mfmg_hip_context_set_communicator_box(mfmg_hip_context_t ctx, int32_t rank, const int32_t grid[3], const int32_t ghost_low[3],
                                          const int32_t ghost_high[3]);
/* Ghost cell layers EVERY rank of the run holds towards a lower neighbour: 2 (default, one agglomerate) or 4 (two; then
 * ghost_low[d] = 4 above).  The same value on every rank, also on ranks without a lower neighbour, after set_communicator*:
 * the interface plane belongs to the upper box, so a box holds three ghost node planes above it and `cells` below, and a
 * sweep of K smoother terms (one exchange of x, K planes deep, ghost DoFs computed redundantly) needs K on every side with a
 * neighbour -- 4 lets the whole Chebyshev(3) smoother of a rank run as ONE sweep with ONE exchange. */
int mfmg_hip_context_set_low_ghost_cells(mfmg_hip_context_t ctx, int32_t cells);
int mfmg_hip_rccl_unique_id(unsigned char out[128]);
/* MFMG_HIP_SUCCESS when librccl and the entry points the transport uses can be resolved in this process (dlopen / dlsym
 * only, no RCCL call is made).  Callers agree on the transport with a collective over this flag before they choose. */
int mfmg_hip_rccl_available(void);
int mfmg_hip_context_use_rccl(mfmg_hip_context_t ctx, const unsigned char unique_id[128]);
int mfmg_hip_context_use_host_transport(mfmg_hip_context_t ctx, mfmg_hip_host_exchange_fn exchange,
                                        mfmg_hip_host_allreduce_fn allreduce, mfmg_hip_host_allgather_fn allgather, void *user);
/* MEASUREMENT of one rank's share of a distributed cycle on one GPU, for a hierarchy that was set up with a real transport:
 * from this call on every message this rank sends is copied back on the device as the message it would have received, an
 * all-gather repeats its block, an all-reduce is the identity -- all kernels, packings, stream joins and replicated levels of
 * the partition with a wire that costs nothing (scratch/rank_cycle_on_one_gpu.py).  What the rank iterates on afterwards is
 * not a solution of anything. */
int mfmg_hip_context_use_reflecting_transport(mfmg_hip_context_t ctx);
/* ... the same with a price on the wire: every grouped send/recv and every collective holds its stream for
 * `microseconds_per_group` (a one-thread kernel) before the reflected data arrive -- the latency of a real group, e.g. what
 * mfmg_hip_context_transport_loopback_time measured for the RCCL transport */
int mfmg_hip_context_use_reflecting_transport_delay(mfmg_hip_context_t ctx, double microseconds_per_group);
/* stream time of one loop-back group (send to self + receive from self, n doubles) of the registered transport: `reps` groups
 * back to back between two events */
int mfmg_hip_context_transport_loopback_time(mfmg_hip_context_t ctx, int64_t n, int reps, double *microseconds);
/* "rccl", "host", "reflecting" or "" (none) */
int mfmg_hip_context_transport_name(mfmg_hip_context_t ctx, char *buffer, size_t buffer_size);
/* ranks the registered transport's own communicator reports (RCCL: ncclCommCount; 1 without a transport) */
int mfmg_hip_context_transport_ranks(mfmg_hip_context_t ctx, int *n_ranks);
/* exercises the registered transport: `n` doubles sent to this rank itself and back, an all-gather and sum / max
 * all-reduces over all ranks; returns the largest deviation from the known answers (0 when everything arrived) */
int mfmg_hip_context_transport_selftest(mfmg_hip_context_t ctx, int64_t n, double *max_error);
/* point-to-point exchanges issued through the context so far (diagnostics) */
int mfmg_hip_context_exchange_count(mfmg_hip_context_t ctx, int64_t *n_exchanges);
/* ... the doubles this rank has sent in them, and (n_overlapped, may be NULL) how many of the exchanges ran on the second
 * stream beside the operator tiles that read no ghost plane */
int mfmg_hip_context_exchange_volume(mfmg_hip_context_t ctx, int64_t *n_doubles_sent, int64_t *n_overlapped);
/* one halo exchange of a device vector of `space` (1 fine DoFs, 2 first coarse level, 3.. aggregation levels):
 * reverse = 0 owner -> ghost, 1 ghost -> owner (added).  The cycle does this by itself; exposed for tests. */
int mfmg_hip_context_exchange(mfmg_hip_context_t ctx, int32_t space, double *vector, int reverse);
/* sum over the ranks of the dot product over the owned entries of two vectors of `space` */
int mfmg_hip_context_owned_dot(mfmg_hip_context_t ctx, int32_t space, const double *x, const double *y, double *result);
/* Distributed runs: overlap the exchange of the fine-level ghost planes with the operator tiles that do not read
 * them (second HIP stream; default on).  Off: exchange first, then one launch over all tiles.  Same results. */
int mfmg_hip_context_set_overlap_exchange(mfmg_hip_context_t ctx, int enable);
/* Matrix-free operators created afterwards: when the eight quadrature coefficients of every cell are equal
 * (material_property constant, the reference's default, or any cell-wise constant material) keep ONE value per
 * cell (20 instead of 76 bytes per cell in FP64; default on).  Off forces the general eight-value layout. */
int mfmg_hip_context_set_cell_constant_layout(mfmg_hip_context_t ctx, int enable);
/* With one coefficient per cell the diagonal of a DoF is kd * (sum of the coefficients of its eight cells); by default the
 * operator kernel derives D^-1 from that on the fly and the chunk records hold no D^-1 (8 bytes per DoF and smoother launch
 * less to read, 768 instead of 1280 B per chunk).  enable != 0: operators created afterwards keep D^-1 in the records. */
int mfmg_hip_context_set_stored_diagonal(mfmg_hip_context_t ctx, int enable);
int mfmg_hip_mf_laplace_diagonal_in_record(mfmg_hip_mf_laplace_t op, int *in_record);
/* Polynomial terms of the Chebyshev smoother that ONE sweep of a matrix-free operator may run (1, 2 or 3; default 3: the whole
 * Chebyshev(3) apply of DealIIMatrixFreeSmoother::apply, source/dealii/dealii_matrix_free_smoother.cc:63-76, reads x, b and the
 * coefficients once).  Operators created afterwards cut their rows into chunks with that many halo columns on either side
 * (1 = the layout of the one-term kernels, which a distributed run always takes). */
int mfmg_hip_context_set_mf_fused_terms(mfmg_hip_context_t ctx, int n_terms);
/* Measurement switches of the distributed fine operator (the environment variables MFMG_MF_SHELL / MFMG_MF_EMULATE_SPLIT give the
 * initial values when the context is created; nothing reads the environment per application).  shell mode: 0 = the shell of
 * tiles around the interior as one launch on the exchange stream beside the interior tiles (default), 1 = after them, 2 = slab
 * by slab.  emulate split (one rank): the launches a rank of 1x1x2 (1), 1x2x2 (2), 2x2x2 (3) would make, without an exchange. */
int mfmg_hip_context_set_mf_shell(mfmg_hip_context_t ctx, int mode);
int mfmg_hip_context_set_mf_emulate_split(mfmg_hip_context_t ctx, int axes);
/* 1 when the operator kernel COMPUTES the DoF ids instead of reading them from its records: a numbering that is affine
 * on the node grid (any lexicographic one), Dirichlet DoFs on whole faces of the box, ghost DoFs on whole z-layers,
 * checked slot by slot at construction; used by the eight-coefficient kernels (where it pays).  MFMG_MF_AFFINE_IDS=0
 * keeps the ids of the records. */
int mfmg_hip_mf_laplace_ids_computed(mfmg_hip_mf_laplace_t op, int *computed);
/* Hierarchies created afterwards form the coarse operator R A R^T of a matrix-free A on the device by probing
 * (27 n_eig applications of R^T, A and R over colour classes of agglomerates; the reference's fast_ap idea,
 * source/dealii/dealii_matrix_free_hierarchy_helpers.cc:77-288) -- the default where the restrictor has the block
 * structure, distributed runs included (colours on global coordinates, every rank keeps its own rows); 0: the host
 * triple product.  Same matrix to rounding. */
int mfmg_hip_context_set_galerkin_on_device(mfmg_hip_context_t ctx, int enable);
/* layout of a distributed space after the hierarchy was built: entries per layer, local layers, owned range;
 * mfmg_hip_context_halo_space adds {global index of local layer 0, global layers, exchange width, number of spaces} */
int mfmg_hip_context_halo_layout(mfmg_hip_context_t ctx, int32_t space, int64_t *layer_elems, int64_t *n_layers,
                                 int64_t *owned_begin, int64_t *owned_count);
int mfmg_hip_context_halo_space(mfmg_hip_context_t ctx, int32_t space, int64_t out[8]);
/* the same per axis (x, y, z): out = {entries per node, local nodes[3], first owned[3], owned[3], global index of local node 0 [3],
 * global nodes[3]}: a vector of the space is the lexicographic array of the local nodes, `entries per node` values each */
int mfmg_hip_context_halo_box(mfmg_hip_context_t ctx, int32_t space, int64_t out[16]);

/* BASELINE.json configs[4]: the cell-local evaluation of the Q1 Laplace as a batched dense contraction,
 * v[m][cell] = c[cell] * sum_k K_ref[m][k] u[k][cell] (K_ref: reference matrix of a Cartesian cell of size `cell_size`,
 * what LaplaceOperator::local_apply computes per cell for a cell-wise constant coefficient, tests/laplace_matrix_free.hpp:129-156)
 * on planar device operands u, v: [8][n_cells], c: [n_cells].  fp32 != 0: float operands.  variant 0: vector ALU,
 * 1: MFMA (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64).  Timed under the kernel names "cell_contraction_valu" /
 * "cell_contraction_mfma" (17 values moved per cell). */
int mfmg_hip_cell_contraction(mfmg_hip_context_t ctx, int fp32, int variant, int64_t n_cells, const void *u, const void *c,
                              void *v, const double cell_size[3]);

/* Per-kernel timing with HIP events recorded on the launch stream (the reference only has the
 * wall-clock dealii::TimerOutput sections, include/mfmg/common/hierarchy.hpp:36-47).  Kernel names:
 * "mf_laplace_kernel", "csr_spmv_kernel".  `algorithmic_bytes` sums SURVEY.md 8d's per-launch figures. */
int mfmg_hip_profile_enable(mfmg_hip_context_t ctx, int enabled);
/* restrict the timing to launches of one kernel name (NULL or "" = all); every timed launch costs two event records */
int mfmg_hip_profile_select(mfmg_hip_context_t ctx, const char *kernel_name);
int mfmg_hip_profile_query(mfmg_hip_context_t ctx, const char *kernel_name, int64_t *n_launches, double *total_ms,
                           double *algorithmic_bytes);

/* ---- host<->device marshalling (source/cuda/utils.cu:484-510, include/mfmg/cuda/utils.cuh:66-99) ---- */
int mfmg_hip_malloc(void **dev_ptr, size_t bytes);                      /* cuda_malloc */
int mfmg_hip_free(void *dev_ptr);                                       /* cuda_free */
int mfmg_hip_copy_to_dev(void *dst_dev, const void *src_host, size_t bytes);  /* cuda_mem_copy_to_dev */
int mfmg_hip_copy_to_host(void *dst_host, const void *src_dev, size_t bytes); /* cuda_mem_copy_to_host */

/* ---- vector kernels used inside the cycle (deal.II CUDA vector ops called at
 *      include/mfmg/common/hierarchy.hpp:258,286,302 and source/cuda/cuda_smoother.cu:50-59) ---- */
int mfmg_hip_vector_set(mfmg_hip_context_t ctx, int64_t n, double value, double *x);                 /* x = value        */
int mfmg_hip_vector_add(mfmg_hip_context_t ctx, int64_t n, double a, const double *v, double *x);    /* x.add(a, v)      */
int mfmg_hip_vector_sadd(mfmg_hip_context_t ctx, int64_t n, double s, double a, const double *v, double *x); /* x.sadd(s,a,v) */
int mfmg_hip_vector_dot(mfmg_hip_context_t ctx, int64_t n, const double *x, const double *y, double *result_host); /* synchronous */
int mfmg_hip_vector_l2_norm(mfmg_hip_context_t ctx, int64_t n, const double *x, double *result_host);              /* synchronous, x.l2_norm() */

/* ---- CSR matrix on device: SparseMatrixDevice<double> ---- */
/* convert_matrix (source/cuda/utils.cu:39-168): upload a host CSR (0-based, int32). */
int mfmg_hip_csr_create(mfmg_hip_context_t ctx, int64_t n_rows, int64_t n_cols, int64_t nnz,
                        const int32_t *row_ptr_host, const int32_t *col_host,
                        const double *val_host, mfmg_hip_csr_t *out);
int mfmg_hip_csr_destroy(mfmg_hip_csr_t a);
int mfmg_hip_csr_shape(mfmg_hip_csr_t a, int64_t *n_rows, int64_t *n_cols, int64_t *nnz); /* m(), n(), n_nonzero_elements() */
/* tuning knob of the SpMV kernel: lanes of a wavefront per row (power of two 1..64; 256 = a workgroup per row, the
 * choice for at most 4096 rows of 256 entries or more; 0 = keep) and the storage variant (0 plain CSR, 1 LDS-cached
 * CSR, 2 block-diagonal (get reports 3 when only the upper half of a symmetric matrix is stored), 4 row-base storage
 * of rectangular stencil-like matrices, 5 node classes (rectangular matrices whose rows repeat a few stencils: per-class
 * tables instead of stored values), -1 = keep; a variant is only taken where its data was built at construction);
 * the summation order, hence the last bits of the result, follows both */
int mfmg_hip_csr_set_kernel(mfmg_hip_csr_t a, int lanes_per_row, int use_lds);
int mfmg_hip_csr_get_kernel(mfmg_hip_csr_t a, int *lanes_per_row, int *use_lds);
/* Rows of a symmetric block-diagonal matrix that repeat one stencil bit for bit (interior rows of the coarse operators
 * of a constant-coefficient problem on a uniform mesh) are evaluated from a table of constants instead of stored
 * values.  in_use: whether such rows were found and the path is on; enable = 0 switches it off (same results to
 * rounding). */
int mfmg_hip_csr_regular_rows(mfmg_hip_csr_t a, int *in_use);
int mfmg_hip_csr_set_regular_rows(mfmg_hip_csr_t a, int enable);
/* Of the rows that are not regular: n_classes stencils that groups of nodes share among themselves (the shells
 * next to the boundary; evaluated from per-class tables) and listed_rows rows left to the stored values. */
int mfmg_hip_csr_stencil_classes(mfmg_hip_csr_t a, int *n_classes, int64_t *listed_rows);
/* 1 when the values the kernels read (block-diagonal planes; the agglomerate-wise planes of a restrictor) are kept in
 * float: every one of them is representable in it -- a hierarchy built with "setup value precision" float -- so the
 * storage is lossless; products and sums stay FP64. */
int mfmg_hip_csr_float_storage(mfmg_hip_csr_t a, int *in_float);
/* SparseMatrixDevice::vmult  (…templates.cuh:351-371): y = A x */
int mfmg_hip_csr_vmult(mfmg_hip_csr_t a, const double *x, double *y);
/* CudaMatrixOperator::apply (source/cuda/cuda_matrix_operator.cu:80-91); TRANS uses the
 * explicit transpose built once by transpose() (:93-130) -- here by device kernels at first use
 * (csr_algebra.hip; rows of the transpose beyond 4096 entries take the host algorithm). */
int mfmg_hip_csr_apply(mfmg_hip_csr_t a, const double *x, double *y, int mode);
/* CudaMatrixOperator::transpose (:93-130) */
int mfmg_hip_csr_transpose(mfmg_hip_csr_t a, mfmg_hip_csr_t *out);
/* SparseMatrixDevice::mmult / CudaMatrixOperator::multiply (…templates.cuh:373-434, cuda_matrix_operator.cu:132-149): C = A B
 * (setup; row-wise hash SpGEMM in LDS, sums in the order of the host product; rows with more than 2048 candidate
 * columns take the host algorithm; MFMG_CSR_ALGEBRA=host|device|device_only selects) */
int mfmg_hip_csr_multiply(mfmg_hip_csr_t a, mfmg_hip_csr_t b, mfmg_hip_csr_t *out);
/* download (copy_from_dev / convert_to_trilinos_matrix, source/cuda/utils.cu:170-204) */
int mfmg_hip_csr_download(mfmg_hip_csr_t a, int32_t *row_ptr_host, int32_t *col_host, double *val_host);
/* CudaSolver(handle, op, params)->apply(b, x) (source/cuda/cuda_solver.cu:204-515, the standalone use of
 * tests/test_direct_solver_device.cu:23-110): builds the coarse solver `solver.type` names for this matrix -- lu_dense |
 * cholesky | lu_sparse_host (dense LU factored here), pcg, amg -- applies it once from a zero guess and drops it. */
int mfmg_hip_csr_solve(mfmg_hip_csr_t a, const char *params_info, const double *b, double *x);
/* extract_inv_diag (source/cuda/cuda_smoother.cu:86-96): dinv[i] = 1/A_ii on device */
int mfmg_hip_csr_inverse_diagonal(mfmg_hip_csr_t a, double *dinv);
/* One fused Jacobi/Chebyshev step on an assembled operator:
 * out = x + alpha (x - x_prev) - beta * dinv .* (A x - b).   x_prev may be NULL when alpha == 0.
 * Replaces the r=Ax-b / tmp=B^-1 r / x-=tmp sequence of source/cuda/cuda_smoother.cu:48-59. `out` must not alias `x`. */
int mfmg_hip_csr_smoother_step(mfmg_hip_csr_t a, const double *dinv, const double *b, const double *x,
                               const double *x_prev, double alpha, double beta, double *out);
/* res = A x - b  (include/mfmg/common/hierarchy.hpp:284-286, negative residual) */
int mfmg_hip_csr_residual(mfmg_hip_csr_t a, const double *x, const double *b, double *res);

/* ---- matrix-free Q1 Laplace operator on a logically structured hex mesh ---- */
/* What the deal.II driver hands over (tests/laplace_matrix_free.hpp:243-313):
 * the cell->DoF index array, the coefficient table _coefficient(cell,q)
 * (:65,100-119), the constrained DoFs (AffineConstraints) and the Cartesian cell size. */
typedef struct mfmg_hip_mesh_desc
{
  int32_t dim;             /* 3 or 2 (2-D: the assembled path, and a plain matrix-free kernel for the small meshes of the reference tests) */
  int32_t n_cells[3];      /* cells per direction, lexicographic cell order, x fastest */
  double cell_size[3];     /* h_x, h_y, h_z (J = diag(h)) */
  int64_t n_dofs;          /* (n_cells+1) product */
  const int32_t *cell_dofs;     /* [n_cells_total][2^dim]  cell->get_dof_indices */
  const double *coefficient;    /* [n_cells_total][2^dim]  _coefficient(cell,q) */
  const uint8_t *constrained;   /* [n_dofs] 1 = Dirichlet-constrained DoF */
  int32_t arrays_on_device;     /* 0: host pointers, 1: device pointers */
} mfmg_hip_mesh_desc;

int mfmg_hip_mf_laplace_create(mfmg_hip_context_t ctx, const mfmg_hip_mesh_desc *mesh, mfmg_hip_mf_laplace_t *out);
int mfmg_hip_mf_laplace_destroy(mfmg_hip_mf_laplace_t op);
int mfmg_hip_mf_laplace_size(mfmg_hip_mf_laplace_t op, int64_t *n_dofs);
/* LaplaceOperator::vmult via matrix_free_evaluate_global (tests/test_hierarchy_helpers.hpp:370-375,
 * source/dealii/dealii_matrix_free_operator.cc:31-36): y = A x, constrained rows y_c = x_c */
int mfmg_hip_mf_laplace_vmult(mfmg_hip_mf_laplace_t op, const double *x, double *y);
/* matrix_free_get_diagonal_inverse (tests/test_hierarchy_helpers.hpp:377-382; compute_diagonal
 * tests/laplace_matrix_free.hpp:75-98): copies the inverse diagonal (constrained entries 1) */
int mfmg_hip_mf_laplace_diagonal_inverse(mfmg_hip_mf_laplace_t op, double *dinv);
int mfmg_hip_mf_laplace_diagonal(mfmg_hip_mf_laplace_t op, double *diag);
/* res = A x - b */
int mfmg_hip_mf_laplace_residual(mfmg_hip_mf_laplace_t op, const double *x, const double *b, double *res);
/* fused smoother step, as mfmg_hip_csr_smoother_step (DealIIMatrixFreeSmoother::apply,
 * source/dealii/dealii_matrix_free_smoother.cc:63-76, one polynomial term per call) */
int mfmg_hip_mf_laplace_smoother_step(mfmg_hip_mf_laplace_t op, const double *b, const double *x,
                                      const double *x_prev, double alpha, double beta, double *out);
/* n_terms (2 or 3) smoother terms in one sweep: x_1 = x - beta[0] dinv (A x - b), x_s = x_{s-1} + alpha[s-1] (x_{s-1} - x_{s-2})
 * - beta[s-1] dinv (A x_{s-1} - b); out = x_{n_terms}, out_prev (may be NULL) = x_{n_terms - 1}.  Bit for bit what n_terms calls
 * of mfmg_hip_mf_laplace_smoother_step return.  alpha[0] must be 0; x, out and out_prev must be different vectors.
 * MFMG_HIP_ERROR_NOT_IMPLEMENTED when the operator cannot run it (mfmg_hip_mf_laplace_sweep_available: cell-constant layout,
 * a numbering the kernel can compute, at least n_terms halo columns, one rank).
 * x == NULL: the sweep from x_0 = 0, which is then not read and whose first term needs no operator application (x_1 = beta[0] dinv b:
 * the pre-smoother of a preconditioner application, include/mfmg/common/hierarchy.hpp:253-259) -- three terms, default arithmetic;
 * the result of the sweep run on a zeroed vector.  Hierarchy::apply uses it when "is preconditioner" is true. */
int mfmg_hip_mf_laplace_sweep_available(mfmg_hip_mf_laplace_t op, int n_terms, int *available);
int mfmg_hip_mf_laplace_smoother_sweep(mfmg_hip_mf_laplace_t op, int n_terms, const double *alpha, const double *beta,
                                       const double *b, const double *x, double *out, double *out_prev);
/* The sweep's cell arithmetic.  Default (0): the cell matrix in mode space -- per direction a Q1 cell acts on the sum and the
 * difference of its two nodes alone, the 8 x 8 cell matrix is diagonal on the eight modes and the butterflies are shared between
 * neighbouring cells: ~30 % fewer FP64 operations, the same operator with its own rounding (1e-15 per cell).  on != 0: the
 * arithmetic of the one-term kernel, bit for bit (tests compare the two kernels that way). */
int mfmg_hip_mf_laplace_set_sweep_reference(mfmg_hip_mf_laplace_t op, int on);
/* tile of the sweep: n_waves wavefronts of tile_y cell rows (2, 3 or 4), tile_z owned layers; 0, 0, 0 = chosen from the mesh */
int mfmg_hip_mf_laplace_set_sweep_tile(mfmg_hip_mf_laplace_t op, int n_waves, int tile_y, int tile_z);
int mfmg_hip_mf_laplace_get_sweep_tile(mfmg_hip_mf_laplace_t op, int n_terms, int *n_waves, int *tile_y, int *tile_z);
/* FP32 instance of the same operator (BASELINE.json configs[4]; the coefficient table is converted once,
 * vectors are device `float`).  The cell kernel stays on the vector ALU: at ~11 flop/B it sits below the
 * FP32-MFMA ridge, so a batched-GEMM reformulation would not lift the HBM bound. */
typedef struct mfmg_hip_mf_laplace_f32_s *mfmg_hip_mf_laplace_f32_t;
int mfmg_hip_mf_laplace_f32_create(mfmg_hip_context_t ctx, const mfmg_hip_mesh_desc *mesh, mfmg_hip_mf_laplace_f32_t *out);
int mfmg_hip_mf_laplace_f32_destroy(mfmg_hip_mf_laplace_f32_t op);
int mfmg_hip_mf_laplace_f32_cell_constant_layout(mfmg_hip_mf_laplace_f32_t op, int *in_use);
int mfmg_hip_mf_laplace_f32_ids_computed(mfmg_hip_mf_laplace_f32_t op, int *computed); /* as mfmg_hip_mf_laplace_ids_computed */
int mfmg_hip_mf_laplace_f32_vmult(mfmg_hip_mf_laplace_f32_t op, const float *x, float *y);
int mfmg_hip_mf_laplace_f32_diagonal_inverse(mfmg_hip_mf_laplace_f32_t op, float *dinv);
int mfmg_hip_mf_laplace_f32_residual(mfmg_hip_mf_laplace_f32_t op, const float *x, const float *b, float *res);
int mfmg_hip_mf_laplace_f32_smoother_step(mfmg_hip_mf_laplace_f32_t op, const float *b, const float *x,
                                          const float *x_prev, float alpha, float beta, float *out);
int mfmg_hip_mf_laplace_f32_sweep_available(mfmg_hip_mf_laplace_f32_t op, int n_terms, int *available);
int mfmg_hip_mf_laplace_f32_set_sweep_reference(mfmg_hip_mf_laplace_f32_t op, int on);
int mfmg_hip_mf_laplace_f32_smoother_sweep(mfmg_hip_mf_laplace_f32_t op, int n_terms, const float *alpha, const float *beta,
                                           const float *b, const float *x, float *out, float *out_prev);
/* tuning knob: owned DoF rows / planes per workgroup tile (0 = heuristic) */
int mfmg_hip_mf_laplace_set_tile(mfmg_hip_mf_laplace_t op, int tile_y, int tile_z);
/* wavefronts per workgroup (1..8) stacked in y that hand their boundary sums on through LDS (0 = heuristic) */
int mfmg_hip_mf_laplace_set_tile_waves(mfmg_hip_mf_laplace_t op, int n_waves);
/* 1 when the operator keeps one coefficient per cell (see mfmg_hip_context_set_cell_constant_layout) */
int mfmg_hip_mf_laplace_cell_constant_layout(mfmg_hip_mf_laplace_t op, int *in_use);
/* the tile in use */
int mfmg_hip_mf_laplace_get_tile(mfmg_hip_mf_laplace_t op, int *n_waves, int *tile_y, int *tile_z);

/* ---- hierarchy: Hierarchy<VectorType> ---- */
/* Evaluator tag strings accepted by the string switch (create_hierarchy_helpers,
 * include/mfmg/common/hierarchy.hpp:49-107):
 *   "HipMeshEvaluator"            assembled CSR path (twin of "CudaMeshEvaluator")
 *   "HipMatrixFreeMeshEvaluator"  matrix-free path   (the tag "CudaMatrixFreeMeshEvaluator",
 *                                  source/cuda/cuda_matrix_free_mesh_evaluator.cu:21-24, is never dispatched upstream)
 * `params_info` is the text of a boost::property_tree INFO file
 * (tests/data/hierarchy_input.info); keys as consumed at hierarchy.hpp:168-172,215,
 * dealii_matrix_free_smoother.cc:24-51, cuda_smoother.cu:105, cuda_solver.cu:215. */
int mfmg_hip_hierarchy_create(mfmg_hip_context_t ctx, const char *evaluator_type,
                              const mfmg_hip_mesh_desc *mesh, const char *params_info,
                              mfmg_hip_hierarchy_t *out);
int mfmg_hip_hierarchy_destroy(mfmg_hip_hierarchy_t h);
/* Hierarchy::apply(b, x, 0)  (hierarchy.hpp:246-309) */
int mfmg_hip_hierarchy_apply(mfmg_hip_hierarchy_t h, const double *b, double *x);
/* Hierarchy::vmult(x, b)     (hierarchy.hpp:238-244) */
/* The same cycle with the fine level in FP32 (BASELINE.json configs[4]): float device vectors; pre-smoother,
 * residual and post-smoother through the FP32 instance of the matrix-free operator, restriction, coarse solve and
 * prolongation in FP64.  Needs the parameter "fine level precision" float at creation (matrix-free evaluator, two
 * levels, one process). */
int mfmg_hip_hierarchy_apply_f32(mfmg_hip_hierarchy_t h, const float *b, float *x);
int mfmg_hip_hierarchy_vmult(mfmg_hip_hierarchy_t h, double *x, const double *b);
/* Outer Krylov driver of tests/hierarchy_driver.cc:103-116: dealii::SolverCG on the fine-level operator with
 * Hierarchy::vmult as preconditioner, SolverControl(max_iterations, tolerance) on the absolute l2 norm of the
 * residual.  x holds the initial guess on entry.  residual_history (may be NULL) receives ||r_0||, ||r_1||, ... as
 * far as history_len reaches.  Returns MFMG_HIP_ERROR when max_iterations is reached without convergence
 * (SolverControl::NoConvergence upstream), with the outputs filled. */
int mfmg_hip_hierarchy_solve_cg(mfmg_hip_hierarchy_t h, const double *b, double *x, double tolerance,
                                int32_t max_iterations, int32_t *n_iterations, double *final_residual,
                                double *residual_history, int32_t history_len);
int mfmg_hip_hierarchy_n_levels(mfmg_hip_hierarchy_t h, int32_t *n_levels);
int mfmg_hip_hierarchy_level_size(mfmg_hip_hierarchy_t h, int32_t level, int64_t *n);
/* Level::get_operator()->apply (level.hpp:30-33) */
int mfmg_hip_hierarchy_operator_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *x, double *y, int mode);
/* Level::get_smoother()->apply(b, x) (level.hpp:40-43) */
int mfmg_hip_hierarchy_smoother_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *b, double *x);
/* levels[level].get_restrictor()->apply(in, out, mode): level >= 1 (level.hpp:35-38) */
int mfmg_hip_hierarchy_restrictor_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *in, double *out, int mode);
/* out = (A R^T) in for the A R^T the coarse operator of `level` was formed from -- `fast_multiply_transpose()` when the
 * parameter `fast_ap` is true (include/mfmg/common/hierarchy.hpp:214-221), `a->multiply_transpose(restrictor)` otherwise.
 * Kept only by a hierarchy built with `keep_ap = true` (tests: the comparison of tests/test_hierarchy.cc:507-642). */
int mfmg_hip_hierarchy_ap_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *in, double *out);
/* coarsest Level::get_solver()->apply(b, x) (level.hpp:45-48) */
int mfmg_hip_hierarchy_coarse_apply(mfmg_hip_hierarchy_t h, const double *b, double *x);
/* Replace the restrictor (and re-derive the Galerkin coarse operator + coarse solver)
 * from a host CSR so that CPU and GPU runs can share the identical R. */
int mfmg_hip_hierarchy_set_restrictor(mfmg_hip_hierarchy_t h, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                      const int32_t *row_ptr_host, const int32_t *col_host, const double *val_host);
/* b_coarse = R (A x - b): the residual of hierarchy.hpp:281-286 and its restriction (:288-290) as the cycle computes them.
 * Where the rows of R A repeat themselves from agglomerate to agglomerate (constant coefficient) this is ONE kernel
 * over x and b (residual_restriction.hip; `restrictor.fused_residual false` or MFMG_FUSED_RESIDUAL=0 switch it off);
 * otherwise the fused residual kernel followed by the restriction.  *_classes: number of agglomerate classes of the
 * one-pass form, 0 when the two-step form is in use. */
int mfmg_hip_hierarchy_restrict_residual(mfmg_hip_hierarchy_t h, int32_t level, const double *x, const double *b, double *b_coarse);
int mfmg_hip_hierarchy_residual_restriction_classes(mfmg_hip_hierarchy_t h, int32_t level, int32_t *n_classes);
/* restrictor / coarse operator download for inspection: query sizes with *_shape first */
int mfmg_hip_hierarchy_get_restrictor(mfmg_hip_hierarchy_t h, mfmg_hip_csr_t *r_borrowed);
int mfmg_hip_hierarchy_get_coarse_operator(mfmg_hip_hierarchy_t h, mfmg_hip_csr_t *ac_borrowed);
/* the operator of level 0 when it is an assembled matrix ("HipMeshEvaluator": evaluate_global,
 * include/mfmg/cuda/cuda_mesh_evaluator.cuh:39-45); borrowed like the two above; an error for a matrix-free hierarchy */
int mfmg_hip_hierarchy_get_fine_operator(mfmg_hip_hierarchy_t h, mfmg_hip_csr_t *a_borrowed);
/* levels of the multilevel coarse solver (0 when the coarse solver is direct / pcg): operator A_l, prolongator P_l,
 * its transpose as stored for the restriction (which = 0 / 1 / 2; a borrowed handle, valid until the next call) and the
 * Chebyshev bounds of its smoother */
int mfmg_hip_hierarchy_coarse_amg_levels(mfmg_hip_hierarchy_t h, int32_t *n_levels);
/* distributed runs: index of the first level of the multilevel coarse solver that is gathered and solved redundantly on
 * every rank (-1: one rank); the levels before it are coupled across the ranks by halo exchanges */
int mfmg_hip_hierarchy_coarse_amg_gather_level(mfmg_hip_hierarchy_t h, int32_t *level);
int mfmg_hip_hierarchy_coarse_amg_get(mfmg_hip_hierarchy_t h, int32_t level, int32_t which, mfmg_hip_csr_t *borrowed);
int mfmg_hip_hierarchy_coarse_amg_smoother(mfmg_hip_hierarchy_t h, int32_t level, int32_t *degree, double *lambda_min,
                                           double *lambda_max);
/* smoother polynomial actually used (degree, lambda_min, lambda_max) */
int mfmg_hip_hierarchy_smoother_info(mfmg_hip_hierarchy_t h, int32_t *degree, double *lambda_min, double *lambda_max);
/* polynomial terms the fine-level smoother runs as ONE sweep over the mesh (mf_cheb_fused.hip): by Smoother::apply (in place: the last
 * term stays a launch of its own) and by the out-of-place form Hierarchy::apply uses where the whole polynomial fits a sweep;
 * 0 = one launch per term (eight coefficients per cell, a numbering the kernel cannot compute, smoother.fused_terms 1) */
int mfmg_hip_hierarchy_smoother_sweep_terms(mfmg_hip_hierarchy_t h, int *terms_in_place, int *terms_out_of_place);
/* tile of that sweep for n_terms terms: n_waves wavefronts of tile_y cell rows, tile_z owned layers (0, 0, 0 sets the heuristic) */
int mfmg_hip_hierarchy_sweep_tile(mfmg_hip_hierarchy_t h, int n_terms, int *n_waves, int *tile_y, int *tile_z);
int mfmg_hip_hierarchy_set_sweep_tile(mfmg_hip_hierarchy_t h, int n_waves, int tile_y, int tile_z);
/* tile of the matrix-free fine-level operator (n_waves, rows per wavefront, layers), see mfmg_hip_mf_laplace_get_tile */
int mfmg_hip_hierarchy_operator_tile(mfmg_hip_hierarchy_t h, int *n_waves, int *tile_y, int *tile_z);
int mfmg_hip_hierarchy_set_operator_tile(mfmg_hip_hierarchy_t h, int n_waves, int tile_y, int tile_z); /* 0 = automatic */
/* TimerOutput-style accumulated wall times of the sections of hierarchy.hpp:164-271 as a text table */
int mfmg_hip_hierarchy_timer_report(mfmg_hip_hierarchy_t h, char *buf, size_t buf_size);

/* ---- host-side setup pieces (no GPU needed; SURVEY.md 8f rank 1-2, kept on the host cores) ----
 * Opaque host CSR results are returned through a handle and copied out with *_host_csr_get. */
typedef struct mfmg_hip_host_csr_s *mfmg_hip_host_csr_t;
int mfmg_hip_host_csr_shape(mfmg_hip_host_csr_t m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
int mfmg_hip_host_csr_get(mfmg_hip_host_csr_t m, int32_t *row_ptr, int32_t *col, double *val);
int mfmg_hip_host_csr_destroy(mfmg_hip_host_csr_t m);
/* Laplace<dim>::assemble_system on the structured mesh (tests/laplace.hpp:154-204); semantics 0: assembled
 * (AffineConstraints elimination), 1: matrix-free (identity rows on constrained DoFs). Host arrays only. */
int mfmg_hip_host_assemble_matrix(const mfmg_hip_mesh_desc *mesh, int semantics, mfmg_hip_host_csr_t *out);
/* AMGe restrictor (include/mfmg/common/amge.templates.hpp:271-325,412-499;
 * include/mfmg/dealii/amge_host.templates.hpp:278-483; include/mfmg/cuda/amge_device.templates.cuh:217-310).
 * `params_info`: INFO text with the eigensolver / agglomeration sections; `matrix_free` picks the evaluator flavour. */
int mfmg_hip_host_build_restrictor(const mfmg_hip_mesh_desc *mesh, const char *params_info, int matrix_free,
                                   mfmg_hip_host_csr_t *out);
/* A_c = R (A R^T) (include/mfmg/common/hierarchy.hpp:214-233) from operator rows generated on the fly */
int mfmg_hip_host_galerkin(const mfmg_hip_mesh_desc *mesh, int semantics, int64_t n_rows, int64_t nnz,
                           const int32_t *r_row_ptr, const int32_t *r_col, const double *r_val,
                           mfmg_hip_host_csr_t *out);
/* Multilevel coarse solver setup ("solver.type amg": smoothed aggregation, the role ML plays at
 * source/dealii/dealii_solver.cc:48-66): level operators A_l and prolongators P_l on the host.
 * which = 0: A_l, 1: P_l (from level l+1 to l; empty on the last level). */
typedef struct mfmg_hip_host_amg_s *mfmg_hip_host_amg_t;
/* grid_dims / node_of_row / component_of_row (optional, may be NULL): rows live on nodes of a structured
 * grid (for the AMGe coarse level: the agglomerates, x fastest), several rows (components) per node;
 * aggregates are 2x2x2 blocks of nodes, one aggregate per component. */
int mfmg_hip_host_amg_build(int64_t n_rows, int64_t nnz, const int32_t *row_ptr, const int32_t *col, const double *val,
                            const double *near_null, const int32_t *grid_dims, const int32_t *node_of_row,
                            const int32_t *component_of_row, const char *params_info, mfmg_hip_host_amg_t *out);
int mfmg_hip_host_amg_n_levels(mfmg_hip_host_amg_t amg, int32_t *n_levels);
int mfmg_hip_host_amg_get(mfmg_hip_host_amg_t amg, int32_t level, int32_t which, mfmg_hip_host_csr_t *out);
int mfmg_hip_host_amg_destroy(mfmg_hip_host_amg_t amg);

/* boost::property_tree INFO round trip of the parameter reader (tests/test_utils.cc:23-60 exercises ptree2plist) */
int mfmg_hip_host_params_get(const char *params_info, const char *path, char *value_buf, size_t buf_size);


#ifdef __cplusplus
}
#endif
#endif /* MFMG_HIP_H */
