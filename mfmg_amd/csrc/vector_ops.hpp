// Vector kernels used inside the cycle: the deal.II CUDA vector operations the
// reference calls at include/mfmg/common/hierarchy.hpp:258,286,302 and
// source/cuda/cuda_smoother.cu:50-59 (x = 0, add, sadd, copy, l2_norm), plus the
// device-scalar building blocks of the coarse Krylov iteration.
#pragma once

#include "common.hpp"

namespace mfmg
{
namespace vec
{
template <typename T>
void set(HipHandle &h, int64_t n, T value, T *x); // x = value
template <typename T>
void copy(HipHandle &h, int64_t n, T const *src, T *dst);
template <typename T>
void add(HipHandle &h, int64_t n, T a, T const *v, T *x); // x += a v
template <typename T>
void sadd(HipHandle &h, int64_t n, T s, T a, T const *v, T *x); // x = s x + a v
template <typename T>
void scale_pointwise(HipHandle &h, int64_t n, T const *d, T const *v, T *out); // out = d .* v
template <typename T>
void scaled_pointwise(HipHandle &h, int64_t n, T s, T const *d, T const *v, T *out); // out = (s d) .* v

// probing vector of the Galerkin product on device: u[(a, e)] = 1 where e == e0 and the agglomerate a (x fastest on a
// grid na) has a_d mod k_d == o_d in every direction, else 0
// (offset: global index of the local agglomerate 0 per axis in a distributed run)
void probing_vector(HipHandle &h, int const na[3], int n_eig, int const k[3], int const o[3], int e0, double *u, int const offset[3]);

// out[r] = in[r] (1 when in is null) on the rows r = node * n_comp + comp of the nodes whose block coordinates
// ((local + offset) / block: global coordinates) are congruent to `phase` modulo `period`, with comp == `comp`;
// 0 elsewhere (probing vectors of the aggregation-hierarchy setup)
void select_rows(HipHandle &h, int const dims[3], int n_comp, int block, int const offset[3], int const period[3], int const phase[3],
                 int comp, double const *in, double *out);

// fine level in FP32 around an FP64 coarse hierarchy: out = (double) in, and x -= (float) correction
void widen(HipHandle &h, int64_t n, float const *in, double *out);
void subtract_narrowed(HipHandle &h, int64_t n, double const *correction, float *x);

// Deterministic two-stage dot product; the result lands in device slot
// `result_dev[slot]` (no host synchronisation).
template <typename T>
void dot_async(HipHandle &h, int64_t n, T const *x, T const *y, double *result_dev, int slot);
// synchronous host result
template <typename T>
double dot(HipHandle &h, int64_t n, T const *x, T const *y);
template <typename T>
double l2_norm(HipHandle &h, int64_t n, T const *x);

// ---- device-scalar CG building blocks (no host round trip, graph-capturable) ----
// scal[] slots hold dot products computed by dot_async.
// x += (scal[rz]/scal[pap]) p ; r -= (scal[rz]/scal[pap]) Ap ; z = dinv .* r (dinv == nullptr: z is left alone)
template <typename T>
void cg_update(HipHandle &h, int64_t n, T const *p, T const *Ap, T const *dinv, T *x, T *r, T *z,
               double const *scal, int slot_rz, int slot_pap);
// p = z + (scal[rz_new]/scal[rz_old]) p
template <typename T>
void cg_direction(HipHandle &h, int64_t n, T const *z, T *p, double const *scal, int slot_rz_new,
                  int slot_rz_old);
} // namespace vec
} // namespace mfmg
