// Cell-local evaluation as a batched dense contraction (BASELINE.json configs[4]); cell_contraction.hip.
#pragma once

#include "common.hpp"

namespace mfmg
{
// v[m][cell] = c[cell] * sum_k K_ref[m][k] u[k][cell] on planar operands (eight planes of n_cells values);
// variant 0: vector ALU, 1: MFMA (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64); h = cell size
template <typename T>
void cell_contraction(HipHandle &handle, int variant, int64_t n_cells, T const *u, T const *c, T *v, double const h[3]);
} // namespace mfmg
