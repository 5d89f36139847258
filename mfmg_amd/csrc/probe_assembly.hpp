// CSR matrices of the setup assembled on the device from colour probes (probe_assembly.hip).
// `Y` / `Z` are device arrays [colour][owned row]; the geometry arguments are those of the host loops these
// functions replace (hip_hierarchy.hip: HipMatrixOperator::multiply; amg_device_setup.hip).
#pragma once

#include "amge_structured.hpp"
#include "sparse_matrix_device.hpp"

namespace mfmg
{
// A_c = R A R^T on an agglomerate grid `na` with `ne` rows per agglomerate: colours of period k (z on global layers,
// offset zoff), rows of the agglomerate layers [z_own0, z_own1) only; every neighbour inside the grid is stored.
std::shared_ptr<SparseMatrixDevice<double>> galerkin_from_probes(HipHandle &h, int const na[3], int ne, int const k[3], int64_t zoff,
                                                                 int64_t z_own0, int64_t z_own1, double const *Y);

// P = (I - w D^-1 A) P_tent: Z[colour][q] = (A y_colour)(row0 + q), t the tentative prolongator (ghosts exchanged),
// dinv the inverse diagonal; zeros are dropped.
std::shared_ptr<SparseMatrixDevice<double>> prolongator_from_probes(HipHandle &h, int const fdims[3], int const cdims[3],
                                                                    int const gdims_c[3], int n_comp, int blk, int reach,
                                                                    int const period[3], int64_t f_global_begin, int64_t c_global_begin,
                                                                    int64_t row0, int64_t n_own, double w, double const *Z,
                                                                    double const *t, double const *dinv);

// A_c = P^T A P: Y[colour][q] = (P^T A P u_colour)(crow0 + q); zeros are dropped.
std::shared_ptr<SparseMatrixDevice<double>> coarse_operator_from_probes(HipHandle &h, int const cdims[3], int const gdims_c[3], int n_comp,
                                                                        int reach, int const period[3], int64_t c_global_begin,
                                                                        int64_t crow0, int64_t cn_own, double const *Y);
// The assembled fine operator of a structured mesh (amge_structured.cpp: assemble_global_matrix) formed on the device:
// the same rows, bit for bit, sorted by column.
std::shared_ptr<SparseMatrixDevice<double>> fine_operator_on_device(HipHandle &h, StructuredMesh const &mesh, bool matrix_free_semantics);
} // namespace mfmg
