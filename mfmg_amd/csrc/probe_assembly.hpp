// CSR matrices of the setup assembled on the device from colour probes (probe_assembly.hip).
// `Y` / `Z` are device arrays [colour][local row]; the geometry arguments are those of the host loops these
// functions replace (hip_hierarchy.hip: HipMatrixOperator::multiply; amg_device_setup.hip).
#pragma once

#include "amge_structured.hpp"
#include "sparse_matrix_device.hpp"

namespace mfmg
{
// A_c = R A R^T on the local agglomerate grid `na` with `ne` rows per agglomerate: colours of period k on GLOBAL agglomerate
// coordinates (local + off), rows of the owned agglomerates [own0, own1) only; every neighbour inside the grid is stored.
// Y[colour][local row].
std::shared_ptr<SparseMatrixDevice<double>> galerkin_from_probes(HipHandle &h, int const na[3], int ne, int const k[3], int const off[3],
                                                                 int64_t const own0[3], int64_t const own1[3], double const *Y);

// P = (I - w D^-1 A) P_tent: Z[colour][local fine row] = (A y_colour)(row), t the tentative prolongator (ghosts exchanged),
// dinv the inverse diagonal; rows of the owned nodes of `fine` only, zeros are dropped.  `fine` / `coarse`: the two levels as
// halo spaces (local box, owned box, global position and size per axis; one rank: all three coincide).
std::shared_ptr<SparseMatrixDevice<double>> prolongator_from_probes(HipHandle &h, HaloSpace const &fine, HaloSpace const &coarse, int blk,
                                                                    int reach, int const period[3], double w, double const *Z,
                                                                    double const *t, double const *dinv, double const *Yp = nullptr);
// (Yp != nullptr: the smoothed prolongator of the cycle, P~ = (I - w D^-1 A) P, from probes with the columns of P themselves:
// Yp[colour][row] = (P e_colour)(row), Z = A Yp; `reach` = the reach of P~ in fine nodes, twice that of A)

// A_c = P^T A P: Y[colour][local coarse row] = (P^T A P u_colour)(row); rows of the owned nodes only, zeros are dropped.
std::shared_ptr<SparseMatrixDevice<double>> coarse_operator_from_probes(HipHandle &h, HaloSpace const &coarse, int reach, int const period[3],
                                                                        double const *Y);
// The assembled fine operator of a structured mesh (amge_structured.cpp: assemble_global_matrix) formed on the device:
// the same rows, bit for bit, sorted by column.
std::shared_ptr<SparseMatrixDevice<double>> fine_operator_on_device(HipHandle &h, StructuredMesh const &mesh, bool matrix_free_semantics);
} // namespace mfmg
