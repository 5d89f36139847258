// Batched agglomerate eigenproblems of the spectral AMGe restrictor on the device (amge_device.hip).
#pragma once

#include "amge_structured.hpp"

namespace mfmg
{
// agglomerates of at most 64 nodes (one lane of a wavefront per node)
bool amge_device_supported(StructuredMesh const &mesh, RestrictorOptions const &opts);
// weights[(a * n_eig + e) * nmax + l] = diag_loc[l] * (eigenvector e of agglomerate a)[l], n_vec[a] vectors selected
void amge_device_eigen(HipHandle &handle, StructuredMesh const &mesh, RestrictorOptions const &opts, int const cnt[3],
                       std::vector<double> &weights, std::vector<int32_t> &n_vec, int &nmax);
} // namespace mfmg
