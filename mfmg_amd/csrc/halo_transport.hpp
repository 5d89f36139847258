// Transports of the slab halo exchange (common.hpp: HaloTransport).
//
// The reference's CUDA path has one communication primitive, an MPI all-gather of the whole vector through the
// host in front of every SpMV (source/cuda/utils.cu:363-482, all_gather_dev).  Here neighbours exchange boundary
// layers point to point:
//   * RcclTransport: ncclGroupStart; ncclSend/ncclRecv x <= 2 neighbours; ncclGroupEnd on the caller's HIP stream
//     (RCCL over xGMI, one process per GPU).  RCCL is resolved at run time (the copy torch has already loaded,
//     else /opt/rocm/lib/librccl.so.1), so the library itself does not link against it.
//   * HostTransport: the library stages the layers through pinned host buffers and calls the registered callbacks
//     (gloo through torch.distributed in the tests, where several ranks share one card).
//   * ReflectingTransport: measurement of one rank's share of a distributed cycle on one GPU (no wire).
#pragma once

#include "common.hpp"

namespace mfmg
{
// 128-byte RCCL unique id created on rank 0 (ncclGetUniqueId) and handed to every rank by the caller
void rccl_unique_id(unsigned char out[128]);
// resolves librccl and the entry points used (dlopen / dlsym only: no RCCL call); throws if that fails
void rccl_available();
std::shared_ptr<HaloTransport> make_rccl_transport(int rank, int n_ranks, unsigned char const unique_id[128]);
std::shared_ptr<HaloTransport> make_host_transport(int rank, int n_ranks, mfmg_hip_host_exchange_fn sendrecv,
                                                   mfmg_hip_host_allreduce_fn allreduce, mfmg_hip_host_allgather_fn allgather,
                                                   void *user);
// measurement: one rank of a grid on its own, every message reflected on the device (halo_transport.cpp)
std::shared_ptr<HaloTransport> make_reflecting_transport(int n_ranks, double delay_us = 0.);
} // namespace mfmg
