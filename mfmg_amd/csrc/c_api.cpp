// extern "C" boundary of libmfmg_hip.so (include/mfmg_hip.h): exceptions of the C++
// mirror are mapped to status codes; messages are kept per thread.
#include <cmath>
#include <cstring>
#include <exception>
#include <string>

#include "cell_contraction.hpp"
#include "halo_transport.hpp"
#include "mfmg/hierarchy.hpp"

using namespace mfmg;

namespace
{
thread_local std::string g_last_error;

template <typename F>
int guarded(F &&f)
{
  try
  {
    f();
    return MFMG_HIP_SUCCESS;
  }
  catch (NotImplementedExc const &e)
  {
    g_last_error = e.what();
    return MFMG_HIP_ERROR_NOT_IMPLEMENTED;
  }
  catch (InvalidArgumentExc const &e)
  {
    g_last_error = e.what();
    return MFMG_HIP_ERROR_INVALID_ARGUMENT;
  }
  catch (DeviceExc const &e)
  {
    g_last_error = e.what();
    return MFMG_HIP_ERROR_DEVICE;
  }
  catch (std::exception const &e)
  {
    g_last_error = e.what();
    return MFMG_HIP_ERROR_RUNTIME;
  }
  catch (...)
  {
    g_last_error = "unknown exception";
    return MFMG_HIP_ERROR_RUNTIME;
  }
}

void require(bool cond, char const *what)
{
  if (!cond)
    throw InvalidArgumentExc(what);
}
} // namespace

struct mfmg_hip_context_s
{
  std::unique_ptr<HipHandle> handle;
};

struct mfmg_hip_csr_s
{
  std::shared_ptr<HipMatrixOperator> op; // owns the SparseMatrixDevice
  bool borrowed = false;
};

struct mfmg_hip_mf_laplace_s
{
  std::shared_ptr<MatrixFreeLaplaceDevice<double>> op;
};

struct mfmg_hip_mf_laplace_f32_s
{
  std::shared_ptr<MatrixFreeLaplaceDevice<float>> op;
};

struct mfmg_hip_host_csr_s
{
  HostCsr m;
};

struct mfmg_hip_host_amg_s
{
  std::vector<AmgLevelHost> levels;
};

struct mfmg_hip_hierarchy_s
{
  HipHandle *handle = nullptr;
  std::shared_ptr<HipMeshEvaluator> evaluator;
  std::shared_ptr<TimerOutput> timer;
  std::unique_ptr<Hierarchy<DVector>> hierarchy;
  mfmg_hip_csr_s restrictor_view, coarse_view, amg_view, fine_view;
  // "fine level precision" float: the matrix-free operator and its smoother in FP32 around the FP64 coarse levels
  std::shared_ptr<MatrixFreeLaplaceDevice<float>> fine_f32;
  DeviceBuffer<float> f32_a, f32_b, f32_c, f32_res;
  bool setup_values_float = false; // "setup value precision" of THIS hierarchy: on the handle only while one of its setups runs
  std::shared_ptr<DVector> f32_res64, f32_bc, f32_xc, f32_corr64;
};

extern "C" {

const char *mfmg_hip_last_error(void) { return g_last_error.c_str(); }
const char *mfmg_hip_version(void) { return "mfmg-hip 0.3.0 (gfx950)"; }
int mfmg_hip_abi_version(void) { return MFMG_HIP_ABI_VERSION; }

int mfmg_hip_memory_inventory(char *buffer, size_t buffer_size)
{
  return guarded([&] {
    require(buffer != nullptr && buffer_size > 0, "null argument");
    const std::string text = DeviceMemoryLedger::get().report();
    std::snprintf(buffer, buffer_size, "%s", text.c_str());
  });
}

// ---- context -------------------------------------------------------------------
int mfmg_hip_context_create(void *hip_stream, mfmg_hip_context_t *ctx)
{
  return guarded([&] {
    require(ctx != nullptr, "null output handle");
    int n_dev = 0;
    hipError_t err = hipGetDeviceCount(&n_dev);
    if (err != hipSuccess || n_dev == 0)
      throw DeviceExc("no HIP device available: the mfmg HIP path has no CPU fallback");
    auto c = new mfmg_hip_context_s;
    const bool own = (hip_stream == MFMG_HIP_OWN_STREAM);
    c->handle.reset(new HipHandle(own ? nullptr : static_cast<hipStream_t>(hip_stream), own));
    *ctx = c;
  });
}

int mfmg_hip_context_destroy(mfmg_hip_context_t ctx)
{
  return guarded([&] { delete ctx; });
}

int mfmg_hip_context_synchronize(mfmg_hip_context_t ctx)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    MFMG_HIP_CHECK(hipStreamSynchronize(ctx->handle->stream));
  });
}

void *mfmg_hip_context_stream(mfmg_hip_context_t ctx) { return ctx ? ctx->handle->stream : nullptr; }

// ---- distributed runs -----------------------------------------------------------------
int mfmg_hip_context_set_communicator(mfmg_hip_context_t ctx, int32_t rank, int32_t n_ranks, int32_t ghost_cells_low,
                                      int32_t ghost_cells_high)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank out of range");
    require((ghost_cells_low == 0 || ghost_cells_low == 2 || ghost_cells_low == 4) && (ghost_cells_high == 0 || ghost_cells_high == 2),
            "ghost cell layers must be 0 or 2 (one agglomerate layer; 4 = two of them towards the lower neighbour)");
    require((ghost_cells_low > 0) == (rank > 0) && (ghost_cells_high == 2) == (rank + 1 < n_ranks),
            "ghost layers must be present exactly towards existing neighbours");
    HaloCommunicator &c = ctx->handle->comm;
    auto transport = c.transport;
    c = HaloCommunicator();
    c.transport = transport;
    c.rank = rank;
    c.n_ranks = n_ranks;
    c.ghost_cells_low = ghost_cells_low;
    c.ghost_cells_high = ghost_cells_high;
    c.grid[2] = n_ranks;
    c.coord[2] = rank;
    c.ghost_lo[2] = ghost_cells_low;
    c.ghost_hi[2] = ghost_cells_high;
  });
}

int mfmg_hip_context_set_communicator_box(mfmg_hip_context_t ctx, int32_t rank, const int32_t grid[3], const int32_t ghost_low[3],
                                          const int32_t ghost_high[3])
{
  return guarded([&] {
    require(ctx != nullptr && grid && ghost_low && ghost_high, "null argument");
    require(grid[0] >= 1 && grid[1] >= 1 && grid[2] >= 1, "the grid of ranks needs at least one rank per axis");
    const int n_ranks = grid[0] * grid[1] * grid[2];
    require(rank >= 0 && rank < n_ranks, "rank out of range");
    const int coord[3] = {rank % grid[0], (rank / grid[0]) % grid[1], rank / (grid[0] * grid[1])};
    for (int d = 0; d < 3; ++d)
    {
      require((ghost_low[d] == 0 || ghost_low[d] == 2 || ghost_low[d] == 4) && (ghost_high[d] == 0 || ghost_high[d] == 2),
              "ghost cell layers must be 0 or 2 (one agglomerate layer; 4 = two of them towards the lower neighbour)");
      require((ghost_low[d] > 0) == (coord[d] > 0) && (ghost_high[d] == 2) == (coord[d] + 1 < grid[d]),
              "ghost layers must be present exactly towards existing neighbours");
    }
    HaloCommunicator &c = ctx->handle->comm;
    auto transport = c.transport;
    c = HaloCommunicator();
    c.transport = transport;
    c.rank = rank;
    c.n_ranks = n_ranks;
    for (int d = 0; d < 3; ++d)
    {
      c.grid[d] = grid[d];
      c.coord[d] = coord[d];
      c.ghost_lo[d] = ghost_low[d];
      c.ghost_hi[d] = ghost_high[d];
    }
    c.ghost_cells_low = ghost_low[2];
    c.ghost_cells_high = ghost_high[2];
  });
}

int mfmg_hip_context_set_low_ghost_cells(mfmg_hip_context_t ctx, int32_t cells)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(cells == 2 || cells == 4, "ghost cell layers towards a lower neighbour: 2 or 4");
    HaloCommunicator &c = ctx->handle->comm;
    for (int d = 0; d < 3; ++d)
      require(c.ghost_lo[d] == 0 || c.ghost_lo[d] == cells, "the local mesh of this rank holds another number of ghost cell layers below");
    c.low_ghost_cells = cells;
  });
}

int mfmg_hip_rccl_available(void)
{
  return guarded([&] { rccl_available(); });
}

int mfmg_hip_rccl_unique_id(unsigned char out[128])
{
  return guarded([&] {
    require(out != nullptr, "null output");
    rccl_unique_id(out);
  });
}

int mfmg_hip_context_use_rccl(mfmg_hip_context_t ctx, const unsigned char unique_id[128])
{
  return guarded([&] {
    require(ctx != nullptr && unique_id != nullptr, "null argument");
    HaloCommunicator &c = ctx->handle->comm;
    c.transport = make_rccl_transport(c.rank, c.n_ranks, unique_id);
    // the exchange stream gets its communicator HERE, where every rank is (ncclCommSplit is collective; created lazily it
    // would be created wherever a rank first overlaps an exchange)
    if (ctx->handle->comm_stream != nullptr)
      c.transport->bind_exchange_stream(ctx->handle->comm_stream);
    else
      (void)ctx->handle->exchange_stream();
  });
}

int mfmg_hip_context_use_host_transport(mfmg_hip_context_t ctx, mfmg_hip_host_exchange_fn exchange,
                                        mfmg_hip_host_allreduce_fn allreduce, mfmg_hip_host_allgather_fn allgather, void *user)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(exchange && allreduce && allgather, "null transport callbacks");
    HaloCommunicator &c = ctx->handle->comm;
    c.transport = make_host_transport(c.rank, c.n_ranks, exchange, allreduce, allgather, user);
  });
}

int mfmg_hip_context_use_reflecting_transport(mfmg_hip_context_t ctx)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    HaloCommunicator &c = ctx->handle->comm;
    require(c.enabled(), "no communicator was set");
    c.transport = make_reflecting_transport(c.n_ranks);
  });
}

int mfmg_hip_context_use_reflecting_transport_delay(mfmg_hip_context_t ctx, double microseconds_per_group)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(microseconds_per_group >= 0. && microseconds_per_group <= 1e4, "delay out of range");
    HaloCommunicator &c = ctx->handle->comm;
    require(c.enabled(), "no communicator was set");
    c.transport = make_reflecting_transport(c.n_ranks, microseconds_per_group);
  });
}

// `reps` loop-back groups (send to self + receive from self) of n doubles back to back on the context's stream, bracketed by
// one event pair: microseconds of stream time per group -- what a grouped send/recv costs before any byte crosses a wire
int mfmg_hip_context_transport_loopback_time(mfmg_hip_context_t ctx, int64_t n, int reps, double *microseconds)
{
  return guarded([&] {
    require(ctx != nullptr && microseconds != nullptr && n > 0 && reps > 0, "bad argument");
    HipHandle &h = *ctx->handle;
    require(h.comm.transport != nullptr, "no transport registered");
    DeviceBuffer<double> a((size_t)n), b((size_t)n);
    MFMG_HIP_CHECK(hipMemsetAsync(a.data(), 0, (size_t)n * sizeof(double), h.stream));
    for (int i = 0; i < 3; ++i)
      h.comm.transport->loopback(a.data(), b.data(), n, h.stream);
    hipEvent_t e0, e1;
    MFMG_HIP_CHECK(hipEventCreate(&e0));
    MFMG_HIP_CHECK(hipEventCreate(&e1));
    MFMG_HIP_CHECK(hipEventRecord(e0, h.stream));
    for (int i = 0; i < reps; ++i)
      h.comm.transport->loopback(a.data(), b.data(), n, h.stream);
    MFMG_HIP_CHECK(hipEventRecord(e1, h.stream));
    MFMG_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    MFMG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *microseconds = 1e3 * ms / reps;
  });
}

int mfmg_hip_context_transport_name(mfmg_hip_context_t ctx, char *buffer, size_t buffer_size)
{
  return guarded([&] {
    require(ctx != nullptr && buffer != nullptr && buffer_size > 0, "null argument");
    std::string name = ctx->handle->comm.transport ? ctx->handle->comm.transport->name() : "";
    std::snprintf(buffer, buffer_size, "%s", name.c_str());
  });
}

int mfmg_hip_context_transport_ranks(mfmg_hip_context_t ctx, int *n_ranks)
{
  return guarded([&] {
    require(ctx != nullptr && n_ranks != nullptr, "null argument");
    *n_ranks = ctx->handle->comm.transport ? ctx->handle->comm.transport->comm_ranks() : 1;
  });
}

int mfmg_hip_context_transport_selftest(mfmg_hip_context_t ctx, int64_t n, double *max_error)
{
  return guarded([&] {
    require(ctx != nullptr && max_error != nullptr && n > 0, "bad argument");
    HipHandle &h = *ctx->handle;
    require(h.comm.transport != nullptr, "no transport registered");
    const int nr = h.comm.n_ranks, rk = h.comm.rank;
    std::vector<double> host((size_t)n);
    for (int64_t i = 0; i < n; ++i)
      host[i] = 1000. * rk + double(i % 977);
    DeviceBuffer<double> a((size_t)n), b((size_t)n), g((size_t)n * nr);
    MFMG_HIP_CHECK(hipMemcpyAsync(a.data(), host.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h.stream));
    MFMG_HIP_CHECK(hipMemsetAsync(b.data(), 0, (size_t)n * sizeof(double), h.stream));
    double err = 0.;
    h.comm.transport->loopback(a.data(), b.data(), n, h.stream);
    std::vector<double> back = b.download(h.stream);
    for (int64_t i = 0; i < n; ++i)
      err = std::max(err, std::abs(back[i] - host[i]));
    h.comm.transport->allgather(a.data(), n, g.data(), h.stream);
    std::vector<double> all = g.download(h.stream);
    for (int r = 0; r < nr; ++r)
      for (int64_t i = 0; i < n; ++i)
        err = std::max(err, std::abs(all[(size_t)r * n + i] - (1000. * r + double(i % 977))));
    double v[2] = {double(rk + 1), double(rk + 1)};
    h.comm.transport->allreduce(v, 1, 0, h.stream);
    h.comm.transport->allreduce(v + 1, 1, 1, h.stream);
    err = std::max(err, std::abs(v[0] - 0.5 * nr * (nr + 1)));
    err = std::max(err, std::abs(v[1] - double(nr)));
    *max_error = err;
  });
}

int mfmg_hip_context_exchange_count(mfmg_hip_context_t ctx, int64_t *n_exchanges)
{
  return guarded([&] {
    require(ctx != nullptr && n_exchanges != nullptr, "null argument");
    *n_exchanges = ctx->handle->comm.n_exchanges;
  });
}

int mfmg_hip_context_exchange_volume(mfmg_hip_context_t ctx, int64_t *n_doubles_sent, int64_t *n_overlapped)
{
  return guarded([&] {
    require(ctx != nullptr && n_doubles_sent != nullptr, "null argument");
    *n_doubles_sent = ctx->handle->comm.n_doubles_sent;
    if (n_overlapped)
      *n_overlapped = ctx->handle->comm.n_overlapped;
  });
}

int mfmg_hip_context_exchange(mfmg_hip_context_t ctx, int32_t space, double *vector, int reverse)
{
  return guarded([&] {
    require(ctx != nullptr && vector != nullptr, "null argument");
    if (reverse)
      ctx->handle->exchange_reverse_add(space, vector);
    else
      ctx->handle->exchange(space, vector);
  });
}

int mfmg_hip_context_owned_dot(mfmg_hip_context_t ctx, int32_t space, const double *x, const double *y, double *result)
{
  return guarded([&] {
    require(ctx != nullptr && x && y && result, "null argument");
    HipHandle &h = *ctx->handle;
    require(space > 0 && space < (int)h.comm.spaces.size() && h.comm.spaces[space].configured(), "unknown vector space");
    DVector vx(h, h.comm.spaces[space].n_local(), const_cast<double *>(x)), vy(h, h.comm.spaces[space].n_local(), const_cast<double *>(y));
    *result = distributed_dot(h, space, vx, vy);
  });
}

int mfmg_hip_context_set_cell_constant_layout(mfmg_hip_context_t ctx, int enable)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    ctx->handle->allow_cell_constant = enable != 0;
  });
}
int mfmg_hip_context_set_stored_diagonal(mfmg_hip_context_t ctx, int enable)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    ctx->handle->stored_diagonal = enable != 0;
  });
}
int mfmg_hip_context_set_mf_fused_terms(mfmg_hip_context_t ctx, int n_terms)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(n_terms >= 1 && n_terms <= 3, "1, 2 or 3 terms per sweep");
    ctx->handle->mf_fused_terms = n_terms;
  });
}
int mfmg_hip_context_set_mf_shell(mfmg_hip_context_t ctx, int mode)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(mode >= 0 && mode <= 2, "shell mode: 0 beside the interior tiles, 1 after them, 2 slab by slab");
    ctx->handle->mf_shell_mode = mode;
  });
}
int mfmg_hip_context_set_mf_emulate_split(mfmg_hip_context_t ctx, int axes)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(axes >= 0 && axes <= 3, "emulated split: 0 off, 1 = z, 2 = yz, 3 = xyz");
    ctx->handle->mf_emulate_split = axes;
  });
}
int mfmg_hip_mf_laplace_diagonal_in_record(mfmg_hip_mf_laplace_t op, int *in_record)
{
  return guarded([&] {
    require(op && in_record, "null argument");
    *in_record = op->op->diagonal_in_record() ? 1 : 0;
  });
}
int mfmg_hip_mf_laplace_ids_computed(mfmg_hip_mf_laplace_t op, int *computed)
{
  return guarded([&] {
    require(op && computed, "null argument");
    *computed = op->op->ids_computed() ? 1 : 0;
  });
}
int mfmg_hip_context_set_galerkin_on_device(mfmg_hip_context_t ctx, int enable)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    ctx->handle->galerkin_on_device = enable != 0;
  });
}

int mfmg_hip_mf_laplace_cell_constant_layout(mfmg_hip_mf_laplace_t op, int *in_use)
{
  return guarded([&] {
    require(op && in_use, "null argument");
    *in_use = op->op->cell_constant_layout() ? 1 : 0;
  });
}

int mfmg_hip_mf_laplace_f32_cell_constant_layout(mfmg_hip_mf_laplace_f32_t op, int *in_use)
{
  return guarded([&] {
    require(op && in_use, "null argument");
    *in_use = op->op->cell_constant_layout() ? 1 : 0;
  });
}

int mfmg_hip_mf_laplace_f32_ids_computed(mfmg_hip_mf_laplace_f32_t op, int *computed)
{
  return guarded([&] {
    require(op && computed, "null argument");
    *computed = op->op->ids_computed() ? 1 : 0;
  });
}

int mfmg_hip_context_set_overlap_exchange(mfmg_hip_context_t ctx, int enable)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    ctx->handle->overlap_exchange = enable != 0;
  });
}

int mfmg_hip_context_halo_layout(mfmg_hip_context_t ctx, int32_t space, int64_t *layer_elems, int64_t *n_layers,
                                 int64_t *owned_begin, int64_t *owned_count)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    require(space >= 1 && space < (int)ctx->handle->comm.spaces.size(), "unknown vector space");
    HaloSpace const &s = ctx->handle->comm.spaces[space];
    if (layer_elems)
      *layer_elems = s.layer_elems;
    if (n_layers)
      *n_layers = s.n_layers;
    if (owned_begin)
      *owned_begin = s.owned_begin;
    if (owned_count)
      *owned_count = s.owned_count;
  });
}

int mfmg_hip_context_halo_space(mfmg_hip_context_t ctx, int32_t space, int64_t out[8])
{
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "null argument");
    require(space >= 1 && space < (int)ctx->handle->comm.spaces.size(), "unknown vector space");
    HaloSpace const &s = ctx->handle->comm.spaces[space];
    out[0] = s.layer_elems;
    out[1] = s.n_layers;
    out[2] = s.owned_begin;
    out[3] = s.owned_count;
    out[4] = s.global_begin;
    out[5] = s.global_layers;
    out[6] = s.width;
    out[7] = (int64_t)ctx->handle->comm.spaces.size();
  });
}

int mfmg_hip_context_halo_box(mfmg_hip_context_t ctx, int32_t space, int64_t out[16])
{
  return guarded([&] {
    require(ctx != nullptr && out != nullptr, "null argument");
    require(space >= 1 && space < (int)ctx->handle->comm.spaces.size(), "unknown vector space");
    HaloSpace const &s = ctx->handle->comm.spaces[space];
    out[0] = s.comps;
    for (int d = 0; d < 3; ++d)
    {
      out[1 + d] = s.dim(d);
      out[4 + d] = s.own0(d);
      out[7 + d] = s.own_n(d);
      out[10 + d] = s.g0(d);
      out[13 + d] = s.gn(d);
    }
  });
}

int mfmg_hip_cell_contraction(mfmg_hip_context_t ctx, int fp32, int variant, int64_t n_cells, const void *u, const void *c,
                              void *v, const double cell_size[3])
{
  return guarded([&] {
    require(ctx != nullptr && u && c && v && cell_size, "null argument");
    if (fp32)
      cell_contraction<float>(*ctx->handle, variant, n_cells, static_cast<float const *>(u), static_cast<float const *>(c),
                              static_cast<float *>(v), cell_size);
    else
      cell_contraction<double>(*ctx->handle, variant, n_cells, static_cast<double const *>(u), static_cast<double const *>(c),
                               static_cast<double *>(v), cell_size);
  });
}

// ---- per-kernel HIP-event timing (bench.py roofline leg) ---------------------------
int mfmg_hip_profile_select(mfmg_hip_context_t ctx, const char *kernel_name)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    ctx->handle->profiler.only = kernel_name ? kernel_name : "";
  });
}

int mfmg_hip_profile_enable(mfmg_hip_context_t ctx, int enabled)
{
  return guarded([&] {
    require(ctx != nullptr, "null context");
    ctx->handle->profiler.enabled = enabled != 0;
    ctx->handle->profiler.reset();
  });
}

int mfmg_hip_profile_query(mfmg_hip_context_t ctx, const char *kernel_name, int64_t *n_launches, double *total_ms,
                           double *algorithmic_bytes)
{
  return guarded([&] {
    require(ctx && kernel_name, "null argument");
    int64_t n = 0;
    double ms = 0., bytes = 0.;
    ctx->handle->profiler.query(kernel_name, n, ms, bytes);
    if (n_launches)
      *n_launches = n;
    if (total_ms)
      *total_ms = ms;
    if (algorithmic_bytes)
      *algorithmic_bytes = bytes;
  });
}

// ---- marshalling ---------------------------------------------------------------
int mfmg_hip_malloc(void **dev_ptr, size_t bytes)
{
  return guarded([&] {
    require(dev_ptr != nullptr, "null output pointer");
    MFMG_HIP_CHECK(hipMalloc(dev_ptr, bytes));
  });
}
int mfmg_hip_free(void *dev_ptr)
{
  return guarded([&] { MFMG_HIP_CHECK(hipFree(dev_ptr)); });
}
int mfmg_hip_copy_to_dev(void *dst_dev, const void *src_host, size_t bytes)
{
  return guarded([&] { MFMG_HIP_CHECK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice)); });
}
int mfmg_hip_copy_to_host(void *dst_host, const void *src_dev, size_t bytes)
{
  return guarded([&] { MFMG_HIP_CHECK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); });
}

// ---- vector kernels ------------------------------------------------------------
int mfmg_hip_vector_set(mfmg_hip_context_t ctx, int64_t n, double value, double *x)
{
  return guarded([&] {
    require(ctx && x, "null argument");
    vec::set<double>(*ctx->handle, n, value, x);
  });
}
int mfmg_hip_vector_add(mfmg_hip_context_t ctx, int64_t n, double a, const double *v, double *x)
{
  return guarded([&] {
    require(ctx && x && v, "null argument");
    vec::add<double>(*ctx->handle, n, a, v, x);
  });
}
int mfmg_hip_vector_sadd(mfmg_hip_context_t ctx, int64_t n, double s, double a, const double *v, double *x)
{
  return guarded([&] {
    require(ctx && x && v, "null argument");
    vec::sadd<double>(*ctx->handle, n, s, a, v, x);
  });
}
int mfmg_hip_vector_dot(mfmg_hip_context_t ctx, int64_t n, const double *x, const double *y, double *result_host)
{
  return guarded([&] {
    require(ctx && x && y && result_host, "null argument");
    *result_host = vec::dot<double>(*ctx->handle, n, x, y);
  });
}
int mfmg_hip_vector_l2_norm(mfmg_hip_context_t ctx, int64_t n, const double *x, double *result_host)
{
  return guarded([&] {
    require(ctx && x && result_host, "null argument");
    *result_host = vec::l2_norm<double>(*ctx->handle, n, x);
  });
}

// ---- CSR -------------------------------------------------------------------------
int mfmg_hip_csr_create(mfmg_hip_context_t ctx, int64_t n_rows, int64_t n_cols, int64_t nnz,
                        const int32_t *row_ptr_host, const int32_t *col_host, const double *val_host,
                        mfmg_hip_csr_t *out)
{
  return guarded([&] {
    require(ctx && out && row_ptr_host, "null argument");
    require(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "negative size");
    require(nnz == 0 || (col_host && val_host), "null column / value array");
    std::vector<int32_t> rp(row_ptr_host, row_ptr_host + n_rows + 1);
    require(rp[n_rows] == nnz, "row_ptr[n_rows] != nnz");
    std::vector<int32_t> cl(col_host, col_host + nnz);
    std::vector<double> vl(val_host, val_host + nnz);
    auto m = std::make_shared<SparseMatrixDevice<double>>(*ctx->handle, n_rows, n_cols, std::move(rp), std::move(cl),
                                                          std::move(vl));
    auto h = new mfmg_hip_csr_s;
    h->op = std::make_shared<HipMatrixOperator>(m);
    *out = h;
  });
}

int mfmg_hip_csr_destroy(mfmg_hip_csr_t a)
{
  return guarded([&] {
    if (a && !a->borrowed)
      delete a;
  });
}

int mfmg_hip_csr_set_kernel(mfmg_hip_csr_t a, int lanes_per_row, int use_lds)
{
  return guarded([&] {
    require(a != nullptr, "null matrix");
    require(lanes_per_row >= 0 && (lanes_per_row <= 64 || lanes_per_row == 256) &&
                (lanes_per_row & (lanes_per_row - 1)) == 0,
            "lanes_per_row must be 0, a power of two up to 64, or 256 (a workgroup per row)");
    a->op->get_matrix()->set_kernel(lanes_per_row, use_lds);
  });
}
int mfmg_hip_csr_regular_rows(mfmg_hip_csr_t a, int *in_use)
{
  return guarded([&] {
    require(a && in_use, "null argument");
    *in_use = a->op->get_matrix()->regular_rows() ? 1 : 0;
  });
}
int mfmg_hip_csr_stencil_classes(mfmg_hip_csr_t a, int *n_classes, int64_t *listed_rows)
{
  return guarded([&] {
    require(a && n_classes && listed_rows, "null argument");
    *n_classes = a->op->get_matrix()->stencil_classes();
    *listed_rows = a->op->get_matrix()->listed_rows();
  });
}
int mfmg_hip_csr_float_storage(mfmg_hip_csr_t a, int *in_float)
{
  return guarded([&] {
    require(a && in_float, "null argument");
    auto op = a->op;
    // block-diagonal planes of the matrix, or -- for a restrictor with its agglomerate-wise form -- its planes
    *in_float = (op->get_matrix()->float_storage() || op->structured_float_planes()) ? 1 : 0;
  });
}
int mfmg_hip_csr_set_regular_rows(mfmg_hip_csr_t a, int enable)
{
  return guarded([&] {
    require(a != nullptr, "null matrix");
    a->op->get_matrix()->set_regular_rows(enable != 0);
  });
}
int mfmg_hip_csr_get_kernel(mfmg_hip_csr_t a, int *lanes_per_row, int *use_lds)
{
  return guarded([&] {
    require(a && lanes_per_row && use_lds, "null argument");
    *lanes_per_row = a->op->get_matrix()->lanes_per_row();
    *use_lds = a->op->get_matrix()->kernel_kind();
  });
}
int mfmg_hip_csr_shape(mfmg_hip_csr_t a, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
  return guarded([&] {
    require(a != nullptr, "null matrix");
    auto m = a->op->get_matrix();
    if (n_rows)
      *n_rows = m->m();
    if (n_cols)
      *n_cols = m->n();
    if (nnz)
      *nnz = m->n_nonzero_elements();
  });
}

int mfmg_hip_csr_vmult(mfmg_hip_csr_t a, const double *x, double *y)
{
  return guarded([&] {
    require(a && x && y, "null argument");
    a->op->get_matrix()->vmult(y, x);
  });
}

int mfmg_hip_csr_apply(mfmg_hip_csr_t a, const double *x, double *y, int mode)
{
  return guarded([&] {
    require(a && x && y, "null argument");
    require(mode == MFMG_HIP_NO_TRANS || mode == MFMG_HIP_TRANS, "unknown operator mode");
    if (mode == MFMG_HIP_NO_TRANS)
      a->op->get_matrix()->vmult(y, x);
    else
      a->op->get_transposed_matrix()->vmult(y, x);
  });
}

int mfmg_hip_csr_transpose(mfmg_hip_csr_t a, mfmg_hip_csr_t *out)
{
  return guarded([&] {
    require(a && out, "null argument");
    auto h = new mfmg_hip_csr_s;
    h->op = std::dynamic_pointer_cast<HipMatrixOperator>(a->op->transpose());
    *out = h;
  });
}

int mfmg_hip_csr_multiply(mfmg_hip_csr_t a, mfmg_hip_csr_t b, mfmg_hip_csr_t *out)
{
  return guarded([&] {
    require(a && b && out, "null argument");
    auto h = new mfmg_hip_csr_s;
    h->op = std::dynamic_pointer_cast<HipMatrixOperator>(a->op->multiply(b->op));
    *out = h;
  });
}

int mfmg_hip_csr_download(mfmg_hip_csr_t a, int32_t *row_ptr_host, int32_t *col_host, double *val_host)
{
  return guarded([&] {
    require(a && row_ptr_host, "null argument");
    std::vector<int32_t> rp, cl;
    std::vector<double> vl;
    a->op->get_matrix()->download(rp, cl, vl);
    std::memcpy(row_ptr_host, rp.data(), rp.size() * sizeof(int32_t));
    if (!cl.empty())
    {
      require(col_host && val_host, "null column / value array");
      std::memcpy(col_host, cl.data(), cl.size() * sizeof(int32_t));
      std::memcpy(val_host, vl.data(), vl.size() * sizeof(double));
    }
  });
}

int mfmg_hip_csr_solve(mfmg_hip_csr_t a, const char *params_info, const double *b, double *x)
{
  return guarded([&] {
    require(a && b && x, "null argument");
    auto m = a->op->get_matrix();
    require(m->m() == m->n(), "the solver needs a square matrix");
    auto params = std::make_shared<ptree>(ptree::parse_info(params_info ? params_info : ""));
    HipSolver solver(m->handle(), a->op, params);
    DVector bv(m->handle(), m->m(), const_cast<double *>(b)), xv(m->handle(), m->m(), x);
    solver.apply(bv, xv);
    MFMG_HIP_CHECK(hipStreamSynchronize(m->handle().stream)); // (the solver and its factors go out of scope)
  });
}

int mfmg_hip_csr_inverse_diagonal(mfmg_hip_csr_t a, double *dinv)
{
  return guarded([&] {
    require(a && dinv, "null argument");
    a->op->get_matrix()->inverse_diagonal(dinv);
  });
}

int mfmg_hip_csr_smoother_step(mfmg_hip_csr_t a, const double *dinv, const double *b, const double *x,
                               const double *x_prev, double alpha, double beta, double *out)
{
  return guarded([&] {
    require(a && dinv && b && x && out, "null argument");
    auto m = a->op->get_matrix();
    require(m->m() == m->n(), "the smoother needs a square matrix");
    m->smoother_step(dinv, b, x, x_prev, alpha, beta, out);
  });
}

int mfmg_hip_csr_residual(mfmg_hip_csr_t a, const double *x, const double *b, double *res)
{
  return guarded([&] {
    require(a && x && b && res, "null argument");
    a->op->get_matrix()->residual(x, b, res);
  });
}

// ---- matrix-free Laplace -----------------------------------------------------------
int mfmg_hip_mf_laplace_create(mfmg_hip_context_t ctx, const mfmg_hip_mesh_desc *mesh, mfmg_hip_mf_laplace_t *out)
{
  return guarded([&] {
    require(ctx && mesh && out, "null argument");
    auto h = new mfmg_hip_mf_laplace_s;
    try
    {
      h->op = std::make_shared<MatrixFreeLaplaceDevice<double>>(*ctx->handle, *mesh, ctx->handle->allow_cell_constant);
    }
    catch (...)
    {
      delete h;
      throw;
    }
    *out = h;
  });
}

int mfmg_hip_mf_laplace_destroy(mfmg_hip_mf_laplace_t op)
{
  return guarded([&] { delete op; });
}

int mfmg_hip_mf_laplace_size(mfmg_hip_mf_laplace_t op, int64_t *n_dofs)
{
  return guarded([&] {
    require(op && n_dofs, "null argument");
    *n_dofs = op->op->n_dofs();
  });
}

int mfmg_hip_mf_laplace_vmult(mfmg_hip_mf_laplace_t op, const double *x, double *y)
{
  return guarded([&] {
    require(op && x && y, "null argument");
    op->op->vmult(x, y);
  });
}

int mfmg_hip_mf_laplace_diagonal_inverse(mfmg_hip_mf_laplace_t op, double *dinv)
{
  return guarded([&] {
    require(op && dinv, "null argument");
    vec::copy<double>(op->op->handle(), op->op->n_dofs(), op->op->diagonal_inverse(), dinv);
  });
}

int mfmg_hip_mf_laplace_diagonal(mfmg_hip_mf_laplace_t op, double *diag)
{
  return guarded([&] {
    require(op && diag, "null argument");
    vec::copy<double>(op->op->handle(), op->op->n_dofs(), op->op->diagonal(), diag);
  });
}

int mfmg_hip_mf_laplace_residual(mfmg_hip_mf_laplace_t op, const double *x, const double *b, double *res)
{
  return guarded([&] {
    require(op && x && b && res, "null argument");
    op->op->residual(x, b, res);
  });
}

int mfmg_hip_mf_laplace_smoother_step(mfmg_hip_mf_laplace_t op, const double *b, const double *x,
                                      const double *x_prev, double alpha, double beta, double *out)
{
  return guarded([&] {
    require(op && b && x && out, "null argument");
    op->op->smoother_step(b, x, x_prev, alpha, beta, out);
  });
}

int mfmg_hip_mf_laplace_sweep_available(mfmg_hip_mf_laplace_t op, int n_terms, int *available)
{
  return guarded([&] {
    require(op && available, "null argument");
    *available = op->op->fused_sweep_available(n_terms) ? 1 : 0;
  });
}
int mfmg_hip_mf_laplace_smoother_sweep(mfmg_hip_mf_laplace_t op, int n_terms, const double *alpha, const double *beta,
                                       const double *b, const double *x, double *out, double *out_prev)
{
  return guarded([&] {
    require(op && alpha && beta && b && out, "null argument");
    if (!op->op->fused_sweep_available(n_terms))
      ASSERT_THROW_NOT_IMPLEMENTED("the multi-term smoother sweep is not available for this operator");
    // x == NULL: the sweep from x_0 = 0, which is then not read (three terms, default arithmetic)
    if (x == nullptr && !op->op->fused_zero_guess_available(n_terms))
      ASSERT_THROW_NOT_IMPLEMENTED("the sweep from a zero guess is not available for this operator / tile / number of terms");
    op->op->smoother_sweep(n_terms, alpha, beta, b, x, out, out_prev);
  });
}
int mfmg_hip_mf_laplace_set_sweep_reference(mfmg_hip_mf_laplace_t op, int on)
{
  return guarded([&] {
    require(op != nullptr, "null operator");
    op->op->set_fused_reference(on != 0);
  });
}
int mfmg_hip_mf_laplace_f32_set_sweep_reference(mfmg_hip_mf_laplace_f32_t op, int on)
{
  return guarded([&] {
    require(op != nullptr, "null operator");
    op->op->set_fused_reference(on != 0);
  });
}
int mfmg_hip_mf_laplace_set_sweep_tile(mfmg_hip_mf_laplace_t op, int n_waves, int tile_y, int tile_z)
{
  return guarded([&] {
    require(op != nullptr, "null operator");
    require(n_waves >= 0 && n_waves <= 8 && tile_y >= 0 && tile_y <= 4 && tile_z >= 0 && tile_z <= 4096, "sweep tile out of range");
    op->op->set_fused_tile(n_waves, tile_y, tile_z);
  });
}
int mfmg_hip_mf_laplace_get_sweep_tile(mfmg_hip_mf_laplace_t op, int n_terms, int *n_waves, int *tile_y, int *tile_z)
{
  return guarded([&] {
    require(op && n_waves && tile_y && tile_z, "null argument");
    op->op->get_fused_tile(n_terms, *n_waves, *tile_y, *tile_z);
  });
}

// ---- FP32 instance ----------------------------------------------------------------------
int mfmg_hip_mf_laplace_f32_create(mfmg_hip_context_t ctx, const mfmg_hip_mesh_desc *mesh, mfmg_hip_mf_laplace_f32_t *out)
{
  return guarded([&] {
    require(ctx && mesh && out, "null argument");
    std::unique_ptr<mfmg_hip_mf_laplace_f32_s> h(new mfmg_hip_mf_laplace_f32_s);
    h->op = std::make_shared<MatrixFreeLaplaceDevice<float>>(*ctx->handle, *mesh, ctx->handle->allow_cell_constant);
    *out = h.release();
  });
}
int mfmg_hip_mf_laplace_f32_destroy(mfmg_hip_mf_laplace_f32_t op)
{
  return guarded([&] { delete op; });
}
int mfmg_hip_mf_laplace_f32_vmult(mfmg_hip_mf_laplace_f32_t op, const float *x, float *y)
{
  return guarded([&] {
    require(op && x && y, "null argument");
    op->op->vmult(x, y);
  });
}
int mfmg_hip_mf_laplace_f32_diagonal_inverse(mfmg_hip_mf_laplace_f32_t op, float *dinv)
{
  return guarded([&] {
    require(op && dinv, "null argument");
    vec::copy<float>(op->op->handle(), op->op->n_dofs(), op->op->diagonal_inverse(), dinv);
  });
}
int mfmg_hip_mf_laplace_f32_residual(mfmg_hip_mf_laplace_f32_t op, const float *x, const float *b, float *res)
{
  return guarded([&] {
    require(op && x && b && res, "null argument");
    op->op->residual(x, b, res);
  });
}
int mfmg_hip_mf_laplace_f32_smoother_step(mfmg_hip_mf_laplace_f32_t op, const float *b, const float *x,
                                          const float *x_prev, float alpha, float beta, float *out)
{
  return guarded([&] {
    require(op && b && x && out, "null argument");
    op->op->smoother_step(b, x, x_prev, alpha, beta, out);
  });
}

int mfmg_hip_mf_laplace_f32_sweep_available(mfmg_hip_mf_laplace_f32_t op, int n_terms, int *available)
{
  return guarded([&] {
    require(op && available, "null argument");
    *available = op->op->fused_sweep_available(n_terms) ? 1 : 0;
  });
}
int mfmg_hip_mf_laplace_f32_smoother_sweep(mfmg_hip_mf_laplace_f32_t op, int n_terms, const float *alpha, const float *beta,
                                           const float *b, const float *x, float *out, float *out_prev)
{
  return guarded([&] {
    require(op && alpha && beta && b && x && out, "null argument");
    if (!op->op->fused_sweep_available(n_terms))
      ASSERT_THROW_NOT_IMPLEMENTED("the multi-term smoother sweep is not available for this operator");
    op->op->smoother_sweep(n_terms, alpha, beta, b, x, out, out_prev);
  });
}

int mfmg_hip_mf_laplace_get_tile(mfmg_hip_mf_laplace_t op, int *n_waves, int *tile_y, int *tile_z)
{
  return guarded([&] {
    require(op && n_waves && tile_y && tile_z, "null argument");
    op->op->get_tile(*n_waves, *tile_y, *tile_z);
  });
}
int mfmg_hip_mf_laplace_set_tile_waves(mfmg_hip_mf_laplace_t op, int n_waves)
{
  return guarded([&] {
    require(op != nullptr, "null operator");
    require(n_waves >= 0 && n_waves <= 8, "0..8 wavefronts per workgroup");
    op->op->set_tile_waves(n_waves);
  });
}
int mfmg_hip_mf_laplace_set_tile(mfmg_hip_mf_laplace_t op, int tile_y, int tile_z)
{
  return guarded([&] {
    require(op != nullptr, "null argument");
    require(tile_y >= 0 && tile_z >= 0 && tile_y <= 64 && tile_z <= 1024, "tile size out of range");
    op->op->set_tile(tile_y, tile_z);
  });
}

// ---- hierarchy -----------------------------------------------------------------------
int mfmg_hip_hierarchy_create(mfmg_hip_context_t ctx, const char *evaluator_type, const mfmg_hip_mesh_desc *mesh,
                              const char *params_info, mfmg_hip_hierarchy_t *out)
{
  return guarded([&] {
    require(ctx && evaluator_type && mesh && out, "null argument");
    auto params = std::make_shared<ptree>(ptree::parse_info(params_info ? params_info : ""));
    std::unique_ptr<mfmg_hip_hierarchy_s> h(new mfmg_hip_hierarchy_s);
    h->handle = ctx->handle.get();
    std::string type(evaluator_type);
    if (type == "HipMatrixFreeMeshEvaluator")
      h->evaluator = std::make_shared<HipMatrixFreeMeshEvaluator>(*ctx->handle, *mesh);
    else if (type == "HipMeshEvaluator")
      h->evaluator = std::make_shared<HipMeshEvaluator>(*ctx->handle, *mesh);
    else
      ASSERT_THROW_NOT_IMPLEMENTED("mesh evaluator type \"" + type + "\" is not available in the HIP build");
    h->timer = std::make_shared<TimerOutput>();
    {
      // "setup value precision" reaches the probing kernels through the handle (HipHierarchyHelpers::build_restrictor sets it):
      // it must not outlive this setup, or the next hierarchy / user-built operator of the context inherits it
      struct Restore
      {
        HipHandle &hd;
        bool &keep;
        ~Restore()
        {
          keep = hd.setup_values_float;
          hd.setup_values_float = false;
        }
      } restore{*ctx->handle, h->setup_values_float};
      ctx->handle->setup_values_float = false;
      h->hierarchy.reset(new Hierarchy<DVector>(nullptr, h->evaluator, params, h->timer));
    }
    // "release setup matrices" true: once the hierarchy stands, the table-driven operators (A_c, the operators of the aggregation
    // levels) free their CSR arrays -- 12 B per entry that only the setup algebra and the exports (get_coarse_operator, the
    // level matrices) read: those calls throw afterwards.  One rank.
    if (params->get("release setup matrices", false))
    {
      require(!ctx->handle->comm.enabled(), "\"release setup matrices\" is not available in a distributed run");
      auto const &lv = h->hierarchy->levels();
      for (size_t l = 1; l < lv.size(); ++l)
        if (auto op = std::dynamic_pointer_cast<HipMatrixOperator const>(lv[l].get_operator()))
          op->get_matrix()->release_csr();
      if (auto solver = std::dynamic_pointer_cast<HipSolver const>(lv.back().get_solver()))
        std::const_pointer_cast<HipSolver>(solver)->release_setup_matrices();
    }
    const std::string precision = params->get("fine level precision", "double");
    if (precision == "float")
    {
      // built here: the mesh arrays of the caller need not outlive this call
      require(type == "HipMatrixFreeMeshEvaluator", "\"fine level precision\" float needs the matrix-free evaluator");
      require(h->hierarchy->levels().size() == 2, "\"fine level precision\" float needs the two-level hierarchy");
      require(!ctx->handle->comm.enabled(), "\"fine level precision\" float is not available in a distributed run");
      h->fine_f32 = std::make_shared<MatrixFreeLaplaceDevice<float>>(*ctx->handle, *mesh, ctx->handle->allow_cell_constant);
    }
    else
      require(precision == "double", "\"fine level precision\" must be double or float");
    MFMG_HIP_CHECK(hipStreamSynchronize(ctx->handle->stream));
    *out = h.release();
  });
}

int mfmg_hip_hierarchy_destroy(mfmg_hip_hierarchy_t h)
{
  return guarded([&] { delete h; });
}

namespace
{
int64_t level_size(mfmg_hip_hierarchy_t h, int level)
{
  auto const &levels = h->hierarchy->levels();
  require(level >= 0 && level < (int)levels.size(), "level out of range");
  auto op = levels[level].get_operator();
  if (auto m = std::dynamic_pointer_cast<HipMatrixOperator const>(op))
    return m->get_matrix()->m();
  return (int64_t)op->grid_complexity();
}
} // namespace

int mfmg_hip_hierarchy_apply(mfmg_hip_hierarchy_t h, const double *b, double *x)
{
  return guarded([&] {
    require(h && b && x, "null argument");
    const int64_t n = level_size(h, 0);
    DVector bv(*h->handle, n, const_cast<double *>(b)), xv(*h->handle, n, x);
    h->hierarchy->apply(bv, xv);
  });
}

// Hierarchy::apply (hierarchy.hpp:246-309) with the fine level in FP32: pre-smoother, residual and post-smoother run
// on float vectors through the FP32 instance of the matrix-free operator (same polynomial coefficients as the FP64
// smoother), the residual is widened, restricted, solved for and prolongated in FP64, and the correction is
// subtracted from the float iterate.
int mfmg_hip_hierarchy_apply_f32(mfmg_hip_hierarchy_t h, const float *b, float *x)
{
  return guarded([&] {
    require(h && b && x, "null argument");
    require(h->fine_f32 != nullptr, "the hierarchy was not built with \"fine level precision\" float");
    HipHandle &hd = *h->handle;
    auto const &levels = h->hierarchy->levels();
    const int64_t n = level_size(h, 0), nc = level_size(h, 1);
    auto smoother = std::dynamic_pointer_cast<HipSmoother const>(levels[0].get_smoother());
    require(smoother != nullptr, "unexpected smoother type");
    auto const &coef = smoother->coefficients();
    const int d = (int)coef.size();
    if (h->f32_a.size() == 0)
    {
      h->f32_a.resize(n);
      h->f32_b.resize(n);
      h->f32_res.resize(n);
      h->f32_res64 = levels[0].get_operator()->build_range_vector();
      h->f32_corr64 = levels[0].get_operator()->build_range_vector();
    }
    if (!h->f32_bc)
    {
      h->f32_bc = levels[1].get_operator()->build_range_vector();
      h->f32_xc = levels[1].get_operator()->build_range_vector();
    }
    auto const &op = *h->fine_f32;
    if (h->f32_c.size() == 0)
      h->f32_c.resize(n);
    // x_out <- x_in - B^-1 (A x_in - b) on two different vectors: the first K terms of the polynomial in one sweep where the
    // FP32 operator offers it (all of them for degree <= smoother.fused_terms), else one fused kernel per term
    const int fused = smoother->fused_terms();
    auto smooth_to = [&](float const *x_in, float *x_out) {
      float const *cur = x_in, *prev = nullptr;
      int k0 = 0;
      const int K = std::min(d, fused);
      if (K >= 2 && coef[0].first == 0. && op.fused_sweep_available(K))
      {
        float alpha[3], beta[3];
        for (int k = 0; k < K; ++k)
        {
          alpha[k] = (float)coef[k].first;
          beta[k] = (float)coef[k].second;
        }
        if (K == d)
        {
          op.smoother_sweep(K, alpha, beta, b, x_in, x_out, nullptr);
          return;
        }
        op.smoother_sweep(K, alpha, beta, b, x_in, h->f32_a.data(), h->f32_b.data());
        cur = h->f32_a.data();
        prev = h->f32_b.data();
        k0 = K;
      }
      for (int k = k0; k < d; ++k)
      {
        // any scratch vector that is not x_k; x_{k-1} may be overwritten in place unless it is the caller's x_in
        float *target = x_out;
        if (k + 1 < d)
        {
          target = h->f32_a.data();
          if (target == cur)
            target = h->f32_b.data();
        }
        op.smoother_step(b, cur, prev, (float)coef[k].first, (float)coef[k].second, target);
        prev = cur;
        cur = target;
      }
    };
    // the iterate alternates between x and a work vector, so that no application ends in a copy
    float *it = x, *other = h->f32_c.data();
    auto smooth = [&]() {
      smooth_to(it, other);
      std::swap(it, other);
    };
    if (h->hierarchy->is_preconditioner())
      MFMG_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(float) * n, hd.stream));
    for (unsigned int i = 0; i < h->hierarchy->n_smoothing_steps(); ++i)
      smooth();
    auto restrictor = levels[1].get_restrictor();
    auto hip_restrictor = std::dynamic_pointer_cast<HipMatrixOperator const>(restrictor);
    // b_c = R (A x - b): one pass over the FP32 vectors where the restrictor holds the rows of R A, otherwise the FP32
    // residual, widened, and the restriction
    if (!(hip_restrictor && hip_restrictor->restrict_residual_f32(*levels[0].get_operator(), it, b, *h->f32_bc)))
    {
      op.residual(it, b, h->f32_res.data());
      vec::widen(hd, n, h->f32_res.data(), h->f32_res64->get_values());
      restrictor->apply(*h->f32_res64, *h->f32_bc);
    }
    h->hierarchy->apply(*h->f32_bc, *h->f32_xc, 1);
    restrictor->apply(*h->f32_xc, *h->f32_corr64, OperatorMode::TRANS);
    vec::subtract_narrowed(hd, n, h->f32_corr64->get_values(), it);
    for (unsigned int i = 0; i < h->hierarchy->n_smoothing_steps(); ++i)
      smooth();
    if (it != x)
      MFMG_HIP_CHECK(hipMemcpyAsync(x, it, sizeof(float) * n, hipMemcpyDeviceToDevice, hd.stream));
    (void)nc;
  });
}

int mfmg_hip_hierarchy_vmult(mfmg_hip_hierarchy_t h, double *x, const double *b)
{
  return guarded([&] {
    require(h && b && x, "null argument");
    const int64_t n = level_size(h, 0);
    DVector bv(*h->handle, n, const_cast<double *>(b)), xv(*h->handle, n, x);
    h->hierarchy->vmult(xv, bv);
  });
}

int mfmg_hip_hierarchy_solve_cg(mfmg_hip_hierarchy_t h, const double *b, double *x, double tolerance,
                                int32_t max_iterations, int32_t *n_iterations, double *final_residual,
                                double *residual_history, int32_t history_len)
{
  return guarded([&] {
    require(h && b && x, "null argument");
    require(tolerance >= 0. && max_iterations >= 0, "bad stopping criterion");
    // preconditioned CG as dealii::SolverCG runs it (third-party, restated): r = b - A x, z = M^-1 r, p = z;
    // alpha = (r,z)/(p,Ap); x += alpha p; r -= alpha Ap; stop on ||r||_2 <= tolerance; beta = (r,z)_new/(r,z)_old
    HipHandle &handle = *h->handle;
    const int64_t n = level_size(h, 0);
    auto op = h->hierarchy->levels()[0].get_operator();
    auto hop = std::dynamic_pointer_cast<HipOperator const>(op);
    const int space = hop ? hop->domain_space() : 0;
    DVector bv(handle, n, const_cast<double *>(b)), xv(handle, n, x);
    DVector r(handle, n), z(handle, n), p(handle, n), ap(handle, n);
    auto dot = [&](DVector const &u, DVector const &v) { return distributed_dot(handle, space, u, v); };
    op->apply(xv, r);
    r.sadd(-1., 1., bv); // r = b - A x
    double res = std::sqrt(dot(r, r));
    int it = 0;
    auto record = [&](int k, double v) {
      if (residual_history && k < history_len)
        residual_history[k] = v;
    };
    record(0, res);
    double rz = 0.;
    bool converged = res <= tolerance;
    if (!handle.comm.enabled())
    {
      // One rank: the scalars of the iteration stay on the device (vec::cg_direction / cg_update read them there) -- ONE host
      // round trip per iteration, for the stopping test, instead of three; x and r are updated in one pass.  The same sums in the
      // same order as below: the same iterates.
      DeviceBuffer<double> scal(4); // [0], [1]: (r, z) of this / the previous iteration, alternating; [2]: (p, A p); [3]: (r, r)
      int cur = 0;
      while (!converged && it < max_iterations)
      {
        h->hierarchy->vmult(z, r);
        vec::dot_async<double>(handle, n, r.get_values(), z.get_values(), scal.data(), cur);
        if (it == 0)
          p = z;
        else
          vec::cg_direction<double>(handle, n, z.get_values(), p.get_values(), scal.data(), cur, 1 - cur); // p = z + beta p
        op->apply(p, ap);
        vec::dot_async<double>(handle, n, p.get_values(), ap.get_values(), scal.data(), 2);
        vec::cg_update<double>(handle, n, p.get_values(), ap.get_values(), nullptr, xv.get_values(), r.get_values(), nullptr, scal.data(), cur, 2);
        vec::dot_async<double>(handle, n, r.get_values(), r.get_values(), scal.data(), 3);
        MFMG_HIP_CHECK(hipMemcpyAsync(handle.host_result, scal.data() + 3, sizeof(double), hipMemcpyDeviceToHost, handle.stream));
        MFMG_HIP_CHECK(hipStreamSynchronize(handle.stream));
        res = std::sqrt(handle.host_result[0]);
        cur = 1 - cur;
        ++it;
        record(it, res);
        converged = res <= tolerance;
      }
    }
    while (!converged && it < max_iterations)
    {
      h->hierarchy->vmult(z, r);
      const double rz_new = dot(r, z);
      if (it == 0)
        p = z;
      else
        p.sadd(rz_new / rz, 1., z); // p = z + beta p
      rz = rz_new;
      op->apply(p, ap);
      const double alpha = rz / dot(p, ap);
      xv.add(alpha, p);
      r.add(-alpha, ap);
      res = std::sqrt(dot(r, r));
      ++it;
      record(it, res);
      converged = res <= tolerance;
    }
    if (n_iterations)
      *n_iterations = it;
    if (final_residual)
      *final_residual = res;
    if (!converged)
      throw std::runtime_error("CG did not reach the tolerance within max_iterations (SolverControl::NoConvergence)");
  });
}

int mfmg_hip_hierarchy_n_levels(mfmg_hip_hierarchy_t h, int32_t *n_levels)
{
  return guarded([&] {
    require(h && n_levels, "null argument");
    *n_levels = (int32_t)h->hierarchy->levels().size();
  });
}

int mfmg_hip_hierarchy_level_size(mfmg_hip_hierarchy_t h, int32_t level, int64_t *n)
{
  return guarded([&] {
    require(h && n, "null argument");
    *n = level_size(h, level);
  });
}

int mfmg_hip_hierarchy_operator_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *x, double *y, int mode)
{
  return guarded([&] {
    require(h && x && y, "null argument");
    require(mode == MFMG_HIP_NO_TRANS || mode == MFMG_HIP_TRANS, "unknown operator mode");
    const int64_t n = level_size(h, level);
    DVector xv(*h->handle, n, const_cast<double *>(x)), yv(*h->handle, n, y);
    h->hierarchy->levels()[level].get_operator()->apply(
        xv, yv, mode == MFMG_HIP_TRANS ? OperatorMode::TRANS : OperatorMode::NO_TRANS);
  });
}

int mfmg_hip_hierarchy_smoother_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *b, double *x)
{
  return guarded([&] {
    require(h && b && x, "null argument");
    const int64_t n = level_size(h, level);
    auto smoother = h->hierarchy->levels()[level].get_smoother();
    require(smoother != nullptr, "this level has no smoother");
    DVector bv(*h->handle, n, const_cast<double *>(b)), xv(*h->handle, n, x);
    smoother->apply(bv, xv);
  });
}

int mfmg_hip_hierarchy_restrictor_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *in, double *out,
                                        int mode)
{
  return guarded([&] {
    require(h && in && out, "null argument");
    require(mode == MFMG_HIP_NO_TRANS || mode == MFMG_HIP_TRANS, "unknown operator mode");
    require(level >= 1 && level < (int)h->hierarchy->levels().size(), "restrictors live on levels >= 1");
    const int64_t n_fine = level_size(h, level - 1), n_coarse = level_size(h, level);
    auto r = h->hierarchy->levels()[level].get_restrictor();
    if (mode == MFMG_HIP_NO_TRANS)
    {
      DVector iv(*h->handle, n_fine, const_cast<double *>(in)), ov(*h->handle, n_coarse, out);
      r->apply(iv, ov, OperatorMode::NO_TRANS);
    }
    else
    {
      DVector iv(*h->handle, n_coarse, const_cast<double *>(in)), ov(*h->handle, n_fine, out);
      r->apply(iv, ov, OperatorMode::TRANS);
    }
  });
}

int mfmg_hip_hierarchy_ap_apply(mfmg_hip_hierarchy_t h, int32_t level, const double *in, double *out)
{
  return guarded([&] {
    require(h && in && out, "null argument");
    auto const &aps = h->hierarchy->ap_operators();
    require(level >= 1 && level <= (int)aps.size(), "no A R^T kept for this level (build the hierarchy with keep_ap = true)");
    DVector iv(*h->handle, level_size(h, level), const_cast<double *>(in)), ov(*h->handle, level_size(h, level - 1), out);
    aps[level - 1]->apply(iv, ov);
  });
}

int mfmg_hip_hierarchy_residual_restriction_classes(mfmg_hip_hierarchy_t h, int32_t level, int32_t *n_classes)
{
  return guarded([&] {
    require(h && n_classes, "null argument");
    require(level >= 1 && level < (int)h->hierarchy->levels().size(), "restrictors live on levels >= 1");
    auto r = std::dynamic_pointer_cast<HipMatrixOperator const>(h->hierarchy->levels()[level].get_restrictor());
    *n_classes = (r && r->has_residual_restriction()) ? (int32_t)r->residual_restriction_classes() : 0;
  });
}

int mfmg_hip_hierarchy_restrict_residual(mfmg_hip_hierarchy_t h, int32_t level, const double *x, const double *b, double *b_coarse)
{
  return guarded([&] {
    require(h && x && b && b_coarse, "null argument");
    require(level >= 1 && level < (int)h->hierarchy->levels().size(), "restrictors live on levels >= 1");
    const int64_t n_fine = level_size(h, level - 1), n_coarse = level_size(h, level);
    auto r = h->hierarchy->levels()[level].get_restrictor();
    auto a = h->hierarchy->levels()[level - 1].get_operator();
    DVector xv(*h->handle, n_fine, const_cast<double *>(x)), bv(*h->handle, n_fine, const_cast<double *>(b)),
        bc(*h->handle, n_coarse, b_coarse);
    if (!r->restrict_residual(*a, xv, bv, bc))
    {
      // the two steps of hierarchy.hpp:281-290
      DVector res(*h->handle, n_fine);
      a->residual(xv, bv, res);
      r->apply(res, bc);
    }
  });
}

int mfmg_hip_hierarchy_coarse_apply(mfmg_hip_hierarchy_t h, const double *b, double *x)
{
  return guarded([&] {
    require(h && b && x, "null argument");
    const int last = (int)h->hierarchy->levels().size() - 1;
    const int64_t n = level_size(h, last);
    DVector bv(*h->handle, n, const_cast<double *>(b)), xv(*h->handle, n, x);
    h->hierarchy->levels()[last].get_solver()->apply(bv, xv);
  });
}

int mfmg_hip_hierarchy_set_restrictor(mfmg_hip_hierarchy_t h, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                      const int32_t *row_ptr_host, const int32_t *col_host, const double *val_host)
{
  return guarded([&] {
    require(h && row_ptr_host && col_host && val_host, "null argument");
    require(n_rows >= 1 && n_cols >= 1 && nnz >= 0 && nnz < (int64_t(1) << 31), "matrix shape out of range");
    require(n_cols == level_size(h, 0), "the restrictor must have one column per fine DoF");
    require(!h->handle->comm.enabled(), "set_restrictor is not available in a distributed run (the restrictor carries "
                                        "the halo layout of the coarse space)");
    require(row_ptr_host[0] == 0 && row_ptr_host[n_rows] == nnz, "row_ptr[0] != 0 or row_ptr[n_rows] != nnz");
    std::vector<int32_t> rp(row_ptr_host, row_ptr_host + n_rows + 1);
    std::vector<int32_t> cl(col_host, col_host + nnz);
    std::vector<double> vl(val_host, val_host + nnz);
    auto m = std::make_shared<SparseMatrixDevice<double>>(*h->handle, n_rows, n_cols, std::move(rp), std::move(cl),
                                                          std::move(vl));
    {
      // the Galerkin product and the coarse solver are set up again, in the value precision this hierarchy was built with
      struct Restore
      {
        HipHandle &hd;
        ~Restore() { hd.setup_values_float = false; }
      } restore{*h->handle};
      h->handle->setup_values_float = h->setup_values_float;
      h->hierarchy->set_restrictor(std::make_shared<HipMatrixOperator>(m));
    }
    // scratch vectors of apply_f32 were sized for the old coarse space
    h->f32_bc.reset();
    h->f32_xc.reset();
    MFMG_HIP_CHECK(hipStreamSynchronize(h->handle->stream));
  });
}

int mfmg_hip_hierarchy_get_restrictor(mfmg_hip_hierarchy_t h, mfmg_hip_csr_t *r_borrowed)
{
  return guarded([&] {
    require(h && r_borrowed, "null argument");
    require(h->hierarchy->levels().size() >= 2, "the hierarchy has a single level");
    auto r = std::dynamic_pointer_cast<HipMatrixOperator const>(h->hierarchy->levels()[1].get_restrictor());
    require(r != nullptr, "the restrictor is not a matrix operator");
    h->restrictor_view.op = std::const_pointer_cast<HipMatrixOperator>(r);
    h->restrictor_view.borrowed = true;
    *r_borrowed = &h->restrictor_view;
  });
}

int mfmg_hip_hierarchy_get_fine_operator(mfmg_hip_hierarchy_t h, mfmg_hip_csr_t *a_borrowed)
{
  return guarded([&] {
    require(h && a_borrowed, "null argument");
    auto a = std::dynamic_pointer_cast<HipMatrixOperator const>(h->hierarchy->levels().front().get_operator());
    require(a != nullptr, "the fine operator is matrix-free");
    h->fine_view.op = std::const_pointer_cast<HipMatrixOperator>(a);
    h->fine_view.borrowed = true;
    *a_borrowed = &h->fine_view;
  });
}

int mfmg_hip_hierarchy_get_coarse_operator(mfmg_hip_hierarchy_t h, mfmg_hip_csr_t *ac_borrowed)
{
  return guarded([&] {
    require(h && ac_borrowed, "null argument");
    require(h->hierarchy->levels().size() >= 2, "the hierarchy has a single level");
    auto a = std::dynamic_pointer_cast<HipMatrixOperator const>(h->hierarchy->levels().back().get_operator());
    require(a != nullptr, "the coarse operator is not a matrix operator");
    h->coarse_view.op = std::const_pointer_cast<HipMatrixOperator>(a);
    h->coarse_view.borrowed = true;
    *ac_borrowed = &h->coarse_view;
  });
}

namespace
{
HipSolver const *coarse_solver_of(mfmg_hip_hierarchy_t h)
{
  require(h != nullptr, "null argument");
  auto s = std::dynamic_pointer_cast<HipSolver const>(h->hierarchy->levels().back().get_solver());
  require(s != nullptr, "the hierarchy has no HIP coarse solver");
  return s.get();
}
} // namespace

int mfmg_hip_hierarchy_coarse_amg_levels(mfmg_hip_hierarchy_t h, int32_t *n_levels)
{
  return guarded([&] {
    require(n_levels != nullptr, "null argument");
    *n_levels = (int32_t)coarse_solver_of(h)->amg_levels().size();
  });
}

int mfmg_hip_hierarchy_coarse_amg_gather_level(mfmg_hip_hierarchy_t h, int32_t *level)
{
  return guarded([&] {
    require(level != nullptr, "null argument");
    *level = (int32_t)coarse_solver_of(h)->amg_gather_level();
  });
}

int mfmg_hip_hierarchy_coarse_amg_get(mfmg_hip_hierarchy_t h, int32_t level, int32_t which, mfmg_hip_csr_t *borrowed)
{
  return guarded([&] {
    require(borrowed != nullptr, "null argument");
    auto const &lv = coarse_solver_of(h)->amg_levels();
    require(level >= 0 && level < (int)lv.size(), "level out of range");
    require(which >= 0 && which <= 2, "which must be 0 (A), 1 (P) or 2 (P^T as stored for the restriction)");
    auto op = which == 0 ? lv[level].a : which == 1 ? lv[level].prolongator : lv[level].restrictor;
    require(op != nullptr, "the last level has no transfer operators");
    h->amg_view.op = op;
    h->amg_view.borrowed = true;
    *borrowed = &h->amg_view;
  });
}

int mfmg_hip_hierarchy_coarse_amg_smoother(mfmg_hip_hierarchy_t h, int32_t level, int32_t *degree, double *lambda_min,
                                           double *lambda_max)
{
  return guarded([&] {
    auto const &lv = coarse_solver_of(h)->amg_levels();
    require(level >= 0 && level + 1 < (int)lv.size(), "level out of range (the last level is solved directly)");
    auto const &s = lv[level].smoother;
    if (degree)
      *degree = s->degree();
    if (lambda_min)
      *lambda_min = s->lambda_min();
    if (lambda_max)
      *lambda_max = s->lambda_max();
  });
}

int mfmg_hip_hierarchy_smoother_info(mfmg_hip_hierarchy_t h, int32_t *degree, double *lambda_min, double *lambda_max)
{
  return guarded([&] {
    require(h != nullptr, "null argument");
    auto s = std::dynamic_pointer_cast<HipSmoother const>(h->hierarchy->levels()[0].get_smoother());
    require(s != nullptr, "level 0 has no HIP smoother");
    if (degree)
      *degree = s->degree();
    if (lambda_min)
      *lambda_min = s->lambda_min();
    if (lambda_max)
      *lambda_max = s->lambda_max();
  });
}

int mfmg_hip_hierarchy_smoother_sweep_terms(mfmg_hip_hierarchy_t h, int *terms_in_place, int *terms_out_of_place)
{
  return guarded([&] {
    require(h != nullptr && terms_in_place != nullptr && terms_out_of_place != nullptr, "null argument");
    auto s = std::dynamic_pointer_cast<HipSmoother const>(h->hierarchy->levels()[0].get_smoother());
    require(s != nullptr, "level 0 has no HIP smoother");
    s->sweep_terms(*terms_in_place, *terms_out_of_place);
  });
}

int mfmg_hip_hierarchy_operator_tile(mfmg_hip_hierarchy_t h, int *n_waves, int *tile_y, int *tile_z)
{
  return guarded([&] {
    require(h && n_waves && tile_y && tile_z, "null argument");
    auto op = std::dynamic_pointer_cast<HipMatrixFreeOperator const>(h->hierarchy->levels()[0].get_operator());
    require(op != nullptr, "the fine-level operator is not matrix-free");
    op->get_mesh_evaluator()->get_device_operator()->get_tile(*n_waves, *tile_y, *tile_z);
  });
}

int mfmg_hip_hierarchy_sweep_tile(mfmg_hip_hierarchy_t h, int n_terms, int *n_waves, int *tile_y, int *tile_z)
{
  return guarded([&] {
    require(h && n_waves && tile_y && tile_z, "null argument");
    auto op = std::dynamic_pointer_cast<HipMatrixFreeOperator const>(h->hierarchy->levels()[0].get_operator());
    require(op != nullptr, "the fine-level operator is not matrix-free");
    op->get_mesh_evaluator()->get_device_operator()->get_fused_tile(n_terms, *n_waves, *tile_y, *tile_z);
  });
}

int mfmg_hip_hierarchy_set_sweep_tile(mfmg_hip_hierarchy_t h, int n_waves, int tile_y, int tile_z)
{
  return guarded([&] {
    require(h != nullptr, "null argument");
    require(n_waves >= 0 && n_waves <= 8 && tile_y >= 0 && tile_y <= 4 && tile_z >= 0, "bad tile");
    auto op = std::dynamic_pointer_cast<HipMatrixFreeOperator const>(h->hierarchy->levels()[0].get_operator());
    require(op != nullptr, "the fine-level operator is not matrix-free");
    op->get_mesh_evaluator()->get_device_operator()->set_fused_tile(n_waves, tile_y, tile_z);
  });
}

int mfmg_hip_hierarchy_set_operator_tile(mfmg_hip_hierarchy_t h, int n_waves, int tile_y, int tile_z)
{
  return guarded([&] {
    require(h != nullptr, "null argument");
    require(n_waves >= 0 && n_waves <= 8 && tile_y >= 0 && tile_z >= 0, "bad tile");
    auto op = std::dynamic_pointer_cast<HipMatrixFreeOperator const>(h->hierarchy->levels()[0].get_operator());
    require(op != nullptr, "the fine-level operator is not matrix-free");
    auto dev = op->get_mesh_evaluator()->get_device_operator();
    dev->set_tile(tile_y, tile_z);
    dev->set_tile_waves(n_waves);
  });
}

int mfmg_hip_hierarchy_timer_report(mfmg_hip_hierarchy_t h, char *buf, size_t buf_size)
{
  return guarded([&] {
    require(h && buf && buf_size > 0, "null argument");
    std::string s = h->timer->summary();
    std::strncpy(buf, s.c_str(), buf_size - 1);
    buf[buf_size - 1] = '\0';
  });
}

// ---- host-side setup pieces -------------------------------------------------------------
extern "C++"
{
namespace
{
StructuredMesh host_mesh(const mfmg_hip_mesh_desc *mesh)
{
  require(mesh != nullptr, "null mesh");
  require(mesh->arrays_on_device == 0, "the host entry points need host arrays");
  return StructuredMesh::from_desc(*mesh, nullptr);
}
} // namespace
}

int mfmg_hip_host_csr_shape(mfmg_hip_host_csr_t m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
  return guarded([&] {
    require(m != nullptr, "null matrix");
    if (n_rows)
      *n_rows = m->m.n_rows;
    if (n_cols)
      *n_cols = m->m.n_cols;
    if (nnz)
      *nnz = m->m.nnz();
  });
}

int mfmg_hip_host_csr_get(mfmg_hip_host_csr_t m, int32_t *row_ptr, int32_t *col, double *val)
{
  return guarded([&] {
    require(m && row_ptr, "null argument");
    std::memcpy(row_ptr, m->m.row_ptr.data(), m->m.row_ptr.size() * sizeof(int32_t));
    if (m->m.nnz() > 0)
    {
      require(col && val, "null column / value array");
      std::memcpy(col, m->m.col.data(), m->m.col.size() * sizeof(int32_t));
      std::memcpy(val, m->m.val.data(), m->m.val.size() * sizeof(double));
    }
  });
}

int mfmg_hip_host_csr_destroy(mfmg_hip_host_csr_t m)
{
  return guarded([&] { delete m; });
}

int mfmg_hip_host_assemble_matrix(const mfmg_hip_mesh_desc *mesh, int semantics, mfmg_hip_host_csr_t *out)
{
  return guarded([&] {
    require(out != nullptr, "null output handle");
    require(semantics == 0 || semantics == 1, "unknown constraint semantics");
    auto sm = host_mesh(mesh);
    auto h = new mfmg_hip_host_csr_s;
    h->m = assemble_global_matrix(sm, semantics == 0 ? ConstraintSemantics::assembled
                                                     : ConstraintSemantics::matrix_free);
    *out = h;
  });
}

int mfmg_hip_host_build_restrictor(const mfmg_hip_mesh_desc *mesh, const char *params_info, int matrix_free,
                                   mfmg_hip_host_csr_t *out)
{
  return guarded([&] {
    require(out != nullptr, "null output handle");
    auto sm = host_mesh(mesh);
    ptree params = ptree::parse_info(params_info ? params_info : "");
    RestrictorOptions o;
    o.agglomerate[0] = params.get("agglomeration.nx", 2);
    o.agglomerate[1] = params.get("agglomeration.ny", 2);
    o.agglomerate[2] = params.get("agglomeration.nz", 2);
    o.n_eigenvectors = params.get("eigensolver.number of eigenvectors", 1);
    o.variant = params.get("eigensolver.variant", matrix_free ? "mf" : "device");
    o.selection = params.get("eigensolver.selection", "krylov");
    o.use_coefficient = params.get("eigensolver.use_coefficient", true);
    auto diag = operator_diagonal(sm, matrix_free ? ConstraintSemantics::matrix_free : ConstraintSemantics::assembled);
    auto h = new mfmg_hip_host_csr_s;
    h->m = build_restrictor_structured(sm, diag, o);
    *out = h;
  });
}

int mfmg_hip_host_galerkin(const mfmg_hip_mesh_desc *mesh, int semantics, int64_t n_rows, int64_t nnz,
                           const int32_t *r_row_ptr, const int32_t *r_col, const double *r_val,
                           mfmg_hip_host_csr_t *out)
{
  return guarded([&] {
    require(out && r_row_ptr && r_col && r_val, "null argument");
    require(semantics == 0 || semantics == 1, "unknown constraint semantics");
    auto sm = host_mesh(mesh);
    HostCsr R, Rt;
    R.n_rows = n_rows;
    R.n_cols = sm.n_dofs;
    R.row_ptr.assign(r_row_ptr, r_row_ptr + n_rows + 1);
    require(R.row_ptr[n_rows] == nnz, "row_ptr[n_rows] != nnz");
    R.col.assign(r_col, r_col + nnz);
    R.val.assign(r_val, r_val + nnz);
    Rt.n_rows = R.n_cols;
    Rt.n_cols = R.n_rows;
    csr_transpose_host<double>(R.n_rows, R.n_cols, R.row_ptr, R.col, R.val, Rt.row_ptr, Rt.col, Rt.val);
    auto h = new mfmg_hip_host_csr_s;
    h->m = galerkin_triple_product(
        sm, semantics == 0 ? ConstraintSemantics::assembled : ConstraintSemantics::matrix_free, R, Rt);
    *out = h;
  });
}

int mfmg_hip_host_amg_build(int64_t n_rows, int64_t nnz, const int32_t *row_ptr, const int32_t *col, const double *val,
                            const double *near_null, const int32_t *grid_dims, const int32_t *node_of_row,
                            const int32_t *component_of_row, const char *params_info, mfmg_hip_host_amg_t *out)
{
  return guarded([&] {
    require(row_ptr && col && val && out, "null argument");
    HostCsr A;
    A.n_rows = A.n_cols = n_rows;
    A.row_ptr.assign(row_ptr, row_ptr + n_rows + 1);
    require(A.row_ptr[n_rows] == nnz, "row_ptr[n_rows] != nnz");
    A.col.assign(col, col + nnz);
    A.val.assign(val, val + nnz);
    std::vector<double> b0 = near_null ? std::vector<double>(near_null, near_null + n_rows)
                                       : std::vector<double>(n_rows, 1.);
    ptree params = ptree::parse_info(params_info ? params_info : "");
    AmgOptions opts;
    opts.max_levels = params.get("solver.amg.max_levels", 10);
    opts.coarsest_size = params.get("solver.amg.coarsest_size", 1100);
    opts.strength = params.get("solver.amg.strength", 0.08);
    opts.smooth_prolongator = params.get("solver.amg.smooth_prolongator", true);
    AmgGridHint grid;
    if (grid_dims && node_of_row)
    {
      for (int d = 0; d < 3; ++d)
        grid.dims[d] = grid_dims[d];
      grid.node_of_row.assign(node_of_row, node_of_row + n_rows);
      const int blk = params.get("solver.amg.aggregate_block", 2);
      require(blk >= 2 && blk <= 8, "solver.amg.aggregate_block must be in 2..8");
      for (int d = 0; d < 3; ++d)
        grid.block[d] = blk;
      if (component_of_row)
      {
        grid.component_of_row.assign(component_of_row, component_of_row + n_rows);
        for (auto c : grid.component_of_row)
          grid.n_components = std::max(grid.n_components, c + 1);
      }
    }
    auto h = new mfmg_hip_host_amg_s;
    h->levels = build_aggregation_hierarchy(std::move(A), std::move(b0), opts, grid.valid(n_rows) ? &grid : nullptr);
    *out = h;
  });
}

int mfmg_hip_host_amg_n_levels(mfmg_hip_host_amg_t amg, int32_t *n_levels)
{
  return guarded([&] {
    require(amg && n_levels, "null argument");
    *n_levels = (int32_t)amg->levels.size();
  });
}

int mfmg_hip_host_amg_get(mfmg_hip_host_amg_t amg, int32_t level, int32_t which, mfmg_hip_host_csr_t *out)
{
  return guarded([&] {
    require(amg && out, "null argument");
    require(level >= 0 && level < (int)amg->levels.size(), "level out of range");
    require(which == 0 || which == 1, "which must be 0 (A) or 1 (P)");
    auto h = new mfmg_hip_host_csr_s;
    h->m = which == 0 ? amg->levels[level].A : amg->levels[level].P;
    *out = h;
  });
}

int mfmg_hip_host_amg_destroy(mfmg_hip_host_amg_t amg)
{
  return guarded([&] { delete amg; });
}

int mfmg_hip_host_params_get(const char *params_info, const char *path, char *value_buf, size_t buf_size)
{
  return guarded([&] {
    require(params_info && path && value_buf && buf_size > 0, "null argument");
    ptree params = ptree::parse_info(params_info);
    std::string v = params.get<std::string>(path);
    std::strncpy(value_buf, v.c_str(), buf_size - 1);
    value_buf[buf_size - 1] = '\0';
  });
}

} // extern "C"
