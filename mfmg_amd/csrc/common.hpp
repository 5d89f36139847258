// Shared plumbing of libmfmg_hip: error conventions, device buffers, launch helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mfmg_hip.h"

namespace mfmg
{
// ---- error conventions of include/mfmg/common/exceptions.hpp:33-73 ----
// ASSERT_THROW throws std::runtime_error always; NotImplementedExc for
// unsupported paths; HIP status checks are always on (the reference's
// ASSERT_CUDA is debug-only, exceptions.hpp:193-217 -- a faulting kernel on
// this pool is too expensive to leave unchecked).
class NotImplementedExc : public std::exception
{
public:
  explicit NotImplementedExc(std::string what = "The function is not implemented")
      : _what(std::move(what))
  {
  }
  const char *what() const noexcept override { return _what.c_str(); }

private:
  std::string _what;
};

class InvalidArgumentExc : public std::runtime_error
{
public:
  using std::runtime_error::runtime_error;
};

class DeviceExc : public std::runtime_error
{
public:
  using std::runtime_error::runtime_error;
};

inline void ASSERT_THROW(bool cond, std::string const &message)
{
  if (!cond)
    throw std::runtime_error(message);
}

[[noreturn]] inline void ASSERT_THROW_NOT_IMPLEMENTED(std::string const &what = "")
{
  throw NotImplementedExc(what.empty() ? "The function is not implemented" : what);
}

inline void ASSERT_HIP(hipError_t err, const char *file, int line)
{
  if (err != hipSuccess)
    throw DeviceExc(std::string("HIP error: ") + hipGetErrorString(err) + " at " + file + ":" +
                    std::to_string(line));
}
#define MFMG_HIP_CHECK(expr) ::mfmg::ASSERT_HIP((expr), __FILE__, __LINE__)

// block_size of include/mfmg/cuda/utils.cuh:35 is 512 (warp-32 era); 256 = 4 waves of 64.
constexpr int block_size = 256;

inline unsigned int n_blocks_for(int64_t n, int bs = block_size, int64_t cap = 1 << 20)
{
  int64_t nb = (n + bs - 1) / bs;
  if (nb < 1)
    nb = 1;
  if (nb > cap)
    nb = cap;
  return static_cast<unsigned int>(nb);
}

// ---- owning device buffer (cuda_malloc/cuda_free, include/mfmg/cuda/utils.cuh:66-99) ----
template <typename T>
class DeviceBuffer
{
public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n) { resize(n); }
  DeviceBuffer(DeviceBuffer const &) = delete;
  DeviceBuffer &operator=(DeviceBuffer const &) = delete;
  DeviceBuffer(DeviceBuffer &&o) noexcept : _ptr(o._ptr), _n(o._n)
  {
    o._ptr = nullptr;
    o._n = 0;
  }
  DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
  {
    if (this != &o)
    {
      release();
      _ptr = o._ptr;
      _n = o._n;
      o._ptr = nullptr;
      o._n = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }

  void resize(size_t n)
  {
    release();
    if (n > 0)
    {
      MFMG_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&_ptr), n * sizeof(T)));
      _n = n;
    }
  }
  void release()
  {
    if (_ptr)
      (void)hipFree(_ptr);
    _ptr = nullptr;
    _n = 0;
  }
  void upload(T const *host, size_t n, hipStream_t stream = nullptr)
  {
    if (n != _n)
      resize(n);
    if (n)
    {
      MFMG_HIP_CHECK(hipMemcpyAsync(_ptr, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
      MFMG_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
  std::vector<T> download(hipStream_t stream = nullptr) const
  {
    std::vector<T> h(_n);
    if (_n)
    {
      MFMG_HIP_CHECK(hipMemcpyAsync(h.data(), _ptr, _n * sizeof(T), hipMemcpyDeviceToHost, stream));
      MFMG_HIP_CHECK(hipStreamSynchronize(stream));
    }
    return h;
  }
  T *data() { return _ptr; }
  T const *data() const { return _ptr; }
  size_t size() const { return _n; }

private:
  T *_ptr = nullptr;
  size_t _n = 0;
};

// ---- per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) ----
// Off by default; when enabled every instrumented launch is bracketed by an event pair.
struct KernelProfiler
{
  struct Entry
  {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t used = 0;
    double algorithmic_bytes = 0.;
  };
  bool enabled = false;
  std::string only; // when not empty: time launches of this kernel name only (two event records cost ~5 us)
  std::map<std::string, Entry> entries;

  ~KernelProfiler()
  {
    for (auto &kv : entries)
      for (auto &ev : kv.second.events)
      {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
      }
  }
  void reset()
  {
    for (auto &kv : entries)
    {
      kv.second.used = 0;
      kv.second.algorithmic_bytes = 0.;
    }
  }
  // returns the stop event to record after the launch (nullptr when disabled)
  hipEvent_t begin(char const *name, double bytes, hipStream_t stream)
  {
    if (!enabled || (!only.empty() && only != name))
      return nullptr;
    Entry &e = entries[name];
    if (e.used == e.events.size())
    {
      hipEvent_t a, b;
      MFMG_HIP_CHECK(hipEventCreate(&a));
      MFMG_HIP_CHECK(hipEventCreate(&b));
      e.events.emplace_back(a, b);
    }
    auto &ev = e.events[e.used++];
    e.algorithmic_bytes += bytes;
    MFMG_HIP_CHECK(hipEventRecord(ev.first, stream));
    return ev.second;
  }
  static void end(hipEvent_t stop, hipStream_t stream)
  {
    if (stop)
      MFMG_HIP_CHECK(hipEventRecord(stop, stream));
  }
  // synchronises; total milliseconds and launches of one kernel since the last reset
  void query(std::string const &name, int64_t &launches, double &total_ms, double &bytes)
  {
    launches = 0;
    total_ms = 0.;
    bytes = 0.;
    auto it = entries.find(name);
    if (it == entries.end())
      return;
    for (size_t i = 0; i < it->second.used; ++i)
    {
      MFMG_HIP_CHECK(hipEventSynchronize(it->second.events[i].second));
      float ms = 0.f;
      MFMG_HIP_CHECK(hipEventElapsedTime(&ms, it->second.events[i].first, it->second.events[i].second));
      total_ms += ms;
    }
    launches = (int64_t)it->second.used;
    bytes = it->second.algorithmic_bytes;
  }
};

// ---- distributed vectors: slab decomposition along the slowest index ----
// The reference gets its ghost exchange from deal.II's distributed::Vector inside
// MatrixFree::cell_loop / Epetra Import (and all-gathers the whole vector on the CUDA path,
// source/cuda/utils.cu:363-482).  Here a rank's vector is [ghost layers | owned layers | ghost layers],
// layers contiguous; exchange() refreshes the one ghost layer on each side that operator applications
// read, through staging buffers and a caller-provided transport (RCCL send/recv via torch.distributed).
struct HaloSpace
{
  int64_t layer_elems = 0;   // entries per layer (a DoF plane, or a layer of agglomerates)
  int64_t n_layers = 0;      // layers of the local vector
  int64_t owned_begin = 0;   // first owned layer
  int64_t owned_count = 0;   // owned layers
  bool has_low = false, has_high = false;
  double *send_low = nullptr, *send_high = nullptr, *recv_low = nullptr, *recv_high = nullptr; // device staging
  int64_t staging_elems = 0;
  bool configured() const { return layer_elems > 0; }
};

struct HaloCommunicator
{
  int rank = 0, n_ranks = 1;
  int ghost_cells_low = 0, ghost_cells_high = 0; // ghost cell layers of the local mesh along z
  int (*exchange_fn)(void *user, int space, void *stream) = nullptr;
  int (*allreduce_fn)(void *user, double *values, int n) = nullptr;
  void *user = nullptr;
  HaloSpace spaces[3]; // [1] fine DoFs, [2] first coarse level
  bool enabled() const { return n_ranks > 1; }
};

// ---- HipHandle: stream + reduction scratch; twin of CudaHandle
//      (include/mfmg/cuda/cuda_handle.cuh:25-48): borrowed by every object built from it ----
struct HipHandle
{
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // second stream of the overlapped halo exchange (created at first use) and its two events
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_unpacked = nullptr;
  bool overlap_exchange = true;
  // matrix-free operators built from this handle may keep one coefficient per cell when a cell's eight are equal
  bool allow_cell_constant = true;
  // R A R^T of a matrix-free A by probing on the device (hip_hierarchy.hip, HipMatrixOperator::multiply) instead of
  // the host triple product
  bool galerkin_on_device = true;
  // scratch for two-stage deterministic reductions
  DeviceBuffer<double> reduce_partials;
  DeviceBuffer<double> reduce_result;
  double *host_result = nullptr; // pinned
  KernelProfiler profiler;
  HaloCommunicator comm;

  // refresh the ghost layers of a distributed vector (no-op on one rank / for local spaces)
  void exchange(int space, double *v)
  {
    if (!comm.enabled() || space <= 0)
      return;
    HaloSpace &s = comm.spaces[space];
    if (!s.configured())
      throw std::runtime_error("halo exchange requested for an unconfigured vector space");
    if (s.staging_elems < s.layer_elems || comm.exchange_fn == nullptr)
      throw std::runtime_error("halo staging buffers / transport were not registered");
    const size_t bytes = (size_t)s.layer_elems * sizeof(double);
    if (s.has_low)
      MFMG_HIP_CHECK(hipMemcpyAsync(s.send_low, v + s.owned_begin * s.layer_elems, bytes, hipMemcpyDeviceToDevice, stream));
    if (s.has_high)
      MFMG_HIP_CHECK(hipMemcpyAsync(s.send_high, v + (s.owned_begin + s.owned_count - 1) * s.layer_elems, bytes,
                                    hipMemcpyDeviceToDevice, stream));
    if (comm.exchange_fn(comm.user, space, stream) != 0)
      throw std::runtime_error("halo exchange transport failed");
    if (s.has_low)
      MFMG_HIP_CHECK(hipMemcpyAsync(v + (s.owned_begin - 1) * s.layer_elems, s.recv_low, bytes, hipMemcpyDeviceToDevice, stream));
    if (s.has_high)
      MFMG_HIP_CHECK(hipMemcpyAsync(v + (s.owned_begin + s.owned_count) * s.layer_elems, s.recv_high, bytes,
                                    hipMemcpyDeviceToDevice, stream));
  }
  // The same exchange split in two, so that work which does not read the ghost layers can run in between
  // (north_star: "halo exchange ... overlapped with interior smoothing on a second HIP stream"):
  //   begin: the boundary layers are packed on `stream`; `comm_stream` waits for the packing, runs the
  //          transport and unpacks into the ghost layers;
  //   end:   `stream` waits for the unpacking.
  // Between the two calls `stream` must not read the ghost layers of `v` nor write its boundary layers.
  void exchange_begin(int space, double *v)
  {
    if (!comm.enabled() || space <= 0)
      return;
    HaloSpace &s = comm.spaces[space];
    if (!s.configured())
      throw std::runtime_error("halo exchange requested for an unconfigured vector space");
    if (s.staging_elems < s.layer_elems || comm.exchange_fn == nullptr)
      throw std::runtime_error("halo staging buffers / transport were not registered");
    if (comm_stream == nullptr)
    {
      MFMG_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
      MFMG_HIP_CHECK(hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming));
      MFMG_HIP_CHECK(hipEventCreateWithFlags(&ev_unpacked, hipEventDisableTiming));
    }
    const size_t bytes = (size_t)s.layer_elems * sizeof(double);
    if (s.has_low)
      MFMG_HIP_CHECK(hipMemcpyAsync(s.send_low, v + s.owned_begin * s.layer_elems, bytes, hipMemcpyDeviceToDevice, stream));
    if (s.has_high)
      MFMG_HIP_CHECK(hipMemcpyAsync(s.send_high, v + (s.owned_begin + s.owned_count - 1) * s.layer_elems, bytes,
                                    hipMemcpyDeviceToDevice, stream));
    MFMG_HIP_CHECK(hipEventRecord(ev_packed, stream));
    MFMG_HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_packed, 0));
    if (comm.exchange_fn(comm.user, space, comm_stream) != 0)
      throw std::runtime_error("halo exchange transport failed");
    if (s.has_low)
      MFMG_HIP_CHECK(hipMemcpyAsync(v + (s.owned_begin - 1) * s.layer_elems, s.recv_low, bytes, hipMemcpyDeviceToDevice,
                                    comm_stream));
    if (s.has_high)
      MFMG_HIP_CHECK(hipMemcpyAsync(v + (s.owned_begin + s.owned_count) * s.layer_elems, s.recv_high, bytes,
                                    hipMemcpyDeviceToDevice, comm_stream));
    MFMG_HIP_CHECK(hipEventRecord(ev_unpacked, comm_stream));
  }
  void exchange_end(int space)
  {
    if (!comm.enabled() || space <= 0)
      return;
    MFMG_HIP_CHECK(hipStreamWaitEvent(stream, ev_unpacked, 0));
  }
  double allreduce_sum(double v)
  {
    if (!comm.enabled())
      return v;
    if (comm.allreduce_fn == nullptr || comm.allreduce_fn(comm.user, &v, 1) != 0)
      throw std::runtime_error("all-reduce transport failed");
    return v;
  }

  // `s` is borrowed (nullptr = the legacy default stream the reference runs on); with
  // `create_own` the handle creates and owns a non-blocking stream instead.
  explicit HipHandle(hipStream_t s, bool create_own = false)
  {
    if (create_own)
    {
      MFMG_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      owns_stream = true;
    }
    else
      stream = s;
    reduce_partials.resize(4096);
    reduce_result.resize(16);
    MFMG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&host_result), 16 * sizeof(double)));
  }
  ~HipHandle()
  {
    if (comm_stream)
    {
      (void)hipStreamSynchronize(comm_stream);
      (void)hipEventDestroy(ev_packed);
      (void)hipEventDestroy(ev_unpacked);
      (void)hipStreamDestroy(comm_stream);
    }
    if (host_result)
      (void)hipHostFree(host_result);
    if (owns_stream && stream)
      (void)hipStreamDestroy(stream);
  }
  HipHandle(HipHandle const &) = delete;
  HipHandle &operator=(HipHandle const &) = delete;
};
} // namespace mfmg
