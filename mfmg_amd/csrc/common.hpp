// Shared plumbing of libmfmg_hip: error conventions, device buffers, launch helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
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
// (string literals bind here: the std::string overload BUILDS its message -- a heap allocation -- on every call, passed
// or not; inside the per-entry validation loops of a 223 M-entry matrix that was 4.4 s per matrix, 13 s of a 22 s setup)
inline void ASSERT_THROW(bool cond, char const *message)
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

// Zero-filled host array whose pages are first touched by all threads (a std::vector value-initialises serially: 0.4 s
// for the 1.8 GB of planes of the first coarse operator, a second over the matrices of a setup)
template <typename T>
class ZeroedHostArray
{
public:
  explicit ZeroedHostArray(size_t n) : _p(new T[n]), _n(n)
  {
    T *p = _p.get();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i)
      p[i] = T(0);
  }
  T &operator[](size_t i) { return _p[i]; }
  T const &operator[](size_t i) const { return _p[i]; }
  T *data() { return _p.get(); }
  T const *data() const { return _p.get(); }
  T *begin() { return _p.get(); }
  T *end() { return _p.get() + _n; }
  size_t size() const { return _n; }

private:
  std::unique_ptr<T[]> _p;
  size_t _n;
};

// ---- what the device memory is spent on: live bytes of the DeviceBuffers per (setup phase | kind of structure) ----
// The phase is the section of the reference's TimerOutput that is open when a buffer is allocated (mfmg/timer.hpp:
// "Setup: build restrictor", ...; the levels of the aggregation hierarchy add their own), the kind is set by the class that
// allocates (MemoryKind scopes: CSR arrays, derived layouts, chunk records, restrictor planes, probe vectors ...).  A buffer
// remembers the entry it was counted under and leaves it when it is released.  mfmg_hip_memory_inventory prints the table
// (160 GB at 513^3 DoFs with no inventory).
struct DeviceMemoryLedger
{
  struct Entry
  {
    int64_t live = 0, peak = 0;
  };
  std::mutex mutex;
  std::map<std::string, Entry> entries;
  static DeviceMemoryLedger &get()
  {
    static DeviceMemoryLedger l;
    return l;
  }
  static std::string &phase()
  {
    static thread_local std::string p = "outside any setup section";
    return p;
  }
  static char const *&kind()
  {
    static thread_local char const *k = "other";
    return k;
  }
  Entry *add(int64_t bytes)
  {
    std::lock_guard<std::mutex> lock(mutex);
    Entry &e = entries[phase() + " | " + kind()];
    e.live += bytes;
    e.peak = std::max(e.peak, e.live);
    return &e; // (std::map nodes do not move)
  }
  void sub(Entry *e, int64_t bytes)
  {
    std::lock_guard<std::mutex> lock(mutex);
    e->live -= bytes;
  }
  std::string report(int64_t at_least = 1 << 20)
  {
    std::lock_guard<std::mutex> lock(mutex);
    std::string out;
    int64_t total = 0;
    for (auto const &kv : entries)
    {
      total += kv.second.live;
      if (kv.second.live >= at_least)
      {
        char buf[320];
        snprintf(buf, sizeof(buf), "%10.3f GB  %s\n", double(kv.second.live) * 1e-9, kv.first.c_str());
        out += buf;
      }
    }
    char buf[96];
    snprintf(buf, sizeof(buf), "%10.3f GB  total in device buffers of the library\n", double(total) * 1e-9);
    return out + buf;
  }
};
struct MemoryKind
{
  explicit MemoryKind(char const *k) : _saved(DeviceMemoryLedger::kind()) { DeviceMemoryLedger::kind() = k; }
  ~MemoryKind() { DeviceMemoryLedger::kind() = _saved; }
  char const *_saved;
};
struct MemoryPhase
{
  explicit MemoryPhase(std::string const &p) : _saved(DeviceMemoryLedger::phase()) { DeviceMemoryLedger::phase() = p; }
  ~MemoryPhase() { DeviceMemoryLedger::phase() = _saved; }
  std::string _saved;
};

// ---- owning device buffer (cuda_malloc/cuda_free, include/mfmg/cuda/utils.cuh:66-99) ----
template <typename T>
class DeviceBuffer
{
public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n) { resize(n); }
  DeviceBuffer(DeviceBuffer const &) = delete;
  DeviceBuffer &operator=(DeviceBuffer const &) = delete;
  DeviceBuffer(DeviceBuffer &&o) noexcept : _ptr(o._ptr), _n(o._n), _ledger(o._ledger)
  {
    o._ptr = nullptr;
    o._n = 0;
    o._ledger = nullptr;
  }
  DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
  {
    if (this != &o)
    {
      release();
      _ptr = o._ptr;
      _n = o._n;
      _ledger = o._ledger;
      o._ptr = nullptr;
      o._n = 0;
      o._ledger = nullptr;
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
      _ledger = DeviceMemoryLedger::get().add((int64_t)(n * sizeof(T)));
    }
  }
  void release()
  {
    if (_ptr)
    {
      (void)hipFree(_ptr);
      if (_ledger)
        DeviceMemoryLedger::get().sub(_ledger, (int64_t)(_n * sizeof(T)));
    }
    _ptr = nullptr;
    _n = 0;
    _ledger = nullptr;
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
  DeviceMemoryLedger::Entry *_ledger = nullptr;
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
  std::string only; // when not empty: time launches of these kernel names only ("a" or "a,b": two event records cost ~5 us)
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
    if (!enabled || (!only.empty() && only != name && ("," + only + ",").find(std::string(",") + name + ",") == std::string::npos))
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

// ---- distributed vectors: slabs along the slowest index, or boxes ----
// The reference gets its ghost exchange from deal.II's distributed::Vector inside
// MatrixFree::cell_loop / Epetra Import (and all-gathers the whole vector on the CUDA path,
// source/cuda/utils.cu:363-482).  Here a rank's vector is the lexicographic array of its local box of nodes --
// the nodes it owns plus the ghost layers that belong to its neighbours -- with `comps` entries per node.  Every
// level of the cycle has its own space (fine DoFs, agglomerates, the nodes of the aggregation levels):
//   forward exchange     owner -> ghost: the `width` owned layers next to each neighbour refresh that
//                        neighbour's nearest `width` ghost layers (before an operator reads them);
//   reverse-add exchange ghost -> owner: the `width` ghost layers are sent back and ADDED to the owner's
//                        boundary layers (after a transposed prolongator scattered partial sums into them).
// The ranks form a grid[0] x grid[1] x grid[2] arrangement (rank = cx + grid[0] (cy + grid[1] cz)); 1 x 1 x N are slabs
// along z, whose layers are contiguous runs of the vector and travel without packing.  A box exchanges with ALL its
// neighbours at once -- up to 26 in general, 7 on a 2 x 2 x 2 grid: 3 faces, 3 edges, 1 corner -- one packing kernel, one
// grouped send/recv, one unpacking kernel: towards the neighbour at offset o in {-1, 0, 1}^3 travel the `width` owned
// layers next to it along the axes with o_d != 0, over the OWNED range of the axes with o_d = 0.
struct HaloSpace
{
  // z (the slowest axis)
  int64_t layer_elems = 0;  // entries per z layer (a DoF plane, a layer of agglomerates / aggregates)
  int64_t n_layers = 0;     // z layers of the local vector
  int64_t owned_begin = 0;  // first owned layer
  int64_t owned_count = 0;  // owned layers
  int64_t global_begin = 0; // global index of local layer 0
  int64_t global_layers = 0;
  int width = 1;            // layers moved per side by an exchange (<= the ghost layers present on that side)
  bool has_low = false, has_high = false;
  // x and y ([0], [1]): whole in a slab run (own = local = global)
  int comps = 1;                        // entries per node
  int64_t n_xy[2] = {0, 0};             // local nodes (layer_elems = comps n_xy[0] n_xy[1])
  int64_t own0_xy[2] = {0, 0}, own_n_xy[2] = {0, 0};
  int64_t g0_xy[2] = {0, 0}, gn_xy[2] = {0, 0}; // global index of local node 0, global nodes
  bool low_xy[2] = {false, false}, high_xy[2] = {false, false};
  bool configured() const { return layer_elems > 0; }
  int64_t ghost_low() const { return owned_begin; }
  int64_t ghost_high() const { return n_layers - owned_begin - owned_count; }
  bool split_xy() const { return low_xy[0] || high_xy[0] || low_xy[1] || high_xy[1]; }
  // x and y not split: one rank holds every node of a layer
  void set_whole_xy(int64_t nx, int64_t ny, int n_comps)
  {
    comps = n_comps;
    n_xy[0] = own_n_xy[0] = gn_xy[0] = nx;
    n_xy[1] = own_n_xy[1] = gn_xy[1] = ny;
    own0_xy[0] = own0_xy[1] = g0_xy[0] = g0_xy[1] = 0;
    low_xy[0] = low_xy[1] = high_xy[0] = high_xy[1] = false;
  }
  // the three axes alike (d = 2: z)
  int64_t dim(int d) const { return d == 2 ? n_layers : n_xy[d]; }
  int64_t own0(int d) const { return d == 2 ? owned_begin : own0_xy[d]; }
  int64_t own_n(int d) const { return d == 2 ? owned_count : own_n_xy[d]; }
  int64_t g0(int d) const { return d == 2 ? global_begin : g0_xy[d]; }
  int64_t gn(int d) const { return d == 2 ? global_layers : gn_xy[d]; }
  bool low(int d) const { return d == 2 ? has_low : low_xy[d]; }
  bool high(int d) const { return d == 2 ? has_high : high_xy[d]; }
  int64_t n_local() const { return layer_elems * n_layers; }
  int64_t n_owned() const { return comps * own_n_xy[0] * own_n_xy[1] * owned_count; }
  int64_t n_global() const { return comps * gn_xy[0] * gn_xy[1] * global_layers; }
  void check() const
  {
    if (layer_elems != comps * n_xy[0] * n_xy[1])
      throw std::runtime_error("internal: halo space without its x / y description");
  }
  // entry i of a local vector: its node, whether this rank owns it, its position in the global lexicographic vector
  void node_of(int64_t i, int64_t c[3]) const
  {
    const int64_t nd = i / comps;
    c[0] = nd % n_xy[0];
    c[1] = (nd / n_xy[0]) % n_xy[1];
    c[2] = nd / (n_xy[0] * n_xy[1]);
  }
  bool owned(int64_t i) const
  {
    int64_t c[3];
    node_of(i, c);
    return c[0] >= own0_xy[0] && c[0] < own0_xy[0] + own_n_xy[0] && c[1] >= own0_xy[1] && c[1] < own0_xy[1] + own_n_xy[1] &&
           c[2] >= owned_begin && c[2] < owned_begin + owned_count;
  }
  int64_t global_id(int64_t i) const
  {
    int64_t c[3];
    node_of(i, c);
    return (((c[2] + global_begin) * gn_xy[1] + (c[1] + g0_xy[1])) * gn_xy[0] + (c[0] + g0_xy[0])) * comps + i % comps;
  }
};

// the sub-boxes of a local array that one box exchange packs (or unpacks): region r = nodes [b, b + n) per axis, its entries
// at off[r] .. off[r + 1] of the packed buffer
struct HaloRegions
{
  int count = 0;
  int b[26][3], n[26][3];
  int64_t off[27];
};

// Point-to-point transport between neighbours + the few collectives of the setup.  Two implementations:
// RCCL send/recv over xGMI on the caller's stream (one process per GPU), and host callbacks (the library stages
// through pinned host buffers; gloo in the tests, where several ranks share one card).
struct HaloTransport
{
  virtual ~HaloTransport() = default;
  // exchange n_low doubles with rank `peer_low` and n_high with rank `peer_high` (device pointers; a count of 0 = no
  // neighbour); enqueued on `stream` (the host transport synchronises the stream around its callbacks)
  virtual void sendrecv(int peer_low, int peer_high, double const *send_low, double *recv_low, int64_t n_low, double const *send_high,
                        double *recv_high, int64_t n_high, hipStream_t stream) = 0;
  // the same with any number of partners in one group: message i = count[i] doubles to and from rank peers[i]
  virtual void exchange_many(int n, int const *peers, double const *const *send, double *const *recv, int64_t const *count,
                             hipStream_t stream) = 0;
  virtual void allreduce(double *host_values, int n, int op /* 0 sum, 1 max */, hipStream_t stream) = 0;
  // every rank contributes n doubles (device), `out` (device) receives n * n_ranks in rank order
  virtual void allgather(double const *in, int64_t n, double *out, hipStream_t stream) = 0;
  // send n doubles to this rank itself and receive them (exercises the point-to-point path on a single GPU)
  virtual void loopback(double const *send, double *recv, int64_t n, hipStream_t stream) = 0;
  virtual char const *name() const = 0;
  // ranks the transport's own communicator reports (RCCL: ncclCommCount), for the record of a multi-GPU run
  virtual int comm_ranks() const = 0;
  // the stream the overlapped exchanges run on: a transport may give it a communicator of its own, so that exchanges on the
  // compute stream and on the exchange stream do not queue behind each other inside ONE communicator (RCCL orders the
  // operations of a communicator)
  virtual void bind_exchange_stream(hipStream_t) {}
};

struct HaloCommunicator
{
  int rank = 0, n_ranks = 1;
  int grid[3] = {1, 1, 1};  // ranks along x, y, z (slabs: 1 x 1 x n_ranks)
  int coord[3] = {0, 0, 0}; // this rank's position
  int ghost_lo[3] = {0, 0, 0}, ghost_hi[3] = {0, 0, 0}; // ghost cell layers of the local mesh per axis
  int ghost_cells_low = 0, ghost_cells_high = 0;        // ... those along z
  // ghost cell layers EVERY rank of the run holds towards a lower neighbour (2 = one agglomerate, 4 = two; the same value on all
  // ranks, also on those without a lower neighbour: what the ranks may do together follows from it).  The interface plane
  // belongs to the upper box, so a box holds ghost_hi + 1 = 3 ghost node planes above but only low_ghost_cells below: a sweep of K
  // smoother terms needs K on every side with a neighbour (mf_cheb_fused.hip)
  int low_ghost_cells = 2;
  int sweep_terms() const { return low_ghost_cells >= 4 ? 3 : 2; }
  std::shared_ptr<HaloTransport> transport;
  std::vector<HaloSpace> spaces = std::vector<HaloSpace>(3); // [0] rank-local, [1] fine DoFs, [2] first coarse level, then the aggregation levels
  int64_t n_exchanges = 0; // (diagnostics) point-to-point exchanges issued so far
  int64_t n_doubles_sent = 0; // ... and the doubles this rank sent in them
  int64_t n_overlapped = 0;   // ... and how many of them ran on the second stream beside operator tiles
  // The spaces from [2] on describe the levels of ONE hierarchy (its operators hold indices into `spaces`): the
  // hierarchy helpers that configured them own them until they are destroyed; a second hierarchy on the same
  // communicator is refused while the first is alive (it would re-purpose spaces the first one still exchanges with).
  void const *spaces_owner = nullptr;
  bool enabled() const { return n_ranks > 1; }
  bool split_xy() const { return grid[0] > 1 || grid[1] > 1; }
  int stride(int d) const { return d == 0 ? 1 : (d == 1 ? grid[0] : grid[0] * grid[1]); }
  bool has_lower(int d) const { return coord[d] > 0; }
  bool has_upper(int d) const { return coord[d] + 1 < grid[d]; }
  int add_space(HaloSpace const &s)
  {
    spaces.push_back(s);
    return (int)spaces.size() - 1;
  }
};

void halo_add_layers(double *dst, double const *src, int64_t n, hipStream_t stream); // dst += src (vector_ops.hip)
void stream_delay(double microseconds, hipStream_t stream); // a one-thread kernel that holds the stream that long (vector_ops.hip)
struct HaloRegions;
struct HaloSpace;
// the regions of a vector of space `s` against the packed buffer: mode 0 buf = v (pack), 1 v = buf (unpack), 2 v += buf (vector_ops.hip)
void halo_regions_copy(double *v, HaloSpace const &s, HaloRegions const &regions, double *buf, int mode, hipStream_t stream);
// the owned sub-box (or the whole local box) of a vector of space `s` against the contiguous buf: mode 0 buf = v, 1 v = buf
void halo_box_copy(double *v, HaloSpace const &s, bool owned_only, double *buf, int mode, hipStream_t stream);
void gather_indexed(int64_t n, double const *in, int32_t const *index, double *out, hipStream_t stream); // out[i] = in[index[i]]

// ---- HipHandle: stream + reduction scratch; twin of CudaHandle
//      (include/mfmg/cuda/cuda_handle.cuh:25-48): borrowed by every object built from it ----
struct HipHandle
{
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // second stream of the overlapped halo exchange (created at first use) and its two events
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_unpacked = nullptr, ev_async = nullptr;
  bool overlap_exchange = true;
  // matrix-free operators built from this handle may keep one coefficient per cell when a cell's eight are equal
  bool allow_cell_constant = true;
  // ... and then keep D^-1 in the chunk records (8 more bytes per DoF and smoother launch) instead of deriving it
  // in the kernel from the cell coefficients
  bool stored_diagonal = false;
  // polynomial terms of the Chebyshev smoother that one sweep of the matrix-free operator may run (mf_cheb_fused.hip); the chunk
  // records of operators built from this handle carry that many halo lanes (1 = one term per launch, the layout of rounds 1-3)
  int mf_fused_terms = 3;
  // measurement switches of the distributed fine operator (hip_hierarchy.hip: HipMatrixFreeOperator::apply_mode), read from the
  // environment ONCE when the handle is built (MFMG_MF_SHELL = slabs | after, MFMG_MF_EMULATE_SPLIT = z | yz | xyz | 1) and
  // set at run time through mfmg_hip_context_set_mf_shell / _set_mf_emulate_split -- not read per application
  int mf_shell_mode = 0;    // 0: the shell beside the interior tiles on the exchange stream; 1: after them; 2: slab by slab
  int mf_emulate_split = 0; // one rank only: 0 off, 1 = z, 2 = yz, 3 = xyz: the launches of a rank of 1x1x2 / 1x2x2 / 2x2x2, no exchange
  // R A R^T of a matrix-free A by probing on the device (hip_hierarchy.hip, HipMatrixOperator::multiply) instead of
  // the host triple product
  bool galerkin_on_device = true;
  // "setup value precision" float (parameter of the hierarchy): the matrices the setup forms -- R, R A R^T, the
  // prolongators and operators of the aggregation hierarchy -- are rounded to float-representable values when they are
  // assembled; the layouts then keep them in float (half the bytes per application), arithmetic stays FP64
  bool setup_values_float = false;
  // scratch for two-stage deterministic reductions
  DeviceBuffer<double> reduce_partials;
  DeviceBuffer<double> reduce_result;
  double *host_result = nullptr; // pinned
  KernelProfiler profiler;
  HaloCommunicator comm;
  // staging of the reverse (adding) exchanges and of the packed regions of a box exchange:
  // [send_low | send_high | recv_low | recv_high] resp. [send (2 segments) | recv (2 segments)], grown on demand
  // One buffer per STREAM: an exchange on the exchange stream (the overlapped fine exchange, prefetch_rhs) and one on the compute
  // stream (a rank without interior tiles, the x exchange of the residual restriction) may be in flight together, and a shared
  // buffer let their packed regions overwrite each other.  Exchanges of one stream are ordered by the stream.
  DeviceBuffer<double> halo_staging, halo_staging_comm;
  int64_t halo_staging_each = 0, halo_staging_comm_each = 0;
  DeviceBuffer<double> dot_scratch; // owned entries of two box vectors, packed for a dot product
  // The right-hand side of the cycle in flight (Hierarchy::apply -> Operator::prefetch_rhs): the vector whose ghost entries are
  // travelling on the exchange stream / are known to be fresh until the cycle ends (release_rhs).  The multi-term smoother sweep
  // and the one-pass residual restriction both read b at ghost DoFs; outside a cycle both exchange it themselves.
  double const *rhs_in_flight = nullptr, *rhs_fresh = nullptr, *rhs_of_cycle = nullptr;
  bool rhs_ghosts_wanted = false; // the fine smoother of this context reads b at ghost DoFs (distributed multi-term sweep)
  int rhs_ghost_width = 1;        // ... that many planes deep (a sweep of K terms: K - 1; the residual restriction: 1)
  int rhs_fresh_width = 0;        // depth of the exchange that made rhs_fresh / rhs_in_flight current
  // the fine DoF space with ghost layers `width` planes deep
  HaloSpace fine_space(int width) const
  {
    HaloSpace s = comm.spaces[1];
    s.width = width;
    return s;
  }
  // make the ghost entries of the fine-level vector b current on `stream`, `width` planes deep
  void need_rhs_ghosts(double const *b, int width = 1)
  {
    if (!comm.enabled())
      return;
    if (rhs_in_flight == b)
    {
      exchange_async_wait();
      rhs_fresh = b;
      rhs_in_flight = nullptr;
    }
    if (rhs_fresh != b || rhs_fresh_width < width)
    {
      if (!comm.transport)
        throw std::runtime_error("no halo transport was registered with the context");
      exchange_on(fine_space(std::max(width, rhs_ghost_width)), const_cast<double *>(b), stream, stream, false);
      rhs_fresh_width = std::max(width, rhs_ghost_width);
      rhs_fresh = rhs_of_cycle == b ? b : nullptr; // (fresh only while the cycle that announced this vector is running)
    }
  }

  HaloSpace &space_checked(int space)
  {
    if (space <= 0 || space >= (int)comm.spaces.size() || !comm.spaces[space].configured())
      throw std::runtime_error("halo exchange requested for an unconfigured vector space");
    if (!comm.transport)
      throw std::runtime_error("no halo transport was registered with the context");
    return comm.spaces[space];
  }
  // staging of the exchanges enqueued on `st` (grown on demand; both streams are drained before a buffer is replaced)
  double *staging_reserve(int64_t each, hipStream_t st, int64_t &each_now)
  {
    const bool on_comm = comm_stream != nullptr && st == comm_stream;
    DeviceBuffer<double> &buf = on_comm ? halo_staging_comm : halo_staging;
    int64_t &have = on_comm ? halo_staging_comm_each : halo_staging_each;
    if (each > have)
    {
      MFMG_HIP_CHECK(hipStreamSynchronize(stream));
      if (comm_stream)
        MFMG_HIP_CHECK(hipStreamSynchronize(comm_stream));
      buf.resize((size_t)4 * each);
      have = each;
    }
    each_now = have;
    return buf.data();
  }
  // A box exchange on `st` (forward: owner -> ghost; reverse: ghost -> owner, added): every existing neighbour at an offset
  // o in {-1, 0, 1}^3 gets one message.  Along an axis with o_d = -1 / +1 the message spans the `width` owned layers next to that
  // neighbour (forward: sent; reverse: added to) resp. the `width` ghost layers beyond them (forward: received; reverse: sent);
  // along an axis with o_d = 0 it spans the owned range.
  void exchange_box(HaloSpace const &s, double *v, bool reverse, hipStream_t st)
  {
    HaloRegions own, ghost;
    int peers[26];
    int64_t counts[26];
    int64_t total = 0;
    for (int oz = -1; oz <= 1; ++oz)
      for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox)
        {
          const int o[3] = {ox, oy, oz};
          if (ox == 0 && oy == 0 && oz == 0)
            continue;
          bool exists = true;
          for (int d = 0; d < 3; ++d)
            if ((o[d] < 0 && !s.low(d)) || (o[d] > 0 && !s.high(d)))
              exists = false;
          if (!exists)
            continue;
          const int r = own.count;
          int64_t n = s.comps;
          for (int d = 0; d < 3; ++d)
          {
            const int o0 = (int)s.own0(d), o1 = (int)(s.own0(d) + s.own_n(d)), w = s.width;
            own.b[r][d] = o[d] < 0 ? o0 : (o[d] > 0 ? o1 - w : o0);
            ghost.b[r][d] = o[d] < 0 ? o0 - w : (o[d] > 0 ? o1 : o0);
            own.n[r][d] = ghost.n[r][d] = o[d] == 0 ? o1 - o0 : w;
            n *= own.n[r][d];
          }
          own.off[r] = ghost.off[r] = total;
          peers[r] = comm.rank + ox * comm.stride(0) + oy * comm.stride(1) + oz * comm.stride(2);
          counts[r] = n;
          total += n;
          own.count = ghost.count = r + 1;
        }
    if (own.count == 0)
      return;
    own.off[own.count] = ghost.off[own.count] = total;
    int64_t each = 0;
    double *send = staging_reserve((total + 1) / 2 + 1, st, each), *recv = send + 2 * each;
    double const *send_ptr[26];
    double *recv_ptr[26];
    for (int r = 0; r < own.count; ++r)
    {
      send_ptr[r] = send + own.off[r];
      recv_ptr[r] = recv + own.off[r];
    }
    halo_regions_copy(v, s, reverse ? ghost : own, send, 0, st);
    comm.transport->exchange_many(own.count, peers, send_ptr, recv_ptr, counts, st);
    ++comm.n_exchanges;
    comm.n_doubles_sent += total;
    halo_regions_copy(v, s, reverse ? own : ghost, recv, reverse ? 2 : 1, st);
  }
  // forward exchange on `st`.  Along z the `width` owned layers next to a neighbour and the ghost layers they refresh are
  // contiguous runs of the vector (lexicographic layers): the transport sends from and receives into the vector itself
  // -- no packing, no staging copies (round 2 moved every layer through a staging buffer: four device copies per exchange).
  // A box packs the regions of all its neighbours (exchange_box).
  void exchange_on(HaloSpace const &s, double *v, hipStream_t pack_stream, hipStream_t st, bool split)
  {
    const int64_t n = (int64_t)s.width * s.layer_elems;
    double const *send_low = v + s.owned_begin * s.layer_elems, *send_high = v + (s.owned_begin + s.owned_count - s.width) * s.layer_elems;
    double *recv_low = v + (s.owned_begin - s.width) * s.layer_elems, *recv_high = v + (s.owned_begin + s.owned_count) * s.layer_elems;
    if (split)
    {
      // the boundary layers are final on `pack_stream` at this point: the transport stream may read them from here on
      MFMG_HIP_CHECK(hipEventRecord(ev_packed, pack_stream));
      MFMG_HIP_CHECK(hipStreamWaitEvent(st, ev_packed, 0));
    }
    if (s.split_xy())
    {
      exchange_box(s, v, false, st);
      return;
    }
    if (!s.has_low && !s.has_high)
      return;
    comm.transport->sendrecv(comm.rank - comm.stride(2), comm.rank + comm.stride(2), send_low, recv_low, s.has_low ? n : 0, send_high,
                             recv_high, s.has_high ? n : 0, st);
    ++comm.n_exchanges;
    comm.n_doubles_sent += (s.has_low ? n : 0) + (s.has_high ? n : 0);
  }
  // refresh the ghost layers of a distributed vector (no-op on one rank / for local spaces)
  void exchange(int space, double *v)
  {
    if (!comm.enabled() || space <= 0)
      return;
    exchange_on(space_checked(space), v, stream, stream, false);
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
    HaloSpace &s = space_checked(space);
    (void)exchange_stream();
    exchange_on(s, v, stream, comm_stream, true);
    ++comm.n_overlapped;
    MFMG_HIP_CHECK(hipEventRecord(ev_unpacked, comm_stream));
  }
  void exchange_end(int space)
  {
    if (!comm.enabled() || space <= 0)
      return;
    MFMG_HIP_CHECK(hipStreamWaitEvent(stream, ev_unpacked, 0));
  }
  // Work enqueued on the exchange stream behind the unpacking (the shell tiles of the fine operator: they run beside the
  // interior tiles on `stream` instead of after them); join_exchange_stream makes `stream` wait for it.
  hipStream_t exchange_stream()
  {
    if (comm_stream == nullptr)
    {
      {
        // (highest priority: the packing, the transport and the shell tiles enqueued here must not queue behind the workgroups the
        // interior launch still has to dispatch on the compute stream)
        int pr_low = 0, pr_high = 0;
        MFMG_HIP_CHECK(hipDeviceGetStreamPriorityRange(&pr_low, &pr_high));
        MFMG_HIP_CHECK(hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, pr_high));
      }
      MFMG_HIP_CHECK(hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming));
      MFMG_HIP_CHECK(hipEventCreateWithFlags(&ev_unpacked, hipEventDisableTiming));
      if (comm.transport)
        comm.transport->bind_exchange_stream(comm_stream);
    }
    return comm_stream;
  }
  void join_exchange_stream()
  {
    MFMG_HIP_CHECK(hipEventRecord(ev_unpacked, exchange_stream()));
    MFMG_HIP_CHECK(hipStreamWaitEvent(stream, ev_unpacked, 0));
  }
  // A whole exchange on the exchange stream, behind what `stream` has enqueued so far; `stream` goes on and waits for it with
  // exchange_async_wait() where it needs the ghost entries (the right-hand side of a cycle: prefetch_rhs).  In between `stream`
  // must not read the ghost entries of `v` nor write its boundary layers.
  void exchange_async(int space, double *v)
  {
    if (!comm.enabled() || space <= 0)
      return;
    exchange_async(space_checked(space), v);
  }
  void exchange_async(HaloSpace const &s, double *v)
  {
    hipStream_t cs = exchange_stream();
    if (ev_async == nullptr)
      MFMG_HIP_CHECK(hipEventCreateWithFlags(&ev_async, hipEventDisableTiming));
    exchange_on(s, v, stream, cs, true);
    ++comm.n_overlapped;
    MFMG_HIP_CHECK(hipEventRecord(ev_async, cs));
  }
  void exchange_async_wait()
  {
    if (ev_async != nullptr)
      MFMG_HIP_CHECK(hipStreamWaitEvent(stream, ev_async, 0));
  }
  // (one rank, measurement only: the exchange stream waits for what `stream` has enqueued so far)
  void fork_exchange_stream()
  {
    hipStream_t cs = exchange_stream();
    MFMG_HIP_CHECK(hipEventRecord(ev_packed, stream));
    MFMG_HIP_CHECK(hipStreamWaitEvent(cs, ev_packed, 0));
  }
  // ghost -> owner: the ghost layers hold partial sums that belong to the neighbours' boundary layers
  void exchange_reverse_add(int space, double *v)
  {
    if (!comm.enabled() || space <= 0)
      return;
    HaloSpace &s = space_checked(space);
    if (s.split_xy())
    {
      exchange_box(s, v, true, stream);
      return;
    }
    if (s.has_low || s.has_high)
    {
      const int64_t n = (int64_t)s.width * s.layer_elems;
      int64_t each = 0;
      double *staging = staging_reserve(n, stream, each);
      // the ghost layers are sent as they lie; what comes back is added to the owned boundary layers, so it is received in staging
      double const *send_low = v + (s.owned_begin - s.width) * s.layer_elems, *send_high = v + (s.owned_begin + s.owned_count) * s.layer_elems;
      double *recv_low = staging + 2 * each, *recv_high = recv_low + each;
      comm.transport->sendrecv(comm.rank - comm.stride(2), comm.rank + comm.stride(2), send_low, recv_low, s.has_low ? n : 0, send_high,
                               recv_high, s.has_high ? n : 0, stream);
      ++comm.n_exchanges;
      comm.n_doubles_sent += (s.has_low ? n : 0) + (s.has_high ? n : 0);
      if (s.has_low)
        halo_add_layers(v + s.owned_begin * s.layer_elems, recv_low, n, stream);
      if (s.has_high)
        halo_add_layers(v + (s.owned_begin + s.owned_count - s.width) * s.layer_elems, recv_high, n, stream);
    }
  }
  double allreduce_sum(double v)
  {
    if (!comm.enabled())
      return v;
    if (!comm.transport)
      throw std::runtime_error("no halo transport was registered with the context");
    comm.transport->allreduce(&v, 1, 0, stream);
    return v;
  }
  double allreduce_max(double v)
  {
    if (!comm.enabled())
      return v;
    if (!comm.transport)
      throw std::runtime_error("no halo transport was registered with the context");
    comm.transport->allreduce(&v, 1, 1, stream);
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
    if (char const *e = std::getenv("MFMG_MF_SHELL"))
      mf_shell_mode = std::string(e) == "after" ? 1 : (std::string(e) == "slabs" ? 2 : 0);
    if (char const *e = std::getenv("MFMG_MF_EMULATE_SPLIT"))
    {
      const std::string v(e);
      mf_emulate_split = v == "z" ? 1 : (v == "yz" ? 2 : ((v == "xyz" || v == "1") ? 3 : 0));
    }
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
      if (ev_async)
        (void)hipEventDestroy(ev_async);
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
