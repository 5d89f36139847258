#include "halo_transport.hpp"

#include <dlfcn.h>
#include <link.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace mfmg
{
namespace
{
// ---- the handful of RCCL entry points used, resolved with dlsym (rccl/rccl.h, NCCL 2.x ABI) ----
struct ncclComm;
typedef ncclComm *ncclComm_t;
struct ncclUniqueId
{
  char internal[128];
};
constexpr int ncclSuccess = 0;
constexpr int ncclFloat64 = 8; // ncclDataType_t
constexpr int ncclSum = 0, ncclMax = 2;

struct RcclApi
{
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(ncclComm_t, int *) = nullptr;
  int (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr; // (optional: NCCL >= 2.18)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  void *lib = nullptr;
};

RcclApi &rccl()
{
  static RcclApi api = [] {
    RcclApi a;
    // the copy already in the process (torch ships its own librccl.so) comes first: two RCCL instances in one
    // process would each bring their own topology state
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    // (torch's bundled copy sits under its own directory: the bare soname may not match it, so the loaded objects are
    // searched for a librccl first and that very file is re-opened)
    std::string loaded;
    dl_iterate_phdr(
        [](struct dl_phdr_info *info, size_t, void *data) {
          if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so") != nullptr)
          {
            *static_cast<std::string *>(data) = info->dlpi_name;
            return 1;
          }
          return 0;
        },
        &loaded);
    if (!loaded.empty())
      a.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
      if (a.lib == nullptr && (a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr)
        break;
    if (!a.lib)
      for (const char *n : names)
        if ((a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr)
          break;
    if (!a.lib)
      throw std::runtime_error("RCCL transport: librccl.so could not be loaded");
    auto sym = [&](const char *name) {
      void *p = dlsym(a.lib, name);
      if (!p)
        throw std::runtime_error(std::string("RCCL transport: missing symbol ") + name);
      return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.CommSplit = reinterpret_cast<decltype(a.CommSplit)>(dlsym(a.lib, "ncclCommSplit"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    return a;
  }();
  return api;
}

void rccl_check(int status, const char *what)
{
  if (status != ncclSuccess)
    throw std::runtime_error(std::string("RCCL error in ") + what + ": " + rccl().GetErrorString(status));
}

class RcclTransport : public HaloTransport
{
public:
  RcclTransport(int rank, int n_ranks, unsigned char const unique_id[128]) : _rank(rank)
  {
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, 128);
    rccl_check(rccl().CommInitRank(&_comm, n_ranks, id, rank), "ncclCommInitRank");
    _scalars.resize(16);
  }
  ~RcclTransport() override
  {
    if (_comm_x)
      (void)rccl().CommDestroy(_comm_x);
    if (_comm)
      (void)rccl().CommDestroy(_comm);
  }
  // A second communicator over the same ranks for the exchange stream (ncclCommSplit with one colour: collective, every rank
  // binds its exchange stream at the same point of the setup).  RCCL serialises the operations of ONE communicator, so the
  // overlapped fine exchange on the exchange stream and a coarse exchange on the compute stream would otherwise wait for each
  // other through the communicator's internal events (multi-GPU (i)).  MFMG_RCCL_ONE_COMM=1 keeps one.
  void bind_exchange_stream(hipStream_t s) override
  {
    _exchange_stream = s;
    char const *one = std::getenv("MFMG_RCCL_ONE_COMM");
    if (_comm_x == nullptr && !_split_tried && rccl().CommSplit != nullptr && !(one && std::string(one) == "1"))
    {
      // (a refusal is not fatal: the exchanges of both streams then share the first communicator, as in round 3)
      _split_tried = true;
      const int status = rccl().CommSplit(_comm, 0, _rank, &_comm_x, nullptr);
      if (status != ncclSuccess)
      {
        std::fprintf(stderr, "mfmg-hip: ncclCommSplit refused (%s): one communicator for both streams\n", rccl().GetErrorString(status));
        _comm_x = nullptr;
      }
    }
  }
  ncclComm_t comm_for(hipStream_t s) const { return (_comm_x != nullptr && s == _exchange_stream) ? _comm_x : _comm; }
  int comm_ranks() const override
  {
    int n = 0;
    rccl_check(rccl().CommCount(_comm, &n), "ncclCommCount");
    return n;
  }
  void sendrecv(int peer_low, int peer_high, double const *send_low, double *recv_low, int64_t n_low, double const *send_high,
                double *recv_high, int64_t n_high, hipStream_t stream) override
  {
    if (n_low <= 0 && n_high <= 0)
      return;
    RcclApi &r = rccl();
    ncclComm_t comm = comm_for(stream);
    rccl_check(r.GroupStart(), "ncclGroupStart");
    if (n_low > 0)
    {
      rccl_check(r.Send(send_low, (size_t)n_low, ncclFloat64, peer_low, comm, stream), "ncclSend");
      rccl_check(r.Recv(recv_low, (size_t)n_low, ncclFloat64, peer_low, comm, stream), "ncclRecv");
    }
    if (n_high > 0)
    {
      rccl_check(r.Send(send_high, (size_t)n_high, ncclFloat64, peer_high, comm, stream), "ncclSend");
      rccl_check(r.Recv(recv_high, (size_t)n_high, ncclFloat64, peer_high, comm, stream), "ncclRecv");
    }
    rccl_check(r.GroupEnd(), "ncclGroupEnd");
  }
  void exchange_many(int n, int const *peers, double const *const *send, double *const *recv, int64_t const *count,
                     hipStream_t stream) override
  {
    if (n <= 0)
      return;
    RcclApi &r = rccl();
    ncclComm_t comm = comm_for(stream);
    rccl_check(r.GroupStart(), "ncclGroupStart");
    for (int i = 0; i < n; ++i)
      if (count[i] > 0)
      {
        rccl_check(r.Send(send[i], (size_t)count[i], ncclFloat64, peers[i], comm, stream), "ncclSend");
        rccl_check(r.Recv(recv[i], (size_t)count[i], ncclFloat64, peers[i], comm, stream), "ncclRecv");
      }
    rccl_check(r.GroupEnd(), "ncclGroupEnd");
  }
  void allreduce(double *host_values, int n, int op, hipStream_t stream) override
  {
    ASSERT_THROW(n >= 1 && n <= 16, "all-reduce of at most 16 scalars");
    MFMG_HIP_CHECK(hipMemcpyAsync(_scalars.data(), host_values, n * sizeof(double), hipMemcpyHostToDevice, stream));
    rccl_check(rccl().AllReduce(_scalars.data(), _scalars.data(), (size_t)n, ncclFloat64, op == 1 ? ncclMax : ncclSum, _comm, stream),
               "ncclAllReduce");
    MFMG_HIP_CHECK(hipMemcpyAsync(host_values, _scalars.data(), n * sizeof(double), hipMemcpyDeviceToHost, stream));
    MFMG_HIP_CHECK(hipStreamSynchronize(stream));
  }
  void allgather(double const *in, int64_t n, double *out, hipStream_t stream) override
  {
    rccl_check(rccl().AllGather(in, out, (size_t)n, ncclFloat64, _comm, stream), "ncclAllGather");
  }
  void loopback(double const *send, double *recv, int64_t n, hipStream_t stream) override
  {
    RcclApi &r = rccl();
    rccl_check(r.GroupStart(), "ncclGroupStart");
    rccl_check(r.Send(send, (size_t)n, ncclFloat64, _rank, _comm, stream), "ncclSend");
    rccl_check(r.Recv(recv, (size_t)n, ncclFloat64, _rank, _comm, stream), "ncclRecv");
    rccl_check(r.GroupEnd(), "ncclGroupEnd");
  }
  char const *name() const override { return "rccl"; }

private:
  int _rank;
  ncclComm_t _comm = nullptr, _comm_x = nullptr;
  bool _split_tried = false;
  hipStream_t _exchange_stream = nullptr;
  DeviceBuffer<double> _scalars;
};

class HostTransport : public HaloTransport
{
public:
  HostTransport(int rank, int n_ranks, mfmg_hip_host_exchange_fn sr, mfmg_hip_host_allreduce_fn ar, mfmg_hip_host_allgather_fn ag,
                void *user)
      : _n(n_ranks), _sr(sr), _ar(ar), _ag(ag), _user(user)
  {
    ASSERT_THROW(sr && ar && ag, "null transport callbacks");
  }
  ~HostTransport() override
  {
    if (_host)
      (void)hipHostFree(_host);
  }
  void sendrecv(int peer_low, int peer_high, double const *send_low, double *recv_low, int64_t n_low, double const *send_high,
                double *recv_high, int64_t n_high, hipStream_t stream) override
  {
    int peers[2], n = 0;
    double const *send[2];
    double *recv[2];
    int64_t count[2];
    if (n_low > 0)
    {
      peers[n] = peer_low;
      send[n] = send_low;
      recv[n] = recv_low;
      count[n++] = n_low;
    }
    if (n_high > 0)
    {
      peers[n] = peer_high;
      send[n] = send_high;
      recv[n] = recv_high;
      count[n++] = n_high;
    }
    exchange_many(n, peers, send, recv, count, stream);
  }
  void exchange_many(int n, int const *peers, double const *const *send, double *const *recv, int64_t const *count,
                     hipStream_t stream) override
  {
    if (n <= 0)
      return;
    ASSERT_THROW(n <= 26, "at most 26 neighbours");
    int64_t total = 0;
    for (int i = 0; i < n; ++i)
      total += count[i];
    reserve(2 * total);
    const double *h_send[26];
    double *h_recv[26];
    int32_t h_peers[26];
    int64_t off = 0;
    for (int i = 0; i < n; ++i)
    {
      h_send[i] = _host + off;
      h_recv[i] = _host + total + off;
      h_peers[i] = peers[i];
      MFMG_HIP_CHECK(hipMemcpyAsync(_host + off, send[i], (size_t)count[i] * sizeof(double), hipMemcpyDeviceToHost, stream));
      off += count[i];
    }
    MFMG_HIP_CHECK(hipStreamSynchronize(stream));
    if (_sr(_user, n, h_peers, h_send, h_recv, count) != 0)
      throw std::runtime_error("halo exchange transport failed");
    for (int i = 0; i < n; ++i)
      MFMG_HIP_CHECK(hipMemcpyAsync(recv[i], h_recv[i], (size_t)count[i] * sizeof(double), hipMemcpyHostToDevice, stream));
    MFMG_HIP_CHECK(hipStreamSynchronize(stream)); // the pinned buffers are reused by the next call
  }
  void allreduce(double *host_values, int n, int op, hipStream_t) override
  {
    if (_ar(_user, host_values, n, op) != 0)
      throw std::runtime_error("all-reduce transport failed");
  }
  void allgather(double const *in, int64_t n, double *out, hipStream_t stream) override
  {
    reserve(n * (int64_t)(_n + 1));
    MFMG_HIP_CHECK(hipMemcpyAsync(_host, in, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    MFMG_HIP_CHECK(hipStreamSynchronize(stream));
    if (_ag(_user, _host, n, _host + n) != 0)
      throw std::runtime_error("all-gather transport failed");
    MFMG_HIP_CHECK(hipMemcpyAsync(out, _host + n, n * _n * sizeof(double), hipMemcpyHostToDevice, stream));
    MFMG_HIP_CHECK(hipStreamSynchronize(stream));
  }
  void loopback(double const *send, double *recv, int64_t n, hipStream_t stream) override
  {
    MFMG_HIP_CHECK(hipMemcpyAsync(recv, send, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  }
  char const *name() const override { return "host"; }
  int comm_ranks() const override { return _n; }

private:
  void reserve(int64_t n)
  {
    if (n <= _host_n)
      return;
    if (_host)
      (void)hipHostFree(_host);
    MFMG_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&_host), (size_t)n * sizeof(double)));
    _host_n = n;
  }
  int _n;
  mfmg_hip_host_exchange_fn _sr;
  mfmg_hip_host_allreduce_fn _ar;
  mfmg_hip_host_allgather_fn _ag;
  void *_user;
  double *_host = nullptr;
  int64_t _host_n = 0;
};
// One rank of a grid on its own, for MEASUREMENT of an existing hierarchy: every message this rank sends comes straight back as
// the message it would have received (a device copy on the stream of the exchange), an all-gather repeats this rank's block, an
// all-reduce is the identity.  Registered AFTER the hierarchy was set up with a real transport (the setup needs the real
// neighbours), the rank then runs every kernel, packing, unpacking, stream join and replicated level of its share of a
// distributed cycle with a wire that costs nothing -- what the partition costs by itself (scratch/rank_cycle_on_one_gpu.py).
// The numbers it iterates on are not a solution of anything.
class ReflectingTransport : public HaloTransport
{
public:
  // delay_us: what one grouped send/recv (resp. one collective) holds its stream for before the reflected data are there -- the
  // latency of the wire the harness otherwise leaves out (a free wire cannot show what 20-40 us per group do to a
  // cycle of 17 exchanges).  A one-thread kernel on the stream of the exchange, in FRONT of the copy.
  explicit ReflectingTransport(int n_ranks, double delay_us) : _n(n_ranks), _delay_us(delay_us) {}
  void sendrecv(int, int, double const *send_low, double *recv_low, int64_t n_low, double const *send_high, double *recv_high,
                int64_t n_high, hipStream_t stream) override
  {
    if (n_low > 0 || n_high > 0)
      stream_delay(_delay_us, stream);
    if (n_low > 0)
      MFMG_HIP_CHECK(hipMemcpyAsync(recv_low, send_low, (size_t)n_low * sizeof(double), hipMemcpyDeviceToDevice, stream));
    if (n_high > 0)
      MFMG_HIP_CHECK(hipMemcpyAsync(recv_high, send_high, (size_t)n_high * sizeof(double), hipMemcpyDeviceToDevice, stream));
  }
  void exchange_many(int n, int const *, double const *const *send, double *const *recv, int64_t const *count,
                     hipStream_t stream) override
  {
    // (the segments of a box exchange are consecutive in the staging buffers: one copy when they are)
    int64_t total = 0;
    bool contiguous = n > 0;
    for (int i = 0; i < n; ++i)
    {
      contiguous = contiguous && send[i] == send[0] + total && recv[i] == recv[0] + total;
      total += count[i];
    }
    if (total > 0)
      stream_delay(_delay_us, stream);
    if (contiguous && total > 0)
      MFMG_HIP_CHECK(hipMemcpyAsync(recv[0], send[0], (size_t)total * sizeof(double), hipMemcpyDeviceToDevice, stream));
    else
      for (int i = 0; i < n; ++i)
        if (count[i] > 0)
          MFMG_HIP_CHECK(hipMemcpyAsync(recv[i], send[i], (size_t)count[i] * sizeof(double), hipMemcpyDeviceToDevice, stream));
  }
  void allreduce(double *, int, int, hipStream_t) override {}
  void allgather(double const *in, int64_t n, double *out, hipStream_t stream) override
  {
    stream_delay(_delay_us, stream);
    for (int r = 0; r < _n; ++r)
      MFMG_HIP_CHECK(hipMemcpyAsync(out + (size_t)r * n, in, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  }
  void loopback(double const *send, double *recv, int64_t n, hipStream_t stream) override
  {
    MFMG_HIP_CHECK(hipMemcpyAsync(recv, send, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  }
  char const *name() const override { return "reflecting"; }
  int comm_ranks() const override { return _n; }

private:
  int _n;
  double _delay_us;
};
} // namespace

std::shared_ptr<HaloTransport> make_reflecting_transport(int n_ranks, double delay_us)
{
  return std::make_shared<ReflectingTransport>(n_ranks, delay_us);
}

void rccl_available() { (void)rccl(); } // throws when librccl or one of its entry points cannot be resolved

void rccl_unique_id(unsigned char out[128])
{
  ncclUniqueId id;
  rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out, id.internal, 128);
}

std::shared_ptr<HaloTransport> make_rccl_transport(int rank, int n_ranks, unsigned char const unique_id[128])
{
  return std::make_shared<RcclTransport>(rank, n_ranks, unique_id);
}

std::shared_ptr<HaloTransport> make_host_transport(int rank, int n_ranks, mfmg_hip_host_exchange_fn sendrecv,
                                                   mfmg_hip_host_allreduce_fn allreduce, mfmg_hip_host_allgather_fn allgather,
                                                   void *user)
{
  return std::make_shared<HostTransport>(rank, n_ranks, sendrecv, allreduce, allgather, user);
}
} // namespace mfmg
