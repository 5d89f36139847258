// Device vector with the member names Hierarchy and the drivers use on
// dealii::LinearAlgebra::distributed::Vector<double, MemorySpace::CUDA>
// (include/mfmg/common/hierarchy.hpp:258,286,302; tests/test_hierarchy.cc:103-114):
// operator=(scalar), add, sadd, l2_norm, local_size, size, get_values, copy-construction.
// Owning (hipMalloc) or a borrowed view over caller memory (the C ABI hands in raw
// device pointers).
#pragma once

#include "../common.hpp"
#include "../vector_ops.hpp"

namespace mfmg
{
template <typename T>
class Vector
{
public:
  using value_type = T;

  Vector(HipHandle &handle, int64_t n) : _handle(&handle), _n(n), _storage(n), _ptr(_storage.data())
  {
    vec::set<T>(*_handle, _n, T(0), _ptr);
  }
  // borrowed view
  Vector(HipHandle &handle, int64_t n, T *device_ptr) : _handle(&handle), _n(n), _ptr(device_ptr) {}
  // deep copy (vector_type r(b) in the smoother wrappers)
  Vector(Vector const &other) : _handle(other._handle), _n(other._n), _storage(other._n), _ptr(_storage.data())
  {
    vec::copy<T>(*_handle, _n, other._ptr, _ptr);
  }
  Vector(Vector &&) = default;
  Vector &operator=(Vector const &other)
  {
    ASSERT_THROW(_n == other._n, "Vector size mismatch in assignment");
    vec::copy<T>(*_handle, _n, other._ptr, _ptr);
    return *this;
  }
  Vector &operator=(T s)
  {
    vec::set<T>(*_handle, _n, s, _ptr);
    return *this;
  }
  void add(T a, Vector const &v)
  {
    ASSERT_THROW(_n == v._n, "Vector size mismatch in add");
    vec::add<T>(*_handle, _n, a, v._ptr, _ptr);
  }
  void sadd(T s, T a, Vector const &v)
  {
    ASSERT_THROW(_n == v._n, "Vector size mismatch in sadd");
    vec::sadd<T>(*_handle, _n, s, a, v._ptr, _ptr);
  }
  double l2_norm() const { return vec::l2_norm<T>(*_handle, _n, _ptr); }
  double operator*(Vector const &v) const { return vec::dot<T>(*_handle, _n, _ptr, v._ptr); }
  int64_t size() const { return _n; }
  int64_t local_size() const { return _n; }
  T *get_values() { return _ptr; }
  T const *get_values() const { return _ptr; }
  HipHandle &handle() const { return *_handle; }
  bool owns() const { return _storage.size() > 0; }

private:
  HipHandle *_handle;
  int64_t _n;
  DeviceBuffer<T> _storage;
  T *_ptr;
};
} // namespace mfmg
