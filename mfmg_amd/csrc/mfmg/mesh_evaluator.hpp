// MeshEvaluator: mirror of include/mfmg/common/mesh_evaluator.hpp:19-32.
#pragma once

#include <string>
#include <type_traits>

namespace mfmg
{
class MeshEvaluator
{
public:
  virtual ~MeshEvaluator() = default;

  virtual int get_dim() const = 0;

  virtual std::string get_mesh_evaluator_type() const = 0;
};

template <typename T>
struct is_matrix_free : std::false_type
{
};
} // namespace mfmg
