/* TEST INFRASTRUCTURE -- host stand-in for <torch/extension.h>: just enough of at::Tensor / c10::ScalarType for the
 * host helpers at the top of the reference's common.cuh (PTR, NUM_OF_ELEMENT, CheckTensor) to PARSE.  None of them is
 * called by oracle/ref_common_shim.cc; only the scalar device functions are. */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <exception>
#include <string>
namespace c10 { enum class ScalarType : int { Float = 6, Int = 3 }; }
namespace at {
using c10::ScalarType;
constexpr ScalarType kFloat = ScalarType::Float;
constexpr ScalarType kInt = ScalarType::Int;
struct TypeMeta { ScalarType t; };
inline ScalarType typeMetaToScalarType(TypeMeta m) { return m.t; }
struct Tensor {
    void* p = nullptr; int64_t n = 0; ScalarType t = ScalarType::Float;
    template <typename T> T* data_ptr() const { return static_cast<T*>(p); }
    int64_t numel() const { return n; }
    TypeMeta dtype() const { return TypeMeta{t}; }
};
}  // namespace at
