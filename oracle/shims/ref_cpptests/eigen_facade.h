// eigen_facade.h -- test infrastructure only.  The reference's C++ tests spell array types through Eigen
// (`Eigen::Ref<img_t<T>>`, `Eigen::Array<T, Dynamic, Dynamic, RowMajor>`), which this image does not have; the mirror of
// the ouster_core API under include/ouster/core/ ships Eigen-free stand-ins instead (typedefs.h).  This header, force-
// included when oracle/Makefile compiles the reference's tests from where they lie, gives those Eigen spellings a
// meaning on top of the stand-ins -- nothing more than what the tests use -- so that the test sources need no edit:
//   Eigen::Ref<img_t<T>>        : ImgRef<T>        (+ assignment of a scalar = fill, as for an Eigen block)
//   Eigen::Ref<const img_t<T>>  : ImgRef<const T>
//   Eigen::Array<T, Dynamic, Dynamic, RowMajor> = img_t<T>
// and tells visit_field / foreach_channel_field (lidar_frame.h) to hand operations the Eigen::Ref spelling.
#pragma once

#include "ouster/core/typedefs.h"

namespace Eigen {

enum : int { Dynamic = -1, ColMajor = 0, RowMajor = 1 };
using Index = std::ptrdiff_t;

template <typename X>
class Ref;

template <typename T>
class Ref<ouster::sdk::core::ArrayXXR<T>> : public ouster::sdk::core::ImgRef<T> {
   public:
    using Base = ouster::sdk::core::ImgRef<T>;
    using Scalar = T;
    Ref(const Base& b) : Base(b) {}
    Ref(ouster::sdk::core::ArrayXXR<T>& a) : Base(a) {}
    template <typename F, typename = decltype(static_cast<Base>(std::declval<F&>()))>
    Ref(F& field) : Base(static_cast<Base>(field)) {}
    const Ref& operator=(T v) const {   // an Eigen block assigned a scalar is filled with it
        this->setConstant(v);
        return *this;
    }
};

template <typename T>
class Ref<const ouster::sdk::core::ArrayXXR<T>> : public ouster::sdk::core::ImgRef<const T> {
   public:
    using Base = ouster::sdk::core::ImgRef<const T>;
    using Scalar = T;
    Ref(const Base& b) : Base(b) {}
    Ref(const ouster::sdk::core::ImgRef<T>& b) : Base(b) {}
    Ref(const ouster::sdk::core::ArrayXXR<T>& a) : Base(a) {}
    template <typename F, typename = decltype(static_cast<Base>(std::declval<const F&>()))>
    Ref(const F& field) : Base(static_cast<Base>(field)) {}
};

namespace facade {
template <typename T, int R, int C, int O>
struct ArraySel;
template <typename T>
struct ArraySel<T, Dynamic, Dynamic, RowMajor> {
    using type = ouster::sdk::core::ArrayXXR<T>;
};
}  // namespace facade
template <typename T, int R, int C, int O = ColMajor>
using Array = typename facade::ArraySel<T, R, C, O>::type;

}  // namespace Eigen

namespace ouster {
namespace sdk {
namespace core {
namespace impl {
template <typename T> struct is_array_like<Eigen::Ref<ArrayXXR<T>>> : std::true_type {};
template <typename T> struct is_array_like<Eigen::Ref<const ArrayXXR<T>>> : std::true_type {};
}  // namespace impl
}  // namespace core
}  // namespace sdk
}  // namespace ouster

#define OUSTER_FIELD_REF(T) ::Eigen::Ref<::ouster::sdk::core::ArrayXXR<T>>
#define OUSTER_CONST_FIELD_REF(T) ::Eigen::Ref<const ::ouster::sdk::core::ArrayXXR<T>>
