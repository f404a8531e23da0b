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

#include <array>

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
/** Eigen::Array<T, Dynamic, 1>: an owning vector; converts to whatever 1-D view the mirror's signatures take. */
template <typename T>
class Arr1 {
   public:
    Arr1() = default;
    explicit Arr1(size_t n) : d_(n) {}
    size_t size() const { return d_.size(); }
    size_t rows() const { return d_.size(); }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    T& operator[](size_t i) { return d_[i]; }
    const T& operator[](size_t i) const { return d_[i]; }
    T& operator()(size_t i) { return d_[i]; }
    const T& operator()(size_t i) const { return d_[i]; }
    template <typename V, typename = decltype(V(std::declval<const T*>(), size_t{}))>
    operator V() const {
        return V(d_.data(), d_.size());
    }

   private:
    std::vector<T> d_;
};
template <typename T, int R, int C, int O>
struct ArraySel;
template <typename T>
struct ArraySel<T, Dynamic, Dynamic, RowMajor> {
    using type = ouster::sdk::core::ArrayXXR<T>;
};
template <typename T, int O>
struct ArraySel<T, Dynamic, 1, O> {
    using type = Arr1<T>;
};
}  // namespace facade
template <typename T, int R, int C, int O = ColMajor>
using Array = typename facade::ArraySel<T, R, C, O>::type;
template <typename T>
using ArrayX = facade::Arr1<T>;

/** Eigen::Tensor<T, N, RowMajor>: a dense n-d buffer; converts to the mirror's ArrayView<T, N> / ConstArrayView<T, N>. */
template <typename T, int N, int O = ColMajor>
class Tensor {
    static_assert(O == RowMajor, "the reference's tensors on this path are row-major");

   public:
    template <typename... D>
    explicit Tensor(D... dims) : shape_{static_cast<size_t>(dims)...} {
        static_assert(sizeof...(D) == N, "one extent per dimension");
        size_t n = 1;
        for (size_t d : shape_) n *= d;
        d_.assign(n, T{});
    }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    size_t dimension(size_t i) const { return shape_[i]; }
    size_t size() const { return d_.size(); }
    template <typename V, typename = decltype(V(std::declval<T*>(), std::declval<const std::array<size_t, N>&>()))>
    operator V() {
        return V(d_.data(), shape_);
    }
    template <typename V, typename = decltype(V(std::declval<const T*>(), std::declval<const std::array<size_t, N>&>())),
              typename = void>
    operator V() const {
        return V(d_.data(), shape_);
    }

   private:
    std::array<size_t, N> shape_;
    std::vector<T> d_;
};

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

// array arithmetic as the reference's tests write it on Eigen arrays: scalar * A, A + B (evaluated at once)
namespace ouster {
namespace sdk {
namespace core {
template <typename T>
ArrayX3R<T> operator*(double s, const ArrayX3R<T>& a) {
    ArrayX3R<T> r(a.rows());
    for (size_t i = 0; i < a.size(); ++i) r.data()[i] = static_cast<T>(s * a.data()[i]);
    return r;
}
template <typename T>
ArrayX3R<T> operator+(const ArrayX3R<T>& a, const ArrayX3R<T>& b) {
    if (a.rows() != b.rows()) throw std::invalid_argument("sum of arrays of different size");
    ArrayX3R<T> r(a.rows());
    for (size_t i = 0; i < a.size(); ++i) r.data()[i] = a.data()[i] + b.data()[i];
    return r;
}
}  // namespace core
}  // namespace sdk
}  // namespace ouster

#define OUSTER_FIELD_REF(T) ::Eigen::Ref<::ouster::sdk::core::ArrayXXR<T>>
#define OUSTER_CONST_FIELD_REF(T) ::Eigen::Ref<const ::ouster::sdk::core::ArrayXXR<T>>
