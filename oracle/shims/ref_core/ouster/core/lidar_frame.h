// TEST INFRASTRUCTURE (oracle/): the handful of ouster / Eigen names the reference's impl/cartesian.h and the
// destagger_into<T> of impl/lidar_frame_impl.h touch, so that both compile from where they lie in /root/reference
// (oracle/Makefile, target _ref/libcore_ref.so; Eigen3 is absent from this image).  Both functions only ever use
// .data(), .rows(), .cols(), .size() of their arguments, which is all these stand-ins provide.  Never used by the product.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace ouster {
namespace sdk {
namespace core {

// row-major [rows][cols] view (Eigen::Array<T, Dynamic, Dynamic, RowMajor> in the reference)
template <typename T>
struct img_t {
    using Scalar = T;
    T* p = nullptr;
    long r = 0, c = 0;
    img_t() = default;
    img_t(T* data, long rows, long cols) : p(data), r(rows), c(cols) {}
    T* data() { return p; }
    const T* data() const { return p; }
    long rows() const { return r; }
    long cols() const { return c; }
    long size() const { return r * c; }
};

// [n][3] row-major (Eigen::Array<T, Dynamic, 3, RowMajor>)
template <typename T>
struct ArrayX3R {
    using Scalar = T;
    T* p = nullptr;
    long r = 0;
    std::vector<T> own;
    ArrayX3R() = default;
    ArrayX3R(T* data, long rows) : p(data), r(rows) {}
    ArrayX3R(long rows, int) : r(rows), own(static_cast<size_t>(rows) * 3) { p = own.data(); }
    T* data() { return p; }
    const T* data() const { return p; }
    long rows() const { return r; }
    long cols() const { return 3; }
    long size() const { return r * 3; }
};
template <typename T>
using PointCloudXYZ = ArrayX3R<T>;

namespace ChanField {
static const char* const RANGE = "RANGE";
}

struct LidarFrame {
    img_t<uint32_t> range;
    const img_t<uint32_t>& field(const std::string&) const { return range; }
};

}  // namespace core
}  // namespace sdk
}  // namespace ouster

namespace Eigen {
// Eigen::Ref<M> / Eigen::Ref<const M>: a non-owning reference to M's storage
template <typename M>
class Ref {
    using Plain = typename std::remove_const<M>::type;
    using Scalar = typename Plain::Scalar;
    using Ptr = typename std::conditional<std::is_const<M>::value, const Scalar*, Scalar*>::type;
    Ptr p_;
    long r_, c_;

   public:
    Ref(Plain& m) : p_(m.data()), r_(m.rows()), c_(m.cols()) {}
    template <typename Q = M, typename = typename std::enable_if<std::is_const<Q>::value>::type>
    Ref(const Plain& m) : p_(m.data()), r_(m.rows()), c_(m.cols()) {}
    Ptr data() const { return p_; }
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
};
}  // namespace Eigen
