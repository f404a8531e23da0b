// TEST INFRASTRUCTURE (oracle/): stand-ins for the few types the REFERENCE's
//   /root/reference/ouster_core/include/ouster/core/impl/dewarp_impl.h
// touches, so that header -- the reference's own range-gated frame dewarp, a template over Eigen and LidarFrame --
// can be compiled from where it lies into oracle/_ref/libdewarp_ref.so (see oracle/dewarp_ref.cpp, oracle/Makefile)
// and the oracle's restatement ora_dewarp_frame_* can be checked against it.  Eigen3 is not installed in this
// image: the namespace Eigen below implements exactly the expressions dewarp_impl.h writes (Map<const MatrixX16dR>,
// Map<const Matrix4dR>, topLeftCorner<3,3>().cast<T>(), topRightCorner<3,1>().cast<T>(), Matrix3 * Vector3 + Vector3,
// Ref<const Header<T>>), with Eigen's evaluation order for the fixed-size 3-term dot product (its unrolled
// reduction is a binary tree: p0 + (p1 + p2)).  Nothing here is shipped or used by the product.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

namespace Eigen {
using Index = std::ptrdiff_t;

template <typename T>
struct Vector3 {
    T v[3];
    T& operator[](Index i) { return v[i]; }
    const T& operator[](Index i) const { return v[i]; }
    Vector3 operator+(const Vector3& o) const { return Vector3{{v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}}; }
};

template <typename X>
class Map;
template <typename X>
class Ref;
}  // namespace Eigen

namespace ouster {
namespace sdk {
namespace core {

namespace ChanField {
static constexpr const char* RANGE = "RANGE";
}

struct MatrixX16dR {};   // Eigen::Matrix<double, Dynamic, 16, RowMajor> in the reference (typedefs.h)
struct Matrix4dR {};     // Eigen::Matrix<double, 4, 4, RowMajor>

template <typename T>
struct Matrix3R {        // Eigen::Matrix<T, 3, 3, RowMajor>
    T m[9];
    Eigen::Vector3<T> operator*(const Eigen::Vector3<T>& p) const {
        Eigen::Vector3<T> r;
        for (int i = 0; i < 3; ++i) {
            const T p0 = m[3 * i] * p.v[0], p1 = m[3 * i + 1] * p.v[1], p2 = m[3 * i + 2] * p.v[2];
            r.v[i] = p0 + (p1 + p2);   // Eigen's redux_novec_unroller<.., 0, 3>: func(term 0, func(term 1, term 2))
        }
        return r;
    }
};

template <typename T>
class PointCloudXYZ {    // Eigen::Array<T, Dynamic, 3> in the reference
   public:
    explicit PointCloudXYZ(size_t n) : d_(n * 3) {}
    T* data() { return d_.data(); }
    Eigen::Vector3<T> row(Eigen::Index i) const {
        return Eigen::Vector3<T>{{d_[3 * i], d_[3 * i + 1], d_[3 * i + 2]}};
    }

   private:
    std::vector<T> d_;
};

template <typename T>
struct img_view {        // what LidarFrame::field<T>() hands out: a row-major h x w view
    const T* p;
    Eigen::Index h, w;
    Eigen::Index rows() const { return h; }
    Eigen::Index cols() const { return w; }
    const T& operator()(Eigen::Index y, Eigen::Index x) const { return p[y * w + x]; }
};

struct pose_field {
    const double* p;
    template <typename T>
    const T* get() const { return reinterpret_cast<const T*>(p); }
};

class LidarFrame {
   public:
    template <typename T>
    struct Header {
        const T* p;
        Eigen::Index n;
        Eigen::Index size() const { return n; }
        const T& operator[](Eigen::Index i) const { return p[i]; }
    };
    size_t w = 0, h = 0;
    const uint32_t* range_ = nullptr;
    const uint32_t* status_ = nullptr;
    const uint64_t* timestamp_ = nullptr;
    const double* poses_ = nullptr;

    template <typename T>
    img_view<T> field(const char*) const { return img_view<T>{reinterpret_cast<const T*>(range_), (Eigen::Index)h, (Eigen::Index)w}; }
    pose_field body_to_world() const { return pose_field{poses_}; }
    Header<uint32_t> status() const { return Header<uint32_t>{status_, (Eigen::Index)w}; }
    Header<uint64_t> timestamp() const { return Header<uint64_t>{timestamp_, (Eigen::Index)w}; }
    // ouster_core/src/lidar_frame.cpp:907-925
    int get_first_valid_column() const {
        for (int i = 0; i < (int)w; ++i)
            if ((status_[i] & 1) > 0) return i;
        throw std::runtime_error("No valid columns in LidarFrame");
    }
    int get_last_valid_column() const {
        for (int i = (int)w - 1; i >= 0; --i)
            if ((status_[i] & 1) > 0) return i;
        throw std::runtime_error("No valid columns in LidarFrame");
    }
};

class FrameSet {
   public:
    std::vector<std::shared_ptr<LidarFrame>> frames;
    size_t size() const { return frames.size(); }
    std::vector<size_t> valid_indices() const {
        std::vector<size_t> v;
        for (size_t i = 0; i < frames.size(); ++i)
            if (frames[i]) v.push_back(i);
        return v;
    }
    const std::shared_ptr<LidarFrame>& operator[](size_t i) const { return frames[i]; }
};

}  // namespace core
}  // namespace sdk
}  // namespace ouster

namespace Eigen {
template <>
class Map<const ouster::sdk::core::MatrixX16dR> {
   public:
    struct RowView {
        const double* p;
        const double* data() const { return p; }
    };
    Map(const double* p, Index rows, Index cols) : p_(p), rows_(rows), cols_(cols) {}
    RowView row(Index x) const { return RowView{p_ + x * cols_}; }

   private:
    const double* p_;
    Index rows_, cols_;
};

template <>
class Map<const ouster::sdk::core::Matrix4dR> {
   public:
    template <int R, int C>
    struct Block {
        const double* p;   // row stride 4
        template <typename T>
        auto cast() const {
            if constexpr (R == 3 && C == 3) {
                ouster::sdk::core::Matrix3R<T> m;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) m.m[3 * i + j] = static_cast<T>(p[4 * i + j]);
                return m;
            } else {
                static_assert(R == 3 && C == 1, "only the blocks dewarp_impl.h takes");
                return Vector3<T>{{static_cast<T>(p[0]), static_cast<T>(p[4]), static_cast<T>(p[8])}};
            }
        }
    };
    explicit Map(const double* p) : p_(p) {}
    template <int R, int C>
    Block<R, C> topLeftCorner() const { return Block<R, C>{p_}; }
    template <int R, int C>
    Block<R, C> topRightCorner() const { return Block<R, C>{p_ + (4 - C)}; }

   private:
    const double* p_;
};

template <typename T>
class Ref<const ouster::sdk::core::LidarFrame::Header<T>> {
   public:
    Ref(const ouster::sdk::core::LidarFrame::Header<T>& h) : h_(h) {}
    const T& operator[](Index i) const { return h_[i]; }
    Index size() const { return h_.size(); }

   private:
    ouster::sdk::core::LidarFrame::Header<T> h_;
};
}  // namespace Eigen
