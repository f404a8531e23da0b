// typedefs.h -- array types of the ouster::sdk::core API.
//
// The reference (ouster_core/include/ouster/core/typedefs.h) aliases Eigen types:
//   img_t<T>           = Eigen::Array<T, Dynamic, Dynamic, RowMajor>
//   ArrayX3R<T>        = Eigen::Array<T, Dynamic, 3, RowMajor>
//   PointCloudXYZ<T>   = ArrayX3R<T>
//   mat4d              = Eigen::Matrix<double, 4, 4>   (used with operator()(r, c))
// Eigen is only a container on this path (the hot loops use .data(), see
// impl/cartesian.h:43-48, impl/lidar_frame_impl.h:750-758), so this header ships a minimal
// row-major stand-in with the same observable interface (.data() .rows() .cols() .size()
// operator()(r,c)).  Row-major storage is part of the API contract: XYZ index i = row*W+col.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace ouster {
namespace sdk {
namespace core {

/** Dense row-major 2-D array owning its storage (zero initialised). */
template <typename T>
class ArrayXXR {
   public:
    using Scalar = T;
    ArrayXXR() = default;
    ArrayXXR(size_t rows, size_t cols) : rows_(rows), cols_(cols), d_(rows * cols) {}
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    size_t size() const { return d_.size(); }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    T& operator()(size_t r, size_t c) { return d_[r * cols_ + c]; }
    const T& operator()(size_t r, size_t c) const { return d_[r * cols_ + c]; }
    void resize(size_t rows, size_t cols) {
        rows_ = rows;
        cols_ = cols;
        d_.assign(rows * cols, T{});
    }
    void setZero() { std::fill(d_.begin(), d_.end(), T{}); }
    void setConstant(T v) { std::fill(d_.begin(), d_.end(), v); }
    bool operator==(const ArrayXXR& o) const {
        return rows_ == o.rows_ && cols_ == o.cols_ && d_ == o.d_;
    }
    bool operator!=(const ArrayXXR& o) const { return !(*this == o); }
    template <typename U>
    ArrayXXR<U> cast() const {
        ArrayXXR<U> r(rows_, cols_);
        for (size_t i = 0; i < d_.size(); ++i) r.data()[i] = static_cast<U>(d_[i]);
        return r;
    }

   private:
    size_t rows_ = 0, cols_ = 0;
    std::vector<T> d_;
};

template <typename T>
using img_t = ArrayXXR<T>;

/** N x 3 row-major array (direction / offset tables and point clouds). */
template <typename T>
class ArrayX3R : public ArrayXXR<T> {
   public:
    ArrayX3R() = default;
    explicit ArrayX3R(size_t rows) : ArrayXXR<T>(rows, 3) {}
    ArrayX3R(size_t rows, size_t cols) : ArrayXXR<T>(rows, cols) {
        if (cols != 3) throw std::invalid_argument("ArrayX3R needs 3 columns");
    }
    template <typename U>
    ArrayX3R<U> cast() const {
        ArrayX3R<U> r(this->rows());
        for (size_t i = 0; i < this->size(); ++i) r.data()[i] = static_cast<U>(this->data()[i]);
        return r;
    }
};

template <typename T>
using PointCloudXYZ = ArrayX3R<T>;
using PointCloudXYZf = PointCloudXYZ<float>;
using PointCloudXYZd = PointCloudXYZ<double>;

/** Non-owning row-major 2-D view; stands in for Eigen::Ref<img_t<T>>. */
template <typename T>
class ImgRef {
   public:
    using NC = typename std::remove_const<T>::type;
    ImgRef(T* data, size_t rows, size_t cols) : p_(data), rows_(rows), cols_(cols) {}
    ImgRef(ArrayXXR<NC>& a) : p_(a.data()), rows_(a.rows()), cols_(a.cols()) {}
    template <typename U = T, typename = typename std::enable_if<std::is_const<U>::value>::type>
    ImgRef(const ArrayXXR<NC>& a) : p_(a.data()), rows_(a.rows()), cols_(a.cols()) {}
    template <typename U = T, typename = typename std::enable_if<std::is_const<U>::value>::type>
    ImgRef(const ImgRef<NC>& o) : p_(o.data()), rows_(o.rows()), cols_(o.cols()) {}
    T* data() const { return p_; }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    size_t size() const { return rows_ * cols_; }
    T& operator()(size_t r, size_t c) const { return p_[r * cols_ + c]; }

   private:
    T* p_;
    size_t rows_, cols_;
};

/** 4x4 double matrix with (row, col) access. */
struct mat4d {
    double m[16];
    static mat4d Zero() {
        mat4d r;
        std::memset(r.m, 0, sizeof r.m);
        return r;
    }
    static mat4d Identity() {
        mat4d r = Zero();
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
        return r;
    }
    static mat4d FromRowMajor(const double* v) {
        mat4d r;
        std::memcpy(r.m, v, sizeof r.m);
        return r;
    }
    double& operator()(int r, int c) { return m[r * 4 + c]; }
    double operator()(int r, int c) const { return m[r * 4 + c]; }
    const double* data() const { return m; }  ///< row-major
    mat4d operator*(const mat4d& o) const {
        mat4d r = Zero();
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 4; ++k) r.m[i * 4 + j] += m[i * 4 + k] * o.m[k * 4 + j];
        return r;
    }
    bool operator==(const mat4d& o) const { return std::memcmp(m, o.m, sizeof m) == 0; }
};

}  // namespace core
}  // namespace sdk
}  // namespace ouster
