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
//
// -DOUSTER_HIP_USE_EIGEN (with Eigen3 on the include path) adds the real Eigen types at the API
// boundary for callers written against the reference:  EigenImg<T> = the reference's img_t<T>,
// EigenX3R<T> = its ArrayX3R<T>.  Every stand-in then converts from / to them implicitly
// (ArrayXXR <-> EigenImg by copy, ImgRef from any dense row-major Eigen expression or Eigen::Ref
// without a copy, `.eigen()` maps a stand-in's storage as an Eigen::Map), and Field grows the
// reference's `operator Eigen::Ref<img_t<T>>` (field.h:433-470).  The compiled library's ABI does
// not change with the switch: it always speaks the stand-ins.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <new>
#include <stdexcept>
#include <type_traits>
#include <vector>

#ifdef OUSTER_HIP_USE_EIGEN
#include <Eigen/Core>
#endif

namespace ouster {
namespace sdk {
namespace core {

#ifdef OUSTER_HIP_USE_EIGEN
/** The reference's img_t<T> / ArrayX3R<T> (ouster_core/include/ouster/core/typedefs.h). */
template <typename T>
using EigenImg = Eigen::Array<T, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
template <typename T>
using EigenX3R = Eigen::Array<T, Eigen::Dynamic, 3, Eigen::RowMajor>;
#endif

template <typename T> class ImgRef;
template <typename T> class VecRef;

namespace impl {
/** Host memory of the containers below and of Field.  From OUSTER_HIP_HOST_POOL_MIN bytes on it comes from the library's
 *  pool of page-locked blocks (include/ouster_hip.h, ouster_hip_host_alloc): the GPU reads and writes such memory in place,
 *  so destagger / XYZLut() / FrameBatcher on these containers are one kernel launch with no staging copy, and a block that
 *  is freed is handed out again instead of being returned to the system.  Smaller requests, and every request on a machine
 *  without a GPU, are plain heap memory. */
void* host_alloc(size_t bytes, bool zero);
void host_free(void* p, size_t bytes) noexcept;
/** Tag: allocate without clearing (the caller overwrites every element). */
struct uninitialized_t {};
constexpr uninitialized_t uninitialized{};

/** The storage of ArrayXXR: a fixed-size run of trivially copyable elements in host_alloc memory. */
template <typename T>
class HostArray {
    static_assert(std::is_trivially_copyable<T>::value, "HostArray holds plain data");

   public:
    HostArray() = default;
    explicit HostArray(size_t n) : p_(n ? static_cast<T*>(host_alloc(n * sizeof(T), true)) : nullptr), n_(n) { check(); }
    HostArray(size_t n, uninitialized_t) : p_(n ? static_cast<T*>(host_alloc(n * sizeof(T), false)) : nullptr), n_(n) { check(); }
    HostArray(const HostArray& o) : HostArray(o.n_, uninitialized) {
        if (n_) std::memcpy(static_cast<void*>(p_), o.p_, n_ * sizeof(T));
    }
    HostArray(HostArray&& o) noexcept : p_(o.p_), n_(o.n_) {
        o.p_ = nullptr;
        o.n_ = 0;
    }
    HostArray& operator=(HostArray o) noexcept {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        return *this;
    }
    ~HostArray() { host_free(p_, n_ * sizeof(T)); }
    size_t size() const { return n_; }
    T* data() { return p_; }
    const T* data() const { return p_; }
    T* begin() { return p_; }
    T* end() { return p_ + n_; }
    const T* begin() const { return p_; }
    const T* end() const { return p_ + n_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
    void assign(size_t n, const T& v) {
        if (n != n_) *this = HostArray(n, uninitialized);
        std::fill(p_, p_ + n_, v);
    }
    bool operator==(const HostArray& o) const { return n_ == o.n_ && std::equal(p_, p_ + n_, o.p_); }

   private:
    void check() const {
        if (n_ && !p_) throw std::bad_alloc();
    }
    T* p_ = nullptr;
    size_t n_ = 0;
};
}  // namespace impl

/** Dense row-major 2-D array owning its storage (zero initialised). */
template <typename T>
class ArrayXXR {
   public:
    using Scalar = T;
    ArrayXXR() = default;
    ArrayXXR(size_t rows, size_t cols) : rows_(rows), cols_(cols), d_(rows * cols) {}
    /** Not cleared: for results that are overwritten in full (what the library's own calls return). */
    ArrayXXR(size_t rows, size_t cols, impl::uninitialized_t) : rows_(rows), cols_(cols), d_(rows * cols, impl::uninitialized) {}
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
    // ---- the handful of Eigen::Array conveniences the reference's callers lean on -------------------------------
    static ArrayXXR Zero(size_t rows, size_t cols) { return ArrayXXR(rows, cols); }
    static ArrayXXR Constant(size_t rows, size_t cols, T v) {
        ArrayXXR a(rows, cols);
        a.setConstant(v);
        return a;
    }
    /** Pseudo-random fill: integers over their whole range, floating point in [-1, 1] (as Eigen's Random()). */
    static ArrayXXR Random(size_t rows, size_t cols) {
        ArrayXXR a(rows, cols);
        for (auto& v : a.d_) v = random_value();
        return a;
    }
    /** Equal shapes and element-wise equal (integers) / equal within `prec` relative to the smaller norm (floating point). */
    template <typename O>
    bool isApprox(const O& o, double prec = 1e-12) const {
        if (rows() != o.rows() || cols() != o.cols()) return false;
        if (std::is_integral<T>::value) {
            for (size_t i = 0; i < size(); ++i)
                if (!(d_[i] == o.data()[i])) return false;
            return true;
        }
        if (std::is_same<T, float>::value && prec == 1e-12) prec = 1e-5;
        double diff = 0, na = 0, nb = 0;
        for (size_t i = 0; i < size(); ++i) {
            const double a = static_cast<double>(d_[i]), b = static_cast<double>(o.data()[i]);
            diff += (a - b) * (a - b);
            na += a * a;
            nb += b * b;
        }
        return diff <= prec * prec * std::min(na, nb);
    }
    /** op(element) for every element, as a new array. */
    template <typename F>
    ArrayXXR unaryExpr(F f) const {
        ArrayXXR r(rows_, cols_);
        for (size_t i = 0; i < d_.size(); ++i) r.d_[i] = static_cast<T>(f(d_[i]));
        return r;
    }
    VecRef<T> row(size_t r);
    VecRef<const T> row(size_t r) const;
    VecRef<T> col(size_t c);
    VecRef<const T> col(size_t c) const;
    /** View of `r` whole rows from row `i` on (a block that is not one dense run of memory is refused).
     *  @throw std::invalid_argument */
    ImgRef<T> block(size_t i, size_t j, size_t r, size_t c);
    T& operator()(size_t i) { return d_[i]; }   ///< flat, row-major
    const T& operator()(size_t i) const { return d_[i]; }
    T& operator[](size_t i) { return d_[i]; }
    const T& operator[](size_t i) const { return d_[i]; }
#ifdef OUSTER_HIP_USE_EIGEN
    /** from any dense Eigen expression (copied element by element: any storage order) */
    template <typename D>
    ArrayXXR(const Eigen::DenseBase<D>& e)
        : rows_(static_cast<size_t>(e.rows())), cols_(static_cast<size_t>(e.cols())), d_(rows_ * cols_) {
        for (size_t r = 0; r < rows_; ++r)
            for (size_t c = 0; c < cols_; ++c) d_[r * cols_ + c] = static_cast<T>(e.derived().coeff(r, c));
    }
    /** the storage as an Eigen array, no copy */
    Eigen::Map<EigenImg<T>> eigen() { return Eigen::Map<EigenImg<T>>(d_.data(), rows_, cols_); }
    Eigen::Map<const EigenImg<T>> eigen() const { return Eigen::Map<const EigenImg<T>>(d_.data(), rows_, cols_); }
    operator EigenImg<T>() const { return eigen(); }
#endif

   private:
    static T random_value() {
        static uint64_t state = 0x9E3779B97F4A7C15ull;   // splitmix64: deterministic, good enough for test data
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (std::is_floating_point<T>::value) return static_cast<T>(static_cast<double>(z >> 11) / 4503599627370496.0 - 1.0);
        return static_cast<T>(z);
    }
    size_t rows_ = 0, cols_ = 0;
    impl::HostArray<T> d_;
};

template <typename T>
using img_t = ArrayXXR<T>;

/** N x 3 row-major array (direction / offset tables and point clouds). */
template <typename T>
class ArrayX3R : public ArrayXXR<T> {
   public:
    ArrayX3R() = default;
    explicit ArrayX3R(size_t rows) : ArrayXXR<T>(rows, 3) {}
    ArrayX3R(size_t rows, impl::uninitialized_t) : ArrayXXR<T>(rows, 3, impl::uninitialized) {}
    ArrayX3R(size_t rows, size_t cols) : ArrayXXR<T>(rows, cols) {
        if (cols != 3) throw std::invalid_argument("ArrayX3R needs 3 columns");
    }
    template <typename U>
    ArrayX3R<U> cast() const {
        ArrayX3R<U> r(this->rows());
        for (size_t i = 0; i < this->size(); ++i) r.data()[i] = static_cast<U>(this->data()[i]);
        return r;
    }
    ArrayX3R(const ArrayXXR<T>& a) : ArrayXXR<T>(a) {
        if (a.cols() != 3) throw std::invalid_argument("ArrayX3R needs 3 columns");
    }
    static ArrayX3R Zero(size_t rows, size_t cols = 3) { return ArrayX3R(rows, cols); }
    static ArrayX3R Constant(size_t rows, size_t cols, T v) { return ArrayX3R(ArrayXXR<T>::Constant(rows, cols, v)); }
    static ArrayX3R Random(size_t rows, size_t cols = 3) { return ArrayX3R(ArrayXXR<T>::Random(rows, cols)); }
#ifdef OUSTER_HIP_USE_EIGEN
    template <typename D>
    ArrayX3R(const Eigen::DenseBase<D>& e) : ArrayXXR<T>(e) {
        if (e.cols() != 3) throw std::invalid_argument("ArrayX3R needs 3 columns");
    }
    operator EigenX3R<T>() const {
        return Eigen::Map<const EigenX3R<T>>(this->data(), static_cast<Eigen::Index>(this->rows()), 3);
    }
#endif
};

using ArrayX3dR = ArrayX3R<double>;   ///< typedefs.h of the reference
using ArrayX3fR = ArrayX3R<float>;

template <typename T>
using PointCloudXYZ = ArrayX3R<T>;
using PointCloudXYZf = PointCloudXYZ<float>;
using PointCloudXYZd = PointCloudXYZ<double>;

/** Result of an element-wise comparison (`a == b`, `a != 7`): the reductions the reference's callers write on Eigen
 *  expressions -- `(a == b).all()`, `.any()`, `.count()`. */
class BoolMask {
   public:
    explicit BoolMask(size_t n) : v_(n, 0) {}
    size_t size() const { return v_.size(); }
    bool operator()(size_t i) const { return v_[i] != 0; }
    void set(size_t i, bool b) { v_[i] = b ? 1 : 0; }
    bool all() const { return std::all_of(v_.begin(), v_.end(), [](uint8_t b) { return b != 0; }); }
    bool any() const { return std::any_of(v_.begin(), v_.end(), [](uint8_t b) { return b != 0; }); }
    size_t count() const { return static_cast<size_t>(std::count_if(v_.begin(), v_.end(), [](uint8_t b) { return b != 0; })); }

   private:
    std::vector<uint8_t> v_;
};

/** Non-owning strided 1-D view: a column or row of an image, a segment of a header (Eigen's .col() / .row() / .segment()). */
template <typename T>
class VecRef {
   public:
    VecRef(T* p, size_t n, size_t stride = 1) : p_(p), n_(n), stride_(stride) {}
    size_t size() const { return n_; }
    size_t rows() const { return n_; }
    T* data() const { return p_; }           ///< the first element; the next one is stride() elements further on
    size_t stride() const { return stride_; }
    size_t innerStride() const { return stride_; }
    T& operator()(size_t i) const { return p_[i * stride_]; }
    T& operator[](size_t i) const { return p_[i * stride_]; }
    VecRef segment(size_t start, size_t n) const {
        if (start + n > n_) throw std::out_of_range("segment outside the vector");
        return VecRef(p_ + start * stride_, n, stride_);
    }
    const VecRef& operator=(typename std::remove_const<T>::type v) const {   ///< fill, as assigning a scalar to an Eigen block does
        for (size_t i = 0; i < n_; ++i) p_[i * stride_] = v;
        return *this;
    }
    /** element-wise copy from a view of the same length (Eigen: block = block).  @throw std::invalid_argument */
    template <typename U>
    const VecRef& assign(const VecRef<U>& o) const {
        if (o.size() != n_) throw std::invalid_argument("assignment between views of different length");
        for (size_t i = 0; i < n_; ++i) p_[i * stride_] = o(i);
        return *this;
    }
    const VecRef& operator=(const VecRef& o) const { return assign(o); }
    template <typename U, typename = typename std::enable_if<!std::is_same<U, T>::value>::type>
    const VecRef& operator=(const VecRef<U>& o) const { return assign(o); }
    VecRef(const VecRef&) = default;
    void setZero() const { *this = T{}; }
    size_t count() const {   ///< non-zero entries
        size_t c = 0;
        for (size_t i = 0; i < n_; ++i) c += p_[i * stride_] != T{};
        return c;
    }
    bool all() const { return count() == n_; }   ///< every entry non-zero (Eigen's .all() of an integer array)
    bool any() const { return count() != 0; }

   private:
    T* p_;
    size_t n_, stride_;
};

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
#ifdef OUSTER_HIP_USE_EIGEN
    /** A view of an Eigen array / Map / Ref / block whose storage is one dense row-major block
     *  (what Eigen::Ref<img_t<T>> guarantees in the reference's signatures).
     *  @throw std::invalid_argument for a strided or column-major expression */
    template <typename D, typename = typename std::enable_if<
                              std::is_same<typename D::Scalar, NC>::value &&
                              (std::is_const<T>::value || !std::is_const<D>::value)>::type>
    ImgRef(D& e, decltype(std::declval<D&>().data())* = nullptr)
        : p_(e.data()), rows_(static_cast<size_t>(e.rows())), cols_(static_cast<size_t>(e.cols())) {
        if (!(D::IsRowMajor || e.rows() == 1 || e.cols() == 1) ||
            (e.rows() > 1 && static_cast<size_t>(e.outerStride()) != cols_) || e.innerStride() != 1)
            throw std::invalid_argument("ImgRef: the Eigen expression is not a dense row-major image");
    }
    Eigen::Map<EigenImg<NC>> eigen() const { return Eigen::Map<EigenImg<NC>>(const_cast<NC*>(p_), rows_, cols_); }
#endif
    T* data() const { return p_; }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    size_t size() const { return rows_ * cols_; }
    T& operator()(size_t r, size_t c) const { return p_[r * cols_ + c]; }
    T& operator()(size_t i) const { return p_[i]; }   ///< flat, row-major
    VecRef<T> col(size_t c) const { return VecRef<T>(p_ + c, rows_, cols_); }
    VecRef<T> row(size_t r) const { return VecRef<T>(p_ + r * cols_, cols_, 1); }
    void setConstant(NC v) const { std::fill(p_, p_ + size(), v); }
    const ImgRef& operator=(NC v) const {   ///< fill, as assigning a scalar to an Eigen::Ref of an array does
        setConstant(v);
        return *this;
    }
    ImgRef(const ImgRef&) = default;
    void setZero() const { setConstant(NC{}); }
    size_t count() const { return static_cast<size_t>(size() - std::count(p_, p_ + size(), NC{})); }
    bool all() const { return count() == size(); }
    bool any() const { return count() != 0; }

   private:
    T* p_;
    size_t rows_, cols_;
};

template <typename T> VecRef<T> ArrayXXR<T>::row(size_t r) { return VecRef<T>(data() + r * cols_, cols_, 1); }
template <typename T> VecRef<const T> ArrayXXR<T>::row(size_t r) const { return VecRef<const T>(data() + r * cols_, cols_, 1); }
template <typename T> VecRef<T> ArrayXXR<T>::col(size_t c) { return VecRef<T>(data() + c, rows_, cols_); }
template <typename T> VecRef<const T> ArrayXXR<T>::col(size_t c) const { return VecRef<const T>(data() + c, rows_, cols_); }
template <typename T>
ImgRef<T> ArrayXXR<T>::block(size_t i, size_t j, size_t r, size_t c) {
    if (i + r > rows_ || j + c > cols_) throw std::out_of_range("block outside the array");
    if (j != 0 || c != cols_) throw std::invalid_argument("block: only whole rows form one dense run of memory");
    return ImgRef<T>(data() + i * cols_, r, c);
}

// ---- element-wise comparisons of the array stand-ins -> BoolMask -------------------------------------------------
namespace impl {
template <typename X> struct is_array_like : std::false_type {};
template <typename T> struct is_array_like<ArrayXXR<T>> : std::true_type {};
template <typename T> struct is_array_like<ArrayX3R<T>> : std::true_type {};
template <typename T> struct is_array_like<ImgRef<T>> : std::true_type {};
template <typename T> struct is_array_like<VecRef<T>> : std::true_type {};
template <typename T> const T& flat_at(const ArrayXXR<T>& a, size_t i) { return a.data()[i]; }
template <typename T> T& flat_at(const ImgRef<T>& a, size_t i) { return a.data()[i]; }
template <typename T> T& flat_at(const VecRef<T>& a, size_t i) { return a(i); }
template <typename A, typename B, typename F>
BoolMask compare_arrays(const A& a, const B& b, F f) {
    if (a.size() != b.size()) throw std::invalid_argument("element-wise comparison of arrays of different size");
    BoolMask m(a.size());
    for (size_t i = 0; i < a.size(); ++i) m.set(i, f(flat_at(a, i), flat_at(b, i)));
    return m;
}
template <typename A, typename S, typename F>
BoolMask compare_scalar(const A& a, S s, F f) {
    BoolMask m(a.size());
    for (size_t i = 0; i < a.size(); ++i) m.set(i, f(flat_at(a, i), s));
    return m;
}
}  // namespace impl

#define OUSTER_ARRAY_CMP_(op)                                                                                          \
    template <typename A, typename B,                                                                                  \
              typename = typename std::enable_if<impl::is_array_like<A>::value && impl::is_array_like<B>::value &&     \
                                                 !std::is_same<A, B>::value>::type>                                    \
    BoolMask operator op(const A& a, const B& b) {                                                                     \
        return impl::compare_arrays(a, b, [](const auto& x, const auto& y) { return x op y; });                        \
    }                                                                                                                  \
    template <typename T>                                                                                              \
    BoolMask operator op(const ImgRef<T>& a, const ImgRef<T>& b) {                                                     \
        return impl::compare_arrays(a, b, [](const auto& x, const auto& y) { return x op y; });                        \
    }                                                                                                                  \
    template <typename T>                                                                                              \
    BoolMask operator op(const VecRef<T>& a, const VecRef<T>& b) {                                                     \
        return impl::compare_arrays(a, b, [](const auto& x, const auto& y) { return x op y; });                        \
    }                                                                                                                  \
    template <typename A, typename S,                                                                                  \
              typename = typename std::enable_if<impl::is_array_like<A>::value && std::is_arithmetic<S>::value>::type, \
              typename = void>                                                                                         \
    BoolMask operator op(const A& a, S s) {                                                                            \
        return impl::compare_scalar(a, s, [](const auto& x, const auto& y) { return x op y; });                        \
    }
OUSTER_ARRAY_CMP_(==)
OUSTER_ARRAY_CMP_(!=)
#undef OUSTER_ARRAY_CMP_

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
