// array_view.h -- non-owning n-d strided views of Field memory.
//
// Mirrors the part of ouster_core/include/ouster/core/array_view.h:185-420 that callers of the
// hot path use: ArrayView<T, Dim> with public `shape` / `strides` arrays (in ELEMENTS, like the
// reference), element access view(i, j, ...), `data()`, `sparse()`, leading-index `subview(i)` and
// the ArrayViewN / ConstArrayViewN aliases.  Field converts to it through the FieldView-style
// operators in lidar_frame.h (field.h:374-470), which throw std::invalid_argument on an element
// type or rank mismatch.  The reference's range()/keep() sub-slicing and reshape are not mirrored.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <type_traits>

namespace ouster {
namespace sdk {
namespace core {

template <typename T, size_t Dim>
class ArrayView {
    static_assert(Dim > 0, "ArrayView needs at least one dimension");

   public:
    int32_t strides[Dim];  ///< elements to skip per unit step of each dimension
    uint32_t shape[Dim];

    /** Dense row-major view of `shape`. */
    template <typename ShapeT>
    ArrayView(T* ptr, const ShapeT& shape_in) : ptr_(ptr) {
        size_t i = 0;
        for (auto v : shape_in) {
            if (i >= Dim) throw std::invalid_argument("ArrayView: too many dimensions");
            shape[i++] = static_cast<uint32_t>(v);
        }
        if (i != Dim) throw std::invalid_argument("ArrayView: too few dimensions");
        int64_t s = 1;
        for (size_t d = Dim; d-- > 0;) {
            strides[d] = static_cast<int32_t>(s);
            s *= shape[d];
        }
    }
    /** Explicit layout. */
    template <typename ShapeT, typename StridesT>
    ArrayView(T* ptr, const ShapeT& shape_in, const StridesT& strides_in) : ptr_(ptr) {
        size_t i = 0;
        for (auto v : shape_in) {
            if (i >= Dim) throw std::invalid_argument("ArrayView: too many dimensions");
            shape[i++] = static_cast<uint32_t>(v);
        }
        if (i != Dim) throw std::invalid_argument("ArrayView: too few dimensions");
        i = 0;
        for (auto v : strides_in) {
            if (i >= Dim) break;
            strides[i++] = static_cast<int32_t>(v);
        }
        if (i != Dim) throw std::invalid_argument("ArrayView: too few strides");
    }
    /** const view of a mutable one */
    template <typename U, typename = typename std::enable_if<std::is_same<const U, T>::value>::type>
    ArrayView(const ArrayView<U, Dim>& o) : ptr_(o.data()) {
        for (size_t d = 0; d < Dim; ++d) {
            shape[d] = o.shape[d];
            strides[d] = o.strides[d];
        }
    }

    T* data() const { return ptr_; }
    size_t size() const {
        size_t n = 1;
        for (size_t d = 0; d < Dim; ++d) n *= shape[d];
        return n;
    }
    /** true when the elements are not one dense row-major block */
    bool sparse() const {
        int64_t s = 1;
        for (size_t d = Dim; d-- > 0;) {
            if (strides[d] != s) return true;
            s *= shape[d];
        }
        return false;
    }

    template <typename... Idx>
    T& operator()(Idx... idx) const {
        static_assert(sizeof...(Idx) == Dim, "ArrayView: one index per dimension");
        const int64_t ix[Dim] = {static_cast<int64_t>(idx)...};
        int64_t off = 0;
        for (size_t d = 0; d < Dim; ++d) off += ix[d] * strides[d];
        return ptr_[off];
    }

    /** view(i) of everything behind the leading index. @throw std::invalid_argument out of bounds */
    template <size_t D = Dim, typename = typename std::enable_if<(D > 1)>::type>
    ArrayView<T, Dim - 1> subview(size_t i) const {
        if (i >= shape[0]) throw std::invalid_argument("ArrayView invalid subview ranges");
        uint32_t sh[Dim - 1];
        int32_t st[Dim - 1];
        for (size_t d = 1; d < Dim; ++d) {
            sh[d - 1] = shape[d];
            st[d - 1] = strides[d];
        }
        return ArrayView<T, Dim - 1>(ptr_ + static_cast<int64_t>(i) * strides[0], sh, st);
    }

   private:
    T* ptr_;
};

template <typename T, size_t Dim>
using ConstArrayView = ArrayView<const T, Dim>;
template <typename T> using ArrayView4 = ArrayView<T, 4>;
template <typename T> using ArrayView3 = ArrayView<T, 3>;
template <typename T> using ArrayView2 = ArrayView<T, 2>;
template <typename T> using ArrayView1 = ArrayView<T, 1>;
template <typename T> using ConstArrayView4 = ConstArrayView<T, 4>;
template <typename T> using ConstArrayView3 = ConstArrayView<T, 3>;
template <typename T> using ConstArrayView2 = ConstArrayView<T, 2>;
template <typename T> using ConstArrayView1 = ConstArrayView<T, 1>;

}  // namespace core
}  // namespace sdk
}  // namespace ouster
