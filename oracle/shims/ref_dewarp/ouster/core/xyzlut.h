// TEST INFRASTRUCTURE (oracle/): XYZLutT<T> stand-in for compiling the reference's impl/dewarp_impl.h (see frame_set.h
// next to this file).  operator() is impl::cartesianT<T> (ouster_core/include/ouster/core/impl/cartesian.h:36-66):
// xyz = range * direction + offset in T, (0, 0, 0) for a zero range; separate multiply and add (-ffp-contract=off).
#pragma once
#include "ouster/core/frame_set.h"

namespace ouster {
namespace sdk {
namespace core {

template <typename T>
struct XYZLutT {
    const T* direction = nullptr;   // [h*w][3]
    const T* offset = nullptr;
    PointCloudXYZ<T> operator()(const img_view<uint32_t>& range) const {
        const size_t n = static_cast<size_t>(range.rows() * range.cols());
        PointCloudXYZ<T> pts(n);
        T* o = pts.data();
        for (size_t i = 0; i < n; ++i) {
            const uint32_t r = range.p[i];
            for (int k = 0; k < 3; ++k)
                o[3 * i + k] = r ? static_cast<T>(r) * direction[3 * i + k] + offset[3 * i + k] : T(0);
        }
        return pts;
    }
};

}  // namespace core
}  // namespace sdk
}  // namespace ouster
