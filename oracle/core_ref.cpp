// TEST INFRASTRUCTURE (oracle/_ref): C entry points around the REFERENCE's own CPU loops of the hot path,
//   cartesianT<T>      /root/reference/ouster_core/include/ouster/core/impl/cartesian.h:36-66      (the header, as it lies)
//   destagger_into<T>  /root/reference/ouster_core/include/ouster/core/impl/lidar_frame_impl.h:733-760 (the one function,
//                      staged at build time by oracle/stage_slice.py: the rest of that header needs Eigen::Tensor)
// compiled against oracle/shims/ref_core (the few Eigen / ouster names they touch).  tests/test_oracle_ref_core.py pins the
// oracle's restatements (ora_cartesian_*, ora_destagger) on these bit for bit, and bench.py times them as the CPU baseline's
// reference legs.  Never used by the product.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ouster/core/impl/cartesian.h"

namespace ouster {
namespace sdk {
namespace core {
#include "destagger_into_staged.inc"
}  // namespace core
}  // namespace sdk
}  // namespace ouster

using namespace ouster::sdk::core;

template <typename T>
static void cartesian(T* pts, const uint32_t* range, const T* dir, const T* ofs, size_t h, size_t w) {
    PointCloudXYZ<T> points(pts, static_cast<long>(h * w));
    img_t<uint32_t> rng(const_cast<uint32_t*>(range), static_cast<long>(h), static_cast<long>(w));
    const ArrayX3R<T> direction(const_cast<T*>(dir), static_cast<long>(h * w)), offset(const_cast<T*>(ofs), static_cast<long>(h * w));
    impl::cartesianT<T>(Eigen::Ref<PointCloudXYZ<T>>(points), Eigen::Ref<const img_t<uint32_t>>(rng), direction, offset);
}

template <typename T>
static int destagger_t(const void* img, void* out, size_t h, size_t w, const int* shifts, size_t n_shifts, int inverse) {
    img_t<T> in(static_cast<T*>(const_cast<void*>(img)), static_cast<long>(h), static_cast<long>(w));
    img_t<T> dst(static_cast<T*>(out), static_cast<long>(h), static_cast<long>(w));
    const std::vector<int> sh(shifts, shifts + n_shifts);
    try {
        destagger_into<T>(Eigen::Ref<const img_t<T>>(in), sh, inverse != 0, Eigen::Ref<img_t<T>>(dst));
    } catch (const std::invalid_argument&) {
        return -1;
    }
    return 0;
}

extern "C" {
void ref_cartesian_f64(double* pts, const uint32_t* range, const double* dir, const double* ofs, size_t h, size_t w) {
    cartesian<double>(pts, range, dir, ofs, h, w);
}
void ref_cartesian_f32(float* pts, const uint32_t* range, const float* dir, const float* ofs, size_t h, size_t w) {
    cartesian<float>(pts, range, dir, ofs, h, w);
}
// elem: bytes per pixel (1, 2, 4, 8); returns -1 where the reference throws (shifts size != image height)
int ref_destagger(const void* img, void* out, size_t h, size_t w, size_t elem, const int* shifts, size_t n_shifts, int inverse) {
    switch (elem) {
        case 1: return destagger_t<uint8_t>(img, out, h, w, shifts, n_shifts, inverse);
        case 2: return destagger_t<uint16_t>(img, out, h, w, shifts, n_shifts, inverse);
        case 4: return destagger_t<uint32_t>(img, out, h, w, shifts, n_shifts, inverse);
        case 8: return destagger_t<uint64_t>(img, out, h, w, shifts, n_shifts, inverse);
        default: return -2;
    }
}
// The destagger + cartesian half of one frame of the benchmark workload, `reps` times on one core: n_dst planes of
// dst_elem[i] bytes are destaggered, n_xyz range planes are projected (cartesianT<double>).  Returns seconds.
double ref_bench_frame_legs(const void* const* dst_planes, const size_t* dst_elem, size_t n_dst, const uint32_t* const* ranges,
                            size_t n_xyz, const double* dir, const double* ofs, size_t h, size_t w, const int* shifts, int reps,
                            double* seconds_destagger, double* seconds_cartesian) {
    std::vector<uint8_t> scratch(h * w * 8);
    std::vector<double> pts(h * w * 3);
    double td = 0, tc = 0;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n_dst; ++i) ref_destagger(dst_planes[i], scratch.data(), h, w, dst_elem[i], shifts, h, 0);
        auto t1 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n_xyz; ++i) cartesian<double>(pts.data(), ranges[i], dir, ofs, h, w);
        auto t2 = std::chrono::steady_clock::now();
        td += std::chrono::duration<double>(t1 - t0).count();
        tc += std::chrono::duration<double>(t2 - t1).count();
    }
    if (seconds_destagger) *seconds_destagger = td;
    if (seconds_cartesian) *seconds_cartesian = tc;
    return td + tc;
}
}
