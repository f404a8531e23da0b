// TEST INFRASTRUCTURE (oracle/_ref): C entry points around the REFERENCE's own frame dewarp,
//   /root/reference/ouster_core/include/ouster/core/impl/dewarp_impl.h:23-115
// compiled from where it lies (oracle/Makefile; -Ishims/ref_dewarp supplies the few ouster / Eigen types it touches).
// tests/test_oracle_ref_dewarp.py checks the oracle's restatement (ora_dewarp_frame_*) against these: counts, order,
// column / frame indices and timestamps exactly, points to the last bits.  Never used by the product.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "ouster/core/impl/dewarp_impl.h"

namespace ouster {
namespace sdk {
namespace core {
namespace impl {
// declared by dewarp_impl.h, defined in ouster_core/src/pose_util.cpp in the reference: any upper bound will do for the
// reserve() calls it feeds
size_t max_number_of_valid_points(const FrameSet& frame_set) {
    size_t n = 0;
    for (size_t i : frame_set.valid_indices()) n += frame_set[i]->w * frame_set[i]->h;
    return n;
}
}  // namespace impl
}  // namespace core
}  // namespace sdk
}  // namespace ouster

using namespace ouster::sdk::core;

template <typename T>
static size_t run_frames(T* out, uint32_t* frame_idx, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                         const uint32_t* status, const uint64_t* timestamp, const double* poses, const T* dir,
                         const T* ofs, const uint8_t* present, size_t n_frames, size_t h, size_t w, double min_range,
                         double max_range) {
    FrameSet fs;
    std::vector<XYZLutT<T>> luts(n_frames);
    for (size_t f = 0; f < n_frames; ++f) {
        luts[f].direction = dir;
        luts[f].offset = ofs;
        if (present && !present[f]) {
            fs.frames.push_back(nullptr);
            continue;
        }
        auto fr = std::make_shared<LidarFrame>();
        fr->w = w;
        fr->h = h;
        fr->range_ = range + f * h * w;
        fr->status_ = status + f * w;
        fr->timestamp_ = timestamp + f * w;
        fr->poses_ = poses + f * w * 16;
        fs.frames.push_back(fr);
    }
    std::vector<uint32_t> fi, ci;
    std::vector<uint64_t> tn;
    std::vector<Eigen::Vector3<T>> pts =
        impl::dewarp_impl<T>(fs, luts, min_range, max_range, frame_idx ? &fi : nullptr, col_idx ? &ci : nullptr, ts ? &tn : nullptr);
    for (size_t i = 0; i < pts.size(); ++i)
        for (int k = 0; k < 3; ++k) out[3 * i + k] = pts[i][k];
    if (frame_idx) std::memcpy(frame_idx, fi.data(), fi.size() * 4);
    if (col_idx) std::memcpy(col_idx, ci.data(), ci.size() * 4);
    if (ts) std::memcpy(ts, tn.data(), tn.size() * 8);
    return pts.size();
}

extern "C" {
// one frame: dewarp_impl<T>(const LidarFrame&, ...)
size_t ref_dewarp_frame_f64(double* out, uint32_t* col_idx, uint64_t* ts, const uint32_t* range, const uint32_t* status,
                            const uint64_t* timestamp, const double* poses, const double* dir, const double* ofs, size_t h,
                            size_t w, double min_range, double max_range) {
    LidarFrame fr;
    fr.w = w; fr.h = h; fr.range_ = range; fr.status_ = status; fr.timestamp_ = timestamp; fr.poses_ = poses;
    XYZLutT<double> lut{dir, ofs};
    std::vector<uint32_t> ci;
    std::vector<uint64_t> tn;
    auto pts = impl::dewarp_impl<double>(fr, lut, min_range, max_range, col_idx ? &ci : nullptr, ts ? &tn : nullptr);
    for (size_t i = 0; i < pts.size(); ++i)
        for (int k = 0; k < 3; ++k) out[3 * i + k] = pts[i][k];
    if (col_idx) std::memcpy(col_idx, ci.data(), ci.size() * 4);
    if (ts) std::memcpy(ts, tn.data(), tn.size() * 8);
    return pts.size();
}
size_t ref_dewarp_frame_f32(float* out, uint32_t* col_idx, uint64_t* ts, const uint32_t* range, const uint32_t* status,
                            const uint64_t* timestamp, const double* poses, const float* dir, const float* ofs, size_t h,
                            size_t w, double min_range, double max_range) {
    LidarFrame fr;
    fr.w = w; fr.h = h; fr.range_ = range; fr.status_ = status; fr.timestamp_ = timestamp; fr.poses_ = poses;
    XYZLutT<float> lut{dir, ofs};
    std::vector<uint32_t> ci;
    std::vector<uint64_t> tn;
    auto pts = impl::dewarp_impl<float>(fr, lut, min_range, max_range, col_idx ? &ci : nullptr, ts ? &tn : nullptr);
    for (size_t i = 0; i < pts.size(); ++i)
        for (int k = 0; k < 3; ++k) out[3 * i + k] = pts[i][k];
    if (col_idx) std::memcpy(col_idx, ci.data(), ci.size() * 4);
    if (ts) std::memcpy(ts, tn.data(), tn.size() * 8);
    return pts.size();
}
// a FrameSet: dewarp_impl<T>(const FrameSet&, ...) -- present[f] == 0 makes frame f an empty slot of the set
size_t ref_dewarp_frames_f64(double* out, uint32_t* frame_idx, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                             const uint32_t* status, const uint64_t* timestamp, const double* poses, const double* dir,
                             const double* ofs, const uint8_t* present, size_t n_frames, size_t h, size_t w,
                             double min_range, double max_range) {
    return run_frames<double>(out, frame_idx, col_idx, ts, range, status, timestamp, poses, dir, ofs, present, n_frames, h,
                              w, min_range, max_range);
}
size_t ref_dewarp_frames_f32(float* out, uint32_t* frame_idx, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                             const uint32_t* status, const uint64_t* timestamp, const double* poses, const float* dir,
                             const float* ofs, const uint8_t* present, size_t n_frames, size_t h, size_t w,
                             double min_range, double max_range) {
    return run_frames<float>(out, frame_idx, col_idx, ts, range, status, timestamp, poses, dir, ofs, present, n_frames, h, w,
                             min_range, max_range);
}
}
