// pose_util.h -- dense dewarp: apply each column's pose to the points of that column.
// Same signatures as ouster_core/include/ouster/core/pose_util.h:38-103 (dewarp<T>(points,
// poses)) and pose_util.h:456-493 + impl/dewarp_impl.h:23-115 (range-gated, compacting
// dewarp(LidarFrame | FrameSet, XYZLut, min_range, max_range) with optional provenance); the
// per-point work runs on the GPU (ouster_hip_dewarp / ouster_hip_dewarp_frames).
#pragma once

#include <array>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ouster/core/lidar_frame.h"
#include "ouster/core/typedefs.h"
#include "ouster/core/xyzlut.h"

namespace ouster {
namespace sdk {
namespace core {

/** W x 16: one flattened row-major 4x4 pose per column (MatrixX16dR in the reference). */
using Poses = ArrayXXR<double>;

/** Stand-in for Eigen::Vector3<T> (three contiguous T, like the reference's element type). */
template <typename T>
using Vector3 = std::array<T, 3>;

/** The part of FrameSet the dewarp needs: frames by index, null = invalid index
 *  (FrameSet::valid_indices, frame_set.h). */
using FrameSet = std::vector<std::shared_ptr<LidarFrame>>;

namespace impl {
void dewarp_device(const void* points, const double* poses, void* out, bool f64, size_t h, size_t w);

/** Batched GPU implementation behind the frame dewarps: frames[i] uses luts[i]; results are
 *  appended to the output vectors (pointers may be null).  Returns the points, x/y/z of T. */
void dewarp_frames_device(const std::vector<const LidarFrame*>& frames,
                          const std::vector<const DeviceLut*>& luts,
                          const std::vector<uint32_t>& frame_index, double min_range,
                          double max_range, bool f64, std::vector<unsigned char>& points,
                          std::vector<uint32_t>* frame_idxs, std::vector<uint32_t>* col_idxs,
                          std::vector<uint64_t>* timestamps_ns);

/** impl/dewarp_impl.h:23-81 */
template <typename T>
std::vector<Vector3<T>> dewarp_impl(const LidarFrame& lidar_frame, const XYZLutT<T>& xyzlut,
                                    double min_range, double max_range,
                                    std::vector<uint32_t>* col_idxs,
                                    std::vector<uint64_t>* timestamps_ns) {
    std::vector<unsigned char> raw;
    dewarp_frames_device({&lidar_frame}, {&xyzlut.device()}, {0}, min_range, max_range,
                         sizeof(T) == 8, raw, nullptr, col_idxs, timestamps_ns);
    std::vector<Vector3<T>> out(raw.size() / sizeof(Vector3<T>));
    if (!raw.empty()) std::memcpy(out.data(), raw.data(), raw.size());
    return out;
}

/** impl/dewarp_impl.h:86-115 */
template <typename T>
std::vector<Vector3<T>> dewarp_impl(const FrameSet& frame_set, const std::vector<XYZLutT<T>>& xyzluts,
                                    double min_range, double max_range,
                                    std::vector<uint32_t>* frame_idxs,
                                    std::vector<uint32_t>* col_idxs,
                                    std::vector<uint64_t>* timestamps_ns) {
    if (frame_set.size() != xyzluts.size())
        throw std::invalid_argument("Number of frames and number of XYZLuts must be the same");
    std::vector<const LidarFrame*> frames;
    std::vector<const DeviceLut*> luts;
    std::vector<uint32_t> index;
    for (size_t i = 0; i < frame_set.size(); ++i) {
        if (!frame_set[i]) continue;
        frames.push_back(frame_set[i].get());
        luts.push_back(&xyzluts[i].device());
        index.push_back(static_cast<uint32_t>(i));
    }
    std::vector<unsigned char> raw;
    dewarp_frames_device(frames, luts, index, min_range, max_range, sizeof(T) == 8, raw, frame_idxs,
                         col_idxs, timestamps_ns);
    std::vector<Vector3<T>> out(raw.size() / sizeof(Vector3<T>));
    if (!raw.empty()) std::memcpy(out.data(), raw.data(), raw.size());
    return out;
}
}  // namespace impl

/** @throw std::invalid_argument on shape mismatches. */
template <typename T>
void dewarp(ImgRef<T> dewarped, const ImgRef<const T>& points, const Poses& poses) {
    const size_t W = poses.rows();
    if (poses.cols() != 16 || W == 0 || points.cols() != 3 || dewarped.cols() != 3 ||
        points.rows() != dewarped.rows() || points.rows() % W != 0)
        throw std::invalid_argument("dewarp: unexpected dimensions");
    impl::dewarp_device(points.data(), poses.data(), dewarped.data(), sizeof(T) == 8,
                        points.rows() / W, W);
}

template <typename T>
PointCloudXYZ<T> dewarp(const PointCloudXYZ<T>& points, const Poses& poses) {
    PointCloudXYZ<T> out(points.rows());
    dewarp<T>(ImgRef<T>(out), ImgRef<const T>(points), poses);
    return out;
}

/** Range-gated dewarp of one frame with its per-column body_to_world poses
 *  (pose_util.h:456-485): points of valid columns with min_range <= r <= max_range [m]. */
template <typename T>
std::vector<Vector3<T>> dewarp(const LidarFrame& lidar_frame, const XYZLutT<T>& xyzlut,
                               double min_range, double max_range) {
    return impl::dewarp_impl<T>(lidar_frame, xyzlut, min_range, max_range, nullptr, nullptr);
}

/** FrameSet form (pose_util.h:475-493): frames concatenated in index order. */
template <typename T>
std::vector<Vector3<T>> dewarp(const FrameSet& frame_set, const std::vector<XYZLutT<T>>& xyzluts,
                               double min_range, double max_range) {
    return impl::dewarp_impl<T>(frame_set, xyzluts, min_range, max_range, nullptr, nullptr, nullptr);
}

}  // namespace core
}  // namespace sdk
}  // namespace ouster
