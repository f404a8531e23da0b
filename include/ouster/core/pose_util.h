// pose_util.h -- dense dewarp: apply each column's pose to the points of that column.
// Same signatures as ouster_core/include/ouster/core/pose_util.h:38-103 (dewarp<T>(points,
// poses)); the per-point work runs on the GPU (ouster_hip_dewarp).  The range-gated, compacting
// dewarp(LidarFrame, XYZLut, min_range, max_range) (impl/dewarp_impl.h:23-81) is not built yet.
#pragma once

#include <stdexcept>

#include "ouster/core/typedefs.h"

namespace ouster {
namespace sdk {
namespace core {

/** W x 16: one flattened row-major 4x4 pose per column (MatrixX16dR in the reference). */
using Poses = ArrayXXR<double>;

namespace impl {
void dewarp_device(const void* points, const double* poses, void* out, bool f64, size_t h, size_t w);
}

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

}  // namespace core
}  // namespace sdk
}  // namespace ouster
