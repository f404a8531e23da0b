// host_internal.h -- declarations shared by the host sources (not installed).
#pragma once

#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "ouster/core/chanfield.h"
#include "ouster/core/data_format.h"
#include "ouster/hip/context.h"
#include "ouster_hip.h"

namespace ouster {
namespace sdk {
namespace core {
namespace impl {
/** default LidarFrame planes of a (built-in or custom) profile */
std::vector<std::pair<std::string, ChanFieldType>> default_planes(UDPProfileLidar profile);

/** One plane of a released frame whose destaggered form is still in HBM (include/ouster/core/lidar_frame.h, impl::mirrors_live):
 *  written by the FrameBatcher's release launch as a by-product (the fused decode kernel has the pixels in LDS anyway),
 *  served by destagger() as one copy out.  Keyed by the plane's host storage. */
struct DeviceLut;
/** What XYZLut()(frame) was last asked for on a frame of one FrameBatcher: its next release launch projects the range planes
 *  with that LUT ahead of the call (the fused kernel has the ranges in registers; the clouds are HBM stores), and the call
 *  itself becomes one copy out (6.3 MB at the link rate: 0.125 ms instead of 0.147 for a kernel that also reads the range over
 *  the link).  Shared between the batcher and the mirror entries of the frames it released. */
struct XyzWish {
    std::mutex mu;
    std::shared_ptr<const DeviceLut> lut;   ///< keeps the device tables alive
    bool f64 = true;
};
struct MirrorPlane {
    const void* host = nullptr;
    const void* d_destaggered = nullptr;
    size_t h = 0, w = 0, elem = 0;
    int device = 0;
    std::shared_ptr<const std::vector<int>> shifts;   ///< the pixel_shift_by_row the destaggered form was made with
    std::shared_ptr<void> keep;                       ///< the device block
    // a 32-bit range plane may also have its cloud in HBM: lut(plane) with `xyz_lut`, element type per xyz_f64
    const void* d_xyz = nullptr;
    std::shared_ptr<const DeviceLut> xyz_lut;         ///< the LUT it was projected with; owned, so that its address cannot be
                                                      ///< handed to another LUT while this entry lives
    bool xyz_f64 = true;
    std::shared_ptr<XyzWish> wish;                    ///< where a call that found no cloud leaves its LUT for the next release
};
void mirror_register(const MirrorPlane& m);
bool mirror_find(const void* host_storage, MirrorPlane& out);
}  // namespace impl
}  // namespace core

namespace hip {
/** Throws std::invalid_argument / std::runtime_error for a failed C ABI call. */
void check(int rc);
/** C handle of Context::current(): the innermost ScopedContext of the calling thread, else the
 *  thread's default context on its current device (include/ouster/hip/context.h).  Also makes that
 *  device the HIP-current one. */
ouster_hip_ctx* default_ctx();
}  // namespace hip
}  // namespace sdk
}  // namespace ouster
