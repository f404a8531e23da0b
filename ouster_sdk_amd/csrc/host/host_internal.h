// host_internal.h -- declarations shared by the host sources (not installed).
#pragma once

#include <memory>
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
