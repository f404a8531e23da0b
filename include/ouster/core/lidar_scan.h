// lidar_scan.h -- pre-1.0 spellings kept as aliases so callers written against
// `ouster::LidarScan`, `ouster::sensor::packet_format`, `ouster::destagger<T>()`,
// `ouster::cartesian()` and `ouster::make_xyz_lut()` keep compiling.  The reference keeps
// LidarScan / ScanBatcher / LidarScanFieldTypes as deprecated typedefs
// (ouster_core/include/ouster/core/lidar_frame.h:1157-1160; lidar_scan.h is a forwarding
// header); the ouster:: / ouster::sensor:: aliases restore the 0.x namespaces named in the
// project brief.
#pragma once

#include "ouster/core/lidar_frame.h"
#include "ouster/core/pose_util.h"
#include "ouster/core/profile_extension.h"
#include "ouster/core/xyzlut.h"

namespace ouster {
namespace sdk {
namespace core {
using LidarScan = LidarFrame;
using LidarScanFieldTypes = LidarFrameFieldTypes;
using ScanBatcher = FrameBatcher;
}  // namespace core
}  // namespace sdk

using LidarScan = sdk::core::LidarFrame;
using LidarScanFieldTypes = sdk::core::LidarFrameFieldTypes;
using ScanBatcher = sdk::core::FrameBatcher;
using XYZLut = sdk::core::XYZLut;
template <typename T>
using img_t = sdk::core::img_t<T>;
using sdk::core::cartesian;
using sdk::core::destagger;
using sdk::core::stagger;
using sdk::core::impl::make_xyz_lut;

namespace sensor {
using packet_format = sdk::core::PacketFormat;
using sensor_info = sdk::core::SensorInfo;
using data_format = sdk::core::DataFormat;
using sdk::core::get_format;
using sdk::core::ChanFieldType;
using sdk::core::UDPProfileLidar;
namespace ChanField = sdk::core::ChanField;
constexpr double range_unit = sdk::core::RANGE_UNIT;
}  // namespace sensor
}  // namespace ouster
