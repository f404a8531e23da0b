// The shape of the reference's examples/representations_example.cpp (fields of a scan as typed
// images, destagger, cartesian) against the mirror -- compiled twice by tests/cpp/Makefile: plain
// (stand-in containers) and with -DOUSTER_HIP_USE_EIGEN (Eigen types at the boundary; Eigen itself is
// mocked by tests/cpp/mock_eigen in this image).  Runs on the GPU box; exit code 0 = all equal.
#include <cstdio>
#include <cstring>

#include "ouster/core/lidar_scan.h"

using namespace ouster::sdk::core;

int main() {
    SensorInfo info;
    info.format.pixels_per_column = 16;
    info.format.columns_per_packet = 16;
    info.format.columns_per_frame = 64;
    info.format.udp_profile_lidar = UDPProfileLidar::RNG15_RFL8_NIR8;
    for (int i = 0; i < 16; ++i) {
        info.format.pixel_shift_by_row.push_back((i % 4) * 3 - 4);
        info.beam_altitude_angles.push_back(10.0 - i);
        info.beam_azimuth_angles.push_back(1.5 * (i % 4));
    }
    info.prod_line = "OS-1-16";
    info.beam_to_lidar_transform = default_beam_to_lidar_transform(info.prod_line);
    info.lidar_to_sensor_transform = DEFAULT_LIDAR_TO_SENSOR;
    info.sensor_to_body = mat4d::Identity();

    LidarFrame scan(info);
    uint32_t* raw = scan.field(ChanField::RANGE);             // FieldView: operator T*
    for (size_t i = 0; i < scan.h * scan.w; ++i) raw[i] = static_cast<uint32_t>(1000 + 7 * i);
    int bad = 0;

    ArrayView2<uint32_t> view = scan.field(ChanField::RANGE);  // FieldView: operator ArrayView<T, 2>
    bad += view(3, 5) != raw[3 * scan.w + 5] || view.shape[1] != scan.w || view.sparse();
    try {
        ArrayView2<uint16_t> wrong = scan.field(ChanField::RANGE);
        (void)wrong;
        ++bad;
    } catch (const std::invalid_argument&) {
    }

#ifdef OUSTER_HIP_USE_EIGEN
    // exactly the reference's spelling: Eigen::Ref<img_t<T>> of a field, Eigen arrays in and out
    Eigen::Ref<EigenImg<uint32_t>> range = scan.field(ChanField::RANGE);
    EigenImg<uint32_t> destaggered = destagger<uint32_t>(info, ImgRef<const uint32_t>(range));
    XYZLut lut = impl::make_xyz_lut(info, false);
    EigenX3R<double> cloud = lut(ImgRef<const uint32_t>(range));
    bad += destaggered.rows() != 16 || destaggered.cols() != 64 || cloud.rows() != 16 * 64 || cloud.cols() != 3;
    img_t<uint32_t> back = destagger<uint32_t>(info, ImgRef<const uint32_t>(destaggered), true);
    bad += std::memcmp(back.data(), raw, 16 * 64 * 4) != 0;
    std::printf("eigen switch ON: %d mismatches\n", bad);
#else
    ImgRef<uint32_t> range = scan.field(ChanField::RANGE);
    img_t<uint32_t> destaggered = destagger<uint32_t>(info, range);
    XYZLut lut = impl::make_xyz_lut(info, false);
    PointCloudXYZd cloud = lut(range);
    bad += destaggered.rows() != 16 || destaggered.cols() != 64 || cloud.rows() != 16 * 64;
    img_t<uint32_t> back = destagger<uint32_t>(info, destaggered, true);
    bad += std::memcmp(back.data(), raw, 16 * 64 * 4) != 0;
    std::printf("eigen switch OFF: %d mismatches\n", bad);
#endif
    return bad ? 1 : 0;
}
