// bench_host_api.cpp -- latency of the drop-in, frame-at-a-time host API (host LidarFrame in,
// host results out; every call crosses PCIe): FrameBatcher::batch x128 -> destagger<uint32_t>
// -> XYZLut::operator().  This is what an unmodified reference caller gets; the batched,
// device-resident path (bench.py, DeviceFrameBatch) is two orders of magnitude faster.
// Build: make -C tests/cpp ../../tools/bench_host_api   (or see tools/Makefile rule below)
#include <chrono>
#include <cstdio>
#include <random>

#include "ouster/core/lidar_scan.h"

using namespace ouster::sdk::core;
using clk = std::chrono::steady_clock;

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 20;
    SensorInfo info;
    info.format.pixels_per_column = 128;
    info.format.columns_per_packet = 16;
    info.format.columns_per_frame = 2048;
    info.format.column_window = {0, 2047};
    info.format.udp_profile_lidar = UDPProfileLidar::RNG15_RFL8_NIR8_DUAL;
    for (int i = 0; i < 128; ++i) {
        static const int pat[4] = {24, 8, -8, -24};
        info.format.pixel_shift_by_row.push_back(pat[i % 4]);
        info.beam_altitude_angles.push_back(21.0 - 42.0 * i / 127);
        info.beam_azimuth_angles.push_back(4.2 - 2.8 * (i % 4));
    }
    info.beam_to_lidar_transform = default_beam_to_lidar_transform("OS-2-128");
    info.lidar_to_sensor_transform = DEFAULT_LIDAR_TO_SENSOR;
    info.fw_rev = "v3.2.0";
    info.init_id = 77;
    auto sinfo = std::make_shared<SensorInfo>(info);
    auto pf = std::make_shared<PacketFormat>(info);
    LidarFrame src(sinfo);
    std::mt19937 g(1);
    for (auto it = pf->begin(); it != pf->end(); ++it) {
        if (!src.has_field(it->first)) continue;
        Field& f = src.field(it->first);
        const uint64_t mask = pf->field_value_mask(it->first);
        uint8_t* p = static_cast<uint8_t*>(f.get());
        for (size_t i = 0; i < f.size(); ++i) {
            uint64_t v = g() & mask;
            std::memcpy(p + i * f.element_size(), &v, f.element_size());
        }
    }
    for (size_t i = 0; i < src.w; ++i) { src.status()[i] = 1; src.measurement_id()[i] = i; src.timestamp()[i] = i; }
    for (size_t i = 0; i < src.packet_count(); ++i) src.packet_timestamp()[i] = 1 + i;
    XYZLut lut(info, false);
    LidarFrame frame(sinfo);
    FrameBatcher batcher(sinfo);
    double t_batch = 0, t_dst = 0, t_xyz = 0;
    for (int f = -2; f < frames; ++f) {  // two warm-up frames
        src.frame_id = 100 + f + 2;
        auto packets = impl::frame_to_packets(src, pf, info.init_id, 1);
        auto t0 = clk::now();
        bool done = false;
        for (auto& p : packets) done = batcher(p, frame);
        auto t1 = clk::now();
        auto d = destagger<uint32_t>(info, frame.field<uint32_t>("RANGE"));
        auto t2 = clk::now();
        auto pts = lut(frame);
        auto t3 = clk::now();
        if (!done || pts.rows() != 128 * 2048 || d.rows() != 128) return 1;
        if (f >= 0) {
            t_batch += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t_dst += std::chrono::duration<double, std::milli>(t2 - t1).count();
            t_xyz += std::chrono::duration<double, std::milli>(t3 - t2).count();
        }
    }
    std::printf("{\"frames\": %d, \"ms_per_frame\": {\"FrameBatcher_128_packets\": %.3f, \"destagger_u32\": %.3f, "
                "\"XYZLut_f64\": %.3f}, \"note\": \"host containers in/out, one PCIe round trip per call\"}\n",
                frames, t_batch / frames, t_dst / frames, t_xyz / frames);
    return 0;
}
