// bench_placement.cpp -- the placement setup sequence of DESIGN 3.2 through the C++ API (no Python): a DeviceFrameBatch of 256
// dual-return 128 x 2048 frames, full output set; milliseconds per decode() on the buffers as allocated, after
// refine_placement(3), after tune_placement(10, 4 GB), after refine_placement(3) again -- and what BatchOptions::placement_thorough
// gives at construction.  Usage: bench_placement [frames=256]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "ouster/core/lidar_scan.h"
#include "ouster/hip/device_batch.h"

using namespace ouster::sdk::core;
using clk = std::chrono::steady_clock;

static double ms_per_decode(ouster::sdk::hip::DeviceFrameBatch& b, int n = 20) {
    for (int i = 0; i < 30; ++i) b.decode();
    b.sync();
    const auto t0 = clk::now();
    for (int i = 0; i < n; ++i) b.decode();
    b.sync();
    return std::chrono::duration<double, std::milli>(clk::now() - t0).count() / n;
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 256;
    SensorInfo info;
    info.format.pixels_per_column = 128;
    info.format.columns_per_frame = 2048;
    info.format.columns_per_packet = 16;
    info.format.column_window = {0, 2047};
    info.format.udp_profile_lidar = UDPProfileLidar::RNG15_RFL8_NIR8_DUAL;
    for (int i = 0; i < 128; ++i) {
        info.format.pixel_shift_by_row.push_back((int[]){24, 8, -8, -24}[i % 4]);
        info.beam_azimuth_angles.push_back((double[]){4.2, 1.4, -1.4, -4.2}[i % 4]);
        info.beam_altitude_angles.push_back(21.0 - 42.0 * i / 127.0);
    }
    info.prod_line = "OS-2-128";
    info.beam_to_lidar_transform = default_beam_to_lidar_transform(info.prod_line);
    info.lidar_to_sensor_transform = DEFAULT_LIDAR_TO_SENSOR;
    info.sensor_to_body = mat4d::Identity();
    info.fw_rev = "v3.2.0";
    auto pf = std::make_shared<PacketFormat>(info);
    std::mt19937 g(3);
    std::vector<std::vector<LidarPacket>> pool;
    for (int f = 0; f < 4; ++f) {
        LidarFrame fr(info);
        for (auto it = pf->begin(); it != pf->end(); ++it) {
            if (!fr.has_field(it->first)) continue;
            Field& fld = fr.field(it->first);
            const uint64_t mask = pf->field_value_mask(it->first);
            uint8_t* p = static_cast<uint8_t*>(fld.get());
            for (size_t i = 0; i < fld.size(); ++i) {
                uint64_t v = g() & mask;
                std::memcpy(p + i * fld.element_size(), &v, fld.element_size());
            }
        }
        for (size_t i = 0; i < fr.w; ++i) { fr.timestamp()[i] = 1000 + i; fr.measurement_id()[i] = i; fr.status()[i] = 1; }
        fr.frame_id = 700 + f;
        pool.push_back(impl::frame_to_packets(fr, pf, 0, 0));
    }
    const double bytes = 14974976.0 * n;
    auto frac = [&](double ms) { return bytes / (ms * 1e-3) / 8e12; };
    auto fill = [&](ouster::sdk::hip::DeviceFrameBatch& b) {
        for (uint32_t f = 0; f < n; ++f) {
            std::vector<const uint8_t*> ptrs;
            for (auto& p : pool[f % pool.size()]) ptrs.push_back(p.buf.data());
            b.upload_frame_packets(f, ptrs);
        }
    };
    ouster::sdk::hip::BatchOptions opt;
    opt.destagger = {"RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"};
    opt.xyz = true;
    std::printf("{\"frames\": %u", n);
    {
        opt.auto_placement = false;
        ouster::sdk::hip::DeviceFrameBatch b({info}, n, opt);
        fill(b);
        const double m0 = ms_per_decode(b);
        auto t0 = clk::now();
        b.refine_placement(3, nullptr, 0);
        const double s1 = std::chrono::duration<double>(clk::now() - t0).count();
        fill(b);
        const double m1 = ms_per_decode(b);
        t0 = clk::now();
        std::vector<double> draws;
        b.tune_placement(10, &draws, size_t{4} << 30);
        const double s2 = std::chrono::duration<double>(clk::now() - t0).count();
        fill(b);
        const double m2 = ms_per_decode(b);
        t0 = clk::now();
        b.refine_placement(3, nullptr, 0);
        const double s3 = std::chrono::duration<double>(clk::now() - t0).count();
        fill(b);
        const double m3 = ms_per_decode(b);
        std::printf(", \"step_by_step\": {\"as_allocated_ms\": %.4f, \"refine_ms\": %.4f, \"tune10_ms\": %.4f, \"refine_again_ms\": %.4f, "
                    "\"frac_as_allocated\": %.4f, \"frac_final\": %.4f, \"seconds\": [%.2f, %.2f, %.2f], \"whole_set_draws_ms\": [",
                    m0, m1, m2, m3, frac(m0), frac(m3), s1, s2, s3);
        for (size_t i = 0; i < draws.size(); ++i) std::printf("%s%.4f", i ? ", " : "", draws[i]);
        std::printf("]}");
    }
    for (int thorough = 0; thorough < 2; ++thorough) {
        opt.auto_placement = true;
        opt.placement_thorough = thorough != 0;
        const auto t0 = clk::now();
        ouster::sdk::hip::DeviceFrameBatch b({info}, n, opt);
        const double s = std::chrono::duration<double>(clk::now() - t0).count();
        fill(b);
        const double m = ms_per_decode(b);
        std::printf(", \"%s\": {\"construction_s\": %.2f, \"ms\": %.4f, \"frac\": %.4f}", thorough ? "placement_thorough" : "default_construction", s, m, frac(m));
    }
    std::printf("}\n");
    return 0;
}
