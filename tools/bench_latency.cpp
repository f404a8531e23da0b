// bench_latency.cpp -- what ONE call of the fused path costs a C++ caller (no Python in the loop): hip::DeviceFrameBatch of 1,
// 4 (four sensors: one BASELINE configs[4] tick) and 16 dual-return 128 x 2048 frames, full output set (8 planes + 4
// destaggered + 2 x XYZ f32 + headers), packets resident in HBM.  Per batch size: microseconds per decode() issued back to back
// (a streaming caller), per decode() + sync() (what one tick waits for, median), and the host time of the decode() call
// itself.  bench.py's `latency` rows time the same calls from Python (ctypes); this is the number without the interpreter.
// Usage: bench_latency [calls=2000]
#include <algorithm>
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
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main(int argc, char** argv) {
    const int calls = argc > 1 ? std::atoi(argv[1]) : 2000;
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
    std::printf("{\"what\": \"hip::DeviceFrameBatch::decode() from C++, dual-return 128x2048 frames, full output set\"");
    for (uint32_t n : {1u, 4u, 16u}) {
        std::vector<SensorInfo> sensors(n == 4 ? 4 : 1, info);
        for (size_t k = 0; k < sensors.size(); ++k) sensors[k].sensor_to_body(0, 3) = 0.5 * k;   // four different extrinsics
        ouster::sdk::hip::BatchOptions opt;
        opt.destagger = {"RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"};
        opt.xyz = true;
        ouster::sdk::hip::DeviceFrameBatch b(sensors, n, opt);
        for (uint32_t f = 0; f < n; ++f) {
            std::vector<const uint8_t*> ptrs;
            for (auto& p : pool[f % pool.size()]) ptrs.push_back(p.buf.data());
            b.upload_frame_packets(f, ptrs);
        }
        for (int i = 0; i < 40; ++i) b.decode();
        b.sync();
        double host = 0;
        const auto t0 = clk::now();
        for (int i = 0; i < calls; ++i) {
            const auto h0 = clk::now();
            b.decode();
            host += us(h0, clk::now());
        }
        b.sync();
        const double pipelined = us(t0, clk::now()) / calls;
        std::vector<double> lat;
        for (int i = 0; i < 200; ++i) {
            const auto s0 = clk::now();
            b.decode();
            b.sync();
            lat.push_back(us(s0, clk::now()));
        }
        std::sort(lat.begin(), lat.end());
        const double bytes = 14974976.0 * n;
        std::printf(", \"%u\": {\"frames\": %u, \"us_per_call_pipelined\": %.2f, \"us_per_call_sync\": %.2f, \"host_us_per_decode_call\": %.2f, "
                    "\"frac_pipelined\": %.4f}", n, n, pipelined, lat[lat.size() / 2], host / calls, bytes / (pipelined * 1e-6) / 8e12);
    }
    std::printf("}\n");
    return 0;
}
