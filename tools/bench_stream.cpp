// bench_stream.cpp -- PCIe-inclusive throughput of the hot path: host packets in, host results out,
// through ouster::sdk::hip::FrameStream (pinned staging, H2D / decode / D2H overlapped).
// Usage: bench_stream [frames=2048] [frames_per_batch=32] [in_flight=3] [what=xyz|xyz+planes|none|compact]
// compact: the range-gated compacting route (StreamOptions::dewarp_*, gate 0.5 - 400 m): the kept points of the first return
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "ouster/core/lidar_scan.h"
#include "ouster/hip/frame_stream.h"

using namespace ouster::sdk::core;

int main(int argc, char** argv) {
    const uint32_t total = argc > 1 ? atoi(argv[1]) : 2048;
    const uint32_t fpb = argc > 2 ? atoi(argv[2]) : 32;
    const uint32_t depth = argc > 3 ? atoi(argv[3]) : 3;
    const std::string what = argc > 4 ? argv[4] : "xyz";
    SensorInfo info;
    info.format.pixels_per_column = 128;
    info.format.columns_per_frame = 2048;
    info.format.columns_per_packet = 16;
    info.format.column_window = {0, 2047};
    info.format.udp_profile_lidar = UDPProfileLidar::RNG15_RFL8_NIR8_DUAL;
    info.format.pixel_shift_by_row.assign(128, 0);
    for (int i = 0; i < 128; ++i) info.format.pixel_shift_by_row[i] = (int[]){24, 8, -8, -24}[i % 4];
    info.beam_azimuth_angles.assign(128, 0.0);
    info.beam_altitude_angles.assign(128, 0.0);
    for (int i = 0; i < 128; ++i) {
        info.beam_azimuth_angles[i] = (double[]){4.2, 1.4, -1.4, -4.2}[i % 4];
        info.beam_altitude_angles[i] = 21.0 - 42.0 * i / 127.0;
    }
    info.prod_line = "OS-2-128";
    info.beam_to_lidar_transform = default_beam_to_lidar_transform(info.prod_line);
    info.lidar_to_sensor_transform = DEFAULT_LIDAR_TO_SENSOR;
    info.sensor_to_body = mat4d::Identity();
    info.fw_rev = "v3.2.0";
    auto pf = std::make_shared<PacketFormat>(info);
    // a pool of 8 synthetic frames -> packets
    std::vector<std::vector<LidarPacket>> pool;
    std::mt19937 g(1);
    for (int f = 0; f < 8; ++f) {
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
    ouster::sdk::hip::StreamOptions opt;
    opt.frames_per_batch = fpb;
    opt.batches_in_flight = depth;
    opt.download_xyz = what != "none";
    opt.download_headers = what != "none";
    if (what == "none") opt.outputs.xyz = true;
    if (what == "compact") {
        opt.download_xyz = false;
        opt.download_headers = false;
        opt.dewarp_min_range = 0.5;
        opt.dewarp_max_range = 400.0;
    }
    if (what == "xyz+planes") {
        opt.outputs.destagger = {"RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"};
        opt.download_planes = {"RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2", "NEAR_IR"};
        opt.download_destaggered = {"RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"};
    }
    double checksum = 0;
    uint64_t got = 0, kept = 0;
    ouster::sdk::hip::FrameStream stream({info}, opt, [&](const ouster::sdk::hip::BatchResult& r) {
        got += r.n_frames;
        kept += r.n_points;
        if (r.xyz[0]) checksum += static_cast<const float*>(r.xyz[0])[12345];
        if (r.points && r.n_points > 12345) checksum += static_cast<const float*>(r.points)[12345];
    });
    std::vector<std::vector<const uint8_t*>> ptrs(pool.size());
    for (size_t f = 0; f < pool.size(); ++f)
        for (auto& p : pool[f]) ptrs[f].push_back(p.buf.data());
    for (uint32_t f = 0; f < 4 * fpb; ++f) stream.push_frame(ptrs[f % ptrs.size()]);  // warm-up (tuner, allocs)
    stream.finish();
    const uint64_t warm = got;
    kept = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t f = 0; f < total; ++f) stream.push_frame(ptrs[f % ptrs.size()]);
    stream.finish();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double in_b = 128.0 * 16640, xyz_b = 2.0 * 128 * 2048 * 12, pl_b = what == "xyz+planes" ? 128.0 * 2048 * (4 + 4 + 1 + 1 + 2 + 10) : 0;
    const double out_b = what == "none" ? 0 : what == "compact" ? 12.0 * kept / total + (fpb + 1) * 8.0 / fpb : xyz_b + pl_b + 2048 * 14;
    std::printf("{\"frames\": %u, \"frames_per_batch\": %u, \"in_flight\": %u, \"download\": \"%s\", \"seconds\": %.4f, "
                "\"frames_per_s\": %.1f, \"Mpoints_per_s\": %.1f, \"H2D_GBps\": %.2f, \"D2H_GBps\": %.2f, \"delivered\": %llu, \"checksum\": %.3f, \"Mpixels_per_s\": %.1f, \"kept_points_per_frame\": %.0f}\n",
                total, fpb, depth, what.c_str(), s, total / s, total / s * 524288 / 1e6, total / s * in_b / 1e9,
                total / s * out_b / 1e9, (unsigned long long)(got - warm), checksum, total / s * 262144 / 1e6, (double)kept / total);
    return 0;
}
