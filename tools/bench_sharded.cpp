// bench_sharded.cpp -- BASELINE.json configs[3] (a batch of 512 OS-1-128 2048x128 single-return frames sharded over the GPUs
// of one node) through ouster::sdk::hip::ShardedBatch: ONE C++ process, no torch, no launcher.  The batch is staged on the
// root GPU, scattered to the shards (peer copies over xGMI, every shard on its own stream), decoded (fused decode + destagger
// + cartesian per shard), and the clouds are gathered back on the root GPU.  Prints one JSON line: per-phase milliseconds on
// the slowest shard, Mpoints/s decode-only and with the exchange, and a checksum of the gathered clouds that does not depend
// on the number of shards (the Python face of the same sharding is bench.py --gpus N --workload batch512).
// Usage: bench_sharded [frames=512] [shards=0 (one per visible GPU); k > 0: k shards, round-robin over the GPUs] [reps=5]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "ouster/core/lidar_scan.h"
#include "ouster/hip/sharded_batch.h"

using namespace ouster::sdk::core;

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? std::atoi(argv[1]) : 512;
    const int want_shards = argc > 2 ? std::atoi(argv[2]) : 0;
    const int reps = argc > 3 ? std::atoi(argv[3]) : 5;
    SensorInfo info;
    info.format.pixels_per_column = 128;
    info.format.columns_per_frame = 2048;
    info.format.columns_per_packet = 16;
    info.format.column_window = {0, 2047};
    info.format.udp_profile_lidar = UDPProfileLidar::RNG19_RFL8_SIG16_NIR16;
    for (int i = 0; i < 128; ++i) {
        info.format.pixel_shift_by_row.push_back((int[]){24, 8, -8, -24}[i % 4]);
        info.beam_azimuth_angles.push_back((double[]){4.2, 1.4, -1.4, -4.2}[i % 4]);
        info.beam_altitude_angles.push_back(21.0 - 42.0 * i / 127.0);
    }
    info.prod_line = "OS-1-128";
    info.beam_to_lidar_transform = default_beam_to_lidar_transform(info.prod_line);
    info.lidar_to_sensor_transform = DEFAULT_LIDAR_TO_SENSOR;
    info.sensor_to_body = mat4d::Identity();
    info.fw_rev = "v3.2.0";
    auto pf = std::make_shared<PacketFormat>(info);
    // a pool of 8 synthetic frames ("recorded" frames come through tools/config4_recorded.py / pcap::IndexedPcapReader)
    std::vector<std::vector<LidarPacket>> pool;
    std::mt19937 g(7);
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
    const int ndev = ouster::sdk::hip::device_count();
    if (ndev < 1) { std::fprintf(stderr, "bench_sharded needs a GPU\n"); return 2; }
    std::vector<int> devices;
    for (int s = 0; s < want_shards; ++s) devices.push_back(s % ndev);
    ouster::sdk::hip::BatchOptions opt;
    opt.destagger = {"RANGE", "REFLECTIVITY"};
    opt.xyz = true;
    ouster::sdk::hip::ShardedBatch sb({info}, n, opt, devices);
    for (uint32_t f = 0; f < n; ++f) {
        std::vector<const uint8_t*> ptrs;
        for (auto& p : pool[f % pool.size()]) ptrs.push_back(p.buf.data());
        sb.upload_frame_packets(f, ptrs);
    }
    double ms[3] = {0, 0, 0};
    for (int r = -2; r < reps; ++r) {   // two warm-up rounds (tuner, peer mappings)
        sb.scatter();
        sb.decode();
        sb.gather_xyz(0);
        sb.sync();
        if (r >= 0) { ms[0] += sb.last_scatter_ms(); ms[1] += sb.last_decode_ms(); ms[2] += sb.last_gather_ms(); }
    }
    for (double& m : ms) m /= reps;
    // checksum of the gathered cloud: every 16th frame
    uint64_t h = 1469598103934665603ull;
    std::vector<float> xyz(static_cast<size_t>(128) * 2048 * 3);
    for (uint32_t f = 0; f < n; f += 16) {
        sb.download_xyz_root(0, f, xyz.data());
        const uint32_t* w = reinterpret_cast<const uint32_t*>(xyz.data());
        for (size_t i = 0; i < xyz.size(); ++i) h = (h ^ w[i]) * 1099511628211ull;
    }
    const double pts = static_cast<double>(n) * 128 * 2048;
    const double in_b = static_cast<double>(n) * 128 * 24832, out_b = pts * 12;
    const int ns = sb.n_shards();
    std::printf("{\"workload\": \"configs[3]: %u OS-1-128 2048x128 RNG19_RFL8_SIG16_NIR16 frames over %d shard(s) on %d GPU(s), hip::ShardedBatch\", "
                "\"frames\": %u, \"shards\": %d, \"gpus_visible\": %d, \"scatter_ms\": %.3f, \"decode_ms\": %.3f, \"gather_xyz_ms\": %.3f, "
                "\"Mpoints_per_s_decode\": %.1f, \"Mpoints_per_s_with_exchange\": %.1f, \"scatter_GBps\": %.1f, \"gather_GBps\": %.1f, "
                "\"xyz_checksum\": \"%016llx\"}\n",
                n, ns, ndev, n, ns, ndev, ms[0], ms[1], ms[2], pts / ms[1] / 1e3, pts / (ms[0] + ms[1] + ms[2]) / 1e3,
                in_b * (ns - 1) / ns / std::max(ms[0], 1e-6) / 1e6, out_b * (ns - 1) / ns / std::max(ms[2], 1e-6) / 1e6,
                (unsigned long long)h);
    return 0;
}
