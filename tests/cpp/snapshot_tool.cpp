// snapshot_tool.cpp -- pcap -> PcapReader -> FrameBatcher (GPU decode) -> per-field hashes, the
// C++ twin of the reference's FrameBatcherSnapshotTest (tests/frame_batcher_test.cpp:612-642):
// the hash is its matrix_hash (:600-610) with libstdc++'s identity std::hash.
// usage: snapshot_tool <capture.pcap> <profile> <header_type> <H> <W> <cpp> <init_id> <fw_rev> [stream]
// With "stream" the same packets go through hip::FrameStream::push_packet (pinned staging, batched GPU
// decode, results back in pinned memory) and the hashes are taken from the first delivered frame.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ouster/core/lidar_scan.h"
#include "ouster/hip/frame_stream.h"
#include "ouster/pcap/pcap.h"

using namespace ouster::sdk::core;

template <typename T>
static size_t raw_hash(const void* data, size_t n) {
    size_t seed = 0;
    const T* p = static_cast<const T*>(data);
    for (size_t i = 0; i < n; ++i)
        seed ^= static_cast<size_t>(p[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
}

template <typename T>
static size_t matrix_hash(const Field& f) {
    size_t seed = 0;
    const T* p = static_cast<const T*>(f.get());
    for (size_t i = 0; i < f.size(); ++i)
        seed ^= static_cast<size_t>(p[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
}

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    SensorInfo info;
    info.format.udp_profile_lidar = udp_profile_lidar_of_string(argv[2]).value_or(UDPProfileLidar::UNKNOWN);
    info.format.header_type = std::atoi(argv[3]) ? HeaderType::FUSA : HeaderType::STANDARD;
    info.format.pixels_per_column = static_cast<uint32_t>(std::atoi(argv[4]));
    info.format.columns_per_frame = static_cast<uint32_t>(std::atoi(argv[5]));
    info.format.columns_per_packet = static_cast<uint32_t>(std::atoi(argv[6]));
    info.format.column_window = {0, std::atoi(argv[5]) - 1};
    info.format.pixel_shift_by_row.assign(info.format.pixels_per_column, 0);
    info.init_id = static_cast<uint32_t>(std::strtoul(argv[7], nullptr, 10));
    info.fw_rev = argv[8];
    auto sinfo = std::make_shared<SensorInfo>(info);
    PacketFormat pf(info);
    ouster::sdk::pcap::PcapReader pcap(argv[1]);
    if (argc > 9 && std::string(argv[9]) == "stream") {
        LidarFrame layout(sinfo);  // which planes a frame of this sensor has, and their element types
        ouster::sdk::hip::StreamOptions opt;
        opt.frames_per_batch = 1;
        opt.batches_in_flight = 2;
        opt.download_xyz = false;
        for (auto it = pf.begin(); it != pf.end(); ++it)
            if (layout.has_field(it->first)) opt.download_planes.push_back(it->first);
        int delivered = 0;
        ouster::sdk::hip::FrameStream stream({info}, opt, [&](const ouster::sdk::hip::BatchResult& r) {
            if (delivered++) return;
            std::printf("stream frames %u\n", r.n_frames);
            const size_t npx = static_cast<size_t>(r.h) * r.w;
            for (const auto& name : opt.download_planes) {
                const Field& f = layout.field(name);
                size_t h = 0;
                switch (f.tag()) {
                    case ChanFieldType::UINT8: h = raw_hash<uint8_t>(r.planes.at(name), npx); break;
                    case ChanFieldType::UINT16: h = raw_hash<uint16_t>(r.planes.at(name), npx); break;
                    case ChanFieldType::UINT32: h = raw_hash<uint32_t>(r.planes.at(name), npx); break;
                    default: continue;
                }
                std::printf("%s %zu\n", name.c_str(), h);
            }
        });
        while (pcap.next_packet()) {
            if (pcap.current_info().dst_port != 7502) continue;
            LidarPacket packet;
            packet.host_timestamp = 1234;
            packet.buf.assign(pcap.current_data(), pcap.current_data() + pcap.current_length());
            if (packet.buf.size() != pf.lidar_packet_size) continue;
            packet.buf.resize(packet.buf.size() + 8, 0);
            stream.push_packet(packet);
        }
        stream.finish();
        if (!delivered) std::printf("stream frames 0\n");
        return 0;
    }
    LidarFrame frame(sinfo);
    FrameBatcher batcher(sinfo);
    int lidar_packets = 0, complete_at = -1;
    while (pcap.next_packet()) {
        if (pcap.current_info().dst_port != 7502) continue;
        LidarPacket packet;
        packet.host_timestamp = 1234;
        packet.buf.assign(pcap.current_data(), pcap.current_data() + pcap.current_length());
        if (packet.buf.size() != pf.lidar_packet_size) continue;
        packet.buf.resize(packet.buf.size() + 8, 0);  // headroom for 64-bit field windows
        if (batcher(packet, frame) && complete_at < 0) complete_at = lidar_packets;
        ++lidar_packets;
    }
    // an incomplete capture (the FUSA fixture has only 8 packets): the reference's batcher has
    // decoded those packets already; ours decodes on release, so ask for the partial frame
    if (complete_at < 0) batcher.flush(frame);
    std::printf("packets %d complete_at %d frame_id %lld\n", lidar_packets, complete_at,
                static_cast<long long>(frame.frame_id));
    for (auto it = pf.begin(); it != pf.end(); ++it) {
        if (!frame.has_field(it->first)) continue;
        const Field& f = frame.field(it->first);
        size_t h = 0;
        switch (f.tag()) {
            case ChanFieldType::UINT8: h = matrix_hash<uint8_t>(f); break;
            case ChanFieldType::UINT16: h = matrix_hash<uint16_t>(f); break;
            case ChanFieldType::UINT32: h = matrix_hash<uint32_t>(f); break;
            case ChanFieldType::UINT64: h = matrix_hash<uint64_t>(f); break;
            default: continue;
        }
        std::printf("%s %zu\n", it->first.c_str(), h);
    }
    return 0;
}
