// os_pcap.h -- the reference keeps its capture playback / recording helpers here (ouster_pcap/include/ouster/pcap/os_pcap.h).
// Of those this mirror has the reader (pcap.h) and the stream survey that tells which UDP streams of a capture belong to
// which sensor: get_stream_info (os_pcap.h:250-283) and guess_ports (os_pcap.h:296-310).  Recording and the deprecated
// replay handles are out of scope.  Like the reference's header it brings the core types along.
#pragma once
#include <chrono>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "ouster/core/lidar_frame.h"
#include "ouster/core/types.h"
#include "ouster/pcap/pcap.h"

namespace ouster {
namespace sdk {
namespace pcap {

/** One UDP stream of a capture: the address / port four-tuple. */
struct StreamKey {
    std::string dst_ip;
    std::string src_ip;
    int src_port = 0;
    int dst_port = 0;
    bool operator==(const StreamKey& o) const {
        return dst_ip == o.dst_ip && src_ip == o.src_ip && src_port == o.src_port && dst_port == o.dst_port;
    }
};

}  // namespace pcap
}  // namespace sdk
}  // namespace ouster

template <>
struct std::hash<ouster::sdk::pcap::StreamKey> {
    std::size_t operator()(const ouster::sdk::pcap::StreamKey& k) const noexcept {
        std::size_t h = std::hash<std::string>{}(k.src_ip);
        h = h * 1000003u ^ std::hash<std::string>{}(k.dst_ip);
        h = h * 1000003u ^ static_cast<std::size_t>(k.src_port);
        return h * 1000003u ^ static_cast<std::size_t>(k.dst_port);
    }
};

namespace ouster {
namespace sdk {
namespace pcap {

using ts = std::chrono::microseconds;

struct GuessedPorts {
    int lidar;  ///< 0 = no lidar stream in this guess
    int imu;    ///< 0 = no IMU stream in this guess
};

/** What was seen of one stream: datagram count, and histograms of payload size, fragment count and IP version. */
struct StreamData {
    uint64_t count = 0;
    std::map<uint64_t, uint64_t> payload_size_counts;
    std::map<uint64_t, uint64_t> fragment_counts;
    std::map<uint64_t, uint64_t> ip_version_counts;
};

struct StreamInfo {
    uint64_t total_packets = 0;
    uint32_t encapsulation_protocol = 0;  ///< pcap link type
    ts timestamp_max{0};
    ts timestamp_min{0};
    std::unordered_map<StreamKey, StreamData> udp_streams;
};

/** Survey the first `packets_to_process` datagrams of a capture (all of them when negative). */
std::shared_ptr<StreamInfo> get_stream_info(const std::string& file, int packets_to_process = -1);
/** The same with a progress callback (current offset, delta, file size) every `packets_per_callback` datagrams. */
std::shared_ptr<StreamInfo> get_stream_info(
    const std::string& file, const std::function<void(uint64_t current, uint64_t delta, uint64_t total)>& progress_callback,
    int packets_per_callback, int packets_to_process = -1);

/** Candidate (lidar port, imu port) pairs of ONE sensor: streams whose payload sizes match, lidar and IMU paired when they
 *  come from the same source address, filtered by the expected ports (0 = not known).  Pairs first, then lidar-only, then
 *  IMU-only candidates. */
std::vector<GuessedPorts> guess_ports(StreamInfo& info, int lidar_packet_size, int imu_packet_size, int expected_lidar_port,
                                      int expected_imu_port);

}  // namespace pcap
}  // namespace sdk
}  // namespace ouster
