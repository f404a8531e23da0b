// packet.h -- packet containers (input side of FrameBatcher).
// Subset of ouster_core/include/ouster/core/packet.h: buffer + host timestamp + format.
// IMU / zone packets are out of scope of this path.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "ouster/core/types.h"

namespace ouster {
namespace sdk {
namespace core {

enum class PacketType { Unknown = 0, Lidar = 1, Imu = 2, Zone = 3 };

struct Packet {
    PacketType type_ = PacketType::Unknown;
    uint64_t host_timestamp = 0;
    std::vector<uint8_t> buf;
    std::shared_ptr<PacketFormat> format;

    Packet() = default;
    explicit Packet(PacketType t) : type_(t) {}
    Packet(PacketType t, int size) : type_(t) {
        buf.reserve(size + 1);
        buf.resize(size, 0);
    }
    PacketType type() const { return type_; }
};

struct LidarPacket : public Packet {
    LidarPacket() : Packet(PacketType::Lidar) {}
    explicit LidarPacket(int size) : Packet(PacketType::Lidar, size) {}
    explicit LidarPacket(std::shared_ptr<PacketFormat> f)
        : Packet(PacketType::Lidar, static_cast<int>(f->lidar_packet_size)) {
        format = std::move(f);
    }
};

}  // namespace core
}  // namespace sdk
}  // namespace ouster
