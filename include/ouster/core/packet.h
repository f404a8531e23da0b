// packet.h -- packet containers (input side of FrameBatcher).
// The lidar-packet side of ouster_core/include/ouster/core/packet.h: buffer + host timestamp + format, the header
// accessors that forward to the PacketFormat, validate(), as<T>().  IMU / zone packets are out of scope of this path:
// their PacketType values exist (a FrameBatcher ignores such packets), their classes do not.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ouster/core/types.h"

namespace ouster {
namespace sdk {
namespace core {

enum class PacketType { Unknown = 0, Lidar = 1, Imu = 2, Zone = 3 };

/** Reasons for failure of packet validation (packet.h:32-36). */
enum class PacketValidationFailure { NONE = 0, PACKET_SIZE = 1, ID = 2 };

/** Size and init id / serial number check of a lidar packet against the metadata (packet.cpp:28-73). */
PacketValidationFailure validate_packet(const SensorInfo& info, const PacketFormat& format, const uint8_t* buf,
                                        uint64_t buf_size, PacketType type = PacketType::Unknown);

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

    PacketValidationFailure validate(const SensorInfo& info) const {
        return validate_packet(info, *format, buf.data(), buf.size(), type_);
    }
    PacketValidationFailure validate(const SensorInfo& info, const PacketFormat& pf) const {
        return validate_packet(info, pf, buf.data(), buf.size(), type_);
    }

    // header fields through the packet's own format (packet.h:75-150)
    auto packet_type() const { return format->packet_type(buf.data()); }
    auto frame_id() const { return format->frame_id(buf.data()); }
    auto init_id() const { return format->init_id(buf.data()); }
    auto prod_sn() const { return format->prod_sn(buf.data()); }
    auto alert_flags() const { return format->alert_flags(buf.data()); }
    auto countdown_thermal_shutdown() const { return format->countdown_thermal_shutdown(buf.data()); }
    auto countdown_shot_limiting() const { return format->countdown_shot_limiting(buf.data()); }
    auto thermal_shutdown() const { return format->thermal_shutdown(buf.data()); }
    auto shot_limiting() const { return format->shot_limiting(buf.data()); }
    auto crc() const { return format->crc(buf.data(), buf.size()); }
    auto calculate_crc() const { return format->calculate_crc(buf.data(), buf.size()); }

    /** The packet as its concrete kind.  @throw std::runtime_error when it is of another kind (packet.h:172-190) */
    template <typename Type>
    Type& as() {
        if (type() != Type::MY_TYPE) throw std::runtime_error("Tried to cast packet to incorrect type.");
        return static_cast<Type&>(*this);
    }
    template <typename Type>
    const Type& as() const {
        if (type() != Type::MY_TYPE) throw std::runtime_error("Tried to cast packet to incorrect type.");
        return static_cast<const Type&>(*this);
    }
};

struct LidarPacket : public Packet {
    static constexpr PacketType MY_TYPE = PacketType::Lidar;
    LidarPacket() : Packet(PacketType::Lidar) {}
    explicit LidarPacket(int size) : Packet(PacketType::Lidar, size) {}
    explicit LidarPacket(std::shared_ptr<PacketFormat> f)
        : Packet(PacketType::Lidar, static_cast<int>(f->lidar_packet_size)) {
        format = std::move(f);
    }

    // measurement blocks through the packet's own format (packet.h:236-300)
    auto nth_col(int n) const { return format->nth_col(n, buf.data()); }
    auto nth_px(int n, const uint8_t* col_buf) const { return format->nth_px(n, col_buf); }
    auto col_timestamp(const uint8_t* col_buf) const { return format->col_timestamp(col_buf); }
    auto col_measurement_id(const uint8_t* col_buf) const { return format->col_measurement_id(col_buf); }
    auto col_status(const uint8_t* col_buf) const { return format->col_status(col_buf); }
    template <typename T>
    void col_field(const uint8_t* col_buf, const std::string& f, T* dst, int dst_stride = 1) const {
        format->col_field<T>(col_buf, f, dst, dst_stride);
    }
    template <typename T, int BlockDim>
    void block_field(T* data, int cols, const std::string& f) const {
        format->block_field<T, BlockDim>(data, cols, f, buf.data());
    }
};

}  // namespace core
}  // namespace sdk
}  // namespace ouster
