// data_format.h -- UDP profile enums and DataFormat.
// Same names / values as ouster_core/include/ouster/core/data_format.h:24-137;
// method semantics from ouster_core/src/data_format.cpp:79-161.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "nonstd/optional.hpp"

namespace ouster {
namespace sdk {
namespace core {

constexpr int MAX_NUM_PROFILES = 32;
constexpr uint32_t DEFAULT_COLUMNS_PER_PACKET = 16;

enum class UDPProfileLidar {
    UNKNOWN = 0,
    LEGACY,
    RNG19_RFL8_SIG16_NIR16_DUAL,
    RNG19_RFL8_SIG16_NIR16,
    RNG15_RFL8_NIR8,
    FIVE_WORD_PIXEL,
    FUSA_RNG15_RFL8_NIR8_DUAL,
    RNG15_RFL8_NIR8_DUAL,
    RNG15_RFL8_NIR8_ZONE16,
    RNG19_RFL8_SIG16_NIR16_ZONE16,
    RNG15_RFL8_WIN8,
    RNG19_RFL8_SIG16_ZONE16_DUAL,
    RNG19_RFL8_SIG16_NIR16_RGB16,
    RNG19_RFL8_SIG16_NIR16_RGB16_DUAL,
    OFF = 100,
};

enum class UDPProfileIMU { LEGACY = 0, ACCEL32_GYRO32_NMEA = 1, OFF = 100 };

enum class HeaderType { STANDARD = 0, FUSA = 1 };

using ColumnWindow = std::pair<int, int>;

struct DataFormat {
    uint32_t pixels_per_column{};
    uint32_t columns_per_packet{};
    uint32_t columns_per_frame{};
    uint32_t imu_measurements_per_packet{};
    uint32_t imu_packets_per_frame{};
    std::vector<int> pixel_shift_by_row;
    ColumnWindow column_window{0, 0};
    UDPProfileLidar udp_profile_lidar{};
    UDPProfileIMU udp_profile_imu{};
    HeaderType header_type{};
    uint16_t fps{};
    bool zone_monitoring_enabled{false};

    int valid_columns_per_frame() const;
    int lidar_packets_per_frame() const;
    uint32_t max_frame_id() const;
};

bool operator==(const DataFormat& lhs, const DataFormat& rhs);
bool operator!=(const DataFormat& lhs, const DataFormat& rhs);

/** Defaults for a lidar mode of `columns` x `fps` (data_format.cpp:79-125). */
DataFormat default_data_format(uint32_t columns, uint16_t fps = 10);

std::string to_string(UDPProfileLidar profile);
/** @return nullopt when the name is not registered (data_format.h:179). */
nonstd::optional<UDPProfileLidar> udp_profile_lidar_of_string(const std::string& s);
std::string to_string(UDPProfileIMU profile);
nonstd::optional<UDPProfileIMU> udp_profile_imu_of_string(const std::string& s);
std::string to_string(HeaderType profile);
nonstd::optional<HeaderType> udp_profile_type_of_string(const std::string& s);

}  // namespace core
}  // namespace sdk
}  // namespace ouster
