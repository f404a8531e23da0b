// types.h -- PacketFormat, SensorInfo (calibration subset) and related types.
//
// PacketFormat keeps the public surface of the reference class
// (ouster_core/include/ouster/core/types.h:109-1000, implementation
// ouster_core/src/parsing.cpp:386-1321): const geometry members, header accessors,
// column accessors, field iteration, setters, CRC.  Header / column accessors run on the
// host (they touch a handful of bytes per packet and drive FrameBatcher's state machine);
// the per-pixel loops col_field / block_field and everything FrameBatcher decodes run on
// the GPU through the C ABI in include/ouster_hip.h.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "ouster/core/visibility.h"
#include "ouster/core/chanfield.h"
#include "ouster/core/data_format.h"
#include "ouster/core/field_decode_info.h"
#include "ouster/core/typedefs.h"

struct ouster_hip_format_desc;  // include/ouster_hip.h

namespace ouster {
namespace sdk {
namespace core {

/** Unit of range from sensor packet, in meters (types.h:46). */
constexpr double RANGE_UNIT = 0.001;

enum class ThermalShutdownStatus { NORMAL = 0x00, IMMINENT = 0x01 };
enum class ShotLimitingStatus {
    NORMAL = 0x00, IMMINENT = 0x01,
    REDUCTION_0_10 = 0x02, REDUCTION_10_20 = 0x03, REDUCTION_20_30 = 0x04,
    REDUCTION_30_40 = 0x05, REDUCTION_40_50 = 0x06, REDUCTION_50_60 = 0x07,
    REDUCTION_60_70 = 0x08, REDUCTION_70_75 = 0x09
};

std::string to_string(ThermalShutdownStatus status);   ///< sensor_info.cpp:505-514
std::string to_string(ShotLimitingStatus status);

/** Firmware version triple (lidar_frame.cpp:1097-1110 compares against 3.2.0). */
struct Version {
    uint16_t major = 0, minor = 0, patch = 0;
    Version() = default;
    Version(uint16_t a, uint16_t b, uint16_t c) : major(a), minor(b), patch(c) {}
    bool operator<(const Version& o) const {
        if (major != o.major) return major < o.major;
        if (minor != o.minor) return minor < o.minor;
        return patch < o.patch;
    }
};

/** Output resolution and frame rate of the lidar (sensor_config.h:27-63). */
struct LidarMode {
    /** From "COLUMNSxFPS".  @throw std::invalid_argument("Invalid lidar mode string ...") */
    explicit LidarMode(const std::string& mode);
    LidarMode(unsigned int cols, unsigned int framerate) : columns(cols), fps(framerate) {}
    unsigned int columns;
    unsigned int fps;
    static const LidarMode _512x10, _512x20, _1024x10, _1024x20, _2048x10, _4096x5;
};
inline bool operator==(const LidarMode& a, const LidarMode& b) { return a.columns == b.columns && a.fps == b.fps; }
inline bool operator!=(const LidarMode& a, const LidarMode& b) { return !(a == b); }
std::string to_string(LidarMode mode);
nonstd::optional<LidarMode> lidar_mode_of_string(const std::string& s);
inline uint32_t n_cols_of_lidar_mode(LidarMode mode) { return mode.columns; }
inline unsigned int frequency_of_lidar_mode(LidarMode mode) { return mode.fps; }

/** The two entries of the reference's SensorConfig (sensor_config.h:233-420) that describe the data this path decodes;
 *  the rest of the sensor configuration (ports, sync, NMEA ...) is out of scope. */
struct SensorConfig {
    nonstd::optional<LidarMode> lidar_mode;
    nonstd::optional<UDPProfileLidar> udp_profile_lidar;
    nonstd::optional<UDPProfileIMU> udp_profile_imu;
    nonstd::optional<int> udp_port_lidar;   ///< destination ports of the sensor's streams (sensor_config.h); 0 = stream disabled
    nonstd::optional<int> udp_port_imu;
    nonstd::optional<int> udp_port_zm;
};

template <typename T> class XYZLutT;

/**
 * Sensor metadata needed by the hot path: data format + calibration.  Field names match
 * ouster_core/include/ouster/core/sensor_info.h:187-211.  Fill the members directly, or read the
 * data-format / calibration subset of a metadata JSON (SensorInfo(json_text), metadata_from_json(path),
 * csrc/host/metadata.cpp); the rest of the reference's metadata handling is out of scope.
 */
class SensorInfo {
   public:
    uint64_t sn{};
    std::string fw_rev{};
    std::string image_rev{};   ///< image_rev of the metadata (sensor_info.h:192); the same string as fw_rev on current firmware
    std::string prod_line{};
    DataFormat format{};
    SensorConfig config{};
    std::vector<double> beam_azimuth_angles{};
    std::vector<double> beam_altitude_angles{};
    double lidar_origin_to_beam_origin_mm{};
    mat4d beam_to_lidar_transform = mat4d::Zero();
    mat4d imu_to_sensor_transform = mat4d::Zero();
    mat4d lidar_to_sensor_transform = mat4d::Zero();
    mat4d sensor_to_body = mat4d::Identity();
    uint32_t init_id{};

    SensorInfo() = default;
    /** From the text of a metadata JSON, either generation (sensor_info.h:229, metadata.cpp:482-842).
     *  @throw std::runtime_error on malformed JSON or an unknown lidar profile */
    explicit SensorInfo(const std::string& metadata_json);
    Version get_version() const;  ///< parsed from fw_rev ("v2.3.0" ...)
    uint32_t w() const { return format.columns_per_frame; }
    uint32_t h() const { return format.pixels_per_column; }
    int num_returns() const;

    /** Lazily built, cached lookup table (sensor_info.cpp:260-275; use_extrinsics = true). */
    template <typename T>
    std::shared_ptr<const XYZLutT<T>> xyzlut() const;

   private:
    struct Cache;
    mutable std::shared_ptr<Cache> cache_;
};

/** Member-wise equality over what this mirror holds (sensor_info.cpp:128-151). */
bool operator==(const SensorInfo& lhs, const SensorInfo& rhs);
inline bool operator!=(const SensorInfo& lhs, const SensorInfo& rhs) { return !(lhs == rhs); }

/** Read a metadata JSON file (sensor_info.h:413).  `skip_beam_validation` is accepted and ignored: nothing is validated. */
SensorInfo metadata_from_json(const std::string& json_file, bool skip_beam_validation = false);

/** sensor_info.cpp:89-105 */
double default_lidar_origin_to_beam_origin(const std::string& prod_line);
mat4d default_beam_to_lidar_transform(const std::string& prod_line);
/** metadata.cpp:54-55 */
extern const mat4d DEFAULT_LIDAR_TO_SENSOR;

class PacketFormat {
   public:
    struct Impl;

   protected:
    std::shared_ptr<const Impl> impl_;
    std::vector<std::pair<std::string, std::pair<ChanFieldType, int>>> field_types_;

   public:
    PacketFormat(const DataFormat& format);
    PacketFormat(const SensorInfo& info);

    using FieldIter = decltype(field_types_)::const_iterator;

    const UDPProfileLidar udp_profile_lidar;
    const UDPProfileIMU udp_profile_imu;
    const HeaderType header_type;
    const size_t lidar_packet_size;
    const size_t imu_packet_size;    ///< sizes only: IMU / zone packets are recognised (packet type by size), not parsed
    const size_t zone_packet_size;
    const uint32_t columns_per_packet;
    const uint32_t pixels_per_column;
    const size_t packet_header_size;
    const size_t col_header_size;
    const size_t col_footer_size;
    const size_t col_size;
    const size_t packet_footer_size;
    const uint32_t max_frame_id;

    // packet headers (parsing.cpp:736-770)
    uint16_t packet_type(const uint8_t* packet_buf) const;
    uint32_t frame_id(const uint8_t* packet_buf) const;
    uint32_t init_id(const uint8_t* packet_buf) const;
    uint64_t prod_sn(const uint8_t* packet_buf) const;
    uint8_t alert_flags(const uint8_t* lidar_buf) const;
    uint16_t countdown_thermal_shutdown(const uint8_t* lidar_buf) const;
    uint16_t countdown_shot_limiting(const uint8_t* lidar_buf) const;
    ThermalShutdownStatus thermal_shutdown(const uint8_t* lidar_buf) const;
    ShotLimitingStatus shot_limiting(const uint8_t* lidar_buf) const;

    ChanFieldType field_type(const std::string& f) const;
    FieldIter begin() const;
    FieldIter end() const;

    const uint8_t* footer(const uint8_t* lidar_buf) const;
    uint8_t* footer(uint8_t* lidar_buf) const;

    // measurement block access (parsing.cpp:786-836)
    const uint8_t* nth_col(size_t n, const uint8_t* lidar_buf) const;
    uint8_t* nth_col(size_t n, uint8_t* lidar_buf) const;
    uint64_t col_timestamp(const uint8_t* col_buf) const;
    uint16_t col_measurement_id(const uint8_t* col_buf) const;
    uint32_t col_status(const uint8_t* col_buf) const;
    uint32_t col_encoder(const uint8_t* col_buf) const;
    uint16_t col_frame_id(const uint8_t* col_buf) const;
    const uint8_t* nth_px(size_t n, const uint8_t* col_buf) const;
    uint8_t* nth_px(size_t n, uint8_t* col_buf) const;

    /**
     * Copy one field of one column into a strided destination (parsing.cpp:659-675).
     * Runs on the GPU: the packet column is staged, decoded by the decode kernel and the H
     * values are copied back.  Meant for API compatibility; bulk decode should use
     * FrameBatcher / the batched C ABI.
     * @throw std::invalid_argument("Dest type too small for specified field")
     */
    template <typename T>
    void col_field(const uint8_t* col_buf, const std::string& f, T* dst, int dst_stride = 1) const;

    /** Largest of 16/8/4 dividing both H and columns_per_packet, else 0 (parsing.cpp:958-966). */
    int block_parsable() const;

    /**
     * Decode one field of a whole packet into a row-major H x cols plane at the packet's
     * measurement ids (parsing.cpp:628-657).  GPU-backed like col_field.
     */
    template <typename T, int BlockDim>
    void block_field(T* data, int cols, const std::string& f, const uint8_t* lidar_buf) const;

    uint64_t field_value_mask(const std::string& f) const;
    int field_bitness(const std::string& f) const;

    // setters (parsing.cpp:1007-1090)
    void set_col_status(uint8_t* col_buf, uint32_t status) const;
    void set_col_timestamp(uint8_t* col_buf, uint64_t ts) const;
    void set_col_measurement_id(uint8_t* col_buf, uint16_t m_id) const;
    void set_frame_id(uint8_t* lidar_buf, uint32_t frame_id) const;
    void set_init_id(uint8_t* lidar_buf, uint32_t init_id) const;
    void set_packet_type(uint8_t* packet_buf, uint16_t packet_type) const;
    void set_prod_sn(uint8_t* lidar_buf, uint64_t sn) const;
    void set_alert_flags(uint8_t* lidar_buf, uint8_t alert_flags) const;
    void set_shutdown(uint8_t* lidar_buf, uint8_t status) const;
    void set_shot_limiting(uint8_t* lidar_buf, uint8_t status) const;
    void set_shutdown_countdown(uint8_t* lidar_buf, uint8_t v) const;
    void set_shot_limiting_countdown(uint8_t* lidar_buf, uint8_t v) const;

    /** Encode one field of a row-major H x cols plane into a packet (host; test-side packet
     * synthesis, parsing.cpp:1056-1090).  Skips columns whose status bit 0 is clear. */
    template <typename T>
    void set_block(const T* data, int cols, const std::string& f, uint8_t* lidar_buf) const;

    /** Stored CRC64 of a packet; `has == false` for LEGACY / FUSA (parsing.cpp:1219-1228). */
    bool crc(const uint8_t* buffer, size_t buffer_size, uint64_t& out) const;
    /** The reference's spelling (types.h:618): nullopt where the format carries no CRC. */
    nonstd::optional<uint64_t> crc(const uint8_t* buffer, size_t buffer_size) const {
        uint64_t v = 0;
        if (!crc(buffer, buffer_size, v)) return nonstd::nullopt;
        return v;
    }
    uint64_t calculate_crc(const uint8_t* buffer, size_t buffer_size) const;

    int frame_id_difference(uint32_t current, uint32_t other) const;

    /** Bit layout of a channel field of this format. @throw std::out_of_range if absent. */
    const FieldDecodeInfo& field_decode_info(const std::string& f) const;
    size_t channel_data_size() const;

    /**
     * POD description handed to ouster_hip_format_create: geometry, header bit fields and
     * the given (field name, destination element bytes) pairs.
     */
    void fill_hip_desc(uint32_t columns_per_frame,
                       const std::vector<std::pair<std::string, uint32_t>>& fields,
                       const std::vector<bool>& f16_nan, ::ouster_hip_format_desc& out) const;
};

const PacketFormat& get_format(const SensorInfo& info);
const PacketFormat& get_format(const DataFormat& format);

namespace impl {
/** 3 x float16 packed pixel (RGB planes), impl/lidar_frame_impl.h:35-41. */
#pragma pack(push, 1)
struct float3x16_t {
    uint16_t a, b, c;
};
#pragma pack(pop)
}  // namespace impl

}  // namespace core
}  // namespace sdk
}  // namespace ouster
