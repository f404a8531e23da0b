// packet_format.cpp -- host side of ouster::sdk::core::PacketFormat and friends.
//
// Behaviour follows the reference (paths relative to the reference checkout):
//   field_info                      ouster_core/src/parsing.cpp:57-122
//   per-profile bit layouts          ouster_core/src/parsing.cpp:170-363
//   geometry + header bit fields     ouster_core/src/parsing.cpp:453-538
//   accessors / setters / crc        ouster_core/src/parsing.cpp:736-836, 1007-1090, 1183-1234
//   custom profiles                  ouster_core/src/profile_extension.cpp:130-183
//   DataFormat helpers               ouster_core/src/data_format.cpp:79-161
// The structure is this project's own: layouts live in one flat table keyed by profile,
// and the per-pixel loops (col_field / block_field) are GPU calls (hip_runtime.cpp).
#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <tuple>

#include "host_internal.h"
#include "ouster/core/profile_extension.h"
#include "ouster/core/packet.h"
#include "ouster/core/types.h"
#include "ouster_hip.h"

namespace ouster {
namespace sdk {
namespace core {

// ---------------------------------------------------------------------------------------
// chanfield / data format helpers
// ---------------------------------------------------------------------------------------
std::string to_string(ChanFieldType t) {
    switch (t) {
        case ChanFieldType::VOID: return "VOID";
        case ChanFieldType::UINT8: return "UINT8";
        case ChanFieldType::UINT16: return "UINT16";
        case ChanFieldType::UINT32: return "UINT32";
        case ChanFieldType::UINT64: return "UINT64";
        case ChanFieldType::INT8: return "INT8";
        case ChanFieldType::INT16: return "INT16";
        case ChanFieldType::INT32: return "INT32";
        case ChanFieldType::INT64: return "INT64";
        case ChanFieldType::FLOAT32: return "FLOAT32";
        case ChanFieldType::FLOAT64: return "FLOAT64";
        case ChanFieldType::CHAR: return "CHAR";
        case ChanFieldType::FLOAT16: return "FLOAT16";
        default: return "UNKNOWN";
    }
}

bool operator==(const DataFormat& a, const DataFormat& b) {
    return a.pixels_per_column == b.pixels_per_column &&
           a.columns_per_packet == b.columns_per_packet &&
           a.columns_per_frame == b.columns_per_frame &&
           a.imu_measurements_per_packet == b.imu_measurements_per_packet &&
           a.pixel_shift_by_row == b.pixel_shift_by_row && a.column_window == b.column_window &&
           a.udp_profile_lidar == b.udp_profile_lidar && a.header_type == b.header_type &&
           a.udp_profile_imu == b.udp_profile_imu && a.fps == b.fps &&
           a.zone_monitoring_enabled == b.zone_monitoring_enabled;
}
bool operator!=(const DataFormat& a, const DataFormat& b) { return !(a == b); }

DataFormat default_data_format(uint32_t columns, uint16_t fps) {
    int unit;
    switch (columns) {
        case 512: unit = 3; break;
        case 1024: unit = 6; break;
        case 2048: unit = 12; break;
        case 4096: unit = 24; break;
        default: throw std::invalid_argument{"default_data_format"};
    }
    DataFormat f;
    f.pixels_per_column = 64;
    f.columns_per_packet = DEFAULT_COLUMNS_PER_PACKET;
    f.columns_per_frame = columns;
    for (int i = 0; i < 16; ++i)
        for (int k = 3; k >= 0; --k) f.pixel_shift_by_row.push_back(k * unit);
    f.column_window = {0, static_cast<int>(columns) - 1};
    f.udp_profile_lidar = UDPProfileLidar::LEGACY;
    f.udp_profile_imu = UDPProfileIMU::LEGACY;
    f.header_type = HeaderType::STANDARD;
    f.fps = fps;
    return f;
}

int DataFormat::valid_columns_per_frame() const {
    const int a = column_window.first, b = column_window.second;
    return a <= b ? b - a + 1 : static_cast<int>(b + (columns_per_frame - a) + 1);
}

int DataFormat::lidar_packets_per_frame() const {
    if (udp_profile_lidar == UDPProfileLidar::OFF) return 0;
    const int first = static_cast<int>(column_window.first / columns_per_packet);
    const int last = static_cast<int>(column_window.second / columns_per_packet);
    if (column_window.second >= column_window.first) return last - first + 1;
    // window wraps through column 0
    const int all = static_cast<int>(columns_per_frame / columns_per_packet) +
                    ((columns_per_frame % columns_per_packet) != 0u ? 1 : 0);
    return first == last ? all : (all - first) + 1 + last;
}

uint32_t DataFormat::max_frame_id() const {
    if (header_type == HeaderType::FUSA && udp_profile_lidar != UDPProfileLidar::LEGACY)
        return std::numeric_limits<uint32_t>::max();
    return std::numeric_limits<uint16_t>::max();
}

// ---------------------------------------------------------------------------------------
// field_info
// ---------------------------------------------------------------------------------------
FieldDecodeInfo field_info(size_t bit_start, size_t bit_size, size_t upshift, size_t max_length,
                           size_t num_elements) {
    const size_t total_bits = bit_size + upshift;
    if (total_bits > 64)
        throw std::invalid_argument(
            "failed creating FieldDecodeInfo: value cannot store more than 64 bits");
    FieldDecodeInfo f{};
    f.offset = bit_start / 8;
    const size_t first_bit = bit_start % 8;
    f.mask = bit_size == 0 ? 0
                           : ((bit_size >= 64 ? ~uint64_t{0} : ((uint64_t{1} << bit_size) - 1))
                              << first_bit);
    f.shift = static_cast<int>(first_bit) - static_cast<int>(upshift);
    f.num_elements = static_cast<int>(num_elements);
    size_t bytes = (total_bits + 7) / 8 / num_elements;
    f.ty_tag = bytes == 1   ? ChanFieldType::UINT8
               : bytes == 2 ? ChanFieldType::UINT16
               : bytes <= 4 && bytes >= 3 ? ChanFieldType::UINT32
               : bytes >= 5 && bytes <= 8 ? ChanFieldType::UINT64
                                          : ChanFieldType::VOID;
    if (max_length > 0) {
        if (f.offset + bytes > max_length)
            throw std::invalid_argument(
                "failed creating FieldDecodeInfo: asked to read past end of packet");
        // keep the 8-byte access inside the buffer by reading from further back
        const int back = static_cast<int>(f.offset) + 8 - static_cast<int>(max_length);
        if (back > 0) {
            f.offset -= back;
            f.mask <<= back * 8;
            f.shift += back * 8;
        }
    }
    return f;
}

namespace impl {
uint64_t get_value_mask(const FieldDecodeInfo& f) {
    const uint64_t tm = field_type_mask(f.ty_tag);
    uint64_t m = f.mask ? f.mask : tm;
    if (f.shift > 0) m >>= f.shift;
    if (f.shift < 0) m <<= -f.shift;
    return m & tm;
}
int get_bitness(const FieldDecodeInfo& f) {
    return __builtin_popcountll(get_value_mask(f));
}
}  // namespace impl

// ---------------------------------------------------------------------------------------
// profile registry
// ---------------------------------------------------------------------------------------
namespace {

struct Bit {  // one row of the wire layout table: value at bits [start, start+size) << up
    const char* name;
    uint16_t start, size, up, n;
};

using namespace ChanField;
#define RAW_WORDS_1 {RAW32_WORD1, 0, 32, 0, 1}
#define RAW_WORDS_2 RAW_WORDS_1, {RAW32_WORD2, 32, 32, 0, 1}
#define RAW_WORDS_3 RAW_WORDS_2, {RAW32_WORD3, 64, 32, 0, 1}
#define RAW_WORDS_4 RAW_WORDS_3, {RAW32_WORD4, 96, 32, 0, 1}
#define RAW_WORDS_5 RAW_WORDS_4, {RAW32_WORD5, 128, 32, 0, 1}
// first/second return words shared by the 19-bit dual layouts
#define DUAL19_HEAD                                                                        \
    {RANGE, 0, 19, 0, 1}, {FLAGS, 19, 5, 0, 1}, {REFLECTIVITY, 24, 8, 0, 1},               \
        {RANGE2, 32, 19, 0, 1}, {FLAGS2, 51, 5, 0, 1}, {REFLECTIVITY2, 56, 8, 0, 1},       \
        {SIGNAL, 64, 16, 0, 1}, {SIGNAL2, 80, 16, 0, 1}
#define LB_HEAD {RANGE, 0, 15, 3, 1}, {FLAGS, 15, 1, 0, 1}, {REFLECTIVITY, 16, 8, 0, 1}

struct Layout {
    UDPProfileLidar profile;
    size_t chan_bytes;
    std::vector<Bit> bits;
    std::vector<std::pair<const char*, ChanFieldType>> planes;  // default LidarFrame planes
};

const std::vector<Layout>& builtin_layouts() {
    using P = UDPProfileLidar;
    using T = ChanFieldType;
    static const std::vector<Layout> tbl = {
        {P::LEGACY, 12,
         {{RANGE, 0, 20, 0, 1}, {FLAGS, 28, 4, 0, 1}, {REFLECTIVITY, 32, 8, 0, 1},
          {SIGNAL, 48, 16, 0, 1}, {NEAR_IR, 64, 16, 0, 1}, RAW_WORDS_3},
         {{RANGE, T::UINT32}, {SIGNAL, T::UINT16}, {NEAR_IR, T::UINT16},
          {REFLECTIVITY, T::UINT8}, {FLAGS, T::UINT8}}},
        {P::RNG19_RFL8_SIG16_NIR16_DUAL, 16,
         {DUAL19_HEAD, {NEAR_IR, 96, 16, 0, 1}, {WINDOW, 120, 8, 0, 1}, RAW_WORDS_4},
         {{RANGE, T::UINT32}, {RANGE2, T::UINT32}, {SIGNAL, T::UINT16}, {SIGNAL2, T::UINT16},
          {REFLECTIVITY, T::UINT8}, {REFLECTIVITY2, T::UINT8}, {FLAGS, T::UINT8},
          {FLAGS2, T::UINT8}, {NEAR_IR, T::UINT16}, {WINDOW, T::UINT8}}},
        {P::RNG19_RFL8_SIG16_NIR16, 12,
         {{RANGE, 0, 19, 0, 1}, {FLAGS, 19, 5, 0, 1}, {REFLECTIVITY, 32, 8, 0, 1},
          {SIGNAL, 48, 16, 0, 1}, {NEAR_IR, 64, 16, 0, 1}, {WINDOW, 88, 8, 0, 1}, RAW_WORDS_3},
         {{RANGE, T::UINT32}, {SIGNAL, T::UINT16}, {REFLECTIVITY, T::UINT8}, {FLAGS, T::UINT8},
          {NEAR_IR, T::UINT16}, {WINDOW, T::UINT8}}},
        {P::RNG15_RFL8_NIR8, 4,
         {LB_HEAD, {NEAR_IR, 24, 8, 4, 1}, RAW_WORDS_1},
         {{RANGE, T::UINT32}, {REFLECTIVITY, T::UINT8}, {NEAR_IR, T::UINT16}, {FLAGS, T::UINT8}}},
        {P::RNG15_RFL8_WIN8, 4,
         {LB_HEAD, {WINDOW, 24, 8, 0, 1}, RAW_WORDS_1},
         {{RANGE, T::UINT32}, {REFLECTIVITY, T::UINT8}, {WINDOW, T::UINT8}, {FLAGS, T::UINT8}}},
        {P::FIVE_WORD_PIXEL, 20,
         {DUAL19_HEAD, {NEAR_IR, 96, 16, 0, 1}, RAW_WORDS_5},
         {{RAW32_WORD1, T::UINT32}, {RAW32_WORD2, T::UINT32}, {RAW32_WORD3, T::UINT32},
          {RAW32_WORD4, T::UINT32}, {RAW32_WORD5, T::UINT32}}},
        {P::FUSA_RNG15_RFL8_NIR8_DUAL, 8, {}, {}},  // filled from RNG15_RFL8_NIR8_DUAL below
        {P::RNG15_RFL8_NIR8_DUAL, 8,
         {LB_HEAD, {NEAR_IR, 24, 8, 4, 1}, {RANGE2, 32, 15, 3, 1}, {FLAGS2, 47, 1, 0, 1},
          {REFLECTIVITY2, 48, 8, 0, 1}, {WINDOW, 56, 8, 0, 1}, RAW_WORDS_2},
         {{RANGE, T::UINT32}, {REFLECTIVITY, T::UINT8}, {NEAR_IR, T::UINT16}, {RANGE2, T::UINT32},
          {REFLECTIVITY2, T::UINT8}, {FLAGS, T::UINT8}, {FLAGS2, T::UINT8}, {WINDOW, T::UINT8}}},
        {P::RNG15_RFL8_NIR8_ZONE16, 8,
         {LB_HEAD, {NEAR_IR, 24, 8, 4, 1}, {ZONE_MASK, 32, 16, 0, 1}, {WINDOW, 48, 8, 0, 1},
          RAW_WORDS_2},
         {{RANGE, T::UINT32}, {REFLECTIVITY, T::UINT8}, {NEAR_IR, T::UINT16}, {FLAGS, T::UINT8},
          {ZONE_MASK, T::UINT16}, {WINDOW, T::UINT8}}},
        {P::RNG19_RFL8_SIG16_NIR16_ZONE16, 12,
         {{RANGE, 0, 19, 0, 1}, {FLAGS, 19, 5, 0, 1}, {REFLECTIVITY, 32, 8, 0, 1},
          {WINDOW, 40, 8, 0, 1}, {SIGNAL, 48, 16, 0, 1}, {NEAR_IR, 64, 16, 0, 1},
          {ZONE_MASK, 80, 16, 0, 1}, RAW_WORDS_3},
         {{RANGE, T::UINT32}, {SIGNAL, T::UINT16}, {REFLECTIVITY, T::UINT8}, {FLAGS, T::UINT8},
          {NEAR_IR, T::UINT16}, {ZONE_MASK, T::UINT16}, {WINDOW, T::UINT8}}},
        {P::RNG19_RFL8_SIG16_ZONE16_DUAL, 16,
         {DUAL19_HEAD, {ZONE_MASK, 96, 16, 0, 1}, {WINDOW, 120, 8, 0, 1}, RAW_WORDS_4},
         {{RANGE, T::UINT32}, {RANGE2, T::UINT32}, {SIGNAL, T::UINT16}, {SIGNAL2, T::UINT16},
          {REFLECTIVITY, T::UINT8}, {REFLECTIVITY2, T::UINT8}, {FLAGS, T::UINT8},
          {FLAGS2, T::UINT8}, {ZONE_MASK, T::UINT16}, {WINDOW, T::UINT8}}},
        {P::RNG19_RFL8_SIG16_NIR16_RGB16, 16,
         {{RANGE, 0, 19, 0, 1}, {FLAGS, 19, 5, 0, 1}, {REFLECTIVITY, 24, 8, 0, 1},
          {SIGNAL, 32, 16, 0, 1}, {NEAR_IR, 48, 16, 0, 1}, {R, 64, 16, 0, 1}, {G, 80, 16, 0, 1},
          {B, 96, 16, 0, 1}, {RGB, 64, 48, 0, 3}, RAW_WORDS_4},
         {{RANGE, T::UINT32}, {SIGNAL, T::UINT16}, {REFLECTIVITY, T::UINT8}, {NEAR_IR, T::UINT16},
          {RGB, T::FLOAT16}, {FLAGS, T::UINT8}}},
        {P::RNG19_RFL8_SIG16_NIR16_RGB16_DUAL, 20,
         {DUAL19_HEAD, {NEAR_IR, 96, 16, 0, 1}, {R, 112, 16, 0, 1}, {G, 128, 16, 0, 1},
          {B, 144, 16, 0, 1}, {RGB, 112, 48, 0, 3}, RAW_WORDS_5},
         {{RANGE, T::UINT32}, {RANGE2, T::UINT32}, {SIGNAL, T::UINT16}, {SIGNAL2, T::UINT16},
          {REFLECTIVITY, T::UINT8}, {REFLECTIVITY2, T::UINT8}, {NEAR_IR, T::UINT16},
          {RGB, T::FLOAT16}, {FLAGS, T::UINT8}, {FLAGS2, T::UINT8}}},
        {P::OFF, 0, {}, {}},
    };
    return tbl;
}

struct Registered {
    int profile;
    std::string name;
    size_t chan_bytes;
    std::vector<std::pair<std::string, FieldDecodeInfo>> fields;
    std::vector<std::pair<std::string, ChanFieldType>> planes;
};

std::mutex& registry_mutex() {
    static std::mutex m;
    return m;
}

std::vector<Registered>& registry() {
    static std::vector<Registered> reg = [] {
        static const std::pair<UDPProfileLidar, const char*> names[] = {
            {UDPProfileLidar::LEGACY, "LEGACY"},
            {UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_DUAL, "RNG19_RFL8_SIG16_NIR16_DUAL"},
            {UDPProfileLidar::RNG19_RFL8_SIG16_NIR16, "RNG19_RFL8_SIG16_NIR16"},
            {UDPProfileLidar::RNG15_RFL8_NIR8, "RNG15_RFL8_NIR8"},
            {UDPProfileLidar::FIVE_WORD_PIXEL, "FIVE_WORD_PIXEL"},
            {UDPProfileLidar::FUSA_RNG15_RFL8_NIR8_DUAL, "FUSA_RNG15_RFL8_NIR8_DUAL"},
            {UDPProfileLidar::RNG15_RFL8_NIR8_DUAL, "RNG15_RFL8_NIR8_DUAL"},
            {UDPProfileLidar::RNG15_RFL8_NIR8_ZONE16, "RNG15_RFL8_NIR8_ZONE16"},
            {UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_ZONE16, "RNG19_RFL8_SIG16_NIR16_ZONE16"},
            {UDPProfileLidar::RNG15_RFL8_WIN8, "RNG15_RFL8_WIN8"},
            {UDPProfileLidar::RNG19_RFL8_SIG16_ZONE16_DUAL, "RNG19_RFL8_SIG16_ZONE16_DUAL"},
            {UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_RGB16, "RNG19_RFL8_SIG16_NIR16_RGB16"},
            {UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_RGB16_DUAL,
             "RNG19_RFL8_SIG16_NIR16_RGB16_DUAL"},
            {UDPProfileLidar::OFF, "OFF"},
        };
        std::vector<Registered> r;
        const auto& tbl = builtin_layouts();
        const Layout* dual_lb = nullptr;
        for (const auto& l : tbl)
            if (l.profile == UDPProfileLidar::RNG15_RFL8_NIR8_DUAL) dual_lb = &l;
        for (const auto& l0 : tbl) {
            const Layout& l = (l0.profile == UDPProfileLidar::FUSA_RNG15_RFL8_NIR8_DUAL) ? *dual_lb : l0;
            Registered e;
            e.profile = static_cast<int>(l0.profile);
            for (const auto& n : names)
                if (n.first == l0.profile) e.name = n.second;
            e.chan_bytes = l.chan_bytes;
            for (const auto& b : l.bits)
                e.fields.emplace_back(b.name, field_info(b.start, b.size, b.up, 0, b.n));
            for (const auto& p : l.planes) e.planes.emplace_back(p.first, p.second);
            r.push_back(std::move(e));
        }
        return r;
    }();
    return reg;
}

const Registered& lookup(UDPProfileLidar profile) {
    std::lock_guard<std::mutex> lk(registry_mutex());
    for (const auto& e : registry())
        if (e.profile == static_cast<int>(profile) && profile != UDPProfileLidar::UNKNOWN) return e;
    throw std::invalid_argument("Unknown lidar udp profile");
}

}  // namespace

std::string to_string(UDPProfileLidar profile) {
    std::lock_guard<std::mutex> lk(registry_mutex());
    for (const auto& e : registry())
        if (e.profile == static_cast<int>(profile)) return e.name;
    return "UNKNOWN";
}

nonstd::optional<UDPProfileLidar> udp_profile_lidar_of_string(const std::string& s) {
    std::lock_guard<std::mutex> lk(registry_mutex());
    for (const auto& e : registry())
        if (e.name == s) return static_cast<UDPProfileLidar>(e.profile);
    return nonstd::nullopt;
}

// data_format.cpp:183-230 of the reference: the IMU profile and header type name tables
std::string to_string(UDPProfileIMU profile) {
    switch (profile) {
        case UDPProfileIMU::LEGACY: return "LEGACY";
        case UDPProfileIMU::ACCEL32_GYRO32_NMEA: return "ACCEL32_GYRO32_NMEA";
        case UDPProfileIMU::OFF: return "OFF";
    }
    return "UNKNOWN";
}
nonstd::optional<UDPProfileIMU> udp_profile_imu_of_string(const std::string& s) {
    for (auto p : {UDPProfileIMU::LEGACY, UDPProfileIMU::ACCEL32_GYRO32_NMEA, UDPProfileIMU::OFF})
        if (to_string(p) == s) return p;
    return nonstd::nullopt;
}
std::string to_string(HeaderType profile) {
    switch (profile) {
        case HeaderType::STANDARD: return "STANDARD";
        case HeaderType::FUSA: return "FUSA";
    }
    return "UNKNOWN";
}
nonstd::optional<HeaderType> udp_profile_type_of_string(const std::string& s) {
    for (auto p : {HeaderType::STANDARD, HeaderType::FUSA})
        if (to_string(p) == s) return p;
    return nonstd::nullopt;
}

void add_custom_profile(int profile_nr, const std::string& name,
                        const std::vector<std::pair<std::string, FieldDecodeInfo>>& fields,
                        size_t chan_data_size) {
    if (profile_nr == 0) throw std::invalid_argument("profile_nr of 0 are not allowed");
    std::lock_guard<std::mutex> lk(registry_mutex());
    auto& reg = registry();
    for (const auto& e : reg) {
        if (e.profile == profile_nr)
            throw std::invalid_argument("Lidar profile of given number already exists");
        if (e.name == name)
            throw std::invalid_argument("Lidar profile of given name already exists");
    }
    if (reg.size() >= static_cast<size_t>(MAX_NUM_PROFILES))
        throw std::runtime_error("Limit of lidar profiles has been reached");
    Registered e;
    e.profile = profile_nr;
    e.name = name;
    e.chan_bytes = chan_data_size;
    for (const auto& kv : fields) {
        FieldDecodeInfo f = kv.second;
        if (f.mask == 0) f.mask = field_type_mask(f.ty_tag);  // "whole type" shorthand
        if (f.num_elements <= 0) f.num_elements = 1;
        e.fields.emplace_back(kv.first, f);
        e.planes.emplace_back(kv.first, kv.second.ty_tag);
    }
    reg.push_back(std::move(e));
}

UDPProfileLidar add_custom_profile(
    const std::string& name, const std::vector<std::pair<std::string, FieldDecodeInfo>>& fields,
    size_t chan_data_size) {
    int next = 0;
    {
        std::lock_guard<std::mutex> lk(registry_mutex());
        for (const auto& e : registry())
            if (e.profile != static_cast<int>(UDPProfileLidar::OFF)) next = std::max(next, e.profile);
    }
    add_custom_profile(next + 1, name, fields, chan_data_size);
    return static_cast<UDPProfileLidar>(next + 1);
}

namespace impl {
std::vector<std::pair<std::string, ChanFieldType>> default_planes(UDPProfileLidar profile) {
    return lookup(profile).planes;
}

// The profile table as the reference's tests look at it (tests/packet_format_test.cpp:27-36 declares this function and
// the entry layout itself; parsing.cpp:365-420 holds the reference's table): MAX_NUM_PROFILES slots, the registered
// profiles first, the rest zero.  The field pointers stay valid while no profile is added.
struct ProfileEntry {
    const std::pair<std::string, FieldDecodeInfo>* fields;
    size_t n_fields;
    size_t chan_data_size;
};
std::array<std::pair<UDPProfileLidar, ProfileEntry>, MAX_NUM_PROFILES> get_profiles() {
    std::array<std::pair<UDPProfileLidar, ProfileEntry>, MAX_NUM_PROFILES> out{};
    std::lock_guard<std::mutex> lk(registry_mutex());
    size_t i = 0;
    for (const auto& e : registry()) {
        if (i == out.size()) break;
        out[i++] = {static_cast<UDPProfileLidar>(e.profile), ProfileEntry{e.fields.data(), e.fields.size(), e.chan_bytes}};
    }
    return out;
}
}  // namespace impl

// ---------------------------------------------------------------------------------------
// PacketFormat
// ---------------------------------------------------------------------------------------
struct PacketFormat::Impl {
    size_t packet_header_size{}, col_header_size{}, channel_data_size{}, col_footer_size{},
        packet_footer_size{}, col_size{}, lidar_packet_size{}, imu_packet_size{}, zone_packet_size{};
    uint32_t max_frame_id{};
    std::map<std::string, FieldDecodeInfo> fields;
    FieldDecodeInfo packet_type, frame_id, init_id, prod_sn, alert_flags, countdown_thermal,
        countdown_shot, thermal, shot, col_status, col_timestamp, col_measurement_id;

    explicit Impl(const DataFormat& fmt) {
        const bool legacy = fmt.udp_profile_lidar == UDPProfileLidar::LEGACY;
        const bool fusa = fmt.header_type == HeaderType::FUSA && !legacy;
        const Registered& e = lookup(fmt.udp_profile_lidar);
        packet_header_size = legacy ? 0 : 32;
        col_header_size = legacy ? 16 : 12;
        channel_data_size = e.chan_bytes;
        col_footer_size = legacy ? 4 : 0;
        packet_footer_size = legacy ? 0 : 32;
        col_size = col_header_size + fmt.pixels_per_column * channel_data_size + col_footer_size;
        lidar_packet_size = packet_header_size + fmt.columns_per_packet * col_size + packet_footer_size;
        if (lidar_packet_size > 65535)
            throw std::invalid_argument("lidar_packet_size cannot exceed 65535");
        // parsing.cpp:540-596: legacy IMU packets are 48 bytes; the newer profile frames a 100-byte NMEA block and 36-byte
        // measurements between header and footer; a zone packet carries 8 + 32 bytes and sixteen 36-byte zone records
        // (any other imu profile -- OFF -- leaves the size at 0: no datagram is taken for an IMU packet)
        imu_packet_size = fmt.udp_profile_imu == UDPProfileIMU::LEGACY ? 48
                          : fmt.udp_profile_imu == UDPProfileIMU::ACCEL32_GYRO32_NMEA
                              ? packet_header_size + 100 + fmt.imu_measurements_per_packet * 36 + packet_footer_size
                              : 0;
        zone_packet_size = packet_header_size + 8 + 32 + 36 * 16 + packet_footer_size;
        for (const auto& kv : e.fields) fields.emplace(kv.first, kv.second);
        max_frame_id = fmt.max_frame_id();

        const FieldDecodeInfo absent = field_info(0, 0);
        packet_type = init_id = prod_sn = alert_flags = countdown_thermal = countdown_shot =
            thermal = shot = absent;
        if (legacy) {
            frame_id = field_info(80, 16);  // lives in the first column header
            // status word trails the column; anchor the 8-byte access at the column end
            const size_t status_bit = 8 * (col_size - col_footer_size);
            col_status = field_info(status_bit, 32, 0, (status_bit + 32) / 8);
        } else {
            if (fusa) {
                packet_type = field_info(0, 8);
                init_id = field_info(8, 24);
                frame_id = field_info(32, 32);
                alert_flags = field_info(64, 8);
                prod_sn = field_info(88, 40);
            } else {
                packet_type = field_info(0, 16);
                frame_id = field_info(16, 16);
                init_id = field_info(32, 24);
                prod_sn = field_info(56, 40);
                alert_flags = field_info(96, 8);
            }
            countdown_thermal = field_info(128, 8);
            countdown_shot = field_info(136, 8);
            thermal = field_info(144, 4);
            shot = field_info(152, 4);
            col_status = field_info(80, 16);
        }
        col_timestamp = field_info(0, 64);
        col_measurement_id = field_info(64, 16);
    }
};

PacketFormat::PacketFormat(const DataFormat& format)
    : impl_{std::make_shared<Impl>(format)},
      udp_profile_lidar{format.udp_profile_lidar},
      udp_profile_imu{format.udp_profile_imu},
      header_type{format.header_type},
      lidar_packet_size{impl_->lidar_packet_size},
      imu_packet_size{impl_->imu_packet_size},
      zone_packet_size{impl_->zone_packet_size},
      columns_per_packet{format.columns_per_packet},
      pixels_per_column{format.pixels_per_column},
      packet_header_size{impl_->packet_header_size},
      col_header_size{impl_->col_header_size},
      col_footer_size{impl_->col_footer_size},
      col_size{impl_->col_size},
      packet_footer_size{impl_->packet_footer_size},
      max_frame_id{impl_->max_frame_id} {
    for (const auto& kv : impl_->fields)
        field_types_.emplace_back(kv.first, std::make_pair(kv.second.ty_tag, kv.second.num_elements));
}

PacketFormat::PacketFormat(const SensorInfo& info) : PacketFormat(info.format) {}

uint16_t PacketFormat::packet_type(const uint8_t* b) const { return impl_->packet_type.get<uint16_t>(b); }
uint32_t PacketFormat::frame_id(const uint8_t* b) const { return impl_->frame_id.get<uint32_t>(b); }
uint32_t PacketFormat::init_id(const uint8_t* b) const { return impl_->init_id.get<uint32_t>(b); }
uint64_t PacketFormat::prod_sn(const uint8_t* b) const { return impl_->prod_sn.get<uint64_t>(b); }
uint8_t PacketFormat::alert_flags(const uint8_t* b) const { return impl_->alert_flags.get<uint8_t>(b); }
uint16_t PacketFormat::countdown_thermal_shutdown(const uint8_t* b) const {
    return impl_->countdown_thermal.get<uint16_t>(b);
}
uint16_t PacketFormat::countdown_shot_limiting(const uint8_t* b) const {
    return impl_->countdown_shot.get<uint16_t>(b);
}
ThermalShutdownStatus PacketFormat::thermal_shutdown(const uint8_t* b) const {
    return static_cast<ThermalShutdownStatus>(impl_->thermal.get<uint8_t>(b));
}
ShotLimitingStatus PacketFormat::shot_limiting(const uint8_t* b) const {
    return static_cast<ShotLimitingStatus>(impl_->shot.get<uint8_t>(b));
}

ChanFieldType PacketFormat::field_type(const std::string& f) const {
    auto it = impl_->fields.find(f);
    return it == impl_->fields.end() ? ChanFieldType::VOID : it->second.ty_tag;
}
PacketFormat::FieldIter PacketFormat::begin() const { return field_types_.cbegin(); }
PacketFormat::FieldIter PacketFormat::end() const { return field_types_.cend(); }

uint8_t* PacketFormat::footer(uint8_t* lidar_buf) const {
    if (impl_->packet_footer_size == 0) return nullptr;
    return lidar_buf + impl_->packet_header_size + columns_per_packet * impl_->col_size;
}
const uint8_t* PacketFormat::footer(const uint8_t* lidar_buf) const {
    return footer(const_cast<uint8_t*>(lidar_buf));
}
uint8_t* PacketFormat::nth_col(size_t n, uint8_t* lidar_buf) const {
    return lidar_buf + impl_->packet_header_size + n * impl_->col_size;
}
const uint8_t* PacketFormat::nth_col(size_t n, const uint8_t* lidar_buf) const {
    return nth_col(n, const_cast<uint8_t*>(lidar_buf));
}
uint32_t PacketFormat::col_status(const uint8_t* c) const { return impl_->col_status.get<uint32_t>(c); }
uint64_t PacketFormat::col_timestamp(const uint8_t* c) const { return impl_->col_timestamp.get<uint64_t>(c); }
uint16_t PacketFormat::col_measurement_id(const uint8_t* c) const {
    return impl_->col_measurement_id.get<uint16_t>(c);
}
uint32_t PacketFormat::col_encoder(const uint8_t* c) const {
    uint32_t v = 0;
    if (udp_profile_lidar == UDPProfileLidar::LEGACY) std::memcpy(&v, c + 12, sizeof v);
    return v;
}
uint16_t PacketFormat::col_frame_id(const uint8_t* c) const {
    uint16_t v = 0;
    if (udp_profile_lidar == UDPProfileLidar::LEGACY) std::memcpy(&v, c + 10, sizeof v);
    return v;
}
uint8_t* PacketFormat::nth_px(size_t n, uint8_t* col_buf) const {
    return col_buf + impl_->col_header_size + n * impl_->channel_data_size;
}
const uint8_t* PacketFormat::nth_px(size_t n, const uint8_t* col_buf) const {
    return nth_px(n, const_cast<uint8_t*>(col_buf));
}

int PacketFormat::block_parsable() const {
    for (int dim : {16, 8, 4})
        if (pixels_per_column % dim == 0 && columns_per_packet % dim == 0) return dim;
    return 0;
}

uint64_t PacketFormat::field_value_mask(const std::string& f) const {
    return impl::get_value_mask(impl_->fields.at(f));
}
int PacketFormat::field_bitness(const std::string& f) const {
    return impl::get_bitness(impl_->fields.at(f));
}
const FieldDecodeInfo& PacketFormat::field_decode_info(const std::string& f) const {
    return impl_->fields.at(f);
}
size_t PacketFormat::channel_data_size() const { return impl_->channel_data_size; }

void PacketFormat::set_col_status(uint8_t* c, uint32_t v) const { impl_->col_status.set(c, v); }
void PacketFormat::set_col_timestamp(uint8_t* c, uint64_t v) const { impl_->col_timestamp.set(c, v); }
void PacketFormat::set_col_measurement_id(uint8_t* c, uint16_t v) const {
    impl_->col_measurement_id.set(c, v);
}
void PacketFormat::set_frame_id(uint8_t* b, uint32_t v) const { impl_->frame_id.set(b, v); }
void PacketFormat::set_init_id(uint8_t* b, uint32_t v) const { impl_->init_id.set(b, v); }
void PacketFormat::set_packet_type(uint8_t* b, uint16_t v) const { impl_->packet_type.set(b, v); }
void PacketFormat::set_prod_sn(uint8_t* b, uint64_t v) const { impl_->prod_sn.set(b, v); }
void PacketFormat::set_alert_flags(uint8_t* b, uint8_t v) const { impl_->alert_flags.set(b, v); }
void PacketFormat::set_shutdown(uint8_t* b, uint8_t v) const { impl_->thermal.set(b, v); }
void PacketFormat::set_shot_limiting(uint8_t* b, uint8_t v) const { impl_->shot.set(b, v); }
void PacketFormat::set_shutdown_countdown(uint8_t* b, uint8_t v) const { impl_->countdown_thermal.set(b, v); }
void PacketFormat::set_shot_limiting_countdown(uint8_t* b, uint8_t v) const {
    impl_->countdown_shot.set(b, v);
}

template <typename T>
void PacketFormat::set_block(const T* data, int cols, const std::string& f, uint8_t* lidar_buf) const {
    if (columns_per_packet > 32) throw std::runtime_error("Recompile set_block_impl with larger N");
    const FieldDecodeInfo info = impl_->fields.at(f);
    const uint16_t m_id0 = col_measurement_id(nth_col(0, lidar_buf));
    for (uint32_t x = 0; x < columns_per_packet; ++x) {
        uint8_t* col = nth_col(x, lidar_buf);
        if (!(col_status(col) & 0x01)) continue;
        for (uint32_t px = 0; px < pixels_per_column; ++px)
            info.set(nth_px(px, col), data[static_cast<ptrdiff_t>(cols) * px + m_id0 + x]);
    }
}
#define OUSTER_INST_SET_BLOCK(T) \
    template void PacketFormat::set_block<T>(const T*, int, const std::string&, uint8_t*) const;
OUSTER_INST_SET_BLOCK(uint8_t)
OUSTER_INST_SET_BLOCK(uint16_t)
OUSTER_INST_SET_BLOCK(uint32_t)
OUSTER_INST_SET_BLOCK(uint64_t)
OUSTER_INST_SET_BLOCK(int8_t)
OUSTER_INST_SET_BLOCK(int16_t)
OUSTER_INST_SET_BLOCK(int32_t)
OUSTER_INST_SET_BLOCK(int64_t)
OUSTER_INST_SET_BLOCK(float)
OUSTER_INST_SET_BLOCK(double)
OUSTER_INST_SET_BLOCK(impl::float3x16_t)
#undef OUSTER_INST_SET_BLOCK

namespace {
// CRC-64/XZ (ECMA-182 polynomial, reflected), byte-table driven
uint64_t crc64(const uint8_t* p, size_t n) {
    static const std::array<uint64_t, 256> table = [] {
        std::array<uint64_t, 256> t{};
        for (uint64_t i = 0; i < 256; ++i) {
            uint64_t r = i;
            for (int b = 0; b < 8; ++b) r = (r & 1) ? (r >> 1) ^ 0xC96C5795D7870F42ull : (r >> 1);
            t[i] = r;
        }
        return t;
    }();
    uint64_t crc = ~uint64_t{0};
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
}  // namespace

bool PacketFormat::crc(const uint8_t* buffer, size_t buffer_size, uint64_t& out) const {
    if (udp_profile_lidar == UDPProfileLidar::LEGACY ||
        udp_profile_lidar == UDPProfileLidar::FUSA_RNG15_RFL8_NIR8_DUAL ||
        header_type == HeaderType::FUSA)
        return false;
    std::memcpy(&out, buffer + buffer_size - 8, 8);
    return true;
}
uint64_t PacketFormat::calculate_crc(const uint8_t* buffer, size_t buffer_size) const {
    return crc64(buffer, buffer_size - 8);
}

int PacketFormat::frame_id_difference(uint32_t current, uint32_t other) const {
    const int64_t span = static_cast<int64_t>(max_frame_id) + 1, half = max_frame_id >> 1;
    int64_t d = static_cast<int64_t>(other) - current;
    if (d < -half) d += span;
    else if (d > half) d -= span;
    return static_cast<int>(d);
}

static ouster_hip_bits to_bits(const FieldDecodeInfo& f) {
    ouster_hip_bits b;
    b.mask = f.mask;
    b.offset = static_cast<uint32_t>(f.offset);
    b.shift = f.shift;
    return b;
}

void PacketFormat::fill_hip_desc(uint32_t columns_per_frame,
                                 const std::vector<std::pair<std::string, uint32_t>>& fields,
                                 const std::vector<bool>& f16_nan,
                                 ouster_hip_format_desc& d) const {
    std::memset(&d, 0, sizeof d);
    d.pixels_per_column = pixels_per_column;
    d.columns_per_packet = columns_per_packet;
    d.columns_per_frame = columns_per_frame;
    d.packet_header_size = static_cast<uint32_t>(packet_header_size);
    d.col_header_size = static_cast<uint32_t>(col_header_size);
    d.channel_data_size = static_cast<uint32_t>(impl_->channel_data_size);
    d.col_footer_size = static_cast<uint32_t>(col_footer_size);
    d.packet_footer_size = static_cast<uint32_t>(packet_footer_size);
    d.col_size = static_cast<uint32_t>(col_size);
    d.lidar_packet_size = static_cast<uint32_t>(lidar_packet_size);
    d.col_timestamp = to_bits(impl_->col_timestamp);
    d.col_measurement_id = to_bits(impl_->col_measurement_id);
    d.col_status = to_bits(impl_->col_status);
    d.frame_id = to_bits(impl_->frame_id);
    d.alert_flags = to_bits(impl_->alert_flags);
    d.thermal_shutdown = to_bits(impl_->thermal);
    d.shot_limiting = to_bits(impl_->shot);
    d.countdown_thermal_shutdown = to_bits(impl_->countdown_thermal);
    d.countdown_shot_limiting = to_bits(impl_->countdown_shot);
    if (fields.size() > OUSTER_HIP_MAX_FIELDS) throw std::invalid_argument("too many fields");
    d.n_fields = static_cast<uint32_t>(fields.size());
    for (size_t i = 0; i < fields.size(); ++i) {
        const FieldDecodeInfo& f = impl_->fields.at(fields[i].first);
        if (fields[i].second < field_type_size(f.ty_tag) * static_cast<size_t>(f.num_elements))
            throw std::invalid_argument("Dest type too small for specified field");
        d.fields[i].bits = to_bits(f);
        d.fields[i].dst_elem_size = fields[i].second;
        d.fields[i].f16_nan_fill = (i < f16_nan.size() && f16_nan[i]) ? 1u : 0u;
    }
}

// get_format cache (parsing.cpp:981-995)
namespace {
bool format_less(const DataFormat& a, const DataFormat& b) {
    return std::tie(a.pixels_per_column, a.columns_per_packet, a.columns_per_frame,
                    a.imu_measurements_per_packet, a.pixel_shift_by_row, a.column_window,
                    a.udp_profile_lidar, a.udp_profile_imu, a.header_type) <
           std::tie(b.pixels_per_column, b.columns_per_packet, b.columns_per_frame,
                    b.imu_measurements_per_packet, b.pixel_shift_by_row, b.column_window,
                    b.udp_profile_lidar, b.udp_profile_imu, b.header_type);
}
struct FormatLess {
    bool operator()(const DataFormat& a, const DataFormat& b) const { return format_less(a, b); }
};
}  // namespace

const PacketFormat& get_format(const DataFormat& format) {
    static std::map<DataFormat, std::unique_ptr<PacketFormat>, FormatLess> cache;
    static std::mutex mx;
    std::lock_guard<std::mutex> lk(mx);
    auto it = cache.find(format);
    if (it == cache.end()) it = cache.emplace(format, std::make_unique<PacketFormat>(format)).first;
    return *it->second;
}
const PacketFormat& get_format(const SensorInfo& info) { return get_format(info.format); }

// ---------------------------------------------------------------------------------------
// SensorInfo helpers
// ---------------------------------------------------------------------------------------
const mat4d DEFAULT_LIDAR_TO_SENSOR = [] {
    mat4d m = mat4d::Identity();
    m(0, 0) = -1;
    m(1, 1) = -1;
    m(2, 3) = 36.18;
    return m;
}();

double default_lidar_origin_to_beam_origin(const std::string& prod_line) {
    if (prod_line.rfind("OS-0-", 0) == 0) return 27.67;
    if (prod_line.rfind("OS-1-", 0) == 0) return 15.806;
    if (prod_line.rfind("OS-2-", 0) == 0) return 13.762;
    return 12.163;  // gen 1
}

mat4d default_beam_to_lidar_transform(const std::string& prod_line) {
    mat4d m = mat4d::Identity();
    m(0, 3) = default_lidar_origin_to_beam_origin(prod_line);
    return m;
}

bool operator==(const SensorInfo& a, const SensorInfo& b) {
    auto same_opt = [](const auto& x, const auto& y) { return bool(x) == bool(y) && (!x || *x == *y); };
    return a.sn == b.sn && a.fw_rev == b.fw_rev && a.image_rev == b.image_rev && a.prod_line == b.prod_line &&
           a.format == b.format && a.beam_azimuth_angles == b.beam_azimuth_angles &&
           a.beam_altitude_angles == b.beam_altitude_angles &&
           a.lidar_origin_to_beam_origin_mm == b.lidar_origin_to_beam_origin_mm &&
           a.beam_to_lidar_transform == b.beam_to_lidar_transform && a.imu_to_sensor_transform == b.imu_to_sensor_transform &&
           a.lidar_to_sensor_transform == b.lidar_to_sensor_transform && a.sensor_to_body == b.sensor_to_body &&
           a.init_id == b.init_id && same_opt(a.config.lidar_mode, b.config.lidar_mode) &&
           same_opt(a.config.udp_profile_lidar, b.config.udp_profile_lidar) &&
           same_opt(a.config.udp_profile_imu, b.config.udp_profile_imu) && same_opt(a.config.udp_port_lidar, b.config.udp_port_lidar) &&
           same_opt(a.config.udp_port_imu, b.config.udp_port_imu) && same_opt(a.config.udp_port_zm, b.config.udp_port_zm);
}

Version SensorInfo::get_version() const {
    // sensor_info.cpp:391-393 parses image_rev ("ousteros-image-prod-aries-v2.3.0+2022...", "v3.2.0", "3.2.1"); a SensorInfo
    // filled by hand with only fw_rev set is read from that
    auto parse = [](const std::string& text, Version& out) {
        for (const char* s = text.c_str(); *s; ++s) {
            if (*s < '0' || *s > '9') continue;
            unsigned a = 0, b = 0, c = 0;
            if (std::sscanf(s, "%u.%u.%u", &a, &b, &c) == 3) {
                out = Version(a, b, c);
                return true;
            }
            while (s[1] >= '0' && s[1] <= '9') ++s;   // not a dotted triple: skip this run of digits
        }
        return false;
    };
    Version v;
    if (!parse(image_rev, v)) parse(fw_rev, v);
    return v;
}

int SensorInfo::num_returns() const {
    switch (format.udp_profile_lidar) {
        case UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_DUAL:
        case UDPProfileLidar::FUSA_RNG15_RFL8_NIR8_DUAL:
        case UDPProfileLidar::RNG15_RFL8_NIR8_DUAL:
        case UDPProfileLidar::RNG19_RFL8_SIG16_ZONE16_DUAL:
        case UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_RGB16_DUAL:
        case UDPProfileLidar::FIVE_WORD_PIXEL:
            return 2;
        default:
            return 1;
    }
}

// packet.cpp:28-73 of the reference, lidar packets (an IMU / zone sized buffer has no counterpart here: every buffer that
// is not explicitly of another kind is held against the lidar packet size)
PacketValidationFailure validate_packet(const SensorInfo& info, const PacketFormat& format, const uint8_t* buf,
                                        uint64_t buf_size, PacketType type) {
    if (type == PacketType::Unknown)
        type = buf_size == format.imu_packet_size    ? PacketType::Imu
               : buf_size == format.zone_packet_size ? PacketType::Zone
                                                     : PacketType::Lidar;
    const size_t want = type == PacketType::Imu ? format.imu_packet_size
                        : type == PacketType::Zone ? format.zone_packet_size
                                                   : format.lidar_packet_size;
    if (buf_size != want) return PacketValidationFailure::PACKET_SIZE;
    if (type == PacketType::Imu && format.udp_profile_imu == UDPProfileIMU::LEGACY) return PacketValidationFailure::NONE;
    const uint32_t init_id = format.init_id(buf);
    if (info.init_id != 0 && init_id != 0 && init_id != info.init_id) return PacketValidationFailure::ID;
    if (info.sn != 0) {
        const uint64_t sn = format.prod_sn(buf);
        if (sn != 0 && sn != info.sn) return PacketValidationFailure::ID;
    }
    return PacketValidationFailure::NONE;
}

}  // namespace core
}  // namespace sdk
}  // namespace ouster
