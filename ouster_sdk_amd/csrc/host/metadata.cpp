// metadata.cpp -- sensor metadata JSON -> SensorInfo: the data-format and calibration subset the hot path needs.
//
// Both generations of the file are read, as the reference does (ouster_core/src/metadata.cpp:831-842 rewrites the
// legacy flat keys into the nested form, :482-592 parses the data format, :725-771 the intrinsics):
//   nested: beam_intrinsics.*, lidar_data_format.*, lidar_intrinsics.lidar_to_sensor_transform, sensor_info.*,
//           config_params.{lidar_mode, udp_profile_lidar}, optional 'ouster-sdk'.extrinsic;
//   flat:   beam_altitude_angles, beam_azimuth_angles, data_format.*, lidar_to_sensor_transform,
//           lidar_origin_to_beam_origin_mm, prod_line, prod_sn, initialization_id, build_rev, lidar_mode.
// Defaults follow default_data_format(mode) (data_format.cpp:79-125), pixel_shift_by_row is zero-padded / cut to H
// (metadata.cpp:530-534), a FUSA profile without header_type implies FUSA headers (:545-555), a missing
// beam_to_lidar_transform is identity with (0, 3) = lidar_origin_to_beam_origin_mm (sensor_info.cpp:89-105).
// Everything else of the reference's metadata handling (sensor config, calibration status, zone sets, validation
// reports) is out of scope.  The JSON reader below is a small recursive-descent parser (the reference uses jsoncons).
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "ouster/core/types.h"

namespace ouster {
namespace sdk {
namespace core {

namespace {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;   // string value, or the literal text of a number (64-bit integers survive)
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const std::string& key) const {
        if (kind != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(const std::string& key) const {
        static const Json none;
        const Json* j = find(key);
        return j ? *j : none;
    }
    bool is_null() const { return kind == Null; }
    uint64_t as_u64() const {
        if (kind == Number) return str.find_first_of(".eE") == std::string::npos ? std::strtoull(str.c_str(), nullptr, 10)
                                                                                 : static_cast<uint64_t>(num);
        if (kind == String) return std::strtoull(str.c_str(), nullptr, 10);
        return 0;
    }
    int64_t as_i64() const { return kind == Number ? static_cast<int64_t>(num) : static_cast<int64_t>(as_u64()); }
    double as_double() const { return kind == Number ? num : (kind == String ? std::strtod(str.c_str(), nullptr) : 0.0); }
    std::string as_string() const { return kind == String ? str : std::string{}; }
    /** numbers of a (possibly nested) array, flattened */
    void flatten(std::vector<double>& out) const {
        if (kind == Number) out.push_back(num);
        else if (kind == Array)
            for (const auto& e : arr) e.flatten(out);
    }
};

class Parser {
   public:
    explicit Parser(const std::string& s) : s_(s) {}
    Json parse() {
        Json j = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return j;
    }

   private:
    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("metadata JSON: ") + what + " at offset " + std::to_string(i_));
    }
    void ws() {
        while (i_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[i_]))) ++i_;
    }
    bool eat(char c) {
        ws();
        if (i_ < s_.size() && s_[i_] == c) {
            ++i_;
            return true;
        }
        return false;
    }
    struct Depth {   // metadata files nest four or five levels; a crafted one must not be able to exhaust the stack
        int& d;
        explicit Depth(int& depth) : d(depth) { ++d; }
        ~Depth() { --d; }
    };
    Json value() {
        Depth guard(depth_);
        if (depth_ > 64) fail("nesting too deep");
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        Json j;
        if (c == '{') {
            ++i_;
            j.kind = Json::Object;
            if (eat('}')) return j;
            do {
                ws();
                if (i_ >= s_.size() || s_[i_] != '"') fail("expected a key");
                std::string k = string();
                if (!eat(':')) fail("expected ':'");
                j.obj.emplace_back(std::move(k), value());
            } while (eat(','));
            if (!eat('}')) fail("expected '}'");
        } else if (c == '[') {
            ++i_;
            j.kind = Json::Array;
            if (eat(']')) return j;
            do j.arr.push_back(value());
            while (eat(','));
            if (!eat(']')) fail("expected ']'");
        } else if (c == '"') {
            j.kind = Json::String;
            j.str = string();
        } else if (s_.compare(i_, 4, "true") == 0) {
            i_ += 4;
            j.kind = Json::Bool;
            j.b = true;
        } else if (s_.compare(i_, 5, "false") == 0) {
            i_ += 5;
            j.kind = Json::Bool;
        } else if (s_.compare(i_, 4, "null") == 0) {
            i_ += 4;
        } else {
            const size_t b = i_;
            while (i_ < s_.size() && s_[i_] != '\0' &&
                   (std::isdigit(static_cast<unsigned char>(s_[i_])) || std::strchr("+-.eE", s_[i_])))
                ++i_;
            if (b == i_) fail("unexpected character");
            j.kind = Json::Number;
            j.str = s_.substr(b, i_ - b);
            j.num = std::strtod(j.str.c_str(), nullptr);
        }
        return j;
    }
    std::string string() {
        std::string out;
        ++i_;   // opening quote
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\') {
                if (i_ >= s_.size()) fail("unterminated escape");
                const char e = s_[i_++];
                switch (e) {
                    case 'n': c = '\n'; break;
                    case 't': c = '\t'; break;
                    case 'r': c = '\r'; break;
                    case 'b': c = '\b'; break;
                    case 'f': c = '\f'; break;
                    case 'u': {   // BMP code point -> UTF-8
                        if (i_ + 4 > s_.size()) fail("short \\u escape");
                        const unsigned cp = static_cast<unsigned>(std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16));
                        i_ += 4;
                        if (cp < 0x80) out.push_back(static_cast<char>(cp));
                        else if (cp < 0x800) {
                            out.push_back(static_cast<char>(0xC0 | (cp >> 6)));
                            out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                        } else {
                            out.push_back(static_cast<char>(0xE0 | (cp >> 12)));
                            out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
                            out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                        }
                        continue;
                    }
                    default: c = e;   // \" \\ \/
                }
            }
            out.push_back(c);
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return out;
    }
    const std::string& s_;
    size_t i_ = 0;
    int depth_ = 0;
};

mat4d mat_from(const Json& j, const mat4d& fallback) {
    std::vector<double> v;
    j.flatten(v);
    if (v.size() != 16) return fallback;
    return mat4d::FromRowMajor(v.data());
}

}  // namespace

LidarMode::LidarMode(const std::string& mode) {
    const size_t split = mode.find('x');
    try {
        if (split == std::string::npos) throw std::invalid_argument("");
        const int c = std::stoi(mode.substr(0, split)), f = std::stoi(mode.substr(split + 1));
        if (c < 0 || f < 0) throw std::invalid_argument("");
        columns = static_cast<unsigned int>(c);
        fps = static_cast<unsigned int>(f);
    } catch (const std::invalid_argument&) {
        throw std::invalid_argument("Invalid lidar mode string \"" + mode + "\".");
    } catch (const std::out_of_range&) {
        throw std::invalid_argument("Invalid lidar mode string \"" + mode + "\".");
    }
}
const LidarMode LidarMode::_512x10 = {512, 10};
const LidarMode LidarMode::_512x20 = {512, 20};
const LidarMode LidarMode::_1024x10 = {1024, 10};
const LidarMode LidarMode::_1024x20 = {1024, 20};
const LidarMode LidarMode::_2048x10 = {2048, 10};
const LidarMode LidarMode::_4096x5 = {4096, 5};
std::string to_string(LidarMode mode) { return std::to_string(mode.columns) + "x" + std::to_string(mode.fps); }
nonstd::optional<LidarMode> lidar_mode_of_string(const std::string& s) {
    try {
        return LidarMode(s);
    } catch (const std::invalid_argument&) {
        return {};
    }
}

std::string to_string(ThermalShutdownStatus status) {
    switch (status) {
        case ThermalShutdownStatus::NORMAL: return "NORMAL";
        case ThermalShutdownStatus::IMMINENT: return "IMMINENT";
    }
    return "UNKNOWN";
}
std::string to_string(ShotLimitingStatus status) {
    static const char* const names[] = {"NORMAL", "IMMINENT", "REDUCTION_0_10", "REDUCTION_10_20", "REDUCTION_20_30",
                                        "REDUCTION_30_40", "REDUCTION_40_50", "REDUCTION_50_60", "REDUCTION_60_70",
                                        "REDUCTION_70_75"};
    const unsigned v = static_cast<unsigned>(status);
    return v < sizeof names / sizeof names[0] ? names[v] : "UNKNOWN";
}

SensorInfo::SensorInfo(const std::string& metadata_json) {
    const Json d = Parser(metadata_json).parse();
    if (d.kind != Json::Object) throw std::runtime_error("metadata JSON: not an object");
    const bool nested = d.find("lidar_data_format") || d.find("beam_intrinsics");
    const Json& df = nested ? d.at("lidar_data_format") : d.at("data_format");
    const Json& bi = nested ? d.at("beam_intrinsics") : d;
    const Json& si = nested ? d.at("sensor_info") : d;
    const Json& cfg = nested ? d.at("config_params") : d;
    const Json& l2s = nested ? d.at("lidar_intrinsics").at("lidar_to_sensor_transform") : d.at("lidar_to_sensor_transform");
    const Json& imu2s = nested ? d.at("imu_intrinsics").at("imu_to_sensor_transform") : d.at("imu_to_sensor_transform");

    std::string mode = cfg.at("lidar_mode").as_string();
    if (mode.empty()) mode = d.at("lidar_mode").as_string();
    uint32_t w_mode = 1024;
    uint16_t fps = 10;
    const size_t x = mode.find('x');
    if (x != std::string::npos) {
        w_mode = static_cast<uint32_t>(std::strtoul(mode.substr(0, x).c_str(), nullptr, 10));
        fps = static_cast<uint16_t>(std::strtoul(mode.substr(x + 1).c_str(), nullptr, 10));
    }
    config.lidar_mode = lidar_mode_of_string(mode);
    format = default_data_format(w_mode, fps);
    if (!df.at("pixels_per_column").is_null()) format.pixels_per_column = static_cast<uint32_t>(df.at("pixels_per_column").as_u64());
    if (!df.at("columns_per_packet").is_null()) format.columns_per_packet = static_cast<uint32_t>(df.at("columns_per_packet").as_u64());
    if (!df.at("columns_per_frame").is_null()) format.columns_per_frame = static_cast<uint32_t>(df.at("columns_per_frame").as_u64());
    if (!df.at("fps").is_null()) format.fps = static_cast<uint16_t>(df.at("fps").as_u64());
    if (!df.at("imu_measurements_per_packet").is_null())
        format.imu_measurements_per_packet = static_cast<uint32_t>(df.at("imu_measurements_per_packet").as_u64());
    if (!df.at("imu_packets_per_frame").is_null()) format.imu_packets_per_frame = static_cast<uint32_t>(df.at("imu_packets_per_frame").as_u64());
    const uint32_t h = format.pixels_per_column, w = format.columns_per_frame;
    if (df.at("pixel_shift_by_row").kind == Json::Array) {
        format.pixel_shift_by_row.clear();
        for (const auto& e : df.at("pixel_shift_by_row").arr) format.pixel_shift_by_row.push_back(static_cast<int>(e.as_i64()));
    }
    format.pixel_shift_by_row.resize(h, 0);
    if (df.at("column_window").kind == Json::Array && df.at("column_window").arr.size() == 2)
        format.column_window = {static_cast<int>(df.at("column_window").arr[0].as_i64()),
                                static_cast<int>(df.at("column_window").arr[1].as_i64())};
    else
        format.column_window = {0, static_cast<int>(w) - 1};
    std::string profile = df.at("udp_profile_lidar").as_string();
    if (profile.empty()) profile = cfg.at("udp_profile_lidar").as_string();
    if (!profile.empty()) {
        const auto p = udp_profile_lidar_of_string(profile);
        if (!p) throw std::runtime_error("metadata JSON: unknown udp_profile_lidar " + profile);
        format.udp_profile_lidar = *p;
    }
    if (const auto p = udp_profile_lidar_of_string(cfg.at("udp_profile_lidar").as_string())) config.udp_profile_lidar = *p;
    std::string imu_profile = df.at("udp_profile_imu").as_string();
    if (imu_profile.empty()) imu_profile = cfg.at("udp_profile_imu").as_string();
    if (const auto p = udp_profile_imu_of_string(imu_profile)) format.udp_profile_imu = *p;
    if (const auto p = udp_profile_imu_of_string(cfg.at("udp_profile_imu").as_string())) config.udp_profile_imu = *p;
    if (!cfg.at("udp_port_lidar").is_null()) config.udp_port_lidar = static_cast<int>(cfg.at("udp_port_lidar").as_i64());
    if (!cfg.at("udp_port_imu").is_null()) config.udp_port_imu = static_cast<int>(cfg.at("udp_port_imu").as_i64());
    if (!cfg.at("udp_port_zm").is_null()) config.udp_port_zm = static_cast<int>(cfg.at("udp_port_zm").as_i64());
    std::string header = df.at("header_type").as_string();
    if (header.empty()) header = cfg.at("header_type").as_string();
    if (const auto t = udp_profile_type_of_string(header)) format.header_type = *t;
    else format.header_type = format.udp_profile_lidar == UDPProfileLidar::FUSA_RNG15_RFL8_NIR8_DUAL ? HeaderType::FUSA : HeaderType::STANDARD;

    bi.at("beam_altitude_angles").flatten(beam_altitude_angles);
    bi.at("beam_azimuth_angles").flatten(beam_azimuth_angles);
    prod_line = si.at("prod_line").as_string();
    image_rev = si.at("image_rev").as_string();
    fw_rev = si.at("build_rev").as_string();
    if (fw_rev.empty()) fw_rev = image_rev;
    std::vector<double> b2l;
    bi.at("beam_to_lidar_transform").flatten(b2l);
    const Json& origin = bi.at("lidar_origin_to_beam_origin_mm");
    if (b2l.size() == 16) {
        beam_to_lidar_transform = mat4d::FromRowMajor(b2l.data());
        lidar_origin_to_beam_origin_mm = origin.is_null() ? beam_to_lidar_transform(0, 3) : origin.as_double();
    } else {
        lidar_origin_to_beam_origin_mm = origin.is_null() ? default_lidar_origin_to_beam_origin(prod_line) : origin.as_double();
        beam_to_lidar_transform = mat4d::Identity();
        beam_to_lidar_transform(0, 3) = lidar_origin_to_beam_origin_mm;
    }
    lidar_to_sensor_transform = mat_from(l2s, DEFAULT_LIDAR_TO_SENSOR);
    imu_to_sensor_transform = mat_from(imu2s, mat4d::Zero());
    sensor_to_body = mat_from(d.at("ouster-sdk").at("extrinsic"), mat4d::Identity());
    init_id = static_cast<uint32_t>(si.at("initialization_id").as_u64());
    sn = si.at("prod_sn").as_u64();
}

SensorInfo metadata_from_json(const std::string& json_file, bool /*skip_beam_validation*/) {
    std::ifstream f(json_file);
    if (!f) throw std::runtime_error("Failed to read metadata file: " + json_file);
    std::stringstream ss;
    ss << f.rdbuf();
    return SensorInfo(ss.str());
}

}  // namespace core
}  // namespace sdk
}  // namespace ouster
