// capi_helpers.cpp -- extern "C" conveniences over the C++ host API for non-C++ callers
// (the Python test / bench plumbing): build an ouster_hip_format_desc for a named profile
// from the same PacketFormat tables the C++ FrameBatcher uses.
#include <cstring>
#include <stdexcept>
#include <string>

#include "host_internal.h"
#include "ouster/core/lidar_frame.h"
#include "ouster/core/types.h"
#include "ouster/pcap/pcap.h"

using namespace ouster::sdk::core;

extern "C" {

/** names: n_fields NUL-terminated field names (e.g. "RANGE"); elem_sizes: bytes per plane
 * element (0 = the profile's default plane type).  Returns 0, or -1 with msg filled. */
int ouster_core_format_desc(const char* profile_name, int header_type, uint32_t pixels_per_column,
                            uint32_t columns_per_packet, uint32_t columns_per_frame,
                            const char* const* names, const uint32_t* elem_sizes, uint32_t n_fields,
                            ouster_hip_format_desc* out, char* msg, size_t msg_len) {
    try {
        DataFormat df;
        df.pixels_per_column = pixels_per_column;
        df.columns_per_packet = columns_per_packet;
        df.columns_per_frame = columns_per_frame;
        df.udp_profile_lidar = udp_profile_lidar_of_string(profile_name).value_or(UDPProfileLidar::UNKNOWN);
        df.header_type = header_type ? HeaderType::FUSA : HeaderType::STANDARD;
        df.column_window = {0, static_cast<int>(columns_per_frame) - 1};
        PacketFormat pf(df);
        auto planes = impl::default_planes(df.udp_profile_lidar);
        std::vector<std::pair<std::string, uint32_t>> fields;
        std::vector<bool> nan;
        for (uint32_t i = 0; i < n_fields; ++i) {
            uint32_t es = elem_sizes ? elem_sizes[i] : 0;
            bool f16 = false;
            for (const auto& p : planes)
                if (p.first == names[i]) {
                    f16 = p.second == ChanFieldType::FLOAT16;
                    if (!es) es = static_cast<uint32_t>(field_type_size(p.second)) * (f16 ? 3 : 1);
                }
            if (!es) {
                const FieldDecodeInfo& f = pf.field_decode_info(names[i]);
                es = static_cast<uint32_t>(field_type_size(f.ty_tag)) * f.num_elements;
            }
            fields.emplace_back(names[i], es);
            nan.push_back(f16);
        }
        pf.fill_hip_desc(columns_per_frame, fields, nan, *out);
        return 0;
    } catch (const std::exception& e) {
        if (msg && msg_len) {
            std::strncpy(msg, e.what(), msg_len - 1);
            msg[msg_len - 1] = 0;
        }
        return -1;
    }
}

/** Default plane names of a profile, ';'-separated, with element sizes.  Returns count. */
int ouster_core_default_planes(const char* profile_name, int with_window, char* names_out,
                               size_t names_len, uint32_t* elem_sizes, uint32_t max_n) {
    try {
        auto prof = udp_profile_lidar_of_string(profile_name).value_or(UDPProfileLidar::UNKNOWN);
        auto planes = impl::default_planes(prof);
        std::string s;
        uint32_t n = 0;
        for (const auto& p : planes) {
            if (!with_window && p.first == "WINDOW") continue;
            if (n >= max_n) break;
            if (n) s += ";";
            s += p.first;
            elem_sizes[n++] = static_cast<uint32_t>(field_type_size(p.second)) *
                              (p.second == ChanFieldType::FLOAT16 ? 3 : 1);
        }
        if (s.size() + 1 > names_len) return -1;
        std::memcpy(names_out, s.c_str(), s.size() + 1);
        return static_cast<int>(n);
    } catch (const std::exception&) {
        return -1;
    }
}

/** Read every UDP payload with destination port `port` (0 = any) and, if `exact_size` != 0,
 * exactly that size, from a classic pcap into out (capacity cap bytes, payloads back to back);
 * sizes[i] receives each payload's length (up to max_n).  Returns the number of payloads, or
 * -1 on error (msg filled); *truncated (nullable) is set when matching payloads were left behind
 * because `cap` or `max_n` was reached. */
int ouster_pcap_read_udp(const char* path, int port, size_t exact_size, uint8_t* out, size_t cap,
                         uint32_t* sizes, int* ports, uint32_t max_n, char* msg, size_t msg_len,
                         int* truncated) {
    if (truncated) *truncated = 0;
    try {
        ouster::sdk::pcap::PcapReader r(path);
        size_t used = 0;
        uint32_t n = 0;
        while (r.next_packet()) {
            const auto& info = r.current_info();
            if (port && info.dst_port != port) continue;
            if (exact_size && r.current_length() != exact_size) continue;
            if (n >= max_n || used + r.current_length() > cap) {  // more matching payloads than room
                if (truncated) *truncated = 1;
                break;
            }
            std::memcpy(out + used, r.current_data(), r.current_length());
            used += r.current_length();
            if (sizes) sizes[n] = static_cast<uint32_t>(r.current_length());
            if (ports) ports[n] = info.dst_port;
            ++n;
        }
        return static_cast<int>(n);
    } catch (const std::exception& e) {
        if (msg && msg_len) {
            std::strncpy(msg, e.what(), msg_len - 1);
            msg[msg_len - 1] = 0;
        }
        return -1;
    }
}

}  // extern "C"
