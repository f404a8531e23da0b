// pcap_index.cpp -- stream survey, port guessing and the per-sensor frame index of a capture
// (include/ouster/pcap/os_pcap.h, include/ouster/pcap/indexed_pcap_reader.h; reference behaviour:
// ouster_pcap/src/os_pcap.cpp:188-353, ouster_pcap/src/indexed_pcap_reader.cpp).
#include <algorithm>
#include <set>

#include "ouster/core/packet.h"
#include "ouster/pcap/indexed_pcap_reader.h"

namespace ouster {
namespace sdk {
namespace pcap {

using core::PacketFormat;
using core::PacketType;
using core::PacketValidationFailure;
using core::SensorInfo;

// ---------------------------------------------------------------------------------------
// stream survey
// ---------------------------------------------------------------------------------------
std::shared_ptr<StreamInfo> get_stream_info(const std::string& file,
                                            const std::function<void(uint64_t, uint64_t, uint64_t)>& progress_callback,
                                            int packets_per_callback, int packets_to_process) {
    PcapReader reader(file);
    auto out = std::make_shared<StreamInfo>();
    const uint64_t total = static_cast<uint64_t>(reader.file_size());
    uint64_t last_reported = 0;
    bool first = true;
    while ((packets_to_process < 0 || out->total_packets < static_cast<uint64_t>(packets_to_process)) &&
           reader.next_packet() != 0) {
        const PacketInfo& pi = reader.current_info();
        ++out->total_packets;
        out->encapsulation_protocol = static_cast<uint32_t>(pi.encapsulation_protocol);
        if (first || pi.timestamp < out->timestamp_min) out->timestamp_min = pi.timestamp;
        if (first || pi.timestamp > out->timestamp_max) out->timestamp_max = pi.timestamp;
        first = false;
        StreamData& sd = out->udp_streams[StreamKey{pi.dst_ip, pi.src_ip, pi.src_port, pi.dst_port}];
        ++sd.count;
        ++sd.payload_size_counts[pi.payload_size];
        ++sd.fragment_counts[static_cast<uint64_t>(pi.fragments_in_packet)];
        ++sd.ip_version_counts[static_cast<uint64_t>(pi.ip_version)];
        if (progress_callback && packets_per_callback > 0 && out->total_packets % packets_per_callback == 0) {
            const uint64_t now = static_cast<uint64_t>(reader.current_offset());
            progress_callback(now, now - last_reported, total);
            last_reported = now;
        }
    }
    if (progress_callback) progress_callback(total, total - last_reported, total);
    return out;
}

std::shared_ptr<StreamInfo> get_stream_info(const std::string& file, int packets_to_process) {
    return get_stream_info(file, nullptr, 0, packets_to_process);
}

std::vector<GuessedPorts> guess_ports(StreamInfo& info, int lidar_packet_size, int imu_packet_size, int expected_lidar_port,
                                      int expected_imu_port) {
    // streams that carried at least one datagram of the right size, by role
    std::vector<const StreamKey*> lidar, imu;
    std::set<std::string> lidar_sources, imu_sources;
    for (const auto& kv : info.udp_streams) {
        if (kv.second.payload_size_counts.count(static_cast<uint64_t>(lidar_packet_size))) {
            lidar.push_back(&kv.first);
            lidar_sources.insert(kv.first.src_ip);
        }
        if (kv.second.payload_size_counts.count(static_cast<uint64_t>(imu_packet_size))) {
            imu.push_back(&kv.first);
            imu_sources.insert(kv.first.src_ip);
        }
    }
    std::vector<GuessedPorts> paired, lidar_only, imu_only;
    for (const StreamKey* l : lidar) {
        for (const StreamKey* i : imu)
            if (i->src_ip == l->src_ip) paired.push_back({l->dst_port, i->dst_port});
        if (!imu_sources.count(l->src_ip)) lidar_only.push_back({l->dst_port, 0});
    }
    for (const StreamKey* i : imu)   // an IMU stream whose source sends no lidar stream stands alone
        if (!lidar_sources.count(i->src_ip)) imu_only.push_back({0, i->dst_port});
    std::vector<GuessedPorts> out;
    auto fits = [](int port, int expected) { return port == expected || expected == 0 || port == 0; };
    for (const auto* group : {&paired, &lidar_only, &imu_only})
        for (const GuessedPorts& g : *group)
            if (fits(g.lidar, expected_lidar_port) && fits(g.imu, expected_imu_port)) out.push_back(g);
    return out;
}

// ---------------------------------------------------------------------------------------
// PcapIndex
// ---------------------------------------------------------------------------------------
void PcapIndex::clear() {
    for (auto& v : frame_indices) v.clear();
    for (auto& m : frame_timestamp_indices) m.clear();
    for (auto& m : frame_id_indices) m.clear();
}

size_t PcapIndex::frame_count(size_t sensor_index) const { return frame_indices.at(sensor_index).size(); }

void PcapIndex::seek_to_frame(PcapReader& reader, size_t sensor_index, unsigned int frame_number) {
    reader.seek(frame_indices.at(sensor_index).at(frame_number));
}

// ---------------------------------------------------------------------------------------
// IndexedPcapReader
// ---------------------------------------------------------------------------------------
IndexedPcapReader::IndexedPcapReader(const std::string& pcap_filename, const std::vector<std::string>& metadata_filenames)
    : PcapReader(pcap_filename),
      index_(metadata_filenames.size()),
      previous_frame_ids_(metadata_filenames.size()),
      filename_(pcap_filename) {
    for (const std::string& name : metadata_filenames) sensor_infos_.push_back(core::metadata_from_json(name));
    init_();
}

IndexedPcapReader::IndexedPcapReader(const std::string& pcap_filename, const std::vector<SensorInfo>& sensor_infos)
    : PcapReader(pcap_filename),
      sensor_infos_(sensor_infos),
      index_(sensor_infos.size()),
      previous_frame_ids_(sensor_infos.size()),
      filename_(pcap_filename) {
    init_();
}

void IndexedPcapReader::init_() {
    // ports a metadata file does not name are guessed from the first datagrams of the capture (old single-sensor recordings)
    const std::shared_ptr<StreamInfo> survey = get_stream_info(filename_, 1000);
    for (size_t idx = 0; idx < sensor_infos_.size(); ++idx) {
        SensorInfo& info = sensor_infos_[idx];
        packet_formats_.emplace_back(info);
        const PacketFormat& pf = packet_formats_.back();

        std::vector<GuessedPorts> guesses =
            guess_ports(*survey, static_cast<int>(pf.lidar_packet_size), static_cast<int>(pf.imu_packet_size),
                        info.config.udp_port_lidar.value_or(0), info.config.udp_port_imu.value_or(0));
        // prefer a guess with a lidar port, then one with an IMU port, then the larger port numbers
        std::stable_sort(guesses.begin(), guesses.end(), [](const GuessedPorts& a, const GuessedPorts& b) {
            if ((a.lidar != 0) != (b.lidar != 0)) return a.lidar != 0;
            if ((a.imu != 0) != (b.imu != 0)) return a.imu != 0;
            return std::make_pair(a.lidar, a.imu) > std::make_pair(b.lidar, b.imu);
        });
        if (!guesses.empty()) {
            if (!info.config.udp_port_lidar) info.config.udp_port_lidar = guesses[0].lidar;
            if (!info.config.udp_port_imu) info.config.udp_port_imu = guesses[0].imu;
        }

        // streams without ids in their packets cannot be told apart by content: one such stream per port
        const std::string sn = std::to_string(info.sn);
        const bool legacy_lidar = info.config.udp_profile_lidar && *info.config.udp_profile_lidar == core::UDPProfileLidar::LEGACY;
        const bool legacy_imu = info.config.udp_profile_imu && *info.config.udp_profile_imu == core::UDPProfileIMU::LEGACY;
        auto claim = [&](const char* what, const nonstd::optional<int>& port, const std::string& who) {
            if (!port || *port == 0) return;   // unknown, or the stream is switched off
            auto& owners = port_map_[static_cast<uint16_t>(*port)];
            if (owners.count(who))
                throw PcapDuplicatePortException(std::string("Duplicate ") + what + " port/sn found in pcap: " + who + ":" +
                                                 std::to_string(*port));
            owners[who] = idx;
        };
        claim("lidar", info.config.udp_port_lidar, legacy_lidar ? "LEGACY_LIDAR" : sn);
        claim("imu", info.config.udp_port_imu, legacy_imu ? "LEGACY_IMU" : sn);
        claim("zm", info.config.udp_port_zm, sn);
    }
}

nonstd::optional<size_t> IndexedPcapReader::sensor_idx_for_current_packet(bool soft_id_check) const {
    return check_sensor_idx_for_current_packet(soft_id_check).second;
}

std::pair<IdxErrorType, nonstd::optional<size_t>> IndexedPcapReader::check_sensor_idx_for_current_packet(
    bool soft_id_check) const {
    nonstd::optional<size_t> soft;
    IdxErrorType error = IdxErrorType::NONE;
    const PacketInfo& pi = current_info();
    const auto owners = port_map_.find(static_cast<uint16_t>(pi.dst_port));
    if (owners == port_map_.end()) return {error, soft};
    for (const auto& kv : owners->second) {
        const size_t idx = static_cast<size_t>(kv.second);
        const PacketFormat& pf = packet_formats_[idx];
        const PacketType type = pi.payload_size == pf.imu_packet_size    ? PacketType::Imu
                                : pi.payload_size == pf.zone_packet_size ? PacketType::Zone
                                                                         : PacketType::Lidar;
        switch (core::validate_packet(sensor_infos_[idx], pf, current_data(), pi.payload_size, type)) {
            case PacketValidationFailure::NONE: return {IdxErrorType::NONE, idx};
            case PacketValidationFailure::ID:
                if (soft_id_check) {
                    if (soft) throw std::runtime_error("Soft ID Checking Does NOT Work With Multiple Sensors");
                    soft = idx;
                }
                error = IdxErrorType::ID;
                break;
            case PacketValidationFailure::PACKET_SIZE:
                if (error == IdxErrorType::NONE) error = IdxErrorType::SIZE;   // an id mismatch is the stronger finding
                break;
        }
    }
    return {error, soft};
}

nonstd::optional<uint32_t> IndexedPcapReader::current_frame_id() const {
    if (current_info().payload_size == 48) return nonstd::nullopt;   // a legacy IMU packet
    if (const nonstd::optional<size_t> idx = sensor_idx_for_current_packet())
        return packet_formats_[*idx].frame_id(current_data());
    return nonstd::nullopt;
}

int IndexedPcapReader::update_index_for_current_packet() {
    if (const nonstd::optional<size_t> idx = sensor_idx_for_current_packet()) {
        if (const nonstd::optional<uint32_t> fid = current_frame_id()) {
            nonstd::optional<uint32_t>& prev = previous_frame_ids_[*idx];
            if (!prev || packet_formats_[*idx].frame_id_difference(*prev, *fid) > 0) {
                const PacketInfo& pi = current_info();
                const uint64_t stamp = static_cast<uint64_t>(pi.timestamp.count());
                index_.frame_indices[*idx].push_back(pi.file_offset);
                index_.frame_timestamp_indices[*idx].insert({stamp, pi.file_offset});
                index_.frame_id_indices[*idx].insert({static_cast<int32_t>(*fid), pi.file_offset});
                index_.global_frame_indices.push_back({pi.file_offset, *idx, stamp});
                prev = *fid;
            }
        }
    }
    const int64_t size = file_size();
    if (size <= 0) return 100;
    return static_cast<int>(100.0f * static_cast<float>(current_offset()) / static_cast<float>(size));
}

void IndexedPcapReader::build_index() {
    index_.clear();
    index_.global_frame_indices.clear();
    std::fill(previous_frame_ids_.begin(), previous_frame_ids_.end(), nonstd::nullopt);
    reset();
    while (next_packet() != 0) update_index_for_current_packet();
    reset();
}

const std::vector<SensorInfo>& IndexedPcapReader::sensor_info() const { return sensor_infos_; }
const PcapIndex& IndexedPcapReader::get_index() const { return index_; }

}  // namespace pcap
}  // namespace sdk
}  // namespace ouster
