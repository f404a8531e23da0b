// indexed_pcap_reader.h -- a capture reader that knows which sensor every datagram belongs to and where every frame of
// every sensor starts (ouster_pcap/include/ouster/pcap/indexed_pcap_reader.h:33-260; behaviour:
// ouster_pcap/src/indexed_pcap_reader.cpp).  This is the demultiplexer in front of the multi-sensor batch (SURVEY section 8
// f-1, configs[4]): sensors may share a destination port; a datagram is routed by port, payload size and the init_id /
// serial number of its packet header (validate_packet).  Host only.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "nonstd/optional.hpp"
#include "ouster/core/types.h"
#include "ouster/pcap/os_pcap.h"
#include "ouster/pcap/pcap.h"

namespace ouster {
namespace sdk {
namespace pcap {

/** A frame start in capture order over all sensors. */
struct GlobalIndex {
    uint64_t file_offset;
    uint64_t sensor_index;
    uint64_t timestamp;
};

/** Frame number -> file offset, per sensor. */
class PcapIndex {
   public:
    using frame_index = std::vector<uint64_t>;
    using timestamp_index = std::unordered_map<uint64_t, uint64_t>;
    using frame_id_index = std::unordered_map<int32_t, uint64_t>;

    std::vector<frame_index> frame_indices;
    std::vector<GlobalIndex> global_frame_indices;
    std::vector<timestamp_index> frame_timestamp_indices;  ///< first-packet capture time -> offset
    std::vector<frame_id_index> frame_id_indices;          ///< frame id -> offset

    explicit PcapIndex(size_t num_sensors)
        : frame_indices(num_sensors), frame_timestamp_indices(num_sensors), frame_id_indices(num_sensors) {}

    void clear();
    /** @throw std::out_of_range if there is no such sensor */
    size_t frame_count(size_t sensor_index) const;
    /** Position `reader` on the first packet of a frame. @throw std::out_of_range for an unknown sensor or frame */
    void seek_to_frame(PcapReader& reader, size_t sensor_index, unsigned int frame_number);
};

enum class IdxErrorType { NONE, SIZE, ID, None = NONE, Size = SIZE, Id = ID };

/** Two sensors claim the same (port, serial number / legacy stream) pair. */
class PcapDuplicatePortException : public std::runtime_error {
   public:
    explicit PcapDuplicatePortException(const std::string& msg) : std::runtime_error(msg) {}
};

class IndexedPcapReader : public PcapReader {
   public:
    /** @throw PcapDuplicatePortException when two metadata entries cannot be told apart on a port */
    IndexedPcapReader(const std::string& pcap_filename, const std::vector<std::string>& metadata_filenames);
    IndexedPcapReader(const std::string& pcap_filename, const std::vector<core::SensorInfo>& sensor_infos);

    /** Read the whole capture once and fill the index; the reader is rewound afterwards. */
    void build_index();
    const PcapIndex& get_index() const;

    /** Index into sensor_info() of the sensor that sent the current datagram, if any.  With `soft_id_check` a datagram
     *  whose size fits but whose init_id / serial number does not is still attributed (single-sensor captures only). */
    nonstd::optional<size_t> sensor_idx_for_current_packet(bool soft_id_check = false) const;
    std::pair<IdxErrorType, nonstd::optional<size_t>> check_sensor_idx_for_current_packet(bool soft_id_check) const;
    /** Frame id of the current datagram when it is a lidar packet of a known sensor. */
    nonstd::optional<uint32_t> current_frame_id() const;
    /** Record the current datagram in the index when it starts a new frame. @return progress through the file, percent */
    int update_index_for_current_packet();
    const std::vector<core::SensorInfo>& sensor_info() const;

   protected:
    void init_();

    std::vector<core::SensorInfo> sensor_infos_;
    std::vector<core::PacketFormat> packet_formats_;
    PcapIndex index_;
    std::vector<nonstd::optional<uint32_t>> previous_frame_ids_;
    /// destination port -> (serial number, or LEGACY_LIDAR / LEGACY_IMU for streams without ids) -> sensor index
    std::unordered_map<uint16_t, std::map<std::string, uint64_t>> port_map_;
    std::string filename_;
};

}  // namespace pcap
}  // namespace sdk
}  // namespace ouster
