// pcap.h -- classic-pcap reader feeding UDP payloads to the batcher.
// Same class and method names as the reference reader
// (ouster_pcap/include/ouster/pcap/pcap.h:30-185: PacketInfo, PcapReader::next_packet /
// current_data / current_length / current_info / reset / seek); the reference builds on
// libtins + libpcap (ouster_pcap/src/pcap.cpp, ip_reassembler.cpp), this one parses the
// container and the Ethernet / VLAN / Linux-cooked / raw-IP, IPv4 (with fragment reassembly),
// IPv6 and UDP headers itself.  Host only; pcapng is not supported.
#pragma once

#include <chrono>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace ouster {
namespace sdk {
namespace pcap {

struct PacketInfo {
    using ts = std::chrono::microseconds;
    std::string dst_ip;
    std::string src_ip;
    int dst_port = 0;
    int src_port = 0;
    size_t payload_size = 0;
    size_t packet_size = 0;
    ts timestamp{0};
    int fragments_in_packet = 0;
    int ip_version = 0;
    int encapsulation_protocol = 0;  ///< pcap link type
    uint64_t file_offset = 0;        ///< of the record the datagram starts in (its first fragment when it was reassembled)
    int network_protocol = 17;       ///< always UDP
};

class PcapReader {
   public:
    /** @throw std::runtime_error if the file cannot be opened or is not a classic pcap. */
    explicit PcapReader(const std::string& file);
    PcapReader(const PcapReader&) = delete;
    PcapReader& operator=(const PcapReader&) = delete;
    PcapReader(PcapReader&&) noexcept;
    PcapReader& operator=(PcapReader&&) noexcept;
    virtual ~PcapReader();

    /** Advance to the next complete UDP datagram. @return payload size, 0 at end of file. */
    size_t next_packet();
    const uint8_t* current_data() const;
    size_t current_length() const;
    const PacketInfo& current_info() const;
    int64_t file_size() const;
    void reset();
    void seek(uint64_t offset);
    int64_t current_offset() const;

   private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace pcap
}  // namespace sdk
}  // namespace ouster
