// pcap.cpp -- classic pcap -> UDP payloads (see include/ouster/pcap/pcap.h).
// Behaviour modelled on the reference reader: only UDP is surfaced, IPv4 fragments are
// reassembled before delivery (ouster_pcap/src/ip_reassembler.cpp), timestamps in microseconds.
#include "ouster/pcap/pcap.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <tuple>

namespace ouster {
namespace sdk {
namespace pcap {

namespace {
uint16_t be16(const uint8_t* p) { return static_cast<uint16_t>(p[0] << 8 | p[1]); }
std::string ipv4_str(const uint8_t* p) {
    char b[32];
    std::snprintf(b, sizeof b, "%u.%u.%u.%u", p[0], p[1], p[2], p[3]);
    return b;
}
std::string ipv6_str(const uint8_t* p) {
    char b[64];
    std::snprintf(b, sizeof b, "%x:%x:%x:%x:%x:%x:%x:%x", be16(p), be16(p + 2), be16(p + 4),
                  be16(p + 6), be16(p + 8), be16(p + 10), be16(p + 12), be16(p + 14));
    return b;
}
struct FragKey {
    uint32_t src, dst;
    uint16_t id;
    bool operator<(const FragKey& o) const {
        return std::tie(src, dst, id) < std::tie(o.src, o.dst, o.id);
    }
};
struct FragBuf {
    std::vector<uint8_t> data;       // IP payload being rebuilt
    std::vector<bool> have;          // per 8-byte block
    size_t total = 0;                // known once the last fragment arrived
    int count = 0;
    uint64_t first_seen = 0;         // IP packet number of its first fragment (incomplete ones expire)
    uint64_t first_offset = 0;       // file offset of the record that carried the first fragment seen
};
}  // namespace

struct PcapReader::Impl {
    FILE* f = nullptr;
    bool swap = false, nanos = false;
    uint32_t linktype = 0;
    int64_t size = 0, start = 24;
    std::vector<uint8_t> rec, payload;
    PacketInfo info;
    std::map<FragKey, FragBuf> frags;
    uint64_t n_ipv4 = 0;

    uint32_t u32(const uint8_t* p) const {
        uint32_t v;
        std::memcpy(&v, p, 4);
        return swap ? __builtin_bswap32(v) : v;
    }

    // UDP header + payload inside `ip_payload`; fills payload/info.  false if not usable.
    bool take_udp(const uint8_t* udp, size_t len) {
        if (len < 8) return false;
        info.src_port = be16(udp);
        info.dst_port = be16(udp + 2);
        size_t ulen = be16(udp + 4);
        if (ulen < 8 || ulen > len) ulen = len;  // jumbo / truncated: take what is there
        payload.assign(udp + 8, udp + ulen);
        info.payload_size = payload.size();
        return true;
    }

    bool take_ipv4(const uint8_t* ip, size_t len) {
        if (len < 20 || (ip[0] >> 4) != 4) return false;
        const size_t ihl = static_cast<size_t>(ip[0] & 0x0f) * 4;
        size_t tot = be16(ip + 2);
        if (ihl < 20 || ihl > len) return false;
        if (tot < ihl || tot > len) tot = len;
        if (ip[9] != 17) return false;
        info.ip_version = 4;
        info.src_ip = ipv4_str(ip + 12);
        info.dst_ip = ipv4_str(ip + 16);
        const uint16_t ff = be16(ip + 6);
        const bool more = (ff & 0x2000) != 0;
        const size_t frag_off = static_cast<size_t>(ff & 0x1fff) * 8;
        info.fragments_in_packet = 1;
        if (!more && frag_off == 0) return take_udp(ip + ihl, tot - ihl);
        // fragment: collect until every 8-byte block up to the end is present
        FragKey key{0, 0, be16(ip + 4)};
        std::memcpy(&key.src, ip + 12, 4);
        std::memcpy(&key.dst, ip + 16, 4);
        // a datagram that lost a fragment never completes: forget what has been waiting for more than 4096
        // IPv4 packets once a few hundred are pending (lossy captures would otherwise grow without bound)
        ++n_ipv4;
        if (frags.size() > 256)
            for (auto it = frags.begin(); it != frags.end();)
                it = (n_ipv4 - it->second.first_seen > 4096) ? frags.erase(it) : std::next(it);
        FragBuf& fb = frags[key];
        if (fb.count == 0) {
            fb.first_seen = n_ipv4;
            fb.first_offset = info.file_offset;
        }
        const size_t plen = tot - ihl;
        if (fb.data.size() < frag_off + plen) {
            fb.data.resize(frag_off + plen);
            fb.have.resize((frag_off + plen + 7) / 8, false);
        }
        std::memcpy(fb.data.data() + frag_off, ip + ihl, plen);
        for (size_t b = frag_off / 8; b < (frag_off + plen + 7) / 8; ++b) fb.have[b] = true;
        fb.count++;
        if (!more) fb.total = frag_off + plen;
        if (fb.total == 0) return false;
        for (size_t b = 0; b < (fb.total + 7) / 8; ++b)
            if (!fb.have[b]) return false;
        std::vector<uint8_t> whole(fb.data.begin(), fb.data.begin() + static_cast<long>(fb.total));
        info.fragments_in_packet = fb.count;
        // a reassembled datagram is located by its first fragment, so that seek(file_offset) + next_packet() reads it again
        // (the reference reports the record of the last fragment: ouster_pcap/src/pcap.cpp:166-169)
        info.file_offset = fb.first_offset;
        frags.erase(key);
        return take_udp(whole.data(), whole.size());
    }

    bool take_ipv6(const uint8_t* ip, size_t len) {
        if (len < 40 || (ip[0] >> 4) != 6 || ip[6] != 17) return false;
        info.ip_version = 6;
        info.src_ip = ipv6_str(ip + 8);
        info.dst_ip = ipv6_str(ip + 24);
        info.fragments_in_packet = 1;
        return take_udp(ip + 40, len - 40);
    }

    bool take_record() {
        const uint8_t* p = rec.data();
        size_t len = rec.size();
        uint16_t ethertype = 0;
        switch (linktype) {
            case 1:  // Ethernet, optional 802.1Q tags
                if (len < 14) return false;
                ethertype = be16(p + 12);
                p += 14; len -= 14;
                while ((ethertype == 0x8100 || ethertype == 0x88a8) && len >= 4) {
                    ethertype = be16(p + 2);
                    p += 4; len -= 4;
                }
                break;
            case 113:  // Linux cooked v1
                if (len < 16) return false;
                ethertype = be16(p + 14);
                p += 16; len -= 16;
                break;
            case 101: case 228: case 229: case 12:  // raw IP
                ethertype = (len && (p[0] >> 4) == 6) ? 0x86dd : 0x0800;
                break;
            default:
                return false;
        }
        if (ethertype == 0x0800) return take_ipv4(p, len);
        if (ethertype == 0x86dd) return take_ipv6(p, len);
        return false;
    }
};

PcapReader::PcapReader(const std::string& file) : impl_(new Impl()) {
    impl_->f = std::fopen(file.c_str(), "rb");
    if (!impl_->f) throw std::runtime_error("PcapReader: cannot open " + file);
    std::fseek(impl_->f, 0, SEEK_END);
    impl_->size = std::ftell(impl_->f);
    std::fseek(impl_->f, 0, SEEK_SET);
    uint8_t hdr[24];
    if (std::fread(hdr, 1, 24, impl_->f) != 24) {
        if (impl_->size == 0) return;  // empty placeholder capture: no packets
        throw std::runtime_error("PcapReader: truncated pcap header in " + file);
    }
    uint32_t magic;
    std::memcpy(&magic, hdr, 4);
    switch (magic) {
        case 0xa1b2c3d4u: break;
        case 0xa1b23c4du: impl_->nanos = true; break;
        case 0xd4c3b2a1u: impl_->swap = true; break;
        case 0x4d3cb2a1u: impl_->swap = true; impl_->nanos = true; break;
        default: throw std::runtime_error("PcapReader: not a classic pcap file: " + file);
    }
    impl_->linktype = impl_->u32(hdr + 20);
    impl_->info.encapsulation_protocol = static_cast<int>(impl_->linktype);
}
PcapReader::PcapReader(PcapReader&&) noexcept = default;
PcapReader& PcapReader::operator=(PcapReader&&) noexcept = default;
PcapReader::~PcapReader() {
    if (impl_ && impl_->f) std::fclose(impl_->f);
}

size_t PcapReader::next_packet() {
    Impl& s = *impl_;
    if (!s.f) return 0;
    for (;;) {
        const int64_t off = std::ftell(s.f);
        uint8_t rh[16];
        if (std::fread(rh, 1, 16, s.f) != 16) {
            s.info.file_offset = static_cast<uint64_t>(off);   // at the end the cached info points past the last record
            s.payload.clear();
            return 0;
        }
        const uint32_t sec = s.u32(rh), frac = s.u32(rh + 4), incl = s.u32(rh + 8);
        if (incl > (64u << 20)) return 0;  // corrupt record
        s.rec.resize(incl);
        if (incl && std::fread(s.rec.data(), 1, incl, s.f) != incl) return 0;
        s.info.file_offset = static_cast<uint64_t>(off);
        s.info.packet_size = incl;
        s.info.timestamp = PacketInfo::ts(static_cast<int64_t>(sec) * 1000000 +
                                          (s.nanos ? frac / 1000 : frac));
        if (s.take_record()) return s.payload.size();
    }
}

const uint8_t* PcapReader::current_data() const { return impl_->payload.data(); }
size_t PcapReader::current_length() const { return impl_->payload.size(); }
const PacketInfo& PcapReader::current_info() const { return impl_->info; }
int64_t PcapReader::file_size() const { return impl_->size; }
void PcapReader::reset() { seek(static_cast<uint64_t>(impl_->start)); }
void PcapReader::seek(uint64_t offset) {
    if (!impl_->f) return;
    if (offset < static_cast<uint64_t>(impl_->start)) offset = static_cast<uint64_t>(impl_->start);
    std::fseek(impl_->f, static_cast<long>(offset), SEEK_SET);
    impl_->frags.clear();
}
int64_t PcapReader::current_offset() const { return impl_->f ? std::ftell(impl_->f) : 0; }

}  // namespace pcap
}  // namespace sdk
}  // namespace ouster
