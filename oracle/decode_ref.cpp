// TEST INFRASTRUCTURE (oracle/_ref): C entry points around the REFERENCE's own packet-decode loop,
//   FieldDecodeInfo::get<T>            /root/reference/ouster_core/include/ouster/core/field_decode_info.h:41-54 (the header, as it lies)
//   PacketFormat::block_field<T,B>     /root/reference/ouster_core/src/parsing.cpp:628-657 (the one member function, staged at
//                                      build time by oracle/stage_slice.py: the rest of parsing.cpp needs Eigen, jsoncons, ...)
// compiled against oracle/shims/ref_decode and the stand-in PacketFormat below, which holds exactly the members that function
// reads (the geometry and field tables are handed in by the caller: they are the oracle's, pinned on the reference's bit-width
// table and header KATs in tests/test_oracle_golden.py).  tests/test_oracle_ref_decode.py pins the oracle's ora_block_field on
// this bit for bit; bench.py times it as the CPU baseline's decode leg (kind "reference").  Never used by the product.
#include <chrono>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ouster/core/field_decode_info.h"

namespace ouster {
namespace sdk {
namespace core {

static size_t field_type_size(ChanFieldType t) { return static_cast<size_t>(t); }

class PacketFormat {
    struct Impl {
        std::map<std::string, FieldDecodeInfo> fields;
        size_t channel_data_size = 0, packet_header_size = 0, col_size = 0;
        FieldDecodeInfo col_measurement_id_info{};
    };
    std::shared_ptr<Impl> impl_ = std::make_shared<Impl>();

   public:
    int columns_per_packet = 0, pixels_per_column = 0;
    size_t col_header_size = 0;

    PacketFormat(size_t packet_header, size_t col_header, size_t col, size_t chan, int cpp, int h, FieldDecodeInfo mid)
        : columns_per_packet(cpp), pixels_per_column(h), col_header_size(col_header) {
        impl_->channel_data_size = chan;
        impl_->packet_header_size = packet_header;
        impl_->col_size = col;
        impl_->col_measurement_id_info = mid;
    }
    void add_field(const std::string& name, FieldDecodeInfo f) { impl_->fields[name] = f; }
    // parsing.cpp:786-792, :803-805 (pointer arithmetic and one get<uint16_t>)
    const uint8_t* nth_col(size_t col_idx, const uint8_t* lidar_buf) const {
        return lidar_buf + impl_->packet_header_size + (col_idx * impl_->col_size);
    }
    uint16_t col_measurement_id(const uint8_t* col_buf) const { return impl_->col_measurement_id_info.get<uint16_t>(col_buf); }
    template <typename T, int BlockDim>
    void block_field(T* data, int cols, const std::string& field_name, const uint8_t* lidar_buf) const;
};

#include "block_field_staged.inc"

}  // namespace core
}  // namespace sdk
}  // namespace ouster

using namespace ouster::sdk::core;

struct ref_fdi {   // the caller's view of one FieldDecodeInfo
    uint64_t offset, mask;
    int32_t shift, type_bytes;
};
static FieldDecodeInfo to_fdi(const ref_fdi& f) {
    FieldDecodeInfo o{};
    o.ty_tag = static_cast<ChanFieldType>(f.type_bytes);
    o.offset = f.offset;
    o.mask = f.mask;
    o.shift = f.shift;
    o.num_elements = 1;
    return o;
}

template <typename T>
static void block_t(const PacketFormat& pf, void* data, int cols, const std::string& name, const uint8_t* buf, int bd) {
    switch (bd) {
        case 16: pf.block_field<T, 16>(static_cast<T*>(data), cols, name, buf); break;
        case 8: pf.block_field<T, 8>(static_cast<T*>(data), cols, name, buf); break;
        default: pf.block_field<T, 4>(static_cast<T*>(data), cols, name, buf); break;
    }
}
static int block_any(const PacketFormat& pf, void* data, size_t elem, int cols, const std::string& name, const uint8_t* buf, int bd) {
    try {
        switch (elem) {
            case 1: block_t<uint8_t>(pf, data, cols, name, buf, bd); break;
            case 2: block_t<uint16_t>(pf, data, cols, name, buf, bd); break;
            case 4: block_t<uint32_t>(pf, data, cols, name, buf, bd); break;
            case 8: block_t<uint64_t>(pf, data, cols, name, buf, bd); break;
            default: return -3;
        }
    } catch (const std::invalid_argument&) {
        return -2;   // "Dest type too small for specified field"
    }
    return 0;
}

extern "C" {
// geometry: {packet_header_size, col_header_size, col_size, channel_data_size, columns_per_packet, pixels_per_column}
void* ref_pf_new(const uint64_t* geometry, const ref_fdi* measurement_id) {
    return new PacketFormat(geometry[0], geometry[1], geometry[2], geometry[3], (int)geometry[4], (int)geometry[5], to_fdi(*measurement_id));
}
void ref_pf_add_field(void* pf, const char* name, const ref_fdi* f) { static_cast<PacketFormat*>(pf)->add_field(name, to_fdi(*f)); }
void ref_pf_free(void* pf) { delete static_cast<PacketFormat*>(pf); }
// block_field<T, block_dim>(data, cols, name, lidar_buf): T by its size; -2 where the reference throws
int ref_block_field(const void* pf, void* data, size_t elem, int cols, const char* name, const uint8_t* lidar_buf, int block_dim) {
    return block_any(*static_cast<const PacketFormat*>(pf), data, elem, cols, name, lidar_buf, block_dim);
}
// The decode half of one frame of the benchmark workload: block_field of every plane for every packet (parse_by_block,
// lidar_frame.cpp:1492-1528), `reps` times on one core.  Returns seconds.
double ref_bench_decode_frame(const void* pf, const uint8_t* packets, size_t n_packets, size_t packet_stride, const char* const* names,
                              void* const* planes, const size_t* elem, size_t n_planes, int cols, int block_dim, int reps) {
    const PacketFormat& f = *static_cast<const PacketFormat*>(pf);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (size_t p = 0; p < n_packets; ++p)
            for (size_t i = 0; i < n_planes; ++i) block_any(f, planes[i], elem[i], cols, names[i], packets + p * packet_stride, block_dim);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}
