// field_decode_info.h -- bit-field descriptor of a packet value.
// Same layout and get/set semantics as
// ouster_core/include/ouster/core/field_decode_info.h:24-78.  These host-side
// accessors serve packet/column HEADER reads and test-side packet synthesis; bulk pixel
// decode goes through the HIP kernels (include/ouster_hip.h).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ouster/core/chanfield.h"

namespace ouster {
namespace sdk {
namespace core {

struct FieldDecodeInfo {
    ChanFieldType ty_tag;
    size_t offset;
    uint64_t mask;
    int shift;
    int num_elements = 1;

    /** NOTE: reads 8 bytes at buffer + offset. */
    template <typename T>
    T get(const uint8_t* buffer) const {
        uint64_t word;
        std::memcpy(&word, buffer + offset, sizeof word);
        word &= mask;
        if (shift > 0) word >>= shift;
        else if (shift < 0) word <<= -shift;
        T out{};
        std::memcpy(&out, &word, sizeof(out) < sizeof(word) ? sizeof(out) : sizeof(word));
        return out;
    }

    /** NOTE: read-modify-writes 8 bytes at buffer + offset. */
    template <typename T>
    void set(uint8_t* buffer, T value) const {
        uint64_t word = 0;
        std::memcpy(&word, &value, sizeof(value) < sizeof(word) ? sizeof(value) : sizeof(word));
        if (shift > 0) word <<= shift;
        if (shift < 0) word >>= -shift;
        word &= mask;
        uint64_t cur;
        std::memcpy(&cur, buffer + offset, sizeof cur);
        cur = (cur & ~mask) | word;
        std::memcpy(buffer + offset, &cur, sizeof cur);
    }
};

/** Build a descriptor from a bit position (ouster_core/src/parsing.cpp:57-122). */
FieldDecodeInfo field_info(size_t bit_start, size_t bit_size, size_t upshift = 0,
                           size_t max_length = 0, size_t num_elements = 1);

namespace impl {
uint64_t get_value_mask(const FieldDecodeInfo& f);
int get_bitness(const FieldDecodeInfo& f);
}  // namespace impl

}  // namespace core
}  // namespace sdk
}  // namespace ouster
