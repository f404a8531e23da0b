// TEST INFRASTRUCTURE (oracle/): what the reference's include/ouster/core/field_decode_info.h needs from its own
// "ouster/core/types.h" -- the ChanFieldType tag and a few std headers -- so that the header compiles from where it lies in
// /root/reference (oracle/Makefile, target _ref/libdecode_ref.so; the real types.h pulls in Eigen, absent from this image).
// Never used by the product.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace ouster {
namespace sdk {
namespace core {
// only the tag's size matters to the decode loop (field_type_size); the values are this shim's own: bytes per element
enum ChanFieldType { VOID = 0, UINT8 = 1, UINT16 = 2, UINT32 = 4, UINT64 = 8 };
}  // namespace core
}  // namespace sdk
}  // namespace ouster
