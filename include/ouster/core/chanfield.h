// chanfield.h -- channel field names and element types.
// Same names and numeric values as the reference
// (ouster_core/include/ouster/core/chanfield.h:20-128, src/chanfield.cpp:54-89).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

namespace ouster {
namespace sdk {
namespace core {

namespace ChanField {
using cf_type = const char*;
static constexpr cf_type RANGE = "RANGE";
static constexpr cf_type RANGE2 = "RANGE2";
static constexpr cf_type SIGNAL = "SIGNAL";
static constexpr cf_type SIGNAL2 = "SIGNAL2";
static constexpr cf_type REFLECTIVITY = "REFLECTIVITY";
static constexpr cf_type REFLECTIVITY2 = "REFLECTIVITY2";
static constexpr cf_type NEAR_IR = "NEAR_IR";
static constexpr cf_type FLAGS = "FLAGS";
static constexpr cf_type FLAGS2 = "FLAGS2";
static constexpr cf_type WINDOW = "WINDOW";
static constexpr cf_type R = "R";
static constexpr cf_type G = "G";
static constexpr cf_type B = "B";
static constexpr cf_type RGB = "RGB";
static constexpr cf_type ZONE_MASK = "ZONE_MASK";
static constexpr cf_type RAW_HEADERS = "RAW_HEADERS";
static constexpr cf_type RAW32_WORD1 = "RAW32_WORD1";
static constexpr cf_type RAW32_WORD2 = "RAW32_WORD2";
static constexpr cf_type RAW32_WORD3 = "RAW32_WORD3";
static constexpr cf_type RAW32_WORD4 = "RAW32_WORD4";
static constexpr cf_type RAW32_WORD5 = "RAW32_WORD5";
static constexpr cf_type RAW32_WORD6 = "RAW32_WORD6";
static constexpr cf_type RAW32_WORD7 = "RAW32_WORD7";
static constexpr cf_type RAW32_WORD8 = "RAW32_WORD8";
static constexpr cf_type RAW32_WORD9 = "RAW32_WORD9";
}  // namespace ChanField

/** Element type of a field; numeric values match the reference enum. */
enum class ChanFieldType {
    VOID = 0,
    UINT8 = 1,
    UINT16 = 2,
    UINT32 = 3,
    UINT64 = 4,
    INT8 = 5,
    INT16 = 6,
    INT32 = 7,
    INT64 = 8,
    FLOAT32 = 9,
    FLOAT64 = 10,
    CHAR = 11,
    FLOAT16 = 12,
    ZONE_STATE = 30,
    UNREGISTERED = 100
};

inline size_t field_type_size(ChanFieldType t) {
    switch (t) {
        case ChanFieldType::INT8:
        case ChanFieldType::UINT8:
        case ChanFieldType::CHAR:
            return 1;
        case ChanFieldType::INT16:
        case ChanFieldType::UINT16:
        case ChanFieldType::FLOAT16:
            return 2;
        case ChanFieldType::INT32:
        case ChanFieldType::UINT32:
        case ChanFieldType::FLOAT32:
            return 4;
        case ChanFieldType::INT64:
        case ChanFieldType::UINT64:
        case ChanFieldType::FLOAT64:
            return 8;
        default:
            return 0;
    }
}

inline uint64_t field_type_mask(ChanFieldType t) {
    switch (field_type_size(t)) {
        case 1: return 0xffull;
        case 2: return 0xffffull;
        case 4: return 0xffffffffull;
        case 8: return 0xffffffffffffffffull;
        default: throw std::runtime_error("field_type_mask error: wrong ChanFieldType");
    }
}

std::string to_string(ChanFieldType t);

/** 16-bit float storage type (bit pattern only; no arithmetic on this path). */
struct float16_t {
    uint16_t bits;
};

/** Map C++ element types to their tag. */
template <typename T> struct FieldTag { static constexpr ChanFieldType tag = ChanFieldType::VOID; };
template <> struct FieldTag<uint8_t> { static constexpr ChanFieldType tag = ChanFieldType::UINT8; };
template <> struct FieldTag<uint16_t> { static constexpr ChanFieldType tag = ChanFieldType::UINT16; };
template <> struct FieldTag<uint32_t> { static constexpr ChanFieldType tag = ChanFieldType::UINT32; };
template <> struct FieldTag<uint64_t> { static constexpr ChanFieldType tag = ChanFieldType::UINT64; };
template <> struct FieldTag<int8_t> { static constexpr ChanFieldType tag = ChanFieldType::INT8; };
template <> struct FieldTag<int16_t> { static constexpr ChanFieldType tag = ChanFieldType::INT16; };
template <> struct FieldTag<int32_t> { static constexpr ChanFieldType tag = ChanFieldType::INT32; };
template <> struct FieldTag<int64_t> { static constexpr ChanFieldType tag = ChanFieldType::INT64; };
template <> struct FieldTag<float> { static constexpr ChanFieldType tag = ChanFieldType::FLOAT32; };
template <> struct FieldTag<double> { static constexpr ChanFieldType tag = ChanFieldType::FLOAT64; };
template <> struct FieldTag<float16_t> { static constexpr ChanFieldType tag = ChanFieldType::FLOAT16; };

}  // namespace core
}  // namespace sdk
}  // namespace ouster
