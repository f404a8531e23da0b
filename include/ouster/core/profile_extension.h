// profile_extension.h -- runtime registration of custom UDP lidar profiles.
// The one real plugin hook of the hot path; same signatures as
// ouster_core/include/ouster/core/profile_extension.h:35-55
// (behaviour: ouster_core/src/profile_extension.cpp:130-183).  A registered profile is
// decoded on the GPU by the generic (descriptor-driven) decode kernel.
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "ouster/core/data_format.h"
#include "ouster/core/field_decode_info.h"

namespace ouster {
namespace sdk {
namespace core {

/** @throw std::invalid_argument if profile_nr is 0 or number / name already exist. */
void add_custom_profile(int profile_nr, const std::string& name,
                        const std::vector<std::pair<std::string, FieldDecodeInfo>>& fields,
                        size_t chan_data_size);

/** Allocates the next free profile number.
 * @throw std::runtime_error("Limit of lidar profiles has been reached") */
UDPProfileLidar add_custom_profile(
    const std::string& name, const std::vector<std::pair<std::string, FieldDecodeInfo>>& fields,
    size_t chan_data_size);

}  // namespace core
}  // namespace sdk
}  // namespace ouster
