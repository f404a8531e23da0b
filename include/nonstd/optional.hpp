// nonstd/optional.hpp -- the reference's API spells its optionals nonstd::optional (optional-lite, a third-party header
// it vendors; with C++17 that library is itself an alias of std::optional).  This header keeps that spelling compiling.
#pragma once

#include <optional>

namespace nonstd {
using std::bad_optional_access;
using std::in_place;
using std::make_optional;
using std::nullopt;
using std::nullopt_t;
using std::optional;
}  // namespace nonstd
