// visibility.h -- the export macros the reference decorates its API with
// (ouster_core/include/ouster/core/visibility.h:18-93).  This mirror builds with default visibility, so they expand to
// the plain GCC attribute (or nothing); they exist so that code written against the reference -- its own tests declare
// `OUSTER_API_FUNCTION ... get_profiles();` -- compiles unchanged.
#pragma once

#if defined(__GNUC__) || defined(__clang__)
#define OUSTER_API_CLASS __attribute__((visibility("default")))
#define OUSTER_API_FUNCTION __attribute__((visibility("default")))
#define OUSTER_API_VAR __attribute__((visibility("default")))
#else
#define OUSTER_API_CLASS
#define OUSTER_API_FUNCTION
#define OUSTER_API_VAR
#endif
#define OUSTER_API_IGNORE
#define OUSTER_API_DEFAULT
#define OUSTER_API_INTERFACE OUSTER_API_CLASS
