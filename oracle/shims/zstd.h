/* Declarations of the four libzstd entry points the reference's thirdparty/zpng/zpng.cpp calls.
 * The image ships libzstd.so.1 without its headers; these prototypes are the library's stable C ABI
 * (written for oracle/_ref only -- test infrastructure, see oracle/Makefile). */
#ifndef ORACLE_SHIM_ZSTD_H
#define ORACLE_SHIM_ZSTD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
size_t ZSTD_compress(void* dst, size_t dst_capacity, const void* src, size_t src_size, int level);
size_t ZSTD_decompress(void* dst, size_t dst_capacity, const void* src, size_t compressed_size);
size_t ZSTD_compressBound(size_t src_size);
unsigned ZSTD_isError(size_t code);
#ifdef __cplusplus
}
#endif
#endif
