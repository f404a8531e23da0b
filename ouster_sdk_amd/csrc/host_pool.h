// host_pool.h -- counted allocation calls and the pinned host pool (internal to libouster_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>

namespace ouster_hip_dev {

struct AllocCounters {
    std::atomic<uint64_t> device_allocs{0}, device_frees{0}, pinned_allocs{0}, pinned_frees{0}, pool_requests{0}, pool_hits{0};
};
AllocCounters& alloc_counters();

// every hipMalloc / hipFree of the library goes through these two (ouster_hip_alloc_stats)
inline hipError_t counted_malloc(void** p, size_t bytes) {
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) alloc_counters().device_allocs.fetch_add(1, std::memory_order_relaxed);
    return e;
}
inline void counted_free(void* p) {
    if (!p) return;
    (void)hipFree(p);
    alloc_counters().device_frees.fetch_add(1, std::memory_order_relaxed);
}

}  // namespace ouster_hip_dev
