// host_pool.hip -- the process-wide pool of page-locked host memory behind ouster_hip_host_alloc
// (include/ouster_hip.h, "host containers"), and the library's allocation counters.
//
// Why a pool: one hipHostMalloc + hipHostFree of a 6 MB cloud costs 1.1 ms on the round-6 boxes
// (tools/copybench), eight times the PCIe time of the cloud itself; a block that is handed out
// again costs a mutex and a map lookup.  Why page-locked at all: a kernel reaches such memory in
// place, so a frame-at-a-time call is ONE launch with no staging copy on either side (1 MB in + 1 MB
// out: 50 us -- the link's two directions do not overlap for a kernel --, against 64 us for copy-in /
// kernel / copy-out and 177 us for what round 5 did), a copy out of HBM into it runs at the full link
// rate (26.5 us per MB), and pages that stay resident are never first-touched again.
//
// Blocks are never returned to the system at process exit (the HIP runtime may be gone by then).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/ouster_hip.h"
#include "host_pool.h"

namespace ouster_hip_dev {
AllocCounters& alloc_counters() {
    static AllocCounters* c = new AllocCounters();  // leaked on purpose: used by static destructors of callers
    return *c;
}
}  // namespace ouster_hip_dev

using ouster_hip_dev::alloc_counters;

namespace {

std::atomic<bool> g_exiting{false};   // set by an atexit handler: from then on nothing is handed back to the HIP runtime

struct Block {
    size_t cap;   // bytes of the block (its size class)
    bool live;    // handed out (false: cached in `free_`)
};

struct Pool {
    std::mutex mu;
    std::map<uintptr_t, Block> blocks;               // every block the pool owns, by base address
    std::map<size_t, std::vector<void*>> free_;      // cached blocks by size class
    size_t live_bytes = 0, cached_bytes = 0;
    size_t cache_limit = size_t{1} << 30;            // OUSTER_HIP_HOST_POOL_MB
    int have_gpu = -1;                               // -1: not probed yet

    static size_t size_class(size_t n) {
        auto up = [](size_t x, size_t g) { return (x + g - 1) / g * g; };
        if (n <= (64u << 10)) return up(n, 4u << 10);
        if (n <= (4u << 20)) return up(n, 64u << 10);
        return up(n, 1u << 20);
    }
    bool gpu() {
        if (have_gpu < 0) {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess) {
                (void)hipGetLastError();
                n = 0;
            }
            have_gpu = n > 0;
            // registered AFTER the probe above has initialised the runtime (and with it the runtime's own exit handlers):
            // handlers run in reverse order, so this one runs before the runtime is torn down
            std::atexit([] { g_exiting.store(true); });
            if (const char* e = std::getenv("OUSTER_HIP_HOST_POOL_MB")) cache_limit = (size_t)std::atoll(e) << 20;
            if (const char* e = std::getenv("OUSTER_HIP_HOST_POOL")) if (std::atoi(e) == 0) have_gpu = 0;   // A/B: plain memory
        }
        return have_gpu > 0;
    }
};

Pool& pool() {
    static Pool* p = new Pool();  // leaked on purpose
    return *p;
}

}  // namespace

extern "C" {

void* ouster_hip_host_alloc(size_t bytes, int zero) {
    if (bytes == 0) bytes = 1;
    if (bytes < OUSTER_HIP_HOST_POOL_MIN) return zero ? std::calloc(1, bytes) : std::malloc(bytes);
    Pool& P = pool();
    void* p = nullptr;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (!P.gpu()) return zero ? std::calloc(1, bytes) : std::malloc(bytes);
        alloc_counters().pool_requests.fetch_add(1, std::memory_order_relaxed);
        cap = Pool::size_class(bytes);
        auto it = P.free_.find(cap);
        if (it != P.free_.end() && !it->second.empty()) {
            p = it->second.back();
            it->second.pop_back();
            P.blocks[(uintptr_t)p].live = true;
            P.cached_bytes -= cap;
            P.live_bytes += cap;
            alloc_counters().pool_hits.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (!p) {
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess || !p) {
            (void)hipGetLastError();  // out of lockable memory: plain memory still works (the *_host calls then stage)
            return zero ? std::calloc(1, bytes) : std::malloc(bytes);
        }
        alloc_counters().pinned_allocs.fetch_add(1, std::memory_order_relaxed);
        std::lock_guard<std::mutex> g(P.mu);
        P.blocks[(uintptr_t)p] = Block{cap, true};
        P.live_bytes += cap;
    }
    if (zero) std::memset(p, 0, bytes);
    return p;
}

void ouster_hip_host_free(void* p) {
    if (!p) return;
    Pool& P = pool();
    bool release = false;
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.blocks.find((uintptr_t)p);
        if (it == P.blocks.end()) {
            std::free(p);  // a small or a fallback allocation
            return;
        }
        const size_t cap = it->second.cap;
        P.live_bytes -= cap;
        if (P.cached_bytes + cap <= P.cache_limit) {
            it->second.live = false;
            P.free_[cap].push_back(p);
            P.cached_bytes += cap;
        } else {
            P.blocks.erase(it);
            release = true;
        }
    }
    if (release && !g_exiting.load()) {   // a container freed by a static destructor at exit: the runtime may be gone
        (void)hipHostFree(p);
        alloc_counters().pinned_frees.fetch_add(1, std::memory_order_relaxed);
    }
}

int ouster_hip_host_is_pinned(const void* p, size_t bytes) {
    if (!p) return 0;
    Pool& P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    if (P.blocks.empty()) return 0;
    auto it = P.blocks.upper_bound((uintptr_t)p);
    if (it == P.blocks.begin()) return 0;
    --it;
    return it->second.live && (uintptr_t)p + bytes <= it->first + it->second.cap;
}

void ouster_hip_host_pool_trim(size_t keep_bytes) {
    Pool& P = pool();
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> g(P.mu);
        for (auto it = P.free_.rbegin(); it != P.free_.rend() && P.cached_bytes > keep_bytes; ++it)   // biggest classes first
            while (!it->second.empty() && P.cached_bytes > keep_bytes) {
                void* p = it->second.back();
                it->second.pop_back();
                P.blocks.erase((uintptr_t)p);
                P.cached_bytes -= it->first;
                drop.push_back(p);
            }
    }
    for (void* p : drop) {
        if (g_exiting.load()) break;
        (void)hipHostFree(p);
        alloc_counters().pinned_frees.fetch_add(1, std::memory_order_relaxed);
    }
}

void ouster_hip_alloc_stats_read(ouster_hip_alloc_stats* out) {
    if (!out) return;
    auto& c = alloc_counters();
    out->device_allocs = c.device_allocs.load();
    out->device_frees = c.device_frees.load();
    out->pinned_allocs = c.pinned_allocs.load();
    out->pinned_frees = c.pinned_frees.load();
    out->pool_requests = c.pool_requests.load();
    out->pool_hits = c.pool_hits.load();
    Pool& P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    out->pool_live_bytes = P.live_bytes;
    out->pool_cached_bytes = P.cached_bytes;
}

}  // extern "C"
