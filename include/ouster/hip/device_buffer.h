// device_buffer.h -- RAII HBM allocation on the default context's device (host-side helper
// used by the GPU-backed host API; not part of the reference's surface).
#pragma once

#include <cstddef>
#include <cstdint>

namespace ouster {
namespace sdk {
namespace hip {

class DeviceBuffer {
   public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes);
    ~DeviceBuffer();
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&&) noexcept;
    DeviceBuffer& operator=(DeviceBuffer&&) noexcept;

    void resize(size_t bytes);  ///< grows the allocation if needed (contents undefined)
    void* data() const { return p_; }
    size_t size() const { return n_; }
    /** Synchronous copies ordered on the default context's stream. */
    void upload(const void* src, size_t bytes, size_t offset = 0);
    void download(void* dst, size_t bytes, size_t offset = 0) const;
    /** The same without the wait: complete after Context::sync() (the host memory must stay untouched until then). */
    void upload_async(const void* src, size_t bytes, size_t offset = 0);
    void download_async(void* dst, size_t bytes, size_t offset = 0) const;
    void fill(int byte_value);

   private:
    void* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

/** Allocation calls this process has made through the library (include/ouster_hip.h, ouster_hip_alloc_stats): the
 *  frame-at-a-time API allocates nothing once it has seen its shapes, and this is how a caller (or a test) checks. */
struct AllocStats {
    uint64_t device_allocs = 0, device_frees = 0, pinned_allocs = 0, pinned_frees = 0, pool_requests = 0, pool_hits = 0,
             pool_live_bytes = 0, pool_cached_bytes = 0;
};
AllocStats alloc_stats();
/** true when the GPU reaches [p, p + bytes) in place: memory of the library's containers (Field, img_t, PointCloudXYZ). */
bool is_device_accessible(const void* p, size_t bytes);

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
