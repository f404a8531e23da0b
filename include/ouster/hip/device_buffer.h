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
    void fill(int byte_value);

   private:
    void* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
