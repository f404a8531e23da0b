// context.h -- which GPU, which stream: the C++ face of ouster_hip_ctx (include/ouster_hip.h).
//
// The reference's types are "one object per sensor stream, externally serialised, distinct objects
// usable from distinct threads" (SURVEY.md section 8b).  The GPU-backed mirror keeps that contract:
//   * every FrameBatcher / DeviceFrameBatch / FrameStream owns its OWN context (HIP stream + scratch)
//     on the device that was current when it was constructed, so two batchers on two threads never
//     share mutable state;
//   * the free functions (destagger<T>, cartesian, XYZLutT::operator(), dewarp) run on the calling
//     thread's default context for its current device (one per thread and device, created on first
//     use);
//   * set_device() selects the GPU for the calling thread, like hipSetDevice: a process that drives
//     several GPUs uses one thread (or one explicit Context) per GPU -- frames are independent, so
//     that is all multi-GPU takes below the Python / torchrun launcher.
#pragma once

#include <memory>

struct ouster_hip_ctx;

namespace ouster {
namespace sdk {
namespace hip {

/** Number of visible GPUs (0: none -- every GPU-backed call will throw std::runtime_error). */
int device_count();
/** GPU used by objects and free-function calls the calling thread makes from now on.
 *  @throw std::invalid_argument when out of range. */
void set_device(int device);
int current_device();

class Context {
   public:
    /** A fresh context (its own stream and scratch) on `device`.
     *  @throw std::runtime_error without a GPU, std::invalid_argument for a bad ordinal. */
    explicit Context(int device);
    ~Context();
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;

    int device() const { return device_; }
    ::ouster_hip_ctx* handle() const { return ctx_; }
    void* stream() const;  ///< the hipStream_t the context's work is ordered on
    void sync() const;     ///< wait for everything queued on it

    /** The calling thread's default context on its current device. */
    static std::shared_ptr<Context> current();
    /** The calling thread's default context on `device`. */
    static std::shared_ptr<Context> for_device(int device);

   private:
    ::ouster_hip_ctx* ctx_ = nullptr;
    int device_ = 0;
};

/** While alive, GPU-backed calls of the calling thread run on `ctx` (nests; restores on exit).
 *  Objects that own a context wrap their member functions in one of these. */
class ScopedContext {
   public:
    explicit ScopedContext(std::shared_ptr<Context> ctx);
    ~ScopedContext();
    ScopedContext(const ScopedContext&) = delete;
    ScopedContext& operator=(const ScopedContext&) = delete;

   private:
    std::shared_ptr<Context> prev_;
};

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
