// sharded_batch.h -- one batch of frames over several GPUs, in one C++ process.
//
// Frames are independent: a batch is cut into contiguous blocks of frames, one ouster::sdk::hip::DeviceFrameBatch (its
// own context = its own HIP stream) per GPU, and there is NO collective on the data path.  The only exchange step --
// when the packets arrive on one GPU and the clouds are wanted on one GPU -- is a scatter of raw packet buffers and a
// gather of XYZ results: point-to-point copies over xGMI (hipMemcpyPeerAsync, every shard on its own stream, so the
// root's copies to its peers leave on their own links concurrently and each GPU receives only its shard).
// This is the C++ face of ouster_sdk_amd/parallel.py (shard_range / scatter_frames / gather_frames over RCCL) for callers
// in the reference's one-object-per-sensor-stream pattern (ouster_sensor/src/sensor_frame_set_source.cpp:177-223) that
// have no Python launcher around them.
#pragma once

#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

#include "ouster/hip/device_batch.h"

namespace ouster {
namespace sdk {
namespace hip {

/** Frames [first, second) of an n_frames batch that shard `rank` of `world` owns: contiguous blocks, sizes differing
 *  by at most one frame (the same rule as ouster_sdk_amd.parallel.shard_range). */
std::pair<uint32_t, uint32_t> shard_range(uint32_t n_frames, int rank, int world);

class ShardedBatch {
   public:
    /** `devices`: the GPU of every shard (empty: one shard per visible GPU; a GPU may be listed more than once, e.g. to
     *  exercise the exchange on a one-GPU box).  The staging buffers of scatter() / gather_xyz() live on `root_device`
     *  (-1: devices[0]).  Frame f of the batch uses sensor f % sensors.size(), whichever shard it lands in. */
    ShardedBatch(const std::vector<core::SensorInfo>& sensors, uint32_t n_frames, const BatchOptions& options,
                 std::vector<int> devices = {}, int root_device = -1);
    ~ShardedBatch();
    ShardedBatch(const ShardedBatch&) = delete;
    ShardedBatch& operator=(const ShardedBatch&) = delete;

    uint32_t n_frames() const { return n_frames_; }
    int n_shards() const { return static_cast<int>(shards_.size()); }
    int root_device() const { return root_; }
    std::pair<uint32_t, uint32_t> range(int shard) const { return shard_range(n_frames_, shard, n_shards()); }
    DeviceFrameBatch& shard(int i) { return *shards_.at(i); }
    /** Shard and local frame index of batch frame f. */
    std::pair<int, uint32_t> locate(uint32_t frame) const;

    /** Stage one frame's packets in the ROOT GPU's copy of the batch (home slots, like DeviceFrameBatch). */
    void upload_frame_packets(uint32_t frame, const std::vector<const uint8_t*>& packets);
    /** Root GPU -> every shard's packet buffer (asynchronous, one peer copy per shard on the shard's stream). */
    void scatter();
    /** decode() of every shard (asynchronous: the shards' streams run concurrently). */
    void decode();
    /** Every shard's XYZ cloud `return_index` -> the root GPU's [n_frames][H*W][3] buffer (asynchronous). */
    void gather_xyz(int return_index);
    void sync();
    /** The gathered cloud on the root GPU / one frame of it on the host (synchronous). */
    void* xyz_root(int return_index) { return d_xyz_root_[return_index].data(); }
    void download_xyz_root(int return_index, uint32_t frame, void* host);
    /** Milliseconds the last scatter() / decode() / gather_xyz() took on the slowest shard (HIP events on the shards'
     *  streams; valid after sync()). */
    double last_scatter_ms() const { return ms_[0]; }
    double last_decode_ms() const { return ms_[1]; }
    double last_gather_ms() const { return ms_[2]; }

   private:
    struct Events;
    void measure();
    uint32_t n_frames_;
    int root_;
    std::vector<int> devices_;
    std::vector<std::unique_ptr<DeviceFrameBatch>> shards_;
    std::shared_ptr<Context> root_ctx_;
    DeviceBuffer d_packets_root_;
    DeviceBuffer d_xyz_root_[2];
    size_t frame_packet_bytes_ = 0;
    std::unique_ptr<Events> ev_;
    double ms_[3] = {0, 0, 0};
};

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
