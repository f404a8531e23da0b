// device_batch.h -- device-resident, batched form of the hot path for C++ callers.
//
// The reference API works one host LidarFrame at a time; on a GPU that costs a PCIe round
// trip per frame.  DeviceFrameBatch keeps a batch of frames in HBM end to end:
//     raw packets (device) --decode+destagger+cartesian--> planes / destaggered planes / XYZ (device)
// It is a thin owner of device buffers around ouster_hip_decode (include/ouster_hip.h); results
// can be read back selectively or handed to other device code through the raw pointers.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ouster/core/lidar_frame.h"
#include "ouster/core/xyzlut.h"
#include "ouster/hip/context.h"
#include "ouster/hip/device_buffer.h"

struct ouster_hip_format;

namespace ouster {
namespace sdk {
namespace hip {

struct BatchOptions {
    std::vector<std::string> planes;      ///< staggered planes to produce (empty: the profile's defaults)
    std::vector<std::string> destagger;   ///< planes to also produce destaggered
    bool xyz = false;                     ///< project RANGE (and RANGE2 when present)
    bool xyz_f64 = false;                 ///< XYZ element type (default float)
    bool use_extrinsics = true;           ///< fold SensorInfo::sensor_to_body into the LUT
    /// XYZ in the world frame: decode() applies every column's body_to_world pose (upload_poses(); identity
    /// until set) to the point while it is in registers -- dewarp<T>(cartesian(range), poses) of
    /// pose_util.h:38-56 without the separate pass over the cloud.
    bool xyz_world_frame = false;
    /// Treat every packet slot of every frame as filled (missing packets = zeroed slots, whose
    /// columns are invalid by their status word) instead of uploading per-frame packet counts:
    /// keeps decode() free of host synchronisation, for the streaming pipeline (FrameStream).
    bool all_slots = false;
    /// Range gate of the dewarp() that will follow decode(): when set (max >= min), decode() also counts
    /// the gated RANGE pixels per column (they are in registers there) and dewarp(min, max) with the same
    /// gate skips its counting pass over the RANGE planes.
    double gate_min_range = 0.0, gate_max_range = -1.0;
    /// On by default in its FRUGAL form (round 5; round 4 had it opt-in): batches of >= 64 frames settle WHERE their output
    /// buffers live when they are constructed, one DeviceFrameBatch::refine_placement(placement_draws, nullptr,
    /// placement_ballast_bytes) on an all-zero packet buffer (same store pattern).  With the defaults below: two more copies
    /// of the output set back to back, every buffer group kept at the fastest of its three locations -- 0.1 s and a transient
    /// 2 x output set (6.6 GB for 256 dual-return frames); a draw is skipped when it would take more than three quarters of
    /// the free device memory.  What it buys: the decode's store rate differs by 3 - 20 % between allocations of the same
    /// buffers (+8.7 % on the box of profiles/r05).  Set it to false where construction time or the transient
    /// memory matter more; placement_ballast_bytes > 0 / placement_draws = 4 is the thorough (tens of GB) form.
    /// Pointers handed out afterwards stay valid for the life of the batch.
    /// Only a batch that OWNS its context searches (`context` below left empty): the search re-times kernel variants and would
    /// otherwise reset the tuner state of other batches sharing the context.
    bool auto_placement = true;
    int placement_draws = 3;
    size_t placement_ballast_bytes = 0;
    /// Round 6, off by default (2 - 9 s and ~75 GB transient at construction): behind the search above, tune_placement(10, nullptr,
    /// 4 GB) -- ten whole output sets drawn, the memory given back -- and the group-wise search once more.  About one fresh process
    /// in three gets only slow placements for its first allocations (0.645 of the HBM roofline per decode where the same GPU
    /// gives 0.70 - 0.75) and back-to-back draws do not leave them; the group-wise search run again in the memory the whole-set
    /// draws gave back is what finds the fast groups (profiles/r06_latency/placement_notes.txt).  For pipelines
    /// that live long enough to pay for it; needs auto_placement and an owned context like the search above.
    bool placement_thorough = false;
    /// File with the kernel-variant tuner's verdicts of earlier processes (include/ouster_hip.h,
    /// ouster_hip_ctx_set_tuning_cache): a batch that finds its workload there launches the recorded variant from its first
    /// decode() and times nothing; one that has to measure appends its verdict.  Empty: the process-wide
    /// OUSTER_HIP_TUNING_CACHE, if set, else none.
    std::string tuning_cache;
    int device = -1;                      ///< GPU to work on (-1: hip::current_device() of the constructing thread)
    std::shared_ptr<Context> context;     ///< share this context (stream + scratch) instead of owning one
};

class DeviceFrameBatch {
   public:
    /** One LUT per sensor; frame f of the batch uses sensor f % sensors.size(). */
    DeviceFrameBatch(const std::vector<core::SensorInfo>& sensors, uint32_t n_frames,
                     const BatchOptions& options);
    DeviceFrameBatch(const core::SensorInfo& sensor, uint32_t n_frames, const BatchOptions& options)
        : DeviceFrameBatch(std::vector<core::SensorInfo>{sensor}, n_frames, options) {}
    ~DeviceFrameBatch();
    DeviceFrameBatch(const DeviceFrameBatch&) = delete;
    DeviceFrameBatch& operator=(const DeviceFrameBatch&) = delete;

    /** The context (GPU, stream) this batch works on: its own unless BatchOptions::context was given. */
    const std::shared_ptr<Context>& context() const { return ctx_; }
    const core::PacketFormat& packet_format() const { return pf_; }
    /** Slot of a frame's packet buffer a lidar packet belongs in: the index of the packet inside its
     *  frame, measurement_id(first column) / columns_per_packet -- its "home" slot, where the decode
     *  kernels find it without any mapping work.  -1: outside the frame (the packet is dropped, as
     *  the reference drops its columns, lidar_frame.cpp:1432-1434). */
    int home_slot(const uint8_t* lidar_packet) const {
        const uint32_t p = pf_.col_measurement_id(pf_.nth_col(0, lidar_packet)) / pf_.columns_per_packet;
        return p < slots_ ? static_cast<int>(p) : -1;
    }
    /** Put a lidar packet into a staging slot (at least lidar_packet_size bytes).  An empty slot takes the packet as it
     *  is.  A slot that already holds a packet -- the same packet sent again -- is merged the way the reference ends up
     *  after batching both (parse_by_col, lidar_frame.cpp:1422-1466): every valid, in-range column of the later packet
     *  replaces the earlier one's, the others stay; the packet-level bytes (header, footer: alert flags, timestamps) are the
     *  later packet's (batch_lidar_packet :1534-1539). */
    void stage_packet(uint8_t* slot, bool occupied, const uint8_t* lidar_packet) const;
    uint32_t n_frames() const { return n_frames_; }
    size_t packet_stride() const { return stride_; }
    uint32_t slots_per_frame() const { return slots_; }

    /** Device buffer for the raw packets, laid out [n_frames][slots_per_frame][packet_stride]. */
    uint8_t* packets_device() { return static_cast<uint8_t*>(d_packets_.data()); }
    /** Copy one frame's packets (host, each lidar_packet_size bytes) into its slots. */
    void upload_frame_packets(uint32_t frame, const std::vector<const uint8_t*>& packets);
    /** The packet buffer was filled on the device (packets in their home slots, missing ones zeroed -- e.g. by a peer
     *  copy, ShardedBatch::scatter): decode() treats every slot of every frame as present. */
    void assume_all_slots_filled() { counts_.assign(n_frames_, slots_); }

    /** Setup-time choice of WHERE the batch's buffers live.  The physical placement of a device allocation
     *  is drawn when it is made and the decode's achieved write rate differs by 10 - 20 % between draws of
     *  the same size (tools/ab/alloc_lottery.py).  A batch allocates once and is reused for
     *  the life of a pipeline, so it can afford to draw: this re-allocates the output buffers `tries` times
     *  (and the packet buffer, contents preserved, up to 6 times), times decode() into each draw and keeps
     *  the fastest.  Output contents are undefined afterwards (decode() again).  Returns the seconds per
     *  decode() of the kept draw; `all_ms` (optional) receives every draw's time in ms, outputs first.
     *  `ballast_bytes_between_draws` of device memory are held between two draws so that the draws scan the
     *  memory instead of its first few GB: which mode a draw gets follows where it lands (tools/ab/ballast.py). */
    double tune_placement(int tries = 16, std::vector<double>* all_ms = nullptr,
                          size_t ballast_bytes_between_draws = size_t{4} << 30);
    /** The cheap form, and what a large batch does by itself when it is constructed (BatchOptions::auto_placement):
     *  the lottery is mostly an interaction between a few heavy output streams (tools/ab/hybrid_sets.py: exchanging
     *  the two XYZ buffers of a slow set for those of a fast one recovers 90 % of the difference, one buffer alone
     *  nothing), and fast / slow regions of the device memory are tens of GB wide: `draws - 1` further copies of the
     *  output set are allocated `ballast_bytes_between_draws` apart, then GROUP BY GROUP -- XYZ clouds, 32-bit planes,
     *  destaggered planes, narrow planes -- decode() is timed with that group's buffers at each location and the fastest
     *  is kept (the buffers the batch has are a candidate too); everything else is freed.  Output contents are undefined
     *  afterwards; device pointers obtained BEFORE the call are invalid.  Returns seconds per decode() of what is kept;
     *  `all_ms`: the first allocation's time, then every candidate's (groups x (draws - 1)). */
    double refine_placement(int draws = 4, std::vector<double>* all_ms = nullptr,
                            size_t ballast_bytes_between_draws = size_t{8} << 30);

    /** Run the fused kernels on everything uploaded so far (asynchronous; sync() to wait). */
    void decode();
    void sync();

    /** Device pointers of the results ([n_frames][H][W] elements; xyz [n_frames][H*W][3]). */
    void* plane_device(const std::string& name);
    void* destaggered_device(const std::string& name);
    void* xyz_device(int return_index);
    uint64_t* timestamp_device() { return static_cast<uint64_t*>(d_ts_.data()); }
    uint16_t* measurement_id_device() { return static_cast<uint16_t*>(d_mid_.data()); }
    uint32_t* status_device() { return static_cast<uint32_t*>(d_status_.data()); }
    size_t plane_bytes_per_frame(const std::string& name) const;
    size_t xyz_bytes_per_frame() const { return static_cast<size_t>(h_) * w_ * 3 * (opt_.xyz_f64 ? 8 : 4); }
    size_t lidar_packet_size() const { return pf_.lidar_packet_size; }
    uint32_t h() const { return h_; }
    uint32_t w() const { return w_; }
    /** Read one frame's result back (synchronous). */
    void download_plane(const std::string& name, uint32_t frame, void* host, bool destaggered = false);
    void download_xyz(int return_index, uint32_t frame, void* host);
    void download_headers(uint32_t frame, uint64_t* timestamp, uint16_t* measurement_id, uint32_t* status);

    /** Per-column body_to_world poses of one frame (w x 16 doubles, row-major 4x4 each; identity
     *  until set), the input of dewarp(). */
    void upload_poses(uint32_t frame, const double* poses_w_by_16);
    /** Range-gated, compacting dewarp of the whole decoded batch, on the device
     *  (core::dewarp(FrameSet, luts, min_range, max_range) with provenance, pose_util.h:475-493 /
     *  impl/dewarp_impl.h:86-115): RANGE planes + status + poses -> world-frame points of type
     *  float (or double with xyz_f64), frames concatenated in index order.  Needs the RANGE plane
     *  and options.xyz (for the LUTs).  Synchronous; returns the number of points. */
    uint64_t dewarp(double min_range, double max_range, bool provenance = false);
    /** The same without the wait: the kernels are queued on the batch's stream and the per-frame offsets stay on the device
     *  (`dewarped_offsets_device()`, [n_frames + 1] u64, the last one the total) -- for pipelines that copy them out with
     *  their own stream (hip::FrameStream).  dewarped_frame_offsets() is NOT updated. */
    void dewarp_async(double min_range, double max_range, bool provenance = false);
    uint64_t* dewarped_offsets_device() { return static_cast<uint64_t*>(d_dw_off_.data()); }
    uint32_t* dewarped_frame_idxs_device() { return static_cast<uint32_t*>(d_dw_fi_.data()); }
    uint32_t* dewarped_col_idxs_device() { return static_cast<uint32_t*>(d_dw_ci_.data()); }
    uint64_t* dewarped_timestamps_device() { return static_cast<uint64_t*>(d_dw_ts_.data()); }
    /** Results of the last dewarp(): device pointers and per-frame exclusive offsets [n_frames+1]. */
    void* dewarped_points_device() { return d_dw_pts_.data(); }
    const std::vector<uint64_t>& dewarped_frame_offsets() const { return dw_offsets_; }
    /** Copy the compacted results to the host (null pointers are skipped). */
    void download_dewarped(void* points, uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns);

   private:
    std::shared_ptr<Context> ctx_;
    core::PacketFormat pf_;
    uint32_t n_frames_, h_, w_, slots_;
    size_t stride_;
    BatchOptions opt_;
    std::vector<std::pair<std::string, uint32_t>> fields_;
    ::ouster_hip_format* fmt_ = nullptr;
    std::vector<core::XYZLut> luts_;
    std::vector<int32_t> shifts_;
    std::vector<uint32_t> counts_;
    DeviceBuffer d_packets_, d_ts_, d_mid_, d_status_;
    std::map<std::string, DeviceBuffer> d_planes_, d_dst_;
    DeviceBuffer d_xyz_[2];
    int xyz_field_[2] = {-1, -1};
    DeviceBuffer d_gate_;   // u16 [n_frames][8][w] kept counts per column (gate by-product of decode)
    bool gate_valid_ = false;
    DeviceBuffer d_poses_, d_dw_pts_, d_dw_fi_, d_dw_ci_, d_dw_ts_, d_dw_off_;
    DeviceBuffer d_pose_rows_;   // float output: rows 0..2 of the poses cast to float, [n_frames][w][12]
    std::vector<uint64_t> dw_offsets_;
    bool dw_prov_ = false;
};

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
