// frame_stream.h -- streaming form of the hot path for callers whose packets arrive on the HOST
// (pcap replay, live sensors: PcapFrameSetSource / SensorFrameSetSource in the reference).
//
// Frames are collected into batches in pinned host memory; each batch then travels
//     pinned packets --H2D copy stream--> HBM --decode (ctx stream)--> HBM --D2H copy stream--> pinned results
// with `batches_in_flight` batches overlapping, so the link, not the host calls, bounds the rate:
// the PCIe-inclusive counterpart of DeviceFrameBatch (which assumes the data already is in HBM).
// Results are handed to a callback as pointers into pinned memory, in submission order.
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ouster/hip/context.h"
#include "ouster/hip/device_batch.h"

namespace ouster {
namespace sdk {
namespace hip {

struct StreamOptions {
    uint32_t frames_per_batch = 32;   ///< multiple of the number of sensors
    uint32_t batches_in_flight = 3;   ///< pinned + device buffer sets (>= 2 to overlap)
    BatchOptions outputs;             ///< what the GPU produces (planes / destaggered / xyz)
    bool download_xyz = true;         ///< which of it comes back to the host
    std::vector<std::string> download_planes;
    std::vector<std::string> download_destaggered;
    bool download_headers = true;     ///< timestamp / measurement_id / status per column
    int device = -1;                  ///< GPU to stream through (-1: hip::current_device())
    /// Round 6: the range-gated, COMPACTING route (core::dewarp(FrameSet, luts, min_range, max_range),
    /// impl/dewarp_impl.h:23-115) as what comes back: with max >= min every batch is also run through
    /// DeviceFrameBatch::dewarp on the device and the compacted point list (points of the first return inside the gate, frames
    /// in push order, the reference's order inside a frame; 12 B per KEPT point instead of 24 B per pixel for two dense
    /// clouds) is returned in BatchResult::points -- the stream is bound by the link from the device, and this is the route
    /// that moves fewer bytes over it (set download_xyz = false with it).  Poses are the identity: the points are in the frame
    /// of the LUT (sensor, or body with the extrinsics folded in).
    double dewarp_min_range = 0.0, dewarp_max_range = -1.0;
    bool dewarp_provenance = false;   ///< also frame / column index and timestamp per point
};

/** One finished batch; the pointers stay valid during the callback only. */
struct BatchResult {
    uint64_t first_frame = 0;         ///< index (in push order) of frame 0 of this batch
    uint32_t n_frames = 0;
    uint32_t h = 0, w = 0;
    const void* xyz[2] = {nullptr, nullptr};                 ///< [n_frames][h*w][3] float (or double)
    std::map<std::string, const void*> planes, destaggered;  ///< [n_frames][h][w] elements
    const uint64_t* timestamp = nullptr;                     ///< [n_frames][w]
    const uint16_t* measurement_id = nullptr;
    const uint32_t* status = nullptr;
    // the compacting route (StreamOptions::dewarp_*): n_points points [3] float (double with outputs.xyz_f64), the points of
    // frame i of this batch are [frame_offsets[i], frame_offsets[i + 1]); provenance arrays [n_points] when asked for
    const void* points = nullptr;
    uint64_t n_points = 0;
    const uint64_t* frame_offsets = nullptr;                 ///< [frames_per_batch + 1]
    const uint32_t* point_frame_idxs = nullptr;              ///< index within the batch
    const uint32_t* point_col_idxs = nullptr;
    const uint64_t* point_timestamps_ns = nullptr;
};

class FrameStream {
   public:
    using Callback = std::function<void(const BatchResult&)>;

    FrameStream(const std::vector<core::SensorInfo>& sensors, const StreamOptions& options,
                Callback on_batch);
    ~FrameStream();
    FrameStream(const FrameStream&) = delete;
    FrameStream& operator=(const FrameStream&) = delete;

    /** Append one frame (the lidar packets that arrived for it, any order, gaps allowed).  Submits
     *  the batch when it is full; blocks only when every buffer set is still in flight, and then
     *  delivers the oldest batch to the callback first. */
    void push_frame(const std::vector<const uint8_t*>& lidar_packets);
    /** Packet-level entry for a single-sensor stream: runs the reference's FrameBatcher state
     *  machine (frame-id boundaries, reorder cache, init-id changes) on the host and pushes every
     *  released frame.  A trailing incomplete frame is not emitted, like the reference's sources. */
    void push_packet(const core::Packet& lidar_packet);
    /** Packet-level entry for several sensors (what follows pcap::IndexedPcapReader::sensor_idx_for_current_packet): one
     *  FrameBatcher per sensor; the frames they release are pushed in TICKS -- one frame of every sensor, in sensor order,
     *  which is the frame order a multi-sensor batch expects (frame i belongs to sensor i % n).  A sensor that has released
     *  nothing while another is `max_skew_frames` frames ahead gets an empty (all-invalid) frame in that tick.
     *  @throw std::out_of_range for an unknown sensor */
    void push_packet(size_t sensor, const core::Packet& lidar_packet);
    /** Frames a sensor may run ahead of the slowest one before ticks stop waiting for it (default 2). */
    void set_max_skew_frames(size_t n) { max_skew_ = n ? n : 1; }
    /** Submit a partial batch and deliver everything still in flight. */
    void finish();

    uint64_t frames_pushed() const { return pushed_; }
    uint64_t frames_delivered() const { return delivered_; }

   private:
    struct Slot;
    void submit(Slot& s);
    void deliver(Slot& s);
    void request_points(Slot& s);   // compacting route: queue the copy of a batch's point list once its size is known
    void poll_points();

    std::shared_ptr<Context> ctx_;    // compute stream + scratch, shared by the buffer sets
    StreamOptions opt_;
    Callback cb_;
    std::vector<std::unique_ptr<Slot>> slots_;
    size_t cur_ = 0;
    uint64_t pushed_ = 0, delivered_ = 0;
    void* stream_h2d_ = nullptr;
    void* stream_d2h_ = nullptr;
    std::vector<core::SensorInfo> sensors_;
    std::unique_ptr<core::FrameBatcher> splitter_;   // push_packet only
    std::unique_ptr<core::LidarFrame> splitter_frame_;
    // push_packet(sensor, ...): per-sensor splitters and the frames (packet bytes) they released but no tick took yet
    struct SensorLane;
    std::vector<std::unique_ptr<SensorLane>> lanes_;
    size_t max_skew_ = 2;
    void emit_ticks(bool flush);
};

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
