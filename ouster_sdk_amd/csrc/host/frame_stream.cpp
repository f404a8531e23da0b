// frame_stream.cpp -- see include/ouster/hip/frame_stream.h
#include "ouster/hip/frame_stream.h"

#include <algorithm>
#include <cstdint>
#include <deque>

#include <hip/hip_runtime_api.h>

#include <cstring>
#include <stdexcept>

#include "host_internal.h"

namespace ouster {
namespace sdk {
namespace hip {

namespace {
void ok(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("ouster_hip: ") + what + ": " + hipGetErrorString(e));
}
struct Pinned {
    void* p = nullptr;
    size_t n = 0;
    void alloc(size_t bytes) {
        n = bytes;
        if (bytes) ok(hipHostMalloc(&p, bytes, hipHostMallocDefault), "hipHostMalloc");
    }
    ~Pinned() {
        if (p) (void)hipHostFree(p);
    }
};
}  // namespace

struct FrameStream::Slot {
    std::unique_ptr<DeviceFrameBatch> batch;
    Pinned packets;                      // [frames][slots][stride]
    Pinned xyz[2], ts, mid, status;
    std::map<std::string, Pinned> planes, destaggered;
    Pinned dw_pts, dw_off, dw_fi, dw_ci, dw_ts;   // the compacting route: capacity H * W points per frame
    hipEvent_t e_h2d = nullptr, e_dec = nullptr, e_done = nullptr, e_pts = nullptr;
    bool points_requested = false;       // the compacted list's copy has been queued (its size is known once the offsets are here)
    uint32_t filled = 0;                 // frames copied into `packets`
    bool in_flight = false;
    uint64_t first_frame = 0;
    uint32_t n_frames = 0;
};

struct FrameStream::SensorLane {
    std::unique_ptr<core::FrameBatcher> splitter;
    std::unique_ptr<core::LidarFrame> frame;
    std::deque<std::vector<std::vector<uint8_t>>> released;   // frames, each the packets the batcher handed over
};

FrameStream::FrameStream(const std::vector<core::SensorInfo>& sensors, const StreamOptions& options,
                         Callback on_batch)
    : ctx_(std::make_shared<Context>(options.device >= 0 ? options.device : current_device())),
      opt_(options), cb_(std::move(on_batch)), sensors_(sensors) {
    ScopedContext on_my_context(ctx_);
    if (sensors.empty()) throw std::invalid_argument("FrameStream: no sensors");
    if (opt_.frames_per_batch == 0 || opt_.frames_per_batch % sensors.size() != 0)
        throw std::invalid_argument("FrameStream: frames_per_batch must be a multiple of the sensor count");
    if (opt_.batches_in_flight == 0) throw std::invalid_argument("FrameStream: batches_in_flight must be > 0");
    opt_.outputs.all_slots = true;
    if (opt_.download_xyz) opt_.outputs.xyz = true;
    const bool compact = opt_.dewarp_max_range >= opt_.dewarp_min_range;
    if (compact) {   // the decode leaves the gate's per-column counts behind (no counting pass), and the LUTs are needed
        opt_.outputs.xyz = true;
        opt_.outputs.gate_min_range = opt_.dewarp_min_range;
        opt_.outputs.gate_max_range = opt_.dewarp_max_range;
    }
    opt_.outputs.context = ctx_;
    hipStream_t a = nullptr, b = nullptr;
    ok(hipStreamCreateWithFlags(&a, hipStreamNonBlocking), "hipStreamCreate");
    ok(hipStreamCreateWithFlags(&b, hipStreamNonBlocking), "hipStreamCreate");
    stream_h2d_ = a;
    stream_d2h_ = b;
    for (uint32_t i = 0; i < opt_.batches_in_flight; ++i) {
        auto s = std::make_unique<Slot>();
        s->batch = std::make_unique<DeviceFrameBatch>(sensors, opt_.frames_per_batch, opt_.outputs);
        DeviceFrameBatch& bt = *s->batch;
        const size_t n = opt_.frames_per_batch;
        s->packets.alloc(n * bt.slots_per_frame() * bt.packet_stride());
        std::memset(s->packets.p, 0, s->packets.n);
        if (opt_.download_xyz)
            for (int k = 0; k < 2; ++k)
                if (bt.xyz_device(k)) s->xyz[k].alloc(n * bt.xyz_bytes_per_frame());
        for (const auto& name : opt_.download_planes) s->planes[name].alloc(n * bt.plane_bytes_per_frame(name));
        for (const auto& name : opt_.download_destaggered) {
            (void)bt.destaggered_device(name);  // throws if it is not produced
            s->destaggered[name].alloc(n * bt.plane_bytes_per_frame(name));
        }
        if (compact) {
            const size_t cap = n * bt.h() * bt.w();
            s->dw_pts.alloc(cap * (opt_.outputs.xyz_f64 ? 24 : 12));
            s->dw_off.alloc((n + 1) * 8);
            if (opt_.dewarp_provenance) {
                s->dw_fi.alloc(cap * 4);
                s->dw_ci.alloc(cap * 4);
                s->dw_ts.alloc(cap * 8);
            }
        }
        if (opt_.download_headers) {
            s->ts.alloc(n * bt.w() * 8);
            s->mid.alloc(n * bt.w() * 2);
            s->status.alloc(n * bt.w() * 4);
        }
        ok(hipEventCreateWithFlags(&s->e_h2d, hipEventDisableTiming), "hipEventCreate");
        ok(hipEventCreateWithFlags(&s->e_dec, hipEventDisableTiming), "hipEventCreate");
        ok(hipEventCreateWithFlags(&s->e_done, hipEventDisableTiming), "hipEventCreate");
        ok(hipEventCreateWithFlags(&s->e_pts, hipEventDisableTiming), "hipEventCreate");
        slots_.push_back(std::move(s));
    }
}

FrameStream::~FrameStream() {
    ScopedContext on_my_context(ctx_);
    for (auto& s : slots_) {
        if (s->in_flight) (void)hipEventSynchronize(s->e_done);
        for (hipEvent_t e : {s->e_h2d, s->e_dec, s->e_done, s->e_pts})
            if (e) (void)hipEventDestroy(e);
    }
    if (stream_h2d_) (void)hipStreamDestroy(static_cast<hipStream_t>(stream_h2d_));
    if (stream_d2h_) (void)hipStreamDestroy(static_cast<hipStream_t>(stream_d2h_));
}

void FrameStream::push_frame(const std::vector<const uint8_t*>& lidar_packets) {
    ScopedContext on_my_context(ctx_);
    poll_points();
    Slot& s = *slots_[cur_];
    if (s.in_flight) deliver(s);  // every buffer set busy: the oldest batch has to come home first
    DeviceFrameBatch& bt = *s.batch;
    if (s.filled == 0) s.first_frame = pushed_;
    uint8_t* base = static_cast<uint8_t*>(s.packets.p) +
                    static_cast<size_t>(s.filled) * bt.slots_per_frame() * bt.packet_stride();
    // home slots: packet p of the frame in slot p, so the decode pass needs no mapping even when
    // packets were lost (holes stay zero = invalid columns); a packet sent twice is merged column by
    // column (DeviceFrameBatch::stage_packet), a packet outside the frame is dropped
    std::vector<bool> have(bt.slots_per_frame(), false);
    for (const uint8_t* pkt : lidar_packets) {
        const int p = bt.home_slot(pkt);
        if (p < 0) continue;
        bt.stage_packet(base + static_cast<size_t>(p) * bt.packet_stride(), have[static_cast<size_t>(p)], pkt);
        have[static_cast<size_t>(p)] = true;
    }
    for (size_t p = 0; p < have.size(); ++p)
        if (!have[p]) std::memset(base + p * bt.packet_stride(), 0, bt.packet_stride());
    ++s.filled;
    ++pushed_;
    if (s.filled == opt_.frames_per_batch) submit(s);
}

void FrameStream::push_packet(const core::Packet& lidar_packet) {
    if (sensors_.size() != 1)
        throw std::logic_error("FrameStream::push_packet serves single-sensor streams; push_frame otherwise");
    if (!splitter_) {
        splitter_ = std::make_unique<core::FrameBatcher>(sensors_[0]);
        splitter_frame_ = std::make_unique<core::LidarFrame>(sensors_[0]);
        splitter_->set_packet_sink([this](const std::vector<const uint8_t*>& packets) { push_frame(packets); });
    }
    (void)splitter_->batch(lidar_packet, *splitter_frame_);
}

void FrameStream::push_packet(size_t sensor, const core::Packet& lidar_packet) {
    if (sensor >= sensors_.size()) throw std::out_of_range("FrameStream::push_packet: no such sensor");
    if (lanes_.empty()) {
        for (const core::SensorInfo& info : sensors_) {
            auto lane = std::make_unique<SensorLane>();
            lane->splitter = std::make_unique<core::FrameBatcher>(info);
            lane->frame = std::make_unique<core::LidarFrame>(info);
            SensorLane* raw = lane.get();
            const size_t size = core::PacketFormat(info).lidar_packet_size;
            // the sink's pointers are only good during the call: keep the bytes until the frame's tick
            lane->splitter->set_packet_sink([raw, size](const std::vector<const uint8_t*>& packets) {
                std::vector<std::vector<uint8_t>> fr;
                fr.reserve(packets.size());
                for (const uint8_t* p : packets) fr.emplace_back(p, p + size);
                raw->released.push_back(std::move(fr));
            });
            lanes_.push_back(std::move(lane));
        }
    }
    (void)lanes_[sensor]->splitter->batch(lidar_packet, *lanes_[sensor]->frame);
    emit_ticks(false);
}

void FrameStream::emit_ticks(bool flush) {
    for (;;) {
        size_t shortest = SIZE_MAX, longest = 0;
        for (const auto& lane : lanes_) {
            shortest = std::min(shortest, lane->released.size());
            longest = std::max(longest, lane->released.size());
        }
        if (longest == 0) return;
        // a tick goes out when every sensor has a frame for it, when somebody is too far ahead, or at the end
        if (shortest == 0 && longest < max_skew_ && !flush) return;
        for (const auto& lane : lanes_) {
            std::vector<const uint8_t*> ptrs;
            if (!lane->released.empty())
                for (const auto& pkt : lane->released.front()) ptrs.push_back(pkt.data());
            push_frame(ptrs);   // no packets: an empty frame keeps the sensor order of the batch
            if (!lane->released.empty()) lane->released.pop_front();
        }
    }
}

void FrameStream::submit(Slot& s) {
    ScopedContext on_my_context(ctx_);
    DeviceFrameBatch& bt = *s.batch;
    const size_t per_frame = static_cast<size_t>(bt.slots_per_frame()) * bt.packet_stride();
    if (s.filled < opt_.frames_per_batch)  // partial batch: the unused frames decode to "empty"
        std::memset(static_cast<uint8_t*>(s.packets.p) + s.filled * per_frame, 0,
                    (opt_.frames_per_batch - s.filled) * per_frame);
    auto h2d = static_cast<hipStream_t>(stream_h2d_);
    auto d2h = static_cast<hipStream_t>(stream_d2h_);
    auto comp = static_cast<hipStream_t>(ctx_->stream());
    ok(hipMemcpyAsync(bt.packets_device(), s.packets.p, s.packets.n, hipMemcpyHostToDevice, h2d), "H2D");
    ok(hipEventRecord(s.e_h2d, h2d), "hipEventRecord");
    ok(hipStreamWaitEvent(comp, s.e_h2d, 0), "hipStreamWaitEvent");
    bt.decode();
    if (s.dw_off.p) bt.dewarp_async(opt_.dewarp_min_range, opt_.dewarp_max_range, opt_.dewarp_provenance);
    ok(hipEventRecord(s.e_dec, comp), "hipEventRecord");
    ok(hipStreamWaitEvent(d2h, s.e_dec, 0), "hipStreamWaitEvent");
    const size_t n = opt_.frames_per_batch;
    for (int k = 0; k < 2; ++k)
        if (s.xyz[k].p)
            ok(hipMemcpyAsync(s.xyz[k].p, bt.xyz_device(k), n * bt.xyz_bytes_per_frame(),
                              hipMemcpyDeviceToHost, d2h), "D2H xyz");
    for (auto& kv : s.planes)
        ok(hipMemcpyAsync(kv.second.p, bt.plane_device(kv.first), kv.second.n, hipMemcpyDeviceToHost, d2h), "D2H plane");
    for (auto& kv : s.destaggered)
        ok(hipMemcpyAsync(kv.second.p, bt.destaggered_device(kv.first), kv.second.n, hipMemcpyDeviceToHost, d2h),
           "D2H destaggered");
    if (opt_.download_headers) {
        ok(hipMemcpyAsync(s.ts.p, bt.timestamp_device(), s.ts.n, hipMemcpyDeviceToHost, d2h), "D2H ts");
        ok(hipMemcpyAsync(s.mid.p, bt.measurement_id_device(), s.mid.n, hipMemcpyDeviceToHost, d2h), "D2H m_id");
        ok(hipMemcpyAsync(s.status.p, bt.status_device(), s.status.n, hipMemcpyDeviceToHost, d2h), "D2H status");
    }
    // the compacted list: only its offsets now; how many points there are is known when they have arrived (deliver())
    if (s.dw_off.p)
        ok(hipMemcpyAsync(s.dw_off.p, bt.dewarped_offsets_device(), s.dw_off.n, hipMemcpyDeviceToHost, d2h), "D2H offsets");
    ok(hipEventRecord(s.e_done, d2h), "hipEventRecord");
    s.in_flight = true;
    s.points_requested = false;
    s.n_frames = s.filled;
    s.filled = 0;
    cur_ = (cur_ + 1) % slots_.size();
}

void FrameStream::deliver(Slot& s) {
    ScopedContext on_my_context(ctx_);
    ok(hipEventSynchronize(s.e_done), "hipEventSynchronize");
    s.in_flight = false;
    BatchResult r;
    r.first_frame = s.first_frame;
    r.n_frames = s.n_frames;
    r.h = s.batch->h();
    r.w = s.batch->w();
    for (int k = 0; k < 2; ++k) r.xyz[k] = s.xyz[k].p;
    for (auto& kv : s.planes) r.planes[kv.first] = kv.second.p;
    for (auto& kv : s.destaggered) r.destaggered[kv.first] = kv.second.p;
    r.timestamp = static_cast<const uint64_t*>(s.ts.p);
    r.measurement_id = static_cast<const uint16_t*>(s.mid.p);
    r.status = static_cast<const uint32_t*>(s.status.p);
    if (s.dw_off.p) {
        request_points(s);                                   // no-op when poll_points() already did it
        ok(hipEventSynchronize(s.e_pts), "hipEventSynchronize");
        const uint64_t* off = static_cast<const uint64_t*>(s.dw_off.p);
        r.points = s.dw_pts.p;
        r.n_points = off[opt_.frames_per_batch];
        r.frame_offsets = off;
        if (opt_.dewarp_provenance) {
            r.point_frame_idxs = static_cast<const uint32_t*>(s.dw_fi.p);
            r.point_col_idxs = static_cast<const uint32_t*>(s.dw_ci.p);
            r.point_timestamps_ns = static_cast<const uint64_t*>(s.dw_ts.p);
        }
    }
    delivered_ += s.n_frames;
    if (cb_) cb_(r);
}

// The compacted list of a batch: exactly the kept points cross the link.  How many there are is known when the batch's offsets
// have arrived (e_done); the copy is queued as soon as somebody notices -- every push looks (poll_points), so it normally runs
// under the host's staging of later frames -- and waited for when the batch is delivered.
void FrameStream::request_points(Slot& s) {
    if (s.points_requested) return;
    ok(hipEventSynchronize(s.e_done), "hipEventSynchronize");
    const uint64_t total = static_cast<const uint64_t*>(s.dw_off.p)[opt_.frames_per_batch];
    auto d2h = static_cast<hipStream_t>(stream_d2h_);
    DeviceFrameBatch& bt = *s.batch;
    if (total) {
        ok(hipMemcpyAsync(s.dw_pts.p, bt.dewarped_points_device(), total * (opt_.outputs.xyz_f64 ? 24 : 12), hipMemcpyDeviceToHost, d2h),
           "D2H points");
        if (opt_.dewarp_provenance) {
            ok(hipMemcpyAsync(s.dw_fi.p, bt.dewarped_frame_idxs_device(), total * 4, hipMemcpyDeviceToHost, d2h), "D2H frame idx");
            ok(hipMemcpyAsync(s.dw_ci.p, bt.dewarped_col_idxs_device(), total * 4, hipMemcpyDeviceToHost, d2h), "D2H col idx");
            ok(hipMemcpyAsync(s.dw_ts.p, bt.dewarped_timestamps_device(), total * 8, hipMemcpyDeviceToHost, d2h), "D2H timestamps");
        }
    }
    ok(hipEventRecord(s.e_pts, d2h), "hipEventRecord");
    s.points_requested = true;
}

void FrameStream::poll_points() {
    for (auto& sp : slots_) {
        Slot& s = *sp;
        if (!s.in_flight || !s.dw_off.p || s.points_requested) continue;
        const hipError_t q = hipEventQuery(s.e_done);
        if (q == hipSuccess) request_points(s);
        else if (q != hipErrorNotReady) ok(q, "hipEventQuery");
        else (void)hipGetLastError();
    }
}

void FrameStream::finish() {
    ScopedContext on_my_context(ctx_);
    if (!lanes_.empty()) emit_ticks(true);   // frames some sensors released after the last complete tick
    if (slots_[cur_]->filled) {
        if (slots_[cur_]->in_flight) deliver(*slots_[cur_]);  // cannot happen (filled implies free), kept for safety
        submit(*slots_[cur_]);
    }
    // oldest first: the slot after the last submitted one
    for (size_t i = 0; i < slots_.size(); ++i) {
        Slot& s = *slots_[(cur_ + i) % slots_.size()];
        if (s.in_flight) deliver(s);
    }
}

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
