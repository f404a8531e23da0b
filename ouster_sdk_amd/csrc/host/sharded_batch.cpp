// sharded_batch.cpp -- see include/ouster/hip/sharded_batch.h
#include "ouster/hip/sharded_batch.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

#include "host_internal.h"

namespace ouster {
namespace sdk {
namespace hip {

using namespace core;

std::pair<uint32_t, uint32_t> shard_range(uint32_t n_frames, int rank, int world) {
    if (world <= 0 || rank < 0 || rank >= world) throw std::invalid_argument("shard_range: bad rank / world");
    const uint32_t base = n_frames / static_cast<uint32_t>(world), rem = n_frames % static_cast<uint32_t>(world);
    const uint32_t r = static_cast<uint32_t>(rank);
    const uint32_t begin = r * base + std::min(r, rem);
    return {begin, begin + base + (r < rem ? 1u : 0u)};
}

struct ShardedBatch::Events {
    // [shard][phase 0 scatter / 1 decode / 2 gather][begin, end]
    std::vector<hipEvent_t> ev;
    std::vector<uint8_t> used;
    int n = 0;
    explicit Events(int shards) : ev(static_cast<size_t>(shards) * 6, nullptr), used(static_cast<size_t>(shards) * 3, 0), n(shards) {}
    ~Events() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    hipEvent_t& at(int shard, int phase, int which) { return ev[(static_cast<size_t>(shard) * 3 + phase) * 2 + which]; }
};

static void hip_try(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("ouster_hip: ") + what + ": " + hipGetErrorString(e));
}

ShardedBatch::ShardedBatch(const std::vector<SensorInfo>& sensors, uint32_t n_frames, const BatchOptions& options,
                           std::vector<int> devices, int root_device)
    : n_frames_(n_frames), devices_(std::move(devices)) {
    if (n_frames == 0) throw std::invalid_argument("ShardedBatch: n_frames must be > 0");
    if (sensors.empty()) throw std::invalid_argument("ShardedBatch: at least one sensor");
    if (options.context) throw std::invalid_argument("ShardedBatch: every shard owns its context (BatchOptions::context must be empty)");
    if (devices_.empty())
        for (int d = 0; d < device_count(); ++d) devices_.push_back(d);
    if (devices_.empty()) throw std::runtime_error("ouster_hip: no HIP device visible");
    if (devices_.size() > n_frames) devices_.resize(n_frames);   // never an empty shard
    root_ = root_device >= 0 ? root_device : devices_[0];
    const int world = static_cast<int>(devices_.size());
    for (int i = 0; i < world; ++i) {
        const auto r = shard_range(n_frames_, i, world);
        // frame f of the batch uses sensor f % S: rotate the list so that the shard's local frame 0 finds its own
        std::vector<SensorInfo> rot(sensors.size());
        for (size_t k = 0; k < sensors.size(); ++k) rot[k] = sensors[(r.first + k) % sensors.size()];
        BatchOptions o = options;
        o.device = devices_[i];
        shards_.push_back(std::make_unique<DeviceFrameBatch>(rot, r.second - r.first, o));
        // direct peer copies over xGMI where the topology allows them (a refusal only means staged copies)
        if (devices_[i] != root_) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices_[i], root_) == hipSuccess && can) {
                int before = 0;
                const bool restore = hipGetDevice(&before) == hipSuccess;
                (void)hipSetDevice(devices_[i]);
                (void)hipDeviceEnablePeerAccess(root_, 0);
                (void)hipSetDevice(root_);
                (void)hipDeviceEnablePeerAccess(devices_[i], 0);
                (void)hipGetLastError();   // "already enabled" is fine
                if (restore) (void)hipSetDevice(before);   // the calling thread keeps its device
            }
        }
    }
    frame_packet_bytes_ = static_cast<size_t>(shards_[0]->slots_per_frame()) * shards_[0]->packet_stride();
    root_ctx_ = std::make_shared<Context>(root_);
    {
        ScopedContext on_root(root_ctx_);
        d_packets_root_.resize(frame_packet_bytes_ * n_frames_);
        d_packets_root_.fill(0);
    }
    ev_ = std::make_unique<Events>(world);
}

ShardedBatch::~ShardedBatch() {
    try {
        sync();
    } catch (...) {
    }
}

std::pair<int, uint32_t> ShardedBatch::locate(uint32_t frame) const {
    if (frame >= n_frames_) throw std::out_of_range("ShardedBatch: frame index");
    for (int i = 0; i < n_shards(); ++i) {
        const auto r = range(i);
        if (frame < r.second) return {i, frame - r.first};
    }
    throw std::out_of_range("ShardedBatch: frame index");
}

void ShardedBatch::upload_frame_packets(uint32_t frame, const std::vector<const uint8_t*>& packets) {
    if (frame >= n_frames_) throw std::out_of_range("ShardedBatch: frame index");
    DeviceFrameBatch& any = *shards_[0];
    const size_t stride = any.packet_stride();
    std::vector<uint8_t> staging(frame_packet_bytes_, 0);
    std::vector<bool> have(any.slots_per_frame(), false);
    for (const uint8_t* pkt : packets) {
        const int p = any.home_slot(pkt);
        if (p < 0) continue;
        any.stage_packet(staging.data() + static_cast<size_t>(p) * stride, have[static_cast<size_t>(p)], pkt);
        have[static_cast<size_t>(p)] = true;
    }
    ScopedContext on_root(root_ctx_);
    d_packets_root_.upload(staging.data(), staging.size(), static_cast<size_t>(frame) * frame_packet_bytes_);
}

void ShardedBatch::scatter() {
    for (int i = 0; i < n_shards(); ++i) {
        const auto r = range(i);
        DeviceFrameBatch& s = *shards_[i];
        ScopedContext on_shard(s.context());
        auto st = static_cast<hipStream_t>(s.context()->stream());
        for (int k = 0; k < 2; ++k)
            if (!ev_->at(i, 0, k)) hip_try(hipEventCreate(&ev_->at(i, 0, k)), "hipEventCreate");
        hip_try(hipEventRecord(ev_->at(i, 0, 0), st), "hipEventRecord");
        hip_try(hipMemcpyPeerAsync(s.packets_device(), devices_[i],
                                   static_cast<const uint8_t*>(d_packets_root_.data()) + static_cast<size_t>(r.first) * frame_packet_bytes_,
                                   root_, static_cast<size_t>(r.second - r.first) * frame_packet_bytes_, st),
                "hipMemcpyPeerAsync(scatter)");
        hip_try(hipEventRecord(ev_->at(i, 0, 1), st), "hipEventRecord");
        ev_->used[static_cast<size_t>(i) * 3 + 0] = 1;
        s.assume_all_slots_filled();
    }
}

void ShardedBatch::decode() {
    for (int i = 0; i < n_shards(); ++i) {
        DeviceFrameBatch& s = *shards_[i];
        ScopedContext on_shard(s.context());
        auto st = static_cast<hipStream_t>(s.context()->stream());
        for (int k = 0; k < 2; ++k)
            if (!ev_->at(i, 1, k)) hip_try(hipEventCreate(&ev_->at(i, 1, k)), "hipEventCreate");
        hip_try(hipEventRecord(ev_->at(i, 1, 0), st), "hipEventRecord");
        s.decode();
        hip_try(hipEventRecord(ev_->at(i, 1, 1), st), "hipEventRecord");
        ev_->used[static_cast<size_t>(i) * 3 + 1] = 1;
    }
}

void ShardedBatch::gather_xyz(int k) {
    if (k < 0 || k > 1) throw std::out_of_range("ShardedBatch: return index");
    const size_t per_frame = shards_[0]->xyz_bytes_per_frame();
    {
        ScopedContext on_root(root_ctx_);
        d_xyz_root_[k].resize(per_frame * n_frames_);
    }
    for (int i = 0; i < n_shards(); ++i) {
        const auto r = range(i);
        DeviceFrameBatch& s = *shards_[i];
        ScopedContext on_shard(s.context());
        auto st = static_cast<hipStream_t>(s.context()->stream());
        for (int e = 0; e < 2; ++e)
            if (!ev_->at(i, 2, e)) hip_try(hipEventCreate(&ev_->at(i, 2, e)), "hipEventCreate");
        hip_try(hipEventRecord(ev_->at(i, 2, 0), st), "hipEventRecord");
        hip_try(hipMemcpyPeerAsync(static_cast<uint8_t*>(d_xyz_root_[k].data()) + static_cast<size_t>(r.first) * per_frame, root_,
                                   s.xyz_device(k), devices_[i], static_cast<size_t>(r.second - r.first) * per_frame, st),
                "hipMemcpyPeerAsync(gather)");
        hip_try(hipEventRecord(ev_->at(i, 2, 1), st), "hipEventRecord");
        ev_->used[static_cast<size_t>(i) * 3 + 2] = 1;
    }
}

void ShardedBatch::measure() {
    for (int phase = 0; phase < 3; ++phase) {
        double worst = 0;
        bool any = false;
        for (int i = 0; i < n_shards(); ++i) {
            if (!ev_->used[static_cast<size_t>(i) * 3 + phase]) continue;
            (void)hipSetDevice(devices_[i]);
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev_->at(i, phase, 0), ev_->at(i, phase, 1)) == hipSuccess) {
                worst = std::max(worst, static_cast<double>(ms));
                any = true;
            } else {
                (void)hipGetLastError();
            }
            ev_->used[static_cast<size_t>(i) * 3 + phase] = 0;
        }
        if (any) ms_[phase] = worst;
    }
}

void ShardedBatch::sync() {
    for (auto& s : shards_) s->sync();
    if (root_ctx_) root_ctx_->sync();
    if (ev_) measure();
}

void ShardedBatch::download_xyz_root(int k, uint32_t frame, void* host) {
    if (frame >= n_frames_) throw std::out_of_range("ShardedBatch: frame index");
    sync();
    const size_t per_frame = shards_[0]->xyz_bytes_per_frame();
    ScopedContext on_root(root_ctx_);
    d_xyz_root_[k].download(host, per_frame, per_frame * frame);
}

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
