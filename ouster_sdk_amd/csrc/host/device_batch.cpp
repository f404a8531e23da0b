// device_batch.cpp -- see include/ouster/hip/device_batch.h
#include "ouster/hip/device_batch.h"

#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <map>

#include <cstring>
#include <stdexcept>

#include "host_internal.h"

namespace ouster {
namespace sdk {
namespace hip {

using namespace core;

DeviceFrameBatch::DeviceFrameBatch(const std::vector<SensorInfo>& sensors, uint32_t n_frames,
                                   const BatchOptions& options)
    : ctx_(options.context ? options.context
                           : std::make_shared<Context>(options.device >= 0 ? options.device : current_device())),
      pf_(sensors.at(0)), n_frames_(n_frames), opt_(options) {
    ScopedContext on_my_context(ctx_);
    if (n_frames == 0) throw std::invalid_argument("DeviceFrameBatch: n_frames must be > 0");
    if (!opt_.tuning_cache.empty()) check(ouster_hip_ctx_set_tuning_cache(ctx_->handle(), opt_.tuning_cache.c_str()));
    const SensorInfo& s0 = sensors[0];
    h_ = s0.format.pixels_per_column;
    w_ = s0.format.columns_per_frame;
    for (const auto& s : sensors)
        if (s.format.pixels_per_column != h_ || s.format.columns_per_frame != w_ ||
            s.format.udp_profile_lidar != s0.format.udp_profile_lidar ||
            s.format.header_type != s0.format.header_type)
            throw std::invalid_argument("DeviceFrameBatch: sensors of one batch must share a data format");
    slots_ = w_ / pf_.columns_per_packet;
    stride_ = (pf_.lidar_packet_size + 15) & ~size_t{15};

    // which planes: requested ones (must exist in the profile) or the frame defaults
    LidarFrameFieldTypes defaults = get_field_types(s0);
    std::vector<std::string> names = opt_.planes;
    if (names.empty())
        for (const auto& ft : defaults) names.push_back(ft.name);
    auto want = [&](const std::string& n) {
        for (const auto& x : names)
            if (x == n) return true;
        for (const auto& x : opt_.destagger)
            if (x == n) return true;
        return opt_.xyz && (n == ChanField::RANGE || n == ChanField::RANGE2);
    };
    std::vector<bool> nan;
    for (auto it = pf_.begin(); it != pf_.end(); ++it) {  // format order, like the batcher
        if (!want(it->first)) continue;
        uint32_t es = 0;
        bool f16 = false;
        for (const auto& ft : defaults)
            if (ft.name == it->first) {
                f16 = ft.element_type == ChanFieldType::FLOAT16;
                es = static_cast<uint32_t>(field_type_size(ft.element_type)) * (f16 ? 3 : 1);
            }
        if (!es) es = static_cast<uint32_t>(field_type_size(it->second.first)) * it->second.second;
        fields_.emplace_back(it->first, es);
        nan.push_back(f16);
    }
    ouster_hip_format_desc d;
    pf_.fill_hip_desc(w_, fields_, nan, d);
    check(ouster_hip_format_create(default_ctx(), &d, &fmt_));

    const size_t npx = static_cast<size_t>(h_) * w_;
    for (size_t i = 0; i < fields_.size(); ++i) {
        const auto& f = fields_[i];
        bool is_plane = opt_.planes.empty();
        for (const auto& x : opt_.planes) is_plane |= x == f.first;
        if (is_plane) d_planes_[f.first].resize(npx * f.second * n_frames_);
        for (const auto& x : opt_.destagger)
            if (x == f.first) d_dst_[f.first].resize(npx * f.second * n_frames_);
        if (opt_.xyz && f.first == ChanField::RANGE) xyz_field_[0] = static_cast<int>(i);
        if (opt_.xyz && f.first == ChanField::RANGE2) xyz_field_[1] = static_cast<int>(i);
    }
    for (const auto& x : opt_.destagger)
        if (!d_dst_.count(x)) throw std::invalid_argument("DeviceFrameBatch: unknown plane '" + x + "'");
    const size_t xes = opt_.xyz_f64 ? 8 : 4;
    for (int k = 0; k < 2; ++k)
        if (xyz_field_[k] >= 0) d_xyz_[k].resize(npx * 3 * xes * n_frames_);
    if (opt_.xyz)
        for (const auto& s : sensors) luts_.emplace_back(s, opt_.use_extrinsics);
    shifts_.assign(s0.format.pixel_shift_by_row.begin(), s0.format.pixel_shift_by_row.end());
    if (!opt_.destagger.empty() && shifts_.size() != h_)
        throw std::invalid_argument("image height does not match shifts size");
    d_packets_.resize(static_cast<size_t>(n_frames_) * slots_ * stride_);
    d_ts_.resize(static_cast<size_t>(n_frames_) * w_ * 8);
    d_mid_.resize(static_cast<size_t>(n_frames_) * w_ * 2);
    d_status_.resize(static_cast<size_t>(n_frames_) * w_ * 4);
    counts_.assign(n_frames_, 0);
    if (opt_.auto_placement && n_frames_ >= 64 && !options.context) {   // a shared context's tuner state is not ours to reset
        if (hipMemsetAsync(d_packets_.data(), 0, d_packets_.size(), static_cast<hipStream_t>(ctx_->stream())) != hipSuccess)
            throw std::runtime_error("ouster_hip: hipMemset(packets) failed");
        refine_placement(opt_.placement_draws, nullptr, opt_.placement_ballast_bytes);
        if (opt_.placement_thorough) {
            tune_placement(10, nullptr, size_t{4} << 30);
            refine_placement(opt_.placement_draws, nullptr, opt_.placement_ballast_bytes);
        }
    }
}

DeviceFrameBatch::~DeviceFrameBatch() {
    ScopedContext on_my_context(ctx_);
    if (fmt_) ouster_hip_format_destroy(fmt_);
}

// A packet that arrives twice is merged into its home slot in arrival order: the later copy's valid columns over the earlier's,
// the later packet's header and footer.  That is what the reference ends up with except in one corner (DESIGN.md section 5 (ii):
// a first copy that ends in invalid columns, re-sent whole before any later packet -- the reference's next_valid bookkeeping then
// zeroes the second copy's tail again, the merged slot keeps it); a caller that needs that corner, or arrival-order semantics for
// packets with rewritten ids, hands ouster_hip_decode the buffer in arrival order with packet counts (general mapping).
void DeviceFrameBatch::stage_packet(uint8_t* slot, bool occupied, const uint8_t* pkt) const {
    if (!occupied) {
        std::memcpy(slot, pkt, pf_.lidar_packet_size);
        return;
    }
    const size_t cols_end = pf_.packet_header_size + static_cast<size_t>(pf_.columns_per_packet) * pf_.col_size;
    std::memcpy(slot, pkt, pf_.packet_header_size);
    std::memcpy(slot + cols_end, pkt + cols_end, pf_.lidar_packet_size - cols_end);
    for (uint32_t ic = 0; ic < pf_.columns_per_packet; ++ic) {
        const uint8_t* col = pf_.nth_col(ic, pkt);
        if ((pf_.col_status(col) & 0x01) != 0u && pf_.col_measurement_id(col) < w_)
            std::memcpy(slot + (col - pkt), col, pf_.col_size);
    }
}

void DeviceFrameBatch::upload_frame_packets(uint32_t frame, const std::vector<const uint8_t*>& packets) {
    ScopedContext on_my_context(ctx_);
    if (frame >= n_frames_) throw std::out_of_range("DeviceFrameBatch: frame index");
    // every packet goes to its home slot (a packet sent twice is merged column by column, stage_packet); slots
    // without a packet are zero = invalid columns
    std::vector<uint8_t> staging(static_cast<size_t>(slots_) * stride_, 0);
    std::vector<bool> have(slots_, false);
    for (const uint8_t* pkt : packets) {
        const int p = home_slot(pkt);
        if (p < 0) continue;
        stage_packet(staging.data() + static_cast<size_t>(p) * stride_, have[static_cast<size_t>(p)], pkt);
        have[static_cast<size_t>(p)] = true;
    }
    d_packets_.upload(staging.data(), staging.size(), static_cast<size_t>(frame) * slots_ * stride_);
    counts_[frame] = slots_;
}

void DeviceFrameBatch::decode() {
    ScopedContext on_my_context(ctx_);
    ouster_hip_frame_out out{};
    for (size_t i = 0; i < fields_.size(); ++i) {
        auto p = d_planes_.find(fields_[i].first);
        if (p != d_planes_.end()) out.planes[i] = p->second.data();
        auto q = d_dst_.find(fields_[i].first);
        if (q != d_dst_.end()) out.destaggered[i] = q->second.data();
    }
    out.timestamp = static_cast<uint64_t*>(d_ts_.data());
    out.measurement_id = static_cast<uint16_t*>(d_mid_.data());
    out.status = static_cast<uint32_t*>(d_status_.data());
    out.xyz_dtype = opt_.xyz_f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32;
    std::vector<const ouster_hip_lut*> luts;
    for (int k = 0; k < 2; ++k) {
        out.xyz_field[k] = xyz_field_[k];
        out.xyz[k] = xyz_field_[k] >= 0 ? d_xyz_[k].data() : nullptr;
    }
    for (const auto& l : luts_) luts.push_back(l.device().handle);
    if (opt_.xyz_world_frame && opt_.xyz) {
        if (d_poses_.size() == 0) upload_poses(0, nullptr);   // identity, like a fresh LidarFrame
        out.xyz_poses = static_cast<const double*>(d_poses_.data());
    }
    gate_valid_ = false;
    if (opt_.gate_max_range >= opt_.gate_min_range) {
        int gf = -1;
        for (size_t i = 0; i < fields_.size(); ++i)
            if (fields_[i].first == ChanField::RANGE) gf = static_cast<int>(i);
        int empty = 0;
        check(ouster_hip_range_gate(opt_.gate_min_range, opt_.gate_max_range, &out.gate_min_r, &out.gate_max_r, &empty));
        if (gf >= 0 && !empty) {
            d_gate_.resize(static_cast<size_t>(n_frames_) * OUSTER_HIP_GATE_CHUNKS * w_ * 2);
            out.gate_counts = static_cast<uint16_t*>(d_gate_.data());
            out.gate_field = gf;
            gate_valid_ = true;
        }
    }
    // per-frame packet counts are only news when a frame has not been uploaded (count 0: it decodes as zeros); a batch whose
    // frames are all there says "every slot counts" with a null pointer and spares the call an upload of the counts
    const bool every_frame = std::all_of(counts_.begin(), counts_.end(), [&](uint32_t c) { return c == slots_; });
    check(ouster_hip_decode(default_ctx(), fmt_, static_cast<const uint8_t*>(d_packets_.data()), stride_,
                            slots_, (opt_.all_slots || every_frame) ? nullptr : counts_.data(), n_frames_, nullptr, &out,
                            d_dst_.empty() ? nullptr : shifts_.data(), luts.empty() ? nullptr : luts.data(),
                            static_cast<uint32_t>(luts.size())));
}

void DeviceFrameBatch::sync() { ctx_->sync(); }

double DeviceFrameBatch::tune_placement(int tries, std::vector<double>* all_ms, size_t ballast_bytes) {
    ScopedContext on_my_context(ctx_);
    auto st = static_cast<hipStream_t>(ctx_->stream());
    struct Events {   // released on every way out, decode() may throw
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev;
    if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess)
        throw std::runtime_error("ouster_hip: hipEventCreate failed");
    hipEvent_t e0 = ev.a, e1 = ev.b;
    const std::vector<uint32_t> kept_counts = counts_;
    struct RestoreCounts {   // the caller's per-frame packet counts come back whatever happens
        std::vector<uint32_t>& dst;
        const std::vector<uint32_t>& src;
        ~RestoreCounts() { dst = src; }
    } restore{counts_, kept_counts};
    counts_.assign(n_frames_, slots_);   // time the full-frame work whatever has been uploaded so far
    auto clock = [&]() {
        constexpr int launches = 12;
        decode();
        decode();
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < launches; ++i) decode();
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return static_cast<double>(ms) / launches;
    };
    struct Outputs {
        std::map<std::string, DeviceBuffer> planes, dst;
        DeviceBuffer xyz[2], ts, mid, status;
    };
    auto fresh = [&]() {
        Outputs o;
        for (const auto& kv : d_planes_) o.planes[kv.first].resize(kv.second.size());
        for (const auto& kv : d_dst_) o.dst[kv.first].resize(kv.second.size());
        for (int k = 0; k < 2; ++k)
            if (d_xyz_[k].size()) o.xyz[k].resize(d_xyz_[k].size());
        o.ts.resize(d_ts_.size());
        o.mid.resize(d_mid_.size());
        o.status.resize(d_status_.size());
        return o;
    };
    auto exchange = [&](Outputs& o) {
        std::swap(o.planes, d_planes_);
        std::swap(o.dst, d_dst_);
        for (int k = 0; k < 2; ++k) std::swap(o.xyz[k], d_xyz_[k]);
        std::swap(o.ts, d_ts_);
        std::swap(o.mid, d_mid_);
        std::swap(o.status, d_status_);
    };
    for (int i = 0; i < 16; ++i) decode();   // let the library's variant tuner settle first
    double best = clock();
    if (all_ms) all_ms->push_back(best);
    // the rejected draws stay allocated until the search is over: memory that has just been freed is what the
    // next allocation of the same size gets back, and it would be the same draw again
    std::vector<Outputs> rejected;
    std::vector<DeviceBuffer> rejected_packets;
    for (int t = 1; t < tries; ++t) {
        Outputs cand;
        try {
            cand = fresh();
        } catch (const std::exception&) {   // out of device memory: choose among the draws made so far
            (void)hipGetLastError();        // ... and do not leave the failed hipMalloc behind as the runtime's last error
            break;
        }
        exchange(cand);                  // members = candidate, cand = incumbent
        const double ms = clock();
        if (all_ms) all_ms->push_back(ms);
        if (ms < best) best = ms;        // keep the candidate
        else exchange(cand);             // put the incumbent back
        rejected.push_back(std::move(cand));
        if (ballast_bytes) {
            try {
                rejected_packets.emplace_back(ballast_bytes);
            } catch (const std::exception&) {
                (void)hipGetLastError();
                break;
            }
        }
    }
    ctx_->sync();
    rejected_packets.clear();            // the packet buffer looks for its place in what the ballast occupied
    for (int t = 1; t < std::min(tries, 6); ++t) {
        DeviceBuffer cand;
        try {
            cand.resize(d_packets_.size());
        } catch (const std::exception&) {
            break;
        }
        if (hipMemcpyAsync(cand.data(), d_packets_.data(), d_packets_.size(), hipMemcpyDeviceToDevice, st) != hipSuccess)
            throw std::runtime_error("ouster_hip: packet buffer copy failed");
        std::swap(cand, d_packets_);
        const double ms = clock();
        if (all_ms) all_ms->push_back(ms);
        if (ms < best) best = ms;
        else std::swap(cand, d_packets_);
        rejected_packets.push_back(std::move(cand));
    }
    ctx_->sync();
    rejected.clear();
    rejected_packets.clear();
    // the kernel variant was chosen on the first draw: let the tuner look again on the buffers that stay
    check(ouster_hip_ctx_set_knob(default_ctx(), "retune", 1));
    for (int i = 0; i < 24; ++i) decode();
    best = std::min(best, clock());
    return best * 1e-3;
}

double DeviceFrameBatch::refine_placement(int draws, std::vector<double>* all_ms, size_t ballast_bytes) {
    ScopedContext on_my_context(ctx_);
    auto st = static_cast<hipStream_t>(ctx_->stream());
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev;
    if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess)
        throw std::runtime_error("ouster_hip: hipEventCreate failed");
    const std::vector<uint32_t> kept_counts = counts_;
    struct RestoreCounts {
        std::vector<uint32_t>& dst;
        const std::vector<uint32_t>& src;
        ~RestoreCounts() { dst = src; }
    } restore{counts_, kept_counts};
    counts_.assign(n_frames_, slots_);   // the full-frame store pattern whatever has been uploaded so far
    auto clock = [&]() {
        constexpr int launches = 10;
        decode();
        decode();
        (void)hipEventRecord(ev.a, st);
        for (int i = 0; i < launches; ++i) decode();
        (void)hipEventRecord(ev.b, st);
        (void)hipEventSynchronize(ev.b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev.a, ev.b);
        return static_cast<double>(ms) / launches;
    };
    // the groups, heaviest first: pointers to the batch's own buffers
    std::vector<std::vector<DeviceBuffer*>> groups(4);
    for (int k = 0; k < 2; ++k)
        if (d_xyz_[k].size()) groups[0].push_back(&d_xyz_[k]);
    auto elem_of = [&](const std::string& name) {
        for (const auto& f : fields_)
            if (f.first == name) return f.second;
        return 0u;
    };
    for (auto& kv : d_planes_) groups[elem_of(kv.first) >= 4 ? 1 : 3].push_back(&kv.second);
    for (auto& kv : d_dst_) groups[2].push_back(&kv.second);
    // draws - 1 further copies of the whole output set, `ballast_bytes` of device memory apart: copies[c][g][i]
    std::vector<std::vector<std::vector<DeviceBuffer>>> copies;
    std::vector<DeviceBuffer> ballast;
    size_t set_bytes = 0;
    for (const auto& grp : groups)
        for (const DeviceBuffer* b : grp) set_bytes += b->size();
    for (int d = 1; d < draws; ++d) {
        // never take the device to its last byte: a draw needs its copy of the output set plus the ballast, and a quarter
        // of what is free stays free for whoever else uses this GPU
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
        if (set_bytes + ballast_bytes > free_b - free_b / 4) break;
        try {
            if (ballast_bytes) ballast.emplace_back(ballast_bytes);
            std::vector<std::vector<DeviceBuffer>> set(groups.size());
            for (size_t g = 0; g < groups.size(); ++g) {
                set[g].resize(groups[g].size());
                for (size_t i = 0; i < groups[g].size(); ++i) set[g][i].resize(groups[g][i]->size());
            }
            copies.push_back(std::move(set));
        } catch (const std::exception&) {
            (void)hipGetLastError();   // the failed hipMalloc must not surface as the next launch's error
            break;   // out of device memory: decide among what has been drawn
        }
    }
    for (int i = 0; i < 24; ++i) decode();   // the library's variant tuner settles first
    double best = clock();
    if (all_ms) all_ms->push_back(best);
    for (size_t g = 0; g < groups.size(); ++g) {
        if (groups[g].empty()) continue;
        for (auto& set : copies) {
            for (size_t i = 0; i < groups[g].size(); ++i) std::swap(set[g][i], *groups[g][i]);   // members = this location
            const double ms = clock();
            if (all_ms) all_ms->push_back(ms);
            if (ms < best) best = ms;                                                            // keep it (the incumbent moves into the copy)
            else for (size_t i = 0; i < groups[g].size(); ++i) std::swap(set[g][i], *groups[g][i]);
        }
    }
    ctx_->sync();
    copies.clear();
    ballast.clear();
    check(ouster_hip_ctx_set_knob(default_ctx(), "retune", 1));
    for (int i = 0; i < 24; ++i) decode();
    ctx_->sync();
    return best * 1e-3;
}

size_t DeviceFrameBatch::plane_bytes_per_frame(const std::string& name) const {
    for (const auto& f : fields_)
        if (f.first == name) return static_cast<size_t>(h_) * w_ * f.second;
    throw std::out_of_range("DeviceFrameBatch: unknown plane '" + name + "'");
}

void* DeviceFrameBatch::plane_device(const std::string& name) { return d_planes_.at(name).data(); }
void* DeviceFrameBatch::destaggered_device(const std::string& name) { return d_dst_.at(name).data(); }
void* DeviceFrameBatch::xyz_device(int k) { return d_xyz_[k].data(); }

void DeviceFrameBatch::download_plane(const std::string& name, uint32_t frame, void* host, bool destaggered) {
    ScopedContext on_my_context(ctx_);
    DeviceBuffer& b = destaggered ? d_dst_.at(name) : d_planes_.at(name);
    const size_t bytes = b.size() / n_frames_;
    sync();
    b.download(host, bytes, bytes * frame);
}
void DeviceFrameBatch::download_xyz(int k, uint32_t frame, void* host) {
    ScopedContext on_my_context(ctx_);
    const size_t bytes = d_xyz_[k].size() / n_frames_;
    sync();
    d_xyz_[k].download(host, bytes, bytes * frame);
}
void DeviceFrameBatch::download_headers(uint32_t frame, uint64_t* ts, uint16_t* mid, uint32_t* st) {
    ScopedContext on_my_context(ctx_);
    sync();
    if (ts) d_ts_.download(ts, static_cast<size_t>(w_) * 8, static_cast<size_t>(frame) * w_ * 8);
    if (mid) d_mid_.download(mid, static_cast<size_t>(w_) * 2, static_cast<size_t>(frame) * w_ * 2);
    if (st) d_status_.download(st, static_cast<size_t>(w_) * 4, static_cast<size_t>(frame) * w_ * 4);
}

void DeviceFrameBatch::upload_poses(uint32_t frame, const double* poses) {
    ScopedContext on_my_context(ctx_);
    if (frame >= n_frames_) throw std::out_of_range("DeviceFrameBatch: frame index");
    const size_t per = static_cast<size_t>(w_) * 128;
    if (d_poses_.size() == 0) {  // identity for every column, like a fresh LidarFrame
        std::vector<double> ident(static_cast<size_t>(n_frames_) * w_ * 16, 0.0);
        for (size_t i = 0; i < ident.size(); i += 16) ident[i] = ident[i + 5] = ident[i + 10] = ident[i + 15] = 1.0;
        d_poses_.resize(ident.size() * 8);
        d_poses_.upload(ident.data(), ident.size() * 8);
    }
    if (poses) d_poses_.upload(poses, per, per * frame);
    // float output: the range-gated dewarp takes the poses as rows 0..2 cast to float (what dewarp<float> multiplies with,
    // pose_util.h:38-56), converted here on the host: 48 B per column on the device instead of 128
    if (!opt_.xyz_f64) {
        const size_t rows_per = static_cast<size_t>(w_) * 12;
        if (d_pose_rows_.size() == 0) {
            std::vector<float> ident(static_cast<size_t>(n_frames_) * rows_per, 0.0f);
            for (size_t i = 0; i < ident.size(); i += 12) ident[i] = ident[i + 5] = ident[i + 10] = 1.0f;
            d_pose_rows_.resize(ident.size() * 4);
            d_pose_rows_.upload(ident.data(), ident.size() * 4);
        }
        if (poses) {
            std::vector<float> rows(rows_per);
            for (size_t c = 0; c < w_; ++c)
                for (size_t k = 0; k < 12; ++k) rows[c * 12 + k] = static_cast<float>(poses[c * 16 + k]);
            d_pose_rows_.upload(rows.data(), rows_per * 4, rows_per * 4 * frame);
        }
    }
}

uint64_t DeviceFrameBatch::dewarp(double min_range, double max_range, bool provenance) {
    dewarp_async(min_range, max_range, provenance);
    ScopedContext on_my_context(ctx_);
    dw_offsets_.resize(static_cast<size_t>(n_frames_) + 1);
    d_dw_off_.download(dw_offsets_.data(), dw_offsets_.size() * 8);  // synchronous
    return dw_offsets_.back();
}

void DeviceFrameBatch::dewarp_async(double min_range, double max_range, bool provenance) {
    ScopedContext on_my_context(ctx_);
    auto rp = d_planes_.find(ChanField::RANGE);
    if (rp == d_planes_.end() || luts_.empty())
        throw std::invalid_argument("DeviceFrameBatch::dewarp needs the RANGE plane and options.xyz");
    if (d_poses_.size() == 0) upload_poses(0, nullptr);
    const size_t cap = static_cast<size_t>(n_frames_) * h_ * w_;
    d_dw_pts_.resize(cap * (opt_.xyz_f64 ? 24 : 12));
    d_dw_off_.resize((static_cast<size_t>(n_frames_) + 1) * 8);
    dw_prov_ = provenance;
    if (provenance) {
        d_dw_fi_.resize(cap * 4);
        d_dw_ci_.resize(cap * 4);
        d_dw_ts_.resize(cap * 8);
    }
    std::vector<const ouster_hip_lut*> luts;
    for (const auto& l : luts_) luts.push_back(l.device().handle);
    // decode() already counted the gated pixels per column when it ran with this very gate
    const bool counted = gate_valid_ && min_range == opt_.gate_min_range && max_range == opt_.gate_max_range;
    if (!opt_.xyz_f64 && d_pose_rows_.size())
        check(ouster_hip_dewarp_frames_rows(
            default_ctx(), luts.data(), static_cast<uint32_t>(luts.size()),
            static_cast<const uint32_t*>(rp->second.data()), static_cast<const uint32_t*>(d_status_.data()),
            static_cast<const uint64_t*>(d_ts_.data()), static_cast<const float*>(d_pose_rows_.data()), n_frames_, min_range,
            max_range, d_dw_pts_.data(), provenance ? static_cast<uint32_t*>(d_dw_fi_.data()) : nullptr,
            provenance ? static_cast<uint32_t*>(d_dw_ci_.data()) : nullptr,
            provenance ? static_cast<uint64_t*>(d_dw_ts_.data()) : nullptr, cap, static_cast<uint64_t*>(d_dw_off_.data()),
            counted ? static_cast<const uint16_t*>(d_gate_.data()) : nullptr));
    else
    check(ouster_hip_dewarp_frames_counted(
        default_ctx(), luts.data(), static_cast<uint32_t>(luts.size()),
        static_cast<const uint32_t*>(rp->second.data()), static_cast<const uint32_t*>(d_status_.data()),
        static_cast<const uint64_t*>(d_ts_.data()), static_cast<const double*>(d_poses_.data()), n_frames_,
        min_range, max_range, opt_.xyz_f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32, d_dw_pts_.data(),
        provenance ? static_cast<uint32_t*>(d_dw_fi_.data()) : nullptr,
        provenance ? static_cast<uint32_t*>(d_dw_ci_.data()) : nullptr,
        provenance ? static_cast<uint64_t*>(d_dw_ts_.data()) : nullptr, cap,
        static_cast<uint64_t*>(d_dw_off_.data()), counted ? static_cast<const uint16_t*>(d_gate_.data()) : nullptr));
}

void DeviceFrameBatch::download_dewarped(void* points, uint32_t* fi, uint32_t* ci, uint64_t* ts) {
    ScopedContext on_my_context(ctx_);
    if (dw_offsets_.empty()) throw std::logic_error("DeviceFrameBatch: dewarp() has not run");
    const size_t n = dw_offsets_.back();
    if (!n) return;
    if (points) d_dw_pts_.download(points, n * (opt_.xyz_f64 ? 24 : 12));
    if ((fi || ci || ts) && !dw_prov_) throw std::logic_error("DeviceFrameBatch: dewarp() ran without provenance");
    if (fi) d_dw_fi_.download(fi, n * 4);
    if (ci) d_dw_ci_.download(ci, n * 4);
    if (ts) d_dw_ts_.download(ts, n * 8);
}

}  // namespace hip
}  // namespace sdk
}  // namespace ouster
