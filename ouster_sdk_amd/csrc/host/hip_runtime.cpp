// hip_runtime.cpp -- glue between the C++ host API and the C ABI (include/ouster_hip.h):
// default context, error translation, staging buffers, and the GPU-backed implementations
// of destagger, cartesian and the single-packet PacketFormat::col_field / block_field.
//
// There is deliberately no CPU fallback here: if the HIP library cannot create a context
// (no GPU), every entry point throws std::runtime_error.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

#include "host_internal.h"
#include "ouster/core/lidar_frame.h"
#include "ouster/core/pose_util.h"
#include "ouster/core/xyzlut.h"
#include "ouster/hip/context.h"
#include "ouster/hip/device_buffer.h"

namespace ouster {
namespace sdk {
namespace hip {

void check(int rc) {
    if (rc == OUSTER_HIP_OK) return;
    const std::string msg = ouster_hip_last_error();
    if (rc == OUSTER_HIP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error("ouster_hip: " + msg);
}

// ---------------------------------------------------------------------------------------
// contexts: one per object that owns GPU state, one default per (thread, device) for the rest
// ---------------------------------------------------------------------------------------
namespace {
thread_local int t_device = -1;                                   // -1: not chosen yet
thread_local std::shared_ptr<Context> t_bound;                    // innermost ScopedContext
thread_local std::map<int, std::shared_ptr<Context>> t_defaults;  // this thread's default contexts

int initial_device() {
    if (const char* e = std::getenv("OUSTER_HIP_DEVICE")) return std::atoi(e);  // read once per thread
    return 0;
}
}  // namespace

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int current_device() {
    if (t_device < 0) t_device = initial_device();
    return t_device;
}

void set_device(int device) {
    const int n = device_count();
    if (device < 0 || device >= std::max(n, 1))
        throw std::invalid_argument("ouster_hip: device " + std::to_string(device) + " out of range [0," +
                                    std::to_string(n) + ")");
    t_device = device;
}

Context::Context(int device) : device_(device) { check(ouster_hip_ctx_create(device, nullptr, &ctx_)); }
Context::~Context() {
    if (ctx_) ouster_hip_ctx_destroy(ctx_);
}
void* Context::stream() const { return ouster_hip_ctx_stream(ctx_); }
void Context::sync() const { check(ouster_hip_sync(ctx_)); }

std::shared_ptr<Context> Context::for_device(int device) {
    auto& slot = t_defaults[device];
    if (!slot) slot = std::make_shared<Context>(device);
    return slot;
}
std::shared_ptr<Context> Context::current() { return t_bound ? t_bound : for_device(current_device()); }

ScopedContext::ScopedContext(std::shared_ptr<Context> ctx) : prev_(std::move(t_bound)) {
    t_bound = std::move(ctx);
    if (t_bound) (void)hipSetDevice(t_bound->device());
}
ScopedContext::~ScopedContext() {
    t_bound = std::move(prev_);
    if (t_bound) (void)hipSetDevice(t_bound->device());
}

ouster_hip_ctx* default_ctx() {
    const std::shared_ptr<Context> c = Context::current();
    (void)hipSetDevice(c->device());  // allocations of the caller land on the context's GPU
    return c->handle();
}

static void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("ouster_hip: ") + what + ": " + hipGetErrorString(e));
}

DeviceBuffer::DeviceBuffer(size_t bytes) { resize(bytes); }
DeviceBuffer::~DeviceBuffer() { ouster_hip_device_free(p_); }
DeviceBuffer::DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) {
    o.p_ = nullptr;
    o.n_ = o.cap_ = 0;
}
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
    std::swap(p_, o.p_);
    std::swap(n_, o.n_);
    std::swap(cap_, o.cap_);
    return *this;
}
void DeviceBuffer::resize(size_t bytes) {
    ouster_hip_ctx* ctx = default_ctx();  // selects the device
    if (bytes > cap_) {
        ouster_hip_device_free(p_);
        p_ = nullptr;
        cap_ = 0;
        check(ouster_hip_device_alloc(ctx, bytes ? bytes : 1, &p_));
        cap_ = bytes;
    }
    n_ = bytes;
}
void DeviceBuffer::upload_async(const void* src, size_t bytes, size_t offset) {
    check(ouster_hip_copy_in(default_ctx(), static_cast<uint8_t*>(p_) + offset, src, bytes));
}
void DeviceBuffer::download_async(void* dst, size_t bytes, size_t offset) const {
    check(ouster_hip_copy_out(default_ctx(), dst, static_cast<const uint8_t*>(p_) + offset, bytes));
}
void DeviceBuffer::upload(const void* src, size_t bytes, size_t offset) {
    upload_async(src, bytes, offset);
    check(ouster_hip_sync(default_ctx()));
}
void DeviceBuffer::download(void* dst, size_t bytes, size_t offset) const {
    download_async(dst, bytes, offset);
    check(ouster_hip_sync(default_ctx()));
}
void DeviceBuffer::fill(int byte_value) {
    auto st = static_cast<hipStream_t>(ouster_hip_ctx_stream(default_ctx()));
    hip_ok(hipMemsetAsync(p_, byte_value, n_, st), "hipMemsetAsync");
}

AllocStats alloc_stats() {
    ouster_hip_alloc_stats c{};
    ouster_hip_alloc_stats_read(&c);
    AllocStats s;
    s.device_allocs = c.device_allocs;
    s.device_frees = c.device_frees;
    s.pinned_allocs = c.pinned_allocs;
    s.pinned_frees = c.pinned_frees;
    s.pool_requests = c.pool_requests;
    s.pool_hits = c.pool_hits;
    s.pool_live_bytes = c.pool_live_bytes;
    s.pool_cached_bytes = c.pool_cached_bytes;
    return s;
}
bool is_device_accessible(const void* p, size_t bytes) { return ouster_hip_host_is_pinned(p, bytes) != 0; }

}  // namespace hip

namespace core {

// ---------------------------------------------------------------------------------------
// destagger (impl/lidar_frame_impl.h:733-811)
// ---------------------------------------------------------------------------------------
namespace impl {
void* host_alloc(size_t bytes, bool zero) { return ouster_hip_host_alloc(bytes, zero ? 1 : 0); }
void host_free(void* p, size_t bytes) noexcept {
    if (!p) return;
    if (bytes < OUSTER_HIP_HOST_POOL_MIN) std::free(p);   // never a pool block: skip the registry
    else ouster_hip_host_free(p);
}

// One launch on the images where they lie when they are pool memory (Field, img_t: one pass over the link in each
// direction, at the same time); any other memory goes through the context's grow-only scratch (ouster_hip_destagger_host).
void destagger_bytes(const void* img, void* out, size_t h, size_t w, size_t elem_bytes,
                     const std::vector<int>& shifts, bool inverse, size_t out_h, size_t out_w) {
    if (shifts.size() != h) throw std::invalid_argument{"image height does not match shifts size"};
    if (h != out_h || w != out_w)
        throw std::invalid_argument{"image and destaggered must have the same shape"};
    if (h == 0 || w == 0) return;
    // a plane of a frame a FrameBatcher has just released: its destaggered form is still in HBM (impl::MirrorPlane) -- one copy out
    if (!inverse && mirrors_live()) {
        MirrorPlane m;
        if (mirror_find(img, m) && m.h == h && m.w == w && m.elem == elem_bytes && img != out && *m.shifts == shifts) {
            const std::shared_ptr<hip::Context> cur = hip::Context::current();
            hip::ScopedContext on_device(cur->device() == m.device ? cur : hip::Context::for_device(m.device));
            ouster_hip_ctx* ctx = hip::default_ctx();
            hip::check(ouster_hip_copy_out(ctx, out, m.d_destaggered, h * w * elem_bytes));
            hip::check(ouster_hip_sync(ctx));
            return;
        }
    }
    static_assert(sizeof(int) == sizeof(int32_t), "pixel_shift_by_row is handed over as int32");
    hip::check(ouster_hip_destagger_host(hip::default_ctx(), img, out, static_cast<uint32_t>(h),
                                         static_cast<uint32_t>(w), static_cast<uint32_t>(elem_bytes),
                                         reinterpret_cast<const int32_t*>(shifts.data()),
                                         static_cast<uint32_t>(shifts.size()), inverse ? 1 : 0));
}
}  // namespace impl

Field destagger(const SensorInfo& info, const Field& field, bool inverse) {
    const auto& shape = field.shape();
    if (shape.size() < 2 || shape[0] != info.format.pixels_per_column ||
        shape[1] != info.format.columns_per_frame ||
        shape[0] != info.format.pixel_shift_by_row.size())
        throw std::invalid_argument{"Image resolution must match SensorInfo."};
    Field out(field.tag(), shape, field.field_class());
    size_t extra = 1;
    for (size_t i = 2; i < shape.size(); ++i) extra *= shape[i];
    impl::destagger_bytes(field.get(), out.get(), shape[0], shape[1], field.element_size() * extra,
                          info.format.pixel_shift_by_row, inverse, shape[0], shape[1]);
    return out;
}

// ---------------------------------------------------------------------------------------
// XYZ lookup tables (xyzlut.cpp:11-124)
// ---------------------------------------------------------------------------------------
namespace impl {

DeviceLut::~DeviceLut() {
    if (handle) ouster_hip_lut_destroy(handle);
}

std::shared_ptr<DeviceLut> device_lut_from_arrays(const void* direction, const void* offset,
                                                  size_t h, size_t w, bool f64) {
    auto d = std::make_shared<DeviceLut>();
    d->device = hip::Context::current()->device();
    hip::check(ouster_hip_lut_create_from_arrays(hip::default_ctx(), direction, offset,
                                                 static_cast<uint32_t>(h), static_cast<uint32_t>(w),
                                                 f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32, &d->handle));
    return d;
}

std::shared_ptr<DeviceLut> device_lut_from_calib(size_t w, size_t h, double range_unit,
                                                 const mat4d& b2l, const mat4d& transform,
                                                 const std::vector<double>& az,
                                                 const std::vector<double>& alt,
                                                 ArrayX3R<double>* direction,
                                                 ArrayX3R<double>* offset) {
    if (w <= 0 || h <= 0) throw std::invalid_argument("lut dimensions must be greater than zero");
    if ((az.size() != h || alt.size() != h) && (az.size() != w * h || alt.size() != w * h))
        throw std::invalid_argument("unexpected frame dimensions");
    ouster_hip_calib c{};
    c.w = static_cast<uint32_t>(w);
    c.h = static_cast<uint32_t>(h);
    c.range_unit = range_unit;
    std::memcpy(c.beam_to_lidar_transform, b2l.data(), sizeof c.beam_to_lidar_transform);
    std::memcpy(c.transform, transform.data(), sizeof c.transform);
    c.azimuth_angles_deg = az.data();
    c.altitude_angles_deg = alt.data();
    c.n_angles = az.size();
    auto d = std::make_shared<DeviceLut>();
    d->device = hip::Context::current()->device();
    hip::check(ouster_hip_lut_create(hip::default_ctx(), &c, &d->handle));
    if (direction && offset) {
        *direction = ArrayX3R<double>(w * h);
        *offset = ArrayX3R<double>(w * h);
        hip::check(ouster_hip_lut_export(d->handle, direction->data(), offset->data()));
    }
    return d;
}

XYZLut make_xyz_lut(size_t w, size_t h, double range_unit, const mat4d& beam_to_lidar_transform,
                    const mat4d& transform, const std::vector<double>& azimuth_angles_deg,
                    const std::vector<double>& altitude_angles_deg) {
    ArrayX3R<double> direction, offset;
    auto dev = device_lut_from_calib(w, h, range_unit, beam_to_lidar_transform, transform,
                                     azimuth_angles_deg, altitude_angles_deg, &direction, &offset);
    XYZLut lut(std::move(direction), std::move(offset), h, w);
    lut.attach_device(std::move(dev));
    return lut;
}

XYZLut make_xyz_lut(const SensorInfo& sensor, bool use_extrinsics) {
    mat4d transform = sensor.lidar_to_sensor_transform;
    if (use_extrinsics) {
        // extrinsics are in metres, the LUT works in range units (mm)
        mat4d ext = sensor.sensor_to_body;
        for (int r = 0; r < 3; ++r) ext(r, 3) /= RANGE_UNIT;
        transform = ext * sensor.lidar_to_sensor_transform;
    }
    return make_xyz_lut(sensor.format.columns_per_frame, sensor.format.pixels_per_column, RANGE_UNIT,
                        sensor.beam_to_lidar_transform, transform, sensor.beam_azimuth_angles,
                        sensor.beam_altitude_angles);
}

void cartesian_device(const DeviceLut& dev, const uint32_t* range, size_t n, void* points,
                      bool points_f64) {
    if (n == 0) return;
    hip::ScopedContext on_lut_device(hip::Context::current()->device() == dev.device
                                         ? hip::Context::current() : hip::Context::for_device(dev.device));
    // the range plane of a frame a FrameBatcher has just released: its cloud may already be in HBM (impl::XyzWish) -- one copy
    // out; if not, the batcher learns which LUT its frames are projected with and its next release produces it
    if (mirrors_live()) {
        MirrorPlane m;
        if (mirror_find(range, m) && m.elem == 4 && m.h * m.w == n && m.device == dev.device) {
            if (m.d_xyz && m.xyz_lut.get() == &dev && m.xyz_f64 == points_f64) {
                ouster_hip_ctx* ctx = hip::default_ctx();
                hip::check(ouster_hip_copy_out(ctx, points, m.d_xyz, n * 3 * (points_f64 ? 8 : 4)));
                hip::check(ouster_hip_sync(ctx));
                return;
            }
            if (m.wish) {
                std::lock_guard<std::mutex> g(m.wish->mu);
                if (m.wish->lut.get() != &dev || m.wish->f64 != points_f64) {
                    try {
                        m.wish->lut = dev.shared_from_this();
                        m.wish->f64 = points_f64;
                    } catch (const std::bad_weak_ptr&) {   // a DeviceLut that is not owned by a shared_ptr: nothing to remember
                    }
                }
            }
        }
    }
    hip::check(ouster_hip_cartesian_host(hip::default_ctx(), dev.handle, range, points,
                                         points_f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32));
}

}  // namespace impl

namespace impl {
void dewarp_device(const void* points, const double* poses, void* out, bool f64, size_t h, size_t w) {
    hip::check(ouster_hip_dewarp_host(hip::default_ctx(), points, poses, out, f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32,
                                      static_cast<uint32_t>(h), static_cast<uint32_t>(w)));
}
}  // namespace impl

namespace impl {
// dewarp(LidarFrame | FrameSet, ...) (impl/dewarp_impl.h:23-115): stage RANGE / status /
// timestamp / body_to_world of the frames, one ouster_hip_dewarp_frames per run of frames with
// equal dimensions, append the compacted results.  The buffers are the thread's and only grow.
void dewarp_frames_device(const std::vector<const LidarFrame*>& frames,
                          const std::vector<const DeviceLut*>& luts,
                          const std::vector<uint32_t>& frame_index, double min_range,
                          double max_range, bool f64, std::vector<unsigned char>& points,
                          std::vector<uint32_t>* frame_idxs, std::vector<uint32_t>* col_idxs,
                          std::vector<uint64_t>* timestamps_ns) {
    const size_t esz = f64 ? 24 : 12;
    ouster_hip_ctx* ctx = hip::default_ctx();
    struct Bufs {
        hip::DeviceBuffer d_rng, d_st, d_ts, d_po, d_off, d_pts, d_ci, d_tn;
        int device = -1;
    };
    static thread_local Bufs tl;
    if (tl.device != ouster_hip_ctx_device(ctx)) {   // the thread moved to another GPU: fresh buffers there
        tl = Bufs();
        tl.device = ouster_hip_ctx_device(ctx);
    }
    size_t i = 0;
    while (i < frames.size()) {
        const size_t h = frames[i]->h, w = frames[i]->w;
        size_t j = i;
        while (j < frames.size() && frames[j]->h == h && frames[j]->w == w) ++j;
        const size_t n = j - i, npx = h * w;
        hip::DeviceBuffer &d_rng = tl.d_rng, &d_st = tl.d_st, &d_ts = tl.d_ts, &d_po = tl.d_po, &d_off = tl.d_off,
                          &d_pts = tl.d_pts, &d_ci = tl.d_ci, &d_tn = tl.d_tn;
        d_rng.resize(n * npx * 4);
        d_st.resize(n * w * 4);
        d_ts.resize(n * w * 8);
        d_po.resize(n * w * 128);
        d_off.resize((n + 1) * 8);
        d_pts.resize(n * npx * esz);
        if (col_idxs) d_ci.resize(n * npx * 4);
        if (timestamps_ns) d_tn.resize(n * npx * 8);
        std::vector<const ouster_hip_lut*> handles(n);
        for (size_t k = 0; k < n; ++k) {
            const LidarFrame& fr = *frames[i + k];
            const auto range = fr.field<uint32_t>(ChanField::RANGE);
            d_rng.upload_async(range.data(), npx * 4, k * npx * 4);
            d_st.upload_async(fr.status().data(), w * 4, k * w * 4);
            d_ts.upload_async(fr.timestamp().data(), w * 8, k * w * 8);
            d_po.upload_async(fr.body_to_world().get<double>(), w * 128, k * w * 128);
            handles[k] = luts[i + k]->handle;
        }
        hip::check(ouster_hip_dewarp_frames(
            ctx, handles.data(), static_cast<uint32_t>(n),
            static_cast<const uint32_t*>(d_rng.data()), static_cast<const uint32_t*>(d_st.data()),
            static_cast<const uint64_t*>(d_ts.data()), static_cast<const double*>(d_po.data()),
            static_cast<uint32_t>(n), min_range, max_range, f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32,
            d_pts.data(), nullptr, col_idxs ? static_cast<uint32_t*>(d_ci.data()) : nullptr,
            timestamps_ns ? static_cast<uint64_t*>(d_tn.data()) : nullptr, n * npx, static_cast<uint64_t*>(d_off.data())));
        std::vector<uint64_t> off(n + 1);
        d_off.download(off.data(), (n + 1) * 8);
        const size_t total = off[n];
        if (total) {
            const size_t p0 = points.size();
            points.resize(p0 + total * esz);
            d_pts.download_async(points.data() + p0, total * esz);
            if (col_idxs) {
                const size_t c0 = col_idxs->size();
                col_idxs->resize(c0 + total);
                d_ci.download_async(col_idxs->data() + c0, total * 4);
            }
            if (timestamps_ns) {
                const size_t t0 = timestamps_ns->size();
                timestamps_ns->resize(t0 + total);
                d_tn.download_async(timestamps_ns->data() + t0, total * 8);
            }
            hip::check(ouster_hip_sync(ctx));
            if (frame_idxs) {  // batch-local index -> caller's FrameSet index
                const size_t f0 = frame_idxs->size();
                frame_idxs->resize(f0 + total);
                for (size_t k = 0; k < n; ++k)
                    std::fill(frame_idxs->begin() + f0 + off[k], frame_idxs->begin() + f0 + off[k + 1],
                              frame_index[i + k]);
            }
        }
        i = j;
    }
}
}  // namespace impl

PointCloudXYZd cartesian(const ImgRef<const uint32_t>& range, const XYZLut& lut) {
    if (range.cols() * range.rows() != lut.direction.rows())
        throw std::invalid_argument("unexpected image dimensions");
    return lut(range);
}

PointCloudXYZd cartesian(const LidarFrame& frame, const XYZLut& lut) {
    return cartesian(frame.field<uint32_t>(ChanField::RANGE), lut);
}

// ---------------------------------------------------------------------------------------
// PacketFormat::col_field / block_field: one-packet decodes through the decode kernel
// ---------------------------------------------------------------------------------------
namespace {
// Decode `name` of the packet in `lidar_buf` into a device plane of H x cols elements of
// `elem` bytes, then copy back only the columns the packet actually covers.
void decode_one_packet(const PacketFormat& pf, const std::string& name, size_t elem,
                       const uint8_t* lidar_buf, int cols, uint8_t* data) {
    const FieldDecodeInfo& info = pf.field_decode_info(name);
    if (elem < field_type_size(info.ty_tag) * static_cast<size_t>(info.num_elements))
        throw std::invalid_argument("Dest type too small for specified field");
    ouster_hip_format_desc d;
    pf.fill_hip_desc(static_cast<uint32_t>(cols), {{name, static_cast<uint32_t>(elem)}}, {false}, d);
    ouster_hip_ctx* ctx = hip::default_ctx();
    ouster_hip_format* fmt = nullptr;
    hip::check(ouster_hip_format_create(ctx, &d, &fmt));
    const size_t H = pf.pixels_per_column, plane_bytes = H * static_cast<size_t>(cols) * elem;
    try {
        const size_t stride = (pf.lidar_packet_size + 15) & ~size_t{15};
        void *d_pkt = nullptr, *d_plane = nullptr;   // the context's scratch: nothing is allocated per call
        hip::check(ouster_hip_ctx_scratch(ctx, 4, stride, &d_pkt));
        hip::check(ouster_hip_ctx_scratch(ctx, 5, plane_bytes, &d_plane));
        hip::check(ouster_hip_copy_in(ctx, d_pkt, lidar_buf, pf.lidar_packet_size));
        ouster_hip_frame_out out{};
        out.planes[0] = d_plane;
        out.xyz_field[0] = out.xyz_field[1] = -1;
        hip::check(ouster_hip_decode(ctx, fmt, static_cast<const uint8_t*>(d_pkt), stride, 1, nullptr, 1, nullptr, &out,
                                     nullptr, nullptr, 0));
        std::vector<uint8_t> host(plane_bytes);
        hip::check(ouster_hip_copy_out(ctx, host.data(), d_plane, plane_bytes));
        hip::check(ouster_hip_sync(ctx));
        for (uint32_t icol = 0; icol < pf.columns_per_packet; ++icol) {
            const uint8_t* col = pf.nth_col(icol, lidar_buf);
            const uint16_t m_id = pf.col_measurement_id(col);
            if (!(pf.col_status(col) & 0x01) || m_id >= cols) continue;
            for (size_t px = 0; px < H; ++px)
                std::memcpy(data + (px * cols + m_id) * elem, host.data() + (px * cols + m_id) * elem,
                            elem);
        }
    } catch (...) {
        ouster_hip_format_destroy(fmt);
        throw;
    }
    ouster_hip_format_destroy(fmt);
}
}  // namespace

template <typename T, int BlockDim>
void PacketFormat::block_field(T* data, int cols, const std::string& f,
                               const uint8_t* lidar_buf) const {
    static_assert(BlockDim == 4 || BlockDim == 8 || BlockDim == 16, "BlockDim must be 4, 8 or 16");
    // block_field trusts its caller: every column is written, at the measurement id of its
    // block's first column plus its position in the block (parsing.cpp:641-655).  Make a copy of
    // the packet that says exactly that, so the decode kernel reproduces it.
    std::vector<uint8_t> pkt(lidar_buf, lidar_buf + lidar_packet_size);
    pkt.resize(lidar_packet_size + 8, 0);
    for (uint32_t icol = 0; icol < columns_per_packet; icol += BlockDim) {
        const uint16_t head = col_measurement_id(nth_col(icol, lidar_buf));
        for (uint32_t x = 0; x < static_cast<uint32_t>(BlockDim) && icol + x < columns_per_packet; ++x) {
            uint8_t* col = nth_col(icol + x, pkt.data());
            set_col_measurement_id(col, static_cast<uint16_t>(head + x));
            set_col_status(col, col_status(col) | 0x01);
        }
    }
    decode_one_packet(*this, f, sizeof(T), pkt.data(), cols, reinterpret_cast<uint8_t*>(data));
}

template <typename T>
void PacketFormat::col_field(const uint8_t* col_buf, const std::string& f, T* dst,
                             int dst_stride) const {
    // wrap the column into a one-column packet image: [packet header][column][...]
    const FieldDecodeInfo& info = field_decode_info(f);
    if (sizeof(T) < field_type_size(info.ty_tag) * static_cast<size_t>(info.num_elements))
        throw std::invalid_argument("Dest type too small for specified field");
    std::vector<uint8_t> pkt(lidar_packet_size + 8, 0);
    std::memcpy(pkt.data() + packet_header_size, col_buf, col_size);
    uint8_t* col0 = nth_col(0, pkt.data());
    set_col_measurement_id(col0, 0);
    // force the staged copy valid; only column 0 is read back
    uint32_t st = col_status(col0);
    set_col_status(col0, st | 0x01);
    std::vector<T> plane(static_cast<size_t>(pixels_per_column) * columns_per_packet);
    decode_one_packet(*this, f, sizeof(T), pkt.data(), static_cast<int>(columns_per_packet),
                      reinterpret_cast<uint8_t*>(plane.data()));
    for (uint32_t px = 0; px < pixels_per_column; ++px)
        dst[static_cast<size_t>(px) * dst_stride] = plane[static_cast<size_t>(px) * columns_per_packet];
}

#define OUSTER_INST_FIELD(T)                                                                      \
    template void PacketFormat::col_field<T>(const uint8_t*, const std::string&, T*, int) const;  \
    template void PacketFormat::block_field<T, 4>(T*, int, const std::string&, const uint8_t*) const;  \
    template void PacketFormat::block_field<T, 8>(T*, int, const std::string&, const uint8_t*) const;  \
    template void PacketFormat::block_field<T, 16>(T*, int, const std::string&, const uint8_t*) const;
OUSTER_INST_FIELD(uint8_t)
OUSTER_INST_FIELD(uint16_t)
OUSTER_INST_FIELD(uint32_t)
OUSTER_INST_FIELD(uint64_t)
OUSTER_INST_FIELD(int8_t)
OUSTER_INST_FIELD(int16_t)
OUSTER_INST_FIELD(int32_t)
OUSTER_INST_FIELD(int64_t)
OUSTER_INST_FIELD(float)
OUSTER_INST_FIELD(double)
OUSTER_INST_FIELD(impl::float3x16_t)
#undef OUSTER_INST_FIELD

}  // namespace core
}  // namespace sdk
}  // namespace ouster
