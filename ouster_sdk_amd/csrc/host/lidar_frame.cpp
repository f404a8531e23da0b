// lidar_frame.cpp -- Field, LidarFrame, FrameBatcher, frame_to_packets (host side).
//
// Reference behaviour (paths relative to the reference checkout):
//   Field                      ouster_core/src/field.cpp:247-296
//   LidarFrame ctor / fields   ouster_core/src/lidar_frame.cpp:309-359, 1038-1117
//   FrameBatcher               ouster_core/src/lidar_frame.cpp:1248-1959 (lidar packets)
//   frame_to_packets           ouster_core/include/ouster/core/impl/lidar_frame_impl.h:435-531
// FrameBatcher keeps the reference's host state machine (frame boundaries, reorder cache,
// init-id changes, completeness) but defers ALL per-pixel work: the packets of the frame
// being assembled are collected in a staging buffer and decoded by one GPU launch when the
// frame is released.
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <deque>
#include <mutex>
#include <sstream>
#include <unordered_map>

#include "host_internal.h"
#include "ouster/core/lidar_frame.h"
#include "ouster/core/xyzlut.h"
#include "ouster/hip/device_buffer.h"

namespace ouster {
namespace sdk {
namespace core {

// ---------------------------------------------------------------------------------------
// mirror registry (impl::mirrors_live / mirror_forget / mirror_register / mirror_find)
// ---------------------------------------------------------------------------------------
namespace impl {
namespace {
struct MirrorRegistry {
    std::mutex mu;
    std::unordered_map<const void*, MirrorPlane> planes;
    std::atomic<int> live{0};
};
MirrorRegistry& mirrors() {
    static MirrorRegistry* r = new MirrorRegistry();   // leaked: Field destructors of static frames may run late
    return *r;
}
}  // namespace
bool mirrors_live() noexcept { return mirrors().live.load(std::memory_order_relaxed) != 0; }
void mirror_forget(const void* host) noexcept {
    MirrorRegistry& r = mirrors();
    std::shared_ptr<void> drop;   // released outside the lock
    std::lock_guard<std::mutex> g(r.mu);
    auto it = r.planes.find(host);
    if (it == r.planes.end()) return;
    drop = std::move(it->second.keep);
    r.planes.erase(it);
    r.live.store(static_cast<int>(r.planes.size()), std::memory_order_relaxed);
}
void mirror_register(const MirrorPlane& m) {
    MirrorRegistry& r = mirrors();
    std::lock_guard<std::mutex> g(r.mu);
    r.planes[m.host] = m;
    r.live.store(static_cast<int>(r.planes.size()), std::memory_order_relaxed);
}
bool mirror_find(const void* host, MirrorPlane& out) {
    MirrorRegistry& r = mirrors();
    std::lock_guard<std::mutex> g(r.mu);
    auto it = r.planes.find(host);
    if (it == r.planes.end()) return false;
    out = it->second;
    return true;
}
}  // namespace impl

// ---------------------------------------------------------------------------------------
// Field
// ---------------------------------------------------------------------------------------
Field::Field(ChanFieldType tag, std::vector<size_t> shape, FieldClass c)
    : tag_(tag), shape_(std::move(shape)), class_(c) {
    count_ = 1;
    for (size_t d : shape_) count_ *= d;
    if (shape_.empty()) count_ = 1;
    const size_t b = count_ * field_type_size(tag_);
    ptr_ = b ? impl::host_alloc(b, true) : nullptr;   // planes come from the pinned pool: the GPU writes them in place
    if (b && !ptr_) throw std::runtime_error("Field: host memory allocation failed");
}
Field::Field(const Field& o) : tag_(o.tag_), shape_(o.shape_), class_(o.class_), count_(o.count_) {
    const size_t b = o.bytes();
    if (b) {
        ptr_ = impl::host_alloc(b, false);
        if (!ptr_) throw std::runtime_error("Field: host memory allocation failed");
        std::memcpy(ptr_, o.ptr_, b);
    }
}
Field::Field(Field&& o) noexcept
    : tag_(o.tag_), shape_(std::move(o.shape_)), class_(o.class_), count_(o.count_), ptr_(o.ptr_), escaped_(o.escaped_) {
    o.ptr_ = nullptr;
    o.escaped_ = false;
    o.count_ = 0;
    o.tag_ = ChanFieldType::VOID;
}
Field& Field::operator=(Field o) noexcept {
    std::swap(tag_, o.tag_);
    std::swap(shape_, o.shape_);
    std::swap(class_, o.class_);
    std::swap(count_, o.count_);
    std::swap(ptr_, o.ptr_);
    std::swap(escaped_, o.escaped_);
    return *this;
}
Field::~Field() {
    if (ptr_ && impl::mirrors_live()) impl::mirror_forget(ptr_);   // the block goes back to the pool: its address will be reused
    impl::host_free(ptr_, bytes());
}
void Field::set_zero() {
    if (ptr_ && impl::mirrors_live()) impl::mirror_forget(ptr_);
    if (ptr_) std::memset(ptr_, 0, bytes());
}
bool Field::operator==(const Field& o) const {
    return tag_ == o.tag_ && shape_ == o.shape_ &&
           (bytes() == 0 || std::memcmp(ptr_, o.ptr_, bytes()) == 0);
}

// ---------------------------------------------------------------------------------------
// default planes
// ---------------------------------------------------------------------------------------
LidarFrameFieldTypes get_field_types(UDPProfileLidar profile) {
    LidarFrameFieldTypes out;
    for (const auto& p : impl::default_planes(profile)) {
        FieldType ft(p.first, p.second, {}, FieldClass::PIXEL_FIELD);
        if (p.first == ChanField::RGB) ft.extra_dims.push_back(3);  // H x W x 3 float16
        out.push_back(std::move(ft));
    }
    return out;
}

LidarFrameFieldTypes get_field_types(const DataFormat& format, const Version& fw_version) {
    LidarFrameFieldTypes out = get_field_types(format.udp_profile_lidar);
    // WINDOW only exists from fw 3.2 (zone profiles: 3.2.1)
    const bool zone = format.udp_profile_lidar == UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_ZONE16 ||
                      format.udp_profile_lidar == UDPProfileLidar::RNG15_RFL8_NIR8_ZONE16;
    if (fw_version < Version(3, 2, 0) || (zone && fw_version < Version(3, 2, 1)))
        for (size_t i = 0; i < out.size(); ++i)
            if (out[i].name == ChanField::WINDOW) {
                out.erase(out.begin() + static_cast<std::ptrdiff_t>(i));
                break;
            }
    return out;
}

LidarFrameFieldTypes get_field_types(const SensorInfo& info) {
    return get_field_types(info.format, info.get_version());
}

// ---------------------------------------------------------------------------------------
// LidarFrame
// ---------------------------------------------------------------------------------------
LidarFrame::LidarFrame() = default;
// a copy is a snapshot: whatever a batcher still owes the source is decoded first, the copy owes nothing
LidarFrame::LidarFrame(const LidarFrame& o)
    : w(o.w), h(o.h), frame_id(o.frame_id), frame_status(o.frame_status), shutdown_countdown(o.shutdown_countdown),
      shot_limiting_countdown(o.shot_limiting_countdown), sensor_info(o.sensor_info),
      timestamp_((o.sync_(), o.timestamp_)), measurement_id_(o.measurement_id_), status_(o.status_),
      packet_timestamp_(o.packet_timestamp_), body_to_world_(o.body_to_world_), alert_flags_(o.alert_flags_),
      fields_(o.fields_), packet_count_(o.packet_count_) {}
LidarFrame::LidarFrame(LidarFrame&&) noexcept = default;
LidarFrame& LidarFrame::operator=(const LidarFrame& o) {
    if (this != &o) {
        LidarFrame tmp(o);
        *this = std::move(tmp);
    }
    return *this;
}
LidarFrame& LidarFrame::operator=(LidarFrame&&) noexcept = default;
LidarFrame::~LidarFrame() = default;

void LidarFrame::run_pending_() const {
    has_pending_ = false;   // first: the decode itself goes through the accessors
    std::shared_ptr<impl::PendingDecode> p = std::move(pending_);
    pending_.reset();
    if (p) p->flush(const_cast<LidarFrame&>(*this));
}

LidarFrame::LidarFrame(size_t h_, size_t w_, const LidarFrameFieldTypes& field_types,
                       size_t columns_per_packet)
    : w(w_), h(h_) {
    if (w * h == 0)
        throw std::invalid_argument("Cannot construct LidarFrame with zero width or height");
    if (columns_per_packet == 0)
        throw std::invalid_argument("columns_per_packet must be greater than 0");
    packet_count_ = (w + columns_per_packet - 1) / columns_per_packet;
    for (const auto& ft : field_types) add_field(ft);
    timestamp_ = Field(ChanFieldType::UINT64, {w}, FieldClass::COLUMN_FIELD);
    measurement_id_ = Field(ChanFieldType::UINT16, {w}, FieldClass::COLUMN_FIELD);
    status_ = Field(ChanFieldType::UINT32, {w}, FieldClass::COLUMN_FIELD);
    packet_timestamp_ = Field(ChanFieldType::UINT64, {packet_count_}, FieldClass::PACKET_FIELD);
    alert_flags_ = Field(ChanFieldType::UINT8, {packet_count_}, FieldClass::PACKET_FIELD);
    body_to_world_ = Field(ChanFieldType::FLOAT64, {w, 4, 4}, FieldClass::NONE);
    double* poses = body_to_world_.get<double>();
    for (size_t i = 0; i < w; ++i)
        for (int d = 0; d < 4; ++d) poses[i * 16 + d * 5] = 1.0;
}

LidarFrame::LidarFrame(size_t h_, size_t w_)
    : LidarFrame(h_, w_, get_field_types(UDPProfileLidar::LEGACY), DEFAULT_COLUMNS_PER_PACKET) {}

LidarFrame::LidarFrame(size_t h_, size_t w_, UDPProfileLidar profile, size_t columns_per_packet)
    : LidarFrame(h_, w_, get_field_types(profile), columns_per_packet) {}

LidarFrame::LidarFrame(const DataFormat& format)
    : LidarFrame(format.pixels_per_column, format.columns_per_frame,
                 get_field_types(format, Version(0, 0, 0)), format.columns_per_packet) {}

LidarFrame::LidarFrame(const SensorInfo& info) : LidarFrame(std::make_shared<SensorInfo>(info)) {}

LidarFrame::LidarFrame(std::shared_ptr<SensorInfo> info)
    : LidarFrame(info->format.pixels_per_column, info->format.columns_per_frame,
                 get_field_types(*info), info->format.columns_per_packet) {
    sensor_info = std::move(info);
}

LidarFrame::LidarFrame(std::shared_ptr<SensorInfo> info, const LidarFrameFieldTypes& field_types)
    : LidarFrame(info->format.pixels_per_column, info->format.columns_per_frame, field_types,
                 info->format.columns_per_packet) {
    sensor_info = std::move(info);
}

Field& LidarFrame::field(const std::string& name) {
    sync_();
    auto it = fields_.find(name);
    if (it == fields_.end()) throw std::out_of_range("Field '" + name + "' not found in LidarFrame");
    return it->second;
}
const Field& LidarFrame::field(const std::string& name) const {
    sync_();
    auto it = fields_.find(name);
    if (it == fields_.end()) throw std::out_of_range("Field '" + name + "' not found in LidarFrame");
    return it->second;
}
bool LidarFrame::has_field(const std::string& name) const { return fields_.count(name) > 0; }

Field& LidarFrame::add_field(const FieldType& type) {
    std::vector<size_t> dims;
    switch (type.field_class) {
        case FieldClass::PIXEL_FIELD: dims = {h, w}; break;
        case FieldClass::COLUMN_FIELD: dims = {w}; break;
        case FieldClass::PACKET_FIELD: dims = {packet_count_}; break;
        default: break;
    }
    dims.insert(dims.end(), type.extra_dims.begin(), type.extra_dims.end());
    return add_field(type.name, FieldDescriptor::array(type.element_type, dims), type.field_class);   // same checks
}
Field& LidarFrame::add_field(const std::string& name, ChanFieldType type,
                             std::vector<size_t> extra_dims, FieldClass c) {
    return add_field(FieldType(name, type, std::move(extra_dims), c));
}
Field& LidarFrame::add_field(const std::string& name, const FieldDescriptor& desc, FieldClass field_class) {
    if (has_field(name)) throw std::invalid_argument("Duplicated field '" + name + "'");
    const auto& sh = desc.shape;
    if (field_class == FieldClass::PIXEL_FIELD) {
        if (sh.size() < 2) throw std::invalid_argument("Pixel fields must have at least 2 dimensions");
        if (sh[0] != h || sh[1] != w)
            throw std::invalid_argument("Pixel field shape must match LidarFrame's width and height. Was " +
                                        std::to_string(sh[0]) + "x" + std::to_string(sh[1]) + " vs " + std::to_string(h) +
                                        "x" + std::to_string(w));
        for (size_t d : sh)
            if (d == 0) throw std::invalid_argument("Cannot add pixel field with 0 elements.");
    }
    if (field_class == FieldClass::COLUMN_FIELD && (sh.empty() || sh[0] != w))
        throw std::invalid_argument("Column field shape must match LidarFrame's height. Width was " +
                                    std::to_string(sh.empty() ? 0 : sh[0]) + " vs required width of " + std::to_string(w));
    if (field_class == FieldClass::PACKET_FIELD && (sh.empty() || sh[0] != packet_count_))
        throw std::invalid_argument("Packet field shape must match number of packets. Width was " +
                                    std::to_string(sh.empty() ? 0 : sh[0]) + " vs required width of " +
                                    std::to_string(packet_count_));
    return fields_.emplace(name, Field(desc.element_type, sh, field_class)).first->second;
}

Field LidarFrame::del_field(const std::string& name) {
    sync_();
    auto it = fields_.find(name);
    if (it == fields_.end())
        throw std::invalid_argument("Attempted deleting non existing field '" + name + "'");
    Field f = std::move(it->second);
    fields_.erase(it);
    return f;
}

LidarFrameFieldTypes LidarFrame::field_types() const {
    LidarFrameFieldTypes out;
    for (const auto& kv : fields_) {
        const auto& shape = kv.second.shape();
        size_t skip = 0;
        switch (kv.second.field_class()) {
            case FieldClass::PIXEL_FIELD: skip = 2; break;
            case FieldClass::COLUMN_FIELD:
            case FieldClass::PACKET_FIELD: skip = 1; break;
            default: break;
        }
        out.emplace_back(kv.first, kv.second.tag(),
                         std::vector<size_t>(shape.begin() + std::min(skip, shape.size()), shape.end()),
                         kv.second.field_class());
    }
    return out;
}

FieldType LidarFrame::field_type(const std::string& name) const {
    for (const auto& ft : field_types())
        if (ft.name == name) return ft;
    throw std::out_of_range("Field '" + name + "' not found in LidarFrame");
}

namespace {
// element-wise static_cast between two fields of the same shape (the reference's impl::copy_and_cast)
template <typename D>
void cast_into(D* dst, const Field& src) {
    const size_t n = src.size();
    switch (src.tag()) {
#define OUSTER_CAST_CASE(TAG, S) \
        case ChanFieldType::TAG: { const S* p = static_cast<const S*>(src.get()); for (size_t i = 0; i < n; ++i) dst[i] = static_cast<D>(p[i]); break; }
        OUSTER_CAST_CASE(UINT8, uint8_t) OUSTER_CAST_CASE(UINT16, uint16_t) OUSTER_CAST_CASE(UINT32, uint32_t)
        OUSTER_CAST_CASE(UINT64, uint64_t) OUSTER_CAST_CASE(INT8, int8_t) OUSTER_CAST_CASE(INT16, int16_t)
        OUSTER_CAST_CASE(INT32, int32_t) OUSTER_CAST_CASE(INT64, int64_t) OUSTER_CAST_CASE(FLOAT32, float)
        OUSTER_CAST_CASE(FLOAT64, double)
#undef OUSTER_CAST_CASE
        default: throw std::invalid_argument("LidarFrame: cannot cast a field of element type " + to_string(src.tag()));
    }
}
void copy_and_cast(Field& dst, const Field& src) {
    switch (dst.tag()) {
        case ChanFieldType::UINT8: cast_into(static_cast<uint8_t*>(dst.storage_()), src); break;
        case ChanFieldType::UINT16: cast_into(static_cast<uint16_t*>(dst.storage_()), src); break;
        case ChanFieldType::UINT32: cast_into(static_cast<uint32_t*>(dst.storage_()), src); break;
        case ChanFieldType::UINT64: cast_into(static_cast<uint64_t*>(dst.storage_()), src); break;
        case ChanFieldType::INT8: cast_into(static_cast<int8_t*>(dst.storage_()), src); break;
        case ChanFieldType::INT16: cast_into(static_cast<int16_t*>(dst.storage_()), src); break;
        case ChanFieldType::INT32: cast_into(static_cast<int32_t*>(dst.storage_()), src); break;
        case ChanFieldType::INT64: cast_into(static_cast<int64_t*>(dst.storage_()), src); break;
        case ChanFieldType::FLOAT32: cast_into(static_cast<float*>(dst.storage_()), src); break;
        case ChanFieldType::FLOAT64: cast_into(static_cast<double*>(dst.storage_()), src); break;
        default: throw std::invalid_argument("LidarFrame: cannot cast to element type " + to_string(dst.tag()));
    }
}
}  // namespace

// lidar_frame.cpp:361-401
LidarFrame::LidarFrame(const LidarFrame& other, const LidarFrameFieldTypes& fields)
    // a frame still being assembled decodes what has arrived first (the reference parses eagerly: its copy is consistent)
    : w((other.sync_(), other.w)), h(other.h), frame_id(other.frame_id), frame_status(other.frame_status),
      shutdown_countdown(other.shutdown_countdown), shot_limiting_countdown(other.shot_limiting_countdown),
      sensor_info(other.sensor_info), timestamp_(other.timestamp_), measurement_id_(other.measurement_id_),
      status_(other.status_), packet_timestamp_(other.packet_timestamp_), body_to_world_(other.body_to_world_),
      alert_flags_(other.alert_flags_), packet_count_(other.packet_count_) {
    for (const auto& ft : fields) {
        Field& dst = add_field(ft);
        if (!other.has_field(ft.name)) continue;             // zero padded
        const Field& src = other.field(ft.name);
        if (src.shape() != dst.shape())
            throw std::invalid_argument("Field '" + ft.name +
                                        "' from source frame has dimensions that don't match desired.");
        if (src.tag() == dst.tag()) std::memcpy(dst.storage_(), src.get(), src.bytes());
        else copy_and_cast(dst, src);
    }
}

void LidarFrame::set_column_pose(int index, const mat4d& pose) {
    if (index < 0 || static_cast<size_t>(index) >= w) throw std::out_of_range("LidarFrame: column index out of bounds");
    std::memcpy(body_to_world_.get<double>() + static_cast<size_t>(index) * 16, pose.data(), 16 * sizeof(double));
}
mat4d LidarFrame::get_column_pose(int index) const {
    if (index < 0 || static_cast<size_t>(index) >= w) throw std::out_of_range("LidarFrame: column index out of bounds");
    return mat4d::FromRowMajor(body_to_world_.get<double>() + static_cast<size_t>(index) * 16);
}

bool LidarFrame::complete() const {
    if (!sensor_info)
        throw std::runtime_error("LidarFrame must have a valid SensorInfo in order to compute completeness");
    return complete(sensor_info->format.column_window);
}

bool LidarFrame::complete(ColumnWindow window) const {
    const uint32_t* st = status_.get<uint32_t>();
    auto valid = [&](int a, int b) {
        for (int i = a; i <= b; ++i)
            if (!(st[i] & 0x01)) return false;
        return true;
    };
    if (window.first <= window.second) return valid(window.first, window.second);
    return valid(0, window.second) && valid(window.first, static_cast<int>(w) - 1);
}

int LidarFrame::get_first_valid_column() const {
    const auto st = status();
    for (size_t i = 0; i < w; ++i)
        if (st[i] & 1u) return static_cast<int>(i);
    throw std::runtime_error("No valid columns in LidarFrame");
}

int LidarFrame::get_last_valid_column() const {
    const auto st = status();
    for (size_t i = w; i-- > 0;)
        if (st[i] & 1u) return static_cast<int>(i);
    throw std::runtime_error("No valid columns in LidarFrame");
}

namespace {
// index of the packets that carry at least one valid column, in the order asked for
// (lidar_frame.cpp:735-797: columns_per_packet = w / number of packet slots)
template <class F>
void for_valid_packets(const LidarFrame& f, F&& fn) {
    const size_t total = f.packet_count();
    if (!total) return;
    const size_t cpp = f.w / total;
    const auto st = f.status();
    for (size_t i = 0; i < total; ++i) {
        bool any = false;
        for (size_t c = i * cpp; c < (i + 1) * cpp && c < f.w; ++c) any |= (st[c] & 1u) != 0;
        if (any) fn(i);
    }
}
uint64_t ts_or_throw(bool found, uint64_t v) {
    if (!found) throw std::runtime_error("No valid packets in LidarFrame");
    return v;
}
}  // namespace

uint64_t LidarFrame::get_first_valid_lidar_packet_timestamp() const {
    bool found = false;
    uint64_t v = 0;
    for_valid_packets(*this, [&](size_t i) { if (!found) { found = true; v = packet_timestamp()[i]; } });
    return v;
}
uint64_t LidarFrame::get_last_valid_lidar_packet_timestamp() const {
    uint64_t v = 0;
    for_valid_packets(*this, [&](size_t i) { v = packet_timestamp()[i]; });
    return v;
}
uint64_t LidarFrame::get_first_valid_packet_timestamp() const {
    bool found = false;
    uint64_t v = 0;
    for_valid_packets(*this, [&](size_t i) { if (!found) { found = true; v = packet_timestamp()[i]; } });
    return ts_or_throw(found, v);
}
uint64_t LidarFrame::get_last_valid_packet_timestamp() const {
    bool found = false;
    uint64_t v = 0;
    for_valid_packets(*this, [&](size_t i) { found = true; v = packet_timestamp()[i]; });
    return ts_or_throw(found, v);
}
namespace {
// The other packet streams a frame may carry stamps of, as plain fields (lidar_frame.cpp:839-891): IMU_PACKET_TIMESTAMP is
// valid where any of its packet's IMU_STATUS entries has bit 0 set, ZONE_PACKET_TIMESTAMP[0] where it is non-zero.
template <class F>
void for_other_stream_stamps(const LidarFrame& f, F&& fn) {
    if (f.has_field("IMU_PACKET_TIMESTAMP") && f.has_field("IMU_STATUS")) {
        const Field& ts = f.field("IMU_PACKET_TIMESTAMP");
        const Field& st = f.field("IMU_STATUS");
        if (ts.tag() == ChanFieldType::UINT64 && st.tag() == ChanFieldType::UINT16 && ts.size() > 0) {
            const size_t per = st.size() / ts.size();
            const uint64_t* t = ts.get<uint64_t>();
            const uint16_t* s = st.get<uint16_t>();
            for (size_t i = 0; i < ts.size(); ++i) {
                bool any = false;
                for (size_t k = i * per; k < (i + 1) * per; ++k) any |= (s[k] & 1u) != 0;
                if (any) fn(t[i]);
            }
        }
    }
    if (f.has_field("ZONE_PACKET_TIMESTAMP")) {
        const Field& z = f.field("ZONE_PACKET_TIMESTAMP");
        if (z.tag() == ChanFieldType::UINT64 && z.size() > 0 && z.get<uint64_t>()[0] != 0) fn(z.get<uint64_t>()[0]);
    }
}
}  // namespace

uint64_t LidarFrame::get_min_valid_packet_timestamp() const {
    bool found = false;
    uint64_t v = 0;
    auto take = [&](uint64_t t) {
        v = found ? std::min(v, t) : t;
        found = true;
    };
    for_valid_packets(*this, [&](size_t i) { take(packet_timestamp()[i]); });
    for_other_stream_stamps(*this, take);
    return ts_or_throw(found, v);
}
uint64_t LidarFrame::get_max_valid_packet_timestamp() const {
    bool found = false;
    uint64_t v = 0;
    auto take = [&](uint64_t t) {
        v = found ? std::max(v, t) : t;
        found = true;
    };
    for_valid_packets(*this, [&](size_t i) { take(packet_timestamp()[i]); });
    for_other_stream_stamps(*this, take);
    return ts_or_throw(found, v);
}

bool LidarFrame::equals(const LidarFrame& o) const {
    sync_();
    o.sync_();
    return w == o.w && h == o.h && frame_id == o.frame_id && frame_status == o.frame_status &&
           shutdown_countdown == o.shutdown_countdown &&
           shot_limiting_countdown == o.shot_limiting_countdown && fields_ == o.fields_ &&
           timestamp_ == o.timestamp_ && measurement_id_ == o.measurement_id_ &&
           status_ == o.status_ && packet_timestamp_ == o.packet_timestamp_ &&
           alert_flags_ == o.alert_flags_ && body_to_world_ == o.body_to_world_;
}
bool operator==(const LidarFrame& a, const LidarFrame& b) { return a.equals(b); }

std::string to_string(FieldClass c) {
    switch (c) {
        case FieldClass::PIXEL_FIELD: return "PIXEL_FIELD";
        case FieldClass::COLUMN_FIELD: return "COLUMN_FIELD";
        case FieldClass::PACKET_FIELD: return "PACKET_FIELD";
        case FieldClass::FRAME_FIELD: return "FRAME_FIELD";
        default: return "UNKNOWN";
    }
}
std::string to_string(const FieldType& ft) {
    std::string out = ft.name + ": " + to_string(ft.element_type) + " (";
    for (size_t i = 0; i < ft.extra_dims.size(); ++i) out += (i ? ", " : "") + std::to_string(ft.extra_dims[i]);
    return out + ") " + to_string(ft.field_class);
}
std::string to_string(const LidarFrameFieldTypes& fts) {
    std::string out = "(";
    for (size_t i = 0; i < fts.size(); ++i) out += (i ? ", " : "") + to_string(fts[i]);
    return out + ")";
}

namespace {
template <typename T>
void min_mean_max(const Field& f, std::ostream& os) {
    const T* p = static_cast<const T*>(f.get());
    double lo = static_cast<double>(p[0]), hi = lo, sum = 0;
    for (size_t i = 0; i < f.size(); ++i) {
        const double v = static_cast<double>(p[i]);
        lo = std::min(lo, v);
        hi = std::max(hi, v);
        sum += v;
    }
    os << "min: " << lo << "; mean: " << sum / static_cast<double>(f.size()) << "; max: " << hi;
}
}  // namespace

std::string to_string(const LidarFrame& frame) {
    std::ostringstream os;
    os << "LidarFrame: {h = " << frame.h << ", w = " << frame.w << ", packets_per_frame = " << frame.packet_timestamp().size()
       << ", fid = " << frame.frame_id << "," << std::endl
       << " frame status = " << std::hex << frame.frame_status << std::dec
       << ", thermal_shutdown status = " << to_string(frame.thermal_shutdown())
       << ", shot_limiting status = " << to_string(frame.shot_limiting()) << "," << std::endl
       << "  field_types = " << to_string(frame.field_types()) << "," << std::endl;
    for (const auto& kv : frame.fields()) {
        const Field& f = kv.second;
        os << "     " << kv.first << " type:" << to_string(f.tag()) << " shape: (";
        for (size_t i = 0; i < f.shape().size(); ++i) os << (i ? ", " : "") << f.shape()[i];
        os << ") ";
        if (f.bytes() > 0) {
            switch (f.tag()) {
                case ChanFieldType::UINT8: min_mean_max<uint8_t>(f, os); break;
                case ChanFieldType::UINT16: min_mean_max<uint16_t>(f, os); break;
                case ChanFieldType::UINT32: min_mean_max<uint32_t>(f, os); break;
                case ChanFieldType::UINT64: min_mean_max<uint64_t>(f, os); break;
                case ChanFieldType::INT8: min_mean_max<int8_t>(f, os); break;
                case ChanFieldType::INT16: min_mean_max<int16_t>(f, os); break;
                case ChanFieldType::INT32: min_mean_max<int32_t>(f, os); break;
                case ChanFieldType::INT64: min_mean_max<int64_t>(f, os); break;
                case ChanFieldType::FLOAT32: min_mean_max<float>(f, os); break;
                case ChanFieldType::FLOAT64: min_mean_max<double>(f, os); break;
                default: break;   // text, half floats and zone states have no summary here
            }
        }
        os << std::endl;
    }
    os << "}";
    return os.str();
}

uint64_t column_timestamp_at_destaggered_pixel(size_t row, size_t col,
                                               const std::vector<int>& pixel_shift_by_row,
                                               const HeaderRef<const uint64_t>& column_timestamps) {
    const size_t width = column_timestamps.size();
    if (row >= pixel_shift_by_row.size() || col >= width)
        throw std::invalid_argument("row or column is out of range");
    // the reference's signed int arithmetic: offset = (w + shift % w) % w
    const int w = static_cast<int>(width);
    const int offset = (w + pixel_shift_by_row[row] % w) % w;
    const int staggered_col = (static_cast<int>(col) - offset + w) % w;
    return column_timestamps[static_cast<size_t>(staggered_col)];
}

uint64_t column_timestamp_at_destaggered_pixel(const LidarFrame& frame, const SensorInfo& info,
                                               size_t row, size_t col) {
    return column_timestamp_at_destaggered_pixel(row, col, info.format.pixel_shift_by_row,
                                                 frame.timestamp());
}

// ---------------------------------------------------------------------------------------
// FrameBatcher
// ---------------------------------------------------------------------------------------
namespace {
struct CachedPacket {
    std::vector<uint8_t> buf;
    uint64_t host_timestamp;
    uint64_t seq;
};
}  // namespace

struct FrameBatcher::State : std::enable_shared_from_this<FrameBatcher::State> {
    std::shared_ptr<SensorInfo> info;
    size_t max_cache_size = 4;
    std::deque<CachedPacket> cache;  // small (<= max_cache_size); searched linearly
    uint64_t seq = 0;
    int64_t finished_frame_id = -1;
    int64_t last_frame_id = -1;
    int64_t last_init_id = 0;
    bool reset_frame = true;
    size_t expected_lidar_packets = 0;
    size_t batched_lidar_packets = 0;
    size_t dropped_packets = 0;
    // the reference's bookkeeping of what a partially assembled frame holds (lidar_frame.cpp:1455, 1497, 1438): channel
    // columns below next_valid_m_id are decoded or zeroed, the ones above still carry the frame's previous contents
    uint32_t next_valid_m_id = 0, next_headers_m_id = 0;
    std::weak_ptr<impl::PendingDecode> pending;     // what frames being assembled are told to call (they own it, and through it this state)

    // packets of the frame being assembled, in arrival order -- in pool (page-locked) memory: the decode kernel reads them
    // where they were staged, no upload (include/ouster_hip.h, "host containers")
    impl::HostArray<uint8_t> staged;
    size_t staged_count = 0;
    size_t stride = 0;

    // device side, created lazily for the frame layout in use
    ouster_hip_format* fmt = nullptr;
    std::vector<std::string> fmt_fields;
    std::vector<uint32_t> fmt_elems;
    hip::DeviceBuffer d_packets, d_out;
    // Frames of 64 packets or more: the staged packets are also sent to the device in pieces of UPLOAD_CHUNK while the rest
    // of the frame is still arriving (asynchronous copies out of the page-locked staging buffer), so that the release launch
    // reads them from HBM and only its stores cross the link: a copy-engine upload and a kernel's stores to host memory share
    // the link (2.1 MB up + 3.96 MB down together 93 us, tools/copybench), a kernel's own reads and stores do not (0.145 ms).
    static constexpr size_t UPLOAD_CHUNK = 32;
    size_t uploaded = 0;           // packets of the frame under assembly already queued for upload into d_packets
    bool chunked() const { return expected_lidar_packets >= 64 && !sink; }
    // destaggered planes of the last released frame, kept in HBM (impl::MirrorPlane) and the host planes they belong to
    std::shared_ptr<hip::DeviceBuffer> d_mirror;
    std::vector<const void*> mirror_keys;
    std::shared_ptr<const std::vector<int>> mirror_shifts;
    std::shared_ptr<impl::XyzWish> xyz_wish = std::make_shared<impl::XyzWish>();   // the LUT the caller projects released frames with
    FrameBatcher::PacketSink sink;  // set: released frames go here instead of being decoded
    // this batcher's own context (stream + scratch) on the GPU that was current when it first needed
    // one: distinct batchers never share mutable GPU state, like the reference's CPU batchers
    std::shared_ptr<hip::Context> ctx;
    int device = -1;
    const std::shared_ptr<hip::Context>& context() {
        if (!ctx) ctx = std::make_shared<hip::Context>(device >= 0 ? device : hip::current_device());
        return ctx;
    }

    ~State() {
        for (const void* k : mirror_keys) impl::mirror_forget(k);
        if (fmt) ouster_hip_format_destroy(fmt);
    }
};

FrameBatcher::FrameBatcher(const std::shared_ptr<SensorInfo>& info)
    : pf(get_format(*info)), s_(new State()) {
    if (info->format.columns_per_packet == 0)
        throw std::invalid_argument("unexpected columns_per_packet: 0");
    if (info->format.pixels_per_column == 0)
        throw std::invalid_argument("unexpected pixels_per_column: 0");
    s_->info = info;
    s_->last_init_id = info->init_id;
    s_->expected_lidar_packets = static_cast<size_t>(info->format.lidar_packets_per_frame());
    s_->stride = (pf.lidar_packet_size + 15) & ~size_t{15};
}
FrameBatcher::FrameBatcher(const SensorInfo& info) : FrameBatcher(std::make_shared<SensorInfo>(info)) {}
FrameBatcher::FrameBatcher(FrameBatcher&&) noexcept = default;
FrameBatcher::~FrameBatcher() = default;
void FrameBatcher::set_device(int device) {
    if (s_->ctx && s_->ctx->device() != device)
        throw std::logic_error("FrameBatcher::set_device: the batcher already works on another GPU");
    s_->device = device;
}

void FrameBatcher::reset() {
    s_->reset_frame = true;
    s_->finished_frame_id = -1;
    s_->batched_lidar_packets = 0;
    s_->staged_count = 0;
    s_->uploaded = 0;
    s_->next_valid_m_id = s_->next_headers_m_id = 0;
    s_->pending.reset();
    s_->cache.clear();
}
void FrameBatcher::set_packet_sink(PacketSink sink) { s_->sink = std::move(sink); }
size_t FrameBatcher::batched_packets() const { return s_->batched_lidar_packets; }
size_t FrameBatcher::dropped_packets() const { return s_->dropped_packets; }
void FrameBatcher::set_max_cache_size(size_t n) {
    if (n == 0) throw std::invalid_argument("max_cache_size must be > 0");
    s_->max_cache_size = n;
}
size_t FrameBatcher::get_max_cache_size() const { return s_->max_cache_size; }

// start_frame :1709-1741, batch_lidar_packet :1530-1576, finalize_frame :1905-1927
struct BatcherOps {
    static void start_frame(FrameBatcher::State& s, const PacketFormat& pf, int64_t f_id,
                            const uint8_t* packet_buf, LidarFrame& frame) {
        s.finished_frame_id = -1;
        s.batched_lidar_packets = 0;
        s.staged_count = 0;
        s.uploaded = 0;
        s.next_valid_m_id = s.next_headers_m_id = 0;
        s.pending.reset();              // notes left on frames of an earlier assembly are void: their packets are gone
        frame.clear_pending_decode();
        frame.frame_id = f_id;
        frame.timestamp().setZero();
        frame.measurement_id().setZero();
        frame.status().setZero();
        frame.packet_timestamp().setZero();
        const auto th = static_cast<uint8_t>(pf.thermal_shutdown(packet_buf));
        const auto sl = static_cast<uint8_t>(pf.shot_limiting(packet_buf));
        frame.frame_status = static_cast<uint64_t>(th & 0x0f) | (static_cast<uint64_t>(sl & 0x0f) << 4);
        frame.shutdown_countdown = pf.countdown_thermal_shutdown(packet_buf);
        frame.shot_limiting_countdown = pf.countdown_shot_limiting(packet_buf);
        frame.sensor_info = s.info;
    }

    // batch_lidar_packet (lidar_frame.cpp:1530-1576): packet-level values now, pixels later
    static void stage_packet(FrameBatcher::State& s, const PacketFormat& pf, const uint8_t* buf,
                             size_t len, uint64_t host_ts, LidarFrame& frame) {
        const uint16_t packet_id = static_cast<uint16_t>(
            pf.col_measurement_id(pf.nth_col(0, buf)) / pf.columns_per_packet);
        if (packet_id < frame.packet_timestamp().rows()) {
            frame.packet_timestamp()[packet_id] = host_ts;
            frame.alert_flags()[packet_id] = pf.alert_flags(buf);
        }
        if ((s.staged_count + 1) * s.stride > s.staged.size()) {   // first packet: room for a whole frame; more only for duplicates
            impl::HostArray<uint8_t> grown(std::max<size_t>({s.staged.size() * 2, (s.staged_count + 8) * s.stride,
                                                             (s.expected_lidar_packets + 4) * s.stride}),
                                           impl::uninitialized);
            if (s.staged_count) std::memcpy(grown.data(), s.staged.data(), s.staged_count * s.stride);
            if (s.uploaded) {   // copies may still be reading the block that is about to go back to the pool
                hip::ScopedContext on_my_context(s.context());
                hip::check(ouster_hip_sync(hip::default_ctx()));
            }
            s.staged = std::move(grown);
        }
        uint8_t* dst = s.staged.data() + s.staged_count * s.stride;
        std::memcpy(dst, buf, std::min(len, pf.lidar_packet_size));
        if (len < s.stride) std::memset(dst + std::min(len, pf.lidar_packet_size), 0, s.stride - std::min(len, pf.lidar_packet_size));
        s.staged_count++;
        s.batched_lidar_packets++;
        track_columns(s, pf, dst, frame);
        if (!s.sink) frame.set_pending_decode(pending_note(s, pf));
        if (s.chunked() && s.staged_count - s.uploaded >= FrameBatcher::State::UPLOAD_CHUNK && s.staged_count < s.expected_lidar_packets)
            upload_staged(s, s.staged_count);
    }
    // queue staged packets [uploaded, upto) for upload (asynchronous, the batcher's stream; the staging block is page-locked)
    static void upload_staged(FrameBatcher::State& s, size_t upto) {
        if (upto <= s.uploaded) return;
        hip::ScopedContext on_my_context(s.context());
        const size_t need = std::max<size_t>(s.staged.size(), (s.expected_lidar_packets + 4) * s.stride);
        if (s.d_packets.size() < need) {
            if (s.uploaded) hip::check(ouster_hip_sync(hip::default_ctx()));
            s.d_packets.resize(need);
            s.uploaded = 0;       // a new buffer: everything staged so far goes up again
        }
        s.d_packets.upload_async(s.staged.data() + s.uploaded * s.stride, (upto - s.uploaded) * s.stride, s.uploaded * s.stride);
        s.uploaded = upto;
    }
    static std::shared_ptr<impl::PendingDecode> pending_note(FrameBatcher::State& s, const PacketFormat& pf);

    // Which columns of the frame this packet settles, as the reference's two parse paths do it (batch_lidar_packet
    // :1541-1573 chooses; parse_by_block :1492-1500, parse_by_col :1422-1466), and the RAW_HEADERS plane, which is header
    // bytes only and is written here, on the host, packet by packet (PackRawHeadersCol :1328-1363).
    static void track_columns(FrameBatcher::State& s, const PacketFormat& pf, const uint8_t* buf, LidarFrame& frame) {
        const bool raw = impl::raw_headers_enabled(pf, frame);
        const uint32_t cpp = pf.columns_per_packet, W = static_cast<uint32_t>(frame.w);
        uint32_t block = static_cast<uint32_t>(pf.block_parsable());
        for (uint32_t ic = 0; ic < cpp && block; ++ic) {
            const uint8_t* col = pf.nth_col(ic, buf);
            if (!(pf.col_status(col) & 1u) || pf.col_measurement_id(col) >= W) block = 0;
        }
        for (uint32_t ic = 0; ic < cpp && block; ic += block)
            if (pf.col_measurement_id(pf.nth_col(ic, buf)) + block > W) block = 0;
        if (block && !raw) {
            const uint32_t first = pf.col_measurement_id(pf.nth_col(0, buf));
            if (first >= s.next_valid_m_id) s.next_valid_m_id = first + cpp;
            return;
        }
        Field* rh = raw ? frame.peek_field(ChanField::RAW_HEADERS) : nullptr;
        for (uint32_t ic = 0; ic < cpp; ++ic) {
            const uint8_t* col = pf.nth_col(ic, buf);
            const uint32_t m_id = pf.col_measurement_id(col);
            if (m_id >= W) continue;
            if (rh) {
                if (m_id >= s.next_headers_m_id) {
                    zero_raw_header_cols(*rh, s.next_headers_m_id, m_id);
                    s.next_headers_m_id = m_id + 1;
                }
                // rows of column m_id: column header, column footer, packet header, packet footer, in elements of the field
                const size_t es = rh->element_size(), Wf = rh->shape()[1];
                uint8_t* base = static_cast<uint8_t*>(rh->get());
                size_t row = 0;
                auto put = [&](const uint8_t* src, size_t bytes) {
                    for (size_t k = 0; k < bytes / es; ++k, ++row) std::memcpy(base + (row * Wf + m_id) * es, src + k * es, es);
                };
                put(col, pf.col_header_size);
                put(col + pf.col_size - pf.col_footer_size, pf.col_footer_size);
                put(buf, pf.packet_header_size);
                put(pf.footer(buf), pf.packet_footer_size);
            }
            if ((pf.col_status(col) & 1u) && m_id >= s.next_valid_m_id) s.next_valid_m_id = m_id + 1;
        }
    }
    static void zero_raw_header_cols(Field& rh, size_t from, size_t to) {
        const size_t es = rh.element_size(), H = rh.shape()[0], Wf = rh.shape()[1];
        uint8_t* base = static_cast<uint8_t*>(rh.get());
        if (to > Wf) to = Wf;
        for (size_t r = 0; r < H && from < to; ++r) std::memset(base + (r * Wf + from) * es, 0, (to - from) * es);
    }

    static bool frame_complete(const FrameBatcher::State& s, const PacketFormat& pf,
                               const LidarFrame& frame) {
        if (pf.udp_profile_lidar == UDPProfileLidar::OFF) return true;
        return s.batched_lidar_packets >= s.expected_lidar_packets &&
               frame.packet_timestamp().count() == s.expected_lidar_packets;
    }

    // finalize_frame (lidar_frame.cpp:1905-1927) + the deferred decode of the whole frame
    static void finalize_frame(FrameBatcher::State& s, const PacketFormat& pf, LidarFrame& frame) {
        if (s.sink) {
            std::vector<const uint8_t*> ptrs(s.staged_count);
            for (size_t i = 0; i < s.staged_count; ++i) ptrs[i] = s.staged.data() + i * s.stride;
            s.sink(ptrs);
        } else {
            frame.clear_pending_decode();
            decode_staged(s, pf, frame, frame.w);
        }
        if (impl::raw_headers_enabled(pf, frame))
            zero_raw_header_cols(*frame.peek_field(ChanField::RAW_HEADERS), s.next_headers_m_id, frame.w);
        if (frame.sensor_info && frame.sensor_info->init_id == s.last_init_id &&
            frame.frame_id <= s.last_frame_id && pf.header_type == HeaderType::FUSA)
            throw std::runtime_error("32-bit frame id did not increase since the last frame");
        s.finished_frame_id = frame.frame_id;
        s.last_frame_id = frame.frame_id;
        s.batched_lidar_packets = 0;
        s.staged_count = 0;
        s.uploaded = 0;
    }

    // one GPU launch: every plane the frame shares with the packet format + column headers
    // `col_limit` < W: a look at a frame that is still being assembled -- only the channel columns the reference would
    // have settled by now (below next_valid_m_id) are taken over, the rest of every plane keeps what it held
    static void decode_staged(FrameBatcher::State& s, const PacketFormat& pf, LidarFrame& frame, size_t col_limit) {
        hip::ScopedContext on_my_context(s.context());
        std::vector<std::pair<std::string, uint32_t>> fields;
        std::vector<bool> nan;
        std::vector<Field*> dst;
        for (auto it = pf.begin(); it != pf.end(); ++it) {  // foreach_channel_field order
            if (!frame.has_field(it->first)) continue;
            Field& f = frame.field(it->first);
            size_t extra = 1;
            for (size_t i = 2; i < f.shape().size(); ++i) extra *= f.shape()[i];
            fields.emplace_back(it->first, static_cast<uint32_t>(f.element_size() * extra));
            nan.push_back(f.tag() == ChanFieldType::FLOAT16);
            dst.push_back(&f);
        }
        std::vector<std::string> names;
        std::vector<uint32_t> elems;
        for (const auto& p : fields) {
            names.push_back(p.first);
            elems.push_back(p.second);
        }
        ouster_hip_ctx* ctx = hip::default_ctx();
        if (!s.fmt || names != s.fmt_fields || elems != s.fmt_elems) {
            if (s.fmt) ouster_hip_format_destroy(s.fmt);
            s.fmt = nullptr;
            ouster_hip_format_desc d;
            pf.fill_hip_desc(static_cast<uint32_t>(frame.w), fields, nan, d);
            hip::check(ouster_hip_format_create(ctx, &d, &s.fmt));
            s.fmt_fields = names;
            s.fmt_elems = elems;
        }
        const size_t W = frame.w, H = frame.h;
        const uint32_t count = static_cast<uint32_t>(s.staged_count);
        const size_t slots = std::max<size_t>(s.staged_count, 1);
        ouster_hip_frame_out out{};
        out.xyz_field[0] = out.xyz_field[1] = -1;
        void* h_ts = frame.timestamp().data();
        void* h_st = frame.status().data();
        void* h_mid = frame.measurement_id().data();
        // the planes are about to change: nothing mirrored from them (by this batcher's last release, or anybody's) stays valid
        for (const void* k : s.mirror_keys) impl::mirror_forget(k);
        s.mirror_keys.clear();
        if (impl::mirrors_live())
            for (Field* f : dst) impl::mirror_forget(f->storage_());

        // (1) The frame's planes and headers and the staged packets are pool memory (what LidarFrame and this batcher
        // allocate): ONE launch that reads the packets and writes the frame where they lie.  Input and output cross the link
        // at the same time; nothing is copied, nothing is allocated.
        bool in_place = col_limit >= W && s.staged_count && hip::is_device_accessible(s.staged.data(), s.staged_count * s.stride) &&
                        hip::is_device_accessible(h_ts, W * 8) && hip::is_device_accessible(h_st, W * 4) &&
                        hip::is_device_accessible(h_mid, W * 2);
        for (size_t i = 0; i < dst.size() && in_place; ++i) in_place = hip::is_device_accessible(dst[i]->storage_(), H * W * elems[i]);
        if (in_place) {
            for (size_t i = 0; i < dst.size(); ++i) out.planes[i] = dst[i]->storage_();
            out.timestamp = static_cast<uint64_t*>(h_ts);
            out.status = static_cast<uint32_t*>(h_st);
            out.measurement_id = static_cast<uint16_t*>(h_mid);
            // By-product kept in HBM: the destaggered form of every plane nobody holds a writable pointer to (the kernel has
            // the tile in LDS; the extra stores are HBM stores).  destagger(plane) of the released frame is then one copy out
            // (26 us per MB) instead of a kernel over the link in both directions (50 us per MB, tools/copybench).
            static const bool mirror_on = [] { const char* e = std::getenv("OUSTER_HIP_MIRROR"); return !e || std::atoi(e) != 0; }();
            const std::vector<int>& shifts = s.info->format.pixel_shift_by_row;
            std::vector<size_t> moff(dst.size(), SIZE_MAX);
            size_t mtotal = 0;
            if (mirror_on && shifts.size() == H) {
                for (size_t i = 0; i < dst.size(); ++i) {
                    if (dst[i]->writable_escaped_() || (elems[i] != 1 && elems[i] != 2 && elems[i] != 4 && elems[i] != 8)) continue;
                    moff[i] = mtotal;
                    mtotal += (H * W * elems[i] + 255) & ~size_t{255};
                }
            }
            // ... and the clouds of the range planes, when the caller has been projecting this batcher's frames (XyzWish)
            std::shared_ptr<const impl::DeviceLut> xl;
            bool xf64 = true;
            {
                std::lock_guard<std::mutex> g(s.xyz_wish->mu);
                xl = s.xyz_wish->lut;
                xf64 = s.xyz_wish->f64;
            }
            size_t xoff[2] = {SIZE_MAX, SIZE_MAX};
            int xfield[2] = {-1, -1};
            const ouster_hip_lut* xl_handle = nullptr;
            if (mtotal && xl && xl->handle && xl->device == s.context()->device()) {
                int k = 0;
                for (size_t i = 0; i < dst.size() && k < 2; ++i)
                    if (moff[i] != SIZE_MAX && elems[i] == 4 && (names[i] == ChanField::RANGE || names[i] == ChanField::RANGE2)) {
                        xfield[k] = static_cast<int>(i);
                        xoff[k] = mtotal;
                        mtotal += (H * W * 3 * (xf64 ? 8 : 4) + 255) & ~size_t{255};
                        ++k;
                    }
                if (k) xl_handle = xl->handle;
            }
            if (mtotal) {
                if (!s.d_mirror || s.d_mirror->size() < mtotal) s.d_mirror = std::make_shared<hip::DeviceBuffer>(mtotal);
                if (!s.mirror_shifts || *s.mirror_shifts != shifts) s.mirror_shifts = std::make_shared<const std::vector<int>>(shifts);
                for (size_t i = 0; i < dst.size(); ++i)
                    if (moff[i] != SIZE_MAX) out.destaggered[i] = static_cast<uint8_t*>(s.d_mirror->data()) + moff[i];
            }
            if (xl_handle) {
                for (int k = 0; k < 2; ++k)
                    if (xfield[k] >= 0) {
                        out.xyz[k] = static_cast<uint8_t*>(s.d_mirror->data()) + xoff[k];
                        out.xyz_field[k] = xfield[k];
                    }
                out.xyz_dtype = xf64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32;
            }
            static_assert(sizeof(int) == sizeof(int32_t), "pixel_shift_by_row is handed over as int32");
            const uint8_t* pk = s.staged.data();
            if (s.chunked()) {   // most of the frame is in HBM already: the rest follows, the launch reads all of it there
                upload_staged(s, s.staged_count);
                pk = static_cast<const uint8_t*>(s.d_packets.data());
            }
            hip::check(ouster_hip_decode(ctx, s.fmt, pk, s.stride, static_cast<uint32_t>(slots), &count, 1, nullptr,
                                         &out, mtotal ? reinterpret_cast<const int32_t*>(shifts.data()) : nullptr,
                                         xl_handle ? &xl_handle : nullptr, xl_handle ? 1 : 0));
            hip::check(ouster_hip_sync(ctx));
            for (size_t i = 0; i < dst.size() && mtotal; ++i) {
                if (moff[i] == SIZE_MAX) continue;
                impl::MirrorPlane m;
                m.host = dst[i]->storage_();
                m.d_destaggered = out.destaggered[i];
                m.h = H;
                m.w = W;
                m.elem = elems[i];
                m.device = s.context()->device();
                m.shifts = s.mirror_shifts;
                m.keep = s.d_mirror;
                m.wish = s.xyz_wish;
                for (int k = 0; k < 2; ++k)
                    if (xl_handle && xfield[k] == static_cast<int>(i)) {
                        m.d_xyz = out.xyz[k];
                        m.xyz_lut = xl;
                        m.xyz_f64 = xf64;
                    }
                impl::mirror_register(m);
                s.mirror_keys.push_back(m.host);
            }
            return;
        }

        // (2) Anything else (a frame whose fields the caller adopted from foreign memory, a tiny frame below the pool's block
        // size, a look at a frame under assembly): device block -- planes, then timestamp / status / measurement_id -- in this
        // batcher's grow-only buffers, asynchronous copies, one synchronisation.
        auto al = [](size_t x) { return (x + 255) & ~size_t{255}; };
        std::vector<size_t> off(dst.size());
        size_t total = 0;
        for (size_t i = 0; i < dst.size(); ++i) {
            off[i] = total;
            total += al(H * W * elems[i]);
        }
        const size_t off_ts = total;
        total += al(W * 8);
        const size_t off_st = total;
        total += al(W * 4);
        const size_t off_mid = total;
        total += al(W * 2);
        s.d_out.resize(total);
        if (s.chunked()) {
            upload_staged(s, s.staged_count);
            if (s.d_packets.size() < slots * s.stride) s.d_packets.resize(slots * s.stride);
        } else {
            s.d_packets.resize(slots * s.stride);
            if (s.staged_count) s.d_packets.upload_async(s.staged.data(), s.staged_count * s.stride);
        }
        uint8_t* base = static_cast<uint8_t*>(s.d_out.data());
        for (size_t i = 0; i < dst.size(); ++i) out.planes[i] = base + off[i];
        out.timestamp = reinterpret_cast<uint64_t*>(base + off_ts);
        out.status = reinterpret_cast<uint32_t*>(base + off_st);
        out.measurement_id = reinterpret_cast<uint16_t*>(base + off_mid);
        hip::check(ouster_hip_decode(ctx, s.fmt, static_cast<const uint8_t*>(s.d_packets.data()),
                                     s.stride, static_cast<uint32_t>(slots), &count, 1, nullptr, &out,
                                     nullptr, nullptr, 0));
        s.d_out.download_async(h_ts, W * 8, off_ts);
        s.d_out.download_async(h_st, W * 4, off_st);
        s.d_out.download_async(h_mid, W * 2, off_mid);
        if (col_limit >= W) {
            for (size_t i = 0; i < dst.size(); ++i) s.d_out.download_async(dst[i]->storage_(), H * W * elems[i], off[i]);
            hip::check(ouster_hip_sync(ctx));
        } else if (col_limit > 0) {
            hip::check(ouster_hip_sync(ctx));
            std::vector<uint8_t> tmp;
            for (size_t i = 0; i < dst.size(); ++i) {
                tmp.resize(H * W * elems[i]);
                s.d_out.download(tmp.data(), tmp.size(), off[i]);
                uint8_t* out_plane = static_cast<uint8_t*>(dst[i]->storage_());
                for (size_t r = 0; r < H; ++r)
                    std::memcpy(out_plane + r * W * elems[i], tmp.data() + r * W * elems[i], col_limit * elems[i]);
            }
        } else {
            hip::check(ouster_hip_sync(ctx));
        }
    }

    static int top_of_cache(const FrameBatcher::State& s, const PacketFormat& pf) {
        int best = 0;  // lowest frame id first, ties in arrival order (lidar_frame.h:970-993)
        for (size_t i = 1; i < s.cache.size(); ++i) {
            const int d = pf.frame_id_difference(pf.frame_id(s.cache[best].buf.data()),
                                                 pf.frame_id(s.cache[i].buf.data()));
            if (d < 0 || (d == 0 && s.cache[i].seq < s.cache[best].seq)) best = static_cast<int>(i);
        }
        return best;
    }
};

namespace {
// what a frame under assembly calls when somebody looks at it (lidar_frame.h: impl::PendingDecode).  It refers to the
// batcher's heap state, which stays where it is when the FrameBatcher object is moved.
struct BatcherPending : impl::PendingDecode {
    std::shared_ptr<FrameBatcher::State> state;   // shared: the frame can still be looked at after the batcher is gone
    PacketFormat pf;
    bool standalone;   // FrameBatcher::flush(): an explicit request, not a note left on a frame
    BatcherPending(std::shared_ptr<FrameBatcher::State> s, const PacketFormat& f, bool alone = false)
        : state(std::move(s)), pf(f), standalone(alone) {}
    void flush(LidarFrame& frame) override {
        // a note is good for the frame the batcher is assembling NOW: start_frame / reset() retire the earlier ones
        if (!standalone && state->pending.lock().get() != this) return;
        if (frame.frame_id != -1 && state->finished_frame_id < 0 && !state->sink && state->staged_count) {
            frame.clear_pending_decode();
            BatcherOps::decode_staged(*state, pf, frame, std::min<size_t>(state->next_valid_m_id, frame.w));
        }
    }
};
}  // namespace

std::shared_ptr<impl::PendingDecode> BatcherOps::pending_note(FrameBatcher::State& s, const PacketFormat& pf) {
    std::shared_ptr<impl::PendingDecode> p = s.pending.lock();
    if (!p) {
        p = std::make_shared<BatcherPending>(s.shared_from_this(), pf);
        s.pending = p;
    }
    return p;
}

void FrameBatcher::flush(LidarFrame& frame) { BatcherPending(s_, pf, true).flush(frame); }

bool FrameBatcher::batch(const Packet& packet, LidarFrame& frame) {
    State& s = *s_;
    if (s.reset_frame) {
        frame.frame_id = -1;
        s.reset_frame = false;
    }
    if (packet.type() == PacketType::Imu || packet.type() == PacketType::Zone) return false;
    if (frame.w != s.info->format.columns_per_frame || frame.h != s.info->format.pixels_per_column)
        throw std::invalid_argument("unexpected frame dimensions");
    if (frame.packet_timestamp().rows() != frame.w / pf.columns_per_packet)
        throw std::invalid_argument("unexpected frame columns_per_packet: " +
                                    std::to_string(pf.columns_per_packet));
    const uint8_t* buf = packet.buf.data();
    const size_t len = packet.buf.size();
    auto push_cache = [&] { s.cache.push_back({packet.buf, packet.host_timestamp, s.seq++}); };
    auto maybe_release = [&]() {
        if (BatcherOps::frame_complete(s, pf, frame)) {
            BatcherOps::finalize_frame(s, pf, frame);
            return true;
        }
        return false;
    };

    // sensor re-initialisation (lidar_frame.cpp:1795-1822)
    if (pf.udp_profile_lidar != UDPProfileLidar::LEGACY &&
        static_cast<int64_t>(pf.init_id(buf)) != s.last_init_id) {
        s.last_init_id = pf.init_id(buf);
        if (frame.frame_id == -1 || s.finished_frame_id >= 0) {
            reset();
            s.reset_frame = false;
            BatcherOps::start_frame(s, pf, pf.frame_id(buf), buf, frame);
            BatcherOps::stage_packet(s, pf, buf, len, packet.host_timestamp, frame);
            return maybe_release();
        }
        BatcherOps::finalize_frame(s, pf, frame);
        reset();
        push_cache();
        return true;
    }

    const int64_t f_id = pf.frame_id(buf);
    if (s.cache.empty()) {
        if (s.finished_frame_id >= 0 &&
            pf.frame_id_difference(static_cast<uint32_t>(s.finished_frame_id),
                                   static_cast<uint32_t>(f_id)) <= 0) {
            s.dropped_packets++;  // late packet of a frame already released
            return false;
        }
        if (frame.frame_id == -1 || s.finished_frame_id >= 0) {
            BatcherOps::start_frame(s, pf, f_id, buf, frame);
            BatcherOps::stage_packet(s, pf, buf, len, packet.host_timestamp, frame);
            return maybe_release();
        }
    }
    if (frame.frame_id == f_id && s.finished_frame_id < 0) {
        BatcherOps::stage_packet(s, pf, buf, len, packet.host_timestamp, frame);
        return maybe_release();
    }

    // batch_with_caching (lidar_frame.cpp:1743-1793)
    push_cache();
    while (!s.cache.empty()) {
        const int t = BatcherOps::top_of_cache(s, pf);
        const uint8_t* tb = s.cache[t].buf.data();
        const int64_t tf = pf.frame_id(tb);
        if (s.finished_frame_id >= 0 &&
            pf.frame_id_difference(static_cast<uint32_t>(s.finished_frame_id),
                                   static_cast<uint32_t>(tf)) <= 0) {
            s.dropped_packets++;
            s.cache.erase(s.cache.begin() + t);
            continue;
        }
        if (frame.frame_id == -1 || s.finished_frame_id >= 0)
            BatcherOps::start_frame(s, pf, tf, tb, frame);
        const int diff = pf.frame_id_difference(static_cast<uint32_t>(frame.frame_id),
                                                static_cast<uint32_t>(tf));
        if (diff < 0) {
            s.dropped_packets++;
            s.cache.erase(s.cache.begin() + t);
        } else if (diff > 0) {
            if (s.cache.size() >= s.max_cache_size) {  // give up waiting for the current frame
                BatcherOps::finalize_frame(s, pf, frame);
                return true;
            }
            return false;
        } else {
            BatcherOps::stage_packet(s, pf, tb, s.cache[t].buf.size(), s.cache[t].host_timestamp,
                                     frame);
            s.cache.erase(s.cache.begin() + t);
            if (maybe_release()) return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------
// frame_to_packets (host; test-side packet synthesis)
// ---------------------------------------------------------------------------------------
namespace impl {

bool raw_headers_enabled(const PacketFormat& pf, const LidarFrame& frame) {
    const Field* fp = frame.peek_field(ChanField::RAW_HEADERS);
    if (!fp) return false;
    const Field& f = *fp;
    if (f.shape().size() != 2) return false;
    return pf.pixels_per_column * f.element_size() >=
           pf.packet_header_size + pf.col_header_size + pf.col_footer_size + pf.packet_footer_size;
}

namespace {
template <typename T>
void pack_plane(const PacketFormat& pf, const Field& f, const std::string& name, int cols,
                uint8_t* buf) {
    pf.set_block<T>(static_cast<const T*>(f.get()), cols, name, buf);
}
}  // namespace

std::vector<LidarPacket> frame_to_packets(const LidarFrame& frame,
                                          std::shared_ptr<PacketFormat> pf, uint32_t init_id,
                                          uint64_t prod_sn) {
    if (!pf) throw std::invalid_argument("Null PacketFormat pointer");
    const size_t total = frame.packet_timestamp().size();
    if (frame.w / pf->columns_per_packet != total)
        throw std::invalid_argument(
            "Mismatch between expected number of packets and PacketFormat.columns_per_packet");
    std::vector<LidarPacket> out;
    const uint32_t cpp = pf->columns_per_packet;
    for (size_t p = 0; p < total; ++p) {
        LidarPacket pkt(pf);
        uint8_t* buf = pkt.buf.data();
        pkt.host_timestamp = frame.packet_timestamp()[p];
        pf->set_shutdown(buf, static_cast<uint8_t>(frame.thermal_shutdown()));
        pf->set_shot_limiting(buf, static_cast<uint8_t>(frame.shot_limiting()));
        pf->set_shutdown_countdown(buf, static_cast<uint8_t>(frame.shutdown_countdown));
        pf->set_shot_limiting_countdown(buf, static_cast<uint8_t>(frame.shot_limiting_countdown));
        pf->set_frame_id(buf, static_cast<uint32_t>(frame.frame_id));
        pf->set_init_id(buf, init_id);
        pf->set_prod_sn(buf, prod_sn);
        pf->set_packet_type(buf, 0x1);
        pf->set_alert_flags(buf, frame.alert_flags()[p]);
        bool any_valid = false;
        for (uint32_t icol = 0; icol < cpp; ++icol) {
            uint8_t* col = pf->nth_col(icol, buf);
            const size_t id = p * cpp + icol;
            pf->set_col_status(col, frame.status()[id]);
            pf->set_col_measurement_id(col, static_cast<uint16_t>(id));
            pf->set_col_timestamp(col, frame.timestamp()[id]);
            any_valid |= (frame.status()[id] & 0x01) != 0;
        }
        if (!any_valid && !pkt.host_timestamp) continue;  // nothing to say: not emitted
        for (auto it = pf->begin(); it != pf->end(); ++it) {
            if (!frame.has_field(it->first)) continue;
            const Field& f = frame.field(it->first);
            const int cols = static_cast<int>(frame.w);
            if (f.shape().size() == 3) {
                pack_plane<float3x16_t>(*pf, f, it->first, cols, buf);
                continue;
            }
            switch (f.tag()) {
                case ChanFieldType::UINT8: pack_plane<uint8_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::UINT16: pack_plane<uint16_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::UINT32: pack_plane<uint32_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::UINT64: pack_plane<uint64_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::INT8: pack_plane<int8_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::INT16: pack_plane<int16_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::INT32: pack_plane<int32_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::INT64: pack_plane<int64_t>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::FLOAT32: pack_plane<float>(*pf, f, it->first, cols, buf); break;
                case ChanFieldType::FLOAT64: pack_plane<double>(*pf, f, it->first, cols, buf); break;
                default: break;
            }
        }
        if (raw_headers_enabled(*pf, frame)) {
            // the recorded headers and footers replace the ones written above, checksum included
            // (PacketFormat::unpack_raw_headers, types.h:1122-1164; element types wider than 32 bit are refused there)
            const Field& rh = frame.field(ChanField::RAW_HEADERS);
            const size_t es = rh.element_size(), Wf = rh.shape()[1];
            if (es > 4)
                throw std::invalid_argument("RAW_HEADERS field should be of typeuint32_t or smaller to work correctly");
            const uint8_t* base = static_cast<const uint8_t*>(rh.get());
            const size_t ch = pf->col_header_size / es, cf = pf->col_footer_size / es, ph = pf->packet_header_size / es,
                         pfo = pf->packet_footer_size / es;
            auto get = [&](uint8_t* dstp, size_t row0, size_t n, size_t m_id) {
                for (size_t k = 0; k < n; ++k) std::memcpy(dstp + k * es, base + ((row0 + k) * Wf + m_id) * es, es);
            };
            size_t m0 = pf->col_measurement_id(pf->nth_col(0, buf));
            get(buf, ch + cf, ph, m0);
            get(pf->footer(buf), ch + cf + ph, pfo, m0);
            for (uint32_t icol = 0; icol < cpp; ++icol) {
                uint8_t* col = pf->nth_col(icol, buf);
                const size_t m_id = pf->col_measurement_id(col);   // read BEFORE the header is replaced
                get(col, 0, ch, m_id);
                get(col + pf->col_size - pf->col_footer_size, ch, cf, m_id);
            }
        } else if (pf->udp_profile_lidar != UDPProfileLidar::LEGACY &&
            pf->header_type == HeaderType::STANDARD) {
            const uint64_t crc = pf->calculate_crc(buf, pkt.buf.size());
            std::memcpy(buf + pkt.buf.size() - sizeof crc, &crc, sizeof crc);
        }
        out.push_back(std::move(pkt));
    }
    return out;
}

}  // namespace impl

}  // namespace core
}  // namespace sdk
}  // namespace ouster
