// lidar_frame.h -- LidarFrame, FrameBatcher, destagger.
//
// Public surface follows the reference ouster_core/include/ouster/core/lidar_frame.h
// (LidarFrame :124-821, destagger :917-935, FrameBatcher :966-1145) and
// ouster_core/include/ouster/core/field.h (Field).  LidarFrame stays a host container of
// named row-major H x W planes (calloc'd, value semantics); what changes is who fills it:
// FrameBatcher collects a frame's packets and decodes them on the GPU in one launch,
// destagger<T>() runs the HIP kernel.  Device-resident, batched variants that avoid the
// PCIe round trip live in ouster/hip/device_batch.h.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>

#include "ouster/core/array_view.h"
#include "ouster/core/packet.h"
#include "ouster/core/types.h"

namespace ouster {
namespace sdk {
namespace core {

enum class FieldClass { NONE = 0, PIXEL_FIELD = 1, COLUMN_FIELD = 2, PACKET_FIELD = 3, FRAME_FIELD = 4 };
std::string to_string(FieldClass c);   ///< field.cpp:81-87

/** Element type and full shape of a field to create (the array part of field.h:116-297). */
struct FieldDescriptor {
    ChanFieldType element_type = ChanFieldType::VOID;
    std::vector<size_t> shape;
    ChanFieldType tag() const { return element_type; }
    size_t size() const {
        size_t n = 1;
        for (size_t d : shape) n *= d;
        return n;
    }
    size_t bytes() const { return size() * field_type_size(element_type); }
    template <typename T>
    static FieldDescriptor array(std::vector<size_t> shape_in) {
        return FieldDescriptor{FieldTag<T>::tag, std::move(shape_in)};
    }
    static FieldDescriptor array(ChanFieldType tag, std::vector<size_t> shape_in) {
        return FieldDescriptor{tag, std::move(shape_in)};
    }
};
/** fd_array<T>(d0, d1, ...) / fd_array(tag, d0, d1, ...) (field.h:305-316) */
template <typename T, typename... Args>
FieldDescriptor fd_array(Args&&... args) {
    return FieldDescriptor::array<T>({static_cast<size_t>(args)...});
}
template <typename... Args>
FieldDescriptor fd_array(ChanFieldType tag, Args&&... args) {
    return FieldDescriptor::array(tag, {static_cast<size_t>(args)...});
}

/** Name, element type and extra dimensions of a LidarFrame field (lidar_frame.h:36-74). */
struct FieldType {
    std::string name;
    ChanFieldType element_type = ChanFieldType::VOID;
    std::vector<size_t> extra_dims;
    FieldClass field_class = FieldClass::PIXEL_FIELD;
    FieldType() = default;
    FieldType(std::string n, ChanFieldType t, std::vector<size_t> extra = {},
              FieldClass c = FieldClass::PIXEL_FIELD)
        : name(std::move(n)), element_type(t), extra_dims(std::move(extra)), field_class(c) {}
    bool operator==(const FieldType& o) const {
        return name == o.name && element_type == o.element_type && extra_dims == o.extra_dims &&
               field_class == o.field_class;
    }
    bool operator<(const FieldType& o) const { return name < o.name; }
};
using LidarFrameFieldTypes = std::vector<FieldType>;
std::string to_string(const FieldType& field_type);              ///< lidar_frame.cpp:1119-1147
std::string to_string(const LidarFrameFieldTypes& field_types);

/** Owning, zero-initialised, typed n-d buffer (field.h:828-905, field.cpp:247-296). */
namespace impl {
/** Device-resident by-products of a FrameBatcher's release (host_internal.h: FrameMirror): the destaggered form of every
 *  plane the batcher just decoded stays in HBM, keyed by the plane's host storage, so that destagger(plane) of the released
 *  frame is one copy out instead of a round trip.  An entry dies the moment the plane's storage can change: a writable
 *  pointer or view is handed out, the storage is freed or replaced, the batcher (or another one) decodes into it again.
 *  A Field that has EVER handed out a writable pointer is never mirrored -- such a pointer may be written through at any
 *  time without the library noticing.  mirrors_live() keeps the hooks at one relaxed load when nothing is mirrored. */
bool mirrors_live() noexcept;
void mirror_forget(const void* host_storage) noexcept;
}  // namespace impl

class Field {
   public:
    Field() = default;
    Field(ChanFieldType tag, std::vector<size_t> shape, FieldClass c = FieldClass::NONE);
    Field(const Field& o);
    Field(Field&& o) noexcept;
    Field& operator=(Field o) noexcept;
    ~Field();

    ChanFieldType tag() const { return tag_; }
    const std::vector<size_t>& shape() const { return shape_; }
    FieldClass field_class() const { return class_; }
    size_t element_size() const { return field_type_size(tag_); }
    size_t size() const { return count_; }   ///< number of elements
    size_t bytes() const { return count_ * element_size(); }
    void* get() { return writable_(); }
    const void* get() const { return ptr_; }
    /** Typed pointer. @throw std::invalid_argument on element type mismatch. */
    template <typename T> T* get() {
        check<T>();
        return static_cast<T*>(writable_());
    }
    template <typename T> const T* get() const {
        check<T>();
        return static_cast<const T*>(ptr_);
    }
    /** 2-D typed view. @throw std::invalid_argument on type / rank mismatch. */
    template <typename T> ImgRef<T> img() {
        check<T>();
        if (shape_.size() != 2) throw std::invalid_argument("Field: cannot convert to 2d image");
        return ImgRef<T>(static_cast<T*>(writable_()), shape_[0], shape_[1]);
    }
    template <typename T> ImgRef<const T> img() const {
        check<T>();
        if (shape_.size() != 2) throw std::invalid_argument("Field: cannot convert to 2d image");
        return ImgRef<const T>(static_cast<const T*>(ptr_), shape_[0], shape_[1]);
    }
    void set_zero();
    bool operator==(const Field& o) const;
    bool operator!=(const Field& o) const { return !(*this == o); }

    // ---- the reference's FieldView conversions (ouster_core/include/ouster/core/field.h:374-470) ----
    /** Typed pointer; `void` always converts.
     *  @throw std::invalid_argument("FieldView: ineligible dereference type ...") on a type mismatch */
    template <typename T> operator T*() { return conv_ptr<T>(std::is_const<T>::value); }
    template <typename T> operator const T*() const { return conv_ptr<const T>(); }
    /** n-d view.  @throw std::invalid_argument on a type or dimension mismatch */
    template <typename T, size_t Dim> operator ArrayView<T, Dim>() {
        check_rank<Dim>();
        return ArrayView<T, Dim>(conv_ptr<T>(std::is_const<T>::value), shape_);
    }
    template <typename T, size_t Dim> operator ConstArrayView<T, Dim>() const {
        check_rank<Dim>();
        return ConstArrayView<T, Dim>(conv_ptr<const T>(), shape_);
    }
    /** 2-D image view (the stand-in for the reference's `operator Eigen::Ref<img_t<T>>`). */
    template <typename T> operator ImgRef<T>() {
        check_2d();
        return ImgRef<T>(conv_ptr<T>(std::is_const<T>::value), shape_[0], shape_[1]);
    }
    template <typename T> operator ImgRef<const T>() const {
        check_2d();
        return ImgRef<const T>(conv_ptr<const T>(), shape_[0], shape_[1]);
    }
#ifdef OUSTER_HIP_USE_EIGEN
    template <typename T> operator Eigen::Ref<EigenImg<T>>() {
        check_2d();
        return Eigen::Map<EigenImg<T>>(conv_ptr<T>(false), static_cast<Eigen::Index>(shape_[0]),
                                       static_cast<Eigen::Index>(shape_[1]));
    }
    template <typename T> operator Eigen::Ref<const EigenImg<T>>() const {
        check_2d();
        return Eigen::Map<const EigenImg<T>>(conv_ptr<const T>(), static_cast<Eigen::Index>(shape_[0]),
                                             static_cast<Eigen::Index>(shape_[1]));
    }
#endif

   private:
    template <typename T> void check() const {
        if (FieldTag<T>::tag != tag_)
            throw std::invalid_argument("Field: ineligible dereference type");
    }
   public:
    /** Library-internal: the storage, without declaring a caller-held writable pointer (FrameBatcher decodes through it
     *  and settles the mirror itself). */
    void* storage_() const { return ptr_; }
    bool writable_escaped_() const { return escaped_; }

   private:
    template <typename T> T* conv_ptr(bool read_only = true) const {
        using NC = typename std::remove_const<T>::type;
        if (!std::is_void<NC>::value && FieldTag<NC>::tag != tag_)
            throw std::invalid_argument("FieldView: ineligible dereference type for field of element type " +
                                        to_string(tag_) + ". Dereference type must match or be void.");
        if (!read_only) const_cast<Field*>(this)->writable_();
        return static_cast<T*>(ptr_);
    }
    /** a writable pointer leaves the Field: from now on its contents may change behind the library's back */
    void* writable_() {
        escaped_ = true;
        if (impl::mirrors_live()) impl::mirror_forget(ptr_);
        return ptr_;
    }
    template <size_t Dim> void check_rank() const {
        if (shape_.size() != Dim)
            throw std::invalid_argument("FieldView: ArrayView conversion failed due to dimension mismatch. Expected " +
                                        std::to_string(shape_.size()) + " got " + std::to_string(Dim) + " dimensions.");
    }
    void check_2d() const {
        if (shape_.size() != 2)
            throw std::invalid_argument("Field: Eigen array conversion failed due to dimension mismatch. "
                                        "Underlying data has " + std::to_string(shape_.size()) +
                                        " dimensions but must have 2 dimensions.");
    }
    ChanFieldType tag_ = ChanFieldType::VOID;
    std::vector<size_t> shape_;
    FieldClass class_ = FieldClass::NONE;
    size_t count_ = 0;
    void* ptr_ = nullptr;
    bool escaped_ = false;   ///< a writable pointer / view of the storage was handed out at some point (travels with the storage)
};

/** 1-D header view (stands in for the Eigen headers returned by the reference): a contiguous VecRef. */
template <typename T>
class HeaderRef : public VecRef<T> {
   public:
    HeaderRef(T* p, size_t n) : VecRef<T>(p, n, 1), p_(p) {}
    T* data() const { return p_; }
    using VecRef<T>::operator=;

   private:
    T* p_;
};
namespace impl {
template <typename T> struct is_array_like<HeaderRef<T>> : std::true_type {};
template <typename T> T& flat_at(const HeaderRef<T>& a, size_t i) { return a(i); }
}  // namespace impl
template <typename T>
BoolMask operator==(const HeaderRef<T>& a, const HeaderRef<T>& b) {
    return impl::compare_arrays(a, b, [](const auto& x, const auto& y) { return x == y; });
}
template <typename T>
BoolMask operator!=(const HeaderRef<T>& a, const HeaderRef<T>& b) {
    return impl::compare_arrays(a, b, [](const auto& x, const auto& y) { return x != y; });
}

/** Default planes of a profile / data format (lidar_frame.cpp:73-259, :1038-1117). */
LidarFrameFieldTypes get_field_types(UDPProfileLidar profile);
LidarFrameFieldTypes get_field_types(const DataFormat& format, const Version& fw_version);
LidarFrameFieldTypes get_field_types(const SensorInfo& info);

class LidarFrame;
namespace impl {
/** A FrameBatcher's promise to a frame it is assembling: the packets collected so far are decoded (one GPU launch) the
 *  first time somebody looks at the frame's planes or column headers -- the reference parses every packet on arrival
 *  (lidar_frame.cpp:1530-1576), so a partially assembled frame shows what has been received; this mirror defers the
 *  pixel work to the release of the frame and pays for an early look only when one is taken. */
struct PendingDecode {
    virtual ~PendingDecode() = default;
    virtual void flush(LidarFrame& frame) = 0;
};
}  // namespace impl

class LidarFrame {
   public:
    size_t w{0};
    size_t h{0};
    int64_t frame_id{-1};
    uint64_t frame_status{0};
    uint16_t shutdown_countdown{0};
    uint16_t shot_limiting_countdown{0};
    std::shared_ptr<SensorInfo> sensor_info;

    LidarFrame();
    LidarFrame(const LidarFrame&);
    LidarFrame(LidarFrame&&) noexcept;
    LidarFrame& operator=(const LidarFrame&);
    LidarFrame& operator=(LidarFrame&&) noexcept;
    ~LidarFrame();

    explicit LidarFrame(const DataFormat& format);
    explicit LidarFrame(const SensorInfo& info);
    explicit LidarFrame(std::shared_ptr<SensorInfo> info);
    LidarFrame(std::shared_ptr<SensorInfo> info, const LidarFrameFieldTypes& field_types);
    /** Deprecated in the reference too: the LEGACY profile's fields, 16 columns per packet (lidar_frame.h:229). */
    LidarFrame(size_t h, size_t w);
    /** @throw std::invalid_argument for zero dims / zero columns_per_packet. */
    LidarFrame(size_t h, size_t w, const LidarFrameFieldTypes& field_types,
               size_t columns_per_packet = DEFAULT_COLUMNS_PER_PACKET);
    LidarFrame(size_t h, size_t w, UDPProfileLidar profile,
               size_t columns_per_packet = DEFAULT_COLUMNS_PER_PACKET);
    /** A frame like `other` with only the indicated fields: copied where the type matches, cast where only the element
     *  type differs, zero where `other` has no such field (lidar_frame.h:297-308, lidar_frame.cpp:361-401).
     *  @throw std::invalid_argument if a field's dimensions are incompatible */
    LidarFrame(const LidarFrame& other, const LidarFrameFieldTypes& fields);

    /** @throw std::out_of_range if the field does not exist (lidar_frame.cpp:422-436). */
    Field& field(const std::string& name);
    const Field& field(const std::string& name) const;
    template <typename T> ImgRef<T> field(const std::string& name) { return field(name).img<T>(); }
    template <typename T> ImgRef<const T> field(const std::string& name) const {
        return field(name).img<T>();
    }
    bool has_field(const std::string& name) const;
    Field& add_field(const FieldType& type);
    /** A field of the descriptor's full shape; pixel / column / packet fields must start with the frame's own extents
     *  (lidar_frame.cpp:463-510).  @throw std::invalid_argument with the reference's texts */
    Field& add_field(const std::string& name, const FieldDescriptor& desc, FieldClass field_class = FieldClass::PIXEL_FIELD);
    Field& add_field(const std::string& name, ChanFieldType type, std::vector<size_t> extra_dims = {},
                     FieldClass c = FieldClass::PIXEL_FIELD);
    Field del_field(const std::string& name);
    std::map<std::string, Field>& fields() { sync_(); return fields_; }
    const std::map<std::string, Field>& fields() const { sync_(); return fields_; }
    LidarFrameFieldTypes field_types() const;
    /** Type of one field (lidar_frame.h:476-484).  @throw std::out_of_range if the field does not exist */
    FieldType field_type(const std::string& name) const;

    HeaderRef<uint64_t> timestamp() { sync_(); return {timestamp_.get<uint64_t>(), w}; }
    HeaderRef<const uint64_t> timestamp() const { sync_(); return {timestamp_.get<uint64_t>(), w}; }
    HeaderRef<uint16_t> measurement_id() { sync_(); return {measurement_id_.get<uint16_t>(), w}; }
    HeaderRef<const uint16_t> measurement_id() const { sync_(); return {measurement_id_.get<uint16_t>(), w}; }
    HeaderRef<uint32_t> status() { sync_(); return {status_.get<uint32_t>(), w}; }
    HeaderRef<const uint32_t> status() const { sync_(); return {status_.get<uint32_t>(), w}; }
    HeaderRef<uint64_t> packet_timestamp() { return {packet_timestamp_.get<uint64_t>(), packet_count_}; }
    HeaderRef<const uint64_t> packet_timestamp() const {
        return {packet_timestamp_.get<uint64_t>(), packet_count_};
    }
    HeaderRef<uint8_t> alert_flags() { return {alert_flags_.get<uint8_t>(), packet_count_}; }
    HeaderRef<const uint8_t> alert_flags() const { return {alert_flags_.get<uint8_t>(), packet_count_}; }
    /** per-column 4x4 poses, identity initialised (lidar_frame.cpp:353-358) */
    Field& body_to_world() { return body_to_world_; }
    const Field& body_to_world() const { return body_to_world_; }
    /** deprecated spelling of body_to_world() (lidar_frame.h:740-753) */
    [[deprecated("LidarFrame::body_to_world()")]] Field& pose() { return body_to_world_; }
    [[deprecated("LidarFrame::body_to_world()")]] const Field& pose() const { return body_to_world_; }
    /** One column's 4x4 pose (lidar_frame.h:755-773).  @throw std::out_of_range if index is out of bounds */
    void set_column_pose(int index, const mat4d& pose);
    mat4d get_column_pose(int index) const;
    size_t packet_count() const { return packet_count_; }

    ThermalShutdownStatus thermal_shutdown() const {
        return static_cast<ThermalShutdownStatus>(frame_status & 0x0f);
    }
    ShotLimitingStatus shot_limiting() const {
        return static_cast<ShotLimitingStatus>((frame_status & 0xf0) >> 4);
    }
    /** true when every column in `window` has status bit 0 set */
    bool complete(ColumnWindow window) const;
    /** ... in the column window of the frame's SensorInfo (lidar_frame.h:783-791).
     *  @throw std::runtime_error if the frame has no sensor_info */
    bool complete() const;
    /** first / last column with status bit 0 set (lidar_frame.cpp:907-925).
     *  @throw std::runtime_error("No valid columns in LidarFrame") */
    int get_first_valid_column() const;
    int get_last_valid_column() const;
    /** host timestamp of the first / last / earliest / latest lidar packet with a valid column
     *  (lidar_frame.cpp:643-797; this mirror has no IMU / zone streams, so the stream-less overloads
     *  look at lidar packets only).  @throw std::runtime_error("No valid packets in LidarFrame") */
    uint64_t get_first_valid_packet_timestamp() const;
    uint64_t get_last_valid_packet_timestamp() const;
    uint64_t get_min_valid_packet_timestamp() const;
    uint64_t get_max_valid_packet_timestamp() const;
    /** same, 0 instead of throwing (lidar_frame.cpp:727-733) */
    uint64_t get_first_valid_lidar_packet_timestamp() const;
    uint64_t get_last_valid_lidar_packet_timestamp() const;

    bool equals(const LidarFrame& other) const;

    /** Used by FrameBatcher: a field without triggering a pending decode (nullptr: no such field). */
    Field* peek_field(const std::string& name) const {
        auto it = fields_.find(name);
        return it == fields_.end() ? nullptr : const_cast<Field*>(&it->second);
    }
    /** Used by FrameBatcher: the decode that the next look at this frame's data triggers (none: empty pointer). */
    void set_pending_decode(std::shared_ptr<impl::PendingDecode> p) const {
        pending_ = std::move(p);
        has_pending_ = true;
    }
    void clear_pending_decode() const {
        pending_.reset();
        has_pending_ = false;
    }

   private:
    void sync_() const {
        if (has_pending_) run_pending_();
    }
    void run_pending_() const;
    mutable std::shared_ptr<impl::PendingDecode> pending_;   ///< keeps the batcher's staged packets alive, even past the batcher
    mutable bool has_pending_ = false;
    Field timestamp_, measurement_id_, status_, packet_timestamp_, body_to_world_, alert_flags_;
    std::map<std::string, Field> fields_;
    size_t packet_count_{0};
};

/** Human-readable summary: extents, frame id and status, field types, min / mean / max of every numeric field
 *  (lidar_frame.cpp:1149-1250). */
std::string to_string(const LidarFrame& lidar_frame);
bool operator==(const LidarFrame& a, const LidarFrame& b);
inline bool operator!=(const LidarFrame& a, const LidarFrame& b) { return !(a == b); }

// ---------------------------------------------------------------------------------------
// destagger / stagger (lidar_frame.h:917-935, impl/lidar_frame_impl.h:733-989): GPU-backed
// ---------------------------------------------------------------------------------------
namespace impl {
/** elem_bytes = sizeof(T) * trailing dims.  Throws the reference's std::invalid_argument
 * messages for size mismatches. */
void destagger_bytes(const void* img, void* out, size_t h, size_t w, size_t elem_bytes,
                     const std::vector<int>& pixel_shift_by_row, bool inverse, size_t out_h,
                     size_t out_w);
}  // namespace impl

template <typename T>
inline void destagger_into(const ImgRef<const T>& img, const std::vector<int>& pixel_shift_by_row,
                           bool inverse, ImgRef<T> destaggered) {
    impl::destagger_bytes(img.data(), destaggered.data(), img.rows(), img.cols(), sizeof(T),
                          pixel_shift_by_row, inverse, destaggered.rows(), destaggered.cols());
}

/** n-d form: the dimensions behind (row, column) travel with the pixel (impl/lidar_frame_impl.h:776-811).
 *  @throw std::invalid_argument("image height does not match shifts size" / "image and destaggered must have the same shape") */
template <typename T, int ndim>
inline void destagger_into(const ConstArrayView<T, static_cast<size_t>(ndim)>& img, const std::vector<int>& pixel_shift_by_row,
                           bool inverse, ArrayView<T, static_cast<size_t>(ndim)> destaggered) {
    static_assert(ndim >= 2, "an image has rows and columns");
    if (pixel_shift_by_row.size() != img.shape[0]) throw std::invalid_argument{"image height does not match shifts size"};
    size_t extra = 1;
    for (int d = 0; d < ndim; ++d) {
        if (img.shape[d] != destaggered.shape[d]) throw std::invalid_argument{"image and destaggered must have the same shape"};
        if (d >= 2) extra *= img.shape[d];
    }
    if (img.sparse() || destaggered.sparse()) throw std::invalid_argument{"destagger needs dense row-major images"};
    if (extra == 0) return;
    impl::destagger_bytes(img.data(), destaggered.data(), img.shape[0], img.shape[1], sizeof(T) * extra,
                          pixel_shift_by_row, inverse, destaggered.shape[0], destaggered.shape[1]);
}

template <typename T>
inline img_t<T> destagger(const ImgRef<const T>& img, const std::vector<int>& pixel_shift_by_row,
                          bool inverse = false) {
    if (pixel_shift_by_row.size() != img.rows()) throw std::invalid_argument{"image height does not match shifts size"};
    img_t<T> out(img.rows(), img.cols(), impl::uninitialized);   // every element is written; pool memory, filled in place
    destagger_into<T>(img, pixel_shift_by_row, inverse, ImgRef<T>(out));
    return out;
}
template <typename T>
inline img_t<T> destagger(const img_t<T>& img, const std::vector<int>& pixel_shift_by_row,
                          bool inverse = false) {
    return destagger<T>(ImgRef<const T>(img), pixel_shift_by_row, inverse);
}

template <typename T>
inline img_t<T> destagger(const SensorInfo& info, const ImgRef<const T>& img, bool inverse = false) {
    if (img.rows() != info.format.pixels_per_column || img.cols() != info.format.columns_per_frame ||
        img.rows() != info.format.pixel_shift_by_row.size())
        throw std::invalid_argument{"Image resolution must match SensorInfo."};
    return destagger<T>(img, info.format.pixel_shift_by_row, inverse);
}
template <typename T>
inline img_t<T> destagger(const SensorInfo& info, const img_t<T>& img, bool inverse = false) {
    return destagger<T>(info, ImgRef<const T>(img), inverse);
}
template <typename T>
inline img_t<T> stagger(const SensorInfo& info, const img_t<T>& img) {
    return destagger<T>(info, img, true);
}
/** Destagger a whole Field (any element type, trailing dims allowed; field.cpp:329-340). */
Field destagger(const SensorInfo& info, const Field& field, bool inverse = false);

/** Timestamp of the staggered column a destaggered pixel came from (lidar_frame.cpp:893-905).
 *  @throw std::invalid_argument("row or column is out of range") */
uint64_t column_timestamp_at_destaggered_pixel(size_t row, size_t col,
                                               const std::vector<int>& pixel_shift_by_row,
                                               const HeaderRef<const uint64_t>& column_timestamps);
/** Convenience form over a frame and its sensor's shifts. */
uint64_t column_timestamp_at_destaggered_pixel(const LidarFrame& frame, const SensorInfo& info,
                                               size_t row, size_t col);

// ---------------------------------------------------------------------------------------
// FrameBatcher (lidar_frame.h:966-1145, lidar_frame.cpp:1248-1959; lidar packets only)
// ---------------------------------------------------------------------------------------
class FrameBatcher {
   public:
    PacketFormat pf;

    explicit FrameBatcher(const SensorInfo& info);
    explicit FrameBatcher(const std::shared_ptr<SensorInfo>& info);
    FrameBatcher(const FrameBatcher&) = delete;
    FrameBatcher& operator=(const FrameBatcher&) = delete;
    FrameBatcher(FrameBatcher&&) noexcept;
    ~FrameBatcher();

    /**
     * Add a packet to the frame; returns true when `lidar_frame` is ready.  packet_timestamp / alert_flags /
     * frame-level values are updated per packet exactly as the reference does; the frame's planes and column headers
     * are written by ONE GPU decode when the frame completes -- or earlier, when somebody looks: a frame that is still
     * being assembled decodes what has arrived the first time its planes or headers are accessed (impl::PendingDecode),
     * and then shows what the reference's packet-by-packet parse would (columns below the highest settled one decoded
     * or zeroed, the others unchanged).
     * @throw std::invalid_argument("unexpected frame dimensions") etc. as the reference
     * @throw std::runtime_error for a non-increasing FUSA frame id
     */
    bool batch(const Packet& packet, LidarFrame& lidar_frame);
    bool operator()(const Packet& packet, LidarFrame& lidar_frame) { return batch(packet, lidar_frame); }
    void reset();
    /** Extension: decode the packets collected so far into `lidar_frame` now (what the first access would do). */
    void flush(LidarFrame& lidar_frame);
    /**
     * Extension: hand every released frame's packets (arrival order, each lidar_packet_size bytes,
     * valid during the call) to `sink` instead of decoding them into `lidar_frame`.  The state
     * machine (frame boundaries, reorder cache, init-id changes, completeness) runs unchanged and
     * `lidar_frame` still receives the frame-level values and packet timestamps; its planes and
     * column headers are left alone.  Used by hip::FrameStream to batch whole frames for the GPU.
     */
    using PacketSink = std::function<void(const std::vector<const uint8_t*>& packets)>;
    void set_packet_sink(PacketSink sink);
    /**
     * Extension: the GPU this batcher decodes on (default: ouster::sdk::hip::current_device() of the
     * thread that triggers its first decode).  Every batcher owns its own HIP stream and scratch, so
     * distinct batchers may be driven from distinct threads, one per sensor, like the reference's.
     * @throw std::logic_error once it has started working on another GPU
     */
    void set_device(int device);
    size_t batched_packets() const;
    size_t dropped_packets() const;
    void set_max_cache_size(size_t n);
    size_t get_max_cache_size() const;

    struct State;  ///< implementation detail (host state machine + staging + device buffers)

   private:
    std::shared_ptr<State> s_;   ///< shared with the frames that still owe a decode of its staged packets
};

namespace impl {
/** LidarFrame -> wire packets (test-side "fake sensor", impl/lidar_frame_impl.h:435-531). */
std::vector<LidarPacket> frame_to_packets(const LidarFrame& frame,
                                          std::shared_ptr<PacketFormat> packet_format,
                                          uint32_t init_id, uint64_t prod_sn);
/** RAW_HEADERS present and tall enough for one column's headers and footers (lidar_frame.cpp:260-279). */
bool raw_headers_enabled(const PacketFormat& pf, const LidarFrame& frame);
/** The reference's form: the packets go to an STL output iterator over Packet (impl/lidar_frame_impl.h:433-437). */
template <typename OutputItT>
void frame_to_packets(const LidarFrame& frame, std::shared_ptr<PacketFormat> packet_format, OutputItT iter,
                      uint32_t init_id, uint64_t prod_sn) {
    for (auto& p : frame_to_packets(frame, std::move(packet_format), init_id, prod_sn)) *iter++ = std::move(p);
}

// ---- visiting fields by their run-time element type (impl/lidar_frame_impl.h:57-375) ----------------------------
// The operation receives a typed 2-D view of the field: OUSTER_FIELD_REF(T), by default this mirror's ImgRef<T>
// (the reference hands out Eigen::Ref<img_t<T>>; a build that has an Eigen with that spelling can define the macro to it
// before including this header).  FLOAT16 / ZONE_STATE / CHAR / VOID fields are skipped, as in the reference.
#ifndef OUSTER_FIELD_REF
#define OUSTER_FIELD_REF(T) ::ouster::sdk::core::ImgRef<T>
#define OUSTER_CONST_FIELD_REF(T) ::ouster::sdk::core::ImgRef<const T>
#endif
template <typename FIELD, typename OP, typename... Args>
void visit_field_2d(FIELD&& field, OP&& op, Args&&... args) {
    constexpr bool is_const = std::is_const<typename std::remove_reference<FIELD>::type>::value;
#define OUSTER_VISIT_CASE_(TAG, T)                                                                                  \
    case ChanFieldType::TAG: {                                                                                      \
        using E = typename std::conditional<is_const, const T, T>::type;                                            \
        using R = typename std::conditional<is_const, OUSTER_CONST_FIELD_REF(T), OUSTER_FIELD_REF(T)>::type;        \
        op(R(static_cast<ImgRef<E>>(field)), std::forward<Args>(args)...);                                          \
        break;                                                                                                      \
    }
    switch (field.tag()) {
        OUSTER_VISIT_CASE_(UINT8, uint8_t)
        OUSTER_VISIT_CASE_(UINT16, uint16_t)
        OUSTER_VISIT_CASE_(UINT32, uint32_t)
        OUSTER_VISIT_CASE_(UINT64, uint64_t)
        OUSTER_VISIT_CASE_(INT8, int8_t)
        OUSTER_VISIT_CASE_(INT16, int16_t)
        OUSTER_VISIT_CASE_(INT32, int32_t)
        OUSTER_VISIT_CASE_(INT64, int64_t)
        OUSTER_VISIT_CASE_(FLOAT32, float)
        OUSTER_VISIT_CASE_(FLOAT64, double)
        case ChanFieldType::FLOAT16:
        case ChanFieldType::ZONE_STATE:
        case ChanFieldType::CHAR:
        case ChanFieldType::VOID:
        case ChanFieldType::UNREGISTERED:
            break;
        default:
            throw std::invalid_argument("Invalid field for LidarFrame");
    }
#undef OUSTER_VISIT_CASE_
}
template <typename FRAME, typename OP, typename... Args>
void visit_field(FRAME&& frame, const std::string& name, OP&& op, Args&&... args) {
    if (!frame.has_field(name)) throw std::invalid_argument("Invalid field for LidarFrame");
    visit_field_2d(frame.field(name), std::forward<OP>(op), std::forward<Args>(args)...);
}
/** op(field view, field name, args...) for every channel field of the packet format the frame has. */
template <typename FRAME, typename OP, typename... Args>
void foreach_channel_field(FRAME&& frame, const PacketFormat& pf, OP&& op, Args&&... args) {
    for (const auto& ft : pf)
        if (frame.has_field(ft.first)) visit_field(frame, ft.first, std::forward<OP>(op), ft.first, std::forward<Args>(args)...);
}
}  // namespace impl

}  // namespace core
}  // namespace sdk
}  // namespace ouster
