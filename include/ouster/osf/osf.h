// osf.h -- reading LidarFrame field planes out of OSF files.
//
// Mirrors, for the hot path only, what the reference's ouster_osf does between an .osf file and a
// LidarFrame: the container walk (ouster_osf/src/file.cpp, reader.cpp, fb_utils.cpp; flatbuffer
// schemas ouster_osf/fb/*.fbs) and restore_lidar_frame (ouster_osf/src/stream_lidar_frame.cpp:165-340)
// with decode_field (ouster_osf/src/png_tools.cpp:664-745).  The split is by what each processor is
// good at: the HOST walks the flatbuffers and undoes the entropy coding (zlib inflate + PNG scanline
// filters, zstd for ZPNG) into pinned staging; the GPU (ouster_hip_osf_unpack) turns the pixel bytes
// of every field of every frame of a batch into typed planes -- ZPNG left-delta prefix sums, PNG
// sample byte order, stagger() -- in the [frames][H][W] layout the rest of the path (destagger,
// cartesian, dewarp) consumes unchanged.
// Out of scope: writing OSF, non-lidar streams, object lists, the streaming-info index.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ouster/core/lidar_frame.h"
#include "ouster/core/xyzlut.h"
#include "ouster/hip/context.h"

namespace ouster {
namespace sdk {
namespace osf {

/** A whole .osf file in memory with its blocks located and checked. */
class OsfFile {
   public:
    /** @throw std::runtime_error when the file cannot be read, is not an OSF file, was not finished
     *  (header status != VALID) or its header / metadata block fails its CRC32 */
    explicit OsfFile(const std::string& path);

    struct MetadataEntry {
        uint32_t id = 0;
        std::string type;        ///< e.g. "ouster/v1/os_sensor/LidarSensor"
        const uint8_t* buffer = nullptr;  ///< size-prefixed flatbuffer of that type
        size_t size = 0;
    };
    struct Message {
        uint64_t ts = 0;        ///< StampedMessage.ts (host nanoseconds)
        uint32_t id = 0;        ///< metadata entry id of the stream it belongs to
        const uint8_t* buffer = nullptr;  ///< size-prefixed flatbuffer of the stream's message type
        size_t size = 0;
    };

    uint64_t version() const { return version_; }
    const std::string& id() const { return id_; }
    const std::vector<MetadataEntry>& metadata_entries() const { return entries_; }
    /** sensor metadata id -> the sensor's metadata JSON (LidarSensor entries) */
    std::map<uint32_t, std::string> sensor_metadata_json() const;
    /** LidarScanStream id -> sensor metadata id */
    std::map<uint32_t, uint32_t> lidar_scan_streams() const;
    /** Every message of every chunk, sorted by ts.  @throw std::runtime_error on a chunk CRC mismatch */
    std::vector<Message> messages() const;

   private:
    std::vector<uint8_t> buf_;
    uint64_t version_ = 0, metadata_offset_ = 0, chunks_base_ = 0;
    std::string id_;
    std::vector<MetadataEntry> entries_;
    std::vector<uint64_t> chunk_offsets_;
};

/** One encoded field of a LidarScanMsg. */
struct EncodedField {
    std::string name;
    core::ChanFieldType type = core::ChanFieldType::VOID;
    const uint8_t* data = nullptr;
    size_t size = 0;
};

/** The parts of a LidarScanMsg the hot path uses (views into the file's memory). */
struct LidarScanMsgView {
    int32_t frame_id = 0;
    uint64_t frame_status = 0;
    uint8_t shutdown_countdown = 0, shot_limiting_countdown = 0;
    std::vector<EncodedField> fields;
    /** LidarScanMsg.custom_fields (every field whose name is not in the CHAN_FIELD enum: WINDOW, ZONE_MASK, user-added
     *  pixel / column / packet / frame fields; stream_lidar_frame.cpp:100-127 writes them, fb_restore_fields :322 reads
     *  them back): name, element type, full shape, class and the encoded bytes (1-D: raw; otherwise PNG / ZPNG of the
     *  field collapsed to rows x (size / rows), never staggered). */
    struct CustomField {
        std::string name;
        core::ChanFieldType type = core::ChanFieldType::VOID;
        std::vector<size_t> shape;
        core::FieldClass field_class = core::FieldClass::NONE;
        const uint8_t* data = nullptr;
        size_t size = 0;
    };
    std::vector<CustomField> custom_fields;
    const uint64_t* timestamp = nullptr;      size_t n_timestamp = 0;
    const uint16_t* measurement_id = nullptr; size_t n_measurement_id = 0;
    const uint32_t* status = nullptr;         size_t n_status = 0;
    const uint64_t* packet_timestamp = nullptr; size_t n_packet_timestamp = 0;
    const uint8_t* alert_flags = nullptr;     size_t n_alert_flags = 0;
    const double* pose = nullptr;             size_t n_pose = 0;
    /** @throw std::runtime_error on a malformed buffer */
    static LidarScanMsgView parse(const OsfFile::Message& msg);
};

/** What the host leaves for the GPU of one encoded field: pixel bytes + how to read them. */
struct StagedField {
    uint32_t encoding = 0;         ///< OUSTER_HIP_OSF_*
    uint32_t src_pixel_bytes = 0;
    bool filtered = false;         ///< PNG: `bytes` is the inflated stream, filter-type byte per scanline (device_unfilter)
    std::vector<uint8_t> bytes;    ///< PNG: unfiltered scanlines (or the inflated stream); ZPNG: zstd-decompressed residuals
};
/** Host half of decode_field: inflate (PNG; + the scanline filters unless device_unfilter) or zstd (ZPNG).  h, w: expected
 *  image size.  device_unfilter (round 5): leave the PNG scanline filters to the GPU (ouster_hip_osf_plane::flags); the
 *  filter-type bytes are checked here.
 *  @throw std::runtime_error("decodeField: could not decode field") like the reference */
StagedField stage_field(const EncodedField& f, size_t h, size_t w, bool device_unfilter = false);

/** A batch of OSF frames whose planes stay in HBM: plane `name` is [n_frames][H][W] elements, the layout
 *  ouster_hip_decode produces, so ouster_hip_destagger / ouster_hip_cartesian / ouster_hip_dewarp_frames (or
 *  their C++ faces below) run on it unchanged and nothing crosses PCIe but the inflated pixel bytes. */
class OsfDeviceBatch {
   public:
    uint32_t n_frames() const { return n_; }
    size_t h() const { return h_; }
    size_t w() const { return w_; }
    const std::vector<std::pair<std::string, core::ChanFieldType>>& fields() const { return fields_; }
    /** device pointer of plane `name`, [n_frames][h][w].  @throw std::out_of_range */
    void* plane_device(const std::string& name) const;
    /** host copies of the per-frame headers (they come straight from the flatbuffer) */
    const std::vector<int32_t>& frame_ids() const { return frame_ids_; }
    const std::vector<uint64_t>& timestamps() const { return ts_; }            ///< [n_frames][w]
    const std::vector<uint32_t>& status() const { return status_; }            ///< [n_frames][w]
    /** destagger<T>() of plane `name` for every frame, on the device; returns the device pointer of
     *  the result (owned by the batch, valid until the next call for the same name). */
    void* destagger_device(const std::string& name);
    /** XYZLutT<float|double>::operator() of plane `range_field` for every frame, on the device:
     *  [n_frames][h*w][3]; owned by the batch. */
    void* cartesian_device(const core::XYZLut& lut, bool f64 = false, const std::string& range_field = core::ChanField::RANGE);
    void download(const void* device_ptr, void* host, size_t bytes) const;
    const std::shared_ptr<hip::Context>& context() const { return ctx_; }

   private:
    friend class OsfFrameDecoder;
    std::shared_ptr<hip::Context> ctx_;
    core::SensorInfo info_;
    uint32_t n_ = 0;
    size_t h_ = 0, w_ = 0;
    std::vector<std::pair<std::string, core::ChanFieldType>> fields_;
    std::shared_ptr<void> planes_;               // one HBM allocation, planes at plane_off_
    std::map<std::string, size_t> plane_off_;
    std::map<std::string, std::shared_ptr<void>> derived_;
    std::vector<int32_t> frame_ids_;
    std::vector<uint64_t> ts_;
    std::vector<uint32_t> status_;
};

/** Decodes LidarScan messages of one sensor into LidarFrames, the pixel work on the GPU. */
class OsfFrameDecoder {
   public:
    explicit OsfFrameDecoder(const core::SensorInfo& info, int device = -1);
    ~OsfFrameDecoder();
    /** restore_lidar_frame for a batch: one ouster_hip_osf_unpack launch covers every field of every
     *  message.  Frames get exactly the fields their message carries (restore_lidar_frame :176-199).
     *  @throw std::runtime_error / std::invalid_argument like the reference's reader */
    std::vector<core::LidarFrame> decode(const std::vector<OsfFile::Message>& msgs);
    core::LidarFrame decode(const OsfFile::Message& msg) { return std::move(decode(std::vector<OsfFile::Message>{msg})[0]); }
    /** decode_field() of png_tools.cpp:664-745 for standalone encoded fields (each an H x W image of its
     *  type), all in one launch: returns the staggered planes as raw little-endian bytes, one per field.
     *  @throw std::runtime_error("decodeField: could not decode field") */
    std::vector<std::vector<uint8_t>> decode_fields(const std::vector<EncodedField>& fields);
    /** The same decode, results left in HBM (all messages must carry the same fields).
     *  @throw std::invalid_argument when the messages' field lists differ */
    OsfDeviceBatch decode_device(const std::vector<OsfFile::Message>& msgs);
    /** Where the PNG scanline filters are reversed: on the GPU (default since round 5: the host is left with inflate only) or
     *  on the host as in rounds 2 - 4.  Results are byte-identical. */
    void set_device_unfilter(bool on);
    bool device_unfilter() const;

   private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace osf
}  // namespace sdk
}  // namespace ouster
