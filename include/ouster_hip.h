/*
 * ouster_hip.h -- C ABI of the MI355X (gfx950) implementation of the Ouster
 * SDK's per-pixel hot path:
 *     packet-format field decode -> LidarFrame planes, destagger, cartesian.
 *
 * This is the drop-in boundary.  The reference (ouster-sdk 1.0.1) has no C ABI
 * for this path -- its boundary is the C++ API of ouster_core -- so each entry
 * point below names the reference C++ interface it replaces (paths relative to
 * the reference checkout).  The C++ classes in include/ouster/core/ (same names
 * and semantics as the reference: PacketFormat, LidarFrame, FrameBatcher,
 * destagger<T>, XYZLutT<T>, cartesian) are thin host code over these calls; see
 * INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns OUSTER_HIP_OK (0) or a negative error code and
 *     never throws; ouster_hip_last_error() returns a thread-local message.
 *   - all "device" pointers are HIP device pointers on the context's GPU; all
 *     work is ordered on the context's stream (ouster_hip_ctx_stream) and is
 *     asynchronous unless stated otherwise; call ouster_hip_sync() to wait.
 *   - plain pointers and sizes only; no C++/torch types.
 *   - a context is not thread-safe (like the reference's FrameBatcher, one per stream of work,
 *     externally serialised); distinct contexts may be used from distinct threads.  Formats and
 *     LUTs are immutable after creation and may be shared by the calls of their context.
 */
#ifndef OUSTER_HIP_H
#define OUSTER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OUSTER_HIP_OK 0
#define OUSTER_HIP_ERR_INVALID_ARGUMENT (-1) /* maps to std::invalid_argument */
#define OUSTER_HIP_ERR_RUNTIME (-2)          /* HIP runtime failure -> std::runtime_error */
#define OUSTER_HIP_ERR_NO_DEVICE (-3)
#define OUSTER_HIP_ERR_UNSUPPORTED (-4)

#define OUSTER_HIP_MAX_FIELDS 32

/* element type tags == ouster::sdk::core::ChanFieldType
 * (ouster_core/include/ouster/core/chanfield.h:111-128) */
#define OUSTER_HIP_U8 1
#define OUSTER_HIP_U16 2
#define OUSTER_HIP_U32 3
#define OUSTER_HIP_U64 4
#define OUSTER_HIP_F32 9
#define OUSTER_HIP_F64 10
#define OUSTER_HIP_F16 12

typedef struct ouster_hip_ctx ouster_hip_ctx;
typedef struct ouster_hip_format ouster_hip_format;
typedef struct ouster_hip_lut ouster_hip_lut;

/* One bit-field: POD mirror of ouster::sdk::core::FieldDecodeInfo
 * (ouster_core/include/ouster/core/field_decode_info.h:24-54):
 *   value = ((*(u64*)(base + offset)) & mask) >> shift   (shift<0: << -shift) */
typedef struct ouster_hip_bits {
    uint64_t mask;
    uint32_t offset;
    int32_t shift;
} ouster_hip_bits;

/* One decoded channel field and the LidarFrame plane it lands in.
 * dst_elem_size = sizeof(plane element) * extra dims (1,2,4,8; 6 for the
 * packed 3 x float16 RGB plane, impl/lidar_frame_impl.h:36-40).  The decoded
 * 64-bit word is truncated little-endian to dst_elem_size bytes exactly as
 * FieldDecodeInfo::get<T> does. */
typedef struct ouster_hip_field_desc {
    ouster_hip_bits bits;    /* offset is relative to the pixel's channel data */
    uint32_t dst_elem_size;
    uint32_t f16_nan_fill;   /* !=0: "zero" for this plane is 0x7e00 per 16-bit lane
                                (ouster_core/src/lidar_frame.cpp:1397-1401) */
} ouster_hip_field_desc;

/* Packet geometry + bit layout: POD mirror of ouster::sdk::core::PacketFormat
 * (ouster_core/include/ouster/core/types.h:137-160, src/parsing.cpp:453-538).
 * Covers the built-in profiles and anything add_custom_profile() registers
 * (ouster_core/include/ouster/core/profile_extension.h:52-55). */
typedef struct ouster_hip_format_desc {
    uint32_t pixels_per_column;   /* H */
    uint32_t columns_per_packet;
    uint32_t columns_per_frame;   /* W */
    uint32_t packet_header_size;
    uint32_t col_header_size;
    uint32_t channel_data_size;
    uint32_t col_footer_size;
    uint32_t packet_footer_size;
    uint32_t col_size;
    uint32_t lidar_packet_size;
    /* column header fields, offsets relative to the column start */
    ouster_hip_bits col_timestamp;
    ouster_hip_bits col_measurement_id;
    ouster_hip_bits col_status;
    /* packet header fields, offsets relative to the packet start */
    ouster_hip_bits frame_id;
    ouster_hip_bits alert_flags;
    ouster_hip_bits thermal_shutdown;
    ouster_hip_bits shot_limiting;
    ouster_hip_bits countdown_thermal_shutdown;
    ouster_hip_bits countdown_shot_limiting;
    uint32_t n_fields;
    uint32_t reserved;
    ouster_hip_field_desc fields[OUSTER_HIP_MAX_FIELDS];
} ouster_hip_format_desc;

/* Sensor calibration for the XYZ lookup table: the arguments of
 * impl::make_xyz_lut (ouster_core/include/ouster/core/xyzlut.h:53-56).
 * Matrices are 4x4 row-major.  n_angles is either H (OS sensors: per-beam
 * angles) or W*H (DF sensors: per-pixel angles, xyzlut.cpp:49-59). */
typedef struct ouster_hip_calib {
    uint32_t w, h;
    double range_unit;
    double beam_to_lidar_transform[16];
    double transform[16];
    const double* azimuth_angles_deg;
    const double* altitude_angles_deg;
    size_t n_angles;
} ouster_hip_calib;

/* frame-level values latched from the first packet of a frame
 * (FrameBatcher::start_frame, ouster_core/src/lidar_frame.cpp:1709-1741) */
typedef struct ouster_hip_frame_meta {
    int64_t frame_id;          /* -1 if the frame has no packets */
    uint64_t frame_status;     /* thermal&0xf | (shot&0xf)<<4, lidar_frame.cpp:1310-1323 */
    uint16_t shutdown_countdown;
    uint16_t shot_limiting_countdown;
    uint32_t n_valid_columns;  /* columns of the frame that received a valid column (not in the reference struct) */
} ouster_hip_frame_meta;

/* Device output pointers of one batched decode.  Every array is dense over
 * the batch: plane i is [n_frames][H][W] elements of fields[i].dst_elem_size
 * bytes, row-major like Field (ouster_core/include/ouster/core/field.h:828-905).
 * Any pointer may be NULL to skip that output. */
typedef struct ouster_hip_frame_out {
    void* planes[OUSTER_HIP_MAX_FIELDS];       /* staggered planes               */
    void* destaggered[OUSTER_HIP_MAX_FIELDS];  /* destagger(plane) fused in       */
    uint64_t* timestamp;        /* [n_frames][W]      LidarFrame::timestamp()      */
    uint16_t* measurement_id;   /* [n_frames][W]      LidarFrame::measurement_id() */
    uint32_t* status;           /* [n_frames][W]      LidarFrame::status()         */
    uint64_t* packet_timestamp; /* [n_frames][W/cpp]  needs host_timestamps        */
    uint8_t* alert_flags;       /* [n_frames][W/cpp]  only touched where a packet exists */
    ouster_hip_frame_meta* frame_meta; /* [n_frames] */
    /* cartesian fused in: xyz[k] = lut(plane of field xyz_field[k]); [n_frames][H*W][3] */
    void* xyz[2];
    int32_t xyz_field[2];       /* index into fields[]; -1 = unused */
    int32_t xyz_dtype;          /* OUSTER_HIP_F32 or OUSTER_HIP_F64 */
    int32_t reserved;
    /* Optional by-product for a range-gated dewarp that follows (ouster_hip_dewarp_frames_counted): how
     * many pixels of every column have gate_min_r <= value <= gate_max_r in the plane of field
     * gate_field (raw range units, like the u32 min_r / max_r of impl/dewarp_impl.h:33-34).  The decode
     * kernel has those values in registers, so the dewarp's counting pass over the RANGE plane
     * disappears.  Layout: u16 [n_frames][OUSTER_HIP_GATE_CHUNKS][W] partial counts (a column's count is
     * the sum over the chunk axis; unused chunks are written as zero).  NULL: not produced. */
    uint16_t* gate_counts;
    uint32_t gate_min_r, gate_max_r;
    int32_t gate_field;         /* index into fields[] of a 32-bit range field */
    int32_t reserved2;
    /* dewarp<T>(points, poses) (ouster_core/include/ouster/core/pose_util.h:38-56) fused behind the
     * cartesian: device array [n_frames][W][16] of row-major 4x4 per-column poses (the frame's
     * body_to_world).  When set, xyz[k] receives R_col * lut(r) + t_col computed in the element type of xyz
     * (the poses are cast to it, as the reference does) while the point is still in registers -- the
     * separate pass over the cloud (read 12 B + write 12 B per point) disappears.  Needs the separable
     * LUT tables (ouster_hip_lut_create).  NULL: xyz stays in the sensor / body frame of the LUT.
     * 16-byte aligned arrays take the wide-tile kernels; an array that is only 8-byte aligned is decoded by the
     * 64-column kernel (same result, slower). */
    const double* xyz_poses;
} ouster_hip_frame_out;
#define OUSTER_HIP_GATE_CHUNKS 8

/* ---- context ------------------------------------------------------------ */
/* stream: an existing hipStream_t to order on (e.g. torch's current stream),
 * OUSTER_HIP_STREAM_NULL to order on the device's null (legacy default) stream,
 * or NULL to let the context create and own a non-blocking stream. */
#define OUSTER_HIP_STREAM_NULL ((void*)(intptr_t)-1)
int ouster_hip_ctx_create(int device, void* stream, ouster_hip_ctx** out);
void ouster_hip_ctx_destroy(ouster_hip_ctx* ctx);
void* ouster_hip_ctx_stream(ouster_hip_ctx* ctx);
int ouster_hip_ctx_device(ouster_hip_ctx* ctx); /* the HIP device ordinal the context was created on */
/* Experiment / test knobs (not part of the behavioural contract; results are identical for every
 * setting).  Their defaults come from OUSTER_HIP_<NAME> environment variables read once in
 * ouster_hip_ctx_create -- the call path itself never touches the environment.
 *   "wide"  -1 auto | 0 k_decode only | 64/128/256/512 force that k_decode_wide tile width
 *   "tile"  force k_decode's tile width (64/32/16)      "wide_kb"  LDS budget of a wide tile
 *   "wide_min_blocks"  smallest launch that may use wide tiles   "tune"  0: no variant timing
 *   "xcd"   0: plain block -> frame mapping   "fast"  0: every frame through the general mapping
 *   "retune" 1: forget the variant tuner's verdicts (a caller that re-allocated its buffers re-learns)
 *   "stream" -1 auto | 0 never | 128/256 force k_decode_stream's tile width   "stream_rows" rows of its tiles
 *   "stream_wait" 0: trust the in-order vmcnt instead of draining it before a prefetched tile is used
 *   "stream_min_tiles" tiles per persistent workgroup below which a launch stays on the one-tile kernels */
int ouster_hip_ctx_set_knob(ouster_hip_ctx* ctx, const char* name, int value);
int ouster_hip_sync(ouster_hip_ctx* ctx);
const char* ouster_hip_last_error(void);
const char* ouster_hip_version(void);

/* ---- packet format -> device descriptor ---------------------------------- */
/* Replaces constructing ouster::sdk::core::PacketFormat for device use
 * (ouster_core/src/parsing.cpp:600-626). */
int ouster_hip_format_create(ouster_hip_ctx* ctx, const ouster_hip_format_desc* desc,
                             ouster_hip_format** out);
void ouster_hip_format_destroy(ouster_hip_format* fmt);

/* ---- XYZ lookup table ---------------------------------------------------- */
/* Replaces impl::make_xyz_lut(w,h,range_unit,b2l,transform,az,alt)
 * (ouster_core/src/xyzlut.cpp:11-89).  Per-beam (n_angles == h) calibrations
 * are kept as separable per-beam x per-column tables (the LUT is never
 * materialised on the device); per-pixel calibrations fall back to a full
 * double LUT.  Returns INVALID_ARGUMENT for the same dimension errors the
 * reference throws for (xyzlut.cpp:14-21). */
int ouster_hip_lut_create(ouster_hip_ctx* ctx, const ouster_hip_calib* calib,
                          ouster_hip_lut** out);
/* Replaces XYZLutT<T>(direction, offset, h, w) (xyzlut.h:134-135): adopt
 * caller-provided host arrays [n][3] of dtype F32 or F64 (copied to the device). */
int ouster_hip_lut_create_from_arrays(ouster_hip_ctx* ctx, const void* direction,
                                      const void* offset, uint32_t h, uint32_t w,
                                      int dtype, ouster_hip_lut** out);
/* Host copy of the full double LUT, [w*h][3] each == XYZLut::direction/offset
 * (xyzlut.h:82-89). */
int ouster_hip_lut_export(const ouster_hip_lut* lut, double* direction, double* offset);
void ouster_hip_lut_destroy(ouster_hip_lut* lut);

/* ---- batched decode (+ fused destagger + fused cartesian) ---------------- */
/* Replaces a sequence of FrameBatcher::batch(packet, frame) calls
 * (ouster_core/src/lidar_frame.cpp:1824-1884; pixel work :1422-1576, which
 * calls PacketFormat::block_field / col_field, src/parsing.cpp:628-675),
 * optionally followed by destagger<T>() (impl/lidar_frame_impl.h:733-760) and
 * XYZLutT::operator() (xyzlut.h:139-150) on the result.
 *
 * packets: device buffer laid out [n_frames][slots_per_frame][packet_stride]
 *   bytes; packet_stride >= lidar_packet_size.  packet_counts (nullable: all
 *   slots filled) gives the number of leading slots filled per frame; it may be a
 *   DEVICE array (used in place, values clamped to slots_per_frame; the form to use
 *   under stream capture), a pinned host array (copied on the stream; must stay
 *   unchanged until the call has executed) or a plain host array (copied through the
 *   context's pinned staging before the call returns).  No form synchronises the stream.
 *   Packets of a frame may be in any order; a frame's result is "all planes and
 *   column headers zero, then every received column with status&1 and
 *   measurement_id < W written at its measurement_id" -- the memset-then-scatter
 *   equivalent of the reference's incremental zero-fill.
 *   When two received columns carry the same measurement_id the one later in
 *   the buffer wins (the reference: the one batched later).
 *   Fast path: when slots_per_frame * columns_per_packet == W and slot p holds the
 *   packet with columns [p*cpp, (p+1)*cpp) (what a sensor sends, in order; missing
 *   packets may be left as holes with status 0), no mapping work is done at all; any
 *   other arrangement is detected on the device and redone by a second pass.
 *   The call keeps no state between invocations and can be captured in a HIP graph
 *   and replayed on changed packet contents (do one eager call first: scratch
 *   allocations cannot happen during capture).
 * host_timestamps: device array [n_frames][slots_per_frame] of
 *   Packet::host_timestamp values, or NULL (packet_timestamp is then not written).
 * pixel_shift_by_row: HOST array [H] (needed iff any destaggered[] is set).
 * luts: HOST array of n_luts LUT handles; frame f uses luts[f % n_luts]
 *   (multi-sensor batches interleave sensors); NULL iff no xyz output. */
int ouster_hip_decode(ouster_hip_ctx* ctx, const ouster_hip_format* fmt,
                      const uint8_t* packets, size_t packet_stride,
                      uint32_t slots_per_frame, const uint32_t* packet_counts,
                      uint32_t n_frames, const uint64_t* host_timestamps,
                      const ouster_hip_frame_out* out,
                      const int32_t* pixel_shift_by_row,
                      const ouster_hip_lut* const* luts, uint32_t n_luts);

/* ---- standalone destagger ------------------------------------------------ */
/* Replaces destagger_into<T>(img, pixel_shift_by_row, inverse, out)
 * (impl/lidar_frame_impl.h:733-760, N-d variant :776-811): n_images images of
 * h x w elements of elem_bytes bytes (element = T times trailing dims).
 * shifts: HOST array; n_shifts != h -> INVALID_ARGUMENT ("image height does
 * not match shifts size"). */
int ouster_hip_destagger(ouster_hip_ctx* ctx, const void* src, void* dst, uint32_t h,
                         uint32_t w, uint32_t elem_bytes, const int32_t* shifts,
                         uint32_t n_shifts, int inverse, uint32_t n_images);

/* ---- standalone cartesian ------------------------------------------------ */
/* Replaces impl::cartesianT<T>(points, range, direction, offset)
 * (impl/cartesian.h:36-66) / XYZLutT<T>::operator()(range) / cartesian():
 * range [n_images][h*w] u32 (staggered) -> xyz [n_images][h*w][3] of
 * xyz_dtype (F32 or F64). */
int ouster_hip_cartesian(ouster_hip_ctx* ctx, const ouster_hip_lut* lut,
                         const uint32_t* range, void* xyz, int xyz_dtype,
                         uint32_t n_images);

/* ---- dense dewarp ---------------------------------------------------------- */
/* Replaces dewarp<T>(dewarped, points, poses) (ouster_core/include/ouster/core/pose_util.h:38-56):
 * points [n_images][h*w][3] (row-major pixel order, i = row*w + col) are moved by the pose of
 * their column: p' = R_col * p + t_col.  poses: device array [n_images][w][16] doubles, each a
 * row-major 4x4 (LidarFrame::body_to_world layout, lidar_frame.cpp:350-358).  dtype F32/F64 is
 * the element type of points/dewarped; the arithmetic is done in that type like the
 * reference.  points and dewarped may alias. */
int ouster_hip_dewarp(ouster_hip_ctx* ctx, const void* points, const double* poses, void* dewarped,
                      int dtype, uint32_t h, uint32_t w, uint32_t n_images);

/* ---- range-gated, compacting frame dewarp ---------------------------------- */
/* Replaces dewarp<T>(const LidarFrame&, const XYZLutT<T>&, min_range, max_range) and its FrameSet
 * form with provenance (ouster_core/include/ouster/core/pose_util.h:456-493,
 * impl/dewarp_impl.h:23-115) for a batch of frames resident in HBM.
 *   range     [n_frames][h][w] u32   staggered RANGE planes (h, w taken from the LUTs)
 *   status    [n_frames][w]    u32   column status; timestamp [n_frames][w] u64 (nullable unless
 *                                    timestamps_ns is requested); poses [n_frames][w][16] f64
 *   luts      host array, frame f uses luts[f % n_luts] (one XYZLut per sensor of a FrameSet)
 * Per frame: columns first_valid..last_valid (status & 1, lidar_frame.cpp:907-925), skipping
 * status == 0; rows top to bottom; a point is kept when ceil(min_range*1e3) <= r <=
 * floor(max_range*1e3); xyz = lut(r) (f64 tables, or r*dir+ofs in a user LUT's precision), then
 * p' = R_col*p + t_col evaluated in `dtype` with the pose cast to it.  Frames are concatenated in
 * index order.
 * Outputs (device): points [capacity][3] of dtype; optional frame_idxs / col_idxs [capacity] u32,
 * timestamps_ns [capacity] u64; frame_offsets [n_frames + 1] u64 = exclusive prefix of the kept
 * points per frame (frame_offsets[n_frames] = total).  Points beyond `capacity` are dropped, the
 * offsets still report the full counts: h*w*n_frames is always enough
 * (impl::max_number_of_valid_points, pose_util.cpp:15-29). */
int ouster_hip_dewarp_frames(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                             const uint32_t* range, const uint32_t* status,
                             const uint64_t* timestamp, const double* poses, uint32_t n_frames,
                             double min_range, double max_range, int dtype, void* points,
                             uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                             uint64_t capacity, uint64_t* frame_offsets);

/* The same with the per-column kept counts already known: `gate_counts` as written by an
 * ouster_hip_decode of the same frames with the same gate (ceil(min_range*1e3) / floor(max_range*1e3) on
 * RANGE); the counting kernel is skipped.  gate_counts == NULL is ouster_hip_dewarp_frames. */
int ouster_hip_dewarp_frames_counted(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                                     const uint32_t* range, const uint32_t* status,
                                     const uint64_t* timestamp, const double* poses, uint32_t n_frames,
                                     double min_range, double max_range, int dtype, void* points,
                                     uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                                     uint64_t capacity, uint64_t* frame_offsets, const uint16_t* gate_counts);
/* The same with the poses given the way dewarp<float> uses them: pose_rows [n_frames][w][12] float = rows 0..2 of every
 * column's body_to_world (row-major 3 x 4), already cast to float -- dewarp<T> casts the pose to T before it multiplies
 * (pose_util.h:38-56), so for float output this table IS what the arithmetic sees, at 48 B per column instead of the 128 B
 * of a double 4 x 4 (67 MB -> 25 MB per 256 frames of 2048 columns; a caller that stages poses on the host converts them
 * there).  float output only; gate_counts as above (may be NULL). */
int ouster_hip_dewarp_frames_rows(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                                  const uint32_t* range, const uint32_t* status, const uint64_t* timestamp,
                                  const float* pose_rows, uint32_t n_frames, double min_range, double max_range,
                                  void* points, uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                                  uint64_t capacity, uint64_t* frame_offsets, const uint16_t* gate_counts);
/* The raw gate ouster_hip_dewarp_frames derives from metres: min_r = ceil(min_range*1e3), max_r =
 * floor(max_range*1e3), clamped to u32; returns 0 and sets *empty when nothing can pass. */
int ouster_hip_range_gate(double min_range, double max_range, uint32_t* min_r, uint32_t* max_r, int* empty);

/* ---- OSF field planes -------------------------------------------------------- */
/* The device half of decode_field (ouster_osf/src/png_tools.cpp:664-745) for a batch of encoded
 * LidarFrame fields whose entropy coding the host has already undone (PNG: zlib inflate + scanline
 * filters; ZPNG: zstd): pixel bytes -> typed plane elements, and for PNG planes -- which the OSF
 * writer stored destaggered (png_lidarframe_encoder.cpp:112-125) -- the stagger() back, so that the
 * planes land in the same [H][W] staggered layout ouster_hip_decode produces and
 * ouster_hip_destagger / ouster_hip_cartesian / ouster_hip_dewarp_frames run on them unchanged.
 *   PNG_*   src = h rows of w pixels, unfiltered, no filter bytes.  Value = little-endian composition
 *           of the pixel's bytes; 16-bit samples are stored byte-swapped (png_set_swap in the writer).
 *   ZPNG    src = the zstd-decompressed residuals (thirdparty/zpng/zpng.cpp:101-352): per row and
 *           byte lane a running sum mod 256 of the left deltas; 3- and 4-byte pixels come as colour
 *           planes with the GB-RG transform.  Undone on the device (row-wise prefix sums).  Never
 *           staggered: ZPNG planes are stored as they are.
 * The decoded value is truncated / zero-extended to dst_elem_size like the reference's
 * static_cast<T>.  pixel_shift_by_row: HOST array [h] (needed iff any job is a PNG plane). */
#define OUSTER_HIP_OSF_PNG_GRAY8 1
#define OUSTER_HIP_OSF_PNG_GRAY16 2
#define OUSTER_HIP_OSF_PNG_RGB8 3    /* 24-bit values */
#define OUSTER_HIP_OSF_PNG_RGBA8 4   /* 32-bit values */
#define OUSTER_HIP_OSF_PNG_RGBA16 5  /* 64-bit values */
#define OUSTER_HIP_OSF_ZPNG 6        /* src_pixel_bytes = channels * bytes per channel (1..8) */
typedef struct ouster_hip_osf_plane {
    const void* src;          /* device */
    void* dst;                /* device, [h][w] elements of dst_elem_size bytes */
    uint32_t encoding;        /* OUSTER_HIP_OSF_* */
    uint32_t src_pixel_bytes; /* bytes of one source pixel */
    uint32_t dst_elem_size;   /* 1, 2, 4 or 8 */
    uint32_t flags;           /* OUSTER_HIP_OSF_FLAG_* (0: src holds the pixel bytes) */
} ouster_hip_osf_plane;
/* PNG encodings only: `src` is the INFLATED IDAT stream as libpng sees it -- h scanlines of 1 filter-type byte followed by
 * w * src_pixel_bytes filtered bytes (PNG specification section 9: None / Sub / Up / Average / Paeth) -- and the library
 * reverses the scanline filters on the device (k_osf_png_unfilter, round 5) before it unpacks the pixels.  The caller checks
 * the h filter-type bytes (<= 4: the reference's libpng refuses anything else); the kernel treats an unknown type as None. */
#define OUSTER_HIP_OSF_FLAG_FILTERED 1u
int ouster_hip_osf_unpack(ouster_hip_ctx* ctx, const ouster_hip_osf_plane* planes, uint32_t n_planes,
                          uint32_t h, uint32_t w, const int32_t* pixel_shift_by_row);

/* ---- host containers: memory the GPU reaches in place, frame-at-a-time calls -------------------- */
/* The reference's callers work one frame at a time on HOST containers: destagger<T>(img, shifts)
 * (ouster_core/include/ouster/core/lidar_frame.h:917-919), XYZLutT<T>::operator()(range)
 * (xyzlut.h:139-150), dewarp<T>(points, poses) (pose_util.h:38-56), FrameBatcher::batch(packet, frame)
 * (lidar_frame.h:1043-1144).  For those the cost is the PCIe crossing, not the kernel, so this
 * group removes everything else from it:
 *   - ouster_hip_host_alloc hands out page-locked host memory from a process-wide pool (freed
 *     blocks are kept and handed out again: no hipHostMalloc in steady state).  The GPU reads
 *     and writes such memory IN PLACE: every "device pointer" of this header may be a pointer into a
 *     pool block, and a kernel launched on it moves its input and its output over the full-duplex
 *     link at the same time, in one launch, with no staging copy on either side.  The C++ mirror's
 *     containers (Field, img_t<T>, PointCloudXYZ<T>) allocate from it.
 *   - the *_host entry points take HOST pointers of any provenance and return when the result is
 *     in place: pool memory is used where it lies; other memory (a caller's own Eigen array, a
 *     numpy buffer) goes through grow-only device scratch of the context with asynchronous copies
 *     and ONE stream synchronisation.  Neither route allocates once the context has seen the size.
 * Without a GPU ouster_hip_host_alloc returns plain calloc'd memory (containers still work; every
 * compute entry point fails with OUSTER_HIP_ERR_NO_DEVICE as before). */
#define OUSTER_HIP_HOST_POOL_MIN 2048 /* smaller requests are plain malloc / calloc memory (free() or host_free) */
void* ouster_hip_host_alloc(size_t bytes, int zero);
void ouster_hip_host_free(void* p);
/* 1 when [p, p + bytes) lies inside one live block of the pool (device-accessible), else 0 */
int ouster_hip_host_is_pinned(const void* p, size_t bytes);
/* give cached (free) pool blocks back to the system until at most keep_bytes stay cached */
void ouster_hip_host_pool_trim(size_t keep_bytes);

/* Process-wide allocation counters of this library (device memory: every hipMalloc / hipFree it makes,
 * ouster_hip_device_alloc included; pinned: the pool's hipHostMalloc / hipHostFree calls).  The
 * steady-state contract of the frame-at-a-time calls -- no allocation once the shapes have been
 * seen -- is checked by reading these before and after (tests/cpp/test_core_api.cpp, test_dropin_host_containers). */
typedef struct ouster_hip_alloc_stats {
    uint64_t device_allocs, device_frees;
    uint64_t pinned_allocs, pinned_frees;
    uint64_t pool_requests, pool_hits;       /* host_alloc calls at pool sizes / served from cached blocks */
    uint64_t pool_live_bytes, pool_cached_bytes;
} ouster_hip_alloc_stats;
void ouster_hip_alloc_stats_read(ouster_hip_alloc_stats* out);
/* counted hipMalloc / hipFree on the context's device, for bindings that keep their own device buffers */
int ouster_hip_device_alloc(ouster_hip_ctx* ctx, size_t bytes, void** out);
void ouster_hip_device_free(void* p);

/* destagger_into<T> on host images (impl/lidar_frame_impl.h:733-760): src / dst are HOST pointers. */
int ouster_hip_destagger_host(ouster_hip_ctx* ctx, const void* src, void* dst, uint32_t h, uint32_t w,
                              uint32_t elem_bytes, const int32_t* shifts, uint32_t n_shifts, int inverse);
/* XYZLutT<T>::operator()(range) / cartesianT<T> on host arrays (xyzlut.h:139-150, impl/cartesian.h:36-66) */
int ouster_hip_cartesian_host(ouster_hip_ctx* ctx, const ouster_hip_lut* lut, const uint32_t* range,
                              void* xyz, int xyz_dtype);
/* dewarp<T>(dewarped, points, poses) on host arrays (pose_util.h:38-56); poses [w][16] f64 */
int ouster_hip_dewarp_host(ouster_hip_ctx* ctx, const void* points, const double* poses, void* dewarped,
                           int dtype, uint32_t h, uint32_t w);
/* Asynchronous copies between host memory of any provenance and device memory, ordered on the context's
 * stream; complete after ouster_hip_sync.  (Pool memory: one DMA.  Other memory: the runtime's
 * pageable route, which tools/copybench measures at the pinned rate from 2 MB on.) */
int ouster_hip_copy_in(ouster_hip_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int ouster_hip_copy_out(ouster_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
/* Grow-only device scratch owned by the context (slot 0..7; contents undefined; valid until the same slot is
 * asked for more).  What the *_host calls stage through; bindings that stage themselves use slots 4..7. */
int ouster_hip_ctx_scratch(ouster_hip_ctx* ctx, uint32_t slot, size_t bytes, void** out);

/* ---- instrumentation ------------------------------------------------------ */
/* Average duration in ms of the dominant decode kernel over the launches made
 * since the last reset, measured with HIP events on the context's stream
 * (enabled by ouster_hip_timing_enable; adds two event records per timed launch: on == 1 times every launch,
 * on == N > 1 every N-th -- an event pair costs the stream 2 - 3 us, a caller inside a timed region samples). */
int ouster_hip_timing_enable(ouster_hip_ctx* ctx, int on);
int ouster_hip_timing_read(ouster_hip_ctx* ctx, double* avg_ms, uint32_t* n_launches);
/* Tile (columns x rows) of the decode kernel variant the last ouster_hip_decode launched: 64/32/16
 * columns x all rows (k_decode) or 64..512 columns x a row chunk (k_decode_wide).  The variant is
 * picked per workload by timing each candidate twice on the first six calls (knob "tune" = 0 disables
 * that, knob "wide" forces one). */
int ouster_hip_last_decode_tile(ouster_hip_ctx* ctx, int* tile_cols, int* tile_rows);
/* Name of the optimistic-pass kernel the last ouster_hip_decode launched: "k_decode" (64/32/16-column tiles,
 * also every general-mapping launch), "k_decode_wide", or one of the two persistent kernels (tiles double-buffered
 * through LDS-DMA; knob "stream" = 0 disables them, 128 / 256 force that tile width): "k_decode_stream2" (dedicated
 * loader waves, the default) or "k_decode_stream" (every wave fetches; knob "stream_loader" = 0). */
const char* ouster_hip_last_decode_kernel(ouster_hip_ctx* ctx);

/* Persisted verdicts of the variant tuner.  ouster_hip_decode picks its kernel variant per workload shape by timing up to
 * five candidates four times each on the first calls (above); with a cache file a process that finds its workload -- keyed by
 * device (arch, CU count, name), library version, profile, H, W, channel bytes, batch-size class and output set -- launches the
 * recorded variant from its FIRST call and times nothing, and a process that had to measure appends its verdict (one O_APPEND
 * write per verdict: the ranks of one job may share the file).  path NULL or "": no cache (the default; the environment
 * variable OUSTER_HIP_TUNING_CACHE, read in ouster_hip_ctx_create, sets one for every context of the process).  Results do
 * not depend on the variant, only the speed does.  What is NOT persisted: where output buffers live (the placement search of
 * hip::DeviceFrameBatch) -- that is a property of the physical pages an allocation drew, gone with the process. */
int ouster_hip_ctx_set_tuning_cache(ouster_hip_ctx* ctx, const char* path);
/* How the last ouster_hip_decode chose its variant: "cache" (read from the file), "measured" (timed by this context),
 * "measuring" (still timing: the call ran a candidate) or "none" (forced by a knob / a shape with one variant). */
const char* ouster_hip_last_decode_tuner(ouster_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* OUSTER_HIP_H */
