/*
 * ouster_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's per-pixel hot path
 *   packet-format field decode -> LidarFrame, destagger, make_xyz_lut, cartesian
 * so that the HIP path can be checked against it.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's cpu_baseline leg may link or call this.
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.
 *
 * Parity pin: this oracle reproduces every md5 in
 * tests/pcaps/<capture>_digest.json (per-packet and first-frame) and the per-field
 * hash snapshots of tests/frame_batcher_test.cpp:553-595 -- see
 * tests/test_oracle_golden.py.  XYZ is pinned against the closed-form formula
 * of python/src/ouster/sdk/examples/reference.py:19-70.
 */
#ifndef OUSTER_ORACLE_H
#define OUSTER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ChanFieldType values, ouster_core/include/ouster/core/chanfield.h:111-128 */
enum {
    ORA_VOID = 0, ORA_U8 = 1, ORA_U16 = 2, ORA_U32 = 3, ORA_U64 = 4,
    ORA_I8 = 5, ORA_I16 = 6, ORA_I32 = 7, ORA_I64 = 8,
    ORA_F32 = 9, ORA_F64 = 10, ORA_CHAR = 11, ORA_F16 = 12
};

/* UDPProfileLidar values, ouster_core/include/ouster/core/data_format.h:27-72 */
enum {
    ORA_PROFILE_UNKNOWN = 0,
    ORA_PROFILE_LEGACY = 1,
    ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_DUAL = 2,
    ORA_PROFILE_RNG19_RFL8_SIG16_NIR16 = 3,
    ORA_PROFILE_RNG15_RFL8_NIR8 = 4,
    ORA_PROFILE_FIVE_WORD_PIXEL = 5,
    ORA_PROFILE_FUSA_RNG15_RFL8_NIR8_DUAL = 6,
    ORA_PROFILE_RNG15_RFL8_NIR8_DUAL = 7,
    ORA_PROFILE_RNG15_RFL8_NIR8_ZONE16 = 8,
    ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_ZONE16 = 9,
    ORA_PROFILE_RNG15_RFL8_WIN8 = 10,
    ORA_PROFILE_RNG19_RFL8_SIG16_ZONE16_DUAL = 11,
    ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_RGB16 = 12,
    ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_RGB16_DUAL = 13,
    ORA_PROFILE_CUSTOM_BASE = 14 /* first id handed out by ora_add_custom_profile */
};

enum { ORA_HEADER_STANDARD = 0, ORA_HEADER_FUSA = 1 };

#define ORA_MAX_FIELDS 32
#define ORA_NAME_LEN 24

/* FieldDecodeInfo, ouster_core/include/ouster/core/field_decode_info.h:24-30 */
typedef struct {
    int32_t ty_tag;
    int32_t shift;
    int32_t num_elements;
    int32_t pad_;
    uint64_t offset;
    uint64_t mask;
} ora_fdi;

typedef struct {
    char name[ORA_NAME_LEN];
    ora_fdi info;
} ora_field;

/* PacketFormat (lidar part), ouster_core/src/parsing.cpp:386-626 */
typedef struct {
    int32_t profile;
    int32_t header_type;
    uint32_t pixels_per_column;
    uint32_t columns_per_packet;
    uint32_t columns_per_frame;
    uint32_t max_frame_id;
    uint64_t packet_header_size;
    uint64_t col_header_size;
    uint64_t channel_data_size;
    uint64_t col_footer_size;
    uint64_t packet_footer_size;
    uint64_t col_size;
    uint64_t lidar_packet_size;
    int32_t n_fields;
    int32_t pad_;
    ora_field fields[ORA_MAX_FIELDS]; /* sorted by name, like std::map */
    ora_fdi packet_type_info, frame_id_info, init_id_info, prod_sn_info,
        alert_flags_info, countdown_thermal_shutdown_info,
        countdown_shot_limiting_info, thermal_shutdown_info, shot_limiting_info,
        col_status_info, col_timestamp_info, col_measurement_id_info;
} ora_pf;

/* one named plane of a LidarFrame (PIXEL_FIELD), row-major h x w x n_extra */
typedef struct {
    char name[ORA_NAME_LEN];
    int32_t ty_tag;
    int32_t n_extra; /* 1, or 3 for RGB (h x w x 3 float16) */
    void* data;
} ora_plane;

/* minimal LidarFrame, ouster_core/include/ouster/core/lidar_frame.h:124-821 */
typedef struct {
    uint32_t h, w, cpp, n_packets;
    int32_t n_planes;
    int32_t pad_;
    ora_plane planes[ORA_MAX_FIELDS];
    uint64_t* timestamp;        /* [w] */
    uint16_t* measurement_id;   /* [w] */
    uint32_t* status;           /* [w] */
    uint64_t* packet_timestamp; /* [n_packets] */
    uint8_t* alert_flags;       /* [n_packets] */
    int64_t frame_id;
    uint64_t frame_status;
    uint16_t shutdown_countdown;
    uint16_t shot_limiting_countdown;
    uint32_t pad2_;
} ora_frame;

typedef struct ora_batcher ora_batcher;

/* ---- field decode ---- */
int ora_field_info(uint64_t bit_start, uint64_t bit_size, uint64_t upshift,
                   uint64_t max_length, uint64_t num_elements, ora_fdi* out);
uint64_t ora_fdi_get(const ora_fdi* f, const uint8_t* buf);
void ora_fdi_set(const ora_fdi* f, uint8_t* buf, uint64_t value);
uint64_t ora_value_mask(const ora_fdi* f);
size_t ora_type_size(int ty_tag);

/* ---- packet format ---- */
int ora_add_custom_profile(const ora_field* fields, int n_fields,
                           uint64_t chan_data_size, const int32_t* slot_types);
int ora_pf_init(ora_pf* pf, int profile, int header_type, uint32_t h,
                uint32_t cpp, uint32_t w);
const ora_fdi* ora_pf_field(const ora_pf* pf, const char* name);
int ora_default_planes(int profile, char names[][ORA_NAME_LEN], int32_t* types,
                       int32_t* n_extra, int max_n);
int ora_block_parsable(const ora_pf* pf);
const uint8_t* ora_nth_col(const ora_pf* pf, size_t i, const uint8_t* lidar_buf);
uint32_t ora_frame_id(const ora_pf* pf, const uint8_t* buf);
uint32_t ora_init_id(const ora_pf* pf, const uint8_t* buf);
uint64_t ora_prod_sn(const ora_pf* pf, const uint8_t* buf);
uint16_t ora_packet_type(const ora_pf* pf, const uint8_t* buf);
uint8_t ora_alert_flags(const ora_pf* pf, const uint8_t* buf);
uint16_t ora_col_measurement_id(const ora_pf* pf, const uint8_t* col_buf);
uint64_t ora_col_timestamp(const ora_pf* pf, const uint8_t* col_buf);
uint32_t ora_col_status(const ora_pf* pf, const uint8_t* col_buf);
uint32_t ora_col_encoder(const ora_pf* pf, const uint8_t* col_buf);
uint16_t ora_col_frame_id(const ora_pf* pf, const uint8_t* col_buf);
int ora_frame_id_difference(const ora_pf* pf, uint32_t current, uint32_t other);
int ora_col_field(const ora_pf* pf, const uint8_t* col_buf, const char* name,
                  void* dst, size_t dst_elem_size, int dst_stride);
int ora_block_field(const ora_pf* pf, void* data, size_t dst_elem_size,
                    int cols, const char* name, const uint8_t* lidar_buf,
                    int block_dim);
int ora_set_block(const ora_pf* pf, const void* data, size_t elem_size,
                  int cols, const char* name, uint8_t* lidar_buf);
uint64_t ora_crc64(const uint8_t* buf, size_t len);

/* ---- frame ---- */
ora_frame* ora_frame_new(uint32_t h, uint32_t w, uint32_t cpp);
void ora_frame_free(ora_frame* f);
int ora_frame_add_plane(ora_frame* f, const char* name, int ty_tag, int n_extra);
int ora_frame_add_default_planes(ora_frame* f, int profile, int with_window);
void* ora_frame_plane(ora_frame* f, const char* name);
int ora_frame_plane_type(const ora_frame* f, const char* name);
void ora_frame_fill(ora_frame* f, int byte_value); /* memset every plane+header */

/* ---- batcher ---- */
ora_batcher* ora_batcher_new(const ora_pf* pf, int64_t init_id,
                             uint32_t expected_lidar_packets);
void ora_batcher_free(ora_batcher* b);
void ora_batcher_reset(ora_batcher* b);
void ora_batcher_force_col_path(ora_batcher* b, int on);
/* returns 1 when the frame is complete, 0 otherwise, <0 on error */
int ora_batcher_batch(ora_batcher* b, const uint8_t* packet, size_t len,
                      uint64_t host_timestamp, ora_frame* frame);
uint64_t ora_batcher_dropped(const ora_batcher* b);
/* test helper: finalize_frame() of the frame in progress (tail zero-fill) */
int ora_batcher_finalize(ora_batcher* b, ora_frame* frame);

/* ---- frame -> packets (test-side packet synthesis) ---- */
/* writes up to n_packets packets of pf->lidar_packet_size bytes into out,
 * host timestamps into out_ts; returns number emitted */
int ora_frame_to_packets(const ora_frame* f, const ora_pf* pf, uint32_t init_id,
                         uint64_t prod_sn, uint8_t* out, uint64_t* out_ts);

/* ---- destagger ---- */
int ora_destagger(const void* img, void* out, size_t h, size_t w,
                  size_t elem_size, const int32_t* pixel_shift_by_row,
                  size_t n_shifts, int inverse);

/* ---- xyz lut / cartesian ---- */
/* direction/offset: [w*h][3] doubles row-major.  n_angles = h (OS sensors) or
 * w*h (DF sensors).  mats are 4x4 row-major. returns 0, or <0 on bad dims */
int ora_make_xyz_lut(size_t w, size_t h, double range_unit,
                     const double* beam_to_lidar, const double* transform,
                     const double* azimuth_deg, const double* altitude_deg,
                     size_t n_angles, double* direction, double* offset);
void ora_cartesian_f64(double* points, const uint32_t* range, const double* dir,
                       const double* ofs, size_t n);
void ora_cartesian_f32(float* points, const uint32_t* range, const float* dir,
                       const float* ofs, size_t n);
/* same loops with an OpenMP parallel-for (impl/cartesian.h:50-52, -DOUSTER_OMP) */
void ora_cartesian_f64_omp(double* points, const uint32_t* range,
                           const double* dir, const double* ofs, size_t n);
void ora_cartesian_f32_omp(float* points, const uint32_t* range,
                           const float* dir, const float* ofs, size_t n);

/* ---- dense dewarp (pose_util.h:38-56) ---- */
void ora_dewarp_f64(double* out, const double* pts, const double* poses, size_t h, size_t w);
void ora_dewarp_f32(float* out, const float* pts, const double* poses, size_t h, size_t w);
/* ---- range-gated, compacting frame dewarp (impl/dewarp_impl.h:23-81) ---- */
size_t ora_dewarp_frame_f64(double* out, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                            const uint32_t* status, const uint64_t* timestamp, const double* poses,
                            const double* dir, const double* ofs, size_t h, size_t w,
                            double min_range, double max_range);
size_t ora_dewarp_frame_f32(float* out, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                            const uint32_t* status, const uint64_t* timestamp, const double* poses,
                            const float* dir, const float* ofs, size_t h, size_t w,
                            double min_range, double max_range);

/* ---- whole hot path for the CPU baseline (decode + destagger + cartesian) ---- */
/* Runs n_frames frames (frame f = pool frame f % pool_frames) of `ppf` packets each through
 * batcher -> destagger of the named planes -> cartesian (f64 if xyz_f64 else f32) for RANGE
 * (+RANGE2).  Frames are independent; `threads` > 1 distributes frames with OpenMP.
 * Returns seconds elapsed for the timed loop (reps passes over the pool). */
double ora_bench_hot_path(const ora_pf* pf, int with_window,
                          const uint8_t* packets, uint32_t pool_frames, uint32_t n_frames,
                          uint32_t ppf, const int32_t* pixel_shift_by_row,
                          const double* lut_dir, const double* lut_ofs,
                          int xyz_f64, int reps, int threads,
                          uint64_t* checksum_out);
/* flags: 1 = per-thread first-touched copies of LUT / packets, 2 = static schedule (see the .c file) */
double ora_bench_hot_path2(const ora_pf* pf, int with_window, const uint8_t* packets,
                           uint32_t pool_frames, uint32_t n_frames, uint32_t ppf, const int32_t* shifts,
                           const double* lut_dir, const double* lut_ofs, int xyz_f64,
                           int reps, int threads, uint64_t* checksum_out, int flags);
double ora_bench_stream_copy(size_t bytes_per_thread, int reps, int threads);

#ifdef __cplusplus
}
#endif
#endif
