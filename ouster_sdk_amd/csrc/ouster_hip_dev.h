// ouster_hip_dev.h -- kernel argument blocks shared by the kernels and the C ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/ouster_hip.h"

namespace ouster_hip_dev {

enum SpecId { SPEC_GENERIC = 0, SPEC_DUAL_LB, SPEC_LB, SPEC_SINGLE, SPEC_DUAL, SPEC_LEGACY };

struct Geometry {
    uint32_t pixels_per_column, columns_per_packet, columns_per_frame;
    uint32_t packet_header_size, col_header_size, channel_data_size, col_footer_size,
        packet_footer_size, col_size, lidar_packet_size;
    ouster_hip_bits col_timestamp, col_measurement_id, col_status;
    ouster_hip_bits frame_id, alert_flags, thermal_shutdown, shot_limiting,
        countdown_thermal_shutdown, countdown_shot_limiting;
};

// device view of one XYZ lookup table
struct LutDev {
    const double* beam_tab;  // [h][9]  U(3) V(3) Wb(3), x range_unit, rotated   (separable mode)
    const double* col_tab;   // [w][5]  cos(theta_e) sin(theta_e) Kc(3)           (separable mode)
    const void* full_dir;    // [w*h][3] full LUT (full mode)
    const void* full_ofs;
    double n;                // beam_to_lidar euclidean distance (raw range units)
    int32_t full_dtype;      // OUSTER_HIP_F32 / F64 for full_dir/full_ofs, 0 when separable
    int32_t pad;
};

// how a decode launch finds the source column of a destination column
enum DecodeMode : uint32_t {
    MODE_FAST = 0,     // optimistic: slot s holds column s; verified against the staged headers, strays
                       //   flag their frame in frame_state
    MODE_FIXUP = 1,    // second pass: redo the frames the fast pass flagged (the others return at once)
    MODE_GENERAL = 2,  // every frame through the scan-the-frame path (slots_per_frame*cpp != W)
    MODE_RESOLVED = 3, // small batches, one launch: every wide tile resolves its frame's column maps itself (k_decode_wide_resolved)
};

// frame_state words (kernels_common.h): sequence, tag, 2 x 8 ticket counters, the launch-wide "a frame was flagged" word
// (its own cache line: only ever touched by atomics), then one word per frame (from a line boundary on)
constexpr uint32_t FS_SEQ = 0, FS_TAG = 1, FS_TICKET = 2, FS_ANY = 20, FS_WORDS = 32;

// fix-up crew (wide_tile.h): frames listed per round, and the crew's bookkeeping in dynamic LDS (DecodeArgs::crew_lds_off)
constexpr uint32_t FIXUP_CHUNK = 512;
struct CrewLds {
    uint16_t list[FIXUP_CHUNK];      // flagged frames of the chunk, in frame order
    uint32_t cnt[FIXUP_CHUNK / 64];
    uint32_t n, nvalid, dirty, pad;
    unsigned long long ticket, ready;
};

struct DecodeArgs {
    Geometry g;
    const uint8_t* packets;
    size_t packet_stride;
    uint32_t slots_per_frame;
    uint32_t n_frames;
    uint32_t tiles_per_frame;
    uint32_t xcd_map;        // 1: blockIdx -> (frame, tile) keeps a frame on one XCD
    uint32_t vec_ok;         // W % 4 == 0 and all output bases/strides 16 B aligned
    uint32_t any_destagger;
    uint32_t rows_per_tile;   // k_decode_wide: rows of a tile, row chunks per column tile,
    uint32_t row_chunks;      //   bytes of one column's LDS slot (tiles_per_frame = column tiles)
    uint32_t lds_col_slot;
    uint32_t mode;            // DecodeMode
    uint32_t beam_lds;        // k_decode: the per-beam xyz table is staged in LDS (it fits without costing a workgroup per CU)
    uint32_t n_packets_out;   // W / cpp: length of the packet-level outputs
    const uint32_t* packet_counts;    // device [n_frames], nullable (= slots_per_frame)
    const uint64_t* host_timestamps;  // device [n_frames][slots_per_frame], nullable
    uint64_t* frame_state;            // device [2 + n_frames] scratch, see FS_* in kernels_common.h
    uint16_t* gate_counts;            // device [n_frames][OUSTER_HIP_GATE_CHUNKS][W] or nullptr (range gate by-product)
    uint32_t gate_min, gate_max;
    int32_t gate_field;               // desc index of the gated range field
    uint16_t* tile_valid;             // device [n_frames][column tiles]: valid columns per tile (fast pass)
    const int32_t* dst_offsets;  // [H] destination column offset per row (device)
    const LutDev* luts;          // [n_luts] (device)
    uint32_t n_luts;
    uint32_t n_fields;
    void* planes[OUSTER_HIP_MAX_FIELDS];
    void* destaggered[OUSTER_HIP_MAX_FIELDS];
    ouster_hip_bits bits[OUSTER_HIP_MAX_FIELDS];
    uint8_t elem[OUSTER_HIP_MAX_FIELDS];
    uint32_t f16_nan_mask;    // bit i: plane i's "zero" is the f16 NaN pattern (a register operand, never a memory read in the row loop)
    int8_t desc_of_spec[16];  // static spec field k -> index into planes[] (-1: not requested)
    uint64_t* timestamp;
    uint16_t* measurement_id;
    uint32_t* status;
    uint64_t* packet_timestamp;
    uint8_t* alert_flags;
    ouster_hip_frame_meta* frame_meta;
    void* xyz[2];
    int32_t xyz_field[2];
    int32_t xyz_dtype;
    int32_t* slot_map;        // device [n_frames][W] or nullptr.  General mapping on wide tiles: k_slotmap writes, per
                              //   destination column, the LAST buffer slot whose live column carries that measurement_id
                              //   (-1: none), k_decode_wide takes its source columns from it instead of "slot c holds column c"
    int32_t* hdr_map;         // device [n_frames][W], with slot_map: the slot whose HEADER lands in destination column c (differs from
                              //   slot_map only for all-valid packets with non-consecutive ids: the reference's block path)
    uint32_t fix_rows_small;  // fix-up pass: rows of a tile when few frames are flagged (rows_per_tile otherwise)
    uint32_t ready_off;       // frame_state word index of the fix-up pass's per-frame ready words (k_decode_wide_fixup)
    uint32_t* hdr_words;      // device [n_frames][W] or nullptr.  The optimistic pass leaves every slot's (measurement_id | valid << 16)
                              //   here, packed: the fix-up pass resolves a flagged frame from 8 KB of consecutive words instead of
                              //   2048 column headers a kilobyte apart (7 us of a workgroup's address unit per tile)
    uint32_t wide_img_words;  // k_decode_wide: LDS words of the tile image (set by the launcher; the fix-up pass keeps resolve_frame's scratch there)
    uint32_t fast_tiles;      // fix-up pass: column tiles of the optimistic pass before it (slots of tile_valid per frame)
    const double* xyz_poses;  // device [n_frames][W][16] or nullptr: per-column pose applied to the xyz outputs
    uint32_t pose_lds_off;    // byte offset of the tile's pose table in dynamic LDS (set by the launchers)
    uint32_t crew_lds_off;    // byte offset of the crew's bookkeeping (CrewLds) in dynamic LDS (set by the launchers)
#ifdef OUSTER_PHASE_TIMING
    uint64_t* phase_times;   // experiment builds only (tools/ab/phase_timing.sh): [workgroup][8] s_memtime stamps of k_decode_wide
#endif
};

// k_decode_stream (persistent, double-buffered tiles filled by LDS-DMA): what the host works out
// once per launch.  A tile context in LDS = the pixel image (TW column blocks of `ncell` 16 B cells, block of column j at
// index j/4 + (TW/4)*(j%4)), then the column-header dwords [n_hdr][TW], the packet-level dwords [n_pkt + 2][64], the
// destagger offsets of the tile's rows and their per-beam xyz constants.
struct FieldPlan {   // a 64-bit field window assembled from fetched dwords: window dword k comes from slot[k] (-1: not needed)
    int8_t slot[3];
    uint8_t sh;      // bit position of the window inside window dword 0
};
struct StreamArgs {
    uint32_t tr, nch;          // rows per tile, row chunks per frame
    uint32_t ncell;            // 16 B cells per column block (the piece of tr rows plus its 16 B phase)
    uint32_t npix_instr;       // 1 KB wave-instructions that fill the pixel image
    uint32_t hdr_off, pkt_off, off_off, beam_off, ctx_bytes;  // byte offsets inside a tile context / its size
    uint32_t fixed_off;        // byte offset of the per-workgroup tables behind the two contexts
    uint32_t n_hdr, n_pkt;     // dwords fetched per column / per packet
    uint32_t hdr_dw[8];        //   their dword offsets from the column start
    uint32_t pkt_dw[4];        //   ... from the packet start
    FieldPlan mid, st, ts, alert;
    uint32_t groups;           // workgroups per (XCD, column tile)
    uint32_t wait0;            // 1: vmcnt(0) before a prefetched tile is used; 0: rely on the in-order counter (>= 63 stores since)
    uint32_t lds_bytes;
    uint32_t order;            // how a group walks the XCD's (frame, row chunk) items, see the kernel
    uint32_t loader;           // > 0: k_decode_stream2 with that many loader waves behind the eight decoding ones (1..4)
};

struct DestaggerArgs {
    const void* src;
    void* dst;
    uint32_t h, w, elem;
    const int32_t* offsets;  // [h] device, reference arithmetic applied on the host
};

struct CartesianArgs {
    LutDev lut;
    const uint32_t* range;
    void* xyz;
    uint32_t w, h, n_images;
    int32_t xyz_dtype;
    uint32_t vec_ok;
    uint32_t rows_per_block;  // tiled kernels: rows of a 64-column tile handled by one workgroup
    uint32_t images_per_block;  // full-LUT mode: images that share one fetch of a tile's LUT rows (blockIdx.y = image group)
};

struct DewarpArgs {
    const void* points;
    void* out;
    const double* poses;  // [n_images][w][16]
    uint32_t w, h, n_images;
    int32_t dtype;
    uint32_t rows_per_block;
};

// range-gated, compacting frame dewarp (impl/dewarp_impl.h:23-115)
struct DewarpFramesArgs {
    const uint32_t* range;      // [n_frames][h][w] staggered RANGE planes
    const uint32_t* status;     // [n_frames][w]
    const uint64_t* timestamp;  // [n_frames][w], nullable unless timestamps_ns is set
    const double* poses;        // [n_frames][w][16]
    const float* pose_rows;     // [n_frames][w][12]: rows 0..2 already cast to float (poses unused then; float output only)
    const LutDev* luts;         // device array, luts[f % n_luts]
    uint32_t n_luts;
    uint32_t w, h, n_frames;
    uint32_t min_r, max_r;      // raw range units (mm), inclusive
    int32_t dtype;              // element type of points
    uint32_t* col_off;          // scratch [n_frames][w + 1] (three-kernel path)
    const uint16_t* gate_counts; // [n_frames][OUSTER_HIP_GATE_CHUNKS][w] from the decode kernel, or nullptr (k_dwf_count runs)
    uint64_t* tile_state;       // scratch [128 + n_frames * tiles] (single-pass path; nullptr: three kernels)
    uint64_t* frame_off;        // out [n_frames + 1]: exclusive prefix of points per frame
    void* points;               // [capacity][3]
    uint32_t* frame_idxs;       // nullable provenance outputs, [capacity]
    uint32_t* col_idxs;
    uint64_t* timestamps_ns;
    uint64_t capacity;
};

// OSF field planes (ouster_hip_osf_unpack): the jobs live in a device array
struct OsfUnpackArgs {
    const ouster_hip_osf_plane* planes;  // device [n]
    const int32_t* offsets;              // device [h]: stagger = destagger with inverse offsets
    uint32_t h, w;
};

// PNG scanline filters reversed on the device (ouster_hip_osf_plane::flags & OUSTER_HIP_OSF_FLAG_FILTERED): one job per image
struct OsfUnfilterJob {
    const uint8_t* raw;   // device: h x (1 + w * bpp) bytes, filter type first
    uint8_t* out;         // device: h x w * bpp bytes
    uint32_t bpp, pad;
};
struct OsfUnfilterArgs {
    const OsfUnfilterJob* jobs;   // device [n]
    uint32_t h, w;
};

// one compile-time field of a standard profile (see the Spec* tables in the kernels file)
struct FieldC {
    uint32_t offset;
    uint64_t mask;
    int32_t shift;
    uint32_t elem;
};
const FieldC* spec_fields(int spec_id, int* nf, uint32_t* chan, int* r1, int* r2);

// LDS of one k_decode workgroup (the general modes add the per-frame packet map and valid bitmap)
size_t decode_lds_bytes(const Geometry& g, int tile, bool general, bool beam_lds, uint32_t slots_per_frame = 0);
size_t decode_wide_lds_bytes(int tw, uint32_t rows_per_tile, uint32_t img_words);
// device: HIP device ordinal of the stream (per-device cache of the one-off kernel attributes)
hipError_t launch_decode(const DecodeArgs& a, int spec_id, int tile, int xyzm, int device, hipStream_t st);
hipError_t launch_decode_wide(const DecodeArgs& a, int spec_id, int tw, int xyzm, int device, hipStream_t st, uint32_t resident = 0);
hipError_t launch_decode_stream(const DecodeArgs& a, const StreamArgs& sp, int spec_id, int tw, int xyzm, int device, hipStream_t st);
size_t slotmap_lds_bytes(uint32_t W, uint32_t cpp, uint32_t slots_per_frame);   // resolve_frame's LDS scratch
hipError_t launch_slotmap(const DecodeArgs& a, int device, hipStream_t st);
hipError_t launch_destagger(const DestaggerArgs& a, uint32_t n_images, hipStream_t st);
hipError_t launch_cartesian(const CartesianArgs& a, int mode, hipStream_t st);
hipError_t launch_dewarp(const DewarpArgs& a, hipStream_t st);
// stream_mode: -1 auto | 0 never | 1 always where eligible -- the persistent k_dwf_emit_stream instead of k_dwf_emit (knob "dwf_stream")
hipError_t launch_dewarp_frames(const DewarpFramesArgs& a, bool separable, hipStream_t st, int stream_mode = -1);
hipError_t launch_osf_unpack(const OsfUnpackArgs& a, uint32_t n_planes, hipStream_t st);
hipError_t launch_osf_png_unfilter(const OsfUnfilterArgs& a, uint32_t n_jobs, uint32_t max_row_bytes, hipStream_t st);
// dynamic LDS that launch needs (per-lane rings sized by the widest pixel + the band's last row): what ouster_hip_osf_unpack checks
size_t osf_png_unfilter_lds_bytes(uint32_t w, uint32_t max_row_bytes);

}  // namespace ouster_hip_dev
