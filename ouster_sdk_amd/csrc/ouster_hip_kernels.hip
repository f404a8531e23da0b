// ouster_hip_kernels.hip -- CDNA4 (gfx950) kernels for the Ouster per-pixel hot path.
//
//   k_colmap        column headers -> per-frame "destination column -> source column" map
//   k_decode        fused field decode (+ destagger + cartesian), one workgroup per
//                   64/32/16-column tile of one frame (whole columns staged in LDS)
//   k_decode_wide   the same on wide, short tiles (128/256 columns x a chunk of rows); picked per
//                   workload against k_decode by the tuner in ouster_hip_capi.hip
//   k_destagger     standalone per-row circular shift
//   k_cartesian(_tiled)   standalone range image -> XYZ
//   k_dewarp(_tiled)      standalone per-column pose applied to a dense point cloud
//   k_dwf_*         range-gated, compacting frame dewarp (count, scans, emit)
//
// What the kernels compute is defined by the reference loops
//   PacketFormat::col_field/block_field      ouster_core/src/parsing.cpp:628-675
//   FieldDecodeInfo::get                     ouster_core/include/ouster/core/field_decode_info.h:41-54
//   FrameBatcher::parse_by_col/_by_block     ouster_core/src/lidar_frame.cpp:1422-1528
//   destagger_into<T>                        ouster_core/include/ouster/core/impl/lidar_frame_impl.h:733-760
//   impl::make_xyz_lut / cartesianT<T>       ouster_core/src/xyzlut.cpp:11-89, impl/cartesian.h:36-66
// How they compute it is MI355X-specific; see DESIGN.md.
//
// Memory-bound byte/bit work: no MFMA anywhere.  The wire format is column
// major (one column = H consecutive pixels), the LidarFrame planes are row
// major H x W, so each workgroup stages a tile of columns in LDS with wide
// coalesced loads and then walks it row-wise, 4 consecutive columns per lane,
// so that every global store is a 16 B (u32 planes / xyz) or packed (u8/u16
// planes) vector store of a contiguous row segment.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ouster_hip_dev.h"

namespace ouster_hip_dev {

// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t apply_bits(uint64_t word, uint64_t mask, int shift) {
    word &= mask;
    if (shift > 0) word >>= shift;
    else if (shift < 0) word <<= -shift;
    return word;
}

__device__ __forceinline__ uint64_t funnel3(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t sh) {
    uint64_t lo = ((uint64_t)d1 << 32) | d0;
    uint64_t v = lo >> sh;
    if (sh) v |= ((uint64_t)d2) << (64 - sh);
    return v;
}

// 64-bit little-endian window at an arbitrary byte address in global memory
__device__ __forceinline__ uint64_t window_global(const uint8_t* p) {
    uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    uint32_t sh = (uint32_t)(a & 3) * 8;
    uint32_t d0 = q[0], d1 = q[1];
    uint32_t d2 = sh ? q[2] : 0u;
    return funnel3(d0, d1, d2, sh);
}

// like window_global, but only the dwords that `mask` (applied to the window) can see are
// loaded; header fields are 1-4 bytes wide, so this is usually a single dword
__device__ __forceinline__ uint64_t window_global_masked(const uint8_t* p, uint64_t mask) {
    if (mask == 0) return 0;
    const uint32_t lo = (uint32_t)__builtin_ctzll(mask) >> 3, hi = (63u - (uint32_t)__builtin_clzll(mask)) >> 3;
    const uintptr_t a = (uintptr_t)p;
    const uintptr_t first = (a + lo) & ~(uintptr_t)3, last = (a + hi) & ~(uintptr_t)3;
    uint64_t v = 0;
    for (uintptr_t q = first; q <= last; q += 4) {
        const uint64_t d = *(const uint32_t*)q;
        const long sh = (long)(q - a) * 8;  // bit position of this dword inside the window
        v |= sh >= 0 ? (sh < 64 ? d << sh : 0) : d >> (-sh);
    }
    return v;
}

// same, from the LDS tile (byte offset into the tile)
__device__ __forceinline__ uint64_t window_lds(const uint32_t* tile, uint32_t byte_off) {
    const uint32_t* q = tile + (byte_off >> 2);
    uint32_t sh = (byte_off & 3) * 8;
    return funnel3(q[0], q[1], q[2], sh);
}

__device__ __forceinline__ uint64_t trunc_elem(uint64_t v, uint32_t elem) {
    return elem >= 8 ? v : (v & ((1ull << (elem * 8)) - 1));
}

// unaligned-capable vector stores (gfx950 global stores only need the HW
// "unaligned access mode", which amdhsa enables; the compiler emits single
// global_store_dword{,x2,x4} for these packed types)
struct __attribute__((packed, aligned(1))) pk4 { uint32_t a; };
struct __attribute__((packed, aligned(1))) pk8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) pk16 { uint32_t a, b, c, d; };

__device__ __forceinline__ void st4(void* p, uint32_t a) { ((pk4*)p)->a = a; }
__device__ __forceinline__ void st8(void* p, uint32_t a, uint32_t b) {
    pk8 v{a, b};
    *((pk8*)p) = v;
}
__device__ __forceinline__ void st16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    pk16 v{a, b, c, d};
    *((pk16*)p) = v;
}

// store 4 consecutive elements of `elem` bytes each starting at byte pointer p
__device__ __forceinline__ void store4(uint8_t* p, const uint64_t v[4], uint32_t elem) {
    switch (elem) {
        case 1:
            st4(p, (uint32_t)(v[0] & 0xff) | ((uint32_t)(v[1] & 0xff) << 8) |
                       ((uint32_t)(v[2] & 0xff) << 16) | ((uint32_t)(v[3] & 0xff) << 24));
            break;
        case 2:
            st8(p, (uint32_t)(v[0] & 0xffff) | ((uint32_t)(v[1] & 0xffff) << 16),
                (uint32_t)(v[2] & 0xffff) | ((uint32_t)(v[3] & 0xffff) << 16));
            break;
        case 4:
            st16(p, (uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
            break;
        case 6: {
            // 4 x 48 bit, little endian, 24 contiguous bytes
            uint64_t a = (v[0] & 0xffffffffffffull) | (v[1] << 48);
            uint64_t b = ((v[1] >> 16) & 0xffffffffull) | (v[2] << 32);
            uint64_t c = ((v[2] >> 32) & 0xffffull) | (v[3] << 16);
            st8(p, (uint32_t)a, (uint32_t)(a >> 32));
            st8(p + 8, (uint32_t)b, (uint32_t)(b >> 32));
            st8(p + 16, (uint32_t)c, (uint32_t)(c >> 32));
            break;
        }
        default:  // 8
            st16(p, (uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32));
            st16(p + 16, (uint32_t)v[2], (uint32_t)(v[2] >> 32), (uint32_t)v[3],
                 (uint32_t)(v[3] >> 32));
    }
}

__device__ __forceinline__ void store1(uint8_t* p, uint64_t v, uint32_t elem) {
    switch (elem) {
        case 1: *p = (uint8_t)v; break;
        case 2: *(uint16_t*)p = (uint16_t)v; break;
        case 4: *(uint32_t*)p = (uint32_t)v; break;
        case 6:
            *(uint16_t*)p = (uint16_t)v;
            *(uint16_t*)(p + 2) = (uint16_t)(v >> 16);
            *(uint16_t*)(p + 4) = (uint16_t)(v >> 32);
            break;
        default: *(uint64_t*)p = v;
    }
}

__device__ __forceinline__ uint64_t zero_value(uint32_t f16_nan) {
    return f16_nan ? 0x7e007e007e007e00ull : 0ull;
}

// ------------------------------------------------------------------------------------
// k_colmap: one thread per received column slot.
//   map[f][m_id] = max over received valid columns of (slot index)   ("last in buffer wins")
//   + packet level outputs + frame meta
// Semantics: FrameBatcher::parse_by_col, ouster_core/src/lidar_frame.cpp:1422-1466
//   (m_id >= W dropped :1432-1434, invalid status dropped :1447-1450) and
//   batch_lidar_packet :1534-1539 (packet_timestamp / alert_flags per packet),
//   start_frame :1709-1741 (frame meta from the first packet).
// ------------------------------------------------------------------------------------
template <int COLMAP_U>
__global__ __launch_bounds__(256) void k_colmap(ColmapArgs a) {
    const uint32_t cpp = a.g.columns_per_packet;
    const uint32_t slots = a.slots_per_frame * cpp;
    const uint32_t f = blockIdx.y;
    const uint32_t s0 = blockIdx.x * (256u * COLMAP_U) + threadIdx.x;
    const uint32_t count = a.packet_counts ? a.packet_counts[f] : a.slots_per_frame;
    if (s0 == 0 && a.frame_meta) {
        ouster_hip_frame_meta m;
        m.frame_id = -1; m.frame_status = 0; m.shutdown_countdown = 0;
        m.shot_limiting_countdown = 0; m.n_valid_columns = 0;
        if (count > 0) {
            const uint8_t* pkt = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
            m.frame_id = (int64_t)(uint32_t)apply_bits(window_global(pkt + a.g.frame_id.offset),
                                                        a.g.frame_id.mask, a.g.frame_id.shift);
            uint8_t th = (uint8_t)apply_bits(window_global(pkt + a.g.thermal_shutdown.offset),
                                             a.g.thermal_shutdown.mask, a.g.thermal_shutdown.shift);
            uint8_t sl = (uint8_t)apply_bits(window_global(pkt + a.g.shot_limiting.offset),
                                             a.g.shot_limiting.mask, a.g.shot_limiting.shift);
            m.frame_status = (uint64_t)(th & 0x0f) | ((uint64_t)(sl & 0x0f) << 4);
            m.shutdown_countdown = (uint16_t)apply_bits(
                window_global(pkt + a.g.countdown_thermal_shutdown.offset),
                a.g.countdown_thermal_shutdown.mask, a.g.countdown_thermal_shutdown.shift);
            m.shot_limiting_countdown = (uint16_t)apply_bits(
                window_global(pkt + a.g.countdown_shot_limiting.offset),
                a.g.countdown_shot_limiting.mask, a.g.countdown_shot_limiting.shift);
        }
        a.frame_meta[f] = m;
    }
    // COLMAP_U column headers per thread, all loads issued before the first use: the reads are
    // scattered 2-4 B accesses (one cache line each), so memory-level parallelism is what counts
    const uint8_t* fpk = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    uint64_t w_mid[COLMAP_U], w_st[COLMAP_U];
    uint32_t sl[COLMAP_U];
#pragma unroll
    for (int u = 0; u < COLMAP_U; ++u) {
        sl[u] = s0 + (uint32_t)u * 256u;
        const uint32_t p = sl[u] / cpp, icol = sl[u] - p * cpp;
        w_mid[u] = w_st[u] = 0;
        if (sl[u] < slots && p < count) {
            const uint8_t* col = fpk + (size_t)p * a.packet_stride + a.g.packet_header_size +
                                 (size_t)icol * a.g.col_size;
            w_mid[u] = window_global_masked(col + a.g.col_measurement_id.offset, a.g.col_measurement_id.mask);
            w_st[u] = window_global_masked(col + a.g.col_status.offset, a.g.col_status.mask);
        }
    }
#pragma unroll
    for (int u = 0; u < COLMAP_U; ++u) {
        const uint32_t p = sl[u] / cpp, icol = sl[u] - p * cpp;
        if (sl[u] >= slots || p >= count) continue;
        const uint32_t m_id = (uint16_t)apply_bits(w_mid[u], a.g.col_measurement_id.mask,
                                                   a.g.col_measurement_id.shift);
        const uint32_t status = (uint32_t)apply_bits(w_st[u], a.g.col_status.mask, a.g.col_status.shift);
        if (icol == 0) {
            const uint8_t* pkt = fpk + (size_t)p * a.packet_stride;
            const uint32_t packet_id = m_id / cpp;
            if (packet_id < a.n_packets_out) {
                if (a.packet_timestamp && a.host_timestamps)
                    a.packet_timestamp[(size_t)f * a.n_packets_out + packet_id] =
                        a.host_timestamps[(size_t)f * a.slots_per_frame + p];
                if (a.alert_flags)
                    a.alert_flags[(size_t)f * a.n_packets_out + packet_id] = (uint8_t)apply_bits(
                        window_global(pkt + a.g.alert_flags.offset), a.g.alert_flags.mask,
                        a.g.alert_flags.shift);
            }
        }
        if ((status & 1u) && m_id < a.g.columns_per_frame)
            atomicMax(&a.map[(size_t)f * a.g.columns_per_frame + m_id], (int32_t)((a.epoch << 20) | sl[u]));
    }
}

// ------------------------------------------------------------------------------------
// global -> LDS staging of one contiguous byte range, all threads of the block
// ------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void stage_range(uint32_t* lds_tile, uint32_t lds_byte_off,
                                            const uint8_t* __restrict__ src, uint32_t nbytes,
                                            uint32_t tid) {
    if ((((uintptr_t)src | lds_byte_off | nbytes) & 15u) == 0) {
        const u32x4* s = (const u32x4*)src;
        u32x4* d = (u32x4*)(lds_tile + (lds_byte_off >> 2));
        const uint32_t n = nbytes >> 4;
#pragma unroll 4
        for (uint32_t i = tid; i < n; i += NT) d[i] = __builtin_nontemporal_load(s + i);
    } else {  // packets are 4-byte granular (parsing.cpp:459-469), so is everything in them
        const uint32_t* s = (const uint32_t*)src;
        uint32_t* d = lds_tile + (lds_byte_off >> 2);
        const uint32_t n = nbytes >> 2;
#pragma unroll 4
        for (uint32_t i = tid; i < n; i += NT) d[i] = s[i];
    }
}

// ------------------------------------------------------------------------------------
// static field tables for the standard profiles (bit layouts: parsing.cpp:170-363,
// plane element sizes: lidar_frame.cpp:73-187).  A runtime format descriptor is
// matched against these at format_create; anything else runs the generic spec.
// ------------------------------------------------------------------------------------
struct SpecDualLB {  // RNG15_RFL8_NIR8_DUAL / FUSA_RNG15_RFL8_NIR8_DUAL, 8 B/px
    static constexpr bool is_static = true;
    static constexpr uint32_t chan = 8;
    static constexpr int nf = 8;
    static constexpr int range_idx = 0, range2_idx = 4;
    static constexpr FieldC f[8] = {{0, 0x7fff, -3, 4}, {1, 0x80, 7, 1},  {2, 0xff, 0, 1},
                                    {3, 0xff, -4, 2},   {4, 0x7fff, -3, 4}, {5, 0x80, 7, 1},
                                    {6, 0xff, 0, 1},    {7, 0xff, 0, 1}};
};
struct SpecLB {  // RNG15_RFL8_NIR8, 4 B/px
    static constexpr bool is_static = true;
    static constexpr uint32_t chan = 4;
    static constexpr int nf = 4;
    static constexpr int range_idx = 0, range2_idx = -1;
    static constexpr FieldC f[4] = {{0, 0x7fff, -3, 4}, {1, 0x80, 7, 1}, {2, 0xff, 0, 1},
                                    {3, 0xff, -4, 2}};
};
struct SpecSingle {  // RNG19_RFL8_SIG16_NIR16, 12 B/px
    static constexpr bool is_static = true;
    static constexpr uint32_t chan = 12;
    static constexpr int nf = 6;
    static constexpr int range_idx = 0, range2_idx = -1;
    static constexpr FieldC f[6] = {{0, 0x7ffff, 0, 4}, {2, 0xf8, 3, 1},   {4, 0xff, 0, 1},
                                    {6, 0xffff, 0, 2},  {8, 0xffff, 0, 2}, {11, 0xff, 0, 1}};
};
struct SpecDual {  // RNG19_RFL8_SIG16_NIR16_DUAL, 16 B/px
    static constexpr bool is_static = true;
    static constexpr uint32_t chan = 16;
    static constexpr int nf = 10;
    static constexpr int range_idx = 0, range2_idx = 3;
    static constexpr FieldC f[10] = {{0, 0x7ffff, 0, 4},  {2, 0xf8, 3, 1},   {3, 0xff, 0, 1},
                                     {4, 0x7ffff, 0, 4},  {6, 0xf8, 3, 1},   {7, 0xff, 0, 1},
                                     {8, 0xffff, 0, 2},   {10, 0xffff, 0, 2}, {12, 0xffff, 0, 2},
                                     {15, 0xff, 0, 1}};
};
struct SpecLegacy {  // LEGACY, 12 B/px
    static constexpr bool is_static = true;
    static constexpr uint32_t chan = 12;
    static constexpr int nf = 5;
    static constexpr int range_idx = 0, range2_idx = -1;
    static constexpr FieldC f[5] = {{0, 0xfffff, 0, 4}, {3, 0xf0, 4, 1}, {4, 0xff, 0, 1},
                                    {6, 0xffff, 0, 2},  {8, 0xffff, 0, 2}};
};
struct SpecGeneric {  // everything else: descriptors read from the kernel arguments
    static constexpr bool is_static = false;
    static constexpr uint32_t chan = 0;
    static constexpr int nf = 0;
    static constexpr int range_idx = -1, range2_idx = -1;
};

const FieldC* spec_fields(int spec_id, int* nf, uint32_t* chan, int* r1, int* r2) {
    switch (spec_id) {
        case SPEC_DUAL_LB: *nf = SpecDualLB::nf; *chan = SpecDualLB::chan; *r1 = 0; *r2 = 4; return SpecDualLB::f;
        case SPEC_LB: *nf = SpecLB::nf; *chan = SpecLB::chan; *r1 = 0; *r2 = -1; return SpecLB::f;
        case SPEC_SINGLE: *nf = SpecSingle::nf; *chan = SpecSingle::chan; *r1 = 0; *r2 = -1; return SpecSingle::f;
        case SPEC_DUAL: *nf = SpecDual::nf; *chan = SpecDual::chan; *r1 = 0; *r2 = 3; return SpecDual::f;
        case SPEC_LEGACY: *nf = SpecLegacy::nf; *chan = SpecLegacy::chan; *r1 = 0; *r2 = -1; return SpecLegacy::f;
        default: *nf = 0; *chan = 0; *r1 = *r2 = -1; return nullptr;
    }
}

// compile-time field extraction from the pixel's dwords held in registers
template <class S, int K, int CW>
__device__ __forceinline__ uint64_t extract_static(const uint32_t (&w)[CW]) {
    constexpr uint32_t off = S::f[K].offset;
    constexpr uint32_t i0 = off / 4, sh = (off % 4) * 8;
    uint64_t lo = w[i0];
    if constexpr (i0 + 1 < CW) lo |= (uint64_t)w[i0 + 1] << 32;
    uint64_t win = lo >> sh;
    if constexpr (sh != 0 && i0 + 2 < CW) win |= (uint64_t)w[i0 + 2] << (64 - sh);
    return trunc_elem(apply_bits(win, S::f[K].mask, S::f[K].shift), S::f[K].elem);
}

// ------------------------------------------------------------------------------------
// XYZ projection of 4 consecutive pixels of one row.
//   separable tables (per-beam x per-column, double math, one rounding on store):
//     dir  = cx*U + sx*V + Wb          (already x range_unit and rotated by `transform`)
//     xyz  = (r - n) * dir + Kc        (Kc: per-column part of the offset)
//   == r*direction + offset of make_xyz_lut (xyzlut.cpp:63-86) up to ~1e-13 m.
//   full LUT (user arrays / per-pixel angle sensors): xyz = r*dir + ofs in the
//   LUT's own precision, as cartesianT<T> does (cartesian.h:53-65).
// ------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void store_xyz4(T* dst, const double (&p)[4][3]);

template <>
__device__ __forceinline__ void store_xyz4<float>(float* dst, const double (&p)[4][3]) {
    float4* d = (float4*)dst;  // 48 B, 16 B aligned (pixel index multiple of 4)
    d[0] = make_float4((float)p[0][0], (float)p[0][1], (float)p[0][2], (float)p[1][0]);
    d[1] = make_float4((float)p[1][1], (float)p[1][2], (float)p[2][0], (float)p[2][1]);
    d[2] = make_float4((float)p[2][2], (float)p[3][0], (float)p[3][1], (float)p[3][2]);
}
template <>
__device__ __forceinline__ void store_xyz4<double>(double* dst, const double (&p)[4][3]) {
    double2* d = (double2*)dst;
    d[0] = make_double2(p[0][0], p[0][1]);
    d[1] = make_double2(p[0][2], p[1][0]);
    d[2] = make_double2(p[1][1], p[1][2]);
    d[3] = make_double2(p[2][0], p[2][1]);
    d[4] = make_double2(p[2][2], p[3][0]);
    d[5] = make_double2(p[3][1], p[3][2]);
}

// f32 xyz of 4 consecutive pixels per lane = 48 contiguous bytes per lane.  Stored directly,
// each of the three 16 B store instructions would touch every third 16 B chunk of the row
// segment.  Instead the wave transposes through a private LDS scratch so that instruction k
// writes chunks [k*LPR, (k+1)*LPR) of the row segment: LPR x 16 B contiguous per row.
//   row_base: xyz address of the first pixel of this lane's row segment (tile column 0)
template <int LPR>
__device__ __forceinline__ void store_xyz4_coalesced(float4* s_xyz, uint32_t tid, float* row_base,
                                                     uint32_t q, const double (&p)[4][3]) {
    const uint32_t wave = tid >> 6, lane = tid & 63u, rho = lane / LPR;
    float4* sc = s_xyz + wave * 192 + rho * (3 * LPR);
    sc[3 * q + 0] = make_float4((float)p[0][0], (float)p[0][1], (float)p[0][2], (float)p[1][0]);
    sc[3 * q + 1] = make_float4((float)p[1][1], (float)p[1][2], (float)p[2][0], (float)p[2][1]);
    sc[3 * q + 2] = make_float4((float)p[2][2], (float)p[3][0], (float)p[3][1], (float)p[3][2]);
    // same-wave LDS write -> read: DS ops of a wave execute in order; keep the compiler from
    // reordering and wait for the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float4* d = (float4*)row_base;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 v = sc[k * LPR + q];
        d[k * LPR + q] = v;
    }
    // the next use of the scratch (second return / next row) must not overtake these reads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Generic forms of the same transpose for a "lane owns 4 consecutive pixels x NV 16 B chunks"
// register block (NV = 3: 4 x f32 xyz, NV = 6: 4 x f64 xyz).  sc = this lane-row's private
// scratch of NV*LPR float4; row_base = address of tile column 0 of the lane's row.
template <int NV, int LPR>
__device__ __forceinline__ void store_quad_coalesced(float4* sc, float4* row_base, uint32_t q,
                                                     const float4 (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sc[NV * q + k] = v[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NV; ++k) row_base[k * LPR + q] = sc[k * LPR + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int NV, int LPR>
__device__ __forceinline__ void load_quad_coalesced(float4* sc, const float4* row_base, uint32_t q,
                                                    float4 (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sc[k * LPR + q] = row_base[k * LPR + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = sc[NV * q + k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <class T>
__device__ __forceinline__ void store_xyz1(T* dst, const double (&p)[3]) {
    dst[0] = (T)p[0]; dst[1] = (T)p[1]; dst[2] = (T)p[2];
}

// full-LUT projection of one pixel; LT = LUT element type
template <class LT>
__device__ __forceinline__ void project_full(const LT* dir, const LT* ofs, size_t pix, uint32_t r,
                                             double (&p)[3]) {
    if (r == 0) { p[0] = p[1] = p[2] = 0.0; return; }
    const LT rr = (LT)r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // separate multiply and add like a default (non-FMA) host build of cartesianT
        LT m = rr * dir[pix * 3 + k];
        asm volatile("" : "+v"(m));  // keep the compiler from contracting into an fma
        p[k] = (double)(LT)(m + ofs[pix * 3 + k]);
    }
}

// ------------------------------------------------------------------------------------
// k_decode: fused decode + destagger + cartesian for one (frame, column tile)
// ------------------------------------------------------------------------------------
// XYZM: 0 no xyz, 1 separable tables -> f32, 2 separable -> f64, 3 full LUT (runtime dtypes)
template <class S, int TILE, int XYZM>
#ifndef OUSTER_DECODE_NT
#define OUSTER_DECODE_NT 256
#endif
__global__ __launch_bounds__(OUSTER_DECODE_NT) void k_decode(DecodeArgs a) {
    constexpr int NT = OUSTER_DECODE_NT;
    constexpr int LPR = TILE / 4;    // lanes per row segment
    constexpr int RPP = NT / LPR;    // rows per pass of the workgroup
    extern __shared__ __align__(16) uint32_t smem[];

    // ---- which (frame, tile)?  XCD-aware: all tiles of a frame on one XCD so that
    // neighbouring tiles' partial cache lines (destaggered rows, u8 planes) merge in
    // that XCD's L2 before they are written back.  Block b is dispatched to XCD b%8.
    uint32_t f, tile;
    const uint32_t tpf = a.tiles_per_frame;
    if (a.xcd_map) {
        const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        f = (i / tpf) * 8u + xcd;
        tile = i % tpf;
        if (f >= a.n_frames) return;
    } else {
        f = blockIdx.x / tpf;
        tile = blockIdx.x - f * tpf;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, H = a.g.pixels_per_column;
    const uint32_t cpp = a.g.columns_per_packet, col_size = a.g.col_size;
    const uint32_t c0 = tile * TILE;

    // ---- LDS carve-up (all offsets 16 B aligned)
    uint32_t* s_tile = smem;                                    // TILE * col_size (+16) bytes
    const uint32_t tile_bytes = (TILE * col_size + 16 + 15) & ~15u;
    int32_t* s_src = (int32_t*)(smem + (tile_bytes >> 2));      // [TILE]
    uint64_t* s_masks = (uint64_t*)(s_src + TILE);              // [0] valid, [1] group-ok
    int32_t* s_off = (int32_t*)(s_masks + 2);                   // [H] destagger offsets
    float4* s_xyz = (float4*)(s_off + ((H + 3) & ~3u));         // [4 waves][192] xyz transpose

    // ---- phase 0: source map of this tile
    if (tid < TILE) {
        const uint32_t c = c0 + tid;
        const int32_t raw = (c < W) ? a.map[(size_t)f * W + c] : -1;
        const int32_t src = (raw >= 0 && ((uint32_t)raw >> 20) == a.epoch) ? (raw & 0xfffff) : -1;  // stale epoch = absent
        s_src[tid] = src;
        const uint32_t j0 = tid - tid % cpp;  // first column of my packet group in the tile
        // "group ok": my packet's cpp columns sit in order, packet-aligned, all present
        int32_t head = __shfl(src, (int)(j0 & 63u));
        bool grp = (TILE % cpp == 0) && src >= 0 && head >= 0 && (uint32_t)head % cpp == 0 &&
                   src == head + (int32_t)(tid - j0);
        uint64_t vb = __ballot(src >= 0);
        uint64_t gb = __ballot(grp);
        if (tid == 0) { s_masks[0] = vb; s_masks[1] = gb; }
    }
    if (a.any_destagger)
        for (uint32_t r = tid; r < H; r += NT) s_off[r] = a.dst_offsets[r];
    const LutDev lut = (XYZM != 0) ? a.luts[f % a.n_luts] : LutDev{};
    __syncthreads();

    // ---- phase 1: stage the tile's columns in LDS, column j at byte j*col_size
    const uint64_t validmask = s_masks[0], groupmask = s_masks[1];
    const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    const uint64_t fullmask = ~0ull >> (64 - TILE);
    const uint32_t gbytes_all = cpp * col_size;
    const bool flat = (TILE % cpp == 0) && (TILE / cpp <= 4) && (groupmask == fullmask) &&
                      (((gbytes_all | a.packet_stride | a.g.packet_header_size |
                         (uint32_t)(uintptr_t)fbase) & 15u) == 0);
    if (flat) {
        // the common case: the tile is G whole packets, all present and in order.  One flat
        // copy with every load of the thread in flight before the first LDS write.
        const uint32_t G = TILE / cpp, n16 = gbytes_all >> 4, total = G * n16;
        const u32x4* src[4];
#pragma unroll
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t p = (g < G) ? (uint32_t)s_src[g * cpp] / cpp : 0u;
            src[g] = (const u32x4*)(fbase + (size_t)p * a.packet_stride + a.g.packet_header_size);
        }
        u32x4* dst = (u32x4*)s_tile;
        constexpr int DEPTH = 17;  // 17 x 256 x 16 B = 68 KB: a 64-column dual-LB tile in one pass
        for (uint32_t base = 0; base < total; base += NT * DEPTH) {
            u32x4 t[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const uint32_t idx = base + k * NT + tid;
                if (idx < total) {
                    const uint32_t g = (idx >= n16) + (idx >= 2 * n16) + (idx >= 3 * n16);
                    const u32x4* sp = g == 0 ? src[0] : g == 1 ? src[1] : g == 2 ? src[2] : src[3];
                    t[k] = __builtin_nontemporal_load(sp + (idx - g * n16));
                }
            }
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const uint32_t idx = base + k * NT + tid;
                if (idx < total) dst[idx] = t[k];
            }
        }
    } else if (TILE % cpp == 0) {
        const uint32_t gbytes = cpp * col_size;
        for (uint32_t j0 = 0; j0 < TILE; j0 += cpp) {
            const uint64_t gm = (cpp >= 64 ? ~0ull : ((1ull << cpp) - 1)) << j0;
            if ((groupmask & gm) == gm) {  // whole packet, one linear copy
                const uint32_t p = (uint32_t)s_src[j0] / cpp;
                stage_range<NT>(s_tile, j0 * col_size,
                                fbase + (size_t)p * a.packet_stride + a.g.packet_header_size,
                                gbytes, tid);
            } else if (validmask & gm) {
                for (uint32_t j = j0; j < j0 + cpp; ++j) {
                    const int32_t s = s_src[j];
                    if (s < 0) continue;
                    const uint32_t p = (uint32_t)s / cpp, ic = (uint32_t)s - p * cpp;
                    stage_range<NT>(s_tile, j * col_size,
                                    fbase + (size_t)p * a.packet_stride +
                                        a.g.packet_header_size + (size_t)ic * col_size,
                                    col_size, tid);
                }
            }
        }
    } else {
        for (uint32_t j = 0; j < TILE; ++j) {
            const int32_t s = s_src[j];
            if (s < 0) continue;
            const uint32_t p = (uint32_t)s / cpp, ic = (uint32_t)s - p * cpp;
            stage_range<NT>(s_tile, j * col_size,
                            fbase + (size_t)p * a.packet_stride + a.g.packet_header_size +
                                (size_t)ic * col_size,
                            col_size, tid);
        }
    }
    if (tid < 4) s_tile[(TILE * col_size >> 2) + tid] = 0;  // slack read by 64-bit windows
    __syncthreads();

    // ---- phase 2a: column headers (timestamp / measurement_id / status), one lane per column
    if (tid < TILE && c0 + tid < W) {
        const uint32_t c = c0 + tid;
        const bool v = (validmask >> tid) & 1;
        const uint32_t cb = tid * col_size;
        if (a.timestamp)
            a.timestamp[(size_t)f * W + c] =
                v ? apply_bits(window_lds(s_tile, cb + a.g.col_timestamp.offset),
                               a.g.col_timestamp.mask, a.g.col_timestamp.shift) : 0ull;
        if (a.measurement_id) a.measurement_id[(size_t)f * W + c] = v ? (uint16_t)c : (uint16_t)0;
        if (a.status)
            a.status[(size_t)f * W + c] =
                v ? (uint32_t)apply_bits(window_lds(s_tile, cb + a.g.col_status.offset),
                                         a.g.col_status.mask, a.g.col_status.shift) : 0u;
    }

    // ---- phase 2b: pixels.  lane = (row within pass, quad of 4 consecutive columns)
    const uint32_t q = tid % LPR, ty = tid / LPR;
    const uint32_t jq = q * 4;                 // first of my 4 columns inside the tile
    const uint32_t col = c0 + jq;              // ... inside the frame
    const uint32_t vq = (uint32_t)(validmask >> jq) & 0xfu;
    const bool vec = a.vec_ok && (col + 3 < W);
    if (col >= W) return;
    const uint32_t ncol = (W - col) < 4 ? (W - col) : 4;  // <4 only when W%4 != 0
    const uint32_t hdr = a.g.col_header_size;
    const size_t plane_px = (size_t)H * W;

    // per-column constants of the separable LUT for my 4 columns
    double cx[4], sx[4], kc[4][3];
    if (XYZM == 1 || XYZM == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t cc = (col + c < W) ? col + c : W - 1;
            const double* t = lut.col_tab + (size_t)cc * 5;
            cx[c] = t[0]; sx[c] = t[1]; kc[c][0] = t[2]; kc[c][1] = t[3]; kc[c][2] = t[4];
        }
    }

    for (uint32_t r = ty; r < H; r += RPP) {
        const size_t rowpix = (size_t)r * W + col;  // pixel index of my first column
        uint32_t doff = 0;                          // destaggered column of my first column
        bool dvec = false;
        if (a.any_destagger) {
            doff = col + (uint32_t)s_off[r];
            if (doff >= W) doff -= W;
            dvec = vec && (doff + 3 < W);
        }
        uint32_t rng[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

        if constexpr (S::is_static) {
            constexpr int CW = S::chan / 4;
            uint32_t w[4][CW];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t* px = s_tile + (((jq + c) * col_size + hdr + r * S::chan) >> 2);
#pragma unroll
                for (int k = 0; k < CW; ++k) w[c][k] = px[k];
            }
            auto do_field = [&](auto kc_) {
                constexpr int K = decltype(kc_)::value;
                const int di = a.desc_of_spec[K];
                if (di < 0) return;
                uint64_t v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[c] = ((vq >> c) & 1) ? extract_static<S, K, CW>(w[c])
                                           : trunc_elem(zero_value(a.f16_nan[di]), S::f[K].elem);
                if (K == S::range_idx) { rng[0][0] = v[0]; rng[0][1] = v[1]; rng[0][2] = v[2]; rng[0][3] = v[3]; }
                if (K == S::range2_idx) { rng[1][0] = v[0]; rng[1][1] = v[1]; rng[1][2] = v[2]; rng[1][3] = v[3]; }
                constexpr uint32_t e = S::f[K].elem;
                uint8_t* pl = (uint8_t*)a.planes[di];
                if (pl) {
                    uint8_t* d = pl + ((size_t)f * plane_px + rowpix) * e;
                    if (vec) store4(d, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) store1(d + c * e, v[c], e);
                }
                uint8_t* dp = (uint8_t*)a.destaggered[di];
                if (dp) {
                    uint8_t* drow = dp + ((size_t)f * plane_px + (size_t)r * W) * e;
                    if (dvec) store4(drow + (size_t)doff * e, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) {
                        uint32_t dc = doff + c; if (dc >= W) dc -= W;
                        store1(drow + (size_t)dc * e, v[c], e);
                    }
                }
            };
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                (do_field(std::integral_constant<int, Ks>{}), ...);
            }(std::make_integer_sequence<int, S::nf>{});
        } else {
            const uint32_t chan = a.g.channel_data_size;
            for (uint32_t i = 0; i < a.n_fields; ++i) {
                const bool want_xyz = (XYZM != 0) && ((int)i == a.xyz_field[0] || (int)i == a.xyz_field[1]);
                uint8_t* pl = (uint8_t*)a.planes[i];
                uint8_t* dp = (uint8_t*)a.destaggered[i];
                if (!pl && !dp && !want_xyz) continue;
                const uint32_t e = a.elem[i];
                uint64_t v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t bo = (jq + c) * col_size + hdr + r * chan + a.bits[i].offset;
                    v[c] = ((vq >> c) & 1)
                               ? trunc_elem(apply_bits(window_lds(s_tile, bo), a.bits[i].mask,
                                                       a.bits[i].shift), e)
                               : trunc_elem(zero_value(a.f16_nan[i]), e);
                }
                if ((int)i == a.xyz_field[0]) { rng[0][0] = v[0]; rng[0][1] = v[1]; rng[0][2] = v[2]; rng[0][3] = v[3]; }
                if ((int)i == a.xyz_field[1]) { rng[1][0] = v[0]; rng[1][1] = v[1]; rng[1][2] = v[2]; rng[1][3] = v[3]; }
                if (pl) {
                    uint8_t* d = pl + ((size_t)f * plane_px + rowpix) * e;
                    if (vec) store4(d, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) store1(d + c * e, v[c], e);
                }
                if (dp) {
                    uint8_t* drow = dp + ((size_t)f * plane_px + (size_t)r * W) * e;
                    if (dvec) store4(drow + (size_t)doff * e, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) {
                        uint32_t dc = doff + c; if (dc >= W) dc -= W;
                        store1(drow + (size_t)dc * e, v[c], e);
                    }
                }
            }
        }

        if constexpr (XYZM == 1 || XYZM == 2) {
            using XT = typename std::conditional<XYZM == 1, float, double>::type;
            const double* b = lut.beam_tab + (size_t)r * 9;  // 9 KB table, L1/L2 resident
            const double u0 = b[0], u1 = b[1], u2 = b[2], v0 = b[3], v1 = b[4], v2 = b[5],
                         w0 = b[6], w1 = b[7], w2 = b[8];
            double d[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d[c][0] = fma(cx[c], u0, fma(sx[c], v0, w0));
                d[c][1] = fma(cx[c], u1, fma(sx[c], v1, w1));
                d[c][2] = fma(cx[c], u2, fma(sx[c], v2, w2));
            }
#pragma unroll
            for (int ret = 0; ret < 2; ++ret) {
                XT* out = (XT*)a.xyz[ret];
                if (!out) continue;
                double p[4][3];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t rr = rng[ret][c];
                    const double rm = (double)rr - lut.n;
#pragma unroll
                    for (int k = 0; k < 3; ++k) p[c][k] = rr ? fma(rm, d[c][k], kc[c][k]) : 0.0;
                }
                XT* dst = out + ((size_t)f * plane_px + rowpix) * 3;
                if constexpr (XYZM == 1) {
                    if (a.vec_ok && c0 + TILE <= W) {  // full tile: every lane of the row is here
                        store_xyz4_coalesced<LPR>(s_xyz, tid, dst - (size_t)jq * 3, q, p);
                        continue;
                    }
                }
                if (vec) store_xyz4<XT>(dst, p);
                else for (uint32_t c = 0; c < ncol; ++c) store_xyz1<XT>(dst + c * 3, p[c]);
            }
        } else if constexpr (XYZM == 3) {
#pragma unroll
            for (int ret = 0; ret < 2; ++ret) {
                if (!a.xyz[ret]) continue;
                double p[4][3];
                for (uint32_t c = 0; c < ncol; ++c) {
                    if (lut.full_dtype == OUSTER_HIP_F32)
                        project_full<float>((const float*)lut.full_dir, (const float*)lut.full_ofs,
                                            rowpix + c, rng[ret][c], p[c]);
                    else
                        project_full<double>((const double*)lut.full_dir, (const double*)lut.full_ofs,
                                             rowpix + c, rng[ret][c], p[c]);
                }
                if (a.xyz_dtype == OUSTER_HIP_F32) {
                    float* dst = (float*)a.xyz[ret] + ((size_t)f * plane_px + rowpix) * 3;
                    if (vec) store_xyz4<float>(dst, p);
                    else for (uint32_t c = 0; c < ncol; ++c) store_xyz1<float>(dst + c * 3, p[c]);
                } else {
                    double* dst = (double*)a.xyz[ret] + ((size_t)f * plane_px + rowpix) * 3;
                    if (vec) store_xyz4<double>(dst, p);
                    else for (uint32_t c = 0; c < ncol; ++c) store_xyz1<double>(dst + c * 3, p[c]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// k_decode_wide: the same fused decode + destagger + cartesian with WIDE, SHORT tiles:
// a workgroup owns TW columns x TR rows (TW = 128/256/512, TW*TR*chan ~ 64 KB of LDS) instead of
// 64 columns x all rows.  Every output row segment is then TW/64 times longer (1 KB of a u32 plane,
// 256 B of a u8 plane, 3 KB of xyz for TW = 256), which is what HBM wants: with 64-column tiles the
// achieved write rate swings between 3.4 and 4.9 TB/s with the physical placement of the output
// planes (tools/storebench.hip), with 256-column tiles it stays at 5.1-6.1 TB/s.
// The price is on the (8x smaller) input side: a column is no longer read whole but in TR-row pieces
// (TR*chan bytes, 256 B for dual-LB at TW = 256), staged with dword loads -- one wave per column
// piece -- into per-column LDS slots padded by one dword (bank spread for the 4-columns-per-lane
// reads).  The column tiles of one row chunk are consecutive blocks of one XCD, so neighbouring
// workgroups write whole rows together.  Column headers are read directly by the first row chunk.
// Used when W % TW == 0 and the batch is large enough; everything else runs k_decode.
// ------------------------------------------------------------------------------------
template <class S, int TW, int XYZM>
__global__ __launch_bounds__(256) void k_decode_wide(DecodeArgs a) {
    constexpr int NT = 256;
    constexpr int QPR = TW / 4;                    // quads (lanes) per tile row
    constexpr int LPR = QPR < 64 ? QPR : 64;       // lanes of one wave in a row segment
    constexpr int RPP = NT / QPR > 0 ? NT / QPR : 1;  // rows per pass (TW <= 1024)
    static_assert(TW % 64 == 0 && QPR <= NT, "tile width");
    extern __shared__ __align__(16) uint32_t smem[];

    const uint32_t TR = a.rows_per_tile, nch = a.row_chunks;
    const uint32_t tpf = a.tiles_per_frame * nch;  // blocks per frame
    uint32_t f, sub;
    if (a.xcd_map) {
        const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        f = (i / tpf) * 8u + xcd;
        sub = i % tpf;
        if (f >= a.n_frames) return;
    } else {
        f = blockIdx.x / tpf;
        sub = blockIdx.x - f * tpf;
    }
    // the column tiles of one row chunk are neighbouring blocks of an XCD: together they write whole
    // 8 KB rows at the same time (dbg 3, experiment: the row chunks of a column tile are neighbours
    // instead, which shares input cache lines but measured 6 % slower)
    const uint32_t tile = a.dbg == 3 ? sub / nch : sub % a.tiles_per_frame;
    const uint32_t rc = a.dbg == 3 ? sub - tile * nch : sub / a.tiles_per_frame;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, H = a.g.pixels_per_column;
    const uint32_t cpp = a.g.columns_per_packet, col_size = a.g.col_size;
    const uint32_t chan = S::is_static ? S::chan : a.g.channel_data_size;
    const uint32_t hdr = a.g.col_header_size;
    const uint32_t c0 = tile * TW, r0 = rc * TR;
    const uint32_t nrows = min(TR, H - r0);
    const uint32_t PD = (TR * chan) >> 2;          // dwords of one column piece
    const uint32_t slot = a.lds_col_slot >> 2;     // LDS dwords per column (PD + pad)

    uint32_t* s_tile = smem;                                  // [TW][slot]
    uint32_t* s_colofs = smem + TW * slot + 4;                // [TW] byte offset of the column in the frame buffer
    int32_t* s_off = (int32_t*)(s_colofs + TW);               // [TR] destagger offsets of my rows
    float4* s_xyz = (float4*)(s_off + ((TR + 3) & ~3u));      // [4 waves][192]

    // ---- phase 0: where do my columns live?
    const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    for (uint32_t j = tid; j < (uint32_t)TW; j += NT) {
        const uint32_t c = c0 + j;
        const int32_t raw = (c < W) ? a.map[(size_t)f * W + c] : -1;
        const int32_t src = (raw >= 0 && ((uint32_t)raw >> 20) == a.epoch) ? (raw & 0xfffff) : -1;  // stale epoch = absent
        uint32_t ofs = 0xffffffffu;
        if (src >= 0) {
            const uint32_t p = (uint32_t)src / cpp, ic = (uint32_t)src - p * cpp;
            ofs = p * (uint32_t)a.packet_stride + a.g.packet_header_size + ic * col_size;
        }
        s_colofs[j] = ofs;
    }
    if (a.any_destagger)
        for (uint32_t r = tid; r < nrows; r += NT) s_off[r] = a.dst_offsets[r0 + r];
    const LutDev lut = (XYZM != 0) ? a.luts[f % a.n_luts] : LutDev{};
    __syncthreads();

    // ---- phase 1: stage my TR-row piece of every column, dword granular (packets are 4 B granular)
    {
        // 16 B aligned loads that keep each column piece's own 16 B phase: chunk ch of column j is
        // the aligned 16 B at (piece start - delta) + 16*ch; its dwords land at piece-relative
        // positions 4*ch - delta/4 + {0..3}, those outside [0, piece) are dropped.  Thread t owns
        // chunks t, t + NT, ...; a wave reads 1 KB of (almost) consecutive bytes per instruction.
        const uint32_t rowofs = hdr + r0 * chan;
        const uint32_t piece = (nrows * chan) >> 2;        // dwords of a column piece in this chunk
        const uint32_t NCH = (piece * 4u + 15u + 15u) >> 4;  // aligned 16 B chunks that can touch it
        const uint32_t total = TW * NCH;
        const uint8_t* fend = fbase + (size_t)a.slots_per_frame * a.packet_stride;
        uint32_t j = tid / NCH, ch = tid - j * NCH;
        const uint32_t dj = NT / NCH, dc = NT - dj * NCH;
        constexpr int DEPTH = 18;  // 256 columns x 17 chunks = 17 per thread for 256 B pieces
        for (uint32_t base = 0; base < total; base += NT * DEPTH) {
            u32x4 t[DEPTH];
            int32_t p0[DEPTH];   // piece-relative dword index of t[k].x, or a value that drops all four
            uint32_t sj[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                p0[k] = -1000000;
                sj[k] = 0;
                t[k] = u32x4{0, 0, 0, 0};
                if (base + k * NT + tid < total) {
                    const uint32_t ofs = s_colofs[j];
                    if (ofs != 0xffffffffu && a.dbg != 2) {
                        const uint8_t* src = fbase + ofs + rowofs;
                        const uint32_t delta = (uint32_t)((uintptr_t)src & 15u);
                        const u32x4* q = (const u32x4*)(src - delta) + ch;
                        if (ch * 16u < delta + piece * 4u) {
                            if ((const uint8_t*)(q + 1) <= fend) t[k] = *q;
                            else {  // last chunk of the frame buffer: stay inside it
                                const uint32_t* qd = (const uint32_t*)q;
                                for (int w = 0; w < 4; ++w)
                                    if ((const uint8_t*)(qd + w + 1) <= fend) t[k][w] = qd[w];
                            }
                            p0[k] = (int32_t)(ch * 4u) - (int32_t)(delta >> 2);
                            sj[k] = j * slot;
                        }
                    }
                }
                j += dj; ch += dc;
                if (ch >= NCH) { ch -= NCH; ++j; }
            }
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int32_t pp = p0[k] + w;
                    if (pp >= 0 && pp < (int32_t)piece) s_tile[sj[k] + (uint32_t)pp] = t[k][w];
                }
            }
        }
        // missing columns and the rows past H keep whatever the LDS held: nothing reads them (vq / nrows)
        if (tid < 4) s_tile[TW * slot + tid] = 0;  // slack read by 64-bit windows
    }
    __syncthreads();

    if (a.dbg == 1) { if (s_tile[tid] == 0x12345678u) a.map[0] = 0; return; }
    // ---- phase 2a: column headers, by the first row chunk, straight from the packets
    if (rc == 0) {
        for (uint32_t j = tid; j < (uint32_t)TW && c0 + j < W; j += NT) {
            const uint32_t c = c0 + j, ofs = s_colofs[j];
            const bool v = ofs != 0xffffffffu;
            uint64_t w_ts = 0, w_st = 0;
            if (v) {
                const uint8_t* colp = fbase + ofs;
                if (a.timestamp) w_ts = window_global_masked(colp + a.g.col_timestamp.offset, a.g.col_timestamp.mask);
                if (a.status) w_st = window_global_masked(colp + a.g.col_status.offset, a.g.col_status.mask);
            }
            if (a.timestamp)
                a.timestamp[(size_t)f * W + c] = v ? apply_bits(w_ts, a.g.col_timestamp.mask, a.g.col_timestamp.shift) : 0ull;
            if (a.measurement_id) a.measurement_id[(size_t)f * W + c] = v ? (uint16_t)c : (uint16_t)0;
            if (a.status)
                a.status[(size_t)f * W + c] = v ? (uint32_t)apply_bits(w_st, a.g.col_status.mask, a.g.col_status.shift) : 0u;
        }
    }

    // ---- phase 2b: pixels.  lane = (row within pass, quad of 4 consecutive columns)
    const uint32_t q = tid % QPR, ty = tid / QPR;
    const uint32_t jq = q * 4, col = c0 + jq;
    if (col >= W) return;
    uint32_t vq = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) vq |= (jq + c < (uint32_t)TW && s_colofs[jq + c] != 0xffffffffu) ? (1u << c) : 0u;
    const bool vec = a.vec_ok && (col + 3 < W);
    const uint32_t ncol = (W - col) < 4 ? (W - col) : 4;
    const size_t plane_px = (size_t)H * W;
    const uint32_t ql = q % LPR;                       // lane position inside its wave's row segment
    const uint32_t seg0 = c0 + (q - ql) * 4;           // first column of that segment

    double cx[4], sx[4], kc[4][3];
    if (XYZM == 1 || XYZM == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t cc = (col + c < W) ? col + c : W - 1;
            const double* t = lut.col_tab + (size_t)cc * 5;
            cx[c] = t[0]; sx[c] = t[1]; kc[c][0] = t[2]; kc[c][1] = t[3]; kc[c][2] = t[4];
        }
    }

    for (uint32_t rrel = ty; rrel < nrows; rrel += RPP) {
        const uint32_t r = r0 + rrel;
        const size_t rowpix = (size_t)r * W + col;
        uint32_t doff = 0;
        bool dvec = false;
        if (a.any_destagger) {
            doff = col + (uint32_t)s_off[rrel];
            if (doff >= W) doff -= W;
            dvec = vec && (doff + 3 < W);
        }
        uint32_t rng[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

        if constexpr (S::is_static) {
            constexpr int CW = S::chan / 4;
            uint32_t w[4][CW];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t* px = s_tile + (jq + c) * slot + rrel * CW;
#pragma unroll
                for (int k = 0; k < CW; ++k) w[c][k] = px[k];
            }
            auto do_field = [&](auto kc_) {
                constexpr int K = decltype(kc_)::value;
                const int di = a.desc_of_spec[K];
                if (di < 0) return;
                uint64_t v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[c] = ((vq >> c) & 1) ? extract_static<S, K, CW>(w[c])
                                           : trunc_elem(zero_value(a.f16_nan[di]), S::f[K].elem);
                if (K == S::range_idx) { rng[0][0] = v[0]; rng[0][1] = v[1]; rng[0][2] = v[2]; rng[0][3] = v[3]; }
                if (K == S::range2_idx) { rng[1][0] = v[0]; rng[1][1] = v[1]; rng[1][2] = v[2]; rng[1][3] = v[3]; }
                constexpr uint32_t e = S::f[K].elem;
                uint8_t* pl = (uint8_t*)a.planes[di];
                if (pl) {
                    uint8_t* d = pl + ((size_t)f * plane_px + rowpix) * e;
                    if (vec) store4(d, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) store1(d + c * e, v[c], e);
                }
                uint8_t* dp = (uint8_t*)a.destaggered[di];
                if (dp) {
                    uint8_t* drow = dp + ((size_t)f * plane_px + (size_t)r * W) * e;
                    if (dvec) store4(drow + (size_t)doff * e, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) {
                        uint32_t dc = doff + c; if (dc >= W) dc -= W;
                        store1(drow + (size_t)dc * e, v[c], e);
                    }
                }
            };
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                (do_field(std::integral_constant<int, Ks>{}), ...);
            }(std::make_integer_sequence<int, S::nf>{});
        } else {
            for (uint32_t i = 0; i < a.n_fields; ++i) {
                const bool want_xyz = (XYZM != 0) && ((int)i == a.xyz_field[0] || (int)i == a.xyz_field[1]);
                uint8_t* pl = (uint8_t*)a.planes[i];
                uint8_t* dp = (uint8_t*)a.destaggered[i];
                if (!pl && !dp && !want_xyz) continue;
                const uint32_t e = a.elem[i];
                uint64_t v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t bo = ((jq + c) * slot << 2) + rrel * chan + a.bits[i].offset;
                    v[c] = ((vq >> c) & 1)
                               ? trunc_elem(apply_bits(window_lds(s_tile, bo), a.bits[i].mask, a.bits[i].shift), e)
                               : trunc_elem(zero_value(a.f16_nan[i]), e);
                }
                if ((int)i == a.xyz_field[0]) { rng[0][0] = v[0]; rng[0][1] = v[1]; rng[0][2] = v[2]; rng[0][3] = v[3]; }
                if ((int)i == a.xyz_field[1]) { rng[1][0] = v[0]; rng[1][1] = v[1]; rng[1][2] = v[2]; rng[1][3] = v[3]; }
                if (pl) {
                    uint8_t* d = pl + ((size_t)f * plane_px + rowpix) * e;
                    if (vec) store4(d, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) store1(d + c * e, v[c], e);
                }
                if (dp) {
                    uint8_t* drow = dp + ((size_t)f * plane_px + (size_t)r * W) * e;
                    if (dvec) store4(drow + (size_t)doff * e, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) {
                        uint32_t dc = doff + c; if (dc >= W) dc -= W;
                        store1(drow + (size_t)dc * e, v[c], e);
                    }
                }
            }
        }

        if constexpr (XYZM == 1 || XYZM == 2) {
            using XT = typename std::conditional<XYZM == 1, float, double>::type;
            const double* b = lut.beam_tab + (size_t)r * 9;
            const double u0 = b[0], u1 = b[1], u2 = b[2], v0 = b[3], v1 = b[4], v2 = b[5],
                         w0 = b[6], w1 = b[7], w2 = b[8];
            double d[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d[c][0] = fma(cx[c], u0, fma(sx[c], v0, w0));
                d[c][1] = fma(cx[c], u1, fma(sx[c], v1, w1));
                d[c][2] = fma(cx[c], u2, fma(sx[c], v2, w2));
            }
#pragma unroll
            for (int ret = 0; ret < 2; ++ret) {
                XT* out = (XT*)a.xyz[ret];
                if (!out) continue;
                double p[4][3];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t rr = rng[ret][c];
                    const double rm = (double)rr - lut.n;
#pragma unroll
                    for (int k = 0; k < 3; ++k) p[c][k] = rr ? fma(rm, d[c][k], kc[c][k]) : 0.0;
                }
                XT* dst = out + ((size_t)f * plane_px + rowpix) * 3;
                if constexpr (XYZM == 1) {
                    if (a.vec_ok && seg0 + 4 * LPR <= W) {  // my wave's whole row segment exists
                        store_xyz4_coalesced<LPR>(s_xyz, tid, dst - (size_t)(4 * ql) * 3, ql, p);
                        continue;
                    }
                }
                if (vec) store_xyz4<XT>(dst, p);
                else for (uint32_t c = 0; c < ncol; ++c) store_xyz1<XT>(dst + c * 3, p[c]);
            }
        } else if constexpr (XYZM == 3) {
#pragma unroll
            for (int ret = 0; ret < 2; ++ret) {
                if (!a.xyz[ret]) continue;
                double p[4][3];
                for (uint32_t c = 0; c < ncol; ++c) {
                    if (lut.full_dtype == OUSTER_HIP_F32)
                        project_full<float>((const float*)lut.full_dir, (const float*)lut.full_ofs,
                                            rowpix + c, rng[ret][c], p[c]);
                    else
                        project_full<double>((const double*)lut.full_dir, (const double*)lut.full_ofs,
                                             rowpix + c, rng[ret][c], p[c]);
                }
                if (a.xyz_dtype == OUSTER_HIP_F32) {
                    float* dst = (float*)a.xyz[ret] + ((size_t)f * plane_px + rowpix) * 3;
                    if (vec) store_xyz4<float>(dst, p);
                    else for (uint32_t c = 0; c < ncol; ++c) store_xyz1<float>(dst + c * 3, p[c]);
                } else {
                    double* dst = (double*)a.xyz[ret] + ((size_t)f * plane_px + rowpix) * 3;
                    if (vec) store_xyz4<double>(dst, p);
                    else for (uint32_t c = 0; c < ncol; ++c) store_xyz1<double>(dst + c * 3, p[c]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// k_destagger: dst[img][u][(v + off[u]) % w] = src[img][u][v], element = elem bytes.
// One workgroup per (row, image); lanes walk the DESTINATION row in 16 B chunks so every
// store is an aligned, coalesced 16 B vector; the source bytes for a chunk start at an
// arbitrary byte offset of the source row and are fetched with three aligned dword loads
// per output dword pair (the row is L1/L2 resident after first touch).
//   offset arithmetic: destagger_into, impl/lidar_frame_impl.h:753-759
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_destagger(DestaggerArgs a) {
    const uint32_t u = blockIdx.x, img = blockIdx.y;
    const size_t row_bytes = (size_t)a.w * a.elem;
    const uint8_t* srow = (const uint8_t*)a.src + ((size_t)img * a.h + u) * row_bytes;
    uint8_t* drow = (uint8_t*)a.dst + ((size_t)img * a.h + u) * row_bytes;
    const size_t shift_bytes = (size_t)a.offsets[u] * a.elem;  // dst byte b <- src byte (b - shift) mod row
    const bool fast = ((row_bytes & 15) == 0) && ((((uintptr_t)a.src | (uintptr_t)a.dst) & 15) == 0);
    if (fast) {
        const uint32_t nchunk = (uint32_t)(row_bytes >> 4);
        for (uint32_t i = threadIdx.x; i < nchunk; i += blockDim.x) {
            const size_t db = (size_t)i << 4;
            size_t sb = db + row_bytes - shift_bytes;
            if (sb >= row_bytes) sb -= row_bytes;
            uint32_t o[4];
            if (sb + 16 <= row_bytes) {
                const uint32_t sh = (uint32_t)(sb & 3) * 8;
                const uint32_t* q = (const uint32_t*)(srow + (sb & ~(size_t)3));
                if (sh == 0) {
                    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
                } else {
                    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                    o[0] = (d0 >> sh) | (d1 << (32 - sh));
                    o[1] = (d1 >> sh) | (d2 << (32 - sh));
                    o[2] = (d2 >> sh) | (d3 << (32 - sh));
                    o[3] = (d3 >> sh) | (d4 << (32 - sh));
                }
            } else {  // the chunk straddles the wrap point of the source row
                uint8_t b[16];
                for (int k = 0; k < 16; ++k) {
                    size_t s = sb + k;
                    if (s >= row_bytes) s -= row_bytes;
                    b[k] = srow[s];
                }
                for (int k = 0; k < 4; ++k)
                    o[k] = b[4 * k] | (b[4 * k + 1] << 8) | (b[4 * k + 2] << 16) |
                           ((uint32_t)b[4 * k + 3] << 24);
            }
            *(uint4*)(drow + db) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (size_t b = threadIdx.x; b < row_bytes; b += blockDim.x) {
            size_t s = b + row_bytes - shift_bytes;
            if (s >= row_bytes) s -= row_bytes;
            drow[b] = srow[s];
        }
    }
}

// ------------------------------------------------------------------------------------
// k_cartesian: standalone range image -> xyz, 4 consecutive pixels per lane
//   (cartesianT<T>, impl/cartesian.h:36-66)
// ------------------------------------------------------------------------------------
template <int MODE /*1 sep->f32, 2 sep->f64, 3 full*/>
__global__ __launch_bounds__(256) void k_cartesian(CartesianArgs a) {
    const uint32_t W = a.w, H = a.h;
    const size_t npix = (size_t)W * H;
    const size_t quads = (npix + 3) / 4;
    const LutDev lut = a.lut;
    for (size_t qi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qi < quads * a.n_images;
         qi += (size_t)gridDim.x * blockDim.x) {
        const size_t img = qi / quads, q = qi - img * quads;
        const size_t pix = q * 4;
        const uint32_t n = (npix - pix) < 4 ? (uint32_t)(npix - pix) : 4;
        const uint32_t* rp = a.range + img * npix + pix;
        uint32_t r[4] = {0, 0, 0, 0};
        const bool vec = (n == 4) && a.vec_ok;
        if (vec) { uint4 t = *(const uint4*)rp; r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
        else for (uint32_t c = 0; c < n; ++c) r[c] = rp[c];
        double p[4][3];
        if constexpr (MODE == 1 || MODE == 2) {
            for (uint32_t c = 0; c < n; ++c) {
                const size_t i = pix + c;
                const uint32_t row = (uint32_t)(i / W), cc = (uint32_t)(i - (size_t)row * W);
                const double* b = lut.beam_tab + (size_t)row * 9;
                const double* t = lut.col_tab + (size_t)cc * 5;
                const double cxx = t[0], sxx = t[1];
                const double rm = (double)r[c] - lut.n;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double d = fma(cxx, b[k], fma(sxx, b[3 + k], b[6 + k]));
                    p[c][k] = r[c] ? fma(rm, d, t[2 + k]) : 0.0;
                }
            }
        } else {
            for (uint32_t c = 0; c < n; ++c) {
                if (lut.full_dtype == OUSTER_HIP_F32)
                    project_full<float>((const float*)lut.full_dir, (const float*)lut.full_ofs,
                                        pix + c, r[c], p[c]);
                else
                    project_full<double>((const double*)lut.full_dir, (const double*)lut.full_ofs,
                                         pix + c, r[c], p[c]);
            }
        }
        if (a.xyz_dtype == OUSTER_HIP_F32) {
            float* dst = (float*)a.xyz + (img * npix + pix) * 3;
            if (vec) store_xyz4<float>(dst, p);
            else for (uint32_t c = 0; c < n; ++c) store_xyz1<float>(dst + c * 3, p[c]);
        } else {
            double* dst = (double*)a.xyz + (img * npix + pix) * 3;
            if (vec) store_xyz4<double>(dst, p);
            else for (uint32_t c = 0; c < n; ++c) store_xyz1<double>(dst + c * 3, p[c]);
        }
    }
}

// ------------------------------------------------------------------------------------
// k_dewarp: p' = R_col * p + t_col for every point (pose_util.h:38-56).  One thread per point;
// a wave reads 64 consecutive points = 768 contiguous bytes (f32) of one row, the 64 column
// poses come from the L2-resident pose table.  HBM bound: 2 x 3 x sizeof(T) B/point.
// ------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_dewarp(DewarpArgs a) {
    const size_t npix = (size_t)a.w * a.h, total = npix * a.n_images;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t img = i / npix, pix = i - img * npix;
        const uint32_t col = (uint32_t)(pix % a.w);
        const double* m = a.poses + (img * a.w + col) * 16;
        const T* p = (const T*)a.points + i * 3;
        const T x = p[0], y = p[1], z = p[2];
        T* o = (T*)a.out + i * 3;
        // rotation * s + translation in T, row by row, like the reference's Eigen expression
        o[0] = (T)m[0] * x + (T)m[1] * y + (T)m[2] * z + (T)m[3];
        o[1] = (T)m[4] * x + (T)m[5] * y + (T)m[6] * z + (T)m[7];
        o[2] = (T)m[8] * x + (T)m[9] * y + (T)m[10] * z + (T)m[11];
    }
}

// ------------------------------------------------------------------------------------
// k_cartesian_tiled: the fast standalone form for W % 4 == 0 and 16 B aligned buffers.
// Same lane mapping as k_decode's compute phase: a workgroup owns 64 columns x RC rows of one
// image, lane = (row within pass, quad of 4 consecutive columns).  Per-column constants live in
// registers for the whole row loop, the row's 9 beam constants come from the L1-resident table,
// the range quad is one 16 B load and f32 XYZ goes out through the wave-private LDS transpose
// (256 contiguous bytes per row per store instruction).
//   MODE 1: separable tables -> f32, 2: separable -> f64, 3: full LUT (runtime dtypes)
// ------------------------------------------------------------------------------------
// full-LUT projection of a lane's 4 pixels from registers; LT = LUT element type
template <class LT>
__device__ __forceinline__ void project_full4(const LT (&dir)[12], const LT (&ofs)[12],
                                              const uint32_t (&rng)[4], double (&p)[4][3]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const LT rr = (LT)rng[c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            LT m = rr * dir[3 * c + k];
            asm volatile("" : "+v"(m));  // no fma contraction (cartesianT host build)
            p[c][k] = rng[c] ? (double)(LT)(m + ofs[3 * c + k]) : 0.0;
        }
    }
}

template <int MODE, int TILE>
__global__ __launch_bounds__(256) void k_cartesian_tiled(CartesianArgs a) {
    constexpr int LPR = TILE / 4, RPP = 256 / LPR;
    __shared__ float4 s_xyz[RPP * 6 * LPR];  // one 6-chunk scratch per lane-row
    const uint32_t W = a.w, H = a.h;
    const uint32_t tiles = (W + TILE - 1) / TILE;
    const uint32_t tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
    const uint32_t img = blockIdx.y, tid = threadIdx.x;
    const uint32_t q = tid % LPR, ty = tid / LPR;
    const uint32_t c0 = tile * TILE, col = c0 + 4 * q;
    const uint32_t r_begin = chunk * a.rows_per_block;
    const uint32_t r_end = min(H, r_begin + a.rows_per_block);
    const bool full_tile = c0 + TILE <= W;
    const bool live = col < W;  // W % 4 == 0: a quad is entirely inside or outside
    const LutDev lut = a.lut;
    const size_t npix = (size_t)W * H;
    float4* sc = s_xyz + ty * (6 * LPR);

    double cx[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0}, kc[4][3] = {};
    if constexpr (MODE == 1 || MODE == 2) {
        if (live) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double* t = lut.col_tab + (size_t)(col + c) * 5;
                cx[c] = t[0]; sx[c] = t[1]; kc[c][0] = t[2]; kc[c][1] = t[3]; kc[c][2] = t[4];
            }
        }
    }
    for (uint32_t r = r_begin + ty; r < r_end; r += RPP) {
        const size_t rowpix = (size_t)r * W + col;
        const size_t rowpix0 = (size_t)r * W + c0;
        uint32_t rng[4] = {0, 0, 0, 0};
        if (live) {
            const uint4 t = *(const uint4*)(a.range + (size_t)img * npix + rowpix);
            rng[0] = t.x; rng[1] = t.y; rng[2] = t.z; rng[3] = t.w;
        }
        double p[4][3];
        if constexpr (MODE == 1 || MODE == 2) {
            const double* b = lut.beam_tab + (size_t)r * 9;
            const double u0 = b[0], u1 = b[1], u2 = b[2], v0 = b[3], v1 = b[4], v2 = b[5],
                         w0 = b[6], w1 = b[7], w2 = b[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double d0 = fma(cx[c], u0, fma(sx[c], v0, w0));
                const double d1 = fma(cx[c], u1, fma(sx[c], v1, w1));
                const double d2 = fma(cx[c], u2, fma(sx[c], v2, w2));
                const double rm = (double)rng[c] - lut.n;
                p[c][0] = rng[c] ? fma(rm, d0, kc[c][0]) : 0.0;
                p[c][1] = rng[c] ? fma(rm, d1, kc[c][1]) : 0.0;
                p[c][2] = rng[c] ? fma(rm, d2, kc[c][2]) : 0.0;
            }
        } else {
            // the LUT rows are streamed like the output: coalesced row segments, transposed
            // to "lane owns 4 pixels" through the scratch
            if (lut.full_dtype == OUSTER_HIP_F32) {
                union { float4 f4[3]; float t[12]; } d, o;
                if (full_tile) {
                    load_quad_coalesced<3, LPR>(sc, (const float4*)((const float*)lut.full_dir + rowpix0 * 3), q, d.f4);
                    load_quad_coalesced<3, LPR>(sc, (const float4*)((const float*)lut.full_ofs + rowpix0 * 3), q, o.f4);
                } else if (live) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        d.f4[k] = ((const float4*)((const float*)lut.full_dir + rowpix * 3))[k];
                        o.f4[k] = ((const float4*)((const float*)lut.full_ofs + rowpix * 3))[k];
                    }
                }
                project_full4<float>(d.t, o.t, rng, p);
            } else {
                union { float4 f4[6]; double t[12]; } d, o;
                if (full_tile) {
                    load_quad_coalesced<6, LPR>(sc, (const float4*)((const double*)lut.full_dir + rowpix0 * 3), q, d.f4);
                    load_quad_coalesced<6, LPR>(sc, (const float4*)((const double*)lut.full_ofs + rowpix0 * 3), q, o.f4);
                } else if (live) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        d.f4[k] = ((const float4*)((const double*)lut.full_dir + rowpix * 3))[k];
                        o.f4[k] = ((const float4*)((const double*)lut.full_ofs + rowpix * 3))[k];
                    }
                }
                project_full4<double>(d.t, o.t, rng, p);
            }
        }
        if (a.xyz_dtype == OUSTER_HIP_F32) {
            float* dst = (float*)a.xyz + ((size_t)img * npix + rowpix) * 3;
            if (full_tile) {
                union { float4 f4[3]; float t[12]; } o;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) o.t[3 * c + k] = (float)p[c][k];
                store_quad_coalesced<3, LPR>(sc, (float4*)(dst - (size_t)(4 * q) * 3), q, o.f4);
            } else if (live) store_xyz4<float>(dst, p);
        } else {
            double* dst = (double*)a.xyz + ((size_t)img * npix + rowpix) * 3;
            if (full_tile) {
                union { float4 f4[6]; double t[12]; } o;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) o.t[3 * c + k] = p[c][k];
                store_quad_coalesced<6, LPR>(sc, (float4*)(dst - (size_t)(4 * q) * 3), q, o.f4);
            } else if (live) store_xyz4<double>(dst, p);
        }
    }
}

// ------------------------------------------------------------------------------------
// k_dewarp_tiled: p' = R_col * p + t_col (pose_util.h:38-56) with the k_decode lane mapping:
// each lane keeps the 3x4 poses of its 4 columns in registers for the whole row loop.
// ------------------------------------------------------------------------------------
template <class T, int TILE>
__global__ __launch_bounds__(256) void k_dewarp_tiled(DewarpArgs a) {
    constexpr int LPR = TILE / 4, RPP = 256 / LPR;
    constexpr int NV = 3 * sizeof(T) / 4;  // 16 B chunks per lane quad: 3 (f32) or 6 (f64)
    __shared__ float4 s_xyz[RPP * NV * LPR];
    const uint32_t W = a.w, H = a.h;
    const uint32_t tiles = (W + TILE - 1) / TILE;
    const uint32_t tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
    const uint32_t img = blockIdx.y, tid = threadIdx.x;
    const uint32_t q = tid % LPR, ty = tid / LPR;
    const uint32_t c0 = tile * TILE, col = c0 + 4 * q;
    const bool full_tile = c0 + TILE <= W;
    const bool live = col < W;
    const uint32_t r_begin = chunk * a.rows_per_block;
    const uint32_t r_end = min(H, r_begin + a.rows_per_block);
    const size_t npix = (size_t)W * H;
    float4* sc = s_xyz + ty * (NV * LPR);
    T m[4][12];
    if (live) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double* pm = a.poses + ((size_t)img * W + col + c) * 16;
#pragma unroll
            for (int k = 0; k < 12; ++k) m[c][k] = (T)pm[k];
        }
    }
    for (uint32_t r = r_begin + ty; r < r_end; r += RPP) {
        const size_t i = ((size_t)img * npix + (size_t)r * W + col) * 3;
        union { float4 f4[NV]; T t[12]; } v, o;
        if (full_tile) {
            // coalesced row-segment read (LPR x 16 B contiguous per instruction), transposed
            // to "lane owns 4 points" through the lane-row's private scratch
            load_quad_coalesced<NV, LPR>(sc, (const float4*)((const T*)a.points + i - (size_t)(4 * q) * 3), q, v.f4);
        } else if (live) {
            const float4* src = (const float4*)((const T*)a.points + i);
#pragma unroll
            for (int k = 0; k < NV; ++k) v.f4[k] = src[k];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const T x = v.t[3 * c], y = v.t[3 * c + 1], z = v.t[3 * c + 2];
            o.t[3 * c + 0] = m[c][0] * x + m[c][1] * y + m[c][2] * z + m[c][3];
            o.t[3 * c + 1] = m[c][4] * x + m[c][5] * y + m[c][6] * z + m[c][7];
            o.t[3 * c + 2] = m[c][8] * x + m[c][9] * y + m[c][10] * z + m[c][11];
        }
        if (full_tile) {
            store_quad_coalesced<NV, LPR>(sc, (float4*)((T*)a.out + i - (size_t)(4 * q) * 3), q, o.f4);
        } else if (live) {
            float4* dst = (float4*)((T*)a.out + i);
#pragma unroll
            for (int k = 0; k < NV; ++k) dst[k] = o.f4[k];
        }
    }
}

// ------------------------------------------------------------------------------------
// Range-gated, compacting frame dewarp: dewarp(LidarFrame|FrameSet, XYZLut, min_range, max_range)
// (impl/dewarp_impl.h:23-115).  Output order is the reference's: frame, then column
// first_valid..last_valid with status != 0, then row; a point is kept when min_r <= r <= max_r.
//   k_dwf_count       kept points per (frame, column) from the range plane (ignores status)
//   k_dwf_scan        per frame: first/last valid column (status & 1, lidar_frame.cpp:907-925),
//                     mask, exclusive scan over the columns; frame total
//   k_dwf_frame_scan  exclusive scan of the frame totals
//   k_dwf_emit        stage a 64-row x 64-column range tile in LDS, wave = column, lane = row:
//                     ballot-rank the kept rows, project (f64 tables or the full LUT), apply the
//                     column pose in T, and write the compacted run through a wave-private LDS
//                     buffer so the global stores are contiguous dwords.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dwf_count(DewarpFramesArgs a) {
    constexpr int TILE = 64, LPR = 16, RPP = 16;
    __shared__ uint32_t s_cnt[TILE];
    const uint32_t W = a.w, H = a.h, f = blockIdx.y, tid = threadIdx.x;
    const uint32_t q = tid % LPR, ty = tid / LPR;
    const uint32_t c0 = blockIdx.x * TILE, col = c0 + 4 * q;
    if (tid < TILE) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t* rp = a.range + (size_t)f * W * H;
    const bool vec = (W % 4 == 0) && ((((uintptr_t)a.range) & 15) == 0);
    uint32_t cnt[4] = {0, 0, 0, 0};
    if (col < W) {
        for (uint32_t r = ty; r < H; r += RPP) {
            uint32_t v[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if (vec) {
                const uint4 t = *(const uint4*)(rp + (size_t)r * W + col);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                for (uint32_t c = 0; c < 4 && col + c < W; ++c) v[c] = rp[(size_t)r * W + col + c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                cnt[c] += (col + c < W && v[c] >= a.min_r && v[c] <= a.max_r) ? 1u : 0u;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cnt[c]) atomicAdd(&s_cnt[4 * q + c], cnt[c]);
    }
    __syncthreads();
    if (tid < TILE && c0 + tid < W) a.col_off[(size_t)f * (W + 1) + c0 + tid] = s_cnt[tid];
}

// block-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix,
// *total = block sum
__device__ __forceinline__ uint32_t block_exscan_256(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wave; ++k) base += s_wave[k];
    *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_dwf_scan(DewarpFramesArgs a) {
    __shared__ int s_lo, s_hi;
    __shared__ uint32_t s_wave[4];
    const uint32_t W = a.w, f = blockIdx.x, tid = threadIdx.x;
    const uint32_t* st = a.status + (size_t)f * W;
    uint32_t* off = a.col_off + (size_t)f * (W + 1);
    if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; }
    __syncthreads();
    int lo = 0x7fffffff, hi = -1;
    for (uint32_t x = tid; x < W; x += 256)
        if (st[x] & 1u) { lo = min(lo, (int)x); hi = max(hi, (int)x); }
    if (hi >= 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    lo = s_lo; hi = s_hi;
    // contiguous segment per thread
    const uint32_t seg = (W + 255) / 256;
    const uint32_t x0 = tid * seg, x1 = min(W, x0 + seg);
    uint32_t sum = 0;
    for (uint32_t x = x0; x < x1; ++x) {
        const bool keep = (int)x >= lo && (int)x <= hi && st[x] != 0;
        sum += keep ? off[x] : 0u;
    }
    uint32_t total;
    uint32_t run = block_exscan_256(sum, s_wave, &total);
    for (uint32_t x = x0; x < x1; ++x) {
        const bool keep = (int)x >= lo && (int)x <= hi && st[x] != 0;
        const uint32_t c = keep ? off[x] : 0u;
        off[x] = run;
        run += c;
    }
    if (tid == 0) {
        off[W] = total;
        a.frame_off[f + 1] = total;
    }
}

__global__ __launch_bounds__(256) void k_dwf_frame_scan(DewarpFramesArgs a) {
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < a.n_frames; base += 256) {
        const uint32_t f = base + tid;
        const uint32_t v = f < a.n_frames ? (uint32_t)a.frame_off[f + 1] : 0u;
        uint32_t total;
        const uint32_t ex = block_exscan_256(v, s_wave, &total);
        if (f < a.n_frames) a.frame_off[f + 1] = carry + ex + v;
        carry += total;
    }
    if (tid == 0) a.frame_off[0] = 0;
}

// wave-uniform broadcast of a register of lane `src` (v_readlane_b32 with an SGPR lane select)
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}
__device__ __forceinline__ float bcast(float v, uint32_t src) {
    return __uint_as_float(bcast_u32(__float_as_uint(v), src));
}
__device__ __forceinline__ double bcast(double v, uint32_t src) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint64_t r = (uint64_t)bcast_u32((uint32_t)u, src) | ((uint64_t)bcast_u32((uint32_t)(u >> 32), src) << 32);
    return __longlong_as_double((long long)r);
}
template <class T> struct __attribute__((packed, aligned(4))) Pt3 { T x, y, z; };

template <class T, bool SEP, int TILE>
__global__ __launch_bounds__(256) void k_dwf_emit(DewarpFramesArgs a) {
    // tile = TILE columns, rows in chunks of ROWS = 4096 / TILE (64 x 64 or 32 x 128).  A wave owns
    // CPW columns; lane l < CPW keeps the metadata of column l (output base, pose cast to T, table
    // row, timestamp) in registers and each column iteration broadcasts it with v_readlane -- no
    // dependent global loads and no LDS traffic in the column loop apart from the transposed
    // range read.  Lane = row (NR rows per lane): the kept rows are ranked with ballots and every
    // lane stores its own 12 / 24 B point, so one store instruction writes one dense run.
    constexpr int LPR = TILE / 4, ROWS = 4096 / TILE, PITCH = TILE + 1, CPW = TILE / 4, NR = ROWS / 64;
    constexpr int RPP = 256 / LPR;  // rows staged per pass
    __shared__ uint32_t s_rng[ROWS * PITCH];
    const uint32_t W = a.w, H = a.h, f = blockIdx.y, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t q = tid % LPR, ty = tid / LPR;
    const uint32_t c0 = blockIdx.x * TILE;
    const uint32_t ncol = min((uint32_t)TILE, W - c0);
    const uint32_t* rp = a.range + (size_t)f * W * H;
    const uint32_t* off = a.col_off + (size_t)f * (W + 1);
    // nothing kept in this tile: leave before touching the range plane (uniform over the workgroup)
    if (off[c0 + ncol] == off[c0]) return;
    const uint64_t fbase = a.frame_off[f];
    const LutDev lut = a.luts[f % a.n_luts];
    const bool vec = (W % 4 == 0) && ((((uintptr_t)a.range) & 15) == 0);
    // column metadata: lane l of the wave holds column wave*CPW + l % CPW
    const uint32_t ml = lane % CPW, mj = wave * CPW + ml, mx = c0 + mj;
    uint32_t m_base = 0, m_cnt = 0;
    uint64_t m_ts = 0;
    T m_pose[12];
    double m_col[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 12; ++k) m_pose[k] = (T)0;
    if (mj < ncol) {
        m_base = off[mx];
        m_cnt = off[mx + 1] - m_base;
        if (m_cnt) {
            const double* pm = a.poses + ((size_t)f * W + mx) * 16;
#pragma unroll
            for (int k = 0; k < 12; ++k) m_pose[k] = (T)pm[k];
            if constexpr (SEP) {
#pragma unroll
                for (int k = 0; k < 5; ++k) m_col[k] = lut.col_tab[(size_t)mx * 5 + k];
            }
            if (a.timestamps_ns) m_ts = a.timestamp[(size_t)f * W + mx];
        }
    }
    uint32_t m_run = 0;  // points of column l already written (previous row chunks)
    for (uint32_t r0 = 0; r0 < H; r0 += ROWS) {
        __syncthreads();  // previous chunk consumed
#pragma unroll
        for (uint32_t rr = ty; rr < (uint32_t)ROWS; rr += RPP) {
            const uint32_t r = r0 + rr, col = c0 + 4 * q;
            uint32_t v[4] = {0, 0, 0, 0};
            if (r < H && col < W) {
                if (vec) {
                    const uint4 t = *(const uint4*)(rp + (size_t)r * W + col);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
                    for (uint32_t c = 0; c < 4 && col + c < W; ++c) v[c] = rp[(size_t)r * W + col + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) s_rng[rr * PITCH + 4 * q + c] = v[c];
        }
        __syncthreads();
        uint32_t row[NR];
        double bt[NR][9];
#pragma unroll
        for (int hh = 0; hh < NR; ++hh) {
            row[hh] = r0 + lane + 64 * hh;
            if constexpr (SEP) {
#pragma unroll
                for (int k = 0; k < 9; ++k) bt[hh][k] = row[hh] < H ? lut.beam_tab[(size_t)row[hh] * 9 + k] : 0.0;
            }
        }
        for (uint32_t jj = 0; jj < (uint32_t)CPW; ++jj) {
            const uint32_t j = wave * CPW + jj, x = c0 + j;  // wave-uniform
            if (j >= ncol) break;
            if (bcast_u32(m_cnt, jj) == 0) continue;  // masked out or empty column
            uint32_t r[NR], rank[NR];
            bool keep[NR];
            uint32_t n_keep = 0;
#pragma unroll
            for (int hh = 0; hh < NR; ++hh) {
                r[hh] = s_rng[(lane + 64 * hh) * PITCH + j];
                keep[hh] = row[hh] < H && r[hh] >= a.min_r && r[hh] <= a.max_r;
                const uint64_t mask = __ballot(keep[hh]);
                rank[hh] = n_keep + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                n_keep += (uint32_t)__popcll(mask);
            }
            if (n_keep == 0) continue;
            const uint64_t g0 = fbase + bcast_u32(m_base, jj) + bcast_u32(m_run, jj);  // first point of this run
            const uint64_t room = g0 < a.capacity ? a.capacity - g0 : 0;
#pragma unroll
            for (int hh = 0; hh < NR; ++hh) {
                if (!(keep[hh] && rank[hh] < room)) continue;
                double p[3];
                if constexpr (SEP) {
                    const double cx = bcast(m_col[0], jj), sx = bcast(m_col[1], jj);
                    const double rm = (double)r[hh] - lut.n;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double d = fma(cx, bt[hh][k], fma(sx, bt[hh][3 + k], bt[hh][6 + k]));
                        p[k] = r[hh] ? fma(rm, d, bcast(m_col[2 + k], jj)) : 0.0;
                    }
                } else {
                    const size_t pix = (size_t)row[hh] * W + x;
                    if (lut.full_dtype == OUSTER_HIP_F32)
                        project_full<float>((const float*)lut.full_dir, (const float*)lut.full_ofs, pix, r[hh], p);
                    else
                        project_full<double>((const double*)lut.full_dir, (const double*)lut.full_ofs, pix, r[hh], p);
                }
                const T px = (T)p[0], py = (T)p[1], pz = (T)p[2];
                Pt3<T> o;
                o.x = bcast(m_pose[0], jj) * px + bcast(m_pose[1], jj) * py + bcast(m_pose[2], jj) * pz + bcast(m_pose[3], jj);
                o.y = bcast(m_pose[4], jj) * px + bcast(m_pose[5], jj) * py + bcast(m_pose[6], jj) * pz + bcast(m_pose[7], jj);
                o.z = bcast(m_pose[8], jj) * px + bcast(m_pose[9], jj) * py + bcast(m_pose[10], jj) * pz + bcast(m_pose[11], jj);
                ((Pt3<T>*)a.points)[g0 + rank[hh]] = o;
            }
            // every point of the run carries the same provenance: dense lanes 0..n_keep-1
            if (a.col_idxs || a.frame_idxs || a.timestamps_ns) {
                const uint64_t ts = (uint64_t)bcast_u32((uint32_t)m_ts, jj) |
                                    ((uint64_t)bcast_u32((uint32_t)(m_ts >> 32), jj) << 32);
                for (uint32_t i = lane; i < n_keep && i < room; i += 64) {
                    if (a.col_idxs) a.col_idxs[g0 + i] = x;
                    if (a.frame_idxs) a.frame_idxs[g0 + i] = f;
                    if (a.timestamps_ns) a.timestamps_ns[g0 + i] = ts;
                }
            }
            if (ml == jj) m_run += n_keep;
        }
    }
}

// ------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------
size_t decode_lds_bytes(const Geometry& g, int tile) {
    size_t tile_bytes = ((size_t)tile * g.col_size + 16 + 15) & ~(size_t)15;
    size_t h4 = (g.pixels_per_column + 3) & ~3u;
    return tile_bytes + (size_t)tile * 4 + 16 + h4 * 4 + (OUSTER_DECODE_NT / 64) * 192 * 16;
}

template <class S, int TILE>
static hipError_t launch_decode_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds,
                                  hipStream_t st) {
    void (*k)(DecodeArgs) = nullptr;
    switch (xyzm) {
        case 0: k = k_decode<S, TILE, 0>; break;
        case 1: k = k_decode<S, TILE, 1>; break;
        case 2: k = k_decode<S, TILE, 2>; break;
        default: k = k_decode<S, TILE, 3>; break;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, grid, dim3(OUSTER_DECODE_NT), lds, st, a);
    return hipGetLastError();
}

template <class S>
static hipError_t launch_decode_s(const DecodeArgs& a, int tile, int xyzm, dim3 grid, size_t lds,
                                  hipStream_t st) {
    switch (tile) {
        case 64: return launch_decode_t<S, 64>(a, xyzm, grid, lds, st);
        case 32: return launch_decode_t<S, 32>(a, xyzm, grid, lds, st);
        default: return launch_decode_t<S, 16>(a, xyzm, grid, lds, st);
    }
}

hipError_t launch_decode(const DecodeArgs& a, int spec_id, int tile, int xyzm, hipStream_t st) {
    const uint32_t tpf = a.tiles_per_frame;
    const uint32_t nblocks = a.xcd_map ? ((a.n_frames + 7) / 8) * 8 * tpf : a.n_frames * tpf;
    const dim3 grid(nblocks);
    const size_t lds = decode_lds_bytes(a.g, tile);
    switch (spec_id) {
        case SPEC_DUAL_LB: return launch_decode_s<SpecDualLB>(a, tile, xyzm, grid, lds, st);
        case SPEC_LB: return launch_decode_s<SpecLB>(a, tile, xyzm, grid, lds, st);
        case SPEC_SINGLE: return launch_decode_s<SpecSingle>(a, tile, xyzm, grid, lds, st);
        case SPEC_DUAL: return launch_decode_s<SpecDual>(a, tile, xyzm, grid, lds, st);
        case SPEC_LEGACY: return launch_decode_s<SpecLegacy>(a, tile, xyzm, grid, lds, st);
        default: return launch_decode_s<SpecGeneric>(a, tile, xyzm, grid, lds, st);
    }
}

template <class S, int TW>
static hipError_t launch_decode_wide_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds, hipStream_t st) {
    void (*k)(DecodeArgs) = nullptr;
    switch (xyzm) {
        case 0: k = k_decode_wide<S, TW, 0>; break;
        case 1: k = k_decode_wide<S, TW, 1>; break;
        case 2: k = k_decode_wide<S, TW, 2>; break;
        default: k = k_decode_wide<S, TW, 3>; break;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

template <class S>
static hipError_t launch_decode_wide_s(const DecodeArgs& a, int tw, int xyzm, dim3 grid, size_t lds, hipStream_t st) {
    switch (tw) {
        case 128: return launch_decode_wide_t<S, 128>(a, xyzm, grid, lds, st);
        case 512: return launch_decode_wide_t<S, 512>(a, xyzm, grid, lds, st);
        default: return launch_decode_wide_t<S, 256>(a, xyzm, grid, lds, st);
    }
}

size_t decode_wide_lds_bytes(int tw, uint32_t rows_per_tile, uint32_t lds_col_slot) {
    return ((size_t)tw * (lds_col_slot >> 2) + 4 + tw + ((rows_per_tile + 3) & ~3u)) * 4 + 4 * 192 * 16;
}

hipError_t launch_decode_wide(const DecodeArgs& a, int spec_id, int tw, int xyzm, hipStream_t st) {
    const uint32_t bpf = a.tiles_per_frame * a.row_chunks;
    const uint32_t nblocks = a.xcd_map ? ((a.n_frames + 7) / 8) * 8 * bpf : a.n_frames * bpf;
    const dim3 grid(nblocks);
    const size_t lds = decode_wide_lds_bytes(tw, a.rows_per_tile, a.lds_col_slot);
    switch (spec_id) {
        case SPEC_DUAL_LB: return launch_decode_wide_s<SpecDualLB>(a, tw, xyzm, grid, lds, st);
        case SPEC_LB: return launch_decode_wide_s<SpecLB>(a, tw, xyzm, grid, lds, st);
        case SPEC_SINGLE: return launch_decode_wide_s<SpecSingle>(a, tw, xyzm, grid, lds, st);
        case SPEC_DUAL: return launch_decode_wide_s<SpecDual>(a, tw, xyzm, grid, lds, st);
        case SPEC_LEGACY: return launch_decode_wide_s<SpecLegacy>(a, tw, xyzm, grid, lds, st);
        default: return launch_decode_wide_s<SpecGeneric>(a, tw, xyzm, grid, lds, st);
    }
}

hipError_t launch_colmap(const ColmapArgs& a, uint32_t n_frames, hipStream_t st) {
    const uint32_t slots = a.slots_per_frame * a.g.columns_per_packet;
    constexpr int U = 4;  // column headers per thread (1/2/4/8 measured within 1 us of each other)
    dim3 grid((slots + 256 * U - 1) / (256 * U), n_frames);
    hipLaunchKernelGGL(k_colmap<U>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_destagger(const DestaggerArgs& a, uint32_t n_images, hipStream_t st) {
    dim3 grid(a.h, n_images);
    hipLaunchKernelGGL(k_destagger, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// tile width of the standalone tiled kernels.  Unlike the 14-stream k_decode these one/two-stream
// kernels gain nothing from 256-column tiles (same-box A/B: equal for f32, 5-10 % slower for f64 and
// full-LUT), so 64 stays the default; OUSTER_HIP_CT_TILE=256 is kept for experiments.
static uint32_t standalone_tile_width(uint32_t w) {
    static const int env = [] { const char* e = getenv("OUSTER_HIP_CT_TILE"); return e ? atoi(e) : 0; }();
    (void)w;
    return env == 256 ? 256u : 64u;
}

hipError_t launch_cartesian(const CartesianArgs& a_in, int mode, hipStream_t st) {
    CartesianArgs a = a_in;
    if (a.vec_ok && a.w % 4 == 0) {
        // enough workgroups to fill the chip: split the rows when the batch is small
        const uint32_t tw = standalone_tile_width(a.w);
        const uint32_t tiles = (a.w + tw - 1) / tw;
        uint32_t rpb = a.h;
        while (rpb > 16 && (size_t)tiles * a.n_images * ((a.h + rpb - 1) / rpb) < 1024) rpb = (rpb + 1) / 2;
        rpb = (rpb + 15) / 16 * 16;
        a.rows_per_block = rpb;
        dim3 grid(tiles * ((a.h + rpb - 1) / rpb), a.n_images);
        if (tw == 256) {
            switch (mode) {
                case 1: hipLaunchKernelGGL((k_cartesian_tiled<1, 256>), grid, dim3(256), 0, st, a); break;
                case 2: hipLaunchKernelGGL((k_cartesian_tiled<2, 256>), grid, dim3(256), 0, st, a); break;
                default: hipLaunchKernelGGL((k_cartesian_tiled<3, 256>), grid, dim3(256), 0, st, a); break;
            }
        } else {
            switch (mode) {
                case 1: hipLaunchKernelGGL((k_cartesian_tiled<1, 64>), grid, dim3(256), 0, st, a); break;
                case 2: hipLaunchKernelGGL((k_cartesian_tiled<2, 64>), grid, dim3(256), 0, st, a); break;
                default: hipLaunchKernelGGL((k_cartesian_tiled<3, 64>), grid, dim3(256), 0, st, a); break;
            }
        }
        return hipGetLastError();
    }
    const size_t quads = ((size_t)a.w * a.h + 3) / 4 * a.n_images;
    size_t blocks = (quads + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) blocks = 1;
    dim3 grid((uint32_t)blocks);
    switch (mode) {
        case 1: hipLaunchKernelGGL(k_cartesian<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(k_cartesian<2>, grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(k_cartesian<3>, grid, dim3(256), 0, st, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_dewarp(const DewarpArgs& a_in, hipStream_t st) {
    DewarpArgs a = a_in;
    if (a.w % 4 == 0 && (((uintptr_t)a.points | (uintptr_t)a.out) & 15) == 0) {
        const uint32_t tw = standalone_tile_width(a.w);
        const uint32_t tiles = (a.w + tw - 1) / tw;
        uint32_t rpb = a.h;
        while (rpb > 16 && (size_t)tiles * a.n_images * ((a.h + rpb - 1) / rpb) < 1024) rpb = (rpb + 1) / 2;
        rpb = (rpb + 15) / 16 * 16;
        a.rows_per_block = rpb;
        dim3 grid(tiles * ((a.h + rpb - 1) / rpb), a.n_images);
        if (tw == 256) {
            if (a.dtype == OUSTER_HIP_F32) hipLaunchKernelGGL((k_dewarp_tiled<float, 256>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_dewarp_tiled<double, 256>), grid, dim3(256), 0, st, a);
        } else {
            if (a.dtype == OUSTER_HIP_F32) hipLaunchKernelGGL((k_dewarp_tiled<float, 64>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_dewarp_tiled<double, 64>), grid, dim3(256), 0, st, a);
        }
        return hipGetLastError();
    }
    const size_t total = (size_t)a.w * a.h * a.n_images;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) blocks = 1;
    if (a.dtype == OUSTER_HIP_F32) hipLaunchKernelGGL(k_dewarp<float>, dim3((uint32_t)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_dewarp<double>, dim3((uint32_t)blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_dewarp_frames(const DewarpFramesArgs& a, bool separable, hipStream_t st) {
    const uint32_t tiles = (a.w + 63) / 64;
    hipLaunchKernelGGL(k_dwf_count, dim3(tiles, a.n_frames), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dwf_scan, dim3(a.n_frames), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dwf_frame_scan, dim3(1), dim3(256), 0, st, a);
    const dim3 grid(tiles, a.n_frames);  // 64 x 64 emit tiles (32 x 128 measured 15 % slower)
    if (a.dtype == OUSTER_HIP_F32) {
        if (separable) hipLaunchKernelGGL((k_dwf_emit<float, true, 64>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_dwf_emit<float, false, 64>), grid, dim3(256), 0, st, a);
    } else {
        if (separable) hipLaunchKernelGGL((k_dwf_emit<double, true, 64>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_dwf_emit<double, false, 64>), grid, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace ouster_hip_dev
