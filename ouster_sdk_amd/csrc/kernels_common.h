// kernels_common.h -- device helpers shared by the kernel translation units
// (k_decode.hip is compiled once per packet-profile specialisation, k_standalone.hip once).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ouster_hip_dev.h"

// 1: the f32 xyz transpose of the fused kernels goes through ds_bpermute (no LDS scratch);
// 0: through a 12 KB wave-private LDS scratch (the r01 form, kept for A/B builds)
#ifndef OUSTER_XYZ_PERMUTE
#define OUSTER_XYZ_PERMUTE 1
#endif

namespace ouster_hip_dev {

constexpr size_t XYZ_SCRATCH_BYTES = OUSTER_XYZ_PERMUTE ? 0 : 4 * 192 * 16;

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

// wave-wide reductions (every lane gets the result): one LDS / global atomic per wave instead of one per lane -- 256 lanes
// adding to ONE LDS word are served one after the other (20 - 40 us for 2000 of them, tools/ab/phase_fixup.py)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v |= (uint32_t)__shfl_xor((int)v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += (uint32_t)__shfl_xor((int)v, d);
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

// The same two reads split into "issue the loads" and "use them": a consumer right behind a global load
// makes the wave sit out a full memory latency, and the wide decode kernel has better things to put in
// flight first (its tile).  RawWin keeps the dwords as they arrive; window_compose is the old arithmetic.
struct RawWin {
    uint32_t d[3];
    int32_t sh0;   // bit position of d[0] inside the 64-bit window at p (negative: d[0] starts before p)
};
__device__ __forceinline__ RawWin window_global_issue(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    RawWin w;
    w.sh0 = -(int32_t)(a & 3) * 8;
    w.d[0] = q[0]; w.d[1] = q[1];
    w.d[2] = (a & 3) ? q[2] : 0u;
    return w;
}
__device__ __forceinline__ RawWin window_global_masked_issue(const uint8_t* p, uint64_t mask) {
    RawWin w{{0u, 0u, 0u}, 0};
    if (mask == 0) return w;
    const uint32_t lo = (uint32_t)__builtin_ctzll(mask) >> 3, hi = (63u - (uint32_t)__builtin_clzll(mask)) >> 3;
    const uintptr_t a = (uintptr_t)p;
    const uintptr_t first = (a + lo) & ~(uintptr_t)3, last = (a + hi) & ~(uintptr_t)3;
    w.sh0 = (int32_t)((long)(first - a) * 8);
    const uint32_t* q = (const uint32_t*)first;
    w.d[0] = q[0];
    if (first + 4 <= last) w.d[1] = q[1];
    if (first + 8 <= last) w.d[2] = q[2];
    return w;
}
__device__ __forceinline__ uint64_t window_compose(const RawWin& w) {
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t sh = w.sh0 + 32 * k;
        const uint64_t d = w.d[k];
        v |= sh >= 0 ? (sh < 64 ? d << sh : 0ull) : (sh > -32 ? d >> (-sh) : 0ull);
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
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) pk4 { uint32_t a; };
struct __attribute__((packed, aligned(1))) pk8 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) pk16 { uint32_t a, b, c, d; };

#ifndef OUSTER_NT_STORES
#define OUSTER_NT_STORES 0   // experiment switch (tools/ab/nt_variants.sh): non-temporal hint on EVERY plane / xyz store
#endif
#ifndef OUSTER_NT_STANDALONE
#define OUSTER_NT_STANDALONE 1   // the hint on the stores of the standalone kernels (k_standalone.hip): +1.9 ... +5.7 % in-process on cold inputs
#endif
#ifndef OUSTER_NT_U32
#define OUSTER_NT_U32 0      // experiment switch: the hint on the 4-byte plane stores (and whatever else asks for it)
#endif
#ifndef OUSTER_NT_XYZ
#define OUSTER_NT_XYZ 0      // experiment switch: the hint on the xyz stores only
#endif
#ifndef OUSTER_NT_PLANES
#define OUSTER_NT_PLANES 0   // experiment switch: the hint on the plane stores only
#endif
#ifndef OUSTER_NT_LOADS
#define OUSTER_NT_LOADS 0    // experiment switch: non-temporal hint on the tile staging loads
#endif
// NT = non-temporal hint.  Measured in-process (BASELINE section 5): the 12 B/px profile's wide tiles gain 3 - 8 % with
// it (128 x 32: 4.7 % in a fast region, 3 % in a slow one; 256 x 16: 7.5 %) while its 32-column k_decode loses 18 %; the
// dual-return 256 x 32 tiles are indifferent and the 128 x 64 tiles lose 6.5 % -- so it is a property of the profile
// specialisation AND the kernel (Spec::nt_stores, applied by k_decode_wide only), not a global switch.
typedef uint32_t u32x1_u __attribute__((aligned(1)));
typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
#ifndef OUSTER_PLAIN_STORES
#define OUSTER_PLAIN_STORES 0   // experiment switch: ignore every non-temporal request (the A/B partner of Spec::nt_stores)
#endif
template <bool NT = false>
__device__ __forceinline__ void st4(void* p, uint32_t a) {
    if constexpr ((NT || OUSTER_NT_STORES) && !OUSTER_PLAIN_STORES) __builtin_nontemporal_store(a, (u32x1_u*)p);
    else ((pk4*)p)->a = a;
}
template <bool NT = false>
__device__ __forceinline__ void st8(void* p, uint32_t a, uint32_t b) {
    if constexpr ((NT || OUSTER_NT_STORES) && !OUSTER_PLAIN_STORES) {
        typedef uint32_t v2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(v2{a, b}, (u32x2_u*)p);
    } else {
        pk8 v{a, b};
        *((pk8*)p) = v;
    }
}
template <bool NT = false>
__device__ __forceinline__ void st16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    if constexpr ((NT || OUSTER_NT_STORES) && !OUSTER_PLAIN_STORES) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(v4{a, b, c, d}, (u32x4_u*)p);
    } else {
        pk16 v{a, b, c, d};
        *((pk16*)p) = v;
    }
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

// the same for values of at most 4 bytes held as one register quad (the static profiles): a u32 plane's
// four elements ARE the 16 B store operand, nothing is moved
template <bool NT = false>
__device__ __forceinline__ void store4v(uint8_t* p, const u32x4_t& v, uint32_t elem) {
    switch (elem) {
        case 1: st4<NT>(p, v.x | (v.y << 8) | (v.z << 16) | (v.w << 24)); break;
        case 2: st8<NT>(p, v.x | (v.y << 16), v.z | (v.w << 16)); break;
        default: {
            if constexpr ((NT || OUSTER_NT_STORES || OUSTER_NT_U32) && !OUSTER_PLAIN_STORES) {
                __builtin_nontemporal_store(v, (u32x4_u*)p);
            } else {
                struct __attribute__((packed, aligned(1))) pkv { u32x4_t v; };
                ((pkv*)p)->v = v;
            }
        }
    }
}

__device__ __forceinline__ uint64_t zero_value(uint32_t f16_nan) {
    return f16_nan ? 0x7e007e007e007e00ull : 0ull;
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
    static constexpr bool nt_stores = false;  // measured: 256 x 32 tiles +0.4 %, 128 x 64 tiles -6.5 % (the plane stores: -8.7 %)
    static constexpr bool nt_xyz = true;      // the xyz stores alone: +0.6 % / +0.9 % (k_decode_wide only)
    static constexpr uint32_t chan = 8;
    static constexpr int nf = 8;
    static constexpr int range_idx = 0, range2_idx = 4;
    static constexpr FieldC f[8] = {{0, 0x7fff, -3, 4}, {1, 0x80, 7, 1},  {2, 0xff, 0, 1},
                                    {3, 0xff, -4, 2},   {4, 0x7fff, -3, 4}, {5, 0x80, 7, 1},
                                    {6, 0xff, 0, 1},    {7, 0xff, 0, 1}};
};
struct SpecLB {  // RNG15_RFL8_NIR8, 4 B/px
    static constexpr bool is_static = true;
    static constexpr bool nt_stores = false;  // measured +2.5 ... +7.3 % with the buffers in a fast region but -5.7 % in a slow one: off
    static constexpr bool nt_xyz = nt_stores;
    static constexpr uint32_t chan = 4;
    static constexpr int nf = 4;
    static constexpr int range_idx = 0, range2_idx = -1;
    static constexpr FieldC f[4] = {{0, 0x7fff, -3, 4}, {1, 0x80, 7, 1}, {2, 0xff, 0, 1},
                                    {3, 0xff, -4, 2}};
};
struct SpecSingle {  // RNG19_RFL8_SIG16_NIR16, 12 B/px
    static constexpr bool is_static = true;
    static constexpr bool nt_stores = true;   // k_decode_wide only; measured +3 ... +8 %
    static constexpr bool nt_xyz = nt_stores;
    static constexpr uint32_t chan = 12;
    static constexpr int nf = 6;
    static constexpr int range_idx = 0, range2_idx = -1;
    static constexpr FieldC f[6] = {{0, 0x7ffff, 0, 4}, {2, 0xf8, 3, 1},   {4, 0xff, 0, 1},
                                    {6, 0xffff, 0, 2},  {8, 0xffff, 0, 2}, {11, 0xff, 0, 1}};
};
struct SpecDual {  // RNG19_RFL8_SIG16_NIR16_DUAL, 16 B/px
    static constexpr bool is_static = true;
    static constexpr bool nt_stores = true;   // k_decode_wide only; measured +2.4 % (128 x 32) ... +4.8 % (256 x 16)
    static constexpr bool nt_xyz = nt_stores;
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
    static constexpr bool nt_stores = true;   // k_decode_wide only; measured +3.6 % (128 x 32) ... +5.4 % (256 x 16)
    static constexpr bool nt_xyz = nt_stores;
    static constexpr uint32_t chan = 12;
    static constexpr int nf = 5;
    static constexpr int range_idx = 0, range2_idx = -1;
    static constexpr FieldC f[5] = {{0, 0xfffff, 0, 4}, {3, 0xf0, 4, 1}, {4, 0xff, 0, 1},
                                    {6, 0xffff, 0, 2},  {8, 0xffff, 0, 2}};
};
struct SpecGeneric {  // everything else: descriptors read from the kernel arguments
    static constexpr bool is_static = false;
    static constexpr bool nt_stores = false;
    static constexpr bool nt_xyz = nt_stores;
    static constexpr uint32_t chan = 0;
    static constexpr int nf = 0;
    static constexpr int range_idx = -1, range2_idx = -1;
};

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
// P = precision the points are held in (double, or already rounded to the output type)
template <class T, class P>
__device__ __forceinline__ void store_xyz4(T* dst, const P (&p)[4][3]) {
    if constexpr (sizeof(T) == 4) {
        float4* d = (float4*)dst;  // 48 B, 16 B aligned (pixel index multiple of 4)
        d[0] = make_float4((float)p[0][0], (float)p[0][1], (float)p[0][2], (float)p[1][0]);
        d[1] = make_float4((float)p[1][1], (float)p[1][2], (float)p[2][0], (float)p[2][1]);
        d[2] = make_float4((float)p[2][2], (float)p[3][0], (float)p[3][1], (float)p[3][2]);
    } else {
        double2* d = (double2*)dst;
        d[0] = make_double2(p[0][0], p[0][1]);
        d[1] = make_double2(p[0][2], p[1][0]);
        d[2] = make_double2(p[1][1], p[1][2]);
        d[3] = make_double2(p[2][0], p[2][1]);
        d[4] = make_double2(p[2][2], p[3][0]);
        d[5] = make_double2(p[3][1], p[3][2]);
    }
}

// f32 xyz of 4 consecutive pixels per lane = 48 contiguous bytes per lane.  Stored directly,
// each of the three 16 B store instructions would touch every third 16 B chunk of the row
// segment.  Instead the wave transposes through a private LDS scratch so that instruction k
// writes chunks [k*LPR, (k+1)*LPR) of the row segment: LPR x 16 B contiguous per row.
//   row_base: xyz address of the first pixel of this lane's row segment (tile column 0)
template <int LPR, class P>
__device__ __forceinline__ void store_xyz4_coalesced(float4* s_xyz, uint32_t tid, float* row_base,
                                                     uint32_t q, const P (&p)[4][3]) {
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

// The same transpose without any LDS memory: three rounds of ds_bpermute (the LDS crossbar moves
// registers between lanes, nothing is allocated).  A lane's 48 B are chunks 3q, 3q+1, 3q+2 of its
// LPR-lane row segment; store instruction k must write chunk k*LPR + q.  In round j every lane offers
// its chunk j (the same register in all lanes, as bpermute requires): lane d receives chunk
// 3s + j == d (mod LPR) from lane s = (d - j) * 3^-1 mod LPR -- a bijection because LPR is a power
// of two -- and files it under k = (3s + j) / LPR.  Frees the 12 KB scratch (one more 12 B/px
// workgroup per CU).
template <int LPR, bool NT = false, class P>
__device__ __forceinline__ void store_xyz4_permuted(float* row_base, uint32_t q, const P (&p)[4][3]) {
    static_assert(LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32 || LPR == 64, "row segment lanes");
    constexpr uint32_t INV3 = LPR == 64 ? 43u : (LPR >= 16 ? 11u : 3u);  // 3 * INV3 == 1 (mod LPR)
    const uint32_t lane = threadIdx.x & 63u, seg = lane - q;             // first lane of my row segment
    const float v[12] = {(float)p[0][0], (float)p[0][1], (float)p[0][2], (float)p[1][0],
                         (float)p[1][1], (float)p[1][2], (float)p[2][0], (float)p[2][1],
                         (float)p[2][2], (float)p[3][0], (float)p[3][1], (float)p[3][2]};
    float o[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t s = ((q + LPR - j) * INV3) & (LPR - 1);   // source lane within the segment
        const uint32_t k = (3u * s + j) / LPR;                   // which of my three stores it feeds
        const int addr = (int)((seg + s) << 2);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float r = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v[4 * j + w])));
            if (j == 0) { o[0][w] = r; o[1][w] = r; o[2][w] = r; }
            else {
                o[0][w] = k == 0 ? r : o[0][w];
                o[1][w] = k == 1 ? r : o[1][w];
                o[2][w] = k == 2 ? r : o[2][w];
            }
        }
    }
    float4* d = (float4*)row_base;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if constexpr ((NT || OUSTER_NT_STORES) && !OUSTER_PLAIN_STORES) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f4{o[k][0], o[k][1], o[k][2], o[k][3]}, (f4*)(d + k * LPR + q));
        } else {
            d[k * LPR + q] = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
        }
    }
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
    for (int k = 0; k < NV; ++k) {
#if OUSTER_NT_STANDALONE
        typedef float f4 __attribute__((ext_vector_type(4)));
        const float4 t = sc[k * LPR + q];
        __builtin_nontemporal_store(f4{t.x, t.y, t.z, t.w}, (f4*)(row_base + k * LPR + q));
#else
        row_base[k * LPR + q] = sc[k * LPR + q];
#endif
    }
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

template <class T, class P>
__device__ __forceinline__ void store_xyz1(T* dst, const P (&p)[3]) {
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
// frame_state: u64 words shared by the optimistic decode pass and the fix-up pass behind it
// (DESIGN.md section 3.1).
//   [FS_SEQ]        sequence number, advanced by the fix-up pass of every call
//   [FS_TAG]        tag (= sequence + 1) the last optimistic pass flagged with
//   [FS_WORDS + f]  frame f was flagged by the call whose tag this is ("a live column sits outside
//                   its home slot"); any other value means clean.  Flags are never cleared: the next
//                   call uses a larger tag.  64 bits do not wrap.
// ------------------------------------------------------------------------------------
//   [FS_TICKET ..]  2 x 8 ticket counters of k_decode_wide_fixup (tag parity x XCD), see there
//   [FS_ANY]        the launch-wide word: raised to the call's tag with the first frame word (round 5).  A clean batch is
//                   "FS_ANY != tag": the fix-up kernel's workgroups leave after two scalar loads instead of reading and
//                   listing every frame word (8 us behind every optimistic pass in round 4)
// (the indices FS_* are in ouster_hip_dev.h: the host sizes the buffer)

// a stray was seen in frame f: the frame's word and the launch-wide word are raised to this call's tag (nothing is ever cleared)
__device__ __forceinline__ void flag_frame(const DecodeArgs& a, uint32_t f, uint64_t tag) {
    atomicMax((unsigned long long*)&a.frame_state[FS_WORDS + f], (unsigned long long)tag);
    atomicMax((unsigned long long*)&a.frame_state[FS_ANY], (unsigned long long)tag);
}

// (measurement_id, status) of a column header staged in LDS at byte offset cb
__device__ __forceinline__ void col_header_lds(const Geometry& g, const uint32_t* s_tile, uint32_t cb,
                                               uint32_t& m_id, uint32_t& status) {
    m_id = (uint16_t)apply_bits(window_lds(s_tile, cb + g.col_measurement_id.offset),
                                g.col_measurement_id.mask, g.col_measurement_id.shift);
    status = (uint32_t)apply_bits(window_lds(s_tile, cb + g.col_status.offset), g.col_status.mask,
                                  g.col_status.shift);
}

// frame-level values latched from the first packet of a frame (start_frame, lidar_frame.cpp:1709-1741)
__device__ __forceinline__ ouster_hip_frame_meta frame_meta_of(const Geometry& g, const uint8_t* pkt,
                                                               bool any_packet) {
    ouster_hip_frame_meta m;
    m.frame_id = -1; m.frame_status = 0; m.shutdown_countdown = 0;
    m.shot_limiting_countdown = 0; m.n_valid_columns = 0;
    if (any_packet) {
        m.frame_id = (int64_t)(uint32_t)apply_bits(window_global(pkt + g.frame_id.offset), g.frame_id.mask,
                                                    g.frame_id.shift);
        const uint8_t th = (uint8_t)apply_bits(window_global(pkt + g.thermal_shutdown.offset),
                                               g.thermal_shutdown.mask, g.thermal_shutdown.shift);
        const uint8_t sl = (uint8_t)apply_bits(window_global(pkt + g.shot_limiting.offset),
                                               g.shot_limiting.mask, g.shot_limiting.shift);
        m.frame_status = (uint64_t)(th & 0x0f) | ((uint64_t)(sl & 0x0f) << 4);
        m.shutdown_countdown = (uint16_t)apply_bits(window_global(pkt + g.countdown_thermal_shutdown.offset),
                                                    g.countdown_thermal_shutdown.mask,
                                                    g.countdown_thermal_shutdown.shift);
        m.shot_limiting_countdown = (uint16_t)apply_bits(window_global(pkt + g.countdown_shot_limiting.offset),
                                                         g.countdown_shot_limiting.mask,
                                                         g.countdown_shot_limiting.shift);
    }
    return m;
}

// The optimistic pass works on home slots: a lost packet is a zeroed slot, and slot 0 may be one.  The reference latches
// the frame-level values from the first packet it RECEIVES (start_frame, lidar_frame.cpp:1709-1741), so they come from the
// first slot that holds a packet -- anything non-zero in its first 16 bytes (packet header, or a LEGACY column header) or
// in its first column's status word.  Slot 0 almost always does; the scan behind it only runs when it does not.
__device__ __forceinline__ ouster_hip_frame_meta frame_meta_first_present(const Geometry& g, const uint8_t* fbase,
                                                                          size_t packet_stride, uint32_t count) {
    for (uint32_t p = 0; p < count; ++p) {
        const uint8_t* pkt = fbase + (size_t)p * packet_stride;
        const uint32_t* q = (const uint32_t*)pkt;
        uint32_t any = q[0] | q[1] | q[2] | q[3];
        any |= (uint32_t)window_global_masked(pkt + g.packet_header_size + g.col_status.offset, g.col_status.mask) |
               (uint32_t)(window_global_masked(pkt + g.packet_header_size + g.col_status.offset, g.col_status.mask) >> 32);
        if (any) return frame_meta_of(g, pkt, true);
    }
    return frame_meta_of(g, fbase, false);
}

// The general mapping's frame-level values: from the first packet of the buffer -- the first one the reference receives
// (start_frame) -- unless the buffer has one slot per column of the frame, where a lost packet is a zeroed slot (this
// library's own staging convention: FrameStream, DeviceFrameBatch) and the first slot that holds a packet counts.
__device__ __forceinline__ ouster_hip_frame_meta frame_meta_general(const DecodeArgs& a, const uint8_t* fbase, uint32_t count) {
    if ((uint64_t)a.slots_per_frame * a.g.columns_per_packet == a.g.columns_per_frame)
        return frame_meta_first_present(a.g, fbase, a.packet_stride, count);
    return frame_meta_of(a.g, fbase, count > 0);
}

// ------------------------------------------------------------------------------------
// resolve_frame: the general column mapping of ONE frame -- what FrameBatcher leaves behind after batching the frame's
// packets in buffer order (ouster_core/src/lidar_frame.cpp:1422-1576), restated as "last event wins" so that it can be
// computed in parallel.  The reference walks the packets one after the other with one piece of state, next_valid_m_id:
//   block path  (every column valid, measurement_id < W, every block of BD = block_parsable() columns ends inside the frame;
//               batch_lidar_packet :1542-1573, parse_by_block :1492-1528): if m_id(col 0) >= next_valid, columns
//               [next_valid, m_id(col 0)) of planes AND headers are zeroed and next_valid = m_id(col 0) + cpp; column headers
//               go to every column's own m_id; PIXELS of block b go to m_id(first column of b) + x (block_field,
//               parsing.cpp:628-654) -- not to the columns' own ids;
//   column path (anything else; parse_by_col :1422-1466): every valid column with m_id < W: if m_id >= next_valid, columns
//               [next_valid, m_id) are zeroed and next_valid = m_id + 1; header and pixels go to m_id;
//   finalize    planes (not headers) of [next_valid, W) are zeroed (:1612-1617); headers were zeroed at frame start.
// next_valid never decreases and every zeroed range ends where the next can begin, so the ranges are disjoint: a column is
// zeroed at most once, by one trigger slot z.  Its final source is then the LAST slot written to it, if that slot is not
// older than z -- max over the candidates, kept when >= z.  For packets with consecutive ids (every sensor) both maps are
// "the last slot whose live column carries that measurement_id", what rounds 1-3 computed.
// Output, in LDS: s_pix[c] / s_hdr[c] = buffer slot (packet * cpp + column) that supplies destination column c's pixels /
// its header, or -1 (zeros).  s_pkm[i] (may be nullptr) = last packet whose first column's m_id / cpp == i (packet-level
// outputs, batch_lidar_packet :1534-1539).  Returns next_valid at the end of the frame (0 on the common-case path).
// Pixel columns the reference neither writes nor zeroes (BD < cpp and a block's ids not consecutive with the block before)
// keep the previous contents of the caller's LidarFrame there; here they read as zeros (documented).
// All NT threads call it; it ends with a barrier.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_parsable_dev(uint32_t H, uint32_t cpp) {   // parsing.cpp:958-966
    for (uint32_t d : {16u, 8u, 4u})
        if (H % d == 0 && cpp % d == 0) return d;
    return 0u;
}

// hdr_words (optional): the frame's slots' (measurement_id | valid << 16) already packed in global memory.
// s_hd[count * cpp]: (measurement_id | valid << 16) of every buffer slot, left behind for the caller (the fix-up pass looks
// up "is slot c live and at home" there).  LDS words needed: resolve_lds_words().
__host__ __device__ inline size_t resolve_lds_words(uint32_t W, uint32_t npo, uint32_t slots_per_frame, uint32_t cpp) {
    return (size_t)3 * W + npo + 2 * (size_t)slots_per_frame + (size_t)slots_per_frame * cpp + 4;
}
struct ResolveLds {
    int32_t *pix, *hdr, *z, *pkm;
    uint32_t *pkt, *hd;
    __device__ __forceinline__ ResolveLds(uint32_t* base, uint32_t W, uint32_t npo, uint32_t slots_per_frame) {
        pix = (int32_t*)base; hdr = pix + W; z = hdr + W; pkm = z + W;
        pkt = (uint32_t*)(pkm + npo); hd = pkt + 2 * (size_t)slots_per_frame;
    }
};

template <int NT>
__device__ __forceinline__ uint32_t resolve_frame(const Geometry& g, const uint8_t* fbase, size_t packet_stride, uint32_t count,
                                                  uint32_t npo, const ResolveLds& L, bool want_pkm,
                                                  const uint32_t* hdr_words = nullptr, uint64_t* pt = nullptr) {
    const uint32_t tid = threadIdx.x;
#ifdef OUSTER_PHASE_TIMING
#define RSTAMP(i) do { if (pt && tid == 0) pt[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RSTAMP(i) do { (void)pt; } while (0)
#endif
    RSTAMP(8);
    const uint32_t W = g.columns_per_frame, cpp = g.columns_per_packet;
    const uint32_t BD = block_parsable_dev(g.pixels_per_column, cpp);
    int32_t *s_pix = L.pix, *s_hdr = L.hdr, *s_z = L.z, *s_pkm = want_pkm ? L.pkm : nullptr;
    uint32_t *s_pkt = L.pkt, *s_hd = L.hd;
    // ---- 0: every slot's (measurement_id, valid), all threads, eight slots each in flight (the only global reads).
    // Unconditional loads from clamped addresses (a load inside a guarded region is waited for on its own at the end of the
    // region), and only the DISTINCT dwords the two fields live in -- one for the standard column header, two for LEGACY:
    // every lane's column is another cache line, so each load instruction is 64 line requests and a workgroup's 2048 slots
    // x 4 dwords kept its CU's address unit busy for 12 us.
    const uint32_t nslots = count * cpp;
    constexpr int U = 8;
    const uint32_t mid_lo = (uint32_t)__builtin_ctzll(g.col_measurement_id.mask | (1ull << 63)) >> 3,
                   mid_hi = (63u - (uint32_t)__builtin_clzll(g.col_measurement_id.mask | 1ull)) >> 3;
    const uint32_t st_lo = (uint32_t)__builtin_ctzll(g.col_status.mask | (1ull << 63)) >> 3,
                   st_hi = (63u - (uint32_t)__builtin_clzll(g.col_status.mask | 1ull)) >> 3;
    // byte offsets (from the column start, columns are 4-byte aligned) of the dwords the fields touch
    const uint32_t want[4] = {(g.col_measurement_id.offset + mid_lo) & ~3u, (g.col_measurement_id.offset + mid_hi) & ~3u,
                              (g.col_status.offset + st_lo) & ~3u, (g.col_status.offset + st_hi) & ~3u};
    uint32_t dw[4], n_dw = 0, which[4];
    for (int k = 0; k < 4; ++k) {
        uint32_t j = 0;
        while (j < n_dw && dw[j] != want[k]) ++j;
        if (j == n_dw) dw[n_dw++] = want[k];
        which[k] = j;
    }
    auto load_headers = [&](auto ndw_) {
        constexpr int ND = decltype(ndw_)::value;
        for (uint32_t base = 0; base < nslots; base += NT * U) {
            uint32_t d[U][ND];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = min(base + (uint32_t)u * NT + tid, nslots - 1u);
                const uint32_t p = s / cpp, ic = s - p * cpp;
                const uint8_t* colp = fbase + (size_t)p * packet_stride + g.packet_header_size + (size_t)ic * g.col_size;
#pragma unroll
                for (int k = 0; k < ND; ++k) d[u][k] = *(const uint32_t*)(colp + dw[k]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = base + (uint32_t)u * NT + tid;
                if (s >= nslots) continue;
                // the 64-bit window at the field's offset from the (at most two, for fields of up to four bytes) dwords read
                auto window = [&](uint32_t off, uint32_t k0, uint32_t k1) -> uint64_t {
                    uint32_t d0 = 0, d1 = 0;
#pragma unroll
                    for (int k = 0; k < ND; ++k) { d0 = which[k0] == (uint32_t)k ? d[u][k] : d0; d1 = which[k1] == (uint32_t)k ? d[u][k] : d1; }
                    const int32_t sh0 = ((int32_t)want[k0] - (int32_t)off) * 8, sh1 = ((int32_t)want[k1] - (int32_t)off) * 8;
                    uint64_t v = sh0 >= 0 ? (uint64_t)d0 << sh0 : (uint64_t)d0 >> (-sh0);
                    if (want[k1] != want[k0]) v |= sh1 < 64 ? (uint64_t)d1 << sh1 : 0ull;
                    return v;
                };
                const uint32_t m_id = (uint16_t)apply_bits(window(g.col_measurement_id.offset, 0, 1), g.col_measurement_id.mask,
                                                           g.col_measurement_id.shift);
                const uint32_t st = (uint32_t)apply_bits(window(g.col_status.offset, 2, 3), g.col_status.mask, g.col_status.shift);
                s_hd[s] = m_id | ((st & 1u) << 16);
            }
        }
    };
    if (hdr_words) {   // the optimistic pass before this fix-up pass left them packed (DecodeArgs::hdr_words)
        for (uint32_t s = tid; s < nslots; s += NT) s_hd[s] = hdr_words[s];
    } else if (n_dw == 1) load_headers(std::integral_constant<int, 1>{});
    else if (n_dw == 2) load_headers(std::integral_constant<int, 2>{});
    else load_headers(std::integral_constant<int, 4>{});
    for (uint32_t i = tid; i < W; i += NT) { s_pix[i] = -1; s_hdr[i] = -1; s_z[i] = 0; }
    if (s_pkm) for (uint32_t i = tid; i < npo; i += NT) s_pkm[i] = -1;
    __syncthreads();
    RSTAMP(9);
    // ---- The common case, decided first.  The result is more than "the last live slot that carries the column's id" only
    // when (i) the block path puts pixels somewhere else than their own ids -- an all-valid packet with non-consecutive ids --
    // or (ii) a zeroed range or the end-of-frame zeroing reaches something already written.  A column-path write at c leaves
    // next_valid > c, so (ii) needs a block-path packet that arrives when next_valid lies INSIDE its span (F < next_valid <
    // F + cpp: it is written without moving next_valid, and the next jump zeroes its tail) -- e.g. a packet whose first copy
    // ended in invalid columns, sent again complete.  When every packet that has live columns is "whole at its tail and
    // aligned" -- its live columns carry ids base + ic with one base that is a multiple of cpp, and its last column is live --
    // every value next_valid ever takes is a multiple of cpp and neither can happen: any order, loss, duplicates, packets of
    // invalid columns.  Then one pass of atomicMax settles both maps and the bookkeeping below is skipped (18 -> 6 us per frame).
    if (cpp <= 64 && (cpp & (cpp - 1)) == 0) {
        const uint32_t lane_ = tid & 63u, ic_ = lane_ & (cpp - 1u);
        const uint64_t gm = cpp >= 64 ? ~0ull : ((1ull << cpp) - 1ull);
        bool odd = false;
        for (uint32_t base = 0; base < nslots; base += NT * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = base + (uint32_t)u * NT + tid;
                const bool in = s < nslots;
                const uint32_t h = in ? s_hd[s] : 0u, m = h & 0xffffu;
                const bool live = in && (h >> 16) && m < W;
                const uint32_t mybase = m - ic_;                                                   // wraps for ids below ic: never equal to an aligned base then
                const uint32_t ref = (uint32_t)__shfl((int)mybase, (int)(lane_ - ic_ + cpp - 1u));   // the packet's last column
                const uint64_t lv = (__ballot(live) >> (lane_ - ic_)) & gm;
                const uint64_t same = (__ballot(!live || mybase == ref) >> (lane_ - ic_)) & gm;
                odd |= lv != 0 && !(((lv >> (cpp - 1u)) & 1ull) && same == gm && ref % cpp == 0u);
            }
        }
        if (!__syncthreads_or(odd ? 1 : 0)) {
            for (uint32_t base = 0; base < nslots; base += NT * U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t s = base + (uint32_t)u * NT + tid;
                    if (s >= nslots) continue;
                    const uint32_t h = s_hd[s], m = h & 0xffffu;
                    if ((h >> 16) && m < W) atomicMax(&s_pix[m], (int32_t)s);
                    if (s_pkm && ic_ == 0 && m / cpp < npo) atomicMax(&s_pkm[m / cpp], (int32_t)(s / cpp));
                }
            }
            __syncthreads();
            for (uint32_t c = tid; c < W; c += NT) s_hdr[c] = s_pix[c];
            __syncthreads();
            RSTAMP(13);
            return 0u;
        }
    }
    // ---- A: which path does the reference take for a packet, and what does the packet do to next_valid?
    //      s_pkt[2p] = block path: F | 1 << 31 (F = m_id of column 0); column path: M = max over live columns of m_id + 1
    // One lane per slot where a packet's columns are a power-of-two group of lanes (every sensor: 16): consecutive lanes
    // read consecutive words and update consecutive counters.  (One thread per packet -- the fallback below -- walks 16 words
    // 16 words apart from its neighbour's: 16-way bank conflicts on every read and every LDS atomic.)  Eight slots per
    // thread at a time: the shuffle chains of the eight are independent and overlap.
    const bool lanes = cpp <= 64 && (cpp & (cpp - 1)) == 0 && BD != 0;
    const uint32_t lane = tid & 63u, ic_l = lane & (cpp - 1u);
    const uint64_t grp_mask = cpp >= 64 ? ~0ull : ((1ull << cpp) - 1ull);
    if (lanes) {
        for (uint32_t base = 0; base < nslots; base += NT * U) {
            uint32_t m[U], top[U];
            uint64_t lv[U], ft[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = base + (uint32_t)u * NT + tid;
                const bool in = s < nslots;
                const uint32_t h = in ? s_hd[s] : 0u;
                m[u] = h & 0xffffu;
                const bool live = in && (h >> 16) && m[u] < W;
                lv[u] = __ballot(live) >> (lane - ic_l);
                ft[u] = __ballot(!in || (ic_l % BD) != 0 || m[u] + BD <= W) >> (lane - ic_l);
                top[u] = live ? m[u] + 1u : 0u;
            }
            for (uint32_t d = 1; d < cpp; d <<= 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) top[u] = max(top[u], (uint32_t)__shfl_xor((int)top[u], (int)d));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = base + (uint32_t)u * NT + tid;
                if (s < nslots && ic_l == 0) {
                    const uint32_t p = s / cpp;
                    const bool block = (lv[u] & grp_mask) == grp_mask && (ft[u] & grp_mask) == grp_mask;
                    s_pkt[2 * p] = block ? (m[u] | 0x80000000u) : top[u];
                    if (s_pkm && m[u] / cpp < npo) atomicMax(&s_pkm[m[u] / cpp], (int32_t)p);
                }
            }
        }
    } else
    for (uint32_t p = tid; p < count; p += NT) {
        bool allv = true, fit = BD != 0;
        uint32_t first = 0, top = 0;
        for (uint32_t ic = 0; ic < cpp; ++ic) {
            const uint32_t h = s_hd[p * cpp + ic], m = h & 0xffffu;
            if (ic == 0) first = m;
            const bool live = (h >> 16) && m < W;
            allv &= live;
            if (live) top = max(top, m + 1u);
            if (BD && ic % BD == 0) fit &= m + BD <= W;
        }
        s_pkt[2 * p] = (allv && fit) ? (first | 0x80000000u) : top;
        if (s_pkm && first / cpp < npo) atomicMax(&s_pkm[first / cpp], (int32_t)p);
    }
    __syncthreads();
    RSTAMP(10);
    // ---- B: next_valid before every packet.  A block packet moves it to F + cpp when F >= next_valid, a column-path packet
    // to max(next_valid, M).  Guess: the running maximum of those targets (an exclusive prefix maximum, a parallel scan).
    // The guess is exact when every block packet either passes its test under the guess or, failing it, has a target the
    // maximum already covers -- in order, with drops, in any order, with duplicates: always, for packets whose ids are
    // multiples of cpp.  Anything else walks the packets one by one (wave 0, on registers).
    __shared__ uint32_t s_nv_final, s_wmax[NT / 64];
    bool exact = true;
    {
        uint32_t carry = 0;
        for (uint32_t base = 0; base < count; base += NT) {
            const uint32_t p = base + tid;
            const uint32_t f0 = p < count ? s_pkt[2 * p] : 0u;
            const bool blk = (f0 & 0x80000000u) != 0u;
            const uint32_t first = f0 & 0x7fffffffu, target = blk ? first + cpp : first;
            uint32_t inc = target;   // inclusive prefix maximum inside my wave
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)inc, d);
                if (lane >= d) inc = max(inc, y);
            }
            if (lane == 63) s_wmax[tid >> 6] = inc;
            __syncthreads();
            uint32_t before = carry;
            for (uint32_t w = 0; w < (tid >> 6); ++w) before = max(before, s_wmax[w]);
            const uint32_t up = (uint32_t)__shfl_up((int)inc, 1u);
            const uint32_t nv_in = max(before, lane ? up : 0u);
            if (p < count) {
                if (blk && first < nv_in && target > nv_in) exact = false;
                s_pkt[2 * p + 1] = nv_in;
            }
            uint32_t all = carry;
            for (uint32_t w = 0; w < NT / 64; ++w) all = max(all, s_wmax[w]);
            carry = all;
            __syncthreads();
        }
        if (tid == 0) s_nv_final = carry;
    }
    if (!__syncthreads_and(exact ? 1 : 0)) {
        if (tid < 64) {
            uint32_t nv = 0;
            for (uint32_t base = 0; base < count; base += 64) {
                const uint32_t p = base + tid;
                const uint32_t a0 = p < count ? s_pkt[2 * p] : 0u;
                uint32_t mine = 0;
                const uint32_t n = min(64u, count - base);
                for (uint32_t k = 0; k < n; ++k) {
                    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)a0, (int)k);
                    if (tid == k) mine = nv;
                    if (f0 & 0x80000000u) {
                        const uint32_t first = f0 & 0x7fffffffu;
                        if (first >= nv) nv = first + cpp;
                    } else {
                        nv = max(nv, f0);
                    }
                }
                if (p < count) s_pkt[2 * p + 1] = mine;
            }
            if (tid == 0) s_nv_final = nv;
        }
        __syncthreads();
    }
    // s_pkt[2p] bit 31: block path; s_pkt[2p + 1] = next_valid before packet p
    RSTAMP(11);
    // ---- C: the zeroed ranges a packet triggers and the columns it writes (one lane per slot, or one thread per packet)
    if (lanes) {
        for (uint32_t base = 0; base < nslots; base += NT * U) {
            uint32_t m[U], nvb[U], bf[U];
            uint32_t flags[U];   // 1 in, 2 live, 4 block path
            bool anycol = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t s = base + (uint32_t)u * NT + tid;
                const bool in = s < nslots;
                const uint32_t h = in ? s_hd[s] : 0u, p = s / cpp;
                m[u] = h & 0xffffu;
                const bool live = in && (h >> 16) && m[u] < W;
                const bool block = in && (s_pkt[2 * p] & 0x80000000u) != 0u;
                nvb[u] = in ? s_pkt[2 * p + 1] : 0u;
                flags[u] = (in ? 1u : 0u) | (live ? 2u : 0u) | (block ? 4u : 0u);
                anycol |= __ballot(in && !block) != 0ull;
                bf[u] = (uint32_t)__shfl((int)m[u], (int)(lane - ic_l % BD));
            }
            if (anycol) {
                // column path: next_valid before my column = max(before the packet, the live columns before me in it)
                uint32_t pre[U];
#pragma unroll
                for (int u = 0; u < U; ++u) pre[u] = (flags[u] & 2u) ? m[u] + 1u : 0u;
                for (uint32_t d = 1; d < cpp; d <<= 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t y = (uint32_t)__shfl_up((int)pre[u], d, (int)cpp);
                        if (ic_l >= d) pre[u] = max(pre[u], y);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t excl = (uint32_t)__shfl_up((int)pre[u], 1u, (int)cpp);
                    if (!(flags[u] & 4u)) nvb[u] = max(nvb[u], ic_l ? excl : 0u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t s = (int32_t)(base + (uint32_t)u * NT + tid);
                if (flags[u] & 4u) {
                    if (ic_l == 0 && m[u] >= nvb[u])
                        for (uint32_t c = nvb[u]; c < m[u]; ++c) s_z[c] = s;
                    atomicMax(&s_hdr[m[u]], s);
                    atomicMax(&s_pix[bf[u] + ic_l % BD], s);
                } else if (flags[u] & 2u) {
                    if (m[u] >= nvb[u])
                        for (uint32_t c = nvb[u]; c < m[u]; ++c) s_z[c] = s;
                    atomicMax(&s_hdr[m[u]], s);
                    atomicMax(&s_pix[m[u]], s);
                }
            }
        }
    } else
    for (uint32_t p = tid; p < count; p += NT) {
        const bool block = (s_pkt[2 * p] & 0x80000000u) != 0u;
        uint32_t nv = s_pkt[2 * p + 1];
        uint32_t block_first = 0;
        for (uint32_t ic = 0; ic < cpp; ++ic) {
            const uint32_t h = s_hd[p * cpp + ic], m = h & 0xffffu;
            const bool v = (h >> 16) != 0u;
            const int32_t slot = (int32_t)(p * cpp + ic);
            if (block) {
                if (ic == 0 && m >= nv)
                    for (uint32_t c = nv; c < m; ++c) s_z[c] = slot;
                if (ic % BD == 0) block_first = m;
                atomicMax(&s_hdr[m], slot);
                atomicMax(&s_pix[block_first + ic % BD], slot);
            } else if (v && m < W) {
                if (m >= nv) {
                    for (uint32_t c = nv; c < m; ++c) s_z[c] = slot;
                    nv = m + 1u;
                }
                atomicMax(&s_hdr[m], slot);
                atomicMax(&s_pix[m], slot);
            }
        }
    }
    __syncthreads();
    RSTAMP(12);
    const uint32_t nv_final = s_nv_final;
    for (uint32_t c = tid; c < W; c += NT) {
        const int32_t z = s_z[c];
        if (s_hdr[c] < z) s_hdr[c] = -1;
        if (s_pix[c] < z || c >= nv_final) s_pix[c] = -1;
    }
    __syncthreads();
    RSTAMP(13);
    return nv_final;
}

// The per-column poses of a tile, cast to the xyz element type, in LDS (12 values per column); all threads of the
// workgroup call it, a barrier follows.  Returns the table or nullptr when no poses were given.
// POSES is a template parameter of the kernels: the pose arithmetic keeps 48 more registers alive, and a kernel that
// merely CONTAINS it is allocated for it (the 12 B/px profile went from 155 to 171 VGPRs = from three waves per SIMD to two).
template <int XYZM, bool POSES>
__device__ __forceinline__ const void* stage_poses(const DecodeArgs& a, uint32_t* smem, uint32_t f, uint32_t c0,
                                                   uint32_t ncols) {
    if constexpr (!POSES || (XYZM != 1 && XYZM != 2)) {
        return nullptr;
    } else {
        if (!a.xyz_poses) return nullptr;
        using XT = typename std::conditional<XYZM == 1, float, double>::type;
        XT* s_pose = (XT*)((uint8_t*)smem + a.pose_lds_off);
        const uint32_t W = a.g.columns_per_frame;
        // transposed: element k of every column next to each other, so that a lane's four columns are one 16 / 32 B
        // read and a wave's reads are conflict free (column-major rows of 12 cost 0.4 ms per launch in bank conflicts)
        // at most 64 columns here (k_decode's tiles): three values per thread
        for (uint32_t i = threadIdx.x; i < ncols * 12; i += blockDim.x) {
            const uint32_t j = i / 12, k = i - j * 12, c = c0 + j;
#ifdef OUSTER_ABLATE_POSE_LOAD   // experiment builds only: no pose is read
            s_pose[k * ncols + j] = (XT)(k % 5 == 0);
#else
            s_pose[k * ncols + j] = c < W ? (XT)a.xyz_poses[((size_t)f * W + c) * 16 + k] : (XT)0;
#endif
        }
        __syncthreads();
        return s_pose;
    }
}

// ------------------------------------------------------------------------------------
// decode_rows: the pixel phase shared by k_decode and k_decode_wide.
// The workgroup's tile (QPR*4 columns x nrows rows, rows r0.. of frame f, columns c0..) sits in LDS,
// column j at dword  col0_dw + j*colstride_dw, its rows `S::chan` (or a.g.channel_data_size) bytes
// apart.  lane = (row within pass, quad of 4 consecutive columns): every global store is a vector
// store of 4 consecutive elements of a plane row (16 B for u32, 8 B for u16, 4 B for u8).
//   field decode   FieldDecodeInfo::get, field_decode_info.h:41-54 (compile-time masks for the
//                  standard profiles, run-time descriptors for SpecGeneric = add_custom_profile)
//   destagger      destagger_into<T>, impl/lidar_frame_impl.h:733-760: the same 4 values go to column
//                  (col + offset[row]) mod W of the destaggered plane
//   cartesian      cartesianT<T>, impl/cartesian.h:36-66 on the separable tables (XYZM 1/2) or the
//                  full LUT (XYZM 3); f32 xyz leaves through the wave-private LDS transpose
// vq: bit c set = my column c holds a received, valid column (else zeros / f16 NaN are written).
// ------------------------------------------------------------------------------------
// s_beam: the tile rows' per-beam constants [nrows][9] in LDS, or nullptr to read them from the table in
// global memory.  LDS matters far beyond the bytes: gfx950 counts loads and stores in ONE counter
// (vmcnt) and they complete out of order with respect to each other, so a global load inside the row
// loop makes the compiler wait `vmcnt(0)` -- for the load AND for every store the wave has in flight.
// With the table in LDS (lgkmcnt) the loop contains no vector load and the stores of successive rows
// stream without ever being waited for.
// s_gate: [QPR*4] LDS counters, zeroed by the caller before a barrier, or nullptr.  When set, the lanes
// count the pixels of their columns that pass the range gate (a.gate_*), the counts are reduced in LDS
// and the tile's partial counts go to a.gate_counts[f][gate_chunk][c0..] -- the dewarp that follows
// skips its counting pass.  Every thread of the workgroup must call decode_rows (it ends in a barrier
// when s_gate is set).
// The per-column constants of a lane's four columns (separable xyz tables, section 3.3): loaded by decode_rows
// itself, or once per kernel by a persistent caller (k_decode_stream) and handed in.
struct ColConst {
    double cx[4], sx[4], kc[4][3];
};
__device__ __forceinline__ void load_colconst(ColConst& cc, const LutDev& lut, uint32_t col, uint32_t W) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t k = (col + c < W) ? col + c : W - 1;
        const double* t = lut.col_tab + (size_t)k * 5;
        cc.cx[c] = t[0]; cc.sx[c] = t[1]; cc.kc[c][0] = t[2]; cc.kc[c][1] = t[3]; cc.kc[c][2] = t[4];
    }
}

// NT: threads of the workgroup.  px_dw: per-lane dword offsets (into s_tile) of row 0 of the lane's four columns
// (tile_px_offsets for "column j at col0_dw + j*colstride_dw", k_decode_stream has permuted blocks).  cc: the lane's
// column constants (load_colconst; XYZM 1 / 2 only) -- a persistent caller loads them once.
template <int QPR>
__device__ __forceinline__ void tile_px_offsets(uint32_t (&px_dw)[4], uint32_t col0_dw, uint32_t colstride_dw) {
    const uint32_t jq = (threadIdx.x % QPR) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) px_dw[c] = col0_dw + (jq + c) * colstride_dw;
}
// POSEREG: keep the f32 poses of the lane's four columns in registers across the row loop (k_decode_wide; k_decode's
// 16-lane row segments read them as LDS broadcasts cheaply enough, and its fix-up instantiation has no registers to spare).
// BEAMLDS: s_beam is known to be an LDS table (no run-time choice between it and lut.beam_tab: a pointer select would
// turn the table reads into flat loads -- vmcnt AND lgkmcnt -- inside the row loop).
// VECONLY: the caller guarantees a.vec_ok and whole quads (W % 4 == 0, every lane's four columns exist): the
// element-wise fallbacks are not compiled in (a destaggered run that wraps round the row end still is).
template <class S, int QPR, int XYZM, bool DEADZ = false, bool NTS = false, bool NTX = NTS, bool POSES = false, int NT = 256,
          bool BEAMLDS = false, bool VECONLY = false, bool POSEREG = false>
__device__ __forceinline__ void decode_rows(const DecodeArgs& a, const uint32_t* s_tile,
                                            const uint32_t (&px_dw)[4], const ColConst& cc, const int32_t* s_off,
                                            float4* s_xyz, const double* s_beam, uint32_t* s_gate,
                                            const LutDev& lut, uint32_t f, uint32_t c0, uint32_t r0,
                                            uint32_t nrows, uint32_t vq, uint32_t gate_chunk,
                                            uint32_t gate_nchunks, const void* s_pose = nullptr) {
    constexpr int LPR = QPR < 64 ? QPR : 64;             // lanes of one wave in a row segment
    constexpr int RPP = NT / QPR > 0 ? NT / QPR : 1;     // rows per pass of the workgroup
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, H = a.g.pixels_per_column;
    const uint32_t q = tid % QPR, ty = tid / QPR;
    const uint32_t jq = q * 4, col = c0 + jq;
    const bool live = VECONLY || col < W;
    if (!live && !s_gate) return;
    uint32_t gcnt[4] = {0, 0, 0, 0};
    const bool vec = VECONLY || (a.vec_ok && (col + 3 < W));
    const uint32_t ncol = VECONLY ? 4u : ((W - col) < 4 ? (W - col) : 4);  // < 4 only when W % 4 != 0
    const size_t plane_px = (size_t)H * W;
    const uint32_t ql = q % LPR;                 // lane position inside its wave's row segment
    const uint32_t seg0 = c0 + (q - ql) * 4;     // first column of that segment
    const uint32_t chan = S::is_static ? S::chan : a.g.channel_data_size;


    // The poses of my four columns (f32: 12 x 4 values = 48 registers) are read from the LDS table ONCE, before the row
    // loop: left inside it they are re-read for every row and return -- 384 B per lane and row, 64 different quads per
    // wave, which cost k_decode_wide 0.19 ms per launch (the 64-column kernel, whose 16-lane row segments read the same
    // addresses four times over as broadcasts, paid 0.03).  The f64 table (96 registers) stays in LDS and is read once per
    // row for both returns.
    using PXT = typename std::conditional<XYZM == 2, double, float>::type;
    constexpr bool POSE_REGS = POSES && POSEREG && XYZM == 1;
    PXT m_pose[POSE_REGS ? 12 : 1][4];
    if constexpr (POSE_REGS) {
        if (s_pose && live) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) m_pose[k][c] = ((const PXT*)s_pose)[(size_t)k * (QPR * 4) + jq + c];
        }
    }

    for (uint32_t rrel = ty; live && rrel < nrows; rrel += RPP) {
        const uint32_t r = r0 + rrel;
        const size_t rowpix = (size_t)r * W + col;  // pixel index of my first column
        uint32_t doff = 0;                          // destaggered column of my first column
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
                const uint32_t* px = s_tile + px_dw[c] + rrel * CW;
#pragma unroll
                for (int k = 0; k < CW; ++k) w[c][k] = (DEADZ || ((vq >> c) & 1)) ? px[k] : 0u;
            }
            auto do_field = [&](auto kc_) {
                constexpr int K = decltype(kc_)::value;
                const int di = a.desc_of_spec[K];
                if (di < 0) return;
                constexpr uint32_t e = S::f[K].elem;
                static_assert(e <= 4, "static profiles carry fields of at most 4 bytes");
                u32x4_t v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = (uint32_t)extract_static<S, K, CW>(w[c]);
                if (K == S::range_idx) { rng[0][0] = v[0]; rng[0][1] = v[1]; rng[0][2] = v[2]; rng[0][3] = v[3]; }
                if (K == S::range2_idx) { rng[1][0] = v[0]; rng[1][1] = v[1]; rng[1][2] = v[2]; rng[1][3] = v[3]; }
                if (s_gate && di == a.gate_field) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        gcnt[c] += (v[c] >= a.gate_min && v[c] <= a.gate_max) ? 1u : 0u;
                }
                uint8_t* pl = (uint8_t*)a.planes[di];
                if (pl) {
                    uint8_t* d = pl + ((size_t)f * plane_px + rowpix) * e;
                    if (vec) store4v<NTS || OUSTER_NT_PLANES>(d, v, e);
                    else for (uint32_t c = 0; c < ncol; ++c) store1(d + c * e, v[c], e);
                }
                uint8_t* dp = (uint8_t*)a.destaggered[di];
                if (dp) {
                    uint8_t* drow = dp + ((size_t)f * plane_px + (size_t)r * W) * e;
                    if (dvec) store4v<NTS || OUSTER_NT_PLANES>(drow + (size_t)doff * e, v, e);
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
                const bool want_gate = s_gate && (int)i == a.gate_field;
                uint8_t* pl = (uint8_t*)a.planes[i];
                uint8_t* dp = (uint8_t*)a.destaggered[i];
                if (!pl && !dp && !want_xyz && !want_gate) continue;
                const uint32_t e = a.elem[i];
                uint64_t v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t bo = (px_dw[c] << 2) + rrel * chan + a.bits[i].offset;
                    v[c] = ((vq >> c) & 1)
                               ? trunc_elem(apply_bits(window_lds(s_tile, bo), a.bits[i].mask, a.bits[i].shift), e)
                               : trunc_elem(zero_value((a.f16_nan_mask >> i) & 1u), e);
                }
                if ((int)i == a.xyz_field[0]) { rng[0][0] = v[0]; rng[0][1] = v[1]; rng[0][2] = v[2]; rng[0][3] = v[3]; }
                if ((int)i == a.xyz_field[1]) { rng[1][0] = v[0]; rng[1][1] = v[1]; rng[1][2] = v[2]; rng[1][3] = v[3]; }
                if (want_gate) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        gcnt[c] += ((uint32_t)v[c] >= a.gate_min && (uint32_t)v[c] <= a.gate_max) ? 1u : 0u;
                }
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
            const double* b;
            if constexpr (BEAMLDS) b = s_beam + rrel * 9u;
            else b = s_beam ? s_beam + (size_t)rrel * 9 : lut.beam_tab + (size_t)r * 9;
            const double u0 = b[0], u1 = b[1], u2 = b[2], v0 = b[3], v1 = b[4], v2 = b[5],
                         w0 = b[6], w1 = b[7], w2 = b[8];
            double d[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d[c][0] = fma(cc.cx[c], u0, fma(cc.sx[c], v0, w0));
                d[c][1] = fma(cc.cx[c], u1, fma(cc.sx[c], v1, w1));
                d[c][2] = fma(cc.cx[c], u2, fma(cc.sx[c], v2, w2));
            }
#pragma unroll
            for (int ret = 0; ret < 2; ++ret) {
                XT* out = (XT*)a.xyz[ret];
                if (!out) continue;
                XT p[4][3];   // rounded to the output type before the zero-range select (same value, half the selects)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t rr = rng[ret][c];
                    const double rm = (double)rr - lut.n;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
#ifdef OUSTER_ABLATE_XYZ_MATH   // experiment builds only (tools/ab/ablate.sh): how much of the kernel's time is the arithmetic?
                        const XT t = (XT)rr;
                        (void)rm;
#else
                        const XT t = (XT)fma(rm, d[c][k], cc.kc[c][k]);
#endif
                        p[c][k] = rr ? t : (XT)0;
                    }
                }
                if constexpr (POSES) if (s_pose) {   // dewarp<T>(points, poses), pose_util.h:38-56: R_col * p + t_col in T
                    XT m[12][4];
#pragma unroll
                    for (int k = 0; k < 12; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if constexpr (POSE_REGS) m[k][c] = m_pose[k][c];
                            else m[k][c] = ((const XT*)s_pose)[(size_t)k * (QPR * 4) + jq + c];
                        }
#ifdef OUSTER_ABLATE_POSE_MATH   // experiment builds only: the table stays in registers, the arithmetic goes
#pragma unroll
                    for (int k = 0; k < 12; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(m[k][c]));
#else
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const XT x = p[c][0], y = p[c][1], z = p[c][2];
                        p[c][0] = m[0][c] * x + m[1][c] * y + m[2][c] * z + m[3][c];
                        p[c][1] = m[4][c] * x + m[5][c] * y + m[6][c] * z + m[7][c];
                        p[c][2] = m[8][c] * x + m[9][c] * y + m[10][c] * z + m[11][c];
                    }
#endif
                }
                XT* dst = out + ((size_t)f * plane_px + rowpix) * 3;
                if constexpr (XYZM == 1) {
                    if (VECONLY || (a.vec_ok && seg0 + 4 * LPR <= W)) {  // my wave's whole row segment exists
#if OUSTER_XYZ_PERMUTE
                        store_xyz4_permuted<LPR, NTX || OUSTER_NT_XYZ>(dst - (size_t)(4 * ql) * 3, ql, p);
#else
                        store_xyz4_coalesced<LPR>(s_xyz, tid, dst - (size_t)(4 * ql) * 3, ql, p);
#endif
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
    if (s_gate) {   // uniform over the workgroup
        if (live) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (gcnt[c] && col + c < W) atomicAdd(&s_gate[jq + c], gcnt[c]);
        }
        __syncthreads();
        uint16_t* dst = a.gate_counts + ((size_t)f * OUSTER_HIP_GATE_CHUNKS + gate_chunk) * W;
        for (uint32_t j = tid; j < (uint32_t)QPR * 4 && c0 + j < W; j += NT) {
            dst[c0 + j] = (uint16_t)s_gate[j];
            if (gate_chunk == 0)   // the chunk slots this launch does not use read as zero
                for (uint32_t k = gate_nchunks; k < OUSTER_HIP_GATE_CHUNKS; ++k) dst[(size_t)k * W + c0 + j] = 0;
        }
    }
}

}  // namespace ouster_hip_dev
