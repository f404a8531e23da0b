// storebench: the store set of k_decode<SpecDualLB,64,xyz f32> without loads or math, to study how
// plane placement and store order change the achieved write rate.  256 frames of 128x2048:
//   planes   RANGE u32, FLAGS u8, REFL u8, NIR u16, RANGE2 u32, FLAGS2 u8, REFL2 u8, WINDOW u8   (15 B/px)
//   destag.  RANGE, RANGE2 (u32), REFL, REFL2 (u8), rows shifted by {24, 8, -8, -24} columns      (10 B/px)
//   xyz      2 x 12 B/px, written as 3 x 16 B per lane through the coalesced layout                (24 B/px)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/storebench tools/storebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

struct Args {
    uint8_t* p[14];   // 0..7 planes, 8..11 destaggered, 12..13 xyz
    uint32_t frames;
    int variant;
    uint32_t fmod;   // writes go to frame f % fmod (footprint control)
    int rowmode;     // 0: r = ty + 16k; 1: rotated by tile; 2: rotated by tile and frame; 3: r = ty*8 + k
};
__device__ __constant__ int kShift[4] = {24, 8, -8, -24};
constexpr int ELEM[12] = {4, 1, 1, 2, 4, 1, 1, 1, 4, 4, 1, 1};

template <int E>
__device__ __forceinline__ void st(uint8_t* p, uint32_t v) {
    if (E == 4) { uint4 q = {v, v + 1, v + 2, v + 3}; *(uint4*)p = q; }
    else if (E == 2) { uint2 q = {v, v + 1}; *(uint2*)p = q; }
    else *(uint32_t*)p = v;
}
struct __attribute__((packed, aligned(1))) pk16 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) pk4 { uint32_t a; };

__global__ __launch_bounds__(256) void k_store(Args a) {
    constexpr uint32_t W = 2048, H = 128;
    const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const uint32_t f = (i / 32) * 8u + xcd, tile = i % 32;
    if (f >= a.frames) return;
    const uint32_t q = threadIdx.x % 16, ty = threadIdx.x / 16, col = tile * 64 + 4 * q;
    const size_t fpx = (size_t)(f % a.fmod) * H * W;
    for (uint32_t rl = ty; rl < H; rl += 16) {
        uint32_t r = rl;
        if (a.rowmode == 1) r = (rl + tile * 20) % H;
        else if (a.rowmode == 2) r = (rl + tile * 20 + f * 52) % H;
        else if (a.rowmode == 3) r = ty * 8 + rl / 16;
        else if (a.rowmode == 4) r = (rl + (tile & 7) * 16) % H;
        const size_t px = fpx + (size_t)r * W + col;
        const uint32_t v = (uint32_t)px;
        uint32_t dc = (col + W + kShift[r & 3]) % W;
        const size_t dpx = fpx + (size_t)r * W + dc;
        if (a.variant == 0) {  // k_decode order: plane k, then its destaggered copy
            st<4>(a.p[0] + px * 4, v); *(pk16*)(a.p[8] + dpx * 4) = pk16{v, v, v, v};
            st<1>(a.p[1] + px, v);
            st<1>(a.p[2] + px, v); *(pk4*)(a.p[10] + dpx) = pk4{v};
            st<2>(a.p[3] + px * 2, v);
            st<4>(a.p[4] + px * 4, v); *(pk16*)(a.p[9] + dpx * 4) = pk16{v, v, v, v};
            st<1>(a.p[5] + px, v);
            st<1>(a.p[6] + px, v); *(pk4*)(a.p[11] + dpx) = pk4{v};
            st<1>(a.p[7] + px, v);
        } else if (a.variant == 1) {  // planes only
            st<4>(a.p[0] + px * 4, v); st<1>(a.p[1] + px, v); st<1>(a.p[2] + px, v); st<2>(a.p[3] + px * 2, v);
            st<4>(a.p[4] + px * 4, v); st<1>(a.p[5] + px, v); st<1>(a.p[6] + px, v); st<1>(a.p[7] + px, v);
        } else if (a.variant == 2) {  // planes + xyz only
            st<4>(a.p[0] + px * 4, v); st<1>(a.p[1] + px, v); st<1>(a.p[2] + px, v); st<2>(a.p[3] + px * 2, v);
            st<4>(a.p[4] + px * 4, v); st<1>(a.p[5] + px, v); st<1>(a.p[6] + px, v); st<1>(a.p[7] + px, v);
        }
        if (a.variant != 1) {
#pragma unroll
            for (int ret = 0; ret < 2; ++ret) {
                uint4* d = (uint4*)(a.p[12 + ret] + (fpx + (size_t)r * W + tile * 64) * 12);
#pragma unroll
                for (int k = 0; k < 3; ++k) d[k * 16 + q] = uint4{v, v, v, v};
            }
        }
    }
}

// wide tiles: a workgroup owns TW columns x 8192/TW rows
template <int TW>
__global__ __launch_bounds__(256) void k_store_wide(Args a) {
    constexpr uint32_t W = 2048, H = 128, TR = 8192 / TW, TPF = (W / TW) * (H / TR);  // 32 tiles per frame
    constexpr uint32_t QPR = TW / 4, RPP = 256 / QPR;  // quads per row, rows per pass
    const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const uint32_t f = (i / TPF) * 8u + xcd, sub = i % TPF;
    if (f >= a.frames) return;
    const uint32_t tile = sub % (W / TW), rc = sub / (W / TW);
    const uint32_t q = threadIdx.x % QPR, ty = threadIdx.x / QPR, col = tile * TW + 4 * q;
    const size_t fpx = (size_t)f * H * W;
    for (uint32_t k = 0; k < TR / RPP; ++k) {
        const uint32_t r = rc * TR + k * RPP + ty;
        const size_t px = fpx + (size_t)r * W + col;
        const uint32_t v = (uint32_t)px;
        uint32_t dc = (col + W + kShift[r & 3]) % W;
        const size_t dpx = fpx + (size_t)r * W + dc;
        st<4>(a.p[0] + px * 4, v); *(pk16*)(a.p[8] + dpx * 4) = pk16{v, v, v, v};
        st<1>(a.p[1] + px, v);
        st<1>(a.p[2] + px, v); *(pk4*)(a.p[10] + dpx) = pk4{v};
        st<2>(a.p[3] + px * 2, v);
        st<4>(a.p[4] + px * 4, v); *(pk16*)(a.p[9] + dpx * 4) = pk16{v, v, v, v};
        st<1>(a.p[5] + px, v);
        st<1>(a.p[6] + px, v); *(pk4*)(a.p[11] + dpx) = pk4{v};
        st<1>(a.p[7] + px, v);
#pragma unroll
        for (int ret = 0; ret < 2; ++ret) {
            uint4* d = (uint4*)(a.p[12 + ret] + (fpx + (size_t)r * W + tile * TW) * 12);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) d[kk * QPR + q] = uint4{v, v, v, v};
        }
    }
}

// plane-major: the workgroup writes all its rows of one plane before moving to the next plane
template <int TW>
__global__ __launch_bounds__(256) void k_store_wide_pm(Args a) {
    constexpr uint32_t W = 2048, H = 128, TR = 8192 / TW, TPF = (W / TW) * (H / TR);
    constexpr uint32_t QPR = TW / 4, RPP = 256 / QPR;
    const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const uint32_t f = (i / TPF) * 8u + xcd, sub = i % TPF;
    if (f >= a.frames) return;
    const uint32_t tile = sub % (W / TW), rc = sub / (W / TW);
    const uint32_t q = threadIdx.x % QPR, ty = threadIdx.x / QPR, col = tile * TW + 4 * q;
    const size_t fpx = (size_t)f * H * W;
    auto rowpx = [&](uint32_t k, size_t& px, size_t& dpx, uint32_t& r) {
        r = rc * TR + k * RPP + ty;
        px = fpx + (size_t)r * W + col;
        uint32_t dc = (col + W + kShift[r & 3]) % W;
        dpx = fpx + (size_t)r * W + dc;
    };
    size_t px, dpx; uint32_t r;
#define PLANE(stmt) for (uint32_t k = 0; k < TR / RPP; ++k) { rowpx(k, px, dpx, r); const uint32_t v = (uint32_t)px; stmt; }
    PLANE(st<4>(a.p[0] + px * 4, v))
    PLANE(*(pk16*)(a.p[8] + dpx * 4) = (pk16{v, v, v, v}))
    PLANE(st<1>(a.p[1] + px, v))
    PLANE(st<1>(a.p[2] + px, v))
    PLANE(*(pk4*)(a.p[10] + dpx) = pk4{v})
    PLANE(st<2>(a.p[3] + px * 2, v))
    PLANE(st<4>(a.p[4] + px * 4, v))
    PLANE(*(pk16*)(a.p[9] + dpx * 4) = (pk16{v, v, v, v}))
    PLANE(st<1>(a.p[5] + px, v))
    PLANE(st<1>(a.p[6] + px, v))
    PLANE(*(pk4*)(a.p[11] + dpx) = pk4{v})
    PLANE(st<1>(a.p[7] + px, v))
    for (int ret = 0; ret < 2; ++ret)
        for (uint32_t k = 0; k < TR / RPP; ++k) {
            rowpx(k, px, dpx, r);
            const uint32_t v = (uint32_t)px;
            uint4* d = (uint4*)(a.p[12 + ret] + (fpx + (size_t)r * W + tile * TW) * 12);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) d[kk * QPR + q] = uint4{v, v, v, v};
        }
#undef PLANE
}

int main(int argc, char** argv) {
    const uint32_t F = 256;
    const size_t npx = (size_t)F * 128 * 2048;
    size_t sizes[14];
    for (int k = 0; k < 12; ++k) sizes[k] = npx * ELEM[k];
    sizes[12] = sizes[13] = npx * 12;
    const size_t M = 1 << 20;
    uint8_t* slab;
    size_t total = 0;
    for (size_t s : sizes) total += s;
    hipMalloc(&slab, total + 1600 * M);
    std::vector<uint8_t*> sep(14);
    for (int k = 0; k < 14; ++k) hipMalloc(&sep[k], sizes[k]);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes[3] = {npx * 49.0, npx * 15.0, npx * 39.0};
    struct Lay { const char* name; long gap; long kgap; };
    const Lay lays[] = {{"separate hipMalloc", -1, 0}, {"slab packed", 0, 0}, {"slab gap 33M", 33 * M, 0}};
    for (const Lay& L : lays) {
        Args a{};
        a.frames = F;
        size_t o = 0;
        for (int k = 0; k < 14; ++k) {
            if (L.gap < 0) a.p[k] = sep[k];
            else {
                o += (size_t)L.gap + (size_t)L.kgap * k;
                o = (o + 255) & ~(size_t)255;
                a.p[k] = slab + o;
                o += sizes[k];
            }
        }
        printf("%-22s", L.name);
        for (int rm : {0, 128, 256, 1128, 1256}) {
            a.variant = 0; a.fmod = 256; a.rowmode = rm; const int fm = rm;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (rm == 128) hipLaunchKernelGGL(k_store_wide<128>, dim3(32 * F), dim3(256), 0, 0, a);
                else if (rm == 256) hipLaunchKernelGGL(k_store_wide<256>, dim3(32 * F), dim3(256), 0, 0, a);
                else if (rm == 512) hipLaunchKernelGGL(k_store_wide<512>, dim3(32 * F), dim3(256), 0, 0, a);
                else if (rm == 1024) hipLaunchKernelGGL(k_store_wide<1024>, dim3(32 * F), dim3(256), 0, 0, a);
                else if (rm == 1128) hipLaunchKernelGGL(k_store_wide_pm<128>, dim3(32 * F), dim3(256), 0, 0, a);
                else if (rm == 1256) hipLaunchKernelGGL(k_store_wide_pm<256>, dim3(32 * F), dim3(256), 0, 0, a);
                else hipLaunchKernelGGL(k_store, dim3(32 * F), dim3(256), 0, 0, a);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf(" | TW %4d %6.3f", fm, best);
        }
        a.fmod = 256;
        for (int variant = 1; variant < 1; ++variant) {
            a.variant = variant;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_store, dim3(32 * F), dim3(256), 0, 0, a);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf(" | v%d %6.3f ms %6.0f GB/s", variant, best, bytes[variant] / best / 1e6);
        }
        printf("\n");
    }
    return 0;
}
