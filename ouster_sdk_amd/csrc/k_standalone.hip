// k_standalone.hip -- the standalone kernels of the hot path (gfx950) and the launch dispatch.
//
//   k_destagger           per-row circular shift
//   k_cartesian(_tiled)   range image -> XYZ
//   k_dewarp(_tiled)      per-column pose applied to a dense point cloud
//   k_dwf_*               range-gated, compacting frame dewarp (count, scans, emit)
// The fused decode kernels live in k_decode.hip (one object per profile specialisation).
//
// Reference loops:
//   destagger_into<T>                        ouster_core/include/ouster/core/impl/lidar_frame_impl.h:733-760
//   impl::make_xyz_lut / cartesianT<T>       ouster_core/src/xyzlut.cpp:11-89, impl/cartesian.h:36-66
//   dewarp<T>                                ouster_core/include/ouster/core/pose_util.h:38-56, impl/dewarp_impl.h:23-115
// Memory-bound byte/bit work: no MFMA anywhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <stdint.h>
#include <stdlib.h>

#include "kernels_common.h"

namespace ouster_hip_dev {

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

// ------------------------------------------------------------------------------------
// k_destagger: dst[img][u][(v + off[u]) % w] = src[img][u][v], element = elem bytes.
// One workgroup per (row, image).  The source row is fetched ONCE with aligned 16 B loads into LDS
// (every byte crosses the memory system -- or, for a host image worked on in place, the PCIe link --
// exactly once); lanes then walk the DESTINATION row in 16 B chunks so every store is an aligned,
// coalesced 16 B vector, and assemble each chunk from the LDS image of the row at an arbitrary byte
// offset (five dword reads and a funnel shift; the wrap point of the rotation is just an index modulo).
// Rows that do not fit the LDS budget (or are not 16 B granular) take the direct path below.
//   offset arithmetic: destagger_into, impl/lidar_frame_impl.h:753-759
// ------------------------------------------------------------------------------------
constexpr uint32_t DESTAGGER_LDS_MAX = 64u << 10;
// k_destagger_rows: the same for short rows, several consecutive rows per workgroup, software pipelined: the aligned 16 B
// loads of row j + 1 are in flight while row j is assembled from its LDS image and stored.  (Built to overlap the two
// directions of the PCIe link for a host image worked on in place; it does not -- a kernel that reads and writes host memory
// gets 1 MB in + 1 MB out in 50 us whatever its shape, tools/copybench -- but it is the faster form for 8 / 16-bit planes in HBM.)
template <int CH>
__global__ __launch_bounds__(256) void k_destagger_rows(DestaggerArgs a, uint32_t rows_per_wg) {
    extern __shared__ uint4 s_row[];   // 2 x row
    const uint32_t img = blockIdx.y;
    const size_t row_bytes = (size_t)a.w * a.elem;
    const uint32_t nchunk = (uint32_t)(row_bytes >> 4), ndw = nchunk * 4;
    const uint32_t r0 = blockIdx.x * rows_per_wg, r1 = min(a.h, r0 + rows_per_wg);
    const uint8_t* sbase = (const uint8_t*)a.src + (size_t)img * a.h * row_bytes;
    uint8_t* dbase = (uint8_t*)a.dst + (size_t)img * a.h * row_bytes;
    uint4 regs[CH];
    auto fetch = [&](uint32_t u) {
        const uint4* srow = (const uint4*)(sbase + (size_t)u * row_bytes);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const uint32_t i = threadIdx.x + k * 256;
            if (i < nchunk) regs[k] = srow[i];
        }
    };
    if (r0 < r1) fetch(r0);
    for (uint32_t u = r0; u < r1; ++u) {
        uint4* buf = s_row + ((u - r0) & 1) * nchunk;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const uint32_t i = threadIdx.x + k * 256;
            if (i < nchunk) buf[i] = regs[k];
        }
        __syncthreads();
        if (u + 1 < r1) fetch(u + 1);
        const uint32_t* s_dw = (const uint32_t*)buf;
        const uint32_t sb0 = (uint32_t)(row_bytes - (size_t)a.offsets[u] * a.elem);
        const uint32_t sh = (sb0 & 3) * 8;
        uint8_t* drow = dbase + (size_t)u * row_bytes;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const uint32_t i = threadIdx.x + k * 256;
            if (i >= nchunk) break;
            uint32_t dw = (sb0 >> 2) + i * 4;
            if (dw >= ndw) dw -= ndw;
            uint32_t d[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                uint32_t j = dw + q;
                if (j >= ndw) j -= ndw;
                d[q] = s_dw[j];
            }
            typedef uint32_t v4 __attribute__((ext_vector_type(4)));
            v4 o;
            if (sh == 0) {
                o = v4{d[0], d[1], d[2], d[3]};
            } else {
                o = v4{(d[0] >> sh) | (d[1] << (32 - sh)), (d[1] >> sh) | (d[2] << (32 - sh)),
                       (d[2] >> sh) | (d[3] << (32 - sh)), (d[3] >> sh) | (d[4] << (32 - sh))};
            }
#if OUSTER_NT_STANDALONE
            __builtin_nontemporal_store(o, (v4*)(drow + ((size_t)i << 4)));
#else
            *(v4*)(drow + ((size_t)i << 4)) = o;
#endif
        }
    }
}

__global__ __launch_bounds__(256) void k_destagger(DestaggerArgs a, uint32_t lds_row) {
    extern __shared__ uint4 s_row[];
    const uint32_t u = blockIdx.x, img = blockIdx.y;
    const size_t row_bytes = (size_t)a.w * a.elem;
    const uint8_t* srow = (const uint8_t*)a.src + ((size_t)img * a.h + u) * row_bytes;
    uint8_t* drow = (uint8_t*)a.dst + ((size_t)img * a.h + u) * row_bytes;
    const size_t shift_bytes = (size_t)a.offsets[u] * a.elem;  // dst byte b <- src byte (b - shift) mod row
    const bool fast = ((row_bytes & 15) == 0) && ((((uintptr_t)a.src | (uintptr_t)a.dst) & 15) == 0);
    if (fast && lds_row) {
        const uint32_t nchunk = (uint32_t)(row_bytes >> 4), ndw = nchunk * 4;
        for (uint32_t i = threadIdx.x; i < nchunk; i += blockDim.x) s_row[i] = ((const uint4*)srow)[i];
        __syncthreads();
        const uint32_t* s_dw = (const uint32_t*)s_row;
        const uint32_t sb0 = (uint32_t)(row_bytes - shift_bytes);   // source byte of destination byte 0 (== row_bytes: no shift)
        const uint32_t sh = (sb0 & 3) * 8;
        for (uint32_t i = threadIdx.x; i < nchunk; i += blockDim.x) {
            uint32_t dw = (sb0 >> 2) + i * 4;     // first source dword of this chunk, before the wrap
            if (dw >= ndw) dw -= ndw;
            uint32_t d[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                uint32_t j = dw + k;
                if (j >= ndw) j -= ndw;
                d[k] = s_dw[j];
            }
            uint4 o;
            if (sh == 0) {
                o = make_uint4(d[0], d[1], d[2], d[3]);
            } else {
                o.x = (d[0] >> sh) | (d[1] << (32 - sh));
                o.y = (d[1] >> sh) | (d[2] << (32 - sh));
                o.z = (d[2] >> sh) | (d[3] << (32 - sh));
                o.w = (d[3] >> sh) | (d[4] << (32 - sh));
            }
#if OUSTER_NT_STANDALONE
            {
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4{o.x, o.y, o.z, o.w}, (v4*)(drow + ((size_t)i << 4)));
            }
#else
            *(uint4*)(drow + ((size_t)i << 4)) = o;
#endif
        }
    } else if (fast) {
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
    const uint32_t tid = threadIdx.x;
    // full-LUT mode: a workgroup projects its tile for a GROUP of images, so the tile's LUT rows (24 B/px for f32,
    // 1.5 x everything else the kernel moves) are fetched once per group instead of once per image
    const uint32_t ipb = a.images_per_block ? a.images_per_block : 1u;
    const uint32_t img0 = blockIdx.y * ipb, img1 = min(a.n_images, img0 + ipb);
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
    auto store = [&](uint32_t img, size_t rowpix, const double (&p)[4][3]) {
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
    };
    auto load_range = [&](uint32_t img, size_t rowpix, uint32_t (&rng)[4]) {
        rng[0] = rng[1] = rng[2] = rng[3] = 0;
        if (live) {
            const uint4 t = *(const uint4*)(a.range + (size_t)img * npix + rowpix);
            rng[0] = t.x; rng[1] = t.y; rng[2] = t.z; rng[3] = t.w;
        }
    };
    for (uint32_t r = r_begin + ty; r < r_end; r += RPP) {
        const size_t rowpix = (size_t)r * W + col;
        const size_t rowpix0 = (size_t)r * W + c0;
        // the images of the group, four at a time: the next four range quads are in flight while these four are
        // projected and stored (one load per iteration left the full-LUT kernel latency bound: 53 % of the roofline)
        auto project_group = [&](auto&& project) {
            constexpr int B = 4;
            uint32_t cur[B][4], nxt[B][4];
#pragma unroll
            for (int k = 0; k < B; ++k)
                if (img0 + k < img1) load_range(img0 + k, rowpix, cur[k]);
            for (uint32_t img = img0; img < img1; img += B) {
#pragma unroll
                for (int k = 0; k < B; ++k)
                    if (img + B + k < img1) load_range(img + B + k, rowpix, nxt[k]);
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    if (img + k >= img1) break;
                    double p[4][3];
                    project(cur[k], p);
                    store(img + k, rowpix, p);
                }
#pragma unroll
                for (int k = 0; k < B; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) cur[k][c] = nxt[k][c];
            }
        };
        if constexpr (MODE == 1 || MODE == 2) {
            const double* b = lut.beam_tab + (size_t)r * 9;
            const double u0 = b[0], u1 = b[1], u2 = b[2], v0 = b[3], v1 = b[4], v2 = b[5],
                         w0 = b[6], w1 = b[7], w2 = b[8];
            double dd[4][3];   // the row's directions for my four columns: the same for every image of the group
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dd[c][0] = fma(cx[c], u0, fma(sx[c], v0, w0));
                dd[c][1] = fma(cx[c], u1, fma(sx[c], v1, w1));
                dd[c][2] = fma(cx[c], u2, fma(sx[c], v2, w2));
            }
            project_group([&](const uint32_t (&rng)[4], double (&p)[4][3]) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double rm = (double)rng[c] - lut.n;
#pragma unroll
                    for (int k = 0; k < 3; ++k) p[c][k] = rng[c] ? fma(rm, dd[c][k], kc[c][k]) : 0.0;
                }
            });
        } else {
            // the LUT rows are streamed like the output: coalesced row segments, transposed to "lane owns 4 pixels"
            // through the scratch -- once, then every image of the group is projected from registers
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
                project_group([&](const uint32_t (&rng)[4], double (&p)[4][3]) { project_full4<float>(d.t, o.t, rng, p); });
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
                project_group([&](const uint32_t (&rng)[4], double (&p)[4][3]) { project_full4<double>(d.t, o.t, rng, p); });
            }
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
    // One workgroup per frame.  Every global access is coalesced (thread t takes columns t, t + 256, ...) and a block's
    // loads are issued together from clamped addresses; the exclusive scan wants eight CONSECUTIVE columns per thread, so the
    // counts cross an LDS image (2048 columns per block, +1 dword of padding every 32).  The first form of this kernel walked
    // eight consecutive columns per thread straight from memory, twice, behind per-column guards: ~16 dependent round
    // trips, 16 us for 256 frames; this one takes two round trips per block of 2048 columns.
    constexpr uint32_t SEG = 8, BLK = 256 * SEG;
    __shared__ int s_lo, s_hi;
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_c[BLK + BLK / 32];
    const uint32_t W = a.w, f = blockIdx.x, tid = threadIdx.x;
    const uint32_t* st = a.status + (size_t)f * W;
    uint32_t* off = a.col_off + (size_t)f * (W + 1);
    if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; }
    __syncthreads();
    // first / last valid column (status bit 0; lidar_frame.cpp:907-925)
    int lo = 0x7fffffff, hi = -1;
    for (uint32_t base = 0; base < W; base += BLK) {
        uint32_t v[SEG];
#pragma unroll
        for (uint32_t k = 0; k < SEG; ++k) v[k] = st[min(base + tid + 256u * k, W - 1u)];
#pragma unroll
        for (uint32_t k = 0; k < SEG; ++k) {
            const uint32_t x = base + tid + 256u * k;
            if (x < W && (v[k] & 1u)) { lo = min(lo, (int)x); hi = max(hi, (int)x); }
        }
    }
    if (hi >= 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    lo = s_lo; hi = s_hi;
    // kept points per column: k_dwf_count's, or the decode kernel's range-gate by-product (partial counts per row chunk,
    // summed here)
    const uint16_t* ext = a.gate_counts ? a.gate_counts + (size_t)f * OUSTER_HIP_GATE_CHUNKS * W : nullptr;
    auto pad = [](uint32_t i) { return i + (i >> 5); };
    uint32_t carry = 0;
    for (uint32_t base = 0; base < W; base += BLK) {
        uint32_t v[SEG], n[SEG], xc[SEG];
#pragma unroll
        for (uint32_t k = 0; k < SEG; ++k) {
            xc[k] = min(base + tid + 256u * k, W - 1u);
            v[k] = st[xc[k]];
        }
        // the branch on `ext` stays outside the column loop: inside it every column's partial counts were loaded and waited
        // for on their own (eight round trips per block)
        if (ext) {
            uint16_t g[SEG][OUSTER_HIP_GATE_CHUNKS];
#pragma unroll
            for (uint32_t k = 0; k < SEG; ++k)
#pragma unroll
                for (uint32_t c = 0; c < OUSTER_HIP_GATE_CHUNKS; ++c) g[k][c] = ext[(size_t)c * W + xc[k]];
#pragma unroll
            for (uint32_t k = 0; k < SEG; ++k) {
                n[k] = 0;
#pragma unroll
                for (uint32_t c = 0; c < OUSTER_HIP_GATE_CHUNKS; ++c) n[k] += g[k][c];
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < SEG; ++k) n[k] = off[xc[k]];
        }
#pragma unroll
        for (uint32_t k = 0; k < SEG; ++k) {
            const uint32_t x = base + tid + 256u * k;
            const bool keep = x < W && (int)x >= lo && (int)x <= hi && v[k] != 0;
            s_c[pad(tid + 256u * k)] = keep ? n[k] : 0u;
        }
        __syncthreads();
        uint32_t c[SEG], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < SEG; ++i) {
            c[i] = s_c[pad(tid * SEG + i)];
            sum += c[i];
        }
        uint32_t total;
        uint32_t run = carry + block_exscan_256(sum, s_wave, &total);   // (two barriers inside: s_c is read by now)
#pragma unroll
        for (uint32_t i = 0; i < SEG; ++i) {
            s_c[pad(tid * SEG + i)] = run;
            run += c[i];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < SEG; ++k) {
            const uint32_t x = base + tid + 256u * k;
            if (x < W) off[x] = s_c[pad(tid + 256u * k)];
        }
        carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        off[W] = carry;
        a.frame_off[f + 1] = carry;
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
// A pointer that was itself loaded from memory is a generic one and is read with flat_load, which counts on lgkmcnt as well
// as vmcnt: the s_waitcnt lgkmcnt(0) in front of the next barrier then waits for it.  Device tables are global memory.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
    return (const __attribute__((address_space(1))) T*)(uintptr_t)p;
}

// rows 0..2 of a column's pose in the emit kernels' register order (rows 0 and 1 interleaved, see DwfColMeta), cast to T:
// from the caller's float table (48 B per column, three 16 B loads) or from the 4 x 4 doubles (128 B)
template <class T>
__device__ __forceinline__ void load_pose_rows(const DewarpFramesArgs& a, size_t col, T (&pose)[12]) {
    if constexpr (sizeof(T) == 4) {
        if (a.pose_rows) {
            const float4* pr = (const float4*)(a.pose_rows + col * 12);
            const float4 r0 = pr[0], r1 = pr[1], r2 = pr[2];
            pose[0] = r0.x; pose[2] = r0.y; pose[4] = r0.z; pose[6] = r0.w;
            pose[1] = r1.x; pose[3] = r1.y; pose[5] = r1.z; pose[7] = r1.w;
            pose[8] = r2.x; pose[9] = r2.y; pose[10] = r2.z; pose[11] = r2.w;
            return;
        }
    }
    const double* pm = a.poses + col * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pose[2 * k] = (T)pm[k];
        pose[2 * k + 1] = (T)pm[4 + k];
        pose[8 + k] = (T)pm[8 + k];
    }
}

// column metadata of an emit tile, staged once in LDS and read back as wave-uniform broadcasts
template <class T>
struct __attribute__((aligned(16))) DwfColMeta {
    T pose[12];      // rows 0..2 of the column's 4x4 pose, cast to the output type; rows 0 and 1 interleaved
                     // (p0 p4 p1 p5 p2 p6 p3 p7 | p8..p11) so that x and y are formed by packed f32 FMAs on adjacent registers
    double col[5];   // separable LUT: cos, sin of the encoder angle and the column constant
    uint32_t base, cnt;
    uint64_t ts;
};

// The column loop of the emit kernels: wave = CPW columns of a tile that sits in LDS (ranges, pitch TILE + 1) with its
// column metadata (count, output base, pose, table row), lane = row (NR rows per lane).  EASY: everything the tile emits lies
// below `capacity` and no kept range can be zero (the gate's lower bound is at least 1) -- then the loop carries neither the
// per-point room check nor the zero-range select.
// SWZ: the tile image came by LDS-DMA (k_dwf_emit_stream): lane-linear, so it cannot be padded per row; instead the 16 B
// cell of columns 4k..4k+3 of row r sits at cell position k ^ (r & 15) of its row -- a wave reading one column over 64 rows
// then spreads over 16 four-bank groups (4-way conflicts on two reads per column, against 64-way for the plain layout).
template <class T, bool SEP, int TILE, int ROWS, bool EASY, int NWAVES = 4, bool SWZ = false>
__device__ __forceinline__ void dwf_column_loop(const DewarpFramesArgs& a, const LutDev& lut, const uint32_t* s_rng,
                                                const DwfColMeta<T>* s_meta, uint32_t f, uint32_t c0, uint32_t ncol,
                                                uint64_t fbase, const uint32_t (&row)[ROWS / 64],
                                                const double (&bt)[ROWS / 64][9], uint32_t& m_run) {
    constexpr int PITCH = SWZ ? TILE : TILE + 1, CPW = TILE / NWAVES, NR = ROWS / 64;
    constexpr bool roomy = EASY, no_zero = EASY;
    const uint32_t W = a.w, H = a.h;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the column loop is software-pipelined over LDS: a column's ranges and count are read one column ahead, so the
    // keep masks can be formed the moment the iteration starts and no LDS round trip stands between two columns
    uint32_t rn[NR], cntn;
    auto fetch_col = [&](uint32_t jj_next) {
        const uint32_t jn = min(wave * CPW + jj_next, (uint32_t)TILE - 1u);
        cntn = s_meta[jn].cnt;
#pragma unroll
        for (int hh = 0; hh < NR; ++hh) {
            const uint32_t rr = lane + 64 * hh;
            rn[hh] = SWZ ? s_rng[rr * PITCH + ((((jn >> 2) ^ (rr & 15u)) << 2) | (jn & 3u))] : s_rng[rr * PITCH + jn];
        }
    };
    fetch_col(0);
    for (uint32_t jj = 0; jj < (uint32_t)CPW; ++jj) {
        const uint32_t j = wave * CPW + jj, x = c0 + j;  // wave-uniform
        if (j >= ncol) break;
        const DwfColMeta<T>& m = s_meta[j];
        uint32_t r[NR];
#pragma unroll
        for (int hh = 0; hh < NR; ++hh) r[hh] = rn[hh];
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(cntn);
        if (cnt == 0) {  // masked out or empty column
            fetch_col(jj + 1);
            continue;
        }
        // this column's constants first (they are needed soonest), then the next column's ranges
        const uint32_t m_base = m.base;
        T ps[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) ps[k] = m.pose[k];
        double cx = 0, sx = 0, kc[3] = {0, 0, 0};
        if constexpr (SEP) {
            cx = m.col[0]; sx = m.col[1];
            kc[0] = m.col[2]; kc[1] = m.col[3]; kc[2] = m.col[4];
        }
        fetch_col(jj + 1);
        uint32_t rank[NR];
        bool keep[NR];
        uint32_t n_keep = 0;
#pragma unroll
        for (int hh = 0; hh < NR; ++hh) {
            // rows past H were staged as range 0, which the easy path's gate (min_r > 0) rejects by itself: one
            // subtract-and-compare whose result IS the ballot
            if constexpr (no_zero) keep[hh] = r[hh] - a.min_r <= a.max_r - a.min_r;
            else keep[hh] = row[hh] < H && r[hh] >= a.min_r && r[hh] <= a.max_r;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(keep[hh]);
            // kept rows below this lane (v_mbcnt_lo / _hi), on top of the rows kept by the earlier row groups
            rank[hh] = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, n_keep));
            n_keep += (uint32_t)__popcll(mask);
        }
        if (n_keep == 0) continue;
        const uint64_t g0 = fbase + __builtin_amdgcn_readfirstlane(m_base) + bcast_u32(m_run, jj);  // first point of this run
        const uint64_t room = roomy ? ~0ull : (g0 < a.capacity ? a.capacity - g0 : 0);
        T* const run = (T*)a.points + g0 * 3;   // wave-uniform: the stores take it as a scalar base + a 32-bit lane offset
#pragma unroll
        for (int hh = 0; hh < NR; ++hh) {
            if (!(keep[hh] && (roomy || rank[hh] < room))) continue;
            T pt[3];
#ifdef OUSTER_ABLATE_DWF_MATH    // experiment builds only: every load and store stays, the projection and the pose do not
            if constexpr (SEP) {
                Pt3<T> o0;
                o0.x = (T)r[hh]; o0.y = (T)rank[hh]; o0.z = (T)x;
                ((Pt3<T>*)run)[rank[hh]] = o0;
                continue;
            }
#endif
            if constexpr (SEP) {
                const double rm = (double)r[hh] - lut.n;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double d = fma(cx, bt[hh][k], fma(sx, bt[hh][3 + k], bt[hh][6 + k]));
                    const T t = (T)fma(rm, d, kc[k]);
                    pt[k] = (no_zero || r[hh]) ? t : (T)0;
                }
            } else {
                double p[3];
                const size_t pix = (size_t)row[hh] * W + x;
                if (lut.full_dtype == OUSTER_HIP_F32)
                    project_full<float>((const float*)lut.full_dir, (const float*)lut.full_ofs, pix, r[hh], p);
                else
                    project_full<double>((const double*)lut.full_dir, (const double*)lut.full_ofs, pix, r[hh], p);
                pt[0] = (T)p[0]; pt[1] = (T)p[1]; pt[2] = (T)p[2];
            }
            const T px = pt[0], py = pt[1], pz = pt[2];
            Pt3<T> o;
            // three fused multiply-adds per component, the translation as the innermost addend
            o.x = fma(ps[0], px, fma(ps[2], py, fma(ps[4], pz, ps[6])));
            o.y = fma(ps[1], px, fma(ps[3], py, fma(ps[5], pz, ps[7])));
            o.z = fma(ps[8], px, fma(ps[9], py, fma(ps[10], pz, ps[11])));
#ifdef OUSTER_ABLATE_DWF_STORE   // experiment builds only: the arithmetic stays, (almost) nothing is stored
            if (!(o.x == (T)-12345.678 && o.y == (T)8765.4321)) continue;
#endif
#if OUSTER_NT_STANDALONE
            {
                T* pd = run + rank[hh] * 3u;
                __builtin_nontemporal_store(o.x, pd);
                __builtin_nontemporal_store(o.y, pd + 1);
                __builtin_nontemporal_store(o.z, pd + 2);
            }
#else
            ((Pt3<T>*)run)[rank[hh]] = o;
#endif
        }
        // every point of the run carries the same provenance: dense lanes 0..n_keep-1
        if (a.col_idxs || a.frame_idxs || a.timestamps_ns) {
            const uint64_t ts = m.ts;
            for (uint32_t i = lane; i < n_keep && i < room; i += 64) {
                if (a.col_idxs) a.col_idxs[g0 + i] = x;
                if (a.frame_idxs) a.frame_idxs[g0 + i] = f;
                if (a.timestamps_ns) a.timestamps_ns[g0 + i] = ts;
            }
        }
        if (lane == jj) m_run += n_keep;
    }
}

template <class T, bool SEP, int TILE, int ROWS>
__global__ __launch_bounds__(256) void k_dwf_emit(DewarpFramesArgs a) {
    // tile = TILE columns x ROWS rows (64 x 128: a 128-beam sensor's whole columns).  A wave owns CPW
    // columns and walks them one at a time, lane = row (NR rows per lane): the kept rows are ranked
    // with ballots and every lane stores its own 12 / 24 B point, so one store instruction writes one
    // dense run.  The column's metadata (output base, pose cast to T, table row, timestamp) sits in
    // LDS and comes back as broadcast reads -- no dependent global loads in the column loop.
    constexpr int LPR = TILE / 4, PITCH = TILE + 1, CPW = TILE / 4, NR = ROWS / 64;
    constexpr int RPP = 256 / LPR;  // rows staged per pass
    static_assert(ROWS % 64 == 0 && TILE % 4 == 0 && TILE <= 256, "emit tile");
    __shared__ uint32_t s_rng[ROWS * PITCH];
    __shared__ DwfColMeta<T> s_meta[TILE];
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
    if (tid < (uint32_t)TILE) {
        DwfColMeta<T> m;
#pragma unroll
        for (int k = 0; k < 12; ++k) m.pose[k] = (T)0;
#pragma unroll
        for (int k = 0; k < 5; ++k) m.col[k] = 0.0;
        m.base = m.cnt = 0;
        m.ts = 0;
        const uint32_t mx = c0 + tid;
        if (tid < ncol) {
            m.base = off[mx];
            m.cnt = off[mx + 1] - m.base;
            if (m.cnt) {
                load_pose_rows<T>(a, (size_t)f * W + mx, m.pose);
                if constexpr (SEP) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) m.col[k] = as_global(lut.col_tab)[(size_t)mx * 5 + k];
                }
                if (a.timestamps_ns) m.ts = a.timestamp[(size_t)f * W + mx];
            }
        }
        s_meta[tid] = m;
    }
    // The common case, decided once per tile: everything this tile emits lies below `capacity`, and no kept range can be
    // zero (the gate's lower bound is at least 1): then the column loop carries neither the per-point room check nor the
    // zero-range select.
    const bool easy = fbase + off[c0 + ncol] <= a.capacity && a.min_r > 0;
    uint32_t m_run = 0;  // lane jj of the wave: points of its jj-th column already written (previous row chunks)
    for (uint32_t r0 = 0; r0 < H; r0 += ROWS) {
        // the lane's rows of the beam table (a few KB shared by every tile): asked for before the tile is staged, so that the
        // two latencies overlap
        uint32_t row[NR];
        double bt[NR][9];
#pragma unroll
        for (int hh = 0; hh < NR; ++hh) {
            row[hh] = r0 + lane + 64 * hh;
            if constexpr (SEP) {
                const uint32_t rc = min(row[hh], H - 1u);   // unconditional loads: a guarded one is waited for on its own
#pragma unroll
                for (int k = 0; k < 9; ++k) bt[hh][k] = as_global(lut.beam_tab)[(size_t)rc * 9 + k];
            }
        }
        __syncthreads();  // previous chunk consumed (and, first time, the metadata published)
        if (vec) {
            // all passes' 16 B loads are issued before the first LDS write (clamped addresses instead of guards: a guarded
            // load and its guarded write compile to load / s_waitcnt vmcnt(0) / write, one memory round trip per pass).
            // Issuing them above the metadata block as well was tried: 164 VGPRs, three waves per SIMD, no gain.
            constexpr int NP = ROWS / RPP;
            uint4 t[NP];
            const uint32_t col = c0 + 4 * q, cc = min(col, W - 4u);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const uint32_t rc = min(r0 + ty + (uint32_t)i * RPP, H - 1u);
                t[i] = *(const uint4*)(rp + (size_t)rc * W + cc);
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const uint32_t rr = ty + (uint32_t)i * RPP;
                const bool in = r0 + rr < H && col < W;
                uint32_t* d = &s_rng[rr * PITCH + 4 * q];
                d[0] = in ? t[i].x : 0u; d[1] = in ? t[i].y : 0u; d[2] = in ? t[i].z : 0u; d[3] = in ? t[i].w : 0u;
            }
        } else {
            for (uint32_t rr = ty; rr < (uint32_t)ROWS; rr += RPP) {
                const uint32_t r = r0 + rr, col = c0 + 4 * q;
                uint32_t v[4] = {0, 0, 0, 0};
                if (r < H)
                    for (uint32_t c = 0; c < 4 && col + c < W; ++c) v[c] = rp[(size_t)r * W + col + c];
#pragma unroll
                for (int c = 0; c < 4; ++c) s_rng[rr * PITCH + 4 * q + c] = v[c];
            }
        }
        __syncthreads();
        if (easy) dwf_column_loop<T, SEP, TILE, ROWS, true>(a, lut, s_rng, s_meta, f, c0, ncol, fbase, row, bt, m_run);
        else dwf_column_loop<T, SEP, TILE, ROWS, false>(a, lut, s_rng, s_meta, f, c0, ncol, fbase, row, bt, m_run);
    }
}

#ifdef OUSTER_EXPERIMENTS
// ------------------------------------------------------------------------------------
// k_dwf_emit_stream: k_dwf_emit as a PERSISTENT kernel with a loader wave (round 6; k_decode_stream2's skeleton).
// k_dwf_emit's time is the sum of its read side and its write side (0.127 ms without the stores, 0.19 with them, 256 frames):
// a workgroup asks for its 32 KB range tile, waits, and only then computes and stores, and four such workgroups per CU do not
// hide one another's wait behind a saturated memory system.  Here two workgroups per CU live for the whole launch and walk
// the (frame, column tile) list; each has TWO tile contexts in LDS: while eight waves rank, project and store tile i, the
// ninth has tile i+1 arriving by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip; nothing for the column loop to wait on) and
// writes its column metadata.  One workgroup barrier per tile.  Same outputs, order and arithmetic as k_dwf_emit.
// Needs whole tiles (W % 64 == 0), ROWS == H, 16 B aligned range planes and the separable tables; anything else: k_dwf_emit.
// MEASURED (round 6, 256 frames of 128 x 2048, same box, same process): bit-identical output, 0.2173 ms for the chain against
// 0.2143 ms with k_dwf_emit -- hiding the tile's read latency buys nothing, so the emit kernel's time is NOT the sum of an
// exposed read side and a write side (the reading of DESIGN r05 section 10.2); 94 VGPRs, 2 x 79 KB of LDS, 18 waves per CU.
// Kept for A/B work only (make EXPERIMENTS=1, knob "dwf_stream" = 1); a default build does not compile it.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void dwf_glds16(const void* g, uint32_t lds) {   // one LDS-DMA wave instruction (k_decode_stream.hip: glds16)
    uint32_t keep;
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
struct __attribute__((aligned(16))) DwfTileHdr {
    uint64_t fbase;
    uint32_t f, c0, any, easy, pad0, pad1;
};
template <class T, int ROWS>
constexpr uint32_t dwf_stream_ctx_bytes() { return (uint32_t)(ROWS * 64 * 4 + 64 * sizeof(DwfColMeta<T>) + sizeof(DwfTileHdr)); }

template <class T, int ROWS>
__global__ __launch_bounds__(576, 2) void k_dwf_emit_stream(DewarpFramesArgs a, uint32_t n_items) {
    constexpr int TILE = 64, NCW = 8, NR = ROWS / 64;
    constexpr uint32_t RNG_BYTES = ROWS * TILE * 4, META_BYTES = TILE * sizeof(DwfColMeta<T>), CTX = dwf_stream_ctx_bytes<T, ROWS>();
    extern __shared__ __align__(16) uint32_t smem[];
    const uint32_t W = a.w, H = a.h, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles = W / TILE;
    const uint32_t G = gridDim.x;
    if (blockIdx.x >= n_items) return;
    const uint32_t n_mine = (n_items - blockIdx.x + G - 1) / G;
    auto item = [&](uint32_t i) { return blockIdx.x + G * i; };
    auto ctx_rng = [&](uint32_t b) { return (uint32_t*)((uint8_t*)smem + b * CTX); };
    auto ctx_meta = [&](uint32_t b) { return (DwfColMeta<T>*)((uint8_t*)smem + b * CTX + RNG_BYTES); };
    auto ctx_hdr = [&](uint32_t b) { return (DwfTileHdr*)((uint8_t*)smem + b * CTX + RNG_BYTES + META_BYTES); };

    if (wave >= (uint32_t)NCW) {
        // ================= the loader wave: lane = column of the tile for the metadata, 16 B cell for the ranges =================
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        auto fetch = [&](uint32_t it, uint32_t b) {
            const uint32_t f = it / tiles, c0 = (it - f * tiles) * TILE;
            const uint32_t* off = a.col_off + (size_t)f * (W + 1);
            const uint32_t mx = c0 + lane;
            const uint32_t base = off[mx], next = off[mx + 1];
            const uint32_t first = __builtin_amdgcn_readfirstlane(base), last = (uint32_t)__builtin_amdgcn_readlane((int)next, 63);
            const bool any = last != first;
            const uint64_t fbase = a.frame_off[f];
            if (any) {
                // ranges: instruction k brings rows 4k..4k+3; lane l -> row 4k + l/16, LDS cell l%16, i.e. columns 4*(cell ^ (row & 15))..
                const uint8_t* rp = (const uint8_t*)(a.range + (size_t)f * W * H) + (size_t)c0 * 4;
                const uint32_t r_in = lane >> 4, cell = lane & 15u;
#pragma unroll 8
                for (uint32_t k = 0; k < (uint32_t)ROWS / 4u; ++k) {
                    const uint32_t r = 4u * k + r_in;
                    dwf_glds16(rp + (size_t)r * W * 4 + ((cell ^ (r & 15u)) << 4), lds0 + b * CTX + k * 1024u);
                }
            }
            DwfColMeta<T> m;
#pragma unroll
            for (int k = 0; k < 12; ++k) m.pose[k] = (T)0;
#pragma unroll
            for (int k = 0; k < 5; ++k) m.col[k] = 0.0;
            m.base = base;
            m.cnt = next - base;
            m.ts = 0;
            if (m.cnt) {
                load_pose_rows<T>(a, (size_t)f * W + mx, m.pose);
                const LutDev* lp = a.luts + f % a.n_luts;
                const double* ct = lp->col_tab;
#pragma unroll
                for (int k = 0; k < 5; ++k) m.col[k] = as_global(ct)[(size_t)mx * 5 + k];
                if (a.timestamps_ns) m.ts = a.timestamp[(size_t)f * W + mx];
            }
            ctx_meta(b)[lane] = m;
            if (lane == 0) {
                DwfTileHdr h;
                h.fbase = fbase; h.f = f; h.c0 = c0; h.any = any ? 1u : 0u;
                h.easy = (fbase + last <= a.capacity && a.min_r > 0) ? 1u : 0u;
                h.pad0 = h.pad1 = 0;
                *ctx_hdr(b) = h;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile has landed (this wave's queue holds nothing else)
        };
        fetch(item(0), 0);
        for (uint32_t i = 0; i < n_mine; ++i) {
            __syncthreads();   // tile i is in its context; the other one is free (the column loops are done with tile i-1)
            if (i + 1 < n_mine) fetch(item(i + 1), (i + 1) & 1u);
        }
        return;
    }

    // ================= the eight column-loop waves: wave = 8 columns of the tile, lane = row =================
    uint32_t row[NR];
    double bt[NR][9];
    uint32_t cur_lut = 0xffffffffu;
    LutDev lut{};
#pragma unroll
    for (int hh = 0; hh < NR; ++hh) row[hh] = lane + 64 * hh;
    for (uint32_t i = 0; i < n_mine; ++i) {
        const uint32_t b = i & 1u;
        __syncthreads();
        const DwfTileHdr h = *ctx_hdr(b);
        if (!h.any) continue;
        const uint32_t li = h.f % a.n_luts;
        if (li != cur_lut) {   // the lane's rows of the beam table: once per launch for a one-sensor batch
            cur_lut = li;
            lut = a.luts[li];
#pragma unroll
            for (int hh = 0; hh < NR; ++hh)
#pragma unroll
                for (int k = 0; k < 9; ++k) bt[hh][k] = as_global(lut.beam_tab)[(size_t)min(row[hh], H - 1u) * 9 + k];
        }
        uint32_t m_run = 0;
        if (h.easy) dwf_column_loop<T, true, TILE, ROWS, true, NCW, true>(a, lut, ctx_rng(b), ctx_meta(b), h.f, h.c0, TILE, h.fbase, row, bt, m_run);
        else dwf_column_loop<T, true, TILE, ROWS, false, NCW, true>(a, lut, ctx_rng(b), ctx_meta(b), h.f, h.c0, TILE, h.fbase, row, bt, m_run);
    }
}

#endif  // OUSTER_EXPERIMENTS (k_dwf_emit_stream)
#ifdef OUSTER_EXPERIMENTS   // the two single-pass forms of the frame dewarp: measured slower than count / scan / emit (DESIGN_HISTORY.md H3);
                            // out of the default build since round 6 (make EXPERIMENTS=1)
// ------------------------------------------------------------------------------------
// k_dwf_single: the same range-gated, compacting frame dewarp in ONE pass over the range planes.
// A workgroup owns a 64-column tile of one frame for ALL rows (H x 65 dwords of LDS), counts the kept
// points of its columns from LDS, and learns where its output starts from a decoupled look-back over
// the tiles before it (Merrill & Garland: every tile publishes {aggregate | inclusive prefix} in one
// 64-bit word; a wave inspects 64 predecessors per step).  Tiles take their index from ticket
// counters -- eight of them, one per blockIdx & 7 class (tile = 8 * ticket + class): a single counter
// serialises 8192 returning atomics (~90 us measured), eight run side by side.  Within a class a tile's
// predecessors have always started; across classes that holds because workgroups are dispatched in
// index order (a class would have to fall > 1000 workgroups behind the others for a wait to stall).  Output order, counts and provenance are exactly k_dwf_count/scan/emit's (= the reference's,
// impl/dewarp_impl.h:23-115); the range plane is read once instead of twice and three launches
// disappear.
//   tile_state[16 * c]   ticket counter of class c, c < 8  (the buffer is zeroed before the launch)
//   tile_state[128 + t]  [63:62] 0 nothing / 1 aggregate / 2 inclusive prefix, [61:0] points
// ------------------------------------------------------------------------------------
#ifndef OUSTER_DWF_NT
#define OUSTER_DWF_NT 256
#endif
constexpr int DWF_NT = OUSTER_DWF_NT;  // threads per tile of k_dwf_single
constexpr uint32_t DWF_WORDS = 128;  // tile words start behind the eight padded ticket counters
constexpr uint64_t DWF_AGG = 1ull << 62, DWF_INC = 2ull << 62, DWF_VAL = (1ull << 62) - 1;

// NT threads share one tile.  Measured (256 frames 128x2048, gate 0.5-400 m, same box): NT 256 0.428 ms,
// 512 0.500 ms, 1024 0.698 ms against 0.344 ms for the count / scan / emit kernels -- the emit loop is
// instruction bound (rocprofv3 --pmc: ~2600 VALU + ~1600 SALU per wave, the v_readlane broadcasts of the
// column metadata), and the whole-column tile (33 KB of LDS) halves the waves per CU that hide it; the
// range plane read it saves (65-75 us of k_dwf_count) does not pay for that.  Kept behind the
// "dewarp_single_pass" knob for sensors of more than 128 beams (k_dwf_fused below serves the others); the three kernels
// stay the default.
template <class T, bool SEP, int NT>
__global__ __launch_bounds__(NT) void k_dwf_single(DewarpFramesArgs a) {
    constexpr int TILE = 64, PITCH = TILE + 1, LPR = 16, RPP = NT / LPR, CPW = TILE / (NT / 64);
    extern __shared__ __align__(16) uint32_t s_rng[];   // [H][PITCH]
    __shared__ uint32_t s_cnt[TILE], s_ticket;
    __shared__ int s_lo, s_hi;
    __shared__ uint64_t s_base;
    const uint32_t W = a.w, H = a.h, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles = (W + TILE - 1) / TILE;
    if (tid == 0) {
        const uint32_t cls = blockIdx.x & 7u;
        s_ticket = (uint32_t)atomicAdd((unsigned long long*)&a.tile_state[16 * cls], 1ull) * 8u + cls;
        s_lo = 0x7fffffff;
        s_hi = -1;
    }
    if (tid < TILE) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t t = s_ticket, f = t / tiles, tile = t - f * tiles;
    const uint32_t c0 = tile * TILE, ncol = min((uint32_t)TILE, W - c0);
    const uint32_t* rp = a.range + (size_t)f * W * H;
    const uint32_t* st = a.status + (size_t)f * W;
    const LutDev lut = a.luts[f % a.n_luts];
    const bool vec = (W % 4 == 0) && ((((uintptr_t)a.range) & 15) == 0);

    // ---- first / last valid column of the frame (status & 1, lidar_frame.cpp:907-925)
    {
        int lo = 0x7fffffff, hi = -1;
        for (uint32_t x = tid; x < W; x += NT)
            if (st[x] & 1u) { lo = min(lo, (int)x); hi = max(hi, (int)x); }
        if (hi >= 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    // ---- stage the tile's range values, all rows, coalesced row segments
    {
        const uint32_t q = tid % LPR, ty = tid / LPR, col = c0 + 4 * q;
        for (uint32_t r = ty; r < H; r += RPP) {
            uint32_t v[4] = {0, 0, 0, 0};
            if (col < W) {
                if (vec) {
                    const uint4 x = *(const uint4*)(rp + (size_t)r * W + col);
                    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
                } else {
                    for (uint32_t c = 0; c < 4 && col + c < W; ++c) v[c] = rp[(size_t)r * W + col + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) s_rng[r * PITCH + 4 * q + c] = v[c];
        }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;

    // ---- column metadata: lane l of the wave holds column wave*CPW + l % CPW
    const uint32_t ml = lane % CPW, mj = wave * CPW + ml, mx = c0 + mj;
    const bool col_on = mj < ncol && (int)mx >= lo && (int)mx <= hi && st[mx] != 0;
    // kept points per column: wave = 16 columns, lane = row
    for (uint32_t jj = 0; jj < (uint32_t)CPW; ++jj) {
        const uint32_t j = wave * CPW + jj;
        if (j >= ncol) break;
        uint32_t n = 0;
        for (uint32_t r0 = 0; r0 < H; r0 += 64) {
            const uint32_t row = r0 + lane;
            const uint32_t r = row < H ? s_rng[row * PITCH + j] : 0u;
            n += (uint32_t)__popcll(__ballot(row < H && r >= a.min_r && r <= a.max_r));
        }
        if (lane == jj) s_cnt[j] = n;   // lane jj is the metadata lane of column j (ml == jj)
    }
    __syncthreads();
    // exclusive scan of the 64 column counts (masked), by every wave for itself
    uint32_t m_cnt = 0, m_base = 0, tile_total = 0;
    {
        const uint32_t jx = lane;  // column of the tile
        const bool on = jx < ncol && (int)(c0 + jx) >= lo && (int)(c0 + jx) <= hi && st[min(c0 + jx, W - 1)] != 0;
        const uint32_t v = on ? s_cnt[jx] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += up;
        }
        tile_total = __shfl(inc, 63, 64);
        const uint32_t ex = inc - v;
        // hand column mj's numbers to its metadata lane
        m_base = __shfl(ex, (int)mj, 64);
        m_cnt = __shfl(v, (int)mj, 64);
        if (!col_on) m_cnt = 0;
    }

    // ---- where does this tile's output start?  publish, then look back (wave 0)
    if (wave == 0) {
        uint64_t excl = 0;
        unsigned long long* my = (unsigned long long*)&a.tile_state[DWF_WORDS + t];
        if (t == 0) {
            if (lane == 0) __hip_atomic_store(my, DWF_INC | (uint64_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(my, DWF_AGG | (uint64_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t back = (int64_t)t - 1;        // nearest predecessor not yet accounted for
            while (true) {
                const int64_t i = back - (int64_t)lane;
                uint64_t w = DWF_INC;             // lanes before the first tile: a zero inclusive prefix
                if (i >= 0) {
                    do {
                        w = __hip_atomic_load((unsigned long long*)&a.tile_state[DWF_WORDS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((w >> 62) == 0);
                }
                const uint64_t inc_mask = __ballot((w >> 62) == 2);
                const uint32_t first_inc = inc_mask ? (uint32_t)__builtin_ctzll(inc_mask) : 64u;
                uint64_t part = lane <= first_inc ? (w & DWF_VAL) : 0ull;   // aggregates up to (incl.) the first inclusive word
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (first_inc < 64u) break;
                back -= 64;
            }
            if (lane == 0) __hip_atomic_store(my, DWF_INC | (excl + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = excl;
            if (tile == 0) a.frame_off[f] = excl;
            if (t + 1 == a.n_frames * tiles) a.frame_off[a.n_frames] = excl + tile_total;
        }
    }
    __syncthreads();
    const uint64_t fbase = s_base;
    if (tile_total == 0) return;

    // ---- emit (as k_dwf_emit, from the resident tile)
    uint64_t m_ts = 0;
    T m_pose[12];
    double m_col[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 12; ++k) m_pose[k] = (T)0;
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
    uint32_t m_run = 0;  // points of my column already written (previous row chunks)
    for (uint32_t r0 = 0; r0 < H; r0 += 64) {
        const uint32_t row = r0 + lane;
        double bt[9];
        if constexpr (SEP) {
#pragma unroll
            for (int k = 0; k < 9; ++k) bt[k] = row < H ? lut.beam_tab[(size_t)row * 9 + k] : 0.0;
        }
        for (uint32_t jj = 0; jj < (uint32_t)CPW; ++jj) {
            const uint32_t j = wave * CPW + jj, x = c0 + j;  // wave-uniform
            if (j >= ncol) break;
            if (bcast_u32(m_cnt, jj) == 0) continue;  // masked out or empty column
            const uint32_t r = row < H ? s_rng[row * PITCH + j] : 0u;
            const bool keep = row < H && r >= a.min_r && r <= a.max_r;
            const uint64_t mask = __ballot(keep);
            const uint32_t n_keep = (uint32_t)__popcll(mask);
            if (n_keep == 0) continue;
            const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            const uint64_t g0 = fbase + bcast_u32(m_base, jj) + bcast_u32(m_run, jj);  // first point of this run
            const uint64_t room = g0 < a.capacity ? a.capacity - g0 : 0;
            if (keep && rank < room) {
                double p[3];
                if constexpr (SEP) {
                    const double cx = bcast(m_col[0], jj), sx = bcast(m_col[1], jj);
                    const double rm = (double)r - lut.n;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double d = fma(cx, bt[k], fma(sx, bt[3 + k], bt[6 + k]));
                        p[k] = r ? fma(rm, d, bcast(m_col[2 + k], jj)) : 0.0;
                    }
                } else {
                    const size_t pix = (size_t)row * W + x;
                    if (lut.full_dtype == OUSTER_HIP_F32)
                        project_full<float>((const float*)lut.full_dir, (const float*)lut.full_ofs, pix, r, p);
                    else
                        project_full<double>((const double*)lut.full_dir, (const double*)lut.full_ofs, pix, r, p);
                }
                const T px = (T)p[0], py = (T)p[1], pz = (T)p[2];
                Pt3<T> o;
                // the same three nested FMAs per component as k_dwf_emit (the two routes agree bit for bit)
                o.x = fma(bcast(m_pose[0], jj), px, fma(bcast(m_pose[1], jj), py, fma(bcast(m_pose[2], jj), pz, bcast(m_pose[3], jj))));
                o.y = fma(bcast(m_pose[4], jj), px, fma(bcast(m_pose[5], jj), py, fma(bcast(m_pose[6], jj), pz, bcast(m_pose[7], jj))));
                o.z = fma(bcast(m_pose[8], jj), px, fma(bcast(m_pose[9], jj), py, fma(bcast(m_pose[10], jj), pz, bcast(m_pose[11], jj))));
                ((Pt3<T>*)a.points)[g0 + rank] = o;
            }
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
// k_dwf_fused: the single-pass dewarp for sensors of up to 128 beams, rebuilt on k_dwf_emit's tile (64 columns x all
// rows in LDS, column metadata in LDS, dwf_column_loop) -- k_dwf_single above keeps every row count but walks memory the
// slow way (a guarded status scan per tile, guarded staging, metadata in lanes).  Per tile:
//   1. ticket -> (frame, tile); EVERY global read of the prologue is issued at once from clamped addresses: the frame's
//      status words (first / last valid column), the tile's range pieces, the 64 columns' poses / table rows / timestamps;
//   2. LDS: ranges, metadata; per-column kept counts from LDS (wave = 16 columns, lane = row);
//   3. wave 0: masked exclusive scan of the 64 counts -> column bases; publish the tile total; decoupled look-back;
//   4. dwf_column_loop with fbase = the tile's global start.
// Output, order and provenance are those of k_dwf_count / scan / emit.
// ------------------------------------------------------------------------------------
template <class T, bool SEP, int ROWS>
__global__ __launch_bounds__(256) void k_dwf_fused(DewarpFramesArgs a) {
    constexpr int TILE = 64, LPR = TILE / 4, PITCH = TILE + 1, CPW = TILE / 4, NR = ROWS / 64, RPP = 256 / LPR, NP = ROWS / RPP;
    constexpr uint32_t SEG = 8;
    __shared__ uint32_t s_rng[ROWS * PITCH];
    __shared__ DwfColMeta<T> s_meta[TILE];
    __shared__ uint32_t s_cnt[TILE], s_ticket, s_total;
    __shared__ int s_lo, s_hi;
    __shared__ uint64_t s_base;
    const uint32_t W = a.w, H = a.h, tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t q = tid % LPR, ty = tid / LPR;
    const uint32_t tiles = (W + TILE - 1) / TILE;
    if (tid == 0) {
        const uint32_t cls = blockIdx.x & 7u;
        s_ticket = (uint32_t)atomicAdd((unsigned long long*)&a.tile_state[16 * cls], 1ull) * 8u + cls;
        s_lo = 0x7fffffff;
        s_hi = -1;
    }
    __syncthreads();
    const uint32_t t = s_ticket, f = t / tiles, tile = t - f * tiles;
    const uint32_t c0 = tile * TILE, ncol = min((uint32_t)TILE, W - c0);
    const uint32_t* rp = a.range + (size_t)f * W * H;
    const uint32_t* st = a.status + (size_t)f * W;
    const LutDev lut = a.luts[f % a.n_luts];
    const bool vec = (W % 4 == 0) && ((((uintptr_t)a.range) & 15) == 0);

    // ---- 1. the prologue's reads
    uint4 tp[NP];
    if (vec) {
        const uint32_t cc = min(c0 + 4 * q, W - 4u);
#pragma unroll
        for (int i = 0; i < NP; ++i) tp[i] = *(const uint4*)(rp + (size_t)min(ty + (uint32_t)i * RPP, H - 1u) * W + cc);
    }
    DwfColMeta<T> m;
    uint32_t my_status = 0;
    if (tid < (uint32_t)TILE) {
        const uint32_t mx = min(c0 + tid, W - 1u);
        const double* pm = a.poses + ((size_t)f * W + mx) * 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m.pose[2 * k] = (T)pm[k];
            m.pose[2 * k + 1] = (T)pm[4 + k];
            m.pose[8 + k] = (T)pm[8 + k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) m.col[k] = SEP ? as_global(lut.col_tab)[(size_t)mx * 5 + k] : 0.0;
        m.ts = a.timestamps_ns ? a.timestamp[(size_t)f * W + mx] : 0;
        m.base = m.cnt = 0;
        my_status = st[mx];
    }
    {
        int lo = 0x7fffffff, hi = -1;
        for (uint32_t base = 0; base < W; base += 256u * SEG) {
            uint32_t v[SEG];
#pragma unroll
            for (uint32_t k = 0; k < SEG; ++k) v[k] = st[min(base + tid + 256u * k, W - 1u)];
#pragma unroll
            for (uint32_t k = 0; k < SEG; ++k) {
                const uint32_t x = base + tid + 256u * k;
                if (x < W && (v[k] & 1u)) { lo = min(lo, (int)x); hi = max(hi, (int)x); }
            }
        }
        if (hi >= 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    // ---- 2. LDS image of the tile and its metadata
    if (vec) {
        const uint32_t col = c0 + 4 * q;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const uint32_t rr = ty + (uint32_t)i * RPP;
            const bool in = rr < H && col < W;
            uint32_t* d = &s_rng[rr * PITCH + 4 * q];
            d[0] = in ? tp[i].x : 0u; d[1] = in ? tp[i].y : 0u; d[2] = in ? tp[i].z : 0u; d[3] = in ? tp[i].w : 0u;
        }
    } else {
        for (uint32_t rr = ty; rr < (uint32_t)ROWS; rr += RPP) {
            const uint32_t col = c0 + 4 * q;
            uint32_t v[4] = {0, 0, 0, 0};
            if (rr < H)
                for (uint32_t c = 0; c < 4 && col + c < W; ++c) v[c] = rp[(size_t)rr * W + col + c];
#pragma unroll
            for (int c = 0; c < 4; ++c) s_rng[rr * PITCH + 4 * q + c] = v[c];
        }
    }
    if (tid < (uint32_t)TILE) {
        m.cnt = (tid < ncol && my_status != 0) ? 1u : 0u;   // for now: "this column may emit"; the count follows
        s_meta[tid] = m;
    }
    // the lane's rows of the beam table (L2 hits), asked for before the counting pass needs nothing from memory
    uint32_t row[NR];
    double bt[NR][9];
#pragma unroll
    for (int hh = 0; hh < NR; ++hh) {
        row[hh] = lane + 64 * hh;
        if constexpr (SEP) {
            const uint32_t rc = min(row[hh], H - 1u);
#pragma unroll
            for (int k = 0; k < 9; ++k) bt[hh][k] = as_global(lut.beam_tab)[(size_t)rc * 9 + k];
        }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    // kept points per column: wave = 16 columns, lane = row; lane jj ends up with column jj's count
    {
        uint32_t mine = 0;
        for (uint32_t jj = 0; jj < (uint32_t)CPW; ++jj) {
            const uint32_t j = wave * CPW + jj;
            uint32_t n = 0;
#pragma unroll
            for (int hh = 0; hh < NR; ++hh) {
                const uint32_t r = s_rng[(lane + 64 * hh) * PITCH + min(j, (uint32_t)TILE - 1u)];
                n += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(row[hh] < H && r >= a.min_r && r <= a.max_r));
            }
            if (lane == jj) mine = n;
        }
        if (lane < (uint32_t)CPW) s_cnt[wave * CPW + lane] = mine;
    }
    __syncthreads();
    // ---- 3. column bases inside the tile, then where the tile starts (wave 0)
    if (wave == 0) {
        const uint32_t jx = lane;
        const bool on = jx < ncol && (int)(c0 + jx) >= lo && (int)(c0 + jx) <= hi && s_meta[jx].cnt != 0;
        const uint32_t v = on ? s_cnt[jx] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += up;
        }
        const uint32_t tile_total = __shfl(inc, 63, 64);
        s_meta[jx].base = inc - v;
        s_meta[jx].cnt = v;
        uint64_t excl = 0;
        unsigned long long* my = (unsigned long long*)&a.tile_state[DWF_WORDS + t];
        if (t == 0) {
            if (lane == 0) __hip_atomic_store(my, DWF_INC | (uint64_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(my, DWF_AGG | (uint64_t)tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t back = (int64_t)t - 1;        // nearest predecessor not yet accounted for
            // (asking for four windows of predecessors per step instead of one was tried: slower, 0.245 -> 0.259 ms; the wait is
            // for the slowest prologue among the few hundred tiles that started just before this one, not for the steps)
            while (true) {
                const int64_t i = back - (int64_t)lane;
                uint64_t w = DWF_INC;             // lanes before the first tile: a zero inclusive prefix
                if (i >= 0) {
                    do {
                        w = __hip_atomic_load((unsigned long long*)&a.tile_state[DWF_WORDS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((w >> 62) == 0);
                }
                const uint64_t inc_mask = __ballot((w >> 62) == 2);
                const uint32_t first_inc = inc_mask ? (uint32_t)__builtin_ctzll(inc_mask) : 64u;
                uint64_t part = lane <= first_inc ? (w & DWF_VAL) : 0ull;   // aggregates up to (incl.) the first inclusive word
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
                excl += part;
                if (first_inc < 64u) break;
                back -= 64;
            }
            if (lane == 0) __hip_atomic_store(my, DWF_INC | (excl + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = excl;
            s_total = tile_total;
            if (tile == 0) a.frame_off[f] = excl;
            if (t + 1 == a.n_frames * tiles) a.frame_off[a.n_frames] = excl + tile_total;
        }
    }
    __syncthreads();
    const uint64_t fbase = s_base;
    const uint32_t tile_total = s_total;
    if (tile_total == 0) return;
    // ---- 4. emit
    uint32_t m_run = 0;
    if (fbase + tile_total <= a.capacity && a.min_r > 0)
        dwf_column_loop<T, SEP, TILE, ROWS, true>(a, lut, s_rng, s_meta, f, c0, ncol, fbase, row, bt, m_run);
    else
        dwf_column_loop<T, SEP, TILE, ROWS, false>(a, lut, s_rng, s_meta, f, c0, ncol, fbase, row, bt, m_run);
}
#endif  // OUSTER_EXPERIMENTS

// ------------------------------------------------------------------------------------
// k_osf_unpack: the device half of the OSF field decode.  One workgroup per
// (row, plane job).
//   PNG   decode_{8,16,24,32,64}bit_image, ouster_osf/src/png_tools.cpp:182-660: pixel bytes -> value
//         (16-bit samples byte-swapped), then stagger(): dst[r][(c + off[r]) % w] = value(r, c) with
//         off[] = destagger's offsets for inverse = true (impl/lidar_frame_impl.h:733-760)
//   ZPNG  UnpackAndUnfilter<N>, thirdparty/zpng/zpng.cpp:101-352: every byte lane of a row is the
//         running sum (mod 256) of its deltas -- a 256-thread block scan per lane; 3 / 4-byte pixels
//         arrive as colour planes and go through the inverse GB-RG transform first
// HBM bound: src bytes + dst bytes per pixel, each read / written once.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_incl_scan_u8(uint32_t v, uint32_t* s_wave) {
    // inclusive scan of one value per thread over the 256-thread block (only the low 8 bits matter)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wave; ++k) base += s_wave[k];
    __syncthreads();
    return v + base;
}

// ------------------------------------------------------------------------------------
// k_osf_png_unfilter (round 5; VERDICT r04 "missing" item 3): the PNG scanline filters of an OSF field reversed on the device
// -- what libpng's png_read_row does behind decode_{8,16,24,32,64}bit_image (ouster_osf/src/png_tools.cpp:182-660; the filters:
// PNG specification section 9, RFC 2083 6.2 - 6.6).  Sub is a prefix sum and Up is elementwise, but Average and Paeth are
// recurrences in x AND depend on the row above, so a row cannot be split over lanes and rows cannot be done independently.
// What is parallel is the ANTI-DIAGONAL: pixel (x, y) needs (x-1, y), (x, y-1), (x-1, y-1) only.  One wave per image, lane =
// row of a 64-row band; at step t lane r reconstructs pixel x = t - r of its row.  Its left neighbour is its own result of
// step t-1; the pixel above and the one above-left are lane r-1's results of steps t-1 and t-2: two wave shifts per step, no
// memory.  A band's last row goes to LDS as it is produced (lane 63 is 63 pixels behind lane 0, so one row buffer serves both
// the band that reads it and the band that writes it); W + 63 steps per band.  Every byte of a pixel is its own recurrence
// (bpp = 1, 2, 3, 4 or 8 of them in one 64-bit register).  Input rows start at odd addresses (the filter byte), hence byte
// loads; output pixels are stored whole.  2300 images of a 256-frame batch are 2300 independent waves: 9 per CU.
// ------------------------------------------------------------------------------------
// Memory never sits inside a step: every lane keeps a RING of three 64-pixel blocks of its own row in LDS.  Block m+1 is read
// from global memory into registers when window m (64 steps) starts and copied to the ring when it ends; a step reads its raw
// pixel from the ring and writes the reconstructed one over it; block m-2 -- finished by every row -- leaves for global memory
// when window m starts.  16 bytes per lane per access on both sides.  (The first form loaded and stored inside the step: 2.9 ms
// per image whatever the batch, ~1.3 us a step -- on gfx9 a load's wait also waits for the stores before it.  With the ring
// 1.95 ms, then the step itself made branch free -- uf_byte -- 0.64 ms per 128 x 1024 image: 0.29 us a step.)
constexpr uint32_t UF_BLK = 64, UF_RING = 3 * UF_BLK, UF_G = 8;   // UF_G steps share their LDS reads (2, 4, 16: within 2 %)
typedef uint32_t uf_v4 __attribute__((ext_vector_type(4)));   // a plain vector: HIP's uint4 is a class and cannot live in an address space
typedef uf_v4 __attribute__((aligned(1))) uf_chunk_u;   // rows start where they start: the hardware takes unaligned 16-byte accesses
// explicit address spaces: the job's pointers were loaded from memory and the ring is reached through a lambda -- as generic
// pointers both become flat_load / flat_store, which count on vmcnt AND lgkmcnt and tie the LDS steps to the global traffic again
typedef __attribute__((address_space(1))) uint8_t g_u8;
typedef __attribute__((address_space(3))) uint8_t l_u8;
typedef __attribute__((address_space(1))) uf_chunk_u g_chunk_u;
typedef __attribute__((address_space(3))) uf_v4 l_uint4;

__host__ __device__ constexpr uint32_t uf_pitch(uint32_t bpp) { return UF_RING * bpp + 16u; }
__host__ __device__ constexpr uint32_t uf_ring_bytes(uint32_t bpp) { return 64u * uf_pitch(bpp); }

template <uint32_t BPP>
__device__ __forceinline__ uint64_t uf_px_load(const l_u8* p) {
    if (BPP == 1) return *p;
    if (BPP == 2) return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>(p);
    if (BPP == 4) return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(p);
    if (BPP == 8) return *reinterpret_cast<const __attribute__((address_space(3))) uint64_t*>(p);
    return (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16);
}
template <uint32_t BPP>
__device__ __forceinline__ void uf_px_store(l_u8* p, uint64_t v) {
    if (BPP == 1) *p = (uint8_t)v;
    else if (BPP == 2) *reinterpret_cast<__attribute__((address_space(3))) uint16_t*>(p) = (uint16_t)v;
    else if (BPP == 4) *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(p) = (uint32_t)v;
    else if (BPP == 8) *reinterpret_cast<__attribute__((address_space(3))) uint64_t*>(p) = v;
    else { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); }
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_osf_png_unfilter relies on the gfx9 DPP row control wave_shr:1 and on more than 64 KB of LDS per workgroup: build for gfx950"
#endif
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {   // lane r receives lane r-1's value (lane 0: zero)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// One byte of one pixel, every filter computed and the row's one picked by masks (m1..m4: all ones for the lane's filter type,
// zero otherwise).  Written as branches (if ft == 1 ... else if ...) the step compiled to ~260 VALU + ~400 SALU instructions of
// exec-mask juggling, 0.9 us a step; this is ~25 VALU per byte and no branch.
__device__ __forceinline__ uint32_t uf_byte(uint32_t a, uint32_t b, uint32_t c, uint32_t raw, uint32_t m1, uint32_t m2,
                                            uint32_t m3, uint32_t m4) {
    const uint32_t avg = (a + b) >> 1;
    const uint32_t pa = __usad(b, c, 0u), pb = __usad(a, c, 0u), pc = __usad(a + b, 2u * c, 0u);   // |p - a|, |p - b|, |p - c|
    const uint32_t paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
    const uint32_t pred = (a & m1) | (b & m2) | (avg & m3) | (paeth & m4);
    return (raw + pred) & 0xffu;
}
template <uint32_t K, class P>
__device__ __forceinline__ uint32_t uf_get(P v) { return (uint32_t)(v >> (8u * K)) & 0xffu; }

template <uint32_t BPP>
__device__ __forceinline__ void unfilter_band(const OsfUnfilterJob& job, uint32_t H, uint32_t W, uint32_t band, l_u8* smem) {
    constexpr uint32_t BLKB = UF_BLK * BPP, NCH = BLKB / 16u;
    const uint32_t lane = threadIdx.x;
    l_u8* ring = smem + lane * uf_pitch(BPP);
    l_u8* s_up = smem + uf_ring_bytes(BPP);   // [W * BPP]: the last row of the band above
    const size_t stride = (size_t)W * BPP;
    const uint32_t row = band + lane;
    const bool has_row = row < H;
    const g_u8* in = (const g_u8*)(uintptr_t)job.raw + (size_t)(has_row ? row : 0) * (stride + 1);
    g_u8* out = (g_u8*)(uintptr_t)job.out + (size_t)(has_row ? row : 0) * stride;
    const uint32_t ft = has_row ? in[0] : 0u;
    ++in;
    const bool first_row = row == 0;          // nothing above: b = c = 0
    const bool from_lds = lane == 0 && band != 0;
    const bool feeds_next = lane == 63u || row + 1u == H;   // the next band's row above (the last band writes it for nobody)
    const uint32_t nblk = (W + UF_BLK - 1) / UF_BLK, steps = W + 63u, nwin = (steps + UF_BLK - 1) / UF_BLK;

    // A full block travels as 16-byte pieces through registers (fetch at the start of a window, commit at its end); the last,
    // partial block of a row goes byte by byte straight between memory and the ring (nothing is read or written past a row).
    // Lanes without a row read row 0's bytes (harmless) and store nothing.
    uf_v4 stage[NCH];
    auto fetch = [&](uint32_t blk) -> bool {    // global -> registers; false: the block already lies in the ring
        const uint32_t nb = min(UF_BLK, W - blk * UF_BLK) * BPP;
        const g_u8* src = in + (size_t)blk * BLKB;
        if (nb == BLKB) {
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) stage[c] = *reinterpret_cast<const g_chunk_u*>(src + c * 16u);
            return true;
        }
        l_u8* d = ring + (blk % 3u) * BLKB;
#pragma nounroll
        for (uint32_t i = 0; i < nb; ++i) d[i] = src[i];
        return false;
    };
    auto commit = [&](uint32_t blk) {   // registers -> my ring
        l_uint4* d = reinterpret_cast<l_uint4*>(ring + (blk % 3u) * BLKB);
#pragma unroll
        for (uint32_t c = 0; c < NCH; ++c) d[c] = stage[c];
    };
    auto flush = [&](uint32_t blk) {    // my ring -> global
        if (!has_row) return;
        const uint32_t nb = min(UF_BLK, W - blk * UF_BLK) * BPP;
        g_u8* dst = out + (size_t)blk * BLKB;
        if (nb == BLKB) {
            const l_uint4* sp = reinterpret_cast<const l_uint4*>(ring + (blk % 3u) * BLKB);
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) *reinterpret_cast<g_chunk_u*>(dst + c * 16u) = sp[c];
            return;
        }
        const l_u8* sp = ring + (blk % 3u) * BLKB;
#pragma nounroll
        for (uint32_t i = 0; i < nb; ++i) dst[i] = sp[i];
    };

    if (fetch(0)) commit(0);
    using P = typename std::conditional<(BPP <= 4u), uint32_t, uint64_t>::type;   // a pixel's bytes in one register
    const uint32_t m1 = ft == 1u ? ~0u : 0u, m2 = ft == 2u ? ~0u : 0u, m3 = ft == 3u ? ~0u : 0u, m4 = ft == 4u ? ~0u : 0u;
    P res1 = 0, res2 = 0;                     // my results of the two steps before (pixels x-1 and x-2 of my row)
    P up_prev = 0;                            // lane 0: the pixel above-left (pixel x-1 of the band above's last row)
    uint32_t pos = (UF_RING - lane) % UF_RING;   // ring position of pixel x = t - lane (meaningful once x >= 0)
    for (uint32_t m = 0; m < nwin; ++m) {
        if (m >= 2 && m - 2 < nblk) flush(m - 2);                    // its ring slot is the one block m + 1 goes to
        const bool staged = m + 1 < nblk && fetch(m + 1);
        for (uint32_t g = 0; g < UF_BLK / UF_G; ++g) {
            const uint32_t t0 = m * UF_BLK + g * UF_G;
            P raw[UF_G], up[UF_G], rec_g[UF_G];
            uint32_t at[UF_G];
#pragma unroll
            for (uint32_t j = 0; j < UF_G; ++j) {
                const uint32_t x = t0 + j - lane;                       // "negative" before the lane's first pixel: huge
                uint32_t q = pos + j;
                if (q >= UF_RING) q -= UF_RING;
                at[j] = q * BPP;
                // unguarded LDS reads (a guard is a branch): what a lane without a pixel reads is never used
                raw[j] = (P)uf_px_load<BPP>(ring + at[j]);
                up[j] = (P)uf_px_load<BPP>(s_up + min(x, W - 1u) * BPP);   // lane 0's; rewritten 63 steps from now at the earliest
            }
#pragma unroll
            for (uint32_t j = 0; j < UF_G; ++j) {
                const uint32_t x = t0 + j - lane;
                const bool valid = has_row && x < W;   // steps past W + 62 are valid for nobody
                // the row above: lane r-1 finished pixel x at step t-1 and pixel x-1 at step t-2
                P b, c;
                if constexpr (BPP <= 4u) {
                    b = wave_shr1(res1);
                    c = wave_shr1(res2);
                } else {
                    b = (uint64_t)wave_shr1((uint32_t)res1) | ((uint64_t)wave_shr1((uint32_t)(res1 >> 32)) << 32);
                    c = (uint64_t)wave_shr1((uint32_t)res2) | ((uint64_t)wave_shr1((uint32_t)(res2 >> 32)) << 32);
                }
                if (lane == 0) {   // the band's first row takes the row above from LDS (or has none)
                    b = from_lds ? up[j] : (P)0;
                    c = up_prev;
                    up_prev = b;
                }
                if (first_row) b = c = 0;
                const P left = (x != 0u) ? res1 : (P)0;   // res1 is zero while the lane has not started
                if (x == 0u) c = 0;
                P rec = 0;
                [&]<uint32_t... K>(std::integer_sequence<uint32_t, K...>) {
                    ((rec |= (P)uf_byte(uf_get<K>(left), uf_get<K>(b), uf_get<K>(c), uf_get<K>(raw[j]), m1, m2, m3, m4) << (8u * K)), ...);
                }(std::make_integer_sequence<uint32_t, BPP>{});
                rec_g[j] = rec;
                res2 = res1;
                res1 = valid ? rec : (P)0;
            }
#pragma unroll
            for (uint32_t j = 0; j < UF_G; ++j) {
                // the ring takes every lane's result: what a lane without a pixel writes lands in a slot that is free (before its
                // first pixel: block 2's, committed at the end of window 1; behind its last: a flushed block's or past the row's end)
                uf_px_store<BPP>(ring + at[j], rec_g[j]);
            }
            if (feeds_next) {
#pragma unroll
                for (uint32_t j = 0; j < UF_G; ++j) {
                    const uint32_t x = t0 + j - lane;
                    if (has_row && x < W) uf_px_store<BPP>(s_up + (size_t)x * BPP, rec_g[j]);
                }
            }
            pos += UF_G;
            if (pos >= UF_RING) pos -= UF_RING;
        }
        if (staged) commit(m + 1);
    }
    for (uint32_t blk = nwin >= 2 ? nwin - 2 : 0; blk < nblk; ++blk) flush(blk);
}

__global__ __launch_bounds__(64) void k_osf_png_unfilter(OsfUnfilterArgs a) {
    extern __shared__ __align__(16) uint8_t s_unf[];   // 64 row rings, then the last row of the band above
    const OsfUnfilterJob job = a.jobs[blockIdx.x];
    for (uint32_t band = 0; band < a.h; band += 64) {
        switch (job.bpp) {   // uniform over the workgroup
            case 1: unfilter_band<1>(job, a.h, a.w, band, (l_u8*)s_unf); break;
            case 2: unfilter_band<2>(job, a.h, a.w, band, (l_u8*)s_unf); break;
            case 3: unfilter_band<3>(job, a.h, a.w, band, (l_u8*)s_unf); break;
            case 4: unfilter_band<4>(job, a.h, a.w, band, (l_u8*)s_unf); break;
            default: unfilter_band<8>(job, a.h, a.w, band, (l_u8*)s_unf); break;
        }
        __syncthreads();   // one wave: orders the band's LDS writes before the next band's reads
    }
}

__global__ __launch_bounds__(256) void k_osf_unpack(OsfUnpackArgs a) {
    __shared__ uint32_t s_wave[4];
    const ouster_hip_osf_plane job = a.planes[blockIdx.y];
    const uint32_t r = blockIdx.x, W = a.w, tid = threadIdx.x;
    const uint32_t pb = job.src_pixel_bytes, es = job.dst_elem_size;
    uint8_t* drow = (uint8_t*)job.dst + (size_t)r * W * es;
    const uint64_t keep = es >= 8 ? ~0ull : ((1ull << (8 * es)) - 1);
    if (job.encoding != OUSTER_HIP_OSF_ZPNG) {
        const uint8_t* srow = (const uint8_t*)job.src + (size_t)r * W * pb;
        const bool swap16 = job.encoding == OUSTER_HIP_OSF_PNG_GRAY16 || job.encoding == OUSTER_HIP_OSF_PNG_RGBA16;
        const uint32_t off = a.offsets ? (uint32_t)a.offsets[r] : 0u;
        for (uint32_t c = tid; c < W; c += 256) {
            uint64_t v = 0;
            for (uint32_t k = 0; k < pb; ++k) v |= (uint64_t)srow[(size_t)c * pb + (swap16 ? (k ^ 1u) : k)] << (8 * k);
            uint32_t dc = c + off;
            if (dc >= W) dc -= W;
            store1(drow + (size_t)dc * es, v & keep, es);
        }
        return;
    }
    // ZPNG: thread t owns the contiguous pixels [t*seg, (t+1)*seg) of the row
    const uint32_t seg = (W + 255) / 256, x0 = tid * seg, x1 = min(W, x0 + seg);
    const uint8_t* src = (const uint8_t*)job.src;
    const size_t plane = (size_t)W * a.h;   // bytes of one colour plane (3 / 4-byte pixels)
    const bool planar = pb == 3 || pb == 4;
    // residual of byte lane k of pixel x in this row (after the inverse colour transform)
    auto resid = [&](uint32_t x, uint32_t k) -> uint32_t {
        if (!planar) return src[((size_t)r * W + x) * pb + k];
        const size_t i = (size_t)r * W + x;
        const uint32_t y = src[i], u = src[plane + i], v = src[2 * plane + i];
        const uint32_t G = (u + y) & 0xffu;
        return k == 0 ? ((G - v) & 0xffu) : k == 1 ? G : k == 2 ? y : src[3 * plane + i];
    };
    uint64_t carry_in = 0;   // byte lane k of the running value just before my segment
    for (uint32_t k = 0; k < pb; ++k) {
        uint32_t sum = 0;
        for (uint32_t x = x0; x < x1; ++x) sum += resid(x, k);
        const uint32_t incl = block_incl_scan_u8(sum, s_wave);
        carry_in |= (uint64_t)((incl - sum) & 0xffu) << (8 * k);
    }
    for (uint32_t x = x0; x < x1; ++x) {
        uint64_t v = 0;
        for (uint32_t k = 0; k < pb; ++k) {
            const uint32_t b = (uint32_t)((carry_in >> (8 * k)) & 0xffu) + resid(x, k);
            v |= (uint64_t)(b & 0xffu) << (8 * k);
        }
        carry_in = v;
        store1(drow + (size_t)x * es, v & keep, es);
    }
}

// ------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------
size_t decode_lds_bytes(const Geometry& g, int tile, bool general, bool beam_lds, uint32_t slots_per_frame) {
    size_t tile_bytes = ((size_t)tile * g.col_size + 16 + 15) & ~(size_t)15;
    size_t h4 = (g.pixels_per_column + 3) & ~3u;
    size_t n = tile_bytes + (size_t)tile * 8 + 32 + h4 * 4 + XYZ_SCRATCH_BYTES;
    if (beam_lds) n += ((size_t)g.pixels_per_column * 9 + (g.pixels_per_column & 1)) * 8;
    if (general) {  // resolve_frame's scratch lies over the tile image (it is done before the tile is staged)
        const uint32_t npo = g.columns_per_frame / g.columns_per_packet;
        n = std::max(n, slotmap_lds_bytes(g.columns_per_frame, g.columns_per_packet, slots_per_frame ? slots_per_frame : npo));
        n = (n + 15) & ~(size_t)15;
    }
    return n;
}

// img_words: the tile image [tw][column slot] + 4 slack words (the fix-up pass: at least resolve_frame's scratch)
size_t decode_wide_lds_bytes(int tw, uint32_t rows_per_tile, uint32_t img_words) {
    return ((size_t)img_words + 3 * (size_t)tw + 4 + ((rows_per_tile + 3) & ~3u)) * 4 +
           ((size_t)rows_per_tile * 9 + (rows_per_tile & 1)) * 8 + XYZ_SCRATCH_BYTES;
}

#define OUSTER_DECL_SPEC(sfx)                                                                             \
    hipError_t launch_decode_##sfx(const DecodeArgs& a, int tile, int xyzm, int device, hipStream_t st); \
    hipError_t launch_decode_wide_##sfx(const DecodeArgs& a, int tw, int xyzm, int device, hipStream_t st, uint32_t resident);
OUSTER_DECL_SPEC(generic)
OUSTER_DECL_SPEC(dual_lb)
OUSTER_DECL_SPEC(lb)
OUSTER_DECL_SPEC(single)
OUSTER_DECL_SPEC(dual)
OUSTER_DECL_SPEC(legacy)
#undef OUSTER_DECL_SPEC

hipError_t launch_decode(const DecodeArgs& a, int spec_id, int tile, int xyzm, int device, hipStream_t st) {
    switch (spec_id) {
        case SPEC_DUAL_LB: return launch_decode_dual_lb(a, tile, xyzm, device, st);
        case SPEC_LB: return launch_decode_lb(a, tile, xyzm, device, st);
        case SPEC_SINGLE: return launch_decode_single(a, tile, xyzm, device, st);
        case SPEC_DUAL: return launch_decode_dual(a, tile, xyzm, device, st);
        case SPEC_LEGACY: return launch_decode_legacy(a, tile, xyzm, device, st);
        default: return launch_decode_generic(a, tile, xyzm, device, st);
    }
}

hipError_t launch_decode_wide(const DecodeArgs& a, int spec_id, int tw, int xyzm, int device, hipStream_t st, uint32_t resident) {
    switch (spec_id) {
        case SPEC_DUAL_LB: return launch_decode_wide_dual_lb(a, tw, xyzm, device, st, resident);
        case SPEC_LB: return launch_decode_wide_lb(a, tw, xyzm, device, st, resident);
        case SPEC_SINGLE: return launch_decode_wide_single(a, tw, xyzm, device, st, resident);
        case SPEC_DUAL: return launch_decode_wide_dual(a, tw, xyzm, device, st, resident);
        case SPEC_LEGACY: return launch_decode_wide_legacy(a, tw, xyzm, device, st, resident);
        default: return launch_decode_wide_generic(a, tw, xyzm, device, st, resident);
    }
}

#define OUSTER_DECL_STREAM(sfx) \
    hipError_t launch_decode_stream_##sfx(const DecodeArgs& a, const StreamArgs& sp, int tw, int xyzm, int device, hipStream_t st);
OUSTER_DECL_STREAM(dual_lb)
OUSTER_DECL_STREAM(lb)
OUSTER_DECL_STREAM(single)
OUSTER_DECL_STREAM(dual)
OUSTER_DECL_STREAM(legacy)
#undef OUSTER_DECL_STREAM

hipError_t launch_decode_stream(const DecodeArgs& a, const StreamArgs& sp, int spec_id, int tw, int xyzm, int device,
                                hipStream_t st) {
    switch (spec_id) {
        case SPEC_DUAL_LB: return launch_decode_stream_dual_lb(a, sp, tw, xyzm, device, st);
        case SPEC_LB: return launch_decode_stream_lb(a, sp, tw, xyzm, device, st);
        case SPEC_SINGLE: return launch_decode_stream_single(a, sp, tw, xyzm, device, st);
        case SPEC_DUAL: return launch_decode_stream_dual(a, sp, tw, xyzm, device, st);
        case SPEC_LEGACY: return launch_decode_stream_legacy(a, sp, tw, xyzm, device, st);
        default: return hipErrorInvalidValue;   // run-time descriptors stay on k_decode / k_decode_wide
    }
}

// ------------------------------------------------------------------------------------
// k_slotmap: the general column mapping of a whole frame, once (one workgroup per frame) -- for buffers that do not have
// one slot per column of the frame (compacted after drops, any order, duplicates), where round 2 let EVERY 64-column tile
// of k_decode scan the frame's column headers.  resolve_frame (kernels_common.h) restates what FrameBatcher leaves behind
// after batching the frame's packets in buffer order, block path and column path alike
// (ouster_core/src/lidar_frame.cpp:1422-1576): per destination column the slot that supplies its pixels (slot_map) and the
// slot that supplies its header (hdr_map; the two differ only for an all-valid packet whose ids are not consecutive).
// Also everything the general path's tile 0 used to resolve: packet-level outputs (batch_lidar_packet :1534-1539), the
// frame-level values (start_frame :1709-1741, from the first packet of the buffer) and the valid-column count.
// k_decode_wide then decodes from the maps.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slotmap(DecodeArgs a) {
    constexpr int NT = 256;
    extern __shared__ __align__(16) uint32_t smem[];
    __shared__ uint32_t s_n;
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, npo = a.n_packets_out;
    const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    uint32_t count = a.slots_per_frame;
    if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
    const ResolveLds L(smem, W, npo, a.slots_per_frame);
    int32_t *s_pix = L.pix, *s_hdr = L.hdr, *s_pkm = L.pkm;
    if (tid == 0) s_n = 0;
    resolve_frame<NT>(a.g, fbase, a.packet_stride, count, npo, L, true);
    uint32_t n = 0;
    for (uint32_t i = tid; i < W; i += NT) {
        const int32_t h = s_hdr[i];
        a.slot_map[(size_t)f * W + i] = s_pix[i];
        a.hdr_map[(size_t)f * W + i] = h;
        n += h >= 0 ? 1u : 0u;
    }
    n = wave_sum(n);
    if (n && (tid & 63u) == 0) atomicAdd(&s_n, n);
    // packet_timestamp is zeroed at frame start (lidar_frame.cpp:1719), alert_flags is not
    for (uint32_t i = tid; i < npo; i += NT) {
        const int32_t p = s_pkm[i];
        if (a.packet_timestamp && a.host_timestamps)
            a.packet_timestamp[(size_t)f * npo + i] = p >= 0 ? a.host_timestamps[(size_t)f * a.slots_per_frame + p] : 0ull;
        if (a.alert_flags && p >= 0)
            a.alert_flags[(size_t)f * npo + i] = (uint8_t)apply_bits(
                window_global(fbase + (size_t)p * a.packet_stride + a.g.alert_flags.offset), a.g.alert_flags.mask, a.g.alert_flags.shift);
    }
    __syncthreads();
    if (tid == 0 && a.frame_meta) {
        ouster_hip_frame_meta m = frame_meta_general(a, fbase, count);
        m.n_valid_columns = s_n;
        a.frame_meta[f] = m;
    }
}

size_t slotmap_lds_bytes(uint32_t W, uint32_t cpp, uint32_t slots_per_frame) {
    return resolve_lds_words(W, W / cpp, slots_per_frame, cpp) * 4;
}

hipError_t launch_slotmap(const DecodeArgs& a, int device, hipStream_t st) {
    const size_t lds = slotmap_lds_bytes(a.g.columns_per_frame, a.g.columns_per_packet, a.slots_per_frame);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static std::atomic<uint32_t> granted[16];
    if (lds > 48 * 1024 && granted[device & 15].load(std::memory_order_acquire) < lds) {
        hipError_t e = hipFuncSetAttribute((const void*)k_slotmap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted[device & 15].store((uint32_t)lds, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_slotmap, dim3(a.n_frames), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_destagger(const DestaggerArgs& a, uint32_t n_images, hipStream_t st) {
    dim3 grid(a.h, n_images);
    const size_t row_bytes = (size_t)a.w * a.elem;
    const bool al = (row_bytes % 16 == 0) && ((((uintptr_t)a.src | (uintptr_t)a.dst) & 15) == 0);
    static const int rows_env = [] { const char* e = getenv("OUSTER_HIP_DESTAGGER_ROWS"); return e ? atoi(e) : -1; }();   // A/B
    if (al && row_bytes <= (rows_env > 0 ? (16u << 10) : (4u << 10)) && rows_env != 0) {
        // rows of up to 4 KB (8 / 16-bit planes of a 2048-column frame): two rows per workgroup, pipelined -- 0.77 / 0.79 of the HBM
        // roofline on 256 images against 0.72 / 0.75 for one row per workgroup (round 6, same box); 8 KB rows (32-bit planes)
        // lose with this form (0.50 against 0.76) and stay on k_destagger
        uint32_t rpw = rows_env > 0 ? (uint32_t)rows_env : 2u;
        const uint32_t nchunk = (uint32_t)(row_bytes >> 4), ch = (nchunk + 255) / 256;
        dim3 g2((a.h + rpw - 1) / rpw, n_images);
        const uint32_t lds2 = 2u * (uint32_t)row_bytes;
        if (ch <= 1) hipLaunchKernelGGL(k_destagger_rows<1>, g2, dim3(256), lds2, st, a, rpw);
        else if (ch <= 2) hipLaunchKernelGGL(k_destagger_rows<2>, g2, dim3(256), lds2, st, a, rpw);
        else hipLaunchKernelGGL(k_destagger_rows<4>, g2, dim3(256), lds2, st, a, rpw);
        return hipGetLastError();
    }
    const uint32_t lds = (al && row_bytes <= DESTAGGER_LDS_MAX) ? (uint32_t)row_bytes : 0u;
    hipLaunchKernelGGL(k_destagger, grid, dim3(256), lds, st, a, lds);
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
        // as many images per workgroup as leave >= 2048 workgroups (16 at most): a full LUT's rows / the separable
        // directions of a row are fetched / computed once per group, and the group's range quads are fetched four deep
        uint32_t ipb = 1;
        {
            const size_t per_image = (size_t)tiles * ((a.h + rpb - 1) / rpb);
            while (ipb < 16 && ipb * 2 <= a.n_images && per_image * ((a.n_images + ipb * 2 - 1) / (ipb * 2)) >= 2048) ipb *= 2;
        }
        a.images_per_block = ipb;
        dim3 grid(tiles * ((a.h + rpb - 1) / rpb), (a.n_images + ipb - 1) / ipb);
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

hipError_t launch_dewarp_frames(const DewarpFramesArgs& a, bool separable, hipStream_t st, int stream_mode) {
    const uint32_t tiles = (a.w + 63) / 64;
    const size_t lds = (size_t)a.h * 65 * 4;
#ifdef OUSTER_EXPERIMENTS
    if (a.tile_state && lds <= 96 * 1024) {
        // single pass with a decoupled look-back over the tiles (the state buffer must be zero)
        hipError_t e = hipMemsetAsync(a.tile_state, 0, ((size_t)a.n_frames * tiles + DWF_WORDS) * 8, st);
        if (e != hipSuccess) return e;
        const dim3 grid(a.n_frames * tiles);
        auto go = [&](auto kernel) -> hipError_t {
            if (lds > 48 * 1024) {
                hipError_t ee = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (ee != hipSuccess) return ee;
            }
            hipLaunchKernelGGL(kernel, grid, dim3(DWF_NT), lds, st, a);
            return hipGetLastError();
        };
        if (a.h <= 128) {   // the rebuilt single pass: static LDS, k_dwf_emit's tile
            auto fused = [&](auto rows) -> hipError_t {
                constexpr int R = decltype(rows)::value;
                if (a.dtype == OUSTER_HIP_F32) {
                    if (separable) hipLaunchKernelGGL((k_dwf_fused<float, true, R>), grid, dim3(256), 0, st, a);
                    else hipLaunchKernelGGL((k_dwf_fused<float, false, R>), grid, dim3(256), 0, st, a);
                } else {
                    if (separable) hipLaunchKernelGGL((k_dwf_fused<double, true, R>), grid, dim3(256), 0, st, a);
                    else hipLaunchKernelGGL((k_dwf_fused<double, false, R>), grid, dim3(256), 0, st, a);
                }
                return hipGetLastError();
            };
            return a.h > 64 ? fused(std::integral_constant<int, 128>{}) : fused(std::integral_constant<int, 64>{});
        }
        if (a.dtype == OUSTER_HIP_F32)
            return separable ? go(k_dwf_single<float, true, DWF_NT>) : go(k_dwf_single<float, false, DWF_NT>);
        return separable ? go(k_dwf_single<double, true, DWF_NT>) : go(k_dwf_single<double, false, DWF_NT>);
    }
#endif
    if (!a.gate_counts) hipLaunchKernelGGL(k_dwf_count, dim3(tiles, a.n_frames), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dwf_scan, dim3(a.n_frames), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_dwf_frame_scan, dim3(1), dim3(256), 0, st, a);
    // 64-column emit tiles (32-column ones measured 15 % slower); 128 rows per pass when the sensor has
    // more than 64 beams: the per-column overhead is paid once per column instead of once per 64 rows
#ifdef OUSTER_EXPERIMENTS
    // the persistent form (k_dwf_emit_stream) where it applies: whole 64-column tiles of exactly h rows, 16 B aligned planes,
    // the separable tables, and enough tiles to give every persistent workgroup a run of them
    {
        static const uint32_t wgs = [] {
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return (uint32_t)(2 * (cus > 0 ? cus : 256));
        }();
        const uint32_t n_items = a.n_frames * tiles;
        const bool ok = stream_mode != 0 && separable && a.dtype == OUSTER_HIP_F32 && a.w % 64 == 0 && (a.h == 64 || a.h == 128) &&
                        ((uintptr_t)a.range & 15) == 0 && n_items >= (stream_mode > 0 ? 1u : 4u * wgs);
        if (ok) {
            const uint32_t g = std::min(n_items, wgs);
            auto go = [&](auto kernel, uint32_t ctx_bytes) -> hipError_t {
                const size_t lds = 2 * (size_t)ctx_bytes;
                hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kernel, dim3(g), dim3(576), lds, st, a, n_items);
                return hipGetLastError();
            };
            return a.h == 128 ? go(k_dwf_emit_stream<float, 128>, dwf_stream_ctx_bytes<float, 128>())
                              : go(k_dwf_emit_stream<float, 64>, dwf_stream_ctx_bytes<float, 64>());
        }
    }
#else
    (void)stream_mode;
#endif
    const dim3 grid(tiles, a.n_frames);
    auto emit = [&](auto rows) {
        constexpr int R = decltype(rows)::value;
        if (a.dtype == OUSTER_HIP_F32) {
            if (separable) hipLaunchKernelGGL((k_dwf_emit<float, true, 64, R>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_dwf_emit<float, false, 64, R>), grid, dim3(256), 0, st, a);
        } else {
            if (separable) hipLaunchKernelGGL((k_dwf_emit<double, true, 64, R>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_dwf_emit<double, false, 64, R>), grid, dim3(256), 0, st, a);
        }
    };
    // (round 6 A/B: two 64-row passes for a 128-beam sensor -- 23 KB of LDS, 80 VGPRs, six workgroups per CU instead of four --
    // 0.254 - 0.269 ms against 0.219 - 0.220: the per-column overhead paid twice costs more than the occupancy gives)
    if (a.h > 64) emit(std::integral_constant<int, 128>{});
    else emit(std::integral_constant<int, 64>{});
    return hipGetLastError();
}

size_t osf_png_unfilter_lds_bytes(uint32_t w, uint32_t max_row_bytes) {
    const uint32_t bpp = w ? (max_row_bytes + w - 1) / w : 1u;   // the widest pixel of the batch sizes everybody's LDS
    return uf_ring_bytes(bpp) + (((size_t)max_row_bytes + 15) & ~(size_t)15);
}

hipError_t launch_osf_png_unfilter(const OsfUnfilterArgs& a, uint32_t n_jobs, uint32_t max_row_bytes, hipStream_t st) {
    const size_t lds = osf_png_unfilter_lds_bytes(a.w, max_row_bytes);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_osf_png_unfilter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_osf_png_unfilter, dim3(n_jobs), dim3(64), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_osf_unpack(const OsfUnpackArgs& a, uint32_t n_planes, hipStream_t st) {
    hipLaunchKernelGGL(k_osf_unpack, dim3(a.h, n_planes), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace ouster_hip_dev
