// k_decode_stream.hip -- the optimistic decode pass as a PERSISTENT, double-buffered kernel, gfx950.
// Compiled once per static packet-profile specialisation (-DOUSTER_SPEC_ID=1..5, see the Makefile).
//
// Same work, same tiles and same results as k_decode_wide (k_decode.hip): fused field decode + destagger +
// cartesian of the reference loops
//   PacketFormat::block_field            ouster_core/src/parsing.cpp:628-657
//   FrameBatcher::parse_by_block         ouster_core/src/lidar_frame.cpp:1468-1528
//   destagger_into<T>                    ouster_core/include/ouster/core/impl/lidar_frame_impl.h:733-760
//   impl::cartesianT<T>                  ouster_core/include/ouster/core/impl/cartesian.h:36-66
// on TW columns x TR rows per tile -- but a k_decode_wide workgroup spends a third (8 B/px) to almost half
// (12 B/px) of its life waiting for its 48 - 64 KB tile while the memory system is saturated with other
// workgroups' stores (profiles/r02/phase_timing_*.txt), and more workgroups per CU do not hide it.  Here:
//   * ONE 512-thread workgroup per CU lives for the whole launch and walks a list of tiles that share its
//     column range, so everything that depends on the columns only (source offsets of every 16 B cell it
//     fetches, its lanes' xyz column constants, the LDS offsets of its pixels) is computed once;
//   * TWO tile contexts in LDS.  While the row loop works on tile i, tile i+1 arrives by LDS-DMA
//     (global_load_lds_dwordx4 / _dword: no VGPR round trip, nothing for the row loop to wait on) --
//     pixels, column-header words, packet-level words, destagger offsets and the per-beam xyz rows alike,
//     so that between two tiles no wave ever consumes an ordinary vector load (a consumer would wait for
//     everything older in the in-order vmcnt queue, i.e. for the prefetch);
//   * the DMA statements are inline asm: hipcc neither counts them nor guards LDS reads against them, the
//     kernel does (s_waitcnt + workgroup barrier at the top of every tile).
// LDS image of a tile: an LDS-DMA writes wave-uniform base + lane * 16, so the image is lane-linear and
// cannot be padded per column.  Column j's TR-row piece lives in a block of `ncell` consecutive 16 B cells
// (the aligned cells that cover it: the piece keeps its own 16 B phase, as in k_decode_wide), block index
// j/4 + (TW/4)*(j%4): the four columns of a lane are TW/4 blocks apart and the lanes of a row read
// consecutive blocks (ncell is odd for the 8 and 16 B/px profiles -> 4-way bank conflicts at worst instead of 16).
#include <hip/hip_runtime.h>

#include <atomic>

#include "kernels_common.h"

#ifndef OUSTER_SPEC_ID
#error "compile with -DOUSTER_SPEC_ID=1..5"
#endif

namespace ouster_hip_dev {

#if OUSTER_SPEC_ID == 1
using SpecT = SpecDualLB;
#define OUSTER_SPEC_FN(name) name##_dual_lb
#elif OUSTER_SPEC_ID == 2
using SpecT = SpecLB;
#define OUSTER_SPEC_FN(name) name##_lb
#elif OUSTER_SPEC_ID == 3
using SpecT = SpecSingle;
#define OUSTER_SPEC_FN(name) name##_single
#elif OUSTER_SPEC_ID == 4
using SpecT = SpecDual;
#define OUSTER_SPEC_FN(name) name##_dual
#else
using SpecT = SpecLegacy;
#define OUSTER_SPEC_FN(name) name##_legacy
#endif

// One LDS-DMA wave-instruction: lane l's 16 (4) bytes at `g` land at LDS byte address lds + 16 (4) * l.
// M0 carries the LDS base; it is compiler-reserved, so it is saved, written and restored inside the one
// statement that uses it (cdna_hip_programming.md section 5.7).  `lds` must be wave-uniform.
__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
    uint32_t keep;
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds4(const void* g, uint32_t lds) {
    uint32_t keep;
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

// Uniform reads of launch-constant tables inside the persistent loop go through the constant address space: the loop
// also stores, so hipcc cannot prove an ordinary global read unclobbered and would issue it as a VECTOR load -- whose
// consumer then waits on the in-order vmcnt queue, i.e. for the tile prefetch issued just before it.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p) {
    return (const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

// field value from the fetched header dwords of column / packet `j` (slot-major table, `n` entries per slot)
__device__ __forceinline__ uint64_t plan_field(const uint32_t* tab, uint32_t n, uint32_t j, const FieldPlan& p,
                                               const ouster_hip_bits& b) {
    const uint32_t d0 = p.slot[0] >= 0 ? tab[(uint32_t)p.slot[0] * n + j] : 0u;
    const uint32_t d1 = p.slot[1] >= 0 ? tab[(uint32_t)p.slot[1] * n + j] : 0u;
    const uint32_t d2 = p.slot[2] >= 0 ? tab[(uint32_t)p.slot[2] * n + j] : 0u;
    return apply_bits(funnel3(d0, d1, d2, p.sh), b.mask, b.shift);
}

template <class S, int TW, int XYZM>
__global__ __launch_bounds__(512) void k_decode_stream(DecodeArgs a, StreamArgs sp) {
    constexpr int NT = 512, NW = NT / 64, QPR = TW / 4;
    constexpr int MAXC = 9;                        // pixel-image DMA instructions per wave (72 KB image at most)
    constexpr uint32_t chan = S::chan;
    constexpr int NCLS = (TW + NT - 1) / NT;       // columns a thread classifies
    static_assert(S::is_static && TW % 64 == 0, "static profiles, whole waves of columns");
    extern __shared__ __align__(16) uint32_t smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t W = a.g.columns_per_frame;
    const uint32_t cpp = a.g.columns_per_packet, col_size = a.g.col_size;
    const uint32_t TR = sp.tr, nch = sp.nch, npo = a.n_packets_out, ncell = sp.ncell;
    const uint32_t CT = a.tiles_per_frame;         // column tiles of a frame (W / TW)
    // workgroup -> (XCD, column tile, group): block b runs on XCD b % 8; the CT workgroups of a group are
    // neighbours there and write whole rows together, frame f belongs to XCD f % 8 (as in k_decode_wide)
    const uint32_t xcd = blockIdx.x & 7u, kx = blockIdx.x >> 3;
    const uint32_t ct = kx % CT, grp = kx / CT, NG = sp.groups;
    const uint32_t c0 = ct * TW;
    if (xcd >= a.n_frames) return;
    const uint32_t items = ((a.n_frames - xcd + 7u) >> 3) * nch;   // (frame, row chunk) pairs of this XCD
    if (grp >= items) return;
    // the i-th item of this group.  order 0: the groups of an XCD take consecutive items (they work on the row chunks
    // of one frame together and share its packets in L2); order 1: every group owns a contiguous run of items (its own
    // frames, far apart from the other groups' in every output plane)
    const uint32_t per_grp = (items + NG - 1) / NG;
    const uint32_t n_mine = sp.order == 1 ? (grp * per_grp < items ? min(per_grp, items - grp * per_grp) : 0u)
                                          : (items - grp + NG - 1) / NG;
    auto item = [&](uint32_t i) { return sp.order == 1 ? grp * per_grp + i : grp + NG * i; };
    if (n_mine == 0) return;

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;               // LDS byte address of the dynamic segment
    uint32_t* s_colofs = (uint32_t*)((uint8_t*)smem + sp.fixed_off);   // [TW] byte offset of column c0+j in a frame buffer
    uint32_t* s_valid = s_colofs + TW;                                 // [TW] 1 = received, valid, at home
    uint32_t* s_gate = s_valid + TW;                                   // [TW] range-gate counters
    uint32_t* s_acc = s_gate + TW;                                     // [2][4] valid / stray / dead columns, by tile parity

    // ---- once per workgroup -------------------------------------------------------------------
    auto col_ofs = [&](uint32_t c) {
        const uint32_t p = c / cpp, ic = c - p * cpp;
        return p * (uint32_t)a.packet_stride + a.g.packet_header_size + ic * col_size;
    };
    for (uint32_t j = tid; j < (uint32_t)TW; j += NT) s_colofs[j] = col_ofs(c0 + j);
    if (tid < 8) s_acc[tid] = 0;
    // source byte offset (from frame base + row offset) of the cells this lane fetches; the host made sure that
    // frame bases, the frame stride and TR*chan are multiples of 16, so a cell's phase never changes
    uint32_t cellofs[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const uint32_t id = wave + (uint32_t)NW * i, cidx = id * 64u + lane;
        uint32_t ofs = 0;
        if (id < sp.npix_instr && cidx < (uint32_t)TW * ncell) {
            const uint32_t block = cidx / ncell, ch = cidx - block * ncell;
            const uint32_t j = (block % QPR) * 4u + block / QPR;
            const uint32_t base = col_ofs(c0 + j) + a.g.col_header_size;
            ofs = (base & ~15u) + ch * 16u;
        }
        cellofs[i] = ofs;
    }
    // LDS dword offsets of row 0 of my four columns
    const uint32_t q = tid % QPR;
    uint32_t px_dw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t base = col_ofs(c0 + q * 4u + c) + a.g.col_header_size;
        px_dw[c] = (q + (uint32_t)c * QPR) * ncell * 4u + ((base & 15u) >> 2);
    }
    const uint64_t tag = a.frame_state[FS_SEQ] + 1;   // this call's stray tag
    ColConst cc;
    uint32_t cur_lut = 0xffffffffu;
    __syncthreads();

    // ---- the DMA of one tile into context b ------------------------------------------------------
    auto issue = [&](uint32_t s, uint32_t b) {
        const uint32_t m = s / nch, rc = s - m * nch, f = xcd + 8u * m;
        const uint8_t* fb = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
        const uint8_t* src0 = fb + rc * TR * chan;
        const uint32_t ldsb = lds0 + b * sp.ctx_bytes;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const uint32_t id = wave + (uint32_t)NW * i;
            if (id < sp.npix_instr) glds16(src0 + cellofs[i], ldsb + id * 1024u);
        }
        // the small tables, dealt round the waves (4..7 fetch one pixel instruction less)
        uint32_t n = 4;
        auto mine = [&]() { return ((n++) & (uint32_t)(NW - 1)) == wave; };
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((uint32_t)k >= sp.n_hdr) break;
#pragma unroll
            for (int g = 0; g < TW / 64; ++g)
                if (mine()) glds4(fb + s_colofs[g * 64 + lane] + sp.hdr_dw[k] * 4u, ldsb + sp.hdr_off + ((uint32_t)k * TW + g * 64u) * 4u);
        }
        const uint32_t np = (uint32_t)TW / cpp;                       // packets under this tile (<= 64)
        const uint32_t pa = c0 / cpp + (lane < np ? lane : np - 1u);  // lanes past the last packet re-fetch it
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((uint32_t)k >= sp.n_pkt) break;
            if (mine()) glds4(fb + (size_t)pa * a.packet_stride + sp.pkt_dw[k] * 4u, ldsb + sp.pkt_off + (uint32_t)k * 256u);
        }
        if (a.host_timestamps && a.packet_timestamp) {
            const uint8_t* ht = (const uint8_t*)(a.host_timestamps + (size_t)f * a.slots_per_frame + pa);
            if (mine()) glds4(ht, ldsb + sp.pkt_off + 4u * 256u);
            if (mine()) glds4(ht + 4, ldsb + sp.pkt_off + 5u * 256u);
        }
        if (a.any_destagger) {
            for (uint32_t k = 0; k * 64u < TR; ++k) {
                const uint32_t i = k * 64u + lane;
                if (mine()) glds4(a.dst_offsets + rc * TR + (i < TR ? i : TR - 1u), ldsb + sp.off_off + k * 256u);
            }
        }
        if (XYZM == 1 || XYZM == 2) {
            const uint32_t* bt = (const uint32_t*)(as_const(a.luts)[f % a.n_luts].beam_tab + (size_t)rc * TR * 9u);
            const uint32_t nb = TR * 18u;
            for (uint32_t k = 0; k * 64u < nb; ++k) {
                const uint32_t i = k * 64u + lane;
                if (mine()) glds4(bt + (i < nb ? i : nb - 1u), ldsb + sp.beam_off + k * 256u);
            }
        }
    };

    // frame-level values (start_frame, lidar_frame.cpp:1709-1741) of the frames whose first row chunk this workgroup
    // owns: ordinary loads, so all of them up front, one frame per thread
    if (ct == 0 && a.frame_meta) {
        for (uint32_t k = tid; k < n_mine; k += NT) {
            const uint32_t sk = item(k), m = sk / nch;
            if (sk != m * nch) continue;
            const uint32_t f = xcd + 8u * m;
            uint32_t count = a.slots_per_frame;
            if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
            a.frame_meta[f] = frame_meta_first_present(a.g, a.packets + (size_t)f * a.slots_per_frame * a.packet_stride,
                                                       a.packet_stride, count);
        }
    }

    // ---- the tiles of this workgroup ---------------------------------------------------------------
    issue(item(0), 0);
    for (uint32_t it = 0; it < n_mine; ++it) {
        const uint32_t b = it & 1u, s = item(it);
        const uint32_t m = s / nch, rc = s - m * nch, f = xcd + 8u * m;
        const uint32_t r0 = rc * TR;
        // Tile `it` was requested one tile ago (or just now).  vmcnt retires in issue order, loads and stores
        // alike: after >= 63 younger stores of this wave the 6-bit counter itself proves that the DMA has
        // landed (sp.wait0 == 0); otherwise wait for everything.  The barrier then covers the other waves'
        // DMA and, for the context about to be refilled, their reads of tile it-1.
        if (it == 0 || sp.wait0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const uint8_t* fb = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
        uint32_t count = a.slots_per_frame;
        if (a.packet_counts) count = min(as_const(a.packet_counts)[f], a.slots_per_frame);
        LutDev lut{};
        if (XYZM != 0) {
            const auto* lp = as_const(a.luts) + f % a.n_luts;
            lut.beam_tab = lp->beam_tab; lut.col_tab = lp->col_tab; lut.n = lp->n;
        }
        // the one ordinary vector load left in the loop (a new LUT's column constants: once per workgroup unless a
        // batch interleaves sensors on an XCD) is consumed HERE, before the next tile is requested: a wait behind the
        // request would be a wait for the request (in-order vmcnt)
        if (XYZM == 1 || XYZM == 2) {
            const uint32_t li = f % a.n_luts;
            if (li != cur_lut) {
                cur_lut = li;
                load_colconst(cc, lut, c0 + q * 4u, W);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    asm volatile("" : "+v"(cc.cx[c]), "+v"(cc.sx[c]), "+v"(cc.kc[c][0]), "+v"(cc.kc[c][1]), "+v"(cc.kc[c][2]));
                }
            }
        }
        if (it + 1 < n_mine) issue(item(it + 1), b ^ 1u);

        uint32_t* s_pix = (uint32_t*)((uint8_t*)smem + b * sp.ctx_bytes);
        const uint32_t* s_hdr = s_pix + (sp.hdr_off >> 2);
        const uint32_t* s_pkt = s_pix + (sp.pkt_off >> 2);
        const int32_t* s_off = (const int32_t*)(s_pix + (sp.off_off >> 2));
        const double* s_beam = (const double*)(s_pix + (sp.beam_off >> 2));
        uint32_t* acc = s_acc + b * 4u;
        if (tid < 4) s_acc[(b ^ 1u) * 4u + tid] = 0;     // the next tile's counters
        if (a.gate_counts)
            for (uint32_t j = tid; j < (uint32_t)TW; j += NT) s_gate[j] = 0;

        // ---- classify my columns from their fetched header words (slot c holds column c: DESIGN.md 3.1)
#pragma unroll
        for (int kc = 0; kc < NCLS; ++kc) {
            const uint32_t j = tid + (uint32_t)kc * NT;
            if (j >= (uint32_t)TW) break;          // whole waves: TW is a multiple of 64
            const uint32_t c = c0 + j, p = c / cpp;
            const bool present = p < count;
            const uint32_t m_id = (uint16_t)plan_field(s_hdr, TW, j, sp.mid, a.g.col_measurement_id);
            const uint32_t st = (uint32_t)plan_field(s_hdr, TW, j, sp.st, a.g.col_status);
            const bool live = present && (st & 1u) && m_id < W;
            bool stray = live && m_id != c;
            const bool v = live && !stray;
            s_valid[j] = v ? 1u : 0u;
            if (rc == 0) {
                if (a.hdr_words) a.hdr_words[(size_t)f * W + c] = present ? (m_id | ((st & 1u) << 16)) : 0u;
                if (c == p * cpp) {   // batch_lidar_packet, lidar_frame.cpp:1534-1539
                    const bool want_pk = a.packet_timestamp || a.alert_flags;
                    const bool home = present && m_id / cpp == p;
                    if (present && !home && want_pk && m_id / cpp < npo) stray = true;
                    const uint32_t pl = p - c0 / cpp;
                    if (a.packet_timestamp && a.host_timestamps)
                        a.packet_timestamp[(size_t)f * npo + p] =
                            home ? ((uint64_t)s_pkt[5 * 64 + pl] << 32) | s_pkt[4 * 64 + pl] : 0ull;
                    if (a.alert_flags && home)
                        a.alert_flags[(size_t)f * npo + p] = (uint8_t)plan_field(s_pkt, 64, pl, sp.alert, a.g.alert_flags);
                }
                if (a.timestamp)
                    a.timestamp[(size_t)f * W + c] = v ? plan_field(s_hdr, TW, j, sp.ts, a.g.col_timestamp) : 0ull;
                if (a.measurement_id) a.measurement_id[(size_t)f * W + c] = v ? (uint16_t)c : (uint16_t)0;
                if (a.status) a.status[(size_t)f * W + c] = v ? st : 0u;
            }
            const uint64_t bv = __ballot(v), bs = __ballot(stray), bd = __ballot(!v);
            if (lane == 0) {
                if (rc == 0) {
                    if (bv) atomicAdd(&acc[0], (uint32_t)__popcll(bv));
                    if (bs) atomicAdd(&acc[1], (uint32_t)__popcll(bs));
                }
                if (bd) atomicAdd(&acc[2], (uint32_t)__popcll(bd));
            }
        }
        __syncthreads();
        if (rc == 0 && tid == 0) {
            if (acc[1]) flag_frame(a, f, tag);
            if (a.frame_meta) a.tile_valid[(size_t)f * CT + ct] = (uint16_t)acc[0];
            if (f == 0 && ct == 0) a.frame_state[FS_TAG] = tag;
        }
        // columns that were not received (or are invalid / not at home) decode as zeros: blank their blocks once,
        // so that the row loop needs no per-pixel select.  Rare, hence the uniform test.
        if (acc[2] != 0) {
            for (uint32_t j = tid >> 2; j < (uint32_t)TW; j += NT / 4)
                if (!s_valid[j]) {
                    const uint32_t blk = ((j >> 2) + (j & 3u) * QPR) * ncell * 4u;
                    for (uint32_t i = tid & 3u; i < ncell * 4u; i += 4) s_pix[blk + i] = 0;
                }
            __syncthreads();
        }

        // ---- pixels: lane = (row within pass, quad of 4 consecutive columns), kernels_common.h
        uint32_t vq = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) vq |= s_valid[q * 4u + c] ? (1u << c) : 0u;
        decode_rows<S, QPR, XYZM, true, S::nt_stores, S::nt_xyz, false, NT, true, true>(
            a, s_pix, px_dw, cc, s_off, nullptr, (XYZM == 1 || XYZM == 2) ? s_beam : nullptr,
            a.gate_counts ? s_gate : nullptr, lut, f, c0, r0, TR, vq, rc, nch, nullptr);
    }
}

// ------------------------------------------------------------------------------------
// k_decode_stream2: the same pipeline with a DEDICATED LOADER WAVE (9 waves: 8 decode, 1 fetches).
// In k_decode_stream every wave carries its share of the next tile's DMA in its own vmcnt queue; the queue retires in
// issue order, so until those (slow: they queue behind the whole chip's stores) reads have landed, the 6-bit counter has
// only 63 - 12 slots left for the wave's stores, and in a slow buffer placement -- where stores want MORE of them in
// flight -- the kernel lost 4 - 10 % against k_decode_wide (tools/ab/lottery_variants.py).  Here the decoding waves issue
// nothing but stores.  One workgroup barrier per tile: the loader reaches it when tile i+1 has landed, the decoders
// when they are done with tile i-1's context; every lane classifies its own four columns from the fetched header
// words (no shared validity table, hence no second barrier), invalid columns are zeroed per pixel.
// ------------------------------------------------------------------------------------
template <class S, int TW, int XYZM>
__global__ __launch_bounds__(768) void k_decode_stream2(DecodeArgs a, StreamArgs sp) {
    constexpr int NT = 512, QPR = TW / 4;          // decoding threads; thread 512.. = the loader wave
    constexpr uint32_t chan = S::chan;
    static_assert(S::is_static && (TW == 128 || TW == 256), "one wave holds all the quads of a row");
    extern __shared__ __align__(16) uint32_t smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t W = a.g.columns_per_frame;
    const uint32_t cpp = a.g.columns_per_packet, col_size = a.g.col_size;
    const uint32_t TR = sp.tr, nch = sp.nch, npo = a.n_packets_out, ncell = sp.ncell;
    const uint32_t CT = a.tiles_per_frame;
    const uint32_t xcd = blockIdx.x & 7u, kx = blockIdx.x >> 3;
    const uint32_t ct = kx % CT, grp = kx / CT, NG = sp.groups;
    const uint32_t c0 = ct * TW;
    if (xcd >= a.n_frames) return;
    const uint32_t items = ((a.n_frames - xcd + 7u) >> 3) * nch;
    if (grp >= items) return;
    const uint32_t n_mine = (items - grp + NG - 1) / NG;
    auto item = [&](uint32_t i) { return grp + NG * i; };

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t* s_colofs = (uint32_t*)((uint8_t*)smem + sp.fixed_off);   // [TW] byte offset of column c0+j in a frame buffer
    for (uint32_t j = tid; j < (uint32_t)TW; j += blockDim.x) {
        const uint32_t c = c0 + j, p = c / cpp, ic = c - p * cpp;
        s_colofs[j] = p * (uint32_t)a.packet_stride + a.g.packet_header_size + ic * col_size;
    }
    if (ct == 0 && a.frame_meta) {   // frame-level values of my frames, up front (ordinary loads)
        for (uint32_t k = tid; k < n_mine; k += blockDim.x) {
            const uint32_t sk = item(k), m = sk / nch;
            if (sk != m * nch) continue;
            const uint32_t f = xcd + 8u * m;
            uint32_t count = a.slots_per_frame;
            if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
            a.frame_meta[f] = frame_meta_first_present(a.g, a.packets + (size_t)f * a.slots_per_frame * a.packet_stride,
                                                       a.packet_stride, count);
        }
    }
    const uint64_t tag = a.frame_state[FS_SEQ] + 1;
    __syncthreads();

    if (wave >= (uint32_t)(NT / 64)) {
        // ================= the loader waves: instruction k of a tile belongs to loader k % nl =================
        const uint32_t nl = (blockDim.x - NT) >> 6, li = wave - NT / 64;
        const uint32_t db = (64u * nl) / ncell, dc = 64u * nl - db * ncell;
        uint32_t n = 0;
        auto mine = [&]() { return ((n++) % nl) == li; };
        auto fetch = [&](uint32_t s, uint32_t b) {
            const uint32_t m = s / nch, rc = s - m * nch, f = xcd + 8u * m;
            const uint8_t* fb = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
            const uint8_t* src0 = fb + rc * TR * chan;
            const uint32_t ldsb = lds0 + b * sp.ctx_bytes;
            uint32_t block = (li * 64u + lane) / ncell, ch = li * 64u + lane - block * ncell;
            n = 0;
            for (uint32_t id = li; id < sp.npix_instr; id += nl) {
                uint32_t ofs = 0;
                if (block < (uint32_t)TW) {
                    const uint32_t j = (block % QPR) * 4u + block / QPR;
                    ofs = ((s_colofs[j] + a.g.col_header_size) & ~15u) + ch * 16u;
                }
                glds16(src0 + ofs, ldsb + id * 1024u);
                block += db; ch += dc;
                if (ch >= ncell) { ch -= ncell; ++block; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((uint32_t)k >= sp.n_hdr) break;
#pragma unroll
                for (int g = 0; g < TW / 64; ++g)
                    if (mine()) glds4(fb + s_colofs[g * 64 + lane] + sp.hdr_dw[k] * 4u, ldsb + sp.hdr_off + ((uint32_t)k * TW + g * 64u) * 4u);
            }
            const uint32_t np = (uint32_t)TW / cpp;
            const uint32_t pa = c0 / cpp + (lane < np ? lane : np - 1u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((uint32_t)k >= sp.n_pkt) break;
                if (mine()) glds4(fb + (size_t)pa * a.packet_stride + sp.pkt_dw[k] * 4u, ldsb + sp.pkt_off + (uint32_t)k * 256u);
            }
            if (a.host_timestamps && a.packet_timestamp) {
                const uint8_t* ht = (const uint8_t*)(a.host_timestamps + (size_t)f * a.slots_per_frame + pa);
                if (mine()) glds4(ht, ldsb + sp.pkt_off + 4u * 256u);
                if (mine()) glds4(ht + 4, ldsb + sp.pkt_off + 5u * 256u);
            }
            if (a.any_destagger)
                for (uint32_t k = 0; k * 64u < TR; ++k) {
                    const uint32_t i = k * 64u + lane;
                    if (mine()) glds4(a.dst_offsets + rc * TR + (i < TR ? i : TR - 1u), ldsb + sp.off_off + k * 256u);
                }
            if (XYZM == 1 || XYZM == 2) {
                const uint32_t* bt = (const uint32_t*)(as_const(a.luts)[f % a.n_luts].beam_tab + (size_t)rc * TR * 9u);
                const uint32_t nb = TR * 18u;
                for (uint32_t k = 0; k * 64u < nb; ++k) {
                    const uint32_t i = k * 64u + lane;
                    if (mine()) glds4(bt + (i < nb ? i : nb - 1u), ldsb + sp.beam_off + k * 256u);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // landed (only this wave's queue: it holds nothing else)
        };
        fetch(item(0), 0);
        for (uint32_t it = 0; it < n_mine; ++it) {
            __syncthreads();   // tile `it` is in its context; the other context is free (the decoders are done with tile it-1)
            if (it + 1 < n_mine) fetch(item(it + 1), (it + 1) & 1u);
        }
        return;
    }

    // ================= the decoding waves =================
    const uint32_t q = tid % QPR;
    uint32_t px_dw[4], pidx[4];
    uint32_t first_of_packet = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t col = c0 + q * 4u + c, p = col / cpp;
        const uint32_t base = s_colofs[q * 4u + c] + a.g.col_header_size;
        px_dw[c] = (q + (uint32_t)c * QPR) * ncell * 4u + ((base & 15u) >> 2);
        pidx[c] = p;
        first_of_packet |= (col == p * cpp) ? (1u << c) : 0u;
    }
    ColConst cc;
    uint32_t cur_lut = 0xffffffffu;
    const bool hdr_lane = tid < (uint32_t)QPR;   // the lanes that write the column headers / packet-level outputs (first row of wave 0)

    for (uint32_t it = 0; it < n_mine; ++it) {
        const uint32_t b = it & 1u, s = item(it);
        const uint32_t m = s / nch, rc = s - m * nch, f = xcd + 8u * m;
        const uint32_t r0 = rc * TR;
        uint32_t count = a.slots_per_frame;
        if (a.packet_counts) count = min(as_const(a.packet_counts)[f], a.slots_per_frame);
        LutDev lut{};
        if (XYZM != 0) {
            const auto* lp = as_const(a.luts) + f % a.n_luts;
            lut.beam_tab = lp->beam_tab; lut.col_tab = lp->col_tab; lut.n = lp->n;
        }
        if (XYZM == 1 || XYZM == 2) {
            const uint32_t li = f % a.n_luts;
            if (li != cur_lut) {
                cur_lut = li;
                load_colconst(cc, lut, c0 + q * 4u, W);
            }
        }
        __syncthreads();

        const uint32_t* s_pix = (const uint32_t*)((const uint8_t*)smem + b * sp.ctx_bytes);
        const uint32_t* s_hdr = s_pix + (sp.hdr_off >> 2);
        const uint32_t* s_pkt = s_pix + (sp.pkt_off >> 2);
        const int32_t* s_off = (const int32_t*)(s_pix + (sp.off_off >> 2));
        const double* s_beam = (const double*)(s_pix + (sp.beam_off >> 2));

        // ---- my four columns, from their fetched header words (slot c holds column c: DESIGN.md 3.1)
        uint32_t vq = 0, sq = 0, st4[4], hw4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t j = q * 4u + c, col = c0 + j;
            const bool present = pidx[c] < count;
            const uint32_t m_id = (uint16_t)plan_field(s_hdr, TW, j, sp.mid, a.g.col_measurement_id);
            const uint32_t st = (uint32_t)plan_field(s_hdr, TW, j, sp.st, a.g.col_status);
            hw4[c] = present ? (m_id | ((st & 1u) << 16)) : 0u;
            const bool live = present && (st & 1u) && m_id < W;
            bool stray = live && m_id != col;
            const bool v = live && !stray;
            st4[c] = v ? st : 0u;
            if (rc == 0 && hdr_lane && ((first_of_packet >> c) & 1u)) {   // batch_lidar_packet, lidar_frame.cpp:1534-1539
                const uint32_t p = pidx[c], pl = p - c0 / cpp;
                const bool want_pk = a.packet_timestamp || a.alert_flags;
                const bool home = present && m_id / cpp == p;
                if (present && !home && want_pk && m_id / cpp < npo) stray = true;
                if (a.packet_timestamp && a.host_timestamps)
                    a.packet_timestamp[(size_t)f * npo + p] = home ? ((uint64_t)s_pkt[5 * 64 + pl] << 32) | s_pkt[4 * 64 + pl] : 0ull;
                if (a.alert_flags && home)
                    a.alert_flags[(size_t)f * npo + p] = (uint8_t)plan_field(s_pkt, 64, pl, sp.alert, a.g.alert_flags);
            }
            vq |= v ? (1u << c) : 0u;
            sq |= stray ? (1u << c) : 0u;
        }
        if (rc == 0 && hdr_lane) {
            const uint32_t col = c0 + q * 4u;
            if (a.timestamp) {
                uint64_t ts[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    ts[c] = ((vq >> c) & 1u) ? plan_field(s_hdr, TW, q * 4u + c, sp.ts, a.g.col_timestamp) : 0ull;
                uint8_t* d = (uint8_t*)(a.timestamp + (size_t)f * W + col);
                st16(d, (uint32_t)ts[0], (uint32_t)(ts[0] >> 32), (uint32_t)ts[1], (uint32_t)(ts[1] >> 32));
                st16(d + 16, (uint32_t)ts[2], (uint32_t)(ts[2] >> 32), (uint32_t)ts[3], (uint32_t)(ts[3] >> 32));
            }
            if (a.measurement_id) {
                uint32_t mm[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) mm[c] = ((vq >> c) & 1u) ? ((col + c) & 0xffffu) : 0u;
                st8(a.measurement_id + (size_t)f * W + col, mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16));
            }
            if (a.status) st16(a.status + (size_t)f * W + col, st4[0], st4[1], st4[2], st4[3]);
            if (a.hdr_words) st16(a.hdr_words + (size_t)f * W + col, hw4[0], hw4[1], hw4[2], hw4[3]);
            // the tile's valid-column count and stray flag: all of its columns sit in these QPR lanes of wave 0
            const uint64_t lm = QPR >= 64 ? ~0ull : ((1ull << QPR) - 1);
            uint32_t nv = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) nv += (uint32_t)__popcll(__ballot((vq >> c) & 1u) & lm);
            const bool any_stray = (__ballot(sq != 0) & lm) != 0;
            if (tid == 0) {
                if (any_stray) flag_frame(a, f, tag);
                if (a.frame_meta) a.tile_valid[(size_t)f * CT + ct] = (uint16_t)nv;
                if (f == 0 && ct == 0) a.frame_state[FS_TAG] = tag;
            }
        }

        uint32_t pxd[4] = {px_dw[0], px_dw[1], px_dw[2], px_dw[3]};
        decode_rows<S, QPR, XYZM, false, S::nt_stores, S::nt_xyz, false, NT, true, true>(
            a, s_pix, pxd, cc, s_off, nullptr, (XYZM == 1 || XYZM == 2) ? s_beam : nullptr, nullptr, lut, f, c0, r0, TR, vq,
            rc, nch, nullptr);
    }
}

// ------------------------------------------------------------------------------------
// launcher (host)
// ------------------------------------------------------------------------------------
struct LdsGrantS {
    std::atomic<uint32_t> bytes[16];
    LdsGrantS() { for (auto& b : bytes) b.store(0); }
};

template <class S, int TW, int XYZM>
static hipError_t launch_stream_x(const DecodeArgs& a, const StreamArgs& sp, dim3 grid, int device, hipStream_t st) {
    static LdsGrantS done;
    std::atomic<uint32_t>& have = done.bytes[device & 15];
    if (sp.lds_bytes > 48 * 1024 && have.load(std::memory_order_acquire) < sp.lds_bytes) {
        hipError_t e = hipFuncSetAttribute((const void*)k_decode_stream<S, TW, XYZM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_bytes);
        if (e != hipSuccess) return e;
        have.store(sp.lds_bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_decode_stream<S, TW, XYZM>), grid, dim3(512), sp.lds_bytes, st, a, sp);
    return hipGetLastError();
}

template <class S, int TW, int XYZM>
static hipError_t launch_stream2_x(const DecodeArgs& a, const StreamArgs& sp, dim3 grid, int device, hipStream_t st) {
    static LdsGrantS done;
    std::atomic<uint32_t>& have = done.bytes[device & 15];
    if (sp.lds_bytes > 48 * 1024 && have.load(std::memory_order_acquire) < sp.lds_bytes) {
        hipError_t e = hipFuncSetAttribute((const void*)k_decode_stream2<S, TW, XYZM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_bytes);
        if (e != hipSuccess) return e;
        have.store(sp.lds_bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_decode_stream2<S, TW, XYZM>), grid, dim3(512u + 64u * sp.loader), sp.lds_bytes, st, a, sp);
    return hipGetLastError();
}

template <class S, int TW>
static hipError_t launch_stream_t(const DecodeArgs& a, const StreamArgs& sp, int xyzm, dim3 grid, int device, hipStream_t st) {
    if constexpr (TW == 128 || TW == 256) {
        if (sp.loader) {
            switch (xyzm) {
                case 0: return launch_stream2_x<S, TW, 0>(a, sp, grid, device, st);
                case 1: return launch_stream2_x<S, TW, 1>(a, sp, grid, device, st);
                case 2: return launch_stream2_x<S, TW, 2>(a, sp, grid, device, st);
                default: return hipErrorInvalidValue;
            }
        }
    }
    switch (xyzm) {
        case 0: return launch_stream_x<S, TW, 0>(a, sp, grid, device, st);
        case 1: return launch_stream_x<S, TW, 1>(a, sp, grid, device, st);
        case 2: return launch_stream_x<S, TW, 2>(a, sp, grid, device, st);
        default: return hipErrorInvalidValue;   // full LUTs are read in the row loop: not a streaming candidate
    }
}

hipError_t OUSTER_SPEC_FN(launch_decode_stream)(const DecodeArgs& a, const StreamArgs& sp, int tw, int xyzm, int device,
                                                hipStream_t st) {
    if (sp.lds_bytes > 160 * 1024 || sp.npix_instr > 72) return hipErrorInvalidValue;
    const dim3 grid(8u * a.tiles_per_frame * sp.groups);
    switch (tw) {
        case 128: return launch_stream_t<SpecT, 128>(a, sp, xyzm, grid, device, st);
        case 256: return launch_stream_t<SpecT, 256>(a, sp, xyzm, grid, device, st);
        case 512: return launch_stream_t<SpecT, 512>(a, sp, xyzm, grid, device, st);
        case 1024: return launch_stream_t<SpecT, 1024>(a, sp, xyzm, grid, device, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ouster_hip_dev
