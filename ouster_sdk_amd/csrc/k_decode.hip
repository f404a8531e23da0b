// k_decode.hip -- the fused decode (+ destagger + cartesian) kernels of the hot path, gfx950.
// Compiled once per packet-profile specialisation (-DOUSTER_SPEC_ID=0..5, see the Makefile) so the
// six specialisations build in parallel; kernels_common.h holds the pixel phase they all share.
//
//   k_decode        one workgroup per (frame, tile of 64/32/16 destination columns): whole columns
//                   staged in LDS as wire bytes, then walked row-wise (decode_rows)
//   k_decode_wide   the same on wide, short tiles (64..512 columns x a chunk of rows)
//
// What is computed is defined by the reference loops
//   PacketFormat::col_field/block_field      ouster_core/src/parsing.cpp:628-675
//   FrameBatcher::parse_by_col/_by_block     ouster_core/src/lidar_frame.cpp:1422-1528
//   batch_lidar_packet / start_frame         ouster_core/src/lidar_frame.cpp:1530-1576, 1709-1741
//   destagger_into<T>                        ouster_core/include/ouster/core/impl/lidar_frame_impl.h:733-760
//   impl::make_xyz_lut / cartesianT<T>       ouster_core/src/xyzlut.cpp:11-89, impl/cartesian.h:36-66
//
// Column -> source slot ("which received column lands in destination column c?"), DESIGN.md 3.1:
//   MODE_FAST     the optimistic pass.  When a frame's buffer has exactly W column slots, a sensor's
//                 packets in order put column c in slot c.  The workgroup stages slots [c0, c0+TILE)
//                 WITHOUT looking anything up (no map kernel, no dependent global round trip), then
//                 checks the staged column headers: a slot is *dead* (no packet / status&1 == 0 /
//                 m_id >= W  -> zeros, exactly the reference's skip rules lidar_frame.cpp:1432-1450),
//                 *home* (m_id == its slot) or a *stray* (a live column somewhere else: dropped packet
//                 with compacted slots, shuffled order, duplicate).  Every slot is checked by the
//                 workgroup that owns it, so "no workgroup saw a stray" proves the identity mapping
//                 for the whole frame; one stray flags the frame in frame_state.
//   MODE_FIXUP    second launch behind the fast one (k_decode_fixup, a small persistent grid): reads the
//                 flags, redoes the flagged frames with the general mapping below, retires the flags.
//   MODE_GENERAL  destination column <- the LAST slot in buffer order whose live column carries that
//                 measurement_id ("the packet batched later overwrites"): every
//                 workgroup scans the frame's column headers (L2 resident after the first tile) and
//                 keeps the winners of its own tile in LDS.  Used for every frame when the buffer does
//                 not have exactly W slots.
// The only thing that persists between calls is a sequence word in HBM that the fix-up pass advances
// (flags are tagged with it instead of being cleared), so a captured graph can be replayed on changed
// packet contents and eager calls may run in between.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "kernels_common.h"

#ifndef OUSTER_SPEC_ID
#error "compile with -DOUSTER_SPEC_ID=0..5"
#endif

namespace ouster_hip_dev {

#if OUSTER_SPEC_ID == 0
using SpecT = SpecGeneric;
#define OUSTER_SPEC_FN(name) name##_generic
#elif OUSTER_SPEC_ID == 1
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

// blockIdx -> (frame, sub-block).  XCD-aware: block b is dispatched to XCD b % 8, so frame f is given
// to XCD f % 8 and its blocks are consecutive there: neighbouring tiles' partial cache lines
// (unaligned destaggered rows, 64 B u8 segments) merge in that XCD's L2 before they are written back.
__device__ __forceinline__ bool block_to_frame(const DecodeArgs& a, uint32_t blocks_per_frame, uint32_t& f,
                                               uint32_t& sub) {
    if (a.xcd_map) {
        const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        f = (i / blocks_per_frame) * 8u + xcd;
        sub = i % blocks_per_frame;
        return f < a.n_frames;
    }
    f = blockIdx.x / blocks_per_frame;
    sub = blockIdx.x - f * blocks_per_frame;
    return true;
}

// What the workgroup that owns column tile `tile` of frame f reports from the optimistic pass (one lane):
// a stray raises the frame's word to this call's tag; the tile's valid-column count goes to its own
// slot (plain store); the first workgroup of the launch records the tag for the fix-up pass.
__device__ __forceinline__ void fast_publish(const DecodeArgs& a, uint32_t f, uint32_t tile, uint32_t n_valid,
                                             bool stray) {
    const uint64_t tag = a.frame_state[FS_SEQ] + 1;
    if (stray) atomicMax((unsigned long long*)&a.frame_state[FS_WORDS + f], (unsigned long long)tag);
    if (a.frame_meta) a.tile_valid[(size_t)f * a.tiles_per_frame + tile] = (uint16_t)n_valid;
    if (f == 0 && tile == 0) a.frame_state[FS_TAG] = tag;
}

// ------------------------------------------------------------------------------------
// k_decode: fused decode + destagger + cartesian for one (frame, column tile)
// ------------------------------------------------------------------------------------
// XYZM: 0 no xyz, 1 separable tables -> f32, 2 separable -> f64, 3 full LUT (runtime dtypes)
// GENERAL_ONLY: the fix-up kernel's instantiation (no fast-mode code in it)
template <class S, int TILE, int XYZM, bool GENERAL_ONLY, bool POSES>
__device__ __forceinline__ void decode_tile(const DecodeArgs& a, uint32_t* smem, uint32_t f, uint32_t tile) {
    constexpr int NT = 256;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, H = a.g.pixels_per_column;
    const uint32_t cpp = a.g.columns_per_packet, col_size = a.g.col_size;
    const uint32_t npo = a.n_packets_out;
    const uint32_t c0 = tile * TILE;

    // ---- LDS carve-up (all offsets 16 B aligned)
    uint32_t* s_tile = smem;                                    // TILE * col_size (+16) bytes
    const uint32_t tile_bytes = (TILE * col_size + 16 + 15) & ~15u;
    int32_t* s_src = (int32_t*)(smem + (tile_bytes >> 2));      // [TILE] slot that supplies the column's pixels (general modes)
    int32_t* s_hsrc = s_src + TILE;                             // [TILE] slot that supplies its header (general modes)
    uint64_t* s_masks = (uint64_t*)(s_hsrc + TILE);             // [0] valid, [1] group-ok / stray
    int32_t* s_off = (int32_t*)(s_masks + 4);                   // [H] destagger offsets
    double* s_beam = (double*)(s_off + ((H + 3) & ~3u));        // [H][9] per-beam xyz constants (a.beam_lds)
    float4* s_xyz = (float4*)(s_beam + (a.beam_lds ? H * 9 + (H & 1) : 0));  // [4 waves][192] (OUSTER_XYZ_PERMUTE=0 builds)
    // general modes: resolve_frame's scratch starts at smem[0] -- it is done before anything above is written

    const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    uint32_t count = a.slots_per_frame;
    if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
    const bool fast = !GENERAL_ONLY && a.mode == MODE_FAST;

    // ---- phase 0
    uint64_t pk_ts = 0;
    uint32_t pk_alert = 0, my_p = 0;
    bool pk_lane = false;
    if (fast) {
        // packet-level values travel in the lane of the packet's first column; frame meta from slot 0
        if (tid < TILE) {
            const uint32_t c = c0 + tid;
            my_p = c / cpp;
            pk_lane = c < W && c == my_p * cpp;
            if (pk_lane && my_p < count) {
                if (a.packet_timestamp && a.host_timestamps)
                    pk_ts = a.host_timestamps[(size_t)f * a.slots_per_frame + my_p];
                if (a.alert_flags)
                    pk_alert = (uint32_t)apply_bits(
                        window_global(fbase + (size_t)my_p * a.packet_stride + a.g.alert_flags.offset),
                        a.g.alert_flags.mask, a.g.alert_flags.shift);
            }
        }
        if (tile == 0 && tid == 0 && a.frame_meta) a.frame_meta[f] = frame_meta_first_present(a.g, fbase, a.packet_stride, count);
    } else {
        // general mapping: what FrameBatcher leaves behind after batching the frame's packets in buffer order
        // (resolve_frame, kernels_common.h); my tile keeps its columns, tile 0 also writes the packet-level outputs and counts
        const ResolveLds L(smem, W, npo, a.slots_per_frame);
        int32_t *r_pix = L.pix, *r_hdr = L.hdr, *r_pkm = L.pkm;
        __shared__ uint32_t s_nvalid;
        if (tid == 0) s_nvalid = 0;
        resolve_frame<NT>(a.g, fbase, a.packet_stride, count, npo, L, tile == 0,
                          (GENERAL_ONLY && a.hdr_words) ? a.hdr_words + (size_t)f * W : nullptr);
        int32_t my_pix = -1, my_hdr = -1;
        if (tid < TILE && c0 + tid < W) { my_pix = r_pix[c0 + tid]; my_hdr = r_hdr[c0 + tid]; }
        if (tile == 0) {
            // packet_timestamp is zeroed at frame start (lidar_frame.cpp:1719), alert_flags is not
            for (uint32_t i = tid; i < npo; i += NT) {
                const int32_t p = r_pkm[i];
                if (a.packet_timestamp && a.host_timestamps)
                    a.packet_timestamp[(size_t)f * npo + i] =
                        p >= 0 ? a.host_timestamps[(size_t)f * a.slots_per_frame + p] : 0ull;
                if (a.alert_flags && p >= 0)
                    a.alert_flags[(size_t)f * npo + i] = (uint8_t)apply_bits(
                        window_global(fbase + (size_t)p * a.packet_stride + a.g.alert_flags.offset),
                        a.g.alert_flags.mask, a.g.alert_flags.shift);
            }
            if (a.frame_meta) {
                uint32_t n = 0;
                for (uint32_t i = tid; i < W; i += NT) n += r_hdr[i] >= 0 ? 1u : 0u;
                n = wave_sum(n);
                if (n && (tid & 63u) == 0) atomicAdd(&s_nvalid, n);
                __syncthreads();
                if (tid == 0) {
                    ouster_hip_frame_meta m = frame_meta_general(a, fbase, count);
                    m.n_valid_columns = s_nvalid;
                    a.frame_meta[f] = m;
                }
            }
        }
        if constexpr (GENERAL_ONLY) {
            // The fix-up pass: the optimistic pass has already written this tile under "slot c holds column c, anything else
            // reads as zeros".  Where the frame's real mapping says the same for every column of the tile there is nothing to redo.
            bool dirty = false;
            if (tid < TILE && c0 + tid < W) {
                const uint32_t c = c0 + tid, p = c / cpp;
                int32_t expect = -1;
                if (p < count && L.hd[c] == (c | 0x10000u)) expect = (int32_t)c;   // slot c is live and at home
                dirty = my_pix != expect || my_hdr != expect;
            }
            if (!__syncthreads_or(dirty ? 1 : 0)) return;
        } else {
            __syncthreads();   // the scratch is about to be overwritten
        }
        if (tid < TILE) {
            s_src[tid] = my_pix;
            s_hsrc[tid] = my_hdr;
            const int32_t src = my_pix;
            const uint32_t j0 = tid - tid % cpp;  // first column of my packet group in the tile
            // "group ok": my packet's cpp columns sit in order, packet-aligned, all present
            const int32_t head = __shfl(src, (int)(j0 & 63u));
            const bool grp = (TILE % cpp == 0) && src >= 0 && head >= 0 && (uint32_t)head % cpp == 0 &&
                             src == head + (int32_t)(tid - j0);
            const uint64_t vb = __ballot(src >= 0);
            const uint64_t gb = __ballot(grp);
            if (tid == 0) { s_masks[0] = vb; s_masks[1] = gb; }
        }
    }
    if (a.any_destagger)
        for (uint32_t r = tid; r < H; r += NT) s_off[r] = a.dst_offsets[r];
    const LutDev lut = (XYZM != 0) ? a.luts[f % a.n_luts] : LutDev{};
    if ((XYZM == 1 || XYZM == 2) && a.beam_lds)
        for (uint32_t i = tid; i < H * 9; i += NT) s_beam[i] = lut.beam_tab[i];
    if (!fast) __syncthreads();

    // ---- phase 1: stage the tile's columns in LDS, column j at byte j*col_size
    const uint32_t ncols_here = min((uint32_t)TILE, W - c0);
    const uint64_t existmask = ncols_here >= 64 ? ~0ull : ((1ull << ncols_here) - 1);
    uint64_t validmask = fast ? existmask : s_masks[0];          // fast: refined after staging
    const uint64_t groupmask = fast ? existmask : s_masks[1];    // fast: W % cpp == 0 (W slots per frame)
    auto src_of = [&](uint32_t j) -> int32_t {
        return fast ? (c0 + j < W ? (int32_t)(c0 + j) : -1) : s_src[j];
    };
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
            const uint32_t p = (g < G) ? (uint32_t)src_of(g * cpp) / cpp : 0u;
            src[g] = (const u32x4*)(fbase + (size_t)p * a.packet_stride + a.g.packet_header_size);
        }
        u32x4* dst = (u32x4*)s_tile;
        constexpr int DEPTH = 17;  // 17 x 256 x 16 B = 68 KB: a 64-column dual-LB tile in one pass
        for (uint32_t base = 0; base < total; base += NT * DEPTH) {
            u32x4 t[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                // every lane loads (the ones past the end re-read the last piece): a guarded load and its guarded LDS
                // write further down were merged into one block per piece in the fix-up instantiation -- a full memory
                // latency per 16 B piece, fifteen in a row per tile
                const uint32_t idx = min(base + k * NT + tid, total - 1u);
                const uint32_t g = (idx >= n16) + (idx >= 2 * n16) + (idx >= 3 * n16);
                const u32x4* sp = g == 0 ? src[0] : g == 1 ? src[1] : g == 2 ? src[2] : src[3];
                t[k] = __builtin_nontemporal_load(sp + (idx - g * n16));
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
                const uint32_t p = (uint32_t)src_of(j0) / cpp;
                stage_range<NT>(s_tile, j0 * col_size,
                                fbase + (size_t)p * a.packet_stride + a.g.packet_header_size,
                                gbytes, tid);
            } else if (validmask & gm) {
                for (uint32_t j = j0; j < j0 + cpp; ++j) {
                    const int32_t s = src_of(j);
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
            const int32_t s = src_of(j);
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
    int32_t my_ps = 0, my_hs = 0;   // general modes: my column's pixel / header source slot (s_src is about to be reused)
    if (!fast && tid < TILE) { my_ps = s_src[tid]; my_hs = s_hsrc[tid]; }
    uint32_t* s_gate = a.gate_counts ? (uint32_t*)s_src : nullptr;  // the slot map is dead after staging
    if (s_gate) {
        if (tid < TILE) s_gate[tid] = 0;
        __syncthreads();
    }

    if (fast) {
        // ---- the check that makes the optimism safe: classify my slots from the staged headers
        if (tid < TILE) {
            const uint32_t c = c0 + tid;
            const bool present = c < W && my_p < count;
            uint32_t m_id = 0, st = 0;
            if (present) col_header_lds(a.g, s_tile, tid * col_size, m_id, st);
            const bool live = present && (st & 1u) && m_id < W;
            bool stray = live && m_id != c;
            if (a.hdr_words && c < W) a.hdr_words[(size_t)f * W + c] = present ? (m_id | ((st & 1u) << 16)) : 0u;
            if (pk_lane) {
                // batch_lidar_packet (lidar_frame.cpp:1534-1539): packet-level values go to index
                // m_id(first column) / cpp whether or not that column is valid
                const bool want_pk = a.packet_timestamp || a.alert_flags;
                const bool home = present && m_id / cpp == my_p;
                if (present && !home && want_pk && m_id / cpp < npo) stray = true;
                if (a.packet_timestamp && a.host_timestamps)
                    a.packet_timestamp[(size_t)f * npo + my_p] = home ? pk_ts : 0ull;
                if (a.alert_flags && home) a.alert_flags[(size_t)f * npo + my_p] = (uint8_t)pk_alert;
            }
            const uint64_t vb = __ballot(live && !stray);
            const uint64_t sb = __ballot(stray);
            if (tid == 0) {
                s_masks[0] = vb;
                s_masks[1] = sb;
                fast_publish(a, f, tile, (uint32_t)__popcll(vb), sb != 0);
            }
        }
        __syncthreads();
        // a stray column reads as zeros here, like in the wide kernels: the fix-up pass redoes exactly the tiles whose real
        // mapping differs from "slot c holds column c or nothing"
        validmask = s_masks[0];
    }

    // ---- phase 2a: column headers (timestamp / measurement_id / status), one lane per column
    if (tid < TILE && c0 + tid < W) {
        const uint32_t c = c0 + tid;
        bool v = (validmask >> tid) & 1;
        uint64_t ts = 0;
        uint32_t st = 0;
        const int32_t hs = my_hs;
        if (!fast && hs != my_ps) {
            // the header comes from another slot than the pixels (an all-valid packet with non-consecutive ids, parse_by_block)
            v = hs >= 0;
            if (v) {
                const uint32_t p = (uint32_t)hs / cpp;
                const uint8_t* colp = fbase + (size_t)p * a.packet_stride + a.g.packet_header_size + (size_t)((uint32_t)hs - p * cpp) * col_size;
                ts = apply_bits(window_global(colp + a.g.col_timestamp.offset), a.g.col_timestamp.mask, a.g.col_timestamp.shift);
                st = (uint32_t)apply_bits(window_global_masked(colp + a.g.col_status.offset, a.g.col_status.mask), a.g.col_status.mask,
                                          a.g.col_status.shift);
            }
        } else if (v) {
            const uint32_t cb = tid * col_size;
            ts = apply_bits(window_lds(s_tile, cb + a.g.col_timestamp.offset), a.g.col_timestamp.mask, a.g.col_timestamp.shift);
            st = (uint32_t)apply_bits(window_lds(s_tile, cb + a.g.col_status.offset), a.g.col_status.mask, a.g.col_status.shift);
        }
        if (a.timestamp) a.timestamp[(size_t)f * W + c] = ts;
        if (a.measurement_id) a.measurement_id[(size_t)f * W + c] = v ? (uint16_t)c : (uint16_t)0;
        if (a.status) a.status[(size_t)f * W + c] = st;
    }

    // ---- phase 2b: pixels
    const uint32_t q = tid % (TILE / 4);
    const uint32_t vq = (uint32_t)(validmask >> (q * 4)) & 0xfu;
    const void* s_pose = stage_poses<XYZM, POSES>(a, smem, f, c0, (uint32_t)TILE);
    uint32_t px_dw[4];
    tile_px_offsets<TILE / 4>(px_dw, a.g.col_header_size >> 2, col_size >> 2);
    ColConst cc;
    if (XYZM == 1 || XYZM == 2) load_colconst(cc, lut, c0 + q * 4, W);
    decode_rows<S, TILE / 4, XYZM, false, false, false, POSES>(a, s_tile, px_dw, cc, s_off, s_xyz,
                                   ((XYZM == 1 || XYZM == 2) && a.beam_lds) ? s_beam : nullptr, s_gate, lut, f, c0, 0u, H,
                                   vq, 0u, 1u, s_pose);
}

// one workgroup per (frame, tile): the optimistic pass (MODE_FAST) or every frame through the
// general mapping (MODE_GENERAL)
template <class S, int TILE, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode(DecodeArgs a) {
    extern __shared__ __align__(16) uint32_t smem[];
    uint32_t f, tile;
    if (!block_to_frame(a, a.tiles_per_frame, f, tile)) return;
    decode_tile<S, TILE, XYZM, false, POSES>(a, smem, f, tile);
}

// The fix-up pass behind an optimistic pass.  A small persistent grid (all workgroups resident at
// once): every workgroup reads the frame words, lists the frames flagged with this call's tag in LDS
// and takes its share of their tiles through the general mapping; for a clean batch that is one
// coalesced read and out.  The clean frames' valid-column counts are summed from the tiles' slots.
// Workgroup 0 finally advances the sequence word, which retires every flag of this call: nothing is
// cleared, nothing is waited for, and a captured graph replays correctly (the tag lives in HBM).
constexpr uint32_t FIXUP_CHUNK = 512;  // frames listed per round (u16 indices in LDS)
template <class S, int TILE, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode_fixup(DecodeArgs a) {
    constexpr int NT = 256;
    extern __shared__ __align__(16) uint32_t smem[];
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_cnt[FIXUP_CHUNK / 64];   // flagged frames per 64-frame group of the chunk
    const uint32_t tid = threadIdx.x, tpf = a.tiles_per_frame;
    const uint32_t fast_tiles = a.fast_tiles;    // column tiles of the optimistic pass (slots of tile_valid)
    uint16_t* s_list = (uint16_t*)(smem + (a.rows_per_tile >> 2));  // rows_per_tile: byte offset of the list here
    const uint64_t tag = a.frame_state[FS_TAG];
    for (uint32_t base = 0; base < a.n_frames; base += FIXUP_CHUNK) {
        // The list must come out in the SAME order in every workgroup -- item `it` is (s_list[it / tpf], it % tpf) and the
        // workgroups share the items out by index -- so it is compacted in frame order (ballots + a prefix over the
        // 64-frame groups), not in the arrival order of an atomic counter (r02: with flagged frames in more than one wave's
        // share, workgroups disagreed about the order, some tiles were redone twice and others never).
        const uint32_t nfr = min(FIXUP_CHUNK, a.n_frames - base);
        constexpr uint32_t ROUNDS = FIXUP_CHUNK / NT;
        uint64_t mine[ROUNDS];
        bool flagged[ROUNDS];
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            const uint32_t i = r * NT + tid;
            flagged[r] = i < nfr && a.frame_state[FS_WORDS + base + i] == tag;
            mine[r] = __ballot(flagged[r]);
            if ((tid & 63u) == 0) s_cnt[i >> 6] = (uint32_t)__popcll(mine[r]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            const uint32_t i = r * NT + tid, grp = i >> 6;
            uint32_t before = 0;
            for (uint32_t k = 0; k < grp; ++k) before += s_cnt[k];
            if (flagged[r]) s_list[before + (uint32_t)__popcll(mine[r] & ((1ull << (tid & 63u)) - 1ull))] = (uint16_t)i;
        }
        if (tid == 0) {
            uint32_t n = 0;
            for (uint32_t k = 0; k < FIXUP_CHUNK / 64; ++k) n += s_cnt[k];
            s_n = n;
        }
        __syncthreads();
        // A flagged frame stays on ONE XCD (workgroup b runs on XCD b % 8): its 64-column tiles write 64 - 256 B row segments,
        // and neighbouring tiles' halves of a cache line must meet in the same L2 (the reason for xcd_map in the one-tile
        // kernels; spread over all eight XCDs the same tiles cost 5.5 us per frame instead of 3.6).  The k-th flagged frame
        // goes to XCD k % 8, whose workgroups share its tiles out by index.
        const uint32_t n_flagged = s_n;
        if (gridDim.x >= 16 && (gridDim.x & 7u) == 0) {
            const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
            const uint32_t mine = n_flagged > xcd ? (n_flagged - xcd + 7u) / 8u : 0u;   // flagged frames of my XCD
            for (uint32_t it = slot; it < mine * tpf; it += per_xcd) {
                const uint32_t f = base + s_list[(it / tpf) * 8u + xcd], tile = it % tpf;
                decode_tile<S, TILE, XYZM, true, POSES>(a, smem, f, tile);
                __syncthreads();  // the tile image is reused
            }
        } else {
            for (uint32_t it = blockIdx.x; it < n_flagged * tpf; it += gridDim.x) {
                const uint32_t f = base + s_list[it / tpf], tile = it % tpf;
                decode_tile<S, TILE, XYZM, true, POSES>(a, smem, f, tile);
                __syncthreads();  // the tile image is reused
            }
        }
    }
    // valid-column counts of the clean frames (flagged ones got theirs from the general path)
    if (a.frame_meta) {
        for (uint32_t f = blockIdx.x * NT + tid; f < a.n_frames; f += gridDim.x * NT) {
            if (a.frame_state[FS_WORDS + f] == tag) continue;
            uint32_t n = 0;
            for (uint32_t t = 0; t < fast_tiles; ++t) n += a.tile_valid[(size_t)f * fast_tiles + t];
            a.frame_meta[f].n_valid_columns = n;
        }
    }
    if (blockIdx.x == 0 && tid == 0) a.frame_state[FS_SEQ] = tag;  // the next call tags with tag + 1
}

// ------------------------------------------------------------------------------------
// k_decode_wide: the same fused decode + destagger + cartesian with WIDE, SHORT tiles:
// a workgroup owns TW columns x TR rows (TW*TR*chan ~ 48-64 KB of LDS) instead of 64 columns x all
// rows.  Every output row segment is then TW/64 times longer (1 KB of a u32 plane, 256 B of a u8
// plane, 3 KB of xyz for TW = 256), which is what HBM wants: with 64-column tiles the achieved write
// rate swings between 3.4 and 4.9 TB/s with the physical placement of the output planes
// (tools/storebench.hip), with 256-column tiles it stays at 5.1-6.1 TB/s.
// The price is on the (8x smaller) input side: a column is no longer read whole but in TR-row pieces
// (TR*chan bytes, 256 B for dual-LB at TW = 256), staged into per-column LDS slots padded by one dword
// (bank spread for the 4-columns-per-lane reads).  The column tiles of one row chunk are consecutive
// blocks of one XCD, so neighbouring workgroups write whole rows together.
// The optimistic pass (slot c holds column c), or -- a.slot_map set -- the general mapping from k_slotmap's per-frame
// map (the fix-up pass always runs k_decode).  Every row chunk reads the (measurement_id,
// status) words of its columns next to its staging loads, the first row chunk of a column tile does
// the stray check, the column headers and the packet-level outputs.
// ------------------------------------------------------------------------------------
// LMAPS: the column maps of the frame lie in LDS (l_pix / l_hdr, resolve_frame's output in the fix-up pass; they may lie
// under the tile image: they are read before the first barrier); otherwise a.slot_map / a.hdr_map in global memory, or none.
template <class S, int TW, int XYZM, bool POSES, bool LMAPS>
__device__ __forceinline__ void wide_tile(const DecodeArgs& a, uint32_t* smem, uint32_t f, uint32_t tile, uint32_t rc,
                                          const int32_t* l_pix, const int32_t* l_hdr, uint32_t TR, uint32_t nch) {
    constexpr int NT = 256;
    constexpr int NJ = (TW + NT - 1) / NT;        // columns per thread in the per-column phases
    static_assert(TW % 64 == 0 && TW / 4 <= NT * 4, "tile width");

    // TR rows per tile, nch row chunks per frame: a.rows_per_tile / a.row_chunks, or fewer rows (the fix-up pass with few
    // flagged frames; the LDS column stride a.lds_col_slot stays that of the launch's tallest tile)
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, H = a.g.pixels_per_column;
    const uint32_t cpp = a.g.columns_per_packet, col_size = a.g.col_size;
    const uint32_t chan = S::is_static ? S::chan : a.g.channel_data_size;
    const uint32_t hdr = a.g.col_header_size, npo = a.n_packets_out;
    const uint32_t c0 = tile * TW, r0 = rc * TR;
    const uint32_t nrows = min(TR, H - r0);
    const uint32_t slot = a.lds_col_slot >> 2;     // LDS dwords per column (piece + pad)

    uint32_t* s_tile = smem;                                  // [TW][slot]
    uint32_t* s_colofs = smem + a.wide_img_words;             // [TW] byte offset of the column in the frame buffer (TW * slot + 4 words in, or behind resolve_frame's scratch)
    uint32_t* s_valid = s_colofs + TW;                        // [TW] 1 = received, valid, at home
    uint32_t* s_acc = s_valid + TW;                           // [0] valid columns, [1] strays (+2 pad)
    int32_t* s_off = (int32_t*)(s_acc + 4);                   // [TR] destagger offsets of my rows
    uint32_t* s_gate = (uint32_t*)(s_off + ((TR + 3) & ~3u)); // [TW] range-gate counters
    double* s_beam = (double*)(s_gate + TW);                  // [TR][9] per-beam xyz constants of my rows
    float4* s_xyz = (float4*)(s_beam + TR * 9 + (TR & 1));    // [4 waves][192] (OUSTER_XYZ_PERMUTE=0 builds)

    const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    uint32_t count = a.slots_per_frame;
    if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
#ifdef OUSTER_PHASE_TIMING
    uint64_t* pt_ = (a.phase_times && a.mode != MODE_FIXUP) ? a.phase_times + (size_t)blockIdx.x * 16 : nullptr;
#define PHASE_STAMP(i) do { if (pt_ && tid == 0) pt_[i] = __builtin_readcyclecounter(); } while (0)
    if (pt_ && tid == 0) { uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); pt_[7] = xcc; }
#else
#define PHASE_STAMP(i) do {} while (0)
#endif
    PHASE_STAMP(0);

    // ---- phase 0: where my columns live (slot c holds column c): arithmetic only, nothing is loaded
    const bool mapped = LMAPS || a.slot_map != nullptr;
    uint32_t hofs[NJ];   // mapped: byte offset of the column whose HEADER lands in my column (0xffffffff: none)
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const uint32_t j = tid + (uint32_t)k * NT, c = c0 + j;
        hofs[k] = 0xffffffffu;
        if (j >= (uint32_t)TW) continue;
        uint32_t ofs = 0xffffffffu;
        if (c < W) {
            // source slot of destination column c: itself (the optimistic pass), or what resolve_frame found (general mapping)
            int32_t sl = (int32_t)c, hs = (int32_t)c;
            if constexpr (LMAPS) { sl = l_pix[c]; hs = l_hdr[c]; }
            else if (a.slot_map) {   // agent-scope loads: in the fix-up pass another XCD's workgroup wrote them moments ago
                sl = __hip_atomic_load(&a.slot_map[(size_t)f * W + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hs = __hip_atomic_load(&a.hdr_map[(size_t)f * W + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (sl >= 0) {
                const uint32_t p = (uint32_t)sl / cpp, ic = (uint32_t)sl - p * cpp;
                ofs = p * (uint32_t)a.packet_stride + a.g.packet_header_size + ic * col_size;
            }
            if (hs >= 0) {
                const uint32_t p = (uint32_t)hs / cpp, ic = (uint32_t)hs - p * cpp;
                hofs[k] = p * (uint32_t)a.packet_stride + a.g.packet_header_size + ic * col_size;
            }
        }
        s_colofs[j] = ofs;
    }
    if (tid < 4) s_acc[tid] = 0;
    for (uint32_t j = tid; j < (uint32_t)TW; j += NT) s_gate[j] = 0;
    const LutDev lut = (XYZM != 0) ? a.luts[f % a.n_luts] : LutDev{};
    // Everything this workgroup needs from memory besides its tile -- the (measurement_id, status,
    // timestamp) words of its columns, the destagger offsets and the per-beam table rows -- is put in
    // flight BEHIND the tile's loads (issue_small, called from the staging loop) and used after the tile
    // has been written to LDS: a consumer right behind one of these loads would cost the workgroup a
    // full memory latency before its tile is even requested (18 % of its life, tools/ab/phase_timing.sh).
    RawWin w_mid[NJ], w_st[NJ], w_ts[NJ], w_alert[NJ];
    uint64_t pk_ts[NJ];
    constexpr int NBT = 3;   // at most 84 rows per tile (setup_wide): three table doubles per thread
    int32_t r_off = 0;
    double r_beam[NBT];
    // the tile's column poses (xyz_poses): 12 of the 16 doubles of each column as 16 B pieces, all in flight together with
    // the other small tables (a loop of load -> convert -> LDS write would pay one memory latency per piece: 12 in a row
    // for a 256-column tile, measured at +27 % of the kernel's time)
    constexpr bool HAS_POSES = POSES && (XYZM == 1 || XYZM == 2);
    constexpr int NPOSE = HAS_POSES ? (TW * 6 + NT - 1) / NT : 1;
    double2 r_pose[NPOSE];
    auto issue_small = [&]() {
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const uint32_t j = tid + (uint32_t)k * NT, c = c0 + j;
            w_mid[k] = w_st[k] = w_ts[k] = w_alert[k] = RawWin{{0u, 0u, 0u}, 0};
            pk_ts[k] = 0;
            if (j >= (uint32_t)TW || c >= W) continue;
            const uint32_t p = c / cpp, ic = c - p * cpp;
            uint32_t cofs = p * (uint32_t)a.packet_stride + a.g.packet_header_size + ic * col_size;
            if (mapped) {
                cofs = hofs[k];
                if (cofs == 0xffffffffu) continue;
            } else if (p >= count) continue;
            const uint8_t* colp = fbase + cofs;
            w_mid[k] = window_global_masked_issue(colp + a.g.col_measurement_id.offset, a.g.col_measurement_id.mask);
            w_st[k] = window_global_masked_issue(colp + a.g.col_status.offset, a.g.col_status.mask);
            if (rc == 0) {
                if (a.timestamp) w_ts[k] = window_global_masked_issue(colp + a.g.col_timestamp.offset, a.g.col_timestamp.mask);
                if (ic == 0 && !mapped) {   // packet-level outputs of the general mapping come from k_slotmap
                    if (a.packet_timestamp && a.host_timestamps)
                        pk_ts[k] = a.host_timestamps[(size_t)f * a.slots_per_frame + p];
                    if (a.alert_flags)
                        w_alert[k] = window_global_issue(fbase + (size_t)p * a.packet_stride + a.g.alert_flags.offset);
                }
            }
        }
        if (a.any_destagger && tid < nrows) r_off = a.dst_offsets[r0 + tid];
        if (XYZM == 1 || XYZM == 2) {
#pragma unroll
            for (int k = 0; k < NBT; ++k) {
                const uint32_t i = tid + (uint32_t)k * NT;
                r_beam[k] = i < nrows * 9 ? lut.beam_tab[(size_t)r0 * 9 + i] : 0.0;
            }
        }
        if constexpr (HAS_POSES) {
            if (a.xyz_poses) {
#pragma unroll
                for (int k = 0; k < NPOSE; ++k) {
                    const uint32_t i = tid + (uint32_t)k * NT, j = i / 6u, kk = (i - j * 6u) * 2u;
#ifdef OUSTER_ABLATE_POSE_LOAD   // experiment builds only: no pose is read
                    r_pose[k] = double2{kk == 0 ? 1.0 : 0.0, 0.0};
                    (void)j;
#else
                    r_pose[k] = (i < (uint32_t)TW * 6u && c0 + j < W)
                                    ? *(const double2*)(a.xyz_poses + ((size_t)f * W + c0 + j) * 16 + kk) : double2{0.0, 0.0};
#endif
                }
            }
        }
    };
    __syncthreads();
    PHASE_STAMP(1);

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
        constexpr int DEPTH = 9;   // 256 columns x 17 chunks = 17 per thread for 256 B pieces: two batches of loads (18 in one batch cost 231 VGPRs = two waves per SIMD; 9 keep the 155 of the row loop)
        for (uint32_t base = 0; base < total; base += NT * DEPTH) {
            u32x4 t[DEPTH];
            // (column j << 16) | (piece-relative dword index of t[k].x as int16; -32768 drops all four):
            // one register per chunk -- the loads of a whole tile are in flight together and the kernel
            // must stay under 168 VGPRs for three waves per SIMD
            uint32_t pj[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                // Every load is issued, from an address that is always safe to read; what must not land in the tile is dropped
                // through pj[k].  (Guarded loads -- the form this loop had -- are each waited for on their own: the compiler
                // ends every guarded region with s_waitcnt vmcnt(0), eighteen memory round trips per tile instead of one.)
                const bool in = base + k * NT + tid < total;
                const uint32_t ofs = s_colofs[in ? j : 0u];
                bool ok = in && ofs != 0xffffffffu;
                const uint8_t* src = fbase + (ok ? ofs : 0u) + rowofs;
                const uint32_t delta = (uint32_t)((uintptr_t)src & 15u);
                const uint8_t* q = src - delta + ch * 16u;
                ok = ok && ch * 16u < delta + piece * 4u;
                // the last chunk of the frame buffer may reach past its end: read the 16 bytes that end there instead; its
                // dwords then sit `shift` positions later in the register, i.e. the first one belongs `shift` dwords earlier
                // in the piece (those are bytes of the same packet, written with the same values by the chunk before)
                const uint8_t* lim = fend - 16;
                const bool tail = q > lim;
                const uint8_t* qs = ok ? (tail ? lim : q) : fbase;
                const int32_t shift = tail ? (int32_t)((q - lim) >> 2) : 0;
#if OUSTER_NT_LOADS
                t[k] = __builtin_nontemporal_load((const u32x4*)qs);
#else
                t[k] = *(const u32x4*)qs;
#endif
                pj[k] = ok ? ((j << 16) | (uint32_t)(((int32_t)(ch * 4u) - (int32_t)(delta >> 2) - shift) & 0xffff)) : 0x8000u;
                j += dj; ch += dc;
                if (ch >= NCH) { ch -= NCH; ++j; }
            }
            if (base == 0) issue_small();   // behind the tile's loads, ahead of the wait for them
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int32_t p0 = (int32_t)(int16_t)(pj[k] & 0xffffu);
                const uint32_t sj = (pj[k] >> 16) * slot;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int32_t pp = p0 + w;
                    if (pp >= 0 && pp < (int32_t)piece) s_tile[sj + (uint32_t)pp] = t[k][w];
                }
            }
        }
        // dead columns and the rows past H keep whatever the LDS held: nothing reads them (vq / nrows)
        if (tid < 4) s_tile[TW * slot + tid] = 0;  // slack read by 64-bit windows
    }
    // the small tables that were in flight behind the tile
    if (a.any_destagger && tid < nrows) s_off[tid] = r_off;
    if (XYZM == 1 || XYZM == 2) {
#pragma unroll
        for (int k = 0; k < NBT; ++k) {
            const uint32_t i = tid + (uint32_t)k * NT;
            if (i < nrows * 9) s_beam[i] = r_beam[k];
        }
    }
    // pose table, transposed (element k of every column next to each other: a lane's four columns are one 16 / 32 B read,
    // a wave's reads conflict free) and cast to the xyz element type; published by the barrier behind the classification
    const void* s_pose = nullptr;
    if constexpr (HAS_POSES) {
        if (a.xyz_poses) {
            using XT = typename std::conditional<XYZM == 1, float, double>::type;
            XT* sp = (XT*)((uint8_t*)smem + a.pose_lds_off);
#pragma unroll
            for (int k = 0; k < NPOSE; ++k) {
                const uint32_t i = tid + (uint32_t)k * NT, j = i / 6u, kk = (i - j * 6u) * 2u;
                if (i < (uint32_t)TW * 6u) {
                    sp[kk * (uint32_t)TW + j] = (XT)r_pose[k].x;
                    sp[(kk + 1u) * (uint32_t)TW + j] = (XT)r_pose[k].y;
                }
            }
            s_pose = sp;
#ifdef OUSTER_ABLATE_POSE_ALL   // experiment builds only: the POSES instantiation runs, its pose code does not
            s_pose = nullptr;
#endif
        }
    }
    if (!mapped && rc == 0 && tile == 0 && tid == 0 && a.frame_meta) a.frame_meta[f] = frame_meta_first_present(a.g, fbase, a.packet_stride, count);

    PHASE_STAMP(2);
    // ---- classify my columns (every row chunk needs the validity; the first one also publishes)
    {
        uint32_t n_valid = 0, n_stray = 0, n_dead = 0;
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const uint32_t j = tid + (uint32_t)k * NT, c = c0 + j;
            if (j >= (uint32_t)TW) continue;
            const uint32_t p = c / cpp;
            const bool present = c < W && (mapped ? hofs[k] != 0xffffffffu : p < count);
            const uint32_t m_id = (uint16_t)apply_bits(window_compose(w_mid[k]), a.g.col_measurement_id.mask, a.g.col_measurement_id.shift);
            const uint32_t st = (uint32_t)apply_bits(window_compose(w_st[k]), a.g.col_status.mask, a.g.col_status.shift);
            const bool live = present && (st & 1u) && m_id < W;
            bool stray = live && m_id != c;   // never under the general mapping: the header map holds slots whose column IS c
            const bool v = live && !stray;    // the column's header
            const bool vp = mapped ? (c < W && s_colofs[j] != 0xffffffffu) : v;   // its pixels (another slot's under the block path)
            s_valid[j] = vp ? 1u : 0u;
            n_dead += (!vp && c < W) ? 1u : 0u;
            if (rc != 0 || c >= W) continue;
            if (a.hdr_words && !mapped) a.hdr_words[(size_t)f * W + c] = present ? (m_id | ((st & 1u) << 16)) : 0u;
            if (c == p * cpp && !mapped) {  // batch_lidar_packet, lidar_frame.cpp:1534-1539
                const bool want_pk = a.packet_timestamp || a.alert_flags;
                const bool home = present && m_id / cpp == p;
                if (present && !home && want_pk && m_id / cpp < npo) stray = true;
                if (a.packet_timestamp && a.host_timestamps)
                    a.packet_timestamp[(size_t)f * npo + p] = home ? pk_ts[k] : 0ull;
                if (a.alert_flags && home)
                    a.alert_flags[(size_t)f * npo + p] =
                        (uint8_t)apply_bits(window_compose(w_alert[k]), a.g.alert_flags.mask, a.g.alert_flags.shift);
            }
            n_valid += v ? 1u : 0u;
            n_stray += stray ? 1u : 0u;
            if (a.timestamp)
                a.timestamp[(size_t)f * W + c] = v ? apply_bits(window_compose(w_ts[k]), a.g.col_timestamp.mask, a.g.col_timestamp.shift) : 0ull;
            if (a.measurement_id) a.measurement_id[(size_t)f * W + c] = v ? (uint16_t)c : (uint16_t)0;
            if (a.status) a.status[(size_t)f * W + c] = v ? st : 0u;
        }
        // one LDS atomic per wave, not per lane: 256 lanes adding to one word are served one after the other
        const uint32_t packed = wave_sum(n_valid | (n_stray << 10) | (n_dead << 20));   // at most 4 columns per lane: 256 per wave
        if ((tid & 63u) == 0) {
            if (rc == 0) {
                if (packed & 0x3ffu) atomicAdd(&s_acc[0], packed & 0x3ffu);
                if ((packed >> 10) & 0x3ffu) atomicAdd(&s_acc[1], (packed >> 10) & 0x3ffu);
            }
            if (packed >> 20) atomicAdd(&s_acc[2], packed >> 20);
        }
    }
    __syncthreads();
    if (rc == 0 && tid == 0 && !mapped) {
        fast_publish(a, f, tile, s_acc[0], s_acc[1] != 0);
    }
    // Columns that were not received (or are invalid / not at home) decode as zeros: blank their
    // slots once, here, so that the row loop needs no per-pixel select.  Rare, hence the uniform test.
    if (s_acc[2] != 0) {
        const uint32_t piece = ((nrows * chan) >> 2) + 1;   // + the pad dword 64-bit windows may touch
        for (uint32_t j = tid >> 2; j < (uint32_t)TW; j += NT / 4)
            if (!s_valid[j])
                for (uint32_t i = tid & 3u; i < piece && i < slot; i += 4) s_tile[j * slot + i] = 0;
        __syncthreads();
    }

    PHASE_STAMP(3);
    // ---- pixels.  lane = (row within pass, quad of 4 consecutive columns)
    const uint32_t jq = (tid % (TW / 4)) * 4;
    uint32_t vq = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) vq |= (jq + c < (uint32_t)TW && s_valid[jq + c]) ? (1u << c) : 0u;
    uint32_t px_dw[4];
    tile_px_offsets<TW / 4>(px_dw, 0u, slot);
    ColConst cc;
    if (XYZM == 1 || XYZM == 2) load_colconst(cc, lut, c0 + jq, W);
    decode_rows<S, TW / 4, XYZM, S::is_static, S::nt_stores, S::nt_xyz, POSES, 256, false, false, true>(a, s_tile, px_dw, cc, s_off, s_xyz, (XYZM == 1 || XYZM == 2) ? s_beam : nullptr,
                                 a.gate_counts ? s_gate : nullptr, lut, f, c0, r0, nrows, vq, rc, nch, s_pose);
    PHASE_STAMP(4);
#ifdef OUSTER_PHASE_TIMING
    if (pt_ && tid == 0) {   // the stores of this wave have been issued; when are they done?
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pt_[5] = __builtin_readcyclecounter();
    }
#endif
}

template <class S, int TW, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode_wide(DecodeArgs a) {
    extern __shared__ __align__(16) uint32_t smem[];
    uint32_t f, sub;
    if (!block_to_frame(a, a.tiles_per_frame * a.row_chunks, f, sub)) return;
    // the column tiles of one row chunk are neighbouring blocks of an XCD: together they write whole
    // 8 KB rows at the same time (ordering the row chunks of a column tile next to each other instead
    // shares input cache lines but measured 6 % slower)
    wide_tile<S, TW, XYZM, POSES, false>(a, smem, f, sub % a.tiles_per_frame, sub / a.tiles_per_frame, nullptr, nullptr,
                                         a.rows_per_tile, a.row_chunks);
}

// ------------------------------------------------------------------------------------
// k_decode_wide_fixup: the fix-up pass on wide tiles (round 4; k_decode_fixup's 64-column tiles remain for formats the wide
// tiles cannot take).  A persistent grid as before: every workgroup lists the frames flagged with this call's tag and the
// workgroups share the work out through tickets.  Tickets, in this order:
//   one per flagged frame      LEAD: resolve_frame gives the frame's real column maps (from the packed header words the
//                              optimistic pass left behind); they go to a.slot_map / a.hdr_map together with the mask of
//                              column tiles whose maps differ from what the optimistic pass assumed ("slot c holds column c,
//                              anything else reads as zeros"), then the frame's ready word is published (tag | mask); then
//                              the frame's packet-level outputs, frame-level values and valid-column count.
//   one per (frame, row chunk, k)   REDO: decodes the k-th wrong column tile, if there is one -- a frame with two packets
//                              swapped costs one column tile, not the frame (r03: every flagged frame was redone whole).
//                              Where do the maps come from?  ONE look at the ready word: published -- a later round of
//                              tickets -- they are read from global memory, for free; not yet -- the first round, whose LEAD
//                              tickets started when this one did -- the workgroup resolves the frame itself (6 us for whole,
//                              aligned packets) and decodes from its LDS copy.  Nobody ever waits for anybody: a version
//                              that polled sat through the LEAD ticket's resolution, the write-through of the maps and a
//                              poll interval (13 us, tools/ab/phase_fixup.py) and needed a bounded-wait fallback; a version
//                              in which every ticket resolved for itself paid 6 us per ticket at scale (0.60 against 0.63).
// Maps and ready words are written and read with agent-scope atomics: the XCDs' L2s are not coherent with each other for
// plain accesses.  Few flagged frames: short tiles (fix_rows_small), so that the damage spreads over the chip; many: the
// launch's tall tiles.  A workgroup's first ticket is its own number (no atomic on a clean batch: 512 workgroups adding to
// one word are served one after the other, 4 us on every call); the counter hands out the tickets behind those, each asked
// for when the item before it starts.  The counter lives in frame_state behind the sequence words, one per tag parity: this
// call's starts at zero (zeroed by the call before), the other is zeroed for the next call; the ready words ([ready_off + f],
// the buffer's second half) carry the tag and are never cleared.
// ------------------------------------------------------------------------------------
#ifdef OUSTER_PHASE_TIMING   // experiment builds (tools/ab/phase_timing.sh): per workgroup 64 words: [0] start, [1] events, then 4 per ticket
#define FSTAMP_BEGIN() uint64_t fs0_ = __builtin_readcyclecounter(), fs1_ = 0
#define FSTAMP_MID() do { fs1_ = __builtin_readcyclecounter(); } while (0)
#define FSTAMP_END(kind, n) do { if (a.phase_times && tid == 0) { uint64_t* q_ = a.phase_times + (size_t)blockIdx.x * 64; const uint64_t e_ = q_[1]; \
    if (e_ < 15) { q_[2 + 4 * e_] = (kind) | ((uint64_t)(n) << 8); q_[3 + 4 * e_] = fs0_; q_[4 + 4 * e_] = fs1_; q_[5 + 4 * e_] = __builtin_readcyclecounter(); q_[1] = e_ + 1; } } } while (0)
#else
#define FSTAMP_BEGIN() do {} while (0)
#define FSTAMP_MID() do {} while (0)
#define FSTAMP_END(kind, n) do {} while (0)
#endif
template <class S, int TW, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode_wide_fixup(DecodeArgs a) {
    constexpr int NT = 256;
    extern __shared__ __align__(16) uint32_t smem[];
    __shared__ uint16_t s_list[FIXUP_CHUNK];
    __shared__ uint32_t s_cnt[FIXUP_CHUNK / 64];
    __shared__ uint32_t s_n, s_nvalid, s_dirty;
    __shared__ unsigned long long s_ticket, s_ready;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, cpp = a.g.columns_per_packet, npo = a.n_packets_out;
    constexpr uint32_t SPLIT_MAX = 16;   // REDO tickets per (frame, row chunk): ticket s takes every SPLIT-th dirty column tile
    uint32_t TRd = a.rows_per_tile, nchd = a.row_chunks;
    const uint64_t tag = a.frame_state[FS_TAG];
    // One counter for the whole grid: a flagged frame's tiles go wherever a workgroup is free.  (Keeping a frame on one XCD
    // -- right for k_decode_fixup's 64-column tiles, whose partial cache lines must meet in one L2 -- limits ONE damaged frame
    // to an eighth of the chip's bandwidth: 115 us for a frame with eight dirty column tiles, tools/ab/fixup_kinds2.py.)
    const uint32_t xcd = 0u, nx = 1u;
    unsigned long long* ctr = (unsigned long long*)&a.frame_state[FS_TICKET + (tag & 1u) * 8u + xcd];
    unsigned long long* ready = (unsigned long long*)&a.frame_state[a.ready_off];
    if (blockIdx.x < 8 && tid == 0) a.frame_state[FS_TICKET + ((tag + 1u) & 1u) * 8u + blockIdx.x] = 0;   // the next call's counters
    // first ticket = the workgroup's own number; the next one is asked for when an item starts and looked at when it ends
    unsigned long long ahead = 0;
    auto pull_ahead = [&]() { if (tid == 0) ahead = atomicAdd(ctr, 1ull) + gridDim.x; };
    auto take_ahead = [&]() -> unsigned long long {
        __syncthreads();
        if (tid == 0) s_ticket = ahead;
        __syncthreads();
        return s_ticket;
    };
#ifdef OUSTER_PHASE_TIMING
    if (a.phase_times && tid == 0) { a.phase_times[(size_t)blockIdx.x * 64] = __builtin_readcyclecounter(); a.phase_times[(size_t)blockIdx.x * 64 + 1] = 0; }
#endif
    const ResolveLds L(smem, W, npo, a.slots_per_frame);
    // the frame's maps in LDS (L.pix / L.hdr); `lead`: also its packet-level outputs, frame-level values and valid-column count
    auto resolve = [&](uint32_t f, bool lead) {
        const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
        uint32_t count = a.slots_per_frame;
        if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
        if (tid == 0) { s_nvalid = 0; s_dirty = 0; }
        resolve_frame<NT>(a.g, fbase, a.packet_stride, count, npo, L, lead, a.hdr_words ? a.hdr_words + (size_t)f * W : nullptr);
        // which column tiles the optimistic pass got wrong: it wrote slot c where that is live and at home, zeros otherwise;
        // the LEAD ticket also leaves the maps in global memory for the tickets of later rounds
        uint32_t n = 0, dm = 0;
        for (uint32_t c = tid; c < W; c += NT) {
            const int32_t px = L.pix[c], hd = L.hdr[c];
            if (lead) {
                __hip_atomic_store(&a.slot_map[(size_t)f * W + c], px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.hdr_map[(size_t)f * W + c], hd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            n += hd >= 0 ? 1u : 0u;
            int32_t expect = -1;
            if (c / cpp < count && L.hd[c] == (c | 0x10000u)) expect = (int32_t)c;
            if (px != expect || hd != expect) dm |= 1u << (c / TW);
        }
        n = wave_sum(n);
        dm = wave_or(dm);
        if ((tid & 63u) == 0) {
            if (n) atomicAdd(&s_nvalid, n);
            if (dm) atomicOr(&s_dirty, dm);
        }
        // (LEAD) The maps were stored with agent-scope atomics (written through to where the other XCDs can see them); every
        // wave waits for its stores before the barrier, the word that announces them is stored behind it.  No cache-wide
        // release: a buffer_wbl2 here would have to write back every tile the XCD has redone so far.
        if (lead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!lead) return;
        if (tid == 0)
            __hip_atomic_store(&ready[f], (unsigned long long)((tag << 32) | s_dirty), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Nobody needs the rest soon: the frame's packet-level outputs (packet_timestamp is zeroed at frame start,
        // lidar_frame.cpp:1719, alert_flags is not), frame-level values and valid-column count
        for (uint32_t i = tid; i < npo; i += NT) {
            const int32_t p = L.pkm[i];
            if (a.packet_timestamp && a.host_timestamps)
                a.packet_timestamp[(size_t)f * npo + i] = p >= 0 ? a.host_timestamps[(size_t)f * a.slots_per_frame + p] : 0ull;
            if (a.alert_flags && p >= 0)
                a.alert_flags[(size_t)f * npo + i] = (uint8_t)apply_bits(
                    window_global(fbase + (size_t)p * a.packet_stride + a.g.alert_flags.offset), a.g.alert_flags.mask,
                    a.g.alert_flags.shift);
        }
        if (tid == 0 && a.frame_meta) {
            ouster_hip_frame_meta m = frame_meta_general(a, fbase, count);
            m.n_valid_columns = s_nvalid;
            a.frame_meta[f] = m;
        }
    };
    unsigned long long ticket = blockIdx.x, done = 0;
    for (uint32_t base = 0; base < a.n_frames; base += FIXUP_CHUNK) {
        // the flagged frames of this chunk, in frame order, the same list in every workgroup (see k_decode_fixup)
        const uint32_t nfr = min(FIXUP_CHUNK, a.n_frames - base);
        constexpr uint32_t ROUNDS = FIXUP_CHUNK / NT;
        uint64_t mine[ROUNDS];
        bool flagged[ROUNDS];
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            const uint32_t i = r * NT + tid;
            flagged[r] = i < nfr && a.frame_state[FS_WORDS + base + i] == tag;
            mine[r] = __ballot(flagged[r]);
            if ((tid & 63u) == 0) s_cnt[i >> 6] = (uint32_t)__popcll(mine[r]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            const uint32_t i = r * NT + tid, grp = i >> 6;
            uint32_t before = 0;
            for (uint32_t k = 0; k < grp; ++k) before += s_cnt[k];
            if (flagged[r]) s_list[before + (uint32_t)__popcll(mine[r] & ((1ull << (tid & 63u)) - 1ull))] = (uint16_t)i;
        }
        if (tid == 0) {
            uint32_t n = 0;
            for (uint32_t k = 0; k < FIXUP_CHUNK / 64; ++k) n += s_cnt[k];
            s_n = n;
        }
        __syncthreads();
        const uint32_t n_flagged = s_n;
        const uint32_t my_frames = n_flagged > xcd ? (n_flagged - xcd + nx - 1u) / nx : 0u;   // flagged frames of my XCD
        // few damaged frames: short tiles, one ticket per dirty tile (the chip is idle, latency counts); many: the launch's
        // tall tiles (a workgroup moves 1.8 x the bytes per microsecond through a 32-row tile than through four 8-row ones)
        // and fewer tickets that find no work
        // (one ticket per dirty tile whatever the number of flagged frames: with two to four tiles behind one ticket the pass
        // took as long as its unluckiest ticket -- 228 us for 64 compacted frames, tools/ab/fixup_kinds.py; a ticket that finds
        // no work costs 2 us)
        const uint32_t SPLIT = min(a.tiles_per_frame, SPLIT_MAX);
        TRd = n_flagged <= 8u ? min(a.fix_rows_small, a.rows_per_tile) : a.rows_per_tile;
        nchd = (a.g.pixels_per_column + TRd - 1u) / TRd;
        const uint32_t bpf = nchd * SPLIT;
        const unsigned long long items = (unsigned long long)my_frames * (1u + bpf);
        if (items == 0) continue;
        while (ticket < done + items) {
            const uint32_t it = (uint32_t)(ticket - done);
            pull_ahead();
            FSTAMP_BEGIN();
            if (it < my_frames) {
                resolve(base + s_list[it * nx + xcd], true);
                FSTAMP_END(1u, 0u);
            } else {
                // ticket order: every frame's FIRST wrong tile (all its row chunks), then every frame's second, ... -- the
                // tickets most likely to find work are handed out first, while every workgroup is still free; frame by frame,
                // the last frames' tiles all fell into the second round behind workgroups that already had a tile to do
                const uint32_t j = it - my_frames, per_share = my_frames * nchd;
                const uint32_t share = j / per_share, rest = j % per_share;
                const uint32_t f = base + s_list[(rest / nchd) * nx + xcd], rc = rest % nchd;
                // ONE look at the frame's ready word.  There (a later round of tickets): the maps and the mask of wrong tiles are
                // in global memory, nothing to compute.  Not there yet (the first round: the LEAD ticket started when this one
                // did): resolve the frame here -- 6 us for whole, aligned packets -- instead of sitting through the LEAD
                // ticket's resolution, its write-through of the maps and a poll interval (13 us, tools/ab/phase_fixup.py).
                // Nobody ever waits for anybody.
                if (tid == 0) s_ready = __hip_atomic_load(&ready[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                FSTAMP_MID();
                const unsigned long long v = s_ready;
                const bool published = (v >> 32) == (tag & 0xffffffffull);
                uint32_t mask = (uint32_t)v;
                if (!published) {
                    resolve(f, false);
                    mask = s_dirty;
                }
                uint32_t rank = 0, ntl = 0;
                for (uint32_t m = mask; m; m &= m - 1u, ++rank) {
                    if (rank % SPLIT != share) continue;
                    const uint32_t tile = (uint32_t)__builtin_ctz(m);
                    if (published) {
                        wide_tile<S, TW, XYZM, POSES, false>(a, smem, f, tile, rc, nullptr, nullptr, TRd, nchd);   // reads the maps with agent-scope loads
                    } else {
                        if (ntl) resolve(f, false);   // more than SPLIT_MAX column tiles: the maps lay under the tile before
                        wide_tile<S, TW, XYZM, POSES, true>(a, smem, f, tile, rc, L.pix, L.hdr, TRd, nchd);
                    }
                    __syncthreads();
                    ++ntl;
                }
                FSTAMP_END(2u, ntl);
            }
            ticket = take_ahead();
        }
        done += items;
    }
    // valid-column counts of the clean frames (flagged ones got theirs from their RESOLVE ticket)
    if (a.frame_meta) {
        for (uint32_t f = blockIdx.x * NT + tid; f < a.n_frames; f += gridDim.x * NT) {
            if (a.frame_state[FS_WORDS + f] == tag) continue;
            uint32_t n = 0;
            for (uint32_t t = 0; t < a.fast_tiles; ++t) n += a.tile_valid[(size_t)f * a.fast_tiles + t];
            a.frame_meta[f].n_valid_columns = n;
        }
    }
    if (blockIdx.x == 0 && tid == 0) a.frame_state[FS_SEQ] = tag;  // the next call tags with tag + 1
#ifdef OUSTER_PHASE_TIMING
    if (a.phase_times && tid == 0) a.phase_times[(size_t)blockIdx.x * 64 + 63] = __builtin_readcyclecounter();
#endif
}

// ------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------
// dynamic LDS above 48 KB needs the kernel's attribute raised (per kernel and device); remembered so
// the steady state makes no runtime call
struct LdsGrant {
    std::atomic<uint32_t> bytes[16];
    LdsGrant() { for (auto& b : bytes) b.store(0); }
};
template <class K>
static hipError_t allow_lds(K kernel, size_t lds, int device, LdsGrant& g) {
    if (lds <= 48 * 1024) return hipSuccess;
    std::atomic<uint32_t>& have = g.bytes[device & 15];
    if (have.load(std::memory_order_acquire) >= lds) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) have.store((uint32_t)lds, std::memory_order_release);
    return e;
}

template <class S, int TILE, int XYZM>
static hipError_t launch_decode_x(const DecodeArgs& a, dim3 grid, size_t lds, int device, hipStream_t st) {
    if constexpr (XYZM == 1 || XYZM == 2) {
        if (a.xyz_poses) {
            static LdsGrant done_p;
            hipError_t e = allow_lds(k_decode<S, TILE, XYZM, true>, lds, device, done_p);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_decode<S, TILE, XYZM, true>), grid, dim3(256), lds, st, a);
            return hipGetLastError();
        }
    }
    static LdsGrant done;
    hipError_t e = allow_lds(k_decode<S, TILE, XYZM>, lds, device, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode<S, TILE, XYZM>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

template <class S, int TILE>
static hipError_t launch_decode_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds, int device, hipStream_t st) {
    switch (xyzm) {
        case 0: return launch_decode_x<S, TILE, 0>(a, grid, lds, device, st);
        case 1: return launch_decode_x<S, TILE, 1>(a, grid, lds, device, st);
        case 2: return launch_decode_x<S, TILE, 2>(a, grid, lds, device, st);
        default: return launch_decode_x<S, TILE, 3>(a, grid, lds, device, st);
    }
}

template <class S, int TILE, int XYZM>
static hipError_t launch_fixup_x(const DecodeArgs& a, dim3 grid, size_t lds, int device, hipStream_t st) {
    if constexpr (XYZM == 1 || XYZM == 2) {
        if (a.xyz_poses) {
            static LdsGrant done_p;
            hipError_t e = allow_lds(k_decode_fixup<S, TILE, XYZM, true>, lds, device, done_p);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_decode_fixup<S, TILE, XYZM, true>), grid, dim3(256), lds, st, a);
            return hipGetLastError();
        }
    }
    static LdsGrant done;
    hipError_t e = allow_lds(k_decode_fixup<S, TILE, XYZM>, lds, device, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_fixup<S, TILE, XYZM>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

template <class S, int TILE>
static hipError_t launch_fixup_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds, int device, hipStream_t st) {
    switch (xyzm) {
        case 0: return launch_fixup_x<S, TILE, 0>(a, grid, lds, device, st);
        case 1: return launch_fixup_x<S, TILE, 1>(a, grid, lds, device, st);
        case 2: return launch_fixup_x<S, TILE, 2>(a, grid, lds, device, st);
        default: return launch_fixup_x<S, TILE, 3>(a, grid, lds, device, st);
    }
}

// bytes of the tile's pose table (ouster_hip_frame_out::xyz_poses): 12 values of the xyz element type per column
static size_t pose_lds_bytes(const DecodeArgs& a, int xyzm, int cols) {
    return (a.xyz_poses && (xyzm == 1 || xyzm == 2)) ? (size_t)cols * 12 * (xyzm == 1 ? 4 : 8) : 0;
}

hipError_t OUSTER_SPEC_FN(launch_decode)(const DecodeArgs& a_in, int tile, int xyzm, int device, hipStream_t st) {
    const uint32_t tpf = a_in.tiles_per_frame;
    if (a_in.mode == MODE_FIXUP) {
        DecodeArgs a = a_in;
        const size_t body = decode_lds_bytes(a.g, tile, true, a.beam_lds != 0, a.slots_per_frame);
        a.rows_per_tile = (uint32_t)body;  // where the frame list starts
        size_t lds = (body + FIXUP_CHUNK * 2 + 15) & ~(size_t)15;
        a.pose_lds_off = (uint32_t)lds;
        lds += pose_lds_bytes(a, xyzm, tile);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        const uint64_t items = (uint64_t)a.n_frames * tpf;
        // row_chunks carries the number of workgroups the device keeps resident (2 per CU)
        const dim3 grid((uint32_t)std::min<uint64_t>(items, a.row_chunks ? a.row_chunks : 512u));
        switch (tile) {
            case 64: return launch_fixup_t<SpecT, 64>(a, xyzm, grid, lds, device, st);
            case 32: return launch_fixup_t<SpecT, 32>(a, xyzm, grid, lds, device, st);
            default: return launch_fixup_t<SpecT, 16>(a, xyzm, grid, lds, device, st);
        }
    }
    DecodeArgs a = a_in;
    const uint32_t nblocks = a.xcd_map ? ((a.n_frames + 7) / 8) * 8 * tpf : a.n_frames * tpf;
    const dim3 grid(nblocks);
    size_t lds = (decode_lds_bytes(a.g, tile, a.mode != MODE_FAST, a.beam_lds != 0, a.slots_per_frame) + 15) & ~(size_t)15;
    a.pose_lds_off = (uint32_t)lds;
    lds += pose_lds_bytes(a, xyzm, tile);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    switch (tile) {
        case 64: return launch_decode_t<SpecT, 64>(a, xyzm, grid, lds, device, st);
        case 32: return launch_decode_t<SpecT, 32>(a, xyzm, grid, lds, device, st);
        default: return launch_decode_t<SpecT, 16>(a, xyzm, grid, lds, device, st);
    }
}

template <class S, int TW, int XYZM>
static hipError_t launch_decode_wide_x(const DecodeArgs& a, dim3 grid, size_t lds, int device, hipStream_t st) {
    if constexpr (XYZM == 1 || XYZM == 2) {
        if (a.xyz_poses) {
            static LdsGrant done_p;
            hipError_t e = allow_lds(k_decode_wide<S, TW, XYZM, true>, lds, device, done_p);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_decode_wide<S, TW, XYZM, true>), grid, dim3(256), lds, st, a);
            return hipGetLastError();
        }
    }
    static LdsGrant done;
    hipError_t e = allow_lds(k_decode_wide<S, TW, XYZM>, lds, device, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_wide<S, TW, XYZM>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

template <class S, int TW>
static hipError_t launch_decode_wide_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds, int device, hipStream_t st) {
    switch (xyzm) {
        case 0: return launch_decode_wide_x<S, TW, 0>(a, grid, lds, device, st);
        case 1: return launch_decode_wide_x<S, TW, 1>(a, grid, lds, device, st);
        case 2: return launch_decode_wide_x<S, TW, 2>(a, grid, lds, device, st);
        default: return launch_decode_wide_x<S, TW, 3>(a, grid, lds, device, st);
    }
}

// ------------------------------------------------------------------------------------
// k_decode_wide_resolved: ONE launch for small batches (a tick of a few sensors, a single frame): every workgroup resolves
// its frame's column maps itself (resolve_frame: one round of header reads that hit L2 after the first tile, then LDS work)
// and decodes its tile from them.  No optimistic pass, no flags, no second launch: a second launch costs such a batch more
// than its own decode (5 - 7 us on the stream against 6 - 10), while the redundant resolution of a few hundred tiles costs
// a few microseconds of latency and no bandwidth worth counting.  Any buffer shape (it is the general mapping).
// ------------------------------------------------------------------------------------
template <class S, int TW, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode_wide_resolved(DecodeArgs a) {
    constexpr int NT = 256;
    extern __shared__ __align__(16) uint32_t smem[];
    __shared__ uint32_t s_nvalid;
    const uint32_t tid = threadIdx.x;
    uint32_t f, sub;
    if (!block_to_frame(a, a.tiles_per_frame * a.row_chunks, f, sub)) return;
    const uint32_t tile = sub % a.tiles_per_frame, rc = sub / a.tiles_per_frame;
    const uint32_t W = a.g.columns_per_frame, npo = a.n_packets_out;
    const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
    uint32_t count = a.slots_per_frame;
    if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
    const ResolveLds L(smem, W, npo, a.slots_per_frame);
    const bool lead = tile == 0 && rc == 0;
    if (tid == 0) s_nvalid = 0;
#ifdef OUSTER_PHASE_TIMING
    resolve_frame<NT>(a.g, fbase, a.packet_stride, count, npo, L, lead, nullptr, a.phase_times ? a.phase_times + (size_t)blockIdx.x * 16 : nullptr);
#else
    resolve_frame<NT>(a.g, fbase, a.packet_stride, count, npo, L, lead);
#endif
    if (lead) {
        // packet_timestamp is zeroed at frame start (lidar_frame.cpp:1719), alert_flags is not
        for (uint32_t i = tid; i < npo; i += NT) {
            const int32_t p = L.pkm[i];
            if (a.packet_timestamp && a.host_timestamps)
                a.packet_timestamp[(size_t)f * npo + i] = p >= 0 ? a.host_timestamps[(size_t)f * a.slots_per_frame + p] : 0ull;
            if (a.alert_flags && p >= 0)
                a.alert_flags[(size_t)f * npo + i] = (uint8_t)apply_bits(
                    window_global(fbase + (size_t)p * a.packet_stride + a.g.alert_flags.offset), a.g.alert_flags.mask, a.g.alert_flags.shift);
        }
        if (a.frame_meta) {
            uint32_t n = 0;
            for (uint32_t i = tid; i < W; i += NT) n += L.hdr[i] >= 0 ? 1u : 0u;
            n = wave_sum(n);
            if (n && (tid & 63u) == 0) atomicAdd(&s_nvalid, n);
            __syncthreads();
            if (tid == 0) {
                ouster_hip_frame_meta m = frame_meta_general(a, fbase, count);
                m.n_valid_columns = s_nvalid;
                a.frame_meta[f] = m;
            }
        }
    }
    wide_tile<S, TW, XYZM, POSES, true>(a, smem, f, tile, rc, L.pix, L.hdr, a.rows_per_tile, a.row_chunks);
}

template <class S, int TW, int XYZM>
static hipError_t launch_wide_resolved_x(const DecodeArgs& a, dim3 grid, size_t lds, int device, hipStream_t st) {
    if constexpr (XYZM == 1 || XYZM == 2) {
        if (a.xyz_poses) {
            static LdsGrant done_p;
            hipError_t e = allow_lds(k_decode_wide_resolved<S, TW, XYZM, true>, lds, device, done_p);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_decode_wide_resolved<S, TW, XYZM, true>), grid, dim3(256), lds, st, a);
            return hipGetLastError();
        }
    }
    static LdsGrant done;
    hipError_t e = allow_lds(k_decode_wide_resolved<S, TW, XYZM>, lds, device, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_wide_resolved<S, TW, XYZM>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

template <class S, int TW>
static hipError_t launch_wide_resolved_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds, int device, hipStream_t st) {
    switch (xyzm) {
        case 0: return launch_wide_resolved_x<S, TW, 0>(a, grid, lds, device, st);
        case 1: return launch_wide_resolved_x<S, TW, 1>(a, grid, lds, device, st);
        case 2: return launch_wide_resolved_x<S, TW, 2>(a, grid, lds, device, st);
        default: return launch_wide_resolved_x<S, TW, 3>(a, grid, lds, device, st);
    }
}

template <class S, int TW, int XYZM>
static hipError_t launch_wide_fixup_x(const DecodeArgs& a, dim3 grid, size_t lds, int device, hipStream_t st) {
    if constexpr (XYZM == 1 || XYZM == 2) {
        if (a.xyz_poses) {
            static LdsGrant done_p;
            hipError_t e = allow_lds(k_decode_wide_fixup<S, TW, XYZM, true>, lds, device, done_p);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_decode_wide_fixup<S, TW, XYZM, true>), grid, dim3(256), lds, st, a);
            return hipGetLastError();
        }
    }
    static LdsGrant done;
    hipError_t e = allow_lds(k_decode_wide_fixup<S, TW, XYZM>, lds, device, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_decode_wide_fixup<S, TW, XYZM>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

template <class S, int TW>
static hipError_t launch_wide_fixup_t(const DecodeArgs& a, int xyzm, dim3 grid, size_t lds, int device, hipStream_t st) {
    switch (xyzm) {
        case 0: return launch_wide_fixup_x<S, TW, 0>(a, grid, lds, device, st);
        case 1: return launch_wide_fixup_x<S, TW, 1>(a, grid, lds, device, st);
        case 2: return launch_wide_fixup_x<S, TW, 2>(a, grid, lds, device, st);
        default: return launch_wide_fixup_x<S, TW, 3>(a, grid, lds, device, st);
    }
}

// a.mode == MODE_FIXUP: the persistent fix-up grid (a.row_chunks etc. describe its tiles, a.fast_tiles the optimistic pass's
// column tiles, `resident` the workgroups the device keeps resident); otherwise one workgroup per tile
hipError_t OUSTER_SPEC_FN(launch_decode_wide)(const DecodeArgs& a_in, int tw, int xyzm, int device, hipStream_t st, uint32_t resident) {
    DecodeArgs a = a_in;
    const uint32_t bpf = a.tiles_per_frame * a.row_chunks;
    const bool fix = a.mode == MODE_FIXUP, resolved = a.mode == MODE_RESOLVED;
    a.wide_img_words = (uint32_t)tw * (a.lds_col_slot >> 2) + 4u;
    if (fix || resolved)   // resolve_frame's scratch lies under the tile image
        a.wide_img_words = std::max<uint32_t>(a.wide_img_words, (uint32_t)((slotmap_lds_bytes(a.g.columns_per_frame, a.g.columns_per_packet, a.slots_per_frame) / 4 + 3) & ~(size_t)3));
    size_t lds = (decode_wide_lds_bytes(tw, a.rows_per_tile, a.wide_img_words) + 15) & ~(size_t)15;
    a.pose_lds_off = (uint32_t)lds;
    lds += pose_lds_bytes(a, xyzm, tw);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (fix) {
        const uint64_t items = (uint64_t)a.n_frames * bpf;
        uint32_t g = (uint32_t)std::min<uint64_t>(items, resident ? resident : 512u);
        const dim3 grid(g);
        switch (tw) {   // 128 or 256 columns (narrower frames take k_decode_fixup): every width is 12 more kernels per profile to compile
            case 128: return launch_wide_fixup_t<SpecT, 128>(a, xyzm, grid, lds, device, st);
            case 256: return launch_wide_fixup_t<SpecT, 256>(a, xyzm, grid, lds, device, st);
            default: return hipErrorInvalidValue;
        }
    }
    const uint32_t nblocks = a.xcd_map ? ((a.n_frames + 7) / 8) * 8 * bpf : a.n_frames * bpf;
    const dim3 grid(nblocks);
    if (resolved) {
        switch (tw) {
            case 128: return launch_wide_resolved_t<SpecT, 128>(a, xyzm, grid, lds, device, st);
            case 256: return launch_wide_resolved_t<SpecT, 256>(a, xyzm, grid, lds, device, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (tw) {
        case 64: return launch_decode_wide_t<SpecT, 64>(a, xyzm, grid, lds, device, st);
        case 128: return launch_decode_wide_t<SpecT, 128>(a, xyzm, grid, lds, device, st);
        case 512: return launch_decode_wide_t<SpecT, 512>(a, xyzm, grid, lds, device, st);
        default: return launch_decode_wide_t<SpecT, 256>(a, xyzm, grid, lds, device, st);
    }
}

}  // namespace ouster_hip_dev
