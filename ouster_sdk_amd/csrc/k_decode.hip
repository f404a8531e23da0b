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
#include "wide_tile.h"

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
    // the ticket counters of k_decode_wide_fixup are keyed by tag parity and zeroed by the call before: this kernel advances the
    // sequence word like the wide one, so it zeroes the next call's counters like the wide one (a context may alternate)
    if (blockIdx.x < 8 && tid == 0) a.frame_state[FS_TICKET + ((tag + 1u) & 1u) * 8u + blockIdx.x] = 0;
    const bool any_flagged = a.frame_state[FS_ANY] == tag;   // the launch-wide word: a clean batch skips the listing altogether
    for (uint32_t base = 0; any_flagged && base < a.n_frames; base += FIXUP_CHUNK) {
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


template <class S, int TW, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode_wide(DecodeArgs a) {
    extern __shared__ __align__(16) uint32_t smem[];
    uint32_t f, sub;
    if (!block_to_frame(a, a.tiles_per_frame * a.row_chunks, f, sub)) return;
    // the column tiles of one row chunk are neighbouring blocks of an XCD: together they write whole
    // 8 KB rows at the same time (ordering the row chunks of a column tile next to each other instead
    // shares input cache lines but measured 6 % slower)
    wide_tile<S, TW, XYZM, POSES, false>(a, smem, f, sub % a.tiles_per_frame, sub / a.tiles_per_frame, nullptr, nullptr,
                                         a.rows_per_tile, a.row_chunks, a.mode == MODE_GENERAL);   // k_slotmap's maps, or the optimistic pass
}

// ------------------------------------------------------------------------------------
// k_decode_wide_fixup: the fix-up crew on wide tiles (fixup_crew, wide_tile.h): a persistent grid behind an optimistic pass of
// any kind (k_decode's 64-column tiles, k_decode_wide, the persistent k_decode_stream*).  A clean batch -- nobody raised the
// launch-wide word FS_ANY to this call's tag -- is two scalar loads and out (round 5; round 4: every workgroup read and listed
// the frame words, 8 us of dependent round trips behind every optimistic pass).
// ------------------------------------------------------------------------------------
template <class S, int TW, int XYZM, bool POSES = false>
__global__ __launch_bounds__(256) void k_decode_wide_fixup(DecodeArgs a) {
    constexpr int NT = 256;
    extern __shared__ __align__(16) uint32_t smem[];
    CrewLds* C = (CrewLds*)((uint8_t*)smem + a.crew_lds_off);
#ifdef OUSTER_PHASE_TIMING
    if (a.phase_times && threadIdx.x == 0) a.phase_times[(size_t)blockIdx.x * 64 + 62] = PT_NOW();
#endif
    // both asked for at once: one round trip.  (Round 6: also asking for this thread's frame word and tile counts here -- the clean
    // batch's second dependent trip -- shortened this kernel from 1.8 to 1.3 us on the chip-wide clock and made the pipelined
    // one-frame call 0.3 us SLOWER, 14.10 against 13.80 us in alternating runs on one box: profiles/r06_latency.)
    const uint64_t tag = a.frame_state[FS_TAG], any = a.frame_state[FS_ANY];
    // the next call's ticket counters (tag parity; k_decode_fixup does the same)
    if (blockIdx.x < 8 && threadIdx.x == 0) a.frame_state[FS_TICKET + ((tag + 1u) & 1u) * 8u + blockIdx.x] = 0;
    if (any == tag) fixup_crew<S, TW, XYZM, POSES>(a, smem, C, tag);
    // valid-column counts of the clean frames (flagged ones got theirs from their LEAD ticket), then the sequence word
    sum_valid_columns<NT>(a, tag, blockIdx.x, gridDim.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.frame_state[FS_SEQ] = tag;  // the next call tags with tag + 1
#ifdef OUSTER_PHASE_TIMING
    if (a.phase_times && threadIdx.x == 0) a.phase_times[(size_t)blockIdx.x * 64 + 63] = PT_NOW();
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

#ifdef OUSTER_EXPERIMENTS   // the one-launch form for small batches: measured slower than the optimistic pass + fix-up launch
                            // (profiles/r05/one_launch_ab.json); out of the default build since round 6 (make EXPERIMENTS=1)
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
    wide_tile<S, TW, XYZM, POSES, true>(a, smem, f, tile, rc, L.pix, L.hdr, a.rows_per_tile, a.row_chunks, true);
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
#endif  // OUSTER_EXPERIMENTS

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
    if (fix) {
        lds = (lds + 15) & ~(size_t)15;
        a.crew_lds_off = (uint32_t)lds;
        lds += (sizeof(CrewLds) + 15) & ~(size_t)15;
    }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (fix) {
        const uint64_t items = (uint64_t)a.n_frames * bpf;
        const dim3 grid((uint32_t)std::min<uint64_t>(items, resident ? resident : 512u));
        switch (tw) {   // 128 or 256 columns (narrower frames take k_decode_fixup): every width is 12 more kernels per profile to compile
            case 128: return launch_wide_fixup_t<SpecT, 128>(a, xyzm, grid, lds, device, st);
            case 256: return launch_wide_fixup_t<SpecT, 256>(a, xyzm, grid, lds, device, st);
            default: return hipErrorInvalidValue;
        }
    }
    const uint32_t nblocks = a.xcd_map ? ((a.n_frames + 7) / 8) * 8 * bpf : a.n_frames * bpf;
    const dim3 grid(nblocks);
    if (resolved) {
#ifdef OUSTER_EXPERIMENTS
        switch (tw) {
            case 128: return launch_wide_resolved_t<SpecT, 128>(a, xyzm, grid, lds, device, st);
            case 256: return launch_wide_resolved_t<SpecT, 256>(a, xyzm, grid, lds, device, st);
            default: return hipErrorInvalidValue;
        }
#else
        return hipErrorInvalidValue;   // MODE_RESOLVED is never chosen by a default build
#endif
    }
    switch (tw) {
        case 64: return launch_decode_wide_t<SpecT, 64>(a, xyzm, grid, lds, device, st);
        case 128: return launch_decode_wide_t<SpecT, 128>(a, xyzm, grid, lds, device, st);
        case 512: return launch_decode_wide_t<SpecT, 512>(a, xyzm, grid, lds, device, st);
        default: return launch_decode_wide_t<SpecT, 256>(a, xyzm, grid, lds, device, st);
    }
}

}  // namespace ouster_hip_dev
