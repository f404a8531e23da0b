// wide_tile.h -- the wide-tile device code of k_decode.hip (compiled once per packet profile):
//   wide_tile     one wide, short tile (TW columns x TR rows) of the fused decode + destagger + cartesian, 256 threads
//   fixup_crew    the fix-up pass: LEAD / REDO tickets over the frames the optimistic pass flagged
// Reference semantics: FrameBatcher::batch_lidar_packet / parse_by_col / parse_by_block, ouster_core/src/lidar_frame.cpp:1422-1576;
// what one batch() call must leave behind: lidar_frame.cpp:1530-1576.
#pragma once
#include "kernels_common.h"

namespace ouster_hip_dev {

// What the workgroup that owns column tile `tile` of frame f reports from the optimistic pass (one lane):
// a stray raises the frame's word to this call's tag; the tile's valid-column count goes to its own
// slot (plain store); the first workgroup of the launch records the tag for the fix-up pass.
__device__ __forceinline__ void fast_publish(const DecodeArgs& a, uint32_t f, uint32_t tile, uint32_t n_valid,
                                             bool stray) {
    const uint64_t tag = a.frame_state[FS_SEQ] + 1;
    if (stray) flag_frame(a, f, tag);
    if (a.frame_meta) a.tile_valid[(size_t)f * a.tiles_per_frame + tile] = (uint16_t)n_valid;
    if (f == 0 && tile == 0) a.frame_state[FS_TAG] = tag;
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
                                          const int32_t* l_pix, const int32_t* l_hdr, uint32_t TR, uint32_t nch,
                                          bool use_maps) {
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
#ifdef OUSTER_PHASE_WALL   // the chip-wide 100 MHz clock: stamps of different workgroups compare (tools/ab/phase_timing.py small)
#define PT_NOW() wall_clock64()
#else
#define PT_NOW() __builtin_readcyclecounter()
#endif
#define PHASE_STAMP(i) do { if (pt_ && tid == 0) pt_[i] = PT_NOW(); } while (0)
    if (pt_ && tid == 0) { uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); pt_[7] = xcc; }
#else
#define PHASE_STAMP(i) do {} while (0)
#endif
    PHASE_STAMP(0);

    // ---- phase 0: where my columns live (slot c holds column c): arithmetic only, nothing is loaded
    // use_maps = false: an optimistic tile (a.slot_map may be there all the same, for the fix-up crew behind it)
    const bool mapped = LMAPS || (use_maps && a.slot_map != nullptr);
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
            else if (mapped) {   // agent-scope loads: in the fix-up pass another XCD's workgroup wrote them moments ago
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
        pt_[5] = PT_NOW();
    }
#endif
}
// ------------------------------------------------------------------------------------
// The fix-up pass behind an optimistic pass.  A ONE-LAUNCH form -- the optimistic pass's workgroups meeting
// behind their last tile (arrival counter, release fence, the last arriver and bounded-wait volunteers as the fix-up crew) --
// was built in round 5 (commit 34e94d7: bit-exact, 17 GPU tests, two contexts on one device included) and measured slower
// than the two launches it replaces: the per-workgroup release fence (buffer_wbl2, served one workgroup after the other per
// XCD) costs 6 - 30 us per call, while the second launch of a clean batch costs 0 - 1.5 us since its workgroups leave after
// two scalar loads (FS_ANY below).  profiles/r05/one_launch_ab.json; removed again.
// ------------------------------------------------------------------------------------
// (CrewLds, the crew's bookkeeping in dynamic LDS, and FIXUP_CHUNK: ouster_hip_dev.h -- the launchers size the LDS)

// valid-column counts of the clean frames, from the optimistic pass's per-tile counts (flagged frames get theirs from their
// LEAD ticket).  Frames first, first + stride, ... x NT threads.
template <int NT>
__device__ __forceinline__ void sum_valid_columns(const DecodeArgs& a, uint64_t tag, uint32_t first, uint32_t stride) {
    if (!a.frame_meta) return;
    for (uint32_t f = first * NT + threadIdx.x; f < a.n_frames; f += stride * NT) {
        // the frame word and the tile counts are asked for together, eight counts at a time from clamped addresses (as a plain
        // loop every count was its own memory round trip behind the frame word's).  Round 6: measured, and the 13.8 us per
        // pipelined one-frame call did not move -- that time is the two dependent launches, not this kernel's loads.
        const uint64_t word = a.frame_state[FS_WORDS + f];
        const uint16_t* tv = a.tile_valid + (size_t)f * a.fast_tiles;
        uint32_t n = 0;
        for (uint32_t t0 = 0; t0 < a.fast_tiles; t0 += 8) {
            uint16_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) v[k] = tv[min(t0 + k, a.fast_tiles - 1u)];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) n += (t0 + k < a.fast_tiles) ? v[k] : 0u;
        }
        if (word != tag) a.frame_meta[f].n_valid_columns = n;
    }
}

// ------------------------------------------------------------------------------------
// fixup_crew (k_decode_wide_fixup): the fix-up pass on wide tiles (round 4; k_decode_fixup's 64-column tiles remain for formats the wide
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
// launch's tall tiles.  The counter hands out every ticket (a clean batch never gets here: FS_ANY), each asked
// for when the item before it starts.  The counter lives in frame_state behind the sequence words, one per tag parity: this
// call's starts at zero (zeroed by the call before), the other is zeroed for the next call; the ready words ([ready_off + f],
// the buffer's second half) carry the tag and are never cleared.
// ------------------------------------------------------------------------------------
#ifdef OUSTER_PHASE_TIMING   // experiment builds (tools/ab/phase_timing.sh): per workgroup 64 words: [0] start, [1] events, then 4 per ticket
#define FSTAMP_BEGIN() uint64_t fs0_ = PT_NOW(), fs1_ = 0
#define FSTAMP_MID() do { fs1_ = PT_NOW(); } while (0)
#define FSTAMP_END(kind, n) do { if (a.phase_times && tid == 0) { uint64_t* q_ = a.phase_times + (size_t)blockIdx.x * 64; const uint64_t e_ = q_[1]; \
    if (e_ < 15) { q_[2 + 4 * e_] = (kind) | ((uint64_t)(n) << 8); q_[3 + 4 * e_] = fs0_; q_[4 + 4 * e_] = fs1_; q_[5 + 4 * e_] = PT_NOW(); q_[1] = e_ + 1; } } } while (0)
#else
#define FSTAMP_BEGIN() do {} while (0)
#define FSTAMP_MID() do {} while (0)
#define FSTAMP_END(kind, n) do {} while (0)
#endif
template <class S, int TW, int XYZM, bool POSES>
__device__ __forceinline__ void fixup_crew(const DecodeArgs& a, uint32_t* smem, CrewLds* C, const uint64_t tag) {
    constexpr int NT = 256;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = a.g.columns_per_frame, cpp = a.g.columns_per_packet, npo = a.n_packets_out;
    constexpr uint32_t SPLIT_MAX = 16;   // REDO tickets per (frame, row chunk): ticket s takes every SPLIT-th dirty column tile
    uint32_t TRd = a.rows_per_tile, nchd = a.row_chunks;
    // One counter for the whole grid: a flagged frame's tiles go wherever a workgroup is free.  (Keeping a frame on one XCD
    // -- right for k_decode_fixup's 64-column tiles, whose partial cache lines must meet in one L2 -- limits ONE damaged frame
    // to an eighth of the chip's bandwidth: 115 us for a frame with eight dirty column tiles, tools/ab/fixup_kinds2.py.)
    unsigned long long* ctr = (unsigned long long*)&a.frame_state[FS_TICKET + (tag & 1u) * 8u];
    unsigned long long* ready = (unsigned long long*)&a.frame_state[a.ready_off];
    // Every ticket comes from the counter, the first one too -- asked for on entry, looked at behind the listing of the flagged
    // frames (which hides the 4 us that 512 workgroups adding to one word take).  Round 6: with "first ticket = my own number"
    // the pass waited for workgroups that were not there yet: of the 512 workgroups of this launch a quarter to a half start at
    // once, the rest 17 - 59 us later (kernel-entry stamps on the chip-wide clock, tools/ab/phase_timing.py fixup), and the
    // late ones held low-numbered tickets -- every frame's first wrong tile.  In arrival order the workgroups that are there
    // take the tickets with work and the late ones find the tail.
    unsigned long long ahead = 0;
    auto pull_ahead = [&]() { if (tid == 0) ahead = atomicAdd(ctr, 1ull); };
    pull_ahead();
    auto take_ahead = [&]() -> unsigned long long {
        __syncthreads();
        if (tid == 0) C->ticket = ahead;
        __syncthreads();
        return C->ticket;
    };
#ifdef OUSTER_PHASE_TIMING
    if (a.phase_times && tid == 0) { a.phase_times[(size_t)blockIdx.x * 64] = PT_NOW(); a.phase_times[(size_t)blockIdx.x * 64 + 1] = 0; }
#endif
    const ResolveLds L(smem, W, npo, a.slots_per_frame);
    // the frame's maps in LDS (L.pix / L.hdr); `lead`: also its packet-level outputs, frame-level values and valid-column count
    auto resolve = [&](uint32_t f, bool lead) {
        const uint8_t* fbase = a.packets + (size_t)f * a.slots_per_frame * a.packet_stride;
        uint32_t count = a.slots_per_frame;
        if (a.packet_counts) count = min(a.packet_counts[f], a.slots_per_frame);
        if (tid == 0) { C->nvalid = 0; C->dirty = 0; }
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
            if (n) atomicAdd(&C->nvalid, n);
            if (dm) atomicOr(&C->dirty, dm);
        }
        // (LEAD) The maps were stored with agent-scope atomics (written through to where the other XCDs can see them); every
        // wave waits for its stores before the barrier, the word that announces them is stored behind it.  No cache-wide
        // release: a buffer_wbl2 here would have to write back every tile the XCD has redone so far.  This relies on
        // gfx942 / gfx950 behaviour (sc1 stores write through, stores are counted in vmcnt): this file is built for gfx950 only.
        if (lead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!lead) return;
        if (tid == 0)
            __hip_atomic_store(&ready[f], (unsigned long long)((tag << 32) | C->dirty), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            m.n_valid_columns = C->nvalid;
            a.frame_meta[f] = m;
        }
    };
    unsigned long long ticket = 0, done = 0;
    for (uint32_t base = 0; base < a.n_frames; base += FIXUP_CHUNK) {
        // The flagged frames of this chunk.  The list must come out in the SAME order in every workgroup -- the workgroups
        // share the items out by index -- so it is compacted in frame order (ballots + a prefix over the 64-frame groups),
        // not in the arrival order of an atomic counter (r02: workgroups disagreed about the order, some tiles were redone
        // twice and others never).
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
            if ((tid & 63u) == 0) C->cnt[i >> 6] = (uint32_t)__popcll(mine[r]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < ROUNDS; ++r) {
            const uint32_t i = r * NT + tid, grp = i >> 6;
            uint32_t before = 0;
            for (uint32_t k = 0; k < grp; ++k) before += C->cnt[k];
            if (flagged[r]) C->list[before + (uint32_t)__popcll(mine[r] & ((1ull << (tid & 63u)) - 1ull))] = (uint16_t)i;
        }
        if (tid == 0) {
            uint32_t n = 0;
            for (uint32_t k = 0; k < FIXUP_CHUNK / 64; ++k) n += C->cnt[k];
            C->n = n;
        }
        __syncthreads();
        if (base == 0) ticket = take_ahead();
        const uint32_t n_flagged = C->n;
        // few damaged frames: short tiles, one ticket per dirty tile (the chip is idle, latency counts); many: the launch's
        // tall tiles (a workgroup moves 1.8 x the bytes per microsecond through a 32-row tile than through four 8-row ones)
        // and fewer tickets that find no work
        // (one ticket per dirty tile whatever the number of flagged frames: with two to four tiles behind one ticket the pass
        // took as long as its unluckiest ticket -- 228 us for 64 compacted frames, tools/ab/fixup_kinds.py; a ticket that finds
        // no work costs 2 us)
        const uint32_t SPLIT = min(a.tiles_per_frame, SPLIT_MAX);
        // (round 6: short tiles for up to gridDim / 16 flagged frames instead of 8 -- every frame's first wrong tile in one round of
        // the grid -- made bench.py's stray10 pass 99 us instead of 66: an 8-row tile takes 18.5 us where a 32-row one takes 42)
        TRd = n_flagged <= 8u ? min(a.fix_rows_small, a.rows_per_tile) : a.rows_per_tile;
        nchd = (a.g.pixels_per_column + TRd - 1u) / TRd;
        const uint32_t bpf = nchd * SPLIT;
        const unsigned long long items = (unsigned long long)n_flagged * (1u + bpf);
        if (items == 0) continue;
        // Few tickets per workgroup: the next one is asked for when this one is DONE.  Asked for at the start (which hides the
        // atomic's 1 - 2 us, and is what many flagged frames want), the tickets behind the first gridDim are all handed out at once
        // to workgroups that have just started a 38 us tile, while the workgroups whose first ticket found no work run out of
        // tickets and leave: bench.py's stray10 (26 flagged frames, 858 tickets, 304 with work, 512 workgroups) had tiles starting
        // 87 us into a pass that could be over by then -- 117 us per pass (tools/ab/phase_timing.py fixup, profiles/r06_latency).
        const bool prefetch = items > 4ull * gridDim.x;
        while (ticket < done + items) {
            const uint32_t it = (uint32_t)(ticket - done);
            if (prefetch) pull_ahead();
            FSTAMP_BEGIN();
            if (it < n_flagged) {
                resolve(base + C->list[it], true);
                FSTAMP_END(1u, 0u);
            } else {
                // ticket order: every frame's FIRST wrong tile (all its row chunks), then every frame's second, ... -- the
                // tickets most likely to find work are handed out first, while every workgroup is still free; frame by frame,
                // the last frames' tiles all fell into the second round behind workgroups that already had a tile to do
                const uint32_t j = it - n_flagged, per_share = n_flagged * nchd;
                const uint32_t share = j / per_share, rest = j % per_share;
                const uint32_t f = base + C->list[rest / nchd], rc = rest % nchd;
                // ONE look at the frame's ready word.  There (a later round of tickets): the maps and the mask of wrong tiles are
                // in global memory, nothing to compute.  Not there yet (the first round: the LEAD ticket started when this one
                // did): resolve the frame here -- 6 us for whole, aligned packets -- instead of sitting through the LEAD
                // ticket's resolution, its write-through of the maps and a poll interval (13 us, tools/ab/phase_fixup.py).
                // Nobody ever waits for anybody.
                if (tid == 0) C->ready = __hip_atomic_load(&ready[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                FSTAMP_MID();
                const unsigned long long v = C->ready;
                const bool published = (v >> 32) == (tag & 0xffffffffull);
                uint32_t mask = (uint32_t)v;
                if (!published) {
                    resolve(f, false);
                    mask = C->dirty;
                }
                uint32_t rank = 0, ntl = 0;
                for (uint32_t m = mask; m; m &= m - 1u, ++rank) {
                    if (rank % SPLIT != share) continue;
                    const uint32_t tile = (uint32_t)__builtin_ctz(m);
                    if (published) {
                        wide_tile<S, TW, XYZM, POSES, false>(a, smem, f, tile, rc, nullptr, nullptr, TRd, nchd, true);   // reads the maps with agent-scope loads
                    } else {
                        if (ntl) resolve(f, false);   // more than SPLIT_MAX column tiles: the maps lay under the tile before
                        wide_tile<S, TW, XYZM, POSES, true>(a, smem, f, tile, rc, L.pix, L.hdr, TRd, nchd, true);
                    }
                    __syncthreads();
                    ++ntl;
                }
                FSTAMP_END(2u, ntl);
            }
            if (!prefetch) pull_ahead();
            ticket = take_ahead();
        }
        done += items;
    }
}

}  // namespace ouster_hip_dev
