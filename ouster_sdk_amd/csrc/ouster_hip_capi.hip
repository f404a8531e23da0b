// ouster_hip_capi.hip -- host side of the C ABI declared in include/ouster_hip.h.
// Owns the context (stream, scratch), turns format / calibration descriptions into
// device tables and launches the kernels of ouster_hip_kernels.hip.  No compute happens
// on the host except the one-off XYZ table construction (make_xyz_lut is a one-off in the
// reference too: ouster_core/src/xyzlut.cpp:11-89, cached per sensor in sensor_info.cpp:260-275).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <new>
#include <string>
#include <unistd.h>
#include <vector>

#include "host_pool.h"
#include "ouster_hip_dev.h"

using namespace ouster_hip_dev;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(OUSTER_HIP_ERR_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        counted_free(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 2;
        if (counted_malloc(&p, want) != hipSuccess) {
            p = nullptr;
            return -1;
        }
        cap = want;
        return 0;
    }
    void release() {
        counted_free(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

// experiment / test knobs of a context: defaults from OUSTER_HIP_* environment variables read ONCE in
// ouster_hip_ctx_create, changed afterwards with ouster_hip_ctx_set_knob (never getenv on the call path)
struct Knobs {
    int tile = 0;             // OUSTER_HIP_TILE: force k_decode's tile width (64/32/16)
    int wide = -1;            // OUSTER_HIP_WIDE: -1 auto (tuner), 0 narrow, 64/128/256/512 force k_decode_wide
    int wide_kb = 64;         // OUSTER_HIP_WIDE_KB: LDS budget of a wide tile image
    int wide_rows = 0;        // OUSTER_HIP_WIDE_ROWS: force the rows of a wide tile (experiments)
    int wide_min_blocks = 512;  // OUSTER_HIP_WIDE_MIN_BLOCKS: smaller launches stay on k_decode
    int tune = 1;             // OUSTER_HIP_TUNE: 0 pins the default wide variant
    int xcd = 1;              // OUSTER_HIP_XCD: 0 disables the XCD-aware block -> frame mapping
    int fast = 1;             // OUSTER_HIP_FAST: 0 sends every frame through the general mapping
    int dewarp_single_pass = 0;  // OUSTER_HIP_DWF_SINGLE: 1 = k_dwf_single instead of count / scan / emit (slower)
    int beam_lds = 1;         // OUSTER_HIP_BEAM_LDS: 0 keeps k_decode's per-beam table in global memory (A/B)
    int fixup = 1;            // tests only: 0 skips the fix-up pass (flagged frames are then left undone)
    int small = 1;            // OUSTER_HIP_SMALL: small batches: 1 = wide tiles of few rows (optimistic pass + fix-up pass; any other buffer shape: the
                              //   one-launch k_decode_wide_resolved) | 2 = the one-launch form always | 0 = k_decode's narrow tiles as in r03
    int fixup_rows = 0;       // OUSTER_HIP_FIXUP_ROWS: rows of a fix-up tile (0: 8)
    int hdr_words = 1;        // OUSTER_HIP_HDR_WORDS: 0 = the fix-up pass reads the column headers from the packets again (A/B)
    int fixup_wide = 1;       // OUSTER_HIP_FIXUP_WIDE: 1 = the fix-up pass on wide tiles where the format allows | 0 = 64-column tiles | 64 / 128 / 256 force
    int stream = -1;          // OUSTER_HIP_STREAM: -1 auto | 0 never | 128 / 256 force k_decode_stream with that tile width when eligible
    int stream_rows = 0;      // OUSTER_HIP_STREAM_ROWS: force the rows of a streamed tile (experiments)
    int stream_wait = 1;      // OUSTER_HIP_STREAM_WAIT: 1 vmcnt(0) before a prefetched tile is used | 0 rely on the in-order counter
    int stream_min_tiles = 8; // OUSTER_HIP_STREAM_MIN_TILES: tiles per workgroup below which a launch stays on k_decode_wide
    int stream_order = 0;     // OUSTER_HIP_STREAM_ORDER: item order of k_decode_stream's groups (experiments)
    int slotmap = 1;          // OUSTER_HIP_SLOTMAP: 0 = buffers without one slot per column go through k_decode's general tiles
                              //   (every tile scans the frame's headers) instead of k_slotmap + k_decode_wide
    int dwf_stream = -1;      // OUSTER_HIP_DWF_STREAM: the frame dewarp's emit kernel: -1 auto | 0 k_dwf_emit | 1 the persistent k_dwf_emit_stream where eligible
    int stream_loader = 4;    // OUSTER_HIP_STREAM_LOADER: loader waves of k_decode_stream2 (0 = k_decode_stream: every wave fetches)
};

struct ouster_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevBuf state, tile_valid, offsets, luts, counts, scratch, slotmap, hdrw, osf_pixels;
    DevBuf user_scratch[8];              // ouster_hip_ctx_scratch: what the *_host calls and bindings stage through
    uint32_t resident_wgs = 512;         // 2 workgroups (80 KB LDS each) per CU
    uint32_t cus = 256;                  // compute units (k_decode_stream: one persistent workgroup each)
    const char* last_kernel = "";        // name of the decode kernel the last ouster_hip_decode launched
    bool state_dirty = false;            // `state` may hold leftovers (fix-up pass skipped / a failed launch)
    std::vector<int32_t> offsets_host;   // cache key of `offsets`
    std::vector<int32_t> shifts_host;    // pixel_shift_by_row the cached offsets were derived from
    uint32_t shifts_w = 0;               //   ... and the frame width (the offsets are (W + x % W) % W)
    std::vector<LutDev> luts_host;       // cache key of `luts`
    // pageable host packet_counts go through a small pinned ring (no stream synchronisation)
    static constexpr int RING = 4;
    uint32_t* ring_buf[RING] = {nullptr, nullptr, nullptr, nullptr};
    size_t ring_cap[RING] = {0, 0, 0, 0};
    hipEvent_t ring_ev[RING] = {nullptr, nullptr, nullptr, nullptr};
    int ring_next = 0;
    Knobs knobs;
    bool timing = false;
    uint32_t timing_every = 1, timing_calls = 0;   // HIP events around every timing_every-th decode kernel (an event record costs the stream 2 - 3 us)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    // k_decode variant (64-column tiles / wide tiles of 128 or 256 columns) per workload, picked by
    // timing each candidate on the first calls: which one is faster depends on how the output
    // planes happen to be placed in HBM
    struct Tune {
        int best = -2;  // -2: still measuring; 0: narrow; 128 / 256: wide; 1000 + tile width: the persistent kernel
        int calls = 0;
        bool from_cache = false;   // `best` was read from the tuning cache file, not timed by this context
        static constexpr int ROUNDS = 4, NCAND = 5, SAMPLES = NCAND * ROUNDS;
        hipEvent_t ev[SAMPLES][2] = {};  // up to five candidates x ROUNDS consecutive launches
        float ms[NCAND] = {0, 0, 0, 0, 0};  // fastest warm sample of each candidate
    };
    std::map<uint64_t, Tune> tune;
    // verdicts of earlier processes (ouster_hip_ctx_set_tuning_cache): workload key -> chosen variant, for THIS device and library
    std::string tune_cache_path, tune_cache_id;
    std::map<uint64_t, int> tune_cache;
    const char* last_tuner = "none";     // how the last ouster_hip_decode chose its variant
    int last_tile_cols = 0, last_tile_rows = 0;  // tile of the last k_decode launch
};

struct ouster_hip_format {
    ouster_hip_format_desc desc;
    Geometry g;
    int spec_id = SPEC_GENERIC;
    int8_t spec_of_desc[OUSTER_HIP_MAX_FIELDS];
    int spec_nf = 0, spec_r1 = -1, spec_r2 = -1;
};

struct ouster_hip_lut {
    int device = 0;  // the tables outlive the context that uploaded them
    uint32_t w = 0, h = 0;
    bool separable = false;
    LutDev dev{};
    void *d_beam = nullptr, *d_col = nullptr, *d_dir = nullptr, *d_ofs = nullptr;
    // host copy of the full double LUT for ouster_hip_lut_export
    std::vector<double> direction, offset;
};

// ---------------------------------------------------------------------------------------
// host math: impl::make_xyz_lut (ouster_core/src/xyzlut.cpp:11-89), full LUT in double
// ---------------------------------------------------------------------------------------
static void host_full_lut(const ouster_hip_calib& c, std::vector<double>& direction,
                          std::vector<double>& offset) {
    const size_t w = c.w, h = c.h;
    direction.assign(w * h * 3, 0.0);
    offset.assign(w * h * 3, 0.0);
    const double* b2l = c.beam_to_lidar_transform;
    const double* tf = c.transform;
    const double bx = b2l[3], bz = b2l[11];
    double n = bx;
    if (bz != 0) n = std::sqrt(std::pow(bx, 2) + std::pow(bz, 2));
    const bool per_beam = (c.n_angles == h);
    const double step = M_PI * 2.0 / static_cast<double>(w);
    for (size_t row = 0; row < h; ++row) {
        for (size_t col = 0; col < w; ++col) {
            const size_t i = row * w + col;
            double enc, azi, alt;
            if (per_beam) {
                enc = 2.0 * M_PI - static_cast<double>(col) * step;
                azi = -c.azimuth_angles_deg[row] * M_PI / 180.0;
                alt = c.altitude_angles_deg[row] * M_PI / 180.0;
            } else {
                enc = 0;
                azi = c.azimuth_angles_deg[i] * M_PI / 180.0;
                alt = c.altitude_angles_deg[i] * M_PI / 180.0;
            }
            const double d[3] = {std::cos(enc + azi) * std::cos(alt),
                                 std::sin(enc + azi) * std::cos(alt), std::sin(alt)};
            const double o[3] = {std::cos(enc) * bx - d[0] * n, std::sin(enc) * bx - d[1] * n,
                                 -d[2] * n + bz};
            for (int r = 0; r < 3; ++r) {
                double dd = 0, oo = 0;
                for (int k = 0; k < 3; ++k) {
                    dd += d[k] * tf[r * 4 + k];
                    oo += o[k] * tf[r * 4 + k];
                }
                oo += tf[r * 4 + 3];
                direction[i * 3 + r] = dd * c.range_unit;
                offset[i * 3 + r] = oo * c.range_unit;
            }
        }
    }
}

// separable form of the same table: see the derivation in DESIGN.md ("XYZ tables")
static void host_separable_tables(const ouster_hip_calib& c, std::vector<double>& beam,
                                  std::vector<double>& col, double& n_out) {
    const size_t w = c.w, h = c.h;
    const double* b2l = c.beam_to_lidar_transform;
    const double* tf = c.transform;
    const double ru = c.range_unit;
    const double bx = b2l[3], bz = b2l[11];
    double n = bx;
    if (bz != 0) n = std::sqrt(std::pow(bx, 2) + std::pow(bz, 2));
    n_out = n;
    auto rot = [&](const double v[3], double out[3]) {
        for (int r = 0; r < 3; ++r)
            out[r] = v[0] * tf[r * 4 + 0] + v[1] * tf[r * 4 + 1] + v[2] * tf[r * 4 + 2];
    };
    beam.assign(h * 9, 0.0);
    for (size_t row = 0; row < h; ++row) {
        const double azi = -c.azimuth_angles_deg[row] * M_PI / 180.0;
        const double alt = c.altitude_angles_deg[row] * M_PI / 180.0;
        const double A = std::cos(azi) * std::cos(alt), B = std::sin(azi) * std::cos(alt),
                     Cc = std::sin(alt);
        const double u[3] = {A, B, 0}, v[3] = {-B, A, 0}, wv[3] = {0, 0, Cc};
        double ru_[3];
        rot(u, ru_);
        for (int k = 0; k < 3; ++k) beam[row * 9 + k] = ru_[k] * ru;
        rot(v, ru_);
        for (int k = 0; k < 3; ++k) beam[row * 9 + 3 + k] = ru_[k] * ru;
        rot(wv, ru_);
        for (int k = 0; k < 3; ++k) beam[row * 9 + 6 + k] = ru_[k] * ru;
    }
    col.assign(w * 5, 0.0);
    const double step = M_PI * 2.0 / static_cast<double>(w);
    for (size_t cc = 0; cc < w; ++cc) {
        const double enc = 2.0 * M_PI - static_cast<double>(cc) * step;
        const double cx = std::cos(enc), sx = std::sin(enc);
        const double o[3] = {cx * bx, sx * bx, bz};
        double ro[3];
        rot(o, ro);
        col[cc * 5 + 0] = cx;
        col[cc * 5 + 1] = sx;
        for (int k = 0; k < 3; ++k) col[cc * 5 + 2 + k] = (ro[k] + tf[k * 4 + 3]) * ru;
    }
}

static void fill_geometry(const ouster_hip_format_desc& d, Geometry& g) {
    g.pixels_per_column = d.pixels_per_column;
    g.columns_per_packet = d.columns_per_packet;
    g.columns_per_frame = d.columns_per_frame;
    g.packet_header_size = d.packet_header_size;
    g.col_header_size = d.col_header_size;
    g.channel_data_size = d.channel_data_size;
    g.col_footer_size = d.col_footer_size;
    g.packet_footer_size = d.packet_footer_size;
    g.col_size = d.col_size;
    g.lidar_packet_size = d.lidar_packet_size;
    g.col_timestamp = d.col_timestamp;
    g.col_measurement_id = d.col_measurement_id;
    g.col_status = d.col_status;
    g.frame_id = d.frame_id;
    g.alert_flags = d.alert_flags;
    g.thermal_shutdown = d.thermal_shutdown;
    g.shot_limiting = d.shot_limiting;
    g.countdown_thermal_shutdown = d.countdown_thermal_shutdown;
    g.countdown_shot_limiting = d.countdown_shot_limiting;
}

extern "C" {

const char* ouster_hip_last_error(void) { return g_err.c_str(); }
#ifdef OUSTER_EXPERIMENTS
const char* ouster_hip_version(void) { return "ouster_hip 0.3 (gfx950) +experiments"; }
#else
const char* ouster_hip_version(void) { return "ouster_hip 0.3 (gfx950)"; }
#endif

int ouster_hip_ctx_create(int device, void* stream, ouster_hip_ctx** out) {
    if (!out) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(OUSTER_HIP_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    ouster_hip_ctx* c = new (std::nothrow) ouster_hip_ctx();
    if (!c) return fail(OUSTER_HIP_ERR_RUNTIME, "out of memory");
    c->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            c->resident_wgs = 2u * (uint32_t)cus, c->cus = (uint32_t)cus;
    }
    {   // the only place the environment is read
        auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        Knobs& k = c->knobs;
        k.tile = env_int("OUSTER_HIP_TILE", k.tile);
        k.wide = env_int("OUSTER_HIP_WIDE", k.wide);
        k.wide_kb = env_int("OUSTER_HIP_WIDE_KB", k.wide_kb);
        k.wide_rows = env_int("OUSTER_HIP_WIDE_ROWS", k.wide_rows);
        k.wide_min_blocks = env_int("OUSTER_HIP_WIDE_MIN_BLOCKS", k.wide_min_blocks);
        k.tune = env_int("OUSTER_HIP_TUNE", k.tune);
        k.xcd = env_int("OUSTER_HIP_XCD", k.xcd);
        k.fast = env_int("OUSTER_HIP_FAST", k.fast);
        k.beam_lds = env_int("OUSTER_HIP_BEAM_LDS", k.beam_lds);
        k.dewarp_single_pass = env_int("OUSTER_HIP_DWF_SINGLE", k.dewarp_single_pass);
        k.stream = env_int("OUSTER_HIP_STREAM", k.stream);
        k.stream_rows = env_int("OUSTER_HIP_STREAM_ROWS", k.stream_rows);
        k.stream_wait = env_int("OUSTER_HIP_STREAM_WAIT", k.stream_wait);
        k.stream_min_tiles = env_int("OUSTER_HIP_STREAM_MIN_TILES", k.stream_min_tiles);
        k.stream_order = env_int("OUSTER_HIP_STREAM_ORDER", k.stream_order);
        k.stream_loader = env_int("OUSTER_HIP_STREAM_LOADER", k.stream_loader);
        k.dwf_stream = env_int("OUSTER_HIP_DWF_STREAM", k.dwf_stream);
        k.slotmap = env_int("OUSTER_HIP_SLOTMAP", k.slotmap);
        k.fixup_wide = env_int("OUSTER_HIP_FIXUP_WIDE", k.fixup_wide);
        k.small = env_int("OUSTER_HIP_SMALL", k.small);
        k.hdr_words = env_int("OUSTER_HIP_HDR_WORDS", k.hdr_words);
        k.fixup_rows = env_int("OUSTER_HIP_FIXUP_ROWS", k.fixup_rows);
    }
    if (const char* e = getenv("OUSTER_HIP_TUNING_CACHE")) (void)ouster_hip_ctx_set_tuning_cache(c, e);
    if (stream == OUSTER_HIP_STREAM_NULL) {
        c->stream = nullptr;  // the null stream
    } else if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(OUSTER_HIP_ERR_RUNTIME, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    *out = c;
    return OUSTER_HIP_OK;
}

void ouster_hip_ctx_destroy(ouster_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->state.release();
    c->tile_valid.release();
    for (int i = 0; i < ouster_hip_ctx::RING; ++i) {
        if (c->ring_buf[i]) (void)hipHostFree(c->ring_buf[i]);
        if (c->ring_ev[i]) (void)hipEventDestroy(c->ring_ev[i]);
    }
    c->offsets.release();
    c->luts.release();
    c->counts.release();
    c->scratch.release();
    c->slotmap.release();
    c->hdrw.release();
    c->osf_pixels.release();
    for (auto& b : c->user_scratch) b.release();
    for (auto& p : c->ev_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto& kv : c->tune)
        for (auto& pr : kv.second.ev)
            for (hipEvent_t e : pr)
                if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* ouster_hip_ctx_stream(ouster_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int ouster_hip_ctx_device(ouster_hip_ctx* c) { return c ? c->device : -1; }

int ouster_hip_ctx_set_knob(ouster_hip_ctx* c, const char* name, int value) {
    if (!c || !name) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    const std::string n = name;
    Knobs& k = c->knobs;
    if (n == "tile") k.tile = value;
    else if (n == "wide") k.wide = value;
    else if (n == "wide_kb") k.wide_kb = value;
    else if (n == "wide_rows") k.wide_rows = value;
    else if (n == "wide_min_blocks") k.wide_min_blocks = value;
    else if (n == "tune") k.tune = value;
    else if (n == "retune") {   // forget what the variant tuner learnt (the caller moved its buffers: pick_placement)
        if (value) {
            (void)hipStreamSynchronize(c->stream);
            for (auto& kv : c->tune)
                for (auto& pr : kv.second.ev)
                    for (hipEvent_t e : pr)
                        if (e) (void)hipEventDestroy(e);
            c->tune.clear();
            c->tune_cache.clear();   // what this context read from the cache file is void as well: it re-measures and appends anew
        }
    }
    else if (n == "xcd") k.xcd = value;
    else if (n == "fast") k.fast = value;
    else if (n == "fixup") k.fixup = value;
    else if (n == "fixup_wide") k.fixup_wide = value;
    else if (n == "small") k.small = value;
    else if (n == "hdr_words") k.hdr_words = value;
    else if (n == "fixup_rows") k.fixup_rows = value;
    else if (n == "beam_lds") k.beam_lds = value;
    else if (n == "dewarp_single_pass") k.dewarp_single_pass = value;
    else if (n == "dwf_stream") k.dwf_stream = value;
    else if (n == "stream") k.stream = value;
    else if (n == "stream_rows") k.stream_rows = value;
    else if (n == "stream_wait") k.stream_wait = value;
    else if (n == "stream_min_tiles") k.stream_min_tiles = value;
    else if (n == "stream_order") k.stream_order = value;
    else if (n == "stream_loader") k.stream_loader = value;
    else if (n == "slotmap") k.slotmap = value;
    else return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "unknown knob '%s'", name);
    return OUSTER_HIP_OK;
}

// ---- persisted verdicts of the variant tuner -----------------------------------------------------------------------------
// One text line per verdict: "v1 <device and library id> <workload key, hex> <variant> <its fastest sample, ms>".  Lines are
// appended with one write() each (O_APPEND: ranks of one job may share the file), the last line of a key wins at load.
int ouster_hip_ctx_set_tuning_cache(ouster_hip_ctx* c, const char* path) {
    if (!c) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    c->tune_cache.clear();
    c->tune_cache_path = path ? path : "";
    if (c->tune_cache_path.empty()) return OUSTER_HIP_OK;
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) {
        (void)hipGetLastError();
        c->tune_cache_path.clear();
        return fail(OUSTER_HIP_ERR_RUNTIME, "hipGetDeviceProperties failed");
    }
    std::string id = std::string(prop.gcnArchName) + ":" + std::to_string(prop.multiProcessorCount) + ":" + prop.name + ":" + ouster_hip_version();
    for (char& ch : id) if (ch == ' ' || ch == '\t') ch = '_';
    c->tune_cache_id = id;
    if (FILE* f = fopen(c->tune_cache_path.c_str(), "r")) {
        char line[512], ver[8], dev[320];
        unsigned long long key = 0;
        int best = 0;
        float ms = 0;
        while (fgets(line, sizeof line, f))
            if (sscanf(line, "%7s %319s %llx %d %f", ver, dev, &key, &best, &ms) == 5 && !strcmp(ver, "v1") && id == dev)
                c->tune_cache[(uint64_t)key] = best;
        fclose(f);
    }
    return OUSTER_HIP_OK;
}
static void tune_cache_append(ouster_hip_ctx* c, uint64_t key, int best, float ms) {
    if (c->tune_cache_path.empty()) return;
    c->tune_cache[key] = best;
    char line[512];
    const int n = snprintf(line, sizeof line, "v1 %s %016llx %d %.4f\n", c->tune_cache_id.c_str(), (unsigned long long)key, best, ms);
    if (n <= 0 || n >= (int)sizeof line) return;
    const int fd = open(c->tune_cache_path.c_str(), O_WRONLY | O_APPEND | O_CREAT, 0644);
    if (fd < 0) return;   // a cache that cannot be written is not an error of the decode
    (void)!write(fd, line, (size_t)n);
    close(fd);
}
const char* ouster_hip_last_decode_tuner(ouster_hip_ctx* c) { return c ? c->last_tuner : ""; }

int ouster_hip_sync(ouster_hip_ctx* c) {
    if (!c) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OUSTER_HIP_OK;
}

// ---- format ----------------------------------------------------------------------------
int ouster_hip_format_create(ouster_hip_ctx* ctx, const ouster_hip_format_desc* d,
                             ouster_hip_format** out) {
    if (!ctx || !d || !out) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->pixels_per_column == 0)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "unexpected pixels_per_column: 0");
    if (d->columns_per_packet == 0)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "unexpected columns_per_packet: 0");
    if (d->columns_per_frame == 0)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "unexpected columns_per_frame: 0");
    if (d->n_fields > OUSTER_HIP_MAX_FIELDS)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "too many fields (%u)", d->n_fields);
    if (d->col_size != d->col_header_size + d->pixels_per_column * d->channel_data_size +
                           d->col_footer_size ||
        d->lidar_packet_size != d->packet_header_size + d->columns_per_packet * d->col_size +
                                    d->packet_footer_size)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "inconsistent packet geometry");
    if (d->lidar_packet_size > 65535)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "lidar_packet_size cannot exceed 65535");
    if ((d->col_size & 3) || (d->packet_header_size & 3) || (d->lidar_packet_size & 3) ||
        (d->col_header_size & 3))
        return fail(OUSTER_HIP_ERR_UNSUPPORTED, "packet geometry must be 4-byte granular");
    for (uint32_t i = 0; i < d->n_fields; ++i) {
        const uint32_t e = d->fields[i].dst_elem_size;
        if (!(e == 1 || e == 2 || e == 4 || e == 6 || e == 8))
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "field %u: unsupported element size %u", i, e);
    }
    ouster_hip_format* f = new (std::nothrow) ouster_hip_format();
    if (!f) return fail(OUSTER_HIP_ERR_RUNTIME, "out of memory");
    f->desc = *d;
    fill_geometry(*d, f->g);
    // match against the compile-time specialisations
    f->spec_id = SPEC_GENERIC;
    for (int sid = SPEC_DUAL_LB; sid <= SPEC_LEGACY && f->spec_id == SPEC_GENERIC; ++sid) {
        int nf, r1, r2;
        uint32_t chan;
        const FieldC* sf = spec_fields(sid, &nf, &chan, &r1, &r2);
        if (chan != d->channel_data_size || (d->col_header_size & 3) || d->n_fields == 0) continue;
        bool ok = true;
        int8_t map[OUSTER_HIP_MAX_FIELDS];
        uint32_t used = 0;
        for (uint32_t i = 0; i < d->n_fields && ok; ++i) {
            const auto& fd = d->fields[i];
            int hit = -1;
            for (int k = 0; k < nf; ++k)
                if (!(used & (1u << k)) && sf[k].offset == fd.bits.offset &&
                    sf[k].mask == fd.bits.mask && sf[k].shift == fd.bits.shift &&
                    sf[k].elem == fd.dst_elem_size && !fd.f16_nan_fill) {
                    hit = k;
                    break;
                }
            if (hit < 0) ok = false;
            else { used |= 1u << hit; map[i] = (int8_t)hit; }
        }
        if (ok) {
            f->spec_id = sid;
            memcpy(f->spec_of_desc, map, sizeof map);
            f->spec_nf = nf;
            f->spec_r1 = r1;
            f->spec_r2 = r2;
        }
    }
    *out = f;
    return OUSTER_HIP_OK;
}

void ouster_hip_format_destroy(ouster_hip_format* f) { delete f; }

// ---- lut -------------------------------------------------------------------------------
static int upload(void** dptr, const void* src, size_t bytes, hipStream_t st) {
    if (counted_malloc(dptr, bytes) != hipSuccess) return -1;
    if (hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

int ouster_hip_lut_create(ouster_hip_ctx* ctx, const ouster_hip_calib* c, ouster_hip_lut** out) {
    if (!ctx || !c || !out) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (c->w == 0 || c->h == 0)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "lut dimensions must be greater than zero");
    const size_t hw = (size_t)c->w * c->h;
    if ((c->n_angles != c->h && c->n_angles != hw) || !c->azimuth_angles_deg ||
        !c->altitude_angles_deg)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "unexpected frame dimensions");
    HIP_TRY(hipSetDevice(ctx->device));
    ouster_hip_lut* L = new (std::nothrow) ouster_hip_lut();
    if (!L) return fail(OUSTER_HIP_ERR_RUNTIME, "out of memory");
    L->device = ctx->device;
    L->w = c->w;
    L->h = c->h;
    host_full_lut(*c, L->direction, L->offset);
    int rc = 0;
    if (c->n_angles == c->h) {
        std::vector<double> beam, col;
        double n;
        host_separable_tables(*c, beam, col, n);
        L->separable = true;
        rc |= upload(&L->d_beam, beam.data(), beam.size() * 8, ctx->stream);
        rc |= upload(&L->d_col, col.data(), col.size() * 8, ctx->stream);
        L->dev.beam_tab = (const double*)L->d_beam;
        L->dev.col_tab = (const double*)L->d_col;
        L->dev.n = n;
        L->dev.full_dtype = 0;
    } else {
        rc |= upload(&L->d_dir, L->direction.data(), hw * 24, ctx->stream);
        rc |= upload(&L->d_ofs, L->offset.data(), hw * 24, ctx->stream);
        L->dev.full_dir = L->d_dir;
        L->dev.full_ofs = L->d_ofs;
        L->dev.full_dtype = OUSTER_HIP_F64;
    }
    if (rc) {
        ouster_hip_lut_destroy(L);
        return fail(OUSTER_HIP_ERR_RUNTIME, "LUT upload failed");
    }
    *out = L;
    return OUSTER_HIP_OK;
}

int ouster_hip_lut_create_from_arrays(ouster_hip_ctx* ctx, const void* direction,
                                      const void* offset, uint32_t h, uint32_t w, int dtype,
                                      ouster_hip_lut** out) {
    if (!ctx || !direction || !offset || !out)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (w == 0 || h == 0)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "lut dimensions must be greater than zero");
    if (dtype != OUSTER_HIP_F32 && dtype != OUSTER_HIP_F64)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "LUT dtype must be F32 or F64");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t hw = (size_t)w * h, es = dtype == OUSTER_HIP_F32 ? 4 : 8;
    ouster_hip_lut* L = new (std::nothrow) ouster_hip_lut();
    if (!L) return fail(OUSTER_HIP_ERR_RUNTIME, "out of memory");
    L->device = ctx->device;
    L->w = w;
    L->h = h;
    L->direction.resize(hw * 3);
    L->offset.resize(hw * 3);
    for (size_t i = 0; i < hw * 3; ++i) {
        L->direction[i] = es == 4 ? (double)((const float*)direction)[i] : ((const double*)direction)[i];
        L->offset[i] = es == 4 ? (double)((const float*)offset)[i] : ((const double*)offset)[i];
    }
    int rc = upload(&L->d_dir, direction, hw * 3 * es, ctx->stream);
    rc |= upload(&L->d_ofs, offset, hw * 3 * es, ctx->stream);
    if (rc) {
        ouster_hip_lut_destroy(L);
        return fail(OUSTER_HIP_ERR_RUNTIME, "LUT upload failed");
    }
    L->dev.full_dir = L->d_dir;
    L->dev.full_ofs = L->d_ofs;
    L->dev.full_dtype = dtype;
    *out = L;
    return OUSTER_HIP_OK;
}

int ouster_hip_lut_export(const ouster_hip_lut* L, double* direction, double* offset) {
    if (!L || !direction || !offset) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    memcpy(direction, L->direction.data(), L->direction.size() * 8);
    memcpy(offset, L->offset.data(), L->offset.size() * 8);
    return OUSTER_HIP_OK;
}

void ouster_hip_lut_destroy(ouster_hip_lut* L) {
    if (!L) return;
    (void)hipSetDevice(L->device);
    counted_free((void*)L->d_beam);
    counted_free((void*)L->d_col);
    counted_free((void*)L->d_dir);
    counted_free((void*)L->d_ofs);
    delete L;
}

// ---- helpers for per-call device tables -------------------------------------------------
// destination column offset per row with the reference's exact arithmetic:
//   offset = (w + sign*shift % w) % w   with w a size_t, i.e. `sign*shift` is converted to
//   an unsigned 64-bit value BEFORE the modulo (impl/lidar_frame_impl.h:737-756).
static void dest_offsets(const int32_t* shifts, uint32_t h, uint32_t w, int inverse,
                         std::vector<int32_t>& out) {
    out.resize(h);
    const int sign = inverse ? -1 : +1;
    const size_t ws = w;
    for (uint32_t u = 0; u < h; ++u) {
        const size_t x = (size_t)(long long)(sign * shifts[u]);
        out[u] = (int32_t)((ws + x % ws) % ws);
    }
}

static int ensure_offsets(ouster_hip_ctx* c, const std::vector<int32_t>& off) {
    if (off == c->offsets_host && c->offsets.p) return 0;
    if (c->offsets.ensure(off.size() * 4)) return -1;
    if (hipMemcpyAsync(c->offsets.p, off.data(), off.size() * 4, hipMemcpyHostToDevice,
                       c->stream) != hipSuccess)
        return -1;
    // the source vector must outlive the (possibly staged) copy
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    c->offsets_host = off;
    return 0;
}

static int ensure_luts(ouster_hip_ctx* c, const std::vector<LutDev>& l) {
    bool same = c->luts.p && l.size() == c->luts_host.size() &&
                memcmp(l.data(), c->luts_host.data(), l.size() * sizeof(LutDev)) == 0;
    if (same) return 0;
    if (c->luts.ensure(l.size() * sizeof(LutDev))) return -1;
    if (hipMemcpyAsync(c->luts.p, l.data(), l.size() * sizeof(LutDev), hipMemcpyHostToDevice,
                       c->stream) != hipSuccess)
        return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    c->luts_host = l;
    return 0;
}

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- decode ------------------------------------------------------------------------------
// packet_counts may live anywhere: device memory is used in place; host memory is copied with the
// stream (pinned: directly; pageable: through the context's pinned ring) -- never a stream sync
static int stage_counts(ouster_hip_ctx* ctx, const uint32_t* packet_counts, uint32_t n_frames,
                        uint32_t slots_per_frame, const uint32_t** d_counts) {
    hipPointerAttribute_t attr{};
    const hipError_t pe = hipPointerGetAttributes(&attr, packet_counts);
    if (pe != hipSuccess) (void)hipGetLastError();  // an unregistered host pointer is not an error
    if (pe == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)) {
        *d_counts = packet_counts;  // values above slots_per_frame are clamped by the kernels
        return OUSTER_HIP_OK;
    }
    for (uint32_t f = 0; f < n_frames; ++f)
        if (packet_counts[f] > slots_per_frame)
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "packet_counts[%u] > slots_per_frame", f);
    const size_t bytes = (size_t)n_frames * 4;
    if (ctx->counts.ensure(bytes)) return fail(OUSTER_HIP_ERR_RUNTIME, "hipMalloc(counts) failed");
    const void* src = packet_counts;
    const bool pinned = pe == hipSuccess && attr.type == hipMemoryTypeHost;
    int slot = -1;
    if (!pinned) {
        slot = ctx->ring_next;
        ctx->ring_next = (slot + 1) % ouster_hip_ctx::RING;
        if (!ctx->ring_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&ctx->ring_ev[slot], hipEventDisableTiming));
        else HIP_TRY(hipEventSynchronize(ctx->ring_ev[slot]));  // RING calls ago: long done
        if (ctx->ring_cap[slot] < bytes) {
            if (ctx->ring_buf[slot]) (void)hipHostFree(ctx->ring_buf[slot]);
            ctx->ring_buf[slot] = nullptr;
            ctx->ring_cap[slot] = 0;
            HIP_TRY(hipHostMalloc((void**)&ctx->ring_buf[slot], bytes + bytes / 2, hipHostMallocDefault));
            ctx->ring_cap[slot] = bytes + bytes / 2;
        }
        memcpy(ctx->ring_buf[slot], packet_counts, bytes);
        src = ctx->ring_buf[slot];
    }
    HIP_TRY(hipMemcpyAsync(ctx->counts.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ring_ev[slot], ctx->stream));
    *d_counts = (const uint32_t*)ctx->counts.p;
    return OUSTER_HIP_OK;
}

static bool fast_possible(const Knobs& kn, uint32_t slots_per_frame, const Geometry& g) {
    return kn.fast && (uint64_t)slots_per_frame * g.columns_per_packet == g.columns_per_frame;
}

int ouster_hip_decode(ouster_hip_ctx* ctx, const ouster_hip_format* fmt, const uint8_t* packets,
                      size_t packet_stride, uint32_t slots_per_frame,
                      const uint32_t* packet_counts, uint32_t n_frames,
                      const uint64_t* host_timestamps, const ouster_hip_frame_out* out,
                      const int32_t* pixel_shift_by_row, const ouster_hip_lut* const* luts,
                      uint32_t n_luts) {
    if (!ctx || !fmt || !out) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_frames == 0) return OUSTER_HIP_OK;
    if (!packets) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "packets is NULL");
    const Geometry& g = fmt->g;
    const Knobs& kn = ctx->knobs;
    const uint32_t W = g.columns_per_frame, H = g.pixels_per_column;
    if (packet_stride < g.lidar_packet_size || (packet_stride & 3) || ((uintptr_t)packets & 3))
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT,
                    "packet_stride must be >= lidar_packet_size and 4-byte granular");
    if ((uint64_t)slots_per_frame * g.columns_per_packet > 0x7fffffffull || slots_per_frame == 0)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "bad slots_per_frame");
    if ((uint64_t)slots_per_frame * packet_stride > 0xffffffffull)
        return fail(OUSTER_HIP_ERR_UNSUPPORTED, "a frame's packet buffer cannot exceed 4 GiB");
    if (W > 65536 || W / g.columns_per_packet > 8192)
        return fail(OUSTER_HIP_ERR_UNSUPPORTED, "more than 65536 columns or 8192 packets per frame");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    const uint32_t nf = fmt->desc.n_fields;
    bool any_dst = false, any_xyz = (out->xyz[0] || out->xyz[1]);
    for (uint32_t i = 0; i < nf; ++i) any_dst |= out->destaggered[i] != nullptr;
    if (any_dst && !pixel_shift_by_row)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "image height does not match shifts size");
    if (out->gate_counts) {
        const int gf = out->gate_field;
        if (gf < 0 || (uint32_t)gf >= nf || fmt->desc.fields[gf].dst_elem_size != 4)
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "gate_field must name a 32-bit range field");
        if (H > 65535) return fail(OUSTER_HIP_ERR_UNSUPPORTED, "gate counts are 16-bit: more than 65535 rows");
    }
    if (any_xyz) {
        if (!luts || n_luts == 0) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "xyz output needs a LUT");
        if (out->xyz_dtype != OUSTER_HIP_F32 && out->xyz_dtype != OUSTER_HIP_F64)
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "xyz_dtype must be F32 or F64");
        for (int k = 0; k < 2; ++k) {
            if (!out->xyz[k]) continue;
            const int xf = out->xyz_field[k];
            if (xf < 0 || (uint32_t)xf >= nf || fmt->desc.fields[xf].dst_elem_size != 4)
                return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT,
                            "xyz_field[%d] must name a 32-bit range field", k);
        }
        for (uint32_t i = 0; i < n_luts; ++i)
            if (!luts[i] || luts[i]->w != W || luts[i]->h != H)
                return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "unexpected image dimensions");
            else if (luts[i]->device != ctx->device)
                return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "LUT %u lives on GPU %d, the context on GPU %d", i,
                            luts[i]->device, ctx->device);
    }

    // ---- scratch: per-frame state words (all-zero between calls: the fix-up pass cleans up after
    // itself), destagger offsets, LUT descriptors, packet counts
    {
        const void* before = ctx->state.p;
        if (ctx->state.ensure(((size_t)2 * n_frames + FS_WORDS) * 8)) return fail(OUSTER_HIP_ERR_RUNTIME, "hipMalloc(state) failed");
        if (ctx->state.p != before || ctx->state_dirty) HIP_TRY(hipMemsetAsync(ctx->state.p, 0, ctx->state.cap, st));
        ctx->state_dirty = fast_possible(kn, slots_per_frame, g);  // until the fix-up pass has been queued
    }
    if (out->frame_meta && ctx->tile_valid.ensure((size_t)n_frames * ((W + 15) / 16) * 2))
        return fail(OUSTER_HIP_ERR_RUNTIME, "hipMalloc(tile_valid) failed");
    const uint32_t n_packets_out = W / g.columns_per_packet;
    const uint32_t* d_counts = nullptr;
    if (packet_counts) {
        const int rc = stage_counts(ctx, packet_counts, n_frames, slots_per_frame, &d_counts);
        if (rc) return rc;
    }
    if (any_dst) {
        // the offsets only change with the sensor: key the cache on the shifts themselves
        if (!(ctx->offsets.p && ctx->shifts_host.size() == H && ctx->shifts_w == W &&
              memcmp(ctx->shifts_host.data(), pixel_shift_by_row, (size_t)H * 4) == 0 &&
              ctx->offsets_host.size() == H)) {
            std::vector<int32_t> off;
            dest_offsets(pixel_shift_by_row, H, W, 0, off);
            if (ensure_offsets(ctx, off)) return fail(OUSTER_HIP_ERR_RUNTIME, "offset upload failed");
            ctx->shifts_host.assign(pixel_shift_by_row, pixel_shift_by_row + H);
            ctx->shifts_w = W;
        }
    }
    int xyzm = 0;
    if (any_xyz) {
        bool all_sep = true, none_sep = true;
        bool same = ctx->luts.p && ctx->luts_host.size() == n_luts;
        for (uint32_t i = 0; i < n_luts; ++i) {
            all_sep &= luts[i]->separable;
            none_sep &= !luts[i]->separable;
            same = same && memcmp(&luts[i]->dev, &ctx->luts_host[i], sizeof(LutDev)) == 0;
        }
        if (!all_sep && !none_sep)
            return fail(OUSTER_HIP_ERR_UNSUPPORTED, "cannot mix separable and full LUTs in one batch");
        if (!same) {
            std::vector<LutDev> l(n_luts);
            for (uint32_t i = 0; i < n_luts; ++i) l[i] = luts[i]->dev;
            if (ensure_luts(ctx, l)) return fail(OUSTER_HIP_ERR_RUNTIME, "LUT descriptor upload failed");
        }
        xyzm = all_sep ? (out->xyz_dtype == OUSTER_HIP_F32 ? 1 : 2) : 3;
    }

    // ---- kernel arguments
    DecodeArgs da{};
    da.g = g;
    da.packets = packets;
    da.packet_stride = packet_stride;
    da.slots_per_frame = slots_per_frame;
    da.n_frames = n_frames;
    da.n_packets_out = n_packets_out;
    da.packet_counts = d_counts;
    da.host_timestamps = host_timestamps;
    da.frame_state = (uint64_t*)ctx->state.p;
    da.ready_off = FS_WORDS + (uint32_t)((ctx->state.cap / 8 - FS_WORDS) / 2);   // the second half of the buffer: moves only when it is reallocated (and zeroed)
    if (out->xyz_poses && xyzm == 3)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "xyz_poses need LUTs with separable tables (ouster_hip_lut_create)");
    da.xyz_poses = xyzm ? out->xyz_poses : nullptr;
    const size_t pose_per_col = da.xyz_poses ? (size_t)12 * (xyzm == 1 ? 4 : 8) : 0;   // LDS bytes of a column's pose
    da.gate_counts = out->gate_counts;
    da.gate_min = out->gate_min_r;
    da.gate_max = out->gate_max_r;
    da.gate_field = out->gate_counts ? out->gate_field : -1;
    da.tile_valid = (uint16_t*)ctx->tile_valid.p;
#ifdef OUSTER_PHASE_TIMING
    {   // experiment builds only: the tool passes a device buffer's address through the environment
        const char* e = getenv("OUSTER_HIP_PHASE_BUF");
        da.phase_times = e ? (uint64_t*)strtoull(e, nullptr, 0) : nullptr;
    }
    uint64_t* const phase_times_all = da.phase_times;
    if (getenv("OUSTER_HIP_PHASE_FIXUP_ONLY")) da.phase_times = nullptr;   // only the fix-up pass stamps
#endif
    da.dst_offsets = (const int32_t*)ctx->offsets.p;
    da.luts = (const LutDev*)ctx->luts.p;
    da.n_luts = n_luts ? n_luts : 1;
    da.n_fields = nf;
    da.any_destagger = any_dst;
    da.timestamp = out->timestamp;
    da.measurement_id = out->measurement_id;
    da.status = out->status;
    da.packet_timestamp = out->packet_timestamp;
    da.alert_flags = out->alert_flags;
    da.frame_meta = out->frame_meta;
    da.xyz[0] = out->xyz[0];
    da.xyz[1] = out->xyz[1];
    da.xyz_field[0] = out->xyz[0] ? out->xyz_field[0] : -1;
    da.xyz_field[1] = out->xyz[1] ? out->xyz_field[1] : -1;
    da.xyz_dtype = out->xyz_dtype;
    bool vec_ok = (W % 4 == 0);
    const size_t npx = (size_t)H * W;
    for (uint32_t i = 0; i < nf; ++i) {
        da.planes[i] = out->planes[i];
        da.destaggered[i] = out->destaggered[i];
        da.bits[i] = fmt->desc.fields[i].bits;
        da.elem[i] = (uint8_t)fmt->desc.fields[i].dst_elem_size;
        da.f16_nan_mask |= fmt->desc.fields[i].f16_nan_fill ? (1u << i) : 0u;
        const size_t fstride = npx * da.elem[i];
        if (out->planes[i]) vec_ok &= al16(out->planes[i]) && (fstride % 16 == 0);
        // destaggered stores are unaligned-capable; only 4-byte stores of sub-dword
        // elements need the element alignment the caller already guarantees
    }
    const size_t xes = out->xyz_dtype == OUSTER_HIP_F64 ? 8 : 4;
    for (int k = 0; k < 2; ++k)
        if (out->xyz[k]) vec_ok &= al16(out->xyz[k]) && ((npx * 3 * xes) % 16 == 0);
    da.vec_ok = vec_ok;

    int spec = fmt->spec_id;
    if (spec != SPEC_GENERIC) {
        for (int k = 0; k < 16; ++k) da.desc_of_spec[k] = -1;
        for (uint32_t i = 0; i < nf; ++i) da.desc_of_spec[fmt->spec_of_desc[i]] = (int8_t)i;
        // the compile-time kernels take the xyz ranges from their RANGE / RANGE2 slots
        if (out->xyz[0] && fmt->spec_of_desc[out->xyz_field[0]] != fmt->spec_r1) spec = SPEC_GENERIC;
        if (out->xyz[1] && fmt->spec_of_desc[out->xyz_field[1]] != fmt->spec_r2) spec = SPEC_GENERIC;
    }

    // optimistic pass + fix-up pass when the buffer has one slot per column of the frame; everything
    // through the general mapping otherwise
    const bool fast = fast_possible(kn, slots_per_frame, g);

    // tile width: widest tile that still lets two workgroups share a CU's 160 KiB LDS
    int tile = 0;
    for (int t : {64, 32, 16})
        if (decode_lds_bytes(g, t, true, false, slots_per_frame) + 16 + t * pose_per_col <= 80 * 1024) { tile = t; break; }
    if (!tile)
        for (int t : {64, 32, 16})
            if (decode_lds_bytes(g, t, true, false, slots_per_frame) + 2048 + t * pose_per_col <= 160 * 1024) { tile = t; break; }
    if (!tile)   // the general mapping keeps resolve_frame's scratch ((2 + cpp) words per buffer slot) in LDS next to the tile
        return fail(OUSTER_HIP_ERR_UNSUPPORTED,
                    fast ? "column of %u bytes does not fit in LDS"
                         : "column of %u bytes + the column maps of %u buffer slots per frame do not fit in LDS (general mapping: at most "
                           "about 2000 slots of 16 columns; hand the frame over in fewer slots or in home slots)",
                    g.col_size, slots_per_frame);
    // small batches: prefer narrower tiles so that at least ~2 workgroups per CU exist
    // (one 128x2048 frame is only 32 tiles of 64 columns -- latency, not bandwidth, bound)
    while (tile > 16 && (size_t)n_frames * ((W + tile - 1) / tile) < 512) tile /= 2;
    if ((kn.tile == 64 || kn.tile == 32 || kn.tile == 16) &&
        decode_lds_bytes(g, kn.tile, true, false, slots_per_frame) + 2048 + kn.tile * pose_per_col <= 160 * 1024)
        tile = kn.tile;
    // the per-beam xyz table goes to LDS (no vector load left in the row loop: stores are never waited
    // for) whenever that does not cost k_decode a workgroup per CU
    if (xyzm == 1 || xyzm == 2) {
        const size_t extra = 2048 + tile * pose_per_col;   // fix-up frame list + pose table
        const size_t without = decode_lds_bytes(g, tile, true, false, slots_per_frame) + extra, with = decode_lds_bytes(g, tile, true, true, slots_per_frame) + extra;
        da.beam_lds = (kn.beam_lds && with <= 160 * 1024 && (160 * 1024) / with == (160 * 1024) / without) ? 1u : 0u;
    }
    // wide, short tiles (k_decode_wide): TW columns x TR rows with TW*TR*chan <= ~64 KB, TR chosen so that
    // the row chunks are equal.  Needs a batch large enough to fill the chip; fast mode only.
    const uint32_t narrow_tiles = (W + tile - 1) / tile;
    // General mapping on wide tiles: the column -> slot map of every frame is resolved once (k_slotmap), not by every tile
    const size_t resolver_lds = slotmap_lds_bytes(W, g.columns_per_packet, slots_per_frame);
    const bool mapped_ok = !fast && kn.slotmap && kn.tile == 0 && resolver_lds <= 160 * 1024;
    // for_fix: the tiles of the fix-up pass (k_decode_wide_fixup): resolve_frame's scratch lies under the tile image, and the
    // grid is persistent (no minimum number of blocks)
    // small: the tiles of the one-launch form for small batches (k_decode_wide_resolved): the scratch again, and the rows of a
    // tile chosen so that the few frames still make about two workgroups per CU
    auto setup_wide = [&](int want, bool for_fix = false, bool small = false) -> bool {
        const uint32_t chan = g.channel_data_size;
        if (!((fast || mapped_ok || small) && (want == 64 || want == 128 || want == 256 || want == 512) && chan && chan % 4 == 0 &&
              W >= (uint32_t)want))
            return false;
        if ((for_fix || small) && (want == 512 || want == 64 || out->gate_counts || resolver_lds > 64 * 1024)) return false;
        if (da.xyz_poses && ((uintptr_t)da.xyz_poses & 15u)) return false;   // the wide tiles fetch the poses in 16 B pieces
        const uint32_t rpp = 1024u / (uint32_t)want;  // rows per pass of the 256-thread workgroup
        uint32_t budget = (uint32_t)(kn.wide_kb > 0 ? kn.wide_kb : 64) * 1024u;
        // with a pose table next to it the tile shrinks so that two workgroups still share a CU's LDS (a 256 x 32 tile +
        // 12 KB of poses is 84 KB = one workgroup per CU: 1.36 ms instead of 0.7)
        if (pose_per_col) budget = std::min<uint32_t>(budget, 72u * 1024u - (uint32_t)want * (uint32_t)pose_per_col);
        uint32_t tr_max = budget / ((uint32_t)want * chan) / rpp * rpp;
        tr_max = std::min(tr_max, 84u / rpp * rpp);   // k_decode_wide keeps a row chunk's table rows in registers (84 rows at most)
        if (tr_max < rpp) return false;
        auto up = [&](uint32_t v) { return (v + rpp - 1) / rpp * rpp; };
        uint32_t nch = (H + tr_max - 1) / tr_max, tr = std::min(up((H + nch - 1) / nch), tr_max);
        for (uint32_t n2 = nch; n2 <= nch + 8 && n2 <= H; ++n2) {  // prefer equal chunks
            const uint32_t t2 = up((H + n2 - 1) / n2);
            if (t2 <= tr_max && t2 * n2 == H && t2 * 4 >= tr * 3) { nch = n2; tr = t2; break; }   // not at the price of much smaller tiles
        }
        const uint32_t tiles = (W + want - 1) / want;
        if (small) {
            if ((size_t)n_frames * tiles * nch >= (size_t)std::max(kn.wide_min_blocks, 0)) return false;   // not a small batch
            const uint32_t want_blocks = 2u * ctx->cus;
            const uint32_t need = (want_blocks + n_frames * tiles - 1) / (n_frames * tiles);   // row chunks for that many workgroups
            tr = std::max(rpp, std::min(tr, H / std::max(need, 1u) / rpp * rpp));
        }
        if (for_fix) {
            // short tiles: the (few) flagged frames of a batch spread over a whole XCD instead of keeping a handful of
            // workgroups busy for a full-height tile each (tools/ab/fixup_prof.sh: 57 us for ONE flagged frame with 32-row tiles)
            // about 16 KB of packet bytes per tile (8 rows of 256 dual-return columns, 16 rows of 128 single-return ones): smaller
            // tiles are all prologue (the 12 B/px profile's 128 x 8 tiles made its fix-up pass slower than r03's)
            const uint32_t rows = kn.fixup_rows > 0 ? (uint32_t)kn.fixup_rows : (16384u + (uint32_t)want * chan - 1u) / ((uint32_t)want * chan);
            da.fix_rows_small = std::max(rpp, std::min(tr, up(rows)));   // the launch keeps the tall tile; the kernel takes the short one for few flagged frames
            if (tiles > 32) return false;   // the frame's ready word carries one bit per column tile
        }
        if (kn.wide_rows > 0) tr = std::min((uint32_t)kn.wide_rows, H);
        if (tr > 84) return false;   // k_decode_wide keeps a row chunk's table rows in registers (3 doubles per thread) while its tile loads
        nch = (H + tr - 1) / tr;
        if (out->gate_counts && nch > OUSTER_HIP_GATE_CHUNKS) return false;  // one count slot per row chunk
        if (!for_fix && !small && (size_t)n_frames * tiles * nch < (size_t)std::max(kn.wide_min_blocks, 0)) return false;
        da.rows_per_tile = tr;
        da.row_chunks = nch;
        da.lds_col_slot = (tr * chan / 4 + 1) * 4;  // +1 dword: bank spread
        da.tiles_per_frame = tiles;
        size_t img_words = (size_t)want * (da.lds_col_slot >> 2) + 4;
        if (for_fix || small) img_words = std::max(img_words, (resolver_lds / 4 + 3) & ~(size_t)3);
        return decode_wide_lds_bytes(want, tr, (uint32_t)img_words) + 16 + (size_t)want * pose_per_col <= ((for_fix || small) ? 80u : 160u) * 1024;
    };
    // Persistent, double-buffered tiles filled by LDS-DMA (k_decode_stream): the optimistic pass of
    // the static profiles on large batches whose buffers keep every 16 B cell's phase fixed.
    StreamArgs sa{};
    auto plan_field = [&](const ouster_hip_bits& b, uint32_t* dws, uint32_t& n_dw, uint32_t max_dw, FieldPlan& fp) -> bool {
        fp.slot[0] = fp.slot[1] = fp.slot[2] = -1;
        fp.sh = (uint8_t)((b.offset & 3u) * 8u);
        if (b.mask == 0) return true;
        const uint32_t lo = (uint32_t)__builtin_ctzll(b.mask) >> 3, hi = (63u - (uint32_t)__builtin_clzll(b.mask)) >> 3;
        const uint32_t d0 = b.offset >> 2, first = (b.offset + lo) >> 2, last = (b.offset + hi) >> 2;
        for (uint32_t d = first; d <= last; ++d) {
            if (d - d0 > 2) return false;
            uint32_t k = 0;
            while (k < n_dw && dws[k] != d) ++k;
            if (k == n_dw) {
                if (n_dw == max_dw) return false;
                dws[n_dw++] = d;
            }
            fp.slot[d - d0] = (int8_t)k;
        }
        return true;
    };
    auto setup_stream = [&](int tw) -> bool {
        const uint32_t chan = g.channel_data_size, cpp = g.columns_per_packet;
        if (!(fast && spec != SPEC_GENERIC && xyzm != 3 && !da.xyz_poses && vec_ok && (tw == 128 || tw == 256 || tw == 512 || tw == 1024))) return false;
        if (W % (uint32_t)tw || (uint32_t)tw % cpp || (uint32_t)tw / cpp > 64 || chan == 0 || chan % 4) return false;
        const uint32_t ct = W / (uint32_t)tw, per_xcd = std::max(ctx->cus / 8u, 1u);
        if (ct > per_xcd) return false;
        // every 16 B cell keeps its phase from tile to tile: frame bases and the frame stride are multiples of 16
        if (((uintptr_t)packets & 15) || (((uint64_t)slots_per_frame * packet_stride) & 15)) return false;
        // the last column's last cell must stay inside its frame's buffer
        const uint64_t last_end = (uint64_t)(slots_per_frame - 1) * packet_stride + g.packet_header_size +
                                  (uint64_t)(cpp - 1) * g.col_size + g.col_header_size + (uint64_t)H * chan;
        if (last_end + 16 > (uint64_t)slots_per_frame * packet_stride) return false;
        const uint32_t rpp = 2048u / (uint32_t)tw;   // rows per pass of the 512-thread workgroup
        sa = StreamArgs{};
        if (!plan_field(g.col_measurement_id, sa.hdr_dw, sa.n_hdr, 8, sa.mid) ||
            !plan_field(g.col_status, sa.hdr_dw, sa.n_hdr, 8, sa.st) ||
            !plan_field(g.col_timestamp, sa.hdr_dw, sa.n_hdr, 8, sa.ts) ||
            !plan_field(g.alert_flags, sa.pkt_dw, sa.n_pkt, 4, sa.alert))
            return false;
        auto fits = [&](uint32_t tr) -> bool {
            if (tr == 0 || tr % rpp || H % tr || (tr * chan) % 16) return false;
            sa.tr = tr;
            sa.nch = H / tr;
            sa.ncell = tr * chan / 16 + 1;
            sa.npix_instr = ((uint32_t)tw * sa.ncell + 63) / 64;
            if (sa.npix_instr > 72) return false;
            uint32_t o = sa.npix_instr * 1024u;
            sa.hdr_off = o; o += sa.n_hdr * (uint32_t)tw * 4u;
            sa.pkt_off = o; o += 6u * 256u;
            sa.off_off = o; o += ((tr + 63) / 64) * 256u;
            sa.beam_off = o; o += ((tr * 18u + 63) / 64) * 256u;
            sa.ctx_bytes = (o + 1023u) & ~1023u;
            sa.fixed_off = 2u * sa.ctx_bytes;
            sa.lds_bytes = sa.fixed_off + 3u * (uint32_t)tw * 4u + 32u;
            return sa.lds_bytes <= 160u * 1024u;
        };
        bool ok = false;
        if (kn.stream_rows > 0) ok = fits((uint32_t)kn.stream_rows);
        else
            for (uint32_t tr = H / rpp * rpp; tr >= rpp && !ok; tr -= rpp) ok = fits(tr);   // the tallest tile that fits twice
        if (!ok) return false;
        if (out->gate_counts && sa.nch > OUSTER_HIP_GATE_CHUNKS) return false;
        sa.groups = per_xcd / ct;
        const uint64_t tiles = (uint64_t)n_frames * ct * sa.nch, wgs = 8ull * ct * sa.groups;
        if (kn.stream_min_tiles > 0 && tiles < wgs * (uint64_t)kn.stream_min_tiles) return false;
        // stores a wave issues per tile (one per output stream and lane row): with at least 63 of them behind a
        // prefetch the 6-bit in-order vmcnt itself proves the prefetch has landed
        uint32_t streams = 0;
        for (uint32_t i = 0; i < nf; ++i) streams += (da.planes[i] ? 1u : 0u) + (da.destaggered[i] ? 1u : 0u);
        for (int k = 0; k < 2; ++k)
            if (da.xyz[k]) streams += xyzm == 1 ? 3u : 6u;
        const uint32_t stores_per_wave = streams * (sa.tr / rpp);
        sa.wait0 = (kn.stream_wait == 0 && stores_per_wave >= 64) ? 0u : 1u;
        sa.order = (uint32_t)kn.stream_order;
        sa.loader = (kn.stream_loader > 0 && (tw == 128 || tw == 256) && !out->gate_counts && kn.stream_order == 0)
                        ? (uint32_t)std::min(kn.stream_loader, 4) : 0u;
        da.rows_per_tile = sa.tr;
        da.row_chunks = sa.nch;
        da.lds_col_slot = 0;
        da.tiles_per_frame = ct;
        return true;
    };
    // Which optimistic-pass kernel?  Forced by a knob, or -- large batches of a static profile -- timed: k_decode_wide
    // 256 x R, 128 x R, k_decode's 64-column tiles and (where the buffers allow it) the persistent k_decode_stream.
    // Which one wins depends on how the buffers happen to lie in HBM: the persistent kernel is
    // 1.5 - 3 % ahead where the memory system is fastest and up to 10 % behind where it is slowest.
    int stream = 0, wide = 0;
    int stream_auto = 0;   // tile width of the persistent candidate (0: not eligible)
    int stream_alt = 0;    // ... and of the other width, when that is eligible too (round 4: the 12 B/px profile's 256 x 16 tiles
                           // beat its 128 x 32 ones by 3 - 4 % on some boxes, tools/ab/single_variants.py)
    // Small batches (one tick of a few sensors, a single frame): one launch, every wide tile resolves its frame's column maps
    // itself -- neither an optimistic pass nor a second launch, whatever the buffer looks like.
    // Measured (tools/ab/small_batch.py, 4 x 128 x 2048 dual return): optimistic wide tiles of 8 rows + the fix-up launch 27 us
    // per call, k_decode's 16-column tiles + fix-up 33 us, the one-launch form 35 us (its workgroups resolve the frame
    // before they can start: 14 us against the 3 us a second launch costs).  So: a buffer with one slot per column takes the
    // optimistic wide tiles, any other shape the one-launch form (kn.small = 2 forces it for both).
    bool resolved = false, small_wide = false;
#ifdef OUSTER_EXPERIMENTS
    constexpr bool may_resolve = true;
#else
    constexpr bool may_resolve = false;   // default build: a small batch whose buffer is not "one slot per column" takes the general
                                          // mapping like a large one (k_slotmap + k_decode_wide); k_decode_wide_resolved is not compiled
#endif
    if (kn.small && kn.stream <= 0 && kn.wide < 0 && kn.tile == 0 && kn.fast && (fast || may_resolve)) {
        for (int tw : {256, 128})
            if (W >= (uint32_t)tw) {
                if (setup_wide(tw, false, true)) {
                    wide = tw;
                    resolved = may_resolve && (!fast || kn.small == 2);
                    small_wide = !resolved;
                }
                break;
            }
    }
    if (resolved || small_wide) {
        // nothing else to choose
    } else if (kn.stream > 0) {
        if (setup_stream(kn.stream)) stream = kn.stream;
    } else if (kn.stream < 0 && kn.wide < 0 && kn.tile == 0) {
        // column pieces of at least 256 B per tile row chunk: 256 columns for the 8 / 16 / 4 B/px profiles, 128 for 12 B/px
        if (setup_stream(256) && sa.tr * g.channel_data_size >= 256) stream_auto = 256;
        else if (setup_stream(128)) stream_auto = 128;
        else if (setup_stream(256)) stream_auto = 256;
        if (stream_auto) {
            const int other = stream_auto == 256 ? 128 : 256;
            if (setup_stream(other)) stream_alt = other;
        }
    }
    ouster_hip_ctx::Tune* tuning = nullptr;
    int tune_slot = -1;
    ctx->last_tuner = "none";   // forced by a knob, or a shape with nothing to choose
    if (stream || resolved || small_wide) {
        // forced / nothing to choose
    } else if (kn.wide >= 0) {  // forced (experiments, tests)
        if (kn.wide && setup_wide(kn.wide)) wide = kn.wide;
    } else if (!fast) {
        // general mapping on wide tiles (k_slotmap first): column pieces of at least 256 B, like the persistent kernel's choice
        if (setup_wide(256) && da.rows_per_tile * g.channel_data_size >= 256) wide = 256;
        else if (setup_wide(128)) wide = 128;
        else if (setup_wide(256)) wide = 256;
    } else if (setup_wide(256)) {
        int sel = 256;   // >= 1000: the persistent kernel with tile width sel - 1000
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (kn.tune && cap == hipStreamCaptureStatusNone) {
            uint64_t key = 1469598103934665603ull;
            auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
            uint32_t lg = 0;  // batches of similar size share a verdict
            while ((2u << lg) <= n_frames) ++lg;
            mix((uint64_t)spec); mix(W); mix(H); mix(g.channel_data_size); mix(lg); mix((uint64_t)xyzm);
            uint64_t pm = 0, dm = 0;
            for (uint32_t i = 0; i < nf; ++i) {
                pm |= da.planes[i] ? (1ull << i) : 0;
                dm |= da.destaggered[i] ? (1ull << i) : 0;
            }
            mix(pm); mix(dm);
            mix((da.xyz[0] ? 1u : 0u) | (da.xyz[1] ? 2u : 0u) | (da.gate_counts ? 4u : 0u) | (da.xyz_poses ? 8u : 0u));
            mix((uint64_t)stream_auto); mix((uint64_t)stream_alt);
            if (ctx->tune.size() > 64 && !ctx->tune.count(key)) {  // bounded: forget everything, re-learn
                for (auto& kv : ctx->tune)
                    for (auto& pr : kv.second.ev)
                        for (hipEvent_t e : pr)
                            if (e) (void)hipEventDestroy(e);
                ctx->tune.clear();
            }
            const bool fresh = !ctx->tune.count(key);
            ouster_hip_ctx::Tune& t = ctx->tune[key];
            const int cand[ouster_hip_ctx::Tune::NCAND] = {256, 128, 0, 1000 + stream_auto, 1000 + stream_alt};
            const int nc = stream_auto ? (stream_alt ? 5 : 4) : 3;
            if (fresh) {   // an earlier process (or rank) has timed this workload on this device: launch its choice from call 1
                auto hit = ctx->tune_cache.find(key);
                if (hit != ctx->tune_cache.end()) {
                    bool known = false;
                    for (int c = 0; c < nc; ++c) known = known || cand[c] == hit->second;
                    if (known) { t.best = hit->second; t.from_cache = true; }
                }
            }
            // Four launches of each candidate, back to back (a launch that follows a different variant
            // is not representative of the steady state: alternating the candidates made the 64-column
            // kernel look 8 % faster than it then ran).  Single launches vary by ~10 %, mostly upwards,
            // and the candidates can be within 1-5 % of each other: each candidate's fastest sample
            // counts, its first one (cold code, cold TLB, another variant's write-backs still draining)
            // only when nothing else landed.  The clocks are polled, never waited for: until all samples
            // have landed the default variant runs.
            constexpr int R = ouster_hip_ctx::Tune::ROUNDS;
            const int NS = nc * R;
            if (t.best == -2 && t.calls >= NS) {
                bool all = true;
                for (int i = 0; i < NS && all; ++i) all = hipEventQuery(t.ev[i][1]) == hipSuccess;
                if (!all) (void)hipGetLastError();
                else {
                    float cold[ouster_hip_ctx::Tune::NCAND] = {0, 0, 0, 0, 0};
                    for (int i = 0; i < NS; ++i) {
                        float ms = 0;
                        if (hipEventElapsedTime(&ms, t.ev[i][0], t.ev[i][1]) != hipSuccess || ms <= 0) continue;
                        float& slot = (i % R == 0) ? cold[i / R] : t.ms[i / R];
                        if (slot == 0 || ms < slot) slot = ms;
                    }
                    for (int c = 0; c < nc; ++c)
                        if (t.ms[c] == 0) t.ms[c] = cold[c];
                    t.best = 256;
                    float best_ms = 0;
                    for (int c = 0; c < nc; ++c)
                        if (t.ms[c] > 0 && (best_ms == 0 || t.ms[c] < best_ms)) { t.best = cand[c]; best_ms = t.ms[c]; }
                    tune_cache_append(ctx, key, t.best, best_ms);
                }
            }
            if (t.best != -2) {
                sel = t.best;
                ctx->last_tuner = t.from_cache ? "cache" : "measured";
            } else if (t.calls < NS) {
                tune_slot = t.calls;
                sel = cand[tune_slot / R];
                tuning = &t;
                ++t.calls;
                ctx->last_tuner = "measuring";
            } else {
                ctx->last_tuner = "measuring";   // every sample is queued, not all have landed: the default variant runs
            }
        }
        if (sel >= 1000 && setup_stream(sel - 1000)) stream = sel - 1000;
        else if (sel > 0 && sel < 1000 && setup_wide(sel)) wide = sel;
        else if (sel != 0 && setup_wide(256)) wide = 256;
    }
    if (!wide && !stream) {
        da.rows_per_tile = da.row_chunks = da.lds_col_slot = 0;
        da.tiles_per_frame = narrow_tiles;
    }
    da.xcd_map = (kn.xcd && n_frames >= 8) ? 1u : 0u;
    da.mode = resolved ? MODE_RESOLVED : fast ? MODE_FAST : MODE_GENERAL;
    if (resolved) ctx->state_dirty = false;   // this call leaves the frame words alone

    // ---- the fix-up pass behind an optimistic pass: the tiles of the frames the optimistic pass flagged are looked at again
    // with the frame's real column maps and redone where those differ from "slot c holds column c".  Wide tiles
    // (fixup_crew, wide_tile.h) where the format allows them, k_decode_fixup's 64-column tiles otherwise.  Always launched
    // behind the optimistic kernel; the workgroups of a clean batch leave after two scalar loads (FS_ANY).  Its tile shape and
    // every allocation of the call are settled here, before the first launch.
    struct Shape { uint32_t rows, chunks, slot, tiles, rows_small; };
    auto shape_now = [&]() { return Shape{da.rows_per_tile, da.row_chunks, da.lds_col_slot, da.tiles_per_frame, da.fix_rows_small}; };
    auto shape_set = [&](const Shape& sh) {
        da.rows_per_tile = sh.rows; da.row_chunks = sh.chunks; da.lds_col_slot = sh.slot; da.tiles_per_frame = sh.tiles; da.fix_rows_small = sh.rows_small;
    };
    const Shape opt = shape_now();
    const bool want_fix = fast && kn.fixup && !resolved;
    int fix_wide = 0;
    Shape fixs{};
    if (want_fix) {
        auto try_fix = [&](int tw) {
            const bool ok = setup_wide(tw, true);
            if (ok) { fix_wide = tw; fixs = shape_now(); }
            shape_set(opt);
            return ok;
        };
        if (kn.fixup_wide && kn.tile == 0) {
            if (kn.fixup_wide > 1) try_fix(kn.fixup_wide);
            else if (setup_wide(256, true) && da.rows_per_tile * g.channel_data_size >= 256) try_fix(256);
            else if (!try_fix(128)) try_fix(256);
            shape_set(opt);
        }
        if (kn.hdr_words) {   // the optimistic pass leaves the packed column ids for the fix-up pass
            if (ctx->hdrw.ensure((size_t)n_frames * W * sizeof(uint32_t))) return fail(OUSTER_HIP_ERR_RUNTIME, "out of device memory (header words)");
            da.hdr_words = (uint32_t*)ctx->hdrw.p;
        }
        if (fix_wide && ctx->slotmap.ensure((size_t)n_frames * W * sizeof(int32_t) * 2)) fix_wide = 0;   // the 64-column fix-up needs no maps
        da.fast_tiles = opt.tiles;    // column tiles of the optimistic pass (slots of tile_valid)
    }
    if (wide && !fast && !resolved) {
        if (ctx->slotmap.ensure((size_t)n_frames * W * sizeof(int32_t) * 2)) return fail(OUSTER_HIP_ERR_RUNTIME, "out of device memory (slot map)");
        da.slot_map = (int32_t*)ctx->slotmap.p;
        da.hdr_map = da.slot_map + (size_t)n_frames * W;
    }

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->timing && (ctx->timing_calls++ % ctx->timing_every) == 0) {
        if (ctx->ev_used == ctx->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            ctx->ev_pool.emplace_back(a, b);
        }
        e0 = ctx->ev_pool[ctx->ev_used].first;
        e1 = ctx->ev_pool[ctx->ev_used].second;
        ctx->ev_used++;
        HIP_TRY(hipEventRecord(e0, st));
    }
    if (tuning) {
        for (int k = 0; k < 2; ++k)
            if (!tuning->ev[tune_slot][k]) HIP_TRY(hipEventCreate(&tuning->ev[tune_slot][k]));
        HIP_TRY(hipEventRecord(tuning->ev[tune_slot][0], st));
    }
    if (wide && !fast && !resolved) HIP_TRY(launch_slotmap(da, ctx->device, st));
    if (stream) {
        HIP_TRY(launch_decode_stream(da, sa, spec, stream, xyzm, ctx->device, st));
        ctx->last_tile_cols = stream;
        ctx->last_tile_rows = (int)sa.tr;
        ctx->last_kernel = sa.loader ? "k_decode_stream2" : "k_decode_stream";   // dedicated loader waves / every wave fetches
    } else if (wide) {
        HIP_TRY(launch_decode_wide(da, spec, wide, xyzm, ctx->device, st));
        ctx->last_tile_cols = wide;
        ctx->last_tile_rows = (int)opt.rows;
        ctx->last_kernel = resolved ? "k_decode_wide_resolved" : "k_decode_wide";
    } else {
        HIP_TRY(launch_decode(da, spec, tile, xyzm, ctx->device, st));
        ctx->last_tile_cols = tile;
        ctx->last_tile_rows = (int)H;
        ctx->last_kernel = "k_decode";
    }
    if (tuning) HIP_TRY(hipEventRecord(tuning->ev[tune_slot][1], st));
    if (e1) HIP_TRY(hipEventRecord(e1, st));
    if (want_fix) {
        da.mode = MODE_FIXUP;
#ifdef OUSTER_PHASE_TIMING
        da.phase_times = phase_times_all;
#endif
        if (fix_wide) {
            shape_set(fixs);
            da.slot_map = (int32_t*)ctx->slotmap.p;
            da.hdr_map = da.slot_map + (size_t)n_frames * W;
            HIP_TRY(launch_decode_wide(da, spec, fix_wide, xyzm, ctx->device, st, ctx->resident_wgs));
        } else {
            da.rows_per_tile = 0;
            da.lds_col_slot = 0;
            da.row_chunks = ctx->resident_wgs;     // persistent grid: what the device keeps resident
            da.tiles_per_frame = narrow_tiles;
            HIP_TRY(launch_decode(da, spec, tile, xyzm, ctx->device, st));
        }
    }
    if (want_fix) ctx->state_dirty = false;
    return OUSTER_HIP_OK;
}

// ---- standalone destagger -----------------------------------------------------------------
int ouster_hip_destagger(ouster_hip_ctx* ctx, const void* src, void* dst, uint32_t h, uint32_t w,
                         uint32_t elem_bytes, const int32_t* shifts, uint32_t n_shifts,
                         int inverse, uint32_t n_images) {
    if (!ctx || !shifts) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_shifts != h)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "image height does not match shifts size");
    if (n_images == 0 || h == 0 || w == 0) return OUSTER_HIP_OK;
    if (!src || !dst || src == dst) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "bad image pointers");
    if (elem_bytes == 0) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "elem_bytes is 0");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<int32_t> off;
    dest_offsets(shifts, h, w, inverse, off);
    if (ensure_offsets(ctx, off)) return fail(OUSTER_HIP_ERR_RUNTIME, "offset upload failed");
    ctx->shifts_host.clear();  // ouster_hip_decode's cache key no longer describes `offsets`
    DestaggerArgs a{};
    a.src = src;
    a.dst = dst;
    a.h = h;
    a.w = w;
    a.elem = elem_bytes;
    a.offsets = (const int32_t*)ctx->offsets.p;
    HIP_TRY(launch_destagger(a, n_images, ctx->stream));
    return OUSTER_HIP_OK;
}

// ---- standalone cartesian -------------------------------------------------------------------
int ouster_hip_cartesian(ouster_hip_ctx* ctx, const ouster_hip_lut* lut, const uint32_t* range,
                         void* xyz, int xyz_dtype, uint32_t n_images) {
    if (!ctx || !lut) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (xyz_dtype != OUSTER_HIP_F32 && xyz_dtype != OUSTER_HIP_F64)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "xyz_dtype must be F32 or F64");
    if (n_images == 0) return OUSTER_HIP_OK;
    if (!range || !xyz) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL image pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    CartesianArgs a{};
    a.lut = lut->dev;
    a.range = range;
    a.xyz = xyz;
    a.w = lut->w;
    a.h = lut->h;
    a.n_images = n_images;
    a.xyz_dtype = xyz_dtype;
    const size_t npx = (size_t)lut->w * lut->h;
    a.vec_ok = al16(range) && al16(xyz) && (npx % 4 == 0);
    const int mode = lut->separable ? (xyz_dtype == OUSTER_HIP_F32 ? 1 : 2) : 3;
    HIP_TRY(launch_cartesian(a, mode, ctx->stream));
    return OUSTER_HIP_OK;
}

// ---- dense dewarp -----------------------------------------------------------------------------
int ouster_hip_dewarp(ouster_hip_ctx* ctx, const void* points, const double* poses, void* dewarped,
                      int dtype, uint32_t h, uint32_t w, uint32_t n_images) {
    if (!ctx) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (dtype != OUSTER_HIP_F32 && dtype != OUSTER_HIP_F64)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "dtype must be F32 or F64");
    if (n_images == 0 || h == 0 || w == 0) return OUSTER_HIP_OK;
    if (!points || !poses || !dewarped) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    DewarpArgs a{};
    a.points = points;
    a.out = dewarped;
    a.poses = poses;
    a.w = w;
    a.h = h;
    a.n_images = n_images;
    a.dtype = dtype;
    HIP_TRY(launch_dewarp(a, ctx->stream));
    return OUSTER_HIP_OK;
}

// ---- host containers (include/ouster_hip.h, "host containers") -----------------------------------
int ouster_hip_device_alloc(ouster_hip_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    *out = nullptr;
    const hipError_t e = counted_malloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(OUSTER_HIP_ERR_RUNTIME, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return OUSTER_HIP_OK;
}
void ouster_hip_device_free(void* p) { counted_free(p); }

int ouster_hip_ctx_scratch(ouster_hip_ctx* ctx, uint32_t slot, size_t bytes, void** out) {
    if (!ctx || !out || slot >= 8) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "bad scratch request");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->user_scratch[slot].ensure(bytes ? bytes : 1)) return fail(OUSTER_HIP_ERR_RUNTIME, "out of device memory (scratch %u, %zu bytes)", slot, bytes);
    *out = ctx->user_scratch[slot].p;
    return OUSTER_HIP_OK;
}

int ouster_hip_copy_in(ouster_hip_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    if (!ctx || (bytes && (!dev_dst || !host_src))) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (bytes) HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OUSTER_HIP_OK;
}
int ouster_hip_copy_out(ouster_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (!ctx || (bytes && (!host_dst || !dev_src))) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (bytes) HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return OUSTER_HIP_OK;
}

namespace {
// One host array of a *_host call: pool memory is handed to the kernel as it is, anything else gets a slot of the context's
// grow-only scratch (inputs copied in now, outputs copied out by finish()).
struct HostArg {
    void* dev = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    bool staged = false;
};
int host_arg(ouster_hip_ctx* ctx, const void* host, size_t bytes, uint32_t slot, bool is_input, HostArg& a) {
    a.host = const_cast<void*>(host);
    a.bytes = bytes;
    if (ouster_hip_host_is_pinned(host, bytes)) {
        a.dev = a.host;
        return OUSTER_HIP_OK;
    }
    a.staged = true;
    if (ctx->user_scratch[slot].ensure(bytes)) return fail(OUSTER_HIP_ERR_RUNTIME, "out of device memory (scratch %u, %zu bytes)", slot, bytes);
    a.dev = ctx->user_scratch[slot].p;
    if (is_input) HIP_TRY(hipMemcpyAsync(a.dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OUSTER_HIP_OK;
}
int host_finish(ouster_hip_ctx* ctx, const HostArg& out) {
    if (out.staged) HIP_TRY(hipMemcpyAsync(out.host, out.dev, out.bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return OUSTER_HIP_OK;
}
}  // namespace

int ouster_hip_destagger_host(ouster_hip_ctx* ctx, const void* src, void* dst, uint32_t h, uint32_t w,
                              uint32_t elem_bytes, const int32_t* shifts, uint32_t n_shifts, int inverse) {
    if (!ctx || !shifts) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_shifts != h) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "image height does not match shifts size");
    if (h == 0 || w == 0) return OUSTER_HIP_OK;
    if (!src || !dst || elem_bytes == 0) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "bad image arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)h * w * elem_bytes;
    HostArg in, out;
    int rc = host_arg(ctx, src, bytes, 0, true, in);
    if (rc == OUSTER_HIP_OK) rc = host_arg(ctx, dst, bytes, 1, false, out);
    if (rc != OUSTER_HIP_OK) return rc;
    // in place (src == dst) in the reference copies rows through each other: not a supported call there either; go through scratch
    if (in.dev == out.dev) {
        if (ctx->user_scratch[1].ensure(bytes)) return fail(OUSTER_HIP_ERR_RUNTIME, "out of device memory (scratch)");
        out.dev = ctx->user_scratch[1].p;
        out.staged = true;
    }
    rc = ouster_hip_destagger(ctx, in.dev, out.dev, h, w, elem_bytes, shifts, n_shifts, inverse, 1);
    if (rc != OUSTER_HIP_OK) return rc;
    return host_finish(ctx, out);
}

int ouster_hip_cartesian_host(ouster_hip_ctx* ctx, const ouster_hip_lut* lut, const uint32_t* range, void* xyz,
                              int xyz_dtype) {
    if (!ctx || !lut) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (xyz_dtype != OUSTER_HIP_F32 && xyz_dtype != OUSTER_HIP_F64)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "xyz_dtype must be F32 or F64");
    const size_t npx = (size_t)lut->w * lut->h;
    if (npx == 0) return OUSTER_HIP_OK;
    if (!range || !xyz) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL image pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    HostArg in, out;
    int rc = host_arg(ctx, range, npx * 4, 0, true, in);
    if (rc == OUSTER_HIP_OK) rc = host_arg(ctx, xyz, npx * 3 * (xyz_dtype == OUSTER_HIP_F64 ? 8 : 4), 1, false, out);
    if (rc != OUSTER_HIP_OK) return rc;
    rc = ouster_hip_cartesian(ctx, lut, (const uint32_t*)in.dev, out.dev, xyz_dtype, 1);
    if (rc != OUSTER_HIP_OK) return rc;
    return host_finish(ctx, out);
}

int ouster_hip_dewarp_host(ouster_hip_ctx* ctx, const void* points, const double* poses, void* dewarped, int dtype,
                           uint32_t h, uint32_t w) {
    if (!ctx) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (dtype != OUSTER_HIP_F32 && dtype != OUSTER_HIP_F64) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "dtype must be F32 or F64");
    if (h == 0 || w == 0) return OUSTER_HIP_OK;
    if (!points || !poses || !dewarped) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL pointer");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t pbytes = (size_t)h * w * 3 * (dtype == OUSTER_HIP_F64 ? 8 : 4);
    HostArg in, po, out;
    int rc = host_arg(ctx, points, pbytes, 0, true, in);
    if (rc == OUSTER_HIP_OK) rc = host_arg(ctx, poses, (size_t)w * 128, 2, true, po);
    if (rc == OUSTER_HIP_OK) rc = host_arg(ctx, dewarped, pbytes, 1, false, out);
    if (rc != OUSTER_HIP_OK) return rc;
    rc = ouster_hip_dewarp(ctx, in.dev, (const double*)po.dev, out.dev, dtype, h, w, 1);
    if (rc != OUSTER_HIP_OK) return rc;
    return host_finish(ctx, out);
}

// ---- range-gated, compacting frame dewarp ------------------------------------------------------
int ouster_hip_range_gate(double min_range, double max_range, uint32_t* min_r, uint32_t* max_r, int* empty) {
    // uint32_t min_r = ceil(min_range * 1e3), max_r = floor(max_range * 1e3): dewarp_impl.h:33-34
    const double lo = std::ceil(min_range * 1e3), hi = std::floor(max_range * 1e3);
    const bool none = !(hi >= 0) || !(lo <= 4294967295.0) || hi < lo;
    if (empty) *empty = none ? 1 : 0;
    if (min_r) *min_r = (none || lo <= 0) ? 0u : (uint32_t)lo;
    if (max_r) *max_r = none ? 0u : (hi >= 4294967295.0 ? 0xffffffffu : (uint32_t)hi);
    return OUSTER_HIP_OK;
}

int ouster_hip_dewarp_frames(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                             const uint32_t* range, const uint32_t* status,
                             const uint64_t* timestamp, const double* poses, uint32_t n_frames,
                             double min_range, double max_range, int dtype, void* points,
                             uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                             uint64_t capacity, uint64_t* frame_offsets) {
    return ouster_hip_dewarp_frames_counted(ctx, luts, n_luts, range, status, timestamp, poses, n_frames, min_range,
                                            max_range, dtype, points, frame_idxs, col_idxs, timestamps_ns, capacity,
                                            frame_offsets, nullptr);
}

static int dewarp_frames_impl(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                              const uint32_t* range, const uint32_t* status,
                              const uint64_t* timestamp, const double* poses, const float* pose_rows, uint32_t n_frames,
                              double min_range, double max_range, int dtype, void* points,
                              uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                              uint64_t capacity, uint64_t* frame_offsets, const uint16_t* gate_counts) {
    if (!ctx) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (dtype != OUSTER_HIP_F32 && dtype != OUSTER_HIP_F64)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "dtype must be F32 or F64");
    if (!luts || n_luts == 0) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "at least one LUT is required");
    for (uint32_t i = 0; i < n_luts; ++i) {
        if (!luts[i]) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL LUT handle");
        if (luts[i]->w != luts[0]->w || luts[i]->h != luts[0]->h)
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "all LUTs of a batch must have the same dimensions");
        if (luts[i]->separable != luts[0]->separable)
            return fail(OUSTER_HIP_ERR_UNSUPPORTED, "cannot mix separable and full LUTs in one batch");
    }
    if (!frame_offsets) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "frame_offsets is NULL");
    if (timestamps_ns && !timestamp)
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "timestamps_ns requested without column timestamps");
    HIP_TRY(hipSetDevice(ctx->device));
    if (n_frames == 0) {
        HIP_TRY(hipMemsetAsync(frame_offsets, 0, sizeof(uint64_t), ctx->stream));
        return OUSTER_HIP_OK;
    }
    if (pose_rows && (((uintptr_t)pose_rows) & 15u))
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "pose_rows must be 16-byte aligned");
    if (!range || !status || (!poses && !pose_rows) || (!points && capacity))
        return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL pointer");
    const uint32_t w = luts[0]->w, h = luts[0]->h;
    // uint32_t min_r = ceil(min_range * 1e3), max_r = floor(max_range * 1e3): dewarp_impl.h:33-34
    const double lo = std::ceil(min_range * 1e3), hi = std::floor(max_range * 1e3);
    if (!(hi >= 0) || !(lo <= 4294967295.0) || hi < lo) {  // nothing can pass the gate
        HIP_TRY(hipMemsetAsync(frame_offsets, 0, sizeof(uint64_t) * ((size_t)n_frames + 1), ctx->stream));
        return OUSTER_HIP_OK;
    }
    std::vector<LutDev> l(n_luts);
    for (uint32_t i = 0; i < n_luts; ++i) l[i] = luts[i]->dev;
    if (ensure_luts(ctx, l)) return fail(OUSTER_HIP_ERR_RUNTIME, "LUT descriptor upload failed");
    // scratch: the three-kernel path's per-column offsets, then the single-pass path's tile words
    const size_t col_off_bytes = ((size_t)n_frames * (w + 1) * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t tile_words = (size_t)n_frames * ((w + 63) / 64) + 128;
    if (ctx->scratch.ensure(col_off_bytes + tile_words * 8))
        return fail(OUSTER_HIP_ERR_RUNTIME, "scratch allocation failed");
    DewarpFramesArgs a{};
    a.range = range;
    a.status = status;
    a.timestamp = timestamp;
    a.poses = poses;
    a.pose_rows = pose_rows;
    a.luts = (const LutDev*)ctx->luts.p;
    a.n_luts = n_luts;
    a.w = w;
    a.h = h;
    a.n_frames = n_frames;
    a.min_r = lo <= 0 ? 0u : (uint32_t)lo;
    a.max_r = hi >= 4294967295.0 ? 0xffffffffu : (uint32_t)hi;
    a.dtype = dtype;
    a.col_off = (uint32_t*)ctx->scratch.p;
    a.gate_counts = gate_counts;
#ifdef OUSTER_EXPERIMENTS
    a.tile_state = (ctx->knobs.dewarp_single_pass && !gate_counts && !pose_rows) ? (uint64_t*)((uint8_t*)ctx->scratch.p + col_off_bytes) : nullptr;
#else
    a.tile_state = nullptr;   // the single-pass kernels are not in a default build (knob "dewarp_single_pass" is then without effect)
#endif
    a.frame_off = frame_offsets;
    a.points = points;
    a.frame_idxs = frame_idxs;
    a.col_idxs = col_idxs;
    a.timestamps_ns = timestamps_ns;
    a.capacity = capacity;
    HIP_TRY(launch_dewarp_frames(a, luts[0]->separable, ctx->stream, ctx->knobs.dwf_stream));
    return OUSTER_HIP_OK;
}

int ouster_hip_dewarp_frames_counted(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                                     const uint32_t* range, const uint32_t* status,
                                     const uint64_t* timestamp, const double* poses, uint32_t n_frames,
                                     double min_range, double max_range, int dtype, void* points,
                                     uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                                     uint64_t capacity, uint64_t* frame_offsets, const uint16_t* gate_counts) {
    return dewarp_frames_impl(ctx, luts, n_luts, range, status, timestamp, poses, nullptr, n_frames, min_range, max_range, dtype,
                              points, frame_idxs, col_idxs, timestamps_ns, capacity, frame_offsets, gate_counts);
}

int ouster_hip_dewarp_frames_rows(ouster_hip_ctx* ctx, const ouster_hip_lut* const* luts, uint32_t n_luts,
                                  const uint32_t* range, const uint32_t* status, const uint64_t* timestamp,
                                  const float* pose_rows, uint32_t n_frames, double min_range, double max_range,
                                  void* points, uint32_t* frame_idxs, uint32_t* col_idxs, uint64_t* timestamps_ns,
                                  uint64_t capacity, uint64_t* frame_offsets, const uint16_t* gate_counts) {
    if (!pose_rows && n_frames) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL pointer");
    return dewarp_frames_impl(ctx, luts, n_luts, range, status, timestamp, nullptr, pose_rows, n_frames, min_range, max_range,
                              OUSTER_HIP_F32, points, frame_idxs, col_idxs, timestamps_ns, capacity, frame_offsets, gate_counts);
}

// ---- OSF field planes ---------------------------------------------------------------------------
int ouster_hip_osf_unpack(ouster_hip_ctx* ctx, const ouster_hip_osf_plane* planes, uint32_t n_planes,
                          uint32_t h, uint32_t w, const int32_t* pixel_shift_by_row) {
    if (!ctx) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n_planes == 0 || h == 0 || w == 0) return OUSTER_HIP_OK;
    if (!planes) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "planes is NULL");
    bool any_png = false;
    size_t unfiltered_bytes = 0;          // device scratch for the planes that arrive as filtered PNG scanlines
    uint32_t n_filtered = 0, max_row_bytes = 0;
    for (uint32_t i = 0; i < n_planes; ++i) {
        const ouster_hip_osf_plane& p = planes[i];
        if (!p.src || !p.dst) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: NULL pointer", i);
        const uint32_t e = p.dst_elem_size;
        if (!(e == 1 || e == 2 || e == 4 || e == 8))
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: unsupported element size %u", i, e);
        if (p.flags & ~OUSTER_HIP_OSF_FLAG_FILTERED) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: unknown flags", i);
        static const uint32_t want[6] = {0, 1, 2, 3, 4, 8};
        if (p.encoding >= OUSTER_HIP_OSF_PNG_GRAY8 && p.encoding <= OUSTER_HIP_OSF_PNG_RGBA16) {
            if (p.src_pixel_bytes != want[p.encoding])
                return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: pixel size does not match its PNG type", i);
            any_png = true;
            if (p.flags & OUSTER_HIP_OSF_FLAG_FILTERED) {
                ++n_filtered;
                unfiltered_bytes += ((size_t)h * w * p.src_pixel_bytes + 15) & ~(size_t)15;
                max_row_bytes = std::max(max_row_bytes, w * p.src_pixel_bytes);
            }
        } else if (p.encoding == OUSTER_HIP_OSF_ZPNG) {
            if (p.src_pixel_bytes == 0 || p.src_pixel_bytes > 8)
                return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: ZPNG pixels are 1..8 bytes", i);
            if (p.flags) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: ZPNG planes carry no PNG filters", i);
        } else {
            return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "plane %u: unknown encoding %u", i, p.encoding);
        }
    }
    if (n_filtered && osf_png_unfilter_lds_bytes(w, max_row_bytes) > 160u * 1024u)   // the launcher's own figure
        return fail(OUSTER_HIP_ERR_UNSUPPORTED, "a PNG scanline of %u bytes needs %zu bytes of LDS for the device filters (160 KB per workgroup): "
                    "hand the planes over unfiltered (flags = 0)", max_row_bytes, osf_png_unfilter_lds_bytes(w, max_row_bytes));
    HIP_TRY(hipSetDevice(ctx->device));
    OsfUnpackArgs a{};
    a.h = h;
    a.w = w;
    if (any_png && pixel_shift_by_row) {  // stagger(img, px_offset) = destagger with inverse offsets
        std::vector<int32_t> off;
        dest_offsets(pixel_shift_by_row, h, w, 1, off);
        if (ensure_offsets(ctx, off)) return fail(OUSTER_HIP_ERR_RUNTIME, "offset upload failed");
        ctx->shifts_host.clear();
        a.offsets = (const int32_t*)ctx->offsets.p;
    }
    // the job arrays: the unpack jobs (filtered planes read the device-side unfiltered copy), then the unfilter jobs
    std::vector<ouster_hip_osf_plane> jobs(planes, planes + n_planes);
    std::vector<OsfUnfilterJob> unf;
    if (n_filtered) {
        if (ctx->osf_pixels.ensure(unfiltered_bytes)) return fail(OUSTER_HIP_ERR_RUNTIME, "scratch allocation failed");
        size_t at = 0;
        for (uint32_t i = 0; i < n_planes; ++i) {
            if (!(jobs[i].flags & OUSTER_HIP_OSF_FLAG_FILTERED)) continue;
            OsfUnfilterJob j{};
            j.raw = (const uint8_t*)jobs[i].src;
            j.out = (uint8_t*)ctx->osf_pixels.p + at;
            j.bpp = jobs[i].src_pixel_bytes;
            unf.push_back(j);
            jobs[i].src = j.out;
            jobs[i].flags = 0;
            at += ((size_t)h * w * j.bpp + 15) & ~(size_t)15;
        }
    }
    const size_t bytes = (size_t)n_planes * sizeof(ouster_hip_osf_plane), ubytes = unf.size() * sizeof(OsfUnfilterJob);
    if (ctx->scratch.ensure(bytes + ubytes)) return fail(OUSTER_HIP_ERR_RUNTIME, "scratch allocation failed");
    HIP_TRY(hipMemcpyAsync(ctx->scratch.p, jobs.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    if (ubytes) HIP_TRY(hipMemcpyAsync((uint8_t*)ctx->scratch.p + bytes, unf.data(), ubytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the job arrays are this call's memory
    if (n_filtered) {
        OsfUnfilterArgs u{};
        u.jobs = (const OsfUnfilterJob*)((uint8_t*)ctx->scratch.p + bytes);
        u.h = h;
        u.w = w;
        HIP_TRY(launch_osf_png_unfilter(u, n_filtered, max_row_bytes, ctx->stream));
    }
    a.planes = (const ouster_hip_osf_plane*)ctx->scratch.p;
    HIP_TRY(launch_osf_unpack(a, n_planes, ctx->stream));
    return OUSTER_HIP_OK;
}

const char* ouster_hip_last_decode_kernel(ouster_hip_ctx* ctx) { return ctx ? ctx->last_kernel : ""; }

int ouster_hip_last_decode_tile(ouster_hip_ctx* ctx, int* tile_cols, int* tile_rows) {
    if (!ctx) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (tile_cols) *tile_cols = ctx->last_tile_cols;
    if (tile_rows) *tile_rows = ctx->last_tile_rows;
    return OUSTER_HIP_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int ouster_hip_timing_enable(ouster_hip_ctx* ctx, int on) {
    if (!ctx) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "ctx is NULL");
    // on > 1: sample every on-th launch (the event pair around a kernel costs the stream 2 - 3 us: a timed region that wants the
    // kernel's duration without paying that on every step samples)
    ctx->timing = on != 0;
    ctx->timing_every = on > 1 ? (uint32_t)on : 1u;
    ctx->timing_calls = 0;
    ctx->ev_used = 0;
    return OUSTER_HIP_OK;
}

int ouster_hip_timing_read(ouster_hip_ctx* ctx, double* avg_ms, uint32_t* n_launches) {
    if (!ctx || !avg_ms) return fail(OUSTER_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double total = 0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        total += ms;
    }
    *avg_ms = ctx->ev_used ? total / (double)ctx->ev_used : 0.0;
    if (n_launches) *n_launches = (uint32_t)ctx->ev_used;
    ctx->ev_used = 0;
    return OUSTER_HIP_OK;
}

}  // extern "C"
