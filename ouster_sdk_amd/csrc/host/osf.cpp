// osf.cpp -- see include/ouster/osf/osf.h.  Host side: flatbuffer walk, CRC32, zlib / zstd, PNG
// scanline filters; the per-pixel work goes to ouster_hip_osf_unpack.
#include "ouster/osf/osf.h"

#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <fstream>
#include <memory>
#include <functional>
#include <atomic>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "host_internal.h"
#include "ouster/hip/device_buffer.h"

// libzstd ships without headers in this image; these are its stable C entry points
extern "C" {
size_t ZSTD_decompress(void* dst, size_t dst_capacity, const void* src, size_t compressed_size);
unsigned ZSTD_isError(size_t code);
}

namespace ouster {
namespace sdk {
namespace osf {

using namespace core;

namespace {

// ---------------------------------------------------------------------------------------
// bounds-checked FlatBuffers reads (little endian; tables, vectors, strings)
// ---------------------------------------------------------------------------------------
struct Span {
    const uint8_t* p = nullptr;
    size_t n = 0;
};

[[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("OSF: malformed buffer (") + what + ")"); }

template <typename T>
T rd(const Span& b, size_t pos) {
    if (pos > b.n || b.n - pos < sizeof(T)) bad("read past the end");  // subtraction form: a crafted offset cannot wrap
    T v;
    std::memcpy(&v, b.p + pos, sizeof(T));
    return v;
}

struct Table {
    Span b;
    size_t pos = 0, vt = 0;
    uint16_t vt_len = 0;
    Table() = default;
    Table(const Span& buf, size_t at) : b(buf), pos(at) {
        const int32_t so = rd<int32_t>(b, pos);
        const int64_t v = static_cast<int64_t>(pos) - so;
        if (v < 0 || static_cast<size_t>(v) > b.n || b.n - static_cast<size_t>(v) < 4) bad("vtable");
        vt = static_cast<size_t>(v);
        vt_len = rd<uint16_t>(b, vt);
        if (b.n - vt < vt_len) bad("vtable length");
    }
    bool valid() const { return b.p != nullptr; }
    size_t field_offset(int field) const {
        const size_t o = 4 + 2 * static_cast<size_t>(field);
        if (o + 2 > vt_len) return 0;
        return rd<uint16_t>(b, vt + o);
    }
    template <typename T>
    T scalar(int field, T dflt) const {
        const size_t o = field_offset(field);
        return o ? rd<T>(b, pos + o) : dflt;
    }
    size_t indirect(int field) const {  // absolute position of the referenced object, 0 = absent
        const size_t o = field_offset(field);
        if (!o) return 0;
        const size_t p = pos + o;
        return p + rd<uint32_t>(b, p);
    }
    Table table(int field) const {
        const size_t p = indirect(field);
        return p ? Table(b, p) : Table();
    }
    std::string string(int field) const {
        const size_t p = indirect(field);
        if (!p) return {};
        const uint32_t n = rd<uint32_t>(b, p);
        if (b.n - p - 4 < n) bad("string");   // rd<> above proved p + 4 <= b.n
        return std::string(reinterpret_cast<const char*>(b.p + p + 4), n);
    }
    /** vector of `elem` byte scalars / structs: pointer + count */
    Span vector(int field, size_t elem, size_t* count) const {
        *count = 0;
        const size_t p = indirect(field);
        if (!p) return {};
        const uint32_t n = rd<uint32_t>(b, p);
        if ((b.n - p - 4) / elem < n) bad("vector");
        *count = n;
        return Span{b.p + p + 4, static_cast<size_t>(n) * elem};
    }
    std::vector<Table> tables(int field) const {
        std::vector<Table> out;
        const size_t p = indirect(field);
        if (!p) return out;
        const uint32_t n = rd<uint32_t>(b, p);
        if ((b.n - p - 4) / 4 < n) bad("table vector");
        for (uint32_t i = 0; i < n; ++i) {
            const size_t e = p + 4 + 4ull * i;
            out.emplace_back(b, e + rd<uint32_t>(b, e));
        }
        return out;
    }
};

// [u32 size][flatbuffer: u32 root offset, 4-byte identifier, ...][u32 crc32] (fb_utils.cpp:60-110)
Table prefixed_root(const Span& file, size_t pos, size_t* body_size) {
    const uint32_t size = rd<uint32_t>(file, pos);
    if (file.n - pos - 4 < size) bad("block size");
    if (body_size) *body_size = size;
    return Table(file, pos + 4 + rd<uint32_t>(file, pos + 4));
}

bool block_crc_ok(const Span& file, size_t pos) {
    if (pos > file.n || file.n - pos < 8) return false;
    const uint32_t size = rd<uint32_t>(file, pos);
    if (file.n - pos - 8 < size) return false;
    const uint32_t stored = rd<uint32_t>(file, pos + 4 + size);
    return static_cast<uint32_t>(::crc32(0L, file.p + pos, 4 + size)) == stored;
}

// CHAN_FIELD enum of ouster_osf/fb/os_sensor/lidar_scan_stream.fbs
std::string chan_field_name(uint8_t v) {
    switch (v) {
        case 1: return ChanField::RANGE;
        case 2: return ChanField::RANGE2;
        case 3: return ChanField::SIGNAL;
        case 4: return ChanField::SIGNAL2;
        case 5: return ChanField::REFLECTIVITY;
        case 6: return ChanField::REFLECTIVITY2;
        case 7: return ChanField::NEAR_IR;
        case 8: return ChanField::FLAGS;
        case 9: return ChanField::FLAGS2;
        case 40: return "RAW_HEADERS";
        default: break;
    }
    if (v >= 45 && v <= 49) return "RAW32_WORD" + std::to_string(v - 40);
    if (v >= 50 && v <= 59) return "CUSTOM" + std::to_string(v - 50);
    if (v >= 60 && v <= 63) return "RAW32_WORD" + std::to_string(v - 59);
    throw std::runtime_error("OSF: unknown channel field " + std::to_string(v));
}

}  // namespace

// ---------------------------------------------------------------------------------------
// OsfFile
// ---------------------------------------------------------------------------------------
OsfFile::OsfFile(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("OSF: cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    buf_.resize(static_cast<size_t>(std::max<std::streamsize>(n, 0)));
    if (n > 0 && !f.read(reinterpret_cast<char*>(buf_.data()), n)) throw std::runtime_error("OSF: cannot read " + path);
    const Span file{buf_.data(), buf_.size()};
    if (buf_.size() < 16 || std::memcmp(buf_.data() + 8, "OSF$", 4) != 0) throw std::runtime_error("OSF: not an OSF file: " + path);
    size_t hsize = 0;
    const Table hdr = prefixed_root(file, 0, &hsize);
    if (!block_crc_ok(file, 0)) throw std::runtime_error("OSF: header crc32 mismatch");
    version_ = hdr.scalar<uint64_t>(0, 0);
    const uint8_t status = hdr.scalar<uint8_t>(1, 0);
    metadata_offset_ = hdr.scalar<uint64_t>(2, 1);
    if (status != 2) throw std::runtime_error("OSF: file was not finished (header status is not VALID)");
    chunks_base_ = 4 + hsize + 4;
    if (metadata_offset_ > buf_.size() || buf_.size() - metadata_offset_ < 8 || !block_crc_ok(file, metadata_offset_))
        throw std::runtime_error("OSF: metadata crc32 mismatch");
    const Table meta = prefixed_root(file, metadata_offset_, nullptr);
    id_ = meta.string(0);
    size_t n_chunks = 0;
    const Span ch = meta.vector(3, 24, &n_chunks);  // struct ChunkOffset {start_ts, end_ts, offset}
    for (size_t i = 0; i < n_chunks; ++i) chunk_offsets_.push_back(rd<uint64_t>(ch, i * 24 + 16));
    for (const Table& e : meta.tables(4)) {
        MetadataEntry me;
        me.id = e.scalar<uint32_t>(0, 0);
        me.type = e.string(1);
        size_t nb = 0;
        const Span b = e.vector(2, 1, &nb);
        me.buffer = b.p;
        me.size = nb;
        entries_.push_back(std::move(me));
    }
}

std::map<uint32_t, std::string> OsfFile::sensor_metadata_json() const {
    std::map<uint32_t, std::string> out;
    for (const auto& e : entries_) {
        if (e.type.size() < 11 || e.type.compare(e.type.size() - 11, 11, "LidarSensor") != 0) continue;
        const Table t = prefixed_root(Span{e.buffer, e.size}, 0, nullptr);
        out[e.id] = t.string(0);
    }
    return out;
}

std::map<uint32_t, uint32_t> OsfFile::lidar_scan_streams() const {
    std::map<uint32_t, uint32_t> out;
    for (const auto& e : entries_) {
        if (e.type.size() < 15 || e.type.compare(e.type.size() - 15, 15, "LidarScanStream") != 0) continue;
        const Table t = prefixed_root(Span{e.buffer, e.size}, 0, nullptr);
        out[e.id] = t.scalar<uint32_t>(0, 0);
    }
    return out;
}

std::vector<OsfFile::Message> OsfFile::messages() const {
    std::vector<Message> out;
    const Span file{buf_.data(), buf_.size()};
    for (uint64_t off : chunk_offsets_) {
        if (off > buf_.size() || buf_.size() - off < chunks_base_ + 8) throw std::runtime_error("OSF: chunk offset out of range");
        const size_t pos = chunks_base_ + off;
        if (!block_crc_ok(file, pos)) throw std::runtime_error("OSF: chunk crc32 mismatch");
        const Table chunk = prefixed_root(file, pos, nullptr);
        for (const Table& m : chunk.tables(0)) {
            Message msg;
            msg.ts = m.scalar<uint64_t>(0, 0);
            msg.id = m.scalar<uint32_t>(1, 0);
            size_t nb = 0;
            const Span b = m.vector(2, 1, &nb);
            msg.buffer = b.p;
            msg.size = nb;
            out.push_back(msg);
        }
    }
    std::stable_sort(out.begin(), out.end(), [](const Message& a, const Message& b) { return a.ts < b.ts; });
    return out;
}

// ---------------------------------------------------------------------------------------
// LidarScanMsg (ouster_osf/fb/os_sensor/lidar_scan_stream.fbs)
// ---------------------------------------------------------------------------------------
LidarScanMsgView LidarScanMsgView::parse(const OsfFile::Message& msg) {
    if (!msg.buffer || msg.size < 8) bad("empty message");
    const Span b{msg.buffer, msg.size};
    const Table t = prefixed_root(b, 0, nullptr);
    LidarScanMsgView v;
    v.frame_id = t.scalar<int32_t>(5, 0);
    v.frame_status = t.scalar<uint64_t>(9, 0);
    v.shutdown_countdown = t.scalar<uint8_t>(10, 0);
    v.shot_limiting_countdown = t.scalar<uint8_t>(11, 0);
    size_t n_types = 0;
    const Span types = t.vector(1, 2, &n_types);  // struct ChannelField {chan_field u8, chan_field_type u8}
    const std::vector<Table> channels = t.tables(0);
    for (size_t i = 0; i < channels.size() && i < n_types; ++i) {
        EncodedField f;
        f.name = chan_field_name(types.p[2 * i]);
        f.type = static_cast<ChanFieldType>(types.p[2 * i + 1]);
        size_t nb = 0;
        const Span d = channels[i].vector(0, 1, &nb);
        f.data = d.p;
        f.size = nb;
        v.fields.push_back(std::move(f));
    }
    for (const Table& cf : t.tables(8)) {   // custom_fields: table Field {name, tag, shape, field_class, data, bytes}
        LidarScanMsgView::CustomField c;
        c.name = cf.string(0);
        c.type = static_cast<ChanFieldType>(cf.scalar<uint8_t>(1, 0));
        size_t nd = 0;
        const Span sh = cf.vector(2, 8, &nd);
        for (size_t i = 0; i < nd; ++i) c.shape.push_back(static_cast<size_t>(rd<uint64_t>(sh, i * 8)));
        const int64_t cls = cf.scalar<int64_t>(3, 0);
        c.field_class = cls >= 0 && cls <= 4 ? static_cast<FieldClass>(cls) : FieldClass::NONE;
        size_t nb = 0;
        const Span d = cf.vector(4, 1, &nb);
        c.data = d.p;
        c.size = nb;
        v.custom_fields.push_back(std::move(c));
    }
    size_t n = 0;
    v.timestamp = reinterpret_cast<const uint64_t*>(t.vector(2, 8, &n).p); v.n_timestamp = n;
    v.measurement_id = reinterpret_cast<const uint16_t*>(t.vector(3, 2, &n).p); v.n_measurement_id = n;
    v.status = reinterpret_cast<const uint32_t*>(t.vector(4, 4, &n).p); v.n_status = n;
    v.pose = reinterpret_cast<const double*>(t.vector(6, 8, &n).p); v.n_pose = n;
    v.packet_timestamp = reinterpret_cast<const uint64_t*>(t.vector(7, 8, &n).p); v.n_packet_timestamp = n;
    v.alert_flags = t.vector(12, 1, &n).p; v.n_alert_flags = n;
    return v;
}

// ---------------------------------------------------------------------------------------
// host half of decode_field: entropy decoding only
// ---------------------------------------------------------------------------------------
namespace {

[[noreturn]] void cannot_decode() { throw std::runtime_error("decodeField: could not decode field"); }

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// PNG scanline filters 0-4 (PNG specification, section 9), in place over the inflated stream
void png_unfilter(const uint8_t* raw, size_t h, size_t stride, size_t bpp, uint8_t* out) {
    const uint8_t* prev = nullptr;
    for (size_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* in = raw + y * (stride + 1) + 1;
        uint8_t* cur = out + y * stride;
        switch (ft) {
            case 0: std::memcpy(cur, in, stride); break;
            case 1:
                for (size_t x = 0; x < stride; ++x) cur[x] = static_cast<uint8_t>(in[x] + (x >= bpp ? cur[x - bpp] : 0));
                break;
            case 2:
                for (size_t x = 0; x < stride; ++x) cur[x] = static_cast<uint8_t>(in[x] + (prev ? prev[x] : 0));
                break;
            case 3:
                for (size_t x = 0; x < stride; ++x) {
                    const unsigned a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0;
                    cur[x] = static_cast<uint8_t>(in[x] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (size_t x = 0; x < stride; ++x) {
                    const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0,
                              c = (prev && x >= bpp) ? prev[x - bpp] : 0;
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    cur[x] = static_cast<uint8_t>(in[x] + pred);
                }
                break;
            default: cannot_decode();
        }
        prev = cur;
    }
}

}  // namespace

namespace {

// What the header of one encoded field says, before any entropy decoding: how many bytes its staged form takes, so that a
// whole batch can be laid out first and every field inflated straight to its place in ONE pinned staging buffer (round 5:
// the staged bytes used to go through a vector per field, a second vector for the batch and a pageable copy to the GPU --
// three passes over ~0.9 MB per frame that cost more than the inflate itself).
struct FieldPlan {
    uint32_t encoding = 0, src_pixel_bytes = 0;
    bool filtered = false;
    size_t staged_bytes = 0;
    size_t h = 0, stride = 0, bpp = 0;                          // PNG
    std::vector<std::pair<const uint8_t*, size_t>> chunks;      // PNG: the IDAT bodies; ZPNG: the zstd frame
};

FieldPlan plan_field(const EncodedField& f, size_t h, size_t w, bool device_unfilter) {
    FieldPlan out;
    const size_t esz = field_type_size(f.type);
    if (esz != 1 && esz != 2 && esz != 4 && esz != 8) cannot_decode();
    // ZPNG first, as decode_field does (png_tools.cpp:688-706)
    if (f.size >= 8 && f.data[0] == 0xF8 && f.data[1] == 0xFB) {
        const size_t zw = f.data[2] | (f.data[3] << 8), zh = f.data[4] | (f.data[5] << 8);
        const size_t pixel_bytes = static_cast<size_t>(f.data[6]) * f.data[7];
        if (zw != w || zh != h || pixel_bytes != esz) throw std::runtime_error("Invalid allocation");
        out.encoding = OUSTER_HIP_OSF_ZPNG;
        out.src_pixel_bytes = static_cast<uint32_t>(pixel_bytes);
        out.staged_bytes = w * h * pixel_bytes;
        out.chunks.emplace_back(f.data + 8, f.size - 8);
        return out;
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (f.size < 8 || std::memcmp(f.data, sig, 8) != 0) cannot_decode();
    size_t pos = 8;
    uint32_t pw = 0, ph = 0;
    int depth = 0, colour = -1, interlace = 0;
    while (pos + 12 <= f.size) {
        const uint32_t n = be32(f.data + pos);
        const uint8_t* typ = f.data + pos + 4;
        if (pos + 12 + n > f.size) cannot_decode();
        const uint8_t* body = f.data + pos + 8;
        if (!std::memcmp(typ, "IHDR", 4) && n >= 13) {
            pw = be32(body);
            ph = be32(body + 4);
            depth = body[8];
            colour = body[9];
            interlace = body[12];
        } else if (!std::memcmp(typ, "IDAT", 4)) {
            out.chunks.emplace_back(body, n);
        } else if (!std::memcmp(typ, "IEND", 4)) {
            break;
        }
        pos += 12 + n;
    }
    if (pw != w || ph != h || interlace) cannot_decode();
    size_t bpp = 0;
    if (colour == 0 && depth == 8) { out.encoding = OUSTER_HIP_OSF_PNG_GRAY8; bpp = 1; }
    else if (colour == 0 && depth == 16) { out.encoding = OUSTER_HIP_OSF_PNG_GRAY16; bpp = 2; }
    else if (colour == 2 && depth == 8) { out.encoding = OUSTER_HIP_OSF_PNG_RGB8; bpp = 3; }
    else if (colour == 6 && depth == 8) { out.encoding = OUSTER_HIP_OSF_PNG_RGBA8; bpp = 4; }
    else if (colour == 6 && depth == 16) { out.encoding = OUSTER_HIP_OSF_PNG_RGBA16; bpp = 8; }
    else cannot_decode();
    // the reference picks the decoder by the FIELD's type and rejects a PNG of another depth
    // (decode_{8,16,32,64}bit_image: "bad sample depth / color type", png_tools.cpp:232-250, 623-643);
    // 24-bit RGB is what encode_24bit_image writes for 32-bit fields
    const bool ok = (esz == 1 && bpp == 1) || (esz == 2 && bpp == 2) || (esz == 4 && (bpp == 4 || bpp == 3)) ||
                    (esz == 8 && bpp == 8);
    if (!ok) cannot_decode();
    out.src_pixel_bytes = static_cast<uint32_t>(bpp);
    out.h = h;
    out.bpp = bpp;
    out.stride = w * bpp;
    // the GPU reverses the filters (k_osf_png_unfilter) -- unless a scanline is too long for its LDS (64 per-lane rings of 192
    // pixels, sized by the widest pixel a batch may hold, 8 bytes, plus one row: 160 KB per workgroup, i.e. rows up to ~61 KB);
    // such a field is unfiltered on the host as in rounds 2 - 4 (it used to fail in the launch)
    const size_t device_lds = 64 * (size_t{192} * 8 + 16) + ((out.stride + 15) & ~size_t{15});
    if (device_unfilter && device_lds > size_t{160} * 1024) device_unfilter = false;
    out.filtered = device_unfilter;
    out.staged_bytes = device_unfilter ? h * (out.stride + 1) : h * out.stride;
    return out;
}

// inflate a zlib stream that may be split over several IDAT chunks into exactly `want` bytes
void inflate_chunks(const std::vector<std::pair<const uint8_t*, size_t>>& chunks, uint8_t* dst, size_t want) {
    z_stream z{};
    if (inflateInit(&z) != Z_OK) cannot_decode();
    z.next_out = dst;
    z.avail_out = static_cast<uInt>(want);
    int rc = Z_OK;
    for (size_t i = 0; i < chunks.size() && rc == Z_OK; ++i) {
        z.next_in = const_cast<Bytef*>(chunks[i].first);
        z.avail_in = static_cast<uInt>(chunks[i].second);
        while (z.avail_in && rc == Z_OK) rc = inflate(&z, Z_NO_FLUSH);   // more pixels than the header said: Z_BUF_ERROR
    }
    const bool ok = rc == Z_STREAM_END && z.total_out == want;
    inflateEnd(&z);
    if (!ok) cannot_decode();
}

// entropy decoding of one planned field into its staged_bytes at dst
void stage_into(const FieldPlan& p, uint8_t* dst) {
    if (p.encoding == OUSTER_HIP_OSF_ZPNG) {
        const size_t n = ZSTD_decompress(dst, p.staged_bytes, p.chunks[0].first, p.chunks[0].second);
        if (ZSTD_isError(n) || n != p.staged_bytes) cannot_decode();
        return;
    }
    if (p.filtered) {   // what libpng would refuse is refused here
        inflate_chunks(p.chunks, dst, p.staged_bytes);
        for (size_t y = 0; y < p.h; ++y)
            if (dst[y * (p.stride + 1)] > 4) cannot_decode();
        return;
    }
    std::vector<uint8_t> raw(p.h * (p.stride + 1));
    inflate_chunks(p.chunks, raw.data(), raw.size());
    png_unfilter(raw.data(), p.h, p.stride, p.bpp, dst);
}

// A few parked threads for the entropy decoding (zlib and zstd are re-entrant; every field is independent).  Started on first
// use and kept: spawning 128 threads per batch cost about as much as one round of inflates.
class Crew {
   public:
    ~Crew() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& t : th_) t.join();
    }
    // fn(i) for i in [0, n) on up to `want` threads; fn must not throw.  The caller is one of them -- or, given `meanwhile`,
    // runs that instead while the others work (it may wait for items to finish: they are handed out in index order)
    void run(size_t n, size_t want, const std::function<void(size_t)>& fn, const std::function<void()>* meanwhile = nullptr) {
        if (want <= 1 || n <= 1) {
            for (size_t i = 0; i < n; ++i) fn(i);
            if (meanwhile) (*meanwhile)();
            return;
        }
        std::unique_lock<std::mutex> lk(mu_);
        while (th_.size() + 1 < want) th_.emplace_back([this] { worker(); });
        fn_ = &fn;
        n_ = n;
        next_.store(0);
        active_ = th_.size();
        ++gen_;
        lk.unlock();
        wake_.notify_all();
        if (meanwhile) (*meanwhile)();
        else
            for (size_t i; (i = next_.fetch_add(1)) < n;) fn(i);
        lk.lock();
        done_.wait(lk, [&] { return active_ == 0; });
    }

   private:
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            wake_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            const std::function<void(size_t)>* fn = fn_;
            const size_t n = n_;
            lk.unlock();
            for (size_t i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
            lk.lock();
            if (--active_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// host memory the copy engine reads and writes directly; grows, never shrinks
struct PinnedBytes {
    uint8_t* p = nullptr;
    size_t cap = 0;
    PinnedBytes() = default;
    PinnedBytes(const PinnedBytes&) = delete;
    PinnedBytes& operator=(const PinnedBytes&) = delete;
    ~PinnedBytes() {
        if (p) (void)hipHostFree(p);
    }
    uint8_t* need(size_t n) {
        if (n > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = n + n / 4;
            void* q = nullptr;
            if (hipHostMalloc(&q, want, hipHostMallocPortable) != hipSuccess) throw std::runtime_error("ouster_hip: hipHostMalloc failed");
            p = static_cast<uint8_t*>(q);
            cap = want;
        }
        return p;
    }
};

}  // namespace

StagedField stage_field(const EncodedField& f, size_t h, size_t w, bool device_unfilter) {
    const FieldPlan p = plan_field(f, h, w, device_unfilter);
    StagedField out;
    out.encoding = p.encoding;
    out.src_pixel_bytes = p.src_pixel_bytes;
    out.filtered = p.filtered;
    out.bytes.resize(p.staged_bytes);
    stage_into(p, out.bytes.data());
    return out;
}

// Where every planned field goes in the staging buffer, biggest first (256-byte aligned): the crew takes the fields in that
// order -- a 24-bit range image inflates three times as long as an 8-bit plane, and handed out as they come the last thread
// to finish decides the batch -- and the buffer fills front to back, so that finished pieces can be uploaded while the rest
// is inflated.  Returns the order; off[i] = offset of field i; total = bytes.
static std::vector<size_t> lay_out(const std::vector<FieldPlan>& plans, std::vector<size_t>& off, size_t& total) {
    std::vector<size_t> order(plans.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return plans[a].staged_bytes > plans[b].staged_bytes; });
    off.assign(plans.size(), 0);
    total = 0;
    for (size_t i : order) {
        off[i] = total;
        total += (plans[i].staged_bytes + 255) & ~size_t{255};
    }
    return order;
}

// ---------------------------------------------------------------------------------------
// OsfFrameDecoder
// ---------------------------------------------------------------------------------------
struct OsfFrameDecoder::Impl {
    SensorInfo info;
    std::shared_ptr<hip::Context> ctx;
    int device = -1;
    bool device_unfilter = true;   // PNG scanline filters on the GPU (round 5); false: on the host, as in rounds 2 - 4
    hip::DeviceBuffer d_src, d_dst;
    PinnedBytes h_src, h_dst;
    Crew crew;
    // Every planned field inflated to base + off[i] by the crew while this thread sends what is finished to dev + off[i]: the
    // fields are handed out in lay_out's order, so the batch is uploaded as (up to) eight consecutive pieces, each as soon as
    // its last field is there -- the copy engine works under the inflate instead of behind it.  Stream-ordered: the caller's next
    // launch on `st` sees all of it.  The first exception wins and is rethrown.
    void stage_and_upload(const std::vector<FieldPlan>& plans, const std::vector<size_t>& order, const std::vector<size_t>& off,
                          size_t total, uint8_t* base, void* dev, hipStream_t st) {
        const size_t n = plans.size(), G = std::min<size_t>(8, n);
        if (!n) return;
        // `base` is this decoder's reused pinned buffer: a copy queued from it by an earlier call that ended in an exception
        // (after its uploads, before the synchronisation at the end of the unpack) may still be reading it
        if (hipStreamSynchronize(st) != hipSuccess) throw std::runtime_error("ouster_hip: hipStreamSynchronize failed");
        std::vector<uint32_t> grp(n);
        std::vector<size_t> first(G + 1);
        std::unique_ptr<std::atomic<uint32_t>[]> left(new std::atomic<uint32_t>[G]);
        for (size_t g = 0; g <= G; ++g) first[g] = g * n / G;
        for (size_t g = 0; g < G; ++g) {
            left[g].store(static_cast<uint32_t>(first[g + 1] - first[g]));
            for (size_t i = first[g]; i < first[g + 1]; ++i) grp[i] = static_cast<uint32_t>(g);
        }
        std::exception_ptr err;
        std::mutex mu;
        hipError_t copy_err = hipSuccess;
        const std::function<void(size_t)> work = [&](size_t i) {   // i: position in the order
            try {
                stage_into(plans[order[i]], base + off[order[i]]);
            } catch (...) {
                std::lock_guard<std::mutex> lock(mu);
                if (!err) err = std::current_exception();
            }
            left[grp[i]].fetch_sub(1, std::memory_order_release);
        };
        const std::function<void()> uploads = [&] {
            for (size_t g = 0; g < G; ++g) {
                while (left[g].load(std::memory_order_acquire) != 0) std::this_thread::yield();
                const size_t b0 = off[order[first[g]]], b1 = g + 1 < G ? off[order[first[g + 1]]] : total;
                const hipError_t e = hipMemcpyAsync(static_cast<uint8_t*>(dev) + b0, base + b0, b1 - b0, hipMemcpyHostToDevice, st);
                if (e != hipSuccess) copy_err = e;
            }
        };
        const size_t nt = std::min<size_t>({n, std::max(1u, std::thread::hardware_concurrency()), size_t{128}});
        crew.run(n, nt + 1, work, &uploads);
        if (err) std::rethrow_exception(err);
        if (copy_err != hipSuccess) throw std::runtime_error("ouster_hip: upload failed");
    }
    const std::shared_ptr<hip::Context>& context() {
        if (!ctx) ctx = std::make_shared<hip::Context>(device >= 0 ? device : hip::current_device());
        return ctx;
    }
};

void OsfFrameDecoder::set_device_unfilter(bool on) { impl_->device_unfilter = on; }
bool OsfFrameDecoder::device_unfilter() const { return impl_->device_unfilter; }

OsfFrameDecoder::OsfFrameDecoder(const SensorInfo& info, int device) : impl_(new Impl) {
    impl_->info = info;
    impl_->device = device;
}
OsfFrameDecoder::~OsfFrameDecoder() = default;

std::vector<LidarFrame> OsfFrameDecoder::decode(const std::vector<OsfFile::Message>& msgs) {
    Impl& s = *impl_;
    const size_t h = s.info.format.pixels_per_column, w = s.info.format.columns_per_frame;
    std::vector<LidarFrame> frames;
    struct Job { size_t frame; std::string name; size_t esz; size_t dst_off; };
    struct CustomJob { size_t frame; std::string name; size_t esz, rows, cols; EncodedField enc; };
    std::vector<Job> jobs;
    std::vector<CustomJob> custom;
    std::vector<EncodedField> encoded;   // parallel to jobs
    size_t src_total = 0, dst_total = 0;
    auto al = [](size_t x) { return (x + 255) & ~size_t{255}; };
    for (const auto& m : msgs) {
        const LidarScanMsgView v = LidarScanMsgView::parse(m);
        LidarFrameFieldTypes types;
        for (const auto& f : v.fields) types.emplace_back(f.name, f.type, std::vector<size_t>{}, FieldClass::PIXEL_FIELD);
        LidarFrame fr(h, w, types, s.info.format.columns_per_packet);
        fr.frame_id = v.frame_id;
        fr.frame_status = v.frame_status;
        fr.shutdown_countdown = v.shutdown_countdown;
        fr.shot_limiting_countdown = v.shot_limiting_countdown;
        auto header = [&](const char* what, size_t have, size_t want) {  // stream_lidar_frame.cpp:213-301
            if (have == want) return true;
            if (have != 0)
                throw std::runtime_error(std::string("OSF: LidarScanMsg has ") + what + " of length " +
                                         std::to_string(have) + ", expected " + std::to_string(want));
            return false;
        };
        if (header("header_timestamp", v.n_timestamp, w)) std::memcpy(fr.timestamp().data(), v.timestamp, w * 8);
        if (header("header_measurement_id", v.n_measurement_id, w))
            std::memcpy(fr.measurement_id().data(), v.measurement_id, w * 2);
        if (header("header_status", v.n_status, w)) std::memcpy(fr.status().data(), v.status, w * 4);
        if (header("pose", v.n_pose, fr.body_to_world().size()))
            std::memcpy(fr.body_to_world().get<double>(), v.pose, v.n_pose * 8);
        if (header("packet_timestamp", v.n_packet_timestamp, fr.packet_timestamp().size()))
            std::memcpy(fr.packet_timestamp().data(), v.packet_timestamp, v.n_packet_timestamp * 8);
        if (header("alert_flags", v.n_alert_flags, fr.alert_flags().size()))
            std::memcpy(fr.alert_flags().data(), v.alert_flags, v.n_alert_flags);
        for (const auto& f : v.fields) {
            Job j;
            j.frame = frames.size();
            j.name = f.name;
            j.esz = field_type_size(f.type);
            if (f.size == 0) continue;  // empty field: stays zero
            encoded.push_back(f);
            jobs.push_back(std::move(j));
        }
        // custom fields (fb_restore_fields, fb_common.cpp:250-330): added with their own shape and class; 1-D fields are
        // raw bytes, everything else is an image of rows x (size / rows) that was never destaggered
        for (const auto& c : v.custom_fields) {
            const size_t esz = field_type_size(c.type);
            if (esz == 0 || c.shape.empty()) continue;   // unsupported element type (a newer writer): skipped like the reference
            size_t lead = 0;
            switch (c.field_class) {
                case FieldClass::PIXEL_FIELD: lead = 2; break;
                case FieldClass::COLUMN_FIELD:
                case FieldClass::PACKET_FIELD: lead = 1; break;
                default: break;
            }
            if (c.shape.size() < lead) throw std::runtime_error("OSF: custom field '" + c.name + "' has too few dimensions for its class");
            Field& dst = fr.add_field(FieldType(c.name, c.type, std::vector<size_t>(c.shape.begin() + lead, c.shape.end()), c.field_class));
            if (dst.shape() != c.shape)
                throw std::runtime_error("OSF: custom field '" + c.name + "' does not have the frame's dimensions");
            if (c.shape.size() == 1) {
                if (c.size > dst.bytes()) throw std::runtime_error("OSF: custom field '" + c.name + "' holds more bytes than its shape");
                std::memcpy(dst.get(), c.data, c.size);
                continue;
            }
            if (dst.bytes() == 0 || c.size == 0) continue;
            CustomJob cj;
            cj.frame = frames.size();
            cj.name = c.name;
            cj.esz = esz;
            cj.rows = c.shape[0];
            cj.cols = dst.size() / c.shape[0];
            cj.enc.name = c.name;
            cj.enc.type = c.type;
            cj.enc.data = c.data;
            cj.enc.size = c.size;
            custom.push_back(std::move(cj));
        }
        frames.push_back(std::move(fr));
    }
    if (!custom.empty()) {
        // one no-stagger unpack launch per distinct image shape (custom fields are few; shapes rarely differ)
        hip::ScopedContext on_my_context(s.context());
        std::vector<bool> done(custom.size(), false);
        for (size_t i0 = 0; i0 < custom.size(); ++i0) {
            if (done[i0]) continue;
            std::vector<size_t> grp;
            for (size_t i = i0; i < custom.size(); ++i)
                if (!done[i] && custom[i].rows == custom[i0].rows && custom[i].cols == custom[i0].cols) { grp.push_back(i); done[i] = true; }
            const size_t rows = custom[i0].rows, cols = custom[i0].cols;
            std::vector<StagedField> st(grp.size());
            std::vector<size_t> so(grp.size()), dof(grp.size());
            size_t stot = 0, dtot = 0;
            for (size_t k = 0; k < grp.size(); ++k) {
                st[k] = stage_field(custom[grp[k]].enc, rows, cols, s.device_unfilter && rows * cols > 0);
                so[k] = stot; dof[k] = dtot;
                stot += al(st[k].bytes.size());
                dtot += al(rows * cols * custom[grp[k]].esz);
            }
            hip::DeviceBuffer dsrc(stot), ddst(dtot);
            std::vector<uint8_t> stage(stot);
            std::vector<ouster_hip_osf_plane> pl(grp.size());
            for (size_t k = 0; k < grp.size(); ++k) {
                std::memcpy(stage.data() + so[k], st[k].bytes.data(), st[k].bytes.size());
                pl[k].src = static_cast<const uint8_t*>(dsrc.data()) + so[k];
                pl[k].dst = static_cast<uint8_t*>(ddst.data()) + dof[k];
                pl[k].encoding = st[k].encoding;
                pl[k].src_pixel_bytes = st[k].src_pixel_bytes;
                pl[k].dst_elem_size = static_cast<uint32_t>(custom[grp[k]].esz);
                pl[k].flags = st[k].filtered ? OUSTER_HIP_OSF_FLAG_FILTERED : 0u;
            }
            dsrc.upload(stage.data(), stot);
            hip::check(ouster_hip_osf_unpack(s.ctx->handle(), pl.data(), static_cast<uint32_t>(pl.size()),
                                             static_cast<uint32_t>(rows), static_cast<uint32_t>(cols), nullptr));
            std::vector<uint8_t> back(dtot);
            ddst.download(back.data(), dtot);
            for (size_t k = 0; k < grp.size(); ++k)
                std::memcpy(frames[custom[grp[k]].frame].field(custom[grp[k]].name).get(), back.data() + dof[k],
                            rows * cols * custom[grp[k]].esz);
        }
    }
    if (jobs.empty()) return frames;
    std::vector<FieldPlan> plans(jobs.size());
    std::vector<size_t> src_off;
    for (size_t i = 0; i < jobs.size(); ++i) {
        plans[i] = plan_field(encoded[i], h, w, s.device_unfilter);
        jobs[i].dst_off = dst_total;
        dst_total += al(h * w * jobs[i].esz);
    }
    const std::vector<size_t> order = lay_out(plans, src_off, src_total);

    hip::ScopedContext on_my_context(s.context());
    s.d_src.resize(src_total);
    s.d_dst.resize(dst_total);
    // one pinned staging buffer every field is inflated into, one copy in; one launch over every (frame, field); one copy out
    s.stage_and_upload(plans, order, src_off, src_total, s.h_src.need(src_total), s.d_src.data(), static_cast<hipStream_t>(s.ctx->stream()));
    std::vector<ouster_hip_osf_plane> planes(jobs.size());
    bool any_png = false;
    for (size_t i = 0; i < jobs.size(); ++i) {
        planes[i].src = static_cast<const uint8_t*>(s.d_src.data()) + src_off[i];
        planes[i].dst = static_cast<uint8_t*>(s.d_dst.data()) + jobs[i].dst_off;
        planes[i].encoding = plans[i].encoding;
        planes[i].src_pixel_bytes = plans[i].src_pixel_bytes;
        planes[i].dst_elem_size = static_cast<uint32_t>(jobs[i].esz);
        planes[i].flags = plans[i].filtered ? OUSTER_HIP_OSF_FLAG_FILTERED : 0u;
        any_png |= plans[i].encoding != OUSTER_HIP_OSF_ZPNG;
    }
    std::vector<int32_t> shifts(s.info.format.pixel_shift_by_row.begin(), s.info.format.pixel_shift_by_row.end());
    if (any_png && !shifts.empty() && shifts.size() != h)
        throw std::invalid_argument("image height does not match shifts size");
    hip::check(ouster_hip_osf_unpack(s.ctx->handle(), planes.data(), static_cast<uint32_t>(planes.size()),
                                     static_cast<uint32_t>(h), static_cast<uint32_t>(w),
                                     shifts.empty() ? nullptr : shifts.data()));
    const uint8_t* host = s.h_dst.need(dst_total);
    s.d_dst.download(s.h_dst.p, dst_total);
    // into the frames' own planes, on the crew: ~1 MB per frame from one thread was two thirds of this function's time
    std::vector<void*> dst(jobs.size());
    for (size_t i = 0; i < jobs.size(); ++i) dst[i] = frames[jobs[i].frame].field(jobs[i].name).get();
    s.crew.run(jobs.size(), std::min<size_t>({jobs.size(), std::max(1u, std::thread::hardware_concurrency()), size_t{32}}),
               [&](size_t i) { std::memcpy(dst[i], host + jobs[i].dst_off, h * w * jobs[i].esz); });
    return frames;
}

std::vector<std::vector<uint8_t>> OsfFrameDecoder::decode_fields(const std::vector<EncodedField>& fields) {
    Impl& s = *impl_;
    const size_t h = s.info.format.pixels_per_column, w = s.info.format.columns_per_frame;
    auto al = [](size_t x) { return (x + 255) & ~size_t{255}; };
    std::vector<std::vector<uint8_t>> out(fields.size());
    std::vector<FieldPlan> plans;        // the fields that carry data, in order
    std::vector<size_t> src_off, dst_off, idx;
    std::vector<ouster_hip_osf_plane> planes;
    size_t src_total = 0, dst_total = 0;
    for (size_t i = 0; i < fields.size(); ++i) {
        const size_t esz = field_type_size(fields[i].type);
        out[i].assign(h * w * esz, 0);
        if (fields[i].size == 0) continue;
        plans.push_back(plan_field(fields[i], h, w, s.device_unfilter));
        idx.push_back(i);
        dst_off.push_back(dst_total);
        dst_total += al(h * w * esz);
    }
    if (plans.empty()) return out;
    const std::vector<size_t> order = lay_out(plans, src_off, src_total);
    if (!src_total) return out;
    hip::ScopedContext on_my_context(s.context());
    s.d_src.resize(src_total);
    s.d_dst.resize(dst_total);
    s.stage_and_upload(plans, order, src_off, src_total, s.h_src.need(src_total), s.d_src.data(), static_cast<hipStream_t>(s.ctx->stream()));
    bool any_png = false;
    for (size_t k = 0; k < plans.size(); ++k) {
        ouster_hip_osf_plane pl{};
        pl.src = static_cast<const uint8_t*>(s.d_src.data()) + src_off[k];
        pl.dst = static_cast<uint8_t*>(s.d_dst.data()) + dst_off[k];
        pl.encoding = plans[k].encoding;
        pl.src_pixel_bytes = plans[k].src_pixel_bytes;
        pl.dst_elem_size = static_cast<uint32_t>(field_type_size(fields[idx[k]].type));
        pl.flags = plans[k].filtered ? OUSTER_HIP_OSF_FLAG_FILTERED : 0u;
        planes.push_back(pl);
        any_png |= plans[k].encoding != OUSTER_HIP_OSF_ZPNG;
    }
    std::vector<int32_t> shifts(s.info.format.pixel_shift_by_row.begin(), s.info.format.pixel_shift_by_row.end());
    if (any_png && !shifts.empty() && shifts.size() != h)
        throw std::invalid_argument("image height does not match shifts size");
    hip::check(ouster_hip_osf_unpack(s.ctx->handle(), planes.data(), static_cast<uint32_t>(planes.size()),
                                     static_cast<uint32_t>(h), static_cast<uint32_t>(w),
                                     shifts.empty() ? nullptr : shifts.data()));
    const uint8_t* host = s.h_dst.need(dst_total);
    s.d_dst.download(s.h_dst.p, dst_total);
    for (size_t k = 0; k < plans.size(); ++k) std::memcpy(out[idx[k]].data(), host + dst_off[k], out[idx[k]].size());
    return out;
}

// ---------------------------------------------------------------------------------------
// device-resident batches
// ---------------------------------------------------------------------------------------
namespace {
std::shared_ptr<void> device_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) throw std::runtime_error("ouster_hip: hipMalloc failed");
    return std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
}
}  // namespace

OsfDeviceBatch OsfFrameDecoder::decode_device(const std::vector<OsfFile::Message>& msgs) {
    Impl& s = *impl_;
    OsfDeviceBatch b;
    b.info_ = s.info;
    b.h_ = s.info.format.pixels_per_column;
    b.w_ = s.info.format.columns_per_frame;
    b.n_ = static_cast<uint32_t>(msgs.size());
    if (msgs.empty()) return b;
    const size_t h = b.h_, w = b.w_, n = msgs.size();
    auto al = [](size_t x) { return (x + 255) & ~size_t{255}; };
    struct Job { size_t frame, field; };
    std::vector<Job> jobs;
    std::vector<EncodedField> encoded;   // parallel to jobs
    size_t src_total = 0;
    for (size_t m = 0; m < n; ++m) {
        const LidarScanMsgView v = LidarScanMsgView::parse(msgs[m]);
        if (m == 0) {
            for (const auto& f : v.fields) b.fields_.emplace_back(f.name, f.type);
        } else {
            bool same = v.fields.size() == b.fields_.size();
            for (size_t i = 0; same && i < v.fields.size(); ++i)
                same = v.fields[i].name == b.fields_[i].first && v.fields[i].type == b.fields_[i].second;
            if (!same) throw std::invalid_argument("OsfFrameDecoder::decode_device: messages carry different fields");
        }
        b.frame_ids_.push_back(v.frame_id);
        b.ts_.resize((m + 1) * w, 0);
        b.status_.resize((m + 1) * w, 0);
        if (v.n_timestamp == w) std::memcpy(b.ts_.data() + m * w, v.timestamp, w * 8);
        if (v.n_status == w) std::memcpy(b.status_.data() + m * w, v.status, w * 4);
        for (size_t i = 0; i < v.fields.size(); ++i) {
            if (v.fields[i].size == 0) continue;
            encoded.push_back(v.fields[i]);
            jobs.push_back(Job{m, i});
        }
    }
    std::vector<FieldPlan> plans(jobs.size());
    std::vector<size_t> src_off;
    for (size_t i = 0; i < jobs.size(); ++i) plans[i] = plan_field(encoded[i], h, w, s.device_unfilter);
    const std::vector<size_t> order = lay_out(plans, src_off, src_total);
    hip::ScopedContext on_my_context(s.context());
    b.ctx_ = s.ctx;
    size_t dst_total = 0;
    for (const auto& f : b.fields_) {
        b.plane_off_[f.first] = dst_total;
        dst_total += al(n * h * w * field_type_size(f.second));
    }
    b.planes_ = device_alloc(dst_total);
    auto st = static_cast<hipStream_t>(s.ctx->stream());
    if (hipMemsetAsync(b.planes_.get(), 0, dst_total, st) != hipSuccess) throw std::runtime_error("ouster_hip: memset failed");
    if (jobs.empty()) return b;
    s.d_src.resize(src_total);
    s.stage_and_upload(plans, order, src_off, src_total, s.h_src.need(src_total), s.d_src.data(), static_cast<hipStream_t>(s.ctx->stream()));
    std::vector<ouster_hip_osf_plane> planes(jobs.size());
    bool any_png = false;
    for (size_t i = 0; i < jobs.size(); ++i) {
        const auto& f = b.fields_[jobs[i].field];
        const size_t esz = field_type_size(f.second);
        planes[i].src = static_cast<const uint8_t*>(s.d_src.data()) + src_off[i];
        planes[i].dst = static_cast<uint8_t*>(b.planes_.get()) + b.plane_off_[f.first] + jobs[i].frame * h * w * esz;
        planes[i].encoding = plans[i].encoding;
        planes[i].src_pixel_bytes = plans[i].src_pixel_bytes;
        planes[i].dst_elem_size = static_cast<uint32_t>(esz);
        planes[i].flags = plans[i].filtered ? OUSTER_HIP_OSF_FLAG_FILTERED : 0u;
        any_png |= plans[i].encoding != OUSTER_HIP_OSF_ZPNG;
    }
    std::vector<int32_t> shifts(s.info.format.pixel_shift_by_row.begin(), s.info.format.pixel_shift_by_row.end());
    if (any_png && !shifts.empty() && shifts.size() != h)
        throw std::invalid_argument("image height does not match shifts size");
    hip::check(ouster_hip_osf_unpack(s.ctx->handle(), planes.data(), static_cast<uint32_t>(planes.size()),
                                     static_cast<uint32_t>(h), static_cast<uint32_t>(w),
                                     shifts.empty() ? nullptr : shifts.data()));
    return b;
}

void* OsfDeviceBatch::plane_device(const std::string& name) const {
    return static_cast<uint8_t*>(planes_.get()) + plane_off_.at(name);
}

void* OsfDeviceBatch::destagger_device(const std::string& name) {
    hip::ScopedContext on_my_context(ctx_);
    size_t esz = 0;
    for (const auto& f : fields_)
        if (f.first == name) esz = field_type_size(f.second);
    if (!esz) throw std::out_of_range("OsfDeviceBatch: no plane '" + name + "'");
    auto out = device_alloc(static_cast<size_t>(n_) * h_ * w_ * esz);
    std::vector<int32_t> sh(info_.format.pixel_shift_by_row.begin(), info_.format.pixel_shift_by_row.end());
    hip::check(ouster_hip_destagger(ctx_->handle(), plane_device(name), out.get(), static_cast<uint32_t>(h_),
                                    static_cast<uint32_t>(w_), static_cast<uint32_t>(esz), sh.data(),
                                    static_cast<uint32_t>(sh.size()), 0, n_));
    derived_["destaggered:" + name] = out;
    return out.get();
}

void* OsfDeviceBatch::cartesian_device(const XYZLut& lut, bool f64, const std::string& range_field) {
    hip::ScopedContext on_my_context(ctx_);
    bool ok = false;
    for (const auto& f : fields_) ok |= f.first == range_field && f.second == ChanFieldType::UINT32;
    if (!ok) throw std::invalid_argument("OsfDeviceBatch::cartesian_device needs a uint32 plane '" + range_field + "'");
    if (lut.h != h_ || lut.w != w_) throw std::invalid_argument("unexpected image dimensions");
    auto out = device_alloc(static_cast<size_t>(n_) * h_ * w_ * 3 * (f64 ? 8 : 4));
    hip::check(ouster_hip_cartesian(ctx_->handle(), lut.device().handle,
                                    static_cast<const uint32_t*>(plane_device(range_field)), out.get(),
                                    f64 ? OUSTER_HIP_F64 : OUSTER_HIP_F32, n_));
    derived_[std::string("xyz:") + range_field] = out;
    return out.get();
}

void OsfDeviceBatch::download(const void* device_ptr, void* host, size_t bytes) const {
    hip::ScopedContext on_my_context(ctx_);
    auto st = static_cast<hipStream_t>(ctx_->stream());
    if (hipMemcpyAsync(host, device_ptr, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        throw std::runtime_error("ouster_hip: download failed");
}

}  // namespace osf
}  // namespace sdk
}  // namespace ouster
