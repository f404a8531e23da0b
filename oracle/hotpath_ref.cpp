// TEST INFRASTRUCTURE (oracle/_ref): a timing harness around the REFERENCE's own CPU loops of the hot path -- it holds no
// reference code itself.  It loads the two libraries oracle/Makefile compiles from /root/reference,
//   libdecode_ref.so  PacketFormat::block_field<T,B> + FieldDecodeInfo::get<T>   (parsing.cpp:628-657, field_decode_info.h:41-54)
//   libcore_ref.so    destagger_into<T> + cartesianT<T>                           (impl/lidar_frame_impl.h:733-760, impl/cartesian.h:36-66)
// and runs, per frame of a pool of packet buffers, what the reference's frame-at-a-time caller runs (FrameBatcher's block
// path: block_field of every plane for every packet, lidar_frame.cpp:1492-1528; then destagger of n_dst planes and
// cartesianT<double> of n_xyz range planes) -- on one thread (the reference as it ships), or with the frames of the pool
// spread over OpenMP threads, each with its own first-touched planes / cloud (and, flags & 1, its own LUT and packets).
// bench.py reports the one-thread figure as cpu_baseline.value (kind "reference"); core_benchmark.cpp:29-154 of the reference
// iterates a pool the same way.  Not included: the FrameBatcher's per-packet bookkeeping and the 28 KB of column headers it
// writes per frame (the planes are 3.9 MB).  Never used by the product.
#include <dlfcn.h>
#include <omp.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
using decode_fn = double (*)(const void*, const uint8_t*, size_t, size_t, const char* const*, void* const*, const size_t*, size_t, int, int, int);
using destagger_fn = int (*)(const void*, void*, size_t, size_t, size_t, const int*, size_t, int);
using cartesian_fn = void (*)(double*, const uint32_t*, const double*, const double*, size_t, size_t);
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" {

struct ref_hot_args {
    const void* pf;              // handle of ref_pf_new (libdecode_ref.so)
    const uint8_t* packets;      // [pool_frames][ppf][packet_stride]
    size_t pool_frames, ppf, packet_stride;
    const char* const* names;    // planes decoded per packet
    const size_t* elem;
    size_t n_planes;
    const int* dst_idx;          // planes that are destaggered
    size_t n_dst;
    const int* xyz_idx;          // range planes that are projected
    size_t n_xyz;
    const double* dir;           // [h*w][3]
    const double* ofs;
    size_t h, w;
    const int* shifts;           // [h]
    int block_dim;
    void* const* out_planes;     // optional [n_planes]: thread 0's planes after its last frame (validation)
    double* out_cloud;           // optional [h*w*3]: thread 0's last cloud of xyz_idx[0]
};

// Returns wall seconds for n_frames x reps frames; legs[0..2] = thread 0's seconds in decode / destagger / cartesian.
// -1: a library or symbol is missing.
double ref_bench_hot_path(const char* decode_so, const char* core_so, const ref_hot_args* a, size_t n_frames, int reps, int threads,
                          int flags, double* legs) {
    void* hd = dlopen(decode_so, RTLD_NOW | RTLD_LOCAL);
    void* hc = dlopen(core_so, RTLD_NOW | RTLD_LOCAL);
    if (!hd || !hc) return -1;
    const decode_fn decode = (decode_fn)dlsym(hd, "ref_bench_decode_frame");
    const destagger_fn destagger = (destagger_fn)dlsym(hc, "ref_destagger");
    const cartesian_fn cartesian = (cartesian_fn)dlsym(hc, "ref_cartesian_f64");
    if (!decode || !destagger || !cartesian) return -1;
    if (threads < 1) threads = 1;
    const size_t npx = a->h * a->w, frame_bytes = a->ppf * a->packet_stride;
    double t0 = 0, t1 = 0, l0 = 0, l1 = 0, l2 = 0;
    omp_set_num_threads(threads);
#pragma omp parallel
    {
        std::vector<void*> planes(a->n_planes), dst(a->n_dst);
        for (size_t i = 0; i < a->n_planes; ++i) planes[i] = std::calloc(npx, a->elem[i]);
        for (size_t i = 0; i < a->n_dst; ++i) dst[i] = std::calloc(npx, a->elem[a->dst_idx[i]]);
        double* cloud = (double*)std::calloc(npx * 3, sizeof(double));
        const double *dir = a->dir, *ofs = a->ofs;
        const uint8_t* packets = a->packets;
        void* own[3] = {nullptr, nullptr, nullptr};
        if (flags & 1) {   // this thread's own, first-touched inputs (NUMA-local pages)
            own[0] = std::malloc(npx * 24);
            own[1] = std::malloc(npx * 24);
            own[2] = std::malloc(a->pool_frames * frame_bytes);
            std::memcpy(own[0], a->dir, npx * 24);
            std::memcpy(own[1], a->ofs, npx * 24);
            std::memcpy(own[2], a->packets, a->pool_frames * frame_bytes);
            dir = (const double*)own[0];
            ofs = (const double*)own[1];
            packets = (const uint8_t*)own[2];
        }
        const bool first = omp_get_thread_num() == 0;
        double m0 = 0, m1 = 0, m2 = 0;
#pragma omp barrier
#pragma omp master
        t0 = now();
        for (int rep = 0; rep < reps; ++rep) {
#pragma omp for schedule(static)
            for (size_t f = 0; f < n_frames; ++f) {
                const double s0 = first ? now() : 0;
                decode(a->pf, packets + (f % a->pool_frames) * frame_bytes, a->ppf, a->packet_stride, a->names, planes.data(), a->elem,
                       a->n_planes, (int)a->w, a->block_dim, 1);
                const double s1 = first ? now() : 0;
                for (size_t i = 0; i < a->n_dst; ++i)
                    destagger(planes[a->dst_idx[i]], dst[i], a->h, a->w, a->elem[a->dst_idx[i]], a->shifts, a->h, 0);
                const double s2 = first ? now() : 0;
                for (size_t i = 0; i < a->n_xyz; ++i) cartesian(cloud, (const uint32_t*)planes[a->xyz_idx[a->n_xyz - 1 - i]], dir, ofs, a->h, a->w);
                if (first) {
                    const double s3 = now();
                    m0 += s1 - s0;
                    m1 += s2 - s1;
                    m2 += s3 - s2;
                }
            }
        }
#pragma omp barrier
#pragma omp master
        {
            t1 = now();
            l0 = m0;
            l1 = m1;
            l2 = m2;
            if (a->out_planes)
                for (size_t i = 0; i < a->n_planes; ++i)
                    if (a->out_planes[i]) std::memcpy(a->out_planes[i], planes[i], npx * a->elem[i]);
            if (a->out_cloud) std::memcpy(a->out_cloud, cloud, npx * 24);
        }
        for (void* p : planes) std::free(p);
        for (void* p : dst) std::free(p);
        std::free(cloud);
        for (void* p : own) std::free(p);
    }
    if (legs) {
        legs[0] = l0;
        legs[1] = l1;
        legs[2] = l2;
    }
    return t1 - t0;
}

// cartesianT<double> as the reference parallelises it itself (-DOUSTER_OMP, impl/cartesian.h:15-23,50-52: `#pragma omp parallel
// for schedule(static)` over the points of ONE cloud): `core_omp_so` is libcore_ref.so compiled with -fopenmp -DOUSTER_OMP.
// Returns seconds for `reps` clouds on `threads` OpenMP threads.
double ref_bench_cartesian_omp(const char* core_omp_so, const uint32_t* range, const double* dir, const double* ofs, size_t h, size_t w,
                               int reps, int threads) {
    void* hc = dlopen(core_omp_so, RTLD_NOW | RTLD_LOCAL);
    if (!hc) return -1;
    const cartesian_fn cartesian = (cartesian_fn)dlsym(hc, "ref_cartesian_f64");
    if (!cartesian) return -1;
    omp_set_num_threads(threads < 1 ? 1 : threads);
    std::vector<double> cloud(h * w * 3);
    cartesian(cloud.data(), range, dir, ofs, h, w);   // first touch by the team
    const double t0 = now();
    for (int r = 0; r < reps; ++r) cartesian(cloud.data(), range, dir, ofs, h, w);
    return now() - t0;
}
}
