// launchbench.hip -- what a SMALL launch costs on this box, piece by piece (small batches): microseconds per call,
// calls issued back to back on one stream (pipelined) and call + synchronise, for
//   empty          one empty kernel per call
//   empty x2       two dependent empty kernels per call (the optimistic pass + the fix-up pass of a clean batch)
//   chain D        G workgroups, each a chain of D dependent global loads (pointer chase through a 64 MB table), then one store
//   barrier        G workgroups, a grid-wide barrier (atomic arrive + spin) between two chain-1 phases: what replaces a second launch
//   code          256 workgroups running 1024 x KB one-cycle instructions as KB kilobytes of straight-line code, or as a loop over 1 KB:
//                 what a cold instruction cache costs a kernel whose every CU runs its code once
//   move F         the bytes of F dual-return 128 x 2048 frames (1.05 MB read, 13.9 MB written) by G workgroups, plain streaming
// build: hipcc -O3 --offload-arch=gfx950 tools/launchbench.hip -o tests/cpp/_build/launchbench     usage: launchbench [calls=2000]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            std::exit(1);                                                           \
        }                                                                           \
    } while (0)

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

__global__ void k_empty() {}

// instruction fetch: the same 16384 (x KB / 16) one-cycle instructions as straight-line code of `KB` kilobytes, or as a loop over 1 KB
template <int KB>
__global__ void k_code_straight(uint32_t* out) {
    asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(KB * 256));
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
template <int KB>
__global__ void k_code_loop(uint32_t* out) {
    for (int i = 0; i < KB; ++i) asm volatile(".rept 256\n s_nop 0\n .endr");
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}

__global__ void k_chain(const uint32_t* __restrict__ table, uint32_t mask, int depth, uint32_t* __restrict__ out, uint32_t salt) {
    uint32_t i = (blockIdx.x * 2654435761u + threadIdx.x * 40503u + salt) & mask;
    for (int d = 0; d < depth; ++d) i = table[i] & mask;
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}

__global__ void k_barrier(const uint32_t* __restrict__ table, uint32_t mask, uint32_t* __restrict__ out, uint32_t salt, uint32_t* arrive,
                          uint32_t epoch) {
    uint32_t i = (blockIdx.x * 2654435761u + threadIdx.x * 40503u + salt) & mask;
    i = table[i] & mask;
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t want = epoch * gridDim.x;
        while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    i = table[(i + 1) & mask] & mask;
    out[(gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x] = i;
}

__global__ void k_move(const uint4* __restrict__ src, size_t n_src, uint4* __restrict__ dst, size_t n_dst) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = t; i < n_src; i += stride) {
        const uint4 v = src[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 a = {acc.x, acc.y, acc.z, acc.w};
    for (size_t i = t; i < n_dst; i += stride) __builtin_nontemporal_store(a, reinterpret_cast<v4*>(dst) + i);
}

template <class F>
static void run(const char* name, int calls, hipStream_t st, F&& call, bool first) {
    for (int i = 0; i < 50; ++i) call(i);
    CK(hipStreamSynchronize(st));
    auto t0 = clk::now();
    for (int i = 0; i < calls; ++i) call(i);
    CK(hipStreamSynchronize(st));
    const double pipe = us(t0, clk::now()) / calls;
    std::vector<double> lat;
    for (int i = 0; i < 300; ++i) {
        auto s0 = clk::now();
        call(i);
        CK(hipStreamSynchronize(st));
        lat.push_back(us(s0, clk::now()));
    }
    std::sort(lat.begin(), lat.end());
    std::printf("%s\"%s\": {\"pipelined_us\": %.2f, \"sync_us\": %.2f}", first ? "" : ", ", name, pipe, lat[lat.size() / 2]);
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const int calls = argc > 1 ? std::atoi(argv[1]) : 2000;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const uint32_t n_table = 16u << 20, mask = n_table - 1;
    std::vector<uint32_t> h(n_table);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 4; }
    uint32_t *table, *out, *arrive;
    CK(hipMalloc(&table, n_table * 4));
    CK(hipMemcpy(table, h.data(), n_table * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 4u << 20));
    CK(hipMalloc(&arrive, 4));
    const size_t frame_in = 1052672, frame_out = 13922304;
    uint4 *src, *dst;
    CK(hipMalloc(&src, frame_in * 4));
    CK(hipMalloc(&dst, frame_out * 4));
    CK(hipMemset(src, 1, frame_in * 4));
    std::printf("{");
    run("empty", calls, st, [&](int) { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); }, true);
    run("empty_x2", calls, st, [&](int) { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipLaunchKernelGGL(k_empty, 1, 64, 0, st); }, false);
    run("empty_256wg", calls, st, [&](int) { hipLaunchKernelGGL(k_empty, 256, 256, 0, st); }, false);
    char name[64];
    for (int G : {64, 256})
        for (int D : {0, 1, 2, 3, 5}) {
            std::snprintf(name, sizeof name, "chain_G%d_D%d", G, D);
            run(name, calls, st, [&](int i) { hipLaunchKernelGGL(k_chain, G, 256, 0, st, table, mask, D, out, (uint32_t)i * 977u); }, false);
        }
    for (int G : {64, 256}) {
        CK(hipMemsetAsync(arrive, 0, 4, st));
        CK(hipStreamSynchronize(st));
        uint32_t epoch = 0;
        std::snprintf(name, sizeof name, "barrier_G%d", G);
        run(name, calls, st, [&](int i) { ++epoch; hipLaunchKernelGGL(k_barrier, G, 256, 0, st, table, mask, out, (uint32_t)i * 977u, arrive, epoch); }, false);
        std::snprintf(name, sizeof name, "two_launches_G%d", G);
        run(name, calls, st, [&](int i) {
            hipLaunchKernelGGL(k_chain, G, 256, 0, st, table, mask, 1, out, (uint32_t)i * 977u);
            hipLaunchKernelGGL(k_chain, G, 256, 0, st, table, mask, 1, out + G * 256, (uint32_t)i * 977u);
        }, false);
    }
#define CODE(KB)                                                                                                              \
    std::snprintf(name, sizeof name, "code_straight_%dKB", KB);                                                                   \
    run(name, calls, st, [&](int) { hipLaunchKernelGGL(k_code_straight<KB>, 256, 256, 0, st, out); }, false);                      \
    std::snprintf(name, sizeof name, "code_loop_%dKB", KB);                                                                       \
    run(name, calls, st, [&](int) { hipLaunchKernelGGL(k_code_loop<KB>, 256, 256, 0, st, out); }, false);
    CODE(4) CODE(16) CODE(32) CODE(64)
    for (int F : {1, 4})
        for (int G : {128, 256, 512, 1024}) {
            std::snprintf(name, sizeof name, "move_F%d_G%d", F, G);
            run(name, calls, st, [&](int) { hipLaunchKernelGGL(k_move, G, 256, 0, st, src, frame_in * F / 16, dst, frame_out * F / 16); }, false);
        }
    std::printf("}\n");
    return 0;
}
