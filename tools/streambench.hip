// streambench: N concurrent write streams whose bases are spaced S + D apart inside one slab.
// Shows how the relative placement of equally-indexed output planes changes the achieved write rate
// (channel / bank hashing of the physical address).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ __launch_bounds__(256) void k_fillN(u32x4* base, size_t stride16, size_t n16) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (size_t i = i0; i < n16; i += step) {
        const u32x4 v = {(unsigned)i, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < N; ++k) base[k * stride16 + i] = v;
    }
}

// tile-like pattern: a workgroup writes 128 rows x 256 B segments (row pitch 8 KB) of N planes
template <int N>
__global__ __launch_bounds__(256) void k_tileN(u32x4* base, size_t stride16, unsigned frames) {
    const unsigned tile = blockIdx.x % 32, f = blockIdx.x / 32;
    if (f >= frames) return;
    const unsigned q = threadIdx.x % 16, ty = threadIdx.x / 16;
    for (unsigned r = ty; r < 128; r += 16) {
        const size_t i = ((size_t)f * 128 + r) * 512 + tile * 16 + q;  // 16 B units, 8 KB rows
        const u32x4 v = {(unsigned)i, 1u, 2u, 3u};
#pragma unroll
        for (int k = 0; k < N; ++k) base[k * stride16 + i] = v;
    }
}

int main() {
    const size_t S = 256ull << 20;  // bytes per stream
    const int N = 8;
    char* slab;
    const size_t total = (size_t)N * (S + (80ull << 20)) + (64ull << 20);
    if (hipMalloc(&slab, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t gaps[] = {0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 5 << 20, 9 << 20,
                           17 << 20, 33 << 20, 65 << 20, (33 << 20) + 65536, 7 * 4096 + 256};
    printf("slab %p, %d streams of %zu MB\n", (void*)slab, N, S >> 20);
    for (size_t D : gaps) {
        const size_t stride16 = (S + D) / 16;
        float best1 = 1e9, best2 = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_fillN<N>, dim3(8192), dim3(256), 0, 0, (u32x4*)slab, stride16, S / 16);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best1) best1 = ms;
            hipEventRecord(a);
            hipLaunchKernelGGL(k_tileN<N>, dim3(32 * 256), dim3(256), 0, 0, (u32x4*)slab, stride16, 256u);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b); if (ms < best2) best2 = ms;
        }
        printf("gap %10zu B | linear %7.1f GB/s | tile-pattern %7.1f GB/s\n", D, N * (double)S / best1 / 1e6,
               N * 256.0 * 128 * 8192 / best2 / 1e6);
    }
    return 0;
}
