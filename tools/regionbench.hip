// regionbench: is the achievable bandwidth a function of WHERE in the device memory a buffer lies?
// Allocates 4 GiB chunks one after the other (all held, so they walk through the memory) and times, in each:
//   write   plain streaming 16 B stores over 2 GiB, 2048 workgroups
//   read    plain streaming 16 B loads over the same 2 GiB
//   planes  256 "frames" x 4 planes: each workgroup writes 1 KiB row segments of four different 512 MiB planes at an
//           8 KiB pitch, frames spread over the XCDs as the decode kernels do (a skeleton of their store pattern)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/regionbench tools/regionbench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_write(uint4* dst, size_t per_wg16) {
    uint4* p = dst + (size_t)blockIdx.x * per_wg16;
    const uint4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    for (size_t i = threadIdx.x; i < per_wg16; i += 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_read(const uint4* src, size_t per_wg16, uint32_t* sink) {
    const uint4* p = src + (size_t)blockIdx.x * per_wg16;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < per_wg16; i += 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 planes of [256 frames][128 rows][2048 cols] u32 = 4 x 512 MiB... scaled: 128 rows x 2048 cols x 4 B = 1 MiB per frame
__global__ __launch_bounds__(256) void k_planes(uint4* base) {
    const uint32_t xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;       // 32 blocks per frame: 8 column tiles x 4 row chunks
    const uint32_t f = (i / 32) * 8u + xcd, sub = i % 32, tile = sub % 8, rc = sub / 8;
    const uint32_t q = threadIdx.x % 64, ty = threadIdx.x / 64;      // 64 lanes x 16 B = 1 KiB row segment
    const size_t plane16 = (size_t)256 * 128 * 2048 * 4 / 16;        // uint4 per plane (256 MiB... see main)
    for (uint32_t r = rc * 32 + ty; r < rc * 32 + 32; r += 4) {
        const size_t o = ((size_t)f * 128 + r) * (2048 * 4 / 16) + tile * 64 + q;
        const uint4 v = {f, r, tile, q};
#pragma unroll
        for (int p = 0; p < 4; ++p) base[(size_t)p * plane16 + o] = v;
    }
}

int main(int argc, char** argv) {
    const int max_chunks = argc > 1 ? atoi(argv[1]) : 60;
    const size_t chunk = 4ull << 30, span = 1ull << 30;   // planes: 4 x 256 MiB = 1 GiB inside the chunk
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    std::vector<void*> held;
    printf("{\"chunk_GiB\": 4, \"regions\": [\n");
    for (int c = 0; c < max_chunks; ++c) {
        void* p = nullptr;
        if (hipMalloc(&p, chunk) != hipSuccess) { (void)hipGetLastError(); break; }
        held.push_back(p);
        float best[3] = {1e30f, 1e30f, 1e30f};
        for (int mode = 0; mode < 3; ++mode)
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, (uint4*)p, (2ull << 30) / 16 / 2048);
                else if (mode == 1) hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const uint4*)p, (2ull << 30) / 16 / 2048, sink);
                else hipLaunchKernelGGL(k_planes, dim3(256 * 32), dim3(256), 0, 0, (uint4*)p);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep && t < best[mode]) best[mode] = t;
            }
        printf("  {\"at_GiB\": %d, \"write_GBps\": %.0f, \"read_GBps\": %.0f, \"planes_GBps\": %.0f}%s\n", c * 4,
               (2ull << 30) / 1e9 / (best[0] * 1e-3), (2ull << 30) / 1e9 / (best[1] * 1e-3), span / 1e9 / (best[2] * 1e-3), ",");
        fflush(stdout);
    }
    printf("  {}]}\n");
    return 0;
}
