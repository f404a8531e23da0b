// writebench: how does the achieved HBM write rate of MI355X scale with the number of workgroups
// (= compute units) that stream stores?  Every workgroup (256 threads) owns a contiguous slice of a
// 2 GiB buffer and writes it with 16 B stores, 4 KiB per workgroup instruction round; G workgroups
// are persistent (G <= CUs: one per CU; beyond that several share a CU).  If the rate per workgroup
// stays flat while G grows, the limit sits in the CU's own store path (outstanding write requests x
// L2 latency); where it bends, the shared part (L2 / fabric / HBM) takes over.
// Also a read pass (16 B loads, xor-reduced) and a copy pass for the same sweep.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/writebench tools/writebench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_write(uint4* dst, size_t per_wg16) {
    uint4* p = dst + (size_t)blockIdx.x * per_wg16;
    const uint4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    for (size_t i = threadIdx.x; i < per_wg16; i += 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_write_nt(uint4* dst, size_t per_wg16) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4* p = (v4*)dst + (size_t)blockIdx.x * per_wg16;
    const v4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    for (size_t i = threadIdx.x; i < per_wg16; i += 256) __builtin_nontemporal_store(v, p + i);
}
__global__ __launch_bounds__(256) void k_read(const uint4* src, size_t per_wg16, uint32_t* sink) {
    const uint4* p = src + (size_t)blockIdx.x * per_wg16;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i < per_wg16; i += 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_copy(uint4* dst, const uint4* src, size_t per_wg16) {
    const size_t o = (size_t)blockIdx.x * per_wg16;
    for (size_t i = threadIdx.x; i < per_wg16; i += 256) dst[o + i] = src[o + i];
}

int main() {
    const size_t bytes = 2ull << 30;
    uint4 *a, *b;
    uint32_t* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"bytes\": %zu, \"sweep\": [\n", bytes);
    const int gs[] = {8, 16, 32, 64, 128, 256, 512, 768, 1024, 2048, 4096, 16384};
    for (size_t gi = 0; gi < sizeof gs / sizeof gs[0]; ++gi) {
        const int G = gs[gi];
        const size_t per = bytes / 16 / G;
        float ms[4] = {0, 0, 0, 0};
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_write, dim3(G), dim3(256), 0, 0, a, per);
                else if (mode == 1) hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, (const uint4*)a, per, sink);
                else if (mode == 2) hipLaunchKernelGGL(k_copy, dim3(G), dim3(256), 0, 0, b, (const uint4*)a, per);
                else hipLaunchKernelGGL(k_write_nt, dim3(G), dim3(256), 0, 0, a, per);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (rep && t < best) best = t;
            }
            ms[mode] = best;
        }
        const double gb = bytes / 1e9;
        printf("  {\"workgroups\": %d, \"write_GBps\": %.0f, \"write_GBps_per_wg\": %.2f, \"read_GBps\": %.0f, \"read_GBps_per_wg\": %.2f, \"copy_GBps\": %.0f, \"write_nt_GBps\": %.0f}%s\n",
               G, gb / (ms[0] * 1e-3), gb / (ms[0] * 1e-3) / G, gb / (ms[1] * 1e-3), gb / (ms[1] * 1e-3) / G,
               2 * gb / (ms[2] * 1e-3), gb / (ms[3] * 1e-3), gi + 1 < sizeof gs / sizeof gs[0] ? "," : "");
    }
    printf("]}\n");
    return 0;
}
