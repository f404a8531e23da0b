// membench.hip -- practical HBM ceilings on this box for the access mixes of k_decode:
//   fill   : pure 16 B/lane streaming stores
//   copy   : 16 B/lane read + write (1:1)
//   mix    : 1 read : 6 writes (the decode kernel's byte mix: 2.13 MB in, 12.87 MB out per frame)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_fill(u32x4* dst, size_t n, bool nt) {
    u32x4 v = {1u, 2u, 3u, threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (nt) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* src, u32x4* dst, size_t n, bool nt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 v = nt ? __builtin_nontemporal_load(src + i) : src[i];
        if (nt) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
// each 16 B read fans out to 6 x 16 B writes in 6 separate streams
__global__ __launch_bounds__(256) void k_mix(const u32x4* src, u32x4* dst, size_t n, bool nt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 v = src[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            v.x += k;
            if (nt) __builtin_nontemporal_store(v, dst + (size_t)k * n + i); else dst[(size_t)k * n + i] = v;
        }
    }
}
// narrow stores: 4 B per lane (u8-plane like) vs 16 B
__global__ __launch_bounds__(256) void k_fill4(unsigned* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (unsigned)i;
}

template <class F> float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b));
    CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const size_t R = 512ull << 20;            // 512 MiB read
    const size_t Wb = 6 * R;                  // 3 GiB written
    void *src, *dst; CK(hipMalloc(&src, R)); CK(hipMalloc(&dst, Wb));
    CK(hipMemset(src, 1, R)); CK(hipMemset(dst, 0, Wb));
    for (int grid : {2048, 8192, 65536}) {
        for (int nt = 0; nt < 2; ++nt) {
            float f = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, (u32x4*)dst, Wb / 16, (bool)nt); }, 5);
            float c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, R / 16, (bool)nt); }, 5);
            float m = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, R / 16, (bool)nt); }, 5);
            printf("grid %6d nt %d | fill %7.1f GB/s | copy %7.1f GB/s (r+w) | mix1:6 %7.1f GB/s (r+w)\n", grid, nt,
                   Wb / f / 1e6, 2.0 * R / c / 1e6, 7.0 * R / m / 1e6);
        }
    }
    float f4 = time_ms([&] { hipLaunchKernelGGL(k_fill4, dim3(8192), dim3(256), 0, 0, (unsigned*)dst, Wb / 4); }, 5);
    printf("fill 4B/lane %7.1f GB/s\n", Wb / f4 / 1e6);
    return 0;
}
