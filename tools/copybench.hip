// copybench.hip -- what one host <-> device crossing costs on this box, for the sizes the frame-at-a-time drop-in API moves
// (one 128 x 2048 plane = 1 MB, a frame's packets = 2.1 MB, its planes = 3.9 MB, an f64 cloud = 6.3 MB): pageable vs pinned
// hipMemcpy, a pinned staging ring fed by memcpy, kernels that read / write pinned host memory in place, and the fixed costs
// (launch + sync, allocation calls, first touch of fresh host memory).  Numbers behind DESIGN section 5 (round 6).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/copybench tools/copybench.hip ; run: tools/copybench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_empty() {}

template <typename F>
static double best_of(int reps, F f) {
    double b = 1e30;
    for (int i = 0; i < reps; ++i) {
        auto t0 = clk::now();
        f();
        b = std::min(b, us(t0, clk::now()));
    }
    return b;
}
template <typename F>
static double mean_of(int reps, F f) {
    f();
    auto t0 = clk::now();
    for (int i = 0; i < reps; ++i) f();
    return us(t0, clk::now()) / reps;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t sizes[] = {1u << 20, 2129920, 3932160 + 28672, 6291456};
    const size_t maxb = 8u << 20;
    void *d_a, *d_b;
    CK(hipMalloc(&d_a, maxb));
    CK(hipMalloc(&d_b, maxb));
    uint8_t* pin;
    CK(hipHostMalloc((void**)&pin, maxb, hipHostMallocDefault));
    std::memset(pin, 1, maxb);
    std::vector<uint8_t> page(maxb, 2);
    const size_t chunk = 256u << 10;
    uint8_t* ring;
    CK(hipHostMalloc((void**)&ring, 4 * chunk, hipHostMallocDefault));
    hipEvent_t ev[4];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    std::printf("{\n");
    // fixed costs
    std::printf(" \"launch_sync_us\": %.2f,\n", mean_of(200, [&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); CK(hipStreamSynchronize(st)); }));
    std::printf(" \"copy4B_pinned_h2d_sync_us\": %.2f,\n", mean_of(200, [&] { CK(hipMemcpyAsync(d_a, pin, 4, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }));
    std::printf(" \"copy4B_pinned_d2h_sync_us\": %.2f,\n", mean_of(200, [&] { CK(hipMemcpyAsync(pin, d_a, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }));
    std::printf(" \"eleven_async_d2h_copies_issue_us\": %.2f,\n", mean_of(50, [&] {
        auto t0 = clk::now(); (void)t0;
        for (int i = 0; i < 11; ++i) CK(hipMemcpyAsync(pin + i * 4096, (uint8_t*)d_a + i * 4096, 2048, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st)); }));
    std::printf(" \"h2d_kernel_d2h_1MB_one_sync_us\": %.2f,\n", mean_of(100, [&] {
        CK(hipMemcpyAsync(d_a, pin, 1u << 20, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_copy16, 256, 256, 0, st, (const uint4*)d_a, (uint4*)d_b, (size_t)(1u << 16));
        CK(hipMemcpyAsync(pin, d_b, 1u << 20, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st)); }));
    {   // host -> host through a kernel (what destagger on pool containers is), and how the wait is done
        uint8_t* pin2;
        CK(hipHostMalloc((void**)&pin2, maxb, hipHostMallocDefault));
        const size_t n16 = (1u << 20) / 16;
        for (int wg : {32, 128, 512}) {
            std::printf(" \"inplace_1MB_host_to_host_%dwg_sync_us\": %.2f,\n", wg, mean_of(100, [&] { hipLaunchKernelGGL(k_copy16, wg, 256, 0, st, (const uint4*)pin, (uint4*)pin2, n16); CK(hipStreamSynchronize(st)); }));
            std::printf(" \"inplace_1MB_host_to_host_%dwg_queryspin_us\": %.2f,\n", wg, mean_of(100, [&] { hipLaunchKernelGGL(k_copy16, wg, 256, 0, st, (const uint4*)pin, (uint4*)pin2, n16); while (hipStreamQuery(st) == hipErrorNotReady) {} }));
        }
        std::printf(" \"launch_queryspin_us\": %.2f,\n", mean_of(200, [&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); while (hipStreamQuery(st) == hipErrorNotReady) {} }));
        hipEvent_t e;
        CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        std::printf(" \"inplace_1MB_host_to_host_128wg_eventsync_us\": %.2f,\n", mean_of(100, [&] { hipLaunchKernelGGL(k_copy16, 128, 256, 0, st, (const uint4*)pin, (uint4*)pin2, n16); CK(hipEventRecord(e, st)); CK(hipEventSynchronize(e)); }));
        std::printf(" \"inplace_6MB_dev_to_host_sync_us\": %.2f,\n", mean_of(50, [&] { hipLaunchKernelGGL(k_copy16, 1024, 256, 0, st, (const uint4*)d_a, (uint4*)pin2, (size_t)6291456 / 16); CK(hipStreamSynchronize(st)); }));
        std::printf(" \"inplace_6MB_dev_to_host_queryspin_us\": %.2f,\n", mean_of(50, [&] { hipLaunchKernelGGL(k_copy16, 1024, 256, 0, st, (const uint4*)d_a, (uint4*)pin2, (size_t)6291456 / 16); while (hipStreamQuery(st) == hipErrorNotReady) {} }));
        // do a copy-engine upload and a kernel's stores to host memory share the link at the same time?
        hipStream_t st2;
        CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        const size_t up = 2129920, down = 3960832;
        std::printf(" \"dma_h2d_2MB_alone_us\": %.2f,\n", mean_of(50, [&] { CK(hipMemcpyAsync(d_b, pin, up, hipMemcpyHostToDevice, st2)); CK(hipStreamSynchronize(st2)); }));
        std::printf(" \"kernel_writes_host_4MB_alone_us\": %.2f,\n", mean_of(50, [&] { hipLaunchKernelGGL(k_copy16, 256, 256, 0, st, (const uint4*)d_a, (uint4*)pin2, down / 16); CK(hipStreamSynchronize(st)); }));
        std::printf(" \"dma_h2d_2MB_and_kernel_writes_host_4MB_together_us\": %.2f,\n", mean_of(50, [&] {
            CK(hipMemcpyAsync(d_b, pin, up, hipMemcpyHostToDevice, st2));
            hipLaunchKernelGGL(k_copy16, 256, 256, 0, st, (const uint4*)d_a, (uint4*)pin2, down / 16);
            CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2)); }));
        std::printf(" \"dma_h2d_2MB_and_dma_d2h_4MB_together_us\": %.2f,\n", mean_of(50, [&] {
            CK(hipMemcpyAsync(d_b, pin, up, hipMemcpyHostToDevice, st2));
            CK(hipMemcpyAsync(pin2, d_a, down, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2)); }));
        CK(hipStreamDestroy(st2));
        CK(hipHostFree(pin2));
    }
    {
        void* p = nullptr;
        std::printf(" \"hipMalloc_free_1MB_us\": %.2f,\n", mean_of(50, [&] { CK(hipMalloc(&p, 1u << 20)); CK(hipFree(p)); }));
        std::printf(" \"hipMalloc_free_6MB_us\": %.2f,\n", mean_of(50, [&] { CK(hipMalloc(&p, 6u << 20)); CK(hipFree(p)); }));
        std::printf(" \"hipHostMalloc_free_6MB_us\": %.2f,\n", mean_of(20, [&] { CK(hipHostMalloc(&p, 6u << 20, hipHostMallocDefault)); CK(hipHostFree(p)); }));
        std::printf(" \"malloc_touch_free_6MB_us\": %.2f,\n", mean_of(20, [&] { uint8_t* q = (uint8_t*)std::malloc(6u << 20); for (size_t i = 0; i < (6u << 20); i += 4096) q[i] = 1; std::free(q); }));
        std::printf(" \"calloc_vector_6MB_us\": %.2f,\n", mean_of(20, [&] { std::vector<double> v((6u << 20) / 8); asm volatile("" ::"r"(v.data()) : "memory"); }));
        std::printf(" \"hipHostRegister_unregister_6MB_us\": %.2f,\n", mean_of(10, [&] { CK(hipHostRegister(page.data(), 6u << 20, hipHostRegisterDefault)); CK(hipHostUnregister(page.data())); }));
        std::printf(" \"hipPointerGetAttributes_us\": %.3f,\n", mean_of(1000, [&] { hipPointerAttribute_t a; (void)hipPointerGetAttributes(&a, page.data()); (void)hipGetLastError(); }));
    }
    std::printf(" \"memcpy_host_6MB_us\": %.2f,\n", mean_of(20, [&] { std::memcpy(pin, page.data(), 6u << 20); }));
    std::printf(" \"sizes\": {\n");
    for (size_t si = 0; si < 4; ++si) {
        const size_t n = sizes[si];
        const int R = 30;
        double pg_h2d = mean_of(R, [&] { CK(hipMemcpyAsync(d_a, page.data(), n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
        double pg_d2h = mean_of(R, [&] { CK(hipMemcpyAsync(page.data(), d_a, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
        double pn_h2d = mean_of(R, [&] { CK(hipMemcpyAsync(d_a, pin, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
        double pn_d2h = mean_of(R, [&] { CK(hipMemcpyAsync(pin, d_a, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
        // staging ring: memcpy chunk k while chunk k-1 is on the wire
        double rg_h2d = mean_of(R, [&] {
            size_t off = 0; int k = 0;
            while (off < n) {
                const size_t c = std::min(chunk, n - off);
                const int s = k & 3;
                if (k >= 4) CK(hipEventSynchronize(ev[s]));
                std::memcpy(ring + s * chunk, page.data() + off, c);
                CK(hipMemcpyAsync((uint8_t*)d_a + off, ring + s * chunk, c, hipMemcpyHostToDevice, st));
                CK(hipEventRecord(ev[s], st));
                off += c; ++k;
            }
            CK(hipStreamSynchronize(st)); });
        double rg_d2h = mean_of(R, [&] {
            size_t off = 0; int k = 0; const int nk = (int)((n + chunk - 1) / chunk);
            auto drain = [&](int j) { const size_t o = (size_t)j * chunk; CK(hipEventSynchronize(ev[j & 3])); std::memcpy(page.data() + o, ring + (j & 3) * chunk, std::min(chunk, n - o)); };
            for (; k < nk; ++k) {
                if (k >= 4) drain(k - 4);
                const size_t c = std::min(chunk, n - off);
                CK(hipMemcpyAsync(ring + (k & 3) * chunk, (uint8_t*)d_a + off, c, hipMemcpyDeviceToHost, st));
                CK(hipEventRecord(ev[k & 3], st));
                off += c;
            }
            for (int j = std::max(0, nk - 4); j < nk; ++j) drain(j); });
        // kernels working on pinned host memory in place (no copy engine, one launch)
        const size_t n16 = n / 16;
        double zc_h2d = mean_of(R, [&] { hipLaunchKernelGGL(k_copy16, 256, 256, 0, st, (const uint4*)pin, (uint4*)d_a, n16); CK(hipStreamSynchronize(st)); });
        double zc_d2h = mean_of(R, [&] { hipLaunchKernelGGL(k_copy16, 256, 256, 0, st, (const uint4*)d_a, (uint4*)pin, n16); CK(hipStreamSynchronize(st)); });
        double zc_d2h_1k = mean_of(R, [&] { hipLaunchKernelGGL(k_copy16, 1024, 256, 0, st, (const uint4*)d_a, (uint4*)pin, n16); CK(hipStreamSynchronize(st)); });
        std::printf("  \"%zu\": {\"pageable_h2d_us\": %.1f, \"pageable_d2h_us\": %.1f, \"pinned_h2d_us\": %.1f, \"pinned_d2h_us\": %.1f, "
                    "\"ring_h2d_us\": %.1f, \"ring_d2h_us\": %.1f, \"kernel_reads_host_us\": %.1f, \"kernel_writes_host_us\": %.1f, "
                    "\"kernel_writes_host_1024wg_us\": %.1f, \"pinned_d2h_GBps\": %.1f}%s\n",
                    n, pg_h2d, pg_d2h, pn_h2d, pn_d2h, rg_h2d, rg_d2h, zc_h2d, zc_d2h, zc_d2h_1k, n / pn_d2h / 1e3, si == 3 ? "" : ",");
    }
    std::printf(" }\n}\n");
    return 0;
}
