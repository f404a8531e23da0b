// runbench: the store pattern of k_dwf_emit without loads or math -- what does a compacted run cost to write?
// A wave writes consecutive "runs" of n 12-byte points (n ~ 68 of 128 rows kept), the way the emit kernel's column loop does:
//   mode 0  two global_store_dwordx3, each from the lanes that kept their row (scattered lanes, dense addresses)
//   mode 1  the same points moved to dense lanes first: one 64-lane dwordx3 store + one short one
//   mode 2  the run as a flat byte image: ceil(12 n / 16) lanes store 16 B each
//   mode 3  mode 0 with every lane active (128 points per run): the cost of the instruction itself
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/runbench tools/runbench.hip ; run: tools/runbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };

__global__ __launch_bounds__(256) void k_runs(float* out, const uint64_t* masks, uint32_t runs_per_wave, int mode, uint32_t nt) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    // every wave owns a contiguous slab of the output sized for 128 points per run
    float* base = out + (size_t)wave * runs_per_wave * 128u * 3u;
    uint64_t g = 0;   // points written so far by this wave
    for (uint32_t i = 0; i < runs_per_wave; ++i) {
        const uint64_t m0 = mode == 3 ? ~0ull : masks[(wave * 2654435761u + i * 2u) & 4095u];
        const uint64_t m1 = mode == 3 ? ~0ull : masks[(wave * 2654435761u + i * 2u + 1u) & 4095u];
        const uint32_t n0 = __popcll(m0), n1 = __popcll(m1), n = n0 + n1;
        const P3 v{(float)i, (float)lane, (float)wave};
        float* run = base + g * 3u;
        if (mode == 0 || mode == 3) {
            const uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0));
            const uint32_t r1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, n0));
            if ((m0 >> lane) & 1) { float* d = run + r0 * 3u; if (nt) { __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1); __builtin_nontemporal_store(v.z, d + 2);} else *(P3*)d = v; }
            if ((m1 >> lane) & 1) { float* d = run + r1 * 3u; if (nt) { __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1); __builtin_nontemporal_store(v.z, d + 2);} else *(P3*)d = v; }
        } else if (mode == 1) {
            const uint32_t a = n < 64u ? n : 64u;
            if (lane < a) { float* d = run + lane * 3u; if (nt) { __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1); __builtin_nontemporal_store(v.z, d + 2);} else *(P3*)d = v; }
            if (lane + 64u < n) { float* d = run + (lane + 64u) * 3u; if (nt) { __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1); __builtin_nontemporal_store(v.z, d + 2);} else *(P3*)d = v; }
        } else {
            const uint32_t bytes = n * 12u;
            for (uint32_t o = lane * 16u; o + 16u <= bytes; o += 1024u) {
                float* d = run + o / 4u;
                if (nt) { __builtin_nontemporal_store(v.x, d); __builtin_nontemporal_store(v.y, d + 1); __builtin_nontemporal_store(v.z, d + 2); __builtin_nontemporal_store(v.x, d + 3); }
                else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.x; }
            }
        }
        g += n;
    }
}

int main() {
    const uint32_t waves = 32768, runs = 16;   // 524288 runs = the columns of 256 frames x 2048
    std::vector<uint64_t> h(4096);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (auto& m : h) {   // ~53 % of the bits set
        m = 0;
        for (int b = 0; b < 64; ++b) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            if ((s >> 11) % 100 < 53) m |= 1ull << b;
        }
    }
    uint64_t* masks; float* out;
    hipMalloc(&masks, h.size() * 8); hipMemcpy(masks, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const size_t bytes = (size_t)waves * runs * 128 * 12;
    hipMalloc(&out, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (uint32_t nt = 0; nt < 2; ++nt)
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 3; ++rep) k_runs<<<waves / 4, 256>>>(out, masks, runs, mode, nt);
            hipEventRecord(a);
            for (int rep = 0; rep < 10; ++rep) k_runs<<<waves / 4, 256>>>(out, masks, runs, mode, nt);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
            const double mb = mode == 3 ? bytes / 1e6 : bytes / 1e6 * 0.53;
            printf("nt %u mode %d: %.4f ms  %.0f MB  %.2f TB/s\n", nt, mode, ms, mb, mb / ms / 1e6 * 1e3 / 1e3);
        }
    return 0;
}
