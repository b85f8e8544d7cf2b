// overlap_probe.hip -- do the transcendental unit and the main VALU of a gfx950 SIMD work at the same time for DIFFERENT waves?
// Workgroups of type A run chains of v_log_f32, type B chains of v_mad_u64_u32 (the two big classes of the path kernel's step loop).
// Per CU: (2 A), (2 B), (4 A), (4 B) and the mix (2 A + 2 B) -- if the mix costs max(2A, 2B) the units overlap, if it costs 2A + 2B they
// share one issue port.  Diagnostic tool, not product code.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ab/overlap_probe tools/overlap_probe.hip && tools/ab/overlap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 4000;

// mode: 0 = all A, 1 = all B, 2 = first half of the grid A, second half B
__global__ __launch_bounds__(256) void mix(uint32_t* sink, int mode, int half, uint32_t seed) {
    const bool type_b = mode == 1 || (mode == 2 && static_cast<int>(blockIdx.x) >= half);
    if (!type_b) {
        float r[16];
        for (int i = 0; i < 16; ++i) r[i] = static_cast<float>(threadIdx.x + i + seed) * 0.37f + 2.0f;
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_log_f32_e32 %0, %0" : "+v"(r[i]));
        }
        float acc = 0.f;
        for (int i = 0; i < 16; ++i) acc += r[i];
        if (acc == 12345.678f) sink[0] = 1u;
    } else {
        uint64_t r[16];
        uint32_t a[16];
        for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i + seed; a[i] = threadIdx.x * 7u + i; }
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint64_t carry;
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r[i]), "=s"(carry) : "v"(a[i]), "s"(0xD2511F53u));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));
        }
        uint64_t acc = 0;
        for (int i = 0; i < 16; ++i) acc += r[i];
        if (acc == 12345u) sink[0] = 1u;
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* sink;
    CHECK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](int mode, int wgs_per_cu, const char* what) -> int {
        const int grid = cus * wgs_per_cu;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipExtLaunchKernelGGL(mix, dim3(grid), dim3(256), 0, 0, e0, e1, 0, sink, mode, grid / 2, 1u);
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("{\"what\": \"%s\", \"workgroups_per_cu\": %d, \"us\": %.1f, \"ns_per_wave_instruction_per_simd\": %.3f}\n", what, wgs_per_cu, best * 1e3,
               best * 1e6 / (double(kIters) * 16 * wgs_per_cu));
        return 0;
    };
    for (int warm = 0; warm < 3; ++warm) run(2, 4, "warm-up");
    run(0, 2, "2 waves per SIMD, all v_log_f32");
    run(1, 2, "2 waves per SIMD, all v_mad_u64_u32");
    run(0, 4, "4 waves per SIMD, all v_log_f32");
    run(1, 4, "4 waves per SIMD, all v_mad_u64_u32");
    run(2, 4, "4 waves per SIMD: 2 v_log_f32 + 2 v_mad_u64_u32 (first half of the grid A, second half B)");
    run(0, 1, "1 wave per SIMD, all v_log_f32");
    run(1, 1, "1 wave per SIMD, all v_mad_u64_u32");
    run(2, 2, "2 waves per SIMD: 1 v_log_f32 + 1 v_mad_u64_u32");
    return 0;
}
