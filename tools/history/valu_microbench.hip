// valu_microbench.hip -- measures the VALU issue cost (cycles per wave64 instruction per SIMD) of the
// instructions the path kernel is made of, on the GPU it runs on.  Diagnostic tool, not product code.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_microbench tools/valu_microbench.hip && /tmp/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 2000;
constexpr int kUnroll = 16;   // instructions per loop body (independent chains)

#define BODY16(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

// each kernel: 16 independent destination registers, one op each per iteration
#define KERNEL(NAME, ASM, CONSTRAINT_T, INIT)                                                     \
__global__ __launch_bounds__(256) void NAME(uint64_t* out, uint32_t seed) {                        \
    CONSTRAINT_T r[16];                                                                            \
    for (int i = 0; i < 16; ++i) r[i] = INIT;                                                      \
    uint64_t t0 = __builtin_amdgcn_s_memtime();                                                    \
    for (int it = 0; it < kIters; ++it) {                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) { ASM }                                     \
    }                                                                                              \
    uint64_t t1 = __builtin_amdgcn_s_memtime();                                                    \
    CONSTRAINT_T acc = r[0];                                                                       \
    for (int i = 1; i < 16; ++i) acc = acc + r[i];                                                 \
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }                                 \
    if (acc == (CONSTRAINT_T)12345.678) out[1] = 1;                                                \
}

KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0001f));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(0xD2511F53u));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_mul_hi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(0xD2511F53u));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(0x511F53u));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r[i]) : "v"(0x511F53u));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_log, asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed + 2))
KERNEL(k_sin, asm volatile("v_sin_f32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed) * 0.001f)
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed + 2))
KERNEL(k_cvt_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed + 2))
KERNEL(k_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(1.5));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0000001));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(1.0000001));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_pk_mul_f32, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(1.0000001));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0000001));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r[i]) : "v"(0x7fffffu));, uint32_t, threadIdx.x + i + seed)


KERNEL(k_add_f32, asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(1.0001f));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_mul_f32, asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(1.0001f));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_fmac_f32, asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(r[i]) : "v"(1.0001f));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_fmamk_f32, asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(r[i]) : "v"(1.0001f));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_add_u32, asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_lshl, asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(r[i]));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_and, asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_mov, asm volatile("v_mov_b32_e32 %0, %1" : "=v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_xad, asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_xor_s, asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(r[i]) : "s"(seed));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_cos, asm volatile("v_cos_f32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed) * 0.001f)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed) * 0.001f)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed) + 1.0f)
KERNEL(k_rndne_f64, asm volatile("v_rndne_f64 %0, %0" : "+v"(r[i]));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_ldexp_f64, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(r[i]) : "v"(1));, double, (double)(threadIdx.x + i + seed))
KERNEL(k_pk_add_f32, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(1.0000001));, double, (double)(threadIdx.x + i + seed))


KERNEL(k_mul_f32_lit, asm volatile("v_mul_f32_e32 %0, 0xbfb17218, %0" : "+v"(r[i]));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_and_lit, asm volatile("v_and_b32_e32 %0, 0x7fffff, %0" : "+v"(r[i]));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_or_inl, asm volatile("v_or_b32_e32 %0, 1.0, %0" : "+v"(r[i]));, uint32_t, threadIdx.x + i + seed)
KERNEL(k_fmaak_f32, asm volatile("v_fmaak_f32 %0, %0, %1, 0x42317218" : "+v"(r[i]) : "v"(1.0001f));, float, (float)(threadIdx.x + i + seed))
KERNEL(k_add_f32_s, asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(r[i]) : "s"(1.0001f));, float, (float)(threadIdx.x + i + seed))
__global__ __launch_bounds__(256) void k_mad_u64_s(uint64_t* out, uint32_t seed) {
    uint64_t r[16];
    uint32_t a[16];
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i + seed; a[i] = threadIdx.x * 7 + i; }
    uint32_t m = 0xD2511F53u + seed;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r[i]) : "v"(a[i]), "s"(m) : "vcc");
    }
    uint64_t acc = 0;
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (acc == 12345) out[1] = 1;
}


// ---- mixes: is the cost of a stream the SUM of its instructions' costs? ----
#define MIXKERNEL(NAME, BODY)                                                                      \
__global__ __launch_bounds__(256) void NAME(uint64_t* out, uint32_t seed) {                        \
    uint32_t r[16]; uint64_t q[8];                                                                 \
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x + i + seed;                                    \
    for (int i = 0; i < 8; ++i) q[i] = threadIdx.x + i;                                            \
    uint32_t c = seed * 3 + threadIdx.x;                                                           \
    for (int it = 0; it < kIters; ++it) { BODY }                                                   \
    uint32_t acc = 0;                                                                              \
    for (int i = 0; i < 16; ++i) acc += r[i];                                                      \
    for (int i = 0; i < 8; ++i) acc += (uint32_t)q[i] + (uint32_t)(q[i] >> 32);                    \
    if (acc == 12345) out[1] = 1;                                                                  \
}
#define XORV(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define ADDV(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(c));
#define MADQ(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[(i) & 7]) : "v"(r[i]), "v"(c) : "vcc");
MIXKERNEL(k_mix_xor_add, XORV(0) ADDV(1) XORV(2) ADDV(3) XORV(4) ADDV(5) XORV(6) ADDV(7) XORV(8) ADDV(9) XORV(10) ADDV(11) XORV(12) ADDV(13) XORV(14) ADDV(15))
MIXKERNEL(k_mix_xor_mad, XORV(0) MADQ(1) XORV(2) MADQ(3) XORV(4) MADQ(5) XORV(6) MADQ(7) XORV(8) MADQ(9) XORV(10) MADQ(11) XORV(12) MADQ(13) XORV(14) MADQ(15))
MIXKERNEL(k_mix_xxm, XORV(0) XORV(1) MADQ(2) XORV(3) XORV(4) MADQ(5) XORV(6) XORV(7) MADQ(8) XORV(9) XORV(10) MADQ(11) XORV(12) XORV(13) MADQ(14) XORV(15))
// dependent Philox-like chain: x -> mad -> xor(hi, lo_prev) -> xor(key) -> mad ...   two interleaved chains
__global__ __launch_bounds__(256) void k_philox_chain(uint64_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = threadIdx.x * 3 + seed, la = 1, lb = 2;
    uint32_t k = seed + threadIdx.x;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint64_t pa, pb;
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(pa) : "v"(a), "v"(0xD2511F53u) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(pb) : "v"(b), "v"(0xCD9E8D57u) : "vcc");
            uint32_t ha = (uint32_t)(pa >> 32), hb = (uint32_t)(pb >> 32);
            asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(ha) : "v"(lb));
            asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(hb) : "v"(la));
            asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(ha) : "v"(k));
            asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(hb) : "v"(k));
            la = (uint32_t)pa; lb = (uint32_t)pb; a = hb; b = ha;
        }
    }
    if (a + b + la + lb == 12345) out[1] = 1;
}

// v_mad_u64_u32: 64-bit destination, 32-bit sources
__global__ __launch_bounds__(256) void k_mad_u64(uint64_t* out, uint32_t seed) {
    uint64_t r[16];
    uint32_t a[16];
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i + seed; a[i] = threadIdx.x * 7 + i; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r[i]) : "v"(a[i]), "v"(0xD2511F53u) : "vcc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 12345) out[1] = 1;
}
__global__ __launch_bounds__(256) void k_cvt_f64_f32(uint64_t* out, uint32_t seed) {
    double r[16];
    float a[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i + seed; r[i] = 0; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    double acc = 0;
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 12345.5) out[1] = 1;
}

template <typename K>
int run(const char* name, K kernel, int waves_per_simd, uint64_t* d_out) {
    // one block of 256 threads per (4 SIMDs x 1 wave); waves_per_simd blocks per CU on ONE CU is enough
    int blocks = waves_per_simd;  // all land on distinct CUs possibly; force 1 CU's worth by using many blocks
    int grid = 256 * blocks;      // fill the chip so every CU has `waves_per_simd` blocks
    uint64_t h[2] = {0, 0};
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz on gfx9? measure cycles via wall-clock instead
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_simd = (double)kIters * 16 * waves_per_simd;   // wave-instructions issued on one SIMD
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-20s waves/SIMD %d: %8.3f ms  %6.3f ns per wave-instr per SIMD  = %5.2f cyc @2.4GHz (memtime ticks %llu)\n",
           name, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4, (unsigned long long)h[0]);
    return 0;
}

template <typename K>
int run_n(const char* name, K kernel, int waves_per_simd, int instr_per_iter, uint64_t* d_out) {
    int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double n = (double)kIters * instr_per_iter * waves_per_simd;
    printf("%-20s waves/SIMD %d: %8.3f ms  %6.3f ns per wave-instr per SIMD (%d instr/iter)\n", name, waves_per_simd, ms, ms * 1e6 / n, instr_per_iter);
    return 0;
}

int main() {
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, 64));
    for (int w : {4, 8}) {
        run("v_fma_f32", k_fma_f32, w, d_out);
        run("v_xor_b32", k_xor, w, d_out);
        run("v_bfi_b32", k_bfi, w, d_out);
        run("v_mul_lo_u32", k_mul_lo, w, d_out);
        run("v_mul_hi_u32", k_mul_hi, w, d_out);
        run("v_mad_u64_u32", k_mad_u64, w, d_out);
        run("v_mul_u32_u24", k_mul_u24, w, d_out);
        run("v_mad_u32_u24", k_mad_u24, w, d_out);
        run("v_log_f32", k_log, w, d_out);
        run("v_sin_f32", k_sin, w, d_out);
        run("v_sqrt_f32", k_sqrt, w, d_out);
        run("v_cvt_f32_u32", k_cvt_u32, w, d_out);
        run("v_cvt_f64_f32", k_cvt_f64_f32, w, d_out);
        run("v_add_f64", k_add_f64, w, d_out);
        run("v_mul_f64", k_mul_f64, w, d_out);
        run("v_fma_f64", k_fma_f64, w, d_out);
        run("v_pk_mul_f32", k_pk_mul_f32, w, d_out);
        run("v_pk_fma_f32", k_pk_fma_f32, w, d_out);
        run("v_add_f32_e32", k_add_f32, w, d_out);
        run("v_mul_f32_e32", k_mul_f32, w, d_out);
        run("v_fmac_f32_e32", k_fmac_f32, w, d_out);
        run("v_fmamk_f32", k_fmamk_f32, w, d_out);
        run("v_add_u32", k_add_u32, w, d_out);
        run("v_lshlrev_b32", k_lshl, w, d_out);
        run("v_and_b32", k_and, w, d_out);
        run("v_mov_b32", k_mov, w, d_out);
        run("v_add3_u32", k_add3, w, d_out);
        run("v_xad_u32", k_xad, w, d_out);
        run("v_and_or_b32", k_and_or, w, d_out);
        run("v_xor_b32(sgpr)", k_xor_s, w, d_out);
        run("v_cos_f32", k_cos, w, d_out);
        run("v_exp_f32", k_exp, w, d_out);
        run("v_rcp_f32", k_rcp, w, d_out);
        run("v_rndne_f64", k_rndne_f64, w, d_out);
        run("v_ldexp_f64", k_ldexp_f64, w, d_out);
        run("v_pk_add_f32", k_pk_add_f32, w, d_out);
        run("v_mul_f32 literal", k_mul_f32_lit, w, d_out);
        run("v_and_b32 literal", k_and_lit, w, d_out);
        run("v_or_b32 inline1.0", k_or_inl, w, d_out);
        run("v_fmaak_f32", k_fmaak_f32, w, d_out);
        run("v_add_f32 sgpr", k_add_f32_s, w, d_out);
        run("v_mad_u64_u32 sgpr", k_mad_u64_s, w, d_out);
        run("mix xor+add_u32", k_mix_xor_add, w, d_out);
        run("mix xor+mad_u64", k_mix_xor_mad, w, d_out);
        run("mix xor,xor,mad", k_mix_xxm, w, d_out);
        run_n("philox-like chain", k_philox_chain, w, 24, d_out);
        printf("\n");
    }
    return 0;
}
