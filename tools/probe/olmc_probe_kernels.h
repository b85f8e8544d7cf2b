// olmc_probe_kernels.h -- measurement and validation kernels of the INSTRUMENTED build (libolmc_probe.so).  Nothing here is compiled
// into libolmc.so.  They use the product's own device functions (olmc_kernels.h), so what they measure is the product's code.
#pragma once
#include "olmc_kernels.h"

namespace olmc {

// Power sums of the normal stream (validation tap): out[m-1] = sum over paths and steps of z^m, m = 1..4,
// z = kZScale * z' in fp64.  At 2^36 normals the second moment is resolved to 5e-6: a bias hunt.
__global__ __launch_bounds__(kBlock) void normal_moments_kernel(PathRange pr, ReduceWs ws) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int32_t blocks = (pr.n_steps + 3) >> 2;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        double m1 = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
        for (int32_t b = 0; b < blocks; ++b) {
            float z[4];
            raw_normals4(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), static_cast<uint32_t>(b), 0u, pr.key0, pr.key1, z);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * b + j < pr.n_steps) {
                    const double x = kZScale * static_cast<double>(z[j]), x2 = x * x;
                    m1 += x; m2 += x2; m3 += x2 * x; m4 += x2 * x2;
                }
        }
        acc[0] += m1; acc[1] += m2; acc[2] += m3; acc[3] += m4;
    }
    block_then_grid_reduce<4>(acc, ws);
}

// Measurement tap (bench.py; never on a pricing path): the shader clock the chip HOLDS while every SIMD runs the
// headline kernel's own step loop.  s_memtime counts shader cycles, s_memrealtime a constant 100 MHz, so
// clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  One stamp pair per
// workgroup around path_normal_sum; the stamps go to a buffer of their own and no priced value depends on them.
__global__ __launch_bounds__(kBlock) void clock_probe_kernel(PathRange pr, uint64_t* __restrict__ stamps, double* __restrict__ sink) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t g = pr.first + static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const double zsum = path_normal_sum(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), pr.n_steps, pr.key0, pr.key1);
    if (zsum == 1.2345678e300) sink[0] = zsum;          // never true: keeps the loop from being optimised away
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * static_cast<size_t>(blockIdx.x)] = t1 - t0;
        stamps[2 * static_cast<size_t>(blockIdx.x) + 1] = r1 - r0;
    }
}

// Measurement tap (tools/phase_stamps.py; never on a pricing path): WHERE a launch of the headline kernel spends its time.
// The body is european_path_kernel<1, true, kReduce, false> call for call (same device functions, same launch shape, same
// reduction); wave 0 of every workgroup additionally leaves four s_memrealtime stamps (a constant 100 MHz counter shared by
// the whole device): [0] entry, [1] step loop done, [2] payoffs + workgroup sums done (about to enter the grid reduction),
// [3] back from the grid reduction (ticket taken; for the workgroup that took the LAST ticket of the launch: totals written),
// and [4] where it ran: HW_REG_HW_ID in the low word (cu_id bits 11:8, sh_id 12, se_id 15:13), HW_REG_XCC_ID in the high word.
// The wave that writes the totals also stamps slot kStampWords * gridDim.x: the end of the launch's useful work.
constexpr int kStampWords = 5;
struct StampEpilogue {
    uint64_t* final_stamp;
    __device__ __forceinline__ void operator()(double) const {
        if (threadIdx.x == 0) *final_stamp = __builtin_amdgcn_s_memrealtime();
    }
};

__global__ __launch_bounds__(kBlock) void european_stamp_kernel(PathRange pr, ContractSet<1> cs, ReduceWs ws, uint64_t* __restrict__ stamps) {
    __shared__ double quarter_sum[kWavesPerBlock][kWave];
    const uint64_t t_entry = __builtin_amdgcn_s_memrealtime();
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave), lane = threadIdx.x & (kWave - 1);
    const bool split = static_cast<int32_t>(blockIdx.x) >= pr.split_from;
    const int64_t i = split ? static_cast<int64_t>(pr.split_from) * kBlock + (static_cast<int64_t>(blockIdx.x) - pr.split_from) * kWave + lane
                            : static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const uint64_t g = pr.first + static_cast<uint64_t>(i);
    double zsum = path_normal_quarters(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), pr.n_steps, split ? wave : 0,
                                       split ? wave + 1 : 4, pr.key0, pr.key1);
    if (split) {
        quarter_sum[wave][lane] = zsum;
        __syncthreads();
        zsum = ((quarter_sum[0][lane] + quarter_sum[1][lane]) + quarter_sum[2][lane]) + quarter_sum[3][lane];
    }
    zsum *= kZScale;
    asm volatile("s_nop 0" ::"v"(zsum));                 // the stamp below must follow the loop's last result
    const uint64_t t_loop = __builtin_amdgcn_s_memrealtime();
    double acc[2] = {0.0, 0.0};
    if (!split || wave == 0) european_payoffs<1, true, kReduce>(cs, zsum, i < pr.count, i, pr.count, nullptr, acc);
    constexpr int P = 2;
    __shared__ double stage[kWavesPerBlock][P];
    double w[P] = {acc[0], acc[1]};
    wave_transpose_reduce<P>(w);
    if ((lane & 31) == 0) stage[wave][lane >> 5] = w[0];
    __syncthreads();
    if (wave != 0) return;
    double v = 0.0;
    if (threadIdx.x < 2) v = ((stage[0][threadIdx.x] + stage[1][threadIdx.x]) + stage[2][threadIdx.x]) + stage[3][threadIdx.x];
    asm volatile("s_nop 0" ::"v"(v));
    const uint64_t t_sums = __builtin_amdgcn_s_memrealtime();
    grid_reduce<2, StampEpilogue>(v, ws, StampEpilogue{stamps + kStampWords * static_cast<size_t>(gridDim.x)});
    if (threadIdx.x == 0) {
        uint64_t* mine = stamps + kStampWords * static_cast<size_t>(blockIdx.x);
        const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        mine[0] = t_entry; mine[1] = t_loop; mine[2] = t_sums; mine[3] = __builtin_amdgcn_s_memrealtime();
        mine[4] = (static_cast<uint64_t>(xcc_id) << 32) | hw_id;
    }
}

// Measurement tap (bench.py): the issue cost of ONE instruction class on this device, in the form the path kernels
// use it (SGPR multipliers / keys where hipcc puts them there).  Every wave runs kProbeIters x 16 independent
// instructions of the class; with `w` workgroups per CU (= w waves per SIMD) the kernel's duration / (iters x 16 x w) is
// the time one SIMD needs per wave64 instruction of that class with w waves to pick from.  This calibrates the
// roofline's issue model on the chip and at the clock the benchmark itself runs at.
constexpr int kProbeIters = 2000;

#define OLMC_PROBE(NAME, TYPE, INIT, ASM)                                                                           \
    __global__ __launch_bounds__(kBlock) void NAME(uint32_t* __restrict__ sink, uint32_t seed, uint32_t kconst) {   \
        TYPE r[16];                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) r[i] = INIT;                                                 \
        for (int it = 0; it < kProbeIters; ++it) {                                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) { ASM; }                                                 \
        }                                                                                                           \
        TYPE acc = r[0];                                                                                            \
        _Pragma("unroll") for (int i = 1; i < 16; ++i) acc = acc + r[i];                                            \
        if (acc == static_cast<TYPE>(12345.678)) sink[0] = 1u;                                                      \
    }

#define OLMC_PF (static_cast<float>(threadIdx.x + i + seed) * 0.37f + 2.0f)
#define OLMC_PU (threadIdx.x * 2654435761u + i + seed)
#define OLMC_PD (static_cast<double>(threadIdx.x + i + seed) * 0.37 + 2.0)
OLMC_PROBE(probe_bitop3, uint32_t, OLMC_PU, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(seed + i), "s"(kconst)))
OLMC_PROBE(probe_cvt_f32_u32, float, OLMC_PF, asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_fmamk_f32, float, OLMC_PF, asm volatile("v_fmamk_f32 %0, %0, 0x2f800000, %1" : "+v"(r[i]) : "v"(1.0001f)))
OLMC_PROBE(probe_and_or, uint32_t, OLMC_PU, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "s"(kconst), "v"(0x3F800000u)))
OLMC_PROBE(probe_log_f32, float, OLMC_PF, asm volatile("v_log_f32_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_sqrt_f32, float, OLMC_PF, asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_sin_f32, float, OLMC_PF, asm volatile("v_sin_f32_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_cos_f32, float, OLMC_PF, asm volatile("v_cos_f32_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_exp_f32, float, OLMC_PF, asm volatile("v_exp_f32_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_add_f32, float, OLMC_PF, asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(1.0001f)))
OLMC_PROBE(probe_fma_f32, float, OLMC_PF, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0001f)))
OLMC_PROBE(probe_add_f64, double, OLMC_PD, asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(1.5)))
OLMC_PROBE(probe_fma_f64, double, OLMC_PD, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0000001)))
OLMC_PROBE(probe_rndne_f64, double, OLMC_PD, asm volatile("v_rndne_f64_e32 %0, %0" : "+v"(r[i])))
OLMC_PROBE(probe_ldexp_f64, double, OLMC_PD, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(r[i]) : "v"(1)))
// Mixed bodies (TWO instructions per body: the reported figure is nanoseconds per PAIR): do classes overlap?
// log+add: a transcendental beside a full-rate fp32 add;  log+bitop3: beside a three-operand integer op;
// operand-form variants of the two integer workhorses of Philox.
OLMC_PROBE(probe_mix_log_add, float, OLMC_PF, asm volatile("v_log_f32_e32 %0, %0\n\tv_add_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(1.0001f)))
OLMC_PROBE(probe_mix_log_bitop3, float, OLMC_PF, asm volatile("v_log_f32_e32 %0, %0\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(seed + i), "s"(kconst)))
OLMC_PROBE(probe_bitop3_vvv, uint32_t, OLMC_PU, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[i]) : "v"(seed + i), "v"(kconst + i)))
OLMC_PROBE(probe_bitop3_vvc, uint32_t, OLMC_PU, asm volatile("v_bitop3_b32 %0, %0, %1, 2 bitop3:0x96" : "+v"(r[i]) : "v"(seed + i)))
OLMC_PROBE(probe_xor_vv, uint32_t, OLMC_PU, asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(seed + i)))
OLMC_PROBE(probe_mix_bitop3_add, uint32_t, OLMC_PU, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n\tv_add_u32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(seed + i), "s"(kconst)))
#undef OLMC_PF
#undef OLMC_PU
#undef OLMC_PD

// mad + bitop3 alternating, as in a Philox round (pair cost)
__global__ __launch_bounds__(kBlock) void probe_mix_mad_bitop3(uint32_t* __restrict__ sink, uint32_t seed, uint32_t kconst) {
    uint64_t r[16];
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i + seed; a[i] = threadIdx.x * 7u + i; }
    for (int it = 0; it < kProbeIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint64_t carry;
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0\n\tv_bitop3_b32 %2, %2, %4, %3 bitop3:0x96" : "=v"(r[i]), "=s"(carry), "+v"(a[i]) : "s"(kconst), "v"(seed + i));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i] + a[i];
    if (acc == 12345u) sink[0] = 1u;
}

// v_mad_u64_u32 with the multiplier in a VGPR instead of an SGPR
__global__ __launch_bounds__(kBlock) void probe_mad_u64_u32_vv(uint32_t* __restrict__ sink, uint32_t seed, uint32_t kconst) {
    uint64_t r[16];
    uint32_t a[16];
    const uint32_t kv = kconst + (threadIdx.x >> 8);
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i + seed; a[i] = threadIdx.x * 7u + i; }
    for (int it = 0; it < kProbeIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint64_t carry;
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r[i]), "=s"(carry) : "v"(a[i]), "v"(kv));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (acc == 12345u) sink[0] = 1u;
}

// v_mad_u64_u32 (64-bit destination, SGPR multiplier as in the Philox rounds) and the two width-changing conversions
__global__ __launch_bounds__(kBlock) void probe_mad_u64_u32(uint32_t* __restrict__ sink, uint32_t seed, uint32_t kconst) {
    uint64_t r[16];
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i + seed; a[i] = threadIdx.x * 7u + i; }
    for (int it = 0; it < kProbeIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint64_t carry;         // the carry-out goes to an SGPR pair, as hipcc emits it in the Philox rounds
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r[i]), "=s"(carry) : "v"(a[i]), "s"(kconst));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));      // sixteen live destinations per iteration, no extra instruction
    }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (acc == 12345u) sink[0] = 1u;
}

__global__ __launch_bounds__(kBlock) void probe_cvt_f64_f32(uint32_t* __restrict__ sink, uint32_t seed, uint32_t) {
    double r[16];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = static_cast<float>(threadIdx.x + i + seed); r[i] = 0.0; }
    for (int it = 0; it < kProbeIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (acc == 12345.5) sink[0] = 1u;
}

__global__ __launch_bounds__(kBlock) void probe_cvt_i32_f64(uint32_t* __restrict__ sink, uint32_t seed, uint32_t) {
    int32_t r[16];
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = static_cast<double>(threadIdx.x + i + seed) * 0.37; r[i] = 0; }
    for (int it = 0; it < kProbeIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_i32_f64_e32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(r[i]));
    }
    int32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (acc == 12345) sink[0] = 1u;
}

// exp2_f64 (the fp64 Asian kernel's exponential) on an array: lets the tests pin it against a reference libm point by point.
// which = 0: exp2_f64 (degree-11 polynomial); 1: exp2_f64_tab (table; the argument is scaled by kExp2Entries = 256 here, exactly)
__global__ void exp2_probe_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ y, int which) {
    __shared__ double tab[kExp2Entries];
    exp2_table_to_lds(tab);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        y[i] = which ? exp2_f64_tab(static_cast<double>(kExp2Entries) * x[i], tab) : exp2_f64(x[i]);
}


// The Sobol kernels' inverse normal on an array of probabilities: z[i] = 0 + (the term every Sobol kernel adds to a point's sum for
// the uniform p[i]) -- form 0 = ndtri_w_add (one point), 1 = ndtri_w_regs_add (coefficients in registers), 2 = ndtri_w_regs_pair_add
// (two in lockstep: element i rides in slot i & 1 next to p = 1/2, whose normal is an exact zero), 3 = ndtri_lockstep_add<8>
// (elements 8 j .. 8 j + 7).  All forms must agree bit for bit; tests pin them to mpmath.
__global__ void ndtri_probe_kernel(const double* __restrict__ p, int64_t n, double* __restrict__ z, int form) {
    NdtriRegs regs;
    regs.load();
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        if (form == 0) {
            z[i] = ndtri_w_add(0.0, p[i], opaque_zero());
        } else if (form == 1) {
            z[i] = ndtri_w_regs_add(0.0, p[i], regs, opaque_zero());
        } else if (form == 2) {
            const double in[2] = {(i & 1) ? 0.5 : p[i], (i & 1) ? p[i] : 0.5};
            z[i] = ndtri_w_regs_pair_add(0.0, in, regs, opaque_zero());
        } else {
            const int64_t base = i & ~static_cast<int64_t>(7);
            double in[8], out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                in[j] = p[base + j < n ? base + j : i];
                out[j] = 0.0;
            }
            ndtri_lockstep_add<8>(out, in, opaque_zero());
            z[i] = out[i & 7];
        }
    }
}

// ---------------------------------------------------------------- the price of the reference's own width (round 4) ----
// The European step loop with fp64 NORMALS, as NumPy draws them (gbm_numpy.py:32-33: standard_normal of a PCG64 Generator, fp64):
// the same Philox4x32-10 counter stream, but one block now yields TWO normals -- its four words make two 53-bit uniforms
//   u_a = ((x0 << 21 ^ x1 >> 11) + 1/2) 2^-53  in (0, 1),   u_b = (x2 << 21 ^ x3 >> 11) 2^-53  (turn fraction),
// and one fp64 Box-Muller: r = sqrt(-2 ln u_a) (library log), (sin, cos)(2 pi u_b) (library sincospi), the sum of a path's
// normals carried in fp64 throughout.  Same launch shape, same fused reduction, same payoff code as european_path_kernel<1, true,
// kReduce, false>.  NOT a product path: a labelled number beside the product's "f32 normals"  -- what the
// reference's width would cost on this device, and that it moves the price by nothing measurable.
__global__ __launch_bounds__(kBlock) void european_f64_normals_kernel(PathRange pr, ContractSet<1> cs, ReduceWs ws) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const uint64_t g = pr.first + static_cast<uint64_t>(i);
    const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
    double zsum = 0.0;
    const int32_t pairs = (pr.n_steps + 1) >> 1;
    for (int32_t b = 0; b < pairs; ++b) {
        const Words4 w = philox4x32_10(g_lo, g_hi, static_cast<uint32_t>(b), 0u, pr.key0, pr.key1);
        const uint64_t ma = (static_cast<uint64_t>(w.x0) << 21) ^ (static_cast<uint64_t>(w.x1) >> 11);
        const uint64_t mb = (static_cast<uint64_t>(w.x2) << 21) ^ (static_cast<uint64_t>(w.x3) >> 11);
        const double ua = (static_cast<double>(ma) + 0.5) * 0x1p-53, ub = static_cast<double>(mb) * 0x1p-53;
        const double rad = sqrt(-2.0 * log(ua));
        double sn, cs_;
        sincospi(2.0 * ub, &sn, &cs_);
        zsum += rad * cs_;
        if (2 * b + 1 < pr.n_steps) zsum += rad * sn;
    }
    double acc[2] = {0.0, 0.0};
    european_payoffs<1, true, kReduce>(cs, zsum, i < pr.count, i, pr.count, nullptr, acc);
    block_then_grid_reduce<2>(acc, ws);
}

// A kernel that does nothing: a chain of them on one stream measures what two DEPENDENT launches cost each other (the command
// processor's gap) -- the floor under every per-date launch of the American option.
__global__ void empty_kernel(uint32_t* __restrict__ sink) {
    if (sink == nullptr && threadIdx.x == 12345) __builtin_trap();
}

}  // namespace olmc
