// olmc_kernels.h -- gfx950 device code of the Monte Carlo path engine.
//
// One thread owns one normal stream = one antithetic pair of GBM paths
// (the per-path loop of src/simulation/gbm_numba.py:86-95 in the reference,
// which is also what simulate_gbm_numpy computes as row sums,
// src/simulation/gbm_numpy.py:43-51).  Normals are produced in registers by
// Philox4x32-10 + Box-Muller; HBM sees only per-block partial sums (or, in the
// array-returning mode, the terminal prices).  The step loop is VALU-bound:
// no LDS, no global loads, MFMA unused (there is no contraction to feed it).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "olmc_host_math.h"   // Contract, ContractSet, the fused-Greeks argument structs and their constants (plain C++, shared with the CPU-only test build)

namespace olmc {

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kWavesPerBlock = kBlock / kWave;

// ---------------------------------------------------------------- Philox ----
constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

struct Words4 {
    uint32_t x0, x1, x2, x3;
};

// a ^ b ^ c in ONE instruction: gfx950's v_bitop3_b32 with truth table 0x96.  hipcc does not form it
// from `a ^ b ^ c` on its own (it emits two v_xor_b32), and XORs were the largest instruction class of
// a Philox call: 38 -> 19 per call, kernel 146 -> 117.5 us (A/B on one device, identical prices).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Philox4x32-10 (Salmon et al., SC'11).  The key is wave-uniform; each round is two 32x32->64
// multiplies (v_mad_u64_u32: hi and lo from one instruction, ~4.4 cycles, not quarter rate) and two
// 3-input XORs.  Round 1 and half of round 2 are loop-invariant or wave-uniform in the step loop and are
// hoisted by hipcc.  (Pinning the round keys in VGPRs to avoid SGPR-source XORs was measured: no gain --
// in this mix every non-transcendental VALU instruction costs ~4 cycles of SIMD issue.)
__device__ __forceinline__ Words4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = static_cast<uint64_t>(kPhiloxM0) * c0;
        const uint64_t p1 = static_cast<uint64_t>(kPhiloxM1) * c2;
        const uint32_t n0 = xor3(static_cast<uint32_t>(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3(static_cast<uint32_t>(p0 >> 32), c3, k1);
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    return Words4{c0, c1, c2, c3};
}

// The twenty round keys of one (seed) pinned in VGPRs.  They are wave-uniform, so hipcc keeps them in SGPRs and every
// round's XOR3 reads two VGPRs and an SGPR -- and on gfx950 a VALU instruction with an SGPR operand holds the issue port
// for ~4.4 cycles where the same instruction on three VGPRs takes ~2.6 (olmc_issue_probe: v_bitop3_b32 1.90 vs 1.12 ns
// at 8 waves/SIMD).  Twenty v_mov per kernel buy that back on 19 XOR3s per Philox call: 1M x 252 European 103.4 -> 100.0 us,
// fused 8 / 14 contracts 119.1 -> 115.1 / 133.8 -> 130.5 us (interleaved A/B, one device, identical sums).  66 VGPRs
// instead of 46, i.e. 7 waves per SIMD instead of 8: pinning only 18 / 16 / 12 keys to stay at 64 was slower (100.7 /
// 101.1 / 101.3 us), forcing 64 by launch bounds 100.9.  Also measured and NOT adopted: the mantissa mask of the angle in
// a VGPR (100.5 us, no gain) and the two multipliers in VGPRs (102.3 us, worse: v_mad_u64_u32 costs the same with an SGPR
// multiplier, the copies only add pressure).
struct RoundKeys {
    uint32_t k[20];
};

__device__ __forceinline__ RoundKeys pin_round_keys(uint32_t k0, uint32_t k1) {
    RoundKeys rk;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t a = k0 + static_cast<uint32_t>(r) * kPhiloxW0, b = k1 + static_cast<uint32_t>(r) * kPhiloxW1;
        asm volatile("v_mov_b32 %0, %1" : "=v"(rk.k[2 * r]) : "s"(a));          // volatile: must not be folded back into an SGPR operand
        asm volatile("v_mov_b32 %0, %1" : "=v"(rk.k[2 * r + 1]) : "s"(b));
    }
    return rk;
}

__device__ __forceinline__ Words4 philox4x32_10_pinned(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const RoundKeys& rk) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = static_cast<uint64_t>(kPhiloxM0) * c0;
        const uint64_t p1 = static_cast<uint64_t>(kPhiloxM1) * c2;
        const uint32_t n0 = xor3(static_cast<uint32_t>(p1 >> 32), c1, rk.k[2 * round]);
        const uint32_t n2 = xor3(static_cast<uint32_t>(p0 >> 32), c3, rk.k[2 * round + 1]);
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
    }
    return Words4{c0, c1, c2, c3};
}

// ------------------------------------------------------------ Box-Muller ----
// Radius word:  u_a = (x_a + 0.5) * 2^-32 in fp32 (never 0, so the log is finite; may
//               round to 1); |z| <= sqrt(2*33*ln 2) = 6.76.
// Angle word:   the low 23 bits of x_b become the mantissa of a float in [1, 2);
//               v_sin_f32 / v_cos_f32 take revolutions and are periodic, so that float IS
//               the angle (turn fraction (x_b & 0x7fffff) * 2^-23): one v_and_or_b32.
// v_log_f32 is log2, so sqrt(-2 ln u) = sqrt(2 ln 2) * sqrt(-log2 u).  The kernels work
// with the RAW normals z' = sqrt(-log2 u_a) * {cos, sin}(2 pi u_b) and apply the constant
// kZScale = sqrt(2 ln 2) once per path (to sum z') or once per thread (to vol): two
// multiplies fewer per pair, and the negation is a free source modifier of v_sqrt_f32.
// kZScale = sqrt(2 ln 2): olmc_host_math.h
constexpr float kZScaleF = 1.17741002f;

__device__ __forceinline__ void box_muller_raw(uint32_t xa, uint32_t xb, float& z_cos, float& z_sin) {
    constexpr float kTwoM32 = 2.3283064365386963e-10f;   // 2^-32
    constexpr float kTwoM33 = 1.1641532182693481e-10f;   // 2^-33
    const float ua = __builtin_fmaf(static_cast<float>(xa), kTwoM32, kTwoM33);
    const float turns = __uint_as_float((xb & 0x007FFFFFu) | 0x3F800000u);
    const float rad = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(ua));
    z_cos = rad * __builtin_amdgcn_cosf(turns);
    z_sin = rad * __builtin_amdgcn_sinf(turns);
}

// Four RAW normals of steps 4*block .. 4*block+3 of global path `g`.
__device__ __forceinline__ void raw_normals4(uint32_t g_lo, uint32_t g_hi, uint32_t block, uint32_t tag,
                                             uint32_t k0, uint32_t k1, float (&z)[4]) {
    const Words4 w = philox4x32_10(g_lo, g_hi, block, tag, k0, k1);
    box_muller_raw(w.x0, w.x1, z[0], z[1]);
    box_muller_raw(w.x2, w.x3, z[2], z[3]);
}

// The same with the round keys pinned in VGPRs (pin_round_keys, once per kernel).
__device__ __forceinline__ void raw_normals4_pinned(uint32_t g_lo, uint32_t g_hi, uint32_t block, uint32_t tag, const RoundKeys& rk, float (&z)[4]) {
    const Words4 w = philox4x32_10_pinned(g_lo, g_hi, block, tag, rk);
    box_muller_raw(w.x0, w.x1, z[0], z[1]);
    box_muller_raw(w.x2, w.x3, z[2], z[3]);
}

// acc + (sum of the four RAW normals of one Philox block), factored as
//   fma(rad_b, cos_b + sin_b, fma(rad_a, cos_a + sin_a, acc)):
// two adds and two fmas per block INCLUDING the accumulation (the packed-math form needed
// register-pair moves: 5.25 instructions per block against 4).
__device__ __forceinline__ float raw_block_sum_of_words(float acc, const Words4& w) {
    constexpr float kTwoM32 = 2.3283064365386963e-10f;   // 2^-32
    constexpr float kTwoM33 = 1.1641532182693481e-10f;   // 2^-33
    const float ua = __builtin_fmaf(static_cast<float>(w.x0), kTwoM32, kTwoM33);
    const float ub = __builtin_fmaf(static_cast<float>(w.x2), kTwoM32, kTwoM33);
    const float ta = __uint_as_float((w.x1 & 0x007FFFFFu) | 0x3F800000u);
    const float tb = __uint_as_float((w.x3 & 0x007FFFFFu) | 0x3F800000u);
    const float rad_a = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(ua));
    const float rad_b = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(ub));
    const float sum_a = __builtin_amdgcn_cosf(ta) + __builtin_amdgcn_sinf(ta);
    const float sum_b = __builtin_amdgcn_cosf(tb) + __builtin_amdgcn_sinf(tb);
    return __builtin_fmaf(rad_b, sum_b, __builtin_fmaf(rad_a, sum_a, acc));
}

__device__ __forceinline__ float raw_block_accumulate(float acc, uint32_t g_lo, uint32_t g_hi, uint32_t block, uint32_t tag,
                                                      uint32_t k0, uint32_t k1) {
    return raw_block_sum_of_words(acc, philox4x32_10(g_lo, g_hi, block, tag, k0, k1));
}

// sum_t Z_t (true normals) of one path.  Block b covers steps 4b..4b+3; fp32 within a group of
// kGroup blocks (16 normals, interleaved by hipcc for ILP), fp64 across groups; a trailing
// partial block contributes its first n_steps % 4 normals.
//
// The fp64 sum over groups has ONE canonical association, whoever computes it:
//     sum = ((Q0 + Q1) + Q2) + Q3,    Q_w = the full groups [w gq, (w+1) gq) added in order, gq = ceil(#full groups / 4),
// Q3 additionally taking the trailing partial group and the partial block.  A thread that owns a whole path evaluates
// the four quarters one after the other (path_normal_sum); in a SPLIT workgroup the four waves evaluate one quarter each
// of the same 64 paths (path_normal_quarter) and combine through LDS -- the same additions in the same order, hence
// the same bits.  (Group sums are fp32 values added in fp64: the association would matter only for a group sum below
// ~5e-7, about one path in ten million -- but "equal seeds give equal bits" must not depend on the launch shape.)
constexpr int kGroup = 4;   // measured: 2 is 2 % slower, 8 no faster

// Quarters [w0, w1) of one path's canonical sum, UNSCALED (multiply by kZScale once all four are in).  w0 = 0, w1 = 4 is the
// whole path: ((0 + Q0) + Q1) + Q2, then + Q3.  A single quarter w returns 0 + Q_w = Q_w exactly.  One loop nest serves both.
__device__ __forceinline__ double path_normal_quarters(uint32_t g_lo, uint32_t g_hi, int32_t n_steps, int32_t w0, int32_t w1, uint32_t k0,
                                                       uint32_t k1, uint32_t tag = 0u) {
    const int32_t full = n_steps >> 2;                 // blocks whose four steps all count
    const int32_t n_groups = full / kGroup;            // full groups
    const int32_t gq = (n_groups + 3) >> 2;            // groups per quarter
    const RoundKeys rk = pin_round_keys(k0, k1);
    double acc = 0.0, q = 0.0;
#pragma unroll 1
    for (int32_t w = w0; w < w1; ++w) {
        q = 0.0;
        const int32_t b1 = min((w + 1) * gq, n_groups) * kGroup;
        for (int32_t b = min(w * gq, n_groups) * kGroup; b < b1; b += kGroup) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < kGroup; ++j) s = raw_block_sum_of_words(s, philox4x32_10_pinned(g_lo, g_hi, static_cast<uint32_t>(b + j), tag, rk));
            q += static_cast<double>(s);
        }
        if (w < 3) acc += q;
    }
    if (w1 == 4) {                                     // whoever owns quarter 3 also owns the trailing partial group and block
        int32_t b = n_groups * kGroup;
        if (b < full) {
            float s = 0.0f;
            for (; b < full; ++b) s = raw_block_sum_of_words(s, philox4x32_10_pinned(g_lo, g_hi, static_cast<uint32_t>(b), tag, rk));
            q += static_cast<double>(s);
        }
        const int32_t rem = n_steps & 3;
        if (rem) {
            float z[4];
            const Words4 wd = philox4x32_10_pinned(g_lo, g_hi, static_cast<uint32_t>(full), tag, rk);
            box_muller_raw(wd.x0, wd.x1, z[0], z[1]);
            box_muller_raw(wd.x2, wd.x3, z[2], z[3]);
            float s = z[0];
            if (rem > 1) s += z[1];
            if (rem > 2) s += z[2];
            q += static_cast<double>(s);
        }
        acc += q;
    }
    return acc;
}

__device__ __forceinline__ double path_normal_sum(uint32_t g_lo, uint32_t g_hi, int32_t n_steps, uint32_t k0, uint32_t k1,
                                                  uint32_t tag = 0u) {
    return path_normal_quarters(g_lo, g_hi, n_steps, 0, 4, k0, k1, tag) * kZScale;
}

// ------------------------------------------------------------ reductions ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Grid-wide reduction fused into the path kernels: no second launch.
//   level 0  every workgroup stores its NV block sums (write-through `sc1` stores), drains
//            them (s_waitcnt vmcnt(0)) and takes a ticket on its group's counter
//            (agent-scope atomic add).  A group = kGroupBlocks consecutive workgroups.
//   level 1  the workgroup whose ticket is last in its group acquires, sums the group's
//            rows in index order, stores the group row the same way and takes a ticket on
//            the top counter;
//   level 2  the last group finisher sums the group rows in index order and writes out[].
// Sums run in INDEX order, never arrival order, so equal inputs give equal bits
// (tests/test_monte_carlo.py:153-158 of the reference needs price1 == price2).  Counters
// are zero on entry and each finisher re-zeroes the one it consumed, so back-to-back
// launches on one stream need no memset.  Protocol per cdna_hip_programming.md G16:
// sc1 payload + drained vmcnt before the agent-scope add on the producer, agent acquire
// fence after the returned add on the consumer.  Only wave 0 of a workgroup takes part.
constexpr int kGroupBlocks = 256;
constexpr int kGroupRowsCapacity = ((1 << 18) / kGroupBlocks + 1) * 32;    // doubles behind group_rows (host: kMaxGroups * kMaxNV)
constexpr int kCounterStride = 64;     // uint32 words between ticket counters: one 256-B granule each.  Packed
                                       // into one line, 3,907 tickets serialised at ~88 atomics/us (a 44 us floor
                                       // under every 1M-path launch, whatever its step count)

struct ReduceWs {
    double* block_rows;     // [gridDim.x][NV]
    double* group_rows;     // [n_groups][NV]
    uint32_t* counters;     // [(n_groups + 1) * kCounterStride], one counter per stride, last = top counter
    double* out;            // [NV] (+1 when tail >= 0)
    double tail;            // if >= 0, written to out[NV] (the sample count of the triple)
    uint64_t row_capacity;  // doubles allocated behind block_rows: a launch whose gridDim.x * NV exceeds it
                            // refuses to store (out[] = NaN) instead of writing out of bounds
    uint64_t* done_flag;    // host-mapped; when not NULL the wave that wrote out[] then stores done_value there (system-scope
    uint64_t done_value;    // release): a blocking caller polls it instead of waiting for the kernel's completion signal
};

// After out[] is written (by this very wave): make it visible system-wide, then raise the caller's flag.
__device__ __forceinline__ void signal_done(const ReduceWs& ws) {
    if (ws.done_flag == nullptr) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: out[] has reached host memory before the flag can
    if (threadIdx.x == 0) __hip_atomic_store(ws.done_flag, ws.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// What the workgroup that took the last ticket does before it loads the rows the others stored: an AGENT-scope acquire
// (`s_waitcnt vmcnt(0); buffer_inv sc1`).  Every row is also stored write-through (`sc1`) and drained (`s_waitcnt vmcnt(0)`) by its
// writer before that writer's ticket, the counter is an agent-scope atomic, and every load of a row in the consumer is an `sc1` load
// to registers (load_sc1, L2-served) -- the hand-off for which cdna_hip_programming.md Guideline 16 allows a wavefront-scope fence in
// place of the acquire, but ONLY for launches of one workgroup per CU (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in
// place of the acquire": a hand-off must match one row of that table in every cell).  The launches here run up to 7 workgroups per
// CU plus split workgroups, outside that envelope: round 4 shipped the relaxed form everywhere and nothing ever failed, but a stale
// 16-byte row at 1M paths would move a price by 0.25 sigma, invisible to every statistical gate.  So the acquire is back (round 5).
// Its price, measured (profiles/r04_ab_kernels.txt, r05_ab_acquire.txt): within run-to-run noise at every size (1M x 252: 100.80
// vs 100.73 us) -- the 16 us the relaxed form once saved were in the American option's per-date tickets, which no longer exist
// (lsm_step_kernel sums the previous date's rows across a kernel boundary).  -DOLMC_AGENT_ACQUIRE=0 builds the relaxed form (A/B
// only; refused on any target but gfx942 / gfx950, whose cache behaviour it leans on).
#ifndef OLMC_AGENT_ACQUIRE
#define OLMC_AGENT_ACQUIRE 1
#endif
#if !OLMC_AGENT_ACQUIRE && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "OLMC_AGENT_ACQUIRE=0 relies on gfx942 / gfx950 cache behaviour (sc1 rows are L2-served): build this target with the acquire"
#endif
__device__ __forceinline__ void acquire_rows() {
#if OLMC_AGENT_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

__device__ __forceinline__ void store_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lanes c < NV return sum_{row < rows} src[row * NV + c], rows taken in a fixed order:
// lane = sub * NVP + c sums rows == sub (mod 64/NVP) ascending, then the sub-sums are
// folded by a shuffle tree.  Call with all 64 lanes of a wave active.
template <int NV>
__device__ __forceinline__ double wave_rows_sum(const double* src, int32_t rows) {
    constexpr int NVP = NV <= 2 ? 2 : NV <= 8 ? 8 : NV <= 16 ? 16 : 32;
    constexpr int SUBS = kWave / NVP;
    const int lane = threadIdx.x & (kWave - 1);
    const int c = lane % NVP, sub = lane / NVP;
    double v = 0.0;
    if (c < NV) {
        constexpr int kBatch = 8;      // loads issued back to back, then added in row order: the chain is latency-bound
        for (int32_t row = sub; row < rows; row += SUBS * kBatch) {
            double t[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int32_t rj = row + j * SUBS;
                t[j] = rj < rows ? load_sc1(src + static_cast<size_t>(rj) * NV + c) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) v += t[j];
        }
    }
#pragma unroll
    for (int off = kWave / 2; off >= NVP; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;   // valid in lanes < NV
}

// `v` = this workgroup's sum of component threadIdx.x (threads < NV of wave 0).  Wave 0 only.
// `done(total)` runs in the ONE wave that holds the grid totals (lane c < NV has component c),
// right after out[] is written: device-side post-processing without another launch.
struct NoEpilogue {
    __device__ __forceinline__ void operator()(double) const {}
};

template <int NV, typename Epilogue = NoEpilogue>
__device__ __forceinline__ void grid_reduce(double v, const ReduceWs& ws, Epilogue done = Epilogue()) {
    const int lane = threadIdx.x;       // wave 0: lane == threadIdx.x
    const int32_t n_blocks = static_cast<int32_t>(gridDim.x);
    const int32_t n_groups = (n_blocks + kGroupBlocks - 1) / kGroupBlocks;
    const int32_t group = static_cast<int32_t>(blockIdx.x) / kGroupBlocks;
    const int32_t group_size = min(kGroupBlocks, n_blocks - group * kGroupBlocks);

    if (static_cast<uint64_t>(n_blocks) * NV > ws.row_capacity || n_groups * NV > kGroupRowsCapacity) {   // host/kernel NV mismatch
        if (blockIdx.x == 0) {
            if (lane < NV) ws.out[lane] = __builtin_nan("");
            signal_done(ws);
        }
        return;
    }
    if (lane < NV) store_sc1(ws.block_rows + static_cast<size_t>(blockIdx.x) * NV + lane, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(ws.counters + static_cast<size_t>(group) * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != static_cast<uint32_t>(group_size - 1)) return;

    acquire_rows();
    const double g = wave_rows_sum<NV>(ws.block_rows + static_cast<size_t>(group) * kGroupBlocks * NV, group_size);
    if (n_groups == 1) {            // <= 65,536 paths (the interactive sizes): the only group IS the total, skip level 2
        if (lane < NV) ws.out[lane] = g;
        if (lane == 0) {
            if (ws.tail >= 0.0) ws.out[NV] = ws.tail;
            __hip_atomic_store(ws.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        done(g);
        signal_done(ws);
        return;
    }
    if (lane < NV) store_sc1(ws.group_rows + static_cast<size_t>(group) * NV + lane, g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        __hip_atomic_store(ws.counters + static_cast<size_t>(group) * kCounterStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __hip_atomic_fetch_add(ws.counters + static_cast<size_t>(n_groups) * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != static_cast<uint32_t>(n_groups - 1)) return;

    acquire_rows();
    const double total = wave_rows_sum<NV>(ws.group_rows, n_groups);
    if (lane < NV) ws.out[lane] = total;
    if (lane == 0) {
        if (ws.tail >= 0.0) ws.out[NV] = ws.tail;
        __hip_atomic_store(ws.counters + static_cast<size_t>(n_groups) * kCounterStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    done(total);
    signal_done(ws);
}

// The same reduction for WIDE rows (NV > 2), with the whole workgroup summing.  grid_reduce lets one wave sum a group's rows:
// with NV = 2 that is 8 rows per lane, one batch of loads in flight, but a row of 32 sums (second-order Greeks) leaves only two
// lanes per column -- 128 rows each, sixteen dependent batches of ~1 us behind the LAST workgroup of the launch (round 3 found the
// fused 14-contract kernel paying 21 us over the one-contract kernel for 450 instructions per wave that should cost 8; and the
// American option's per-date launches, NV = 16, spending 8 of their 11 us here).  Here all 256 threads of the workgroup that took
// the last ticket load: thread t sums rows == t / NVP (mod 256 / NVP) of column t % NVP in row order, the 256 / NVP partial
// sums of a column meet in LDS and are added in index order -- 32 rows per lane for NV = 32, four batches.  Call with ALL waves
// of the workgroup; `v` is this workgroup's sum of component threadIdx.x (wave 0, lanes < NV).
template <int NV>
__device__ __forceinline__ double workgroup_rows_sum(const double* src, int32_t rows, double* part /* LDS [kBlock] */) {
    constexpr int NVP = NV <= 8 ? 8 : NV <= 16 ? 16 : 32;
    constexpr int SUBS = kBlock / NVP;
    const int c = threadIdx.x % NVP, sub = threadIdx.x / NVP;
    double v = 0.0;
    if (c < NV) {
        // loads in flight per thread: a group of 256 rows is ONE round trip for every row width (NVP = 16: 16 threads per column, 16 rows
        // each; NVP = 32: 8 threads, 32 rows = two trips).  Round 3 kept 8 in flight: two trips behind the last workgroup of every date
        // of the American option (NV = 16), four behind the 14-contract Greeks'.
        constexpr int kBatch = NVP >= 16 ? 16 : 8;
        for (int32_t row = sub; row < rows; row += SUBS * kBatch) {
            double t[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int32_t rj = row + j * SUBS;
                t[j] = rj < rows ? load_sc1(src + static_cast<size_t>(rj) * NV + c) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) v += t[j];
        }
    }
    part[threadIdx.x] = v;                       // part[sub * NVP + c]
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x < NV) {
#pragma unroll
        for (int k = 0; k < SUBS; ++k) total += part[k * NVP + threadIdx.x];
    }
    __syncthreads();                             // `part` is free again
    return total;                                // valid in threads < NV
}

template <int NV, typename Epilogue = NoEpilogue>
__device__ __forceinline__ void grid_reduce_workgroup(double v, const ReduceWs& ws, Epilogue done = Epilogue()) {
    __shared__ double part[kBlock];
    __shared__ uint32_t last;                    // did this workgroup take the last ticket (of its group / of the launch)?
    const int t = threadIdx.x;
    const bool wave0 = t < kWave;
    const int32_t n_blocks = static_cast<int32_t>(gridDim.x);
    const int32_t n_groups = (n_blocks + kGroupBlocks - 1) / kGroupBlocks;
    const int32_t group = static_cast<int32_t>(blockIdx.x) / kGroupBlocks;
    const int32_t group_size = min(kGroupBlocks, n_blocks - group * kGroupBlocks);

    if (static_cast<uint64_t>(n_blocks) * NV > ws.row_capacity || n_groups * NV > kGroupRowsCapacity) {   // host/kernel NV mismatch
        if (blockIdx.x == 0 && wave0) {
            if (t < NV) ws.out[t] = __builtin_nan("");
            signal_done(ws);
        }
        return;
    }
    if (wave0) {
        if (t < NV) store_sc1(ws.block_rows + static_cast<size_t>(blockIdx.x) * NV + t, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) last = __hip_atomic_fetch_add(ws.counters + static_cast<size_t>(group) * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                           static_cast<uint32_t>(group_size - 1);
    }
    __syncthreads();
    if (!last) return;                           // workgroup-uniform

    acquire_rows();
    const double g = workgroup_rows_sum<NV>(ws.block_rows + static_cast<size_t>(group) * kGroupBlocks * NV, group_size, part);
    if (n_groups == 1) {                         // the only group IS the total
        if (wave0) {
            if (t < NV) ws.out[t] = g;
            if (t == 0) {
                if (ws.tail >= 0.0) ws.out[NV] = ws.tail;
                __hip_atomic_store(ws.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            done(g);
            signal_done(ws);
        }
        return;
    }
    if (wave0) {
        if (t < NV) store_sc1(ws.group_rows + static_cast<size_t>(group) * NV + t, g);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) {
            __hip_atomic_store(ws.counters + static_cast<size_t>(group) * kCounterStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(ws.counters + static_cast<size_t>(n_groups) * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                   static_cast<uint32_t>(n_groups - 1);
        }
    }
    __syncthreads();
    if (!last) return;

    acquire_rows();
    const double total = workgroup_rows_sum<NV>(ws.group_rows, n_groups, part);
    if (wave0) {
        if (t < NV) ws.out[t] = total;
        if (t == 0) {
            if (ws.tail >= 0.0) ws.out[NV] = ws.tail;
            __hip_atomic_store(ws.counters + static_cast<size_t>(n_groups) * kCounterStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        done(total);
        signal_done(ws);
    }
}

// Wave-wide sums of P (a power of two) per-lane values in P-1 + (6 - log2 P) exchange-adds instead of
// 6 P: at every halving step a lane trades half of its values with the lane `off` away and keeps the
// sums of the other half, so after log2 P steps each lane holds ONE value's sum over a lane subset;
// the remaining steps finish that single value.  On return v[0] of every lane holds the total of value
// index  lane >> (6 - log2 P).  Fixed exchange pattern => deterministic.
//
// The two widest exchanges (lane ^ 32, lane ^ 16) are gfx950's v_permlane32_swap / v_permlane16_swap: ONE VALU
// instruction per dword that swaps the upper half-wave (odd rows of 16 lanes) of register A with the lower half-wave
// (even rows) of register B -- with A = v[k], B = v[k + H] that IS the trade of this step, so a kept value costs two swaps
// and an add where the generic form (select what to send, select what to keep, two ds_bpermute through the LDS crossbar,
// add) cost ten instructions and an LDS round trip.  Same operands per addition as the generic form (a + b vs b + a in the
// upper lanes), hence the same bits.  Three quarters of the exchanges of a 32-value reduction happen at these two levels.
__device__ __forceinline__ uint32_t dbl_lo(double x) { return static_cast<uint32_t>(__double_as_longlong(x)); }
__device__ __forceinline__ uint32_t dbl_hi(double x) { return static_cast<uint32_t>(static_cast<uint64_t>(__double_as_longlong(x)) >> 32); }
__device__ __forceinline__ double dbl_of(uint32_t lo, uint32_t hi) {
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}

// a' + b' after swapping the upper (OFF = 32: lanes 32..63; OFF = 16: rows 1 and 3) part of a with the lower part of b:
// lanes with bit OFF clear get a(own) + a(partner), lanes with it set get b(partner) + b(own).
template <int OFF>
__device__ __forceinline__ double swap_add(double a, double b) {
    static_assert(OFF == 32 || OFF == 16, "permlane swaps exist for half-waves and rows");
    if constexpr (OFF == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap(dbl_lo(a), dbl_lo(b), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(dbl_hi(a), dbl_hi(b), false, false);
        return dbl_of(lo[0], hi[0]) + dbl_of(lo[1], hi[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap(dbl_lo(a), dbl_lo(b), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(dbl_hi(a), dbl_hi(b), false, false);
        return dbl_of(lo[0], hi[0]) + dbl_of(lo[1], hi[1]);
    }
}

template <int P, int OFF = kWave / 2>
__device__ __forceinline__ void wave_transpose_reduce(double (&v)[P]) {
    if constexpr (P > 1) {
        constexpr int H = P / 2;
        double kept[H];
        if constexpr (OFF >= 16) {
#pragma unroll
            for (int k = 0; k < H; ++k) kept[k] = swap_add<OFF>(v[k], v[k + H]);
        } else {
            const bool upper = (threadIdx.x & OFF) != 0;
#pragma unroll
            for (int k = 0; k < H; ++k) {
                const double send = upper ? v[k] : v[k + H];
                const double mine = upper ? v[k + H] : v[k];
                kept[k] = mine + __shfl_xor(send, OFF, kWave);
            }
        }
        wave_transpose_reduce<H, OFF / 2>(kept);
        v[0] = kept[0];
    } else if constexpr (OFF >= 1) {
        double one[1];
        if constexpr (OFF >= 16) one[0] = swap_add<OFF>(v[0], v[0]);        // own + partner in every lane
        else one[0] = v[0] + __shfl_xor(v[0], OFF, kWave);
        wave_transpose_reduce<1, OFF / 2>(one);
        v[0] = one[0];
    }
}

constexpr int pow2_ceil(int n) { return n <= 1 ? 1 : 2 * pow2_ceil((n + 1) / 2); }
constexpr int log2_of(int p) { return p <= 1 ? 0 : 1 + log2_of(p / 2); }

// One-thread-per-path kernels: fold NV per-thread values over the workgroup (fixed order,
// LDS-staged across the four waves), then into the grid reduction.
//
// `w` holds the values after the FIRST `DONE` halving steps of wave_transpose_reduce<P> (DONE = 0: the P padded values
// themselves; DONE = 1: the P / 2 sums the lane ^ 32 exchange leaves -- a producer that makes its values two at a time can
// fold that step into its own loop and never hold all P of them, see european_payoffs_folded).
template <int NV, int DONE = 0, typename Epilogue = NoEpilogue>
__device__ __forceinline__ void block_then_grid_reduce_from(double (&w)[pow2_ceil(NV) >> DONE], const ReduceWs& ws, Epilogue done = Epilogue()) {
    constexpr int P = pow2_ceil(NV);
    constexpr int SHIFT = 6 - log2_of(P);           // lanes sharing one value after the transpose-reduce: 2^SHIFT
    static_assert(DONE == 0 || (DONE == 1 && P >= 2), "only the widest exchange can be folded into the producer");
    __shared__ double stage[kWavesPerBlock][P];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    wave_transpose_reduce<(P >> DONE), ((kWave / 2) >> DONE)>(w);
    if ((lane & ((1 << SHIFT) - 1)) == 0) stage[wave][lane >> SHIFT] = w[0];
    __syncthreads();
    if constexpr (NV <= 2) {
        if (wave != 0) return;
    }
    double s = 0.0;
    if (threadIdx.x < NV) {
        s = stage[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < kWavesPerBlock; ++k) s += stage[k][threadIdx.x];
    }
    if constexpr (NV <= 2) grid_reduce<NV, Epilogue>(s, ws, done);                 // narrow rows: one wave is plenty
    else grid_reduce_workgroup<NV, Epilogue>(s, ws, done);                         // wide rows: the whole workgroup sums them
}

// The workgroup half alone: threads < NV return the workgroup's sum of component threadIdx.x (the row block_then_grid_reduce stores).
// Ends on a barrier-free read of `stage`: a caller that reuses LDS afterwards synchronises itself.
template <int NV>
__device__ __forceinline__ double block_row_sum(const double (&v)[NV]) {
    constexpr int P = pow2_ceil(NV);
    constexpr int SHIFT = 6 - log2_of(P);
    __shared__ double stage[kWavesPerBlock][P];
    double w[P];
#pragma unroll
    for (int i = 0; i < P; ++i) w[i] = i < NV ? v[i] : 0.0;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    wave_transpose_reduce<P, kWave / 2>(w);
    if ((lane & ((1 << SHIFT) - 1)) == 0) stage[wave][lane >> SHIFT] = w[0];
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < NV) {
        s = stage[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < kWavesPerBlock; ++k) s += stage[k][threadIdx.x];
    }
    return s;
}

template <int NV, typename Epilogue = NoEpilogue>
__device__ __forceinline__ void block_then_grid_reduce(const double (&v)[NV], const ReduceWs& ws, Epilogue done = Epilogue()) {
    constexpr int P = pow2_ceil(NV);
    double w[P];
#pragma unroll
    for (int i = 0; i < P; ++i) w[i] = i < NV ? v[i] : 0.0;
    block_then_grid_reduce_from<NV, 0, Epilogue>(w, ws, done);
}

// 2^y in fp64, |error| <= 1.5 ulp (the library exp() is ~25 instructions + its argument scaling; this is 15 and takes
// the exponent already in log2 units).  n = rint(y), r = y - n in [-1/2, 1/2] (exact), 2^r by the degree-11 polynomial
// interpolating 2^r at the Chebyshev nodes of [-1/2, 1/2] (fitted in 50-digit arithmetic: interpolation error 4e-18,
// the 1.5 ulp is the Horner chain's rounding, measured against mpmath on 20,001 points), scaled by v_ldexp_f64.
// Finite overflow -> inf, underflow -> 0, NaN -> NaN, as exp2(); an INFINITE argument answers NaN (inf - rint(inf)) -- a
// cumulative log-return is infinite only for infinite parameters, and guarding the case would cost 3 of 15 instructions.
__device__ __forceinline__ double exp2_f64(double y) {
    const double n = __builtin_rint(y);
    const double r = y - n;
    double p = 4.4558179083360645e-10;
    p = __builtin_fma(p, r, 7.074194297288521e-09);
    p = __builtin_fma(p, r, 1.0178057087733941e-07);
    p = __builtin_fma(p, r, 1.3215432535912375e-06);
    p = __builtin_fma(p, r, 1.5252733841556773e-05);
    p = __builtin_fma(p, r, 0.00015403530463724353);
    p = __builtin_fma(p, r, 0.001333355814640647);
    p = __builtin_fma(p, r, 0.009618129107587256);
    p = __builtin_fma(p, r, 0.055504108664821625);
    p = __builtin_fma(p, r, 0.24022650695910158);
    p = __builtin_fma(p, r, 0.6931471805599453);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, static_cast<int>(n));
}

// The same function by a 256-entry table: with c = 256 y,  j = rint(c),  r = c - j in [-1/2, 1/2] (exact),
//     2^y = 2^(j div 256) * T[j mod 256] * 2^(r / 256),      T[k] = 2^(k / 256) rounded to fp64 (2 KB of LDS),
// and 2^(r / 256) - 1 = r q(r) with q of degree 3 (|r ln 2 / 256| <= 0.00135: interpolation error 0.04 ulp; tools/fit_exp2_table.py),
// so the result is fma(T, r q(r), T) scaled by v_ldexp_f64.  12 VALU instructions -- rndne, sub, cvt, three 2-cycle integer ops for
// the table index and the exponent, 3 fma + 1 mul + 1 fma, ldexp -- about 42 issue cycles against the 60 of the degree-11 form,
// plus one ds_read_b64 per call (lanes scatter over 256 entries: a few-way bank conflicts, on a unit the loop does not otherwise use).
// Error: half an ulp of T, half an ulp of the final fma, the polynomial's 0.04: <= 1.1 ulp.  Takes 256 y, so a caller that carries
// its exponent in units of 1/256 (asian_exp64_kernel scales drift and vol once) pays no multiply.  Limits as exp2_f64.
// (Round 3 started with 64 entries and degree 5: 935 -> 876 us at 1M x 1024; 256 entries and degree 4: -> 823 us.  The next fma
// would need 4,096 entries.)
constexpr double kExp2Q[4] = {0.0027076061740622767, 3.665565596910102e-06, 3.308302983711395e-09, 2.2393953276859936e-12};
__constant__ double kExp2Tab[256] = {
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175,
    1.0108892860517005, 1.0136300849514894, 1.016378314910953, 1.019133996077738,
    1.0218971486541166, 1.0246677928971357, 1.0274459491187637, 1.030231637686041,
    1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264,
    1.0556451783605572, 1.0585073227945128, 1.061377227289262, 1.0642549128844645,
    1.0671404006768237, 1.0700337118202419, 1.0729348675259756, 1.075843889062791,
    1.0787607977571199, 1.0816856149932152, 1.0846183622133092, 1.0875590609177697,
    1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175,
    1.1143867425958924, 1.1174081515673693, 1.1204377524096067, 1.12347556733302,
    1.1265216186082418, 1.129575928566288, 1.1326385195987192, 1.1357094141578055,
    1.1387886347566916, 1.1418762039695616, 1.1449721444318042, 1.148076478840179,
    1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373,
    1.1763969916502812, 1.1795865274628758, 1.182784710984341, 1.1859915656609938,
    1.189207115002721, 1.1924313825831512, 1.1956643920398273, 1.1989061670743806,
    1.202156731452703, 1.2054161090051239, 1.2086843236265816, 1.2119613992768012,
    1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165,
    1.241857812073484, 1.245224830175258, 1.2486009771892048, 1.2519862778663162,
    1.255380757024691, 1.2587844395497165, 1.2621973503942507, 1.2656195145788063,
    1.2690509571917332, 1.2724917033894028, 1.275941778396392, 1.2794012075056693,
    1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773,
    1.3109612115247644, 1.3145155879493546, 1.318079601266064, 1.3216532776031575,
    1.3252366431597413, 1.3288297242059544, 1.3324325470831615, 1.3360451382041458,
    1.339667524053303, 1.3432997311868353, 1.3469417862329458, 1.3505937158920345,
    1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238,
    1.383909881963832, 1.387662042298529, 1.3914243757719262, 1.3951969099662003,
    1.3989796725383112, 1.4027726912202048, 1.4065759938190154, 1.4103896082172707,
    1.4142135623730951, 1.4180478843204152, 1.4218926021691656, 1.4257477441054942,
    1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438,
    1.460917794180647, 1.4648787441464057, 1.4688504333369818, 1.4728328908693675,
    1.4768261459394993, 1.4808302278224719, 1.4848451658727524, 1.488870989524397,
    1.4929077282912648, 1.4969554117672355, 1.5010140696264256, 1.5050837316234065,
    1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568,
    1.5422108254079407, 1.5463921831410214, 1.550584877685, 1.5547889397770887,
    1.559004400237837, 1.5632312899713576, 1.567469639965553, 1.5717194812923414,
    1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771,
    1.6104903319492543, 1.6148568142048607, 1.6192351351948637, 1.6236253270173289,
    1.6280274218573478, 1.632441451987275, 1.6368674497669644, 1.6413054476440063,
    1.645755478153965, 1.6502175739206177, 1.6546917676561943, 1.6591780921616162,
    1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326,
    1.7001063537185235, 1.7047158096580513, 1.709337763100463, 1.713972247929926,
    1.718619298122478, 1.723278947746274, 1.7279512309618377, 1.732636182022311,
    1.7373338352737062, 1.7420442251551564, 1.746767386199169, 1.7515033530318782,
    1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401,
    1.7947090750031072, 1.7995750249405351, 1.804454167806624, 1.809346539371032,
    1.8142521755003989, 1.8191711121586085, 1.8241033854070534, 1.8290490314048973,
    1.8340080864093424, 1.8389805867758937, 1.843966568958626, 1.8489660695104508,
    1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392,
    1.8945759815869656, 1.8997126981765553, 1.9048633418176741, 1.9100279502703899,
    1.9152065613971474, 1.9203992131630474, 1.925605943636125, 1.930826790987627,
    1.9360617934922943, 1.9413109895286405, 1.9465744175792332, 1.9518521162309783,
    1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402,
};

// kExp2Entries (olmc_host_math.h) = table entries = workgroup size: one entry per thread on the way to LDS
static_assert(kExp2Entries == kBlock, "one table entry per thread");
constexpr int kExp2Shift = 8;

__device__ __forceinline__ void exp2_table_to_lds(double* tab /* LDS [kExp2Entries] */) {
    for (int k = threadIdx.x; k < kExp2Entries; k += blockDim.x) tab[k] = kExp2Tab[k];
    __syncthreads();
}

__device__ __forceinline__ double exp2_f64_tab(double c, const double* tab /* LDS [kExp2Entries] */) {       // c = kExp2Entries * y
    const double n = __builtin_rint(c);
    const double r = c - n;
    const int j = static_cast<int>(n);
    const double t = tab[j & (kExp2Entries - 1)];
    double q = kExp2Q[3];
    q = __builtin_fma(q, r, kExp2Q[2]);
    q = __builtin_fma(q, r, kExp2Q[1]);
    q = __builtin_fma(q, r, kExp2Q[0]);
    return __builtin_ldexp(__builtin_fma(t, q * r, t), j >> kExp2Shift);
}

#ifndef OLMC_EXP2_TABLE
#define OLMC_EXP2_TABLE 1           // 0 builds the degree-11 polynomial into asian_exp64_kernel (A/B measurements)
#endif

// Contract, ContractSet<NSETS>: olmc_host_math.h
struct PathRange {
    uint64_t first;    // global index of local path 0
    int64_t count;     // paths in this launch
    int32_t n_steps;
    uint32_t key0, key1;
    int32_t split_from;   // european_path_kernel (grid covers every path): workgroups >= split_from are SPLIT workgroups (see
                          // there); INT32_MAX = none.  Other kernels ignore it.
};

enum Mode : int { kReduce = 0, kTerminal = 1, kControlVariate = 2, kSumOnly = 3 };   // kSumOnly: sum x per contract, no sum x^2 (prices only)

template <int MODE>
__device__ __forceinline__ void add_sample(double (&acc)[MODE == kControlVariate ? 5 : 2], double x, double st) {
    if constexpr (MODE == kControlVariate) {
        acc[0] += x; acc[1] += st; acc[2] += x * x; acc[3] += st * st; acc[4] += x * st;
    } else {
        acc[0] += x; acc[1] += x * x;
    }
}

// European terminal payoff, one thread per path (= one antithetic pair).
//   kReduce          out[2s], out[2s+1] = sum x, sum x^2 of contract s (x = UNdiscounted payoff)
//   kTerminal        terminal[i] = S_T^+, terminal[count + i] = S_T^-  (coalesced [pos | neg], gbm_numpy.py:51)
//   kControlVariate  out[0..4] = sum x, sum s, sum x^2, sum s^2, sum x*s  (NSETS == 1)
// (Round 1 built a variant in which EVERY workgroup split one 64-path tile's steps over its four waves: never faster,
// 121 vs 117.5 us, because every path then pays the LDS meeting and three idle waves in the epilogue.  Only the
// remainder of a launch is split now -- see SPLIT workgroups below.)
// STRIDED = false: the grid covers every path (one per thread), so the accumulators are born AFTER
// the step loop and do not occupy registers during it -- with 8 / 16 contracts (16 / 32 fp64 sums)
// that is the difference between 5 and 7-8 waves per SIMD in the loop that matters.
//
// SPLIT workgroups (STRIDED = false only).  A launch of W = ceil(count / 256) equal workgroups on C compute units takes
// ceil(W / C) workgroup-times per CU whatever the remainder is: 1M paths = 3906.25 workgroups = 15.26 per CU costs 16
// (+4.8 %).  So the host lets the first F = floor(W / C) * C workgroups own 256 whole paths each, as ever, and hands the
// remaining paths to split workgroups of 64 paths whose four waves each walk a QUARTER of the steps (one quarter of the
// canonical sum above) and meet in LDS: a quarter of the duration on all four SIMDs of a CU, so the remainder spreads
// over the chip in units four times finer -- and a launch smaller than one round of the chip (the interactive sizes)
// finishes its step loops in a quarter of the time.  Wave 0 of a split workgroup evaluates the payoffs of its 64 paths.
// Payoffs of one path (both legs) for every contract of the set, given its normal sum.
// The dead-lane trick below (a NaN normal sum makes every payoff of the lane an exact zero through fmax(NaN, 0) = 0, IEEE maxnum)
// holds only while the compiler may not assume finite arithmetic: under -ffinite-math-only / -ffast-math the ragged-end lanes
// would add garbage to every price without a diagnostic.
#if defined(__FINITE_MATH_ONLY__) && __FINITE_MATH_ONLY__
#error "olmc_kernels.h relies on NaN propagation through fmax (dead lanes carry a NaN normal sum): do not build with -ffinite-math-only / -ffast-math"
#endif
template <int NSETS, bool ANTI, int MODE>
__device__ __forceinline__ void european_payoffs(const ContractSet<NSETS>& cs, double zsum, bool live, int64_t i, int64_t count,
                                                 double* __restrict__ terminal, double (&acc)[(MODE == kControlVariate) ? 5 : 2 * NSETS]) {
    constexpr int NC = (MODE == kControlVariate) ? 5 : 2;
    // kReduce: a dead lane (the ragged end of the last workgroup) carries a NaN normal sum: every S_T of the lane is NaN and
    // fmax(NaN, 0) = 0 (IEEE maxnum), so its payoffs and their squares are exact zeros without a select per sample (64
    // v_cndmask for 16 contracts).  Contract parameters that are themselves NaN are answered on the host (olmc.hip, poisoned()).
    if constexpr (MODE == kReduce) zsum = live ? zsum : __builtin_nan("");
    double base_st[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        const Contract c = cs.c[s];
        const bool is_base = NSETS == 1 || ((cs.base_mask >> s) & 1u) != 0u;          // wave-uniform, on the scalar unit
        if (is_base) {
            // a real, wave-uniform branch around the two exponentials (the empty volatile asm keeps the block from being
            // speculated into selects): the 14 contracts of second-order Greeks have 4 distinct vols = 8 exps per path, not 28
            if constexpr (NSETS > 1) asm volatile("");
            const double dz = c.vol * zsum;
            // the library exp() (the reference's np.exp); the 15-instruction exp2_f64 here was measured at -0.4 % (1 contract) ... -1.3 %
            // (8 contracts) and not adopted: not worth moving the terminal prices 2-3 ulp away from libm
            base_st[0] = exp(c.a + dz);
            if constexpr (ANTI) base_st[1] = exp(c.a - dz);
        }
#pragma unroll
        for (int leg = 0; leg < (ANTI ? 2 : 1); ++leg) {
            const double st = NSETS == 1 ? base_st[leg] : c.scale * base_st[leg];      // scale = 1 exactly for a base
            if constexpr (MODE == kTerminal) {
                if (live) terminal[leg * count + i] = st;
            } else {
                const double x = fmax(c.sign * (st - c.strike), 0.0);
                double (&slot)[NC] = *reinterpret_cast<double (*)[NC]>(&acc[(MODE == kControlVariate) ? 0 : 2 * s]);
                if constexpr (MODE == kReduce) add_sample<MODE>(slot, x, st);
                else add_sample<MODE>(slot, live ? x : 0.0, live ? st : 0.0);
            }
        }
    }
}

// The two sums (x_u + x_d, x_u^2 + x_d^2) of ONE contract for this lane's path.  `base_st` is the pair of terminal prices of the
// latest base contract of the stream the caller walks (a base refreshes it, the others scale it).
template <bool ANTI, bool SQUARES = true>
__device__ __forceinline__ void contract_sums(const Contract& c, bool is_base, double zsum, double (&base_st)[2], double& sum, double& sumsq) {
    if (is_base) {                                   // wave-uniform scalar branch, kept real (see european_payoffs)
        asm volatile("");
        const double dz = c.vol * zsum;
        base_st[0] = exp(c.a + dz);
        if constexpr (ANTI) base_st[1] = exp(c.a - dz);
    }
    // sign_scale = +-1 exactly for a base: one fma for every contract, no select between "own" and "scaled" prices
    const double xu = fmax(__builtin_fma(c.sign_scale, base_st[0], c.neg_sign_strike), 0.0);
    sum = xu;                                        // add_sample<kReduce> on accumulators born at zero
    if constexpr (SQUARES) sumsq = xu * xu;
    if constexpr (ANTI) {
        const double xd = fmax(__builtin_fma(c.sign_scale, base_st[1], c.neg_sign_strike), 0.0);
        sum += xd;
        if constexpr (SQUARES) sumsq = __builtin_fma(xd, xd, sumsq);      // one rounding fewer than x_d^2 rounded, then added: the squares feed the standard error only
    }
}

// kReduce with 8 / 16 contracts on a launch that covers every path (one path per thread): the 2 NSETS per-lane sums are made
// two CONTRACTS at a time -- s and s + NSETS/2, i.e. values k and k + P/2 of the wave reduction -- and traded across the
// half-waves at once (swap_add<32>, the first step of wave_transpose_reduce), so a lane never holds more than NSETS sums:
// 32 instead of 64 VGPRs of accumulators for second-order Greeks, and the kernel keeps the step loop's own occupancy
// (round 2: 84 VGPRs, 5 waves per SIMD; the loop alone needs 66 = 7 waves).  The two half-sets are walked as two streams,
// each with its own latest base: the host lays the set out so that slot NSETS/2 is a base, or -- when the group that straddles
// the middle is the one of slot 0 -- tells the second stream to start from slot 0's prices (group_contracts).
// `zsum` is NaN in dead lanes (all their payoffs are then exact zeros, see european_payoffs).
//
// SQUARES = false (kSumOnly): only sum x per contract -- what finite-difference Greeks need (prices, no standard errors): NSETS
// values per lane, NSETS / 2 after the fold, two of the seven fp64 operations per contract and half of every exchange gone.
template <int NSETS, bool ANTI, bool SQUARES = true>
__device__ __forceinline__ void european_payoffs_folded(const ContractSet<NSETS>& cs, double zsum, double (&kept)[SQUARES ? NSETS : NSETS / 2]) {
    constexpr int H = NSETS / 2;
    double base_lo[2] = {0.0, 0.0}, base_hi[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < H; ++s) {
        double a0, a1 = 0.0, b0, b1 = 0.0;
        contract_sums<ANTI, SQUARES>(cs.c[s], ((cs.base_mask >> s) & 1u) != 0u, zsum, base_lo, a0, a1);
        if (s == 0 && cs.upper_continues_slot0 != 0u) {         // launch-uniform: the second stream starts inside slot 0's group
            base_hi[0] = base_lo[0];
            base_hi[1] = base_lo[1];
        }
        contract_sums<ANTI, SQUARES>(cs.c[s + H], ((cs.base_mask >> (s + H)) & 1u) != 0u, zsum, base_hi, b0, b1);
        if constexpr (SQUARES) {
            kept[2 * s] = swap_add<32>(a0, b0);
            kept[2 * s + 1] = swap_add<32>(a1, b1);
        } else {
            kept[s] = swap_add<32>(a0, b0);          // value s of NSETS meets value s + NSETS / 2
        }
    }
}

template <int NSETS, bool ANTI, int MODE, bool STRIDED>
__global__ __launch_bounds__(kBlock) void european_path_kernel(PathRange pr, ContractSet<NSETS> cs, ReduceWs ws,
                                                               double* __restrict__ terminal) {
    constexpr int NV = (MODE == kControlVariate) ? 5 : (MODE == kSumOnly ? NSETS : 2 * NSETS);
    static_assert(MODE != kSumOnly || (!STRIDED && NSETS > 1), "kSumOnly exists for the fused sets on launches that cover every path");
    if constexpr (STRIDED) {
        double acc[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] = 0.0;
        const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
            const uint64_t g = pr.first + static_cast<uint64_t>(i);
            const double zsum = path_normal_sum(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), pr.n_steps, pr.key0, pr.key1);
            european_payoffs<NSETS, ANTI, MODE>(cs, zsum, true, i, pr.count, terminal, acc);
        }
        if constexpr (MODE != kTerminal) block_then_grid_reduce<NV>(acc, ws);
    } else {
        __shared__ double quarter_sum[kWavesPerBlock][kWave];
        // readfirstlane: the wave index is uniform, and the compiler must KNOW it -- the Philox block counter derives from it in
        // a split workgroup, and a counter in SGPRs keeps the first round's multiply on the scalar unit (17 instead of 18
        // v_mad_u64_u32 per block)
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave), lane = threadIdx.x & (kWave - 1);
        const bool split = static_cast<int32_t>(blockIdx.x) >= pr.split_from;                    // workgroup-uniform
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
        const bool payer = !split || wave == 0;         // waves 1..3 of a split workgroup carry no path through the payoffs (wave-uniform)
        if constexpr ((MODE == kReduce || MODE == kSumOnly) && NSETS > 1) {
            constexpr int KEPT = NV / 2;                // born after the step loop, and only half of the NV sums (folded first exchange)
            double kept[KEPT];
#pragma unroll
            for (int k = 0; k < KEPT; ++k) kept[k] = 0.0;
            if (payer) european_payoffs_folded<NSETS, ANTI, MODE == kReduce>(cs, i < pr.count ? zsum : __builtin_nan(""), kept);
            block_then_grid_reduce_from<NV, 1>(kept, ws);
        } else {
            double acc[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] = 0.0;      // born after the step loop
            if (payer) european_payoffs<NSETS, ANTI, MODE>(cs, zsum, i < pr.count, i, pr.count, terminal, acc);
            if constexpr (MODE != kTerminal) block_then_grid_reduce<NV>(acc, ws);
        }
    }
}

// Many INDEPENDENT contracts in one launch (MonteCarloPricerUni.price_batch,
// src/pricing_models/monte_carlo_unified.py:562-631): blockIdx.y = contract, each contract has
// its own Philox stream (counter word 3 = its tag) like the reference's per-option slice of Z
// (:320) / per-option seed (:176), and its own last-arriver reduction: the workgroup that takes
// the last ticket of contract j sums that contract's rows in index order and writes out[j].
constexpr int kMultiCounterStride = 32;   // uint32 words between per-contract ticket counters (128 B)

struct MultiOption {
    double a, vol, strike, sign;    // as Contract
    uint32_t tag;                   // stream tag: equal tags => common random numbers
    uint32_t pad;
};

// Completion of a whole batch: out[] is pinned host memory; the finisher of every contract makes its two sums visible
// system-wide and takes an acquire-release ticket on `done_count`; the finisher that takes the LAST ticket of the batch (n_total
// contracts, over all launches of the call) re-zeroes the counter, releases at system scope and raises the host's completion word.
struct MultiDone {
    uint32_t* done_count;
    uint64_t* done_flag;     // NULL: the host waits for the stream instead
    uint64_t done_value;
    uint32_t n_total;
};

template <bool ANTI>
__global__ __launch_bounds__(kBlock) void european_multi_kernel(PathRange pr, const MultiOption* __restrict__ opts,
                                                                int64_t opt_base, double* __restrict__ block_rows,
                                                                uint32_t* __restrict__ counters, double* __restrict__ out, MultiDone md) {
    __shared__ double stage[kWavesPerBlock][2];
    const int64_t opt = opt_base + blockIdx.y;
    const MultiOption o = opts[opt];
    double acc[2] = {0.0, 0.0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const double dz = o.vol * path_normal_sum(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), pr.n_steps,
                                                  pr.key0, pr.key1, o.tag);
#pragma unroll
        for (int leg = 0; leg < (ANTI ? 2 : 1); ++leg) {
            const double x = fmax(o.sign * (exp(leg ? o.a - dz : o.a + dz) - o.strike), 0.0);
            acc[0] += x; acc[1] += x * x;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double s = wave_sum(acc[c]);
        if (lane == 0) stage[wave][c] = s;
    }
    __syncthreads();
    if (wave != 0) return;
    double v = 0.0;
    if (lane < 2) v = ((stage[0][lane] + stage[1][lane]) + stage[2][lane]) + stage[3][lane];
    double* rows = block_rows + static_cast<size_t>(opt) * gridDim.x * 2;
    if (lane < 2) store_sc1(rows + static_cast<size_t>(blockIdx.x) * 2 + lane, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(counters + static_cast<size_t>(opt) * kMultiCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != gridDim.x - 1) return;
    acquire_rows();
    const double total = wave_rows_sum<2>(rows, static_cast<int32_t>(gridDim.x));
    if (lane < 2) out[opt * 2 + lane] = total;
    if (lane == 0) __hip_atomic_store(counters + static_cast<size_t>(opt) * kMultiCounterStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (md.done_flag == nullptr) return;
    // Completion of the batch, by the memory model and not by what a fence happens to drain on gfx950: every contract's finisher
    //   (1) releases its two sums at SYSTEM scope (they are in host memory before anything ordered after the fence),
    //   (2) takes its ticket with an ACQUIRE-RELEASE add at agent scope -- the tickets form a release sequence on done_count, so the
    //       finisher that takes the last one synchronises with every earlier finisher and with all that they released,
    //   (3) the last finisher releases again at system scope and only then raises the host's word: the host's acquire load of the
    //       word therefore sees the sums of ALL contracts, whichever workgroup wrote them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    uint32_t fin = 0;
    if (lane == 0) fin = __hip_atomic_fetch_add(md.done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    fin = __builtin_amdgcn_readfirstlane(fin);
    if (fin != md.n_total - 1u) return;
    if (lane == 0) __hip_atomic_store(md.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope, as signal_done(): one fence, paid by the last workgroup only
    if (lane == 0) __hip_atomic_store(md.done_flag, md.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Asian option: running arithmetic sum of S_t (or sum of ln S_t) over t = 1..M kept in
// registers (exotic_options.py:59-67, 119-122 without the (n_paths, n_steps) matrix).
//
// Two-level arithmetic, like the European normal sum.  Per GROUP of 16 dates the path carries its
// cumulative log-return `base` in fp64 (advanced once per group with the fp64 vol and drift).  Inside
// a group only the prefix sum P_t of the group's RAW normals (parameter-free, fp32) moves:
//     y_t = base + vol * P_t + j * drift                      (j = date index within the group)
// arithmetic:  S_t / S_0 = 2^y_t  by one v_exp_f32 of fma(vol32, P_t, fp32(base)) + fp32(j drift):
//              4 fp32 VALU + 1 transcendental per date per leg, the group's 16 terms summed in fp32,
//              groups in fp64.  The exponent is rounded to fp32 as before (|y| < ~2: 1.2e-7 absolute,
//              unbiased); vol32's own rounding only touches the within-group part (< 1e-9).  The 16
//              values fp32(j drift) live in registers: a running b += drift32 instead would drop the
//              same sub-ulp fraction of drift32 sixteen times in a row -- a 2e-7 bias, measured.
// geometric:   sum_t y_t over a group = n base + vol * (sum_t P_t) + drift n(n+1)/2: two fp32 adds per
//              date for BOTH legs (the mirrored leg shares P), everything else once per group in fp64.
struct AsianContract {
    double log_s0;
    double s0;
    double drift;      // (r - q - sigma^2/2) dt
    double vol;        // sigma sqrt(dt)
    double strike;
    double sign;
    double inv_steps;  // 1 / M
};

constexpr int kAsianGroupBlocks = 4;      // Philox blocks (of four dates) per fp64 update

struct AsianGroup {      // fp32 state of the group in flight
    float p;             // prefix sum of the group's RAW normals
    float pp;            // geometric: sum over the group's dates of p
    float b_u, b_d;      // arithmetic: fp32(base) of the two legs
    float e_u, e_d;      // arithmetic: sum of the group's 2^y terms
};

// One Philox block = four monitoring dates of one path.  LIVE < 4 only for the trailing block.
// `jd` = fp32(j * drift) for the block's four dates (j counts from the start of the group).
template <bool ANTI, bool GEOMETRIC, int LIVE>
__device__ __forceinline__ void asian_block(const float (&z)[4], float vol32, const float* jd, AsianGroup& g) {
#pragma unroll
    for (int j = 0; j < LIVE; ++j) {
        g.p += z[j];
        if constexpr (GEOMETRIC) {
            g.pp += g.p;
        } else {
            g.e_u += __builtin_amdgcn_exp2f(__builtin_fmaf(vol32, g.p, g.b_u) + jd[j]);
            if constexpr (ANTI) g.e_d += __builtin_amdgcn_exp2f(__builtin_fmaf(-vol32, g.p, g.b_d) + jd[j]);
        }
    }
}

template <bool ANTI, bool GEOMETRIC>
__global__ __launch_bounds__(kBlock) void asian_kernel(PathRange pr, AsianContract c, ReduceWs ws) {
    constexpr double kLog2e = 1.4426950408889634;
    double acc[2] = {0.0, 0.0};
    // arithmetic: y = log2(S_t / S_0); geometric: y = ln(S_t / S_0)
    const double unit = GEOMETRIC ? 1.0 : kLog2e;
    const double drift = c.drift * unit;
    const double vol = c.vol * kZScale * unit;      // applied to RAW normals
    const float vol32 = static_cast<float>(vol);
    float jd[4 * kAsianGroupBlocks];
#pragma unroll
    for (int j = 0; j < 4 * kAsianGroupBlocks; ++j) jd[j] = static_cast<float>((j + 1) * drift);
    const int32_t full = pr.n_steps >> 2, rem = pr.n_steps & 3;
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);       // A/B at 1M x 1024: arithmetic 566 -> 550 us (antithetic 704 -> 692), geometric unchanged
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t gp = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(gp), g_hi = static_cast<uint32_t>(gp >> 32);
        double base_u = 0.0, base_d = 0.0;   // cumulative log-return at the start of the group (in `unit`s)
        double run_u = 0.0, run_d = 0.0;     // running sum of S_t / S_0, or of ln(S_t / S_0)
        AsianGroup g{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        // close a group of n dates: fold its fp32 sums into the fp64 running sums, advance the bases
        auto close_group = [&](int32_t n) {
            const double p = static_cast<double>(g.p), nd = static_cast<double>(n);
            if constexpr (GEOMETRIC) {
                const double tri = 0.5 * nd * (nd + 1.0), pp = static_cast<double>(g.pp);
                run_u += __builtin_fma(vol, pp, __builtin_fma(nd, base_u, drift * tri));
                if constexpr (ANTI) run_d += __builtin_fma(-vol, pp, __builtin_fma(nd, base_d, drift * tri));
            } else {
                run_u += static_cast<double>(g.e_u);
                if constexpr (ANTI) run_d += static_cast<double>(g.e_d);
            }
            base_u += __builtin_fma(vol, p, nd * drift);
            if constexpr (ANTI) base_d += __builtin_fma(-vol, p, nd * drift);
            g.p = 0.0f; g.pp = 0.0f; g.e_u = 0.0f; g.e_d = 0.0f;
            g.b_u = static_cast<float>(base_u);
            g.b_d = static_cast<float>(base_d);
        };
        float z[4];
        int32_t b = 0;
        for (; b + kAsianGroupBlocks <= full; b += kAsianGroupBlocks) {     // branch-free body: 16 dates schedule together
#pragma unroll
            for (int k = 0; k < kAsianGroupBlocks; ++k) {
                raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b + k), 0u, rk, z);
                asian_block<ANTI, GEOMETRIC, 4>(z, vol32, jd + 4 * k, g);
            }
            close_group(4 * kAsianGroupBlocks);
        }
        // the trailing partial group: up to three full blocks, then the block holding the last n_steps % 4 dates
        const int32_t tail_blocks = full - b;
#pragma unroll
        for (int k = 0; k < kAsianGroupBlocks; ++k) {
            if (k < tail_blocks) {
                raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b + k), 0u, rk, z);
                asian_block<ANTI, GEOMETRIC, 4>(z, vol32, jd + 4 * k, g);
            } else if (k == tail_blocks && rem) {
                raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(full), 0u, rk, z);
                if (rem == 1) asian_block<ANTI, GEOMETRIC, 1>(z, vol32, jd + 4 * k, g);
                else if (rem == 2) asian_block<ANTI, GEOMETRIC, 2>(z, vol32, jd + 4 * k, g);
                else asian_block<ANTI, GEOMETRIC, 3>(z, vol32, jd + 4 * k, g);
            }
        }
        const int32_t open_dates = 4 * tail_blocks + rem;
        if (open_dates) close_group(open_dates);
#pragma unroll
        for (int leg = 0; leg < (ANTI ? 2 : 1); ++leg) {
            const double mean = (leg ? run_d : run_u) * c.inv_steps;
            const double avg = GEOMETRIC ? exp(c.log_s0 + mean) : c.s0 * mean;
            const double x = fmax(c.sign * (avg - c.strike), 0.0);
            acc[0] += x; acc[1] += x * x;
        }
    }
    block_then_grid_reduce<2>(acc, ws);
}

// Arithmetic Asian at the REFERENCE's precision (exotic_options.py:59-67, 119-122): the cumulative log-return is a
// running fp64 sum advanced date by date (the reference's cumsum), every monitoring date takes a full fp64 exponential
// of it (the reference's np.exp(log_S)), the running sum of S_t / S_0 is fp64.  Only the normals are fp32, as in every
// kernel of this engine.  This is OLMC_AVG_ARITHMETIC; asian_kernel<ANTI, false> above (one v_exp_f32 per date on an
// exponent rounded to fp32) is the opt-in OLMC_AVG_ARITHMETIC_FAST.  Per date and leg: fma + add (cumsum), 15 (exp2),
// add (running sum), + one v_cvt_f64_f32 shared by the legs.
template <bool ANTI, int LIVE>
__device__ __forceinline__ void asian_exp64_block(const float (&z)[4], double drift, double vol, double& cum_u, double& run_u,
                                                  double& cum_d, double& run_d, const double* tab) {
#pragma unroll
    for (int j = 0; j < LIVE; ++j) {
        const double zj = static_cast<double>(z[j]);
        cum_u += __builtin_fma(vol, zj, drift);
        if constexpr (OLMC_EXP2_TABLE) run_u += exp2_f64_tab(cum_u, tab);
        else run_u += exp2_f64(cum_u);
        if constexpr (ANTI) {
            cum_d += __builtin_fma(-vol, zj, drift);
            if constexpr (OLMC_EXP2_TABLE) run_d += exp2_f64_tab(cum_d, tab);
            else run_d += exp2_f64(cum_d);
        }
    }
}

template <bool ANTI>
__global__ __launch_bounds__(kBlock) void asian_exp64_kernel(PathRange pr, AsianContract c, ReduceWs ws) {
    constexpr double kLog2e = 1.4426950408889634;
    constexpr double kUnit = OLMC_EXP2_TABLE ? kExp2Entries * kLog2e : kLog2e;     // the exponent is carried in the units the exponential takes: 1/256 log2 (table) or log2
    __shared__ double tab[kExp2Entries];
    if constexpr (OLMC_EXP2_TABLE) exp2_table_to_lds(tab);
    double acc[2] = {0.0, 0.0};
    const double drift = c.drift * kUnit;               // no argument scaling per date
    const double vol = c.vol * kZScale * kUnit;         // applied to RAW normals
    const int32_t full = pr.n_steps >> 2, rem = pr.n_steps & 3;
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);       // A/B at 1M x 1024: 965 -> 940 us (antithetic 1478 -> 1467)
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t gp = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(gp), g_hi = static_cast<uint32_t>(gp >> 32);
        double cum_u = 0.0, cum_d = 0.0, run_u = 0.0, run_d = 0.0;
        float z[4];
        for (int32_t b = 0; b < full; ++b) {            // branch-free body
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk, z);
            asian_exp64_block<ANTI, 4>(z, drift, vol, cum_u, run_u, cum_d, run_d, tab);
        }
        if (rem) {
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(full), 0u, rk, z);
            if (rem == 1) asian_exp64_block<ANTI, 1>(z, drift, vol, cum_u, run_u, cum_d, run_d, tab);
            else if (rem == 2) asian_exp64_block<ANTI, 2>(z, drift, vol, cum_u, run_u, cum_d, run_d, tab);
            else asian_exp64_block<ANTI, 3>(z, drift, vol, cum_u, run_u, cum_d, run_d, tab);
        }
#pragma unroll
        for (int leg = 0; leg < (ANTI ? 2 : 1); ++leg) {
            const double avg = c.s0 * ((leg ? run_d : run_u) * c.inv_steps);
            const double x = fmax(c.sign * (avg - c.strike), 0.0);
            acc[0] += x; acc[1] += x * x;
        }
    }
    block_then_grid_reduce<2>(acc, ws);
}

// Finite-difference Greeks of the arithmetic Asian in ONE launch (round 4): the 8 / 14 bumped contracts of compute_greeks_unified
// through ExoticAdapter(AsianOption) (src/greeks/unified_greeks.py:177-227, 295-358) on the SAME normals.  A contract enters the
// date loop only through its per-step drift and volatility; the spot scales the average (S_t / S_0 does not depend on S) and the
// strike and the discount act after the loop.  So the 8 / 14 contracts are at most kAsianGroups = 6 distinct PATH RECURSIONS --
// {mid, S+, S-}, sigma+, sigma-, T-, r+, r-; the second-order set adds contracts ((S+-, sigma+-), (S+-, T-)), no recursion -- and
// a date costs the normal (13.5 instructions) + 6 x (fma, add, 12-instruction exp2, add) instead of 8 / 14 x 29 in 8 / 14
// launches.  Each recursion is asian_exp64_kernel's own arithmetic (same scaling of drift and vol into exponent units, same
// exp2_f64_tab, same S_0 (run / M)), so a contract's payoffs are the bits its own launch produces; only the association of the sums
// differs (16 / 32 values per workgroup row).

template <bool ANTI, int NSETS>
__global__ __launch_bounds__(kBlock) void asian_exp64_greeks_kernel(PathRange pr, AsianGreeksSet gs, ReduceWs ws) {
    constexpr int NV = 2 * NSETS, G = kAsianGroups, R = kAsianRealGroups, D = G - R, LEGS = ANTI ? 2 : 1;
    constexpr int32_t kRefreshBlocks = 64;              // the rate factors are re-anchored on the library exponential every 256 dates
    __shared__ double tab[kExp2Entries];
    if constexpr (OLMC_EXP2_TABLE) exp2_table_to_lds(tab);
    double acc[NV];                                     // the grid covers every path (host guarantee): born after the date loop
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    {
        const double* __restrict__ drift = gs.drift;    // already in exponent units (host: the two multiplications of asian_exp64_kernel)
        const double* __restrict__ vol = gs.vol;
        const uint64_t gp = pr.first + static_cast<uint64_t>(i < pr.count ? i : 0);
        const uint32_t g_lo = static_cast<uint32_t>(gp), g_hi = static_cast<uint32_t>(gp >> 32);
        double cum[LEGS][R], run[LEGS][G];
#pragma unroll
        for (int leg = 0; leg < LEGS; ++leg) {
#pragma unroll
            for (int g = 0; g < G; ++g) run[leg][g] = 0.0;
#pragma unroll
            for (int g = 0; g < R; ++g) cum[leg][g] = 0.0;
        }
        // A bump of r moves the per-step drift and nothing else: the price relative of such a contract at date t is slot 0's times
        // exp(rate_step t) = 1 + g_t, the same number for every path.  The rider carries g_t (~1e-5: its rounding is 1e-21 of the factor)
        // from date to date as g += e + g e, e = expm1(rate_step), sums x g beside slot 0's sum of x, and is slot 0's sum plus that.
        // (Two earlier forms and what they cost: f *= exp(rate_step) commits the same 1e-16 at every date -- the r + h price moved by
        // 1.2e-13 at 1,024 dates; f += f e is unbiased but its random walk, ~8 ulp after 256 dates, is common to all paths and an
        // at-the-money payoff K - avg magnifies it a hundredfold: 1.06e-13 in the property hunt.)  g is set afresh from expm1() every
        // 256 dates.
        double growth[D], growth_step[D], rider[LEGS][D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            growth[d] = 0.0;
            growth_step[d] = expm1(gs.rate_step[d]);
#pragma unroll
            for (int leg = 0; leg < LEGS; ++leg) rider[leg][d] = 0.0;
        }
        auto dates = [&](const float (&z)[4], auto live) {      // `live` dates of one Philox block, every recursion
#pragma unroll
            for (int j = 0; j < decltype(live)::value; ++j) {
                const double zj = static_cast<double>(z[j]);
                double x0[LEGS];
#pragma unroll
                for (int g = 0; g < R; ++g) {
                    cum[0][g] += __builtin_fma(vol[g], zj, drift[g]);
                    const double x = OLMC_EXP2_TABLE ? exp2_f64_tab(cum[0][g], tab) : exp2_f64(cum[0][g]);
                    run[0][g] += x;
                    if (g == 0) x0[0] = x;
                    if constexpr (ANTI) {
                        cum[1][g] += __builtin_fma(-vol[g], zj, drift[g]);
                        const double y = OLMC_EXP2_TABLE ? exp2_f64_tab(cum[1][g], tab) : exp2_f64(cum[1][g]);
                        run[1][g] += y;
                        if (g == 0) x0[1] = y;
                    }
                }
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    growth[d] = __builtin_fma(growth[d], growth_step[d], growth[d]) + growth_step[d];
#pragma unroll
                    for (int leg = 0; leg < LEGS; ++leg) rider[leg][d] = __builtin_fma(x0[leg], growth[d], rider[leg][d]);
                }
            }
        };
        const int32_t full = pr.n_steps >> 2, rem = pr.n_steps & 3;
        float z[4];
        for (int32_t b0 = 0; b0 < full; b0 += kRefreshBlocks) {
#pragma unroll
            for (int d = 0; d < D; ++d) growth[d] = expm1(gs.rate_step[d] * static_cast<double>(4 * b0));
            const int32_t b1 = min(b0 + kRefreshBlocks, full);
            for (int32_t b = b0; b < b1; ++b) {                 // branch-free body
                raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk, z);
                dates(z, std::integral_constant<int, 4>{});
            }
        }
        if (rem) {
#pragma unroll
            for (int d = 0; d < D; ++d) growth[d] = expm1(gs.rate_step[d] * static_cast<double>(4 * full));
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(full), 0u, rk, z);
            if (rem == 1) dates(z, std::integral_constant<int, 1>{});
            else if (rem == 2) dates(z, std::integral_constant<int, 2>{});
            else dates(z, std::integral_constant<int, 3>{});
        }
        const bool alive = i < pr.count;
#pragma unroll
        for (int leg = 0; leg < LEGS; ++leg)
#pragma unroll
            for (int d = 0; d < D; ++d) run[leg][R + d] = run[leg][0] + rider[leg][d];
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            const int32_t g = gs.group[s];                       // launch-uniform
            acc[2 * s] = acc[2 * s + 1] = 0.0;
#pragma unroll
            for (int leg = 0; leg < LEGS; ++leg) {
                double r = run[leg][0];
#pragma unroll
                for (int k = 1; k < G; ++k) r = g == k ? run[leg][k] : r;
                const double avg = gs.s0[s] * (r * gs.inv_steps);
                const double x = alive ? fmax(gs.sign * (avg - gs.strike), 0.0) : 0.0;
                acc[2 * s] += x;
                acc[2 * s + 1] += x * x;
            }
        }
    }
    block_then_grid_reduce<NV>(acc, ws);
}

// The geometric average (asian_kernel<., true>) under the same 8 / 14 bumps.  Here even the six recursions share almost everything:
// inside a group of 16 dates only the fp32 prefix sum p of the group's RAW normals and the sum pp of those prefixes move, and neither
// depends on the contract; a recursion enters when a group CLOSES -- run += vol pp + n base + drift n(n+1)/2, base += vol p + n drift,
// five fp64 operations per 16 dates.  All 14 contracts of second-order Greeks therefore cost ONE geometric pricing (+ 6 x 5 fp64
// operations per 16 dates and 14 exponentials per path).  Same operations in the same order as asian_kernel<ANTI, true> per recursion.
template <bool ANTI, int NSETS>
__global__ __launch_bounds__(kBlock) void asian_geometric_greeks_kernel(PathRange pr, AsianGreeksSet gs, ReduceWs ws) {
    constexpr int NV = 2 * NSETS, G = kAsianGroups, LEGS = ANTI ? 2 : 1;
    double acc[NV];                                     // the grid covers every path (host guarantee): born after the date loop
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const uint64_t gp = pr.first + static_cast<uint64_t>(i < pr.count ? i : 0);
    const uint32_t g_lo = static_cast<uint32_t>(gp), g_hi = static_cast<uint32_t>(gp >> 32);
    double base[LEGS][G], run[LEGS][G];
#pragma unroll
    for (int leg = 0; leg < LEGS; ++leg)
#pragma unroll
        for (int g = 0; g < G; ++g) base[leg][g] = run[leg][g] = 0.0;
    float p = 0.0f, pp = 0.0f;                          // the group in flight: prefix sum of its raw normals, sum of those prefixes
    auto close_group = [&](int32_t n) {
        const double pd = static_cast<double>(p), ppd = static_cast<double>(pp), nd = static_cast<double>(n);
        const double tri = 0.5 * nd * (nd + 1.0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            run[0][g] += __builtin_fma(gs.vol[g], ppd, __builtin_fma(nd, base[0][g], gs.drift[g] * tri));
            base[0][g] += __builtin_fma(gs.vol[g], pd, nd * gs.drift[g]);
            if constexpr (ANTI) {
                run[1][g] += __builtin_fma(-gs.vol[g], ppd, __builtin_fma(nd, base[1][g], gs.drift[g] * tri));
                base[1][g] += __builtin_fma(-gs.vol[g], pd, nd * gs.drift[g]);
            }
        }
        p = 0.0f; pp = 0.0f;
    };
    auto dates = [&](const float (&z)[4], auto live) {
#pragma unroll
        for (int j = 0; j < decltype(live)::value; ++j) { p += z[j]; pp += p; }
    };
    const int32_t full = pr.n_steps >> 2, rem = pr.n_steps & 3;
    float z[4];
    int32_t b = 0;
    for (; b + kAsianGroupBlocks <= full; b += kAsianGroupBlocks) {
#pragma unroll
        for (int k = 0; k < kAsianGroupBlocks; ++k) {
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b + k), 0u, rk, z);
            dates(z, std::integral_constant<int, 4>{});
        }
        close_group(4 * kAsianGroupBlocks);
    }
    const int32_t tail_blocks = full - b;
#pragma unroll
    for (int k = 0; k < kAsianGroupBlocks; ++k) {
        if (k < tail_blocks) {
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b + k), 0u, rk, z);
            dates(z, std::integral_constant<int, 4>{});
        } else if (k == tail_blocks && rem) {
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(full), 0u, rk, z);
            if (rem == 1) dates(z, std::integral_constant<int, 1>{});
            else if (rem == 2) dates(z, std::integral_constant<int, 2>{});
            else dates(z, std::integral_constant<int, 3>{});
        }
    }
    const int32_t open_dates = 4 * tail_blocks + rem;
    if (open_dates) close_group(open_dates);
    const bool alive = i < pr.count;
#pragma unroll 1
    for (int s = 0; s < NSETS; ++s) {                   // a real loop: one library exponential per contract and leg
        const int32_t g = gs.group[s];                  // launch-uniform
        double sum = 0.0, sumsq = 0.0;
#pragma unroll
        for (int leg = 0; leg < LEGS; ++leg) {
            double r = run[leg][0];
#pragma unroll
            for (int k = 1; k < G; ++k) r = g == k ? run[leg][k] : r;
            const double avg = exp(gs.log_s0[s] + r * gs.inv_steps);
            const double x = alive ? fmax(gs.sign * (avg - gs.strike), 0.0) : 0.0;
            sum += x;
            sumsq += x * x;
        }
#pragma unroll
        for (int k = 0; k < NSETS; ++k)
            if (k == s) { acc[2 * k] = sum; acc[2 * k + 1] = sumsq; }
    }
    block_then_grid_reduce<NV>(acc, ws);
}

// Barrier and lookback options: both depend on the path only through its terminal value and
// its running extrema, and exp is monotone, so the step loop tracks max / min of the cumulative
// LOG-return (t = 0 included, as the reference's paths[:, 0] = S is: exotic_options.py:64-67,
// 200-204, 379-381) in fp64 -- three fp64 ops per leg per step, no per-step exp.
//   barrier  (exotic_options.py:174-224): crossed = max >= ln(B/S) (up) | min <= ln(B/S) (down);
//            knock-out pays if !crossed, knock-in if crossed; payoff max(+-(S_T - K), 0).
//   lookback (exotic_options.py:359-401):  floating call S_T - S_min, floating put S_max - S_T,
//            fixed call max(S_max - K, 0), fixed put max(K - S_min, 0).

struct ExtremaContract {
    double s0, log_barrier_rel;   // ln(B / S0)
    double drift, vol;            // per step: (r - q - sigma^2/2) dt, sigma sqrt(dt)
    double strike, sign;
    int32_t payoff;               // ExtremaPayoff
    int32_t pad;
};

__device__ __forceinline__ double extrema_payoff(const ExtremaContract& c, double cum, double mx, double mn) {
    const double st = c.s0 * exp(cum);
    if (c.payoff <= kBarrierDownIn) {
        const bool up = c.payoff <= kBarrierUpIn;
        const bool crossed = up ? (mx >= c.log_barrier_rel) : (mn <= c.log_barrier_rel);
        const bool knock_out = (c.payoff == kBarrierUpOut) || (c.payoff == kBarrierDownOut);
        const bool active = knock_out ? !crossed : crossed;
        return active ? fmax(c.sign * (st - c.strike), 0.0) : 0.0;
    }
    if (c.payoff == kLookbackFloating) return c.sign > 0.0 ? st - c.s0 * exp(mn) : c.s0 * exp(mx) - st;
    return c.sign > 0.0 ? fmax(c.s0 * exp(mx) - c.strike, 0.0) : fmax(c.strike - c.s0 * exp(mn), 0.0);
}

// Running extrema: v_max_f64 / v_min_f64 as single instructions.  fmax() / fmin() reach the same instruction, but the compiler
// first CANONICALISES any operand it cannot prove quiet (`v_max_f64 x, x, x`) -- and a running extremum carried around a loop is such
// an operand at the head of every trip: two extra fp64 instructions per recursion per Philox block (a third of the max / min work of
// the barrier / lookback kernels: 6 v_max + 4 v_min where 4 + 4 are needed).  The hardware instruction quiets a signalling NaN by
// itself (IEEE mode), both operands here are sums of finite terms or earlier extrema, and NaN parameters never reach a kernel
// (olmc_host_math.h poisoned()).  Same bits as fmax / fmin for every non-NaN input.
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool ANTI, int LIVE>
__device__ __forceinline__ void extrema_block(const float (&z)[4], double drift, double vol, double& cum_u, double& mx_u,
                                              double& mn_u, double& cum_d, double& mx_d, double& mn_d) {
#pragma unroll
    for (int j = 0; j < LIVE; ++j) {
        const double zj = static_cast<double>(z[j]);
        cum_u += __builtin_fma(vol, zj, drift);
        mx_u = max_f64(mx_u, cum_u);
        mn_u = min_f64(mn_u, cum_u);
        if constexpr (ANTI) {
            cum_d += __builtin_fma(-vol, zj, drift);
            mx_d = max_f64(mx_d, cum_d);
            mn_d = min_f64(mn_d, cum_d);
        }
    }
}

template <bool ANTI>
__global__ __launch_bounds__(kBlock) void extrema_kernel(PathRange pr, ExtremaContract c, ReduceWs ws) {
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    double acc[2] = {0.0, 0.0};
    const double vol = c.vol * kZScale;             // applied to RAW normals
    const int32_t full = pr.n_steps >> 2, rem = pr.n_steps & 3;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double cum_u = 0.0, mx_u = 0.0, mn_u = 0.0, cum_d = 0.0, mx_d = 0.0, mn_d = 0.0;   // t = 0: ln(S_0/S_0) = 0
        float z[4];
        for (int32_t b = 0; b < full; ++b) {           // branch-free body
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk, z);
            extrema_block<ANTI, 4>(z, c.drift, vol, cum_u, mx_u, mn_u, cum_d, mx_d, mn_d);
        }
        if (rem) {
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(full), 0u, rk, z);
            if (rem == 1) extrema_block<ANTI, 1>(z, c.drift, vol, cum_u, mx_u, mn_u, cum_d, mx_d, mn_d);
            else if (rem == 2) extrema_block<ANTI, 2>(z, c.drift, vol, cum_u, mx_u, mn_u, cum_d, mx_d, mn_d);
            else extrema_block<ANTI, 3>(z, c.drift, vol, cum_u, mx_u, mn_u, cum_d, mx_d, mn_d);
        }
        const double xu = extrema_payoff(c, cum_u, mx_u, mn_u);
        acc[0] += xu; acc[1] += xu * xu;
        if constexpr (ANTI) {
            const double xd = extrema_payoff(c, cum_d, mx_d, mn_d);
            acc[0] += xd; acc[1] += xd * xd;
        }
    }
    block_then_grid_reduce<2>(acc, ws);
}

// Finite-difference Greeks of a barrier / lookback option in ONE launch (round 4; the live caller is streamlit_app/pages/
// 7_Exotic_Options.py:266-284: compute_greeks_unified over ExoticAdapter(BarrierOption | LookbackOption)).  As for the Asian
// (asian_exp64_greeks_kernel) a contract enters the step loop only through its drift and vol per step: the 8 / 14 contracts are at most
// six recursions of (cumulative log-return, its running max, its running min); spot, strike and the barrier's level relative to the
// spot act in the epilogue.  Each recursion is extrema_kernel's own arithmetic, each payoff extrema_payoff's.

template <bool ANTI, int NSETS>
__global__ __launch_bounds__(kBlock) void extrema_greeks_kernel(PathRange pr, ExtremaGreeksSet gs, ReduceWs ws) {
    constexpr int NV = 2 * NSETS, G = kAsianGroups, LEGS = ANTI ? 2 : 1;
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;       // the grid covers every path (host guarantee)
    const uint64_t gp = pr.first + static_cast<uint64_t>(i < pr.count ? i : 0);
    const uint32_t g_lo = static_cast<uint32_t>(gp), g_hi = static_cast<uint32_t>(gp >> 32);
    double cum[LEGS][G], mx[LEGS][G], mn[LEGS][G];
#pragma unroll
    for (int leg = 0; leg < LEGS; ++leg)
#pragma unroll
        for (int g = 0; g < G; ++g) cum[leg][g] = mx[leg][g] = mn[leg][g] = 0.0;         // t = 0: ln(S_0 / S_0) = 0
    auto dates = [&](const float (&z)[4], auto live) {
#pragma unroll
        for (int j = 0; j < decltype(live)::value; ++j) {
            const double zj = static_cast<double>(z[j]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cum[0][g] += __builtin_fma(gs.vol[g], zj, gs.drift[g]);
                mx[0][g] = max_f64(mx[0][g], cum[0][g]);
                mn[0][g] = min_f64(mn[0][g], cum[0][g]);
                if constexpr (ANTI) {
                    cum[1][g] += __builtin_fma(-gs.vol[g], zj, gs.drift[g]);
                    mx[1][g] = max_f64(mx[1][g], cum[1][g]);
                    mn[1][g] = min_f64(mn[1][g], cum[1][g]);
                }
            }
            if constexpr (ANTI) __builtin_amdgcn_sched_barrier(0);       // a date at a time: twelve chains already fill the pipeline, and
        }                                                                  // four dates in flight at once cost 40 more registers
    };
    const int32_t full = pr.n_steps >> 2, rem = pr.n_steps & 3;
    float z[4];
    for (int32_t b = 0; b < full; ++b) {                // branch-free body
        raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk, z);
        dates(z, std::integral_constant<int, 4>{});
    }
    if (rem) {                                          // the last one to three dates: ONE copy of the date body in a real loop (three
        raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(full), 0u, rk, z);       // unrolled variants were where the register count peaked)
#pragma unroll 1
        for (int32_t j = 0; j < rem; ++j) {
            const float one[4] = {j == 0 ? z[0] : (j == 1 ? z[1] : z[2]), 0.f, 0.f, 0.f};
            dates(one, std::integral_constant<int, 1>{});
        }
    }
    // Epilogue, a contract at a time.  Round 4 kept every contract's {sum, sumsq} in registers until one transpose-reduce at the end:
    // 32 doubles beside the 36 of the recursions -- 172 VGPRs for the antithetic second-order kernel, two waves per SIMD.  Here a
    // contract's two values are folded over the wave as soon as they exist (wave_sum: fixed order) and parked in LDS; the workgroup
    // row is the four waves' entries added in wave order, as in block_then_grid_reduce.  Nothing but the recursions stays live.
    __shared__ double stage[kWavesPerBlock][NV];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const bool alive = i < pr.count;
#pragma unroll 1
    for (int s = 0; s < NSETS; ++s) {                   // a real loop: the payoff holds up to three library exponentials
        const int32_t g = gs.group[s];                  // launch-uniform
        ExtremaContract c;
        c.s0 = gs.s0[s]; c.log_barrier_rel = gs.log_barrier_rel[s]; c.drift = 0.0; c.vol = 0.0; c.strike = gs.strike; c.sign = gs.sign;
        c.payoff = gs.payoff; c.pad = 0;
        double sum = 0.0, sumsq = 0.0;
#pragma unroll
        for (int leg = 0; leg < LEGS; ++leg) {
            double a = cum[leg][0], b = mx[leg][0], d = mn[leg][0];
#pragma unroll
            for (int k = 1; k < G; ++k) {
                a = g == k ? cum[leg][k] : a;
                b = g == k ? mx[leg][k] : b;
                d = g == k ? mn[leg][k] : d;
            }
            const double x = alive ? extrema_payoff(c, a, b, d) : 0.0;
            sum += x;
            sumsq += x * x;
        }
        const double wave_total = wave_sum(sum), wave_total_sq = wave_sum(sumsq);       // valid in lane 0
        if (lane == 0) { stage[wave][2 * s] = wave_total; stage[wave][2 * s + 1] = wave_total_sq; }
    }
    __syncthreads();
    double row = 0.0;
    if (threadIdx.x < NV) {
        row = stage[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < kWavesPerBlock; ++k) row += stage[k][threadIdx.x];
    }
    grid_reduce_workgroup<NV>(row, ws);
}

// Structured products on the step loop, observation dates counted down in a scalar register.
//   autocallable (exotic_options.py:404-491): on every observation date t = f, 2f, ... <= M an
//     unredeemed path with S_t/S_0 >= autocall_barrier redeems (1 + c (i+1)/n_obs T) e^{-r t dt};
//     at maturity the rest get 1 (+ c T if S_T/S_0 >= coupon_barrier), or S_T/S_0 if the path ever
//     touched ki_barrier (min over ALL steps, t = 0 included) and ends below 1; x e^{-rT}.
//     Barriers are compared in log space; the payoff already carries its discount.
//   cliquet (exotic_options.py:494-554): sum over n_periods of clip(S_end/S_start - 1, floor, cap),
//     clipped globally, payoff max(total, 0) * S_0; periods are M // n_periods steps long.
struct AutocallContract {
    double drift, vol;                     // per step
    double log_autocall, log_coupon, log_ki;
    double coupon_unit;                    // coupon_rate * T / n_obs: coupon accrued per observation
    double final_coupon;                   // coupon_rate * T
    double obs_df, final_df;               // exp(-r dt obs_freq), exp(-r T)
    int32_t obs_freq, n_obs;
};

template <bool ANTI>
__global__ __launch_bounds__(kBlock) void autocall_kernel(PathRange pr, AutocallContract c, ReduceWs ws) {
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    double acc[2] = {0.0, 0.0};
    const double vol = c.vol * kZScale;
    constexpr int LEGS = ANTI ? 2 : 1;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double cum[2] = {0.0, 0.0}, mn[2] = {0.0, 0.0}, pay[2] = {0.0, 0.0};
        bool redeemed[2] = {false, false};
        // wave-uniform running values of the redemption amount (1 + coupon_k) * exp(-r t_k): no exp, no division in the loop
        int32_t until_obs = c.obs_freq;
        double coupon = 0.0, df = 1.0;
        const int32_t blocks = (pr.n_steps + 3) >> 2;
        for (int32_t b = 0; b < blocks; ++b) {
            float z[4];
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk, z);
            if (until_obs > 4 && 4 * b + 4 <= pr.n_steps) {
                // no observation date among these four steps (16 of 21 blocks at monthly observation): only the
                // cumulative return and its running minimum move -- branch-free, the four dates schedule together
                asm volatile("; olmc_fast_trip");       // no instruction: a label in the assembly by which tools/isa_mix.py finds the fast path of this loop
                until_obs -= 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double zj = static_cast<double>(z[j]);
#pragma unroll
                    for (int leg = 0; leg < LEGS; ++leg) {
                        cum[leg] += __builtin_fma(leg ? -vol : vol, zj, c.drift);
                        mn[leg] = min_f64(mn[leg], cum[leg]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (4 * b + j < pr.n_steps) {
                    const double zj = static_cast<double>(z[j]);
                    const bool observe = (--until_obs == 0);
                    double redemption = 0.0;
                    if (observe) {
                        until_obs = c.obs_freq;
                        coupon += c.coupon_unit;
                        df *= c.obs_df;
                        redemption = (1.0 + coupon) * df;
                    }
#pragma unroll
                    for (int leg = 0; leg < LEGS; ++leg) {
                        cum[leg] += __builtin_fma(leg ? -vol : vol, zj, c.drift);
                        mn[leg] = min_f64(mn[leg], cum[leg]);
                        const bool call_now = observe && !redeemed[leg] && cum[leg] >= c.log_autocall;
                        pay[leg] = call_now ? redemption : pay[leg];
                        redeemed[leg] = redeemed[leg] || call_now;
                    }
                }
            }
        }
#pragma unroll
        for (int leg = 0; leg < LEGS; ++leg) {
            double x = pay[leg];
            if (!redeemed[leg]) {
                double fin = 1.0;
                if (cum[leg] >= c.log_coupon) fin += c.final_coupon;
                if (mn[leg] <= c.log_ki && cum[leg] < 0.0) fin = exp(cum[leg]);
                x = fin * c.final_df;
            }
            acc[0] += x; acc[1] += x * x;
        }
    }
    block_then_grid_reduce<2>(acc, ws);
}

struct CliquetContract {
    double s0, drift, vol;
    double local_cap, local_floor, global_cap, global_floor;
    int32_t steps_per_period, n_periods;
};

// A period's return exp(ln S_end - ln S_start) - 1 depends on the path only through the SUM of the period's
// normals: ln S_end - ln S_start = steps_per_period * drift +- vol * sum Z.  So blocks of four steps that hold
// no period end run the terminal-price kernel's own accumulation (raw_block_accumulate: fp32 within 16
// normals, fp64 across) and only the blocks with a reset date look at single steps.
template <bool ANTI>
__global__ __launch_bounds__(kBlock) void cliquet_kernel(PathRange pr, CliquetContract c, ReduceWs ws) {
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    double acc[2] = {0.0, 0.0};
    const double vol = c.vol * kZScale;
    constexpr int LEGS = ANTI ? 2 : 1;
    const int32_t used_steps = c.steps_per_period * c.n_periods;      // trailing steps never enter a period
    const double period_drift = c.steps_per_period * c.drift;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double total[2] = {0.0, 0.0};
        double psum = 0.0;            // RAW normal sum of the period in flight (fp64 part)
        float part = 0.0f;            // ... and the blocks not yet folded into it
        int32_t part_blocks = 0;
        int32_t until_reset = c.steps_per_period;
        const int32_t blocks = (used_steps + 3) >> 2;
        for (int32_t b = 0; b < blocks; ++b) {
            if (until_reset > 4) {                      // no period end among these four steps (and all four are used)
                asm volatile("; olmc_fast_trip");       // (tools/isa_mix.py: the fast path of this loop)
                until_reset -= 4;
                part = raw_block_sum_of_words(part, philox4x32_10_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk));
                if (++part_blocks == kGroup) { psum += static_cast<double>(part); part = 0.0f; part_blocks = 0; }
                continue;
            }
            psum += static_cast<double>(part);
            part = 0.0f;
            part_blocks = 0;
            float z[4];
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), 0u, rk, z);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (4 * b + j < used_steps) {
                    psum += static_cast<double>(z[j]);
                    if (--until_reset == 0) {
                        until_reset = c.steps_per_period;
#pragma unroll
                        for (int leg = 0; leg < LEGS; ++leg) {
                            const double local = exp(__builtin_fma(leg ? -vol : vol, psum, period_drift)) - 1.0;   // (S_end - S_start) / S_start
                            total[leg] += fmin(fmax(local, c.local_floor), c.local_cap);
                        }
                        psum = 0.0;
                    }
                }
            }
        }
#pragma unroll
        for (int leg = 0; leg < LEGS; ++leg) {
            const double clipped = fmin(fmax(total[leg], c.global_floor), c.global_cap);
            const double x = fmax(clipped, 0.0) * c.s0;
            acc[0] += x; acc[1] += x * x;
        }
    }
    block_then_grid_reduce<2>(acc, ws);
}

// American option, Longstaff-Schwartz least-squares Monte Carlo (exotic_options.py:227-305).
// The only path here that must STORE paths: time-major paths[t][i] (t = 0..M) so that a time
// slice is one coalesced stream; n_paths * (M+1) * 8 bytes (20 MB at the reference's defaults).
// Backward induction is one launch per exercise date t = M-1 .. 1:
//   1. finish date t+1: if its regression was valid, in-the-money paths whose intrinsic value
//      beats the fitted continuation value exercise (cash flow := intrinsic);
//   2. discount the cash flow one step;
//   3. accumulate, over paths in the money at t, the normal-equation moments of the polynomial
//      regression of cash flow on x = S_t / K:  sum x^m (m = 0..2d), sum x^k cf (k = 0..d), count
//      -- through the fused deterministic grid reduction.
// The basis spans the reference's raw powers X^k (same polynomial space, so the same least-squares fit in exact arithmetic),
// written in the STANDARDISED regressor z = (S/K - c_t) / w_t, where c_t and w_t are the mean and standard deviation of S_t/K over
// the in-the-money side of the strike under the model's own lognormal law (closed form, computed on the host per date:
// lsm_regressor_scale in olmc.hip).  The reference's lstsq(rcond=None) -- an SVD -- is replaced by the normal equations, which
// square the condition number: in powers of S/K itself the moment matrix of a degree-4 fit over a narrow in-the-money range (deep
// in the money, low vol, an early date) has a condition number beyond 1e14 and fp64 leaves nothing of the coefficients (round 4's
// property hunt: device and checker disagreed on exercise decisions at 236 paths, degree 4); in z it stays below ~1e6.
constexpr int kLsmMaxDegree = 4;
constexpr int kLsmNV = 16;          // 2d+1 + d+1 + 1 <= 15 sums, padded

struct LsmContract {
    double log_s0, drift, vol;      // per step
    double s_first;                 // value stored at t = 0: exp(ln S) for the exotics' path builder (exotic_options.py:64-67),
                                    // S itself for simulate_gbm_paths (gbm_numpy.py:115)
    double strike, inv_strike, sign;
    double discount;                // exp(-r dt)
    int32_t degree;
    int32_t n_steps;
};

// Element (path i, date t) of a path matrix: time-major [t][i] (coalesced; what the device-side consumers
// read) or path-major [i][t] (the reference's (n_paths, n_steps + 1) C order, so a host caller needs no
// transpose: a thread then writes runs of consecutive dates, 32 B per Philox block).
template <bool PATH_MAJOR>
__device__ __forceinline__ size_t path_at(int64_t i, int32_t t, int64_t count, int32_t n_steps) {
    return PATH_MAJOR ? static_cast<size_t>(i) * static_cast<size_t>(n_steps + 1) + static_cast<size_t>(t)
                      : static_cast<size_t>(t) * static_cast<size_t>(count) + static_cast<size_t>(i);
}

template <bool PATH_MAJOR = false>
__global__ __launch_bounds__(kBlock) void lsm_paths_kernel(PathRange pr, LsmContract c, double* __restrict__ paths) {
    const double vol = c.vol * kZScale;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double cum = 0.0;
        paths[path_at<PATH_MAJOR>(i, 0, pr.count, pr.n_steps)] = c.s_first;
        const int32_t blocks = (pr.n_steps + 3) >> 2;
        for (int32_t b = 0; b < blocks; ++b) {
            float z[4];
            raw_normals4(g_lo, g_hi, static_cast<uint32_t>(b), 0u, pr.key0, pr.key1, z);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int32_t t = 4 * b + j + 1;
                if (t <= pr.n_steps) {
                    cum += __builtin_fma(vol, static_cast<double>(z[j]), c.drift);
                    paths[path_at<PATH_MAJOR>(i, t, pr.count, pr.n_steps)] = exp(c.log_s0 + cum);
                }
            }
        }
    }
}

// AmericanOption.early_exercise_boundary (exotic_options.py:309-345): for every date t the 10th (put) /
// 90th (call) percentile of the in-the-money prices, np.percentile's default linear interpolation.
// One 1024-thread workgroup per date over the time-major path matrix.  The two order statistics the
// interpolation needs come from ONE 8-pass byte-wise radix select on the fp64 bit patterns (positive doubles order like their bits): no sort, no
// key buffer, the in-the-money filter applied on the fly.  Exact: the selected elements ARE the sorted
// array's entries, and the interpolation follows NumPy's _lerp term by term.
__device__ __forceinline__ bool boundary_itm(double s, double strike, double sign) { return sign * (s - strike) > 0.0; }

// The k_lo-th and k_hi-th smallest (0-based) bit patterns among the in-the-money entries of `row`, found
// together: one traversal of the row per byte feeds two 256-bin histograms (one per order statistic; they
// coincide until the two prefixes part, which for neighbours is usually the last byte or never).
constexpr int kBoundaryThreads = 1024;

__device__ __forceinline__ void boundary_select2(const double* __restrict__ row, int64_t n, double strike, double sign,
                                                 int64_t k_lo, int64_t k_hi, uint32_t (*hist)[256], uint64_t* shared,
                                                 uint64_t& out_lo, uint64_t& out_hi) {
    uint64_t prefix[2] = {0, 0}, mask = 0;
    int64_t k[2] = {k_lo, k_hi};
    for (int byte = 7; byte >= 0; --byte) {
        for (int b = threadIdx.x; b < 512; b += kBoundaryThreads) hist[b >> 8][b & 255] = 0u;
        __syncthreads();
        const bool same = prefix[0] == prefix[1];
        for (int64_t i = threadIdx.x; i < n; i += kBoundaryThreads) {
            const double s = row[i];
            if (!boundary_itm(s, strike, sign)) continue;
            const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(s));
            const uint32_t digit = static_cast<uint32_t>(bits >> (8 * byte)) & 0xFFu;
            if ((bits & mask) == prefix[0]) atomicAdd(&hist[0][digit], 1u);
            if (!same && (bits & mask) == prefix[1]) atomicAdd(&hist[1][digit], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const uint32_t* h = hist[same ? 0 : threadIdx.x];
            int64_t left = k[threadIdx.x];
            uint32_t bucket = 0;
            for (; bucket < 255u; ++bucket) {
                if (left < static_cast<int64_t>(h[bucket])) break;
                left -= h[bucket];
            }
            shared[2 * threadIdx.x] = bucket;
            shared[2 * threadIdx.x + 1] = static_cast<uint64_t>(left);
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            prefix[w] |= shared[2 * w] << (8 * byte);
            k[w] = static_cast<int64_t>(shared[2 * w + 1]);
        }
        mask |= 0xFFull << (8 * byte);
        __syncthreads();
    }
    out_lo = prefix[0];
    out_hi = prefix[1];
}

__global__ __launch_bounds__(kBoundaryThreads) void exercise_boundary_kernel(const double* __restrict__ paths, int64_t n_paths,
                                                                             double strike, double sign, double quantile,
                                                                             double* __restrict__ boundary) {
    __shared__ uint32_t hist[2][256];
    __shared__ uint64_t shared[4];
    __shared__ unsigned long long count;
    const double* row = paths + static_cast<size_t>(blockIdx.x) * static_cast<size_t>(n_paths);
    if (threadIdx.x == 0) count = 0ull;
    __syncthreads();
    unsigned long long mine = 0;
    for (int64_t i = threadIdx.x; i < n_paths; i += kBoundaryThreads) mine += boundary_itm(row[i], strike, sign) ? 1ull : 0ull;
    if (mine) atomicAdd(&count, mine);
    __syncthreads();
    const int64_t cnt = static_cast<int64_t>(count);
    if (cnt == 0) {                                            // nobody in the money: NaN (:343)
        if (threadIdx.x == 0) boundary[blockIdx.x] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    const double pos = static_cast<double>(cnt - 1) * quantile;       // NumPy: virtual index (n - 1) q
    const int64_t lo = static_cast<int64_t>(floor(pos));
    const int64_t hi = lo + 1 < cnt ? lo + 1 : cnt - 1;
    const double t = pos - static_cast<double>(lo);
    uint64_t bits_a, bits_b;
    boundary_select2(row, n_paths, strike, sign, lo, hi, hist, shared, bits_a, bits_b);
    if (threadIdx.x == 0) {
        const double a = __longlong_as_double(static_cast<long long>(bits_a)), b = __longlong_as_double(static_cast<long long>(bits_b));
        const double diff = b - a;                             // numpy.lib._function_base_impl._lerp
        double v = a + diff * t;
        if (t >= 0.5) v = b - diff * (1.0 - t);
        boundary[blockIdx.x] = v;
    }
}

struct LsmScale {                   // regressor z = (S / K - centre) * inv_width of the date being FITTED and of the date whose fit is APPLIED
    double fit_centre, fit_inv_width, prev_centre, prev_inv_width;
};

struct LsmCoeffs {
    double beta[kLsmMaxDegree + 1];
    int32_t valid;                  // regression at the date being finished was fitted
    int32_t pad;
};

__device__ __forceinline__ double lsm_intrinsic(const LsmContract& c, double s) { return fmax(c.sign * (s - c.strike), 0.0); }

// Runs in the wave that holds the grid totals of one exercise date (all 64 lanes active): lane m < 2d+1 has sum z^m,
// lane 9+k has sum z^k cf, lane 14 the in-the-money count.  The wave solves the (d+1)x(d+1) normal equations (Gaussian
// elimination in the natural order, fp64) and leaves the coefficients in `coef` (LDS of the workgroup that applies them): no host
// round trip per exercise date.
//
// The augmented 5 x 6 matrix lives ONE ELEMENT PER LANE (lane = 8 row + column) instead of thirty doubles in lane 0.  Round 3's
// form indexed its private array with the run-time pivot row (`a[piv][col]`), which put the whole matrix into a 256-byte private
// segment (the only scratch in the library) and ran ~1,200 dependent fp64 instructions in one lane.  Here an elimination step is one
// broadcast from a compile-time lane (v_readlane: the pivot), two gathers (ds_bpermute) and ONE multiply-subtract per lane;
// back-substitution runs on broadcast (wave-uniform) values.  There is no pivot SEARCH any more: since the regression runs in the
// standardised regressor (LsmScale) the moment matrix is symmetric positive definite with a condition number of 1e2 .. 1e6, and for
// such a matrix elimination in the natural order is backward stable (it is the Cholesky order; partial pivoting bought nothing and
// cost fifteen broadcasts, their compare chains and a row exchange per solve -- a third of the ~450 instructions one wave executes
// serially at the head of every per-date launch).
__device__ __forceinline__ double lane_bcast(double v, int src_lane /* compile-time constant */) {
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(dbl_lo(v)), src_lane));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(dbl_hi(v)), src_lane));
    return dbl_of(lo, hi);
}

struct LsmFit {
    LsmCoeffs* coef;
    int32_t degree;
    __device__ __forceinline__ void operator()(double total) const {
        constexpr int D = kLsmMaxDegree;               // rows 0..D, columns 0..D + the right-hand side in column D + 1
        const int lane = threadIdx.x & (kWave - 1);
        const int row = lane >> 3, l = lane & 7;
        const bool live = row <= D && l <= D + 1;
        const int n = degree + 1;
        // a[row][l] = sum x^(row + l)  (l <= D),  a[row][D + 1] = sum x^row cf
        double e = __shfl(total, live ? (l <= D ? row + l : 2 * D + 1 + row) : 0, kWave);
        bool ok = lane_bcast(total, kLsmNV - 2) > static_cast<double>(degree + 1);       // np.sum(itm) > poly_degree + 1 (:279)
        // unknowns beyond `degree` are pinned to 0 by turning their rows / columns into the identity
        if (row >= n || (l <= D && l >= n)) e = row == l ? 1.0 : 0.0;
        if (!live) e = 0.0;
#pragma unroll
        for (int col = 0; col <= D; ++col) {
            // no pivot search: the moment matrix of the standardised regressor is symmetric positive definite with a modest condition
            // number, for which elimination in the natural order is stable (it is the Cholesky order)
            // ... as long as the matrix IS definite.  Few distinct in-the-money prices (near-duplicate paths, a count of exactly
            // degree + 2, a width clamped at 1e-6 of the mean) make it numerically singular: the pivot -- the part of sum z^(2 col) that
            // the lower monomials do not explain -- is then rounding noise, and dividing by it would hand the next date arbitrary
            // coefficients with `valid` = 1.  A pivot below 1e-11 of its own diagonal entry pins that unknown to zero
            // instead (its row and column become the identity; the remaining monomials are fitted), a rule the checker restates.  The
            // reference's lstsq returns the minimum-norm solution there: not the same numbers -- per-seed parity of the American
            // option is statistical (olmc.h), and these cases are where it cannot be anything else.
            double p = lane_bcast(e, col * 8 + col);
            const bool pin = !(fabs(p) > 1e-11 * fabs(lane_bcast(total, 2 * col)));       // wave-uniform
            if (pin) {
                if (row == col) e = l == col ? 1.0 : 0.0;
                else if (l == col) e = 0.0;
                p = 1.0;
            }
            const double inv = 1.0 / p;
            const double f = __shfl(e, row * 8 + col, kWave) * inv;
            const double pr = __shfl(e, col * 8 + l, kWave);
            if (live && row > col && l >= col) e -= f * pr;
        }
        double beta[D + 1];
#pragma unroll
        for (int k = D; k >= 0; --k) {
            double v = lane_bcast(e, k * 8 + D + 1);
#pragma unroll
            for (int j = 0; j <= D; ++j)
                if (j > k) v -= lane_bcast(e, k * 8 + j) * beta[j];
            beta[k] = v / lane_bcast(e, k * 8 + k);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k <= D; ++k) coef->beta[k] = ok ? beta[k] : 0.0;
            coef->valid = ok ? 1 : 0;
        }
    }
};

// One exercise date for this thread's paths.  t_fit: the date whose moments are accumulated (>= 1);
// the date finished first is t_fit + 1 (skipped when init: the terminal payoff needs no regression).
// init != 0: cash flow starts as the terminal intrinsic value.  `fit()` = the regression of date t_fit + 1.
//
// A date is 24 bytes read and 8 written per path and ~40 flops: the loop is bounded by how many loads the few resident waves
// keep in flight, so a thread issues the loads of U of its paths (3 U doubles) before it touches any of them (round 4; round 3
// walked one path at a time: three dependent load round trips per path, eight paths per thread at 1M paths).  The paths of a
// thread are still consumed in ascending order, so the sixteen per-thread sums -- and the bits of the price -- do not depend on U.
// `fit()` is called by every thread, ONCE, after the loads of its first U paths are on their way: the coefficients arrive while
// those loads are in flight.
template <int U, bool INIT, bool FINAL, typename Fit>
__device__ __forceinline__ void lsm_date(int64_t n, const LsmContract& c, const LsmScale& sc, Fit fit, int32_t t_fit,
                                         const double* __restrict__ paths, double* __restrict__ cash, double (&acc)[kLsmNV]) {
#pragma unroll
    for (int k = 0; k < kLsmNV; ++k) acc[k] = 0.0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    const double* __restrict__ row1 = paths + static_cast<size_t>(INIT ? c.n_steps : t_fit + 1) * n;     // the later date
    const double* __restrict__ row0 = paths + static_cast<size_t>(t_fit) * n;                            // the date being fitted
    LsmCoeffs prev;
    bool fitted = false;
    int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    do {                                                // at least one trip: every thread of the workgroup meets the barriers inside fit()
        double cfv[U], s1v[U], s0v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            const bool in = i < n;
            s1v[u] = in ? row1[i] : 0.0;
            cfv[u] = (in && !INIT) ? cash[i] : 0.0;
            s0v[u] = (in && !FINAL) ? row0[i] : 0.0;
        }
        if (!fitted) { prev = fit(); fitted = true; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n) {
                double cf;
                if constexpr (INIT) {
                    cf = lsm_intrinsic(c, s1v[u]);
                } else {
                    cf = cfv[u];
                    const double iv = lsm_intrinsic(c, s1v[u]);
                    if (prev.valid && iv > 0.0) {
                        const double x = (s1v[u] * c.inv_strike - sc.prev_centre) * sc.prev_inv_width;
                        double cont = prev.beta[kLsmMaxDegree];
#pragma unroll
                        for (int k = kLsmMaxDegree - 1; k >= 0; --k) cont = cont * x + prev.beta[k];
                        if (iv > cont) cf = iv;
                    }
                }
                cf *= c.discount;
                cash[i] = cf;
                if constexpr (!FINAL) {
                    // every power up to 2 kLsmMaxDegree is summed whatever the degree: LsmFit replaces the rows and columns beyond
                    // `degree` by the identity, so the surplus sums are never read (and a sum that IS read is the same chain of adds)
                    const bool itm = lsm_intrinsic(c, s0v[u]) > 0.0;
                    const double x = (s0v[u] * c.inv_strike - sc.fit_centre) * sc.fit_inv_width;
                    double p = itm ? 1.0 : 0.0;                 // out-of-the-money paths add exact zeros (x is finite)
                    const double w = itm ? cf : 0.0;
#pragma unroll
                    for (int m = 0; m <= 2 * kLsmMaxDegree; ++m) {
                        acc[m] += p;
                        if (m <= kLsmMaxDegree) acc[2 * kLsmMaxDegree + 1 + m] += p * w;
                        p *= x;
                    }
                    acc[kLsmNV - 2] += itm ? 1.0 : 0.0;       // in-the-money count
                } else {
                    acc[0] += cf;                     // final date (t_fit == 0): moments of the time-0 cash flow
                    acc[1] += cf * cf;
                }
            }
        }
        i0 += stride * U;
    } while (i0 < n);
}

// One launch per exercise date; stream order is the only synchronisation between dates.
// INIT: the cash flow starts as the terminal intrinsic value (first launch); FINAL: t_fit == 0, the launch that leaves the
// moments of the time-0 cash flow (through the grid reduction, for the host) instead of regression sums.
//
// Round 4, second form.  The first form finished a date inside its own launch: row store (sc1) -> drain -> ticket -> the LAST
// workgroup loads the rows, solves, stores five coefficients, which the next launch loads.  That is three dependent memory round
// trips behind the last wave of every date and one in front of the next.  Here a launch only STORES its workgroup rows (plain
// stores: the kernel boundary publishes them) and the NEXT launch -- every workgroup of it, redundantly -- sums those <= 256 rows in
// the same index order (workgroup_rows_sum: one round trip, served by L2, overlapped with the loads of the workgroup's first paths),
// solves the 5 x 5 system in its first wave and hands the coefficients to its threads through LDS.  No ticket, no drain, no
// coefficient buffer, nothing but the row store behind a date's last wave.  The rows alternate between two buffers by date
// parity: a workgroup of date t may store its row while a slower one still reads date t + 1's.  Same sums in the same order
// as the first form, so the same coefficients, exercise decisions and price bits.
template <int U, bool INIT, bool FINAL>
__global__ __launch_bounds__(kBlock) void lsm_step_kernel(int64_t n, LsmContract c, LsmScale sc, double* __restrict__ rows /* [2][gridDim.x][kLsmNV] */,
                                                          int32_t t_fit, const double* __restrict__ paths, double* __restrict__ cash, ReduceWs ws) {
    __shared__ double part[kBlock];
    __shared__ LsmCoeffs shared_fit;
    const size_t buffer = static_cast<size_t>(gridDim.x) * kLsmNV;
    const double* rows_prev = rows + static_cast<size_t>((t_fit + 1) & 1) * buffer;
    auto fit = [&]() -> LsmCoeffs {
        LsmCoeffs f;
        f.valid = 0;
        if constexpr (!INIT) {
            const double total = workgroup_rows_sum<kLsmNV>(rows_prev, static_cast<int32_t>(gridDim.x), part);
            if (threadIdx.x < kWave) LsmFit{&shared_fit, c.degree}(total);
            __syncthreads();
            f = shared_fit;
        }
        return f;
    };
    double acc[kLsmNV];
    lsm_date<U, INIT, FINAL>(n, c, sc, fit, t_fit, paths, cash, acc);
    if constexpr (!FINAL) {
        const double s = block_row_sum<kLsmNV>(acc);
        if (threadIdx.x < kLsmNV) rows[static_cast<size_t>(t_fit & 1) * buffer + static_cast<size_t>(blockIdx.x) * kLsmNV + threadIdx.x] = s;
    } else {
        block_then_grid_reduce<kLsmNV>(acc, ws);
    }
}

// Heston full-truncation Euler (src/pricing_models/heston.py:184-255): per step two normals
// (Z1, and Z2 = rho Z1 + sqrt(1 - rho^2) Z2'), state (ln S, v) in fp64 registers:
//   v+ = max(v, 0);  ln S += (r - q - v+/2) dt + sqrt(v+ dt) Z1;
//   v  += kappa (theta - v+) dt + sigma_v sqrt(v+ dt) Z2;  v = max(v, 0).
// One Philox block feeds two steps: (z0, z1) -> step 2b, (z2, z3) -> step 2b+1, stream tag 1.
constexpr uint32_t kTagHeston = 1u;

struct HestonContract {
    double log_s0, v0;
    double mu_dt;          // (r - q) dt
    double dt, sqrt_dt;
    double kappa_dt, theta, sigma_v;
    double rho, rho_c;     // rho, sqrt(1 - rho^2)
    double strike, sign;
};

// sqrt of a non-negative finite fp64 from an fp32 v_rsq_f32 seed (2^-22) and one coupled Newton/Goldschmidt
// round plus a residual correction: 1 ulp (checked against sqrt over [1e-12, 10]), 10 instructions against
// the 17 + v_rsq_f64 of the library expansion (which also handles inf / NaN / subnormals: not needed for a
// truncated variance).  x = 0 gives 0; x below 1e-30 loses accuracy (sqrt < 1e-15: immaterial here).
__device__ __forceinline__ double sqrt_nonneg(double x) {
    const double r = static_cast<double>(__builtin_amdgcn_rsqf(fmaxf(static_cast<float>(x), 1e-30f)));
    double g = x * r, h = 0.5 * r;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}

// The recursion with its constants folded (v >= 0 holds after every step, so v+ = v):
//   ln S' = ln S - (dt/2) v + sqrt(v) u            u = sqrt(dt) Z1                 (+ (r - q) dt, added once per path as M (r-q) dt)
//   v'    = max(v (1 - kappa dt) + kappa theta dt + sqrt(v) w, 0)   w = sigma_v sqrt(dt) (rho Z1 + rho_c Z2')
// 2 + 3 fp64 ops per leg per step after the square root, against 12 for the literal form.
struct HestonStep {
    double neg_half_dt, one_minus_kdt, kdt_theta;
    double zs, a, b;                 // RAW normal -> u = zs z1;  w = a z1 + b z2
    __device__ __forceinline__ explicit HestonStep(const HestonContract& c)
        : neg_half_dt(-0.5 * c.dt), one_minus_kdt(1.0 - c.kappa_dt), kdt_theta(c.kappa_dt * c.theta),
          zs(kZScale * c.sqrt_dt), a(c.sigma_v * c.rho * (kZScale * c.sqrt_dt)), b(c.sigma_v * c.rho_c * (kZScale * c.sqrt_dt)) {}
    // SIGN = +1 / -1: the antithetic leg flips both normals (free source modifiers)
    template <int SIGN>
    __device__ __forceinline__ void advance(double u, double w, double& ls, double& v) const {
        const double sv = sqrt_nonneg(v);
        ls = __builtin_fma(sv, SIGN > 0 ? u : -u, __builtin_fma(neg_half_dt, v, ls));
        v = fmax(__builtin_fma(sv, SIGN > 0 ? w : -w, __builtin_fma(v, one_minus_kdt, kdt_theta)), 0.0);
    }
};

// v0 < 0 never comes through HestonPricer (heston.py:71-72 rejects it); at the C ABI the reference's recursion
// is kept: its first step sees v+ = 0, so it is deterministic -- ln S += (r - q) dt, v = max(v0 + kappa theta dt, 0)
// -- and is taken before the loop, which then skips date 0.
__device__ __forceinline__ bool heston_start(const HestonContract& c, double& v) {
    if (c.v0 >= 0.0) { v = c.v0; return false; }
    v = fmax(c.v0 + c.kappa_dt * c.theta, 0.0);
    return true;
}

template <bool ANTI>
__global__ __launch_bounds__(kBlock) void heston_kernel(PathRange pr, HestonContract c, ReduceWs ws) {
    const RoundKeys rk = pin_round_keys(pr.key0, pr.key1);
    double acc[2] = {0.0, 0.0};
    const HestonStep hs(c);
    double v_start;
    const bool skip0 = heston_start(c, v_start);
    const double ls_start = c.log_s0 + pr.n_steps * c.mu_dt;        // the drift (r - q) dt of every step, once
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double ls[2] = {ls_start, ls_start}, v[2] = {v_start, v_start};
        const int32_t blocks = (pr.n_steps + 1) >> 1;
        for (int32_t b = 0; b < blocks; ++b) {
            float z[4];
            raw_normals4_pinned(g_lo, g_hi, static_cast<uint32_t>(b), kTagHeston, rk, z);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int32_t t = 2 * b + h;
                if (t < pr.n_steps && !(skip0 && t == 0)) {
                    const double z1 = static_cast<double>(z[2 * h]);
                    const double u = hs.zs * z1;
                    const double w = __builtin_fma(hs.b, static_cast<double>(z[2 * h + 1]), hs.a * z1);
                    hs.advance<1>(u, w, ls[0], v[0]);
                    if constexpr (ANTI) hs.advance<-1>(u, w, ls[1], v[1]);
                }
            }
        }
#pragma unroll
        for (int leg = 0; leg < (ANTI ? 2 : 1); ++leg) {
            const double x = fmax(c.sign * (exp(ls[leg]) - c.strike), 0.0);
            acc[0] += x; acc[1] += x * x;
        }
    }
    block_then_grid_reduce<2>(acc, ws);
}

// HestonPricer.simulate_paths (heston.py:257-305): the same recursion on the same stream, every state
// written out (layouts: path_at): date 0 is (S, v0) as given (:286-287), not exp(log S).
template <bool PATH_MAJOR>
__global__ __launch_bounds__(kBlock) void heston_paths_kernel(PathRange pr, HestonContract c, double s_first,
                                                              double* __restrict__ spot, double* __restrict__ var) {
    const HestonStep hs(c);
    double v_start;
    const bool skip0 = heston_start(c, v_start);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double ls = c.log_s0, v = v_start;       // ls WITHOUT the (r - q) dt terms: added per date below
        spot[path_at<PATH_MAJOR>(i, 0, pr.count, pr.n_steps)] = s_first;
        var[path_at<PATH_MAJOR>(i, 0, pr.count, pr.n_steps)] = c.v0;
        const int32_t blocks = (pr.n_steps + 1) >> 1;
        for (int32_t b = 0; b < blocks; ++b) {
            float z[4];
            raw_normals4(g_lo, g_hi, static_cast<uint32_t>(b), kTagHeston, pr.key0, pr.key1, z);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int32_t t = 2 * b + h;
                if (t < pr.n_steps) {
                    if (!(skip0 && t == 0)) {
                        const double z1 = static_cast<double>(z[2 * h]);
                        hs.advance<1>(hs.zs * z1, __builtin_fma(hs.b, static_cast<double>(z[2 * h + 1]), hs.a * z1), ls, v);
                    }
                    const size_t at = path_at<PATH_MAJOR>(i, t + 1, pr.count, pr.n_steps);
                    spot[at] = exp(__builtin_fma(static_cast<double>(t + 1), c.mu_dt, ls));
                    var[at] = v;
                }
            }
        }
    }
}

// Jump diffusion (src/pricing_models/jump_diffusion.py:160-225 Merton, :325-372 Kou): per step one
// diffusion normal, a Poisson(lambda dt) number of jumps, and the jump sum added to ln S.
// Every step needs a diffusion normal and a Poisson uniform; the jump sizes are needed only where a jump
// occurs (lambda dt ~ 1e-3).  So ONE Philox block (stream tag 2) feeds TWO steps -- block b: (x0, x1) ->
// Box-Muller pair = the diffusion normals of steps 2b, 2b+1; x2, x3 -> their Poisson uniforms (inversion) --
// and the sizes come from a second stream drawn only by the lanes that jump (rare, divergent, cheap):
// block (path, step, tag 3 + j/2).  Merton: n jumps ~ N(n mu_j, n sigma_j^2), drawn exactly as
// n mu_j + sigma_j sqrt(n) z (the sum of n iid normals), z = the cosine normal of words (x0, x1) of block
// (path, step, tag 3).  Kou: each jump is +Exp(eta1) with probability p, else -Exp(eta2); jump j takes its
// two uniforms from words 2(j%2), 2(j%2)+1 of block (path, step, tag 3 + j/2).
constexpr uint32_t kTagJump = 2u, kTagJumpSize = 3u;

struct JumpContract {
    double log_s0, drift, vol;      // drift = (r - q - lambda kappa - sigma^2/2) dt, vol = sigma sqrt(dt)
    double strike, sign;
    double p0, lam_dt;              // exp(-lambda dt), lambda dt
    double mu_j, sigma_j;           // Merton
    double kou_p, inv_eta1, inv_eta2;
    int32_t kou;                    // 0 Merton, 1 Kou
    int32_t pad;
};

__device__ __forceinline__ double unit_open64(uint32_t x) { return (static_cast<double>(x) + 0.5) * 2.3283064365386963e-10; }

// The jump part of step t of one path, given its Poisson uniform: the count by inversion, then the sizes.
__device__ __forceinline__ void jump_sizes(const PathRange& pr, const JumpContract& c, uint32_t g_lo, uint32_t g_hi, int32_t t,
                                           double u, double& ls) {
    if (u < c.p0) return;                              // no jump (almost always)
    int32_t n = 1;
    double pk = c.p0 * c.lam_dt, cdf = c.p0 + pk;
    while (u >= cdf && n < 64) {
        ++n;
        pk *= c.lam_dt / n;
        cdf += pk;
    }
    if (!c.kou) {
        const Words4 k = philox4x32_10(g_lo, g_hi, static_cast<uint32_t>(t), kTagJumpSize, pr.key0, pr.key1);
        float z_jump, unused;
        box_muller_raw(k.x0, k.x1, z_jump, unused);
        ls += n * c.mu_j + c.sigma_j * sqrt(static_cast<double>(n)) * (kZScale * static_cast<double>(z_jump));
    } else {
        for (int32_t j = 0; j < n; ++j) {
            const Words4 k = philox4x32_10(g_lo, g_hi, static_cast<uint32_t>(t), kTagJumpSize + static_cast<uint32_t>(j >> 1),
                                           pr.key0, pr.key1);
            const double ud = unit_open64((j & 1) ? k.x2 : k.x0), um = unit_open64((j & 1) ? k.x3 : k.x1);
            ls += ud < c.kou_p ? -log(um) * c.inv_eta1 : log(um) * c.inv_eta2;
        }
    }
}

// Steps 2b and 2b+1 of one path (the second only if it exists); `after(t, ls)` sees ln S after each step.
template <typename After>
__device__ __forceinline__ void jump_block(const PathRange& pr, const JumpContract& c, double vol, uint32_t g_lo, uint32_t g_hi,
                                           int32_t b, double& ls, After after) {
    const Words4 w = philox4x32_10(g_lo, g_hi, static_cast<uint32_t>(b), kTagJump, pr.key0, pr.key1);
    float z0, z1;
    box_muller_raw(w.x0, w.x1, z0, z1);
    ls += __builtin_fma(vol, static_cast<double>(z0), c.drift);
    jump_sizes(pr, c, g_lo, g_hi, 2 * b, unit_open64(w.x2), ls);
    after(2 * b, ls);
    if (2 * b + 1 < pr.n_steps) {
        ls += __builtin_fma(vol, static_cast<double>(z1), c.drift);
        jump_sizes(pr, c, g_lo, g_hi, 2 * b + 1, unit_open64(w.x3), ls);
        after(2 * b + 1, ls);
    }
}

__global__ __launch_bounds__(kBlock) void jump_kernel(PathRange pr, JumpContract c, ReduceWs ws) {
    double acc[2] = {0.0, 0.0};
    const double vol = c.vol * kZScale;
    const int32_t blocks = (pr.n_steps + 1) >> 1;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double ls = c.log_s0;
        for (int32_t b = 0; b < blocks; ++b) jump_block(pr, c, vol, g_lo, g_hi, b, ls, [](int32_t, double) {});
        const double x = fmax(c.sign * (exp(ls) - c.strike), 0.0);
        acc[0] += x; acc[1] += x * x;
    }
    block_then_grid_reduce<2>(acc, ws);
}

// MertonJumpDiffusion.simulate_path (jump_diffusion.py:227-272), for any number of paths: the pricing
// kernel's recursion with every price written out (layouts: path_at), date 0 = S as given (:254).
template <bool PATH_MAJOR>
__global__ __launch_bounds__(kBlock) void jump_paths_kernel(PathRange pr, JumpContract c, double s_first, double* __restrict__ out) {
    const double vol = c.vol * kZScale;
    const int32_t blocks = (pr.n_steps + 1) >> 1;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double ls = c.log_s0;
        out[path_at<PATH_MAJOR>(i, 0, pr.count, pr.n_steps)] = s_first;
        for (int32_t b = 0; b < blocks; ++b)
            jump_block(pr, c, vol, g_lo, g_hi, b, ls,
                       [&](int32_t t, double l) { out[path_at<PATH_MAJOR>(i, t + 1, pr.count, pr.n_steps)] = exp(l); });
    }
}

// ------------------------------------------------------------------ QMC ----
// Scrambled-Sobol terminal prices (src/simulation/gbm_qmc.py:14-46): point k of the sequence is
//   x_t(k) = shift[t] ^ XOR_{b in gray(k)} sv[t][b],   u = x * 2^-30,
// with sv / shift the host-built (SciPy LMS + digital shift) direction matrix of dimension t,
// so the uniforms equal scipy.stats.qmc.Sobol(d, scramble=True, seed).random(n) bit for bit.
// Then clip to [1e-10, 1-1e-10] (:36), z = Phi^-1(u) in fp64 (ndtri_w), sum over the dims, exp.  No
// antithetic mirror in the pricer (gbm_qmc.py:14-46); QmcRange.mirror serves simulate_gbm_qmc_antithetic (:49-76).
constexpr int kSobolBits = 30;      // SciPy's default `bits`

// Inverse normal CDF for p in [1e-10, 1 - 1e-10], 1 ulp-class (against mpmath on Sobol-shaped probabilities: within 1.5 x 2^-52
// of the value, 1.40e-15 absolute; SciPy's ndtri, the reference's: 1.56e-15 -- tests/test_gpu_instrumented.py):
//   z = x f(w),   x = 2p - 1,   w = -ln(4 p (1 - p)) = -ln(1 - x^2),   f = sqrt(2) erfinv(x) / x,
// with f a degree-24 polynomial in (w - 3.125) for w < 6.25 (|x| < 0.99903: 99.9 % of the points, so a wave almost
// never runs the other branch) and a degree-22 polynomial in (sqrt(w) - 3.6) beyond (the form of Giles, "Approximating
// the erfinv function"; the coefficients are our own Chebyshev fits in 50-digit arithmetic, tools/fit_ndtri.py).  The
// coefficients sit in constant memory so that they reach the fma as SGPR pairs (as literals each needs a v_mov_b64 and
// the kernel 187 VGPRs).  One 24-instruction log + 24 fma on the main path (round 4: 33 + 24), against ~280 instructions of the
// two-region rational AS241 (whose tail, 15 % of the points, made nearly every wave execute both branches).
constexpr double kNdtriSplit = 6.25, kNdtriCentreA = 3.125, kNdtriCentreB = 3.6;
__constant__ double kNdtriA[25] = {
    2.338620710026593, 0.3396349587011389, -0.008532899177267343,
    -0.001047511569485617, 0.00026408204954061674, -1.9632852863696442e-05,
    -1.931065027529881e-06, 5.988894861277318e-07, -4.111174160550445e-08,
    -5.8161801676714855e-09, 1.4866543569083624e-09, -7.65698113492015e-11,
    -1.8354802577559388e-11, 3.7201267209535326e-12, -1.1415427137390895e-13,
    -5.659687073061811e-14, 9.222558296724625e-15, -5.57226748086415e-17,
    -1.7275519790527177e-16, 2.1935565352484576e-17, 8.591419094975e-19,
    -4.912241660340459e-19, 2.8273805909578654e-20, 4.51702010660485e-21,
    -5.081556263217504e-22};
__constant__ double kNdtriB[23] = {
    4.859596171648984, 1.4254796811228716, 0.003842875854461125,
    -0.0023842328222623333, 0.0011466522188153746, -0.0006567166880171974,
    0.0004164386944938703, -0.00023425354061713856, 9.740652820425401e-05,
    -2.0256567899466118e-05, -6.957554201847671e-06, 8.5258821514889e-06,
    -3.747113816251416e-06, 5.696998164072895e-07, 3.3129619045867057e-07,
    -2.596439487036509e-07, 7.981211000255892e-08, -3.4150983553453232e-09,
    -9.995513600060063e-09, 6.228140947976023e-09, -1.2224374850709372e-09,
    -5.137257330967595e-10, 2.2302639189274297e-10};

// w = -ln(4 t) for t = p (1 - p) in [1e-10, 1/4] (neg_log_quad).  t = m 2^e with m in [0.7071066, 1.4142132) -- split on the HIGH
// WORD as the C libraries do (add 0x3ff00000 - 0x3fe6a09e, shift for e, mask and add back for m: four 32-bit integer operations
// where frexp + compare + selects took six, two of them fp64) -- ln m = 2 atanh(s) with s = (m - 1)/(m + 1) as 2s + s^3 Q(s^2), Q of
// degree 6 (5.6e-17 absolute on ln m; tools/fit_ndtri.py), the division by a v_rcp_f64 seed (2^-23) with ONE Newton round (2^-46)
// and a residual correction of the quotient (which squares what is left: below 2^-53), (e + 2) ln 2 in two pieces: 24 instructions
// where the library's correctly rounded log spends ~75.  Round 5 (new bits, the same accuracy: tests/test_gpu_instrumented.py pins
// all forms to mpmath): t by one fma (ndtri_pt), the factor 4 in the exponent, one Newton round instead of two, the integer split.
__constant__ double kLogQ[7] = {0.666666666666667, 0.39999999999886615, 0.28571428631764334, 0.2222221019926421, 0.18182956608063458, 0.15329500754204178, 0.14643628601909797};

// A zero the optimiser cannot see through (see ndtri_lockstep_add): a table indexed [k + opaque_zero()] is loaded where it is used.
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

// A Sobol integer x < 2^30 as the reference's clipped uniform: clip(x 2^-30, 1e-10, 1 - 1e-10) (gbm_qmc.py:36).  The upper clip
// can never bind -- the largest uniform is 1 - 2^-30 = 1 - 9.3e-10 -- so it is not computed; the lower one binds for x = 0 alone.
__device__ __forceinline__ double sobol_uniform(uint32_t x) {
    return fmax(static_cast<double>(x) * 9.313225746154785e-10, 1e-10);      // x 2^-30
}

// p (1 - p) = p - p^2 with ONE rounding (and no cancellation at either end: the fma sees the exact difference).
__device__ __forceinline__ double ndtri_pt(double p) { return __builtin_fma(-p, p, p); }

template <class Q /* q[0 .. 6]: kLogQ, or the caller's registers */>
__device__ __forceinline__ double neg_log_quad(double t, const Q& lq) {
    const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(t));
    // the exponent counted from 1/4 (so that 4 t costs nothing), the mantissa's high word re-based to [sqrt(1/2), sqrt(2))
    const int32_t k = static_cast<int32_t>(static_cast<uint32_t>(bits >> 32) + (0x3ff00000u - 0x3fe6a09eu) - (1021u << 20));
    const int32_t e = k >> 20;                                                                  // arithmetic: floor
    const uint32_t mh = (static_cast<uint32_t>(k) & 0x000fffffu) + 0x3fe6a09eu;
    const double m = __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(mh) << 32) | static_cast<uint32_t>(bits)));
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    double s = num * r;
    s = __builtin_fma(__builtin_fma(-den, s, num), r, s);
    const double u = s * s;
    double q = lq[6];
#pragma unroll
    for (int j = 5; j >= 0; --j) q = __builtin_fma(q, u, lq[j]);
    const double ln_m = __builtin_fma(s * u, q, s + s);
    const double ed = static_cast<double>(e);
    return -__builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, ln_m));
}

// The rare side of the inverse normal (w >= 6.25, 0.1 % of the points): degree 22 in sqrt(w) - 3.6.
__device__ __forceinline__ double ndtri_tail(double w, int z = 0 /* see ndtri_lockstep_add */) {
    const double t = sqrt(w) - kNdtriCentreB;
    double f = kNdtriB[22 + z];
#pragma unroll
    for (int k = 21; k >= 0; --k) f = __builtin_fma(f, t, kNdtriB[k + z]);
    return f;
}

// Every form below ADDS a point's inverse normal to a running sum:  acc + Phi^-1(p) = fma(x, f(w), acc),  x = 2p - 1 (one fma, exact),
// the sqrt(2) inside f's coefficients -- one instruction where z = sqrt(2) * x * f, acc += z were three.  The Sobol kernels only ever
// sum a point's normals; a lone Phi^-1(p) is the same call with acc = 0.
//
// The one-point-per-thread kernels (european_qmc_kernel, european_qmc_batch_kernel<., false>) call ndtri_w_add once per dimension:
// the compiler keeps the 55 coefficients in scalar registers across the dimension loop (what does not fit next to the dimension's
// 30 direction numbers comes back through a few v_readlane_b32).
__device__ __forceinline__ double ndtri_w_add(double acc, double p, int z_tail = 0) {
    const double x = __builtin_fma(p, 2.0, -1.0);
    const double w = neg_log_quad(ndtri_pt(p), kLogQ);
    double f;
    if (w < kNdtriSplit) {
        const double t = w - kNdtriCentreA;
        f = kNdtriA[24];
#pragma unroll
        for (int k = 23; k >= 0; --k) f = __builtin_fma(f, t, kNdtriA[k]);
    } else {
        // 0.1 % of the points: 94 % of the waves have no lane here.  The empty volatile asm keeps this side a branch the wave can
        // skip (s_cbranch_execz) -- without it hipcc flattened both sides into selects in european_qmc_batch_kernel (58 instead of
        // 34 fp64 fma per dimension and the sqrt expansion for every point: 229 vs 160 us at 2^17 x 252)
        asm volatile("");
        f = ndtri_tail(w, z_tail);
    }
    return __builtin_fma(x, f, acc);
}

// The main branch's coefficients held in VECTOR registers by the caller (the aligned one-point Sobol kernels, whose 24 fewer lane
// masks leave the room): every fma then reads three VGPRs -- no scalar-operand limit, no coefficient spilled to lanes and fetched
// back by v_readlane_b32.  Same operations in the same order as ndtri_w_add: the same bits.
struct NdtriRegs {
    double a[25], q[7];
    // Which coefficients are pinned to vector registers: the two LEADING ones -- an fma takes one scalar operand, so a scalar leading
    // coefficient next to a scalar addend costs a v_mov_b64 per point.  The rest is left to the compiler (scalar registers, rebuilt
    // by s_mov where they do not fit).  Pinning the main polynomial's first 4 / 8 / 12 / 25 as well (63 / 71 / 79 / 95 VGPRs against
    // 55) measured even to 1.5 % slower (profiles/r05_ab_kernels.txt): since the fold left the scalar unit, neither occupancy nor
    // the operand kind is what limits the kernel.
    static constexpr int kPinned = 0;
    __device__ __forceinline__ void load() {
#pragma unroll
        for (int k = 0; k < 25; ++k) { a[k] = kNdtriA[k]; if (k < kPinned || k == 24) asm volatile("" : "+v"(a[k])); }
#pragma unroll
        for (int k = 0; k < 7; ++k) { q[k] = kLogQ[k]; if (k == 6) asm volatile("" : "+v"(q[k])); }
    }
};

__device__ __forceinline__ double ndtri_w_regs_add(double acc, double p, const NdtriRegs& c, int z_tail) {
    const double x = __builtin_fma(p, 2.0, -1.0);
    const double w = neg_log_quad(ndtri_pt(p), c.q);
    double f;
    if (w < kNdtriSplit) {
        const double t = w - kNdtriCentreA;
        f = c.a[24];
#pragma unroll
        for (int k = 23; k >= 0; --k) f = __builtin_fma(f, t, c.a[k]);
    } else {
        asm volatile("");
        f = ndtri_tail(w, z_tail);
    }
    return __builtin_fma(x, f, acc);
}

// TWO inverse normals in lockstep from the same register-held coefficients (round 5: two consecutive dimensions of one point in the
// aligned one-point Sobol kernels), added to acc in order: (acc + z0) + z1.  One point per thread left a wave a single chain of ~60
// dependent fp64 operations per dimension.  Both logarithms and both main polynomials run as one basic block here; a tail point is
// repaired afterwards behind a branch the wave almost always skips (as in ndtri_lockstep_add below).  Every point sees exactly the
// operations of ndtri_w_add: the same bits.
__device__ __forceinline__ double ndtri_w_regs_pair_add(double acc, const double (&p)[2], const NdtriRegs& c, int z_tail) {
    double w[2], f[2], t[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w[j] = neg_log_quad(ndtri_pt(p[j]), c.q);
        t[j] = w[j] - kNdtriCentreA;
        f[j] = c.a[24];
    }
#pragma unroll
    for (int k = 23; k >= 0; --k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) f[j] = __builtin_fma(f[j], t[j], c.a[k]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!(w[j] < kNdtriSplit)) {
            asm volatile("");                           // a real branch (see ndtri_w_add)
            f[j] = ndtri_tail(w[j], z_tail);
        }
        acc = __builtin_fma(__builtin_fma(p[j], 2.0, -1.0), f[j], acc);
    }
    return acc;
}

// fma(a, b, c) with the addend taken straight from a scalar register pair (VOP3).  Left to itself hipcc copies a freshly s_load-ed
// coefficient into a VGPR pair (two v_mov_b32) so that it can use the two-address v_fmac_f64.
__device__ __forceinline__ double fma_scalar_addend(double a, double b, double c_uniform) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}

// EIGHT inverse normals per thread, in lockstep (round 4; the eight-points-per-thread kernels), each added to its own running sum.
// Round 3 called the one-point form eight times per dimension: eight basic blocks (each point has its own main / tail branch), so
// nothing of one point overlapped anything of another -- a wave ran eight serial chains of ~45 dependent fp64 operations per
// dimension -- and the 55 coefficients, hoisted in front of the dimension loop, overflowed the scalar registers next to the 28
// direction numbers of the dimension: 511 v_readlane_b32 per trip of european_qmc_block_kernel's loop brought them back one by one
// (the 16-contract batch kernel's spill lanes and accumulation registers took its VGPR count to 270).  Here
//   * the eight logarithms run as one basic block (eight independent chains for the scheduler to interleave);
//   * the main polynomial (99.9 % of the points) is evaluated for ALL eight points, coefficient by coefficient: one scalar load
//     serves eight fma, the chains cover each other's latency;
//   * a point in the tail (w >= 6.25) is repaired afterwards behind a branch the wave almost always skips;
//   * the tables are indexed [k + z] with z an OPAQUE ZERO the caller redefines once per dimension (opaque_zero()): the address is
//     then not invariant in the dimension loop, so the coefficients are s_load-ed where they are used (440 bytes per trip from
//     the scalar data cache) and occupy scalar registers only then.
// Every point sees exactly the operations of ndtri_w_add: the same bits.
template <int NP>
__device__ __forceinline__ void ndtri_lockstep_add(double (&acc)[NP], const double (&p)[NP], int z) {
    double w[NP], f[NP], t[NP];
    // the main polynomial's leading coefficient and first group are asked for BEFORE the logarithms (which do not need them)
    constexpr int kGroup = 4;
    double cur[kGroup], nxt[kGroup];
    const double lead = kNdtriA[24 + z];
#pragma unroll
    for (int i = 0; i < kGroup; ++i) cur[i] = kNdtriA[23 - i + z];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        w[j] = neg_log_quad(ndtri_pt(p[j]), kLogQ);
        t[j] = w[j] - kNdtriCentreA;
    }
    // the main polynomial, its coefficients fetched a GROUP AHEAD of the fma that use them: scalar loads return out of order, so the
    // wait in front of a group's first use is for everything outstanding -- issued right before its use (what the compiler does
    // left alone: nine load-wait pairs per trip) a wave idles a scalar-cache latency each time; issued a group earlier, behind
    // kGroup x NP fma of work, the wait finds the data there.  The scheduling barriers keep the order written here.
#pragma unroll
    for (int j = 0; j < NP; ++j) f[j] = lead;
#pragma unroll
    for (int g = 0; g < 24 / kGroup; ++g) {
        // touch the group in hand FIRST: its wait (for everything outstanding) must come before the next group's loads are issued
#pragma unroll
        for (int i = 0; i < kGroup; ++i) asm volatile("" : "+s"(cur[i]));
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 24 / kGroup) {
#pragma unroll
            for (int i = 0; i < kGroup; ++i) nxt[i] = kNdtriA[23 - kGroup * (g + 1) - i + z];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
#pragma unroll
            for (int j = 0; j < NP; ++j) f[j] = fma_scalar_addend(f[j], t[j], cur[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < kGroup; ++i) cur[i] = nxt[i];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (!(w[j] < kNdtriSplit)) {
            asm volatile("");                           // a real branch (see ndtri_w_add)
            f[j] = ndtri_tail(w[j], z);
        }
        acc[j] = __builtin_fma(__builtin_fma(p[j], 2.0, -1.0), f[j], acc[j]);
    }
}

struct QmcRange {
    uint64_t first;    // index of the first Sobol point of this launch
    int64_t count;
    int32_t dims;      // effective_steps = min(n_steps, 21201)
    int32_t mirror;    // kTerminal only: also write terminal[count + i] = exp(a - vol sum z)  (simulate_gbm_qmc_antithetic, gbm_qmc.py:49-76)
};

// The sum of a point's inverse normals has ONE association in every Sobol kernel (round 4):  ((Q0 + Q1) + Q2) + Q3,  Q_w = the
// dimensions [w D / 4, (w + 1) D / 4) added in order -- so that a point summed by one thread, by a thread that carries eight points,
// and a point whose dimensions are SPLIT over the four waves of a workgroup (below) all give the same bits.
__device__ __forceinline__ int32_t qmc_quarter_begin(int32_t dims, int w) {
    return static_cast<int32_t>(static_cast<int64_t>(dims) * w / 4);
}

// Sum of the inverse normals of one Sobol point over the dimensions [t0, t1).
//
// UNIFORM_HI (round 5): the 64 lanes of the wave hold the 64 CONSECUTIVE points of a 64-aligned block, so bits 6 .. 29 of their
// Gray codes are the same in every lane (gray bit b = k_b ^ k_{b+1}).  The XOR of the direction numbers those bits select is then one
// number per wave and dimension (`gray_hi` comes from v_readfirstlane; the digital shift goes into the same word) and only the six
// low bits are left per point: 6 v_bitop3_b32 per point and dimension instead of 30 (each with an SGPR operand: 4 issue cycles),
// and 6 lane masks in registers instead of 30.
template <bool UNIFORM_HI = false>
__device__ __forceinline__ double qmc_point_sum(const uint32_t (&mask)[kSobolBits], int32_t t0, int32_t t1, const uint32_t* __restrict__ sv,
                                                const uint32_t* __restrict__ shift, uint32_t gray_hi = 0u /* wave-uniform; UNIFORM_HI only */,
                                                const NdtriRegs* regs = nullptr /* UNIFORM_HI only */) {
    double q = 0.0;
    if constexpr (UNIFORM_HI) {
        // Lane-per-dimension fold (round 5, second form): for 64 dimensions at a time, lane l XORs rows 6 .. 29 of dimension c0 + l
        // under the wave's Gray bits -- and the digital shift -- into ONE word: 24 vector and-xors per 64 dimensions where the
        // scalar unit spent 24 and / xor per dimension behind a scalar load it had to wait for.  The dimension loop fetches only
        // the six low rows, one trip AHEAD (s_load, no wait in front of its use).
        const int lane = static_cast<int>(threadIdx.x) & (kWave - 1);
        for (int32_t c0 = t0; c0 < t1; c0 += kWave) {
            const int32_t cn = __builtin_amdgcn_readfirstlane(t1 - c0 < kWave ? t1 - c0 : kWave);
            const int32_t tl = c0 + lane < t1 ? c0 + lane : t1 - 1;
            const uint32_t* __restrict__ mine = sv + static_cast<size_t>(tl) * kSobolBits;
            uint32_t fold = shift[tl];
#pragma unroll
            for (int b = 6; b < kSobolBits; ++b) fold ^= mine[b] & (0u - ((gray_hi >> b) & 1u));
            // two dimensions per trip (their inverse normals in lockstep: ndtri_w_regs_pair_add), added to q in dimension order.  A
            // dimension's folded word reaches every lane through ds_bpermute_b32 (a broadcast of lane j, issued one trip ahead: no
            // vector-unit instruction, where v_readlane_b32 + the v_mov_b32 its scalar result forced on the first row cost two).
            int hop = 0;                                                            // byte address of lane j for ds_bpermute_b32
            asm volatile("" : "+v"(hop));
            uint32_t lo[2][6];
            int xf[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const uint32_t* __restrict__ row = sv + static_cast<size_t>(c0 + d < t1 ? c0 + d : c0) * kSobolBits;
#pragma unroll
                for (int b = 0; b < 6; ++b) lo[d][b] = row[b];
                xf[d] = __builtin_amdgcn_ds_bpermute(hop + 4 * d, static_cast<int>(fold));
            }
            int32_t j = 0;
            for (; j + 2 <= cn; j += 2) {
                double u[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    uint32_t x = static_cast<uint32_t>(xf[d]);
#pragma unroll
                    for (int b = 0; b < 6; ++b) x = __builtin_amdgcn_bitop3_b32(x, lo[d][b], mask[b], 0x78);
                    u[d] = sobol_uniform(x);
                }
                // the next trip's words and low rows, fetched now (lane indices wrap, dimensions are clamped: unused past the end)
                hop += 8;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    xf[d] = __builtin_amdgcn_ds_bpermute(hop + 4 * d, static_cast<int>(fold));
                    const int32_t tn = c0 + j + 2 + d < t1 ? c0 + j + 2 + d : t1 - 1;
                    const uint32_t* __restrict__ next = sv + static_cast<size_t>(tn) * kSobolBits;
#pragma unroll
                    for (int b = 0; b < 6; ++b) lo[d][b] = next[b];
                }
                q = ndtri_w_regs_pair_add(q, u, *regs, opaque_zero());
            }
            if (j < cn) {                                                           // an odd dimension left (xf[0], lo[0] are its)
                uint32_t x = static_cast<uint32_t>(xf[0]);
#pragma unroll
                for (int b = 0; b < 6; ++b) x = __builtin_amdgcn_bitop3_b32(x, lo[0][b], mask[b], 0x78);
                q = ndtri_w_regs_add(q, sobol_uniform(x), *regs, opaque_zero());
            }
        }
        return q;
    }
    for (int32_t t = t0; t < t1; ++t) {
        const uint32_t* __restrict__ row = sv + static_cast<size_t>(t) * kSobolBits;
        uint32_t x = shift[t];
#pragma unroll
        for (int b = 0; b < kSobolBits; ++b) x = __builtin_amdgcn_bitop3_b32(x, row[b], mask[b], 0x78);   // x ^ (row & mask), one v_bitop3_b32
        const double u = sobol_uniform(x);
        q = ndtri_w_add(q, u, opaque_zero());
    }
    return q;
}

// Contract::a = ln S + drift * dims, Contract::vol = sigma sqrt(T / dims) (gbm_qmc.py:38-44).
//
// SPLIT (round 4: launches of at most 2^18 points; round 5: every launch below 2^20 points): a workgroup owns 64 points and each of
// its four waves walks a QUARTER of the dimensions -- the split workgroups of the pseudo-random kernel.  A 2^17-point launch is
// 2,048 waves of one-point threads, two per SIMD, each a serial chain of ~45 dependent fp64 operations per dimension: the vector
// unit idles 30 % of the time (`frac_valu_active_pmc` 0.70, profiles/r04_bench_detail.json).  Split, the same launch is 8,192 waves
// of a quarter of the length.  The quarter sums meet in LDS, wave 0 adds them in the canonical order and prices the point.
// ALIGNED (round 5; SPLIT launches whose point offset is a multiple of 64, from 64 dimensions on): qmc_point_sum<true> -- the wave's
// lanes are an aligned block of 64 points, so the direction numbers of Gray bits 6 .. 29 fold into one word per wave and dimension
// (first on the scalar unit, twelve 64-bit and / xor pairs per dimension behind a scalar load: 119 -> 97 us at 2^17 x 252; then
// lane-per-dimension on the vector unit, 64 dimensions at a time, broadcast by ds_bpermute_b32: see qmc_point_sum), six
// v_bitop3_b32 are left per point and dimension, two dimensions run in lockstep, and with 24 lane masks fewer the inverse normal's
// coefficients sit in registers without v_readlane spill traffic: 110 -> 71 vector instructions per point and dimension with the
// same bits (83 us), 64 with round 5's shorter inverse normal (76 us; round 4: 120).  63 -> 55 VGPRs.  The one-point form
// (SPLIT = false) keeps its 30 lane masks: its aligned variant needed 191 VGPRs and lost.
template <int MODE, bool SPLIT = false, bool ALIGNED = false /* qr.first is a multiple of 64: a wave's lanes are an aligned block of points */>
__global__ __launch_bounds__(kBlock) void european_qmc_kernel(QmcRange qr, Contract c, const uint32_t* __restrict__ sv,
                                                              const uint32_t* __restrict__ shift, ReduceWs ws,
                                                              double* __restrict__ terminal) {
    constexpr int NV = MODE == kControlVariate ? 5 : 2;
    double acc[NV] = {};
    auto price_point = [&](int64_t i, double zsum) {
        const double st = exp(c.a + c.vol * zsum);
        if constexpr (MODE == kTerminal) {
            terminal[i] = st;
            if (qr.mirror) terminal[qr.count + i] = exp(c.a - c.vol * zsum);
        } else {
            add_sample<MODE>(acc, fmax(c.sign * (st - c.strike), 0.0), st);
        }
    };
    if constexpr (SPLIT) {
        __shared__ double quarter_sum[kWavesPerBlock][kWave];
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave), lane = threadIdx.x & (kWave - 1);
        const int64_t i = static_cast<int64_t>(blockIdx.x) * kWave + lane;             // the grid covers every point (host guarantee)
        // a dead lane of the last workgroup keeps ITS index in the aligned form (its sum is never used; the wave's high bits stay uniform)
        const uint64_t k = qr.first + static_cast<uint64_t>((ALIGNED || i < qr.count) ? i : 0);
        const uint32_t gray = static_cast<uint32_t>(k ^ (k >> 1));
        uint32_t mask[kSobolBits];
#pragma unroll
        for (int b = 0; b < kSobolBits; ++b) mask[b] = 0u - ((gray >> b) & 1u);
        const uint32_t gray_hi = ALIGNED ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gray))) : 0u;
        NdtriRegs regs;
        if constexpr (ALIGNED) regs.load();
        quarter_sum[wave][lane] = qmc_point_sum<ALIGNED>(mask, qmc_quarter_begin(qr.dims, wave), qmc_quarter_begin(qr.dims, wave + 1), sv, shift, gray_hi, &regs);
        __syncthreads();
        if (wave == 0 && i < qr.count)
            price_point(i, ((quarter_sum[0][lane] + quarter_sum[1][lane]) + quarter_sum[2][lane]) + quarter_sum[3][lane]);
    } else {
        const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < qr.count; i += stride) {
            const uint64_t k = qr.first + static_cast<uint64_t>(i);
            const uint32_t gray = static_cast<uint32_t>(k ^ (k >> 1));
            uint32_t mask[kSobolBits];
#pragma unroll
            for (int b = 0; b < kSobolBits; ++b) mask[b] = 0u - ((gray >> b) & 1u);
            const uint32_t gray_hi = ALIGNED ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gray))) : 0u;
            NdtriRegs regs;
            if constexpr (ALIGNED) regs.load();
            double zsum = 0.0;
#pragma unroll 1
            for (int w = 0; w < 4; ++w) zsum += qmc_point_sum<ALIGNED>(mask, qmc_quarter_begin(qr.dims, w), qmc_quarter_begin(qr.dims, w + 1), sv, shift, gray_hi, &regs);
            price_point(i, zsum);
        }
    }
    if constexpr (MODE != kTerminal) block_then_grid_reduce<NV>(acc, ws);
}

// Large launches: a thread takes the EIGHT consecutive points k0 .. k0+7 of an aligned block (k0 = 8j).  In Gray-code order
// consecutive points differ in one direction number, and inside an aligned block of eight those are sv[t][0], [1], [0], [2],
// [0], [1], [0] -- the same for every lane, i.e. scalar operands: 28 + 7 v_bitop3 / v_xor per dimension for eight points
// instead of 8 x 30, and eight independent inverse normals in flight per thread.  Same points, same uniforms, same z as
// the one-point kernel; only the order in which a workgroup's payoffs are added differs (1e-16).
constexpr int kQmcBlock = 8;

// Sums of the inverse normals of the eight points of an aligned block over all dimensions, in the canonical association (quarters,
// see qmc_quarter_begin), left in LDS: zs[p][threadIdx.x].  mask[b], b >= B0 = 2, are the Gray-code masks of the block's first point
// (its bits 0 and 1 are clear).  The running totals live in LDS between the quarters (each thread touches only its own column: no
// barrier), so the dimension loop carries the eight quarter sums in registers and nothing else -- the epilogues walk the eight
// points by a run-time index anyway, which registers do not offer.
// UNIFORM_HI (round 5): the wave's 64 lanes carry 64 CONSECUTIVE blocks starting at a multiple of 512 points, so bits 9 .. 29 of
// their first points' Gray codes are wave-uniform: those direction numbers fold into one word per wave and dimension (first on the
// scalar unit per dimension, then lane-per-dimension for 64 dimensions at a time), bits 2 .. 8 stay per-lane work -- 7 + 7
// v_bitop3_b32 / v_xor per dimension for eight points instead of 28 + 7, and 21 lane masks fewer in registers.
template <int B0, bool UNIFORM_HI = false>
__device__ __forceinline__ void qmc_block_sums(const uint32_t (&mask)[kSobolBits], int32_t dims, const uint32_t* __restrict__ sv,
                                               const uint32_t* __restrict__ shift, double (*zs)[kBlock], uint32_t gray_hi = 0u) {
    static_assert(!UNIFORM_HI || B0 == 2, "the fold is laid out for blocks of eight");
#pragma unroll
    for (int p = 0; p < kQmcBlock; ++p) zs[p][threadIdx.x] = 0.0;          // 0 + Q0 = Q0 exactly
    const int lane = static_cast<int>(threadIdx.x) & (kWave - 1);
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        double q[kQmcBlock];
#pragma unroll
        for (int p = 0; p < kQmcBlock; ++p) q[p] = 0.0;
        const int32_t t0 = qmc_quarter_begin(dims, w), t1 = qmc_quarter_begin(dims, w + 1);
        if constexpr (UNIFORM_HI) {
            // as qmc_point_sum<true>: 64 dimensions at a time, lane l folds rows 9 .. 29 of dimension c0 + l under the wave's Gray
            // bits (and the digital shift) into one word, ds_bpermute_b32 hands a dimension's word to every lane one trip ahead, and
            // the nine low rows (2 .. 8 under lane masks, 0 .. 2 for the steps inside a block) come by s_load one trip ahead
            for (int32_t c0 = t0; c0 < t1; c0 += kWave) {
                const int32_t cn = __builtin_amdgcn_readfirstlane(t1 - c0 < kWave ? t1 - c0 : kWave);
                const int32_t tl = c0 + lane < t1 ? c0 + lane : t1 - 1;
                const uint32_t* __restrict__ mine = sv + static_cast<size_t>(tl) * kSobolBits;
                uint32_t fold = shift[tl];
#pragma unroll
                for (int b = 9; b < kSobolBits; ++b) fold ^= mine[b] & (0u - ((gray_hi >> b) & 1u));
                int hop = 0;
                asm volatile("" : "+v"(hop));
                uint32_t lo[9];
                {
                    const uint32_t* __restrict__ row = sv + static_cast<size_t>(c0) * kSobolBits;
#pragma unroll
                    for (int b = 0; b < 9; ++b) lo[b] = row[b];
                }
                int xf = __builtin_amdgcn_ds_bpermute(hop, static_cast<int>(fold));
                for (int32_t j = 0; j < cn; ++j) {
                    const int z0 = opaque_zero();
                    uint32_t x = static_cast<uint32_t>(xf);
#pragma unroll
                    for (int b = B0; b < 9; ++b) x = __builtin_amdgcn_bitop3_b32(x, lo[b], mask[b], 0x78);
                    double u[kQmcBlock];
#pragma unroll
                    for (int p = 0; p < kQmcBlock; ++p) {
                        if (p) x ^= lo[__builtin_ctz(static_cast<unsigned>(p))];          // gray(k + 1) = gray(k) ^ (1 << ctz(k + 1))
                        u[p] = sobol_uniform(x);
                    }
                    hop += 4;
                    xf = __builtin_amdgcn_ds_bpermute(hop, static_cast<int>(fold));
                    {
                        const int32_t tn = c0 + j + 1 < t1 ? c0 + j + 1 : t1 - 1;
                        const uint32_t* __restrict__ next = sv + static_cast<size_t>(tn) * kSobolBits;
#pragma unroll
                        for (int b = 0; b < 9; ++b) lo[b] = next[b];
                    }
                    ndtri_lockstep_add<kQmcBlock>(q, u, z0);
                }
            }
        } else {
            for (int32_t t = t0; t < t1; ++t) {
                const uint32_t* __restrict__ row = sv + static_cast<size_t>(t) * kSobolBits;
                uint32_t x = shift[t];
                const int z0 = opaque_zero();
#pragma unroll
                for (int b = B0; b < kSobolBits; ++b) x = __builtin_amdgcn_bitop3_b32(x, row[b], mask[b], 0x78);
                double u[kQmcBlock];
#pragma unroll
                for (int p = 0; p < kQmcBlock; ++p) {
                    if (p) x ^= row[__builtin_ctz(static_cast<unsigned>(p))];          // gray(k + 1) = gray(k) ^ (1 << ctz(k + 1))
                    u[p] = sobol_uniform(x);
                }
                ndtri_lockstep_add<kQmcBlock>(q, u, z0);
            }
        }
#pragma unroll
        for (int p = 0; p < kQmcBlock; ++p) zs[p][threadIdx.x] += q[p];
    }
}

template <int MODE, bool ALIGNED = false /* qr.first is a multiple of 512: a wave's lanes are 64 consecutive blocks of an aligned run */>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ALIGNED ? 4 : 3, 8))) void european_qmc_block_kernel(QmcRange qr, Contract c, const uint32_t* __restrict__ sv,
                                                                    const uint32_t* __restrict__ shift, ReduceWs ws,
                                                                    double* __restrict__ terminal) {
    constexpr int NV = MODE == kControlVariate ? 5 : 2;
    double acc[NV] = {};
    const uint64_t base = qr.first / kQmcBlock;                                   // first block (may start before qr.first)
    const uint64_t last = qr.first + static_cast<uint64_t>(qr.count);             // one past the last point
    const int64_t n_blocks = static_cast<int64_t>((last + kQmcBlock - 1) / kQmcBlock - base);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    // ALIGNED: a wave stays whole while its FIRST lane has a block -- the lanes fold one another's dimensions (qmc_block_sums<2, true>:
    // lane l serves dimension c0 + l), so a lane past the end keeps going with its own index (the wave's high Gray bits stay uniform)
    // and its points, all >= last, are skipped below like any ragged end
    const int64_t lane = ALIGNED ? static_cast<int64_t>(threadIdx.x & (kWave - 1)) : 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i - lane < n_blocks; i += stride) {
        const uint64_t k0 = (base + static_cast<uint64_t>(i)) * kQmcBlock;
        const uint32_t gray = static_cast<uint32_t>(k0 ^ (k0 >> 1));              // bits 0 and 1 are zero for k0 = 8j
        uint32_t mask[kSobolBits];
#pragma unroll
        for (int b = 2; b < kSobolBits; ++b) mask[b] = 0u - ((gray >> b) & 1u);
        // the eight exponentials one after the other (a real loop over the LDS-staged sums, as in european_qmc_batch_kernel): unrolled,
        // their interleaving set the kernel's register count
        __shared__ double zs[kQmcBlock][kBlock];
        qmc_block_sums<2, ALIGNED>(mask, qr.dims, sv, shift, zs, ALIGNED ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gray))) : 0u);
#pragma unroll 1
        for (int p = 0; p < kQmcBlock; ++p) {
            const uint64_t k = k0 + static_cast<uint64_t>(p);
            if (k < qr.first || k >= last) continue;                            // the ragged ends of the range
            const double zp = zs[p][threadIdx.x];
            const double st = exp(c.a + c.vol * zp);
            if constexpr (MODE == kTerminal) {
                const int64_t at = static_cast<int64_t>(k - qr.first);
                terminal[at] = st;
                if (qr.mirror) terminal[qr.count + at] = exp(c.a - c.vol * zp);
            } else {
                add_sample<MODE>(acc, fmax(c.sign * (st - c.strike), 0.0), st);
            }
        }
    }
    if constexpr (MODE != kTerminal) block_then_grid_reduce<NV>(acc, ws);
}

// k contracts on the SAME Sobol points in one launch (the 8 / 14 bumped contracts of compute_greeks_unified on a
// MCMethod.QMC pricer, unified_greeks.py:295-358; round 2 priced them by 8 / 14 launches): the points, their uniforms and
// the inverse normals do not depend on the contract (every bump keeps dims = min(n_steps, 21201) and the seed), so sum z is
// formed once per point and each contract costs its exp (or, with the base's vol, one multiply) and a payoff -- the
// ContractSet machinery of the pseudo-random kernel.  Contract c prices exp(c.a + c.vol sum z) exactly as european_qmc_kernel
// does for it alone; a contract that shares its vol with a base takes scale * S_T(base) (2-3 ulp from its own exp).
// BLOCK8 = the eight-points-per-thread expansion of european_qmc_block_kernel.  The grid covers every point / block (host
// guarantee), so the 2 NSETS sums are born after the dimension loop.
template <int NSETS, bool BLOCK8, bool SPLIT = false, bool ALIGNED = false /* SPLIT: qr.first is a multiple of 64 (qmc_point_sum<true>); BLOCK8: of 512 (qmc_block_sums<2, true>) */>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((BLOCK8 && ALIGNED) ? 4 : 1, 8))) void european_qmc_batch_kernel(QmcRange qr, ContractSet<NSETS> cs, const uint32_t* __restrict__ sv,
                                                                    const uint32_t* __restrict__ shift, ReduceWs ws) {
    static_assert(!(BLOCK8 && SPLIT), "a thread either carries eight points or a quarter of one point's dimensions");
    static_assert(SPLIT || BLOCK8 || !ALIGNED, "the aligned forms exist for split workgroups and for blocks of eight");
    constexpr int NV = 2 * NSETS;
    double acc[NV];
    if constexpr (SPLIT) {
        // 64 points per workgroup, a quarter of the dimensions per wave (european_qmc_kernel<., true>): launches below 2^20 points
        __shared__ double quarter_sum[kWavesPerBlock][kWave];
        const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave), lane = threadIdx.x & (kWave - 1);
        const int64_t i = static_cast<int64_t>(blockIdx.x) * kWave + lane;
        const uint64_t k = qr.first + static_cast<uint64_t>((ALIGNED || i < qr.count) ? i : 0);
        const uint32_t gray = static_cast<uint32_t>(k ^ (k >> 1));
        uint32_t mask[kSobolBits];
#pragma unroll
        for (int b = 0; b < kSobolBits; ++b) mask[b] = 0u - ((gray >> b) & 1u);
        const uint32_t gray_hi = ALIGNED ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gray))) : 0u;
        {
            NdtriRegs regs;             // dead after the dimension loop: the contracts' epilogue gets the registers back
            if constexpr (ALIGNED) regs.load();
            quarter_sum[wave][lane] = qmc_point_sum<ALIGNED>(mask, qmc_quarter_begin(qr.dims, wave), qmc_quarter_begin(qr.dims, wave + 1), sv, shift, gray_hi, &regs);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] = 0.0;
        if (wave == 0) {
            const double zsum = ((quarter_sum[0][lane] + quarter_sum[1][lane]) + quarter_sum[2][lane]) + quarter_sum[3][lane];
            european_payoffs<NSETS, false, kReduce>(cs, zsum, i < qr.count, 0, 0, nullptr, acc);
        }
    } else if constexpr (BLOCK8) {
        const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
        const uint64_t base = qr.first / kQmcBlock;
        const uint64_t last = qr.first + static_cast<uint64_t>(qr.count);
        const int64_t n_units = static_cast<int64_t>((last + kQmcBlock - 1) / kQmcBlock - base);
        const bool unit_live = i < n_units;
        const uint64_t k0 = (base + static_cast<uint64_t>((ALIGNED || unit_live) ? i : 0)) * kQmcBlock;     // aligned: a dead lane keeps its index (uniform high bits)
        const uint32_t gray = static_cast<uint32_t>(k0 ^ (k0 >> 1));       // an aligned block of eight starts with gray bits 0 and 1 clear
        uint32_t mask[kSobolBits];
#pragma unroll
        for (int b = 2; b < kSobolBits; ++b) mask[b] = 0u - ((gray >> b) & 1u);
        __shared__ double zs[kQmcBlock][kBlock];
        qmc_block_sums<2, ALIGNED>(mask, qr.dims, sv, shift, zs, ALIGNED ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gray))) : 0u);
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] = 0.0;
        // The eight points of the thread are priced ONE AFTER THE OTHER by a real loop (round 4).  Unrolled, the compiler interleaved
        // eight times NSETS payoffs with up to five library exponentials each: 12,600 instructions and 270 VGPRs for 14 contracts
        // (one wave per SIMD in a kernel whose dimension loop needs 127).  A loop needs the normal sums addressable by a run-time
        // index, which registers are not, so they pass through LDS: 16 KB per workgroup, each thread reads back only what it wrote
        // itself (no barrier).  The payoffs of a thread's points are still added in ascending order: same bits.
        uint32_t live_points = 0u;                                          // bit p: point k0 + p lies in the range (its ragged ends)
#pragma unroll
        for (int p = 0; p < kQmcBlock; ++p) {
            const uint64_t k = k0 + static_cast<uint64_t>(p);
            live_points |= (unit_live && k >= qr.first && k < last) ? (1u << p) : 0u;
        }
#pragma unroll 1
        for (int p = 0; p < kQmcBlock; ++p)
            european_payoffs<NSETS, false, kReduce>(cs, zs[p][threadIdx.x], ((live_points >> p) & 1u) != 0u, 0, 0, nullptr, acc);
    } else {
        const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
        const bool live = i < qr.count;
        const uint64_t k = qr.first + static_cast<uint64_t>(live ? i : 0);
        const uint32_t gray = static_cast<uint32_t>(k ^ (k >> 1));
        uint32_t mask[kSobolBits];
#pragma unroll
        for (int b = 0; b < kSobolBits; ++b) mask[b] = 0u - ((gray >> b) & 1u);
        double zsum = 0.0;
#pragma unroll 1
        for (int w = 0; w < 4; ++w) zsum += qmc_point_sum(mask, qmc_quarter_begin(qr.dims, w), qmc_quarter_begin(qr.dims, w + 1), sv, shift);
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] = 0.0;
        european_payoffs<NSETS, false, kReduce>(cs, zsum, live, 0, 0, nullptr, acc);
    }
    block_then_grid_reduce<NV>(acc, ws);
}

// Hands n doubles that earlier work on the stream left in device memory (a shard's all-reduced triple) to the host the way the
// path kernels hand over their own results: written to the pinned buffer, completion word raised behind them.  One wave.
__global__ void publish_kernel(const double* __restrict__ src, int32_t n, double* __restrict__ host_dst, uint64_t* __restrict__ done_flag,
                               uint64_t done_value) {
    if (static_cast<int32_t>(threadIdx.x) < n) host_dst[threadIdx.x] = src[threadIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (threadIdx.x == 0) __hip_atomic_store(done_flag, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------- validation taps ----
__global__ void philox_words_kernel(uint64_t first, int64_t n_paths, int32_t block0, int32_t n_blocks,
                                    uint32_t tag, uint32_t k0, uint32_t k1, uint32_t* __restrict__ out) {
    const int64_t total = n_paths * n_blocks;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const uint64_t g = first + static_cast<uint64_t>(i / n_blocks);
        const uint32_t b = static_cast<uint32_t>(block0 + static_cast<int32_t>(i % n_blocks));
        const Words4 w = philox4x32_10(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), b, tag, k0, k1);
        out[4 * i + 0] = w.x0; out[4 * i + 1] = w.x1; out[4 * i + 2] = w.x2; out[4 * i + 3] = w.x3;
    }
}

__global__ void normals_kernel(uint64_t first, int64_t n_paths, int32_t n_steps, uint32_t k0, uint32_t k1,
                               float* __restrict__ out) {
    const int32_t blocks = (n_steps + 3) >> 2;
    const int64_t total = n_paths * blocks;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t p = i / blocks;
        const int32_t b = static_cast<int32_t>(i % blocks);
        const uint64_t g = first + static_cast<uint64_t>(p);
        float z[4];
        raw_normals4(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), static_cast<uint32_t>(b), 0u, k0, k1, z);
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < n_steps) out[p * n_steps + 4 * b + j] = kZScaleF * z[j];
    }
}

}  // namespace olmc
