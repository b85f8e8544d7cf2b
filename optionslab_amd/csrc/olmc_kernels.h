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
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Philox4x32-10 (Salmon et al., SC'11).  The key (k0,k1) is wave-uniform, so the
// ten bumped round keys live in SGPRs; each round is two 32x32->64 multiplies
// (v_mad_u64_u32) and two three-input XORs.
__device__ __forceinline__ Words4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = static_cast<uint64_t>(kPhiloxM0) * c0;
        const uint64_t p1 = static_cast<uint64_t>(kPhiloxM1) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += kPhiloxW0;
        k1 += kPhiloxW1;
    }
    return Words4{c0, c1, c2, c3};
}

// ------------------------------------------------------------ Box-Muller ----
// u = (x + 0.5) * 2^-32 in fp32 (never 0, so the log is finite; may round to 1).
// v_log_f32 is log2, v_sin_f32 / v_cos_f32 take their argument in revolutions,
// so 2*pi*u needs no multiply.  |z| <= sqrt(2*33*ln 2) = 6.76.
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& z_cos, float& z_sin) {
    constexpr float kTwoM32 = 2.3283064365386963e-10f;   // 2^-32
    constexpr float kTwoM33 = 1.1641532182693481e-10f;   // 2^-33
    constexpr float kMinus2Ln2 = -1.3862943611198906f;   // -2 ln 2
    const float ua = __builtin_fmaf(static_cast<float>(xa), kTwoM32, kTwoM33);
    const float ub = __builtin_fmaf(static_cast<float>(xb), kTwoM32, kTwoM33);
    const float rad = __builtin_amdgcn_sqrtf(kMinus2Ln2 * __builtin_amdgcn_logf(ua));
    z_cos = rad * __builtin_amdgcn_cosf(ub);
    z_sin = rad * __builtin_amdgcn_sinf(ub);
}

// Four normals of steps 4*block .. 4*block+3 of global path `g`.
__device__ __forceinline__ void normals4(uint32_t g_lo, uint32_t g_hi, uint32_t block, uint32_t tag,
                                         uint32_t k0, uint32_t k1, float (&z)[4]) {
    const Words4 w = philox4x32_10(g_lo, g_hi, block, tag, k0, k1);
    box_muller(w.x0, w.x1, z[0], z[1]);
    box_muller(w.x2, w.x3, z[2], z[3]);
}

// sum_t Z_t over n_steps for one path: fp32 inside a block of four, fp64 across
// blocks.  A trailing partial block uses the first n_steps % 4 normals.
__device__ __forceinline__ double path_normal_sum(uint32_t g_lo, uint32_t g_hi, int32_t n_steps,
                                                  uint32_t k0, uint32_t k1) {
    const int32_t full = n_steps >> 2;
    double acc = 0.0;
    float z[4];
    for (int32_t b = 0; b < full; ++b) {
        normals4(g_lo, g_hi, static_cast<uint32_t>(b), 0u, k0, k1, z);
        acc += static_cast<double>((z[0] + z[1]) + (z[2] + z[3]));
    }
    const int32_t rem = n_steps & 3;
    if (rem) {
        normals4(g_lo, g_hi, static_cast<uint32_t>(full), 0u, k0, k1, z);
        float s = z[0];
        if (rem > 1) s += z[1];
        if (rem > 2) s += z[2];
        acc += static_cast<double>(s);
    }
    return acc;
}

// ------------------------------------------------------------ reductions ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Sum NV per-thread values over the block in a fixed order and let thread 0
// store them to dst[0..NV).  LDS-staged across the four waves.
template <int NV>
__device__ __forceinline__ void block_sum_store(const double (&v)[NV], double* __restrict__ dst) {
    __shared__ double stage[kWavesPerBlock][NV];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) stage[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = stage[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) s += stage[w][threadIdx.x];
        dst[threadIdx.x] = s;
    }
}

// ------------------------------------------------------------- contracts ----
// Host-precomputed per-contract constants, in the reference's own arithmetic
// order (gbm_numpy.py:35-39): a = ln S + (r - q - sigma^2/2) dt * M, vol = sigma sqrt(dt).
struct Contract {
    double a;        // log_S0 + total_drift
    double vol;      // sigma * sqrt(dt)
    double strike;
    double sign;     // +1 call, -1 put : payoff = max(sign * (S_T - K), 0)
};

template <int NSETS>
struct ContractSet {
    Contract c[NSETS];
};

struct PathRange {
    uint64_t first;    // global index of local path 0
    int64_t count;     // paths in this launch
    int32_t n_steps;
    uint32_t key0, key1;
};

enum Mode : int { kReduce = 0, kTerminal = 1, kControlVariate = 2 };

// European terminal payoff.  kReduce: partials[block][set][{sum,sumsq}].
// kTerminal (NSETS == 1): terminal[i] = S_T^+, terminal[count + i] = S_T^- (coalesced,
// the [pos | neg] layout of gbm_numpy.py:51).  kControlVariate (NSETS == 1):
// partials[block][{sum_x, sum_s, sum_xx, sum_ss, sum_xs}] with x the UNdiscounted payoff.
template <int NSETS, bool ANTI, int MODE>
__global__ __launch_bounds__(kBlock) void european_kernel(PathRange pr, ContractSet<NSETS> cs,
                                                          double* __restrict__ partials,
                                                          double* __restrict__ terminal) {
    constexpr int NV = (MODE == kControlVariate) ? 5 : 2 * NSETS;
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;

    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const double zsum = path_normal_sum(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32),
                                            pr.n_steps, pr.key0, pr.key1);
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            const Contract c = cs.c[s];
            const double dz = c.vol * zsum;
            const double up = exp(c.a + dz);
            if constexpr (MODE == kTerminal) {
                terminal[i] = up;
                if constexpr (ANTI) terminal[pr.count + i] = exp(c.a - dz);
            } else {
                const double xu = fmax(c.sign * (up - c.strike), 0.0);
                if constexpr (MODE == kControlVariate) {
                    acc[0] += xu; acc[1] += up; acc[2] += xu * xu; acc[3] += up * up; acc[4] += xu * up;
                } else {
                    acc[2 * s] += xu; acc[2 * s + 1] += xu * xu;
                }
                if constexpr (ANTI) {
                    const double dn = exp(c.a - dz);
                    const double xd = fmax(c.sign * (dn - c.strike), 0.0);
                    if constexpr (MODE == kControlVariate) {
                        acc[0] += xd; acc[1] += dn; acc[2] += xd * xd; acc[3] += dn * dn; acc[4] += xd * dn;
                    } else {
                        acc[2 * s] += xd; acc[2 * s + 1] += xd * xd;
                    }
                }
            }
        }
    }
    if constexpr (MODE != kTerminal) block_sum_store<NV>(acc, partials + static_cast<size_t>(blockIdx.x) * NV);
}

// Asian option: running arithmetic sum of S_t (or sum of ln S_t) over t = 1..M
// kept in registers (exotic_options.py:59-67, 119-122 without the path matrix).
struct AsianContract {
    double log_s0;
    double drift;     // (r - q - sigma^2/2) dt
    double vol;       // sigma sqrt(dt)
    double strike;
    double sign;
    double inv_steps; // 1 / M
};

template <bool ANTI, bool GEOMETRIC>
__global__ __launch_bounds__(kBlock) void asian_kernel(PathRange pr, AsianContract c,
                                                       double* __restrict__ partials) {
    double acc[2] = {0.0, 0.0};
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < pr.count; i += stride) {
        const uint64_t g = pr.first + static_cast<uint64_t>(i);
        const uint32_t g_lo = static_cast<uint32_t>(g), g_hi = static_cast<uint32_t>(g >> 32);
        double cum_u = 0.0, cum_d = 0.0;   // cumsum of log-returns
        double run_u = 0.0, run_d = 0.0;   // running sum of S_t or of ln S_t
        const int32_t blocks = (pr.n_steps + 3) >> 2;
        for (int32_t b = 0; b < blocks; ++b) {
            float z[4];
            normals4(g_lo, g_hi, static_cast<uint32_t>(b), 0u, pr.key0, pr.key1, z);
            const int32_t live = min(4, pr.n_steps - 4 * b);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < live) {
                    const double dz = c.vol * static_cast<double>(z[j]);
                    cum_u += c.drift + dz;
                    if constexpr (GEOMETRIC) run_u += c.log_s0 + cum_u;
                    else run_u += exp(c.log_s0 + cum_u);
                    if constexpr (ANTI) {
                        cum_d += c.drift - dz;
                        if constexpr (GEOMETRIC) run_d += c.log_s0 + cum_d;
                        else run_d += exp(c.log_s0 + cum_d);
                    }
                }
            }
        }
        double avg = run_u * c.inv_steps;
        if constexpr (GEOMETRIC) avg = exp(avg);
        const double xu = fmax(c.sign * (avg - c.strike), 0.0);
        acc[0] += xu; acc[1] += xu * xu;
        if constexpr (ANTI) {
            double avd = run_d * c.inv_steps;
            if constexpr (GEOMETRIC) avd = exp(avd);
            const double xd = fmax(c.sign * (avd - c.strike), 0.0);
            acc[0] += xd; acc[1] += xd * xd;
        }
    }
    block_sum_store<2>(acc, partials + static_cast<size_t>(blockIdx.x) * 2);
}

// Second stage: out[v] = sum over blocks of partials[block][v], fixed order
// (thread t takes blocks t, t+256, ...; then an LDS tree), so equal inputs give
// equal bits (tests/test_monte_carlo.py:153-158 requires price1 == price2).
__global__ __launch_bounds__(kBlock) void finalize_kernel(const double* __restrict__ partials, int32_t n_blocks,
                                                          int32_t nv, double* __restrict__ out) {
    __shared__ double tree[kBlock];
    for (int v = 0; v < nv; ++v) {
        double s = 0.0;
        for (int32_t b = threadIdx.x; b < n_blocks; b += kBlock) s += partials[static_cast<size_t>(b) * nv + v];
        tree[threadIdx.x] = s;
        __syncthreads();
        for (int half = kBlock / 2; half > 0; half >>= 1) {
            if (threadIdx.x < half) tree[threadIdx.x] += tree[threadIdx.x + half];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[v] = tree[0];
        __syncthreads();
    }
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
        normals4(static_cast<uint32_t>(g), static_cast<uint32_t>(g >> 32), static_cast<uint32_t>(b), 0u, k0, k1, z);
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < n_steps) out[p * n_steps + 4 * b + j] = z[j];
    }
}

}  // namespace olmc
