/*
 * philox_gbm.c -- CPU restatement (plain C, libm) of the DEVICE algorithm of
 * libolmc: Philox4x32-10 counters -> Box-Muller normals -> GBM payoffs.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as a checker; the product never links or calls it.
 *
 * Why a second oracle: the reference (oracle/numpy_reference.py, pinned bitwise
 * to the reference's own outputs) draws PCG64 + ziggurat normals, so the GPU can
 * match it only statistically (within 3 sigma of the Monte Carlo error).  This
 * file consumes the SAME counter-based stream as the GPU, so GPU results must
 * agree with it to rounding (the GPU's v_log/v_sin/v_cos approximations, ~1e-6
 * relative on a normal), which is a far tighter gate on indexing, antithetic
 * layout, remainder handling, sharding offsets and the reduction.
 *
 * Pinning: the Philox core is checked against the Random123 known-answer
 * vectors (tests/test_philox_oracle.py); the payoff arithmetic follows
 * src/simulation/gbm_numpy.py:35-51 and src/pricing_models/exotic_options.py:54-67,119-131
 * of the reference and is cross-checked against the NumPy oracle statistically.
 *
 * Stream contract (include/olmc.h): key = (seed lo, seed hi); counter =
 * (path lo, path hi, step/4, tag); words (x0,x1) -> normals of steps 4b, 4b+1
 * (cos, sin), (x2,x3) -> steps 4b+2, 4b+3; radius u = fmaf((float)x, 2^-32, 2^-33),
 * angle = (x & 0x7fffff) * 2^-23 turns.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define M0 0xD2511F53u
#define M1 0xCD9E8D57u
#define W0 0x9E3779B9u
#define W1 0xBB67AE85u

void ol_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static float unit_open(uint32_t x) { return fmaf((float)x, 0x1p-32f, 0x1p-33f); }

#define Z_SCALE 1.1774100225154747 /* sqrt(2 ln 2) */

/* RAW Box-Muller pair z' = sqrt(-log2 u_a) * {cos, sin}(2 pi u_b), evaluated in double on the
 * device's fp32 inputs and rounded once to fp32 (true normal = sqrt(2 ln 2) * z').
 * radius: u_a = fmaf(x_a, 2^-32, 2^-33); angle: turn fraction (x_b & 0x7fffff) * 2^-23. */
static void box_muller_raw(uint32_t xa, uint32_t xb, float* zc, float* zs) {
    const double ua = unit_open(xa), ub = (double)(xb & 0x007FFFFFu) * 0x1p-23;
    const double rad = sqrt(-log2(ua)), ang = 6.283185307179586476925 * ub;
    *zc = (float)(rad * cos(ang));
    *zs = (float)(rad * sin(ang));
}

static void raw_normals4(uint64_t path, uint32_t block, uint64_t seed, float z[4]) {
    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), block, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
    ol_philox4x32_10(ctr, key, w);
    box_muller_raw(w[0], w[1], &z[0], &z[1]);
    box_muller_raw(w[2], w[3], &z[2], &z[3]);
}

/* out[p*n_steps + t]: the normals as the olmc_normals tap reports them (fp32 scale of z'). */
void ol_normals(uint64_t seed, int64_t path0, int64_t n_paths, int32_t n_steps, float* out) {
    for (int64_t p = 0; p < n_paths; ++p)
        for (int32_t b = 0; 4 * b < n_steps; ++b) {
            float z[4];
            raw_normals4((uint64_t)(path0 + p), (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < n_steps; ++j) out[p * n_steps + 4 * b + j] = (float)Z_SCALE * z[j];
        }
}

/* sum_t Z in the device's order: fp32 within a Philox block and across a group of four blocks
 * (fma chain), fp64 across groups, a trailing partial block on its own; total * sqrt(2 ln 2). */
/* acc + a block's four RAW normals in the device's factored form:
 * fmaf(rad_b, cos_b + sin_b, fmaf(rad_a, cos_a + sin_a, acc)), every intermediate rounded to fp32. */
static float raw_block_accumulate(float acc, uint64_t path, uint32_t b, uint64_t seed) {
    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), b, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
    ol_philox4x32_10(ctr, key, w);
    for (int h = 0; h < 2; ++h) {
        const double ua = unit_open(w[2 * h]), ang = 6.283185307179586476925 * ((double)(w[2 * h + 1] & 0x007FFFFFu) * 0x1p-23);
        const float rad = (float)sqrt(-log2(ua)), c = (float)cos(ang), sn = (float)sin(ang);
        acc = fmaf(rad, c + sn, acc);
    }
    return acc;
}

static double path_normal_sum(uint64_t path, int32_t n_steps, uint64_t seed) {
    const int32_t full = n_steps >> 2, rem = n_steps & 3;
    double acc = 0.0;
    int32_t b = 0;
    for (; b + 4 <= full; b += 4) {
        float s = 0.0f;
        for (int j = 0; j < 4; ++j) s = raw_block_accumulate(s, path, (uint32_t)(b + j), seed);
        acc += (double)s;
    }
    if (b < full) {
        float s = 0.0f;
        for (; b < full; ++b) s = raw_block_accumulate(s, path, (uint32_t)b, seed);
        acc += (double)s;
    }
    if (rem) {
        float z[4];
        raw_normals4(path, (uint32_t)full, seed, z);
        float s = z[0];
        if (rem > 1) s += z[1];
        if (rem > 2) s += z[2];
        acc += (double)s;
    }
    return acc * Z_SCALE;
}

/* Terminal prices [pos | neg] for global paths path0 .. path0+n-1 (gbm_numpy.py:35-51). */
void ol_european_terminal(double S, double T, double r, double sigma, double q, int64_t path0, int64_t n,
                          int32_t n_steps, uint64_t seed, int antithetic, double* out) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt);
    const double a = log(S) + drift * n_steps;
    for (int64_t i = 0; i < n; ++i) {
        const double dz = vol * path_normal_sum((uint64_t)(path0 + i), n_steps, seed);
        out[i] = exp(a + dz);
        if (antithetic) out[n + i] = exp(a - dz);
    }
}

/* moments[0..4] = sum x, sum x^2, sum s, sum s^2, sum x*s over all payoff samples
 * (x = UNdiscounted payoff, s = terminal price); long double accumulation. */
void ol_european_moments(double S, double K, double T, double r, double sigma, double q, int is_call, int64_t path0,
                         int64_t n, int32_t n_steps, uint64_t seed, int antithetic, double moments[5]) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt);
    const double a = log(S) + drift * n_steps, sign = is_call ? 1.0 : -1.0;
    long double m[5] = {0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const double dz = vol * path_normal_sum((uint64_t)(path0 + i), n_steps, seed);
        for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
            const double s = exp(leg ? a - dz : a + dz), x = fmax(sign * (s - K), 0.0);
            m[0] += x; m[1] += x * x; m[2] += s; m[3] += s * s; m[4] += x * s;
        }
    }
    for (int j = 0; j < 5; ++j) moments[j] = (double)m[j];
}

/* Asian: average over t = 1..M of S_t (arithmetic) or exp(mean ln S_t) (geometric)
 * (exotic_options.py:54-67, 119-131).  moments[0..1] = sum x, sum x^2. */
void ol_asian_moments(double S, double K, double T, double r, double sigma, double q, int is_call, int geometric,
                      int64_t path0, int64_t n, int32_t n_steps, uint64_t seed, int antithetic, double moments[2]) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt);
    const double log_s0 = log(S), sign = is_call ? 1.0 : -1.0;
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        double cum[2] = {0, 0}, run[2] = {0, 0};
        for (int32_t b = 0; 4 * b < n_steps; ++b) {
            float z[4];
            raw_normals4((uint64_t)(path0 + i), (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < n_steps; ++j) {
                const double dz = (vol * Z_SCALE) * (double)z[j];
                cum[0] += drift + dz;
                cum[1] += drift - dz;
                for (int leg = 0; leg < 2; ++leg) run[leg] += geometric ? log_s0 + cum[leg] : exp(log_s0 + cum[leg]);
            }
        }
        for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
            double avg = run[leg] / n_steps;
            if (geometric) avg = exp(avg);
            const double x = fmax(sign * (avg - K), 0.0);
            m0 += x; m1 += x * x;
        }
    }
    moments[0] = (double)m0; moments[1] = (double)m1;
}

/* Barrier / lookback: running max / min of the cumulative log-return, t = 0 included
 * (exotic_options.py:174-224, 359-401).  payoff: 0 up-out, 1 up-in, 2 down-out, 3 down-in,
 * 4 lookback floating, 5 lookback fixed.  moments[0..1] = sum x, sum x^2. */
void ol_extrema_moments(double S, double K, double T, double r, double sigma, double q, int is_call, int payoff,
                        double barrier, int64_t path0, int64_t n, int32_t n_steps, uint64_t seed, int antithetic,
                        double moments[2]) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt) * Z_SCALE;
    const double lb = payoff <= 3 ? log(barrier / S) : 0.0, sign = is_call ? 1.0 : -1.0;
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        double cum[2] = {0, 0}, mx[2] = {0, 0}, mn[2] = {0, 0};
        for (int32_t b = 0; 4 * b < n_steps; ++b) {
            float z[4];
            raw_normals4((uint64_t)(path0 + i), (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < n_steps; ++j) {
                const double dz = vol * (double)z[j];
                cum[0] += drift + dz;
                cum[1] += drift - dz;
                for (int leg = 0; leg < 2; ++leg) {
                    if (cum[leg] > mx[leg]) mx[leg] = cum[leg];
                    if (cum[leg] < mn[leg]) mn[leg] = cum[leg];
                }
            }
        }
        for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
            const double st = S * exp(cum[leg]);
            double x;
            if (payoff <= 3) {
                const int crossed = payoff <= 1 ? (mx[leg] >= lb) : (mn[leg] <= lb);
                const int active = (payoff == 0 || payoff == 2) ? !crossed : crossed;
                x = active ? fmax(sign * (st - K), 0.0) : 0.0;
            } else if (payoff == 4) {
                x = is_call ? st - S * exp(mn[leg]) : S * exp(mx[leg]) - st;
            } else {
                x = is_call ? fmax(S * exp(mx[leg]) - K, 0.0) : fmax(K - S * exp(mn[leg]), 0.0);
            }
            m0 += x; m1 += x * x;
        }
    }
    moments[0] = (double)m0; moments[1] = (double)m1;
}

/* Heston full-truncation Euler (heston.py:184-255), stream tag 1, two steps per Philox block. */
void ol_heston_moments(double S, double K, double T, double r, double q, int is_call, double kappa, double theta,
                       double sigma_v, double rho, double v0, int64_t path0, int64_t n, int32_t n_steps, uint64_t seed,
                       int antithetic, double moments[2]) {
    const double dt = T / n_steps, sqrt_dt = sqrt(dt), rho_c = sqrt(1 - rho * rho), sign = is_call ? 1.0 : -1.0;
    const double zs = Z_SCALE * sqrt_dt;
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        double ls[2] = {log(S), log(S)}, v[2] = {v0, v0};
        for (int32_t b = 0; 2 * b < n_steps; ++b) {
            uint64_t path = (uint64_t)(path0 + i);
            uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)b, 1u};
            uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
            float z[4];
            ol_philox4x32_10(ctr, key, w);
            box_muller_raw(w[0], w[1], &z[0], &z[1]);
            box_muller_raw(w[2], w[3], &z[2], &z[3]);
            for (int h = 0; h < 2 && 2 * b + h < n_steps; ++h) {
                const double w1 = zs * (double)z[2 * h], w2 = rho * w1 + rho_c * (zs * (double)z[2 * h + 1]);
                for (int leg = 0; leg < 2; ++leg) {
                    const double sg = leg ? -1.0 : 1.0, vp = fmax(v[leg], 0.0), sv = sqrt(vp);
                    ls[leg] += ((r - q) * dt - 0.5 * vp * dt) + sv * (sg * w1);
                    v[leg] = fmax(v[leg] + kappa * dt * (theta - vp) + sigma_v * sv * (sg * w2), 0.0);
                }
            }
        }
        for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
            const double x = fmax(sign * (exp(ls[leg]) - K), 0.0);
            m0 += x; m1 += x * x;
        }
    }
    moments[0] = (double)m0; moments[1] = (double)m1;
}

/* HestonPricer.simulate_paths (heston.py:257-305) on the stream of ol_heston_moments' first leg,
 * time-major spot[t * n + i], var[t * n + i]; row 0 = (S, v0). */
void ol_heston_paths(double S, double T, double r, double q, double kappa, double theta, double sigma_v, double rho, double v0,
                     int64_t n, int32_t n_steps, uint64_t seed, double* spot, double* var) {
    const double dt = T / n_steps, sqrt_dt = sqrt(dt), rho_c = sqrt(1 - rho * rho), zs = Z_SCALE * sqrt_dt;
    for (int64_t i = 0; i < n; ++i) {
        double ls = log(S), v = v0;
        spot[i] = S;
        var[i] = v0;
        for (int32_t b = 0; 2 * b < n_steps; ++b) {
            uint32_t ctr[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)b, 1u};
            uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
            float z[4];
            ol_philox4x32_10(ctr, key, w);
            box_muller_raw(w[0], w[1], &z[0], &z[1]);
            box_muller_raw(w[2], w[3], &z[2], &z[3]);
            for (int h = 0; h < 2 && 2 * b + h < n_steps; ++h) {
                const double w1 = zs * (double)z[2 * h], w2 = rho * w1 + rho_c * (zs * (double)z[2 * h + 1]);
                const double vp = fmax(v, 0.0), sv = sqrt(vp);
                ls += ((r - q) * dt - 0.5 * vp * dt) + sv * w1;
                v = fmax(v + kappa * dt * (theta - vp) + sigma_v * sv * w2, 0.0);
                spot[(size_t)(2 * b + h + 1) * n + i] = exp(ls);
                var[(size_t)(2 * b + h + 1) * n + i] = v;
            }
        }
    }
}

/* Autocallable (exotic_options.py:404-491): moments of the per-path DISCOUNTED payoff. */
void ol_autocall_moments(double S, double T, double r, double sigma, double q, double autocall_b, double coupon_b,
                         double coupon_rate, double ki_b, int32_t freq, int64_t path0, int64_t n, int32_t n_steps,
                         uint64_t seed, int antithetic, double moments[2]) {
    (void)S;
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt) * Z_SCALE;
    const int32_t n_obs = n_steps / freq;
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        double cum[2] = {0, 0}, mn[2] = {0, 0}, pay[2] = {0, 0};
        int red[2] = {0, 0};
        for (int32_t b = 0; 4 * b < n_steps; ++b) {
            float z[4];
            raw_normals4((uint64_t)(path0 + i), (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < n_steps; ++j) {
                const int32_t t = 4 * b + j + 1;
                const double dz = vol * (double)z[j];
                for (int leg = 0; leg < 2; ++leg) {
                    cum[leg] += drift + (leg ? -dz : dz);
                    if (cum[leg] < mn[leg]) mn[leg] = cum[leg];
                    if (t % freq == 0 && !red[leg] && cum[leg] >= log(autocall_b)) {
                        red[leg] = 1;
                        pay[leg] = (1.0 + coupon_rate * ((double)(t / freq) / n_obs) * T) * exp(-(r * dt) * t);
                    }
                }
            }
        }
        for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
            double x = pay[leg];
            if (!red[leg]) {
                double fin = 1.0;
                if (cum[leg] >= log(coupon_b)) fin += coupon_rate * T;
                if (mn[leg] <= log(ki_b) && cum[leg] < 0.0) fin = exp(cum[leg]);
                x = fin * exp(-(r * dt) * n_steps);
            }
            m0 += x; m1 += x * x;
        }
    }
    moments[0] = (double)m0; moments[1] = (double)m1;
}

/* Cliquet (exotic_options.py:494-554): moments of the UNdiscounted payoff. */
void ol_cliquet_moments(double S, double T, double r, double sigma, double q, double lcap, double lfloor, double gcap,
                        double gfloor, int32_t n_periods, int64_t path0, int64_t n, int32_t n_steps, uint64_t seed,
                        int antithetic, double moments[2]) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt) * Z_SCALE;
    const int32_t spp = n_steps / n_periods, used = spp * n_periods;
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        double cum[2] = {0, 0}, start[2] = {0, 0}, total[2] = {0, 0};
        for (int32_t b = 0; 4 * b < used; ++b) {
            float z[4];
            raw_normals4((uint64_t)(path0 + i), (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < used; ++j) {
                const int32_t t = 4 * b + j + 1;
                const double dz = vol * (double)z[j];
                for (int leg = 0; leg < 2; ++leg) {
                    cum[leg] += drift + (leg ? -dz : dz);
                    if (t % spp == 0) {
                        double local = exp(cum[leg] - start[leg]) - 1.0;
                        local = local < lfloor ? lfloor : (local > lcap ? lcap : local);
                        total[leg] += local;
                        start[leg] = cum[leg];
                    }
                }
            }
        }
        for (int leg = 0; leg < (antithetic ? 2 : 1); ++leg) {
            double tot = total[leg] < gfloor ? gfloor : (total[leg] > gcap ? gcap : total[leg]);
            const double x = (tot > 0 ? tot : 0.0) * S;
            m0 += x; m1 += x * x;
        }
    }
    moments[0] = (double)m0; moments[1] = (double)m1;
}

/* American LSM (exotic_options.py:237-305) on the device's paths; regression in x = S/K by the
 * normal equations (long double Gaussian elimination in the natural order, singular directions pinned), degree <= 4.
 * moments[0..1] = sum, sum of squares of the time-0 cash flows. */
static int lsm_solve(const long double* mom, const long double* rhs, int degree, double* beta) {
    int n = degree + 1;
    long double a[5][6];
    for (int k = 0; k < n; ++k) {
        for (int l = 0; l < n; ++l) a[k][l] = mom[k + l];
        a[k][n] = rhs[k];
    }
    /* Natural order, as the device (olmc_kernels.h LsmFit): the moment matrix of the standardised regressor is symmetric positive
     * definite.  Where it is numerically singular -- the pivot below 1e-11 of its own diagonal entry sum z^(2 col): few distinct
     * in-the-money prices -- that unknown is pinned to zero and the remaining monomials are fitted (the device's rule; the reference's
     * lstsq gives the minimum-norm solution there, so per-seed parity with the reference is statistical for the American option). */
    for (int col = 0; col < n; ++col) {
        if (!(fabsl(a[col][col]) > 1e-11L * fabsl(mom[2 * col]))) {
            for (int l = 0; l <= n; ++l) a[col][l] = l == col ? 1.0L : 0.0L;
            for (int row = 0; row < n; ++row) if (row != col) a[row][col] = 0.0L;
        }
        for (int row = col + 1; row < n; ++row) {
            long double f = a[row][col] / a[col][col];
            for (int l = col; l <= n; ++l) a[row][l] -= f * a[col][l];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        long double v = a[k][n];
        for (int l = k + 1; l < n; ++l) v -= a[k][l] * beta[l];
        beta[k] = (double)(v / a[k][k]);
    }
    return 1;
}

/* The standardised regressor of the date-t regression: mean and standard deviation of S_t / K over the in-the-money side of the strike
 * under the lognormal law (restates lsm_regressor_scale of the device library's host side; any positive pair spans the same fit). */
static void lsm_scale(double S, double K, double r, double q, double sigma, double dt, int32_t t, int is_call, double* centre, double* inv_width) {
    *centre = 1.0;
    *inv_width = 1.0;
    const double m = log(S) + (r - q - 0.5 * sigma * sigma) * dt * t, s = sigma * sqrt(dt * t), a = log(K);
    if (!(s > 0.0) || !isfinite(m) || !isfinite(a)) return;
    const double side = is_call ? 1.0 : -1.0;
    const double p0 = 0.5 * erfc(-(side * (m - a) / s) * 0.70710678118654752440);
    const double m1 = exp(m + 0.5 * s * s) * (0.5 * erfc(-(side * (m + s * s - a) / s) * 0.70710678118654752440));
    const double m2 = exp(2.0 * m + 2.0 * s * s) * (0.5 * erfc(-(side * (m + 2.0 * s * s - a) / s) * 0.70710678118654752440));
    if (!(p0 > 1e-280) || !isfinite(m1) || !isfinite(m2)) return;
    const double mean = m1 / p0, var = m2 / p0 - mean * mean;
    const double width = fmax(sqrt(fmax(var, 0.0)), 1e-6 * mean);
    if (!(mean > 0.0) || !isfinite(width) || !(width > 0.0)) return;
    *centre = mean / K;
    *inv_width = K / width;
}

int ol_american_lsm(double S, double K, double T, double r, double sigma, double q, int is_call, int64_t n, int32_t n_steps,
                    int32_t degree, uint64_t seed, double moments[2]) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt) * Z_SCALE;
    const double sign = is_call ? 1.0 : -1.0, disc = exp(-r * dt);
    double* paths = (double*)malloc(sizeof(double) * (size_t)n * (n_steps + 1));
    double* cf = (double*)malloc(sizeof(double) * (size_t)n);
    if (!paths || !cf) { free(paths); free(cf); return 1; }
    for (int64_t i = 0; i < n; ++i) {
        double cum = 0.0;
        paths[i] = exp(log(S));
        for (int32_t b = 0; 4 * b < n_steps; ++b) {
            float z[4];
            raw_normals4((uint64_t)i, (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < n_steps; ++j) {
                cum += drift + vol * (double)z[j];
                paths[(size_t)(4 * b + j + 1) * n + i] = exp(log(S) + cum);
            }
        }
    }
    for (int64_t i = 0; i < n; ++i) cf[i] = fmax(sign * (paths[(size_t)n_steps * n + i] - K), 0.0);
    for (int32_t t = n_steps - 1; t >= 1; --t) {
        long double mom[9] = {0}, rhs[5] = {0};
        int64_t count = 0;
        double centre, inv_width;
        lsm_scale(S, K, r, q, sigma, dt, t, is_call, &centre, &inv_width);
        for (int64_t i = 0; i < n; ++i) {
            cf[i] *= disc;
            const double s = paths[(size_t)t * n + i];
            if (fmax(sign * (s - K), 0.0) > 0.0) {
                const double x = (s * (1.0 / K) - centre) * inv_width;
                double p = 1.0;
                for (int m = 0; m <= 2 * degree; ++m) {
                    mom[m] += p;
                    if (m <= degree) rhs[m] += p * cf[i];
                    p *= x;
                }
                ++count;
            }
        }
        double beta[5] = {0, 0, 0, 0, 0};
        if (count > degree + 1 && lsm_solve(mom, rhs, degree, beta)) {
            for (int64_t i = 0; i < n; ++i) {
                const double s = paths[(size_t)t * n + i], iv = fmax(sign * (s - K), 0.0);
                if (iv > 0.0) {
                    const double x = (s * (1.0 / K) - centre) * inv_width;
                    double cont = beta[4];
                    for (int k = 3; k >= 0; --k) cont = cont * x + beta[k];
                    if (iv > cont) cf[i] = iv;
                }
            }
        }
    }
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) { const double x = cf[i] * disc; m0 += x; m1 += x * x; }
    moments[0] = (double)m0; moments[1] = (double)m1;
    free(paths); free(cf);
    return 0;
}

/* Jump diffusion (jump_diffusion.py:160-225, 325-372): one Philox block per TWO steps, tag 2; jump sizes from tag 3 + j/2. */
static double unit_open64(uint32_t x) { return ((double)x + 0.5) * 0x1p-32; }

typedef struct { int kou; double a1, a2, a3, drift, vol, lam_dt, p0; } jump_model;

static jump_model jump_setup(double T, double r, double sigma, double q, int kou, double lambda_j, double a1, double a2, double a3,
                             int32_t n_steps) {
    jump_model m;
    const double dt = T / n_steps;
    const double kappa = kou ? a1 * a2 / (a2 - 1) + (1 - a1) * a3 / (a3 + 1) - 1 : exp(a1 + 0.5 * a2 * a2) - 1;
    m.kou = kou; m.a1 = a1; m.a2 = a2; m.a3 = a3;
    m.drift = (r - q - lambda_j * kappa - 0.5 * sigma * sigma) * dt;
    m.vol = sigma * sqrt(dt) * Z_SCALE;
    m.lam_dt = lambda_j * dt;
    m.p0 = exp(-m.lam_dt);
    return m;
}

/* jump part of step t given its Poisson uniform: count by inversion, sizes from stream tag 3 + j/2 */
static void jump_sizes(const jump_model* m, uint64_t path, int32_t t, uint64_t seed, double u, double* ls) {
    if (u < m->p0) return;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int nj = 1;
    double pk = m->p0 * m->lam_dt, cdf = m->p0 + pk;
    while (u >= cdf && nj < 64) { ++nj; pk *= m->lam_dt / nj; cdf += pk; }
    if (!m->kou) {
        uint32_t c2[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)t, 3u}, k[4];
        float zj, unused;
        ol_philox4x32_10(c2, key, k);
        box_muller_raw(k[0], k[1], &zj, &unused);
        *ls += nj * m->a1 + m->a2 * sqrt((double)nj) * (Z_SCALE * (double)zj);
        return;
    }
    for (int j = 0; j < nj; ++j) {
        uint32_t c2[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)t, 3u + (uint32_t)(j >> 1)}, k[4];
        ol_philox4x32_10(c2, key, k);
        const double ud = unit_open64((j & 1) ? k[2] : k[0]), um = unit_open64((j & 1) ? k[3] : k[1]);
        *ls += ud < m->a1 ? -log(um) / m->a2 : log(um) / m->a3;
    }
}

/* step t of one path: block t/2 of stream tag 2 holds the diffusion normals of steps 2b, 2b+1 (cos, sin of
 * words 0, 1) and their Poisson uniforms (words 2, 3) */
static void jump_step(const jump_model* m, uint64_t path, int32_t t, uint64_t seed, double* ls) {
    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)(t >> 1), 2u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[4];
    float z0, z1;
    ol_philox4x32_10(ctr, key, w);
    box_muller_raw(w[0], w[1], &z0, &z1);
    *ls += m->vol * (double)((t & 1) ? z1 : z0) + m->drift;
    jump_sizes(m, path, t, seed, unit_open64(w[2 + (t & 1)]), ls);
}

void ol_jump_moments(double S, double K, double T, double r, double sigma, double q, int is_call, int kou, double lambda_j,
                     double a1, double a2, double a3, int64_t path0, int64_t n, int32_t n_steps, uint64_t seed,
                     double moments[2]) {
    const double sign = is_call ? 1.0 : -1.0;
    const jump_model m = jump_setup(T, r, sigma, q, kou, lambda_j, a1, a2, a3, n_steps);
    long double m0 = 0, m1 = 0;
    for (int64_t i = 0; i < n; ++i) {
        double ls = log(S);
        for (int32_t t = 0; t < n_steps; ++t) jump_step(&m, (uint64_t)(path0 + i), t, seed, &ls);
        const double x = fmax(sign * (exp(ls) - K), 0.0);
        m0 += x; m1 += x * x;
    }
    moments[0] = (double)m0; moments[1] = (double)m1;
}

/* MertonJumpDiffusion.simulate_path (jump_diffusion.py:227-272) for n paths, time-major out[t * n + i], row 0 = S. */
void ol_jump_paths(double S, double T, double r, double sigma, double q, int kou, double lambda_j, double a1, double a2, double a3,
                   int64_t n, int32_t n_steps, uint64_t seed, double* out) {
    const jump_model m = jump_setup(T, r, sigma, q, kou, lambda_j, a1, a2, a3, n_steps);
    for (int64_t i = 0; i < n; ++i) {
        double ls = log(S);
        out[i] = S;
        for (int32_t t = 0; t < n_steps; ++t) {
            jump_step(&m, (uint64_t)i, t, seed, &ls);
            out[(size_t)(t + 1) * n + i] = exp(ls);
        }
    }
}

/* Full paths, time-major out[t * n + i], t = 0..n_steps (gbm_numpy.py:86-118 transposed). */
void ol_gbm_paths(double S, double T, double r, double sigma, double q, int64_t n, int32_t n_steps, uint64_t seed, double* out) {
    const double dt = T / n_steps, drift = (r - q - 0.5 * sigma * sigma) * dt, vol = sigma * sqrt(dt) * Z_SCALE;
    for (int64_t i = 0; i < n; ++i) {
        double cum = 0.0;
        out[i] = S;
        for (int32_t b = 0; 4 * b < n_steps; ++b) {
            float z[4];
            raw_normals4((uint64_t)i, (uint32_t)b, seed, z);
            for (int j = 0; j < 4 && 4 * b + j < n_steps; ++j) {
                cum += drift + vol * (double)z[j];
                out[(size_t)(4 * b + j + 1) * n + i] = exp(log(S) + cum);
            }
        }
    }
}
