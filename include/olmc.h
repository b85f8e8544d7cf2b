/*
 * olmc.h -- C ABI of libolmc.so, the MI355X (gfx950) Monte Carlo path engine.
 *
 * This is the drop-in boundary for the OptionsLab hot path: every entry point
 * names the reference interface it replaces (paths relative to the reference
 * repository root).  The reference is pure Python, so the binding a maintainer
 * adds is a ctypes stub (INTEGRATION.md shows it); the signatures therefore
 * use only plain C scalars, pointers and sizes.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message
 *     for the calling thread's last failure is olmc_last_error();
 *   - the caller owns every buffer it passes in; the library owns its device
 *     scratch (grown lazily, freed by olmc_shutdown) and returns no pointer
 *     that outlives the call except the thread-local error string;
 *   - Threading: entry points are re-entrant AND concurrent.  A call leases one of up to 8 contexts of its device (its own
 *     stream, reduction workspace, pinned landing buffer and completion word) for its duration, so calls from different
 *     threads overlap on the host and on the device instead of queueing behind one mutex (Streamlit runs a thread per
 *     session; a ninth concurrent caller waits for a lease).  Results do not depend on which context served a call.  A
 *     single-threaded caller always gets context 0.  olmc_shutdown must not race with calls in flight;
 *   - a blocking call waits by polling a completion word in pinned memory: it spins for 200 us, yields between polls up to 2 ms and
 *     naps (20 us) between them from then on.  While it naps, the CALLING thread's timer slack is 1 us (prctl(PR_SET_TIMERSLACK); the
 *     previous value is restored before the call returns) -- at the default 50 us every nap returned as late again as it was long;
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with OLMC_ERR_HIP.
 *
 * Random stream (identical whatever the grid shape / GPU count):
 *   Philox4x32-10, key = (seed_lo32, seed_hi32),
 *   counter = (path_lo32, path_hi32, block, stream_tag), `path` = GLOBAL path
 *   index, `block` = step/4.  The four output words give the four normals of
 *   steps 4*block .. 4*block+3 by two Box-Muller transforms:
 *     u_a = (x_a + 0.5) * 2^-32 (fp32),  rad = sqrt(-2 ln u_a),
 *     u_b = (x_b & 0x7fffff) * 2^-23     (turn fraction, 23 bits),
 *     z_even = rad * cos(2 pi u_b),  z_odd = rad * sin(2 pi u_b),
 *   pair (x0,x1) -> steps 4b,4b+1; pair (x2,x3) -> steps 4b+2,4b+3.
 *   sum_t Z of a path is accumulated in fp32 within 16 normals (4 Philox blocks)
 *   and in fp64 across those groups -- the same bits whatever the grid shape.
 *   Normals are fp32; every quantity that depends on S, K, T, r, sigma, q is
 *   fp64 (finite-difference Greeks under common random numbers stay smooth).
 */
#ifndef OLMC_H
#define OLMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OLMC_ABI_VERSION 6   /* 2: olmc_asian avg_kind 0 is the fp64-exponent kernel (2 = the fp32 one); probes, tune knobs 5-8
                              * 3: additions only -- olmc_european_qmc_batch, olmc_european_qmc_greeks_fd, olmc_multi_capacity,
                              *    olmc_exp2_probe_form, olmc_phase_stamps, olmc_contract_layout, tune knob 9; every v2 entry point keeps its signature and meaning
                              * 4: the measurement entry points (olmc_*probe*, olmc_phase_stamps, olmc_clock_probe, olmc_normal_moments) and the
                              *    fault-injection knobs 5 / 6 LEFT this library for the instrumented build (olmc_probe.h, libolmc_probe.so); added
                              *    olmc_multi_gpu_greeks_fd, olmc_multi_gpu_european_cv, olmc_asian_greeks_fd, olmc_extrema_greeks_fd; every pricing entry point keeps its signature and meaning;
                              *    entry points are now concurrent across threads (a context per caller, olmc.h "Threading")
                              * 5: additions only -- olmc_multi_gpu_european_qmc, olmc_multi_gpu_spans, tune knobs 10, 11 (OLMC_TUNE_MULTI_LAUNCH, OLMC_TUNE_STAGED_COPY); the multi-GPU
                              *    entry points launch their ranks from one launcher thread per device; the grid reduction's consumer side is an
                              *    agent-scope acquire again (numbers unchanged) 
                              * 6: additions only -- olmc_multi_gpu_european_qmc_greeks_fd, olmc_multi_gpu_european_qmc_cv; OLMC_TUNE_QMC_BLOCK takes 2 */

enum {
    OLMC_OK = 0,
    OLMC_ERR_ARG = 1,     /* bad argument (n_paths < 1, n_steps < 1, null pointer, k out of range) */
    OLMC_ERR_HIP = 2,     /* HIP runtime / device failure                                         */
    OLMC_ERR_STATE = 3,   /* olmc_init not called / wrong device                                   */
    OLMC_ERR_RCCL = 4     /* RCCL failure in the multi-GPU entry points                            */
};

enum { OLMC_STREAM_GBM = 0, OLMC_STREAM_HESTON = 1, OLMC_STREAM_JUMP = 2, OLMC_STREAM_KOU = 3 };   /* counter word 3 (stream_tag); batches use tag = contract index */
enum { OLMC_AVG_ARITHMETIC = 0, OLMC_AVG_GEOMETRIC = 1, OLMC_AVG_ARITHMETIC_FAST = 2 };
#define OLMC_MAX_BATCH 16                        /* parameter sets per fused launch      */

/* Result of one reduction.  Mirrors what MonteCarloPricer.price() derives from
 * the 2N payoffs (src/pricing_models/monte_carlo.py:140-150):
 *   price     = exp(-rT) * sum / n
 *   std_error = exp(-rT) * sqrt(sumsq/n - (sum/n)^2) / sqrt(n)      (ddof = 0)
 * sum / sumsq are UNdiscounted payoff moments so shards can be added. */
typedef struct olmc_stats {
    double  sum;
    double  sumsq;
    int64_t n;          /* payoff samples = n_paths * (1 + antithetic)  (MCResult.n_paths) */
    double  price;
    double  std_error;
} olmc_stats;

/* One European contract; `is_call` != 0 -> max(S_T-K,0), else max(K-S_T,0)
 * (monte_carlo.py:140-143: anything that is not "call" prices as a put). */
typedef struct olmc_option {
    double  S, K, T, r, sigma, q;
    int32_t is_call;
    int32_t reserved;
} olmc_option;

/* Five moments of the control-variate estimator
 * (monte_carlo.py:154-186): d = discounted payoff, s = terminal price. */
typedef struct olmc_cv_moments {
    double  sum_d, sum_s, sum_dd, sum_ss, sum_ds;
    int64_t n;
    double  value;      /* mean(d) - beta * (mean(s) - S e^{(r-q)T}), beta = cov/var (ddof=1) */
} olmc_cv_moments;

typedef struct olmc_devinfo {
    char    name[128];
    char    arch[32];
    int32_t compute_units;
    int32_t clock_mhz;
    int32_t wavefront;
    int32_t device;
    int64_t hbm_bytes;
} olmc_devinfo;

/* ---- lifetime ---------------------------------------------------------- */
int         olmc_abi_version(void);
int         olmc_init(int device);             /* idempotent per device; selects it for the calling thread */
int         olmc_shutdown(void);               /* frees scratch + streams of every initialised device      */
const char* olmc_last_error(void);             /* thread-local, never NULL                                 */
int         olmc_device_info(olmc_devinfo* out);

/* ---- European terminal payoff: GBM paths + payoff + on-device reduction --
 * Replaces MonteCarloPricer._simulate + the payoff/mean/std tail of .price()
 * (monte_carlo.py:74-106, 137-150) and simulate_gbm_numpy / _fast
 * (src/simulation/gbm_numpy.py:15-53, 56-83).  n_steps == 1 is the reference's
 * single-step closed form.  Only (sum, sumsq) leave the device. */
int olmc_european(double S, double K, double T, double r, double sigma, double q, int is_call,
                  int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                  olmc_stats* out);

/* Same for the global path range [path_offset, path_offset + n_local): the unit a
 * rank owns when independent path batches are sharded over GPUs.
 * out->price / std_error are those of the shard alone; combine with
 * olmc_combine_stats after summing (sum, sumsq, n) over ranks. */
int olmc_european_shard(double S, double K, double T, double r, double sigma, double q, int is_call,
                        int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                        int antithetic, olmc_stats* out);

/* Device-resident variant for one-process-per-GPU jobs: writes the raw triple
 * {sum, sumsq, (double)n} to `d_triple` (3 doubles of DEVICE memory owned by
 * the caller, e.g. a torch tensor handed to an RCCL all-reduce) on `hip_stream`
 * (a hipStream_t used as given; NULL = the HIP null stream, which is torch's
 * default stream) and does not synchronise: later work on that stream is
 * ordered after it.  Calls on DIFFERENT streams may overlap on the device: the
 * library rotates event-guarded reduction workspaces, so they never share rows
 * or counters (a launch waits, on its own stream, for the previous user of its
 * workspace). */
int olmc_european_shard_dev(double S, double K, double T, double r, double sigma, double q, int is_call,
                            int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                            int antithetic, double* d_triple, void* hip_stream);

/* k <= OLMC_MAX_BATCH contracts priced on the SAME normals (common random
 * numbers) in one pass: one RNG stream, k payoffs per path, 2k sums.  This is
 * the fused form of the 8 / 14 price() calls compute_greeks_unified makes
 * (src/greeks/unified_greeks.py:280-358).  All contracts share n_steps, so
 * they share sum_t Z exactly as the reference's re-seeded calls do.
 * Rounding: a contract's payoff is formed in one of two ways, by launch shape.  Launches whose grid covers every path (up to
 * 2^26 paths) use max(fma(sign * scale, S_T(base), -sign * K), 0) -- ONE rounding of sign * (scale * S_T - K); grid-striding
 * launches beyond that and the Sobol batch kernels use sign * (scale * S_T(base) - K) with scale * S_T rounded first -- two.  For
 * a contract that is its own base (scale = 1: every contract of k = 1, the first of each group of equal vol) the two are the
 * same bits; for a scaled contract (the S- and r-bumps of a Greeks set) they differ by at most one ulp of S_T per sample, i.e.
 * ~1e-16 relative on a price -- below the 1e-13 the tests allow between the fused and the literal forms. */
int olmc_european_batch(const olmc_option* opts, int32_t k,
                        int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                        int antithetic, olmc_stats* out /* [k] */);

/* n_options INDEPENDENT contracts in one launch (grid.y = contract), each with its own
 * Philox stream selected by its tag (counter word 3; tags == NULL -> tag j = j) and
 * its own on-device reduction.  Replaces MonteCarloPricerUni.price_batch /
 * _simulate_terminal_prices (src/pricing_models/monte_carlo_unified.py:298-343, 562-631),
 * where option j consumes its own slice of the normals; equal tags give common random
 * numbers (delta_gamma_batch, :633-689: the S-h / S / S+h copies of option j share tag j). */
int olmc_european_multi(const olmc_option* opts, const uint32_t* tags, int64_t n_options,
                        int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                        olmc_stats* out /* [n_options] */);

/* How a set of k (2 .. OLMC_MAX_BATCH) contracts is laid out for the fused kernels (host arithmetic only; no device needed):
 * nsets = 8 or 16 slots; pos[i] = slot of contract i; bit s of base_mask = slot s evaluates its own exponentials (its
 * sigma * sqrt(dt) differs bit-wise from every earlier base's, or it opens the second half of the set); scale16[s] = 1 for a base,
 * else exp(a_s - a_base), a = ln S + (r - q - sigma^2 / 2) dt n_steps; upper_continues_slot0 != 0: slot nsets / 2 is not a base
 * but belongs to slot 0's group.  For tests of the layout logic. */
int olmc_contract_layout(const olmc_option* opts, int32_t k, int32_t n_steps, int32_t* nsets, int32_t* pos /* [k] */,
                         uint32_t* base_mask, int32_t* upper_continues_slot0, double* scale16 /* [16] */);

/* Capacity of the library-owned workspace behind olmc_european_multi as it stands: out2 = {contracts, workgroups per contract}.
 * The two grow independently (more contracts doubles the first, more paths per contract only widens the rows). */
int olmc_multi_capacity(int64_t* out2);

/* Finite-difference Greeks, bumps exactly as unified_greeks.py:274-277, 295-362.
 * out9 = {price, delta, gamma, vega, theta, rho, vanna, charm, vomma}; the last
 * three are written only when second_order != 0.  `evals` (nullable) receives
 * the 8 or 14 per-evaluation stats in the reference's call order; with evals == NULL
 * the launch reduces the prices only (no sums of squares: nobody would read the
 * standard errors), which is the faster form. */
int olmc_european_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                            int64_t n_paths, int32_t n_steps, uint64_t seed, int second_order,
                            double* out9, olmc_stats* evals /* [14] or NULL */);

/* Terminal prices to caller-owned HOST memory, length n_paths*(1+antithetic),
 * layout [pos(0..N-1) | neg(0..N-1)] (gbm_numpy.py:51).  This is the backend
 * contract simulate_*(S,T,r,sigma,q,n_paths,n_steps,seed) -> ndarray
 * (src/simulation/__init__.py:5-6). */
int olmc_european_terminal(double S, double T, double r, double sigma, double q,
                           int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                           double* out_host);

/* Full GBM paths to caller-owned HOST memory; t = 0 .. n_steps, t = 0 is the spot.  Replaces
 * simulate_gbm_paths (src/simulation/gbm_numpy.py:86-118); no antithetic mirror there either.
 *   path_major != 0: out_host[i * (n_steps + 1) + t] -- the reference's (n_paths, n_steps + 1) C-order
 *                    array, written in that layout by the kernel (no host transpose);
 *   path_major == 0: out_host[t * n_paths + i] (coalesced device writes; one date per row).
 * The same flag has the same meaning in olmc_heston_paths and olmc_jump_paths. */
int olmc_gbm_paths(double S, double T, double r, double sigma, double q, int64_t n_paths,
                   int32_t n_steps, uint64_t seed, int path_major, double* out_host);

/* Control-variate estimator, five moments reduced on device
 * (MonteCarloPricer.price_with_control_variate, monte_carlo.py:154-186). */
int olmc_european_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                     int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                     olmc_cv_moments* out);

/* The same for the global path range [path_offset, path_offset + n_local) (a rank's shard; out->value is
 * the shard's own estimate), and the estimate from moments summed over shards: the all-reduce payload of
 * the control variate is the five sums + n (count = 5). */
int olmc_european_cv_shard(double S, double K, double T, double r, double sigma, double q, int is_call,
                           int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                           int antithetic, olmc_cv_moments* out);
int olmc_combine_cv(const olmc_cv_moments* parts, int32_t n_parts, double S, double T, double r, double q,
                    olmc_cv_moments* out);

/* ---- Asian (average over t = 1..M, t = 0 excluded) ------------------------
 * Replaces ExoticOptionBase._generate_paths + AsianOption.price
 * (src/pricing_models/exotic_options.py:40-67, 97-131): running sum of S_t
 * (arithmetic) or of log S_t (geometric) in registers, no path matrix.
 * avg_kind: OLMC_AVG_ARITHMETIC      the reference's arithmetic -- fp64 cumulative log-return advanced date by date,
 *                                    one full fp64 exponential per date (np.exp(log_S), :62-67), fp64 running sum;
 *           OLMC_AVG_GEOMETRIC       sum of log S_t, fp64 across groups of 16 dates;
 *           OLMC_AVG_ARITHMETIC_FAST opt-in: one hardware v_exp_f32 per date on an exponent rounded to fp32
 *                                    (~1e-7 relative per term, unbiased; 2e-6 on a price against the fp64 form,
 *                                    CRN finite differences smooth: tests/test_gpu_exotics.py) -- 1.6x faster. */
int olmc_asian(double S, double K, double T, double r, double sigma, double q, int is_call,
               int avg_kind, int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
               int antithetic, olmc_stats* out);

/* Finite-difference Greeks of the Asian option -- avg_kind OLMC_AVG_ARITHMETIC (the reference's precision) or OLMC_AVG_GEOMETRIC (whose
 * fused form costs ONE geometric pricing: between group boundaries the path state is the same for every contract) -- in ONE launch: the 8 / 14 bumped
 * contracts compute_greeks_unified prices through ExoticAdapter(AsianOption) (src/greeks/unified_greeks.py:177-227, 295-358: same
 * bumps, same call order, same formulas as olmc_european_greeks_fd) on the SAME normals.  The contracts are at most six distinct
 * path recursions ({mid, S+-}, sigma+-, T-, r+-: the spot only scales the average); in the arithmetic kernel the two r bumps are not
 * even recursions -- their price relative at date t is the mid contract's times exp(+-h_r dt t), one factor per date for every path --
 * so a date costs FOUR exponentials instead of 8 / 14 in 8 / 14 launches.  Every evaluation is its own launch's result to rounding
 * (bitwise but for the two r evaluations, which agree to ~1e-15 relative).  out9 / evals as olmc_european_greeks_fd. */
int olmc_asian_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call, int avg_kind,
                         int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic, int second_order,
                         double* out9, olmc_stats* evals /* [14] or NULL */);

/* ---- barrier and lookback (running extrema of the path, t = 0 included) ------
 * Replace BarrierOption.price (src/pricing_models/exotic_options.py:174-224) and
 * LookbackOption.price (:359-401) on ExoticOptionBase._generate_paths (:40-67): the
 * running max / min of ln S_t live in registers, no path matrix.  barrier_kind:
 * OLMC_BARRIER_*; fixed_strike: 0 = floating (call S_T - S_min, put S_max - S_T),
 * 1 = fixed (call max(S_max - K, 0), put max(K - S_min, 0)). */
enum { OLMC_BARRIER_UP_OUT = 0, OLMC_BARRIER_UP_IN = 1, OLMC_BARRIER_DOWN_OUT = 2, OLMC_BARRIER_DOWN_IN = 3 };
int olmc_barrier(double S, double K, double T, double r, double sigma, double q, int is_call,
                 double barrier, int barrier_kind, int64_t path_offset, int64_t n_local,
                 int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out);
int olmc_lookback(double S, double K, double T, double r, double sigma, double q, int is_call,
                  int fixed_strike, int64_t path_offset, int64_t n_local, int32_t n_steps,
                  uint64_t seed, int antithetic, olmc_stats* out);

/* Finite-difference Greeks of a barrier / lookback option in ONE launch: the 8 / 14 bumped contracts compute_greeks_unified prices
 * through ExoticAdapter(BarrierOption | LookbackOption) (src/greeks/unified_greeks.py:177-227, 295-358; live caller
 * streamlit_app/pages/7_Exotic_Options.py:266-284) on the SAME normals, as at most six recursions of (cumulative log-return, running
 * max, running min) -- see olmc_asian_greeks_fd.  payoff: OLMC_BARRIER_UP_OUT .. OLMC_BARRIER_DOWN_IN (`barrier` = the level) or
 * OLMC_LOOKBACK_FLOATING / OLMC_LOOKBACK_FIXED (`barrier` ignored).  out9 / evals as olmc_european_greeks_fd. */
enum { OLMC_LOOKBACK_FLOATING = 4, OLMC_LOOKBACK_FIXED = 5 };
int olmc_extrema_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call, int payoff, double barrier,
                           int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic, int second_order,
                           double* out9, olmc_stats* evals /* [14] or NULL */);

/* ---- structured products on the step loop ----------------------------------------
 * olmc_autocallable replaces AutocallableOption.price (src/pricing_models/exotic_options.py:404-491):
 * barriers are relative to spot; out->price = mean of the per-path DISCOUNTED payoffs (the reference
 * discounts each redemption at its own date), fraction of notional.
 * olmc_cliquet replaces CliquetOption.price (:494-554): n_periods resets of n_steps // n_periods steps. */
int olmc_autocallable(double S, double T, double r, double sigma, double q, double autocall_barrier,
                      double coupon_barrier, double coupon_rate, double ki_barrier,
                      int32_t observation_freq, int64_t path_offset, int64_t n_local,
                      int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out);
int olmc_cliquet(double S, double T, double r, double sigma, double q, double local_cap,
                 double local_floor, double global_cap, double global_floor, int32_t n_periods,
                 int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                 int antithetic, olmc_stats* out);

/* ---- American option, Longstaff-Schwartz LSM --------------------------------------
 * Replaces AmericanOption.price (src/pricing_models/exotic_options.py:227-305): stores the path
 * matrix in HBM (time-major), one launch per exercise date doing {regression of the later date from the
 * sums the launch before it left (every workgroup sums them in index order and solves the small normal
 * equations itself), exercise decision of that date, one-step discount, regression sums of this date};
 * no host round trip per date, the host waits once.  Polynomial of degree poly_degree in [1, 4] (the
 * reference's default is 3) in S/K -- the space of the reference's raw powers of S, so the same fit in exact
 * arithmetic -- written in the standardised regressor (S/K - c_t) / w_t (mean and width of S_t/K over the
 * in-the-money side under the model's own law), which keeps the normal equations well conditioned where the
 * reference relies on lstsq's SVD.  Where the moment matrix is numerically singular all the same (few distinct
 * in-the-money prices: a pivot below 1e-11 of its diagonal entry) that monomial is left out of the date's fit; the
 * reference's lstsq returns the minimum-norm solution there.  PARITY with the reference is therefore STATISTICAL for
 * this entry point, per-seed parity unpinned: other normals (Philox, not PCG64) AND, in degenerate regressions, another
 * choice among the equally good fits -- gated by 3-sigma tests against the reference's own algorithm over 24 seeds and on
 * ill-conditioned cases (tests/test_gpu_exotics.py), bit-level only against the build's own checker.
 * out->price = mean of the time-0 cash flows; single device. */
int olmc_american_lsm(double S, double K, double T, double r, double sigma, double q, int is_call,
                      int64_t n_paths, int32_t n_steps, int32_t poly_degree, uint64_t seed,
                      olmc_stats* out);

/* AmericanOption.early_exercise_boundary (src/pricing_models/exotic_options.py:309-345): per date the
 * 10th (put) / 90th (call) percentile of the in-the-money prices of the LSM path set (same stream as
 * olmc_american_lsm), NumPy's linear interpolation; NaN where no path is in the money.  The order
 * statistics are selected on the device (radix select per date); boundary_host[n_steps + 1]. */
int olmc_exercise_boundary(double S, double K, double T, double r, double sigma, double q, int is_call,
                           int64_t n_paths, int32_t n_steps, uint64_t seed, double* boundary_host);

/* ---- Heston stochastic volatility, full-truncation Euler ----------------------
 * Replaces HestonPricer.price_monte_carlo (src/pricing_models/heston.py:184-255): two
 * normals per step, (ln S, v) in fp64 registers, Philox stream tag 1.  The reference
 * has no antithetic mirror here (antithetic = 0 reproduces its n = n_paths samples). */
int olmc_heston(double S, double K, double T, double r, double q, int is_call,
                double kappa, double theta, double sigma_v, double rho, double v0,
                int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                int antithetic, olmc_stats* out);

/* HestonPricer.simulate_paths (heston.py:257-305): the states of olmc_heston's recursion on the same
 * stream (non-antithetic leg), t = 0 .. n_steps, date 0 = (S, v0); layout as olmc_gbm_paths
 * (path_major != 0: the reference's two (n_paths, n_steps + 1) arrays). */
int olmc_heston_paths(double S, double T, double r, double q, double kappa, double theta, double sigma_v,
                      double rho, double v0, int64_t n_paths, int32_t n_steps, uint64_t seed,
                      int path_major, double* spot_host, double* var_host);

/* ---- jump diffusion ----------------------------------------------------------------
 * Replaces MertonJumpDiffusion.price_monte_carlo (src/pricing_models/jump_diffusion.py:160-225)
 * and KouJumpDiffusion.price_monte_carlo (:325-372): per step a diffusion normal, a
 * Poisson(lambda dt) jump count and the jump sum, compensated drift.  model = OLMC_JUMP_MERTON:
 * (a1, a2) = (mu_j, sigma_j);  OLMC_JUMP_KOU: (a1, a2, a3) = (p, eta1, eta2).  No antithetic.
 * Streams: Philox block (path, b, tag 2) feeds steps 2b and 2b+1 (words 0,1 -> the two diffusion normals,
 * words 2,3 -> the two Poisson uniforms); the jump sizes of step t come from blocks (path, t, tag 3 + j/2). */
enum { OLMC_JUMP_MERTON = 0, OLMC_JUMP_KOU = 1 };
int olmc_jump_diffusion(double S, double K, double T, double r, double sigma, double q, int is_call,
                        int model, double lambda_j, double a1, double a2, double a3,
                        int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                        olmc_stats* out);

/* MertonJumpDiffusion.simulate_path (jump_diffusion.py:227-272) for n_paths paths (the reference draws
 * one): the prices of olmc_jump_diffusion's recursion on the same stream, t = 0 .. n_steps, date 0 = S;
 * layout as olmc_gbm_paths.  Kou paths come for free (model = 1). */
int olmc_jump_paths(double S, double T, double r, double sigma, double q, int model, double lambda_j,
                    double a1, double a2, double a3, int64_t n_paths, int32_t n_steps, uint64_t seed,
                    int path_major, double* out_host);

/* ---- quasi-Monte Carlo (MCMethod.QMC) --------------------------------------
 * Replaces simulate_gbm_qmc (src/simulation/gbm_qmc.py:14-46): scrambled-Sobol
 * points -> clip [1e-10, 1-1e-10] -> inverse normal (fp64) -> sum over dims ->
 * terminal price; no antithetic mirror, n = n_paths payoffs.  `sv` is the
 * [dims][bits] scrambled direction matrix and `shift` the [dims] digital shift
 * of scipy.stats.qmc.Sobol(d=dims, scramble=True, seed) (host memory, uint32,
 * bits must be 30 = SciPy's default): point k = shift ^ XOR_{b in gray(k)} sv[:, b],
 * u = x * 2^-bits -- the same uniforms as Sobol.random(n), bit for bit.
 * dims = min(n_steps, 21201) as in the reference (:30). */
int olmc_european_qmc(double S, double K, double T, double r, double sigma, double q, int is_call,
                      int64_t point_offset, int64_t n_paths, int32_t dims,
                      const uint32_t* sv, const uint32_t* shift, int32_t bits, olmc_stats* out);
/* price_with_control_variate on the Sobol points (monte_carlo.py:154-186 with MCMethod.QMC): the five moments
 * and the estimate, reduced on the device like olmc_european_cv. */
int olmc_european_qmc_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                         int64_t point_offset, int64_t n_paths, int32_t dims,
                         const uint32_t* sv, const uint32_t* shift, int32_t bits, olmc_cv_moments* out);

/* k <= OLMC_MAX_BATCH contracts on the SAME Sobol points in ONE launch: the points, their uniforms and the inverse normals are
 * formed once, each contract adds its exponential (contracts whose sigma sqrt(T / dims) agrees bit for bit share it) and its
 * payoff.  out[i] = what olmc_european_qmc answers for opts[i] alone, to the last few ulp.  And the finite-difference Greeks of
 * compute_greeks_unified (unified_greeks.py:280-358) over it -- bumps, evaluation order and out9 / evals exactly as
 * olmc_european_greeks_fd -- for a MCMethod.QMC pricer: one launch where the literal form makes 8 or 14. */
int olmc_european_qmc_batch(const olmc_option* opts, int32_t k, int64_t point_offset, int64_t n_paths, int32_t dims,
                            const uint32_t* sv, const uint32_t* shift, int32_t bits, olmc_stats* out /* [k] */);
int olmc_european_qmc_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                                int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
                                int second_order, double* out9, olmc_stats* evals /* [14] or NULL */);

/* antithetic != 0: simulate_gbm_qmc_antithetic (gbm_qmc.py:49-76), 2 * n_paths values [pos | neg]. */
int olmc_european_qmc_terminal(double S, double T, double r, double sigma, double q,
                               int64_t point_offset, int64_t n_paths, int32_t dims,
                               const uint32_t* sv, const uint32_t* shift, int32_t bits,
                               int antithetic, double* out_host /* [n_paths * (1 + antithetic)] */);

/* ---- multi-GPU, single process ------------------------------------------
 * n_paths split into n_gpus contiguous global path ranges (rank d = device d, [d N / P, (d + 1) N / P)).  Per list of devices the
 * library keeps an engine: a stream, a send / receive buffer and a LAUNCHER THREAD per rank (bound to the rank's device once, at
 * birth; parked on a futex between calls) and the list's RCCL communicators (ncclCommInitAll).  A call posts the launch to the
 * launchers -- every rank's path kernel is queued at the same time --, then the calling thread queues ONE grouped RCCL all-reduce
 * over xGMI (only after every rank has launched: a failed rank leaves no peer inside a collective; the group is closed on every
 * error path), and the reduced sums are handed to the host by rank 0's polled completion word (as olmc_fetch_dev) while the
 * launchers drain the other ranks, which hold the same sums.  Identical finalisation on every rank.  On any error
 * return the thread's device and the streams already launched on are restored / drained.  Calls on lists that share no device run
 * concurrently; lists that share a device take turns.  Payload of the all-reduce:
 *   olmc_multi_gpu_european      {sum, sumsq, n}                                   count = 3
 *   olmc_multi_gpu_greeks_fd     the 8 / 14 bumped contracts of olmc_european_greeks_fd on the SAME normals, one launch per rank:
 *                                {sum, sumsq} x 8 or 16 slots, n                   count = 17 / 33
 *   olmc_multi_gpu_european_cv   the five control-variate moments, n              count = 6
 *   olmc_multi_gpu_european_qmc  {sum, sumsq, n} of the rank's block of Sobol POINTS (src/simulation/gbm_qmc.py:14-46)   count = 3
 *                                (inner boundaries on multiples of 512 points where a rank owns >= 4,096: every rank's point offset
 *                                is one the aligned kernels take; sharding.qmc_shard_bounds cuts the same way)
 *   olmc_multi_gpu_european_qmc_greeks_fd   the 8 / 14 bumped contracts of olmc_european_qmc_greeks_fd on the rank's block of the SAME
 *                                Sobol points, one launch per rank: {sum, sumsq} x 8 or 16 slots, n   count = 17 / 33
 *   olmc_multi_gpu_european_qmc_cv   the five control-variate moments of the rank's block of Sobol points, n   count = 6
 * Prices agree with the one-GPU entry points to the rounding of the sums' association (same paths whatever n_gpus is: the Philox
 * counter carries the global path index, the Sobol kernels take the global point index).
 *
 * STATUS: with n_gpus >= 2 the RCCL branch is UNVERIFIED ON HARDWARE -- no box with more than one GPU has been available to this
 * build.  What is verified: n_gpus = 1 through RCCL on a real device; 1 .. 12 ranks REHEARSED on one device in the instrumented
 * build (same engine, same launchers, the collective replaced by a kernel that adds the send buffers in rank order);
 * tests/test_gpu_multi_device.py compares 2 .. N devices with the one-device results and skips itself where only one is visible. */
int olmc_multi_gpu_european(double S, double K, double T, double r, double sigma, double q, int is_call,
                            int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                            int n_gpus, olmc_stats* out);
int olmc_multi_gpu_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                             int64_t n_paths, int32_t n_steps, uint64_t seed, int second_order,
                             int n_gpus, double* out9, olmc_stats* evals /* [14] or NULL */);
int olmc_multi_gpu_european_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                               int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                               int n_gpus, olmc_cv_moments* out);
int olmc_multi_gpu_european_qmc(double S, double K, double T, double r, double sigma, double q, int is_call,
                                int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
                                int n_gpus, olmc_stats* out);
int olmc_multi_gpu_european_qmc_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                                          int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
                                          int second_order, int n_gpus, double* out9, olmc_stats* evals /* [14] or NULL */);
int olmc_multi_gpu_european_qmc_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                                   int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
                                   int n_gpus, olmc_cv_moments* out);
/* Host microseconds of the calling thread's last multi-GPU call: out8 = {launch phase (launch job posted -> every rank's kernel
 * queued), collective queued, result fetched (contains the kernels' run time), other ranks drained, total, the latest launcher's start
 * after the post (wake latency; 0 in the serial form), the longest and the shortest single rank's own launch}. */
int olmc_multi_gpu_spans(double* out8);

/* Blocking fetch of n (1..33) doubles that work ALREADY QUEUED on hip_stream leaves at d_src -- the triple after the caller's RCCL
 * all-reduce in the one-process-per-GPU form: a one-wave kernel behind that work hands them over through the library's pinned
 * buffer and completion word (the hand-over of every blocking pricing), instead of hipMemcpyAsync + hipStreamSynchronize. */
int olmc_fetch_dev(const double* d_src, int32_t n, void* hip_stream, double* out_host);

/* Host-side finalisation shared by every path: fills price / std_error from
 * (sum, sumsq, n) with discount exp(-rT).  Pure function, no device needed. */
int olmc_combine_stats(const olmc_stats* parts, int32_t n_parts, double r, double T, olmc_stats* out);

/* ---- validation taps (what the parity tests compare with the checker; not the product path) ------------------------ */
/* Raw Philox4x32-10 words: out[(p*n_blocks + b)*4 + w], p < n_paths, b < n_blocks. */
int olmc_philox_words(uint64_t seed, int64_t path_offset, int64_t n_paths,
                      int32_t block0, int32_t n_blocks, uint32_t stream_tag, uint32_t* out_host);
/* The fp32 normal stream: out[p*n_steps + t]. */
int olmc_normals(uint64_t seed, int64_t path_offset, int64_t n_paths, int32_t n_steps,
                 float* out_host);
/* Measurement kernels (instruction-issue probes, phase stamps, the clock probe), the moment and exp2 taps and the fault-injection /
 * rehearsal seams are NOT in this library: they live in the instrumented build, include/olmc_probe.h -> libolmc_probe.so. */

/* ---- measurement ----------------------------------------------------------
 * When enabled, every path-kernel launch carries a pair of HIP events attached to the dispatch itself
 * (hipExtLaunchKernelGGL): they take the kernel's own begin / end timestamps on the stream it runs on, the
 * figures rocprofv3 reports.  (hipEventRecord brackets around a launch also time the marker packets on either
 * side: +7..10 us at these durations.)  Multi-launch entry points (olmc_american_lsm, olmc_european_multi with
 * more than 65535 contracts) are bracketed as a whole.  olmc_kernel_time returns the number of launches timed
 * and their total milliseconds since the last reset. */
int olmc_profile_enable(int on);
/* Tuning knob for A/B measurements (results never change, only the launch shape):
 *   OLMC_TUNE_GRID_CAP   max workgroups per launch, 0 = default (larger jobs grid-stride)
 *   OLMC_TUNE_QMC_BLOCK  Sobol kernels: 0 = by size (default): from 16 dimensions on a workgroup takes 64 points and each of its four waves
 *                        a quarter of the dimensions -- where the point offset is a multiple of 64 and there are 32 dimensions or more,
 *                        with the high Gray-code bits' direction numbers folded once per wave and dimension --; from 2^22 points on (2^21
 *                        below 128 dimensions, 2^20 below 64, 2^19 below 32) a thread takes eight consecutive points.  1 = always eight points per thread, 2 = always
 *                        split workgroups, -1 = always one point per thread.  Every shape returns the same terminal prices bit for bit
 *                        (one association of a point's normal sum)
 *   OLMC_TUNE_POLL       blocking calls: 0 = wait by polling the host-mapped flag the kernel raises behind its results
 *                        (default), -1 = hipStreamSynchronize
 *   OLMC_TUNE_SPLIT_TAIL European launches: 0 = the paths beyond a whole number of workgroups per compute unit go to split
 *                        workgroups (64 paths, each wave a quarter of the steps; default), -1 = never (one shape throughout)
 *   OLMC_TUNE_SPLIT_SAT  k in [1, 16]: when the whole workgroups per compute unit leave a last round (of `occupancy` resident
 *                        workgroups) with fewer than k of them, that round is handed to the split workgroups too; 0 = never
 *                        (default: measured at 1M x 252, no gain at any k)
 *   OLMC_TUNE_STAGED_COPY results of 32 MB and more (path matrices, large terminal arrays): 0 = they leave the device in 16 MB chunks by DMA
 *                        into two pinned staging buffers while up to 8 host threads copy the previous chunk into the caller's buffer
 *                        (default), -1 = one hipMemcpyAsync into the caller's pageable buffer (round 4's form).  Same bytes either way
 *   OLMC_TUNE_MULTI_LAUNCH multi-GPU entry points: 0 = one launcher thread per device queues the ranks' kernels in parallel
 *                        (default), -1 = the calling thread queues them one after the other (round 4's form)
 */
enum { OLMC_TUNE_GRID_CAP = 2, OLMC_TUNE_QMC_BLOCK = 4, OLMC_TUNE_SPLIT_TAIL = 7, OLMC_TUNE_POLL = 8, OLMC_TUNE_SPLIT_SAT = 9,
       OLMC_TUNE_MULTI_LAUNCH = 10, OLMC_TUNE_STAGED_COPY = 11 };
int olmc_tune(int knob, int value);
/* The behavioural knobs can also be switched off from the environment, read once by the first olmc_init:
 * OLMC_POLL=0 (as OLMC_TUNE_POLL = -1), OLMC_SPLIT_TAIL=0 (as OLMC_TUNE_SPLIT_TAIL = -1), OLMC_MULTI_LAUNCH=serial (as
 * OLMC_TUNE_MULTI_LAUNCH = -1), OLMC_STAGED_COPY=0 (as OLMC_TUNE_STAGED_COPY = -1).  OLMC_TRACE_COPY=1 prints the phases of every
 * staged copy to stderr. */
int olmc_profile_reset(void);
int olmc_kernel_time(int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* OLMC_H */
