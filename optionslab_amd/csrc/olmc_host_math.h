// olmc_host_math.h -- the pure-host arithmetic of libolmc.so: everything between the C ABI's scalars and a kernel's argument
// structs, and between a kernel's raw sums and the numbers the ABI returns.  No HIP, no device, no global state: plain C++17 that
// olmc.hip includes for the product and tests/test_host_math_sanitizers.py compiles on its own with
// `g++ -fsanitize=address,undefined` (tests/host_math_harness.cpp) -- the 8 / 14-contract layouts, the finite-difference formulas
// and the moment combiners are the code that feeds every fused and every multi-GPU call, and none of it needs a GPU to be checked.
//
// Reference arithmetic restated here (paths relative to the reference repository root):
//   make_contract        src/simulation/gbm_numpy.py:35-39
//   finish_stats         src/pricing_models/monte_carlo.py:140-150   (ddof = 0)
//   cv_finish            src/pricing_models/monte_carlo.py:175-184   (np.cov: ddof = 1)
//   GreeksSet            src/greeks/unified_greeks.py:274-277, 295-358 (bumps, call order, differences)
//   *_greeks_layout      src/pricing_models/exotic_options.py:54-56    (dt, drift, vol per step of the exotic classes)
#ifndef OLMC_HOST_MATH_H
#define OLMC_HOST_MATH_H

#include "olmc.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace olmc {

// ------------------------------------------------------------- contracts ----
// Host-precomputed per-contract constants, in the reference's own arithmetic
// order (gbm_numpy.py:35-39): a = ln S + (r - q - sigma^2/2) dt * M, vol = sigma sqrt(dt).
struct Contract {
    double a;        // log_S0 + total_drift
    double vol;      // sigma * sqrt(dt)
    double strike;
    double sign;     // +1 call, -1 put : payoff = max(sign * (S_T - K), 0)
    double scale;    // a BASE contract (ContractSet::base_mask) evaluates its own exp(a +- vol z) and carries scale = 1.  Any other
                     // shares vol with the nearest base before it in its half of the set and S_T = scale * S_T(base), scale =
                     // exp(a - a_base): the S- and r-bumped contracts of a Greeks batch cost a multiply instead of two fp64 exps
    double neg_sign_strike;   // -sign * K: payoff = max(fma(sign, S_T, -sign K), 0) -- the same bits as sign * (S_T - K) for sign = +-1
                              // (one rounding of +-(S_T - K) either way), one instruction fewer per sample
    double sign_scale;        // sign * scale (exact): the fused-Greeks epilogue forms max(fma(sign_scale, S_T(base), -sign K), 0) -- for a
                              // base (scale 1) the very same bits as above, for a scaled contract one rounding fewer and one multiply fewer
};

template <int NSETS>
struct ContractSet {
    Contract c[NSETS];
    uint32_t base_mask;   // bit s set <=> contract s is a base: it evaluates its own exponentials.  An integer test on the scalar
                          // unit (s_bitcmp) where round 2 compared Contract::scale with 0.0 on the vector unit, once per contract
    uint32_t upper_continues_slot0;   // != 0: slot NSETS/2 is not a base -- it and the non-base slots behind it belong to the group of
                                      // slot 0 (the group straddles the middle: first-order Greeks' {mid, S+, S-, r+, r-}); see european_payoffs_folded
};

constexpr double kZScale = 1.1774100225154747;     // sqrt(2 ln 2): raw Box-Muller normals are in units of it (olmc_kernels.h)
constexpr int kExp2Entries = 256;                  // entries of the exp2 table = workgroup size (olmc_kernels.h: exp2_f64_tab)

// Fused exotic Greeks: the 8 / 14 contracts are at most six path recursions (see the kernels in olmc_kernels.h).
constexpr int kAsianGroups = 6;
constexpr int kAsianRealGroups = 4;                     // arithmetic kernel: slots 0..3 are recursions of their own, 4..5 ride on slot 0

struct AsianGreeksSet {
    double drift[kAsianGroups], vol[kAsianGroups];      // per step, in the exponential's units: AsianContract's drift x kUnit, vol x kZScale x kUnit
                                                        // (host side: the two products asian_exp64_kernel forms; unused groups repeat group 0).
                                                        // Geometric: unit 1 (drift, vol x kZScale), as asian_kernel<., true>
    double s0[16];                                      // spot of contract s (0 for an unused slot)
    double log_s0[16];                                  // geometric: ln of it
    double strike, sign, inv_steps;
    double rate_step[kAsianGroups - kAsianRealGroups];  // arithmetic kernel: slot 4 + d is slot 0 with the per-step drift moved by this much
                                                        // (natural units; 0 for an unused slot)
    int32_t group[16];                                  // recursion of contract s
};

enum ExtremaPayoff : int {
    kBarrierUpOut = 0, kBarrierUpIn = 1, kBarrierDownOut = 2, kBarrierDownIn = 3,
    kLookbackFloating = 4, kLookbackFixed = 5
};

struct ExtremaGreeksSet {
    double drift[kAsianGroups], vol[kAsianGroups];      // per step; vol already x kZScale (host: extrema_kernel's product)
    double s0[16], log_barrier_rel[16];                 // contract s: spot, ln(B / S_s) (0 for lookbacks and unused slots)
    double strike, sign;
    int32_t group[16];
    int32_t payoff, pad;
};

static_assert(OLMC_MAX_BATCH == 16, "the fused Greeks sets carry 16 contract slots");

// ------------------------------------------------------------ small pieces ----
inline olmc_option make_option(double S, double K, double T, double r, double sigma, double q, int is_call) {
    olmc_option o;
    o.S = S; o.K = K; o.T = T; o.r = r; o.sigma = sigma; o.q = q;
    o.is_call = is_call ? 1 : 0;
    o.reserved = 0;
    return o;
}

// gbm_numpy.py:35-39 (multi-step) and :73-75 (single-step; identical when M == 1
// up to the exact product drift*1).
inline Contract make_contract(const olmc_option& o, int32_t n_steps) {
    const double dt = o.T / n_steps;
    const double drift = (o.r - o.q - 0.5 * o.sigma * o.sigma) * dt;
    const double vol = o.sigma * std::sqrt(dt);
    const double total_drift = drift * n_steps;
    Contract c;
    c.a = std::log(o.S) + total_drift;
    c.vol = vol;
    c.strike = o.K;
    c.sign = o.is_call ? 1.0 : -1.0;
    c.scale = 1.0;                                   // a base until group_contracts says otherwise
    c.neg_sign_strike = -c.sign * o.K;
    c.sign_scale = c.sign;
    return c;
}

inline bool same_bits(const double& a, const double& b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// Orders k contracts so that those with bit-identical vol are contiguous, the first of each group being its
// base (its bit in base_mask, scale 1) and the others carrying scale = exp(a - a_base); fills `set` (padded to nsets with
// scale-1 copies of the last contract) and pos[i] = slot of contract i.  The kernel walks the two halves of the set as two
// streams, each with its own "latest base" (european_payoffs_folded), so slot NSETS / 2 is ALWAYS a base: a group that
// straddles the middle gets a second base there (one more pair of exponentials per path; the first-order Greeks set
// {mid, S+, S-, r+, r-} + 3 pays it, the second-order set of 14 does not).
// Precondition (checked by every caller): 1 <= k <= NSETS <= OLMC_MAX_BATCH; pos has room for k entries.
template <int NSETS>
inline void group_contracts(const olmc_option* opts, int32_t k, int32_t n_steps, ContractSet<NSETS>* set, int* pos) {
    static_assert(NSETS >= 1 && NSETS <= OLMC_MAX_BATCH, "set size");
    Contract all[OLMC_MAX_BATCH];
    bool placed[OLMC_MAX_BATCH] = {};
    for (int i = 0; i < k; ++i) all[i] = make_contract(opts[i], n_steps);
    int slot = 0;
    set->base_mask = 0;
    set->upper_continues_slot0 = 0;
    auto put = [&](int j, int base_slot) -> int {   // returns the slot of the base the NEXT member of the group should refer to
        set->c[slot] = all[j];
        // the second stream (slots >= NSETS / 2) starts without a base of its own -- unless the group that straddles the middle is slot
        // 0's, whose prices the kernel hands over to it (one pair of exponentials saved for first-order Greeks)
        const bool needs_own_base = slot == NSETS / 2 && base_slot != 0;
        if (slot == NSETS / 2 && base_slot == 0) set->upper_continues_slot0 = 1;
        if (base_slot < 0 || needs_own_base) {
            set->c[slot].scale = 1.0;
            set->base_mask |= 1u << slot;
            base_slot = slot;
        } else {
            set->c[slot].scale = std::exp(all[j].a - set->c[base_slot].a);
        }
        set->c[slot].sign_scale = set->c[slot].sign * set->c[slot].scale;
        pos[j] = slot++;
        placed[j] = true;
        return base_slot;
    };
    for (int i = 0; i < k; ++i) {
        if (placed[i]) continue;
        int base = put(i, -1);                       // base of a new group
        for (int j = i + 1; j < k; ++j)
            if (!placed[j] && same_bits(all[j].vol, all[i].vol) && std::isfinite(all[j].a - all[i].a))
                base = put(j, base);
    }
    for (; slot < NSETS; ++slot) {                   // padding: cheap non-base copies (a base of its own if it opens the second half)
        set->c[slot] = set->c[slot - 1];
        set->c[slot].scale = 1.0;
        set->c[slot].sign_scale = set->c[slot].sign;
        if (slot == NSETS / 2) set->base_mask |= 1u << slot;
    }
}

inline void finish_stats(double sum, double sumsq, int64_t n, double r, double T, olmc_stats* out) {
    const double disc = std::exp(-r * T);
    const double mean = sum / static_cast<double>(n);
    double var = sumsq / static_cast<double>(n) - mean * mean;   // ddof = 0, monte_carlo.py:149
    if (var < 0.0) var = 0.0;
    out->sum = sum;
    out->sumsq = sumsq;
    out->n = n;
    out->price = disc * mean;
    out->std_error = disc * std::sqrt(var) / std::sqrt(static_cast<double>(n));
}

// The reference validates nothing at call time (tests/test_monte_carlo.py:143-151 skip it): a negative
// spot or a NaN input makes np.log / the arithmetic produce NaN, np.maximum PROPAGATES it, and the price
// is NaN.  Device fmax() would swallow the NaN (payoff 0), so such inputs are answered on the host.
// T < 0 is the same case one step later: sqrt(dt) is NaN in the reference (gbm_numpy.py:37) and so is every price.
inline bool poisoned(double S, double K, double T, double r, double sigma, double q) {
    return std::isnan(S + K + T + r + sigma + q) || S < 0.0 || T < 0.0;
}

// ln of a barrier LEVEL the reference compares in price space (exotic_options.py:455-480): a level <= 0 lies below
// every price, so `S_t >= level` always holds and `S_t < level` never does -- which is what -inf gives in log space
// (log() itself would answer NaN for a negative level and every comparison would be false).
inline double log_level(double level) {
    return level > 0.0 ? std::log(level) : (std::isnan(level) ? level : -INFINITY);
}

inline void nan_stats(int64_t n, olmc_stats* out) {
    const double nan = std::nan("");
    out->sum = nan; out->sumsq = nan; out->n = n; out->price = nan; out->std_error = nan;
}

// Shard sums added in rank order (fixed order => bitwise stable whatever the arrival order).  false: no samples.
inline bool combine_stats(const olmc_stats* parts, int32_t n_parts, double r, double T, olmc_stats* out) {
    double sum = 0.0, sumsq = 0.0;
    int64_t n = 0;
    for (int i = 0; i < n_parts; ++i) {
        sum += parts[i].sum;
        sumsq += parts[i].sumsq;
        n += parts[i].n;
    }
    if (n < 1) return false;
    finish_stats(sum, sumsq, n, r, T, out);
    return true;
}

// ----------------------------------------------------------- control variate ----
// beta, forward and the estimate from the five (already discounted) moments: monte_carlo.py:175-184
inline void cv_finish(double S, double T, double r, double q, olmc_cv_moments* m) {
    const double n = static_cast<double>(m->n);
    const double mean_d = m->sum_d / n, mean_s = m->sum_s / n;
    // np.cov default ddof = 1 (monte_carlo.py:181); the n-1 cancels in beta but not in the 1e-10 guard
    const double cov_ds = (m->sum_ds - n * mean_d * mean_s) / (n - 1.0);
    const double var_s = (m->sum_ss - n * mean_s * mean_s) / (n - 1.0);
    const double beta = (n > 1.0 && var_s > 1e-10) ? cov_ds / var_s : 0.0;        // :182
    const double forward = S * std::exp((r - q) * T);                              // :179
    m->value = mean_d - beta * (mean_s - forward);                                 // :184
}

// Device moments are of the UNdiscounted payoff x; the reference's d = disc * x (monte_carlo.py:175).
inline void cv_from_device(const double* raw5, int64_t n, double S, double T, double r, double q, olmc_cv_moments* out) {
    const double disc = std::exp(-r * T);
    out->sum_d = disc * raw5[0];
    out->sum_s = raw5[1];
    out->sum_dd = disc * disc * raw5[2];
    out->sum_ss = raw5[3];
    out->sum_ds = disc * raw5[4];
    out->n = n;
    cv_finish(S, T, r, q, out);
}

inline bool combine_cv(const olmc_cv_moments* parts, int32_t n_parts, double S, double T, double r, double q, olmc_cv_moments* out) {
    olmc_cv_moments m{};
    for (int i = 0; i < n_parts; ++i) {   // fixed rank order => bitwise stable
        m.sum_d += parts[i].sum_d;  m.sum_s += parts[i].sum_s;  m.sum_dd += parts[i].sum_dd;
        m.sum_ss += parts[i].sum_ss;  m.sum_ds += parts[i].sum_ds;  m.n += parts[i].n;
    }
    if (m.n < 1) return false;
    cv_finish(S, T, r, q, &m);
    *out = m;
    return true;
}

// ------------------------------------------------- finite-difference Greeks ----
// The evaluations of compute_greeks_unified (unified_greeks.py:274-277, 295-358) in the reference's get_price() call order, and
// the finite differences over their prices.  Shared by every fused Greeks entry point: only the pricing of the set differs.
// k is 7 or 8 (first order, without / with the T bump) or 11 / 14 (second order): never more than OLMC_MAX_BATCH = 16 = the length
// of o[], and finish() zero-fills evals[k .. 14): a caller's `evals` buffer holds 14 olmc_stats (olmc.h).
struct GreeksSet {
    static constexpr int kEvalSlots = 14;
    olmc_option o[OLMC_MAX_BATCH];
    int k = 0;
    double h_S, h_v, h_r, h_T;
    bool has_T, second;
    int i_mid, i_su, i_sd, i_vu, i_vd, i_td, i_ru, i_rd, i_uu = -1, i_ud = -1, i_du = -1, i_dd = -1, i_ut = -1, i_dt = -1;

    GreeksSet(double S, double K, double T, double r, double sigma, double q, int is_call, int second_order) {
        h_S = std::max(1e-4, 0.01 * S);                             // :274-277
        h_v = std::max(1e-4, 0.01);
        h_r = 1e-4;
        h_T = 1 / 365.0;
        has_T = T > h_T;                                            // :310
        second = second_order != 0;
        auto add = [&](double S_, double T_, double r_, double v_) { o[k] = make_option(S_, K, T_, r_, v_, q, is_call); return k++; };
        i_mid = add(S, T, r, sigma);
        i_su = add(S + h_S, T, r, sigma); i_sd = add(S - h_S, T, r, sigma);
        i_vu = add(S, T, r, sigma + h_v); i_vd = add(S, T, r, sigma - h_v);
        i_td = has_T ? add(S, T - h_T, r, sigma) : -1;
        i_ru = add(S, T, r + h_r, sigma); i_rd = add(S, T, r - h_r, sigma);
        if (second) {
            i_uu = add(S + h_S, T, r, sigma + h_v); i_ud = add(S + h_S, T, r, sigma - h_v);
            i_du = add(S - h_S, T, r, sigma + h_v); i_dd = add(S - h_S, T, r, sigma - h_v);
            if (has_T) { i_ut = add(S + h_S, T - h_T, r, sigma); i_dt = add(S - h_S, T - h_T, r, sigma); }
        }
    }

    int nsets() const { return k <= 8 ? 8 : 16; }

    void finish(const olmc_stats* st, double T, double* out9, olmc_stats* evals) const {
        auto P = [&](int i) { return st[i].price; };
        const double mid = P(i_mid);
        const double delta = (P(i_su) - P(i_sd)) / (2 * h_S);                       // :301
        out9[0] = mid;
        out9[1] = delta;
        out9[2] = (P(i_su) - 2 * mid + P(i_sd)) / (h_S * h_S);                       // :302
        out9[3] = (P(i_vu) - P(i_vd)) / (2 * h_v);                                   // :307
        out9[4] = has_T ? (P(i_td) - mid) / h_T : -mid / std::max(T, 1e-6);          // :310-314
        out9[5] = (P(i_ru) - P(i_rd)) / (2 * h_r);                                   // :319
        if (second) {
            out9[6] = (P(i_uu) - P(i_ud) - P(i_du) + P(i_dd)) / (4 * h_S * h_v);    // :343-345
            out9[7] = has_T ? ((P(i_ut) - P(i_dt)) / (2 * h_S) - delta) / h_T : 0.0; // :348-354
            out9[8] = (P(i_vu) - 2 * mid + P(i_vd)) / (h_v * h_v);                   // :357
        }
        if (evals) {
            for (int i = 0; i < k; ++i) evals[i] = st[i];
            for (int i = k; i < kEvalSlots; ++i) std::memset(&evals[i], 0, sizeof(olmc_stats));
        }
    }

    // The stats of the k evaluations from one launch's {sum, sumsq} pairs: contract i's pair sits at slot pos[i] (pos == nullptr:
    // slot i), `stride` doubles apart (2; 1 and no sumsq for a sums-only launch).  Poisoned contracts answer NaN (see poisoned()).
    void stats_from_sums(const double* sums, const int* pos, int64_t n, bool sums_only, bool extra_poison, olmc_stats* st) const {
        for (int i = 0; i < k; ++i) {
            const int slot = pos ? pos[i] : i;
            if (extra_poison || poisoned(o[i].S, o[i].K, o[i].T, o[i].r, o[i].sigma, o[i].q)) nan_stats(n, &st[i]);
            else if (sums_only) finish_stats(sums[slot], std::nan(""), n, o[i].r, o[i].T, &st[i]);
            else finish_stats(sums[2 * slot], sums[2 * slot + 1], n, o[i].r, o[i].T, &st[i]);
        }
    }
};

// The layout olmc_european_batch / olmc_european_greeks_fd give a set of k contracts (what olmc_contract_layout reports).
inline void contract_layout(const olmc_option* opts, int32_t k, int32_t n_steps, int32_t* nsets_out, int32_t* pos, uint32_t* base_mask,
                            int32_t* upper_continues_slot0, double* scale16) {
    int p[OLMC_MAX_BATCH];
    for (int i = 0; i < OLMC_MAX_BATCH; ++i) scale16[i] = 0.0;
    if (k <= 8) {
        ContractSet<8> cs;
        group_contracts<8>(opts, k, n_steps, &cs, p);
        *nsets_out = 8; *base_mask = cs.base_mask; *upper_continues_slot0 = static_cast<int32_t>(cs.upper_continues_slot0);
        for (int i = 0; i < 8; ++i) scale16[i] = cs.c[i].scale;
    } else {
        ContractSet<16> cs;
        group_contracts<16>(opts, k, n_steps, &cs, p);
        *nsets_out = 16; *base_mask = cs.base_mask; *upper_continues_slot0 = static_cast<int32_t>(cs.upper_continues_slot0);
        for (int i = 0; i < 16; ++i) scale16[i] = cs.c[i].scale;
    }
    for (int i = 0; i < k; ++i) pos[i] = p[i];
}

// ---------------------------------------------------- fused exotic Greeks sets ----
// Barrier / lookback: a contract enters the step loop only through (drift, vol) per step (exotic_options.py:54-56): the 8 / 14
// contracts of a GreeksSet are at most kAsianGroups recursions.  Returns nullptr, or what is wrong.
inline const char* extrema_greeks_layout(const GreeksSet& gs, int32_t n_steps, int payoff, double barrier, double K, int is_call,
                                         ExtremaGreeksSet* es) {
    std::memset(es, 0, sizeof *es);
    const bool is_barrier = payoff <= kBarrierDownIn;
    int n_groups = 0;
    for (int i = 0; i < gs.k; ++i) {
        const olmc_option& o = gs.o[i];
        const double dt = o.T / n_steps;                             // exotic_options.py:54-56, as run_extrema
        const double drift = (o.r - o.q - 0.5 * o.sigma * o.sigma) * dt, vol = o.sigma * std::sqrt(dt) * kZScale;
        int g = 0;
        while (g < n_groups && !(same_bits(es->drift[g], drift) && same_bits(es->vol[g], vol))) ++g;
        if (g == n_groups) {
            if (n_groups == kAsianGroups) return "more distinct path recursions than the fused Greeks kernel carries";
            es->drift[g] = drift;
            es->vol[g] = vol;
            ++n_groups;
        }
        es->group[i] = g;
        es->s0[i] = o.S;
        es->log_barrier_rel[i] = is_barrier ? std::log(barrier / o.S) : 0.0;
    }
    for (int g = n_groups; g < kAsianGroups; ++g) { es->drift[g] = es->drift[0]; es->vol[g] = es->vol[0]; }
    es->strike = K;
    es->sign = is_call ? 1.0 : -1.0;
    es->payoff = payoff;
    return nullptr;
}

// Asian.  Recursions: contracts that share (drift, vol) per step share one (a spot bump only scales the average).  Geometric: up to
// six, all alike.  Arithmetic: slots 0..3 are recursions of their own; a contract whose vol is slot 0's and whose drift is not (the r
// bumps) rides on slot 0 through a per-date factor (slots 4..5, rate_step = its drift - slot 0's).  `unit` = what the kernel's
// exponential counts in: 1 for the geometric kernel, kExp2Entries log2(e) (table form) or log2(e) for the arithmetic one.
inline const char* asian_greeks_layout(const GreeksSet& gs, int32_t n_steps, bool geometric, double unit, double K, int is_call,
                                       AsianGreeksSet* as) {
    std::memset(as, 0, sizeof *as);
    const int real_slots = geometric ? kAsianGroups : kAsianRealGroups;
    int n_groups = 0, n_riders = 0;
    double rider_drift[kAsianGroups - kAsianRealGroups] = {0.0, 0.0};
    for (int i = 0; i < gs.k; ++i) {
        const olmc_option& o = gs.o[i];
        const double dt = o.T / n_steps;                             // exotic_options.py:54-56, as olmc_asian
        const double drift = (o.r - o.q - 0.5 * o.sigma * o.sigma) * dt, vol = o.sigma * std::sqrt(dt);
        as->s0[i] = o.S;
        as->log_s0[i] = std::log(o.S);
        if (!geometric && n_groups > 0 && same_bits(as->vol[0], vol) && !same_bits(as->drift[0], drift)) {
            int d = 0;
            while (d < n_riders && !same_bits(rider_drift[d], drift)) ++d;
            if (d == n_riders) {
                if (n_riders == kAsianGroups - kAsianRealGroups) return "more drift-only bumps than the fused Asian Greeks kernel carries";
                rider_drift[d] = drift;
                as->rate_step[d] = drift - as->drift[0];
                ++n_riders;
            }
            as->group[i] = kAsianRealGroups + d;
            continue;
        }
        int g = 0;
        while (g < n_groups && !(same_bits(as->drift[g], drift) && same_bits(as->vol[g], vol))) ++g;
        if (g == n_groups) {
            if (n_groups == real_slots) return "more distinct path recursions than the fused Asian Greeks kernel carries";
            as->drift[g] = drift;
            as->vol[g] = vol;
            ++n_groups;
        }
        as->group[i] = g;
    }
    for (int g = n_groups; g < kAsianGroups; ++g) { as->drift[g] = as->drift[0]; as->vol[g] = as->vol[0]; }
    // into the units the kernel sums in, by the very products the one-contract kernels form on the device: arithmetic
    // (asian_exp64_kernel) drift * unit, vol * kZScale * unit; geometric (asian_kernel<., true>) drift * 1.0, vol * kZScale * 1.0
    for (int g = 0; g < kAsianGroups; ++g) {
        as->drift[g] = as->drift[g] * unit;
        as->vol[g] = as->vol[g] * kZScale * unit;
    }
    as->strike = K;
    as->sign = is_call ? 1.0 : -1.0;
    as->inv_steps = 1.0 / n_steps;
    return nullptr;
}

// Contiguous global path ranges of a P-rank call: rank d owns [d N / P, (d + 1) N / P)  (SURVEY §8e).  N < 2^53 / P in practice
// (N <= 2^40 paths, P <= 16), so the products cannot overflow int64.
inline void shard_range(int64_t n_paths, int rank, int n_ranks, int64_t* lo, int64_t* count) {
    const int64_t a = n_paths * rank / n_ranks, b = n_paths * (rank + 1) / n_ranks;
    *lo = a;
    *count = b - a;
}

// Sobol POINTS of a P-rank call: the same tiling with every inner boundary rounded DOWN to a multiple of 512 points, so that every
// rank's point offset is one the aligned Sobol kernels take (olmc.hip qmc_shape: a wave's 64 points, or 64 blocks of eight, then
// share their high Gray bits) -- an unaligned offset costs a rank the 30-mask form, 1.5 x the time.  Only where a rank then still
// owns at least 4,096 points; below, the plain ranges.  Mirrored by optionslab_amd/sharding.py qmc_shard_bounds (the
// one-process-per-GPU form must cut the sequence at the same points).
constexpr int64_t kQmcShardAlign = 512, kQmcShardMinPoints = 4096;
inline void qmc_shard_range(int64_t n_points, int rank, int n_ranks, int64_t* lo, int64_t* count) {
    if (n_points / n_ranks < kQmcShardMinPoints) return shard_range(n_points, rank, n_ranks, lo, count);
    const int64_t a = n_points * rank / n_ranks / kQmcShardAlign * kQmcShardAlign;
    const int64_t b = rank + 1 == n_ranks ? n_points : n_points * (rank + 1) / n_ranks / kQmcShardAlign * kQmcShardAlign;
    *lo = a;
    *count = b - a;
}

}  // namespace olmc

#endif  // OLMC_HOST_MATH_H
