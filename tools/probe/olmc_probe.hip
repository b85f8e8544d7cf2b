// olmc_probe.hip -- libolmc_probe.so, the INSTRUMENTED build of the library (include/olmc_probe.h).
//
// It is the product's own translation unit (optionslab_amd/csrc/olmc.hip, textually included with OLMC_WITH_PROBES defined, so
// that the fault-injection and rehearsal seams inside it are compiled in) plus the measurement / validation kernels and their
// entry points.  libolmc.so is built WITHOUT any of this: `nm -D optionslab_amd/libolmc.so` lists no probe symbol, the product
// carries no fault-injection branch.  Loaded by tests, tools/ and bench.py's calibration only (tools/probe/binding.py); it has its
// own contexts, streams and workspaces, independent of a libolmc.so loaded beside it.
#define OLMC_WITH_PROBES 1
#include "../../optionslab_amd/csrc/olmc.hip"
#include "olmc_probe.h"
#include "olmc_probe_kernels.h"

// ============================================================ validation taps ====
extern "C" int olmc_exp2_probe_form(const double* x_host, int64_t n, double* y_host, int form) {
    if (!x_host || !y_host || n < 1) return fail(OLMC_ERR_ARG, "bad arguments");
    if (form != 0 && form != 1) return fail(OLMC_ERR_ARG, "form must be 0 (polynomial) or 1 (table)");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t bytes = sizeof(double) * static_cast<size_t>(n);
    rc = bulk_reserve(c, 2 * bytes);
    if (rc) return rc;
    double* d_x = static_cast<double*>(c->d_bulk);
    double* d_y = d_x + n;
    HIP_TRY(hipMemcpyAsync(d_x, x_host, bytes, hipMemcpyHostToDevice, c->stream));
    const int grid = static_cast<int>(std::min<int64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(exp2_probe_kernel, dim3(grid), dim3(256), 0, c->stream, d_x, n, d_y, form);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(y_host, d_y, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OLMC_OK;
}

// the form the arithmetic Asian kernel is built with
extern "C" int olmc_exp2_probe(const double* x_host, int64_t n, double* y_host) {
    return olmc_exp2_probe_form(x_host, n, y_host, OLMC_EXP2_TABLE ? 1 : 0);
}

extern "C" int olmc_ndtri_probe(const double* p_host, int64_t n, double* z_host, int form) {
    if (!p_host || !z_host || n < 1) return fail(OLMC_ERR_ARG, "bad arguments");
    if (form < 0 || form > 3) return fail(OLMC_ERR_ARG, "form must be 0 .. 3");
    for (int64_t i = 0; i < n; ++i)
        if (!(p_host[i] >= 1e-10 && p_host[i] <= 1.0 - 1e-10)) return fail(OLMC_ERR_ARG, "a probability outside [1e-10, 1 - 1e-10]");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t bytes = sizeof(double) * static_cast<size_t>(n);
    rc = bulk_reserve(c, 2 * bytes);
    if (rc) return rc;
    double* d_p = static_cast<double*>(c->d_bulk);
    double* d_z = d_p + n;
    HIP_TRY(hipMemcpyAsync(d_p, p_host, bytes, hipMemcpyHostToDevice, c->stream));
    const int grid = static_cast<int>(std::min<int64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(ndtri_probe_kernel, dim3(grid), dim3(256), 0, c->stream, d_p, n, d_z, form);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(z_host, d_z, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OLMC_OK;
}

extern "C" int olmc_normal_moments(uint64_t seed, int64_t path_offset, int64_t n_paths, int32_t n_steps, double* out4) {
    if (!out4) return fail(OLMC_ERR_ARG, "null pointer");
    olmc_stats dummy;
    return run_structured(path_offset, n_paths, n_steps, seed, 0, 0.0, 1.0, false, &dummy,
                          [&](int32_t grid, hipStream_t st, const EventPair* timed, const PathRange& pr, const ReduceWs& ws) {
                              launch_timed(normal_moments_kernel, dim3(grid), dim3(kBlock), st, timed, pr, ws);
                          }, 4, out4);
}


// Shader clock held under the headline kernel's load (see clock_probe_kernel): median over workgroups.
extern "C" int olmc_clock_probe(int64_t n_paths, int32_t n_steps, uint64_t seed, double* out3) {
    if (!out3) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const int32_t grid = static_cast<int32_t>(std::min<int64_t>((n_paths + kBlock - 1) / kBlock, kMaxGrid));
    const size_t bytes = sizeof(uint64_t) * 2 * static_cast<size_t>(grid) + 256;
    rc = bulk_reserve(c, bytes);
    if (rc) return rc;
    uint64_t* d_stamps = static_cast<uint64_t*>(c->d_bulk);
    double* d_sink = reinterpret_cast<double*>(d_stamps + 2 * static_cast<size_t>(grid));
    const PathRange pr = make_range(0, static_cast<int64_t>(grid) * kBlock, n_steps, seed);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(grid), dim3(kBlock), 0, c->stream, pr, d_stamps, d_sink);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> h(2 * static_cast<size_t>(grid));
    HIP_TRY(hipMemcpyAsync(h.data(), d_stamps, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<double> cyc(grid), tick(grid), ghz;
    for (int32_t b = 0; b < grid; ++b) {
        cyc[b] = static_cast<double>(h[2 * b]);
        tick[b] = static_cast<double>(h[2 * b + 1]);
        if (tick[b] > 0) ghz.push_back(cyc[b] / tick[b] * 0.1);     // cycles per 10 ns tick -> GHz
    }
    auto median = [](std::vector<double>& v) {
        if (v.empty()) return 0.0;
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        return v[v.size() / 2];
    };
    out3[0] = median(cyc);
    out3[1] = median(tick);
    out3[2] = median(ghz);
    return OLMC_OK;
}

// Where a launch of the headline kernel spends its time (see european_stamp_kernel): one blocking launch of n_paths x n_steps in
// the production launch shape; stamps_host receives 5 words per workgroup (4 stamps in 100 MHz ticks + where it ran) + the final stamp, info3 = {workgroups,
// split_from (or workgroups when nothing is split), the dispatch's own duration in nanoseconds (begin / end timestamps)}.
extern "C" int olmc_phase_stamps(int64_t n_paths, int32_t n_steps, uint64_t seed, int32_t lead_launches, uint64_t* stamps_host, int64_t capacity,
                                 int64_t* info3) {
    if (!stamps_host || !info3) return fail(OLMC_ERR_ARG, "null pointer");
    if (lead_launches < 0 || lead_launches > 1000) return fail(OLMC_ERR_ARG, "lead_launches must be in [0, 1000]");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    PathRange pr = make_range(0, n_paths, n_steps, seed);
    const int32_t grid = european_launch_shape(c, &pr, european_occupancy<1, kReduce>(true));
    if (static_cast<int64_t>(grid) * kBlock < n_paths) return fail(OLMC_ERR_ARG, "grid-striding launches are not instrumented");
    const int64_t words = kStampWords * static_cast<int64_t>(grid) + 1;
    if (capacity < words) return fail(OLMC_ERR_ARG, "stamp buffer too small: need 5 * workgroups + 1 words");
    rc = bulk_reserve(c, sizeof(uint64_t) * static_cast<size_t>(words));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_bulk, 0, sizeof(uint64_t) * static_cast<size_t>(words), c->stream));
    ContractSet<1> cs;
    cs.c[0] = make_contract(make_option(100.0, 100.0, 1.0, 0.05, 0.2, 0.0, 1), n_steps);
    cs.base_mask = 1u; cs.upper_continues_slot0 = 0;
    EventPair ep{};
    rc = prof_acquire(c, &ep);
    if (rc) return rc;
    struct GiveBack {                                // the pair goes back to the free list however the call leaves
        DeviceCtx* c; EventPair ep;
        ~GiveBack() { c->ev_free.push_back(ep); }
    } give_back{c, ep};
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 2, c->d_result, -1.0, &ws);
    if (rc) return rc;
    // `lead_launches` identical launches go out back to back in front of the recorded one (each overwrites the stamps of the one
    // before; only the last is armed to raise the completion word): the recorded launch then runs on a device that is already under
    // this very load, at the clock it holds there -- a lone launch between a memset and a copy ran 16 % slow
    ReduceWs quiet = ws;
    quiet.done_flag = nullptr;
    for (int32_t k = 0; k < lead_launches; ++k) {
        hipLaunchKernelGGL(european_stamp_kernel, dim3(grid), dim3(kBlock), 0, c->stream, pr, cs, quiet, static_cast<uint64_t*>(c->d_bulk));
        rc = after_launch(c, c->stream);             // a failed launch leaves the self-resetting counters to ws_recover()
        if (rc) return rc;
    }
    hipExtLaunchKernelGGL(european_stamp_kernel, dim3(grid), dim3(kBlock), 0, c->stream, ep.start, ep.stop, 0, pr, cs, ws, static_cast<uint64_t*>(c->d_bulk));
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stamps_host, c->d_bulk, sizeof(uint64_t) * static_cast<size_t>(words), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.start, ep.stop));
    info3[0] = grid;
    info3[1] = pr.split_from == INT32_MAX ? grid : pr.split_from;
    info3[2] = static_cast<int64_t>(static_cast<double>(ms) * 1e6);
    return OLMC_OK;
}

// Issue cost of one instruction class on this device (see the probe kernels): nanoseconds one SIMD needs per wave64
// instruction of the class with `waves_per_simd` waves resident.
extern "C" int olmc_issue_probe(int op, int waves_per_simd, double* ns_per_instr) {
    using Probe = void (*)(uint32_t*, uint32_t, uint32_t);
    static const Probe table[] = {probe_mad_u64_u32, probe_bitop3, probe_cvt_f32_u32, probe_fmamk_f32, probe_and_or, probe_log_f32,
                                  probe_sqrt_f32, probe_sin_f32, probe_cos_f32, probe_exp_f32, probe_add_f32, probe_fma_f32,
                                  probe_cvt_f64_f32, probe_add_f64, probe_fma_f64, probe_rndne_f64, probe_ldexp_f64, probe_cvt_i32_f64,
                                  probe_mix_log_add, probe_mix_log_bitop3, probe_bitop3_vvv, probe_bitop3_vvc, probe_xor_vv, probe_mix_bitop3_add,
                                  probe_mix_mad_bitop3, probe_mad_u64_u32_vv};
    constexpr int kOps = static_cast<int>(sizeof(table) / sizeof(table[0]));
    static_assert(kOps == OLMC_PROBE_COUNT, "include/olmc_probe.h lists the probe classes");
    if (!ns_per_instr) return fail(OLMC_ERR_ARG, "null pointer");
    if (op < 0 || op >= kOps) return fail(OLMC_ERR_ARG, "unknown probe class");
    if (waves_per_simd < 1 || waves_per_simd > 8) return fail(OLMC_ERR_ARG, "waves_per_simd must be in [1, 8]");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    rc = bulk_reserve(c, 256);
    if (rc) return rc;
    const dim3 grid(static_cast<uint32_t>(c->cus * waves_per_simd)), block(kBlock);    // one 4-wave workgroup per (CU, resident wave slot)
    EventPair ep{};
    rc = prof_acquire(c, &ep);
    if (rc) return rc;
    for (int rep = 0; rep < 2; ++rep) {          // first launch warms the instruction cache
        hipExtLaunchKernelGGL(table[op], grid, block, 0, c->stream, ep.start, ep.stop, 0, static_cast<uint32_t*>(c->d_bulk), 1u, 0xD2511F53u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.start, ep.stop));
    c->ev_free.push_back(ep);
    *ns_per_instr = static_cast<double>(ms) * 1e6 / (static_cast<double>(kProbeIters) * 16.0 * waves_per_simd);
    return OLMC_OK;
}



// The European call with fp64 normals (european_f64_normals_kernel): one launch of n_paths x n_steps, one workgroup per 256 paths
// (n_paths <= 2^18 x 256), antithetic.  Timed like every pricing when olmc_profile_enable is on.
extern "C" int olmc_european_f64_normals(double S, double K, double T, double r, double sigma, double q, int is_call, int64_t n_paths,
                                         int32_t n_steps, uint64_t seed, olmc_stats* out) {
    if (n_paths > static_cast<int64_t>(kMaxGrid) * kBlock) return fail(OLMC_ERR_ARG, "n_paths too large for one workgroup per 256 paths");
    const olmc_option o = make_option(S, K, T, r, sigma, q, is_call);
    ContractSet<1> cs;
    cs.c[0] = make_contract(o, n_steps);
    cs.base_mask = 1u; cs.upper_continues_slot0 = 0;
    const int saved_cap = g_grid_cap;
    g_grid_cap = kMaxGrid;                               // one workgroup per 256 paths whatever the step count (no short-path grid cap)
    const int rc = run_structured(0, n_paths, n_steps, seed, 1, r, T, poisoned(S, K, T, r, sigma, q), out,
                                  [&](int32_t grid, hipStream_t st, const EventPair* timed, const PathRange& pr, const ReduceWs& ws) {
                                      launch_timed(european_f64_normals_kernel, dim3(grid), dim3(kBlock), st, timed, pr, cs, ws);
                                  });
    g_grid_cap = saved_cap;
    return rc;
}

// Microseconds one more DEPENDENT launch costs on a stream: wall of a chain of `n` empty kernels, minus nothing, over n.
extern "C" int olmc_launch_gap_probe(int32_t n, double* us_per_launch) {
    if (!us_per_launch || n < 2 || n > 100000) return fail(OLMC_ERR_ARG, "bad arguments");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    for (int rep = 0; rep < 2; ++rep) {                  // the first chain warms the code object
        HIP_TRY(hipStreamSynchronize(c->stream));
        const auto t0 = std::chrono::steady_clock::now();
        for (int32_t k = 0; k < n; ++k) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(kWave), 0, c->stream, reinterpret_cast<uint32_t*>(c->d_result));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        *us_per_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    }
    return OLMC_OK;
}

// Test seams of the instrumented build (none of them exists in libolmc.so).
extern "C" int olmc_probe_tune(int knob, int value) {
    if (knob == OLMC_PROBE_TUNE_FAULT_SHARD && value >= 0 && value <= kMaxDevices) { g_fault_shard = value; return OLMC_OK; }
    if (knob == OLMC_PROBE_TUNE_FORCE_NV && value >= 0 && value <= kMaxNV) { g_force_nv = value; return OLMC_OK; }
    if (knob == OLMC_PROBE_TUNE_MULTI_REHEARSAL && value >= 0 && value <= 1) { g_multi_rehearsal = value; return OLMC_OK; }
    return fail(OLMC_ERR_ARG, "unknown probe knob or value");
}
