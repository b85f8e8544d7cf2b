/* olmc_probe.h -- the INSTRUMENTED build of the library: libolmc_probe.so (tools/probe/olmc_probe.hip).
 *
 * Test and measurement infrastructure, not the product.  libolmc_probe.so is the product's own translation unit compiled with its
 * test seams switched in, plus the kernels and entry points below; it also exports everything include/olmc.h declares (so a test
 * can price through it), with contexts of its own.  libolmc.so exports NONE of the symbols declared here and contains no
 * fault-injection branch.  Same conventions as olmc.h: int status, 0 = ok, olmc_last_error() of the SAME library for the text.
 * Who loads it: tests/ (device-guard, failing-shard, multi-rank rehearsal, exp2 and moment taps), tools/ (phase stamps, issue
 * probes), bench.py (clock and issue-cost calibration of the roofline) -- through tools/probe/binding.py. */
#ifndef OLMC_PROBE_H
#define OLMC_PROBE_H
#include "olmc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- validation taps ------------------------------------------------------ */
/* The device's fp64 base-2 exponential, the per-date exponential of OLMC_AVG_ARITHMETIC (asian_exp64_kernel): y[i] = 2^x[i], host
 * arrays.  olmc_exp2_probe evaluates the form the Asian kernel is built with. */
int olmc_exp2_probe(const double* x_host, int64_t n, double* y_host);
/* Either form the sources carry: form 0 = rint + degree-11 polynomial (round 2; <= 1.5 ulp), form 1 = 256-entry table of 2^(k/256)
 * in LDS + degree-4 correction (1 + r q(r), q cubic; <= 1.1 ulp measured) -- the one OLMC_AVG_ARITHMETIC uses. */
int olmc_exp2_probe_form(const double* x_host, int64_t n, double* y_host, int form);
/* The Sobol kernels' inverse normal (ndtri_w and its register / lockstep forms, olmc_kernels.h) on a host array of probabilities
 * in [1e-10, 1 - 1e-10] (the clip of gbm_qmc.py:36): z[i] = Phi^-1(p[i]).  form 0 = one point, 1 = coefficients in registers,
 * 2 = two in lockstep, 3 = eight in lockstep: the four must agree bit for bit (the Sobol kernels' sums do not depend on the launch
 * shape because of it). */
int olmc_ndtri_probe(const double* p_host, int64_t n, double* z_host, int form);
/* Power sums of the normal stream: out4[m-1] = sum over paths and steps of z^m, m = 1..4 (fp64). */
int olmc_normal_moments(uint64_t seed, int64_t path_offset, int64_t n_paths, int32_t n_steps, double* out4);

/* ---- where a launch spends its time --------------------------------------- */
/* One launch of the headline kernel (European call, antithetic, n_paths x n_steps, production launch shape): wave 0 of every
 * workgroup stamps the device-wide 100 MHz counter (s_memrealtime) at entry, after the step loop, after the workgroup sums and on
 * return from the grid reduction, and notes where it ran (HW_ID | XCC_ID << 32); the wave that writes the totals stamps once
 * more.  stamps_host (caller-owned, `capacity` words >= 5 * workgroups + 1) receives [workgroup][5] then the final stamp; info3 =
 * {workgroups, index of the first split workgroup (= workgroups when none), duration of the dispatch in ns by its own begin / end
 * timestamps}.  `lead_launches` (0..1000) identical launches are queued back to back in front of the recorded one, so that it runs
 * at the clock the device holds under this load. */
int olmc_phase_stamps(int64_t n_paths, int32_t n_steps, uint64_t seed, int32_t lead_launches, uint64_t* stamps_host, int64_t capacity,
                      int64_t* info3);

/* Shader clock the device holds while every SIMD runs the headline kernel's step loop (n_paths x n_steps, one workgroup per 256
 * paths): out3 = {median shader cycles of a workgroup's loop (s_memtime), median 100 MHz ticks of the same interval
 * (s_memrealtime), median of their quotient in GHz}.  Feeds bench.py's roofline. */
int olmc_clock_probe(int64_t n_paths, int32_t n_steps, uint64_t seed, double* out3);

/* Issue cost of one VALU instruction class on this device: *ns_per_instr = nanoseconds one SIMD needs per wave64 instruction of
 * class `op` with waves_per_simd (1..8) waves resident, measured by a kernel of independent instructions of that class in the
 * operand form the path kernels use.  Calibrates bench.py's issue-time roofline live. */
enum { OLMC_PROBE_MAD_U64_U32 = 0, OLMC_PROBE_BITOP3_B32, OLMC_PROBE_CVT_F32_U32, OLMC_PROBE_FMAMK_F32, OLMC_PROBE_AND_OR_B32,
       OLMC_PROBE_LOG_F32, OLMC_PROBE_SQRT_F32, OLMC_PROBE_SIN_F32, OLMC_PROBE_COS_F32, OLMC_PROBE_EXP_F32, OLMC_PROBE_ADD_F32,
       OLMC_PROBE_FMA_F32, OLMC_PROBE_CVT_F64_F32, OLMC_PROBE_ADD_F64, OLMC_PROBE_FMA_F64, OLMC_PROBE_RNDNE_F64,
       OLMC_PROBE_LDEXP_F64, OLMC_PROBE_CVT_I32_F64,
       /* two-instruction bodies (the figure is per PAIR) and operand-form variants: do classes overlap in a mix? */
       OLMC_PROBE_MIX_LOG_ADD, OLMC_PROBE_MIX_LOG_BITOP3, OLMC_PROBE_BITOP3_VVV, OLMC_PROBE_BITOP3_VVC, OLMC_PROBE_XOR_VV,
       OLMC_PROBE_MIX_BITOP3_ADD, OLMC_PROBE_MIX_MAD_BITOP3, OLMC_PROBE_MAD_U64_U32_VV, OLMC_PROBE_COUNT };
int olmc_issue_probe(int op, int waves_per_simd, double* ns_per_instr);

/* ---- what the reference's own width would cost -------------------------------
 * European call with fp64 NORMALS (the reference draws fp64 normals, gbm_numpy.py:32-33): same Philox counter stream, but a block
 * yields two normals from two 53-bit uniforms by a library-precision fp64 Box-Muller (log, sincospi), the path's sum of normals in
 * fp64 throughout; launch shape, payoff and fused reduction of the product's european_path_kernel.  Antithetic.  Not a product
 * path: bench.py prints its time and |price - product price| / se beside the product's "f32 normals" line (key c2_f64_normals). */
int olmc_european_f64_normals(double S, double K, double T, double r, double sigma, double q, int is_call, int64_t n_paths,
                              int32_t n_steps, uint64_t seed, olmc_stats* out);

/* Microseconds one more DEPENDENT kernel launch costs on a stream (a chain of n empty kernels): the command processor's floor under
 * every per-date launch of olmc_american_lsm. */
int olmc_launch_gap_probe(int32_t n, double* us_per_launch);

/* ---- test seams (0 = off, the default) ------------------------------------
 *   OLMC_PROBE_TUNE_FAULT_SHARD      k > 0: rank k - 1 of a multi-GPU call fails before it launches (error-path tests)
 *   OLMC_PROBE_TUNE_FORCE_NV         v > 0: reduction workspaces REPORT a capacity of v values per workgroup row, so a kernel that
 *                                    reduces more than v values trips its device-side bound check (result NaN, nothing written out
 *                                    of bounds, library usable afterwards)
 *   OLMC_PROBE_TUNE_MULTI_REHEARSAL  1: the n ranks of olmc_multi_gpu_* all run on the caller's ONE device, each with its own
 *                                    stream and buffers, and the RCCL all-reduce (which refuses two ranks on one GPU) is replaced by
 *                                    a kernel per rank that adds the n send buffers in rank order behind every rank's path kernel.
 *                                    The engine with its launcher thread per rank, partitioning, payload layout, the hand-over by rank
 *                                    0's completion word, the drain of the ranks and the restoration of the thread's device run
 *                                    exactly as with n devices (except that n launchers then queue on ONE device's runtime locks: the
 *                                    launch phase is no faster than the serial form's there); the call also checks that every rank
 *                                    ended with rank 0's bits. */
enum { OLMC_PROBE_TUNE_FAULT_SHARD = 5, OLMC_PROBE_TUNE_FORCE_NV = 6, OLMC_PROBE_TUNE_MULTI_REHEARSAL = 11 };
int olmc_probe_tune(int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* OLMC_PROBE_H */
