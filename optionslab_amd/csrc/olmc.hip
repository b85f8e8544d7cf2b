// olmc.hip -- host side of libolmc.so: the C ABI declared in include/olmc.h.
//
// One context per HIP device (stream, workspace of the fused grid reduction,
// a pinned host landing buffer).  Every compute entry point is
//   host: derive the per-contract constants in fp64 (reference order)
//   device: ONE path kernel (its last workgroup writes the reduced sums) -> 8*NV bytes D2H
// and fails loudly when no HIP device is usable -- there is no CPU fallback.
#include "olmc.h"
#include "olmc_kernels.h"
#include "olmc_job_board.h"

#include <dlfcn.h>
#include <sched.h>
#include <time.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>      // types and enum values only: the library itself is dlopen()ed on first multi-GPU use

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include <linux/futex.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {

using namespace olmc;

thread_local std::string t_error = "";
thread_local int t_device = -1;

int fail(int code, const std::string& msg) {
    t_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(OLMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

constexpr int kMaxDevices = 16;
constexpr int kMaxNV = 2 * OLMC_MAX_BATCH;       // values per workgroup row
constexpr int32_t kMaxGrid = 1 << 18;            // workgroups per launch; larger jobs grid-stride
constexpr int32_t kMaxGroups = kMaxGrid / kGroupBlocks + 1;
static_assert(kMaxGroups * kMaxNV == kGroupRowsCapacity, "device-side guard must match the group_rows allocation");

struct EventPair {
    hipEvent_t start, stop;
};

struct DeviceCtx {
    int device = -1;
    int cus = 0;
    hipStream_t stream = nullptr;
    // Reduction workspaces.  Each CALLER stream gets one of kSlots of them to itself (stream order guards it), so
    // independent pricings enqueued on DIFFERENT streams may overlap on the device (the tail of
    // one launch hides under the head of the next) without sharing rows or counters; a ninth, tenth ... stream shares
    // slots with the others, rotating and event-guarded.  Launches on the library's
    // OWN stream (every blocking entry point) use slot kSlots and no event at all: the stream is in-order, so a
    // launch finds the workspace free by construction -- and a blocking call is spared the two barrier packets
    // (hipStreamWaitEvent in front of the kernel, hipEventRecord behind it): ~1 us per call, measured.
    struct WsSlot {
        double* block_rows = nullptr;  // [cap] doubles, grown on demand (grid x NV)
        size_t cap = 0;
        double* group_rows = nullptr;  // [kMaxGroups][kMaxNV]
        uint32_t* counters = nullptr;  // [(kMaxGroups + 1) * kCounterStride], zero between launches (self-resetting)
        hipEvent_t done = nullptr;     // recorded after the slot's latest launch (shared slots only)
        bool used = false;
        bool claimed = false;          // `owner` (which may be the NULL stream) is the ONE caller stream that has used this slot so far:
        hipStream_t owner = nullptr;   // its launches need no event (stream order)
        bool shared = false;           // a second stream had to take the slot: event-guarded from then on
    };
    static constexpr int kSlots = 8;    // = the deepest overlap bench.py's `pipelined` pass asks for (--streams 8)
    WsSlot slots[kSlots + 1];
    int next_slot = 0, cur_slot = 0;
    // Completion by polling: a blocking call on the library stream arms done_flag (a word of the same pinned buffer) with a
    // fresh sequence number; the wave that writes the results raises it, and the host reads the results the moment it
    // sees the number instead of waiting for the kernel's completion signal to travel through the runtime.
    uint64_t* h_flag = nullptr;      // host view of the flag word
    uint64_t* d_flag = nullptr;      // device alias
    uint64_t seq = 0;                // last sequence number handed out
    uint64_t armed = 0;              // != 0: the launch just made raises h_flag to this value
    double* h_result = nullptr;      // pinned + mapped [kMaxNV + 1 (+ flag)]: the last workgroup writes the sums straight to the host
    double* d_result = nullptr;      // device alias of h_result (zero-copy: no D2H copy node, only a stream sync)
    void* d_bulk = nullptr;          // terminal prices / validation taps
    size_t bulk_bytes = 0;
    // large results on their way home (copy_to_host): two pinned staging buffers, an event per buffer (allocated by the first large copy)
    void* h_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_landed[2] = {nullptr, nullptr};
    // independent-contract batches (european_multi_kernel)
    void* d_multi = nullptr;         // device [ticket counters | done counter | contracts | rows], see olmc_european_multi
    void* h_multi = nullptr;         // pinned + mapped [contracts | sums]
    void* d_h_multi = nullptr;       // device alias of h_multi
    int64_t multi_cap = 0;           // capacity in contracts
    size_t multi_bpo = 0;            // workgroups per contract the rows are sized for
    size_t multi_off_done = 0, multi_off_opts = 0, multi_off_rows = 0, multi_hoff_out = 0;
    // scrambled-Sobol tables of the last QMC call, kept on the device: FD Greeks and repeated pricings reuse one table
    uint32_t* d_sobol = nullptr;     // [dims x 30 direction numbers | dims shifts]
    size_t sobol_words = 0;          // capacity
    std::vector<uint32_t> sobol_host;   // what d_sobol holds (compared word for word with the caller's table)
    bool busy = false;                  // leased (guarded by the pool's mutex)
    // profiling
    std::vector<EventPair> ev_free, ev_pending;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
};

// Contexts of one device.  A call LEASES a context for its whole duration (CtxLease): it picks the first idle one under the pool's
// mutex -- a single-threaded caller therefore always works on context 0, exactly round 3's one-context library -- and everything
// after that (argument staging, launches, the wait for the result) runs outside any lock, on the context's own stream, workspace,
// pinned landing buffer and completion word.  Concurrent callers (Streamlit runs one thread per session, SURVEY §8b "Threading")
// get a context each, up to kMaxContexts per device (created on demand, never destroyed before olmc_shutdown); a further caller
// waits for a lease to come back.  Round 3 held ONE mutex around launch AND wait, so N sessions' pricings ran strictly one after
// the other, host launch overhead and device ramp / drain included.
constexpr int kMaxContexts = 8;

struct DevicePool {
    std::mutex mu;
    std::condition_variable idle;       // a lease came back
    std::vector<DeviceCtx*> all;        // grows to kMaxContexts, first idle first
    int device = -1;
};

std::mutex g_mu;                        // serialises the WRITERS of g_pool[] and g_default_device (olmc_init / olmc_shutdown);
std::atomic<DevicePool*> g_pool[kMaxDevices] = {};      // readers (every call's ctx_lease, possibly while another thread initialises a
std::atomic<int> g_default_device{-1};                  // device) take no lock
bool g_profile = false;

int ctx_allocate(DeviceCtx* c) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& sl : c->slots) {
        HIP_TRY(hipMalloc(&sl.group_rows, sizeof(double) * kMaxNV * kMaxGroups));
        HIP_TRY(hipMalloc(&sl.counters, sizeof(uint32_t) * (kMaxGroups + 1) * kCounterStride));
        HIP_TRY(hipMemset(sl.counters, 0, sizeof(uint32_t) * (kMaxGroups + 1) * kCounterStride));
        HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    HIP_TRY(hipHostMalloc(&c->h_result, sizeof(double) * (kMaxNV + 1 + 15), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_result), c->h_result, 0));
    c->h_flag = reinterpret_cast<uint64_t*>(c->h_result + kMaxNV + 8);      // a cache line of its own
    c->d_flag = reinterpret_cast<uint64_t*>(c->d_result + kMaxNV + 8);
    *c->h_flag = 0;
    return OLMC_OK;
}

// Frees whatever a context owns (tolerates partially built contexts) and deletes it.
void ctx_release(DeviceCtx* c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& ep : c->ev_pending) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
    for (auto& ep : c->ev_free) { (void)hipEventDestroy(ep.start); (void)hipEventDestroy(ep.stop); }
    if (c->d_bulk) (void)hipFree(c->d_bulk);
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        if (c->stage_landed[i]) (void)hipEventDestroy(c->stage_landed[i]);
    }
    if (c->d_multi) (void)hipFree(c->d_multi);
    if (c->h_multi) (void)hipHostFree(c->h_multi);
    if (c->d_sobol) (void)hipFree(c->d_sobol);
    for (auto& sl : c->slots) {
        if (sl.block_rows) (void)hipFree(sl.block_rows);
        if (sl.group_rows) (void)hipFree(sl.group_rows);
        if (sl.counters) (void)hipFree(sl.counters);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (c->h_result) (void)hipHostFree(c->h_result);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int ctx_create(int device, DeviceCtx** out) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(OLMC_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= count || device >= kMaxDevices)
        return fail(OLMC_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    DeviceCtx* c = new DeviceCtx();
    c->device = device;
    c->cus = prop.multiProcessorCount;
    const int rc = ctx_allocate(c);
    if (rc) {                       // do not leak a half-built context
        ctx_release(c);
        return rc;
    }
    *out = c;
    return OLMC_OK;
}

// RAII lease of one context of the calling thread's device (see DevicePool).
struct CtxLease {
    DeviceCtx* c = nullptr;
    DevicePool* pool = nullptr;
    CtxLease() = default;
    CtxLease(const CtxLease&) = delete;
    CtxLease& operator=(const CtxLease&) = delete;
    ~CtxLease() { release(); }
    void release() {
        if (!c) return;
        {
            std::lock_guard<std::mutex> lock(pool->mu);
            c->busy = false;
        }
        // notify_all: the pool's one condition variable has two kinds of waiter (a lease, in ctx_lease_on; "every context idle", in
        // pool_forget_stream / with_all_contexts) -- notify_one could hand the only wake-up to a waiter whose predicate is still false
        pool->idle.notify_all();
        c = nullptr;
    }
};

int ctx_lease_on(int dev, CtxLease* out) {
    DevicePool* pool = g_pool[dev].load(std::memory_order_acquire);
    if (!pool) return fail(OLMC_ERR_STATE, "device not initialised (call olmc_init)");
    HIP_TRY(hipSetDevice(dev));
    std::unique_lock<std::mutex> lock(pool->mu);
    for (;;) {
        for (DeviceCtx* c : pool->all)
            if (!c->busy) {
                c->busy = true;
                out->c = c;
                out->pool = pool;
                return OLMC_OK;
            }
        if (static_cast<int>(pool->all.size()) < kMaxContexts) {
            DeviceCtx* c = nullptr;
            const int rc = ctx_create(dev, &c);
            if (rc) return rc;
            c->busy = true;
            pool->all.push_back(c);
            out->c = c;
            out->pool = pool;
            return OLMC_OK;
        }
        pool->idle.wait(lock);
    }
}

int ctx_lease(CtxLease* out) {
    int dev = t_device >= 0 ? t_device : g_default_device.load(std::memory_order_acquire);
    if (dev < 0) {
        // lazy init on device 0 so a bare compute call still works (or fails loudly)
        int rc = olmc_init(0);
        if (rc) return rc;
        dev = t_device;
    }
    return ctx_lease_on(dev, out);
}

int bulk_reserve(DeviceCtx* c, size_t bytes) {
    if (bytes <= c->bulk_bytes) return OLMC_OK;
    if (c->d_bulk) HIP_TRY(hipFree(c->d_bulk));
    c->d_bulk = nullptr;
    c->bulk_bytes = 0;
    HIP_TRY(hipMalloc(&c->d_bulk, bytes));
    c->bulk_bytes = bytes;
    return OLMC_OK;
}

// A large result (a path matrix: 100k x 252 doubles = 202 MB; a terminal array of 64M paths = 1 GB) on its way to the CALLER's buffer,
// which is pageable and, coming fresh from NumPy, not even faulted in.  One hipMemcpyAsync into it runs at 10-17 GB/s (round 4: 20 ms for
// the 202 MB): the runtime stages through pinned memory and ONE host thread copies every chunk out, paying every first-touch page
// fault on the way.  Here the device result leaves in 16 MB chunks by DMA into two pinned staging buffers (link rate: ~50 GB/s), and
// while chunk i + 1 is on the link, a handful of host threads copy chunk i out, a slice each -- so the page faults of the destination
// are taken in parallel too.  Work queued on c->stream before the call (the kernel that writes d_src) is ordered before the first DMA;
// the call returns when the last byte is in `dst`.  Bytes are copied, never interpreted: same array as the direct copy, bit for bit.
constexpr size_t kStageBytes = size_t(16) << 20, kStagedFrom = size_t(32) << 20;
int g_staged_copy = 0;       // OLMC_TUNE_STAGED_COPY: 0 = results of 32 MB and more come home through pinned staging buffers and parallel host
                             // copies (default), -1 = always one hipMemcpyAsync into the caller's buffer

int copy_to_host(DeviceCtx* c, void* dst, const void* d_src, size_t bytes) {
    if (bytes < kStagedFrom || g_staged_copy < 0) {
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return OLMC_OK;
    }
    for (int i = 0; i < 2; ++i) {
        if (!c->h_stage[i]) HIP_TRY(hipHostMalloc(&c->h_stage[i], kStageBytes, hipHostMallocDefault));
        if (!c->stage_landed[i]) HIP_TRY(hipEventCreateWithFlags(&c->stage_landed[i], hipEventDisableTiming));
    }
    cpu_set_t cpus;
    int allowed = 4;
    if (sched_getaffinity(0, sizeof cpus, &cpus) == 0) allowed = CPU_COUNT(&cpus);
    const int n_threads = std::max(1, std::min(8, allowed / 2));
    const int64_t n_chunks = static_cast<int64_t>((bytes + kStageBytes - 1) / kStageBytes);
    std::atomic<int64_t> landed{0};                 // chunks whose DMA has completed: [0, landed) may be copied out
    std::atomic<int> copied[2] = {{0}, {0}};        // threads done with the chunk that sits in staging buffer 0 / 1
    std::atomic<bool> give_up{false};
    auto chunk_len = [&](int64_t i) { return std::min(kStageBytes, bytes - static_cast<size_t>(i) * kStageBytes); };
    auto worker = [&](int w) {
        for (int64_t i = 0; i < n_chunks; ++i) {
            for (uint32_t spins = 0; landed.load(std::memory_order_acquire) <= i; ++spins) {
                if (give_up.load(std::memory_order_relaxed)) return;
                if ((spins & 0xFF) == 0xFF) sched_yield(); else __builtin_ia32_pause();
            }
            const size_t len = chunk_len(i), per = (len / n_threads + 4095) & ~size_t(4095);      // page-sized slices: each page of dst has one toucher
            const size_t lo = std::min(len, per * w), hi = std::min(len, lo + per);
            if (hi > lo) std::memcpy(static_cast<char*>(dst) + static_cast<size_t>(i) * kStageBytes + lo, static_cast<const char*>(c->h_stage[i & 1]) + lo, hi - lo);
            copied[i & 1].fetch_add(1, std::memory_order_release);
        }
    };
    static const bool trace = std::getenv("OLMC_TRACE_COPY") != nullptr;
    using clock = std::chrono::steady_clock;
    auto us_since = [](clock::time_point t) { return std::chrono::duration<double, std::micro>(clock::now() - t).count(); };
    const auto t_begin = clock::now();
    double us_spawn = 0, us_first = 0, us_events = 0, us_workers = 0;
    std::vector<std::thread> threads;
    threads.reserve(n_threads);
    try {
        for (int w = 0; w < n_threads; ++w) threads.emplace_back(worker, w);
    } catch (const std::system_error&) {             // no thread to be had: the single copy still works
        give_up.store(true);
        for (std::thread& t : threads) t.join();
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return OLMC_OK;
    }
    us_spawn = us_since(t_begin);
    hipError_t err = hipSuccess;
    for (int64_t i = 0; i < n_chunks && err == hipSuccess; ++i) {
        const int slot = static_cast<int>(i & 1);
        if (i >= 2) {                               // the chunk that sat in this buffer has been copied out by every thread
            const auto t = clock::now();
            for (uint32_t spins = 0; copied[slot].load(std::memory_order_acquire) < n_threads; ++spins)
                if ((spins & 0xFF) == 0xFF) sched_yield(); else __builtin_ia32_pause();
            copied[slot].store(0, std::memory_order_relaxed);
            us_workers += us_since(t);
        }
        err = hipMemcpyAsync(c->h_stage[slot], static_cast<const char*>(d_src) + static_cast<size_t>(i) * kStageBytes, chunk_len(i), hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipEventRecord(c->stage_landed[slot], c->stream);
        if (err == hipSuccess && i >= 1) {
            const auto t = clock::now();
            err = hipEventSynchronize(c->stage_landed[slot ^ 1]);
            if (i == 1) us_first = us_since(t); else us_events += us_since(t);
            if (err == hipSuccess) landed.store(i, std::memory_order_release);
        }
    }
    {
        const auto t = clock::now();
        if (err == hipSuccess) err = hipEventSynchronize(c->stage_landed[(n_chunks - 1) & 1]);
        us_events += us_since(t);
    }
    if (err == hipSuccess) landed.store(n_chunks, std::memory_order_release);
    else give_up.store(true);
    const auto t_join = clock::now();
    for (std::thread& t : threads) t.join();
    if (trace)
        std::fprintf(stderr, "[olmc copy_to_host] %.1f MB, %lld chunks, %d threads: spawn %.0f us, first chunk landed (kernel + DMA) %.0f us, later event waits %.0f us, "
                             "waits for the copy threads %.0f us, last chunk out + join %.0f us, total %.0f us\n", bytes / 1e6, static_cast<long long>(n_chunks), n_threads,
                     us_spawn, us_first, us_events, us_workers, us_since(t_join), us_since(t_begin));
    if (err != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        return fail(OLMC_ERR_HIP, std::string("copy_to_host: ") + hipGetErrorString(err));
    }
    return OLMC_OK;
}

// Tuning knob (olmc_tune): 0 = automatic.
int g_multi_launch = 0;      // OLMC_TUNE_MULTI_LAUNCH: 0 = multi-GPU calls queue their ranks from one launcher thread per device (default), -1 = serial
int g_grid_cap = 0;          // OLMC_TUNE_GRID_CAP: max workgroups per launch (0 = kMaxGrid)
int g_qmc_block = 0;         // OLMC_TUNE_QMC_BLOCK: 0 = by size, 1 = always eight points per thread, -1 = never (and never split), 2 = always split
int g_poll = 0;              // OLMC_TUNE_POLL: 0 = blocking calls poll a host-mapped flag for completion (default), -1 = hipStreamSynchronize
int g_split_tail = 0;        // OLMC_TUNE_SPLIT_TAIL: 0 = split workgroups for the remainder of a European launch (default), -1 = never
int g_split_sat = 0;         // OLMC_TUNE_SPLIT_SAT: k > 0: a last round of fewer than k whole workgroups per CU is split too; 0 = never (default:
                             // measured, no gain -- see european_launch_shape)
#ifdef OLMC_WITH_PROBES      // the instrumented build (tools/probe/olmc_probe.hip -> libolmc_probe.so) only: test seams, see include/olmc_probe.h
int g_fault_shard = 0;       // OLMC_PROBE_TUNE_FAULT_SHARD: k > 0 makes rank k - 1 of a multi-GPU call fail before it launches
int g_force_nv = 0;          // OLMC_PROBE_TUNE_FORCE_NV: > 0 makes workspaces REPORT room for this many values per row (test of the device guard)
int g_multi_rehearsal = 0;   // OLMC_PROBE_TUNE_MULTI_REHEARSAL: != 0 runs the n ranks of a multi-GPU call on the caller's ONE device
#endif

// Launch geometry: one workgroup per 256 paths, handed out by the hardware dispatcher
// (measured faster than a fixed 8-workgroups-per-CU grid-stride); beyond kMaxGrid
// workgroups the kernels grid-stride.
// Short paths (n_steps <= 128, the reference's default is ONE step) are bounded by per-workgroup costs
// (dispatch, row store, ticket), not by the step loop: there 16 workgroups per CU that grid-stride beat
// one workgroup per 256 paths (8M x 4: 63 -> 43 us, 8M x 32: 143 -> 125 us; equal from 128 steps on,
// a one-off probe of round 3, profiles/HISTORY.md).
constexpr int32_t kShortPathSteps = 128, kShortPathGrid = 4096;

int32_t grid_for(int64_t n_paths, int32_t n_steps = INT32_MAX) {
    int64_t cap = g_grid_cap > 0 ? std::min<int64_t>(g_grid_cap, kMaxGrid) : kMaxGrid;
    if (g_grid_cap == 0 && n_steps <= kShortPathSteps) cap = kShortPathGrid;
    return static_cast<int32_t>(std::min<int64_t>((n_paths + kBlock - 1) / kBlock, cap));
}

// Resident workgroups per compute unit of one kernel instantiation (register / LDS limited), asked of the runtime once.
template <typename Kernel>
int resident_workgroups(Kernel kernel) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> cache;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& e : cache)
        if (e.first == key) return e.second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    cache.emplace_back(key, nb);
    return nb;
}

// Launch shape of european_path_kernel for pr.count paths: returns the grid and sets pr->split_from.  When one workgroup
// per 256 paths covers the launch (no grid-striding) the first F workgroups own 256 paths each and the remaining paths go to
// split workgroups of 64 paths (four waves x a quarter of the steps each; see the kernel) -- unless the paths have fewer than
// four full fp32 groups (64 steps) to hand out.
//
// F = floor(W / C) C: every CU gets the same number of whole workgroups, the remainder is spread in quarter-length units.
// Round 3 asked whether a THIN last round of whole workgroups (1M paths = 15 per CU = 7 + 7 + 1 at 7 resident per CU) should be
// split too, on the theory that a lone wave per SIMD issues slowly.  It does not: exactly k C whole workgroups take
// 9.4 + 6.03 k us for every k from 1 to 28 (profiles/r03_occupancy_probe.jsonl) -- one wave per SIMD
// already issues at the full rate (four interleaved Philox blocks are enough ILP), the SIMD serves its waves oldest-first
// rather than in rounds (first workgroup of a 7-per-CU launch done after 9 us, last after 48: profiles/r03_phase_stamps.jsonl),
// and OLMC_TUNE_SPLIT_SAT at 2 / 5 / 7 left the 1M x 252 kernel at 100.35 us +- 0.1 (profiles/r03_ab_kernels.txt).  The knob
// stays for measurements; the default is off.
int32_t european_launch_shape(const DeviceCtx* c, PathRange* pr, int occ) {
    pr->split_from = INT32_MAX;
    const int32_t grid = grid_for(pr->count, pr->n_steps);
    const int64_t wgs = (pr->count + kBlock - 1) / kBlock;
    if (g_split_tail < 0 || g_grid_cap != 0 || wgs != grid) return grid;           // tuned or grid-striding launches keep their shape
    if ((pr->n_steps >> 2) / kGroup < 4 || c->cus < 1) return grid;
    int64_t per_cu = (pr->count / kBlock) / c->cus;                                 // whole 256-path workgroups per CU
    if (occ >= 1 && g_split_sat > 0) {
        const int64_t r = per_cu % occ;
        if (r > 0 && r < std::min(g_split_sat, occ)) per_cu -= r;
    }
    const int64_t full = per_cu * c->cus;
    const int64_t rest = pr->count - full * kBlock;
    if (rest == 0) return grid;
    const int64_t split = (rest + kWave - 1) / kWave;
    if (full + split > kMaxGrid) return grid;
    pr->split_from = static_cast<int32_t>(full);
    return static_cast<int32_t>(full + split);
}

// The kernel launch_european<NSETS, MODE> will pick for a launch that covers every path (the only shape that splits).
template <int NSETS, int MODE>
int european_occupancy(bool anti) {
    if (g_split_sat <= 0) return 0;              // only the measurement knob needs it: nothing on the path of an ordinary call
    return anti ? resident_workgroups(european_path_kernel<NSETS, true, MODE, false>) : resident_workgroups(european_path_kernel<NSETS, false, MODE, false>);
}

// Workspace of the fused grid reduction for a launch of `grid` workgroups x nv values.
int make_ws(DeviceCtx* c, hipStream_t stream, int32_t grid, int nv, double* d_out, double tail, ReduceWs* ws) {
    const bool own = stream == c->stream;
    int idx = DeviceCtx::kSlots;
    if (!own) {
        // a caller stream keeps ONE slot to itself while there are slots to go round (bench.py's 8 streams, a rank's one stream):
        // its launches are ordered by the stream, no event needed.  Only when more streams than slots show up are slots shared,
        // rotating and event-guarded as before.
        idx = -1;
        for (int i = 0; i < DeviceCtx::kSlots && idx < 0; ++i)
            if (c->slots[i].claimed && !c->slots[i].shared && c->slots[i].owner == stream) idx = i;
        for (int i = 0; i < DeviceCtx::kSlots && idx < 0; ++i)
            if (!c->slots[i].claimed) { c->slots[i].claimed = true; c->slots[i].owner = stream; idx = i; }
        if (idx < 0) {
            idx = c->next_slot;
            c->next_slot = (c->next_slot + 1) % DeviceCtx::kSlots;
            DeviceCtx::WsSlot& taken = c->slots[idx];
            if (!taken.shared) {
                // its owner's launches so far carry no event: drain them once, then guard by events.  The owner is a CALLER's stream
                // and may have been destroyed since (a handle this library must not touch again: hipStreamSynchronize on a dead stream
                // is a use-after-free in the runtime, not an error code), so the whole device is drained -- once per slot, ever.
                HIP_TRY(hipDeviceSynchronize());
                taken.shared = true;
                taken.used = false;
            }
        }
    }
    DeviceCtx::WsSlot& sl = c->slots[idx];
    c->cur_slot = idx;
    const bool guarded = !own && sl.shared;
    const size_t need = static_cast<size_t>(grid) * nv;
    if (need > sl.cap) {
        if (!guarded) HIP_TRY(hipStreamSynchronize(stream));
        else if (sl.used) HIP_TRY(hipEventSynchronize(sl.done));
        if (sl.block_rows) HIP_TRY(hipFree(sl.block_rows));
        sl.block_rows = nullptr;
        sl.cap = 0;
        const size_t cap = std::max<size_t>(need, size_t(1) << 16);
        HIP_TRY(hipMalloc(&sl.block_rows, sizeof(double) * cap));
        sl.cap = cap;
    }
    if (guarded && sl.used) HIP_TRY(hipStreamWaitEvent(stream, sl.done, 0));   // previous user of this shared slot, whatever its stream
    ws->block_rows = sl.block_rows;
    ws->group_rows = sl.group_rows;
    ws->counters = sl.counters;
    ws->out = d_out;
    ws->tail = tail;
    ws->row_capacity = sl.cap;
#ifdef OLMC_WITH_PROBES
    if (g_force_nv > 0) ws->row_capacity = static_cast<uint64_t>(grid) * g_force_nv;        // the knob UNDER-reports (test of the guard)
#endif
    ws->done_flag = nullptr;
    ws->done_value = 0;
    c->armed = 0;
    if (own && d_out == c->d_result && g_poll >= 0) {      // a blocking call: results and flag land in the same pinned buffer
        c->armed = ++c->seq;
        ws->done_flag = c->d_flag;
        ws->done_value = c->armed;
    }
    return OLMC_OK;
}

// After a failed launch the self-resetting counters may be left dirty.
void ws_recover(DeviceCtx* c) {
    c->armed = 0;
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    for (auto& sl : c->slots) (void)hipMemset(sl.counters, 0, sizeof(uint32_t) * (kMaxGroups + 1) * kCounterStride);
}

// The same for the batch workspace (self-resetting ticket counters and the batch's done counter).
void multi_recover(DeviceCtx* c) {
    c->armed = 0;
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    if (c->d_multi) (void)hipMemset(c->d_multi, 0, c->multi_off_opts);
}

// Call right after launching a kernel that uses the workspace handed out by make_ws.
int after_launch(DeviceCtx* c, hipStream_t stream) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ws_recover(c); return fail(OLMC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e)); }
    if (c->cur_slot == DeviceCtx::kSlots || !c->slots[c->cur_slot].shared) return OLMC_OK;    // one stream's own slot: stream order is the guard
    DeviceCtx::WsSlot& sl = c->slots[c->cur_slot];
    HIP_TRY(hipEventRecord(sl.done, stream));
    sl.used = true;
    return OLMC_OK;
}

// ---- profiling brackets ---------------------------------------------------
int prof_begin(DeviceCtx* c, hipStream_t s, EventPair* ep) {
    if (c->ev_free.empty()) {
        HIP_TRY(hipEventCreate(&ep->start));
        HIP_TRY(hipEventCreate(&ep->stop));
    } else {
        *ep = c->ev_free.back();
        c->ev_free.pop_back();
    }
    HIP_TRY(hipEventRecord(ep->start, s));
    return OLMC_OK;
}

// An event pair for a dispatch that carries its own timestamps (launch_one): nothing is recorded here.
int prof_acquire(DeviceCtx* c, EventPair* ep) {
    if (c->ev_free.empty()) {
        HIP_TRY(hipEventCreate(&ep->start));
        HIP_TRY(hipEventCreate(&ep->stop));
    } else {
        *ep = c->ev_free.back();
        c->ev_free.pop_back();
    }
    return OLMC_OK;
}

int prof_end(DeviceCtx* c, hipStream_t s, const EventPair& ep) {
    HIP_TRY(hipEventRecord(ep.stop, s));
    c->ev_pending.push_back(ep);
    return OLMC_OK;
}

int prof_drain(DeviceCtx* c) {
    for (const EventPair& ep : c->ev_pending) {
        float ms = 0.f;
        // a pair whose launch failed was never recorded: it is skipped, not an error of the measurement
        if (hipEventSynchronize(ep.stop) == hipSuccess && hipEventElapsedTime(&ms, ep.start, ep.stop) == hipSuccess) {
            c->prof_ms += ms;
            c->prof_launches += 1;
        } else {
            (void)hipGetLastError();
        }
        c->ev_free.push_back(ep);
    }
    c->ev_pending.clear();
    return OLMC_OK;
}

// make_contract, group_contracts, finish_stats, poisoned, log_level, nan_stats, GreeksSet, cv_finish and the moment combiners are
// pure host arithmetic: olmc_host_math.h (compiled on its own, under sanitizers, by tests/test_host_math_sanitizers.py).

int check_paths(int64_t path_offset, int64_t n_local, int32_t n_steps) {
    if (n_local < 1) return fail(OLMC_ERR_ARG, "n_paths must be >= 1");
    if (n_steps < 1) return fail(OLMC_ERR_ARG, "n_steps must be >= 1");
    if (path_offset < 0) return fail(OLMC_ERR_ARG, "path_offset must be >= 0");
    return OLMC_OK;
}

PathRange make_range(int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed) {
    PathRange pr;
    pr.first = static_cast<uint64_t>(path_offset);
    pr.count = n_local;
    pr.n_steps = n_steps;
    pr.key0 = static_cast<uint32_t>(seed);
    pr.key1 = static_cast<uint32_t>(seed >> 32);
    pr.split_from = INT32_MAX;
    return pr;
}

// `timed` != nullptr: the dispatch itself carries the event pair (hipExtLaunchKernelGGL): the events take the
// kernel's own begin / end timestamps, as rocprofv3 reads them.  hipEventRecord brackets around a launch also
// time the marker packets on either side (+7..10 us at these durations: 119 vs 109 us in one and the same run).
template <typename Kernel, int NSETS>
void launch_one(Kernel kernel, int32_t grid, hipStream_t s, const EventPair* timed, const PathRange& pr, const ContractSet<NSETS>& cs,
                const ReduceWs& ws, double* terminal) {
    if (timed) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, s, timed->start, timed->stop, 0, pr, cs, ws, terminal);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, s, pr, cs, ws, terminal);
}

// The same for any kernel and any launch shape.
template <typename Kernel, typename... Args>
void launch_timed(Kernel kernel, dim3 grid, dim3 block, hipStream_t s, const EventPair* timed, Args... args) {
    if (timed) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, timed->start, timed->stop, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
}

// Profiling on: take an event pair for the launch that follows and queue it for prof_drain.
int prof_pair(DeviceCtx* c, EventPair* ep, const EventPair** timed) {
    *timed = nullptr;
    if (!g_profile) return OLMC_OK;
    int rc = prof_acquire(c, ep);
    if (rc) return rc;
    c->ev_pending.push_back(*ep);
    *timed = ep;
    return OLMC_OK;
}

template <int NSETS, int MODE>
void launch_european(bool anti, int32_t grid, hipStream_t s, const PathRange& pr, const ContractSet<NSETS>& cs,
                     const ReduceWs& ws, double* terminal, const EventPair* timed = nullptr) {
    const bool strided = static_cast<int64_t>(grid) * kBlock < pr.count;      // the grid does not cover every path
    if (strided) {
        if (anti) launch_one(european_path_kernel<NSETS, true, MODE, true>, grid, s, timed, pr, cs, ws, terminal);
        else launch_one(european_path_kernel<NSETS, false, MODE, true>, grid, s, timed, pr, cs, ws, terminal);
    } else {
        if (anti) launch_one(european_path_kernel<NSETS, true, MODE, false>, grid, s, timed, pr, cs, ws, terminal);
        else launch_one(european_path_kernel<NSETS, false, MODE, false>, grid, s, timed, pr, cs, ws, terminal);
    }
}

// ONE launch on stream `s` for k contracts: leaves {sum, sumsq} x k in d_out[0 .. 2k)
// (padded to the kernel's NSETS) and, when tail >= 0, `tail` in d_out[2 * nsets].
int run_batch_device(DeviceCtx* c, hipStream_t s, const olmc_option* opts, int32_t k, int64_t path_offset,
                     int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic, double* d_out, double tail,
                     int* pos /* [k]: slot of contract i in d_out, may be NULL when k == 1 */, bool* sums_only = nullptr
                     /* in: the caller needs no sums of squares; out: the launch made left ONE sum per slot (d_out[slot]) */) {
    PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const bool anti = antithetic != 0;
    const int nsets = k == 1 ? 1 : (k <= 8 ? 8 : 16);
    const int occ = nsets == 1 ? european_occupancy<1, kReduce>(anti) : (nsets == 8 ? european_occupancy<8, kReduce>(anti) : european_occupancy<16, kReduce>(anti));
    const int32_t grid = european_launch_shape(c, &pr, occ);
    // prices only (finite-difference Greeks without their evaluations' standard errors): the sum-only form of the fused kernels,
    // where the launch covers every path
    const bool lean = sums_only && *sums_only && nsets > 1 && static_cast<int64_t>(grid) * kBlock >= n_local;
    if (sums_only) *sums_only = lean;
    ReduceWs ws;
    int rc = make_ws(c, s, grid, lean ? nsets : 2 * nsets, d_out, tail, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    if (nsets == 1) {
        ContractSet<1> cs;
        cs.c[0] = make_contract(opts[0], n_steps);
        cs.base_mask = 1u; cs.upper_continues_slot0 = 0;
        if (pos) pos[0] = 0;
        launch_european<1, kReduce>(anti, grid, s, pr, cs, ws, nullptr, timed);
    } else if (nsets == 8) {
        ContractSet<8> cs;
        group_contracts<8>(opts, k, n_steps, &cs, pos);
        if (!lean) launch_european<8, kReduce>(anti, grid, s, pr, cs, ws, nullptr, timed);
        else if (anti) launch_one(european_path_kernel<8, true, kSumOnly, false>, grid, s, timed, pr, cs, ws, nullptr);
        else launch_one(european_path_kernel<8, false, kSumOnly, false>, grid, s, timed, pr, cs, ws, nullptr);
    } else {
        ContractSet<16> cs;
        group_contracts<16>(opts, k, n_steps, &cs, pos);
        if (!lean) launch_european<16, kReduce>(anti, grid, s, pr, cs, ws, nullptr, timed);
        else if (anti) launch_one(european_path_kernel<16, true, kSumOnly, false>, grid, s, timed, pr, cs, ws, nullptr);
        else launch_one(european_path_kernel<16, false, kSumOnly, false>, grid, s, timed, pr, cs, ws, nullptr);
    }
    rc = after_launch(c, s);
    if (rc) return rc;
    return OLMC_OK;
}

// Waits for the launch just made on stream s.  If it was armed (make_ws), the host polls the flag word in pinned memory:
// the results are there as soon as the flag shows the launch's sequence number.  The stream itself is left to drain on its
// own (it is in-order, so the next launch still starts after this kernel has retired).
// The poll is a spin only for as long as a spin is cheap: the first kSpinUs (200 us: every interactive size, the 1M x 252
// headline at 100 us) spin with `pause`; from then on every poll is followed by sched_yield(), so a host thread waiting for a
// 6 ms pricing of 64M paths hands its core to any other runnable thread (Streamlit runs one thread per session) instead of
// burning it; after kYieldUs (2 ms) the polls are kSleepUs apart (nanosleep at a 1 us timer slack: < 1 % on anything that long; the calling thread's
// CPU share over a 6 ms pricing falls from 1.0 to 0.34, profiles/r03_call_overhead.jsonl).  Insurance: from 2 ms on the stream
// is queried as well, about once a millisecond -- an error is reported as such, and a stream that reports completion without
// the flag having shown up falls back to the runtime's own wait.
constexpr int64_t kSpinUs = 200, kYieldUs = 2000, kSleepUs = 20;

// A nanosleep() wakes up to the thread's timer slack late -- 50 us by default, as long again as the nap asked for: a 2.3 ms pricing
// came back 0.16 ms after its kernel had ended.  While a call naps its thread's slack is 1 us (restored on the way out).
struct TimerSlack {
    long old = -1;
    void tighten() {
        if (old >= 0) return;
        old = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
        if (old >= 0 && prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0) != 0) old = -1;
    }
    ~TimerSlack() {
        if (old >= 0) (void)prctl(PR_SET_TIMERSLACK, static_cast<unsigned long>(old), 0, 0, 0);
    }
};

int wait_armed(DeviceCtx* c, hipStream_t s) {
    if (c->armed != 0) {
        const uint64_t want = c->armed;
        c->armed = 0;
        using clock = std::chrono::steady_clock;
        const auto t0 = clock::now();
        bool queried = false;
        int phase = 0;                                                           // 0 spin, 1 yield, 2 sleep + query
        TimerSlack slack;
        for (uint32_t spins = 0;; ++spins) {
            if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == want) {
                if (queried) (void)hipGetLastError();                        // a hipErrorNotReady answer must not linger as this thread's last error
                return OLMC_OK;
            }
            if (phase == 0) {
                __builtin_ia32_pause();
                if ((spins & 0x3F) == 0x3F && clock::now() - t0 >= std::chrono::microseconds(kSpinUs)) phase = 1;
                continue;
            }
            if (phase == 1) {
                sched_yield();
                if ((spins & 0xF) == 0xF && clock::now() - t0 >= std::chrono::microseconds(kYieldUs)) phase = 2;
                continue;
            }
            if ((spins & 0x3F) == 0) {                                       // the stream is asked every 64th nap (~1.5 ms), the word every time
                const hipError_t q = hipStreamQuery(s);
                queried = true;
                if (q == hipSuccess) break;                                  // retired: fall through to the runtime's wait
                if (q != hipErrorNotReady) {
                    ws_recover(c);
                    return fail(OLMC_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
                }
            }
            slack.tighten();
            const timespec nap{0, kSleepUs * 1000};
            nanosleep(&nap, nullptr);
        }
        (void)hipGetLastError();
    }
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { ws_recover(c); return fail(OLMC_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); }
    return OLMC_OK;
}

int sync_or_recover(DeviceCtx* c, hipStream_t s) {
    if (s != c->stream) c->armed = 0;               // only launches on the library's own stream are ever armed
    return wait_armed(c, s);
}

int run_batch(const olmc_option* opts, int32_t k, int64_t path_offset, int64_t n_local, int32_t n_steps,
              uint64_t seed, int antithetic, olmc_stats* out, bool prices_only = false) {
    if (!opts || !out) return fail(OLMC_ERR_ARG, "null pointer");
    if (k < 1 || k > OLMC_MAX_BATCH) return fail(OLMC_ERR_ARG, "batch size must be in [1, OLMC_MAX_BATCH]");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    int pos[OLMC_MAX_BATCH];
    bool lean = prices_only;
    rc = run_batch_device(c, c->stream, opts, k, path_offset, n_local, n_steps, seed, antithetic, c->d_result, -1.0, pos, &lean);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    const int64_t n = n_local * (antithetic ? 2 : 1);
    for (int i = 0; i < k; ++i) {
        if (poisoned(opts[i].S, opts[i].K, opts[i].T, opts[i].r, opts[i].sigma, opts[i].q)) {
            nan_stats(n, &out[i]);
        } else if (lean) {                           // one sum per slot: the price is exact, the standard error was not asked for
            finish_stats(c->h_result[pos[i]], std::nan(""), n, opts[i].r, opts[i].T, &out[i]);
        } else {
            finish_stats(c->h_result[2 * pos[i]], c->h_result[2 * pos[i] + 1], n, opts[i].r, opts[i].T, &out[i]);
        }
    }
    return OLMC_OK;
}

}  // namespace

namespace { void multi_gpu_release(); }      // the multi-GPU engine's streams, buffers and communicators (defined with it)

// =============================================================== lifetime ====
extern "C" int olmc_abi_version(void) { return OLMC_ABI_VERSION; }

extern "C" const char* olmc_last_error(void) { return t_error.c_str(); }

extern "C" int olmc_init(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    // operational safety valves (no rebuild, no code change in the host): OLMC_POLL=0 makes blocking calls wait with
    // hipStreamSynchronize, OLMC_SPLIT_TAIL=0 keeps one launch shape throughout -- the same switches olmc_tune offers
    static const bool env_read = [] {
        const char* p = std::getenv("OLMC_POLL");
        if (p && p[0] == '0') g_poll = -1;
        const char* t = std::getenv("OLMC_SPLIT_TAIL");
        if (t && t[0] == '0') g_split_tail = -1;
        const char* m = std::getenv("OLMC_MULTI_LAUNCH");          // "serial": the multi-GPU entry points queue their ranks from the calling thread
        if (m && m[0] == 's') g_multi_launch = -1;
        const char* sc = std::getenv("OLMC_STAGED_COPY");          // "0": large results come home by one hipMemcpyAsync
        if (sc && sc[0] == '0') g_staged_copy = -1;
        return true;
    }();
    (void)env_read;
    if (device < 0 || device >= kMaxDevices) return fail(OLMC_ERR_ARG, "device index out of range");
    if (!g_pool[device].load(std::memory_order_acquire)) {
        DeviceCtx* c = nullptr;
        int rc = ctx_create(device, &c);            // the first context of the device: a device that cannot be used fails HERE, loudly
        if (rc) return rc;
        DevicePool* pool = new DevicePool();
        pool->device = device;
        pool->all.push_back(c);
        g_pool[device].store(pool, std::memory_order_release);
    } else {
        HIP_TRY(hipSetDevice(device));
    }
    t_device = device;
    if (g_default_device.load() < 0) g_default_device.store(device, std::memory_order_release);
    return OLMC_OK;
}

extern "C" int olmc_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    multi_gpu_release();
    for (int d = 0; d < kMaxDevices; ++d) {         // the caller's contract: no call is in flight on any thread
        DevicePool* pool = g_pool[d].load();
        if (!pool) continue;
        const bool usable = hipSetDevice(d) == hipSuccess;
        for (DeviceCtx* c : pool->all) {
            if (usable) ctx_release(c);
            else delete c;
        }
        delete pool;
        g_pool[d].store(nullptr);
    }
    g_default_device.store(-1);
    t_device = -1;
    return OLMC_OK;
}

extern "C" int olmc_device_info(olmc_devinfo* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->name, sizeof(out->name), "%s", prop.name);
    std::snprintf(out->arch, sizeof(out->arch), "%s", prop.gcnArchName);
    out->compute_units = prop.multiProcessorCount;
    out->clock_mhz = prop.clockRate / 1000;
    out->wavefront = prop.warpSize;
    out->device = c->device;
    out->hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
    return OLMC_OK;
}

// ===================================================== European reductions ====
extern "C" int olmc_european(double S, double K, double T, double r, double sigma, double q, int is_call,
                             int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out) {
    const olmc_option o = make_option(S, K, T, r, sigma, q, is_call);
    return run_batch(&o, 1, 0, n_paths, n_steps, seed, antithetic, out);
}

extern "C" int olmc_european_shard(double S, double K, double T, double r, double sigma, double q, int is_call,
                                   int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                                   int antithetic, olmc_stats* out) {
    const olmc_option o = make_option(S, K, T, r, sigma, q, is_call);
    return run_batch(&o, 1, path_offset, n_local, n_steps, seed, antithetic, out);
}

extern "C" int olmc_european_shard_dev(double S, double K, double T, double r, double sigma, double q, int is_call,
                                       int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                                       int antithetic, double* d_triple, void* hip_stream) {
    if (!d_triple) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);   // used as given: NULL is the HIP null stream
    const olmc_option o = make_option(S, K, T, r, sigma, q, is_call);
    const double n = static_cast<double>(n_local * (antithetic ? 2 : 1));
    // the path kernel's last workgroup writes {sum, sumsq, n} straight into the caller's buffer
    return run_batch_device(c, s, &o, 1, path_offset, n_local, n_steps, seed, antithetic, d_triple, n, nullptr);
}

// Blocking fetch of n (<= 33) doubles that work already queued on `hip_stream` leaves at d_src (e.g. the triple after the caller's
// RCCL all-reduce): a one-wave kernel behind that work copies them into the pinned buffer and raises the completion word, the host
// polls it -- the same hand-over as a blocking pricing, instead of hipMemcpyAsync + hipStreamSynchronize (8 us -> 3 us per step).
extern "C" int olmc_fetch_dev(const double* d_src, int32_t n, void* hip_stream, double* out_host) {
    if (!d_src || !out_host) return fail(OLMC_ERR_ARG, "null pointer");
    if (n < 1 || n > kMaxNV + 1) return fail(OLMC_ERR_ARG, "n must be in [1, 33]");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const uint64_t want = ++c->seq;
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kWave), 0, s, d_src, n, c->d_result, c->d_flag, want);
    HIP_TRY(hipGetLastError());
    if (g_poll >= 0) {
        c->armed = want;
        rc = wait_armed(c, s);
    } else {
        rc = OLMC_OK;
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(OLMC_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    if (rc) return rc;
    for (int32_t i = 0; i < n; ++i) out_host[i] = c->h_result[i];
    return OLMC_OK;
}

extern "C" int olmc_european_batch(const olmc_option* opts, int32_t k, int64_t path_offset, int64_t n_local,
                                   int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out) {
    return run_batch(opts, k, path_offset, n_local, n_steps, seed, antithetic, out);
}

extern "C" int olmc_combine_stats(const olmc_stats* parts, int32_t n_parts, double r, double T, olmc_stats* out) {
    if (!parts || !out || n_parts < 1) return fail(OLMC_ERR_ARG, "bad arguments");
    if (!combine_stats(parts, n_parts, r, T, out)) return fail(OLMC_ERR_ARG, "no samples");     // fixed rank order => bitwise stable
    return OLMC_OK;
}

// ==================================================== independent contracts ====
extern "C" int olmc_european_multi(const olmc_option* opts, const uint32_t* tags, int64_t n_options, int64_t n_paths,
                                   int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out) {
    if (!opts || !out) return fail(OLMC_ERR_ARG, "null pointer");
    if (n_options < 1) return fail(OLMC_ERR_ARG, "n_options must be >= 1");
    if (n_options > (int64_t(1) << 31) - 2) return fail(OLMC_ERR_ARG, "too many contracts in one call");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const int32_t bpo = static_cast<int32_t>(std::min<int64_t>((n_paths + kBlock - 1) / kBlock, 1024));   // workgroups per contract
    auto align = [](size_t b) { return (b + 255) / 256 * 256; };
    // Device workspace [counters | done counter | contracts | rows]: the ticket counters sit FIRST and are sized by the capacity in
    // contracts, so their place does not move with the batch size and they are zeroed once per (re)allocation -- every finisher
    // re-zeroes the counter it consumed.  Pinned host staging [contracts | sums]: contracts go up by one asynchronous DMA from
    // pinned memory, the sums are written into pinned memory by the kernel itself, completion is the polled word of the context.
    if (n_options > c->multi_cap || static_cast<size_t>(bpo) > c->multi_bpo) {
        // the two capacities grow independently: more contracts doubles the contract capacity, more workgroups per contract (a
        // larger n_paths on the same batch) only widens the rows -- a convergence sweep over n_paths must not double `cap` each time
        const int64_t cap = n_options > c->multi_cap ? std::max<int64_t>(n_options, std::max<int64_t>(2 * c->multi_cap, 64)) : c->multi_cap;
        const size_t bpo_cap = std::max<size_t>(bpo, c->multi_bpo);
        const size_t b_cnt = align(sizeof(uint32_t) * cap * kMultiCounterStride), b_done = 256, b_opts = align(sizeof(MultiOption) * cap);
        const size_t b_rows = align(sizeof(double) * 2 * bpo_cap * cap);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_multi) HIP_TRY(hipFree(c->d_multi));
        if (c->h_multi) HIP_TRY(hipHostFree(c->h_multi));
        c->d_multi = nullptr; c->h_multi = nullptr; c->multi_cap = 0; c->multi_bpo = 0;
        HIP_TRY(hipMalloc(&c->d_multi, b_cnt + b_done + b_opts + b_rows));
        HIP_TRY(hipMemsetAsync(c->d_multi, 0, b_cnt + b_done, c->stream));
        HIP_TRY(hipHostMalloc(&c->h_multi, b_opts + align(sizeof(double) * 2 * cap), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&c->d_h_multi, c->h_multi, 0));
        c->multi_cap = cap;
        c->multi_bpo = bpo_cap;
        c->multi_off_done = b_cnt; c->multi_off_opts = b_cnt + b_done; c->multi_off_rows = b_cnt + b_done + b_opts; c->multi_hoff_out = b_opts;
    }
    char* base = static_cast<char*>(c->d_multi);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base);
    uint32_t* d_done = reinterpret_cast<uint32_t*>(base + c->multi_off_done);
    MultiOption* d_opts = reinterpret_cast<MultiOption*>(base + c->multi_off_opts);
    double* d_rows = reinterpret_cast<double*>(base + c->multi_off_rows);
    MultiOption* h = reinterpret_cast<MultiOption*>(c->h_multi);
    const double* h_out = reinterpret_cast<const double*>(static_cast<char*>(c->h_multi) + c->multi_hoff_out);
    double* d_out = reinterpret_cast<double*>(static_cast<char*>(c->d_h_multi) + c->multi_hoff_out);
    for (int64_t j = 0; j < n_options; ++j) {
        const Contract ct = make_contract(opts[j], n_steps);
        h[j].a = ct.a; h[j].vol = ct.vol; h[j].strike = ct.strike; h[j].sign = ct.sign;
        h[j].tag = tags ? tags[j] : static_cast<uint32_t>(j);
        h[j].pad = 0;
    }
    HIP_TRY(hipMemcpyAsync(d_opts, h, sizeof(MultiOption) * n_options, hipMemcpyHostToDevice, c->stream));
    MultiDone md;
    md.done_count = d_done;
    md.n_total = static_cast<uint32_t>(n_options);
    md.done_flag = nullptr;
    md.done_value = 0;
    c->armed = 0;
    if (g_poll >= 0) {
        c->armed = ++c->seq;
        md.done_flag = c->d_flag;
        md.done_value = c->armed;
    }
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    EventPair ep{};
    if (g_profile) { rc = prof_begin(c, c->stream, &ep); if (rc) return rc; }
    for (int64_t base_opt = 0; base_opt < n_options; base_opt += 65535) {
        const unsigned ny = static_cast<unsigned>(std::min<int64_t>(65535, n_options - base_opt));
        if (antithetic) hipLaunchKernelGGL((european_multi_kernel<true>), dim3(bpo, ny), dim3(kBlock), 0, c->stream, pr, d_opts, base_opt, d_rows, d_cnt, d_out, md);
        else hipLaunchKernelGGL((european_multi_kernel<false>), dim3(bpo, ny), dim3(kBlock), 0, c->stream, pr, d_opts, base_opt, d_rows, d_cnt, d_out, md);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            multi_recover(c);
            return fail(OLMC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
        }
    }
    if (g_profile) { rc = prof_end(c, c->stream, ep); if (rc) { multi_recover(c); return rc; } }
    rc = wait_armed(c, c->stream);
    if (rc) { multi_recover(c); return rc; }
    const int64_t n = n_paths * (antithetic ? 2 : 1);
    for (int64_t j = 0; j < n_options; ++j) {
        if (poisoned(opts[j].S, opts[j].K, opts[j].T, opts[j].r, opts[j].sigma, opts[j].q)) nan_stats(n, &out[j]);
        else finish_stats(h_out[2 * j], h_out[2 * j + 1], n, opts[j].r, opts[j].T, &out[j]);
    }
    return OLMC_OK;
}

// The layout olmc_european_batch / olmc_european_greeks_fd give a set of k contracts (pure host arithmetic: needs no device).
extern "C" int olmc_contract_layout(const olmc_option* opts, int32_t k, int32_t n_steps, int32_t* nsets_out, int32_t* pos, uint32_t* base_mask,
                                    int32_t* upper_continues_slot0, double* scale16) {
    if (!opts || !nsets_out || !pos || !base_mask || !upper_continues_slot0 || !scale16) return fail(OLMC_ERR_ARG, "null pointer");
    if (k < 2 || k > OLMC_MAX_BATCH) return fail(OLMC_ERR_ARG, "a set has 2 .. OLMC_MAX_BATCH contracts");
    if (n_steps < 1) return fail(OLMC_ERR_ARG, "n_steps must be >= 1");
    contract_layout(opts, k, n_steps, nsets_out, pos, base_mask, upper_continues_slot0, scale16);
    return OLMC_OK;
}

// Capacity of the batch workspace as it stands: {contracts, workgroups per contract} (0, 0 before the first batch).
extern "C" int olmc_multi_capacity(int64_t* out2) {
    if (!out2) return fail(OLMC_ERR_ARG, "null pointer");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    out2[0] = c->multi_cap;
    out2[1] = static_cast<int64_t>(c->multi_bpo);
    return OLMC_OK;
}

// ================================================== finite-difference Greeks ====
// GreeksSet (olmc_host_math.h): the evaluations of compute_greeks_unified (unified_greeks.py:274-277, 295-358) in the reference's
// get_price() call order, and the finite differences over their prices.
extern "C" int olmc_european_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                                       int64_t n_paths, int32_t n_steps, uint64_t seed, int second_order,
                                       double* out9, olmc_stats* evals) {
    if (!out9) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(T > 0.0)) return fail(OLMC_ERR_ARG, "T must be > 0 (price() returns intrinsic value without simulating)");
    const GreeksSet gs(S, K, T, r, sigma, q, is_call, second_order);
    olmc_stats st[OLMC_MAX_BATCH];
    int rc = run_batch(gs.o, gs.k, 0, n_paths, n_steps, seed, 1, st, /*prices_only=*/evals == nullptr);     // no evaluations asked for: no sums of squares
    if (rc) return rc;
    gs.finish(st, T, out9, evals);
    return OLMC_OK;
}

// ============================================================ terminal array ====
extern "C" int olmc_european_terminal(double S, double T, double r, double sigma, double q, int64_t n_paths,
                                      int32_t n_steps, uint64_t seed, int antithetic, double* out_host) {
    if (!out_host) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t bytes = sizeof(double) * static_cast<size_t>(n_paths) * (antithetic ? 2 : 1);
    rc = bulk_reserve(c, bytes);
    if (rc) return rc;
    PathRange pr = make_range(0, n_paths, n_steps, seed);
    const int32_t grid = european_launch_shape(c, &pr, european_occupancy<1, kTerminal>(antithetic != 0));
    ContractSet<1> cs;
    cs.c[0] = make_contract(make_option(S, 0.0, T, r, sigma, q, 1), n_steps);
    cs.base_mask = 1u; cs.upper_continues_slot0 = 0;
    ReduceWs ws{};   // unused in kTerminal mode
    launch_european<1, kTerminal>(antithetic != 0, grid, c->stream, pr, cs, ws, static_cast<double*>(c->d_bulk));
    HIP_TRY(hipGetLastError());
    return copy_to_host(c, out_host, c->d_bulk, bytes);
}

// =========================================================== control variate ====
extern "C" int olmc_european_cv_shard(double S, double K, double T, double r, double sigma, double q, int is_call,
                                      int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic,
                                      olmc_cv_moments* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const int32_t grid = european_launch_shape(c, &pr, european_occupancy<1, kControlVariate>(antithetic != 0));
    ContractSet<1> cs;
    cs.c[0] = make_contract(make_option(S, K, T, r, sigma, q, is_call), n_steps);
    cs.base_mask = 1u; cs.upper_continues_slot0 = 0;
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 5, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    launch_european<1, kControlVariate>(antithetic != 0, grid, c->stream, pr, cs, ws, nullptr, timed);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    cv_from_device(c->h_result, n_local * (antithetic ? 2 : 1), S, T, r, q, out);       // d = disc * x (monte_carlo.py:175)
    if (poisoned(S, K, T, r, sigma, q)) out->value = std::nan("");
    return OLMC_OK;
}

extern "C" int olmc_european_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                                int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic,
                                olmc_cv_moments* out) {
    return olmc_european_cv_shard(S, K, T, r, sigma, q, is_call, 0, n_paths, n_steps, seed, antithetic, out);
}

extern "C" int olmc_combine_cv(const olmc_cv_moments* parts, int32_t n_parts, double S, double T, double r, double q,
                               olmc_cv_moments* out) {
    if (!parts || !out || n_parts < 1) return fail(OLMC_ERR_ARG, "bad arguments");
    if (!combine_cv(parts, n_parts, S, T, r, q, out)) return fail(OLMC_ERR_ARG, "no samples");     // fixed rank order => bitwise stable
    return OLMC_OK;
}

// ===================================================================== Asian ====
extern "C" int olmc_asian(double S, double K, double T, double r, double sigma, double q, int is_call, int avg_kind,
                          int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic,
                          olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    if (avg_kind != OLMC_AVG_ARITHMETIC && avg_kind != OLMC_AVG_GEOMETRIC && avg_kind != OLMC_AVG_ARITHMETIC_FAST)
        return fail(OLMC_ERR_ARG, "bad avg_kind");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const int32_t grid = grid_for(n_local);
    AsianContract ac;
    const double dt = T / n_steps;                               // exotic_options.py:54-56
    ac.log_s0 = std::log(S);
    ac.s0 = S;
    ac.drift = (r - q - 0.5 * sigma * sigma) * dt;
    ac.vol = sigma * std::sqrt(dt);
    ac.strike = K;
    ac.sign = is_call ? 1.0 : -1.0;
    ac.inv_steps = 1.0 / n_steps;
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 2, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    const bool anti = antithetic != 0, geo = avg_kind == OLMC_AVG_GEOMETRIC, fast = avg_kind == OLMC_AVG_ARITHMETIC_FAST;
    if (anti && geo) launch_timed(asian_kernel<true, true>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ac, ws);
    else if (geo) launch_timed(asian_kernel<false, true>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ac, ws);
    else if (anti && fast) launch_timed(asian_kernel<true, false>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ac, ws);
    else if (fast) launch_timed(asian_kernel<false, false>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ac, ws);
    else if (anti) launch_timed(asian_exp64_kernel<true>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ac, ws);      // reference precision
    else launch_timed(asian_exp64_kernel<false>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ac, ws);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    finish_stats(c->h_result[0], c->h_result[1], n_local * (antithetic ? 2 : 1), r, T, out);
    if (poisoned(S, K, T, r, sigma, q)) nan_stats(out->n, out);
    return OLMC_OK;
}

// The 8 / 14 contracts of compute_greeks_unified over an arithmetic Asian (ExoticAdapter, unified_greeks.py:177-227) in one launch.
extern "C" int olmc_asian_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call, int avg_kind, int64_t n_paths,
                                    int32_t n_steps, uint64_t seed, int antithetic, int second_order, double* out9, olmc_stats* evals) {
    if (!out9) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(T > 0.0)) return fail(OLMC_ERR_ARG, "T must be > 0");
    if (avg_kind != OLMC_AVG_ARITHMETIC && avg_kind != OLMC_AVG_GEOMETRIC)
        return fail(OLMC_ERR_ARG, "avg_kind must be OLMC_AVG_ARITHMETIC or OLMC_AVG_GEOMETRIC (the fp32-exponent form has no fused Greeks)");
    const bool geo = avg_kind == OLMC_AVG_GEOMETRIC;
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    const GreeksSet gs(S, K, T, r, sigma, q, is_call, second_order);
    AsianGreeksSet as;
    constexpr double kLog2e = 1.4426950408889634;
    const double unit = geo ? 1.0 : (OLMC_EXP2_TABLE ? kExp2Entries * kLog2e : kLog2e);     // what each kernel's exponential counts in
    if (const char* bad = asian_greeks_layout(gs, n_steps, geo, unit, K, is_call, &as)) return fail(OLMC_ERR_STATE, bad);
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    if (n_paths > static_cast<int64_t>(kMaxGrid) * kBlock) return fail(OLMC_ERR_ARG, "n_paths beyond one launch of the fused Asian Greeks kernel (2^26)");
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    const int32_t grid = static_cast<int32_t>((n_paths + kBlock - 1) / kBlock);      // the grid covers every path
    const int nsets = gs.k <= 8 ? 8 : 16;
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 2 * nsets, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    const bool anti = antithetic != 0;
    if (geo) {
        if (nsets == 8 && anti) launch_timed(asian_geometric_greeks_kernel<true, 8>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
        else if (nsets == 8) launch_timed(asian_geometric_greeks_kernel<false, 8>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
        else if (anti) launch_timed(asian_geometric_greeks_kernel<true, 16>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
        else launch_timed(asian_geometric_greeks_kernel<false, 16>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
    } else if (nsets == 8 && anti) launch_timed(asian_exp64_greeks_kernel<true, 8>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
    else if (nsets == 8) launch_timed(asian_exp64_greeks_kernel<false, 8>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
    else if (anti) launch_timed(asian_exp64_greeks_kernel<true, 16>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
    else launch_timed(asian_exp64_greeks_kernel<false, 16>, dim3(grid), dim3(kBlock), c->stream, timed, pr, as, ws);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    const int64_t n = n_paths * (anti ? 2 : 1);
    olmc_stats st[OLMC_MAX_BATCH];
    for (int i = 0; i < gs.k; ++i) {
        finish_stats(c->h_result[2 * i], c->h_result[2 * i + 1], n, gs.o[i].r, gs.o[i].T, &st[i]);
        if (poisoned(gs.o[i].S, gs.o[i].K, gs.o[i].T, gs.o[i].r, gs.o[i].sigma, gs.o[i].q)) nan_stats(n, &st[i]);
    }
    gs.finish(st, T, out9, evals);
    return OLMC_OK;
}

// ========================================================== barrier / lookback ====
namespace {
int run_extrema(double S, double K, double T, double r, double sigma, double q, int is_call, int payoff, double barrier,
                int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const int32_t grid = grid_for(n_local);
    ExtremaContract ec;
    const double dt = T / n_steps;                               // exotic_options.py:54-56
    ec.s0 = S;
    ec.log_barrier_rel = payoff <= kBarrierDownIn ? std::log(barrier / S) : 0.0;
    ec.drift = (r - q - 0.5 * sigma * sigma) * dt;
    ec.vol = sigma * std::sqrt(dt);
    ec.strike = K;
    ec.sign = is_call ? 1.0 : -1.0;
    ec.payoff = payoff;
    ec.pad = 0;
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 2, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    if (antithetic) launch_timed(extrema_kernel<true>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ec, ws);
    else launch_timed(extrema_kernel<false>, dim3(grid), dim3(kBlock), c->stream, timed, pr, ec, ws);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    finish_stats(c->h_result[0], c->h_result[1], n_local * (antithetic ? 2 : 1), r, T, out);
    if (poisoned(S, K, T, r, sigma, q) || std::isnan(barrier)) nan_stats(out->n, out);
    return OLMC_OK;
}
}  // namespace

extern "C" int olmc_barrier(double S, double K, double T, double r, double sigma, double q, int is_call, double barrier,
                            int barrier_kind, int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed,
                            int antithetic, olmc_stats* out) {
    if (!(barrier > 0.0)) return fail(OLMC_ERR_ARG, "Barrier must be positive");
    if (barrier_kind < OLMC_BARRIER_UP_OUT || barrier_kind > OLMC_BARRIER_DOWN_IN) return fail(OLMC_ERR_ARG, "bad barrier_kind");
    return run_extrema(S, K, T, r, sigma, q, is_call, barrier_kind, barrier, path_offset, n_local, n_steps, seed, antithetic, out);
}

extern "C" int olmc_lookback(double S, double K, double T, double r, double sigma, double q, int is_call, int fixed_strike,
                             int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic,
                             olmc_stats* out) {
    return run_extrema(S, K, T, r, sigma, q, is_call, fixed_strike ? kLookbackFixed : kLookbackFloating, 0.0, path_offset,
                       n_local, n_steps, seed, antithetic, out);
}

// The 8 / 14 contracts of compute_greeks_unified over a barrier / lookback option (ExoticAdapter, unified_greeks.py:177-227) in one launch.
extern "C" int olmc_extrema_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call, int payoff, double barrier,
                                      int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic, int second_order, double* out9,
                                      olmc_stats* evals) {
    if (!out9) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(T > 0.0)) return fail(OLMC_ERR_ARG, "T must be > 0");
    if (payoff < kBarrierUpOut || payoff > kLookbackFixed) return fail(OLMC_ERR_ARG, "payoff must be a barrier kind (0..3) or OLMC_LOOKBACK_FLOATING / _FIXED (4, 5)");
    const bool is_barrier = payoff <= kBarrierDownIn;
    if (is_barrier && !(barrier > 0.0)) return fail(OLMC_ERR_ARG, "Barrier must be positive");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    if (n_paths > static_cast<int64_t>(kMaxGrid) * kBlock) return fail(OLMC_ERR_ARG, "n_paths beyond one launch of the fused Greeks kernel (2^26)");
    const GreeksSet gs(S, K, T, r, sigma, q, is_call, second_order);
    ExtremaGreeksSet es;
    if (const char* bad = extrema_greeks_layout(gs, n_steps, payoff, barrier, K, is_call, &es)) return fail(OLMC_ERR_STATE, bad);
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    const int32_t grid = static_cast<int32_t>((n_paths + kBlock - 1) / kBlock);      // the grid covers every path
    const int nsets = gs.k <= 8 ? 8 : 16;
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 2 * nsets, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    const bool anti = antithetic != 0;
    if (nsets == 8 && anti) launch_timed(extrema_greeks_kernel<true, 8>, dim3(grid), dim3(kBlock), c->stream, timed, pr, es, ws);
    else if (nsets == 8) launch_timed(extrema_greeks_kernel<false, 8>, dim3(grid), dim3(kBlock), c->stream, timed, pr, es, ws);
    else if (anti) launch_timed(extrema_greeks_kernel<true, 16>, dim3(grid), dim3(kBlock), c->stream, timed, pr, es, ws);
    else launch_timed(extrema_greeks_kernel<false, 16>, dim3(grid), dim3(kBlock), c->stream, timed, pr, es, ws);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    const int64_t n = n_paths * (anti ? 2 : 1);
    olmc_stats st[OLMC_MAX_BATCH];
    for (int i = 0; i < gs.k; ++i) {
        finish_stats(c->h_result[2 * i], c->h_result[2 * i + 1], n, gs.o[i].r, gs.o[i].T, &st[i]);
        if (poisoned(gs.o[i].S, gs.o[i].K, gs.o[i].T, gs.o[i].r, gs.o[i].sigma, gs.o[i].q) || std::isnan(barrier)) nan_stats(n, &st[i]);
    }
    gs.finish(st, T, out9, evals);
    return OLMC_OK;
}

// ======================================================= autocallable / cliquet ====
namespace {
template <typename Launch>
int run_structured(int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic, double r_for_discount,
                   double T, bool poisoned_inputs, olmc_stats* out, Launch launch, int nv = 2, double* raw_sums = nullptr) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const int32_t grid = grid_for(n_local);
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, nv, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    launch(grid, c->stream, timed, pr, ws);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    finish_stats(c->h_result[0], c->h_result[1], n_local * (antithetic ? 2 : 1), r_for_discount, T, out);
    if (raw_sums) for (int m = 0; m < nv; ++m) raw_sums[m] = c->h_result[m];     // the context is still leased
    if (poisoned_inputs) nan_stats(out->n, out);
    return OLMC_OK;
}
}  // namespace

extern "C" int olmc_autocallable(double S, double T, double r, double sigma, double q, double autocall_barrier,
                                 double coupon_barrier, double coupon_rate, double ki_barrier, int32_t observation_freq,
                                 int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic,
                                 olmc_stats* out) {
    if (observation_freq < 1) return fail(OLMC_ERR_ARG, "observation_freq must be >= 1");
    if (n_steps >= 1 && n_steps / observation_freq < 1) return fail(OLMC_ERR_ARG, "no observation date: observation_freq > n_steps");
    AutocallContract ac;
    const double dt = T / n_steps;
    ac.drift = (r - q - 0.5 * sigma * sigma) * dt;
    ac.vol = sigma * std::sqrt(dt);
    ac.log_autocall = log_level(autocall_barrier);
    ac.log_coupon = log_level(coupon_barrier);
    ac.log_ki = log_level(ki_barrier);
    ac.obs_freq = observation_freq;
    ac.n_obs = n_steps / observation_freq;                      // len(range(f, M + 1, f))
    ac.coupon_unit = coupon_rate * T / ac.n_obs;                // coupon_rate * ((i+1)/n_obs) * T, accrued per observation (:459-460)
    ac.final_coupon = coupon_rate * T;
    ac.obs_df = std::exp(-r * dt * observation_freq);           // exp(-r t dt) at t = k f, built up by products (:461)
    ac.final_df = std::exp(-r * T);
    const bool bad = poisoned(S, 1.0, T, r, sigma, q) || std::isnan(autocall_barrier + coupon_barrier + coupon_rate + ki_barrier);
    // payoffs are already discounted path by path (exotic_options.py:463, 489): no outer discount
    return run_structured(path_offset, n_local, n_steps, seed, antithetic, 0.0, T, bad, out,
                          [&](int32_t grid, hipStream_t st, const EventPair* timed, const PathRange& pr, const ReduceWs& ws) {
                              if (antithetic) launch_timed(autocall_kernel<true>, dim3(grid), dim3(kBlock), st, timed, pr, ac, ws);
                              else launch_timed(autocall_kernel<false>, dim3(grid), dim3(kBlock), st, timed, pr, ac, ws);
                          });
}

extern "C" int olmc_cliquet(double S, double T, double r, double sigma, double q, double local_cap, double local_floor,
                            double global_cap, double global_floor, int32_t n_periods, int64_t path_offset,
                            int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic, olmc_stats* out) {
    if (n_periods < 1 || (n_steps >= 1 && n_steps / n_periods < 1)) return fail(OLMC_ERR_ARG, "n_periods must be in [1, n_steps]");
    CliquetContract cc;
    const double dt = T / n_steps;
    cc.s0 = S;
    cc.drift = (r - q - 0.5 * sigma * sigma) * dt;
    cc.vol = sigma * std::sqrt(dt);
    cc.local_cap = local_cap; cc.local_floor = local_floor; cc.global_cap = global_cap; cc.global_floor = global_floor;
    cc.steps_per_period = n_steps / n_periods;                  // exotic_options.py:532
    cc.n_periods = n_periods;
    const bool bad = poisoned(S, 1.0, T, r, sigma, q) || std::isnan(local_cap + local_floor + global_cap + global_floor);
    return run_structured(path_offset, n_local, n_steps, seed, antithetic, r, T, bad, out,
                          [&](int32_t grid, hipStream_t st, const EventPair* timed, const PathRange& pr, const ReduceWs& ws) {
                              if (antithetic) launch_timed(cliquet_kernel<true>, dim3(grid), dim3(kBlock), st, timed, pr, cc, ws);
                              else launch_timed(cliquet_kernel<false>, dim3(grid), dim3(kBlock), st, timed, pr, cc, ws);
                          });
}

// ================================================================== full paths ====
extern "C" int olmc_gbm_paths(double S, double T, double r, double sigma, double q, int64_t n_paths, int32_t n_steps,
                              uint64_t seed, int path_major, double* out_host) {
    if (!out_host) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    const double bytes = 8.0 * static_cast<double>(n_paths) * (n_steps + 1.0);
    if (bytes > 64e9) return fail(OLMC_ERR_ARG, "path matrix would exceed 64 GB");
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    rc = bulk_reserve(c, static_cast<size_t>(bytes));
    if (rc) return rc;
    LsmContract lc{};
    const double dt = T / n_steps;                      // gbm_numpy.py:106-108
    lc.log_s0 = std::log(S);
    lc.s_first = S;
    lc.drift = (r - q - 0.5 * sigma * sigma) * dt;
    lc.vol = sigma * std::sqrt(dt);
    lc.n_steps = n_steps;
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    if (path_major) hipLaunchKernelGGL((lsm_paths_kernel<true>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, lc, static_cast<double*>(c->d_bulk));
    else hipLaunchKernelGGL((lsm_paths_kernel<false>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, lc, static_cast<double*>(c->d_bulk));
    HIP_TRY(hipGetLastError());
    return copy_to_host(c, out_host, c->d_bulk, static_cast<size_t>(bytes));
}

extern "C" int olmc_exercise_boundary(double S, double K, double T, double r, double sigma, double q, int is_call,
                                      int64_t n_paths, int32_t n_steps, uint64_t seed, double* boundary_host) {
    if (!boundary_host) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    const double bytes = 8.0 * static_cast<double>(n_paths) * (n_steps + 1.0);
    if (bytes > 64e9) return fail(OLMC_ERR_ARG, "path matrix would exceed 64 GB");
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t rows = static_cast<size_t>(n_steps) + 1;
    const size_t path_bytes = (static_cast<size_t>(bytes) + 255) / 256 * 256;
    rc = bulk_reserve(c, path_bytes + rows * sizeof(double));
    if (rc) return rc;
    double* d_paths = static_cast<double*>(c->d_bulk);
    double* d_boundary = reinterpret_cast<double*>(static_cast<char*>(c->d_bulk) + path_bytes);
    LsmContract lc{};
    const double dt = T / n_steps;                      // exotic_options.py:54-56
    lc.log_s0 = std::log(S);
    lc.s_first = std::exp(lc.log_s0);                   // :59-65: column 0 is exp(log S)
    lc.drift = (r - q - 0.5 * sigma * sigma) * dt;
    lc.vol = sigma * std::sqrt(dt);
    lc.n_steps = n_steps;
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    hipLaunchKernelGGL((lsm_paths_kernel<false>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, lc, d_paths);
    HIP_TRY(hipGetLastError());
    // np.percentile(x, 10) for a put, 90 for a call (:337-341); NumPy divides q by 100 first
    hipLaunchKernelGGL(exercise_boundary_kernel, dim3(static_cast<uint32_t>(rows)), dim3(kBoundaryThreads), 0, c->stream, d_paths, n_paths, K,
                       is_call ? 1.0 : -1.0, (is_call ? 90.0 : 10.0) / 100.0, d_boundary);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(boundary_host, d_boundary, rows * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OLMC_OK;
}

// ================================================================ American (LSM) ====
namespace {
// Mean and standard deviation of S_t / K over the in-the-money side of the strike under the model's lognormal law (truncated
// lognormal moments, closed form): the affine map that standardises the regressor of the date-t regression (LsmScale).
// Any finite, positive pair gives the same least-squares fit in exact arithmetic; this one keeps the normal equations well conditioned.
void lsm_regressor_scale(double S, double K, double r, double q, double sigma, double dt, int32_t t, bool is_call, double* centre, double* inv_width) {
    *centre = 1.0;
    *inv_width = 1.0;
    const double m = std::log(S) + (r - q - 0.5 * sigma * sigma) * dt * t, s = sigma * std::sqrt(dt * t), a = std::log(K);
    if (!(s > 0.0) || !std::isfinite(m) || !std::isfinite(a)) return;
    const double side = is_call ? 1.0 : -1.0;            // in the money: side * (ln S_t - ln K) > 0
    auto phi = [](double u) { return 0.5 * std::erfc(-u * 0.70710678118654752440); };
    const double p0 = phi(side * (m - a) / s);
    const double m1 = std::exp(m + 0.5 * s * s) * phi(side * (m + s * s - a) / s);
    const double m2 = std::exp(2.0 * m + 2.0 * s * s) * phi(side * (m + 2.0 * s * s - a) / s);
    if (!(p0 > 1e-280) || !std::isfinite(m1) || !std::isfinite(m2)) return;
    const double mean = m1 / p0, var = m2 / p0 - mean * mean;
    const double width = std::max(std::sqrt(std::max(var, 0.0)), 1e-6 * mean);
    if (!(mean > 0.0) || !std::isfinite(width) || !(width > 0.0)) return;
    *centre = mean / K;
    *inv_width = K / width;
}
}  // namespace

extern "C" int olmc_american_lsm(double S, double K, double T, double r, double sigma, double q, int is_call,
                                 int64_t n_paths, int32_t n_steps, int32_t poly_degree, uint64_t seed, olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    if (poly_degree < 1 || poly_degree > kLsmMaxDegree) return fail(OLMC_ERR_ARG, "poly_degree must be in [1, 4]");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    const double path_bytes = 8.0 * static_cast<double>(n_paths) * (n_steps + 2.0) + 8.0 * 2 * 1024 * kLsmNV;   // + two buffers of <= 1024 workgroup rows
    if (path_bytes > 64e9) return fail(OLMC_ERR_ARG, "path matrix would exceed 64 GB: lower n_paths or n_steps");
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    rc = bulk_reserve(c, static_cast<size_t>(path_bytes));
    if (rc) return rc;
    double* d_paths = static_cast<double*>(c->d_bulk);                                  // [n_steps + 1][n_paths]
    double* d_cash = d_paths + static_cast<size_t>(n_steps + 1) * n_paths;              // [n_paths]
    double* d_rows = d_cash + n_paths;                                                  // [2][grid][kLsmNV]: the regression sums a date hands to the next launch
    LsmContract lc;
    const double dt = T / n_steps;                      // exotic_options.py:54-56, 260-261
    lc.log_s0 = std::log(S);
    lc.s_first = std::exp(lc.log_s0);
    lc.drift = (r - q - 0.5 * sigma * sigma) * dt;
    lc.vol = sigma * std::sqrt(dt);
    lc.strike = K;
    lc.inv_strike = 1.0 / K;
    lc.sign = is_call ? 1.0 : -1.0;
    lc.discount = std::exp(-r * dt);
    lc.degree = poly_degree;
    lc.n_steps = n_steps;
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    const int32_t path_grid = grid_for(n_paths);
    // The per-date launches move 32 bytes per path and hand 16 sums per workgroup to the next launch, EVERY workgroup of which sums all
    // the rows: at most ONE workgroup per compute unit (<= 256 rows: one round trip), each thread taking its paths U at a time with all
    // 3 U loads in flight.  With one wave per SIMD the register count buys nothing, so U follows the paths a thread has (1, 2, 4, 8:
    // 98 ... 170 VGPRs).  Measured per call, 51 launches (profiles/r04_lsm_ab.jsonl): 1M x 50 -- 898 us at U = 1, 771 / 658 / 628 / 662 at
    // 2 / 4 / 8 / 16; two workgroups per CU 792 / 717 / 674 (U = 1 / 2 / 4), four 978 / 971 / 952 (a workgroup re-reads every row);
    // 200k x 50 -- 366 / 343 / 328 / 345 at U = 1 / 2 / 4 / 8; 50k x 50, where a thread has one path, 275 - 300 whatever U.
    const int32_t grid = std::min<int32_t>(path_grid, std::min<int32_t>(c->cus, 1024));
    const int64_t per_thread = (n_paths + static_cast<int64_t>(grid) * kBlock - 1) / (static_cast<int64_t>(grid) * kBlock);
    const int lsm_unroll = per_thread <= 1 ? 1 : per_thread <= 2 ? 2 : per_thread <= 4 ? 4 : 8;
    EventPair ep{};
    if (g_profile) { rc = prof_begin(c, c->stream, &ep); if (rc) return rc; }
    hipLaunchKernelGGL((lsm_paths_kernel<false>), dim3(path_grid), dim3(kBlock), 0, c->stream, pr, lc, d_paths);
    HIP_TRY(hipGetLastError());
    // every launch sums the rows the launch before it stored, fits the later date from them and stores its own rows: stream order is
    // the only synchronisation, the host waits once at the end.  Only the LAST launch (t_fit == 0: the moments of the time-0 cash
    // flow) goes through the grid reduction and writes into the pinned host buffer.
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, kLsmNV, c->d_result, -1.0, &ws);          // arms the completion word for the launch that writes d_result
    if (rc) return rc;
    std::vector<double> centre(static_cast<size_t>(n_steps) + 1, 1.0), inv_width(static_cast<size_t>(n_steps) + 1, 1.0);   // while the path kernel runs
    for (int32_t t = 1; t < n_steps; ++t) lsm_regressor_scale(S, K, r, q, sigma, dt, t, is_call != 0, &centre[t], &inv_width[t]);
    int32_t init = 1;
    for (int32_t t_fit = n_steps - 1; t_fit >= 0; --t_fit) {
        const bool first = init != 0, final_date = t_fit == 0;
        const LsmScale sc{centre[t_fit], inv_width[t_fit], centre[t_fit + 1], inv_width[t_fit + 1]};
        auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, c->stream, n_paths, lc, sc, d_rows, t_fit, d_paths, d_cash, ws); };
        auto pick = [&](auto u) {
            constexpr int U = decltype(u)::value;
            if (first && final_date) go(lsm_step_kernel<U, true, true>);
            else if (first) go(lsm_step_kernel<U, true, false>);
            else if (final_date) go(lsm_step_kernel<U, false, true>);
            else go(lsm_step_kernel<U, false, false>);
        };
        if (lsm_unroll == 1) pick(std::integral_constant<int, 1>{});
        else if (lsm_unroll == 2) pick(std::integral_constant<int, 2>{});
        else if (lsm_unroll == 4) pick(std::integral_constant<int, 4>{});
        else pick(std::integral_constant<int, 8>{});
        rc = after_launch(c, c->stream);
        if (rc) return rc;
        init = 0;
    }
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    if (g_profile) { rc = prof_end(c, c->stream, ep); if (rc) return rc; }
    // the time-0 cash flows are already discounted step by step (:302-304): no outer factor
    finish_stats(c->h_result[0], c->h_result[1], n_paths, 0.0, T, out);
    if (poisoned(S, K, T, r, sigma, q)) nan_stats(n_paths, out);
    return OLMC_OK;
}

// ===================================================================== Heston ====
namespace {
HestonContract make_heston(double S, double K, double T, double r, double q, int is_call, double kappa, double theta,
                           double sigma_v, double rho, double v0, int32_t n_steps) {
    HestonContract hc;
    const double dt = T / n_steps;                     // heston.py:218-219
    hc.log_s0 = std::log(S);
    hc.v0 = v0;
    hc.mu_dt = (r - q) * dt;
    hc.dt = dt;
    hc.sqrt_dt = std::sqrt(dt);
    hc.kappa_dt = kappa * dt;
    hc.theta = theta;
    hc.sigma_v = sigma_v;
    hc.rho = rho;
    hc.rho_c = std::sqrt(1 - rho * rho);               // :228
    hc.strike = K;
    hc.sign = is_call ? 1.0 : -1.0;
    return hc;
}
}  // namespace

extern "C" int olmc_heston_paths(double S, double T, double r, double q, double kappa, double theta, double sigma_v, double rho,
                                 double v0, int64_t n_paths, int32_t n_steps, uint64_t seed, int path_major, double* spot_host,
                                 double* var_host) {
    if (!spot_host || !var_host) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(rho >= -1.0 && rho <= 1.0)) return fail(OLMC_ERR_ARG, "rho must be in [-1, 1]");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    const double bytes = 8.0 * static_cast<double>(n_paths) * (n_steps + 1.0);
    if (2 * bytes > 64e9) return fail(OLMC_ERR_ARG, "path matrices would exceed 64 GB");
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    rc = bulk_reserve(c, 2 * static_cast<size_t>(bytes));
    if (rc) return rc;
    double* d_spot = static_cast<double*>(c->d_bulk);
    double* d_var = d_spot + static_cast<size_t>(n_paths) * (n_steps + 1);
    const HestonContract hc = make_heston(S, 0.0, T, r, q, 1, kappa, theta, sigma_v, rho, v0, n_steps);
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    if (path_major) hipLaunchKernelGGL((heston_paths_kernel<true>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, hc, S, d_spot, d_var);
    else hipLaunchKernelGGL((heston_paths_kernel<false>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, hc, S, d_spot, d_var);
    HIP_TRY(hipGetLastError());
    rc = copy_to_host(c, spot_host, d_spot, static_cast<size_t>(bytes));
    if (rc) return rc;
    return copy_to_host(c, var_host, d_var, static_cast<size_t>(bytes));
}

extern "C" int olmc_heston(double S, double K, double T, double r, double q, int is_call, double kappa, double theta,
                           double sigma_v, double rho, double v0, int64_t path_offset, int64_t n_local, int32_t n_steps,
                           uint64_t seed, int antithetic, olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(rho >= -1.0 && rho <= 1.0)) return fail(OLMC_ERR_ARG, "rho must be in [-1, 1]");
    int rc = check_paths(path_offset, n_local, n_steps);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const int32_t grid = grid_for(n_local);
    const HestonContract hc = make_heston(S, K, T, r, q, is_call, kappa, theta, sigma_v, rho, v0, n_steps);
    ReduceWs ws;
    rc = make_ws(c, c->stream, grid, 2, c->d_result, -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    if (antithetic) launch_timed(heston_kernel<true>, dim3(grid), dim3(kBlock), c->stream, timed, pr, hc, ws);
    else launch_timed(heston_kernel<false>, dim3(grid), dim3(kBlock), c->stream, timed, pr, hc, ws);
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    finish_stats(c->h_result[0], c->h_result[1], n_local * (antithetic ? 2 : 1), r, T, out);
    if (poisoned(S, K, T, r, 0.0, q) || std::isnan(kappa + theta + sigma_v + rho + v0)) nan_stats(out->n, out);
    return OLMC_OK;
}

// ============================================================== jump diffusion ====
namespace {
int make_jump(double S, double K, double T, double r, double sigma, double q, int is_call, int model, double lambda_j,
              double a1, double a2, double a3, int32_t n_steps, JumpContract* out) {
    if (model != OLMC_JUMP_MERTON && model != OLMC_JUMP_KOU) return fail(OLMC_ERR_ARG, "bad jump model");
    if (!(lambda_j >= 0.0)) return fail(OLMC_ERR_ARG, "lambda_j must be non-negative");
    if (n_steps < 1) return fail(OLMC_ERR_ARG, "n_steps must be >= 1");
    JumpContract jc;
    const double dt = T / n_steps;
    double kappa;                                            // E[e^Y - 1]
    if (model == OLMC_JUMP_MERTON) {
        jc.mu_j = a1; jc.sigma_j = a2;
        jc.kou_p = 0.0; jc.inv_eta1 = 0.0; jc.inv_eta2 = 0.0;
        kappa = std::exp(a1 + 0.5 * a2 * a2) - 1;            // jump_diffusion.py:64-67
    } else {
        jc.mu_j = 0.0; jc.sigma_j = 0.0;
        jc.kou_p = a1; jc.inv_eta1 = 1.0 / a2; jc.inv_eta2 = 1.0 / a3;
        kappa = a1 * a2 / (a2 - 1) + (1 - a1) * a3 / (a3 + 1) - 1;   // :293-299
    }
    jc.kou = model == OLMC_JUMP_KOU;
    jc.pad = 0;
    jc.log_s0 = std::log(S);
    jc.drift = (r - q - lambda_j * kappa - 0.5 * sigma * sigma) * dt;   // :207 / :346 compensated drift
    jc.vol = sigma * std::sqrt(dt);
    jc.strike = K;
    jc.sign = is_call ? 1.0 : -1.0;
    jc.lam_dt = lambda_j * dt;
    // the device draws the jump count by inversion from p0 = exp(-lambda dt), at most kJumpCap = 64 jumps per step:
    // exact to 1e-15 of probability mass up to lambda dt = 20, silently truncated beyond (and p0 underflows at 745).
    // The reference's np.random.poisson has no such limit, so larger rates are refused rather than mispriced.
    if (jc.lam_dt > 20.0) return fail(OLMC_ERR_ARG, "lambda_j * T / n_steps must be <= 20 jumps per step: raise n_steps");
    jc.p0 = std::exp(-jc.lam_dt);
    *out = jc;
    return OLMC_OK;
}
}  // namespace

extern "C" int olmc_jump_diffusion(double S, double K, double T, double r, double sigma, double q, int is_call, int model,
                                   double lambda_j, double a1, double a2, double a3, int64_t path_offset, int64_t n_local,
                                   int32_t n_steps, uint64_t seed, olmc_stats* out) {
    JumpContract jc;
    int rc = make_jump(S, K, T, r, sigma, q, is_call, model, lambda_j, a1, a2, a3, n_steps, &jc);
    if (rc) return rc;
    const bool bad = poisoned(S, K, T, r, sigma, q) || std::isnan(lambda_j + a1 + a2 + a3);
    return run_structured(path_offset, n_local, n_steps, seed, 0, r, T, bad, out,
                          [&](int32_t grid, hipStream_t st, const EventPair* timed, const PathRange& pr, const ReduceWs& ws) {
                              launch_timed(jump_kernel, dim3(grid), dim3(kBlock), st, timed, pr, jc, ws);
                          });
}

extern "C" int olmc_jump_paths(double S, double T, double r, double sigma, double q, int model, double lambda_j, double a1,
                               double a2, double a3, int64_t n_paths, int32_t n_steps, uint64_t seed, int path_major,
                               double* out_host) {
    if (!out_host) return fail(OLMC_ERR_ARG, "null pointer");
    JumpContract jc;
    int rc = make_jump(S, 0.0, T, r, sigma, q, 1, model, lambda_j, a1, a2, a3, n_steps, &jc);
    if (rc) return rc;
    rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    const double bytes = 8.0 * static_cast<double>(n_paths) * (n_steps + 1.0);
    if (bytes > 64e9) return fail(OLMC_ERR_ARG, "path matrix would exceed 64 GB");
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    rc = bulk_reserve(c, static_cast<size_t>(bytes));
    if (rc) return rc;
    const PathRange pr = make_range(0, n_paths, n_steps, seed);
    if (path_major) hipLaunchKernelGGL((jump_paths_kernel<true>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, jc, S,
                                       static_cast<double*>(c->d_bulk));
    else hipLaunchKernelGGL((jump_paths_kernel<false>), dim3(grid_for(n_paths)), dim3(kBlock), 0, c->stream, pr, jc, S,
                            static_cast<double*>(c->d_bulk));
    HIP_TRY(hipGetLastError());
    return copy_to_host(c, out_host, c->d_bulk, static_cast<size_t>(bytes));
}

// ======================================================================= QMC ====
namespace {
// The scrambled direction matrix and digital shift on the device: [dims x 30 | dims] words.  The table travels only when it
// differs from the one already there (compared word for word: 31 KB at 252 dims, ~1 us, against two pageable uploads): the
// 8 / 14 pricings of literal FD Greeks and every repeated pricing share one upload.  The caller holds the context's lease.
int qmc_table(DeviceCtx* c, const uint32_t* sv, const uint32_t* shift, int32_t dims, bool* uploaded = nullptr) {
    if (uploaded) *uploaded = false;
    const size_t sv_words = static_cast<size_t>(dims) * kSobolBits, table_words = sv_words + dims;
    const bool same = c->sobol_host.size() == table_words && std::memcmp(c->sobol_host.data(), sv, sizeof(uint32_t) * sv_words) == 0 &&
                      std::memcmp(c->sobol_host.data() + sv_words, shift, sizeof(uint32_t) * dims) == 0;
    if (same) return OLMC_OK;
    c->sobol_host.clear();                       // whatever happens below, the device copy is no longer described by it
    if (table_words > c->sobol_words) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_sobol) HIP_TRY(hipFree(c->d_sobol));
        c->d_sobol = nullptr;
        c->sobol_words = 0;
        HIP_TRY(hipMalloc(&c->d_sobol, sizeof(uint32_t) * table_words));
        c->sobol_words = table_words;
    }
    HIP_TRY(hipMemcpyAsync(c->d_sobol, sv, sizeof(uint32_t) * sv_words, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_sobol + sv_words, shift, sizeof(uint32_t) * dims, hipMemcpyHostToDevice, c->stream));
    c->sobol_host.assign(sv, sv + sv_words);
    c->sobol_host.insert(c->sobol_host.end(), shift, shift + dims);
    if (uploaded) *uploaded = true;
    return OLMC_OK;
}

int qmc_check(const uint32_t* sv, const uint32_t* shift, int32_t bits, int32_t dims, int64_t point_offset, int64_t n_paths) {
    if (!sv || !shift) return fail(OLMC_ERR_ARG, "null pointer");
    if (bits != kSobolBits) return fail(OLMC_ERR_ARG, "only 30-bit Sobol tables (SciPy's default) are supported");
    if (dims < 1 || dims > 21201) return fail(OLMC_ERR_ARG, "dims must be in [1, 21201]");
    int rc = check_paths(point_offset, n_paths, dims);
    if (rc) return rc;
    if (point_offset + n_paths > (int64_t(1) << kSobolBits)) return fail(OLMC_ERR_ARG, "at most 2**30 Sobol points");
    return OLMC_OK;
}

// gbm_qmc.py:38-44 as an olmc_option -> Contract: dt = T / dims, a = ln S + drift dims, vol = sigma sqrt(dt) = make_contract(o, dims)
// Launch shape of a Sobol kernel.  A workgroup takes 64 points and each of its four waves a quarter of the dimensions (from 16
// dimensions on; round 4 drew that line at 2^18 points and ran one point per thread in between); from 2^22 points on a thread takes
// an aligned block of eight consecutive points instead -- from 2^21 below 128 dimensions, 2^20 below 64, 2^19 below 32: a split wave
// has its fixed costs (masks, coefficients, the fold's prologue) to spread over a quarter of the dimensions.  Round 5 re-measured the
// crossover as the aligned split kernel got faster (profiles/r05_ab_kernels.txt, profiles/r05_qmc_grid.txt; split / eight, us):
// 2^19 x 16: 36 / 32, 2^20 x 16: 65 / 48, 2^19 x 32: 52 / 52, 2^20 x 32: 97 / 82, 2^19 x 48: 68 / 73, 2^20 x 48: 128 / 117, 2^20 x 64:
// 158 / 171, 2^21 x 64: 310 / 293, 2^20 x 252: 530 / 638, 2^21 x 252: 1050 / 1107, 2^22 x 252: 2106 / 2074.  OLMC_TUNE_QMC_BLOCK:
// 1 = always eight points per thread, 2 = always split, -1 = never eight and never split (one point per thread).
struct QmcShape {
    bool blocks, split, aligned, aligned8;
    int64_t units;       // threads' worth of work: blocks of eight, or points
    int32_t grid;
};
QmcShape qmc_shape(int64_t point_offset, int64_t n_paths, int32_t dims) {
    QmcShape sh;
    // split workgroups whose 64 lanes are a 64-ALIGNED block of points fold the high Gray bits' direction numbers once per wave and
    // dimension and keep the inverse normal's coefficients in registers (olmc_kernels.h qmc_point_sum<true>).  The coefficients'
    // load and the fold's prologue are a fixed cost per wave: it pays from 32 dimensions on (8 per wave; 2^17 x 32: 21.5 -> 19.9 us,
    // 2^18 x 48: 44.6 -> 38.4, 2^14 x 32: even; 16 dimensions: 8.6 -> 8.9, 12.0 -> 12.3).  The one-point form is left to launches of
    // fewer than 16 dimensions and to the knob
    sh.aligned = (point_offset & 63) == 0 && dims >= 32;
    sh.aligned8 = (point_offset & 511) == 0;        // eight points per thread: a wave's 64 blocks start at a multiple of 512 points
    const int blocks_from_log2 = dims < 32 ? 19 : dims < 64 ? 20 : dims < 128 ? 21 : 22;      // fewer dimensions: less for a split wave to spread its fixed costs over
    sh.blocks = g_qmc_block == 1 ? true : (g_qmc_block != 0 ? false : n_paths >= (int64_t(1) << blocks_from_log2));
    sh.split = !sh.blocks && (g_qmc_block == 0 || g_qmc_block == 2) && dims >= 16;
    sh.units = sh.blocks ? (point_offset + n_paths + kQmcBlock - 1) / kQmcBlock - point_offset / kQmcBlock : n_paths;
    sh.grid = sh.split ? static_cast<int32_t>((n_paths + kWave - 1) / kWave) : grid_for(sh.units);
    return sh;
}

int run_qmc(double S, double K, double T, double r, double sigma, double q, int is_call, int64_t point_offset,
            int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
            olmc_stats* out, double* terminal_host, int mirror = 0, olmc_cv_moments* cv = nullptr,
            double* d_triple = nullptr /* a shard of a multi-GPU call: {sum, sumsq, n} are LEFT here, on `shard_stream`, nothing is waited for */,
            hipStream_t shard_stream = nullptr, bool shard_cv = false /* with d_triple: the five control-variate moments and n instead */) {
    int rc = qmc_check(sv, shift, bits, dims, point_offset, n_paths);
    if (rc) return rc;
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t sv_words = static_cast<size_t>(dims) * kSobolBits;
    const size_t term_bytes = terminal_host ? sizeof(double) * static_cast<size_t>(n_paths) * (mirror ? 2 : 1) : 0;
    if (term_bytes) {
        rc = bulk_reserve(c, term_bytes);
        if (rc) return rc;
    }
    bool uploaded = false;
    rc = qmc_table(c, sv, shift, dims, &uploaded);
    if (rc) return rc;
    if (d_triple && uploaded) HIP_TRY(hipStreamSynchronize(c->stream));      // the table travelled on the context's stream, the kernel runs on the rank's
    uint32_t* d_sv = c->d_sobol;
    uint32_t* d_shift = d_sv + sv_words;
    double* d_term = terminal_host ? static_cast<double*>(c->d_bulk) : nullptr;
    // gbm_qmc.py:38-44
    const double dt = T / dims;
    const double drift = (r - q - 0.5 * sigma * sigma) * dt;
    Contract ct;
    ct.vol = sigma * std::sqrt(dt);
    // gbm_qmc.py:44 (drift * steps) vs :70 (the antithetic variant multiplies the rate by T directly)
    ct.a = mirror ? std::log(S) + (r - q - 0.5 * sigma * sigma) * T : std::log(S) + drift * dims;
    ct.strike = K;
    ct.sign = is_call ? 1.0 : -1.0;
    ct.scale = 1.0;
    ct.neg_sign_strike = -ct.sign * K;
    ct.sign_scale = ct.sign;
    QmcRange qr;
    qr.first = static_cast<uint64_t>(point_offset);
    qr.count = n_paths;
    qr.dims = dims;
    qr.mirror = mirror ? 1 : 0;
    const QmcShape sh = qmc_shape(point_offset, n_paths, dims);
    const bool blocks = sh.blocks;
    const int32_t grid = sh.grid;
    ReduceWs ws{};
    EventPair ep{};
    const EventPair* timed = nullptr;
    double* const no_terminal = nullptr;
    if (cv || (d_triple && shard_cv)) {
        hipStream_t s = d_triple ? shard_stream : c->stream;
        rc = make_ws(c, s, grid, 5, d_triple ? d_triple : c->d_result, d_triple ? static_cast<double>(n_paths) : -1.0, &ws);
        if (rc) return rc;
        rc = prof_pair(c, &ep, &timed);
        if (rc) return rc;
        if (blocks && sh.aligned8) launch_timed(european_qmc_block_kernel<kControlVariate, true>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else if (blocks) launch_timed(european_qmc_block_kernel<kControlVariate, false>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else if (sh.split && sh.aligned) launch_timed(european_qmc_kernel<kControlVariate, true, true>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else if (sh.split) launch_timed(european_qmc_kernel<kControlVariate, true>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else launch_timed(european_qmc_kernel<kControlVariate, false>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        if (d_triple) return after_launch(c, s);
    } else if (!terminal_host) {
        hipStream_t s = d_triple ? shard_stream : c->stream;
        rc = make_ws(c, s, grid, 2, d_triple ? d_triple : c->d_result, d_triple ? static_cast<double>(n_paths) : -1.0, &ws);
        if (rc) return rc;
        rc = prof_pair(c, &ep, &timed);
        if (rc) return rc;
        if (blocks && sh.aligned8) launch_timed(european_qmc_block_kernel<kReduce, true>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else if (blocks) launch_timed(european_qmc_block_kernel<kReduce, false>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else if (sh.split && sh.aligned) launch_timed(european_qmc_kernel<kReduce, true, true>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else if (sh.split) launch_timed(european_qmc_kernel<kReduce, true, false>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        else launch_timed(european_qmc_kernel<kReduce, false, false>, dim3(grid), dim3(kBlock), s, timed, qr, ct, d_sv, d_shift, ws, no_terminal);
        if (d_triple) return after_launch(c, s);
    } else {
        if (blocks && sh.aligned8) hipLaunchKernelGGL((european_qmc_block_kernel<kTerminal, true>), dim3(grid), dim3(kBlock), 0, c->stream, qr, ct, d_sv, d_shift, ws, d_term);
        else if (blocks) hipLaunchKernelGGL((european_qmc_block_kernel<kTerminal, false>), dim3(grid), dim3(kBlock), 0, c->stream, qr, ct, d_sv, d_shift, ws, d_term);
        else if (sh.split && sh.aligned) hipLaunchKernelGGL((european_qmc_kernel<kTerminal, true, true>), dim3(grid), dim3(kBlock), 0, c->stream, qr, ct, d_sv, d_shift, ws, d_term);
        else if (sh.split) hipLaunchKernelGGL((european_qmc_kernel<kTerminal, true>), dim3(grid), dim3(kBlock), 0, c->stream, qr, ct, d_sv, d_shift, ws, d_term);
        else hipLaunchKernelGGL((european_qmc_kernel<kTerminal, false>), dim3(grid), dim3(kBlock), 0, c->stream, qr, ct, d_sv, d_shift, ws, d_term);
    }
    if (terminal_host) {            // no reduction workspace was handed out: only the launch status matters
        HIP_TRY(hipGetLastError());
        c->armed = 0;
        return copy_to_host(c, terminal_host, d_term, term_bytes);
    }
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    if (cv) {       // device moments are of the UNdiscounted payoff x; d = disc * x (monte_carlo.py:175)
        cv_from_device(c->h_result, n_paths, S, T, r, q, cv);
        if (poisoned(S, K, T, r, sigma, q)) cv->value = std::nan("");
        return OLMC_OK;
    }
    finish_stats(c->h_result[0], c->h_result[1], n_paths, r, T, out);
    if (poisoned(S, K, T, r, sigma, q)) nan_stats(n_paths, out);
    return OLMC_OK;
}
}  // namespace

extern "C" int olmc_european_qmc_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                                    int64_t point_offset, int64_t n_paths, int32_t dims, const uint32_t* sv,
                                    const uint32_t* shift, int32_t bits, olmc_cv_moments* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    return run_qmc(S, K, T, r, sigma, q, is_call, point_offset, n_paths, dims, sv, shift, bits, nullptr, nullptr, 0, out);
}

extern "C" int olmc_european_qmc(double S, double K, double T, double r, double sigma, double q, int is_call,
                                 int64_t point_offset, int64_t n_paths, int32_t dims, const uint32_t* sv,
                                 const uint32_t* shift, int32_t bits, olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    return run_qmc(S, K, T, r, sigma, q, is_call, point_offset, n_paths, dims, sv, shift, bits, out, nullptr);
}

extern "C" int olmc_european_qmc_terminal(double S, double T, double r, double sigma, double q, int64_t point_offset,
                                          int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift,
                                          int32_t bits, int antithetic, double* out_host) {
    if (!out_host) return fail(OLMC_ERR_ARG, "null pointer");
    return run_qmc(S, 0.0, T, r, sigma, q, 1, point_offset, n_paths, dims, sv, shift, bits, nullptr, out_host, antithetic);
}

namespace {
// k contracts on the same Sobol points, ONE launch (european_qmc_batch_kernel); falls back to k launches beyond the size one
// grid covers.  out[i] = stats of opts[i].
int run_qmc_batch(const olmc_option* opts, int32_t k, int64_t point_offset, int64_t n_paths, int32_t dims, const uint32_t* sv,
                  const uint32_t* shift, int32_t bits, olmc_stats* out,
                  double* d_sums = nullptr /* a shard of a multi-GPU call: the 2 nsets sums and n are LEFT here, on `shard_stream`, nothing is waited for */,
                  hipStream_t shard_stream = nullptr, int* pos_out = nullptr /* with d_sums: slot of contract i */) {
    if (!opts || (!out && !d_sums)) return fail(OLMC_ERR_ARG, "null pointer");
    if (k < 1 || k > OLMC_MAX_BATCH) return fail(OLMC_ERR_ARG, "batch size must be in [1, OLMC_MAX_BATCH]");
    int rc = qmc_check(sv, shift, bits, dims, point_offset, n_paths);
    if (rc) return rc;
    const QmcShape sh = qmc_shape(point_offset, n_paths, dims);
    const bool blocks = sh.blocks;
    const int64_t units = sh.units;
    if (d_sums && (k < 2 || (units + kBlock - 1) / kBlock > kMaxGrid))
        return fail(OLMC_ERR_ARG, "a shard of fused Sobol contracts needs 2 .. 16 contracts and points one grid covers");
    if (k == 1 || (units + kBlock - 1) / kBlock > kMaxGrid) {             // one contract, or more points than a grid covers: literal launches
        for (int i = 0; i < k; ++i) {
            rc = run_qmc(opts[i].S, opts[i].K, opts[i].T, opts[i].r, opts[i].sigma, opts[i].q, opts[i].is_call, point_offset, n_paths, dims, sv, shift,
                         bits, &out[i], nullptr);
            if (rc) return rc;
        }
        return OLMC_OK;
    }
    CtxLease lease;
    rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    bool uploaded = false;
    rc = qmc_table(c, sv, shift, dims, &uploaded);
    if (rc) return rc;
    if (d_sums && uploaded) HIP_TRY(hipStreamSynchronize(c->stream));        // the table travelled on the context's stream, the kernel runs on the rank's
    hipStream_t const st = d_sums ? shard_stream : c->stream;
    const size_t sv_words = static_cast<size_t>(dims) * kSobolBits;
    uint32_t* d_sv = c->d_sobol;
    uint32_t* d_shift = d_sv + sv_words;
    QmcRange qr;
    qr.first = static_cast<uint64_t>(point_offset);
    qr.count = n_paths;
    qr.dims = dims;
    qr.mirror = 0;
    const int32_t grid = sh.split ? sh.grid : static_cast<int32_t>((units + kBlock - 1) / kBlock);     // the grid covers every point / block
    const int nsets = k <= 8 ? 8 : 16;
    ReduceWs ws;
    rc = make_ws(c, st, grid, 2 * nsets, d_sums ? d_sums : c->d_result, d_sums ? static_cast<double>(n_paths) : -1.0, &ws);
    if (rc) return rc;
    EventPair ep{};
    const EventPair* timed = nullptr;
    rc = prof_pair(c, &ep, &timed);
    if (rc) return rc;
    int pos[OLMC_MAX_BATCH];
    // make_contract(o, dims) IS gbm_qmc.py:38-44: dt = T / dims, drift * dims, sigma sqrt(dt)
    if (nsets == 8) {
        ContractSet<8> cs;
        group_contracts<8>(opts, k, dims, &cs, pos);
        if (blocks && sh.aligned8) launch_timed(european_qmc_batch_kernel<8, true, false, true>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else if (blocks) launch_timed(european_qmc_batch_kernel<8, true, false>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else if (sh.split && sh.aligned) launch_timed(european_qmc_batch_kernel<8, false, true, true>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else if (sh.split) launch_timed(european_qmc_batch_kernel<8, false, true>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else launch_timed(european_qmc_batch_kernel<8, false, false>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
    } else {
        ContractSet<16> cs;
        group_contracts<16>(opts, k, dims, &cs, pos);
        if (blocks && sh.aligned8) launch_timed(european_qmc_batch_kernel<16, true, false, true>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else if (blocks) launch_timed(european_qmc_batch_kernel<16, true, false>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else if (sh.split && sh.aligned) launch_timed(european_qmc_batch_kernel<16, false, true, true>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else if (sh.split) launch_timed(european_qmc_batch_kernel<16, false, true>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
        else launch_timed(european_qmc_batch_kernel<16, false, false>, dim3(grid), dim3(kBlock), st, timed, qr, cs, d_sv, d_shift, ws);
    }
    if (d_sums) {
        if (pos_out) std::copy(pos, pos + k, pos_out);
        return after_launch(c, st);
    }
    rc = after_launch(c, c->stream);
    if (rc) return rc;
    rc = sync_or_recover(c, c->stream);
    if (rc) return rc;
    for (int i = 0; i < k; ++i) {
        if (poisoned(opts[i].S, opts[i].K, opts[i].T, opts[i].r, opts[i].sigma, opts[i].q)) nan_stats(n_paths, &out[i]);
        else finish_stats(c->h_result[2 * pos[i]], c->h_result[2 * pos[i] + 1], n_paths, opts[i].r, opts[i].T, &out[i]);
    }
    return OLMC_OK;
}
}  // namespace

extern "C" int olmc_european_qmc_batch(const olmc_option* opts, int32_t k, int64_t point_offset, int64_t n_paths, int32_t dims,
                                       const uint32_t* sv, const uint32_t* shift, int32_t bits, olmc_stats* out) {
    return run_qmc_batch(opts, k, point_offset, n_paths, dims, sv, shift, bits, out);
}

extern "C" int olmc_european_qmc_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call, int64_t n_paths,
                                           int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits, int second_order,
                                           double* out9, olmc_stats* evals) {
    if (!out9) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(T > 0.0)) return fail(OLMC_ERR_ARG, "T must be > 0 (price() returns intrinsic value without simulating)");
    const GreeksSet gs(S, K, T, r, sigma, q, is_call, second_order);
    olmc_stats st[OLMC_MAX_BATCH];
    int rc = run_qmc_batch(gs.o, gs.k, 0, n_paths, dims, sv, shift, bits, st);
    if (rc) return rc;
    gs.finish(st, T, out9, evals);
    return OLMC_OK;
}

// ======================================================== multi-GPU (RCCL) ====
// librccl is resolved lazily (dlopen) so single-GPU users never load it; the TYPES and ENUM VALUES come from
// <rccl/rccl.h> at compile time, so a header / library mismatch is a build-time matter, not a guessed constant.
//
// One process, n ranks (rank d = device d).  An ENGINE exists per list of devices: per rank a stream, a send / receive buffer and a
// LAUNCHER THREAD bound to the rank's device at birth (one hipSetDevice, ever -- SURVEY §8e: "one host thread per device +
// ncclCommInitAll"), plus the list's communicators.  Never a pricing context: a shard launch leases one like any other call and uses
// the rank's stream as a caller stream.  A call
//   1. posts the launch job: every launcher queues its rank's path kernel at once (round 4 queued them one after the other from the
//      calling thread: rank 7's kernel started seven enqueues late);
//   2. queues ONE grouped all-reduce of `count` doubles from the calling thread (3 for a price, 2 nsets + 1 for finite-difference
//      Greeks, 6 for the control variate: SURVEY §8e) -- only after every rank has launched, so a rank that failed leaves no peer
//      waiting inside a collective;
//   3. posts the drain job (launchers 1 .. n-1 wait for their streams) and meanwhile takes the reduced values home through rank 0's
//      polled completion word (olmc_fetch_dev); every rank holds the same values.
// Calls on device lists that share no device run concurrently; lists that share one queue behind that device's mutex (two
// communicators driven at once on one device may deadlock inside RCCL).  OLMC_TUNE_MULTI_LAUNCH = -1 keeps round 4's serial form
// (everything from the calling thread) for A/B; both forms give the same bits.
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return OLMC_OK;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(OLMC_ERR_RCCL, std::string("cannot load librccl: ") + dlerror());
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(h, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.CommInitAll || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.CommDestroy) {
        dlclose(h);
        return fail(OLMC_ERR_RCCL, "librccl is missing required symbols");
    }
    g_rccl.lib = h;
    return OLMC_OK;
}

std::string rccl_message(const char* what, ncclResult_t r) {
    return std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
}

#define RCCL_TRY(expr)                                                                \
    do {                                                                              \
        ncclResult_t r_ = (expr);                                                     \
        if (r_ != ncclSuccess) return fail(OLMC_ERR_RCCL, rccl_message(#expr, r_));   \
    } while (0)

constexpr int kMultiValues = kMaxNV + 1;            // the widest payload: 2 x 16 sums + n

struct MultiEngine;

struct MultiRank {
    int device = -1;
    hipStream_t stream = nullptr;
    double* d_send = nullptr;                       // [kMultiValues] this rank's sums (written by its path kernel's last workgroup)
    double* d_recv = nullptr;                       // [kMultiValues] the reduced sums
    hipEvent_t queued = nullptr;                    // rehearsal only: the rank's path kernel is queued behind this
    std::thread launcher;                           // bound to `device`; runs the engine's jobs for this rank
    int rc = OLMC_OK;                               // of the job it ran last (read by the caller behind the job's completion count)
    std::string error;
    std::chrono::steady_clock::time_point began, ended;     // of that job, on the launcher's clock (olmc_multi_gpu_spans: wake latency, own time)
};

struct MultiEngine {
    std::vector<MultiRank> ranks;                   // sized once, before the launchers start
    std::vector<int> devices;                       // device of every rank: what `ranks` and `comms` were built for
    std::vector<ncclComm_t> comms;                  // empty in a rehearsal
    bool rehearsal = false;
    JobBoard board;                                 // the hand-over between the calling thread and the launchers (olmc_job_board.h)
};

std::mutex g_engines_mu;                            // guards g_engines (never held while a call runs)
std::vector<MultiEngine*> g_engines;                // one per (device list, rehearsal) asked for since the last olmc_shutdown
std::mutex g_multi_dev_mu[kMaxDevices];             // a multi-GPU call holds the mutex of every device of its list (ascending order)
thread_local double t_multi_spans[8] = {0, 0, 0, 0, 0, 0, 0, 0};

void launcher_main(MultiEngine* e, int d, uint32_t seen) {
    MultiRank& rk = e->ranks[d];
    (void)hipSetDevice(rk.device);                  // once: every launch of this thread goes to this device
    t_device = rk.device;
    using clock = std::chrono::steady_clock;
    while (const std::function<int(int)>* work = board_next(e->board, d, seen)) {
        rk.began = clock::now();
        rk.rc = (*work)(d);
        if (rk.rc) rk.error = t_error;
        rk.ended = clock::now();
        board_done(e->board);
    }
}

void engine_post(MultiEngine* e, const std::function<int(int)>* work, int first_rank) { board_post(e->board, work, first_rank); }

// Waits for the posted job; the first failing rank's status and message become the caller's.
int engine_wait(MultiEngine* e) {
    board_wait(e->board);
    for (size_t d = static_cast<size_t>(e->board.first_rank); d < e->ranks.size(); ++d)
        if (e->ranks[d].rc) return fail(e->ranks[d].rc, "rank " + std::to_string(d) + ": " + e->ranks[d].error);
    return OLMC_OK;
}

// A rank stream is about to be destroyed: the workspace slots it claimed in the contexts of its device are handed back (a later
// stream may get the same handle value; and a slot whose owner is gone would otherwise stay taken for good -- the host fault of round 4,
// with every slot claimed by a dead stream, the next caller stream went down the slot-sharing path, which drained the
// slot's "owner" with hipStreamSynchronize on a destroyed handle).
void pool_forget_stream(int dev, hipStream_t stream) {
    DevicePool* pool = g_pool[dev].load(std::memory_order_acquire);
    if (!pool) return;
    // context by context: each is taken out of circulation for the moment its slots are looked at (a wait bounded by one call's
    // duration).  Waiting for ALL contexts to be idle at once could starve behind two busy callers that never pause together.
    for (size_t i = 0;; ++i) {
        DeviceCtx* c = nullptr;
        {
            std::unique_lock<std::mutex> lock(pool->mu);
            if (i >= pool->all.size()) break;
            c = pool->all[i];
            pool->idle.wait(lock, [&] { return !c->busy; });
            c->busy = true;
        }
        for (int k = 0; k < DeviceCtx::kSlots; ++k) {
            DeviceCtx::WsSlot& sl = c->slots[k];
            if (sl.claimed && !sl.shared && sl.owner == stream) { sl.claimed = false; sl.owner = nullptr; sl.used = false; }
        }
        {
            std::lock_guard<std::mutex> lock(pool->mu);
            c->busy = false;
        }
        pool->idle.notify_all();                    // lease waiters share the condition variable with this wait
    }
}

void engine_destroy(MultiEngine* e) {
    if (std::any_of(e->ranks.begin(), e->ranks.end(), [](const MultiRank& rk) { return rk.launcher.joinable(); })) {
        engine_post(e, nullptr, 0);                 // work == nullptr: leave
        for (MultiRank& rk : e->ranks)
            if (rk.launcher.joinable()) rk.launcher.join();
    }
    for (ncclComm_t cm : e->comms)
        if (cm && g_rccl.CommDestroy) g_rccl.CommDestroy(cm);
    for (MultiRank& rk : e->ranks) {
        if (rk.device < 0 || hipSetDevice(rk.device) != hipSuccess) continue;
        if (rk.stream) { (void)hipStreamSynchronize(rk.stream); pool_forget_stream(rk.device, rk.stream); (void)hipStreamDestroy(rk.stream); }
        if (rk.d_send) (void)hipFree(rk.d_send);
        if (rk.queued) (void)hipEventDestroy(rk.queued);
    }
    (void)hipGetLastError();
    delete e;
}

// olmc_shutdown: the caller's contract is that no call is in flight.
void multi_gpu_release() {
    std::vector<MultiEngine*> all;
    {
        std::lock_guard<std::mutex> lock(g_engines_mu);
        all.swap(g_engines);
    }
    for (MultiEngine* e : all) engine_destroy(e);
}

// Streams, buffers, launchers and communicators for exactly this list of devices.  Called with the devices' mutexes held.
int engine_build(const std::vector<int>& devs, bool rehearsal, MultiEngine** out) {
    MultiEngine* e = new MultiEngine();
    e->rehearsal = rehearsal;
    e->devices = devs;
    e->ranks.resize(devs.size());
    auto bail = [&](int rc) { engine_destroy(e); return rc; };
    for (size_t d = 0; d < devs.size(); ++d) {
        MultiRank& rk = e->ranks[d];
        hipError_t err = hipSetDevice(devs[d]);
        if (err == hipSuccess) { rk.device = devs[d]; err = hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking); }
        constexpr int kStride = 64;                  // doubles: the receive buffer starts on its own 512-byte boundary (the collective's vector accesses)
        static_assert(kStride >= kMultiValues, "rank buffers hold the widest payload");
        if (err == hipSuccess) err = hipMalloc(&rk.d_send, sizeof(double) * 2 * kStride);
        if (err == hipSuccess) { rk.d_recv = rk.d_send + kStride; err = hipEventCreateWithFlags(&rk.queued, hipEventDisableTiming); }
        if (err != hipSuccess) return bail(fail(OLMC_ERR_HIP, std::string("multi-GPU engine, rank ") + std::to_string(d) + ": " + hipGetErrorString(err)));
    }
    if (!rehearsal) {
        int rc = rccl_load();
        if (rc) return bail(rc);
        e->comms.assign(devs.size(), nullptr);
        std::vector<int> list = devs;
        const ncclResult_t r = g_rccl.CommInitAll(e->comms.data(), static_cast<int>(list.size()), list.data());
        if (r != ncclSuccess) return bail(fail(OLMC_ERR_RCCL, rccl_message("ncclCommInitAll", r)));
    }
    if (devs.size() > 1) {                           // one rank needs no launcher: the calling thread is as good
        e->board.n_ranks = static_cast<int>(devs.size());
        try {
            for (size_t d = 0; d < devs.size(); ++d) e->ranks[d].launcher = std::thread(launcher_main, e, static_cast<int>(d), e->board.job_no.load());
        } catch (const std::system_error& err) {
            return bail(fail(OLMC_ERR_STATE, std::string("multi-GPU engine: cannot start a launcher thread: ") + err.what()));
        }
    }
    *out = e;
    return OLMC_OK;
}

int engine_for(const std::vector<int>& devs, bool rehearsal, MultiEngine** out) {
    {
        std::lock_guard<std::mutex> lock(g_engines_mu);
        for (MultiEngine* e : g_engines)
            if (e->devices == devs && e->rehearsal == rehearsal) { *out = e; return OLMC_OK; }
    }
    // A caller that keeps changing its device list (a rehearsal sweeping 1 .. 12 ranks on one device) must not pile up streams and
    // parked launchers for ever: beyond kMaxEngines the oldest engines go -- but only engines ALL of whose devices this call has
    // locked (nobody can be inside them), oldest first.
    constexpr size_t kMaxEngines = 4;
    std::vector<MultiEngine*> evicted;
    {
        std::lock_guard<std::mutex> lock(g_engines_mu);
        for (size_t i = 0; i < g_engines.size() && g_engines.size() >= kMaxEngines;) {
            const std::vector<int>& theirs = g_engines[i]->devices;
            const bool ours = std::all_of(theirs.begin(), theirs.end(), [&](int d) { return std::find(devs.begin(), devs.end(), d) != devs.end(); });
            if (!ours) { ++i; continue; }
            evicted.push_back(g_engines[i]);
            g_engines.erase(g_engines.begin() + static_cast<std::ptrdiff_t>(i));
        }
    }
    for (MultiEngine* old : evicted) engine_destroy(old);      // joins its launchers, hands its streams' workspace slots back, destroys them
    MultiEngine* e = nullptr;
    const int rc = engine_build(devs, rehearsal, &e);      // the devices' mutexes are held: nobody else builds this list meanwhile
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g_engines_mu);
    g_engines.push_back(e);
    *out = e;
    return OLMC_OK;
}

// After a stream of a failed call could not be drained: the self-resetting counters of that device's workspaces may be dirty.
void pool_recover(int dev) {
    DevicePool* pool = g_pool[dev].load(std::memory_order_acquire);
    if (!pool) return;
    std::lock_guard<std::mutex> lock(pool->mu);
    for (DeviceCtx* c : pool->all) ws_recover(c);
}

// Whatever way a multi-GPU call leaves (any of its error returns included), the calling thread gets back the library device and
// the HIP current device it came in with, and every rank stream a kernel was already queued on has been drained, so no launch of
// a failed call is still running behind the caller's next one.  It also holds the device mutexes of the call.
struct MultiGpuScope {
    int saved_lib_device, saved_hip_device = -1;
    MultiEngine* engine = nullptr;
    std::vector<int> launched;          // ranks with work queued by this call
    std::vector<int> locked;            // devices whose multi-GPU mutex this call holds
    explicit MultiGpuScope(int lib_device) : saved_lib_device(lib_device) { (void)hipGetDevice(&saved_hip_device); }
    void lock_devices(const std::vector<int>& devs) {
        locked = devs;
        std::sort(locked.begin(), locked.end());
        locked.erase(std::unique(locked.begin(), locked.end()), locked.end());
        for (int d : locked) g_multi_dev_mu[d].lock();
    }
    ~MultiGpuScope() {
        if (engine)
            for (int d : launched) {
                const MultiRank& rk = engine->ranks[d];
                if (hipSetDevice(rk.device) == hipSuccess && hipStreamSynchronize(rk.stream) != hipSuccess) pool_recover(rk.device);
            }
        for (auto it = locked.rbegin(); it != locked.rend(); ++it) g_multi_dev_mu[*it].unlock();
        t_device = saved_lib_device;
        if (saved_hip_device >= 0) (void)hipSetDevice(saved_hip_device);
        (void)hipGetLastError();
    }
};

// The grouped all-reduce of `count` doubles, one per rank stream.  The group is CLOSED on every path out (an error between
// ncclGroupStart and ncclGroupEnd would otherwise leave the calling thread inside an open group for good).
int grouped_allreduce(MultiEngine* e, int count) {
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (size_t d = 0; d < e->ranks.size() && bad == ncclSuccess; ++d) {
        const MultiRank& rk = e->ranks[d];
        bad = g_rccl.AllReduce(rk.d_send, rk.d_recv, static_cast<size_t>(count), ncclFloat64, ncclSum, e->comms[d], rk.stream);
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return fail(OLMC_ERR_RCCL, rccl_message("ncclAllReduce", bad));
    if (end != ncclSuccess) return fail(OLMC_ERR_RCCL, rccl_message("ncclGroupEnd", end));
    return OLMC_OK;
}

#ifdef OLMC_WITH_PROBES
// The instrumented build's stand-in for the collective when n ranks are REHEARSED on one device (RCCL refuses two ranks on one
// GPU): every rank adds the n send buffers in rank order.
struct RankBuffers {
    const double* send[kMaxDevices];
};
__global__ void rehearsal_allreduce_kernel(RankBuffers in, int n_ranks, int count, double* __restrict__ recv) {
    const int t = threadIdx.x;
    if (t >= count) return;
    double sum = 0.0;
    for (int r = 0; r < n_ranks; ++r) sum += in.send[r][t];
    recv[t] = sum;
}

int rehearsal_allreduce(MultiEngine* e, int count) {
    const int n = static_cast<int>(e->ranks.size());
    RankBuffers in{};
    for (int d = 0; d < n; ++d) {
        in.send[d] = e->ranks[d].d_send;
        HIP_TRY(hipEventRecord(e->ranks[d].queued, e->ranks[d].stream));
    }
    for (int d = 0; d < n; ++d) {
        for (int o = 0; o < n; ++o)
            if (o != d) HIP_TRY(hipStreamWaitEvent(e->ranks[d].stream, e->ranks[o].queued, 0));
        hipLaunchKernelGGL(rehearsal_allreduce_kernel, dim3(1), dim3(kWave), 0, e->ranks[d].stream, in, n, count, e->ranks[d].d_recv);
        HIP_TRY(hipGetLastError());
    }
    return OLMC_OK;
}
#endif

// The skeleton every multi-GPU entry point shares.  launch(rank, lo, n_local, stream, d_send) queues rank's path kernel on ITS
// stream (the thread it runs on has the rank's device as its library device) and must leave `count` doubles at d_send; host
// receives their sums.  `launch` runs on the launcher threads, one rank each, at the same time: it must not write shared state.
template <typename Launch>
int multi_gpu_run(int n_gpus, int64_t n_paths, int32_t n_steps, int count, Launch launch, double* host, bool sobol_points = false) {
    if (n_gpus < 1 || n_gpus > kMaxDevices) return fail(OLMC_ERR_ARG, "n_gpus out of range");
    int rc = check_paths(0, n_paths, n_steps);
    if (rc) return rc;
    if (n_paths < n_gpus) return fail(OLMC_ERR_ARG, "fewer paths than GPUs");
    if (count < 1 || count > kMultiValues) return fail(OLMC_ERR_STATE, "payload wider than the rank buffers");
    bool rehearsal = false;
#ifdef OLMC_WITH_PROBES
    rehearsal = g_multi_rehearsal != 0;
#endif
    int visible = 0;
    hipError_t e = hipGetDeviceCount(&visible);
    if (e != hipSuccess || visible < (rehearsal ? 1 : n_gpus))
        return fail(OLMC_ERR_HIP, "requested " + std::to_string(n_gpus) + " GPUs, " + std::to_string(visible) + " visible");
    using clock = std::chrono::steady_clock;
    const int home = t_device >= 0 ? t_device : (g_default_device.load() >= 0 ? g_default_device.load() : 0);
    MultiGpuScope scope(home);
    std::vector<int> devs(n_gpus);
    for (int d = 0; d < n_gpus; ++d) devs[d] = rehearsal ? home : d;
    if (home >= kMaxDevices) return fail(OLMC_ERR_ARG, "device index out of range");
    scope.lock_devices(devs);
    for (int d = 0; d < n_gpus; ++d)
        if (!g_pool[devs[d]].load(std::memory_order_acquire)) { rc = olmc_init(devs[d]); if (rc) return rc; }      // olmc_init moves t_device: the guard puts it back
    MultiEngine* eng = nullptr;
    rc = engine_for(devs, rehearsal, &eng);
    if (rc) return rc;
    scope.engine = eng;
    const bool threaded = n_gpus > 1 && g_multi_launch >= 0;
    const auto t0 = clock::now();
    // 1. every rank's path kernel: contiguous global path ranges, rank d owns [d N / P, (d + 1) N / P)  
    const std::function<int(int)> launch_rank = [&](int d) -> int {
        const MultiRank& rk = eng->ranks[d];
        int64_t lo, n_local;
        if (sobol_points) qmc_shard_range(n_paths, d, n_gpus, &lo, &n_local);
        else shard_range(n_paths, d, n_gpus, &lo, &n_local);
#ifdef OLMC_WITH_PROBES
        if (g_fault_shard == d + 1) return fail(OLMC_ERR_HIP, "injected shard failure (OLMC_PROBE_TUNE_FAULT_SHARD)");
#endif
        return launch(d, lo, n_local, rk.stream, rk.d_send);
    };
    if (threaded) {
        for (int d = 0; d < n_gpus; ++d) scope.launched.push_back(d);       // any of them may have queued work by the time one fails
        engine_post(eng, &launch_rank, 0);
        rc = engine_wait(eng);
        if (rc) return rc;
    } else {
        for (int d = 0; d < n_gpus; ++d) {
            HIP_TRY(hipSetDevice(eng->ranks[d].device));
            t_device = eng->ranks[d].device;
            scope.launched.push_back(d);
            rc = launch_rank(d);
            if (rc) return rc;
        }
    }
    const auto t1 = clock::now();
    auto us = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    double wake_max = 0.0, own_max = 0.0, own_min = 0.0;      // launcher threads: how late the slowest one began, how long a rank's own launch took
    if (threaded) {
        own_min = 1e30;
        for (const MultiRank& rk : eng->ranks) {
            wake_max = std::max(wake_max, us(t0, rk.began));
            own_max = std::max(own_max, us(rk.began, rk.ended));
            own_min = std::min(own_min, us(rk.began, rk.ended));
        }
    } else {
        own_max = own_min = us(t0, t1) / n_gpus;
    }
    // 2. the collective: queued on every rank stream before the first wait
    if (!rehearsal) {
        rc = grouped_allreduce(eng, count);
        if (rc) return rc;
    } else {
#ifdef OLMC_WITH_PROBES
        rc = rehearsal_allreduce(eng, count);
        if (rc) return rc;
#endif
    }
    const auto t2 = clock::now();
    // 3. The sums come home the way every blocking pricing's result does: a one-wave kernel behind the all-reduce on rank 0 writes
    // them into a pinned buffer and raises the completion word the host polls (olmc_fetch_dev); the other ranks hold the same sums
    // and finish with the same collective -- their launchers wait for their streams meanwhile.
    const std::function<int(int)> drain_rank = [&](int d) -> int {
        const hipError_t err = hipStreamSynchronize(eng->ranks[d].stream);
        return err == hipSuccess ? OLMC_OK : fail(OLMC_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(err));
    };
    const MultiRank& first = eng->ranks[0];
    HIP_TRY(hipSetDevice(first.device));
    t_device = first.device;
    // from the post to the wait below NOTHING may return: the job refers to this frame
    if (threaded) engine_post(eng, &drain_rank, 0);     // rank 0's launcher too: every launcher ends the call awake, and spins for the next
    rc = olmc_fetch_dev(first.d_recv, count, first.stream, host);
    const auto t3 = clock::now();
    if (threaded) {
        const int rc_drain = engine_wait(eng);          // always: the job refers to this frame
        if (!rc) rc = rc_drain;
    } else if (!rc) {
        for (int d = n_gpus - 1; d >= 1 && !rc; --d) {
            HIP_TRY(hipSetDevice(eng->ranks[d].device));
            rc = drain_rank(d);
        }
    }
    if (rc) return rc;
    const auto t4 = clock::now();
    t_multi_spans[5] = wake_max; t_multi_spans[6] = own_max; t_multi_spans[7] = own_min;
    t_multi_spans[0] = us(t0, t1); t_multi_spans[1] = us(t1, t2); t_multi_spans[2] = us(t2, t3); t_multi_spans[3] = us(t3, t4); t_multi_spans[4] = us(t0, t4);
#ifdef OLMC_WITH_PROBES
    // instrumented build: every rank must hold rank 0's bits (identical finalisation on every rank)
    for (int d = 1; d < n_gpus; ++d) {
        double other[kMultiValues];
        HIP_TRY(hipSetDevice(eng->ranks[d].device));
        HIP_TRY(hipMemcpy(other, eng->ranks[d].d_recv, sizeof(double) * count, hipMemcpyDeviceToHost));
        if (std::memcmp(other, host, sizeof(double) * count) != 0) return fail(OLMC_ERR_STATE, "rank " + std::to_string(d) + " holds other sums than rank 0");
    }
#endif
    scope.launched.clear();                        // rank 0 handed over by its completion word, the others drained: nothing left for the guard
    return OLMC_OK;
}

// k contracts on this rank's block of the common normals, queued on the rank's stream: {sum, sumsq} x nsets then n at d_out.
int batch_shard_dev(const olmc_option* opts, int32_t k, int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic,
                    double* d_out, hipStream_t s, int* pos) {
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    const double n = static_cast<double>(n_local * (antithetic ? 2 : 1));
    return run_batch_device(lease.c, s, opts, k, path_offset, n_local, n_steps, seed, antithetic, d_out, n, pos);
}

// The five control-variate moments of this rank's block (of the UNdiscounted payoff) then n at d_out.
int cv_shard_dev(const olmc_option& o, int64_t path_offset, int64_t n_local, int32_t n_steps, uint64_t seed, int antithetic, double* d_out, hipStream_t s) {
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    PathRange pr = make_range(path_offset, n_local, n_steps, seed);
    const int32_t grid = european_launch_shape(c, &pr, european_occupancy<1, kControlVariate>(antithetic != 0));
    ContractSet<1> cs;
    cs.c[0] = make_contract(o, n_steps);
    cs.base_mask = 1u; cs.upper_continues_slot0 = 0;
    ReduceWs ws;
    rc = make_ws(c, s, grid, 5, d_out, static_cast<double>(n_local * (antithetic ? 2 : 1)), &ws);
    if (rc) return rc;
    launch_european<1, kControlVariate>(antithetic != 0, grid, s, pr, cs, ws, nullptr);
    return after_launch(c, s);
}
}  // namespace

extern "C" int olmc_multi_gpu_european(double S, double K, double T, double r, double sigma, double q, int is_call,
                                       int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic, int n_gpus,
                                       olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    double host[3] = {0, 0, 0};
    const int rc = multi_gpu_run(n_gpus, n_paths, n_steps, 3, [&](int, int64_t lo, int64_t n_local, hipStream_t s, double* d_send) {
        return olmc_european_shard_dev(S, K, T, r, sigma, q, is_call, lo, n_local, n_steps, seed, antithetic, d_send, s);
    }, host);
    if (rc) return rc;
    finish_stats(host[0], host[1], static_cast<int64_t>(host[2]), r, T, out);
    if (poisoned(S, K, T, r, sigma, q)) nan_stats(out->n, out);
    return OLMC_OK;
}

extern "C" int olmc_multi_gpu_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                                        int64_t n_paths, int32_t n_steps, uint64_t seed, int second_order, int n_gpus,
                                        double* out9, olmc_stats* evals) {
    if (!out9) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(T > 0.0)) return fail(OLMC_ERR_ARG, "T must be > 0 (price() returns intrinsic value without simulating)");
    const GreeksSet gs(S, K, T, r, sigma, q, is_call, second_order);
    const int nsets = gs.k <= 8 ? 8 : 16;
    int pos[OLMC_MAX_BATCH] = {};
    double host[kMultiValues] = {};
    const int rc = multi_gpu_run(n_gpus, n_paths, n_steps, 2 * nsets + 1, [&](int rank, int64_t lo, int64_t n_local, hipStream_t s, double* d_send) {
        int other[OLMC_MAX_BATCH];                                                               // every rank lays the set out alike: rank 0's
        return batch_shard_dev(gs.o, gs.k, lo, n_local, n_steps, seed, 1, d_send, s, rank == 0 ? pos : other);      // layout is the one kept
    }, host);
    if (rc) return rc;
    const int64_t n = static_cast<int64_t>(host[2 * nsets]);
    olmc_stats st[OLMC_MAX_BATCH];
    for (int i = 0; i < gs.k; ++i) {
        finish_stats(host[2 * pos[i]], host[2 * pos[i] + 1], n, gs.o[i].r, gs.o[i].T, &st[i]);
        if (poisoned(gs.o[i].S, gs.o[i].K, gs.o[i].T, gs.o[i].r, gs.o[i].sigma, gs.o[i].q)) nan_stats(n, &st[i]);
    }
    gs.finish(st, T, out9, evals);
    return OLMC_OK;
}

extern "C" int olmc_multi_gpu_european_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                                          int64_t n_paths, int32_t n_steps, uint64_t seed, int antithetic, int n_gpus,
                                          olmc_cv_moments* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    const olmc_option o = make_option(S, K, T, r, sigma, q, is_call);
    double host[6] = {};
    const int rc = multi_gpu_run(n_gpus, n_paths, n_steps, 6, [&](int, int64_t lo, int64_t n_local, hipStream_t s, double* d_send) {
        return cv_shard_dev(o, lo, n_local, n_steps, seed, antithetic, d_send, s);
    }, host);
    if (rc) return rc;
    const double disc = std::exp(-r * T);          // device moments are of the UNdiscounted payoff x; d = disc * x (monte_carlo.py:175)
    out->sum_d = disc * host[0];
    out->sum_s = host[1];
    out->sum_dd = disc * disc * host[2];
    out->sum_ss = host[3];
    out->sum_ds = disc * host[4];
    out->n = static_cast<int64_t>(host[5]);
    cv_finish(S, T, r, q, out);
    if (poisoned(S, K, T, r, sigma, q)) out->value = std::nan("");
    return OLMC_OK;
}

// Scrambled-Sobol pricing over n_gpus devices (gbm_qmc.py:14-46): rank d prices POINTS [d N / P, (d + 1) N / P) of the one sequence --
// the inner boundaries rounded down to multiples of 512 points where a rank owns 4,096 or more (qmc_shard_range: every rank then
// runs the aligned kernels) -- through the point offset every Sobol kernel already takes, the ranks' {sum, sumsq, n} meet in the same all-reduce (count 3).  The
// same points as the one-device call, another association of the sums.
extern "C" int olmc_multi_gpu_european_qmc(double S, double K, double T, double r, double sigma, double q, int is_call,
                                           int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
                                           int n_gpus, olmc_stats* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = qmc_check(sv, shift, bits, dims, 0, n_paths);
    if (rc) return rc;
    double host[3] = {0, 0, 0};
    rc = multi_gpu_run(n_gpus, n_paths, dims, 3, [&](int, int64_t lo, int64_t n_local, hipStream_t s, double* d_send) {
        return run_qmc(S, K, T, r, sigma, q, is_call, lo, n_local, dims, sv, shift, bits, nullptr, nullptr, 0, nullptr, d_send, s);
    }, host, true);
    if (rc) return rc;
    finish_stats(host[0], host[1], static_cast<int64_t>(host[2]), r, T, out);
    if (poisoned(S, K, T, r, sigma, q)) nan_stats(out->n, out);
    return OLMC_OK;
}

// The 8 / 14 bumped contracts of compute_greeks_unified on a MCMethod.QMC pricer (unified_greeks.py:295-358 over gbm_qmc.py:14-46)
// over n_gpus devices: every rank prices ALL contracts on its block of the Sobol points in one launch (european_qmc_batch_kernel),
// the 2 nsets sums and n meet in the one all-reduce (count 17 / 33, as olmc_multi_gpu_greeks_fd).
extern "C" int olmc_multi_gpu_european_qmc_greeks_fd(double S, double K, double T, double r, double sigma, double q, int is_call,
                                                     int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift,
                                                     int32_t bits, int second_order, int n_gpus, double* out9, olmc_stats* evals) {
    if (!out9) return fail(OLMC_ERR_ARG, "null pointer");
    if (!(T > 0.0)) return fail(OLMC_ERR_ARG, "T must be > 0 (price() returns intrinsic value without simulating)");
    int rc = qmc_check(sv, shift, bits, dims, 0, n_paths);
    if (rc) return rc;
    const GreeksSet gs(S, K, T, r, sigma, q, is_call, second_order);
    const int nsets = gs.k <= 8 ? 8 : 16;
    int pos[OLMC_MAX_BATCH] = {};
    double host[kMultiValues] = {};
    rc = multi_gpu_run(n_gpus, n_paths, dims, 2 * nsets + 1, [&](int rank, int64_t lo, int64_t n_local, hipStream_t s, double* d_send) {
        int other[OLMC_MAX_BATCH];                                                               // every rank lays the set out alike: rank 0's
        return run_qmc_batch(gs.o, gs.k, lo, n_local, dims, sv, shift, bits, nullptr, d_send, s, rank == 0 ? pos : other);      // layout is the one kept
    }, host, true);
    if (rc) return rc;
    const int64_t n = static_cast<int64_t>(host[2 * nsets]);
    olmc_stats st[OLMC_MAX_BATCH];
    for (int i = 0; i < gs.k; ++i) {
        finish_stats(host[2 * pos[i]], host[2 * pos[i] + 1], n, gs.o[i].r, gs.o[i].T, &st[i]);
        if (poisoned(gs.o[i].S, gs.o[i].K, gs.o[i].T, gs.o[i].r, gs.o[i].sigma, gs.o[i].q)) nan_stats(n, &st[i]);
    }
    gs.finish(st, T, out9, evals);
    return OLMC_OK;
}

// price_with_control_variate on a MCMethod.QMC pricer (monte_carlo.py:154-186 over gbm_qmc.py:14-46) over n_gpus devices: the five
// moments of the rank's block of Sobol points and n (count 6, as olmc_multi_gpu_european_cv).
extern "C" int olmc_multi_gpu_european_qmc_cv(double S, double K, double T, double r, double sigma, double q, int is_call,
                                              int64_t n_paths, int32_t dims, const uint32_t* sv, const uint32_t* shift, int32_t bits,
                                              int n_gpus, olmc_cv_moments* out) {
    if (!out) return fail(OLMC_ERR_ARG, "null pointer");
    int rc = qmc_check(sv, shift, bits, dims, 0, n_paths);
    if (rc) return rc;
    double host[6] = {};
    rc = multi_gpu_run(n_gpus, n_paths, dims, 6, [&](int, int64_t lo, int64_t n_local, hipStream_t s, double* d_send) {
        return run_qmc(S, K, T, r, sigma, q, is_call, lo, n_local, dims, sv, shift, bits, nullptr, nullptr, 0, nullptr, d_send, s, true);
    }, host, true);
    if (rc) return rc;
    cv_from_device(host, static_cast<int64_t>(host[5]), S, T, r, q, out);
    if (poisoned(S, K, T, r, sigma, q)) out->value = std::nan("");
    return OLMC_OK;
}

// Host microseconds of the calling thread's last multi-GPU call: {launch phase (launch job posted -> every rank's kernel queued),
// collective queued, result fetched (includes the kernels' run time), other ranks drained, total, the latest launcher's start after the
// post (wake latency), the longest and the shortest single rank's own launch}.
extern "C" int olmc_multi_gpu_spans(double* out8) {
    if (!out8) return fail(OLMC_ERR_ARG, "null pointer");
    for (int i = 0; i < 8; ++i) out8[i] = t_multi_spans[i];
    return OLMC_OK;
}

// ============================================================ validation taps ====
extern "C" int olmc_philox_words(uint64_t seed, int64_t path_offset, int64_t n_paths, int32_t block0, int32_t n_blocks,
                                 uint32_t stream_tag, uint32_t* out_host) {
    if (!out_host || n_paths < 1 || n_blocks < 1 || block0 < 0 || path_offset < 0) return fail(OLMC_ERR_ARG, "bad arguments");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t bytes = sizeof(uint32_t) * 4 * static_cast<size_t>(n_paths) * n_blocks;
    rc = bulk_reserve(c, bytes);
    if (rc) return rc;
    const int64_t total = n_paths * n_blocks;
    const int grid = static_cast<int>(std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(philox_words_kernel, dim3(grid), dim3(256), 0, c->stream, static_cast<uint64_t>(path_offset), n_paths,
                       block0, n_blocks, stream_tag, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32),
                       static_cast<uint32_t*>(c->d_bulk));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, c->d_bulk, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OLMC_OK;
}

extern "C" int olmc_normals(uint64_t seed, int64_t path_offset, int64_t n_paths, int32_t n_steps, float* out_host) {
    if (!out_host || n_paths < 1 || n_steps < 1 || path_offset < 0) return fail(OLMC_ERR_ARG, "bad arguments");
    CtxLease lease;
    int rc = ctx_lease(&lease);
    if (rc) return rc;
    DeviceCtx* const c = lease.c;
    const size_t bytes = sizeof(float) * static_cast<size_t>(n_paths) * n_steps;
    rc = bulk_reserve(c, bytes);
    if (rc) return rc;
    const int64_t total = n_paths * ((n_steps + 3) / 4);
    const int grid = static_cast<int>(std::min<int64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(normals_kernel, dim3(grid), dim3(256), 0, c->stream, static_cast<uint64_t>(path_offset), n_paths,
                       n_steps, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32),
                       static_cast<float*>(c->d_bulk));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, c->d_bulk, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OLMC_OK;
}

// ================================================================ measurement ====
extern "C" int olmc_tune(int knob, int value) {
    if (knob == OLMC_TUNE_GRID_CAP && value >= 0) { g_grid_cap = value; return OLMC_OK; }
    if (knob == OLMC_TUNE_QMC_BLOCK && value >= -1 && value <= 2) { g_qmc_block = value; return OLMC_OK; }
    if (knob == OLMC_TUNE_POLL && value >= -1 && value <= 0) { g_poll = value; return OLMC_OK; }
    if (knob == OLMC_TUNE_SPLIT_TAIL && value >= -1 && value <= 0) { g_split_tail = value; return OLMC_OK; }
    if (knob == OLMC_TUNE_SPLIT_SAT && value >= 0 && value <= 16) { g_split_sat = value; return OLMC_OK; }
    if (knob == OLMC_TUNE_MULTI_LAUNCH && value >= -1 && value <= 0) { g_multi_launch = value; return OLMC_OK; }
    if (knob == OLMC_TUNE_STAGED_COPY && value >= -1 && value <= 0) { g_staged_copy = value; return OLMC_OK; }
    return fail(OLMC_ERR_ARG, "unknown tuning knob or value");
}

namespace {
// Runs fn on EVERY context of the calling thread's device, each taken out of circulation first (waits for the calls in flight).
template <typename Fn>
int with_all_contexts(Fn fn) {
    int dev = t_device >= 0 ? t_device : g_default_device.load(std::memory_order_acquire);
    if (dev < 0) {
        int rc = olmc_init(0);
        if (rc) return rc;
        dev = t_device;
    }
    DevicePool* pool = g_pool[dev].load(std::memory_order_acquire);
    if (!pool) return fail(OLMC_ERR_STATE, "device not initialised (call olmc_init)");
    HIP_TRY(hipSetDevice(dev));
    std::vector<DeviceCtx*> mine;
    {
        std::unique_lock<std::mutex> lock(pool->mu);
        pool->idle.wait(lock, [&] { return std::none_of(pool->all.begin(), pool->all.end(), [](const DeviceCtx* c) { return c->busy; }); });
        mine = pool->all;
        for (DeviceCtx* c : mine) c->busy = true;
    }
    int rc = OLMC_OK;
    for (DeviceCtx* c : mine) {
        const int r = fn(c);
        if (r && !rc) rc = r;
    }
    {
        std::lock_guard<std::mutex> lock(pool->mu);
        for (DeviceCtx* c : mine) c->busy = false;
    }
    pool->idle.notify_all();
    return rc;
}
}  // namespace

extern "C" int olmc_profile_enable(int on) {
    g_profile = on != 0;
    if (g_profile && (t_device >= 0 || g_default_device.load() >= 0)) {
        // pre-create the event pairs a measurement pass will consume, so that no hipEventCreate lands inside a timed call
        // (context 0 is the one a single-threaded measurement runs on; the others create theirs on demand)
        CtxLease lease;
        int rc = ctx_lease(&lease);
        if (rc) return rc;
        DeviceCtx* const c = lease.c;
        constexpr size_t kPool = 4096;
        while (c->ev_free.size() + c->ev_pending.size() < kPool) {
            EventPair ep{};
            HIP_TRY(hipEventCreate(&ep.start));
            HIP_TRY(hipEventCreate(&ep.stop));
            c->ev_free.push_back(ep);
        }
    }
    return OLMC_OK;
}

extern "C" int olmc_profile_reset(void) {
    return with_all_contexts([](DeviceCtx* c) {
        const int rc = prof_drain(c);
        c->prof_launches = 0;
        c->prof_ms = 0.0;
        return rc;
    });
}

extern "C" int olmc_kernel_time(int64_t* launches, double* total_ms) {
    if (!launches || !total_ms) return fail(OLMC_ERR_ARG, "null pointer");
    int64_t n = 0;
    double ms = 0.0;
    const int rc = with_all_contexts([&](DeviceCtx* c) {
        const int r = prof_drain(c);
        n += c->prof_launches;
        ms += c->prof_ms;
        return r;
    });
    if (rc) return rc;
    *launches = n;
    *total_ms = ms;
    return OLMC_OK;
}
