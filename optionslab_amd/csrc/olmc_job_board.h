// olmc_job_board.h -- the hand-over between the thread that makes a multi-GPU call and the engine's launcher threads (one per
// device, olmc.hip MultiEngine).  Plain C++17 + futex, no HIP: olmc.hip includes it, and tests/job_board_harness.cpp compiles it on
// its own with g++ -fsanitize=thread (GPU sanitizers are not available on the pool; this code needs no GPU).
//
// Protocol (ONE caller at a time: a multi-GPU call holds the mutex of every device of its list):
//   caller     board_post(work, first_rank): writes the job (plain fields), then moves `job_no` (seq_cst) and wakes sleepers;
//              board_wait(): until `remaining` is 0.
//   launcher d board_next(): waits for `job_no` to move (spin 200 us, then futex), reads the job behind that acquire.  EVERY
//              launcher acknowledges EVERY job through `remaining` -- a launcher below first_rank acknowledges at once, without
//              running it -- so the caller cannot post job k + 1 while any launcher has yet to read job k: the job's plain fields
//              are never written while somebody may read them, and no launcher can run a job twice or skip one.
//              board_done(): the acknowledgement of a launcher that ran the job (release: what it wrote is the caller's to read).
//   leave      work == nullptr: board_next returns nullptr without acknowledging; the caller joins the threads.
#ifndef OLMC_JOB_BOARD_H
#define OLMC_JOB_BOARD_H

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <functional>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace olmc {

// A word threads sleep on (futex): the launchers wait for the board's job number to move.
inline long futex_call(std::atomic<uint32_t>* word, int op, uint32_t value) {
    static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");
    return syscall(SYS_futex, reinterpret_cast<uint32_t*>(word), op, value, nullptr, nullptr, 0);
}

constexpr int64_t kLauncherSpinUs = 200;            // a launcher that finished a job spins this long for the next one before it sleeps
constexpr int64_t kBoardWaitSpinUs = 200;           // the caller spins this long for the acknowledgements before it starts yielding

struct JobBoard {
    int n_ranks = 0;                                // launchers serving this board (set before they start)
    std::atomic<uint32_t> job_no{0};                // moves once per posted job
    std::atomic<int> sleepers{0};                   // launchers inside futex_wait (the caller skips the wake syscall when none)
    std::atomic<int> remaining{0};                  // launchers that have not acknowledged the posted job yet
    const std::function<int(int)>* work = nullptr;  // work(rank); nullptr = leave
    int first_rank = 0;                             // ranks below acknowledge the posted job without running it
};

inline void board_post(JobBoard& b, const std::function<int(int)>* work, int first_rank) {
    b.work = work;
    b.first_rank = first_rank;
    b.remaining.store(b.n_ranks, std::memory_order_relaxed);
    b.job_no.fetch_add(1, std::memory_order_seq_cst);
    if (b.sleepers.load(std::memory_order_seq_cst) > 0) futex_call(&b.job_no, FUTEX_WAKE_PRIVATE, INT_MAX);
}

inline void board_wait(JobBoard& b) {
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    for (uint32_t spins = 0; b.remaining.load(std::memory_order_acquire) > 0; ++spins) {
        if ((spins & 0x3F) == 0x3F && clock::now() - t0 >= std::chrono::microseconds(kBoardWaitSpinUs)) sched_yield();
        else __builtin_ia32_pause();
    }
}

inline void board_done(JobBoard& b) { b.remaining.fetch_sub(1, std::memory_order_release); }

// Launcher d: the next job this rank has to RUN (nullptr: leave).  `seen` is the job number the launcher has dealt with so far (the
// board's number when the thread was started, then whatever this function leaves in it).
inline const std::function<int(int)>* board_next(JobBoard& b, int d, uint32_t& seen) {
    using clock = std::chrono::steady_clock;
    for (;;) {
        auto t0 = clock::now();
        uint32_t now;
        for (uint32_t spins = 0; (now = b.job_no.load(std::memory_order_acquire)) == seen; ++spins) {
            if ((spins & 0x3F) != 0x3F || clock::now() - t0 < std::chrono::microseconds(kLauncherSpinUs)) {
                __builtin_ia32_pause();
                continue;
            }
            b.sleepers.fetch_add(1, std::memory_order_seq_cst);
            if (b.job_no.load(std::memory_order_seq_cst) == seen) futex_call(&b.job_no, FUTEX_WAIT_PRIVATE, seen);
            b.sleepers.fetch_sub(1, std::memory_order_seq_cst);
            t0 = clock::now();
        }
        seen = now;
        const std::function<int(int)>* work = b.work;
        if (!work) return nullptr;
        if (d >= b.first_rank) return work;
        board_done(b);                              // not this rank's job: acknowledged all the same (see the protocol above)
    }
}

}  // namespace olmc

#endif  // OLMC_JOB_BOARD_H
