// Stress driver for optionslab_amd/csrc/olmc_job_board.h (the hand-over between a multi-GPU call and the engine's launcher threads),
// built by tests/test_job_board_tsan.py with g++ -fsanitize=thread.  No GPU, no HIP: the launchers run a stand-in for "queue this
// rank's kernel" that writes PLAIN per-rank data the caller reads after board_wait -- ThreadSanitizer sees every missing
// happens-before edge, the counters see every job run twice or not at all.
//
//   job_board_harness <n_ranks> <n_jobs> <seed>
#include "olmc_job_board.h"

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

using namespace olmc;

int main(int argc, char** argv) {
    const int n_ranks = argc > 1 ? std::atoi(argv[1]) : 8;
    const long n_jobs = argc > 2 ? std::atol(argv[2]) : 20000;
    uint64_t lcg = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(lcg >> 33); };

    JobBoard board;
    board.n_ranks = n_ranks;
    struct alignas(64) Rank { long payload = -1; long runs = 0; int rc = 0; };      // plain data: written by the launcher, read by the caller
    std::vector<Rank> ranks(n_ranks);
    std::vector<std::thread> launchers;
    for (int d = 0; d < n_ranks; ++d)
        launchers.emplace_back([&board, &ranks, d, seen0 = board.job_no.load()]() mutable {
            uint32_t seen = seen0;
            while (const std::function<int(int)>* work = board_next(board, d, seen)) {
                ranks[d].rc = (*work)(d);
                board_done(board);
            }
        });

    long failures = 0, expected_runs_total = 0;
    std::vector<long> expected_runs(n_ranks, 0);
    for (long job = 0; job < n_jobs; ++job) {
        const int first_rank = (rnd() % 4 == 0) ? static_cast<int>(rnd() % n_ranks) : 0;      // mostly every rank, sometimes a tail of them
        long frame_local = job;                                                                  // the job refers to the caller's frame, as multi_gpu_run's do
        const std::function<int(int)> work = [&](int d) {
            ranks[d].payload = frame_local;
            ranks[d].runs += 1;
            if ((d + frame_local) % 97 == 0) std::this_thread::yield();
            return d == 3 && frame_local % 1000 == 999 ? 7 : 0;                                  // a failing rank now and then
        };
        board_post(board, &work, first_rank);
        board_wait(board);
        for (int d = 0; d < n_ranks; ++d) {
            if (d >= first_rank) { expected_runs[d] += 1; ++expected_runs_total; }
            const bool ran_now = ranks[d].payload == job;
            if (ran_now != (d >= first_rank) || ranks[d].runs != expected_runs[d]) ++failures;
            if (d >= first_rank && ranks[d].rc != (d == 3 && job % 1000 == 999 ? 7 : 0)) ++failures;
        }
        // pauses of every length between calls: launchers still spinning, launchers asleep in the futex, launchers in between
        const uint32_t r = rnd() % 1000;
        if (r < 3) std::this_thread::sleep_for(std::chrono::microseconds(400 + rnd() % 600));
        else if (r < 30) std::this_thread::sleep_for(std::chrono::microseconds(150 + rnd() % 100));
        else if (r < 100) std::this_thread::yield();
    }
    board_post(board, nullptr, 0);          // leave
    for (std::thread& t : launchers) t.join();
    std::printf("%s jobs %ld ranks %d runs %ld failures %ld\n", failures ? "FAILED" : "ok", n_jobs, n_ranks, expected_runs_total, failures);
    return failures ? 1 : 0;
}
