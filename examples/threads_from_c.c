/* Concurrent callers of the C ABI from a plain C host: N threads, each pricing back to back through olmc_european for a
 * fixed time -- the library's own scaling with threads, without an interpreter lock in the way (tools/thread_bench.py measures
 * the same through Python, where the GIL serialises the ~4 us of ctypes marshalling per call).  Every thread checks that it
 * gets the bits thread 0 got for the same contract.
 *
 *   gcc -O2 -pthread -Iinclude examples/threads_from_c.c -o /tmp/threads_from_c -Loptionslab_amd -lolmc -Wl,-rpath,$PWD/optionslab_amd
 *   /tmp/threads_from_c [n_paths] [n_steps] [seconds]      -> one JSON line per thread count
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "olmc.h"

enum { MAX_THREADS = 16 };

typedef struct {
    int64_t n_paths;
    int32_t n_steps;
    double seconds;
    volatile int* go;
    long calls;
    int failed;
    olmc_stats first;
} job;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void* worker(void* arg) {
    job* j = (job*)arg;
    olmc_stats st;
    if (olmc_init(0) != 0) { j->failed = 1; return NULL; }
    for (int k = 0; k < 50; ++k)
        if (olmc_european(100, 100, 1.0, 0.05, 0.2, 0.0, 1, j->n_paths, j->n_steps, 42, 1, &st) != 0) { j->failed = 1; return NULL; }
    j->first = st;
    while (!*j->go) { }
    const double t_end = now_s() + j->seconds;
    long calls = 0;
    while (now_s() < t_end) {
        if (olmc_european(100, 100, 1.0, 0.05, 0.2, 0.0, 1, j->n_paths, j->n_steps, 42, 1, &st) != 0 ||
            memcmp(&st, &j->first, sizeof st) != 0) { j->failed = 1; break; }
        ++calls;
    }
    j->calls = calls;
    return NULL;
}

int main(int argc, char** argv) {
    const int64_t n_paths = argc > 1 ? atoll(argv[1]) : 10000;
    const int32_t n_steps = argc > 2 ? atoi(argv[2]) : 50;
    const double seconds = argc > 3 ? atof(argv[3]) : 1.0;
    if (olmc_init(0) != 0) {
        fprintf(stderr, "olmc_init(0) failed: %s\n", olmc_last_error());
        return 1;
    }
    double single = 0.0;
    olmc_stats reference;
    memset(&reference, 0, sizeof reference);
    const int counts[] = {1, 2, 4, 8, 16};
    for (size_t c = 0; c < sizeof counts / sizeof counts[0]; ++c) {
        const int n = counts[c];
        pthread_t th[MAX_THREADS];
        job jobs[MAX_THREADS];
        volatile int go = 0;
        for (int k = 0; k < n; ++k) {
            jobs[k].n_paths = n_paths; jobs[k].n_steps = n_steps; jobs[k].seconds = seconds; jobs[k].go = &go; jobs[k].calls = 0; jobs[k].failed = 0;
            if (pthread_create(&th[k], NULL, worker, &jobs[k]) != 0) return 1;
        }
        struct timespec nap = {0, 200 * 1000 * 1000};
        nanosleep(&nap, NULL);                          /* every thread has warmed up and waits on `go` */
        go = 1;
        long total = 0;
        int failed = 0;
        for (int k = 0; k < n; ++k) {
            pthread_join(th[k], NULL);
            total += jobs[k].calls;
            failed |= jobs[k].failed;
            if (c == 0 && k == 0) reference = jobs[k].first;
            if (memcmp(&jobs[k].first, &reference, sizeof reference) != 0) failed = 1;      /* same bits in every thread, every count */
        }
        const double rate = (double)total / seconds;
        if (n == 1) single = rate;
        printf("{\"host\": \"C (pthreads)\", \"n_paths\": %lld, \"n_steps\": %d, \"threads\": %d, \"calls_per_s\": %.1f, "
               "\"throughput_vs_one_thread\": %.3f, \"mean_latency_us\": %.2f, \"bit_identical_results\": %s}\n",
               (long long)n_paths, n_steps, n, rate, single > 0 ? rate / single : 0.0, rate > 0 ? 1e6 * n / rate : 0.0, failed ? "false" : "true");
        fflush(stdout);
        if (failed) return 2;
    }
    olmc_shutdown();
    return 0;
}
