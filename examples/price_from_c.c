/* The C ABI from a plain C host (no Python, no torch): price the BASELINE contract, its fused
 * finite-difference Greeks and an arithmetic Asian, then two shards combined as two GPUs would.
 *
 *   gcc -O2 -Iinclude examples/price_from_c.c -o /tmp/price_from_c -Loptionslab_amd -lolmc \
 *       -Wl,-rpath,$PWD/optionslab_amd
 *   /tmp/price_from_c [n_paths] [n_steps] [n_gpus]
 */
#include <stdio.h>
#include <stdlib.h>

#include "olmc.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, olmc_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    const int64_t n_paths = argc > 1 ? atoll(argv[1]) : 1000000;
    const int32_t n_steps = argc > 2 ? atoi(argv[2]) : 252;
    const double S = 100, K = 100, T = 1.0, r = 0.05, sigma = 0.2, q = 0.0;

    if (olmc_abi_version() != OLMC_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch\n");
        return 1;
    }
    CHECK(olmc_init(0));
    olmc_devinfo info;
    CHECK(olmc_device_info(&info));
    printf("device %d: %s (%s), %d CUs\n", info.device, info.name, info.arch, info.compute_units);

    olmc_stats st;
    CHECK(olmc_european(S, K, T, r, sigma, q, 1, n_paths, n_steps, 42, 1, &st));
    printf("european call  price %.6f  std_error %.6f  n %lld\n", st.price, st.std_error, (long long)st.n);

    double greeks[9];
    CHECK(olmc_european_greeks_fd(S, K, T, r, sigma, q, 1, n_paths, n_steps, 42, 1, greeks, NULL));
    printf("greeks (one launch)  delta %.5f gamma %.6f vega %.4f theta %.4f rho %.4f\n", greeks[1], greeks[2], greeks[3],
           greeks[4], greeks[5]);

    olmc_stats asian;
    CHECK(olmc_asian(S, K, T, r, sigma, q, 1, OLMC_AVG_ARITHMETIC, 0, n_paths, n_steps, 42, 0, &asian));
    printf("asian call     price %.6f  std_error %.6f\n", asian.price, asian.std_error);

    /* two contiguous shards of the same global path range, combined as two ranks would after their all-reduce */
    olmc_stats part[2], both;
    CHECK(olmc_european_shard(S, K, T, r, sigma, q, 1, 0, n_paths / 2, n_steps, 42, 1, &part[0]));
    CHECK(olmc_european_shard(S, K, T, r, sigma, q, 1, n_paths / 2, n_paths - n_paths / 2, n_steps, 42, 1, &part[1]));
    CHECK(olmc_combine_stats(part, 2, r, T, &both));
    printf("two shards     price %.6f  (whole %.6f)\n", both.price, st.price);
    if (both.n != st.n || both.price < st.price * (1 - 1e-12) || both.price > st.price * (1 + 1e-12)) {
        fprintf(stderr, "shards do not add up\n");
        return 1;
    }
    /* the same two halves priced by the library itself on `n_gpus` devices of this process: one launch per device, ONE RCCL
     * all-reduce of (sum, sumsq, n) over xGMI, no torch, no extra processes (n_gpus = 1 here: any box can run the example) */
    const int n_gpus = argc > 3 ? atoi(argv[3]) : 1;
    olmc_stats multi;
    CHECK(olmc_multi_gpu_european(S, K, T, r, sigma, q, 1, n_paths, n_steps, 42, 1, n_gpus, &multi));
    printf("%d GPU(s)       price %.6f  std_error %.6f  n %lld\n", n_gpus, multi.price, multi.std_error, (long long)multi.n);
    if (multi.n != st.n || multi.price < st.price * (1 - 1e-12) || multi.price > st.price * (1 + 1e-12)) {
        fprintf(stderr, "the multi-GPU form prices other paths\n");
        return 1;
    }
    CHECK(olmc_shutdown());
    return 0;
}
