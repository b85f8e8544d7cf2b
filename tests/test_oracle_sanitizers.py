"""The C checker under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the
pool).  Exercises every entry point of oracle/philox_gbm.c on small ragged inputs."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = textwrap.dedent(r"""
    #include <stdint.h>
    #include <stdio.h>
    #include <stdlib.h>
    void ol_philox4x32_10(const uint32_t*, const uint32_t*, uint32_t*);
    void ol_normals(uint64_t, int64_t, int64_t, int32_t, float*);
    void ol_european_terminal(double, double, double, double, double, int64_t, int64_t, int32_t, uint64_t, int, double*);
    void ol_european_moments(double, double, double, double, double, double, int, int64_t, int64_t, int32_t, uint64_t, int, double*);
    void ol_asian_moments(double, double, double, double, double, double, int, int, int64_t, int64_t, int32_t, uint64_t, int, double*);
    void ol_extrema_moments(double, double, double, double, double, double, int, int, double, int64_t, int64_t, int32_t, uint64_t, int, double*);
    void ol_heston_moments(double, double, double, double, double, int, double, double, double, double, double, int64_t, int64_t, int32_t, uint64_t, int, double*);
    void ol_autocall_moments(double, double, double, double, double, double, double, double, double, int32_t, int64_t, int64_t, int32_t, uint64_t, int, double*);
    void ol_cliquet_moments(double, double, double, double, double, double, double, double, double, int32_t, int64_t, int64_t, int32_t, uint64_t, int, double*);
    int ol_american_lsm(double, double, double, double, double, double, int, int64_t, int32_t, int32_t, uint64_t, double*);
    void ol_jump_moments(double, double, double, double, double, double, int, int, double, double, double, double, int64_t, int64_t, int32_t, uint64_t, double*);
    int main(void) {
        uint32_t c[4] = {1, 2, 3, 4}, k[2] = {5, 6}, w[4];
        ol_philox4x32_10(c, k, w);
        double m[5], acc = w[0];
        for (int steps = 1; steps <= 9; ++steps) {
            float* z = malloc(sizeof(float) * 7 * steps);
            ol_normals(42, (1ll << 32) - 3, 7, steps, z);
            acc += z[7 * steps - 1];
            free(z);
            double* st = malloc(sizeof(double) * 2 * 5);
            ol_european_terminal(100, 1, .05, .2, .01, 3, 5, steps, 9, 1, st);
            acc += st[9];
            free(st);
            ol_european_moments(100, 100, 1, .05, .2, 0, steps & 1, 0, 11, steps, 1, steps & 1, m); acc += m[4];
            ol_asian_moments(100, 100, 1, .05, .2, 0, 1, steps & 1, 5, 11, steps, 1, 1, m); acc += m[1];
            for (int p = 0; p < 6; ++p) { ol_extrema_moments(100, 100, 1, .05, .2, 0, 1, p, 110, 0, 9, steps, 2, p & 1, m); acc += m[0]; }
            ol_heston_moments(100, 100, 1, .05, 0, 1, 2, .04, .3, -.7, .04, 0, 9, steps, 3, 1, m); acc += m[0];
            ol_autocall_moments(100, 1, .05, .2, 0, 1.0, .8, .1, .6, 1 + steps / 3, 0, 9, steps, 4, 1, m); acc += m[0];
            ol_cliquet_moments(100, 1, .05, .2, 0, .05, -.05, .3, 0, 1 + steps / 4, 0, 9, steps, 5, 1, m); acc += m[0];
            ol_jump_moments(100, 100, 1, .05, .2, 0, 1, 0, 30.0, -.1, .2, 0, 0, 50, steps, 7, m); acc += m[0];
            ol_jump_moments(100, 100, 1, .05, .2, 0, 0, 1, 40.0, .4, 10, 5, 0, 50, steps, 7, m); acc += m[0];
            if (ol_american_lsm(100, 100, 1, .05, .2, 0, 0, 200, steps, 1 + steps % 4, 6, m)) return 2;
            acc += m[0];
        }
        printf("%.17g\n", acc);
        return 0;
    }
""")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_checker_is_clean_under_asan_and_ubsan(tmp_path):
    main_c = tmp_path / "main.c"
    main_c.write_text(MAIN)
    exe = tmp_path / "checker_san"
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe),
           str(main_c), os.path.join(ROOT, "oracle", "philox_gbm.c"), "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr.splitlines()[0])
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert run.returncode == 0, run.stderr
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
    assert float(run.stdout.strip()) == float(run.stdout.strip())     # finite, not NaN
