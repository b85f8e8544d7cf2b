"""optionslab_amd/csrc/olmc_job_board.h -- the hand-over between the thread that makes a multi-GPU call and the engine's launcher
threads (one per device) -- compiled on its own by g++ with ThreadSanitizer and stressed by tests/job_board_harness.cpp: jobs that
refer to the caller's frame, random subsets of ranks, pauses that leave the launchers spinning, asleep in the futex or in between,
a failing rank now and then.  Every job must be run exactly once by every rank it is for, and what a launcher wrote must be the
caller's to read behind board_wait (TSan reports any missing happens-before edge).  The multi-rank engine has never run on more
than one real GPU (include/olmc.h): this is the part of it that needs no GPU to be wrong."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("job_board") / "job_board_tsan"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=thread", "-I" + os.path.join(ROOT, "optionslab_amd", "csrc"),
           "-o", str(exe), os.path.join(ROOT, "tests", "job_board_harness.cpp"), "-lpthread"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("ThreadSanitizer runtime not installed: " + build.stderr.splitlines()[0])
    assert build.returncode == 0, build.stderr
    return str(exe)


@pytest.mark.parametrize("n_ranks,n_jobs,seed", [(2, 20000, 1), (8, 12000, 2), (12, 6000, 3)])
def test_every_job_runs_once_per_rank_and_races_nowhere(harness, n_ranks, n_jobs, seed):
    r = subprocess.run([harness, str(n_ranks), str(n_jobs), str(seed)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"})
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.stdout, r.stderr[-2000:])
