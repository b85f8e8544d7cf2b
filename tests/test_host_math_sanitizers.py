"""optionslab_amd/csrc/olmc_host_math.h -- the pure-host arithmetic behind every fused and every multi-GPU call (the 8 / 14
evaluations of compute_greeks_unified and their finite differences, the contract layouts of the fused kernels, the moment
combiners, the shard ranges) -- compiled on its own by g++ with AddressSanitizer + UBSan (GPU sanitizers are not available on
the pool; this code needs no GPU).  tests/host_math_harness.cpp is the driver."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATM = (100.0, 100.0, 1.0, 0.05, 0.2)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("host_math") / "host_math_san"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "optionslab_amd", "csrc"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host_math_harness.cpp")]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr.splitlines()[0])
    assert build.returncode == 0, build.stderr

    def run(*args, stdin=""):
        r = subprocess.run([str(exe), *map(str, args)], input=stdin, capture_output=True, text=True,
                           env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
        assert r.returncode == 0, r.stderr
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
        return r.stdout

    return run


def test_property_sweep_is_clean_under_asan_and_ubsan(harness):
    """Every Greeks set (k = 7 / 8 / 11 / 14) over a parameter sweep: indices distinct and in range, finite differences exact on a
    quadratic surface, European / barrier / lookback / Asian layouts carry each contract's own constants; 4,000 random batches incl.
    duplicates and poisoned members; combiners and shard ranges."""
    out = harness("self")
    assert out.startswith("ok ") and int(out.split()[1]) > 10_000


@pytest.mark.parametrize("T", [1.0, 0.002])                  # with and without the T bump (unified_greeks.py:310)
@pytest.mark.parametrize("second", [0, 1])
@pytest.mark.parametrize("is_call", [1, 0])
def test_greeks_set_is_the_python_side_compute_greeks_unified(harness, T, second, is_call):
    """The C++ GreeksSet (call order, bumps, differences: unified_greeks.py:274-358) against optionslab_amd.greeks.compute_greeks_unified
    -- the line-by-line restatement the GPU tests hold against the reference's golden Greeks -- on the same closed-form prices."""
    import optionslab_amd as ol
    from optionslab_amd.greeks import compute_greeks_unified

    S, K, _, r, v = ATM
    q, kind = 0.01, "call" if is_call else "put"
    calls = []

    class Pricer:
        def price(self, S_, K_, T_, r_, v_, option_type, q_=0.0):
            calls.append((S_, K_, T_, r_, v_, q_))
            return float(ol.black_scholes(S_, K_, T_, r_, v_, option_type, q_))

    want = compute_greeks_unified(Pricer(), S, K, T, r, v, kind, q, bool(second), fused=False)
    # first pass: the evaluation tuples, in the reference's get_price() order
    head = harness("greeks", S, K, T, r, v, q, is_call, second, stdin="0 " * 16).splitlines()
    k = int(head[0])
    tuples = [tuple(float(x) for x in line.split()[:6]) for line in head[1:1 + k]]
    assert tuples == calls                                   # same evaluations, same order, same bits
    prices = [float(ol.black_scholes(S_, K_, T_, r_, v_, kind, q_)) for (S_, K_, T_, r_, v_, q_) in tuples]
    out9 = [float(x) for x in harness("greeks", S, K, T, r, v, q, is_call, second, stdin=" ".join(repr(p) for p in prices)).splitlines()[1 + k].split()]
    for got, key in zip(out9, want):
        assert got == want[key], key                         # the same differences of the same prices: bit-identical


def test_layout_in_the_header_is_the_layout_the_library_reports(harness):
    """olmc_contract_layout of the built libolmc.so (a pure host entry point: no device) == the header compiled by g++."""
    from optionslab_amd import _hip

    S, K, T, r, v = ATM
    sets = {
        "first order": [(S, K, T, r, v, 0.0, 1), (S + 1, K, T, r, v, 0.0, 1), (S - 1, K, T, r, v, 0.0, 1), (S, K, T, r, v + 0.01, 0.0, 1),
                        (S, K, T, r, v - 0.01, 0.0, 1), (S, K, T - 1 / 365.0, r, v, 0.0, 1), (S, K, T, r + 1e-4, v, 0.0, 1), (S, K, T, r - 1e-4, v, 0.0, 1)],
        "pairs": [(S, K, T, r, 0.3, 0.0, 0), (S, K, T, r, 0.2, 0.0, 0), (2 * S, K, T, r, 0.3, 0.0, 1)],
        "eleven": [(S + i, K, T, r, 0.2 + 0.01 * (i % 3), 0.0, i & 1) for i in range(11)],
    }
    for name, opts in sets.items():
        text = "\n".join(" ".join(repr(float(x)) if j < 6 else str(int(x)) for j, x in enumerate(o)) for o in opts)
        lines = harness("layout", 252, len(opts), stdin=text).splitlines()
        nsets, mask, upper = (int(x) for x in lines[0].split())
        pos = [int(x) for x in lines[1].split()]
        scale = [float(x) for x in lines[2].split()]
        lib = _hip.contract_layout([(*o[:6], bool(o[6])) for o in opts], 252)
        assert (nsets, pos, mask, bool(upper), scale) == lib, name


def test_sobol_shard_ranges_are_the_python_side_ranges(harness):
    """qmc_shard_range (what olmc_multi_gpu_european_qmc cuts the sequence by) against optionslab_amd.sharding.qmc_shard_bounds (what
    one process per GPU cuts it by): the same points on every rank whichever form runs -- inner boundaries on multiples of 512 where a
    rank keeps 4,096 points, the plain ranges below."""
    from optionslab_amd.sharding import qmc_shard_bounds, shard_bounds

    for n, p in [(100_000, 8), (1_000_000, 8), (1 << 17, 3), (32_767, 8), (32_768, 8), (4096, 1), (300_001, 7), (1 << 30, 16), (20, 5)]:
        got = [tuple(map(int, line.split())) for line in harness("qmc-shards", n, p).splitlines()]
        want = [qmc_shard_bounds(n, k, p) for k in range(p)]
        assert got == want, (n, p)
        assert want[0][0] == 0 and want[-1][1] == n and all(a[1] == b[0] for a, b in zip(want, want[1:]))
        if n // p >= 4096:
            assert all(lo % 512 == 0 for lo, _ in want) and all(0 <= shard_bounds(n, k, p)[0] - want[k][0] < 512 for k in range(p))
        else:
            assert want == [shard_bounds(n, k, p) for k in range(p)]
