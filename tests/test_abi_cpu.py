"""CPU checks of the C-ABI boundary: libolmc.so builds/loads, exports every
symbol include/olmc.h declares, the ctypes prototypes cover them all, struct
layouts match, the pure host function works, and -- without a GPU -- compute
entry points fail loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import pytest

from optionslab_amd import _hip
from optionslab_amd.build import LIBRARY, build_library
from optionslab_amd.exceptions import AccelerationError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "olmc.h")
PROBE_HEADER = os.path.join(ROOT, "include", "olmc_probe.h")


def declared_symbols(header=HEADER):
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(olmc_[a-z0-9_]+)\s*\(", text)))


def exported_symbols(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return set(re.findall(r"\bT (olmc_[a-z0-9_]+)", out))


@pytest.fixture(scope="module")
def library():
    build_library()
    return _hip.load_library()


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("olmc_init", "olmc_european", "olmc_european_batch", "olmc_european_greeks_fd", "olmc_european_terminal",
                 "olmc_european_cv", "olmc_asian", "olmc_multi_gpu_european", "olmc_last_error", "olmc_shutdown"):
        assert must in names


def test_library_exports_every_declared_symbol(library):
    for name in declared_symbols():
        assert hasattr(library, name), f"{name} declared in include/olmc.h but not exported"
    assert set(declared_symbols()) == exported_symbols(LIBRARY)      # nothing undeclared is exported either


def test_measurement_code_is_not_in_the_product_library():
    """probe / stamp / clock / moment kernels, their entry points and the fault-injection knobs live in the
    instrumented build (include/olmc_probe.h -> tools/probe/libolmc_probe.so), which also exports the whole pricing ABI."""
    from optionslab_amd.build import PROBE_LIBRARY, build_probe_library
    from tools.probe import binding as probe

    build_probe_library()
    product, instrumented = exported_symbols(LIBRARY), exported_symbols(PROBE_LIBRARY)
    only_probe = set(declared_symbols(PROBE_HEADER))
    assert only_probe == set(probe.PROBE_PROTOTYPES) and len(only_probe) == 10
    assert not (only_probe & product)
    assert not [n for n in product if any(w in n for w in ("probe", "stamp", "clock", "moments"))]
    assert instrumented == product | only_probe
    blob = open(LIBRARY, "rb").read()
    for word in (b"probe_mad_u64_u32", b"european_stamp_kernel", b"clock_probe_kernel", b"normal_moments_kernel", b"FAULT_SHARD", b"injected shard failure"):
        assert word not in blob, word
    assert b"european_stamp_kernel" in open(PROBE_LIBRARY, "rb").read()
    lib = probe.hip.load_library()
    for name in only_probe:
        assert hasattr(lib, name)


def test_ctypes_prototypes_cover_the_header():
    assert sorted(_hip.PROTOTYPES) == declared_symbols()


def test_library_carries_gfx950_code_object():
    blob = open(LIBRARY, "rb").read()
    assert b"gfx950" in blob


def test_struct_layouts_match_the_header():
    assert C.sizeof(_hip.Stats) == 40
    assert C.sizeof(_hip.Option) == 56
    assert C.sizeof(_hip.CvMoments) == 56
    assert _hip.Stats.n.offset == 16 and _hip.Stats.price.offset == 24


def test_abi_version_and_error_string(library):
    assert library.olmc_abi_version() == 6
    assert isinstance(library.olmc_last_error(), bytes)


def test_combine_stats_is_a_pure_host_function(library):
    # two shards of payoffs {1,2,3} and {4}: mean 2.5, var(ddof=0) 1.25
    st = _hip.combine_stats([(6.0, 14.0, 3), (4.0, 16.0, 1)], r=0.0, T=1.0)
    assert st.n == 4 and st.sum == 10.0 and st.sumsq == 30.0
    assert st.price == 2.5
    assert abs(st.std_error - (1.25 ** 0.5) / 2.0) < 1e-15
    with pytest.raises(AccelerationError):
        _hip.combine_stats([], 0.0, 1.0)


def _no_gpu():
    return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="GPU present: loud-failure path not reachable")
def test_compute_fails_loudly_without_a_gpu():
    with pytest.raises(AccelerationError) as e:
        _hip.european(100, 100, 1.0, 0.05, 0.2, 0.0, True, 1000, 4, 1)
    assert e.value.backend == "hip"
    import optionslab_amd as ol
    with pytest.raises(AccelerationError):
        ol.MonteCarloPricer(1000, 4, 1).price(100, 100, 1.0, 0.05, 0.2, "call")
    with pytest.raises(ol.GreeksError):
        ol.compute_greeks_unified(ol.MonteCarloPricer(1000, 4, 1), 100, 100, 1.0, 0.05, 0.2)
    with pytest.raises(AccelerationError):
        ol.AsianOption(100, 100, 1.0, 0.05, 0.2, seed=1).price(100, 8)
    with pytest.raises(AccelerationError):
        ol.simulate_gbm_hip(100, 1.0, 0.05, 0.2, 0.0, 100, 4, 1)
    assert ol.hip_available() is False and ol.HIP_AVAILABLE is False


def _build_c_example(tmp_path, name="price_from_c", extra=()):
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    _hip.load_library()                                   # builds libolmc.so if it is stale
    exe = str(tmp_path / name)
    pkg = os.path.join(root, "optionslab_amd")
    subprocess.run([cc, "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", *extra, "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", name + ".c"), "-o", exe, "-L" + pkg, "-lolmc", "-Wl,-rpath," + pkg], check=True)
    return exe


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: the example runs in the gpu suite")
def test_header_is_plain_c_and_a_c_host_fails_loudly_without_a_gpu(tmp_path):
    """include/olmc.h compiles as C99 with -Werror and links against libolmc.so; with no device the host gets a
    status code and a message, not a crash and not a CPU answer."""
    import subprocess

    out = subprocess.run([_build_c_example(tmp_path), "1000", "10"], capture_output=True, text=True)
    assert out.returncode == 1 and "olmc_init(0) failed" in out.stderr and "european" not in out.stdout


@pytest.mark.gpu
def test_c_host_example_prices_on_the_gpu(tmp_path):
    import re
    import subprocess

    out = subprocess.run([_build_c_example(tmp_path), "200000", "50"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    price = float(re.search(r"european call\s+price ([0-9.]+)", out.stdout).group(1))
    assert abs(price - 10.450583572185565) < 0.2 and "two shards" in out.stdout and "asian call" in out.stdout


def test_the_threaded_c_host_compiles_against_the_header(tmp_path):
    _build_c_example(tmp_path, "threads_from_c", extra=("-pthread",))


@pytest.mark.gpu
def test_c_host_threads_price_concurrently_with_identical_bits(tmp_path):
    import json
    import subprocess

    out = subprocess.run([_build_c_example(tmp_path, "threads_from_c", extra=("-pthread",)), "10000", "50", "0.3"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [r["threads"] for r in rows] == [1, 2, 4, 8, 16] and all(r["bit_identical_results"] for r in rows)
    assert max(r["calls_per_s"] for r in rows[1:]) > rows[0]["calls_per_s"]          # concurrent callers are not serialised


def test_the_kernels_refuse_a_finite_math_build():
    """dead lanes are zeroed by fmax(NaN, 0) = 0; -ffinite-math-only / -ffast-math would silently break that."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-ffast-math", "--cuda-device-only", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "optionslab_amd", "csrc"), os.path.join(ROOT, "optionslab_amd", "csrc", "olmc.hip")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "do not build with -ffinite-math-only" in r.stderr
