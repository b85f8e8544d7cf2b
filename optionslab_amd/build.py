"""Builds libolmc.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SOURCES = [os.path.join(PKG, "csrc", "olmc.hip")]
HEADERS = [os.path.join(PKG, "csrc", "olmc_kernels.h"), os.path.join(PKG, "csrc", "olmc_host_math.h"), os.path.join(PKG, "csrc", "olmc_job_board.h"),
           os.path.join(ROOT, "include", "olmc.h")]
LIBRARY = os.path.join(PKG, "libolmc.so")
# the instrumented build (include/olmc_probe.h): the same translation unit with its test seams compiled in + the measurement kernels.
# Test / measurement infrastructure: lives under tools/, loaded by tests, tools and bench.py's calibration, never by the package.
PROBE_DIR = os.path.join(ROOT, "tools", "probe")
PROBE_SOURCES = [os.path.join(PROBE_DIR, "olmc_probe.hip")]
PROBE_HEADERS = [os.path.join(PROBE_DIR, "olmc_probe_kernels.h"), os.path.join(ROOT, "include", "olmc_probe.h")]
PROBE_LIBRARY = os.path.join(PROBE_DIR, "libolmc_probe.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale() -> bool:
    if not os.path.exists(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS)


def _compile(sources, output, extra_includes=(), verbose=False):
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), *["-I" + d for d in extra_includes],
           "-o", output, *sources, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)


def probe_is_stale() -> bool:
    if not os.path.exists(PROBE_LIBRARY):
        return True
    built = os.path.getmtime(PROBE_LIBRARY)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS + PROBE_SOURCES + PROBE_HEADERS)


def build_probe_library(force: bool = False, verbose: bool = False) -> str:
    """libolmc_probe.so (None when tools/probe is absent, e.g. in a packaged copy of the product)."""
    if not os.path.exists(PROBE_SOURCES[0]):
        return None
    if force or probe_is_stale():
        _compile(PROBE_SOURCES, PROBE_LIBRARY, extra_includes=(PROBE_DIR,), verbose=verbose)
    return PROBE_LIBRARY


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIBRARY
    _compile(SOURCES, LIBRARY, verbose=verbose)
    # the static instruction mix of the step loops (bench.py's issue-cycle roofline) belongs to this very build.  It is a
    # by-product: the library above is complete without it, so a failure here (tools/ absent in a packaged copy, a symbol
    # pattern that no longer matches after a kernel was renamed) is reported and leaves the previous isa_mix.json in place
    mix = os.path.join(ROOT, "tools", "isa_mix.py")
    if os.path.exists(mix):
        r = subprocess.run([sys.executable, mix], env=dict(os.environ, HIPCC=_hipcc()), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print(f"[optionslab_amd.build] WARNING: tools/isa_mix.py failed (rc {r.returncode}).  The library is built; bench.py will print NO roofline "
                  f"fraction for the kernels whose mix is missing or belongs to other sources (it checks the sha256 in isa_mix.json):\n"
                  f"{r.stderr[-1500:]}", file=sys.stderr)
    return LIBRARY


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_probe_library(force=True, verbose=True))
