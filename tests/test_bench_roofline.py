"""bench.py's issue-cycle roofline and its inputs (no GPU needed): the static ISA mix that travels with the build, the
pricing of a counter set, and the fallbacks."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("olmc_bench", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_isa_mix_covers_every_kernel_the_bench_prices(bench):
    mix = bench.load_isa_mix()
    assert set(bench.PMC_KERNELS.values()) <= set(mix)
    assert "_stale" not in mix and mix["_sources_sha256"] == bench.device_sources_sha256()      # the mix belongs to THESE device sources
    for key in bench.PMC_KERNELS.values():
        m = mix[key]
        assert m["loop_valu_instructions"] == sum(m["by_class"].values())
        assert set(m["by_class"]) <= set(bench.ISSUE_PASSES), key            # every class has a price
        if key in ("f_qmc", "f_qmc_block"):                                                     # Sobol kernels: no Box-Muller; the rare tail is not in the mix
            # a trip is one dimension of a thread's eight points, or (aligned one-point kernel, round 5) two dimensions of one point
            points = 2 if key == "f_qmc" else 8
            assert m["steps_per_trip"] == (2 if key == "f_qmc" else 1) and "v_log_f32" not in m["by_class"] and m["cold_lines_skipped"] > 100
            # aligned forms (round 5): split workgroups keep the SIX low Gray bits as vector work (bits 6 .. 29 are folded for 64 dimensions at a
            # time, lane per dimension); eight points per thread keep bits 2 .. 8 (7) + the 7 in-block increments (v_xor: class "other")
            assert m["by_class"]["v_bitop3_b32"] == (12 if key == "f_qmc" else 7)
            # an inverse normal: 24 + 6 polynomial fma, 2 + 2 for the quotient, 1 p - p^2, 1 x, 1 accumulate, 1 ln m, 2 e ln 2 = 40 fma; 64 / 59 in all
            assert m["by_mnemonic"]["v_rcp_f64"] == points and m["by_mnemonic"]["v_fma_f64"] + m["by_mnemonic"]["v_fmac_f64"] == 40 * points
            assert m["loop_valu_instructions"] <= (64 if key == "f_qmc" else 60) * points
            assert m["vgprs"] <= (85 if key == "f_qmc" else 128)                               # six / four waves per SIMD
        elif key == "c4_asian_greeks14":                                     # four recursions per date (the two r bumps ride on the mid one): four table exponentials per normal
            assert m["steps_per_trip"] == 4 and m["by_class"]["v_ldexp_f64"] == 16 and m["by_class"]["v_rndne_f64"] == 16
        elif key == "f_heston":                                              # two normals per step
            assert m["steps_per_trip"] == m["by_class"]["v_log_f32"] == 2
        else:
            assert m["steps_per_trip"] in (4, 8, 16) and m["by_class"]["v_log_f32"] * 2 == m["steps_per_trip"]        # one Box-Muller log per two normals
    # round 5's entries: the fused barrier / lookback Greeks carry six recursions (x 2 legs) of {fma + add, max, min} per date, and fit four waves per SIMD
    eg, ega = mix["f_extrema_greeks14"], mix["f_extrema_greeks14a"]
    assert eg["by_class"]["v_fma_f64"] == 24 and ega["by_class"]["v_fma_f64"] == 48 and eg["vgprs"] <= 128 and ega["vgprs"] <= 128
    assert mix["f_cv"]["by_class"]["v_mad_u64_u32"] == 68           # the control-variate shape runs the headline step loop
    # two-path loops (autocallable, cliquet): the trip is the marked fast path, one Philox block of four dates; what the slow trips add is
    # priced at the minimum (rest_passes 2), so the fraction stays a bound
    for key in ("f_autocall", "f_cliquet"):
        assert mix[key]["steps_per_trip"] == 4 and mix[key]["by_class"]["v_mad_u64_u32"] == 17 and mix[key]["rest_passes"] == 2
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mix
    assert isa_mix._PASSES == bench.ISSUE_PASSES                    # the path search prices blocks with bench.py's own table
    # the headline loop: 17 multiplies and 19 XOR3 per Philox block, all but two of the XOR3 on three VGPRs (pinned round keys)
    c2 = mix["c2_european"]["by_class"]
    assert c2["v_mad_u64_u32"] == 68 and c2["v_bitop3_b32(v,v,v)"] + c2["v_bitop3_b32"] == 76 and c2["v_bitop3_b32"] <= 8
    # the reference-precision Asian loop carries the fp64 exponential, table form: per date 3 fma + 1 mul + 1 fma of the degree-4 polynomial
    # around the table entry + 1 fma of the cumsum = 6 fma-class, and the three integer ops of the table index (priced as `other`);
    # the fp32 one an exp per date
    a64 = mix["c4_asian_fp64"]["by_class"]
    assert a64["v_fma_f64"] == 24 and a64["v_ldexp_f64"] == 4 and a64["v_rndne_f64"] == 4 and a64["other"] in (8, 12)
    assert mix["c4_asian_fp32"]["by_class"]["v_exp_f32"] == 16


def test_roofline_pricing_is_bounded_and_uses_the_counters_it_says(bench):
    mix = bench.load_isa_mix()
    with open(os.path.join(ROOT, "profiles", "r02_pmc.json")) as f:
        pmc = json.load(f)
    c = pmc["c2_european"]
    t = 100.1e-6
    r = bench.roofline_for(pmc, "c2_european", t, 252, 1_000_000, None, mix, 2.25)
    loop = 1_000_000 / 64 * 15 * mix["c2_european"]["loop_valu_instructions"]
    cycles = 1_000_000 / 64 * 15 * sum(n * bench.ISSUE_PASSES[k] for k, n in mix["c2_european"]["by_class"].items()) + (c["SQ_INSTS_VALU"] - loop) * 4
    assert r["issue_cycles_needed_per_launch"] == pytest.approx(cycles, rel=1e-12)
    assert r["frac"] == pytest.approx(cycles / t / (1024 * 2.4e9), rel=1e-12) and 0.5 < r["frac"] <= 1.0
    assert r["peak"] == pytest.approx(2457.6) and r["achieved"] == pytest.approx(r["frac"] * r["peak"])
    assert r["frac_valu_active_pmc"] == pytest.approx(c["SQ_ACTIVE_INST_VALU"] * 4 / t / (1024 * 2.4e9))
    assert r["traffic"] == pytest.approx((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) and r["hbm_frac_of_8TBps"] < 1e-3
    assert "frac_vs_isolated_rates" not in r                                 # no probe costs were given
    # a kernel cannot be faster than its own speed of light: at that duration the fraction is exactly 1
    sol = bench.roofline_for(pmc, "c2_european", r["speed_of_light_kernel_ms"] / 1e3, 252, 1_000_000, None, mix)
    assert sol["frac"] == pytest.approx(1.0, rel=1e-12)
    # every committed kernel prices to a fraction in (0, 1] at the duration its own counters were collected next to
    for key, ms in (("c3_fused8", 0.1152), ("c3_fused14", 0.1313), ("c4_asian_fp64", 0.9408), ("c4_asian_fp64_antithetic", 1.4553),
                    ("c4_asian_fp32", 0.5433), ("c4_asian_fp32_antithetic", 0.6871)):
        n_steps = 252 if key.startswith("c3") else 1024
        rr = bench.roofline_for(pmc, key, ms / 1e3, n_steps, 1_000_000, None, mix)
        assert 0.6 < rr["frac"] <= 1.0, (key, rr["frac"])


def test_a_mix_of_other_device_sources_is_refused(bench, tmp_path, monkeypatch):
    """isa_mix.json carries the sha256 of the device sources it was read from; bench.py prices nothing with a mix of
    another build and says why."""
    import shutil
    fake = tmp_path / "repo"
    (fake / "optionslab_amd" / "csrc").mkdir(parents=True)
    for name in ("olmc.hip", "olmc_kernels.h", "olmc_host_math.h"):
        shutil.copy(os.path.join(ROOT, "optionslab_amd", "csrc", name), fake / "optionslab_amd" / "csrc" / name)
    shutil.copy(os.path.join(ROOT, "optionslab_amd", "isa_mix.json"), fake / "optionslab_amd" / "isa_mix.json")
    monkeypatch.setattr(bench, "ROOT", str(fake))
    assert "_stale" not in bench.load_isa_mix()
    with open(fake / "optionslab_amd" / "csrc" / "olmc_kernels.h", "a") as f:
        f.write("\n// a kernel changed\n")
    stale = bench.load_isa_mix()
    assert set(stale) == {"_stale"} and "run tools/isa_mix.py" in stale["_stale"]
    with open(os.path.join(ROOT, "profiles", "r02_pmc.json")) as f:
        pmc = json.load(f)
    r = bench.roofline_for(pmc, "c2_european", 1e-4, 252, 1_000_000, None, stale)
    assert r["frac"] is None and r["why_null"] == stale["_stale"] and r["frac_valu_active_pmc"] > 0


def test_roofline_fallbacks(bench):
    assert bench.roofline_for(None, "c2_european", 1e-4, 252, 1_000_000) is None
    assert bench.roofline_for({"c2_european": {"SQ_ACTIVE_INST_VALU": 1.0}}, "c2_european", None, 252, 1_000_000) is None
    r = bench.roofline_for({"c2_european": {"SQ_ACTIVE_INST_VALU": 6.4e7, "SQ_INSTS_VALU": 5.6e7}}, "c2_european", 1e-4, 252, 1_000_000, None, {})
    assert r["frac"] is None and r["traffic"] is None and r["frac_valu_active_pmc"] > 0      # no mix: only the counter formula
    committed = bench.committed_pmc()
    assert committed and committed["source"].startswith("COMMITTED profiles/") and "c2_european" in committed
