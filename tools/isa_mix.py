#!/usr/bin/env python3
"""Static VALU instruction mix of the hot loop of each kernel bench.py reports a roofline for.

    python tools/isa_mix.py            # hipcc -S the device code (gfx950), parse, write optionslab_amd/isa_mix.json

For every kernel in KERNELS the innermost loop that generates normals (it contains Box-Muller's v_log_f32) and has the most vector
instructions is taken as the hot loop (the step
loop: four Philox blocks per trip in the European kernel, one block per trip in the Asian ones) and its v_* instructions
are binned into the classes olmc_issue_probe measures on the device.  bench.py prices SQ_INSTS_VALU (live PMC) with this
mix and the live per-class issue costs: the issue-time roofline.  Cross-compiles; needs no GPU."""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "optionslab_amd")
OUT = os.path.join(PKG, "isa_mix.json")

KERNELS = {      # key in bench.py's JSON -> mangled-name regex
    "c2_european": r"^_ZN4olmc20european_path_kernelILi1ELb1ELi0ELb0EEE",
    "c3_fused8": r"^_ZN4olmc20european_path_kernelILi8ELb1ELi3ELb0EEE",          # MODE 3 = kSumOnly, the form MonteCarloPricer.greeks() launches
    "c3_fused14": r"^_ZN4olmc20european_path_kernelILi16ELb1ELi3ELb0EEE",
    "c4_asian_fp64": r"^_ZN4olmc18asian_exp64_kernelILb0EEE",
    "c4_asian_fp64_antithetic": r"^_ZN4olmc18asian_exp64_kernelILb1EEE",
    "c4_asian_fp32": r"^_ZN4olmc12asian_kernelILb0ELb0EEE",
    "c4_asian_fp32_antithetic": r"^_ZN4olmc12asian_kernelILb1ELb0EEE",
}

# mnemonic (encoding suffix stripped) -> probe class of optionslab_amd/_hip.py PROBE_CLASSES
CLASS_OF = {
    "v_mad_u64_u32": "v_mad_u64_u32", "v_bitop3_b32": "v_bitop3_b32(v,v,v)", "v_bitop3_b32(sgpr)": "v_bitop3_b32", "v_cvt_f32_u32": "v_cvt_f32_u32", "v_fmamk_f32": "v_fmamk_f32",
    "v_fmaak_f32": "v_fmamk_f32", "v_and_or_b32": "v_and_or_b32", "v_log_f32": "v_log_f32", "v_sqrt_f32": "v_sqrt_f32",
    "v_sin_f32": "v_sin_f32", "v_cos_f32": "v_cos_f32", "v_exp_f32": "v_exp_f32", "v_rcp_f32": "v_exp_f32", "v_rsq_f32": "v_exp_f32",
    "v_add_f32": "v_add_f32", "v_sub_f32": "v_add_f32", "v_subrev_f32": "v_add_f32", "v_mul_f32": "v_add_f32",
    "v_fma_f32": "v_fma_f32", "v_fmac_f32": "v_fma_f32", "v_pk_fma_f32": "v_fma_f32", "v_pk_add_f32": "v_fma_f32", "v_pk_mul_f32": "v_fma_f32",
    "v_cvt_f64_f32": "v_cvt_f64_f32", "v_add_f64": "v_add_f64", "v_fma_f64": "v_fma_f64", "v_fmac_f64": "v_fma_f64", "v_mul_f64": "v_fma_f64",
    "v_rndne_f64": "v_rndne_f64", "v_ldexp_f64": "v_ldexp_f64", "v_cvt_i32_f64": "v_cvt_i32_f64",
}
OTHER = "other"             # anything else (moves, integer adds, compares, selects): priced at the 2-cycle minimum by bench.py


def device_asm():
    tmp = tempfile.mkdtemp(prefix="olmc_isa_")
    out = os.path.join(tmp, "olmc.s")
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(PKG, "csrc"), "-S", "--cuda-device-only", "-o", out, os.path.join(PKG, "csrc", "olmc.hip")],
                   check=True, stderr=subprocess.PIPE, cwd=tmp)
    with open(out) as f:
        return f.read().splitlines()


def function_body(lines, pattern):
    rx = re.compile(pattern)
    start = next(i for i, l in enumerate(lines) if rx.match(l) and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))      # a kernel may hold several s_endpgm (early exits)
    return lines[start:end + 1]


def hot_loop(body):
    """Innermost loop (label ... backward branch to it) with the most vector instructions."""
    label_at = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB[0-9_]+):", l))}
    spans = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            spans.append((label_at[m.group(1)], i))
    inner = [s for s in spans if not any(o != s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]
    count = lambda s: sum(1 for l in body[s[0]:s[1]] if re.match(r"^\s+v_", l))
    steps = [s for s in inner if any("v_log_f32" in l for l in body[s[0]:s[1]])]      # the step loop generates normals (Box-Muller's log)
    return max(steps or inner, key=count)


def mix_of(body, span):
    ops = Counter()
    for l in body[span[0]:span[1]]:
        m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
        if not m:
            continue
        op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))
        if op == "v_bitop3_b32" and re.search(r"\bs\d+\b|\bs\[", l.split(op, 1)[1]):
            op = "v_bitop3_b32(sgpr)"        # an SGPR operand makes it a 4-cycle instruction; three VGPRs issue in ~2.6
        ops[op] += 1
    classes = Counter()
    for op, n in ops.items():
        classes[CLASS_OF.get(op, OTHER)] += n
    return ops, classes


def main():
    lines = device_asm()
    result = {"_what": "VALU instructions of each kernel's hot loop body (hipcc -S, gfx950), binned into the classes olmc_issue_probe measures; "
                       "generated by tools/isa_mix.py, regenerate after changing a kernel"}
    for key, pat in KERNELS.items():
        body = function_body(lines, pat)
        span = hot_loop(body)
        ops, classes = mix_of(body, span)
        vgprs = next((int(m.group(1)) for l in lines[lines.index(body[0]):] if (m := re.match(r"^; NumVgprs: (\d+)", l))), None)
        result[key] = {"loop_valu_instructions": sum(ops.values()), "steps_per_trip": 2 * ops.get("v_log_f32", 0), "by_class": dict(sorted(classes.items())), "by_mnemonic": dict(sorted(ops.items())),
                       "unclassified_priced_as_" + OTHER: sum(n for op, n in ops.items() if op not in CLASS_OF), "vgprs": vgprs}
        print(f"{key:28s} loop {sum(ops.values()):4d} VALU instr, {vgprs} VGPRs: {dict(classes)}", file=sys.stderr)
    with open(OUT, "w") as f:
        json.dump(result, f, indent=1)
    print(OUT)


if __name__ == "__main__":
    main()
