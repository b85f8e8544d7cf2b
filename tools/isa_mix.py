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
    "c4_asian_greeks14": r"^_ZN4olmc25asian_exp64_greeks_kernelILb0ELi16EEE",   # round 4: 14 contracts, six path recursions, one launch
    # round 4: the SURVEY 8(f) kernels
    "f_asian_geometric": r"^_ZN4olmc12asian_kernelILb0ELb1EEE",
    "f_extrema": r"^_ZN4olmc14extrema_kernelILb0EEE",
    "f_heston": r"^_ZN4olmc13heston_kernelILb0EEE",
    "f_multi": r"^_ZN4olmc21european_multi_kernelILb1EEE",
    "f_qmc": r"^_ZN4olmc19european_qmc_kernelILi0ELb1ELb1EEE",       # SPLIT, ALIGNED: what a launch below 2^20 points runs (bench.py's f_qmc workload: 2^17 x 252)
    "f_qmc_block": r"^_ZN4olmc25european_qmc_block_kernelILi0ELb1EEE",
    # round 5: round 4's new kernels (VERDICT r4 "missing" 2) and the control-variate shape of the headline kernel
    "f_extrema_greeks14": r"^_ZN4olmc21extrema_greeks_kernelILb0ELi16EEE",
    "f_extrema_greeks14a": r"^_ZN4olmc21extrema_greeks_kernelILb1ELi16EEE",
    "f_geo_greeks14": r"^_ZN4olmc29asian_geometric_greeks_kernelILb0ELi16EEE",
    "f_autocall": r"^_ZN4olmc15autocall_kernelILb0EEE",
    "f_cliquet": r"^_ZN4olmc14cliquet_kernelILb0EEE",
    "f_cv": r"^_ZN4olmc20european_path_kernelILi1ELb1ELi2ELb0EEE",
}
# steps (monitoring dates, Sobol dimensions) one trip of the hot loop advances a path by, where it is not "two per Box-Muller":
# Heston consumes TWO normals per step; a Sobol kernel's trip is one dimension of a thread's eight points, or -- the aligned one-point
# kernel since round 5 -- TWO dimensions of one point (one v_rcp_f64 per inverse normal)
STEPS_PER_TRIP = {"f_heston": lambda ops: ops.get("v_log_f32", 0), "f_qmc": lambda ops: ops.get("v_rcp_f64", 0), "f_qmc_block": lambda ops: 1}

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


def basic_blocks(body):
    """[(first line, one past the last line, own label or None, header label of the innermost loop it belongs to or None, is that
    loop's header, is an INNER loop's header)] from the compiler's own block comments (`.LBBn_m:` / `; %bb.k:` lines carry
    "in Loop: Header=BBn_m" / "This Inner Loop Header")."""
    starts = [i for i, l in enumerate(body) if re.match(r"^(\.LBB[0-9_]+:|; %bb\.\d+:)", l)]
    blocks = []
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(body)
        line = body[i]
        for j in range(i + 1, min(i + 4, end)):          # the annotation of a nested loop's header continues on comment-only lines
            if re.match(r"^\s*;", body[j]):
                line += " " + body[j]
            else:
                break
        own = (m.group(1) if (m := re.match(r"^\.L(BB[0-9_]+):", line)) else None)
        is_header = "Loop Header" in line
        member = re.search(r"in Loop: Header=(BB[0-9_]+)", line)
        loop = own if is_header else (member.group(1) if member else None)
        blocks.append((i, end, own, loop, is_header, is_header and "Inner Loop Header" in line))
    return blocks


def hot_loop(body):
    """Blocks of the innermost loop with the most vector instructions among those that generate normals (Box-Muller's v_log_f32), or,
    for kernels without one (the Sobol kernels), among all innermost loops."""
    blocks = basic_blocks(body)
    inner = [b[3] for b in blocks if b[5]]
    members = {h: [b for b in blocks if b[3] == h] for h in inner}
    count = lambda h: sum(1 for b in members[h] for l in body[b[0]:b[1]] if re.match(r"^\s+v_", l))
    steps = [h for h in inner if any("v_log_f32" in l for b in members[h] for l in body[b[0]:b[1]])]
    return members[max(steps or inner, key=count)]


# Kernels whose step loop holds a FAST path (Philox blocks without an observation / reset date) and a slow one behind wave-uniform
# branches: the loop's blocks together are not what one trip executes.  The kernel source marks its fast path with an assembly
# comment (`; olmc_fast_trip`, no instruction); a trip is priced along the cheapest path from the loop header THROUGH the marked
# block to the back edge -- what the fast trips execute.  A slow trip executes more; that reaches the model through the counters
# (SQ_INSTS_VALU minus trips x this path) at the 2-cycle minimum, so the fraction stays a bound (bench.py: "rest_passes").
MIN_PATH = {"f_autocall", "f_cliquet"}
FAST_MARK = "olmc_fast_trip"
_PASSES = {"v_mad_u64_u32": 4, "v_bitop3_b32": 4, "v_bitop3_b32(v,v,v)": 2, "v_cvt_f32_u32": 4, "v_fmamk_f32": 2, "v_and_or_b32": 4,
           "v_log_f32": 8, "v_sqrt_f32": 8, "v_sin_f32": 8, "v_cos_f32": 8, "v_exp_f32": 8, "v_add_f32": 2, "v_fma_f32": 2,
           "v_cvt_f64_f32": 4, "v_add_f64": 4, "v_fma_f64": 4, "v_rndne_f64": 4, "v_ldexp_f64": 4, "v_cvt_i32_f64": 4, "other": 2}     # = bench.ISSUE_PASSES (tests/test_bench_roofline.py)


def cheapest_trip(body, blocks):
    """Blocks of the loop on the path of least issue cycles header -> marked block -> back edge (Dijkstra over the loop's own CFG:
    `s_cbranch_* L` -> {L, fall-through}, `s_branch L` -> {L}, else fall-through)."""
    import heapq
    label_of = {b[2]: k for k, b in enumerate(blocks) if b[2]}
    header = blocks[0][2]
    marked = [k for k, (first, end, *_r) in enumerate(blocks) if any(FAST_MARK in l for l in body[first:end])]
    if len(marked) != 1:
        raise RuntimeError(f"expected one block marked `; {FAST_MARK}` in the loop, found {len(marked)}")
    cost, succ = [], []
    for k, (first, end, *_rest) in enumerate(blocks):
        _ops, classes, _cold = mix_of(body, [blocks[k]])
        cost.append(sum(n * _PASSES[c] for c, n in classes.items()))
        out, falls = [], True
        for l in body[first:end]:
            m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+\.L(BB[0-9_]+)", l)
            if m:
                out.append(m.group(2))
                if m.group(1) == "s_branch":
                    falls = False
        if falls and k + 1 < len(blocks):
            out.append(k + 1)
        succ.append([("back" if t == header else label_of.get(t)) if isinstance(t, str) else t for t in out])

    def shortest(src, dst):
        best, heap, done = {src: (cost[src], [src])}, [(cost[src], src)], set()
        while heap:
            d, k = heapq.heappop(heap)
            if k in done:
                continue
            done.add(k)
            if k == dst:
                return best[k][1]
            for j in succ[k]:
                if j == "back":
                    if dst == "back":
                        return best[k][1]
                    continue
                if j is None:
                    continue
                nd = d + cost[j]
                if j not in best or nd < best[j][0]:
                    best[j] = (nd, best[k][1] + [j])
                    heapq.heappush(heap, (nd, j))
        raise RuntimeError("no such path in the loop")

    path = shortest(0, marked[0]) + shortest(marked[0], "back")[1:]
    return [blocks[i] for i in path]


def mix_of(body, blocks):
    """VALU instructions of the loop's blocks by mnemonic and by probe class.  A block that holds an fp64 square root is COLD: it is the
    tail of the inverse normal (0.1 % of the Sobol points, olmc_kernels.h ndtri_tail), skipped by a branch in nearly every wave."""
    ops = Counter()
    n_cold = 0
    for first, end, *_ in blocks:
        if any(re.search(r"v_rsq_f64|v_sqrt_f64", l) for l in body[first:end]):
            n_cold += end - first
            continue
        for l in body[first:end]:
            m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
            if not m:
                continue
            op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))
            if op == "v_bitop3_b32" and re.search(r"\bs\d+\b|\bs\[", l.split(op, 1)[1]):
                op = "v_bitop3_b32(sgpr)"        # an SGPR operand makes it a 4-cycle instruction; three VGPRs issue in ~2.6
            ops[op] += 1
    classes = Counter()
    for op, n in ops.items():
        # any fp64 operation without a probe class of its own (v_max_f64, v_cmp_*_f64, v_frexp_*_f64, v_rcp_f64 ...) is priced as the
        # cheapest fp64 class (4 passes = the fp64 vector peak): never below what it costs
        classes[CLASS_OF.get(op, "v_add_f64" if re.search(r"_f64$|_f64_", op) else OTHER)] += n
    return ops, classes, n_cold


def source_digest():
    """sha256 over the device sources the mix was read from: bench.py refuses a mix that belongs to another build."""
    import hashlib
    h = hashlib.sha256()
    for name in ("olmc.hip", "olmc_kernels.h", "olmc_host_math.h"):
        with open(os.path.join(PKG, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def main():
    lines = device_asm()
    result = {"_what": "VALU instructions of each kernel's hot loop body (hipcc -S, gfx950), binned into the classes olmc_issue_probe measures; "
                       "generated by tools/isa_mix.py, regenerate after changing a kernel"}
    missing = []
    for key, pat in KERNELS.items():
        try:
            body = function_body(lines, pat)
        except StopIteration:           # a kernel was renamed / re-templated: ITS mix is absent (bench.py then prints no fraction for it), the others stay fresh
            missing.append(key)
            print(f"{key:28s} NO KERNEL MATCHES {pat}", file=sys.stderr)
            continue
        span = hot_loop(body)
        if key in MIN_PATH:
            span = cheapest_trip(body, span)
        ops, classes, n_cold = mix_of(body, span)
        vgprs = next((int(m.group(1)) for l in lines[lines.index(body[0]):] if (m := re.match(r"^; NumVgprs: (\d+)", l))), None)
        steps = STEPS_PER_TRIP[key](ops) if key in STEPS_PER_TRIP else 2 * ops.get("v_log_f32", 0)
        result[key] = {"loop_valu_instructions": sum(ops.values()), "steps_per_trip": steps, "by_class": dict(sorted(classes.items())), "by_mnemonic": dict(sorted(ops.items())),
                       "unclassified_priced_as_" + OTHER: sum(n for op, n in ops.items() if op not in CLASS_OF and not re.search(r"_f64$|_f64_", op)),
                       "cold_lines_skipped": n_cold, "vgprs": vgprs}
        if key in MIN_PATH:
            result[key]["rest_passes"] = 2
            result[key]["trip"] = "cheapest path through the loop (the fast trips); slower trips reach the model through SQ_INSTS_VALU at 2 cycles an instruction"
        print(f"{key:28s} loop {sum(ops.values()):4d} VALU instr, {vgprs} VGPRs: {dict(classes)}", file=sys.stderr)
    result["_sources_sha256"] = source_digest()
    with open(OUT, "w") as f:
        json.dump(result, f, indent=1)
    print(OUT)
    if missing:
        raise SystemExit(f"tools/isa_mix.py: no kernel for {missing}: update KERNELS (the other mixes were written)")


if __name__ == "__main__":
    main()
