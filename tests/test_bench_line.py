"""The ONE stdout line of bench.py (no GPU needed).  Round 2's line grew to 29 KB and the driver, which keeps an 8 KB tail
of stdout, could not parse it: `compact_line` is a pure function of the full record, and this file holds it to its budget on
canned records -- round 2's own 29 KB record and a synthetic 8-GPU one -- and checks the launcher's control flow."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline")


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("olmc_bench_line", os.path.join(ROOT, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def _r02_record():
    with open(os.path.join(ROOT, "profiles", "r02_bench.json")) as f:
        return json.load(f)


def _check_line(bench, line):
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.LINE_BUDGET < 4096, len(text)
    back = json.loads(text)
    assert back == line
    for k in CONTRACT_KEYS:
        assert k in back, k
    assert set(back["config"]) == {"workload", "paths_per_gpu", "n_steps", "global_paths", "parallelism"}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    return back


def test_the_29_kb_record_of_round_2_becomes_a_line_under_3_kb(bench):
    full = _r02_record()
    assert len(json.dumps(full)) > 25_000                       # the record that broke the driver's parser
    line = _check_line(bench, bench.compact_line(full, "bench_detail.json"))
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
    assert line["roofline"]["avg_kernel_ms"] <= line["ms_per_step"]
    cpu = line["cpu_baseline"]
    assert set(cpu) >= {"value", "unit", "cores", "kind", "sample"} and cpu["kind"] == "port" and cpu["cores"] == 1
    assert line["gpu_over_cpu"] == pytest.approx(full["value"] / full["cpu_baseline"]["value"], rel=1e-4)
    # configs[2], [3], [4] ride as {value, ms, frac}
    assert set(line["c3"]) == {"fused_8", "fused_14", "literal_8"} and 0 < line["c3"]["fused_14"]["frac"] <= 1
    assert set(line["c4"]) >= {"fp64", "fp64_antithetic"} and line["c4"]["fp64"]["ms"] == pytest.approx(full["c4_asian"]["fp64"]["ms_per_call"], rel=1e-4)
    assert set(line["c5"]) == {"weak", "strong"} and line["c5"]["weak"]["paths_per_gpu"] == 8_000_000
    assert line["detail"] == "bench_detail.json"
    # none of the tables that bloated round 2's line
    text = json.dumps(line)
    for banned in ("issue_costs_ns", "issue_model", "issue_costs_cycles_at_held_clock", "loop_mix", "seconds_each", "what"):
        assert banned not in text, banned


def test_a_multi_gpu_record_with_errors_and_long_strings_stays_in_budget(bench):
    full = _r02_record()
    full.update(n_gpus=8, ranks_seen=8)
    full["config"]["workload"] = "x" * 5000
    full["config"]["parallelism"] = "y" * 5000
    full["roofline"]["pmc_source"] = "z" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["errors"] = ["e" * 1000] * 50
    full["c3_greeks"] = {"error": "RuntimeError: " + "boom " * 400}
    full["n1_basis"] = dict(full["c5_weak"])
    full["c2_1m_per_gpu"] = dict(full["c5_weak"], paths_per_gpu=1_000_000)
    line = _check_line(bench, bench.compact_line(full, "bench_detail.json"))
    assert line["n_gpus"] == 8 and len(line["errors"]) <= 4 and all(len(e) <= 120 for e in line["errors"])
    assert "c3" not in line and any(e.startswith("c3_greeks:") for e in line["errors"])
    assert line["n1_basis"]["value"] == pytest.approx(full["c5_weak"]["value"], rel=1e-5)


def test_a_bare_record_still_gives_the_contract_keys(bench):
    full = {"metric": "m", "value": 1.0, "unit": "path-steps/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 normals / f64 prices", "data": "synthetic", "config": {"workload": "w"},
            "roofline": {"bound": "valu", "frac": None, "why_null": "rocprofv3 not found"}}
    line = _check_line(bench, bench.compact_line(full))
    assert line["roofline"]["frac"] is None and line["roofline"]["pmc_source"] == "rocprofv3 not found" and "cpu_baseline" not in line
    assert bench._num(float("nan")) is None and bench._num(1.23456789012, 4) == 1.235


def test_gpus_n_without_a_launcher_starts_its_own_ranks_and_relays_line_and_exit_code(bench, tmp_path, monkeypatch):
    """`python3 bench.py --gpus 8` the way the driver calls `--gpus 1`: the parent starts torch.distributed.run as a child and
    relays the line.  Here the child is a stand-in (no GPU): the parent's control flow is what is under test."""
    calls = {}

    class Done:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    def fake_run(cmd, stdout=None, env=None, timeout=None):
        calls["cmd"], calls["env"] = cmd, env
        return Done(calls["rc"], calls["out"])

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1"])
    args = type("A", (), {"gpus": 8})()
    calls.update(rc=0, out=b'RCCL banner\n{"metric":"m","value":1}\n')
    assert bench.launch_ranks(args) == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    calls.update(rc=3, out=b'{"metric":"m","value":1,"errors":["section `c5_strong` did not finish"]}\n')
    assert bench.launch_ranks(args) == 3                         # the watchdog's exit code travels
    calls.update(rc=0, out=b"no line here\n")
    assert bench.launch_ranks(args) == 4                         # rc 0 without a line is not success


def test_the_launcher_path_is_taken_before_anything_touches_a_gpu():
    """--gpus 2 with WORLD_SIZE unset and a torch.distributed.run that cannot start ranks (no GPU here): the parent must fail by its
    CHILD's exit code, not by an exception of its own, and must not import torch itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-pmc', '--no-cpu-baseline'];\n"
            "import subprocess\n"
            "real = subprocess.run\n"
            "def fake(cmd, **kw):\n"
            "    assert 'torch' not in sys.modules, 'the parent imported torch'\n"
            "    print('SPAWN', cmd[1:3], file=sys.stderr)\n"
            "    class R: returncode = 7; stdout = b''\n"
            "    return R()\n"
            "subprocess.run = fake\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])
    assert "SPAWN ['-m', 'torch.distributed.run']" in r.stderr and r.stdout.strip() == ""


def test_the_single_process_child_waits_for_go_and_its_line_is_relayed(bench, monkeypatch):
    """c5.single_process: rank 0 starts `bench.py --single-process-child` BEFORE it touches a GPU; the child sits on its stdin, GPU
    untouched, until told to go; its JSON object becomes the section.  A stand-in child here (no GPU): the control flow is under test."""
    # the real child, told something else than "go", leaves without importing the engine
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process-child", "--gpus", "2"], input="no\n", capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == ""
    seen = {}

    class FakeChild:
        returncode = 0

        def communicate(self, text, timeout=None):
            seen["sent"] = text
            return 'banner\n{"value": 5.0e12, "ms_per_step": 0.8, "n_gpus": 8, "paths_per_gpu": 8000000}\n', ""

        def poll(self):
            return 0

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return FakeChild()

    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "8")
    args = type("A", (), {"steps": 100, "warmup": 10, "paths_per_gpu": 0})()
    child = bench.start_single_process_child(args, 8)
    assert seen["cmd"][1:5] == [os.path.join(ROOT, "bench.py"), "--single-process-child", "--gpus", "8"]
    assert "RANK" not in seen["env"] and "WORLD_SIZE" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    res = bench.c5_single_process(child, None, False, False, 1, 0, None)
    assert seen["sent"] == "go\n" and res["n_gpus"] == 8 and res["value"] == 5.0e12
    line = bench.compact_line({"metric": "m", "value": 1.0, "config": {}, "roofline": {}, "c5_single_process": res})
    assert line["c5"]["single_process"] == {"value": 5.0e12, "ms": 0.8, "n_gpus": 8, "paths_per_gpu": 8000000}
    assert bench.c5_single_process(None, None, False, True, 1, 0, None)["error"].startswith("rehearsal")


def test_round_4_record_keeps_every_section_within_the_budget(bench):
    """The round-4 record (profiles/r04_bench_detail.json: f_kernels, c2_f64_normals, c5_single_process on top of round 3's sections) must
    fit the line WITHOUT the belt-and-braces shedding: every section the record holds appears in the line."""
    with open(os.path.join(ROOT, "profiles", "r04_bench_detail.json")) as f:
        full = json.load(f)
    line = _check_line(bench, bench.compact_line(full, "bench_detail.json"))
    assert len(json.dumps(line, separators=(",", ":"))) <= bench.LINE_BUDGET - 64
    assert set(line["f"]) == {"heston", "extrema", "asian_geometric", "multi", "qmc", "qmc_one_point", "qmc_block", "american_lsm"}
    # round 5: every entry is [ms, frac] (six more kernels had to fit the line)
    assert all(v[0] > 0 and 0 < v[1] <= 1 for k, v in line["f"].items() if k != "american_lsm") and set(line["f"]["american_lsm"]) == {"50000x50", "1000000x50"}
    assert line["c2_f64_normals"]["x_product_kernel"] > 3 and line["c5"]["single_process"]["n_gpus"] == 1
    assert {"c3", "c4", "pipelined"} <= set(line)
