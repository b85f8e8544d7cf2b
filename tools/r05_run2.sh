# round 5, second GPU pass: new tests, host spans with per-launcher stamps, A/Bs, every entry point re-measured
set -x
D=gpurun_out/r05b; mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_gpu_instrumented.py tests/test_gpu_multi_device.py tests/test_gpu_exotics.py -m gpu -x -q > $D/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $D/pytest.log
tail -5 $D/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/measure_multi_enqueue.py > $D/multi_enqueue.jsonl 2> $D/multi_enqueue.err; echo "enqueue rc $?"
cut -c1-330 $D/multi_enqueue.jsonl
A=tools/ab/libolmc_r05_relaxed_acquire.so; B=optionslab_amd/libolmc.so; R4=tools/ab/libolmc_r04.so; CP=tools/ab/libolmc_r05_copy.so
{
echo "== American option again: the default build, a byte-for-byte COPY of it under another name, and the relaxed-acquire build (the per-date kernels are identical instruction for instruction in all three)"
timeout -k 10 300 python3 tools/ab_libs.py $B $CP $A --case american --n 50000 --m 50 --rounds 5 || exit 1
echo "== round 5: fused barrier / lookback Greeks, per-contract wave sums + one-copy tail (94 / 120 VGPRs) against round 4's final build (134 / 170 / 172 VGPRs), 1M x 252, 5 rounds"
for c in barrier_greeks14_anti barrier_greeks14 barrier_greeks8_anti lookback_greeks14_anti lookback_greeks8; do
  echo "== $c 1000000 252"; timeout -k 10 200 python3 tools/ab_libs.py $R4 $B --case $c --n 1000000 --m 252 --rounds 5 || exit 1
done
} > $D/ab.txt 2> $D/ab.err
cut -c1-200 $D/ab.txt
timeout -k 10 600 python3 tools/measure_configs.py > $D/configs.jsonl 2> $D/configs.err; echo "configs rc $?"
python3 - <<'PY'
import json
for l in open('gpurun_out/r05b/configs.jsonl'):
    d=json.loads(l)
    if 'config' in d: print(f"{d['config'][:100]:100s} wall {d['wall_ms_median']:9.3f} ms  kernel {d['kernel_us']:9.1f} us x{d['kernel_launches_per_call']}")
PY
