# round 5, first GPU pass: the GPU suite, the multi-GPU engine's host spans (rehearsed ranks), the acquire A/B, the fused extrema Greeks A/B
set -x
D=gpurun_out/r05a; mkdir -p $D
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $D/pytest.log
tail -15 $D/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/measure_multi_enqueue.py > $D/multi_enqueue.jsonl 2> $D/multi_enqueue.err; echo "enqueue rc $?"
cat $D/multi_enqueue.jsonl | cut -c1-260
A=tools/ab/libolmc_r05_relaxed_acquire.so; B=optionslab_amd/libolmc.so; R4=tools/ab/libolmc_r04_agent_acquire.so
{
echo "== round 5: the grid reduction's consumer side is the agent-scope acquire again (default); libolmc_r05_relaxed_acquire.so = the same sources with -DOLMC_AGENT_ACQUIRE=0 (round 4's wavefront-scope fence), 5 rounds"
for c in "european 10000 50" "european 1000000 252" "european 8000000 252" "greeks14_lean 1000000 252" "american 50000 50" "barrier_greeks14_anti 10000 50"; do
  set -- $c; echo "== $1 $2 $3"; timeout -k 10 200 python3 tools/ab_libs.py $A $B --case $1 --n $2 --m $3 --rounds 5 || exit 1
done
echo "== round 5: fused barrier / lookback Greeks, per-contract wave sums + one-copy tail (120 VGPRs) against round 4's build (172 / 170 / 134 VGPRs), 1M x 252, 5 rounds"
for c in barrier_greeks14_anti barrier_greeks14 barrier_greeks8_anti lookback_greeks14_anti lookback_greeks8; do
  echo "== $c 1000000 252"; timeout -k 10 200 python3 tools/ab_libs.py $R4 $B --case $c --n 1000000 --m 252 --rounds 5 || exit 1
done
} > $D/ab.txt 2> $D/ab.err
cat $D/ab.txt | cut -c1-200
