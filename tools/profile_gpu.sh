#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
# Outputs land in gpurun_out/prof_<tag>/{stats,pmc_sq,pmc_sq2,pmc_fetch,pmc_write}; copy summaries to profiles/.
set -e -o pipefail
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --streams 1: launches do not overlap, so per-dispatch durations and counters are one kernel's own
ARGS=${@:---steps 40 --warmup 4 --streams 1 --no-cpu-baseline}
run() {  # name, rocprof args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
run stats --kernel-trace --stats
ARGS_SAVE=$ARGS; ARGS="--steps 200 --warmup 10 --no-cpu-baseline"      # the default (8-stream) command, for the record
run stats_default --kernel-trace --stats
ARGS=$ARGS_SAVE
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run pmc_sq2 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
