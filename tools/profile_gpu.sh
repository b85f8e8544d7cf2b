#!/bin/bash
# Runs on the GPU box (via gpurun): the evidence behind bench.py's numbers, written under gpurun_out/prof_<tag>/.
#   stats        rocprofv3 --kernel-trace --stats of the bench command restricted to the headline pass
#                (--no-extras: every european_path_kernel<1> dispatch is a 1M x 252 launch, so the table's average
#                is the figure bench.py's roofline.avg_kernel_ms must agree with)
#   stats_full   the same for the full default command (C3 / C4 / C5 kernels appear with their own rows)
#   bench.json   the plain command, unprofiled, with its live PMC passes kept (pmc/: rocprofv3 --pmc CSVs + pmc.json);
#                every pass leaves its full record next to its one-line JSON (<pass>.detail.json)
# Usage: tools/profile_gpu.sh <tag> ; then  python tools/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<name>.txt
set -e -o pipefail
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
  local name=$1; shift
  export OLMC_BENCH_DETAIL="$OUT/$name.detail.json"      # the full record behind the one-line JSON (exported: nothing but the program may follow `--`)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" "$@" > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
run stats --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-extras
run stats_full --steps 20 --warmup 5 --no-pmc --no-cpu-baseline
export OLMC_BENCH_DETAIL="$OUT/bench.detail.json"
timeout -k 10 400 python3 "$ROOT/bench.py" --steps 20 --warmup 5 --pmc-keep "$OUT/pmc" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"
