#!/bin/bash
# Resource usage of the device kernels of this build (cross-compiles; no GPU needed):
#   tools/kernel_meta.sh [regex on the demangled name]   ->  VGPRs, AGPRs, SGPRs, private segment, LDS, spills per kernel
# Leaves the device assembly in ${TMPDIR:-/tmp}/olmc_meta/olmc.s for a closer look.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=${TMPDIR:-/tmp}/olmc_meta
mkdir -p "$WORK" && cd "$WORK" || exit 1
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I"$ROOT/include" -I"$ROOT/optionslab_amd/csrc" -S --cuda-device-only \
    -o olmc.s "$ROOT/optionslab_amd/csrc/olmc.hip" 2>&1 | grep -v warning | head -20
python3 - "$1" <<'PY'
import re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ''
txt = open('olmc.s').read()
blocks = re.split(r"\n  - \.", txt[txt.rfind('amdhsa.kernels'):])
rows = []
for b in blocks[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, b) or [None, None])[1]
    name = g('name')
    if name is None:
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace('void olmc::', '')
    if pat and not re.search(pat, dem):
        continue
    rows.append((dem, g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('vgpr_spill_count')))
for r in sorted(rows):
    print("%-70s vgpr %4s agpr %3s sgpr %4s scratch %5s lds %6s spill %s" % r)
PY
