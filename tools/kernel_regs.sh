#!/bin/bash
# VGPRs / spills / occupancy of the kernels of one .hip file whose mangled name matches a pattern.
#   tools/kernel_regs.sh [pattern] [file]       e.g. tools/kernel_regs.sh 'k_final_fast|k_down_march'
cd "$(dirname "$0")/../super-resolution-system_amd" || exit 1
pat=${1:-.}
src=${2:-csrc/sr_engine.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DSR_BUILD \
    -I ../include -I csrc $SR_HIPCC_EXTRA -x hip -c "$src" -o /tmp/kernel_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk '/Function Name:/ {name=$5} /  VGPRs:/ {v=$4} /VGPRs Spill:/ {sp=$5} /SGPRs Spill:/ {ss=$5} /Occupancy/ {occ=$5} /LDS Size/ {printf "%-90s vgpr %3s  spill v%s s%s  waves %s\n", name, v, sp, ss, occ}' |
    grep -E "$pat"
