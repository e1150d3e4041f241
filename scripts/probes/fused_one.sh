#!/bin/bash
# Register / spill figures of ONE instantiation of the one-launch solve (the headline's: six blocks, plain Gauss-Newton, lazy
# arguments; MMX_PROBE_RULE=1 / -1: LM schedule / generic rule) in half a minute instead of a five-minute group build:
#   bash scripts/probes/fused_one.sh [extra hipcc flags ...]        e.g.  -DMMX_PROBE_RULE=1  -mllvm -sink-insts-to-avoid-spills=1
cd "$(dirname "$0")/../../momentum_amd/csrc" || exit 1
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMMX_FUSED_GROUP=9 "$@" -c mmx_fused.hip -o $tmp/one.o -Rpass-analysis=kernel-resource-usage 2> $tmp/one.txt
grep -E "error|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill" $tmp/one.txt | sed 's/.*remark: //; s/ \[-Rpass.*//' | tr '\n' ' '; echo
[ -n "$KEEP_ASM" ] && hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMMX_FUSED_GROUP=9 "$@" -S --cuda-device-only mmx_fused.hip -o "$KEEP_ASM" 2> /dev/null
rm -rf $tmp
