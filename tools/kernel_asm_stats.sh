#!/bin/bash
# Dev tool: compile deepmod_hip.hip to gfx950 assembly (with extra -D flags) and print the instruction mix, register use
# and spills of one kernel.   bash tools/kernel_asm_stats.sh [kernel-name-substring] [-DFLAG ...]
K=${1:-bilstm_f16s}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
S=$(mktemp /tmp/dmasm.XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S -o $S "$@" $R/deepmod_amd/csrc/deepmod_hip.hip 2>/dev/null || { echo "compile failed"; exit 1; }
awk -v k="$K" '/^_Z[A-Za-z0-9_]*:/ {on = index($0, k) > 0} on {print} on && /s_endpgm/ {exit}' $S > $S.k
echo "instructions: $(grep -E '^\s+[a-z]' $S.k | grep -v '^\s*;' | grep -vE '^\s+\.' | wc -l)"
grep -E "^\s+[vsdg][a-z_0-9]+" -o $S.k | sed 's/^\s*//' | sort | uniq -c | sort -rn | head -${TOP:-16}
awk -v k="$K" '$0 ~ "\\.name:.*" k {on=1} on && /vgpr_count|spill_count|sgpr_count/ {print} on && /wavefront_size/ {exit}' $S
[ -n "$KEEP" ] && cp $S.k $KEEP
rm -f $S $S.k
