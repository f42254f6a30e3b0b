#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the bench command.
# Usage: bash tools/profile_round.sh <tag> [f16x3|f16x3r|f32]     -> gpurun_out/prof_<tag>/...
TAG=${1:-final}
PREC=${2:-f16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --precision $PREC"
BKT="python $R/bench.py --steps 32 --warmup 1 --no-cpu-baseline --no-extras --precision $PREC"      # 32 + 1 untimed, 32 timed launches
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BKT > $OUT/kt.log 2>&1; echo "kt rc=$?"
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $B > $OUT/pmc_$N.log 2>&1; echo "$N rc=$?"
done
