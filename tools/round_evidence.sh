#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/<round>/ cites, into gpurun_out/<round>/.   bash tools/round_evidence.sh r06
# (the LAST kernel-touching act of a round: bench.py marks a committed PMC summary stale once the kernel source changes)
RND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$RND
mkdir -p $O
python bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err
python bench.py --precision f16i8 --no-cpu-baseline --no-extras > $O/bench_f16i8.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline --no-extras > $O/bench_f32.json 2>/dev/null
python bench.py --weights trained-like --precision auto > $O/bench_trained_like_auto.json 2>/dev/null
DM_BENCH_FORCE_DIST=1 MASTER_PORT=29999 python bench.py --steps 8 --no-cpu-baseline --no-extras > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
for P in f16x3 f16i8 f32; do
  bash tools/profile_round.sh ${RND}_$P $P >> $O/profile.log 2>&1
  python tools/summarize_profiles.py ${RND}_$P $O/prof_$P > /dev/null
done
DM_N=20000 python tools/i8_check.py > $O/i8_check.txt 2>&1
python tools/len_error.py > $O/len_error.txt 2>&1
python tools/isa_lint.py > $O/isa_lint.json 2>&1
( cd /tmp && export TMPDIR=/tmp; for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/fc_$C -- $R/tools/ubench/fetch_calib > /tmp/fc_$C.log 2>&1; done; tail -3 /tmp/fc_FETCH_SIZE.log; python3 $R/tools/fetch_calib_report.py ) > $O/fetch_calib.txt 2>&1
python tools/i8_tail.py > $O/i8_tail.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $O/pytest_gpu.txt
( python tools/e2e_rate.py packed 200 | sed -n 2,2p; python tools/e2e_rate.py raw 200 | sed -n 2,2p; python tools/e2e_rate.py feat 60 | sed -n 2,2p;
  echo "-- the per-read Python path (DEEPMOD_ROWS_IN_C=0) on the same box:";
  DEEPMOD_ROWS_IN_C=0 python tools/e2e_rate.py packed 200 | sed -n 2,2p; DEEPMOD_ROWS_IN_C=0 python tools/e2e_rate.py raw 200 | sed -n 2,2p ) > $O/e2e_rate.txt 2>&1
# the raw-container command on a 6x repeated input (6e8 base-positions): steady-state rate by number of feeders, rocprofv3 of the command and of the
# signal stage alone (per-kernel GB/s, HBM traffic, the timeline of one request) -> gpurun_out/r06/raw/ + raw_profile.txt
( cd /tmp && export TMPDIR=/tmp; python $R/tools/raw_profile.py 20000 6 2,3,4,6 --rocprof ) > $O/raw_profile.txt 2>&1
[ -f tools/_abl/lib_q_lopack_mix.so ] && python tools/lopack_check.py > $O/ablate_lopack.txt 2>&1
python tools/e2e_detect_packed.py 2,2 > $O/e2e_packed_feeders.txt 2>&1
python tools/e2e_detect_packed.py 2 120 > $O/e2e_packed_120x.txt 2>&1
for P in f16x3 f16i8 f32; do bash tools/power_trace.sh $O/power_$P.txt python tools/bench_loop.py $P 6 > /dev/null; done
tail -c 400 $O/bench_f16x3.json; echo; cat $O/e2e_rate.txt
