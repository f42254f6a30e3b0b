#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/<round>/ cites, into gpurun_out/<round>/.   bash tools/round_evidence.sh r02
RND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$RND
mkdir -p $O
python bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err
python bench.py --precision f32 --no-cpu-baseline > $O/bench_f32.json 2>/dev/null
DM_BENCH_FORCE_DIST=1 MASTER_PORT=29999 python bench.py --steps 8 --no-cpu-baseline --no-extras > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
bash tools/profile_round.sh ${RND}_f16x3 f16x3 > $O/profile.log 2>&1
python tools/summarize_profiles.py ${RND}_f16x3 $O/prof_f16x3 > /dev/null
bash tools/profile_round.sh ${RND}_f32 f32 >> $O/profile.log 2>&1
python tools/summarize_profiles.py ${RND}_f32 $O/prof_f32 > /dev/null
( python tools/e2e_rate.py packed 200 | sed -n 2,2p; python tools/e2e_rate.py raw 60 | sed -n 2,2p; python tools/e2e_rate.py feat 60 | sed -n 2,2p ) > $O/e2e_rate.txt 2>&1
python tools/e2e_detect_raw.py 12000 16 > $O/e2e_detect_raw.txt 2>&1
bash tools/power_trace.sh $O/power_f16x3.txt python tools/bench_loop.py f16x3 6 > /dev/null
bash tools/power_trace.sh $O/power_f32.txt python tools/bench_loop.py f32 6 > /dev/null
tail -c 300 $O/bench_f16x3.json; echo; cat $O/e2e_rate.txt
