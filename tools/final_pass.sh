#!/bin/bash
# One consistent evidence pass on the GPU box: full GPU test suite, profile round, parity report, default bench line.
R=${1:-r06}
mkdir -p gpurun_out/final_$R
(timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/final_$R/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_$R/gputests.log)
tail -4 gpurun_out/final_$R/gputests.log
bash tools/profile_round.sh $R > gpurun_out/final_$R/profile_round.log 2>&1
tail -3 gpurun_out/final_$R/profile_round.log
python tools/prof_summary.py $R > gpurun_out/final_$R/prof_summary.log 2>&1   # (writes profiles/ on the box: not merged back; re-run locally)
bash tools/parity_report.sh gpurun_out/final_$R/parity_report.txt > /dev/null 2>&1
tail -2 gpurun_out/final_$R/parity_report.txt
cp gpurun_out/final_$R/parity_report.txt profiles/${R}_parity_report.txt   # (so that the bench line below sees the report of THESE sources)
timeout 700 python bench.py > gpurun_out/final_$R/bench_default.json 2> gpurun_out/final_$R/bench_default.err; echo "bench rc=$?"
