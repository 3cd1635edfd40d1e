# kernel durations of the PLIF / LIF forward schedules (EVF_FWD_LM) under the tracer: eager launches, kernel-trace only
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lm in ${LMS:-0 1}; do
  rm -rf /tmp/prof_$lm
  EVF_FWD_LM=$lm timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lm -- python $R/tools/bench_firenet.py ${ARGS:---model PLIFFireNet --H 260 --W 346 --B 4} --steps 6 > /tmp/prof_$lm.log 2>&1
  f=$(find /tmp/prof_$lm -name "*kernel_stats.csv" | head -1)
  echo "== EVF_FWD_LM=$lm"; head -12 $f | cut -c1-160
  cp $f $R/gpurun_out/fwdlm_${TAG:-c5}_$lm.csv
done
