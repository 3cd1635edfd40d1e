#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3r; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py tests/test_gpu_training.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/prof_$n.log 2>&1; echo "$n rc=$?"
  f=$(ls $O/prof_$n/*/*kernel_stats.csv | head -1); grep -i "head" $f | cut -c1-110
  grep -o '"ms_per_step": [0-9.]*' $O/prof_$n.log
}
run carry A=1

for hw in 1 0; do
  EVF_HEAD_WIN=$hw timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_hw$hw.json 2> $O/bench_hw$hw.err; echo "hw=$hw rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/bench_hw$hw.json
done
