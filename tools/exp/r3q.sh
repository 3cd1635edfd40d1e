#!/bin/bash
# head layer of a window in one launch each way: tests, A/B of the step, kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py tests/test_gpu_training.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for hw in 1 0; do
  EVF_HEAD_WIN=$hw timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_hw$hw.json 2> $O/bench_hw$hw.err; echo "hw=$hw rc=$?"
  python - $O/bench_hw$hw.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   ms_per_step", round(d["ms_per_step"], 4), "value", round(d["value"], 1), "loss", d.get("loss"))
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/prof.log 2>&1; echo "prof rc=$?"
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-150
