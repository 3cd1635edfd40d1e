#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3v; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_backward" > $O/pytest_k.log 2>&1; echo "kernel test rc=$?"; tail -3 $O/pytest_k.log
for rep in 1 2; do
for cfg in "teams 8" "fused 11" "teams4 11"; do
  set -- $cfg
  EVF_BWD_DIAG=$1 EVF_BWD_COST=$2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "$1 cost=$2 rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$1_$2.json)"
done
done
EVF_LIB=$PWD/event_flow_amd/libevflow_fbstamps.so timeout 300 python tools/probes/fbw_stamps.py > $O/stamps.txt 2> $O/stamps.err; echo "stamps rc=$?"; grep -v '^{' $O/stamps.txt | head -6
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_training.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
