#!/bin/bash
# the A/B switches still give a green network / training suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
for cfg in "EVF_HEAD_WIN=0" "EVF_HEAD_WIN=mem" "EVF_BWD_DIAG=fused" "EVF_BWD_DIAG=teams4" "EVF_DEFER_FWD=0" "EVF_DEFER_BWD=0" "EVF_DGRAD_SPLIT=0"; do
  env $cfg timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_training.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/pytest_${cfg//=/_}.log 2>&1; echo "$cfg rc=$? $(tail -1 $O/pytest_${cfg//=/_}.log)"
done
