#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_general.py -x -q -m gpu > $O/pytest_k.log 2>&1; echo "kernel/general tests rc=$?"; tail -3 $O/pytest_k.log
for m in teams fused teams fused; do
  EVF_BWD_ONE=$m timeout 300 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/c5_$m.json 2> $O/c5_$m.err; echo "c5 one=$m rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c5_$m.json)"
done
EVF_DEFER_BWD=0 EVF_BWD_ONE=teams timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/c3_nodefer_teams.json 2>/dev/null; echo "c3 nodefer teams $(grep -o '"ms_per_step": [0-9.]*' $O/c3_nodefer_teams.json)"
EVF_DEFER_BWD=0 EVF_BWD_ONE=fused timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/c3_nodefer_fused.json 2>/dev/null; echo "c3 nodefer fused $(grep -o '"ms_per_step": [0-9.]*' $O/c3_nodefer_fused.json)"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_training.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
