#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_general.py -x -q -m gpu > $O/pytest.log 2>&1; echo "general tests rc=$?"; tail -5 $O/pytest.log
for p in 1 0 1 0; do
  EVF_CONV_PAIR=$p timeout 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/c4_p$p.json 2> $O/c4_p$p.err; echo "c4 pair=$p rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c4_p$p.json)"
done
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -k "config4 or c4 or evflownet" > $O/pytest_c4.log 2>&1; echo "c4 parity rc=$?"; tail -3 $O/pytest_c4.log
