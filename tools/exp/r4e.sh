#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "head_layer" > $O/pytest_k.log 2>&1; echo "kernel test rc=$?"; tail -5 $O/pytest_k.log
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_training.py tests/test_gpu_bench_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/prof.log 2>&1; echo "prof rc=$?"
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); grep -i "head" $f | cut -c1-110; grep -o '"ms_per_step": [0-9.]*' $O/prof.log
