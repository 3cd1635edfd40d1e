#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in hip prev hip prev; do
  EVF_LIB=$PWD/event_flow_amd/libevflow_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json | head -1)"
done
EVF_LIB=$PWD/event_flow_amd/libevflow_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/prof.log 2>&1
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); grep "k_fwd_diag_p" $f | cut -c1-120
