#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_network.py -x -q -m gpu -k "diagonal or defer or recorded or forward" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in hip fpdma hip fpdma; do
  EVF_LIB=$PWD/event_flow_amd/libevflow_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json)"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/prof.log 2>&1; echo "prof rc=$?"
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); head -5 $f | cut -c1-130
