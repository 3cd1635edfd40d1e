#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3u; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_backward" > $O/pytest_k.log 2>&1; echo "kernel test rc=$?"; tail -5 $O/pytest_k.log
for m in teams teams4 fused; do
  EVF_BWD_DIAG=$m timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$m.json 2> $O/bench_$m.err; echo "$m rc=$?"; grep -o '"ms_per_step": [0-9.]*' $O/bench_$m.json; grep -o '"loss": [0-9.]*' $O/bench_$m.json | head -1
done
EVF_LIB=$PWD/event_flow_amd/libevflow_fbstamps.so timeout 300 python tools/probes/fbw_stamps.py > $O/stamps.txt 2> $O/stamps.err; echo "stamps rc=$?"; grep -v '^{' $O/stamps.txt | head -12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/prof.log 2>&1; echo "prof rc=$?"
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); head -5 $f | cut -c1-130
