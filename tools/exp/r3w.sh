#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3w; rm -rf $O; mkdir -p $O
for v in hip fpd60 fpd110 hip fpd60 fpd110; do
  EVF_LIB=$PWD/event_flow_amd/libevflow_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json)"
done
EVF_LIB=$PWD/event_flow_amd/libevflow_fpd110s.so timeout 300 python tools/probes/fp_stamps.py 4 1 > $O/stamps.txt 2>&1; echo "rc=$?"; head -6 $O/stamps.txt
