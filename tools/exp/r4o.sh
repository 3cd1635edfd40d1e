#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
for v in hip wmnoprio wmlowprio hip wmnoprio wmlowprio; do
  EVF_LIB=$PWD/event_flow_amd/libevflow_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json | head -1)"
done
