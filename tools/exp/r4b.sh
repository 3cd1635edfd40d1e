#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
for s in 1 2 1 2 4; do
  timeout 300 python bench.py --steps 40 --warmup 5 --streams $s --no-cpu-baseline --no-iwe --no-others > $O/bench_s$s.json 2> $O/bench_s$s.err; echo "streams=$s rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_s$s.json) $(grep -o '"value": [0-9.]*' $O/bench_s$s.json | head -1)"
done
