#!/bin/bash
# soak: the default bench line and the test suite repeated
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-others > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.json | head -1) $(grep -o '"value": [0-9.]*' $O/bench_$i.json | head -1)"
done
for i in 1 2; do
  timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_$i.log 2>&1; echo "pytest $i rc=$? $(tail -1 $O/pytest_$i.log)"
done
