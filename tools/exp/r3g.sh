#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g; mkdir -p $O
t() { tag=$1; shift; env "$@" timeout 600 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_network.py -x -q -m gpu -s -k "two_graph_replay or bitwise or diagonal" > $O/$tag.log 2>&1; echo "== $tag rc=$? $(grep -E 'graph vs eager|passed|failed' $O/$tag.log | tr '\n' ' ')"; }
t split_side EVF_HEAD_SIDE=1
bench() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "== $tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json) loss $(grep -o '"loss": [0-9.]*' $O/bench_$tag.json)"; }
bench side EVF_HEAD_SIDE=1
bench noside EVF_HEAD_SIDE=0
bench side2 EVF_HEAD_SIDE=1
bench noside2 EVF_HEAD_SIDE=0
