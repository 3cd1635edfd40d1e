#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l; rm -rf $O; mkdir -p $O
for f in 0.7 0.5 0.45 0.7 0.5; do
  EVF_CONV_TILE_FILL=$f timeout 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > $O/c4_f$f.json 2> $O/c4_f$f.err; echo "c4 fill=$f rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/c4_f$f.json)"
done
