#!/bin/bash
# Block-share / grid knobs of the recorded diagonal launches at 8 x 128 x 128, one replayed step each (60 steps), the default
# setting re-measured between groups (boxes drift by ~0.5 %).   bash tools/debug/balance_sweep.sh > gpurun_out/balance_sweep.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
run() {
  r=$(env $1 timeout 200 python tools/bench_firenet.py --model LIFFireNet --H 128 --W 128 --B 8 --graph --steps 60 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d.get('ms_per_step', d.get('ms', -1)))" 2>&1)
  echo "$1  $r"
}
run X=0
for g in "EVF_BWD_W=10,12,11 EVF_BWD_W=10,14,11 EVF_BWD_W=10,15,11 EVF_BWD_W=10,11,11 EVF_BWD_W=1,1,1" \
         "EVF_BWD_UNITS=16 EVF_BWD_UNITS=24 EVF_BWD_UNITS=32 EVF_BWD_UNITS=43 EVF_BWD_UNITS=64" \
         "EVF_BWD_COST=5 EVF_BWD_COST=11 EVF_BWD_COST=16 EVF_BWD_COST=24" \
         "EVF_FT_W=8,12,13 EVF_FT_W=8,13,13 EVF_FT_W=8,15,13 EVF_FT_W=8,16,13 EVF_FT_W=8,18,13 EVF_FT_W=8,11,13"; do
  for s in $g; do run $s; done
  run X=0
done
