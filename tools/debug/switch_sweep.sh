#!/bin/bash
# Every A/B switch of the library / engine once: the bench step must run and report a finite loss (correctness of the
# combinations is in the tests; this only makes sure no switch is left broken).   bash tools/debug/switch_sweep.sh [c3|c5]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
C=${1:-c3}
for cfg in "X=0" "EVF_FWD_DIAG=persistent" "EVF_FWD_DIAG=tile" "EVF_BWD_DIAG=fused" "EVF_BWD_DIAG=teams4" "EVF_BWD_W=0,0,0" "EVF_BWD_ONE=fused" \
           "EVF_DGRAD_RING=1" "EVF_DGRAD_SPLIT=0" "EVF_DGRAD_DIAG=lds EVF_DGRAD_SPLIT=0" "EVF_HEAD_WIN=0" "EVF_DEFER_FWD=0" "EVF_DEFER_BWD=0" \
           "EVF_DEFER_FWD=0 EVF_DEFER_BWD=0" "EVF_FUSED_TAIL=0" "EVF_CM_MERGE=0" "EVF_PRED_FUSED=0" "EVF_TOP_FUSED=0" "EVF_PAIR_DGRAD=0" \
           "EVF_PARAM_ROWS=0" "EVF_PLIF_BOX=kernel" "EVF_FT_W=4,7,4" "EVF_PLIF_TRACE_FUSED=0" "EVF_HEAD_FWD_WAVES=4" "EVF_HEAD_WIN=mem" \
           "EVF_PLIF_TRACE_FUSED=0 EVF_HEAD_WIN=0" "EVF_PLIF_LAYER_MAJOR=0" "EVF_PLIF_LM_DGRAD=dma" "EVF_LIF_BWD_TOP=0" "EVF_FWD_LM=0" "EVF_FWD_LM=1" "EVF_FWD_LM=top" "EVF_FWD_LM=1 EVF_FWD_WIN_ILV=0" "EVF_DEBUG_POISON_LDS=1"; do
  r=$(env $cfg timeout 300 python bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-iwe --no-others 2>/tmp/sw.err | tail -1 | python -c "
import sys,json,math
try:
    d=json.loads(sys.stdin.read()); l=d['config']['loss']; print('%.3f ms  loss %.6f %s' % (d['ms_per_step'], l, 'ok' if math.isfinite(l) else 'NOT FINITE'))
except Exception as e: print('FAILED', e)")
  echo "$cfg  $r"
  case "$r" in *FAILED*|*NOT*) tail -3 /tmp/sw.err;; esac
done
