#!/bin/bash
# ms/step of the replayed train step under HIP-runtime settings (one process each; same box)
cd "$GRAFT_REPO_ROOT"
for cfg in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" \
           "DEBUG_HIP_GRAPH_BATCH_SIZE=16" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" \
           "GPU_FLUSH_ON_EXECUTION=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "X=1"; do
  r=$(env $cfg timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "$cfg  $r"
done
