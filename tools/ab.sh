#!/bin/bash
# A/B (/C/...) of environments in the replayed train step on ONE GPU box, alternating so that box-to-box and drift noise
# cancel.  One parametrised script instead of one file per experiment (rounds 3-4 left 40 of those):
#
#   bash tools/ab.sh [-n reps] [-s steps] [-c c3|c4|c5] [-p kernel-substring] "ENV_A" "ENV_B" ["ENV_C" ...]
#
#   ENV_x   space-separated VAR=value pairs ("" or "X=0" for the default build), e.g.
#             "EVF_FWD_DIAG=persistent"   "EVF_CM_MERGE=0 EVF_FUSED_TAIL=0"   "EVF_LIB=$PWD/event_flow_amd/libevflow_<name>.so"
#           probe / variant builds: python tools/ab_variant.py <name> <file.hip>:-DFLAG[=v] ...  (prints the library path)
#   -p      also run every environment once under `rocprofv3 --kernel-trace --stats` and print the average duration of the
#           kernels whose name contains the substring
# through gpurun:  gpurun --timeout 1500 -- 'bash tools/ab.sh -n 3 "X=0" "EVF_FWD_DIAG=persistent"'
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=3; S=60; C=c3; P=""
while getopts "n:s:c:p:" o; do case $o in n) R=$OPTARG;; s) S=$OPTARG;; c) C=$OPTARG;; p) P=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { echo "usage: bash tools/ab.sh [-n reps] [-s steps] [-c cfg] [-p kernel] ENV_A ENV_B ..."; exit 2; }
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(d['ms_per_step'],4), 'ms', round(d['value'],1), 'windows/s')
except Exception as e: print('$1 FAILED', e)"; }
for i in $(seq 1 $R); do
  k=0
  for E in "$@"; do
    k=$((k + 1))
    env $E timeout 900 python bench.py --config $C --steps $S --warmup 5 --no-cpu-baseline --no-iwe --no-others 2>/dev/null | line "[$k: $E]"
  done
done
if [ -n "$P" ]; then
  k=0
  for E in "$@"; do
    k=$((k + 1)); O=gpurun_out/ab_prof_$k; rm -rf $O
    env $E timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > /dev/null 2>&1
    f=$(find $O -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep "$P" "$f" | awk -F'","' -v t="[$k: $E]" '{n=split($0,f,"\","); split(f[2],g,","); printf "%s %s: calls %s avg %.1f us\n", t, substr(f[1],2,40), g[1], g[3]/1000}'
  done
fi
