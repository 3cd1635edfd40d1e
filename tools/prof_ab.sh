#!/bin/bash
# A/B of two environment settings inside the hipGraph-replayed train step: rocprofv3 kernel averages + ms/step.
#   bash tools/prof_ab.sh "EVF_BWD=fused" "EVF_BWD=ws" [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="$1"; B="$2"; shift 2
for cfg in "$A" "$B"; do
  tag=$(echo "$cfg" | tr -c 'A-Za-z0-9' '_')
  rm -rf gpurun_out/prof_ab_$tag
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$tag -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe "$@" > gpurun_out/prof_ab_$tag.log 2>&1
  f=$(ls gpurun_out/prof_ab_$tag/*/*kernel_stats.csv | head -1)
  echo "== $cfg  $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_ab_$tag.log)  $(grep -o '"value": [0-9.]*' gpurun_out/prof_ab_$tag.log | head -1)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:10]:
    print("  ", r["Name"][:64].ljust(64), r["Calls"].rjust(5), "%6.1f us %5.1f %%" % (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
PY
done
