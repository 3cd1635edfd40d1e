#!/bin/bash
# Kernels of the hipGraph-replayed train step that are NOT ours (torch / runtime helpers): calls per step and time.
#   bash tools/prof_plumbing.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=20; WARM=3
rm -rf gpurun_out/prof_plumb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_plumb -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-iwe "$@" > gpurun_out/prof_plumb.log 2>&1
f=$(ls gpurun_out/prof_plumb/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/prof_plumb_kernel_stats.csv
echo "$(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_plumb.log)"
python - "$f" $STEPS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = lambda n: n.lstrip("void ").startswith("k_")
o = [r for r in rows if not ours(r["Name"])]
print("ours: %.3f ms/step (all calls / %d)" % (sum(float(r["TotalDurationNs"]) for r in rows if ours(r["Name"])) / steps / 1e6, steps))
print("not ours: %.3f ms/step" % (sum(float(r["TotalDurationNs"]) for r in o) / steps / 1e6))
for r in o:
    print("  ", r["Name"][:110].ljust(110), r["Calls"].rjust(6), "%6.1f us avg  %7.1f us/step-ish" % (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e3))
PY
