#!/bin/bash
# per-kernel averages of one bench configuration under two libraries:  bash tools/debug/ab_kernels.sh c5 event_flow_amd/libevflow_old.so
CFG=${1:-c5}; OLD=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in new old; do
  L=""; [ $v = old ] && L=$OLD
  EVF_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o t -- python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > /dev/null 2>&1
done
python - <<'PY'
import csv, glob
def load(v):
    f = glob.glob(f"/tmp/ab_{v}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"].split("(")[0].replace("void ", ""): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f))}
a, b = load("new"), load("old")
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1])[:16]:
    o = b.get(k, (float("nan"), 0))
    print(f"{k[:60]:60s} new {a[k][0]:8.1f} us x {a[k][1]:5d}   old {o[0]:8.1f} us   {100 * (a[k][0] / o[0] - 1):+5.1f} %")
PY
