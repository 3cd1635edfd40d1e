"""Kernels of ONE hipGraph-replayed train step, from a rocprofv3 kernel trace of bench.py.

    python tools/step_nodes.py r02 > profiles/r02_step_nodes.txt

Reads the newest gpurun_out/prof_<round>/graph/*/*kernel_trace.csv (tools/profile_round.sh), takes the dispatches
between the last two k_clip_adam launches (= one replayed step) and prints name, launches, summed duration.
"""
import collections
import csv
import glob
import os
import re
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{R}", "graph", "*", "*kernel_trace.csv")), key=os.path.getmtime)
if not files:
    sys.exit("no kernel trace under gpurun_out/prof_%s/graph" % R)
# (the default bench run appends the c4 / c5 lines from child processes, each with a trace of its own: the headline step is in the
#  trace that holds the LIF diagonal kernel -- the PLIF child (c5) has a head-window kernel as well)
pick = [f for f in files if "k_bwd_diag_ws<" in open(f).read()]
rows = list(csv.DictReader(open(pick[-1] if pick else files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_clip_adam")]
if len(adam) < 2:
    sys.exit("fewer than two k_clip_adam dispatches in the trace")
step = rows[adam[-2] + 1 : adam[-1] + 1]


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


agg = collections.OrderedDict()
for r in step:
    n = short(r["Kernel_Name"])
    c, t = agg.get(n, (0, 0.0))
    agg[n] = (c + 1, t + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
busy = sum(t for _, t in agg.values())
native = sum(c for n, (c, _) in agg.items() if "at::native" in n or "elementwise" in n)
print("# kernels of ONE hipGraph-replayed train step (the last one of `rocprofv3 --kernel-trace -- python bench.py --steps 10 "
      "--warmup 3 --no-cpu-baseline`,")
print("# between two k_clip_adam dispatches): name, launches, summed duration (us).  %d kernels, busy %.3f ms "
      "(under the profiler)." % (len(step), busy / 1e3))
print("# at::native / elementwise kernels: %d" % native)
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %5d %9.1f" % (n[:60], c, t))
