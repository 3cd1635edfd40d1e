"""One rocprofv3 CSV per POPULATION: the kernels of the hipGraph-REPLAYED train steps only.

    python tools/replayed_step_stats.py r06            # reads gpurun_out/prof_r06/{graph,evfn,plif}/*/*kernel_trace.csv
                                                       # writes profiles/r06_{c3,c4,c5}_replayed_step_kernel_stats.csv

The `*_kernel_stats.csv` files rocprofv3 writes for a bench.py run average every dispatch of the process: eager warm-up steps,
the instrumented capture's replays, side measurements (IWE sweep, c2 line) and the timed replays alike.  This tool keeps only the
dispatches of the TIMED region's replayed steps: the trace is cut at the optimizer kernel (k_clip_adam*: the last kernel of a
step), a step is the run of dispatches between two such cuts, and the replayed steps are the steps whose kernel sequence is the
most frequent one (every replay of a graph launches the same nodes; eager steps carry torch / fill kernels and other counts).
Columns as rocprofv3's stats CSV plus calls per step: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev,
CallsPerStep, Steps.  `roofline.frac` of the bench line = algorithmic work / AverageNs of its kernel in THIS file."""
import collections
import csv
import glob
import math
import os
import re
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def replayed(rows):
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cuts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_clip_adam")]
    steps = [rows[a + 1 : b + 1] for a, b in zip(cuts[:-1], cuts[1:])]
    if not steps:
        return [], None
    sig = [tuple(short(r["Kernel_Name"]) for r in st) for st in steps]
    # the replayed steps: the most frequent kernel sequence(s) -- two graphs alternate, so the two most frequent sequences with the
    # same length count when both occur often
    cnt = collections.Counter(sig)
    best, nbest = cnt.most_common(1)[0]
    keep = {best}
    for s, n in cnt.most_common(3)[1:]:
        if len(s) == len(best) and n >= max(2, nbest // 2):
            keep.add(s)
    return [st for st, sg in zip(steps, sig) if sg in keep], len(best)


def write(tag, trace):
    rows = list(csv.DictReader(open(trace)))
    steps, nk = replayed(rows)
    if not steps:
        print(f"{tag}: no replayed steps found in {trace}")
        return
    dur = collections.defaultdict(list)
    for st in steps:
        for r in st:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in dur.values())
    out = os.path.join(DST, f"{R}_{tag}_replayed_step_kernel_stats.csv")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "CallsPerStep", "Steps"])
        for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            m = sum(v) / len(v)
            sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
            w.writerow([name, len(v), sum(v), round(m, 3), round(100.0 * sum(v) / tot, 4), min(v), max(v), round(sd, 3),
                        round(len(v) / len(steps), 3), len(steps)])
    span = [(int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])) / 1e6 for st in steps]
    print(f"{tag}: {len(steps)} replayed steps of {nk} kernels, busy {tot / len(steps) / 1e6:.3f} ms per step, first-start to last-end "
          f"{sum(span) / len(span):.3f} ms per step (under the profiler) -> {os.path.relpath(out, ROOT)}")


for tag, run, must in (("c3", "graph", "k_bwd_diag_ws<"), ("c4", "evfn", None), ("c5", "plif", None)):
    files = sorted(glob.glob(os.path.join(SRC, run, "*", "*kernel_trace.csv")), key=os.path.getmtime)
    if must:  # (the default bench run's child processes write traces of their own: the headline step holds the LIF diagonal kernel)
        files = [f for f in files if must in open(f).read()] or files
    if files:
        write(tag, files[-1])
    else:
        print(f"{tag}: no kernel trace under {os.path.join(SRC, run)}")
