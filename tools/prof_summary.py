"""Condense the rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<round>/) into the small files
committed under profiles/.   usage: python tools/prof_summary.py r01"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)
    return f[-1] if f else None  # newest (earlier collections of the round may still be lying around)


def short(name):
    name = re.sub(r"\(.*", "", name)  # drop the argument list
    return name.replace("void ", "").strip()


for run, out in (("graph", "bench_hipgraph"), ("eager", "bench_eager"), ("evfn", "evflownet"), ("plif", "plif_firenet"), ("iwe", "iwe_b2048"),
                 ("xlif", "xlif_step"), ("alif", "alif_step")):
    f = one(f"{run}/*/*kernel_stats.csv")
    if run == "graph":  # (the default bench run appends the c4 / c5 lines from child processes, each with files of its own: the
        #  headline step is in the process that ran the head layer's window kernel)
        cand = [g for g in sorted(glob.glob(os.path.join(SRC, f"{run}/*/*kernel_stats.csv")), key=os.path.getmtime)
                if "k_bwd_diag_ws<" in open(g).read()]  # (the LIF diagonal kernel: the c5 child has head-window kernels too)
        f = cand[-1] if cand else f
    if f:
        shutil.copy(f, os.path.join(DST, f"{R}_{out}_kernel_stats.csv"))
    log = os.path.join(SRC, f"{run}.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                open(os.path.join(DST, f"{R}_{out}.json"), "w").write(line)
            elif "ms per step" in line and run in ("xlif", "alif"):
                open(os.path.join(DST, f"{R}_{out}.txt"), "w").write(line)
            elif line.startswith("B="):
                open(os.path.join(DST, f"{R}_{out}.txt"), "w").write(line)


def counters(run):
    f = one(f"{run}/*/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write, mfma, mfma32 = counters("fetch"), counters("write"), counters("mfma"), counters("mfma32")
rows = []
names = sorted(set(fetch) | set(mfma) | set(mfma32), key=lambda n: -sum(mfma.get(n, {}).get("GRBM_GUI_ACTIVE", [0])))
mean = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
for n in names:
    if n.startswith("at::") or n.startswith("__amd") or "elementwise" in n:
        continue
    src = mfma if n in mfma else mfma32
    gui = mean(src[n].get("GRBM_GUI_ACTIVE", []))  # summed over the 8 XCDs
    busy = mean(src[n].get("SQ_VALU_MFMA_BUSY_CYCLES", []))  # summed over all SIMDs (1024)
    f_kb = mean(fetch.get(n, {}).get("FETCH_SIZE", []))
    w_kb = mean(write.get(n, {}).get("WRITE_SIZE", []))
    rows.append({
        "kernel": n, "dispatches": len(src[n].get("GRBM_GUI_ACTIVE", [])), "precision_run": "bf16x3" if n in mfma else "fp32",
        "gui_cycles_per_xcd": round(gui / 8, 0), "mfma_busy_cycles_per_simd": round(busy / 1024, 0) if busy == busy else "",
        "mfma_util_pct": round(100 * (busy / 1024) / (gui / 8), 1) if busy == busy and gui > 0 else "",
        "fetch_size_KB_raw": round(f_kb, 1), "fetch_MB_x2_corrected": round(2 * f_kb / 1024, 2), "write_size_KB": round(w_kb, 1),
    })
# LDS / VALU side of the kernels (separate PMC passes): per launch, per kernel
lds, valu = counters("lds"), counters("valu")
if lds or valu:
    with open(os.path.join(DST, f"{R}_bench_pmc_lds_valu.csv"), "w", newline="") as fh:
        cols = ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
                "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"]
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + [c + "_per_launch" for c in cols] + ["lds_conflict_share_of_lds_active", "valu_per_mfma"])
        for n in sorted(set(lds) | set(valu)):
            if n.startswith("at::") or n.startswith("__amd") or "elementwise" in n:
                continue
            vals = [mean((lds.get(n, {}) if c in lds.get(n, {}) else valu.get(n, {})).get(c, [])) for c in cols]
            nd = max(len(lds.get(n, {}).get("SQ_INSTS_LDS", [])), len(valu.get(n, {}).get("SQ_INSTS_VALU", [])))
            conf = vals[1] / vals[2] if vals[2] == vals[2] and vals[2] else float("nan")
            vpm = vals[5] / vals[6] if vals[6] == vals[6] and vals[6] else float("nan")
            w.writerow([n, nd] + [round(v, 0) if v == v else "" for v in vals] + [round(conf, 4) if conf == conf else "", round(vpm, 2) if vpm == vpm else ""])
    print(open(os.path.join(DST, f"{R}_bench_pmc_lds_valu.csv")).read())
# kernels that only exist on the fp32 path
for n in sorted(set(mfma32) - set(mfma)):
    pass
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash: bench.py refuses these numbers on any other kernel sources)

if not rows:
    print("(no c3 PMC passes in", SRC, "-- c3 summary files left as they are)")
else:
  with open(os.path.join(DST, f"{R}_bench_pmc_summary.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
# per-launch HBM traffic of the bench's dominant kernels for bench.py's roofline.traffic
traffic = {r["kernel"]: {"fetch_MB": r["fetch_MB_x2_corrected"], "write_MB": round(r["write_size_KB"] / 1024, 2),
                         "mfma_busy_pct": r["mfma_util_pct"] if r["mfma_util_pct"] != "" else None} for r in rows
           if r["fetch_size_KB_raw"] == r["fetch_size_KB_raw"]}
if rows:
  json.dump({"round": R, "src_hash": bench.source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --no-graph` (tools/profile_round.sh); "
           "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated",
           "per_launch": traffic}, open(os.path.join(DST, f"{R}_bench_pmc_traffic.json"), "w"), indent=1)
if rows:
    print(open(os.path.join(DST, f"{R}_bench_pmc_summary.csv")).read())


# ---- config 4: the same three PMC passes over `bench.py --config c4 --no-graph` -> per kernel of the LIF-EV-FlowNet step
c4f, c4w, c4m = counters("c4fetch"), counters("c4write"), counters("c4mfma")
if c4f and c4w:
    def nsteps(acc, key):  # steps the pass ran = dispatches of the optimizer kernel (one per step)
        return max(len(acc.get("k_clip_adam", {}).get(key, [])), 1)
    sf, sw, sm = nsteps(c4f, "FETCH_SIZE"), nsteps(c4w, "WRITE_SIZE"), nsteps(c4m, "GRBM_GUI_ACTIVE")
    rows4 = []
    for n in sorted(set(c4f) | set(c4w), key=lambda n: -sum(c4m.get(n, {}).get("GRBM_GUI_ACTIVE", [0]))):
        f_kb, w_kb = mean(c4f.get(n, {}).get("FETCH_SIZE", [])), mean(c4w.get(n, {}).get("WRITE_SIZE", []))
        gui, busy = mean(c4m.get(n, {}).get("GRBM_GUI_ACTIVE", [])), mean(c4m.get(n, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        rows4.append({"kernel": n, "launches_per_step": round(len(c4f.get(n, {}).get("FETCH_SIZE", [])) / sf, 2),
                      "fetch_MB_x2_corrected": round(2 * f_kb / 1024, 2) if f_kb == f_kb else "",
                      "write_MB": round(w_kb / 1024, 2) if w_kb == w_kb else "",
                      "gui_cycles_per_xcd": round(gui / 8, 0) if gui == gui else "",
                      "mfma_util_pct": round(100 * (busy / 1024) / (gui / 8), 1) if busy == busy and gui == gui and gui > 0 else ""})
    with open(os.path.join(DST, f"{R}_c4_pmc_summary.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows4[0].keys()))
        w.writeheader()
        w.writerows(rows4)
    json.dump({"round": R, "src_hash": bench.source_hash(), "steps_in_pass": [sf, sw, sm],
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes over `bench.py --config c4 --no-graph` "
                         "(tools/profile_round.sh); per LAUNCH means, launches per step = dispatches / dispatches of k_clip_adam; FETCH_SIZE "
                         "doubled per MI355X_MICROARCH.md",
               "per_launch": {r["kernel"]: {"launches_per_step": r["launches_per_step"], "fetch_MB": r["fetch_MB_x2_corrected"],
                                            "write_MB": r["write_MB"], "mfma_busy_pct": r["mfma_util_pct"] if r["mfma_util_pct"] != "" else None}
                              for r in rows4 if r["fetch_MB_x2_corrected"] != "" and r["write_MB"] != ""}},
              open(os.path.join(DST, f"{R}_c4_pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(DST, f"{R}_c4_pmc_summary.csv")).read())


# ---- config 5 (PLIF-FireNet): per launch, in the format of the c3 traffic file (bench.py --config c5 reads it)
c5f, c5w, c5m = counters("c5fetch"), counters("c5write"), counters("c5mfma")
if c5f and c5w:
    rows5 = []
    for n in sorted(set(c5f) | set(c5m), key=lambda n: -sum(c5m.get(n, {}).get("GRBM_GUI_ACTIVE", [0]))):
        if n.startswith("at::") or n.startswith("__amd") or "elementwise" in n:
            continue
        f_kb, w_kb = mean(c5f.get(n, {}).get("FETCH_SIZE", [])), mean(c5w.get(n, {}).get("WRITE_SIZE", []))
        gui, busy = mean(c5m.get(n, {}).get("GRBM_GUI_ACTIVE", [])), mean(c5m.get(n, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", []))
        rows5.append({"kernel": n, "dispatches": len(c5f.get(n, {}).get("FETCH_SIZE", [])),
                      "fetch_MB_x2_corrected": round(2 * f_kb / 1024, 2) if f_kb == f_kb else "",
                      "write_MB": round(w_kb / 1024, 2) if w_kb == w_kb else "",
                      "gui_cycles_per_xcd": round(gui / 8, 0) if gui == gui else "",
                      "mfma_util_pct": round(100 * (busy / 1024) / (gui / 8), 1) if busy == busy and gui == gui and gui > 0 else ""})
    with open(os.path.join(DST, f"{R}_c5_pmc_summary.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows5[0].keys()))
        w.writeheader()
        w.writerows(rows5)
    json.dump({"round": R, "src_hash": bench.source_hash(),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes over `bench.py --config c5 --no-graph` "
                         "(tools/profile_round.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md",
               "per_launch": {r["kernel"]: {"fetch_MB": r["fetch_MB_x2_corrected"], "write_MB": r["write_MB"],
                                            "mfma_busy_pct": r["mfma_util_pct"] if r["mfma_util_pct"] != "" else None}
                              for r in rows5 if r["fetch_MB_x2_corrected"] != "" and r["write_MB"] != ""}},
              open(os.path.join(DST, f"{R}_c5_pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(DST, f"{R}_c5_pmc_summary.csv")).read())
