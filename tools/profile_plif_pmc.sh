#!/bin/bash
# PMC passes over the PLIF-FireNet step (config 5), eager launches (counters are per dispatch):  bash tools/profile_plif_pmc.sh r05
set -u
R=${1:-r05}
O=gpurun_out/prof_${R}_plif_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others"
run() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B
run write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B
run mfma  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B
run valu  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/valu -- $B
python - <<PY
import csv, glob, os, re
from collections import defaultdict
O = "$O"
def counters(run):
    f = sorted(glob.glob(os.path.join(O, run, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = defaultdict(lambda: defaultdict(list))
    if not f:
        return acc
    for r in csv.DictReader(open(f[-1])):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
fe, wr, mf, va = counters("fetch"), counters("write"), counters("mfma"), counters("valu")
mean = lambda v: sum(v) / len(v) if v else float("nan")
names = sorted(set(fe) | set(mf), key=lambda n: -sum(mf.get(n, {}).get("GRBM_GUI_ACTIVE", [0])))
with open(os.path.join("gpurun_out", "${R}_plif_pmc_summary.csv"), "w") as f:
    f.write("kernel,launches,fetch_MB_x2corrected,write_MB,mfma_busy_pct,valu_insts_per_mfma,valu_active_pct_of_wave_cycles,wait_any_pct_of_wave_cycles\n")
    for n in names[:14]:
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-B requests at 64 B: x2 on fetch (MI355X_MICROARCH.md, as tools/prof_summary.py)
        fm = 2 * mean(fe.get(n, {}).get("FETCH_SIZE", [])) / 1024 if n in fe else float("nan")
        wm = mean(wr.get(n, {}).get("WRITE_SIZE", [])) / 1024 if n in wr else float("nan")
        m = mf.get(n, {})
        # busy cycles summed over the 1024 SIMDs against the launch's cycles per XCD (GRBM_GUI_ACTIVE summed over 8 XCDs)
        busy = 100.0 * (sum(m.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / 1024) / max(sum(m.get("GRBM_GUI_ACTIVE", [0])) / 8, 1.0) if m else float("nan")
        v = va.get(n, {})
        ipm = sum(v.get("SQ_INSTS_VALU", [0])) / max(sum(v.get("SQ_INSTS_MFMA", [0])), 1.0) if v else float("nan")
        act = 100.0 * sum(v.get("SQ_ACTIVE_INST_VALU", [0])) / max(sum(v.get("SQ_WAVE_CYCLES", [0])), 1.0) if v else float("nan")
        wt = 100.0 * sum(v.get("SQ_WAIT_INST_ANY", [0])) / max(sum(v.get("SQ_WAVE_CYCLES", [0])), 1.0) if v else float("nan")
        f.write(f"{n},{len(m.get('GRBM_GUI_ACTIVE', []))},{fm:.2f},{wm:.2f},{busy:.1f},{ipm:.2f},{act:.1f},{wt:.1f}\n")
print(open(os.path.join("gpurun_out", "${R}_plif_pmc_summary.csv")).read())
PY
