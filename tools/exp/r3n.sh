#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in hip fb_nob16 fb_lut0; do
O=gpurun_out/r3n_$v; rm -rf $O; mkdir -p $O
EVF_LIB=$PWD/event_flow_amd/libevflow_$v.so timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/lds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others > $O/lds.log 2>&1; echo "$v rc=$?"
python - $O <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/lds/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    if "bwd_diag" in n:
        m = {k: sum(v) / len(v) for k, v in d.items()}
        print("   ", n, {k: round(v) for k, v in m.items()}, "conflict share", round(m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1), 3))
PY
f=$(ls $O/lds/*/*kernel_trace.csv | head -1)
python - $f <<'PY'
import csv, sys
t = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "k_bwd_diag" in r["Kernel_Name"]]
print("    k_bwd_diag mean us (under pmc)", round(sum(t) / len(t) / 1e3, 1), len(t))
PY
done
