cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p $O
run() { name=$1; shift; timeout 600 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run valu rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/valu -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run mfma rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
python - <<PY
import csv, glob, collections
for run in ("valu", "mfma"):
    f = glob.glob("$O/%s/*/*counter_collection.csv" % run)
    if not f: print(run, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in acc:
        if "diag" in n:
            print(run, n, {k: round(sum(v)/len(v)) for k, v in acc[n].items()})
PY
