cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_events.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_network.py -x -q 2>&1 | tail -3
for v in new old; do
E=""; if [ $v = old ]; then E="EVF_CM_MERGE=0 EVF_FUSED_TAIL=0 EVF_FUSED_ADAM=0"; fi
env $E timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4j/$v -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4j/bench_$v.json 2> gpurun_out/r4j/bench_$v.err; echo "bench $v rc=$?"
python - <<PY
import json, csv, glob
d=json.loads(open("gpurun_out/r4j/bench_$v.json").read().strip().split("\n")[-1])
print("$v", d["value"], d["ms_per_step"], d.get("other_configs",{}).get("c2"))
f=glob.glob("gpurun_out/r4j/$v/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
ad=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_clip_adam")]
step=rows[ad[-2]+1:ad[-1]+1]
print("  nodes per step:", len(step), " period us:", (int(rows[ad[-1]]["End_Timestamp"])-int(rows[ad[-2]]["End_Timestamp"]))/1e3)
PY
done
