cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4a
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "recorded_forward" > gpurun_out/r4a/t1.log 2>&1; echo "t1 rc=$?"; tail -15 gpurun_out/r4a/t1.log
timeout 900 python -m pytest tests/test_gpu_network.py -x -q -k "diagonal or hipgraph or golden" > gpurun_out/r4a/t2.log 2>&1; echo "t2 rc=$?"; tail -8 gpurun_out/r4a/t2.log
for m in persistent teams; do
EVF_FWD_DIAG=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4a/bench_$m.json 2> gpurun_out/r4a/bench_$m.err; echo "bench $m rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4a/bench_$m.json").read().strip().split("\n")[-1])
print("$m", d["value"], d["ms_per_step"])
for k in d.get("kernels",[]):
    print("  ", k.get("name"), k.get("mean_us"), k.get("launches_per_step"))
PY
done
