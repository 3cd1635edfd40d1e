cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4f
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "recorded_forward" 2>&1 | tail -3
for m in persistent teams; do
EVF_FWD_DIAG=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4f/bench_$m.json 2> gpurun_out/r4f/bench_$m.err; echo "bench $m rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4f/bench_$m.json").read().strip().split("\n")[-1])
print("$m", d["value"], d["ms_per_step"], d["kernels"]["k_fwd_diag"]["mean_us"])
PY
done
