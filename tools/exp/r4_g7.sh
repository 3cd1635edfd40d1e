cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4g
for m in persistent teams; do
EVF_FWD_DIAG=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4g/$m -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4g/bench_$m.json 2> gpurun_out/r4g/bench_$m.err; echo "bench $m rc=$?"
f=$(find gpurun_out/r4g/$m -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-160
python - <<PY
import json
d=json.loads(open("gpurun_out/r4g/bench_$m.json").read().strip().split("\n")[-1])
print("$m", d["value"], d["ms_per_step"])
PY
done
