cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -x -q -s > gpurun_out/r4i/parity.log 2>&1; echo "parity rc=$?"; grep -v "^$" gpurun_out/r4i/parity.log | grep "census\|passed\|failed\|Error\|assert\|cone\|flow rel\|gradient rel\|loss " | cut -c1-330 | tail -40
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4i/bench_default.json 2> gpurun_out/r4i/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4i/bench_default.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"]); print(d.get("other_configs",{}).get("c2"))
PY
