cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4c
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "recorded_forward" 2>&1 | tail -3
python - <<PY
from event_flow_amd import build
print(build.build_variant("ft_a", {"evf_fwd_teams.hip": ["-DFT_STAMPS=6"]}))
PY
for v in a; do
EVF_LIB=$PWD/event_flow_amd/libevflow_ft_$v.so timeout 600 python tools/probes/ft_stamps.py > gpurun_out/r4c/stamps_$v.log 2>&1; echo "$v rc=$?"
grep -v "^{" gpurun_out/r4c/stamps_$v.log | grep -A1 "block 0 wave  [048]" | cut -c1-330
done
for m in persistent teams; do
L=""; M=$m
EVF_LIB=$L EVF_FWD_DIAG=$M timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4c/bench_$m.json 2> gpurun_out/r4c/bench_$m.err; echo "bench $m rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4c/bench_$m.json").read().strip().split("\n")[-1])
print("$m", d["value"], d["ms_per_step"], d["kernels"]["k_fwd_diag"]["mean_us"])
PY
done
