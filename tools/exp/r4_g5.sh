cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4e
python - <<PY
from event_flow_amd import build
build.build_variant("ft_p3", {"evf_fwd_teams.hip": ["-DFT_PF=3"]})
build.build_variant("ft_p3c", {"evf_fwd_teams.hip": ["-DFT_PF=3", "-DFT_PROBE_NOE"]})
build.build_variant("ft_c", {"evf_fwd_teams.hip": ["-DFT_PROBE_NOE"]})
PY
for v in base p3 c p3c base p3; do
L=$PWD/event_flow_amd/libevflow_ft_$v.so; if [ $v = base ]; then L=""; fi
EVF_LIB=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > gpurun_out/r4e/bench_$v.json 2> gpurun_out/r4e/bench_$v.err; echo "bench $v rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4e/bench_$v.json").read().strip().split("\n")[-1])
    print("$v", d["ms_per_step"], d["kernels"]["k_fwd_diag"]["mean_us"])
except Exception as e:
    print("$v failed", e)
PY
done
