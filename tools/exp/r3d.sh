#!/bin/bash
# round 3, experiment D: K-split two-waves-per-SIMD input gradient (k_dgrad_diag_ks) against the team kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -x -q -m gpu -k "gradient or diagonal or hipgraph or exact_split or g7 or G7" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
bench() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe --no-others > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "== $tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json) loss $(grep -o '"loss": [0-9.]*' $O/bench_$tag.json)"; python - $O/bench_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("    ", {k: round(v["mean_us"], 1) for k, v in d.get("kernels", {}).items() if "diag" in k or "head" in k})
except Exception as e:
    print("    parse failed", e)
PY
}
bench team EVF_DGRAD_DMA=team
bench ks EVF_DGRAD_DMA=ks
bench ks2 EVF_DGRAD_DMA=ks
EVF_LIB=$PWD/event_flow_amd/libevflow_wmstamps.so timeout 300 python tools/probes/wm_stamps.py 4 2 > $O/ks_stamps_4_2.txt 2>&1; echo "stamps rc=$?"
EVF_LIB=$PWD/event_flow_amd/libevflow_wmstamps.so timeout 300 python tools/probes/wm_stamps.py 1 0 > $O/ks_stamps_1_0.txt 2>&1
grep -v amdgpu.ids $O/ks_stamps_4_2.txt | head -24
