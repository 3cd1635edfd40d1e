#!/bin/bash
# round 3, experiment B: LDS-DMA fed persistent input gradient on pre-split planes (k_dgrad_diag_dma)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
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
bench ws EVF_DGRAD_SPLIT=0
bench dma EVF_DGRAD_SPLIT=1
bench dma_nodpp EVF_LIB=$PWD/event_flow_amd/libevflow_wm_nodpp.so
bench dma_nopipe EVF_LIB=$PWD/event_flow_amd/libevflow_wm_nopipe.so
for v in wmstamps wmstamps_nodpp; do
  EVF_LIB=$PWD/event_flow_amd/libevflow_$v.so timeout 300 python tools/probes/wm_stamps.py 4 2 > $O/${v}_4_2.txt 2>&1; echo "$v rc=$?"
done
EVF_LIB=$PWD/event_flow_amd/libevflow_wmstamps.so timeout 300 python tools/probes/wm_stamps.py 1 0 > $O/wmstamps_1_0.txt 2>&1
grep -v amdgpu.ids $O/wmstamps_4_2.txt | head -24
