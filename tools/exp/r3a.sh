#!/bin/bash
# round 3, experiment A: the persistent wave-specialised launch of the recorded input-gradient cells (k_dgrad_diag_ws)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "input_gradient" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bench() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "== $tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json)"; python - $O/bench_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k, v in d.get("kernels", {}).items():
        if "diag" in k or "head" in k:
            print("    ", k, v.get("mean_us"), v.get("frac_of_hbm_peak"))
except Exception as e:
    print("    parse failed", e)
PY
}
bench lds EVF_DGRAD_DIAG=lds
bench ws EVF_DGRAD_DIAG=ws
bench ws_nodpp EVF_LIB=$PWD/event_flow_amd/libevflow_wd_nodpp.so
bench ws_pf1 EVF_LIB=$PWD/event_flow_amd/libevflow_wd_pf1.so
bench ws_pf3 EVF_LIB=$PWD/event_flow_amd/libevflow_wd_pf3.so
EVF_LIB=$PWD/event_flow_amd/libevflow_wdstamps.so timeout 300 python tools/probes/wd_stamps.py 4 2 > $O/stamps_4_2.txt 2>&1; echo "stamps rc=$?"
EVF_LIB=$PWD/event_flow_amd/libevflow_wdstamps.so timeout 300 python tools/probes/wd_stamps.py 1 0 > $O/stamps_1_0.txt 2>&1
EVF_LIB=$PWD/event_flow_amd/libevflow_wdstamps_nodpp.so timeout 300 python tools/probes/wd_stamps.py 4 2 > $O/stamps_nodpp_4_2.txt 2>&1
head -30 $O/stamps_4_2.txt
