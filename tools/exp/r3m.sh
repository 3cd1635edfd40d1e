#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py tests/test_gpu_bench_parity.py -x -q -m gpu -k "not config4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bench() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-iwe --no-others ${ARGS} > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "== $tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json) loss $(grep -o '"loss": [0-9.]*' $O/bench_$tag.json)"; python - $O/bench_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("    ", {k: round(v["mean_us"], 1) for k, v in d.get("kernels", {}).items() if v.get("total_ms_per_step", 0) > 0.2})
except Exception as e:
    print("    parse failed", e)
PY
}
ARGS="--steps 30 --warmup 5"
bench c3 A=1
bench c3b A=1
ARGS="--steps 10 --warmup 3 --config c5"
bench c5 A=1
