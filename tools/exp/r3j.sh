#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3j; mkdir -p $O
timeout 2300 python -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; echo "rc=$?"; tail -8 $O/pytest_full.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iwe > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], {k: round(v["mean_us"], 1) for k, v in d["kernels"].items()})
print(json.dumps(d.get("other_configs"))[:1500])
PY
