#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_training.py -x -q -m gpu > $O/pytest_training.log 2>&1; echo "rc=$?"
grep -v "frame #" $O/pytest_training.log | grep -v "^E  *$" | tail -80
