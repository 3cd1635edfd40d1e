cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6; do timeout 900 python -m pytest tests/test_gpu_training.py -x -q -s -k rccl_code_path 2>&1 | grep "loss:\|passed\|failed\|Error" | head -5; done
