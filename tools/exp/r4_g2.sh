cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
python - <<PY
from event_flow_amd import build
print(build.build_variant("ftstamps", {"evf_fwd_teams.hip": ["-DFT_STAMPS=6"]}))
PY
EVF_LIB=$PWD/event_flow_amd/libevflow_ftstamps.so timeout 600 python tools/probes/ft_stamps.py > gpurun_out/r4b/stamps.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r4b/stamps.log | tail -50
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "recorded_forward" 2>&1 | tail -3
