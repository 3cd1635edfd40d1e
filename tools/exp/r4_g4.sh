cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4d
python - <<PY
from event_flow_amd import build
print(build.build_variant("ft_d", {"evf_fwd_teams.hip": ["-DFT_STAMPS=6", "-DFT_PROBE_NOSTORE"]}))
print(build.build_variant("ft_e", {"evf_fwd_teams.hip": ["-DFT_STAMPS=6", "-DFT_PROBE_NOVPREV"]}))
print(build.build_variant("ft_f", {"evf_fwd_teams.hip": ["-DFT_STAMPS=6", "-DFT_PROBE_NOVPREV", "-DFT_PROBE_NOSTORE"]}))
PY
for v in d e f; do
EVF_LIB=$PWD/event_flow_amd/libevflow_ft_$v.so timeout 600 python tools/probes/ft_stamps.py > gpurun_out/r4d/stamps_$v.log 2>&1; echo "$v rc=$?"
grep -v "^{" gpurun_out/r4d/stamps_$v.log | grep -A1 "block 0 wave  [048]" | cut -c1-300
done
