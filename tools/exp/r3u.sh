#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3u; rm -rf $O; mkdir -p $O
EVF_LIB=$PWD/event_flow_amd/libevflow_fbstamps.so timeout 300 python tools/probes/fbw_stamps.py > $O/stamps.txt 2> $O/stamps.err; echo "stamps rc=$?"; grep -v '^{' $O/stamps.txt | head -40
