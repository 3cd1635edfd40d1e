set -u
R=r06; O=gpurun_out/prof_$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
grep -E "^run (xlif|alif)" tools/profile_round.sh > /tmp/lines.sh
source /tmp/lines.sh
tail -1 $O/xlif.log | cut -c1-200; tail -1 $O/alif.log | cut -c1-200
