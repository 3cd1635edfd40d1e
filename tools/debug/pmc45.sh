set -u
R=r06
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
grep -E "^run c[45]" tools/profile_round.sh > /tmp/lines.sh
source /tmp/lines.sh
python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others > gpurun_out/s2_c5.json 2> gpurun_out/s2_c5.err; echo c5 rc=$?
python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/s2_c4.json 2> gpurun_out/s2_c4.err; echo c4 rc=$?
du -sh $O
