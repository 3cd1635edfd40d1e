# A/B of two environments in the replayed step on ONE box, alternating:  bash tools/exp/r4_ab.sh "ENV_A" "ENV_B" [reps] [steps]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="$1"; B="$2"; R=${3:-3}; S=${4:-60}
for i in $(seq 1 $R); do
for v in A B; do
E="$A"; if [ $v = B ]; then E="$B"; fi
env $E timeout 600 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-iwe --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v [$E]', round(d['ms_per_step'],4), round(d['value'],1))"
done; done
