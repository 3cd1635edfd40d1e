# A/B of the recorded forward's schedules on one box (EVF_FWD_LM): replayed train step, alternating settings
#   bash tools/debug/fwd_lm_ab.sh [c3|c5] [rounds]
C=${1:-c3}; N=${2:-3}
if [ $C = c3 ]; then ARGS="--model LIFFireNet --H 128 --W 128 --B 8 --steps 60"; else ARGS="--model PLIFFireNet --H 260 --W 346 --B 4 --steps 20"; fi
for i in $(seq $N); do
  for lm in 0 top 1; do
    r=$(EVF_FWD_LM=$lm timeout 300 python tools/bench_firenet.py $ARGS --graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms loss %.6f' % (d['ms_per_step'], d['loss']))")
    echo "$C EVF_FWD_LM=$lm  $r"
  done
done
