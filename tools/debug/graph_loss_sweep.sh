run() { echo "== $*"; env "$@" python tools/bench_firenet.py --model LIFFireNet --graph --steps 5 --trace-loss $EXTRA 2>&1 | grep -i "loss" | grep -v "^{" | awk '{printf "%s ", $NF} END {print ""}'; }
EXTRA="--H 128 --W 128 --B 8" run A=1
EXTRA="--H 260 --W 346 --B 4" run A=1
EXTRA="--H 256 --W 320 --B 4" run A=1
EXTRA="" run EVF_FUSED_TAIL=0
EXTRA="" run EVF_PARAM_ROWS=0
EXTRA="" run EVF_HEAD_WIN=0
EXTRA="" run EVF_DEFER_FWD=0 EVF_DEFER_BWD=0
EXTRA="" run EVF_TOP_FUSED=0
EXTRA="" run EVF_PRED_FUSED=0
