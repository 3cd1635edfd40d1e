set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_network.py -x -q -k "layer_major_forward or diagonal_launches_equal or plif_cells_recorded" 2>&1 | tail -15 > gpurun_out/t1.log
cat gpurun_out/t1.log
for lm in 0 1 top; do
  echo "== c5 EVF_FWD_LM=$lm"; EVF_FWD_LM=$lm timeout 300 python tools/bench_firenet.py --model PLIFFireNet --H 260 --W 346 --B 4 --graph --steps 20 2>&1 | tail -2
  echo "== c3 EVF_FWD_LM=$lm"; EVF_FWD_LM=$lm timeout 300 python tools/bench_firenet.py --model LIFFireNet --H 128 --W 128 --B 8 --graph --steps 40 2>&1 | tail -2
done 2>&1 | tee gpurun_out/ab1.log
