#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# kernel-trace/stats passes and --pmc passes are separate runs (gpurun refuses --pmc together with API traces).
set -u
R=${1:-r01}
O=gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; }
run graph  rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph  -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run eager  rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager  -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-iwe --no-others --no-others
run fetch  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run write  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run mfma   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run mfma32 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma32 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others --precision fp32
# LDS side of the matrix-bound kernels: instructions, bank-conflict cycles, issue stalls on the LDS pipe
run lds    rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/lds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run valu   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/valu -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run evfn   rocprofv3 --kernel-trace --stats --output-format csv -d $O/evfn -- python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others
# config 4 (the general path): HBM bytes and matrix-pipe busy per kernel (roofline.traffic of the c4 line)
run c4fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c4fetch -- python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-graph
run c4write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c4write -- python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-graph
run c4mfma  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c4mfma -- python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-graph
# config 5 (PLIF-FireNet): the same three passes (what tools/profile_plif_pmc.sh collected in round 5, + the traffic file of the c5 line)
run c5fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c5fetch -- python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run c5write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c5write -- python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run c5mfma  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c5mfma -- python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-iwe --no-others
run plif   rocprofv3 --kernel-trace --stats --output-format csv -d $O/plif -- python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-iwe --no-others
# the other neuron models of the FireNet family on the recorded window kernels, at the headline shape
run xlif   rocprofv3 --kernel-trace --stats --output-format csv -d $O/xlif -- python tools/debug/xlif_step.py XLIFFireNet
run alif   rocprofv3 --kernel-trace --stats --output-format csv -d $O/alif -- python tools/debug/xlif_step.py ALIFFireNet
run iwe    rocprofv3 --kernel-trace --stats --output-format csv -d $O/iwe -- python tools/iwe_bench.py 2048
ls -R $O | head -60
