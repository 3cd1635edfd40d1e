#!/bin/bash
# The measured parity distances of the benched workloads (c3, c4, c5: spike flips per pass, flow / loss / gradient
# distances against the CPU oracle) and of the exact-split kernels against float64 -- `pytest -q` drops these prints.
#   bash tools/parity_report.sh [out]      (on the GPU box; committed as profiles/rNN_parity_report.txt)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=${1:-gpurun_out/parity_report.txt}
mkdir -p "$(dirname "$OUT")"
{
  echo "# parity report: $(date -u +%Y-%m-%dT%H:%M:%SZ)  $(python -c 'import torch; print(torch.cuda.get_device_name(0))' 2>/dev/null)"
  echo "# kernel sources: $(python -c 'import bench; print(bench.source_hash())' 2>/dev/null)"
  python -m pytest -s -q -m gpu tests/test_gpu_bench_parity.py tests/test_gpu_teacher_forced.py \
      "tests/test_gpu_network.py::test_exact_split_input_gradient_is_fp32_equivalent" \
      "tests/test_gpu_network.py::test_exact_split_weight_gradient_is_fp32_equivalent" \
      "tests/test_gpu_network.py::test_bf16x3_forward_is_fp32_equivalent" 2>&1 | grep -v "^$"
} > "$OUT" 2>&1
tail -5 "$OUT"
