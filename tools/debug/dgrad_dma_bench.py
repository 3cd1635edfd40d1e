"""k_dgrad_diag_dma (pre-split planes, a list of products per launch) against k_conv_dgrad_ws (fp32 gradient, one product per launch)
per product at a given shape:  python tools/debug/dgrad_dma_bench.py B H W"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_flow_amd import _lib  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 260, 346)
dev = "cuda:0"
P = lambda t: t.data_ptr()  # noqa: E731
w = torch.randn(32, 32, 3, 3, device=dev) * 0.1
wt = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
_lib.call("evf_pack_conv_weight_b3t", P(w), 32, 32, P(wt))
NP = 8
gs = [torch.randn(B, H, W, 32, device=dev) for _ in range(NP)]
gsp = [torch.randn(3, B, H, W, 32, device=dev).to(torch.bfloat16) for _ in range(NP)]
out = [torch.empty(B, H, W, 32, device=dev) for _ in range(NP)]
gP = torch.randn(B, H, W, device=dev)
xb = torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev)


def single(plif):
    for k in range(NP):
        _lib.call("evf_conv_dgrad_b3_f32", P(gs[k]), P(wt), P(out[k]), 2 if plif else 0, B, H, W, P(gP) if plif else None, P(xb) if plif else None)


def dma():
    assert _lib.raw("evf_bwd_defer_begin") == 0 and _lib.raw("evf_bwd_defer_slot", 1) == 0
    for k in range(NP):
        _lib.call("evf_conv_dgrad_b3", P(gsp[k]), P(wt), P(out[k]), 0, B, H, W, None, None)
    _lib.call("evf_bwd_defer_flush")


for name, fn in (("k_conv_dgrad_ws, fp32 in", lambda: single(False)), ("k_conv_dgrad_ws + PLIF term", lambda: single(True)), ("k_dgrad_diag_dma, 8 products", dma)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-30s %.1f us per product" % (name, e0.elapsed_time(e1) * 1e3 / 10 / NP))
