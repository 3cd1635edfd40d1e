#!/usr/bin/env python
"""Micro-benchmark of the network entry points at the BASELINE config-2/3 shape
(B=8, 128x128, C=32): mean kernel time from HIP events over back-to-back
launches (the queue stays full, so host launch cost is hidden) and the
achieved algorithmic TFLOP/s / GB/s.   python tools/kbench.py [--reps 50]"""

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from event_flow_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=128)
ap.add_argument("--W", type=int, default=128)
args = ap.parse_args()
B, H, W, C = args.B, args.H, args.W, 32
dev = "cuda:0"
npix = B * H * W
torch.manual_seed(0)
f = lambda *s: torch.randn(*s, device=dev)
bits = lambda: torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev)
w = f(32, 32, 3, 3) * 0.1
wp, wpt = torch.empty(9216, device=dev), torch.empty(9216, device=dev)
_lib.call("evf_pack_conv_weight", w.data_ptr(), 32, 32, 0, wp.data_ptr())
_lib.call("evf_pack_conv_weight", w.data_ptr(), 32, 32, 1, wpt.data_ptr())
wb3 = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
_lib.call("evf_pack_conv_weight_b3", w.data_ptr(), 32, 32, wb3.data_ptr())
leak, thresh = f(32) * 0.1 - 4, f(32) * 0.1 + 0.8
x, z = bits(), bits()
v, vo, g1, g2, g3, g4 = (f(B, H, W, C) for _ in range(6))
zo = bits()
nslab = _lib.load().evf_conv_wgrad_slabs(B, H, W)
slab = torch.empty(nslab, 9216, device=dev)
slab2 = torch.empty(nslab, 9216, device=dev)
gsp = torch.zeros(3, B, H, W, 32, dtype=torch.bfloat16, device=dev)
wb3t = torch.empty(54 * 1024, dtype=torch.uint8, device=dev)
_lib.call("evf_pack_conv_weight_b3t", w.data_ptr(), 32, 32, wb3t.data_ptr())
xT = torch.randint(-2**31, 2**31 - 1, (B, H, 32, (W + 31) // 32), dtype=torch.int32, device=dev)
gl, gt = torch.zeros(32, device=dev), torch.zeros(32, device=dev)
xin = f(B, 2, H, W)
wh = f(32, 2, 3, 3)
dwh = torch.zeros(32, 2, 3, 3, device=dev)
hslab = torch.zeros(_lib.load().evf_head_lif_bwd_wgrad_slabs(B, H, W), 32 * 18, device=dev)
P = lambda t: t.data_ptr()
FL = 2 * 9 * 32 * 32 * npix

cases = [
    ("conv_lif_fwd ff", FL, 3 * npix * 128, lambda: _lib.call("evf_conv_lif_fwd", P(x), P(wp), None, P(leak), P(thresh), P(v), P(z), B, H, W, 1, P(vo), P(zo), None)),
    ("conv_lif_fwd rec", 2 * FL, 3 * npix * 128, lambda: _lib.call("evf_conv_lif_fwd", P(x), P(wp), P(wp), P(leak), P(thresh), P(v), P(z), B, H, W, 1, P(vo), P(zo), None)),
    ("conv_lif_fwd_b3 ff", FL, 3 * npix * 128, lambda: _lib.call("evf_conv_lif_fwd_b3", P(x), P(wb3), None, P(leak), P(thresh), P(v), P(z), B, H, W, 1, P(vo), P(zo), None)),
    ("conv_lif_fwd_b3 rec", 2 * FL, 3 * npix * 128, lambda: _lib.call("evf_conv_lif_fwd_b3", P(x), P(wb3), P(wb3), P(leak), P(thresh), P(v), P(z), B, H, W, 1, P(vo), P(zo), None)),
    ("head_lif_fwd", 2 * 18 * 32 * npix, 2 * npix * 128, lambda: _lib.call("evf_head_lif_fwd", P(xin), P(wh), P(leak), P(thresh), P(v), P(z), B, 2, H, W, 1, P(vo), P(zo), None)),
    ("conv_dgrad one", FL, 2 * npix * 128, lambda: _lib.call("evf_conv_dgrad", P(g1), P(wpt), P(g2), 0, None, None, 0, B, H, W)),
    ("conv_dgrad two", 2 * FL, 3 * npix * 128, lambda: _lib.call("evf_conv_dgrad", P(g1), P(wpt), P(g2), 0, P(wpt), P(g3), 0, B, H, W)),
    ("conv_dgrad_b3", FL, 2 * npix * 128, lambda: _lib.call("evf_conv_dgrad_b3", P(gsp), P(wb3t), P(g2), 0, B, H, W, None, None)),
    ("conv_wgrad_bits", FL, npix * 128, lambda: _lib.call("evf_conv_wgrad_bits", P(x), P(g1), B, H, W, P(slab), 1)),
    ("head_lif_bwd_wgrad", 2 * 18 * 32 * npix, 5 * npix * 128, lambda: _lib.call("evf_head_lif_bwd_wgrad", P(g1), P(g2), P(vo), P(v), P(z), P(xin), P(leak), P(thresh), B, 2, H, W, 1, 0, 10.0, None, P(g4), P(gl), P(gt), P(hslab), 0)),
    ("lif_bwd_wgrad ff", FL, 6 * npix * 128, lambda: _lib.call("evf_lif_bwd_wgrad", P(g1), P(g2), P(vo), P(v), P(z), P(xT), None, P(leak), P(thresh), B, H, W, 1, 0, 10.0, P(g3), P(gsp), P(g4), P(gl), P(gt), P(slab), None, 1)),
    ("lif_bwd_wgrad rec", 2 * FL, 6 * npix * 128, lambda: _lib.call("evf_lif_bwd_wgrad", P(g1), P(g2), P(vo), P(v), P(z), P(xT), P(xT), P(leak), P(thresh), B, H, W, 1, 0, 10.0, P(g3), P(gsp), P(g4), P(gl), P(gt), P(slab), P(slab2), 1)),
    ("lif_bwd", 0, 6 * npix * 128, lambda: _lib.call("evf_lif_bwd", P(g1), P(g2), P(vo), P(v), P(z), P(leak), P(thresh), B, H, W, 1, 0, 10.0, P(g3), P(g4), P(gl), P(gt))),
    ("head_wgrad", 2 * 18 * 32 * npix, npix * 128, lambda: _lib.call("evf_head_wgrad", P(xin), P(g1), B, 2, H, W, P(dwh))),
]
print(f"shape B={B} {H}x{W}  ({torch.cuda.get_device_name(0)})")
for name, flop, byts, fn in cases:
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.reps * 1e3
    print(f"{name:18s} {us:8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s ({flop / us / 1e6 / 157.3 * 100:5.1f}% fp32-MFMA)   {byts / us / 1e3:7.1f} GB/s algorithmic")
