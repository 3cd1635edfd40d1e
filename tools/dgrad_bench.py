#!/usr/bin/env python
"""Input-gradient kernels stand-alone: time per launch (HIP events over back-to-back launches, buffers cycled through more
than the Infinity Cache) and a hash of the outputs (A/B runs of two builds / EVF_DGRAD_LDS=1 must print the same hashes).
    python tools/dgrad_bench.py [--B 8 --H 128 --W 128] [--reps 40]"""
import argparse
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from event_flow_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--H", type=int, default=128)
ap.add_argument("--W", type=int, default=128)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
B, H, W = a.B, a.H, a.W
dev = "cuda:0"
torch.manual_seed(0)
P = lambda t: t.data_ptr()
w1, w2 = torch.randn(32, 32, 3, 3, device=dev) * 0.1, torch.randn(32, 32, 3, 3, device=dev) * 0.1
wt1, wt2 = (torch.empty(54 * 1024, dtype=torch.uint8, device=dev) for _ in range(2))
_lib.call("evf_pack_conv_weight_b3t", P(w1), 32, 32, P(wt1))
_lib.call("evf_pack_conv_weight_b3t", P(w2), 32, 32, P(wt2))
NSET = max(2, int(600e6 // (3 * B * H * W * 128)) + 1)
gs = [torch.randn(B, H, W, 32, device=dev) for _ in range(NSET)]
oa = [torch.randn(B, H, W, 32, device=dev) for _ in range(NSET)]
ob = [torch.empty(B, H, W, 32, device=dev) for _ in range(NSET)]
gP = torch.randn(B, H, W, device=dev)
xb = torch.randint(-2**31, 2**31 - 1, (B, H, W), dtype=torch.int32, device=dev)
npix = B * H * W


def h(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]


cases = [
    ("dgrad_f32", 256, lambda k: _lib.call("evf_conv_dgrad_b3_f32", P(gs[k]), P(wt1), P(oa[k]), 0, B, H, W, None, None), lambda: (oa[0],)),
    ("dgrad_f32 acc", 384, lambda k: _lib.call("evf_conv_dgrad_b3_f32", P(gs[k]), P(wt1), P(oa[k]), 1, B, H, W, None, None), lambda: (oa[0],)),
    ("dgrad_f32 plif", 264, lambda k: _lib.call("evf_conv_dgrad_b3_f32", P(gs[k]), P(wt1), P(oa[k]), 0, B, H, W, P(gP), P(xb)), lambda: (oa[0],)),
    ("dgrad_f32_pair", 384, lambda k: _lib.call("evf_conv_dgrad_b3_f32_pair", P(gs[k]), P(wt1), P(oa[k]), 0, P(wt2), P(ob[k]), B, H, W, None, None), lambda: (oa[0], ob[0])),
    ("dgrad_f32_pair acc", 512, lambda k: _lib.call("evf_conv_dgrad_b3_f32_pair", P(gs[k]), P(wt1), P(oa[k]), 1, P(wt2), P(ob[k]), B, H, W, None, None), lambda: (oa[0], ob[0])),
]
print(f"shape B={B} {H}x{W}, {NSET} buffer sets, EVF_DGRAD_LDS={os.environ.get('EVF_DGRAD_LDS', '')}")
for name, bpp, fn, outs in cases:
    for t in oa:
        t.fill_(0.25)
    fn(0)
    torch.cuda.synchronize()
    hs = " ".join(h(t) for t in outs())
    for k in range(3):
        fn(k % NSET)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(a.reps):
        fn(k % NSET)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    print(f"{name:20s} {us:8.2f} us  {npix * bpp / us / 1e3:8.1f} GB/s algorithmic ({npix * bpp / us / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)  out {hs}")
