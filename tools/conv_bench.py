"""Per-layer timing of the general convolution kernels at the LIF-EV-FlowNet (BASELINE configs[3]) layer shapes:
fp32 MFMA vs bf16x3 general vs bf16x3 spatially tiled, forward and input gradient, spike-valued and real inputs;
prints the max difference against the fp32 kernel.  Usage: python tools/conv_bench.py [B] [H]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from event_flow_amd import _lib  # noqa: E402
from event_flow_amd.models import hip_ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
LAYERS = [  # name, Cin, Cout, resolution divisor, k, stride
    ("enc1.rec", 64, 64, 2, 3, 1), ("enc2.conv", 64, 128, 2, 3, 2), ("enc2.rec", 128, 128, 4, 3, 1),
    ("enc3.conv", 128, 256, 4, 3, 2), ("enc3.rec", 256, 256, 8, 3, 1), ("enc4.conv", 256, 512, 8, 3, 2),
    ("enc4.rec", 512, 512, 16, 3, 1), ("dec1", 1024, 256, 8, 3, 1), ("dec2", 516, 128, 4, 3, 1), ("dec3", 260, 64, 2, 3, 1),
    ("dec4", 132, 32, 1, 3, 1),
]


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(mode, x, w, k, s, grad):
    hip_ops.CONV_B3 = mode != "f32"
    _lib.load().evf_conv_tile_select({"b3": 0, "b3tile": -1, "f32": 0}[mode])
    Bn, Hh, Ww, Cin = x.shape
    Cout = w.shape[0]
    cache = hip_ops._PackCache()
    Ho, Wo = hip_ops._out_dim(Hh, k, s), hip_ops._out_dim(Ww, k, s)
    if not grad:
        wp = cache.get(w, 0, 0, Cin)
        y = torch.empty(Bn, Ho, Wo, Cout, device=dev)
        f = lambda: hip_ops.conv_fwd(x, wp, None, y, Cin, Cout, k, s)  # noqa: E731
    else:
        wp = cache.get(w, 1, 0, w.shape[1])
        gy = x  # caller passes the [B,Ho,Wo,Cout] gradient as x
        y = torch.empty(Bn, Hh * s, Ww * s, w.shape[1], device=dev)
        f = lambda: hip_ops.conv_dgrad(gy, wp, y, w.shape[1], Cout, k, s)  # noqa: E731
    us = timed(f)
    return us, y


print(f"{'layer':10s} {'pass':6s} {'input':7s} {'GF':>6s} | {'f32 us':>8s} {'b3 us':>8s} {'tile us':>8s} | b3 TF  tile TF | maxdiff b3 / tile (rel to max|y|)")
for name, cin, cout, div, k, s in LAYERS:
    hh = H // div
    gen = torch.Generator().manual_seed(1)
    w = (torch.randn(cout, cin, k, k, generator=gen) * 0.05).to(dev)
    for grad in (False, True):
        for kind in ("spikes", "real"):
            if grad and kind == "spikes":
                continue
            if not grad:
                x = torch.rand(B, hh, hh, cin, generator=gen)
                x = (x < 0.1).float() if kind == "spikes" else x
            else:
                x = torch.randn(B, hh // s, hh // s, cout, generator=gen)
            x = x.to(dev)
            gf = 2.0 * k * k * cin * cout * B * (hh // s) ** 2 / 1e9
            res = {m: run(m, x, w, k, s, grad) for m in ("f32", "b3", "b3tile")}
            ref = res["f32"][1]
            sc = float(ref.abs().max())
            d = [float((res[m][1] - ref).abs().max()) / sc for m in ("b3", "b3tile")]
            print(f"{name:10s} {'dgrad' if grad else 'fwd':6s} {kind:7s} {gf:6.1f} | {res['f32'][0]:8.1f} {res['b3'][0]:8.1f} "
                  f"{res['b3tile'][0]:8.1f} | {gf / res['b3'][0] * 1e3:6.1f} {gf / res['b3tile'][0] * 1e3:6.1f} | {d[0]:.2e} {d[1]:.2e}")
_lib.load().evf_conv_tile_select(-1)
