"""Per-layer timing of the general-path conv kernels at the EV-FlowNet (BASELINE config 4) layer shapes.
usage: python tools/kbench_conv.py [--B 8] [--res 256] [--iters 20]"""

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_flow_amd import _lib  # noqa: E402
from event_flow_amd.models import hip_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
a = ap.parse_args()
R, B = a.res, a.B
LAYERS = [  # name, Cin, Cout, H(in), k, stride
    ("enc0.conv", 2, 64, R, 3, 2), ("enc0.rec", 64, 64, R // 2, 3, 1),
    ("enc1.conv", 64, 128, R // 2, 3, 2), ("enc1.rec", 128, 128, R // 4, 3, 1),
    ("enc2.conv", 128, 256, R // 4, 3, 2), ("enc2.rec", 256, 256, R // 8, 3, 1),
    ("enc3.conv", 256, 512, R // 8, 3, 2), ("res/enc3.rec", 512, 512, R // 16, 3, 1),
    ("dec0", 1024, 256, R // 8, 3, 1), ("dec1", 514, 128, R // 4, 3, 1), ("dec2", 258, 64, R // 2, 3, 1),
    ("dec3", 130, 32, R, 3, 1), ("pred3", 32, 2, R, 1, 1), ("firenet", 32, 32, 128, 3, 1),
]
dev = "cuda:0"


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3  # us


class Owner:
    pass


print(f"{'layer':14s} {'Cin':>5s} {'Cout':>5s} {'HxW':>9s} {'GFLOP':>8s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
tot = [0.0, 0.0, 0.0]
for name, Cin, Cout, H, k, s in LAYERS:
    if a.only and a.only not in name:
        continue
    Ho = (H + 2 * (k // 2) - k) // s + 1
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    gx = torch.empty(B, H, H, Cin, device=dev)
    gw = torch.empty_like(w)
    o = Owner()
    wp = hip_ops._wcache(o, "w").get(w, 0)
    wt = hip_ops._wcache(o, "wT").get(w, 1)
    gf = 2.0 * B * Ho * Ho * Cin * Cout * k * k / 1e9
    t_f = timeit(lambda: hip_ops.conv_fwd(x, wp, None, y, Cin, Cout, k, s))
    t_d = timeit(lambda: hip_ops.conv_dgrad(y, wt, gx, Cin, Cout, k, s))
    t_w = timeit(lambda: hip_ops.conv_wgrad(x, y, gw, None, Cin, Cout, k, s))
    tot[0] += t_f
    tot[1] += t_d
    tot[2] += t_w
    print(f"{name:14s} {Cin:5d} {Cout:5d} {H:4d}x{H:<4d} {gf:8.2f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} | {t_d:8.1f} {gf / t_d * 1e3:6.1f} | {t_w:8.1f} {gf / t_w * 1e3:6.1f}")
print("sum us", [round(t, 1) for t in tot])
