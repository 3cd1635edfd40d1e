"""Per-layer timing of evf_conv2d_wgrad (3x3, stride 1) at the LIF-EV-FlowNet layer shapes, spike-valued and real inputs.
Run once with EVF_WGRAD=f32 and once without to compare the fp32 kernel with the bf16 one (+ fp32 redo of flagged tiles);
prints a checksum-free comparison against a float64 reference on a sub-sample of the weights.  Usage: python tools/wgrad_bench.py [B] [H]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from event_flow_amd.models import hip_ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
LAYERS = [("enc1.rec", 64, 64, 2), ("enc2.rec", 128, 128, 4), ("enc3.rec", 256, 256, 8), ("enc4.rec", 512, 512, 16),
          ("dec1", 1024, 256, 8), ("dec2", 516, 128, 4), ("dec3", 260, 64, 2), ("dec4", 132, 32, 1)]
print("mode", os.environ.get("EVF_WGRAD", "b3"))
for name, cin, cout, div in LAYERS:
    hh = H // div
    gen = torch.Generator().manual_seed(1)
    for kind in ("spikes", "mixed", "real"):
        x = torch.rand(B, hh, hh, cin, generator=gen)
        if kind != "real":
            xs = (x < 0.1).float()
            if kind == "mixed":
                xs[..., :2] = x[..., :2]
            x = xs
        g = torch.randn(B, hh, hh, cout, generator=gen)
        xd, gd = x.to(dev), g.to(dev)
        gw = torch.empty(cout, cin, 3, 3, device=dev)
        gb = torch.empty(cout, device=dev)
        f = lambda: hip_ops.conv_wgrad(xd, gd, gw, gb, cin, cout, 3, 1)  # noqa: E731
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        # float64 reference of a few weights: gw[co][ci][1][1] = sum x[px][ci] g[px][co]
        ci_s, co_s = [0, 1, cin // 2, cin - 1], [0, cout - 1]
        ref = torch.einsum("bhwi,bhwo->oi", x[..., ci_s].double(), g[..., co_s].double())
        got = gw[co_s][:, ci_s, 1, 1].double().cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        gf = 2.0 * 9 * cin * cout * B * hh * hh / 1e9
        print(f"{name:9s} {kind:7s} {gf:6.1f} GF {us:8.1f} us {gf / us * 1e3:7.1f} TF   centre-tap err {err:.2e}  bias err "
              f"{float((gb.cpu().double() - g.double().sum((0, 1, 2))).abs().max() / g.double().sum((0, 1, 2)).abs().max()):.2e}")
