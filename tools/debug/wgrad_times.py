"""HIP-event times of evf_conv2d_wgrad at the LIF-EV-FlowNet shapes (BASELINE configs[3]), two-team kernel on / off."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd import _lib

dev = torch.device("cuda:0")
L = _lib.load()
SHAPES = [(8, 16, 16, 512, 512), (8, 256, 256, 128, 32), (8, 64, 64, 512, 128), (8, 32, 32, 1024, 256), (8, 128, 128, 256, 64),
          (8, 32, 32, 256, 256), (8, 64, 64, 128, 128), (8, 128, 128, 64, 64)]
gen = torch.Generator().manual_seed(0)
for B, H, W, Cin, Cout in SHAPES:
    x = (torch.rand(B, H, W, Cin, generator=gen) < 0.2).float().to(dev)
    gy = (torch.randn(B, H, W, Cout, generator=gen) * 0.1).to(dev)
    ws = torch.empty(max(L.evf_conv2d_wgrad_ws(B, H, W, Cin, Cout, 3, 1), 1), device=dev)
    g_w = torch.zeros(Cout, Cin, 3, 3, device=dev)
    line = f"{B}x{H}x{W} {Cin:4d}->{Cout:4d}:"
    for mode in (0, 2, 0, 2):
        _lib.call("evf_wgrad_teams_select", mode)
        for _ in range(3):
            _lib.call("evf_conv2d_wgrad", _lib.ptr(x), Cin, _lib.ptr(gy), Cout, _lib.ptr(g_w), None, B, H, W, Cin, Cout, 3, 1, Cin, 0, 4, _lib.ptr(ws))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            _lib.call("evf_conv2d_wgrad", _lib.ptr(x), Cin, _lib.ptr(gy), Cout, _lib.ptr(g_w), None, B, H, W, Cin, Cout, 3, 1, Cin, 0, 4, _lib.ptr(ws))
        e1.record()
        torch.cuda.synchronize()
        line += f"  mode{mode} {e0.elapsed_time(e1) / n * 1e3:7.1f} us"
    print(line, flush=True)
_lib.call("evf_wgrad_teams_select", 0)
