import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd.models import hip_ops
DEV = torch.device("cuda:0")
for (B, C, H, W) in ((8, 32, 256, 256), (8, 256, 32, 32)):
    gen = torch.Generator().manual_seed(1)
    x = (torch.rand(B, C, H, W, generator=gen) < 0.3).float()
    w = torch.randn(2, C, 1, 1, generator=gen) * 0.01
    b = torch.randn(2, generator=gen) * 0.01
    gy = torch.randn(B, 2, H, W, generator=gen)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.tanh(torch.nn.functional.conv2d(xr.double(), wr.double(), br.double()))
    yr.backward(gy.double())
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = hip_ops.conv_act(object(), xd, wd, bd, stride=1, activation="tanh")
    y.backward(gy.to(DEV))
    rel = lambda a, r: float(np.linalg.norm(a.cpu().numpy().astype(np.float64) - r.numpy()) / np.linalg.norm(r.numpy()))
    print((B, C, H, W), "y", rel(y.detach(), yr.detach()), "gw", rel(wd.grad, wr.grad), "gb", rel(bd.grad, br.grad), "gx", rel(xd.grad, xr.grad))
