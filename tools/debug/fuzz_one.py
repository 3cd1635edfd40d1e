"""Per-parameter gradient differences of one fuzz case: python tools/debug/fuzz_one.py LIF|PLIF B H W seed"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools/debug")
import fuzz_firenet as ff  # noqa: E402
from oracle import snn as osnn  # noqa: E402

kind, B, H, W, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
name, cls, neuron = (("LIFFireNet", ff.LIFFireNet, ff.NEURON) if kind == "LIF" else ("PLIFFireNet", ff.PLIFFireNet, ff.PLIF))
torch.manual_seed(seed)
model = cls(ff.cfg(neuron)).to(ff.DEV)
with torch.no_grad():
    for k, p in model.named_parameters():
        if k.endswith("thresh"):
            p.mul_(0.2)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
xs = [(torch.rand(B, 2, H, W) < 0.6).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(2)]


def oracle(dtype):
    params = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k, _ in model.named_parameters():
        params[k].requires_grad_(True)
    states = [None] * 7
    tot = 0
    for x in xs:
        f, states = osnn.firenet_forward(name, params, x.to(dtype), states)
        wgt = (torch.arange(f.numel()).view(f.shape).remainder(5).float() - 2.0).to(dtype)
        tot = tot + (f * wgt).sum()
    tot.backward()
    return {k: (params[k].grad.double().numpy() if params[k].grad is not None else None) for k, _ in model.named_parameters()}


g32 = oracle(torch.float32)
g64 = g32  # (a float64 oracle breaks at the second layer: the reference's spikes are always float32, spiking_util.py:21)
model.train()
tot = 0
for x in xs:
    f = model(x.to(ff.DEV), x.to(ff.DEV))["flow"][0]
    wgt = torch.arange(f.numel()).view(f.shape).remainder(5).float() - 2.0
    tot = tot + (f * wgt.to(ff.DEV)).sum()
tot.backward()
print("param                      |ref64|     hip-vs-32   hip-vs-64   ora32-vs-64")
for k, p in model.named_parameters():
    if g64[k] is None:
        continue
    h = p.grad.cpu().double().numpy()
    n = max(np.linalg.norm(g64[k]), 1e-30)
    print(f"{k:24s} {n:10.3e}  {np.linalg.norm(h - g32[k]) / n:10.2e}  {np.linalg.norm(h - g64[k]) / n:10.2e}  {np.linalg.norm(g32[k] - g64[k]) / n:10.2e}")

# ---- which pixels explain the difference of d pred.conv2d.weight?  (gpre_o * z_c per pixel and pass, oracle side)
k = "pred.conv2d.weight"
diff = (dict(model.named_parameters())[k].grad.cpu().double().numpy() - g32[k]).reshape(2, 32)
print("max |diff| of", k, np.abs(diff).max(), "at", np.unravel_index(np.abs(diff).argmax(), diff.shape))
params = {kk: v.clone() for kk, v in sd.items()}
states = [None] * 7
best = []
for t, x in enumerate(xs):
    f, states = osnn.firenet_forward(name, params, x, states)
    wgt = torch.arange(f.numel()).view(f.shape).remainder(5).float() - 2.0
    gpre = (wgt * (1 - f * f)).detach().double().numpy()  # [B,2,H,W]
    z = states[6][1].detach().double().numpy()             # [B,32,H,W]
    for b in range(B):
        for y in range(H):
            for xx in range(W):
                c = np.outer(gpre[b, :, y, xx], z[b, :, y, xx])
                for sgn in (+1.0, -1.0):
                    r = np.linalg.norm(diff - sgn * c)
                    best.append((r, t, b, y, xx, sgn))
best.sort()
print("|diff| =", np.linalg.norm(diff))
for r in best[:6]:
    print("residual %.4f after removing %+d x (pass %d, b %d, y %d, x %d)" % (r[0], int(r[5]), r[1], r[2], r[3], r[4]))
