"""Relative error of every parameter gradient of a fused XLIF FireNet against the oracle (plain autograd, three passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_xlif as T
from oracle import snn as osnn
B, H, W = 2, 16, 20
torch.manual_seed(5)
model = T.XLIFFireNet(T.cfg()).to(T.DEV)
print("fused", model._fused())
params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
for k, _ in model.named_parameters():
    params[k].requires_grad_(True)
xs = [(torch.rand(B, 2, H, W) < 0.5).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3)]
states = [None] * 7
tot_ref, tot = 0, 0
for x in xs:
    f_ref, states = osnn.firenet_forward("XLIFFireNet", params, x, states, hard_reset=True)
    out = model(x.to(T.DEV), x.to(T.DEV))
    tot_ref = tot_ref + (f_ref * torch.arange(f_ref.numel()).view(f_ref.shape).remainder(7)).sum()
    fl = out["flow"][0]
    tot = tot + (fl * torch.arange(fl.numel(), device=T.DEV).view(fl.shape).remainder(7)).sum()
print("spike rates", [float(s[1].mean()) for s in states])
tot.backward(); tot_ref.backward()
for k, p in model.named_parameters():
    ref = params[k].grad
    ref = ref.numpy() if ref is not None else np.zeros(tuple(p.shape), np.float32)
    got = T.N(p.grad) if p.grad is not None else np.zeros_like(ref)
    print(f"{k:22s} rel {np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12):10.3e}  |ref| {np.linalg.norm(ref):10.3e} |got| {np.linalg.norm(got):10.3e}")
