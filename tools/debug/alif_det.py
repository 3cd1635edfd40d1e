"""Is the ALIF step deterministic pass by pass?  Two EAGER runs of three steps under a deterministic loss, parameters compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_xlif as T
from test_gpu_network import _LinearWindowLoss
from event_flow_amd.train import train_window, FlatAdam
from event_flow_amd.dataloader.encodings import encode_event_list
from event_flow_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "ALIFFireNet"
cls, neuron, _ = T.NETS[name]
B, n, H, W, P = 2, 600, 32, 64, 3
DEV = T.DEV
pool = [[torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 7100 + 100 * w + k)).to(DEV) for k in range(P)] for w in range(2)]
gw = torch.Generator(device="cpu").manual_seed(9)
wts = [(torch.randn(B, 2, H, W, generator=gw) * 0.02).to(DEV) for _ in range(P)]
res = []
for rep in range(3):
    torch.manual_seed(3)
    m = cls(T.cfg(neuron)).to(DEV); m.train()
    opt = FlatAdam(m, lr=2e-4, clip=100.0, device_step=True); opt.zero_grad()
    l = _LinearWindowLoss(wts)
    for i in range(3):
        passes = [encode_event_list(ev, 2, (H, W), want=("cnt", "mask", "pol")) for ev in pool[i % 2]]
        for d in passes: d["event_voxel"] = None
        train_window(m, l, opt, passes)
    torch.cuda.synchronize()
    res.append(opt.flat_param.clone())
    # garbage into freed memory between the runs: a read of an uninitialised buffer would show
    junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(8)]; del junk
print(name, "eager run 0 vs 1:", float((res[0] - res[1]).abs().max()), " 0 vs 2:", float((res[0] - res[2]).abs().max()), " nan:", bool(torch.isnan(res[2]).any()))
