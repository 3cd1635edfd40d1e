"""Debug: 4-scale EventWarping gradient HIP vs oracle at the config-4 shape (256x256, B=8, 50k events)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd import synthetic
from event_flow_amd.loss import flow as hloss
from oracle import encodings as oenc
from oracle import loss as oloss

DEV = torch.device("cuda:0")
torch.set_num_threads(32)
for (B, H, W, n, mag) in ((8, 256, 256, 50000, 0.2), (4, 128, 128, 50000, 0.2), (8, 256, 256, 15000, 0.2)):
    cfg = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
    rng = np.random.default_rng(5)
    lossf = hloss.EventWarping(cfg, DEV)
    win = oloss.Window((H, W))
    ev = synthetic.event_list_batch(B, n, H, W, synthetic.seed_for(4, 0, 0))
    d = oenc.collate([oenc.encode_window(ev[b, :, 2], ev[b, :, 1], ev[b, :, 0], ev[b, :, 3], 2, (H, W)) for b in range(B)])
    gfs, ofs = [], []
    for s in (8, 4, 2, 1):
        f = rng.uniform(-mag, mag, size=(B, 2, H // s, W // s)).astype(np.float32)
        f = np.repeat(np.repeat(f, s, 2), s, 3).copy()
        gfs.append(torch.from_numpy(f).to(DEV).requires_grad_(True))
        ofs.append(torch.from_numpy(f).requires_grad_(True))
    lossf.event_flow_association(gfs, torch.from_numpy(d["event_list"]).to(DEV), torch.from_numpy(d["event_list_pol_mask"]).to(DEV),
                                 torch.from_numpy(d["event_mask"]).to(DEV))
    win.add(ofs, torch.from_numpy(d["event_list"]), torch.from_numpy(d["event_list_pol_mask"]), torch.from_numpy(d["event_mask"]))
    val = lossf()
    val.backward()
    ref = oloss.event_warping_loss(win, max(H, W), 0.001)
    ref.backward()
    print((B, H, W, n), f"loss {float(val.detach()):.8f} vs {float(ref.detach()):.8f}")
    for i, (g, o) in enumerate(zip(gfs, ofs)):
        g, r = g.grad.cpu().numpy(), o.grad.numpy()
        print(f"   scale {i}: grad rel-L2 {np.linalg.norm(g - r) / np.linalg.norm(r):.3e}; max|diff| {np.abs(g - r).max():.3e} of max|g| {np.abs(r).max():.3e}")
