"""Debug: EventWarping gradient HIP vs oracle as the flow magnitude grows (events leaving the image)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd import synthetic
from event_flow_amd.loss import flow as hloss
from oracle import encodings as oenc
from oracle import loss as oloss

DEV = torch.device("cuda:0")
B, H, W, n = 4, 128, 128, 15000
cfg = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
for mag in (0.05, 0.2, 0.5, 1.0):
    for smooth in (False, True):
        rng = np.random.default_rng(5)
        lossf = hloss.EventWarping(cfg, DEV)
        win = oloss.Window((H, W))
        ev = synthetic.event_list_batch(B, n, H, W, 7000)
        d = oenc.collate([oenc.encode_window(ev[b, :, 2], ev[b, :, 1], ev[b, :, 0], ev[b, :, 3], 2, (H, W)) for b in range(B)])
        f = rng.uniform(-mag, mag, size=(B, 2, H, W)).astype(np.float32)
        if smooth:  # piecewise-constant flow (like an up-sampled coarse prediction)
            f = np.repeat(np.repeat(f[:, :, ::8, ::8], 8, 2), 8, 3).copy()
        gf = torch.from_numpy(f).to(DEV).requires_grad_(True)
        of = torch.from_numpy(f).requires_grad_(True)
        lossf.event_flow_association([gf], torch.from_numpy(d["event_list"]).to(DEV), torch.from_numpy(d["event_list_pol_mask"]).to(DEV),
                                     torch.from_numpy(d["event_mask"]).to(DEV))
        win.add([of], torch.from_numpy(d["event_list"]), torch.from_numpy(d["event_list_pol_mask"]), torch.from_numpy(d["event_mask"]))
        val = lossf()
        val.backward()
        ref = oloss.event_warping_loss(win, max(H, W), 0.001)
        ref.backward()
        g, r = gf.grad.cpu().numpy(), of.grad.numpy()
        print(f"mag {mag:4.2f} smooth {smooth!s:5s}: loss {float(val.detach()):.8f} vs {float(ref.detach()):.8f}; grad rel-L2 {np.linalg.norm(g - r) / np.linalg.norm(r):.3e}; "
              f"max|diff| {np.abs(g - r).max():.3e} of max|g| {np.abs(r).max():.3e}")
