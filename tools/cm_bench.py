#!/usr/bin/env python
"""evf_cm_loss_fwd: device-scope-atomic splat vs the LDS-striped splat, at the config-3 and config-4 shapes.
python tools/cm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from event_flow_amd import _lib, synthetic
from event_flow_amd.loss import flow as hloss

DEV = "cuda:0"


def run(B, H, W, P, n, S, min_events):
    hloss.CM_LDS_MIN_EVENTS = min_events
    cfg = {"loader": {"resolution": [H, W], "batch_size": B}, "loss": {"flow_regul_weight": 0.001, "clip_grad": 100.0, "overwrite_intermediate": False},
           "model": {"mask_output": True}}
    lossf = hloss.EventWarping(cfg, DEV)
    rng = np.random.default_rng(0)
    for k in range(P):
        ev = torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 7000 + 100 * k)).to(DEV)
        pol = torch.stack([(ev[:, :, 3] > 0).float(), (ev[:, :, 3] < 0).float()], 2).contiguous()
        fl = [torch.from_numpy(rng.uniform(-0.1, 0.1, size=(B, 2, H, W)).astype(np.float32)).to(DEV).requires_grad_(True) for _ in range(S)]
        mask = torch.ones(B, 1, H, W, device=DEV)
        lossf.event_flow_association(fl, ev, pol, mask)
    v = lossf()
    torch.cuda.synchronize()
    _lib.profile_start(["evf_cm_loss_fwd"])
    vals = []
    for _ in range(5):
        vals.append(float(lossf()))
    t = _lib.profile_stop()[("evf_cm_loss_fwd", "")]
    _lib.profile_start(["evf_cm_loss_bwd"])
    for _ in range(5):
        lossf().backward()
    tb = _lib.profile_stop()[("evf_cm_loss_bwd", "")]
    return float(np.median(t)) * 1e3, vals[0], float(np.median(tb)) * 1e3


for name, shp in (("config 3: B8 128x128 10x1500 ev, 1 scale", (8, 128, 128, 10, 1500, 1)),
                  ("config 4: B8 256x256 1x50000 ev, 4 scales", (8, 256, 256, 1, 50000, 4)),
                  ("config 5: B4 260x346 10x1500 ev, 1 scale", (4, 260, 346, 10, 1500, 1))):
    a, va, ba = run(*shp, 1 << 60)
    l, vl, bl = run(*shp, 1)
    print(f"{name}: atomics {a:7.1f} us  lds {l:7.1f} us   loss {va:.6f} / {vl:.6f}   backward (2 launches) {ba:7.1f} / {bl:7.1f} us")
