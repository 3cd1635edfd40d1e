"""End-to-end demo on synthetic moving dots: train a FireNet-family model with the reference's loop structure
(train_flow.py:98-171: passes -> EventWarping -> backward -> clip -> Adam -> detach) and watch the
contrast-maximisation loss and the AEE against the known motion.
usage: python tools/train_demo.py [--model LIFFireNet] [--steps 150] [--res 64] [--B 4]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_flow_amd import synthetic  # noqa: E402
from event_flow_amd.dataloader.encodings import encode_event_list  # noqa: E402
from event_flow_amd.loss.flow import AEE, EventWarping  # noqa: E402
from event_flow_amd.models import model as M  # noqa: E402
from event_flow_amd.train import FlatAdam, train_window  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="LIFFireNet")
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--passes", type=int, default=4)
ap.add_argument("--events", type=int, default=1500)
ap.add_argument("--lr", type=float, default=1e-3)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
H = W = a.res
neuron = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
if "PLIF" in a.model:
    neuron = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1]}
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"], "spiking_neuron": neuron}
model = M.MODELS[a.model](cfg).to(dev)
model.train()
conf = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
        "model": {"mask_output": True}}
lossf = EventWarping(conf, dev)
opt = FlatAdam(model, lr=a.lr, clip=100.0)
opt.zero_grad()


def window(seed):
    """P passes of one moving-dots window per sample (the window's events split in time)."""
    evs, gts = [], []
    for b in range(a.B):
        xs, ys, ts, ps, uv = synthetic.moving_dots_events(a.events * a.passes, H, W, seed + b, max_disp=6.0, k=60)
        evs.append(np.stack([ts, ys, xs, ps], 1))
        gts.append(uv)
    ev = np.stack(evs).astype(np.float32)  # [B, P*N, 4], ts in [0,1] over the whole window
    lists = []
    for k in range(a.passes):
        part = ev[:, k * a.events : (k + 1) * a.events].copy()
        t0, t1 = part[:, :1, 0], part[:, -1:, 0]
        part[:, :, 0] = (part[:, :, 0] - t0) / np.maximum(t1 - t0, 1e-9)  # per-pass normalised timestamps
        lists.append(torch.from_numpy(part).to(dev))
    return lists, gts


def evaluate():
    """Mean end-point error (pixels per pass) of the predicted flow against the dots' known motion, on event pixels."""
    errs, zero = [], []
    model.eval()
    with torch.no_grad():
        for w in range(8):
            lists, gts = window(1000 + 17 * w)
            model.reset_states()
            for ev in lists:
                d = encode_event_list(ev, 2, (H, W))
                out = model(d["event_voxel"], d["event_cnt"])
            flow = out["flow"][0] * float(max(H, W))  # pixels per pass (EventWarping flow_scaling = max(res))
            m = d["event_mask"][:, 0] > 0
            for b, (u, v) in enumerate(gts):
                gx, gy = u / a.passes, v / a.passes
                e = torch.sqrt((flow[b, 0] - gx) ** 2 + (flow[b, 1] - gy) ** 2)[m[b]]
                errs.append(float(e.mean()))
                zero.append(float(np.hypot(gx, gy)))
    model.train()
    model.reset_states()
    return float(np.mean(errs)), float(np.mean(zero))


log = []
aee0, aee_zero = evaluate()
print(f"before training: AEE {aee0:.3f} px/pass (a zero-flow prediction scores {aee_zero:.3f})", flush=True)
run = []
for step in range(a.steps):
    lists, gts = window(1000 + 17 * (step % 8))
    passes = [encode_event_list(ev, 2, (H, W)) for ev in lists]
    loss = train_window(model, lossf, opt, passes)
    model.reset_states()
    run.append(loss)
    if (step + 1) % 100 == 0 or step == a.steps - 1:
        avg = float(torch.stack(run).mean())
        run = []
        log.append((step, avg))
        print(f"step {step + 1:5d}  mean loss of the last 100 steps {avg:.4f}", flush=True)
aee1, _ = evaluate()
print(json.dumps({"model": a.model, "steps": a.steps, "loss_first_100": log[0][1], "loss_last_100": log[-1][1],
                  "aee_before": aee0, "aee_after": aee1, "aee_zero_flow": aee_zero}))
