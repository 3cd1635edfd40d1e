"""BASELINE config 4: LIF EV-FlowNet (SpikingRecEVFlowNet, base 32), 256x256, 50k events / window, batch 8,
1 x MI355X -- forward + 4-scale EventWarping loss + backward + clip/Adam, with the per-entry-point breakdown
of the general-path kernels.  usage: python tools/bench_evflownet.py [--steps 5] [--B 8] [--res 256] [--prof]"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from event_flow_amd import _lib, synthetic  # noqa: E402
from event_flow_amd.loss.flow import EventWarping  # noqa: E402
from event_flow_amd.models.model import MODELS  # noqa: E402
from event_flow_amd.train import FlatAdam, encode_passes, train_window  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--events", type=int, default=50000)
ap.add_argument("--base", type=int, default=32)
ap.add_argument("--prof", action="store_true")
ap.add_argument("--model", default="SpikingRecEVFlowNet", help="any multi-scale model of models/model.py (EVFlowNet, RecEVFlowNet, ...)")
a = ap.parse_args()

dev = "cuda:0"
torch.manual_seed(0)
cfg = {"num_bins": 2, "base_num_channels": a.base, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
       "mask_output": True, "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True,
                          "hard_reset": True}}
if not a.model.startswith(("Spiking", "PLIF", "ALIF", "XLIF")):  # the non-spiking EV-FlowNets / E2VID
    cfg["activations"] = ["relu", None]
    cfg["spiking_neuron"] = {"leak": [-4.0, 0.1], "learn_leak": True} if a.model.startswith("Leaky") else None
model = MODELS[a.model](cfg).to(dev)
model.train()
H = W = a.res
lossf = EventWarping({"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
                      "model": {"mask_output": True}}, dev)
opt = FlatAdam(model, lr=2e-4, clip=100.0)
opt.zero_grad()
ev = torch.from_numpy(synthetic.event_list_batch(a.B, a.events, H, W, synthetic.seed_for(4, 0, 0))).to(dev)
passes = encode_passes([ev], 2, (H, W))
nparam = sum(p.numel() for p in model.parameters())

NAMES = ["evf_conv2d_fwd", "evf_conv2d_dgrad", "evf_conv2d_wgrad", "evf_neuron_fwd", "evf_neuron_bwd", "evf_upsample2x_fwd",
         "evf_upsample2x_bwd", "evf_cm_loss_fwd", "evf_cm_loss_bwd", "evf_clip_adam_step", "evf_pack_conv2d_weight",
         "evf_act_fwd", "evf_act_bwd", "evf_nchw_to_nhwc"]
for _ in range(a.warmup):
    loss = train_window(model, lossf, opt, passes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = train_window(model, lossf, opt, passes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
res = {"config": f"{a.model} base{a.base} {H}x{W} B{a.B} {a.events}ev", "params": nparam, "ms_per_step": dt * 1e3,
       "windows_per_s": a.B / dt, "loss": float(loss)}
if a.prof:
    _lib.profile_start(NAMES)
    train_window(model, lossf, opt, passes)
    prof = _lib.profile_stop()
    tot = {}
    for (name, _v), ms in prof.items():
        tot[name] = (round(float(np.sum(ms)), 3), len(ms))
    res["kernel_ms(total,calls)"] = dict(sorted(tot.items(), key=lambda kv: -kv[1][0]))
print(json.dumps(res))
