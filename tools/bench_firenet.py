"""Train-step timing of any FireNet-family model at any shape (eager launches), e.g. BASELINE config 5:
  python tools/bench_firenet.py --model PLIFFireNet --H 260 --W 346 --B 4 --passes 10 --events 1500"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_flow_amd import synthetic  # noqa: E402
from event_flow_amd.dataloader.encodings import encode_event_list  # noqa: E402
from event_flow_amd.loss.flow import EventWarping  # noqa: E402
from event_flow_amd.models import model as M  # noqa: E402
from event_flow_amd.train import FlatAdam, train_window  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="PLIFFireNet")
ap.add_argument("--H", type=int, default=260)
ap.add_argument("--W", type=int, default=346)
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--passes", type=int, default=10)
ap.add_argument("--events", type=int, default=1500)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--graph", action="store_true", help="replay the step from hipGraphs (train.GraphedWindowStep; fused LIF/PLIF FireNets)")
ap.add_argument("--trace-loss", action="store_true", help="print the loss of every step (debugging: the same data every step)")
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
neuron = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
if a.model.startswith("PLIF"):
    neuron = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
              "learn_thresh": True, "hard_reset": True}
elif a.model.startswith(("ALIF", "XLIF")):
    neuron = {"leak_v": [-4.0, 0.1], "t0": [0.3, 0.05], "t1": [0.5, 0.1], "learn_leak": True, "learn_thresh": True}
acts = ["arctanspike", "arctanspike"]
ANN = {"FireNet": (["relu", None], None), "RNNFireNet": (["relu", None], None), "FireFlowNet": (["relu", "relu"], None),
       "LeakyFireNet": (["relu", None], {"leak": [-4.0, 0.1], "learn_leak": True}),
       "LeakyFireFlowNet": (["relu", "relu"], {"leak": [-4.0, 0.1], "learn_leak": True})}
if a.model in ANN:
    acts, neuron = ANN[a.model]
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": acts, "spiking_neuron": neuron}
model = M.MODELS[a.model](cfg).to(dev)
model.train()
lossf = EventWarping({"loader": {"resolution": [a.H, a.W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
                      "model": {"mask_output": True}}, dev)
opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=a.graph)
opt.zero_grad()
lists = [torch.from_numpy(synthetic.event_list_batch(a.B, a.events, a.H, a.W, synthetic.seed_for(5, 0, k))).to(dev)
         for k in range(a.passes)]


if a.graph:
    from event_flow_amd.train import GraphedWindowStep

    stepper = GraphedWindowStep(model, lossf, opt, 2, (a.H, a.W))
    a.warmup = max(a.warmup, 4)  # 2 eager steps + capture + first replays


def step():
    if a.graph:
        return stepper.step(lists)
    passes = [encode_event_list(ev, 2, (a.H, a.W)) for ev in lists]
    return train_window(model, lossf, opt, passes)


for _ in range(a.warmup):
    l_ = step()
    if a.trace_loss:
        print("warmup loss", float(l_), file=sys.stderr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
    if a.trace_loss:
        print("loss", float(loss), file=sys.stderr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"model": a.model, "shape": [a.B, a.H, a.W], "passes": a.passes, "events_per_pass": a.events, "launch": "hipgraph" if a.graph else "eager",
                  "ms_per_step": dt * 1e3, "windows_per_s": a.B / dt, "loss": float(loss)}))
