"""Which torch ops (cat / copy / fill / add) run in the LIF-EV-FlowNet train step, and from which source lines."""
import sys
import collections
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from event_flow_amd import synthetic
from event_flow_amd.models.model import SpikingRecEVFlowNet
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.train import FlatAdam, train_window, encode_passes

dev = torch.device("cuda:0")
Hc = Wc = 256; Bc = 8; nev = 50000
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
torch.manual_seed(0)
model = SpikingRecEVFlowNet(dict(cfg)).to(dev)
lossf = EventWarping({"loader": {"resolution": [Hc, Wc]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False, "clip_grad": 100.0}, "model": {"mask_output": True}}, dev)
opt = FlatAdam(model, lr=1e-4, clip=100.0)
pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(Bc, nev, Hc, Wc, 1000 * w)).to(dev)], 2, (Hc, Wc)) for w in range(2)]
for i in range(3):
    train_window(model, lossf, opt, pool[i % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    train_window(model, lossf, opt, pool[1])
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name in ("aten::cat", "aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::stack", "aten::zeros", "aten::mul"):
        st = [f for f in (e.stack or []) if "event_flow_amd" in f or "bench" in f]
        where = st[0].split("/root/repo/")[-1] if st else "?"
        agg[(e.name, where[:90], str(e.input_shapes)[:60])] += 1
for (n, w, sh), c in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d} {n:18s} {w:90s} {sh}")
