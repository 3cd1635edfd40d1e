"""Which torch-native device ops are left in a general-path training step (and who calls them): one eager train_window of a small
spiking EV-FlowNet under torch.profiler with stacks.   python tools/debug/torch_ops_in_step.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from event_flow_amd import synthetic  # noqa: E402
from event_flow_amd.loss.flow import EventWarping  # noqa: E402
from event_flow_amd.models.model import SpikingRecEVFlowNet  # noqa: E402
from event_flow_amd.train import FlatAdam, encode_passes, train_window  # noqa: E402

DEV = "cuda:0"
B, n, H, W = 2, 3000, 64, 64
cfg = {"num_bins": 2, "base_num_channels": 8, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 777 + 13 * w)).to(DEV)], 2, (H, W)) for w in range(2)]
torch.manual_seed(0)
model = SpikingRecEVFlowNet(dict(cfg)).to(DEV)
model.train()
lossf = EventWarping(lc, DEV)
opt = FlatAdam(model, lr=1e-3, clip=100.0, device_step=True)
opt.zero_grad()
for i in range(3):
    train_window(model, lossf, opt, pool[i % 2])
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_window(model, lossf, opt, pool[1])
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_stack_n=8):
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if ev.key.startswith("aten::") and dt > 0:
        st = [f.split("/")[-1] for f in (ev.stack or []) if "event_flow_amd" in f]
        rows.append((ev.count, ev.key, dt, " <- ".join(st[:3]) or "(autograd engine)"))
for c, name, dt, where in sorted(rows, key=lambda r: -r[2]):
    print(f"{c:4d}  {name:24s} {dt:8.1f} us  {where}")
