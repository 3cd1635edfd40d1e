"""Per-launch HIP-event times of the conv entry points in one eager LIF-EV-FlowNet train step (BASELINE configs[3]), by shape."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd import _lib, synthetic
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
model.train()
lossf = EventWarping({"loader": {"resolution": [Hc, Wc]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False, "clip_grad": 100.0}, "model": {"mask_output": True}}, dev)
opt = FlatAdam(model, lr=2e-4, clip=100.0)
opt.zero_grad()
pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(Bc, nev, Hc, Wc, 4000 + 100000 * w)).to(dev)], 2, (Hc, Wc)) for w in range(2)]
for i in range(3):
    train_window(model, lossf, opt, pool[i % 2])
torch.cuda.synchronize()
names = ["evf_conv2d_fwd_b3", "evf_conv2d_fwd_b3_parts", "evf_conv2d_dgrad_b3", "evf_conv2d_wgrad", "evf_neuron_bwd", "evf_lif_fwd_parts"]
_lib.profile_start(names)
for i in range(2):
    train_window(model, lossf, opt, pool[i % 2])
prof = _lib.profile_stop()
for (name, var), ms in sorted(prof.items(), key=lambda kv: -float(np.sum(kv[1]))):
    ms = np.array(ms)
    line = f"{name:28s} {var:28s} n/step {len(ms) / 2:4.1f}  mean {ms.mean() * 1e3:7.1f} us  total/step {ms.sum() / 2 * 1e3:8.1f} us"
    if name.startswith("evf_conv2d_") and var:
        b, h, w, cin, cout, k, st = (int(v) for v in var.split(","))
        ho, wo = (h + 2 * (k // 2) - k) // st + 1, (w + 2 * (k // 2) - k) // st + 1
        fl = 2.0 * k * k * cin * cout * b * ho * wo
        terms = 6 if "dgrad" in name else 3
        line += f"  fp32-eq {fl / (ms.mean() * 1e-3) / 1e12:6.1f} TF  issued>={terms}x: {terms * fl / (ms.mean() * 1e-3) / 1e12 / 2500:5.2f} of bf16 peak"
    print(line)
