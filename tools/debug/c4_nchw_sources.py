import sys, collections, traceback
import torch
sys.path.insert(0, ".")
from event_flow_amd.models import hip_ops
from event_flow_amd import synthetic
from event_flow_amd.models.model import SpikingRecEVFlowNet
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.train import FlatAdam, train_window, encode_passes
cnt = collections.Counter()
orig = hip_ops.to_nhwc
def spy(t):
    p = t.permute(0, 2, 3, 1)
    if not p.is_contiguous():
        st = traceback.extract_stack(limit=4)
        cnt[(tuple(t.shape), t.is_contiguous(), " <- ".join(f"{f.name}:{f.lineno}" for f in st[:-1][-2:]))] += 1
    return orig(t)
hip_ops.to_nhwc = spy
dev = torch.device("cuda:0")
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
torch.manual_seed(0)
model = SpikingRecEVFlowNet(dict(cfg)).to(dev)
lossf = EventWarping({"loader": {"resolution": [256, 256]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False, "clip_grad": 100.0}, "model": {"mask_output": True}}, dev)
opt = FlatAdam(model, lr=1e-4, clip=100.0)
pool = [encode_passes([torch.from_numpy(synthetic.event_list_batch(8, 50000, 256, 256, 1000 * w)).to(dev)], 2, (256, 256)) for w in range(2)]
train_window(model, lossf, opt, pool[0])
cnt.clear()
train_window(model, lossf, opt, pool[1])
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
