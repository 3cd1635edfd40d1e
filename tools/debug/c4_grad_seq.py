"""Debug: config-4 gradient per parameter at thresholds x0.3: default bf16x3 path vs variants, one process."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd import synthetic, _lib
from event_flow_amd.models.model import SpikingRecEVFlowNet
from event_flow_amd.models import hip_ops
from event_flow_amd.train import encode_passes
from event_flow_amd.loss.flow import EventWarping

DEV = torch.device("cuda:0")
B, n, H, W = 8, 50000, 256, 256
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
L = _lib.load()


def run(scale, fwd_mode, bwd_mode):
    torch.manual_seed(0)
    model = SpikingRecEVFlowNet(dict(cfg)).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(scale)
    model.train()
    ev = torch.from_numpy(synthetic.event_list_batch(B, n, H, W, synthetic.seed_for(4, 0, 0))).to(DEV)
    d = encode_passes([ev], 2, (H, W))[0]
    lossf = EventWarping(lc, DEV)
    L.evf_conv_tile_select(fwd_mode[0]); L.evf_conv_split_select(fwd_mode[1])
    out = model(d["event_voxel"], d["event_cnt"])
    lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    loss = lossf()
    torch.cuda.synchronize()
    L.evf_conv_tile_select(bwd_mode[0]); L.evf_conv_split_select(bwd_mode[1])
    loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}, float(loss.detach())


sc = 0.3
run(1.0, (-1, 0), (-1, 0))  # dirty the allocator like the test sequence does
ref, lref = run(sc, (0, 1), (0, 1))  # general kernel, unsplit
for name, fm, bm in (("default", (-1, 0), (-1, 0)), ("fwd default, bwd plain", (-1, 0), (0, 1)), ("fwd plain, bwd default", (0, 1), (-1, 0)),
                     ("tile unsplit", (-1, 1), (-1, 1)), ("general split", (0, 0), (0, 0))):
    g, l = run(sc, fm, bm)
    num = sum(float(((g[k] - ref[k]) ** 2).sum()) for k in ref)
    den = sum(float((ref[k] ** 2).sum()) for k in ref)
    print(f"{name:28s}: loss {l:.8f} (ref {lref:.8f})  grad rel {np.sqrt(num / den):.3e}")
    bad = sorted(((float(np.sqrt(((g[k] - ref[k]) ** 2).sum())), float(np.sqrt((ref[k] ** 2).sum())), k) for k in ref), reverse=True)[:3]
    for e, m, k in bad:
        print(f"      {k:56s} |diff| {e:.3e}  |g| {m:.3e}")
