"""Debug: config 4 x0.3 -- dL/dflow per scale and dL/d(parameters) HIP vs oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from event_flow_amd import synthetic
from event_flow_amd.models.model import SpikingRecEVFlowNet
from event_flow_amd.train import encode_passes
from event_flow_amd.loss.flow import EventWarping
from oracle import train as otrain

DEV = torch.device("cuda:0")
B, n, H, W = 8, 50000, 256, 256
scale = 0.3
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
lc = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}
torch.manual_seed(0)
model = SpikingRecEVFlowNet(dict(cfg)).to(DEV)
with torch.no_grad():
    for k, p in model.named_parameters():
        if k.endswith("thresh"):
            p.mul_(scale)
model.train()
params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
keys = [k for k, p in model.named_parameters() if p.requires_grad]
ev = torch.from_numpy(synthetic.event_list_batch(B, n, H, W, synthetic.seed_for(4, 0, 0))).to(DEV)
d = encode_passes([ev], 2, (H, W))[0]
lossf = EventWarping(lc, DEV)
out = model(d["event_voxel"], d["event_cnt"])
for f in out["flow"]:
    f.retain_grad()
lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
loss = lossf()
loss.backward()
torch.cuda.synchronize()
torch.set_num_threads(32)
leaves = {k: (t.clone().requires_grad_(k in keys)) for k, t in params.items()}
opasses = [{k: v.detach().cpu() for k, v in d.items()}]
oloss_t, oflows, ostates = otrain.forward_window("SpikingRecEVFlowNet", leaves, opasses, [None] * 10, (H, W),
                                                 loss_cfg={"flow_regul_weight": 0.001, "mask_output": True}, model_cfg={"kind": "lif"})
og = torch.autograd.grad(oloss_t, list(oflows[0]) + [leaves[k] for k in keys], allow_unused=True)
print("loss", float(loss.detach()), float(oloss_t.detach()))
for i, f in enumerate(out["flow"]):
    g, r = f.grad.cpu().numpy(), og[i].numpy()
    fr = oflows[0][i].detach().numpy()
    print(f"scale {i}: flow rel {np.linalg.norm(f.detach().cpu().numpy() - fr) / np.linalg.norm(fr):.2e}  dL/dflow rel-L2 {np.linalg.norm(g - r) / np.linalg.norm(r):.3e}  max|diff| {np.abs(g - r).max():.3e} max|g| {np.abs(r).max():.3e}")
    bad = np.argwhere(np.abs(g - r) > 0.01 * np.abs(r).max())
    print("    elements off by > 1% of max:", len(bad), bad[:5].tolist())

# --- the same flows as leaves through both losses
from oracle import loss as oloss
lossf2 = EventWarping(lc, DEV)
win = oloss.Window((H, W))
gfs = [f.detach().clone().requires_grad_(True) for f in out["flow"]]
ofs = [f.detach().cpu().clone().requires_grad_(True) for f in out["flow"]]
lossf2.event_flow_association(gfs, d["event_list"], d["event_list_pol_mask"], d["event_mask"])
win.add(ofs, d["event_list"].cpu(), d["event_list_pol_mask"].cpu(), d["event_mask"].cpu())
v = lossf2(); v.backward()
r = oloss.event_warping_loss(win, max(H, W), 0.001); r.backward()
print("leaf flows: loss", float(v.detach()), float(r.detach()))
for i, (g, o) in enumerate(zip(gfs, ofs)):
    gg, rr = g.grad.cpu().numpy(), o.grad.numpy()
    print(f"   scale {i}: dL/dflow rel-L2 {np.linalg.norm(gg - rr) / np.linalg.norm(rr):.3e}; vs in-model HIP grad {np.linalg.norm(gg - out['flow'][i].grad.cpu().numpy()) / np.linalg.norm(rr):.3e}; "
          f"oracle leaf vs oracle in-model {np.linalg.norm(rr - og[i].numpy()) / np.linalg.norm(rr):.3e}; zero-flow fraction {float((o.detach() == 0).float().mean()):.3f}")
