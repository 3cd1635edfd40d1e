"""Diagnostic: PLIF/LIF FireNet forward spike flips and gradient errors vs the CPU oracle at a given shape."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_flow_amd import synthetic
from event_flow_amd.dataloader.encodings import encode_event_list
from event_flow_amd.loss import flow as hloss
from event_flow_amd.models.model import PLIFFireNet, LIFFireNet
from oracle import snn as osnn, train as otrain

DEV = "cuda:0"
name = sys.argv[1]; H = int(sys.argv[2]); W = int(sys.argv[3]); B = int(sys.argv[4]); P = int(sys.argv[5]); n = int(sys.argv[6])
NEUR = {"LIFFireNet": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True},
        "PLIFFireNet": {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}[name]
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"], "spiking_neuron": NEUR}
torch.manual_seed(1)
model = {"LIFFireNet": LIFFireNet, "PLIFFireNet": PLIFFireNet}[name](cfg).to(DEV)
with torch.no_grad():
    for k, p in model.named_parameters():
        if k.endswith("thresh"): p.mul_(0.25)
params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
keys = osnn.trainable_keys(params)
evs = [synthetic.event_list_batch(B, n, H, W, 5000 + 100 * k) for k in range(P)]
passes = [encode_event_list(torch.from_numpy(ev).to(DEV), 2, (H, W)) for ev in evs]
model.train()
lossf = hloss.EventWarping({"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}, DEV)
opasses = [{k: v.detach().cpu() for k, v in d.items()} for d in passes]
states = [None] * 7
LAY = ["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]
for i, d in enumerate(passes):
    out = model(d["event_voxel"], d["event_cnt"])
    lossf.event_flow_association(out["flow"], d["event_list"], d["event_list_pol_mask"], d["event_mask"])
    with torch.no_grad():
        f_ref, states = osnn.firenet_forward(name, params, opasses[i]["event_cnt"], states)
    st = model.states
    for li, ln in enumerate(LAY):
        z = st[li][1].cpu().numpy(); zr = states[li][1].numpy()
        v = st[li][0].cpu().numpy(); vr = states[li][0].numpy()
        flips = np.argwhere(z != zr)
        print(f"pass {i} {ln}: flips {len(flips)} / {z.size}  rate {zr.mean():.3f}  max|dv| {np.abs(v-vr).max():.2e}", flips[:3].tolist())
    print("  flow max diff", float((out["flow"][0].cpu() - f_ref).abs().max()))
loss = lossf(); loss.backward()
torch.set_num_threads(16)
ol, og, _, _ = otrain.train_step(name, params, keys, opasses, [None] * 7, (H, W), {"step": 0, "m": {}, "v": {}}, loss_cfg={"flow_regul_weight": 0.001, "mask_output": True})
print("loss", float(loss), ol)
for k, p in model.named_parameters():
    ref = og[k].numpy(); got = p.grad.cpu().numpy()
    print(f"{k:18s} rel {np.linalg.norm(got-ref)/max(np.linalg.norm(ref),1e-12):.2e}  norm {np.linalg.norm(ref):.2e}")
