"""GraphedWindowStep against eager train_window in lockstep (different windows, states carried): losses per step.
  python tools/debug/graphed_step_check.py B n H W P [steps] [want: cnt,mask,pol] [same]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from event_flow_amd import synthetic  # noqa: E402
from event_flow_amd.loss.flow import EventWarping  # noqa: E402
from event_flow_amd.models import model as M  # noqa: E402
from event_flow_amd.train import FlatAdam, GraphedWindowStep, encode_passes, train_window  # noqa: E402

B, n, H, W, P = (int(v) for v in sys.argv[1:6])
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 8
want = tuple(sys.argv[7].split(",")) if len(sys.argv) > 7 else ("cnt", "mask", "pol")
same = len(sys.argv) > 8
dev = "cuda:0"
cfg = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"],
       "spiking_neuron": {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}}
lcfg = {"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False}, "model": {"mask_output": True}}


def make():
    torch.manual_seed(int(os.environ.get("SEED", "5")))
    m = M.MODELS["LIFFireNet"](cfg).to(dev)
    m.train()
    return m


wins = [[torch.from_numpy(synthetic.event_list_batch(B, n, H, W, (synthetic.seed_for(5, 0, k) if os.environ.get("TOOLEV") else 9000 + (0 if same else 100 * w) + k))).to(dev) for k in range(P)] for w in range(steps)]
m1 = make()
o1 = FlatAdam(m1, lr=2e-4, clip=100.0, device_step=True)
o1.zero_grad()
if not os.environ.get("ONEMODEL"):
    m2 = make()
    o2 = FlatAdam(m2, lr=2e-4, clip=100.0)
    o2.zero_grad()
st = GraphedWindowStep(m1, EventWarping(lcfg, dev), o1, 2, (H, W), want=want)
l2 = EventWarping(lcfg, dev)
for w, lists in enumerate(wins):
    if os.environ.get("DEVSYNC") and w >= int(os.environ["DEVSYNC"]):
        torch.cuda.synchronize()
    a = float(st.step(lists))
    passes = encode_passes(lists, 2, (H, W), want=want)
    for d in passes:
        d.setdefault("event_voxel", None)
    if os.environ.get("NOEAGER"):
        print(f"step {w}: graphed {a:.6g}")
        continue
    b = float(train_window(m2, l2, o2, passes))
    dv = max(float((x[0] - y[0]).abs().max()) for x, y in zip(m1.states, m2.states))
    print(f"step {w}: graphed {a:.6g} eager {b:.6g}  max|dv| {dv:.3g}  gnorm {float(o1.grad_norm()):.4g} / {float(o2.grad_norm()):.4g}")
