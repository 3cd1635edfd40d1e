"""Does train.GraphedWindowStep follow the eager steps when it is NOT the first network of the process?  python tools/debug/graphed_repeat.py NAME [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_xlif as T
from event_flow_amd.models.model import PLIFFireNet, LIFFireNet
from event_flow_amd.train import GraphedWindowStep, train_window, FlatAdam
from event_flow_amd.loss import flow as hloss
from event_flow_amd.dataloader.encodings import encode_event_list
from event_flow_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "XLIFFireNet"
nets = dict(T.NETS)
nets["PLIFFireNet"] = (PLIFFireNet, {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True}, "leak_pt")
nets["LIFFireNet"] = (LIFFireNet, {"leak": [-4.0, 0.1], "thresh": [0.3, 0.05], "learn_leak": True, "learn_thresh": True, "hard_reset": True}, "leak")
cls, neuron, _ = nets[name]
B, n, H, W, P = 2, 600, 32, 64, 3
DEV = T.DEV
wins = [[torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 7000 + 100 * w + k)).to(DEV) for k in range(P)] for w in range(2)]
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    def make():
        torch.manual_seed(11)
        m = cls(T.cfg(neuron)).to(DEV); m.train(); return m
    m1 = make(); opt1 = FlatAdam(m1, lr=2e-4, clip=100.0, device_step=True); opt1.zero_grad()
    st = GraphedWindowStep(m1, hloss.EventWarping(T.loss_cfg(H, W), DEV), opt1, 2, (H, W))
    got = [float(st.step(wins[i % 2])) for i in range(8)]
    m2 = make(); opt2 = FlatAdam(m2, lr=2e-4, clip=100.0); opt2.zero_grad(); l2 = hloss.EventWarping(T.loss_cfg(H, W), DEV)
    ref = []
    for i in range(8):
        passes = [encode_event_list(ev, 2, (H, W)) for ev in wins[i % 2]]
        ref.append(float(train_window(m2, l2, opt2, passes)))
    print(name, "rep", rep, "rel diff per step", ["%.1e" % (abs(a - b) / abs(b)) for a, b in zip(got, ref)], flush=True)
    del st, m1, m2, opt1, opt2
