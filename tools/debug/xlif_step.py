"""Train-step time of the XLIF-FireNet (hard reset, arctan: configs/train_SNN.yml) at the headline shape (8 x 128 x 128, 10 passes x
1500 events): on the recorded window kernels (models/engine.py), replayed from hipGraphs (train.GraphedWindowStep), and -- EVF_XLIF_FUSED=0
-- cell by cell on the general path (train.capture_window_cycle).  python tools/debug/xlif_step.py [ALIFFireNet | PLIFFireNet]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from event_flow_amd import synthetic
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.models import model as models
from event_flow_amd.train import FlatAdam, GraphedWindowStep, capture_window_cycle, train_window

dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "XLIFFireNet"
bench.set_workload("c3")
cfg = dict(bench.MODEL_CFG)
if name == "XLIFFireNet":
    cfg["spiking_neuron"] = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "t0": [0.8, 0.1], "t1": [0.5, 0.1], "learn_leak": True,
                             "learn_thresh": True, "hard_reset": True}
elif name == "ALIFFireNet":
    cfg["spiking_neuron"] = {"leak_v": [-4.0, 0.1], "leak_t": [-4.0, 0.1], "t0": [0.8, 0.1], "t1": [0.5, 0.1], "learn_leak": True,
                             "learn_thresh": True, "hard_reset": True}
elif name == "PLIFFireNet":
    cfg["spiking_neuron"] = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
                             "learn_thresh": True, "hard_reset": True}
B, P, n, H, W = bench.B_PER_GPU, bench.PASSES, bench.EV_PER_PASS, bench.H, bench.W
wins = [[torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 100 * w + k)).to(dev) for k in range(P)] for w in range(2)]
torch.manual_seed(0)
model = getattr(models, name)(cfg).to(dev)
model.train()
path = model.compute_path
lossf = EventWarping(bench.LOSS_CFG, dev)
opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=True)
opt.zero_grad()
if path[0] == "fused":
    st = GraphedWindowStep(model, lossf, opt, 2, (H, W))
    for i in range(6):
        loss = st.step(wins[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        loss = st.step(wins[i % 2])
    torch.cuda.synchronize()
    print(f"{name} fused (recorded window kernels, hipGraph replay): {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per step, loss {float(loss):.4f}")
else:
    pool = [bench._encode(w) for w in bench.make_windows(0, 2, dev)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            loss = train_window(model, lossf, opt, pool[i % 2])
        graphs, _ = capture_window_cycle(model, lossf, opt, pool, side)
        torch.cuda.synchronize()
        for w in range(2):
            graphs[w][0].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            graphs[i % 2][0].replay()
        torch.cuda.synchronize()
    print(f"{name} general path ({path[1]}), hipGraph replay: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step, loss {float(graphs[1][1]):.4f}")
