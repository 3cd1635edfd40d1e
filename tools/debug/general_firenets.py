"""Train-step time of the FireNet family on the GENERAL path (cell by cell) at the headline shape (8 x 128 x 128, 10 passes x 1500
events), replayed from hipGraphs (train.capture_window_cycle), beside the fused LIF-FireNet engine's figure."""
import sys
import time
import torch
sys.path.insert(0, ".")
import bench
from event_flow_amd import _lib, synthetic
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.models import model as models
from event_flow_amd.train import FlatAdam, train_window, capture_window_cycle

dev = "cuda:0"
names = sys.argv[1:] or ["LIFFireNet", "ALIFFireNet", "XLIFFireNet", "FireNet", "LeakyFireNet", "RNNFireNet"]
bench.set_workload("c3")
for name in names:
    torch.manual_seed(0)
    cfg = dict(bench.MODEL_CFG)
    if name not in ("LIFFireNet", "PLIFFireNet"):
        cfg.pop("spiking_neuron", None)  # (the class defaults: ALIF / XLIF have their own parameter names)
    if name in ("FireNet", "LeakyFireNet", "RNNFireNet", "FireFlowNet", "LeakyFireFlowNet"):
        cfg["activations"] = ["relu", None]
    model = getattr(models, name)(cfg).to(dev)
    model.train()
    path = getattr(model, "compute_path", ("?", ""))
    lossf = EventWarping(bench.LOSS_CFG, dev)
    opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=True)
    opt.zero_grad()
    pool = [bench._encode(w) for w in bench.make_windows(0, 2, dev)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            loss = train_window(model, lossf, opt, pool[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4):
            loss = train_window(model, lossf, opt, pool[i % 2])
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 4 * 1e3
        line = f"{name:14s} path {path[0]:8s} eager {eager:8.2f} ms"
        if path[0] != "fused":
            try:
                graphs, _ = capture_window_cycle(model, lossf, opt, pool, side)
                torch.cuda.synchronize()
                for w in range(2):
                    graphs[w][0].replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(20):
                    graphs[i % 2][0].replay()
                torch.cuda.synchronize()
                line += f"  hipgraph {(time.perf_counter() - t0) / 20 * 1e3:8.2f} ms  loss {float(graphs[1][1]):.4f}"
            except Exception as e:  # noqa: BLE001
                line += f"  capture failed: {type(e).__name__}: {e}"
    print(line, flush=True)
