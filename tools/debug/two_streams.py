"""Experiment: two independent half-batch (B=4) train-step graphs replayed concurrently on two streams vs one B=8 graph.
(Would an intra-GPU pipeline over half batches hide the per-kernel ramp / drain?)"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from event_flow_amd import _lib
from event_flow_amd.loss.flow import EventWarping
from event_flow_amd.models import model as models
from event_flow_amd.parallel import DataParallel
from event_flow_amd.train import FlatAdam

dev = "cuda:0"
torch.cuda.set_device(0)
dp = DataParallel(device=dev)


def build(B, seed):
    bench.B_PER_GPU = B
    torch.manual_seed(seed)
    model = models.LIFFireNet(dict(bench.MODEL_CFG)).to(dev)
    model.train()
    lossf = EventWarping(bench.LOSS_CFG, dev)
    opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=True)
    opt.zero_grad()
    model.use_static_states(True)
    pool = bench.make_windows(seed, 2, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            bench.run_step(model, lossf, opt, dp, pool[i % 2])
        torch.cuda.synchronize()
        graphs = bench.capture_step_graphs(model, lossf, opt, dp, pool, side)
    torch.cuda.synchronize()
    return graphs, side


def timeit(sets, steps=40):
    for g, s in sets:
        with torch.cuda.stream(s):
            g[0].replay(); g[1].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for g, s in sets:
            with torch.cuda.stream(s):
                g[i % 2].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = build(8, 0)
ms8 = timeit([one])
print(f"one stream  B=8: {ms8:.3f} ms/step  {8 / ms8 * 1e3:.0f} windows/s")
del one
a, b = build(4, 1), build(4, 2)
ms4 = timeit([a])
print(f"one stream  B=4: {ms4:.3f} ms/step  {4 / ms4 * 1e3:.0f} windows/s")
ms44 = timeit([a, b])
print(f"two streams B=4+4: {ms44:.3f} ms per pair of steps  {8 / ms44 * 1e3:.0f} windows/s")
del a, b
q = [build(2, 10 + i) for i in range(4)]
ms2 = timeit(q)
print(f"four streams B=2x4: {ms2:.3f} ms per 4 steps  {8 / ms2 * 1e3:.0f} windows/s")
ms22 = timeit(q[:2])
print(f"two streams B=2+2: {ms22:.3f} ms  {4 / ms22 * 1e3:.0f} windows/s")
