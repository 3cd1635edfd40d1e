"""Multi-GPU readiness without a multi-GPU node: the step a rank replays with several ranks -- TWO hipGraphs (binning ... backward,
then clip + Adam ... reset) around the eager RCCL all-reduce of [flat gradient | loss | new_seq] -- against the ONE-graph step of
the single-GPU run, bit for bit.  A one-rank SUM all-reduce is the identity, and with a loss whose backward is order-
deterministic (no float atomics) everything else in the step is too, so after the same number of updates the parameters, Adam
moments and recurrent states must be EQUAL.  Prints one JSON line.  (tests/test_gpu_training.py runs it in its own process:
the process group is initialised with the nccl backend at world size 1.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from event_flow_amd import synthetic  # noqa: E402
from event_flow_amd.models.model import LIFFireNet  # noqa: E402
from event_flow_amd.parallel import DataParallel  # noqa: E402
from event_flow_amd.train import FlatAdam  # noqa: E402

DEV = "cuda:0"
B, n, H, W, P = 2, 600, 32, 64, 3


class LinearWindowLoss:
    """sum over the passes of <flow_t, w_t> through torch ops: a deterministic backward, one upstream gradient tensor per pass
    (same duck type as loss.flow.EventWarping for train.window_backward)."""
    overwrite_intermediate = False

    def __init__(self, weights):
        self.w, self.flows = weights, []

    def event_flow_association(self, flow_list, event_list, pol_mask, event_mask):
        self.flows.append(flow_list[0])

    def __call__(self):
        return sum((f * self.w[k % len(self.w)]).sum() for k, f in enumerate(self.flows))

    def reset(self):
        self.flows = []


def run(force, steps):
    bench.H, bench.W, bench.B_PER_GPU = H, W, B  # (bench._encode bins at the module's resolution)
    dp = DataParallel(device=DEV, force_collectives=force)
    torch.manual_seed(3)
    cfg = dict(bench.MODEL_CFG)
    model = LIFFireNet(cfg).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.25)
    model.train()
    gw = torch.Generator(device="cpu").manual_seed(9)
    lossf = LinearWindowLoss([(torch.randn(B, 2, H, W, generator=gw) * 0.02).to(DEV) for _ in range(P)])
    opt = FlatAdam(model, lr=2e-4, clip=100.0, device_step=True)
    opt.zero_grad()
    model.use_static_states(True)
    pool = [[torch.from_numpy(synthetic.event_list_batch(B, n, H, W, 7100 + 100 * w + k)).to(DEV) for k in range(P)] for w in range(2)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            bench.run_step(model, lossf, opt, dp, pool[i % 2])
        torch.cuda.synchronize()
        graphs = bench.capture_step_graphs(model, lossf, opt, dp, pool, side)
        torch.cuda.synchronize()
        for i in range(steps):
            graphs[i % 2].replay()
        torch.cuda.synchronize()
        model.set_state_buffers(graphs[(steps - 1) % 2].left)
    torch.cuda.current_stream().wait_stream(side)
    two = graphs[0].post is not None
    out = {"param": opt.flat_param.clone(), "m": opt.m.clone(), "v": opt.v.clone(), "states": [s.clone() for s in model.states],
           "two_graphs": two, "norm": float(opt.norm_ws[0].sqrt()), "count": float(opt.norm_ws[1])}
    return out, dp


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    steps = 4
    a, _ = run(False, steps)
    b, dp = run(True, steps)
    res = {
        "one_graph_step_is_one_graph": not a["two_graphs"], "forced_step_is_two_graphs": b["two_graphs"],
        "backend": dp.backend, "updates": [a["count"], b["count"]], "grad_norm": [a["norm"], b["norm"]],
        "params_bitwise_equal": bool(torch.equal(a["param"], b["param"])),
        "moments_bitwise_equal": bool(torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"])),
        "states_bitwise_equal": all(bool(torch.equal(x, y)) for x, y in zip(a["states"], b["states"])),
        "max_abs_param_diff": float((a["param"] - b["param"]).abs().max()),
        "trained": bool((a["param"] != 0).any()),
    }
    print(json.dumps(res), flush=True)
    dp.close()


if __name__ == "__main__":
    main()
