"""Multi-GPU readiness without a multi-GPU node: the step a rank replays with several ranks against the ONE-graph step of the
single-GPU run, bit for bit, in its three forms: (b) ONE hipGraph with evf_allreduce_sum (ncclAllReduce on the library's own RCCL
communicator, bootstrapped from torch's store) captured as a node -- the default; (c) the same collective launched eagerly between
TWO hipGraphs (binning ... backward, then clip + Adam ... reset; EVF_DP_TWO_GRAPHS=1); (d) torch.distributed's all_reduce between
the two graphs (EVF_DP_NATIVE=0: round 4's form).  A one-rank SUM all-reduce is the identity, and with a loss whose backward is order-
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


def run(force, steps, two_graphs=False, native=True):
    bench.H, bench.W, bench.B_PER_GPU = H, W, B  # (bench._encode bins at the module's resolution)
    os.environ["EVF_DP_TWO_GRAPHS"] = "1" if two_graphs else "0"
    os.environ["EVF_DP_NATIVE"] = "1" if native else "0"
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
           "two_graphs": two, "norm": float(opt.norm_ws[0].sqrt()), "count": float(opt.norm_ws[1]), "capturable": bool(dp.capturable),
           "rccl_version": getattr(dp, "native_version", None)}
    if dp.native is not None:  # (the process group itself stays up for the next run)
        from event_flow_amd import _lib

        torch.cuda.synchronize()
        _lib.load().evf_comm_destroy(dp.native)
        dp.native = None
    return out, dp


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    steps = 4
    a, _ = run(False, steps)
    b, dp = run(True, steps)                      # one graph, evf_allreduce_sum captured
    c, _ = run(True, steps, two_graphs=True)      # two graphs, evf_allreduce_sum eager between them
    d, _ = run(True, steps, native=False)         # two graphs, torch.distributed all_reduce between them

    def same(x, y):
        return {"params": bool(torch.equal(x["param"], y["param"])),
                "moments": bool(torch.equal(x["m"], y["m"]) and torch.equal(x["v"], y["v"])),
                "states": all(bool(torch.equal(p, q)) for p, q in zip(x["states"], y["states"]))}

    res = {
        "one_graph_step_is_one_graph": not a["two_graphs"], "captured_rccl_step_is_one_graph": (not b["two_graphs"]) and b["capturable"],
        "forced_step_is_two_graphs": c["two_graphs"] and d["two_graphs"], "torch_path_not_capturable": not d["capturable"],
        "backend": dp.backend, "rccl_version": b["rccl_version"], "updates": [a["count"], b["count"], c["count"], d["count"]],
        "grad_norm": [a["norm"], b["norm"], c["norm"], d["norm"]],
        "captured_vs_plain": same(a, b), "two_graph_native_vs_plain": same(a, c), "two_graph_torch_vs_plain": same(a, d),
        "params_bitwise_equal": bool(torch.equal(a["param"], b["param"]) and torch.equal(a["param"], c["param"]) and torch.equal(a["param"], d["param"])),
        "moments_bitwise_equal": all(bool(torch.equal(a[k], x[k])) for k in ("m", "v") for x in (b, c, d)),
        "states_bitwise_equal": all(bool(torch.equal(p, q)) for x in (b, c, d) for p, q in zip(a["states"], x["states"])),
        "max_abs_param_diff": max(float((a["param"] - x["param"]).abs().max()) for x in (b, c, d)),
        "trained": bool((a["param"] != 0).any()),
    }
    print(json.dumps(res), flush=True)
    dp.close()


if __name__ == "__main__":
    main()
