"""Random windows through train.window_backward with the diagonal launches on and off (EVF_DEFER_FWD / EVF_DEFER_BWD):
python tools/debug/fuzz_diag.py [n] [seed] [lif|plif].  Loss and flat gradient must agree up to the float atomics of the loss (forward and backward);
'==' marks a bit-identical loss."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from event_flow_amd import _lib, train as htrain  # noqa: E402
from event_flow_amd.loss import flow as hloss  # noqa: E402
from event_flow_amd.models.model import LIFFireNet, PLIFFireNet  # noqa: E402
from event_flow_amd.train import FlatAdam  # noqa: E402

DEV = "cuda:0"
NEURON = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
PLIF = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True,
        "learn_thresh": True, "hard_reset": True}
KIND = sys.argv[3] if len(sys.argv) > 3 else "lif"
CLS, NEURON = (PLIFFireNet, PLIF) if KIND == "plif" else (LIFFireNet, NEURON)
CFG = {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False, "mask_output": True,
       "activations": ["arctanspike", "arctanspike"], "spiking_neuron": dict(NEURON)}


def run(defer, lists, H, W, seed):
    htrain.DEFER_FORWARD = htrain.DEFER_BACKWARD = defer
    torch.manual_seed(seed)
    model = CLS(dict(CFG, spiking_neuron=dict(NEURON))).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.3)
    model.train()
    lossf = hloss.EventWarping({"loader": {"resolution": [H, W]}, "loss": {"flow_regul_weight": 0.001, "overwrite_intermediate": False},
                                "model": {"mask_output": True}}, DEV)
    opt = FlatAdam(model, lr=2e-4, clip=100.0)
    opt.zero_grad()
    out = []
    for w in range(2):  # two windows: the second starts from the carried state
        loss = htrain.window_backward(model, lossf, opt, htrain.encode_passes(lists, 2, (H, W)))
        torch.cuda.synchronize()
        out.append((float(loss.detach()), opt.flat_grad.detach().cpu().numpy().copy()))
        model.detach_states()
        lossf.reset()
        opt.zero_grad()
    assert _lib.raw("evf_fwd_defer_pending") == 0 and _lib.raw("evf_bwd_defer_pending") == 0
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        B, H, W, P = int(rng.integers(1, 5)), int(rng.integers(2, 41)), int(rng.integers(2, 161)), int(rng.integers(1, 7))  # (the loss needs H, W >= 2)
        n_ev = int(rng.integers(1, 400))
        gen = torch.Generator().manual_seed(seed * 100 + it)
        lists = []
        for _ in range(P):
            ts = torch.sort(torch.rand(B, n_ev, generator=gen), dim=1).values
            ys = torch.randint(0, H, (B, n_ev), generator=gen).float()
            xs = torch.randint(0, W, (B, n_ev), generator=gen).float()
            ps = torch.randint(0, 2, (B, n_ev), generator=gen).float() * 2 - 1
            lists.append(torch.stack([ts, ys, xs, ps], dim=2).to(DEV))
        try:
            a, b = run(False, lists, H, W, it), run(True, lists, H, W, it)
        except _lib.EvflowError as e:
            print(f"B={B} H={H:2d} W={W:3d} P={P} N={n_ev:3d}  {e}", flush=True)
            bad += 1
            continue
        ok = True
        msg = []
        for w in range(2):
            rel = np.linalg.norm(a[w][1] - b[w][1]) / max(np.linalg.norm(a[w][1]), 1e-30)
            same = a[w][0] == b[w][0]
            close = abs(a[w][0] - b[w][0]) <= 1e-6 * abs(a[w][0])  # (the loss sums its images with float atomics)
            ok = ok and close and rel <= 1e-4 and np.isfinite(b[w][1]).all()
            msg.append(f"loss {'==' if same else '!='} grad rel {rel:.1e}")
        bad += 0 if ok else 1
        print(f"B={B} H={H:2d} W={W:3d} P={P} N={n_ev:3d}  " + " | ".join(msg) + ("  ok" if ok else "  FAIL"), flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
