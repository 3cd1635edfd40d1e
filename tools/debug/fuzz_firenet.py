"""Random-shape sweep of the fused FireNet path against the CPU oracle (forward flow, states, parameter gradients over
two passes): python tools/debug/fuzz_firenet.py [n_shapes] [seed].  Exercises ragged tiles, odd unit counts of the fused
backward, one-pixel images."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from event_flow_amd.models.model import LIFFireNet, PLIFFireNet  # noqa: E402
from oracle import snn as osnn  # noqa: E402

DEV = "cuda:0"
NEURON = {"leak": [-4.0, 0.1], "thresh": [0.8, 0.1], "learn_leak": True, "learn_thresh": True, "hard_reset": True}
PLIF = {"leak_v": [-4.0, 0.1], "leak_pt": [-4.0, 0.1], "add_pt": [-2.0, 0.1], "thresh": [0.8, 0.1],
        "learn_leak": True, "learn_thresh": True, "hard_reset": True}


def cfg(neuron):
    return {"num_bins": 2, "base_num_channels": 32, "kernel_size": 3, "encoding": "cnt", "norm_input": False,
            "mask_output": True, "activations": ["arctanspike", "arctanspike"], "spiking_neuron": dict(neuron)}


def run(name, cls, neuron, B, H, W, seed):
    torch.manual_seed(seed)
    model = cls(cfg(neuron)).to(DEV)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("thresh"):
                p.mul_(0.2)
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k, _ in model.named_parameters():
        params[k].requires_grad_(True)
    xs = [(torch.rand(B, 2, H, W) < 0.6).float() * torch.randint(1, 4, (B, 2, H, W)).float() for _ in range(2)]
    states = [None] * 7
    tot, tot_ref = 0, 0
    model.train()
    nflip, margin = 0, 1.0
    for x in xs:
        f_ref, states = osnn.firenet_forward(name, params, x, states)
        f = model(x.to(DEV), x.to(DEV))["flow"][0]
        wgt = torch.arange(f_ref.numel()).view(f_ref.shape).remainder(5).float() - 2.0
        tot_ref = tot_ref + (f_ref * wgt).sum()
        tot = tot + (f * wgt.to(DEV)).sum()
        for li, ln in enumerate(["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]):  # flips after EVERY pass
            z, z_ref = model.states[li][1].cpu().numpy(), states[li][1].detach().numpy()
            bad = z != z_ref
            if bad.any():
                if nflip == 0:  # the first layer with a flip (inputs still identical): the oracle's distance to the threshold there
                    th = params[ln + ".thresh"].detach().clamp_min(0.01).numpy().reshape(1, -1, 1, 1)
                    margin = float(np.abs(states[li][0].detach().numpy() - th)[bad].max())
                nflip += int(bad.sum())
    ferr = float((f.detach().cpu() - f_ref.detach()).abs().max())
    tot.backward()
    tot_ref.backward()
    worst = 0.0
    refs = {k: (params[k].grad.numpy() if params[k].grad is not None else np.zeros(tuple(p.shape), np.float32))
            for k, p in model.named_parameters()}
    scale = max(np.linalg.norm(r) for r in refs.values())
    for k, p in model.named_parameters():
        # a parameter whose gradient nearly cancels (the 2-element bias of the prediction head under the +-2 weights of
        # this loss) is judged against the scale of the whole gradient, not against its own tiny norm
        denom = max(np.linalg.norm(refs[k]), 1e-4 * scale, 1e-12)
        worst = max(worst, float(np.linalg.norm(p.grad.cpu().numpy() - refs[k]) / denom))
    return nflip, ferr, worst, margin


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 41)), int(rng.integers(1, 201))
        name, cls, neuron = (("LIFFireNet", LIFFireNet, NEURON) if it % 3 else ("PLIFFireNet", PLIFFireNet, PLIF))
        nflip, ferr, worst, margin = run(name, cls, neuron, B, H, W, seed * 1000 + it)
        ok = nflip > 0 or (ferr <= 1e-4 and worst <= 2e-3)
        bad += 0 if ok else 1
        print(f"{name:12s} B={B} H={H:3d} W={W:3d}  flips={nflip:3d}  max|dflow|={ferr:.2e}  worst grad rel={worst:.2e}  {'ok' if ok else 'FAIL'}" + (f"  (largest |v - thresh| among the first layer's flipped neurons: {margin:.1e})" if nflip else ""),
              flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
