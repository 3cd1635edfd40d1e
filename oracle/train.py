"""Oracle (PyTorch-CPU fp32) for one truncated-BPTT training step  --  test
infrastructure only.

Restates the step glue of reference train_flow.py:129-171: P forward passes
with carried state, EventWarping over the window, backward through all
passes, global-norm clip (clip_grad_norm_, :157-158), Adam (:86,162, torch
defaults betas=(0.9,0.999), eps=1e-8), detach_states (:170).
"""

import torch

from . import loss as oloss
from . import snn


def forward_window(name, params, passes, states, res, *, loss_cfg, model_cfg=None):
    """passes: list of dicts with event_cnt/event_voxel [B,C,H,W], event_list
    [B,N,4], event_list_pol_mask [B,N,2], event_mask [B,1,H,W].
    Returns (loss, flows per pass, new_states)."""
    model_cfg = model_cfg or {}
    win = oloss.Window(res)
    flows = []
    for d in passes:
        x = d["event_cnt"] if model_cfg.get("encoding", "cnt") == "cnt" else d["event_voxel"]
        if name in snn.FIRENET_KINDS or name == "FireNet":
            flow, states = snn.firenet_forward(
                name, params, x, states, acts=model_cfg.get("activations", ("arctanspike", "arctanspike")),
                hard_reset=model_cfg.get("hard_reset"),
            )
            flow_list = [flow]
        else:
            flow_list, states = snn.spiking_unet_forward(
                model_cfg["kind"], params, x, states, hard_reset=model_cfg.get("hard_reset")
            )
        win.add(flow_list, d["event_list"], d["event_list_pol_mask"], d["event_mask"])
        flows.append(flow_list)
    if loss_cfg.get("overwrite_intermediate", False):
        win.overwrite(flows[-1])
    loss = oloss.event_warping_loss(
        win,
        loss_cfg.get("flow_scaling", max(res)),
        loss_cfg["flow_regul_weight"],
        smoothing_mask=loss_cfg.get("mask_output", True),
        overwrite=loss_cfg.get("overwrite_intermediate", False),
    )
    return loss, flows, states


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): returns (scaled grads, total_norm)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update, no weight decay / amsgrad."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = v.sqrt() / (bc2**0.5) + eps
    return p - (lr / bc1) * m / denom, m, v


def train_step(name, params, keys, passes, states, res, opt_state, *, loss_cfg, model_cfg=None, lr=2e-4, clip=100.0):
    """One optimizer step.  params: dict name->tensor (leaves).  keys: trainable
    names.  opt_state: dict(step, m{}, v{}) updated in place.
    Returns (loss value, grads dict (pre-clip), new params dict, detached states)."""
    leaves = {k: (t.detach().clone().requires_grad_(k in keys)) for k, t in params.items()}
    loss, _, states = forward_window(name, leaves, passes, states, res, loss_cfg=loss_cfg, model_cfg=model_cfg)
    glist = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaves[k])) for k, g in zip(keys, glist)}
    cl = [grads[k] for k in keys]
    if clip is not None:
        cl, _ = clip_grad_norm(cl, clip)
    opt_state["step"] += 1
    newp = {k: t.detach() for k, t in leaves.items()}
    for k, g in zip(keys, cl):
        m = opt_state["m"].get(k, torch.zeros_like(g))
        v = opt_state["v"].get(k, torch.zeros_like(g))
        newp[k], opt_state["m"][k], opt_state["v"][k] = adam_step(newp[k], g, m, v, opt_state["step"], lr)
    return float(loss.detach()), grads, newp, snn.detach_states(states)
