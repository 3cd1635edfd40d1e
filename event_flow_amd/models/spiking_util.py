"""Spike functions -- mirror of reference models/spiking_util.py.

On the hot path the Heaviside forward and the surrogate-gradient backward are
fused into the neuron kernels of libevflow_hip.so; SURROGATE_ID maps the
reference's activation names onto the kernels' surrogate switch.  The
stand-alone functions below keep the reference's names and signatures
(`arctanspike(x, thresh, width)` ...) for code that calls them on tensors
directly; they run `evf_spike_fwd` / `evf_spike_bwd` on the MI355X (no CPU path).
"""

import torch

from .. import _lib

SURROGATE_ID = {"arctanspike": 0, "superspike": 1, "trianglespike": 2, "mgspike": 3}


class _Spike(torch.autograd.Function):
    """spiking_util.py:13-25 (BaseSpike) with the surrogate of :38-43 / :55-65 / :74-79 / :88-93."""

    @staticmethod
    def forward(ctx, x, thresh, width, name):
        _lib.require_gpu(x, name)
        xc = x.detach().float().contiguous()
        th = torch.as_tensor(thresh, dtype=torch.float32).to(xc.device)
        if th.requires_grad:
            raise NotImplementedError("stand-alone spike functions do not differentiate w.r.t. the threshold; "
                                      "the spiking cells (spiking_submodules.py) do")
        per_elem = th.numel() > 1
        th = th.expand_as(xc).contiguous() if per_elem else th.reshape(1).contiguous()  # broadcast like x - thresh
        z = torch.empty_like(xc)  # spikes are always float32 (spiking_util.py:21)
        _lib.call("evf_spike_fwd", _lib.ptr(xc), _lib.ptr(th), 1 if per_elem else 0, xc.numel(), _lib.ptr(z))
        ctx.save_for_backward(xc, th)
        ctx.width, ctx.name, ctx.per_elem = float(width), name, per_elem
        return z

    @staticmethod
    def backward(ctx, grad_output):
        xc, th = ctx.saved_tensors
        g = grad_output.float().contiguous()
        gx = torch.empty_like(xc)
        _lib.call("evf_spike_bwd", SURROGATE_ID[ctx.name], _lib.ptr(xc), _lib.ptr(th), 1 if ctx.per_elem else 0, _lib.ptr(g),
                  ctx.width, xc.numel(), _lib.ptr(gx))
        return gx, None, None, None


def superspike(x, thresh=torch.tensor(1.0), width=torch.tensor(10.0)):
    return _Spike.apply(x, thresh, width, "superspike")


def mgspike(x, thresh=torch.tensor(1.0), width=torch.tensor(0.5)):
    return _Spike.apply(x, thresh, width, "mgspike")


def trianglespike(x, thresh=torch.tensor(1.0), width=torch.tensor(1.0)):
    return _Spike.apply(x, thresh, width, "trianglespike")


def arctanspike(x, thresh=torch.tensor(1.0), width=torch.tensor(10.0)):
    return _Spike.apply(x, thresh, width, "arctanspike")
