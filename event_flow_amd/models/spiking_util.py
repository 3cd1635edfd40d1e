"""Spike functions -- mirror of reference models/spiking_util.py.

On the hot path the Heaviside forward and the surrogate-gradient backward are
fused into the neuron kernels of libevflow_hip.so (`evf_conv_lif_fwd`,
`evf_lif_bwd`); SURROGATE_ID maps the reference's activation names onto the
kernels' surrogate switch.  The stand-alone functions below keep the reference's
names and semantics for code that calls them directly on tensors (they are a
few elementwise torch ops, not part of the accelerated path)."""

from math import pi

import torch

SURROGATE_ID = {"arctanspike": 0, "superspike": 1, "trianglespike": 2, "mgspike": 3}


def gaussian(x, mu, sigma):
    """Gaussian PDF with broadcasting (spiking_util.py:6-10)."""
    return torch.exp(-((x - mu) * (x - mu)) / (2 * sigma * sigma)) / (sigma * (2 * pi) ** 0.5)


def surrogate_gradient(name, x, width):
    """d spike / d x at x = v - thresh (spiking_util.py:38-43,55-65,74-79,88-93)."""
    if name == "arctanspike":
        return 1 / (1 + width * x * x)
    if name == "superspike":
        return 1 / (1 + width * x.abs()) ** 2
    if name == "trianglespike":
        return torch.relu(1 - width * x.abs())
    if name == "mgspike":
        return 1.15 * gaussian(x, 0.0, width) - 0.15 * gaussian(x, width, 6 * width) - 0.15 * gaussian(x, -width, 6 * width)
    raise AttributeError(name)


class _Spike(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, width, name):
        ctx.save_for_backward(x)
        ctx.width, ctx.name = width, name
        return x.gt(0).float()  # spikes are always float32 (spiking_util.py:21)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        return grad_output * surrogate_gradient(ctx.name, x, ctx.width), None, None


def superspike(x, thresh=torch.tensor(1.0), width=torch.tensor(10.0)):
    return _Spike.apply(x - thresh, width, "superspike")


def mgspike(x, thresh=torch.tensor(1.0), width=torch.tensor(0.5)):
    return _Spike.apply(x - thresh, width, "mgspike")


def trianglespike(x, thresh=torch.tensor(1.0), width=torch.tensor(1.0)):
    return _Spike.apply(x - thresh, width, "trianglespike")


def arctanspike(x, thresh=torch.tensor(1.0), width=torch.tensor(10.0)):
    return _Spike.apply(x - thresh, width, "arctanspike")
