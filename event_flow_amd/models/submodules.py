"""Non-spiking conv layers on the hot path -- mirror of the used part of
reference models/submodules.py: ConvLayer (the 1x1 tanh prediction head of
every model, :12-61), ConvLayer_ (:64-83) and ConvGRU (:377-418, FireNet-ANN).
Parameter containers with the reference's names and initialisation; the
arithmetic runs in libevflow_hip.so (models/engine.py)."""

import torch
import torch.nn as nn

from .spiking_util import SURROGATE_ID


class ConvLayer(nn.Module):
    """Convolutional layer; default bias, ReLU, no downsampling, no norm.
    Reference: models/submodules.py:12-61."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None, BN_momentum=0.1,
                 w_scale=None):
        super().__init__()
        if norm is not None:
            raise NotImplementedError("BN/IN ConvLayers belong to the ANN baselines, outside the accelerated path")
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=True)
        if w_scale is not None:
            nn.init.uniform_(self.conv2d.weight, -w_scale, w_scale)
            nn.init.zeros_(self.conv2d.bias)
        if activation is not None and not hasattr(torch, activation) and activation not in SURROGATE_ID:
            raise AttributeError(activation)
        self.activation = activation
        self.norm = norm


class ConvLayer_(ConvLayer):
    """ConvLayer that takes/returns a (unused) state and allows a residual.
    Reference: models/submodules.py:64-83."""


class ConvGRU(nn.Module):
    """Convolutional GRU cell.  Reference: models/submodules.py:377-418."""

    def __init__(self, input_size, hidden_size, kernel_size, activation=None):
        super().__init__()
        padding = kernel_size // 2
        self.input_size, self.hidden_size = input_size, hidden_size
        self.reset_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.update_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.out_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        assert activation is None, "ConvGRU activation cannot be set (just for compatibility)"
        for g in (self.reset_gate, self.update_gate, self.out_gate):
            nn.init.orthogonal_(g.weight)
        for g in (self.reset_gate, self.update_gate, self.out_gate):
            nn.init.constant_(g.bias, 0.0)
