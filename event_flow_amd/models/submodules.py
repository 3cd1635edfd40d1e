"""Non-spiking conv layers on the hot path -- mirror of the used part of
reference models/submodules.py: ConvLayer (the 1x1 tanh prediction head of
every model, :12-61), ConvLayer_ (:64-83) and ConvGRU (:377-418, FireNet-ANN).
The nn.Conv2d members only hold the parameters under the reference's names
and initialisation; `forward` runs in libevflow_hip.so through the general
path (models/hip_ops.py; the FireNet prediction head on packed spikes through
models/engine.py)."""

import torch
import torch.nn as nn

from . import hip_ops
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
        self.stride = stride

    def _act(self):
        if self.activation in SURROGATE_ID:
            raise NotImplementedError("stateless spike activations on a ConvLayer are not on the accelerated path")
        if self.activation not in hip_ops.ACT_ID:
            raise NotImplementedError(f"ConvLayer activation {self.activation!r} has no HIP kernel (tanh/sigmoid/relu/None)")
        return self.activation

    def forward(self, x):
        return hip_ops.conv_act(self, x, self.conv2d.weight, self.conv2d.bias, self.stride, self._act())


class ConvLayer_(ConvLayer):
    """ConvLayer that takes/returns a (unused) state and allows a residual.
    Reference: models/submodules.py:64-83."""

    def forward(self, x, prev_state, residual=0):
        if prev_state is None:
            prev_state = torch.tensor(0)  # not used (reference :71-72)
        res = residual if torch.is_tensor(residual) else None
        out = hip_ops.conv_act(self, x, self.conv2d.weight, self.conv2d.bias, self.stride, self._act(), residual=res)
        return out, prev_state


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

    def forward(self, input_, prev_state):
        """-> (new_state, new_state), reference :400-418."""
        new = hip_ops.conv_gru(self, input_, prev_state)
        return new, new
