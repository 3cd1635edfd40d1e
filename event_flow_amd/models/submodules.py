"""Non-spiking conv layers on the hot path -- mirror of the used part of
reference models/submodules.py: ConvLayer (the 1x1 tanh prediction head of
every model, :12-61), ConvLayer_ (:64-83), ConvGRU (:377-418, FireNet-ANN) and
the cells of the ANN comparisons ConvRecurrent (:421-451), ConvLeakyRecurrent
(:454-499), ConvLeaky (:502-554).
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


def _zeros_like_state(x, channels):
    B, _, H, W = x.shape
    return torch.zeros((B, H, W, channels), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)


class ConvRecurrent(nn.Module):
    """Convolutional recurrent cell: state = tanh(ff(x) + rec(state)), out = relu(out(state)).
    Reference: models/submodules.py:421-451 (the `rec` / `out` convs are declared on input_size channels there too)."""

    def __init__(self, input_size, hidden_size, kernel_size, activation=None):
        super().__init__()
        padding = kernel_size // 2
        self.input_size, self.hidden_size = input_size, hidden_size
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding)
        self.rec = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding)
        self.out = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding)
        assert activation is None, "ConvRecurrent activation cannot be set (just for compatibility)"

    def forward(self, input_, prev_state):
        if prev_state is None:  # zeros through the recurrent conv = its bias (reference :437-444)
            prev_state = _zeros_like_state(input_, self.hidden_size)
        ff = hip_ops.conv_act(self.ff, input_, self.ff.weight, self.ff.bias)
        state = hip_ops.conv_act(self.rec, prev_state, self.rec.weight, self.rec.bias, activation="tanh", residual=ff)
        out = hip_ops.conv_act(self.out, state, self.out.weight, self.out.bias, activation="relu")
        return out, state


class _LeakParam:
    def _make_leak(self, hidden_size, leak, learn_leak):
        v = torch.randn(hidden_size, 1, 1) * leak[1] + leak[0]
        if learn_leak:
            self.leak = nn.Parameter(v)
        else:
            self.register_buffer("leak", v)


class ConvLeakyRecurrent(nn.Module, _LeakParam):
    """Recurrent cell with leak: state = tanh(state l + (1 - l)(ff(x) + rec(state))), l = sigmoid(leak);
    out = relu(out(state)).  Reference: models/submodules.py:454-499."""

    def __init__(self, input_size, hidden_size, kernel_size, activation=None, leak=(-4.0, 0.1), learn_leak=True, norm=None):
        super().__init__()
        padding = kernel_size // 2
        self.input_size, self.hidden_size = input_size, hidden_size
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding)
        self.rec = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding)
        self.out = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding)
        self._make_leak(hidden_size, leak, learn_leak)
        assert activation is None, "ConvLeakyRecurrent activation cannot be set (just for compatibility)"

    def forward(self, input_, prev_state):
        ff = hip_ops.conv_act(self.ff, input_, self.ff.weight, self.ff.bias)
        if prev_state is None:
            prev_state = _zeros_like_state(input_, self.hidden_size)
        cur = hip_ops.conv_act(self.rec, prev_state, self.rec.weight, self.rec.bias, residual=ff)
        state, _ = hip_ops.leaky_mix(cur, prev_state, 0, self.leak, "tanh")
        out = hip_ops.conv_act(self.out, state, self.out.weight, self.out.bias, activation="relu")
        return out, state


class ConvLeaky(nn.Module, _LeakParam):
    """Stateful cell with leak: state = state l + (1 - l)(ff(x) + residual), out = act(state).
    Reference: models/submodules.py:502-554."""

    def __init__(self, input_size, hidden_size, kernel_size, stride=1, activation="relu", leak=(-4.0, 0.1), learn_leak=True,
                 norm=None):
        super().__init__()
        padding = kernel_size // 2
        self.input_size, self.hidden_size, self.stride = input_size, hidden_size, stride
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, stride=stride, padding=padding)
        self._make_leak(hidden_size, leak, learn_leak)
        if activation is not None and not hasattr(torch, activation) and activation not in SURROGATE_ID:
            raise AttributeError(activation)
        self.activation = activation

    def forward(self, input_, prev_state, residual=0):
        ff = hip_ops.conv_act(self.ff, input_, self.ff.weight, self.ff.bias, self.stride)
        out, state = hip_ops.leaky_mix(ff, prev_state, residual, self.leak, self.activation)
        return out, state
