"""Non-spiking conv layers on the hot path -- mirror of the used part of
reference models/submodules.py: ConvLayer (the 1x1 tanh prediction head of
every model, :12-61), ConvLayer_ (:64-83), ConvGRU (:377-418, FireNet-ANN) and
the cells and blocks of the ANN comparisons: UpsampleConvLayer (:140-185),
RecurrentConvLayer (:188-235), ResidualBlock (:238-311), ConvRecurrent
(:421-451), ConvLeakyRecurrent (:454-499), ConvLeaky (:502-554) and the leaky
blocks (:557-686).
The nn.Conv2d members only hold the parameters under the reference's names
and initialisation; `forward` runs in libevflow_hip.so through the general
path (models/hip_ops.py; the FireNet prediction head on packed spikes through
models/engine.py)."""

import torch
import torch.nn as nn

from .. import _lib
from . import hip_ops
from .spiking_util import SURROGATE_ID


class ConvLayer(nn.Module):
    """Convolutional layer; default bias, ReLU, no downsampling, no norm.
    Reference: models/submodules.py:12-61."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None, BN_momentum=0.1,
                 w_scale=None):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=norm != "BN")
        _make_norm(self, norm, out_channels, BN_momentum)
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
        if self.norm in ("BN", "IN"):  # conv -> norm -> activation (reference :52-61)
            out = hip_ops.conv_act(self, x, self.conv2d.weight, self.conv2d.bias, self.stride, None)
            return hip_ops.activation(hip_ops.norm2d(out, self.norm_layer), self._act())
        return hip_ops.conv_act(self, x, self.conv2d.weight, self.conv2d.bias, self.stride, self._act())


class ConvLayer_(ConvLayer):
    """ConvLayer that takes/returns a (unused) state and allows a residual.
    Reference: models/submodules.py:64-83."""

    def forward(self, x, prev_state, residual=0):
        if prev_state is None:
            prev_state = torch.tensor(0)  # not used (reference :71-72)
        res = residual if torch.is_tensor(residual) else None
        if self.norm in ("BN", "IN"):  # conv -> norm -> + residual -> activation (reference :74-81)
            out = hip_ops.conv_act(self, x, self.conv2d.weight, self.conv2d.bias, self.stride, None)
            return hip_ops.activation(hip_ops.norm2d(out, self.norm_layer), self._act(), res), prev_state
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


class ConvLSTM(nn.Module):
    """Convolutional LSTM cell: (input, (hidden, cell)) -> (hidden', cell').  Reference: models/submodules.py:314-374."""

    def __init__(self, input_size, hidden_size, kernel_size, activation=None):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        assert activation is None, "ConvLSTM activation cannot be set (just for compatibility)"
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=kernel_size // 2)

    def forward(self, input_, prev_state=None):
        prev_hidden, prev_cell = prev_state if prev_state is not None else (None, None)
        return hip_ops.conv_lstm(self, input_, prev_hidden, prev_cell)


def _zeros_like_state(x, channels):
    B, _, H, W = x.shape
    return _lib.zeros((B, H, W, channels), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)


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


def _make_norm(module, norm, channels, BN_momentum=0.1, name="norm_layer"):
    """The reference's norm members (models/submodules.py:46-50): parameter / buffer holders under the reference's names; the
    arithmetic runs in hip_ops.norm2d."""
    if norm == "BN":
        setattr(module, name, nn.BatchNorm2d(channels, momentum=BN_momentum))
    elif norm == "IN":
        setattr(module, name, nn.InstanceNorm2d(channels, track_running_stats=True))


def stack_nhwc(states):
    """torch.stack of logical [B,C,H,W] tensors that keeps their NHWC memory layout (a plain copy)."""
    return torch.stack([s.permute(0, 2, 3, 1) for s in states]).permute(0, 1, 4, 2, 3)


class UpsampleConvLayer(nn.Module):
    """Bilinear x2 up-sampling + ConvLayer (decoder).  Reference: models/submodules.py:140-185."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=norm != "BN")
        _make_norm(self, norm, out_channels)
        if activation is not None and not hasattr(torch, activation) and activation not in SURROGATE_ID:
            raise AttributeError(activation)
        self.activation, self.norm, self.stride = activation, norm, stride

    def forward(self, x):
        if self.activation not in hip_ops.ACT_ID:
            raise NotImplementedError(f"UpsampleConvLayer activation {self.activation!r} has no HIP kernel")
        x_up = hip_ops.upsample2x_bilinear(x)
        if self.norm in ("BN", "IN"):
            out = hip_ops.conv_act(self, x_up, self.conv2d.weight, self.conv2d.bias, self.stride, None)
            return hip_ops.activation(hip_ops.norm2d(out, self.norm_layer), self.activation)
        return hip_ops.conv_act(self, x_up, self.conv2d.weight, self.conv2d.bias, self.stride, self.activation)


class TransposedConvLayer(nn.Module):
    """Transposed conv (x2 up-sampling) decoder layer, use_upsample_conv=False.  Reference: models/submodules.py:86-137."""

    def __init__(self, in_channels, out_channels, kernel_size, activation="relu", norm=None):
        super().__init__()
        self.transposed_conv2d = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=2, padding=kernel_size // 2,
                                                    output_padding=1, bias=norm != "BN")
        if activation is not None and not hasattr(torch, activation) and activation not in SURROGATE_ID:
            raise AttributeError(activation)
        self.activation, self.norm = activation, norm
        _make_norm(self, norm, out_channels)

    def forward(self, x):
        if self.activation not in hip_ops.ACT_ID:
            raise NotImplementedError(f"TransposedConvLayer activation {self.activation!r} has no HIP kernel")
        out = hip_ops.conv_transpose(self, x, self.transposed_conv2d.weight, self.transposed_conv2d.bias)
        if self.norm in ("BN", "IN"):
            out = hip_ops.norm2d(out, self.norm_layer)
        return hip_ops.activation(out, self.activation)


class RecurrentConvLayer(nn.Module):
    """ConvLayer followed by a recurrent block (ConvGRU / ConvRecurrent).  Reference: models/submodules.py:188-235."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, recurrent_block_type="convlstm",
                 activation_ff="relu", activation_rec=None, norm=None, BN_momentum=0.1):
        super().__init__()
        assert recurrent_block_type in ["convlstm", "convgru", "convrnn"]
        self.recurrent_block_type = recurrent_block_type
        block = {"convlstm": ConvLSTM, "convgru": ConvGRU, "convrnn": ConvRecurrent}[recurrent_block_type]
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, activation_ff, norm, BN_momentum=BN_momentum)
        self.recurrent_block = block(input_size=out_channels, hidden_size=out_channels, kernel_size=3,
                                     activation=activation_rec)

    def forward(self, x, prev_state):
        x = self.conv(x)
        x, state = self.recurrent_block(x, prev_state)
        if isinstance(self.recurrent_block, ConvLSTM):
            state = (x, state)  # (hidden, cell), reference :233-234
        return x, state


class ResidualBlock(nn.Module):
    """out1 = act(conv1(x)); out2 = act(conv2(out1) + x) -> (out2, out1).  Reference: models/submodules.py:238-311."""

    def __init__(self, in_channels, out_channels, stride=1, activation="relu", downsample=None, norm=None, BN_momentum=0.1):
        super().__init__()
        if downsample is not None:
            raise NotImplementedError("ResidualBlock(downsample=...) is not used by the reference's networks")
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=norm != "BN")
        _make_norm(self, norm, out_channels, BN_momentum, "bn1")
        _make_norm(self, norm, out_channels, BN_momentum, "bn2")
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=norm != "BN")
        if activation is not None and not hasattr(torch, activation) and activation not in SURROGATE_ID:
            raise AttributeError(activation)
        self.activation, self.norm, self.stride, self.downsample = activation, norm, stride, downsample

    def forward(self, x):
        if self.activation not in hip_ops.ACT_ID:
            raise NotImplementedError(f"ResidualBlock activation {self.activation!r} has no HIP kernel")
        if self.norm in ("BN", "IN"):  # conv -> norm -> act; conv -> norm -> + x -> act (reference :286-311)
            out1 = hip_ops.conv_act(self.conv1, x, self.conv1.weight, self.conv1.bias, self.stride, None)
            out1 = hip_ops.activation(hip_ops.norm2d(out1, self.bn1), self.activation)
            out2 = hip_ops.conv_act(self.conv2, out1, self.conv2.weight, self.conv2.bias, 1, None)
            out2 = hip_ops.activation(hip_ops.norm2d(out2, self.bn2), self.activation, x)
            return out2, out1
        out1 = hip_ops.conv_act(self.conv1, x, self.conv1.weight, self.conv1.bias, self.stride, self.activation)
        out2 = hip_ops.conv_act(self.conv2, out1, self.conv2.weight, self.conv2.bias, 1, self.activation, residual=x)
        return out2, out1


class LeakyResidualBlock(nn.Module):
    """Two ConvLeaky cells, the block input added inside the second.  Reference: models/submodules.py:557-592."""

    def __init__(self, in_channels, out_channels, stride=1, feedforward_block_type="convleaky", activation="relu", **kwargs):
        super().__init__()
        assert feedforward_block_type in ["convleaky"]
        self.conv1 = ConvLeaky(in_channels, out_channels, kernel_size=3, stride=stride, activation=activation, **kwargs)
        self.conv2 = ConvLeaky(out_channels, out_channels, kernel_size=3, stride=1, activation=activation, **kwargs)

    def forward(self, x, prev_state):
        if prev_state is None:
            prev_state = [None, None]
        conv1, conv2 = prev_state
        x1, conv1 = self.conv1(x, conv1)
        x2, conv2 = self.conv2(x1, conv2, residual=x)
        return x2, stack_nhwc([conv1, conv2])


class LeakyUpsampleConvLayer(nn.Module):
    """Bilinear x2 up-sampling + ConvLeaky.  Reference: models/submodules.py:595-623."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, feedforward_block_type="convleaky",
                 activation="relu", **kwargs):
        super().__init__()
        assert feedforward_block_type in ["convleaky"]
        self.conv2d = ConvLeaky(in_channels, out_channels, kernel_size, stride=stride, activation=activation, **kwargs)

    def forward(self, x, prev_state):
        return self.conv2d(hip_ops.upsample2x_bilinear(x), prev_state)


class LeakyTransposedConvLayer(nn.Module):
    """Reference: models/submodules.py:626-641 (raises there as well)."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError


class LeakyRecurrentConvLayer(nn.Module):
    """ConvLeaky followed by ConvLeakyRecurrent.  Reference: models/submodules.py:644-686."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, recurrent_block_type="convleaky",
                 activation_ff="relu", activation_rec=None, **kwargs):
        super().__init__()
        assert recurrent_block_type in ["convleaky"]
        self.conv = ConvLeaky(in_channels, out_channels, kernel_size, stride, activation_ff, **kwargs)
        self.recurrent_block = ConvLeakyRecurrent(out_channels, out_channels, kernel_size, activation=activation_rec, **kwargs)

    def forward(self, x, prev_state):
        if prev_state is None:
            prev_state = [None, None]
        ff, rec = prev_state
        x1, ff = self.conv(x, ff)
        x2, rec = self.recurrent_block(x1, rec)
        return x2, stack_nhwc([ff, rec])
