"""Spiking convolutional cells -- host-side mirror of reference
models/spiking_submodules.py (same class names, constructor arguments,
parameter names and initial distributions, so reference state_dicts load).

`ff` / `rec` are nn.Conv2d modules only so that the state_dict keys
(`ff.weight`, `rec.weight`) and the RNG draw order of the reference
constructors (:63-75, :475-490) are reproduced; they are never called.  The
arithmetic of a cell step -- conv, neuron update, Heaviside, surrogate-gradient
backward -- runs in libevflow_hip.so:
  * whole FireNets are sequenced by models/engine.py (fused 32->32 kernels);
  * a cell called on its own, `cell(input_, prev_state, residual=0) ->
    (out, state)` exactly as in the reference, and the blocks of the spiking
    EV-FlowNet below go through the general path (models/hip_ops.py).
"""

import math

import torch
import torch.nn as nn

from . import hip_ops
from .spiking_util import SURROGATE_ID


def _param_or_buffer(mod, name, value, learn):
    if learn:
        setattr(mod, name, nn.Parameter(value))
    else:
        mod.register_buffer(name, value)


class _SpikingCell(nn.Module):
    kind = None
    recurrent = False
    num_states = 2

    def _common(self, input_size, hidden_size, kernel_size, stride, activation, act_width, hard_reset, detach, norm):
        assert isinstance(
            activation, str
        ), "Spiking neurons need a valid activation, see models/spiking_util.py for choices"
        if activation not in SURROGATE_ID:
            raise AttributeError(activation)  # reference: getattr(spiking, activation)
        # norm (reference spiking_submodules.py:87-94, :502-514): only the LIF cells look at it; the other kinds take the argument
        # and ignore it, as the reference does
        self.wnorm = self.gnorm = False
        if self.kind == "lif" and norm == "weight":
            self.ff = nn.utils.weight_norm(self.ff)  # parameters ff.weight_g / ff.weight_v, the reference's state_dict keys
            if self.recurrent:
                self.rec = nn.utils.weight_norm(self.rec)
            self.wnorm = True
        elif self.kind == "lif" and norm == "group":
            # nn.GroupNorm(1, C) on the input (and, recurrent cell, on the previous spikes): parameter holders under the
            # reference's names (norm | norm_ff, norm_rec); the arithmetic is hip_ops.group_norm1.  `min(1, n // 4)` groups is
            # the reference's expression: 1 for n >= 4, and nn.GroupNorm's own error below that.
            if self.recurrent:
                self.norm_ff = nn.GroupNorm(min(1, input_size // 4), input_size)
                self.norm_rec = nn.GroupNorm(min(1, hidden_size // 4), hidden_size)
            else:
                self.norm = nn.GroupNorm(min(1, input_size // 4), input_size)
            self.gnorm = True
        if self.kind == "lif" and not self.gnorm:
            # attribute parity: the reference's LIF cells always carry the attribute(s), None when no group norm (:87-94, :502-514)
            if self.recurrent:
                self.norm_ff = self.norm_rec = None
            else:
                self.norm = None
        self.input_size, self.hidden_size = input_size, hidden_size
        self.kernel_size, self.stride = kernel_size, stride
        self.activation = activation
        self.register_buffer("act_width", torch.tensor(act_width))
        self.hard_reset, self.detach = hard_reset, detach

    def _init_conv(self, conv, fan):
        w_scale = math.sqrt(1 / fan)
        nn.init.uniform_(conv.weight, -w_scale, w_scale)

    def forward(self, input_, prev_state, residual=0, slots=None):
        """-> (z_out + residual, stack([v_out, z_out(, trace)])), reference :96-126 etc."""
        return hip_ops.cell_forward(self, input_, prev_state, residual, slots)


class ConvLIF(_SpikingCell):
    """Convolutional spiking LIF cell (reference: spiking_submodules.py:24-126):
    arctan surrogate, hard reset, detached reset, per-channel sigmoid leak and
    learnable threshold."""

    kind = "lif"

    def __init__(self, input_size, hidden_size, kernel_size, stride=1, activation="arctanspike", act_width=10.0,
                 leak=(-4.0, 0.1), thresh=(0.8, 0.0), learn_leak=True, learn_thresh=True, hard_reset=True, detach=True,
                 norm=None):
        super().__init__()
        padding = kernel_size // 2
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, stride=stride, padding=padding, bias=False)
        _param_or_buffer(self, "leak", torch.randn(hidden_size, 1, 1) * leak[1] + leak[0], learn_leak)
        _param_or_buffer(self, "thresh", torch.randn(hidden_size, 1, 1) * thresh[1] + thresh[0], learn_thresh)
        self._init_conv(self.ff, input_size)
        self._common(input_size, hidden_size, kernel_size, stride, activation, act_width, hard_reset, detach, norm)


class ConvLIFRecurrent(_SpikingCell):
    """Convolutional recurrent spiking LIF cell (reference: spiking_submodules.py:438-551)."""

    kind = "lif"
    recurrent = True

    def __init__(self, input_size, hidden_size, kernel_size, activation="arctanspike", act_width=10.0, leak=(-4.0, 0.1),
                 thresh=(0.8, 0.0), learn_leak=True, learn_thresh=True, hard_reset=True, detach=True, norm=None):
        super().__init__()
        padding = kernel_size // 2
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, padding=padding, bias=False)
        self.rec = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=padding, bias=False)
        _param_or_buffer(self, "leak", torch.randn(hidden_size, 1, 1) * leak[1] + leak[0], learn_leak)
        _param_or_buffer(self, "thresh", torch.randn(hidden_size, 1, 1) * thresh[1] + thresh[0], learn_thresh)
        self._init_conv(self.ff, input_size)
        self._init_conv(self.rec, hidden_size)
        self._common(input_size, hidden_size, kernel_size, 1, activation, act_width, hard_reset, detach, norm)


class _PLIFBase(_SpikingCell):
    kind = "plif"
    num_states = 3

    def _plif_params(self, hidden_size, leak_v, leak_pt, add_pt, thresh, learn_leak, learn_thresh):
        _param_or_buffer(self, "leak_v", torch.randn(hidden_size, 1, 1) * leak_v[1] + leak_v[0], learn_leak)
        _param_or_buffer(self, "leak_pt", torch.randn(hidden_size, 1, 1) * leak_pt[1] + leak_pt[0], learn_leak)
        _param_or_buffer(self, "add_pt", torch.randn(hidden_size, 1, 1) * add_pt[1] + add_pt[0], learn_leak)
        _param_or_buffer(self, "thresh", torch.randn(hidden_size, 1, 1) * thresh[1] + thresh[0], learn_thresh)


class ConvPLIF(_PLIFBase):
    """LIF cell with adaptation through a pre-synaptic trace (reference: spiking_submodules.py:129-227)."""

    def __init__(self, input_size, hidden_size, kernel_size, stride=1, activation="arctanspike", act_width=10.0,
                 leak_v=(-4.0, 0.1), leak_pt=(-4.0, 0.1), add_pt=(-2.0, 0.1), thresh=(0.8, 0.0), learn_leak=True,
                 learn_thresh=True, hard_reset=True, detach=True, norm=None):
        super().__init__()
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, stride=stride, padding=kernel_size // 2, bias=False)
        self._plif_params(hidden_size, leak_v, leak_pt, add_pt, thresh, learn_leak, learn_thresh)
        self._init_conv(self.ff, input_size)
        self._common(input_size, hidden_size, kernel_size, stride, activation, act_width, hard_reset, detach, norm)


class ConvPLIFRecurrent(_PLIFBase):
    """Recurrent PLIF cell (reference: spiking_submodules.py:554-657)."""

    recurrent = True

    def __init__(self, input_size, hidden_size, kernel_size, activation="arctanspike", act_width=10.0,
                 leak_v=(-4.0, 0.1), leak_pt=(-4.0, 0.1), add_pt=(-2.0, 0.1), thresh=(0.8, 0.0), learn_leak=True,
                 learn_thresh=True, hard_reset=True, detach=True, norm=None):
        super().__init__()
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, padding=kernel_size // 2, bias=False)
        self.rec = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2, bias=False)
        self._plif_params(hidden_size, leak_v, leak_pt, add_pt, thresh, learn_leak, learn_thresh)
        self._init_conv(self.ff, input_size)
        self._init_conv(self.rec, hidden_size)
        self._common(input_size, hidden_size, kernel_size, 1, activation, act_width, hard_reset, detach, norm)


class _ALIFBase(_SpikingCell):
    kind = "alif"
    num_states = 3
    trace_name = "leak_t"

    def _alif_params(self, hidden_size, leak_v, leak_x, t0, t1, learn_leak, learn_thresh):
        _param_or_buffer(self, "leak_v", torch.randn(hidden_size, 1, 1) * leak_v[1] + leak_v[0], learn_leak)
        _param_or_buffer(self, self.trace_name, torch.randn(hidden_size, 1, 1) * leak_x[1] + leak_x[0], learn_leak)
        _param_or_buffer(self, "t0", torch.randn(hidden_size, 1, 1) * t0[1] + t0[0], learn_thresh)
        _param_or_buffer(self, "t1", torch.randn(hidden_size, 1, 1) * t1[1] + t1[0], learn_thresh)


class ConvALIF(_ALIFBase):
    """Adaptive-threshold LIF cell, soft reset by default (reference: spiking_submodules.py:230-334)."""

    def __init__(self, input_size, hidden_size, kernel_size, stride=1, activation="arctanspike", act_width=10.0,
                 leak_v=(-4.0, 0.1), leak_t=(-4.0, 0.1), t0=(0.01, 0.0), t1=(1.8, 0.0), learn_leak=True,
                 learn_thresh=False, hard_reset=False, detach=True, norm=None):
        super().__init__()
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, stride=stride, padding=kernel_size // 2, bias=False)
        self._alif_params(hidden_size, leak_v, leak_t, t0, t1, learn_leak, learn_thresh)
        self._init_conv(self.ff, input_size)
        self._common(input_size, hidden_size, kernel_size, stride, activation, act_width, hard_reset, detach, norm)


class ConvALIFRecurrent(_ALIFBase):
    """Recurrent ALIF cell (reference: spiking_submodules.py:660-768)."""

    recurrent = True

    def __init__(self, input_size, hidden_size, kernel_size, activation="arctanspike", act_width=10.0,
                 leak_v=(-4.0, 0.1), leak_t=(-4.0, 0.1), t0=(0.01, 0.0), t1=(1.8, 0.0), learn_leak=True,
                 learn_thresh=False, hard_reset=False, detach=True, norm=None):
        super().__init__()
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, padding=kernel_size // 2, bias=False)
        self.rec = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2, bias=False)
        self._alif_params(hidden_size, leak_v, leak_t, t0, t1, learn_leak, learn_thresh)
        self._init_conv(self.ff, input_size)
        self._init_conv(self.rec, hidden_size)
        self._common(input_size, hidden_size, kernel_size, 1, activation, act_width, hard_reset, detach, norm)


class ConvXLIF(_ALIFBase):
    """LIF cell whose threshold adapts to the pre-synaptic trace (reference: spiking_submodules.py:337-435)."""

    kind = "xlif"
    trace_name = "leak_pt"

    def __init__(self, input_size, hidden_size, kernel_size, stride=1, activation="arctanspike", act_width=10.0,
                 leak_v=(-4.0, 0.1), leak_pt=(-4.0, 0.1), t0=(0.01, 0.0), t1=(1.8, 0.0), learn_leak=True,
                 learn_thresh=False, hard_reset=False, detach=True, norm=None):
        super().__init__()
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, stride=stride, padding=kernel_size // 2, bias=False)
        self._alif_params(hidden_size, leak_v, leak_pt, t0, t1, learn_leak, learn_thresh)
        self._init_conv(self.ff, input_size)
        self._common(input_size, hidden_size, kernel_size, stride, activation, act_width, hard_reset, detach, norm)


class ConvXLIFRecurrent(_ALIFBase):
    """Recurrent XLIF cell (reference: spiking_submodules.py:771-875)."""

    kind = "xlif"
    recurrent = True
    trace_name = "leak_pt"

    def __init__(self, input_size, hidden_size, kernel_size, stride=1, activation="arctanspike", act_width=10.0,
                 leak_v=(-4.0, 0.1), leak_pt=(-4.0, 0.1), t0=(0.01, 0.0), t1=(1.8, 0.0), learn_leak=True,
                 learn_thresh=False, hard_reset=False, detach=True, norm=None):
        super().__init__()
        self.ff = nn.Conv2d(input_size, hidden_size, kernel_size, stride=stride, padding=kernel_size // 2, bias=False)
        self.rec = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2, bias=False)
        self._alif_params(hidden_size, leak_v, leak_pt, t0, t1, learn_leak, learn_thresh)
        self._init_conv(self.ff, input_size)
        self._init_conv(self.rec, hidden_size)
        self._common(input_size, hidden_size, kernel_size, stride, activation, act_width, hard_reset, detach, norm)


# ---------------------------------------------------------------------------
# blocks of the spiking EV-FlowNet (reference: spiking_submodules.py:878-1013)
# ---------------------------------------------------------------------------
_FF = {"lif": ConvLIF, "alif": ConvALIF, "plif": ConvPLIF, "xlif": ConvXLIF}
_REC = {"lif": ConvLIFRecurrent, "alif": ConvALIFRecurrent, "plif": ConvPLIFRecurrent, "xlif": ConvXLIFRecurrent}


stack_states = hip_ops.stack_states


class SpikingRecurrentConvLayer(nn.Module):
    """Spiking conv cell followed by a recurrent spiking conv cell (reference :878-929)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, recurrent_block_type="lif",
                 activation_ff="arctanspike", activation_rec="arctanspike", **kwargs):
        super().__init__()
        assert recurrent_block_type in ["lif", "alif", "plif", "xlif"]
        kwargs.pop("spiking_feedforward_block_type", None)
        self.conv = _FF[recurrent_block_type](in_channels, out_channels, kernel_size, stride, activation_ff, **kwargs)
        self.recurrent_block = _REC[recurrent_block_type](out_channels, out_channels, kernel_size, activation=activation_rec,
                                                          **kwargs)

    def forward(self, x, prev_state):
        if prev_state is None:
            prev_state = [None, None]
        ff, rec = prev_state
        slots = hip_ops.StateSlots(2)  # both new states in one buffer: the stacked state needs no copy
        x1, ff = self.conv(x, ff, slots=slots)
        x2, rec = self.recurrent_block(x1, rec, slots=slots)
        return x2, stack_states([ff, rec], slots)


class SpikingResidualBlock(nn.Module):
    """Spike-based residual block, Fang et al. 2021: the block input is added to the second
    cell's output spikes (reference :932-975)."""

    def __init__(self, in_channels, out_channels, stride=1, spiking_feedforward_block_type="lif", activation="arctanspike",
                 **kwargs):
        super().__init__()
        assert spiking_feedforward_block_type in ["lif", "alif", "plif", "xlif"]
        cls = _FF[spiking_feedforward_block_type]
        self.conv1 = cls(in_channels, out_channels, kernel_size=3, stride=stride, activation=activation, **kwargs)
        self.conv2 = cls(out_channels, out_channels, kernel_size=3, stride=1, activation=activation, **kwargs)

    def forward(self, x, prev_state):
        if prev_state is None:
            prev_state = [None, None]
        conv1, conv2 = prev_state
        slots = hip_ops.StateSlots(2)
        x1, conv1 = self.conv1(x, conv1, slots=slots)
        x2, conv2 = self.conv2(x1, conv2, residual=hip_ops.twin(x), slots=slots)  # (the block input's second consumer)
        return x2, stack_states([conv1, conv2], slots)


class SpikingUpsampleConvLayer(nn.Module):
    """Bilinear x2 up-sampling followed by a spiking conv cell (reference :978-1013)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, spiking_feedforward_block_type="lif",
                 activation="arctanspike", **kwargs):
        super().__init__()
        assert spiking_feedforward_block_type in ["lif", "alif", "plif", "xlif"]
        self.conv2d = _FF[spiking_feedforward_block_type](in_channels, out_channels, kernel_size, stride=stride,
                                                          activation=activation, **kwargs)

    def forward(self, x, prev_state):
        x_up = hip_ops.upsample2x_bilinear(x)
        return self.conv2d(x_up, prev_state)

    def forward_upsampled(self, x_up, prev_state):
        """The cell on an input that is up-sampled already (models/unet.py builds cat + bilinear x2 in one kernel)."""
        return self.conv2d(x_up, prev_state)


class SpikingTransposedConvLayer(nn.Module):
    """Reference :1016-1065 (use_upsample_conv=False).  Not on the accelerated path."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("transposed-conv decoders (use_upsample_conv=False) are not on the accelerated path")
