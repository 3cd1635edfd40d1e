"""Model zoo -- host-side mirror of reference models/model.py for the models on
the accelerated path: the FireNet family (FireNet.forward, model.py:229-286)
with LIF / PLIF / ALIF / XLIF neurons and the spiking recurrent EV-FlowNets.

Same class names, constructor argument (`unet_kwargs` dict from the YAML
`model:` block, `spiking_neuron` merged in by configs/parser.py:117-127),
`forward(event_voxel, event_cnt, log=False) -> {"flow": [...], "activity": ...}`,
`reset_states()`, `detach_states()`, `.states`, `.mask`, `init_cropping()`, and
the same state_dict keys, so train_flow.py:81-83,130 / eval_flow.py:93-95,134
work unchanged.  All arithmetic runs in libevflow_hip.so through
models/engine.py; the model's tensors must be on the MI355X.
"""

import torch

from .. import _lib
from .base import BaseModel
from .engine import FireNetEngine
from .model_util import copy_states
from .spiking_submodules import (
    ConvALIF,
    ConvALIFRecurrent,
    ConvLIF,
    ConvLIFRecurrent,
    ConvPLIF,
    ConvPLIFRecurrent,
    ConvXLIF,
    ConvXLIFRecurrent,
)
from .submodules import ConvGRU, ConvLayer, ConvLayer_


class FireNet(BaseModel):
    """FireNet architecture (Scheerlinck et al., WACV 2020) adapted for optical
    flow: head -> G1 -> R1a -> R1b -> G2 -> R2a -> R2b -> pred.
    Reference: models/model.py:148-286."""

    head_neuron = ConvLayer_
    ff_neuron = ConvLayer_
    rec_neuron = ConvGRU
    residual = False
    num_recurrent_units = 7
    w_scale_pred = None
    precision = "bf16x3"  # matrix-core path of the 32->32 forward convs: "bf16x3" (exact split) or "fp32"

    def __init__(self, unet_kwargs):
        super().__init__()
        self.num_bins = unet_kwargs["num_bins"]
        base_num_channels = unet_kwargs["base_num_channels"]
        kernel_size = unet_kwargs["kernel_size"]
        self.encoding = unet_kwargs["encoding"]
        self.norm_input = False if "norm_input" not in unet_kwargs.keys() else unet_kwargs["norm_input"]
        self.mask = unet_kwargs["mask_output"]
        ff_act, rec_act = unet_kwargs["activations"]
        # per-instance kwargs (the reference shares one class-level dict, quirk q2)
        kwargs = dict(unet_kwargs["spiking_neuron"]) if type(unet_kwargs.get("spiking_neuron")) is dict else {}

        self.head = self.head_neuron(self.num_bins, base_num_channels, kernel_size, activation=ff_act, **kwargs)
        self.G1 = self.rec_neuron(base_num_channels, base_num_channels, kernel_size, activation=rec_act, **kwargs)
        self.R1a = self.ff_neuron(base_num_channels, base_num_channels, kernel_size, activation=ff_act, **kwargs)
        self.R1b = self.ff_neuron(base_num_channels, base_num_channels, kernel_size, activation=ff_act, **kwargs)
        self.G2 = self.rec_neuron(base_num_channels, base_num_channels, kernel_size, activation=rec_act, **kwargs)
        self.R2a = self.ff_neuron(base_num_channels, base_num_channels, kernel_size, activation=ff_act, **kwargs)
        self.R2b = self.ff_neuron(base_num_channels, base_num_channels, kernel_size, activation=ff_act, **kwargs)
        self.pred = ConvLayer(base_num_channels, out_channels=2, kernel_size=1, activation="tanh", w_scale=self.w_scale_pred)
        self._engine = None
        self.reset_states()

    # -- engine ------------------------------------------------------------
    def _cells(self):
        return [self.head, self.G1, self.R1a, self.R1b, self.G2, self.R2a, self.R2b]

    def _eng(self):
        if self._engine is None:
            if self.residual:
                raise NotImplementedError("residual FireNet variants are not part of the shipped configurations")
            if not hasattr(self.head, "kind"):
                raise NotImplementedError(
                    f"{type(self).__name__}: the ANN FireNet (ConvLayer_/ConvGRU, reference config 1 is a CPU plumbing "
                    "case) has no HIP path yet; only the spiking FireNets are accelerated"
                )
            self._engine = FireNetEngine(self._cells(), self.pred, self.num_bins, precision=self.precision)
        return self._engine

    def invalidate_weight_cache(self):
        """Call after parameters were rewritten outside torch (fused optimizer kernel)."""
        if self._engine is not None:
            self._engine._packed_key = None

    def use_static_states(self, flag=True):
        """Keep the recurrent state in persistent buffers across detach_states()
        (required when a whole training step is replayed from a hipGraph)."""
        self._eng().static_states = bool(flag)

    # -- state API (models/model.py:203-227) -------------------------------
    @property
    def states(self):
        if self._engine is None:
            return [None] * self.num_recurrent_units
        return copy_states(self._engine.get_states())

    @states.setter
    def states(self, states):
        self._eng().set_states(states)

    def detach_states(self):
        if self._engine is not None:
            self._engine.detach_states()

    def reset_states(self):
        if self._engine is not None:
            self._engine.reset_states()

    def init_cropping(self, width, height):
        pass

    def forward(self, event_voxel, event_cnt, log=False):
        """event_voxel [N,num_bins,H,W], event_cnt [N,2,H,W] ->
        {"flow": [[N,2,H,W]], "activity": dict or None}."""
        if self.encoding == "voxel":
            x = event_voxel
        elif self.encoding == "cnt" and self.num_bins == 2:
            x = event_cnt
        else:
            print("Model error: Incorrect input encoding.")
            raise AttributeError

        if self.norm_input:  # models/model.py:247-252, out of place (quirk q4)
            nz = x != 0
            vals = x[nz]
            x = x.clone()
            x[nz] = (vals - vals.mean()) / vals.std()

        eng = self._eng()
        flow = eng.forward(x)

        if log:  # fraction of non-zero outputs per layer, models/model.py:268-284
            names = ["0:input", "1:head", "2:G1", "3:R1a", "4:R1b", "5:G2", "6:R2a", "7:R2b", "8:pred"]
            acts = [x.detach().ne(0).float().mean().item()]
            for st in eng.get_states():
                acts.append(st[1].ne(0).float().mean().item())
            acts.append(flow.detach().ne(0).float().mean().item())
            activity = dict(zip(names, acts))
        else:
            activity = None
        return {"flow": [flow], "activity": activity}


class LIFFireNet(FireNet):
    """Spiking FireNet of LIF neurons (reference: models/model.py:636-645)."""

    head_neuron = ConvLIF
    ff_neuron = ConvLIF
    rec_neuron = ConvLIFRecurrent
    residual = False
    w_scale_pred = 0.01


class PLIFFireNet(FireNet):
    """Spiking FireNet of PLIF neurons (reference: models/model.py:648-657)."""

    head_neuron = ConvPLIF
    ff_neuron = ConvPLIF
    rec_neuron = ConvPLIFRecurrent
    residual = False
    w_scale_pred = 0.01


class ALIFFireNet(FireNet):
    """Spiking FireNet of ALIF neurons (reference: models/model.py:660-669)."""

    head_neuron = ConvALIF
    ff_neuron = ConvALIF
    rec_neuron = ConvALIFRecurrent
    residual = False
    w_scale_pred = 0.01


class XLIFFireNet(FireNet):
    """Spiking FireNet of XLIF neurons (reference: models/model.py:672-681)."""

    head_neuron = ConvXLIF
    ff_neuron = ConvXLIF
    rec_neuron = ConvXLIFRecurrent
    residual = False
    w_scale_pred = 0.01


class LIFFireFlowNet(FireNet):
    """Spiking FireFlowNet: no explicit recurrency (reference: models/model.py:684-693)."""

    head_neuron = ConvLIF
    ff_neuron = ConvLIF
    rec_neuron = ConvLIF
    residual = False
    w_scale_pred = 0.01


MODELS = {
    c.__name__: c for c in (FireNet, LIFFireNet, PLIFFireNet, ALIFFireNet, XLIFFireNet, LIFFireFlowNet)
}
