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

import os

import torch

from .. import _lib
from .base import BaseModel
from .engine import FireNetEngine


def _xlif_fused_ok(c):
    """XLIF and ALIF cells ride on the PLIF kernels (threshold t0 + t1 * trace instead of the trace in the current; ALIF: the trace
    driven by the cell's own previous spikes); their backward forms exist for the default neuron only (configs/train_SNN.yml: hard
    reset, arctan surrogate) -- other XLIF / ALIF cells: general path (EVF_XLIF_FUSED=0: always)."""
    import os

    return bool(c.hard_reset) and c.activation == "arctanspike" and os.environ.get("EVF_XLIF_FUSED", "1") != "0"
from . import hip_ops
from .model_util import CropParameters, copy_states
from .unet import (
    LeakyMultiResUNetRecurrent,
    MultiResUNet,
    MultiResUNetRecurrent,
    SpikingMultiResUNetRecurrent,
    UNetRecurrent,
)
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
from .submodules import ConvGRU, ConvLayer, ConvLayer_, ConvLeaky, ConvLeakyRecurrent, ConvRecurrent


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
        self._use_fused = None
        self._states = [None] * self.num_recurrent_units  # general path only
        self.reset_states()

    # -- engine ------------------------------------------------------------
    def _cells(self):
        return [self.head, self.G1, self.R1a, self.R1b, self.G2, self.R2a, self.R2b]

    def _fused(self):
        """True when the network runs on the fused 32->32 spike kernels of models/engine.py (LIF / PLIF
        FireNets at the reference's width); every other variant -- ANN FireNet (ConvLayer_/ConvGRU), ALIF/XLIF,
        other widths, residual -- is chained cell by cell through the general path (models/hip_ops.py)."""
        if self._use_fused is None:
            cells = self._cells()
            self._use_fused = (
                not self.residual
                and all(getattr(c, "kind", None) in ("lif", "plif", "xlif", "alif") and not getattr(c, "wnorm", False) and not getattr(c, "gnorm", False)
                        for c in cells)
                and all(c.kind not in ("xlif", "alif") or _xlif_fused_ok(c) for c in cells)
                and len({c.kind for c in cells}) == 1
                and all(c.hidden_size == 32 and c.kernel_size == 3 and c.stride == 1 for c in cells)
            )
            if not self._use_fused and os.environ.get("EVF_PATH_NOTICE", "1") != "0":
                # not silently: once per model, say which path serves it and why (both paths are HIP; there is no CPU path)
                import sys

                print(f"[event_flow_amd] {type(self).__name__}: general path (one fused conv + neuron kernel per cell, "
                      f"models/hip_ops.py) -- {self.compute_path[1]}; the recorded 32-channel window kernels (models/engine.py) "
                      "serve LIF / PLIF FireNets (and XLIF / ALIF ones with the hard reset and the arctan surrogate) with base_num_channels=32, "
                      "kernel_size=3, no residual / weight / group norm",
                      file=sys.stderr)
        return self._use_fused

    @property
    def compute_path(self):
        """("fused" | "general", reason): which HIP path serves this network (train.py, bench.py and the tests read it)."""
        cells = self._cells()
        kinds = {getattr(c, "kind", type(c).__name__) for c in cells}
        why = []
        if self.residual:
            why.append("residual connections")
        if any(getattr(c, "kind", None) not in ("lif", "plif", "xlif", "alif") for c in cells):
            why.append("cell kind(s) " + ", ".join(sorted(str(k) for k in kinds)))
        elif any(c.kind in ("xlif", "alif") and not _xlif_fused_ok(c) for c in cells):
            import os

            why.append("EVF_XLIF_FUSED=0" if os.environ.get("EVF_XLIF_FUSED", "1") == "0" else
                       "XLIF / ALIF cells with the soft reset or another surrogate than arctanspike (their fused kernels: hard reset, arctan)")
        elif len(kinds) > 1:
            why.append("mixed cell kinds")
        if any(getattr(c, "wnorm", False) or getattr(c, "gnorm", False) for c in cells):
            why.append("normalised weights / group norm")
        if any(getattr(c, "hidden_size", 32) != 32 or getattr(c, "kernel_size", 3) != 3 or getattr(c, "stride", 1) != 1 for c in cells):
            why.append("width / kernel size other than 32 / 3")
        return ("general", "; ".join(why)) if why else ("fused", "")

    def _eng(self):
        if self._engine is None:
            if not self._fused():
                raise NotImplementedError(f"{type(self).__name__} runs on the general path; it has no fused engine")
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

    def state_buffers(self):
        """Opaque handle on the tensors that currently hold the recurrent state (fused engine)."""
        return list(self._eng()._states)

    def set_state_buffers(self, handle):
        """Point the model at the tensors of `handle` (from state_buffers()) as its current recurrent state -- after
        a hipGraph replay the state lives where the captured step left it, which Python did not see."""
        self._eng()._states = list(handle)

    def final_states_into(self, handle):
        """The last pass of the NEXT window (announced by mark_last_pass()) writes its state straight
        into the tensors of `handle` (from state_buffers()) instead of fresh ones.  With a cycle of
        K >= 2 captured step graphs -- graph 0 starts from `handle`, graph k from the state graph k-1
        left, the last one ends in `handle` -- the state crosses replays without any copy."""
        self._eng()._final_target = handle

    def mark_last_pass(self):
        """The next forward pass is the last one of its window (see final_states_into)."""
        if self._engine is not None:
            self._engine._final_hint = True

    def defer_forward(self, on=True):
        """Fused path only: launch the window's hidden cells diagonal by diagonal (FireNetEngine.defer_forward)."""
        if self._fused():
            self._eng().defer_forward(on)

    def flush_forward(self):
        if self._engine is not None:
            self._engine.flush_forward()

    def defer_backward(self, on=True):
        """Fused path only: launch the window's backward cells diagonal by diagonal (FireNetEngine.defer_backward)."""
        if self._fused():
            self._eng().defer_backward(on)

    # -- state API (models/model.py:203-227) -------------------------------
    @property
    def states(self):
        if not self._fused():
            return copy_states(self._states)
        if self._engine is None:
            return [None] * self.num_recurrent_units
        return copy_states(self._engine.get_states())

    @states.setter
    def states(self, states):
        if not self._fused():
            self._states = states
        else:
            self._eng().set_states(states)

    def detach_states(self):
        if not self._fused():
            self._states = [s.detach() if torch.is_tensor(s) else s for s in self._states]
        elif self._engine is not None:
            self._engine.detach_states()

    def reset_states(self):
        self._states = [None] * self.num_recurrent_units
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
            x = hip_ops.norm_nonzero(x)

        if not self._fused():
            return self._forward_general(x, log)

        eng = self._eng()
        flow = eng.forward(x)

        if log:  # fraction of non-zero outputs per layer, models/model.py:268-284
            eng.flush_forward()  # (recorded cells of this pass have not run yet: torch reads below would see stale memory)
            names = ["0:input", "1:head", "2:G1", "3:R1a", "4:R1b", "5:G2", "6:R2a", "7:R2b", "8:pred"]
            acts = [x.detach().ne(0).float().mean().item()]
            for st in eng.get_states():
                acts.append(st[1].ne(0).float().mean().item())
            acts.append(flow.detach().ne(0).float().mean().item())
            activity = dict(zip(names, acts))
        else:
            activity = None
        return {"flow": [flow], "activity": activity}


def _activity(names, tensors):
    return {n: t.detach().ne(0).float().mean().item() for n, t in zip(names, tensors)}


def _firenet_forward_general(self, x, log):
    """models/model.py:253-286, one libevflow_hip.so cell step per layer."""
    st = self._states
    x1, st[0] = self.head(x, st[0])
    x2, st[1] = self.G1(x1, st[1])
    x3, st[2] = self.R1a(x2, st[2])
    x4, st[3] = self.R1b(x3, st[3], residual=x2 if self.residual else 0)
    x5, st[4] = self.G2(x4, st[4])
    x6, st[5] = self.R2a(x5, st[5])
    x7, st[6] = self.R2b(x6, st[6], residual=x5 if self.residual else 0)
    flow = self.pred(x7).contiguous()
    activity = None
    if log:
        names = ["0:input", "1:head", "2:G1", "3:R1a", "4:R1b", "5:G2", "6:R2a", "7:R2b", "8:pred"]
        activity = _activity(names, [x, x1, x2, x3, x4, x5, x6, x7, flow])
    return {"flow": [flow], "activity": activity}


FireNet._forward_general = _firenet_forward_general


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


class FireFlowNet(FireNet):
    """EV-FireFlowNet: no recurrency, ConvLayer_ everywhere (reference: models/model.py:398-409)."""

    head_neuron = ConvLayer_
    ff_neuron = ConvLayer_
    rec_neuron = ConvLayer_
    residual = False
    w_scale_pred = 0.01


class RNNFireNet(FireNet):
    """Recurrent FireNet of plain convolutional neurons (reference: models/model.py:614-622)."""

    head_neuron = ConvLayer_
    ff_neuron = ConvLayer_
    rec_neuron = ConvRecurrent
    residual = False


class LeakyFireNet(FireNet):
    """Recurrent FireNet of leaky / stateful convolutional neurons (reference: models/model.py:625-633)."""

    head_neuron = ConvLeaky
    ff_neuron = ConvLeaky
    rec_neuron = ConvLeakyRecurrent
    residual = False


class LeakyFireFlowNet(FireNet):
    """FireFlowNet with a leaky internal state (reference: models/model.py:696-704)."""

    head_neuron = ConvLeaky
    ff_neuron = ConvLeaky
    rec_neuron = ConvLeaky
    residual = False


class RecEVFlowNet(BaseModel):
    """Recurrent EV-FlowNet (Zhu et al., RSS 2018) with ConvGRU encoders -- reference: models/model.py:412-547.
    Every variant (ConvGRU / ConvRNN / leaky / spiking) runs cell by cell through the general path."""

    unet_type = MultiResUNetRecurrent
    recurrent_block_type = "convgru"
    spiking_feedforward_block_type = None

    def __init__(self, unet_kwargs):
        super().__init__()
        unet_kwargs = dict(unet_kwargs)  # the reference mutates the caller's dict (quirk q3); we do not
        norm = unet_kwargs.get("norm", None)
        use_upsample_conv = unet_kwargs.get("use_upsample_conv", True)
        net_kwargs = {
            "base_num_channels": unet_kwargs["base_num_channels"],
            "num_encoders": 4,
            "num_residual_blocks": 2,
            "num_output_channels": 2,
            "skip_type": "concat",
            "norm": norm,
            "use_upsample_conv": use_upsample_conv,
            "kernel_size": unet_kwargs["kernel_size"],
            "channel_multiplier": 2,
            "recurrent_block_type": self.recurrent_block_type,
            "final_activation": "tanh",
            "spiking_feedforward_block_type": self.spiking_feedforward_block_type,
            "spiking_neuron": unet_kwargs["spiking_neuron"],
        }
        self.crop = None
        self.mask = unet_kwargs["mask_output"]
        self.norm_input = False if "norm_input" not in unet_kwargs.keys() else unet_kwargs["norm_input"]
        self.encoding = unet_kwargs["encoding"]
        self.num_bins = unet_kwargs["num_bins"]
        self.num_encoders = net_kwargs["num_encoders"]
        unet_kwargs.update(net_kwargs)
        for k in ("name", "encoding", "round_encoding", "norm_input", "mask_output"):
            unet_kwargs.pop(k, None)
        self.multires_unetrec = self.unet_type(unet_kwargs)

    @property
    def states(self):
        return copy_states(self.multires_unetrec.states)

    @states.setter
    def states(self, states):
        self.multires_unetrec.states = states

    def detach_states(self):
        det = []
        for state in self.multires_unetrec.states:
            if type(state) is tuple:
                det.append(tuple(h.detach() for h in state))
            else:
                det.append(state.detach() if state is not None else None)
        self.multires_unetrec.states = det

    def reset_states(self):
        self.multires_unetrec.states = [None] * self.multires_unetrec.num_states

    def init_cropping(self, width, height, safety_margin=0):
        self.crop = CropParameters(width, height, self.num_encoders, safety_margin)

    def forward(self, event_voxel, event_cnt, log=False):
        """-> {"flow": [4 x [N,2,H,W]] (coarse to fine, all at input resolution), "activity": None}."""
        return _multires_forward(self, self.multires_unetrec, event_voxel, event_cnt, log)


def _multires_forward(self, unet, event_voxel, event_cnt, log):
    """Shared body of RecEVFlowNet.forward (models/model.py:476-547) and EVFlowNet.forward (:337-395)."""
    if self.encoding == "voxel":
        x = event_voxel
    elif self.encoding == "cnt" and self.num_bins == 2:
        x = event_cnt
    else:
        print("Model error: Incorrect input encoding.")
        raise AttributeError
    if self.norm_input:  # models/model.py:494-500
        x = hip_ops.norm_nonzero(x)
    if self.crop is not None:
        x = self.crop.pad(x)
    multires_flow = unet.forward(x)
    if log:
        raise NotImplementedError("Activity logging not implemented")  # reference :523-524
    flow_list = []
    for flow in multires_flow:
        fy = multires_flow[-1].shape[2] / flow.shape[2]
        fx = multires_flow[-1].shape[3] / flow.shape[3]
        if fy != fx:
            raise NotImplementedError("anisotropic flow pyramids")
        flow_list.append(hip_ops.upsample_nearest(flow.contiguous(), fy))
    if self.crop is not None:
        for i, flow in enumerate(flow_list):
            flow_list[i] = flow[:, :, self.crop.iy0 : self.crop.iy1, self.crop.ix0 : self.crop.ix1].contiguous()
    return {"flow": flow_list, "activity": None}


class E2VID(BaseModel):
    """E2VID (Rebecq et al., TPAMI 2021) adapted for optical flow: recurrent UNet with ConvLSTM encoders and one
    flow map at full resolution.  Reference: models/model.py:29-145."""

    def __init__(self, unet_kwargs):
        super().__init__()
        unet_kwargs = dict(unet_kwargs)  # the reference mutates the caller's dict (quirk q3); we do not
        net_kwargs = {
            "base_num_channels": unet_kwargs["base_num_channels"],
            "num_encoders": 3,
            "num_residual_blocks": 2,
            "num_output_channels": 2,
            "skip_type": "sum",
            "norm": unet_kwargs.get("norm", None),
            "use_upsample_conv": unet_kwargs.get("use_upsample_conv", True),
            "kernel_size": unet_kwargs["kernel_size"],
            "channel_multiplier": 2,
            "recurrent_block_type": "convlstm",
            "final_activation": "tanh",
        }
        self.crop = None
        self.mask = unet_kwargs["mask_output"]
        self.norm_input = False if "norm_input" not in unet_kwargs.keys() else unet_kwargs["norm_input"]
        self.encoding = unet_kwargs["encoding"]
        self.num_bins = unet_kwargs["num_bins"]
        self.num_encoders = net_kwargs["num_encoders"]
        unet_kwargs.update(net_kwargs)
        for k in ("name", "encoding", "round_encoding", "norm_input", "mask_output", "spiking_neuron"):
            unet_kwargs.pop(k, None)
        self.unetrecurrent = UNetRecurrent(unet_kwargs)

    @property
    def states(self):
        return copy_states(self.unetrecurrent.states)

    @states.setter
    def states(self, states):
        self.unetrecurrent.states = states

    def detach_states(self):
        det = []
        for state in self.unetrecurrent.states:
            if type(state) is tuple:
                det.append(tuple(h.detach() for h in state))
            else:
                det.append(state.detach() if state is not None else None)
        self.unetrecurrent.states = det

    def reset_states(self):
        self.unetrecurrent.states = [None] * self.unetrecurrent.num_states

    def init_cropping(self, width, height, safety_margin=0):
        self.crop = CropParameters(width, height, self.num_encoders, safety_margin)

    def forward(self, event_voxel, event_cnt, log=False):
        """-> {"flow": [[N,2,H,W]], "activity": None}  (reference :100-145)."""
        if self.encoding == "voxel":
            x = event_voxel
        elif self.encoding == "cnt" and self.num_bins == 2:
            x = event_cnt
        else:
            print("Model error: Incorrect input encoding.")
            raise AttributeError
        if self.norm_input:
            x = hip_ops.norm_nonzero(x)
        if self.crop is not None:
            x = self.crop.pad(x)
        flow = self.unetrecurrent.forward(x)
        if log:
            raise NotImplementedError("Activity logging not implemented")
        if self.crop is not None:
            flow = flow[:, :, self.crop.iy0 : self.crop.iy1, self.crop.ix0 : self.crop.ix1]
        return {"flow": [flow.contiguous()], "activity": None}


class EVFlowNet(BaseModel):
    """EV-FlowNet (Zhu et al., RSS 2018): feed-forward multi-resolution UNet, no state.
    Reference: models/model.py:289-395."""

    def __init__(self, unet_kwargs):
        super().__init__()
        unet_kwargs = dict(unet_kwargs)  # the reference mutates the caller's dict (quirk q3); we do not
        net_kwargs = {
            "base_num_channels": unet_kwargs["base_num_channels"],
            "num_encoders": 4,
            "num_residual_blocks": 2,
            "num_output_channels": 2,
            "skip_type": "concat",
            "norm": None,
            "use_upsample_conv": True,
            "kernel_size": unet_kwargs["kernel_size"],
            "channel_multiplier": 2,
            "final_activation": "tanh",
        }
        self.crop = None
        self.mask = unet_kwargs["mask_output"]
        self.norm_input = False if "norm_input" not in unet_kwargs.keys() else unet_kwargs["norm_input"]
        self.encoding = unet_kwargs["encoding"]
        self.num_bins = unet_kwargs["num_bins"]
        self.num_encoders = net_kwargs["num_encoders"]
        unet_kwargs.update(net_kwargs)
        for k in ("name", "eval", "encoding", "round_encoding", "mask_output", "norm_input", "spiking_neuron"):
            unet_kwargs.pop(k, None)
        self.multires_unet = MultiResUNet(unet_kwargs)

    def detach_states(self):
        pass

    def reset_states(self):
        pass

    def init_cropping(self, width, height, safety_margin=0):
        self.crop = CropParameters(width, height, self.num_encoders, safety_margin)

    def forward(self, event_voxel, event_cnt, log=False):
        return _multires_forward(self, self.multires_unet, event_voxel, event_cnt, log)


class RNNRecEVFlowNet(RecEVFlowNet):
    """Recurrent EV-FlowNet with ConvRecurrent encoders (reference: models/model.py:594-601)."""

    unet_type = MultiResUNetRecurrent
    recurrent_block_type = "convrnn"


class LeakyRecEVFlowNet(RecEVFlowNet):
    """Recurrent EV-FlowNet of leaky cells (reference: models/model.py:604-611)."""

    unet_type = LeakyMultiResUNetRecurrent
    recurrent_block_type = "convleaky"


class SpikingRecEVFlowNet(RecEVFlowNet):
    """LIF EV-FlowNet (reference: models/model.py:550-558; BASELINE config 4)."""

    unet_type = SpikingMultiResUNetRecurrent
    recurrent_block_type = "lif"
    spiking_feedforward_block_type = "lif"


class PLIFRecEVFlowNet(RecEVFlowNet):
    """Reference: models/model.py:561-569."""

    unet_type = SpikingMultiResUNetRecurrent
    recurrent_block_type = "plif"
    spiking_feedforward_block_type = "plif"


class ALIFRecEVFlowNet(RecEVFlowNet):
    """Reference: models/model.py:572-580."""

    unet_type = SpikingMultiResUNetRecurrent
    recurrent_block_type = "alif"
    spiking_feedforward_block_type = "alif"


class XLIFRecEVFlowNet(RecEVFlowNet):
    """Reference: models/model.py:583-591."""

    unet_type = SpikingMultiResUNetRecurrent
    recurrent_block_type = "xlif"
    spiking_feedforward_block_type = "xlif"


MODELS = {
    c.__name__: c
    for c in (FireNet, FireFlowNet, RNNFireNet, LeakyFireNet, LeakyFireFlowNet, LIFFireNet, PLIFFireNet, ALIFFireNet,
              XLIFFireNet, LIFFireFlowNet, E2VID, EVFlowNet, RecEVFlowNet, RNNRecEVFlowNet, LeakyRecEVFlowNet, SpikingRecEVFlowNet,
              PLIFRecEVFlowNet, ALIFRecEVFlowNet, XLIFRecEVFlowNet)
}
