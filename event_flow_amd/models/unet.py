"""Multi-resolution UNets -- host-side mirror of reference models/unet.py:
BaseUNet (:28-145), UNetRecurrent (:148-221, E2VID), MultiResUNet (:224-311,
EVFlowNet), MultiResUNetRecurrent
(:314-415, ConvGRU / ConvRNN encoders), SpikingMultiResUNetRecurrent (:418-465)
and LeakyMultiResUNetRecurrent (:468-480): 4 strided encoders (each followed by
a recurrent block in the recurrent nets), 2 residual blocks, 4 bilinear
up-sampling decoders on cat(prediction, x, skip) and a tanh 1x1 prediction per
scale.  Same module tree and parameter names as the reference, so its
state_dicts load.  Every block runs in libevflow_hip.so through the general
path (models/hip_ops.py); torch.cat / pad / stack are the only torch calls
(memory movement)."""

import torch
import torch.nn as nn

from .model_util import skip_concat, skip_sum  # noqa: F401  (resolved by name, reference unet.py:77)
from .spiking_submodules import (
    SpikingRecurrentConvLayer,
    SpikingResidualBlock,
    SpikingTransposedConvLayer,
    SpikingUpsampleConvLayer,
)
from .submodules import (
    ConvLayer,
    LeakyRecurrentConvLayer,
    LeakyResidualBlock,
    LeakyTransposedConvLayer,
    LeakyUpsampleConvLayer,
    RecurrentConvLayer,
    ResidualBlock,
    TransposedConvLayer,
    UpsampleConvLayer,
)


class BaseUNet(nn.Module):
    """Channel plan and layer factories shared by the UNets (reference unet.py:28-145).  The plan for
    `num_encoders` = E, base width C, multiplier m:  encoder i maps C m^i -> C m^(i+1) at stride 2 (the first one
    reads `num_bins` channels in the multi-resolution nets), the residual blocks work at C m^E, decoder i maps the
    (concatenated or summed) skip pair back to the width of encoder E-1-i's input."""

    ff_type = ConvLayer
    res_type = ResidualBlock
    upsample_type = UpsampleConvLayer
    transpose_type = TransposedConvLayer
    rec_type = RecurrentConvLayer
    w_scale_pred = None

    def __init__(self, unet_kwargs):
        super().__init__()
        kw = dict(unet_kwargs)
        self.final_activation = kw.pop("final_activation", None)
        self._plan(**kw)

    def _plan(self, base_num_channels, num_encoders, num_residual_blocks, num_output_channels, skip_type, norm,
              use_upsample_conv, num_bins, recurrent_block_type=None, kernel_size=5, channel_multiplier=2,
              activations=("relu", None), spiking_feedforward_block_type=None, spiking_neuron=None):
        assert num_output_channels > 0
        for name, value in (("base_num_channels", base_num_channels), ("num_encoders", num_encoders),
                            ("num_residual_blocks", num_residual_blocks), ("num_output_channels", num_output_channels),
                            ("kernel_size", kernel_size), ("skip_type", skip_type), ("norm", norm), ("num_bins", num_bins),
                            ("recurrent_block_type", recurrent_block_type), ("channel_multiplier", channel_multiplier)):
            setattr(self, name, value)
        self.ff_act, self.rec_act = activations
        # keyword arguments every cell / block of a spiking or leaky net receives (reference :69-73)
        self.spiking_kwargs = dict(spiking_neuron) if type(spiking_neuron) is dict else {}
        if spiking_feedforward_block_type is not None:
            self.spiking_kwargs = {"spiking_feedforward_block_type": spiking_feedforward_block_type, **self.spiking_kwargs}
        self.skip_ftn = {"concat": skip_concat, "sum": skip_sum}[skip_type]
        self.UpsampleLayer = self.upsample_type if use_upsample_conv else self.transpose_type
        widths = [int(base_num_channels * channel_multiplier ** i) for i in range(num_encoders + 1)]
        self.encoder_input_sizes, self.encoder_output_sizes = widths[:-1], widths[1:]
        self.max_num_channels = widths[-1]

    def _stack(self, make, n):
        return nn.ModuleList(make(i) for i in range(n))

    def build_encoders(self, recurrent, first_reads_bins):
        """Stride-2 encoders: plain `ff_type` layers (unet.py:241-257) or `rec_type` layers with their recurrent
        block (:335-353; :175-190 for the UNet whose head conv comes first)."""
        def make(i):
            cin = self.num_bins if (first_reads_bins and i == 0) else self.encoder_input_sizes[i]
            common = dict(kernel_size=self.kernel_size, stride=2, norm=self.norm)
            if not recurrent:
                return self.ff_type(cin, self.encoder_output_sizes[i], activation=self.ff_act, **common, **self.spiking_kwargs)
            extra = self.spiking_kwargs if first_reads_bins else {}
            return self.rec_type(cin, self.encoder_output_sizes[i], recurrent_block_type=self.recurrent_block_type,
                                 activation_ff=self.ff_act, activation_rec=self.rec_act, **common, **extra)

        return self._stack(make, self.num_encoders)

    def build_resblocks(self):  # unet.py:110-122
        c = self.max_num_channels
        return self._stack(lambda _i: self.res_type(c, c, activation=self.ff_act, norm=self.norm, **self.spiking_kwargs),
                           self.num_residual_blocks)

    def build_decoders(self, with_predictions):
        """Up-sampling decoders, coarse to fine.  Their input is the skip pair (2x the channels for `concat`) plus, in the
        multi-resolution nets, the previous scale's prediction (unet.py:124-138, :371-388)."""
        def make(i):
            cin = self.encoder_output_sizes[-1 - i]
            cin = cin if self.skip_type == "sum" and not with_predictions else 2 * cin
            if with_predictions and i > 0:
                cin += self.num_output_channels
            layer = self.UpsampleLayer(cin, self.encoder_input_sizes[-1 - i], kernel_size=self.kernel_size,
                                       activation=self.ff_act, norm=self.norm, **self.spiking_kwargs)
            cell = getattr(layer, "conv2d", None)
            if with_predictions and i > 0 and hasattr(cell, "kind"):
                # cat(prediction, x, skip): the leading flow channels are the only inputs of a spiking decoder that are
                # not spike-valued (hip_ops.conv_wgrad keeps the rest on the bf16 matrix cores)
                cell.analog_input_channels = self.num_output_channels
            return layer

        return self._stack(make, self.num_encoders)

    def build_predictions(self):  # one 1x1 layer per scale, unet.py:259-266 / :355-369
        return self._stack(lambda i: self.ff_type(self.encoder_input_sizes[-1 - i], self.num_output_channels, 1,
                                                   activation=self.final_activation, norm=self.norm,
                                                   w_scale=self.w_scale_pred), self.num_encoders)

    def _decoder_parts(self, x, skip, prediction, decoder):
        """The parts of cat(prediction, cat(x, skip)) and the alignment padding behind them (see _decoder_input)."""
        from .model_util import _centred

        parts = [_centred(x, skip), skip]
        pad = 0
        if prediction is not None:
            parts = [_centred(prediction, skip)] + parts
            conv = getattr(decoder, "conv2d", None)
            if conv is not None and getattr(conv, "kind", "ann") in ("lif", "alif", "ann"):
                pad = (-sum(p.shape[1] for p in parts)) % 4
        return parts, pad

    def _decoder_input(self, x, skip, prediction, decoder):
        """cat(prediction, cat(x, skip)) (unet.py:303-306); with 2C+2 channels two zero channels keep the activation
        16-byte aligned for the conv kernels (the packed weight is zero there; cells with a pre-synaptic trace
        average over the true channels and are left unpadded).  Concatenating skips: ONE torch.cat of all parts."""
        from .model_util import _centred

        if self.skip_type != "concat":
            x = self.skip_ftn(x, skip)
            return x if prediction is None else self.skip_ftn(prediction, x)
        from . import hip_ops

        parts = [_centred(x, skip), skip]
        pad = 0
        if prediction is not None:
            parts = [_centred(prediction, skip)] + parts
            conv = getattr(decoder, "conv2d", None)  # (None: transposed-conv decoder, which takes its exact channel count)
            if conv is not None and getattr(conv, "kind", "ann") in ("lif", "alif", "ann"):
                pad = (-sum(p.shape[1] for p in parts)) % 4
        return hip_ops.concat_channels(parts, pad)

    def _decode(self, x, blocks, stateful, offset=0):
        """Decoders + per-scale predictions, coarse to fine (unet.py:298-311, :402-415, :455-465)."""
        predictions = []
        from . import hip_ops

        for i, (decoder, pred) in enumerate(zip(self.decoders, self.preds)):
            if stateful and self.skip_type == "concat" and hasattr(decoder, "forward_upsampled"):
                # concatenation and bilinear x2 in one kernel: the low-resolution cat is never written
                parts, pad = self._decoder_parts(x, blocks[-1 - i], predictions[-1] if i else None, decoder)
                x, self.states[offset + i] = decoder.forward_upsampled(hip_ops.concat_up2(parts, pad), self.states[offset + i])
                predictions.append(pred(hip_ops.twin(x)))  # (x feeds the prediction and the next decoder: the cell's twin output)
                continue
            x = self._decoder_input(x, blocks[-1 - i], predictions[-1] if i else None, decoder)
            if stateful:
                x, self.states[offset + i] = decoder(x, self.states[offset + i])
            else:
                x = decoder(x)
            predictions.append(pred(hip_ops.twin(x)))
        return predictions


class UNetRecurrent(BaseUNet):
    """E2VID's UNet: head conv, ConvLayer + ConvLSTM encoders, residual blocks, up-sampling decoders on x + skip,
    one prediction at full resolution (reference unet.py:148-221)."""

    def __init__(self, unet_kwargs):
        kw = dict(unet_kwargs)
        final_activation = kw.pop("final_activation", "none")
        nn.Module.__init__(self)
        self.final_activation = final_activation if hasattr(torch, final_activation) else None
        self._plan(**kw)
        self.head = ConvLayer(self.num_bins, self.base_num_channels, kernel_size=self.kernel_size, stride=1)
        self.encoders = self.build_encoders(recurrent=True, first_reads_bins=False)
        self.resblocks = self.build_resblocks()
        self.decoders = self.build_decoders(with_predictions=False)
        self.pred = self.ff_type(self.base_num_channels if self.skip_type == "sum" else 2 * self.base_num_channels,
                                 self.num_output_channels, 1, activation=None, norm=self.norm)
        self.num_states = self.num_encoders
        self.states = [None] * self.num_states

    def forward(self, x):
        """x [N,num_bins,H,W] -> [N,num_output_channels,H,W].  unet.py:192-221."""
        from . import hip_ops

        x = self.head(x)
        head = x
        blocks = []
        for i, encoder in enumerate(self.encoders):
            x, self.states[i] = encoder(x, self.states[i])
            blocks.append(x)
        for resblock in self.resblocks:
            x, _ = resblock(x)
        for i, decoder in enumerate(self.decoders):
            x = decoder(self.skip_ftn(x, blocks[-1 - i]))
        x = self.skip_ftn(x, head)
        # pred (1x1 ConvLayer, no activation) followed by the final activation = one conv + activation launch
        return hip_ops.conv_act(self.pred, x, self.pred.conv2d.weight, self.pred.conv2d.bias, 1, self.final_activation)


class MultiResUNet(BaseUNet):
    """Feed-forward multi-resolution UNet of EVFlowNet (reference unet.py:224-311); its prediction layers keep the
    default initialisation (:259-266)."""

    def __init__(self, unet_kwargs):
        super().__init__(unet_kwargs)
        self.encoders = self.build_encoders(recurrent=False, first_reads_bins=True)
        self.resblocks = self.build_resblocks()
        self.decoders = self.build_decoders(with_predictions=True)
        self.preds = self.build_predictions()

    def forward(self, x):
        blocks = []
        for encoder in self.encoders:
            x = encoder(x)
            blocks.append(x)
        for resblock in self.resblocks:
            x, _ = resblock(x)
        return self._decode(x, blocks, stateful=False)


class MultiResUNetRecurrent(BaseUNet):
    """Every encoder is a ConvLayer followed by a ConvGRU / ConvRecurrent block; stateless residual blocks and
    decoders (reference unet.py:314-415)."""

    def __init__(self, unet_kwargs):
        super().__init__(unet_kwargs)
        self.encoders = self.build_encoders(recurrent=True, first_reads_bins=True)
        self.resblocks = self.build_resblocks()
        self.decoders = self.build_decoders(with_predictions=True)
        self.preds = self.build_predictions()
        self.num_states = self.num_encoders
        self.states = [None] * self.num_states

    def _encode(self, x):
        from . import hip_ops

        blocks = []
        for i, encoder in enumerate(self.encoders):
            x, self.states[i] = encoder(x, self.states[i])
            blocks.append(hip_ops.twin(x))  # (the skip connection: the second consumer of the encoder's output)
        return x, blocks

    def forward(self, x):
        x, blocks = self._encode(x)
        for resblock in self.resblocks:
            x, _ = resblock(x)
        return self._decode(x, blocks, stateful=False)


class SpikingMultiResUNetRecurrent(MultiResUNetRecurrent):
    """Spiking cells everywhere, 2 * encoders + residual blocks + decoders states (reference unet.py:418-465)."""

    res_type = SpikingResidualBlock
    upsample_type = SpikingUpsampleConvLayer
    transpose_type = SpikingTransposedConvLayer
    rec_type = SpikingRecurrentConvLayer
    w_scale_pred = 0.01

    def __init__(self, unet_kwargs):
        super().__init__(unet_kwargs)
        self.num_states = self.num_encoders * 2 + self.num_residual_blocks
        self.states = [None] * self.num_states

    def forward(self, x):
        """x [N,num_bins,H,W] -> [N,2,H/8..H,W/8..W] x 4 (coarse to fine).  unet.py:437-465."""
        x, blocks = self._encode(x)
        offset = self.num_encoders
        for i, resblock in enumerate(self.resblocks):
            x, self.states[offset + i] = resblock(x, self.states[offset + i])
        return self._decode(x, blocks, stateful=True, offset=offset + self.num_residual_blocks)


class LeakyMultiResUNetRecurrent(SpikingMultiResUNetRecurrent):
    """The spiking UNet's topology with leaky non-spiking cells (reference unet.py:468-480)."""

    res_type = LeakyResidualBlock
    upsample_type = LeakyUpsampleConvLayer
    transpose_type = LeakyTransposedConvLayer
    rec_type = LeakyRecurrentConvLayer
