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
    """Sizes and builders shared by the multi-resolution UNets (reference unet.py:28-145)."""

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
        self._base_init(**kw)

    # reference BaseUNet.__init__, unet.py:40-91
    def _base_init(self, base_num_channels, num_encoders, num_residual_blocks, num_output_channels, skip_type, norm,
                   use_upsample_conv, num_bins, recurrent_block_type=None, kernel_size=5, channel_multiplier=2,
                   activations=("relu", None), spiking_feedforward_block_type=None, spiking_neuron=None):
        self.base_num_channels = base_num_channels
        self.num_encoders = num_encoders
        self.num_residual_blocks = num_residual_blocks
        self.num_output_channels = num_output_channels
        self.kernel_size = kernel_size
        self.skip_type = skip_type
        self.norm = norm
        self.num_bins = num_bins
        self.recurrent_block_type = recurrent_block_type
        self.channel_multiplier = channel_multiplier
        self.ff_act, self.rec_act = activations
        self.spiking_kwargs = {}
        if spiking_feedforward_block_type is not None:
            self.spiking_kwargs["spiking_feedforward_block_type"] = spiking_feedforward_block_type
        if type(spiking_neuron) is dict:
            self.spiking_kwargs.update(spiking_neuron)
        self.skip_ftn = {"concat": skip_concat, "sum": skip_sum}[skip_type]
        self.UpsampleLayer = self.upsample_type if use_upsample_conv else self.transpose_type
        assert self.num_output_channels > 0
        self.encoder_input_sizes = [int(base_num_channels * pow(channel_multiplier, i)) for i in range(num_encoders)]
        self.encoder_output_sizes = [int(base_num_channels * pow(channel_multiplier, i + 1)) for i in range(num_encoders)]
        self.max_num_channels = self.encoder_output_sizes[-1]

    def build_recurrent_encoders(self):  # unet.py:335-353
        encoders = nn.ModuleList()
        for i, (cin, cout) in enumerate(zip(self.encoder_input_sizes, self.encoder_output_sizes)):
            if i == 0:
                cin = self.num_bins
            encoders.append(self.rec_type(cin, cout, kernel_size=self.kernel_size, stride=2,
                                          recurrent_block_type=self.recurrent_block_type, activation_ff=self.ff_act,
                                          activation_rec=self.rec_act, norm=self.norm, **self.spiking_kwargs))
        return encoders

    def build_resblocks(self):  # unet.py:110-122
        blocks = nn.ModuleList()
        for _ in range(self.num_residual_blocks):
            blocks.append(self.res_type(self.max_num_channels, self.max_num_channels, activation=self.ff_act, norm=self.norm,
                                        **self.spiking_kwargs))
        return blocks

    def build_multires_prediction_decoders(self):  # unet.py:371-388
        decoders = nn.ModuleList()
        sizes = zip(reversed(self.encoder_output_sizes), reversed(self.encoder_input_sizes))
        for i, (cin, cout) in enumerate(sizes):
            pred_ch = 0 if i == 0 else self.num_output_channels
            decoders.append(self.UpsampleLayer(2 * cin + pred_ch, cout, kernel_size=self.kernel_size, activation=self.ff_act,
                                               norm=self.norm, **self.spiking_kwargs))
        return decoders

    def build_multires_prediction_layer(self):  # unet.py:355-369
        preds = nn.ModuleList()
        for cout in reversed(self.encoder_input_sizes):
            preds.append(self.ff_type(cout, self.num_output_channels, 1, activation=self.final_activation, norm=self.norm,
                                      w_scale=self.w_scale_pred))
        return preds

    def build_encoders(self):  # unet.py:241-257 (MultiResUNet)
        encoders = nn.ModuleList()
        for i, (cin, cout) in enumerate(zip(self.encoder_input_sizes, self.encoder_output_sizes)):
            if i == 0:
                cin = self.num_bins
            encoders.append(self.ff_type(cin, cout, kernel_size=self.kernel_size, stride=2, activation=self.ff_act,
                                         norm=self.norm, **self.spiking_kwargs))
        return encoders

    def _decoder_input(self, x, skip, prediction, decoder):
        """cat(prediction, cat(x, skip)) (unet.py:303-306); with 2C+2 channels two zero channels keep the activation
        16-byte aligned for the conv kernels (the packed weight is zero there; cells with a pre-synaptic trace
        average over the true channels and are left unpadded)."""
        x = self.skip_ftn(x, skip)
        if prediction is not None:
            x = self.skip_ftn(prediction, x)
            pad = (-x.shape[1]) % 4
            if pad and self.skip_type == "concat" and getattr(decoder.conv2d, "kind", "ann") in ("lif", "alif", "ann"):
                x = torch.cat([x, x.new_zeros((x.shape[0], pad) + tuple(x.shape[2:]))], 1)
        return x


class UNetRecurrent(BaseUNet):
    """E2VID's UNet: head conv, ConvLayer + ConvLSTM encoders, residual blocks, up-sampling decoders on x + skip,
    one prediction at full resolution (reference unet.py:148-221)."""

    def __init__(self, unet_kwargs):
        kw = dict(unet_kwargs)
        final_activation = kw.pop("final_activation", "none")
        nn.Module.__init__(self)
        self.final_activation = final_activation if hasattr(torch, final_activation) else None
        self._base_init(**kw)
        self.head = ConvLayer(self.num_bins, self.base_num_channels, kernel_size=self.kernel_size, stride=1)
        self.encoders = self.build_recurrent_encoders()
        self.resblocks = self.build_resblocks()
        self.decoders = self.build_decoders()
        self.pred = self.ff_type(self.base_num_channels if self.skip_type == "sum" else 2 * self.base_num_channels,
                                 self.num_output_channels, 1, activation=None, norm=self.norm)
        self.num_states = self.num_encoders
        self.states = [None] * self.num_states

    def build_recurrent_encoders(self):  # unet.py:175-190: every encoder input is a feature map (the head comes first)
        encoders = nn.ModuleList()
        for cin, cout in zip(self.encoder_input_sizes, self.encoder_output_sizes):
            encoders.append(self.rec_type(cin, cout, kernel_size=self.kernel_size, stride=2,
                                          recurrent_block_type=self.recurrent_block_type, activation_ff=self.ff_act,
                                          activation_rec=self.rec_act, norm=self.norm))
        return encoders

    def build_decoders(self):  # unet.py:124-138
        decoders = nn.ModuleList()
        for cin, cout in zip(reversed(self.encoder_output_sizes), reversed(self.encoder_input_sizes)):
            decoders.append(self.UpsampleLayer(cin if self.skip_type == "sum" else 2 * cin, cout, kernel_size=self.kernel_size,
                                               activation=self.ff_act, norm=self.norm, **self.spiking_kwargs))
        return decoders

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
            x = decoder(self.skip_ftn(x, blocks[self.num_encoders - i - 1]))
        x = self.skip_ftn(x, head)
        # pred (1x1 ConvLayer, no activation) followed by the final activation = one conv + activation launch
        return hip_ops.conv_act(self.pred, x, self.pred.conv2d.weight, self.pred.conv2d.bias, 1, self.final_activation)


class MultiResUNet(BaseUNet):
    """Feed-forward multi-resolution UNet of EVFlowNet (reference unet.py:224-311); its prediction layers keep the
    default initialisation (:259-266)."""

    def __init__(self, unet_kwargs):
        super().__init__(unet_kwargs)
        self.encoders = self.build_encoders()
        self.resblocks = self.build_resblocks()
        self.decoders = self.build_multires_prediction_decoders()
        self.preds = self.build_multires_prediction_layer()

    def forward(self, x):
        blocks = []
        for encoder in self.encoders:
            x = encoder(x)
            blocks.append(x)
        for resblock in self.resblocks:
            x, _ = resblock(x)
        predictions = []
        for i, (decoder, pred) in enumerate(zip(self.decoders, self.preds)):
            x = self._decoder_input(x, blocks[self.num_encoders - i - 1], predictions[-1] if i else None, decoder)
            x = decoder(x)
            predictions.append(pred(x))
        return predictions


class MultiResUNetRecurrent(BaseUNet):
    """Every encoder is a ConvLayer followed by a ConvGRU / ConvRecurrent block; stateless residual blocks and
    decoders (reference unet.py:314-415)."""

    def __init__(self, unet_kwargs):
        super().__init__(unet_kwargs)
        self.encoders = self.build_recurrent_encoders()
        self.resblocks = self.build_resblocks()
        self.decoders = self.build_multires_prediction_decoders()
        self.preds = self.build_multires_prediction_layer()
        self.num_states = self.num_encoders
        self.states = [None] * self.num_states

    def forward(self, x):
        blocks = []
        for i, encoder in enumerate(self.encoders):
            x, self.states[i] = encoder(x, self.states[i])
            blocks.append(x)
        for resblock in self.resblocks:
            x, _ = resblock(x)
        predictions = []
        for i, (decoder, pred) in enumerate(zip(self.decoders, self.preds)):
            x = self._decoder_input(x, blocks[self.num_encoders - i - 1], predictions[-1] if i else None, decoder)
            x = decoder(x)
            predictions.append(pred(x))
        return predictions


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
        blocks = []
        for i, encoder in enumerate(self.encoders):
            x, self.states[i] = encoder(x, self.states[i])
            blocks.append(x)
        offset = self.num_encoders
        for i, resblock in enumerate(self.resblocks):
            x, self.states[offset + i] = resblock(x, self.states[offset + i])
        predictions = []
        offset += self.num_residual_blocks
        for i, (decoder, pred) in enumerate(zip(self.decoders, self.preds)):
            x = self._decoder_input(x, blocks[self.num_encoders - i - 1], predictions[-1] if i else None, decoder)
            x, self.states[offset + i] = decoder(x, self.states[offset + i])
            predictions.append(pred(x))
        return predictions


class LeakyMultiResUNetRecurrent(SpikingMultiResUNetRecurrent):
    """The spiking UNet's topology with leaky non-spiking cells (reference unet.py:468-480)."""

    res_type = LeakyResidualBlock
    upsample_type = LeakyUpsampleConvLayer
    transpose_type = LeakyTransposedConvLayer
    rec_type = LeakyRecurrentConvLayer
