"""Oracle (PyTorch-CPU fp32) for the spiking / recurrent conv cells and the
FireNet / spiking EV-FlowNet stacks  --  test infrastructure only.

Restates, as pure functions over a `state_dict`-named parameter dictionary:
  models/spiking_util.py:13-109            (Heaviside + surrogate gradients)
  models/spiking_submodules.py:24-875      (LIF / PLIF / ALIF / XLIF cells, ff + recurrent)
  models/spiking_submodules.py:878-1013    (UNet spiking blocks)
  models/submodules.py:12-83,377-418       (ConvLayer, ConvLayer_, ConvGRU)
  models/model.py:148-286,412-558,636-693  (FireNet family, spiking RecEVFlowNet)
  models/unet.py:418-465                   (SpikingMultiResUNetRecurrent)
Parameter names equal the reference's state_dict keys, so reference weights
drop straight in.  float32 throughout; op order of the neuron update follows
the reference expression literally.
"""

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# spike function (models/spiking_util.py)
# ----------------------------------------------------------------------------
def _gauss(x, mu, sigma):
    return torch.exp(-((x - mu) * (x - mu)) / (2 * sigma * sigma)) / (sigma * math.sqrt(2 * math.pi))


def surrogate(kind, x, width):
    """d spike / d x for x = v - thresh.  spiking_util.py:38-43,55-65,74-79,88-93."""
    if kind == "arctanspike":
        return 1 / (1 + width * x * x)
    if kind == "superspike":
        return 1 / (1 + width * x.abs()) ** 2
    if kind == "trianglespike":
        return torch.relu(1 - width * x.abs())
    if kind == "mgspike":
        return 1.15 * _gauss(x, 0.0, width) - 0.15 * _gauss(x, width, 6 * width) - 0.15 * _gauss(x, -width, 6 * width)
    raise AttributeError(kind)


class _Spike(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, width, kind):
        ctx.save_for_backward(x)
        ctx.width, ctx.kind = width, kind
        return x.gt(0).float()  # always float32: spiking_util.py:21

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * surrogate(ctx.kind, x, ctx.width), None, None


def spike(v, thresh, width, kind):
    return _Spike.apply(v - thresh, float(width), kind)


# ----------------------------------------------------------------------------
# neuron cells.  `p` = parameter dict, `pre` = key prefix ("head.", "G1.", ...)
# state: tuple of tensors or None.  Return (out, new_state_tuple).
# ----------------------------------------------------------------------------
def _conv(x, w, stride=1, bias=None):
    return F.conv2d(x, w, bias, stride=stride, padding=w.shape[-1] // 2)


def _pretrace(x, k, stride):
    # mean over channels of |input|, 3x3 average pool (zero padded, count_include_pad)
    return F.avg_pool2d(x.abs().mean(1, keepdim=True), k, stride, padding=k // 2)


def conv_weight(p, name):
    """A conv layer's weight: the plain parameter, or -- norm="weight" cells (spiking_submodules.py:87-88, :502-504:
    nn.utils.weight_norm) -- g * v / ||v|| with the norm over everything but the output channel."""
    if name + ".weight_g" in p:
        v, g = p[name + ".weight_v"], p[name + ".weight_g"]
        return torch._weight_norm(v, g, 0)  # (what nn.utils.weight_norm's hook evaluates: torch/nn/utils/weight_norm.py)
    return p[name + ".weight"]


def cell_step(kind, p, pre, x, state, *, recurrent, stride=1, act="arctanspike", hard_reset=None, detach=True, residual=0):
    """One step of Conv{LIF,PLIF,ALIF,XLIF}[Recurrent].
    spiking_submodules.py:96-126 (LIF), :191-227 (PLIF), :299-334 (ALIF),
    :399-435 (XLIF), :516-551, :618-657, :730-768, :836-875 (recurrent)."""
    if hard_reset is None:
        hard_reset = kind in ("lif", "plif")  # ctor defaults, :51,:151 vs :260,:359
    width = p[pre + "act_width"]
    wff = conv_weight(p, pre + "ff")
    # norm="group" LIF cells (spiking_submodules.py:90-99, :507-529): nn.GroupNorm(1, C) on the input, and -- recurrent cell --
    # on the previous spikes, whose NORMALISED values then also enter the reset
    gn_in = pre + ("norm_ff" if recurrent else "norm") + ".weight"
    if kind == "lif" and gn_in in p:
        x = torch.nn.functional.group_norm(x, 1, p[gn_in], p[gn_in[:-6] + "bias"], 1e-5)
    ff = _conv(x, wff, stride)
    nstate = 2 if kind == "lif" else 3
    if state is None:
        state = tuple(torch.zeros_like(ff) for _ in range(nstate))
    v, z = state[0], state[1]
    if kind == "lif" and recurrent and pre + "norm_rec.weight" in p:
        z = torch.nn.functional.group_norm(z, 1, p[pre + "norm_rec.weight"], p[pre + "norm_rec.bias"], 1e-5)
    cur = ff
    if recurrent:
        cur = ff + _conv(z, conv_weight(p, pre + "rec"))  # z NOT detached here (:530)
    k = wff.shape[-1]

    if kind == "lif":
        thresh = p[pre + "thresh"].clamp_min(0.01)
        leak = torch.sigmoid(p[pre + "leak"])
        zr = z.detach() if detach else z
        if hard_reset:
            v_out = v * leak * (1 - zr) + (1 - leak) * cur
        else:
            v_out = v * leak + (1 - leak) * cur - zr * thresh
        z_out = spike(v_out, thresh, width, act)
        new = (v_out, z_out)
    elif kind == "plif":
        thresh = p[pre + "thresh"].clamp_min(0.01)
        leak_v = torch.sigmoid(p[pre + "leak_v"])
        leak_pt = torch.sigmoid(p[pre + "leak_pt"])
        add_pt = torch.sigmoid(p[pre + "add_pt"])
        pt = state[2]
        pt_out = pt * leak_pt + (1 - leak_pt) * _pretrace(x, k, stride)
        zr = z.detach() if detach else z
        if hard_reset:
            v_out = v * leak_v * (1 - zr) + (1 - leak_v) * (cur - add_pt * pt_out)
        else:
            v_out = v * leak_v + (1 - leak_v) * (cur - add_pt * pt_out) - zr * thresh
        z_out = spike(v_out, thresh, width, act)
        new = (v_out, z_out, pt_out)
    elif kind == "alif":
        t0 = p[pre + "t0"].clamp_min(0.01)
        t1 = p[pre + "t1"].clamp_min(0)
        leak_v = torch.sigmoid(p[pre + "leak_v"])
        leak_t = torch.sigmoid(p[pre + "leak_t"])
        t = state[2]
        t_out = t * leak_t + (1 - leak_t) * z  # z not detached here (:317)
        thresh = t0 + t1 * t_out
        zr = z.detach() if detach else z
        if hard_reset:
            v_out = v * leak_v * (1 - zr) + (1 - leak_v) * cur
        else:
            v_out = v * leak_v + (1 - leak_v) * cur - zr * (t0 + t1 * t)
        z_out = spike(v_out, thresh, width, act)
        new = (v_out, z_out, t_out)
    elif kind == "xlif":
        t0 = p[pre + "t0"].clamp_min(0.01)
        t1 = p[pre + "t1"].clamp_min(0)
        leak_v = torch.sigmoid(p[pre + "leak_v"])
        leak_pt = torch.sigmoid(p[pre + "leak_pt"])
        pt = state[2]
        pt_out = pt * leak_pt + (1 - leak_pt) * _pretrace(x, k, stride)
        thresh = t0 + t1 * pt_out
        zr = z.detach() if detach else z
        if hard_reset:
            v_out = v * leak_v * (1 - zr) + (1 - leak_v) * cur
        else:
            v_out = v * leak_v + (1 - leak_v) * cur - zr * (t0 + t1 * pt)
        z_out = spike(v_out, thresh, width, act)
        new = (v_out, z_out, pt_out)
    else:
        raise AttributeError(kind)
    return z_out + residual, new


def conv_gru_step(p, pre, x, h):
    """ConvGRU.forward, models/submodules.py:400-418."""
    if h is None:
        h = torch.zeros(x.shape[0], p[pre + "reset_gate.weight"].shape[0], *x.shape[2:], dtype=x.dtype)
    s = torch.cat([x, h], 1)
    u = torch.sigmoid(_conv(s, p[pre + "update_gate.weight"], bias=p[pre + "update_gate.bias"]))
    r = torch.sigmoid(_conv(s, p[pre + "reset_gate.weight"], bias=p[pre + "reset_gate.bias"]))
    o = torch.tanh(_conv(torch.cat([x, h * r], 1), p[pre + "out_gate.weight"], bias=p[pre + "out_gate.bias"]))
    n = h * (1 - u) + o * u
    return n, n


def conv_relu_step(p, pre, x, residual=0):
    """ConvLayer_.forward with activation relu, models/submodules.py:69-83."""
    out = _conv(x, p[pre + "conv2d.weight"], bias=p[pre + "conv2d.bias"]) + residual
    return torch.relu(out)


def conv_act_step(p, pre, x, act, residual=0):
    """ConvLayer_.forward, models/submodules.py:69-83 (activation by name or None)."""
    out = _conv(x, p[pre + "conv2d.weight"], bias=p[pre + "conv2d.bias"]) + residual
    return getattr(torch, act)(out) if act is not None else out


def conv_rnn_step(p, pre, x, state):
    """ConvRecurrent.forward, models/submodules.py:437-451."""
    if state is None:
        state = torch.zeros(x.shape[0], p[pre + "ff.weight"].shape[0], *x.shape[2:], dtype=x.dtype)
    ff = _conv(x, p[pre + "ff.weight"], bias=p[pre + "ff.bias"])
    rec = _conv(state, p[pre + "rec.weight"], bias=p[pre + "rec.bias"])
    state = torch.tanh(ff + rec)
    return torch.relu(_conv(state, p[pre + "out.weight"], bias=p[pre + "out.bias"])), state


def conv_leaky_rec_step(p, pre, x, state):
    """ConvLeakyRecurrent.forward, models/submodules.py:485-499."""
    ff = _conv(x, p[pre + "ff.weight"], bias=p[pre + "ff.bias"])
    if state is None:
        state = torch.zeros_like(ff)
    rec = _conv(state, p[pre + "rec.weight"], bias=p[pre + "rec.bias"])
    leak = torch.sigmoid(p[pre + "leak"])
    state = torch.tanh(state * leak + (1 - leak) * (ff + rec))
    return torch.relu(_conv(state, p[pre + "out.weight"], bias=p[pre + "out.bias"])), state


def conv_leaky_step(p, pre, x, state, act, residual=0, stride=1):
    """ConvLeaky.forward, models/submodules.py:538-554."""
    ff = _conv(x, p[pre + "ff.weight"], stride=stride, bias=p[pre + "ff.bias"])
    if state is None:
        state = torch.zeros_like(ff)
    leak = torch.sigmoid(p[pre + "leak"])
    state = state * leak + (1 - leak) * (ff + residual)
    return (getattr(torch, act)(state) if act is not None else state), state


def pred_layer(p, pre, x):
    """1x1 ConvLayer + tanh, models/submodules.py:52-61 via model.py:197-199."""
    return torch.tanh(F.conv2d(x, p[pre + "conv2d.weight"], p[pre + "conv2d.bias"]))


# ----------------------------------------------------------------------------
# FireNet family (models/model.py:148-286, 636-693)
# ----------------------------------------------------------------------------
FIRENET_LAYERS = ["head", "G1", "R1a", "R1b", "G2", "R2a", "R2b"]

# model name -> (neuron kind, G layers recurrent?)
FIRENET_KINDS = {
    "LIFFireNet": ("lif", True),
    "PLIFFireNet": ("plif", True),
    "ALIFFireNet": ("alif", True),
    "XLIFFireNet": ("xlif", True),
    "LIFFireFlowNet": ("lif", False),
}


# ANN comparison FireNets: model name -> (cell of head/R layers, cell of the G layers)
ANN_FIRENETS = {
    "FireFlowNet": ("conv", "conv"),
    "RNNFireNet": ("conv", "rnn"),
    "LeakyFireNet": ("leaky", "leaky_rec"),
    "LeakyFireFlowNet": ("leaky", "leaky"),
}


def firenet_forward(name, p, x, states, *, acts=("arctanspike", "arctanspike"), hard_reset=None, collect=None):
    """One pass head->G1->R1a->R1b->G2->R2a->R2b->pred (model.py:255-265).
    `states`: list of 7 (tuples or None).  Returns (flow [B,2,H,W], new_states).
    `collect`, if a dict, receives every layer's (out, state)."""
    new_states = []
    if name == "FireNet":  # ANN: ConvLayer_ / ConvGRU
        h = conv_relu_step(p, "head.", x)
        new_states.append(None)
        for li, lname in enumerate(FIRENET_LAYERS[1:], start=1):
            if lname.startswith("G"):
                h, st = conv_gru_step(p, lname + ".", h, states[li])
            else:
                h, st = conv_relu_step(p, lname + ".", h), None
            new_states.append(st)
            if collect is not None:
                collect[lname] = (h, st)
        return pred_layer(p, "pred.", h), new_states
    if name in ANN_FIRENETS:  # models/model.py:398-409, 614-633, 696-704
        ff_cell, g_cell = ANN_FIRENETS[name]
        ff_act, rec_act = acts
        h = x
        for li, lname in enumerate(FIRENET_LAYERS):
            is_g = lname.startswith("G")
            cell, act = (g_cell, rec_act) if is_g else (ff_cell, ff_act)
            if cell == "conv":
                h, st = conv_act_step(p, lname + ".", h, act), None
            elif cell == "leaky":
                h, st = conv_leaky_step(p, lname + ".", h, states[li], act)
            elif cell == "rnn":
                h, st = conv_rnn_step(p, lname + ".", h, states[li])
            else:
                h, st = conv_leaky_rec_step(p, lname + ".", h, states[li])
            new_states.append(st)
            if collect is not None:
                collect[lname] = (h, st)
        return pred_layer(p, "pred.", h), new_states
    kind, g_rec = FIRENET_KINDS[name]
    ff_act, rec_act = acts
    h = x
    for li, lname in enumerate(FIRENET_LAYERS):
        rec = g_rec and lname.startswith("G")
        h, st = cell_step(
            kind, p, lname + ".", h, states[li], recurrent=rec, act=(rec_act if lname.startswith("G") else ff_act),
            hard_reset=hard_reset,
        )
        new_states.append(st)
        if collect is not None:
            collect[lname] = (h, st)
    return pred_layer(p, "pred.", h), new_states


# ----------------------------------------------------------------------------
# Spiking recurrent EV-FlowNet (models/model.py:412-558, models/unet.py:418-465)
# ----------------------------------------------------------------------------
def skip_concat(x1, x2):
    """models/model_util.py:14-19: zero-pad x1 to x2's size, concat."""
    dh, dw = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))
    return torch.cat([x1, x2], 1)


def spiking_unet_forward(kind, p, x, states, *, num_encoders=4, num_res=2, acts=("arctanspike", "arctanspike"), hard_reset=None):
    """SpikingMultiResUNetRecurrent.forward (unet.py:437-465) + nearest
    upsampling of every scale to full resolution (model.py:528-539).
    states: list of 2*E + R entries: encoders hold (ff_state, rec_state),
    resblocks (conv1_state, conv2_state), decoders a single cell state."""
    pre = "multires_unetrec."
    ff_act, rec_act = acts
    new_states = [None] * len(states)
    blocks = []
    for i in range(num_encoders):
        st = states[i] if states[i] is not None else (None, None)
        x, s_ff = cell_step(kind, p, f"{pre}encoders.{i}.conv.", x, st[0], recurrent=False, stride=2, act=ff_act, hard_reset=hard_reset)
        x, s_rec = cell_step(kind, p, f"{pre}encoders.{i}.recurrent_block.", x, st[1], recurrent=True, act=rec_act, hard_reset=hard_reset)
        new_states[i] = (s_ff, s_rec)
        blocks.append(x)
    off = num_encoders
    for i in range(num_res):
        st = states[off + i] if states[off + i] is not None else (None, None)
        res = x
        x1, s1 = cell_step(kind, p, f"{pre}resblocks.{i}.conv1.", x, st[0], recurrent=False, act=ff_act, hard_reset=hard_reset)
        x, s2 = cell_step(kind, p, f"{pre}resblocks.{i}.conv2.", x1, st[1], recurrent=False, act=ff_act, hard_reset=hard_reset, residual=res)
        new_states[off + i] = (s1, s2)
    off += num_res
    preds = []
    for i in range(num_encoders):
        x = skip_concat(x, blocks[num_encoders - i - 1])
        if i > 0:
            x = skip_concat(preds[-1], x)
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        x, st = cell_step(kind, p, f"{pre}decoders.{i}.conv2d.", x, states[off + i], recurrent=False, act=ff_act, hard_reset=hard_reset)
        new_states[off + i] = st
        preds.append(pred_layer(p, f"{pre}preds.{i}.", x))
    flows = []
    for f in preds:
        sf = (preds[-1].shape[2] / f.shape[2], preds[-1].shape[3] / f.shape[3])
        flows.append(F.interpolate(f, scale_factor=sf))  # default mode = nearest
    return flows, new_states


def _act(name, v):
    return getattr(torch, name)(v) if name is not None else v


def ann_unet_forward(name, p, x, states, *, num_encoders=4, num_res=2, acts=("relu", None)):
    """Non-spiking EV-FlowNets: EVFlowNet (MultiResUNet.forward, unet.py:287-311), RecEVFlowNet / RNNRecEVFlowNet
    (MultiResUNetRecurrent.forward, unet.py:390-415, ConvGRU / ConvRecurrent encoders) and LeakyRecEVFlowNet
    (LeakyMultiResUNetRecurrent, unet.py:468-480 over the spiking forward :437-465), followed by the nearest
    up-sampling of every scale (model.py:528-539).  states: [] (EVFlowNet), E tensors (recurrent nets) or
    2E + R + E entries (leaky: (ff, rec) per encoder, (conv1, conv2) per residual block, one per decoder)."""
    leaky = name == "LeakyRecEVFlowNet"
    pre = "multires_unet." if name == "EVFlowNet" else "multires_unetrec."
    ff_act, rec_act = acts
    new_states = [None] * len(states)
    blocks = []
    for i in range(num_encoders):
        e = f"{pre}encoders.{i}."
        if name == "EVFlowNet":  # ConvLayer, submodules.py:52-61
            x = _act(ff_act, _conv(x, p[e + "conv2d.weight"], stride=2, bias=p[e + "conv2d.bias"]))
        elif leaky:  # LeakyRecurrentConvLayer.forward, submodules.py:679-686
            st = states[i] if states[i] is not None else (None, None)
            x, s_ff = conv_leaky_step(p, e + "conv.", x, st[0], ff_act, stride=2)
            x, s_rec = conv_leaky_rec_step(p, e + "recurrent_block.", x, st[1])
            new_states[i] = (s_ff, s_rec)
        else:  # RecurrentConvLayer.forward, submodules.py:229-235
            x = _act(ff_act, _conv(x, p[e + "conv.conv2d.weight"], stride=2, bias=p[e + "conv.conv2d.bias"]))
            if name == "RecEVFlowNet":
                x, new_states[i] = conv_gru_step(p, e + "recurrent_block.", x, states[i])
            else:
                x, new_states[i] = conv_rnn_step(p, e + "recurrent_block.", x, states[i])
        blocks.append(x)
    off = num_encoders
    for i in range(num_res):
        r = f"{pre}resblocks.{i}."
        if leaky:  # LeakyResidualBlock.forward, submodules.py:583-592
            st = states[off + i] if states[off + i] is not None else (None, None)
            x1, s1 = conv_leaky_step(p, r + "conv1.", x, st[0], ff_act)
            x, s2 = conv_leaky_step(p, r + "conv2.", x1, st[1], ff_act, residual=x)
            new_states[off + i] = (s1, s2)
        else:  # ResidualBlock.forward, submodules.py:290-311
            o1 = _act(ff_act, _conv(x, p[r + "conv1.weight"], bias=p[r + "conv1.bias"]))
            x = _act(ff_act, _conv(o1, p[r + "conv2.weight"], bias=p[r + "conv2.bias"]) + x)
    off += num_res
    preds = []
    for i in range(num_encoders):
        d = f"{pre}decoders.{i}."
        x = skip_concat(x, blocks[num_encoders - i - 1])
        if i > 0:
            x = skip_concat(preds[-1], x)
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        if leaky:  # LeakyUpsampleConvLayer.forward, submodules.py:619-623
            x, new_states[off + i] = conv_leaky_step(p, d + "conv2d.", x, states[off + i], ff_act)
        else:  # UpsampleConvLayer.forward, submodules.py:175-185
            x = _act(ff_act, _conv(x, p[d + "conv2d.weight"], bias=p[d + "conv2d.bias"]))
        preds.append(pred_layer(p, f"{pre}preds.{i}.", x))
    flows = []
    for f in preds:
        sf = (preds[-1].shape[2] / f.shape[2], preds[-1].shape[3] / f.shape[3])
        flows.append(F.interpolate(f, scale_factor=sf))  # default mode = nearest
    return flows, new_states


def conv_lstm_step(p, pre, x, state):
    """ConvLSTM.forward, models/submodules.py:335-374."""
    ch = p[pre + "Gates.weight"].shape[0] // 4
    if state is None:
        z = torch.zeros(x.shape[0], ch, *x.shape[2:], dtype=x.dtype)
        state = (z, z)
    prev_hidden, prev_cell = state
    gates = _conv(torch.cat((x, prev_hidden), 1), p[pre + "Gates.weight"], bias=p[pre + "Gates.bias"])
    in_gate, remember_gate, out_gate, cell_gate = gates.chunk(4, 1)
    in_gate, remember_gate, out_gate = torch.sigmoid(in_gate), torch.sigmoid(remember_gate), torch.sigmoid(out_gate)
    cell_gate = torch.tanh(cell_gate)
    cell = remember_gate * prev_cell + in_gate * cell_gate
    hidden = out_gate * torch.tanh(cell)
    return hidden, cell


def skip_sum(x1, x2):
    """models/model_util.py:22-27."""
    dh, dw = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2))
    return x1 + x2


def e2vid_forward(p, x, states, *, num_encoders=3, num_res=2, acts=("relu", None)):
    """E2VID: UNetRecurrent.forward (unet.py:192-221) with ConvLSTM encoders (RecurrentConvLayer.forward,
    submodules.py:229-235: the state is (hidden, cell)), skip 'sum', final tanh (model.py:45-56)."""
    pre = "unetrecurrent."
    ff_act, _ = acts
    x = _act("relu", _conv(x, p[pre + "head.conv2d.weight"], bias=p[pre + "head.conv2d.bias"]))  # ConvLayer default relu
    head = x
    new_states, blocks = [], []
    for i in range(num_encoders):
        e = f"{pre}encoders.{i}."
        x = _act(ff_act, _conv(x, p[e + "conv.conv2d.weight"], stride=2, bias=p[e + "conv.conv2d.bias"]))
        x, cell = conv_lstm_step(p, e + "recurrent_block.", x, states[i])
        new_states.append((x, cell))
        blocks.append(x)
    for i in range(num_res):
        r = f"{pre}resblocks.{i}."
        o1 = _act(ff_act, _conv(x, p[r + "conv1.weight"], bias=p[r + "conv1.bias"]))
        x = _act(ff_act, _conv(o1, p[r + "conv2.weight"], bias=p[r + "conv2.bias"]) + x)
    for i in range(num_encoders):
        d = f"{pre}decoders.{i}."
        x = skip_sum(x, blocks[num_encoders - i - 1])
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        x = _act(ff_act, _conv(x, p[d + "conv2d.weight"], bias=p[d + "conv2d.bias"]))
    x = skip_sum(x, head)
    return torch.tanh(F.conv2d(x, p[pre + "pred.conv2d.weight"], p[pre + "pred.conv2d.bias"])), new_states


def detach_states(states):
    def d(s):
        if s is None:
            return None
        if isinstance(s, tuple):
            return tuple(d(t) for t in s)
        return s.detach()

    return [d(s) for s in states]


# ----------------------------------------------------------------------------
# deterministic parameter construction (NOT a replay of torch's RNG stream:
# fixtures carry explicit state_dicts; this only has the reference's shapes
# and distributions: spiking_submodules.py:63-75,475-490, model.py:197-199)
# ----------------------------------------------------------------------------
def make_firenet_params(name, gen, *, num_bins=2, C=32, k=3, neuron=None):
    neuron = neuron or {}
    p = {}

    def U(shape, a):
        return (torch.rand(shape, generator=gen) * 2 - 1) * a

    def N(shape, ms):
        return torch.randn(shape, generator=gen) * ms[1] + ms[0]

    if name == "FireNet":
        cin = num_bins
        for l in FIRENET_LAYERS:
            if l.startswith("G"):
                for g in ("reset_gate", "update_gate", "out_gate"):
                    w = torch.empty(C, 2 * C, k, k)
                    torch.nn.init.orthogonal_(w, generator=gen)
                    p[f"{l}.{g}.weight"] = w
                    p[f"{l}.{g}.bias"] = torch.zeros(C)
            else:
                bound = 1 / math.sqrt(cin * k * k)
                p[f"{l}.conv2d.weight"] = U((C, cin, k, k), bound)
                p[f"{l}.conv2d.bias"] = U((C,), bound)
            cin = C
        bound = 1 / math.sqrt(C)
        p["pred.conv2d.weight"] = U((2, C, 1, 1), bound)
        p["pred.conv2d.bias"] = U((2,), bound)
        return p
    kind, g_rec = FIRENET_KINDS[name]
    cin = num_bins
    for l in FIRENET_LAYERS:
        p[f"{l}.ff.weight"] = U((C, cin, k, k), math.sqrt(1 / cin))
        if g_rec and l.startswith("G"):
            p[f"{l}.rec.weight"] = U((C, C, k, k), math.sqrt(1 / C))
        if kind == "lif":
            p[f"{l}.leak"] = N((C, 1, 1), neuron.get("leak", (-4.0, 0.1)))
            p[f"{l}.thresh"] = N((C, 1, 1), neuron.get("thresh", (0.8, 0.0)))
        elif kind == "plif":
            p[f"{l}.leak_v"] = N((C, 1, 1), neuron.get("leak_v", (-4.0, 0.1)))
            p[f"{l}.leak_pt"] = N((C, 1, 1), neuron.get("leak_pt", (-4.0, 0.1)))
            p[f"{l}.add_pt"] = N((C, 1, 1), neuron.get("add_pt", (-2.0, 0.1)))
            p[f"{l}.thresh"] = N((C, 1, 1), neuron.get("thresh", (0.8, 0.0)))
        elif kind == "alif":
            p[f"{l}.leak_v"] = N((C, 1, 1), neuron.get("leak_v", (-4.0, 0.1)))
            p[f"{l}.leak_t"] = N((C, 1, 1), neuron.get("leak_t", (-4.0, 0.1)))
            p[f"{l}.t0"] = N((C, 1, 1), neuron.get("t0", (0.01, 0.0)))
            p[f"{l}.t1"] = N((C, 1, 1), neuron.get("t1", (1.8, 0.0)))
        elif kind == "xlif":
            p[f"{l}.leak_v"] = N((C, 1, 1), neuron.get("leak_v", (-4.0, 0.1)))
            p[f"{l}.leak_pt"] = N((C, 1, 1), neuron.get("leak_pt", (-4.0, 0.1)))
            p[f"{l}.t0"] = N((C, 1, 1), neuron.get("t0", (0.01, 0.0)))
            p[f"{l}.t1"] = N((C, 1, 1), neuron.get("t1", (1.8, 0.0)))
        p[f"{l}.act_width"] = torch.tensor(10.0)
        cin = C
    p["pred.conv2d.weight"] = U((2, C, 1, 1), 0.01)
    p["pred.conv2d.bias"] = torch.zeros(2)
    return p


# parameters that are buffers (no gradient) per neuron kind with the ctor
# defaults learn_leak=True, learn_thresh=(True for lif/plif, False for alif/xlif)
def trainable_keys(p, learn_thresh_t=False):
    keys = []
    for kname in p:
        if kname.endswith("act_width"):
            continue
        if (kname.endswith(".t0") or kname.endswith(".t1")) and not learn_thresh_t:
            continue
        keys.append(kname)
    return keys
