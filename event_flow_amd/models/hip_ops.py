"""autograd nodes of the general path: every layer that is not one of the fused
32->32 FireNet kernels (models/engine.py) runs through these.  Each node is a
`torch.autograd.Function` whose forward and backward are calls into
libevflow_hip.so (include/evflow.h, "general path"); torch provides device
memory, the stream and the tape -- no torch arithmetic.

Tensors cross the node boundary in the reference's logical NCHW shape; in
memory they are NHWC (`channels_last` strides), which is what the kernels read.
A node output feeds the next node without a copy; only foreign NCHW-contiguous
inputs are transposed once (evf_nchw_to_nhwc).

Reference semantics implemented here:
  spiking cell step   models/spiking_submodules.py:96-126,191-227,299-334,399-435
                      (+ recurrent :516-551,618-657,730-768,836-875)
  ConvLayer / ConvLayer_  models/submodules.py:52-61,69-83
  ConvGRU             models/submodules.py:400-418
  bilinear x2         models/spiking_submodules.py:1011
  nearest xf          models/model.py:529-539
"""

import ctypes
import os
import weakref

import torch

from .. import _lib
from .spiking_util import SURROGATE_ID

KIND_ID = {"lif": 0, "plif": 1, "alif": 2, "xlif": 3}
ACT_ID = {None: 0, "tanh": 1, "sigmoid": 2, "relu": 3}


# ---------------------------------------------------------------------------
# layout helpers (memory plumbing only)
# ---------------------------------------------------------------------------
def to_nhwc(t):
    """Logical [B,C,H,W] (any strides) -> contiguous [B,H,W,C] fp32 device tensor."""
    _lib.require_gpu(t, "general path")
    if t.dtype != torch.float32:
        raise _lib.EvflowError(f"general path: unsupported dtype {t.dtype}")
    p = t.permute(0, 2, 3, 1)
    if p.is_contiguous():
        return p
    if t.is_contiguous():
        B, Cc, H, W = t.shape
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=t.device)
        _lib.call("evf_nchw_to_nhwc", _lib.ptr(t), B, Cc, H, W, _lib.ptr(out))
        return out
    return p.contiguous()


def from_nhwc(t):
    """[B,H,W,C] -> logical [B,C,H,W] view (channels_last strides)."""
    return t.permute(0, 3, 1, 2)


_POISON_NEW = os.environ.get("EVF_DEBUG_POISON_NEW", "0") == "1"


def _new(shape, dev):
    if _POISON_NEW:  # debugging aid: a kernel that leaves part of its output unwritten (or reads it) shows up as NaNs
        return torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _out_dim(n, k, s):
    return (n + 2 * (k // 2) - k) // s + 1


_EPOCH = [0]
# General convolutions (forward / input gradient) on the bf16 matrix cores with exact operand splits
# (csrc/evf_conv_b3gen.hip); EVF_CONV=f32 keeps the fp32-MFMA kernels (csrc/evf_conv_gen.hip).  Packed weights differ.
CONV_B3 = os.environ.get("EVF_CONV", "b3") != "f32"


def invalidate_packed_weights():
    """Parameters were rewritten behind torch's version counters (fused optimizer kernel)."""
    _EPOCH[0] += 1


def _pack_tag(w):
    """Identity of a weight tensor's VALUES for the pack caches.  A parameter: storage pointer + in-place version counter (+ the
    epoch fused optimizer kernels bump).  A weight-normalised weight is a fresh tensor every forward (the allocator may even hand
    out the same address again): its identity is that of the (v, g) pair it was made from (weight_norm()) -- the passes of a
    BPTT window then share one pack per operand instead of re-packing 2-4 times per cell and pass."""
    src = getattr(w, "_evf_src", None)
    if src is not None:
        return (src, w.device, _EPOCH[0])
    return (w.data_ptr(), w._version, w.device, _EPOCH[0])


class _PackCache:
    """Packed matrix-core operands of one weight tensor, re-packed when the
    parameter changes (_pack_tag)."""

    def __init__(self):
        self.store = {}

    def get(self, w, transpose, cin_off=0, cin=None):
        Cout, Ctot, k, _ = w.shape
        cin = Ctot if cin is None else cin
        sfx = "_b3" if CONV_B3 else ""
        key = (transpose, cin_off, cin, sfx)
        tag = _pack_tag(w)
        hit = self.store.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        wc = w.detach()
        if not wc.is_contiguous():
            wc = wc.contiguous()
        n = getattr(_lib.load(), "evf_conv2d" + sfx + "_packed_size")(Cout, cin, k, transpose)
        dst = _new((n,), w.device)
        _lib.call("evf_pack_conv2d_weight" + sfx, _lib.ptr(wc), Cout, cin, k, transpose, Ctot, cin_off, _lib.ptr(dst))
        try:
            wref = weakref.ref(w)  # (repack_all re-packs in place after an optimizer step)
        except TypeError:
            wref = None
        self.store[key] = (tag, dst, wref)
        return dst


def repack_all():
    """Re-pack every cached bf16 operand whose weight tensor is still alive, in ONE launch, into the buffers the caches
    already hold.  Called by train.FlatAdam.step(): after an optimizer step every weight changed, and the lazy path
    would re-pack them one launch at a time (two per layer) during the next forward / backward."""
    if not CONV_B3:
        return
    by_dev = {}
    for _ref, d in list(_CACHES.values()):
        for pc in d.values():
            if not isinstance(pc, _PackCache):
                continue
            for key, ent in pc.store.items():
                w = ent[2]() if ent[2] is not None else None
                if w is None or key[3] != "_b3" or not w.is_contiguous() or w.dtype != torch.float32 or w.device != ent[1].device:
                    continue
                if getattr(w, "_evf_src", None) is not None:
                    continue  # (a weight-normalised weight of the LAST step: stale values; the next forward makes and packs a new one)
                by_dev.setdefault(w.device, []).append((pc, key, w, ent[1], ent[2]))
    for dev, ents in by_dev.items():
        n = len(ents)
        wp = (ctypes.c_void_p * n)(*[e[2].data_ptr() for e in ents])
        dp = (ctypes.c_void_p * n)(*[e[3].data_ptr() for e in ents])
        meta = []
        for _pc, key, w, _dst, _r in ents:
            transpose, cin_off, cin, _sfx = key
            meta += [w.shape[0], cin, w.shape[2], transpose, w.shape[1], cin_off]
        with torch.cuda.device(dev):
            _lib.call("evf_pack_conv2d_weights_b3_multi", wp, dp, (ctypes.c_int * len(meta))(*meta), n)
        for pc, key, w, dst, r in ents:
            pc.store[key] = (_pack_tag(w), dst, r)


_CACHES = {}


def pack_cache(owner):
    """One cache per owning module (kept off the module so state_dict / deepcopy stay clean)."""
    c = _CACHES.get(id(owner))
    if c is None or c[0]() is not owner:
        try:
            ref = weakref.ref(owner, lambda _r, k=id(owner): _CACHES.pop(k, None))
        except TypeError:  # not weak-referenceable: no caching
            return {}
        c = (ref, {})
        _CACHES[id(owner)] = c
    return c[1]


def _wcache(owner, name):
    d = pack_cache(owner)
    if name not in d:
        d[name] = _PackCache()
    return d[name]


def _b3_ws(B, Ho, Wo, Cout, dev):
    """Split-K scratch of the bf16x3 kernels (low-resolution, many-channel layers at small batch); (None, 0) when the
    shape never splits."""
    n = _lib.load().evf_conv2d_b3_ws(B, Ho, Wo, Cout)
    if n <= 0:
        return None, 0
    return _scratch(n, dev), n


def _exact_flags(exact_from):
    """evf_conv2d_fwd_b3[_parts] flag bits of a spike-valued input: bit 2 + the first exact channel in bits 4..8."""
    if exact_from is None or not EXACT_WGRAD or not 0 <= exact_from <= 16:
        return 0
    return 4 | (int(exact_from) << 4)


def conv_fwd(x, wp, bias, y, Cin, Cout, k, stride, accumulate=0, exact_from=None):
    """exact_from: the channels of x from this index (<= 16) on are exactly representable in bf16 by construction (spike_tag
    below): evf_conv2d_fwd_b3 bit 2."""
    B, H, W, ldx = x.shape[0], x.shape[1], x.shape[2], x.stride(2)
    if CONV_B3:
        ws, n = _b3_ws(B, y.shape[1], y.shape[2], Cout, x.device)
        _lib.call("evf_conv2d_fwd_b3", _lib.ptr(x), ldx, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(y), y.stride(2), B, H, W, Cin,
                  Cout, k, stride, (accumulate & 1) | _exact_flags(exact_from), _lib.ptr(ws), n)
        return
    _lib.call("evf_conv2d_fwd", _lib.ptr(x), ldx, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(y), y.stride(2), B, H, W, Cin, Cout,
              k, stride, accumulate)


def conv_fwd_parts(x, wp, y, Cin, Cout, k, stride, exact_from, slot):
    """conv(x) with its K-split partial sums left in parts: -> (tensor, n, stride in floats): n = 1 with y itself when the plan
    does not split, else n slabs in scratch `slot`."""
    import ctypes as _ct

    B, H, W, ldx = x.shape[0], x.shape[1], x.shape[2], x.stride(2)
    n = _lib.load().evf_conv2d_b3_ws(B, y.shape[1], y.shape[2], Cout)
    ws = _scratch(n, x.device, slot) if n > 0 else None
    parts = _ct.c_int(0)
    _lib.call("evf_conv2d_fwd_b3_parts", _lib.ptr(x), ldx, _lib.ptr(wp), _lib.ptr(y), y.stride(2), B, H, W, Cin, Cout, k, stride,
              _exact_flags(exact_from), _lib.ptr(ws), max(n, 0), _ct.byref(parts))
    if parts.value > 0:
        return ws, parts.value, y.numel()
    return y, 1, 0


def conv_dgrad(g_y, wtp, g_x, Cin, Cout, k, stride, accumulate=0):
    B, H, W = g_x.shape[0], g_x.shape[1], g_x.shape[2]
    if CONV_B3:
        ws, n = _b3_ws(B, H, W, Cin, g_x.device)
        _lib.call("evf_conv2d_dgrad_b3", _lib.ptr(g_y), g_y.stride(2), _lib.ptr(wtp), _lib.ptr(g_x), g_x.stride(2), B, H, W,
                  Cin, Cout, k, stride, accumulate, _lib.ptr(ws), n)
        return
    _lib.call("evf_conv2d_dgrad", _lib.ptr(g_y), g_y.stride(2), _lib.ptr(wtp), _lib.ptr(g_x), g_x.stride(2), B, H, W, Cin,
              Cout, k, stride, accumulate)


_SCRATCH = {}
_POISON = os.environ.get("EVF_DEBUG_POISON_SCRATCH", "0") == "1"
# LIF cells: the K-split partial sums of the cell's conv(s) go straight into the neuron update (evf_conv2d_fwd_b3_parts +
# evf_lif_fwd_parts) instead of through a k_b3_reduce launch and a `cur` tensor per conv; 0: conv -> cur -> evf_neuron_fwd
LIF_PARTS = os.environ.get("EVF_LIF_PARTS", "1") != "0"


def _scratch(n, dev, slot=0):
    """Stream-ordered scratch buffer (grown on demand, reused by every weight-gradient launch; `slot`: a second one for a
    consumer that needs two conv outputs in parts at once)."""
    buf = _SCRATCH.get((dev, slot))
    if buf is None or buf.numel() < n:
        buf = _new((max(int(n), 1),), dev)
        _SCRATCH[(dev, slot)] = buf
    if _POISON:  # debugging aid: a kernel that reads scratch it did not write turns its result into NaNs
        buf.fill_(float("nan"))
    return buf


def conv_wgrad(x, g_y, g_w, g_b, Cin, Cout, k, stride, cin_total=None, cin_off=0, accumulate=0, spikes=False, analog_head=0,
               exact_from=None):
    """Weight (+ bias) gradient.  spikes: x is spike-valued (binary spikes, counts, bilinear blends of spikes): the 3x3
    stride-1 contraction runs on the bf16 matrix cores (exact there; anything else it finds is redone in fp32).
    analog_head: the first `analog_head` channels of x are real-valued (a decoder's flow prediction): they are split off
    into an fp32 call of their own so that the rest stays on the fast path.
    exact_from: provenance (spike_tag below) -- the channels of x from this index on are exactly representable in bf16 BY
    CONSTRUCTION: the bf16 kernel then reduces its own partial sums (one launch instead of three; evf_conv2d_wgrad bit 2)."""
    B, H, W = x.shape[0], x.shape[1], x.shape[2]
    ct = Cin if cin_total is None else cin_total
    if spikes and analog_head and k == 3 and stride == 1 and cin_off == 0 and Cin > 8:
        a = (analog_head + 3) // 4 * 4  # (the tail must stay 16-byte aligned)
        if not accumulate:
            _lib.zero_(g_w)
            if g_b is not None:
                _lib.zero_(g_b)
        conv_wgrad(x, g_y, g_w, g_b, a, Cout, k, stride, ct, 0, 1, spikes=False)
        conv_wgrad(x[..., a:], g_y, g_w, None, Cin - a, Cout, k, stride, ct, a, 1, spikes=True,
                   exact_from=None if exact_from is None else max(exact_from - a, 0))
        return
    ws = _scratch(_lib.load().evf_conv2d_wgrad_ws(B, H, W, Cin, Cout, k, stride), x.device)
    exact = 4 if (spikes and EXACT_WGRAD and exact_from is not None and exact_from <= 0) else 0
    _lib.call("evf_conv2d_wgrad", _lib.ptr_strided(x), x.stride(2), _lib.ptr(g_y), g_y.stride(2), _lib.ptr(g_w), _lib.ptr(g_b),
              B, H, W, Cin, Cout, k, stride, ct, cin_off, (accumulate & 1) | (0 if spikes else 2) | exact, _lib.ptr(ws))


# ---------------------------------------------------------------------------
# provenance of spike-valued activations
# ---------------------------------------------------------------------------
# A tensor produced by this module carries `_evf_spike = (bound, exact_from)` when its channels from `exact_from` on are, BY
# CONSTRUCTION, multiples of 1/16 of magnitude <= bound <= 15: output spikes of a cell ({0,1}; + a tagged residual: the bounds
# add), channel concatenations and zero paddings of such tensors, their bilinear x2 blends (weights 1, 3, 3, 9 / 16).  Such values
# have at most 8 significant bits: exactly representable in bf16, which is what lets the weight-gradient kernel skip its fp32
# verification pass (conv_wgrad(exact_from=...)).  Anything that went through a foreign op has no tag and takes the verified path.
EXACT_WGRAD = os.environ.get("EVF_WGRAD_EXACT", "1") != "0"
SPIKE_BOUND_MAX = 15.0


def spike_tag(t):
    return getattr(t, "_evf_spike", None) if torch.is_tensor(t) else None


def set_spike_tag(t, bound, exact_from=0):
    if bound is not None and bound <= SPIKE_BOUND_MAX:
        t._evf_spike = (float(bound), int(exact_from))
    return t


# ---------------------------------------------------------------------------
# norm_input (reference models/model.py:247-252)
# ---------------------------------------------------------------------------
def norm_nonzero(x):
    """x with its non-zero entries standardised (mean / unbiased std over the non-zero entries of the whole
    tensor) -- out of place (the reference writes into the caller's batch, quirk q4), no host sync.  The input
    carries no gradient (it is the event encoding)."""
    _lib.require_gpu(x, "norm_input")
    xc = x.detach().float().contiguous()
    out = torch.empty_like(xc)
    ws = torch.empty(3, dtype=torch.float64, device=xc.device)
    _lib.call("evf_norm_nonzero", _lib.ptr(xc), xc.numel(), _lib.ptr(out), ws.data_ptr())
    return out


# ---------------------------------------------------------------------------
# parameter gradients written in place
# ---------------------------------------------------------------------------
DIRECT_PARAM_GRADS = False  # switched on by train.FlatAdam (whose flat buffer every .grad is a view of)


_GRAD_LISTENERS = []  # weak references to optimizers that want to know when a kernel wrote into a bound .grad (train.FlatAdam)


def on_direct_grads(listener):
    import weakref

    _GRAD_LISTENERS.append(weakref.ref(listener))


def direct_grads_written():
    """A backward kernel is about to add into bound `.grad` buffers (no AccumulateGrad node, hence no autograd hook, sees that)."""
    for r in list(_GRAD_LISTENERS):
        o = r()
        if o is None:
            _GRAD_LISTENERS.remove(r)
        else:
            o.mark_grad_dirty()


def bound_grad(t):
    """The fp32 `.grad` buffer already bound to leaf parameter `t`, if our kernels may accumulate into it
    directly (the backward then returns None for `t`: no temporary, no AccumulateGrad add kernel per tensor
    and step).  Only with DIRECT_PARAM_GRADS: `loss.backward()` semantics are unchanged, `torch.autograd.grad`
    users keep the flag off."""
    if not DIRECT_PARAM_GRADS or t is None or not t.is_leaf or not t.requires_grad:
        return None
    g = t.grad
    if g is None or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous() or g.shape != t.shape:
        return None
    direct_grads_written()  # (the caller is about to let a kernel accumulate into it)
    return g


# ---------------------------------------------------------------------------
# spiking cell step
# ---------------------------------------------------------------------------
def act_width(cell):
    """Surrogate width as a host float, read from the device buffer once (it is never trained)."""
    d = pack_cache(cell)
    if "act_width" not in d:
        d["act_width"] = float(cell.act_width)
    return d["act_width"]


def cell_params(cell):
    """The four per-channel parameter slots of evf_neuron_fwd/bwd for this cell."""
    if cell.kind == "lif":
        return [cell.leak, cell.thresh, None, None]
    if cell.kind == "plif":
        return [cell.leak_v, cell.thresh, cell.leak_pt, cell.add_pt]
    if cell.kind == "alif":
        return [cell.leak_v, cell.t0, cell.t1, cell.leak_t]
    return [cell.leak_v, cell.t0, cell.t1, cell.leak_pt]


_NEURON_WS = {}


def _neuron_ws(dev):
    """Zeroed scratch of evf_neuron_bwd (the kernel hands it back zeroed; one per device, calls are stream-ordered)."""
    if dev not in _NEURON_WS:
        _NEURON_WS[dev] = _lib.zeros(32 * 4096 + 64, dtype=torch.float32, device=dev)
    return _NEURON_WS[dev]


# State routing (train.capture_window_cycle): a step replayed from hipGraphs needs its recurrent state at FIXED addresses, and the
# last graph of a cycle has to leave its final state where the first graph reads it.  Instead of copying (410 MB of state per
# step of the LIF-EV-FlowNet of BASELINE configs[3]: 0.25 ms, and memcpy NODES in the graph) a cell whose PREVIOUS state starts
# at a registered address writes its NEW state straight into the tensor registered for it.
_STATE_OUT = {}  # data_ptr of a previous state -> the tensor (logical layout of that state, NHWC memory) the new state goes into


def route_states(prev, into):
    """prev / into: lists of state tensors (a block's stacked state counts as one).  Until clear_state_routes() a cell (block)
    that is handed prev[k] as its previous state writes its new state into the memory of into[k] (same shape and layout;
    anything else is ignored and the cell allocates as usual -- the caller compares addresses afterwards)."""
    for a, b in zip(prev, into):
        if a is None or b is None:
            continue
        # a cell must never write its new state over the previous one it still has to read in the backward pass (a cycle of ONE
        # window would map every state tensor onto itself): overlapping storages are not routed, the caller copies instead.
        a0, a1 = a.data_ptr(), a.data_ptr() + a.numel() * a.element_size()
        b0, b1 = b.data_ptr(), b.data_ptr() + b.numel() * b.element_size()
        if a0 < b1 and b0 < a1:
            continue
        _STATE_OUT[a.data_ptr()] = b


def clear_state_routes():
    _STATE_OUT.clear()


def _routed(prev, shape_mem):
    if not _STATE_OUT or prev is None:
        return None
    t = _STATE_OUT.get(prev.data_ptr())
    if t is None or t.dim() < 4:
        return None
    nd = t.dim()
    m = t.detach().permute(*range(nd - 3), nd - 2, nd - 1, nd - 3)  # logical [.., C, H, W] -> memory [.., H, W, C]
    if not m.is_contiguous() or tuple(m.shape) != tuple(shape_mem) or m.dtype != torch.float32:
        return None
    return m


class StateSlots:
    """New-state tensors of the n cells of a block in ONE buffer [n, S, B, H, W, C], so that the block's stacked state
    (reference: torch.stack([ff, rec]), spiking_submodules.py:926, :973) exists without a copy."""

    def __init__(self, n):
        self.n, self.base, self.used = n, None, 0

    def take(self, shape, dev, prev=None):
        if self.base is None:
            # (prev: the first cell's previous state = slice 0 of the block's previous stacked state, i.e. that buffer's address)
            self.base = _routed(prev, (self.n,) + tuple(shape))
        if self.base is None:
            self.base = _new((self.n,) + tuple(shape), dev)
        if self.used >= self.n or tuple(self.base.shape[1:]) != tuple(shape):
            self.used = self.n + 1  # (a cell of another shape: this block falls back to torch.stack)
            return _new(shape, dev)
        self.used += 1
        return self.base[self.used - 1]

    def complete(self):
        return self.base is not None and self.used == self.n


class _StackInPlace(torch.autograd.Function):
    """torch.stack of states that already lie in one buffer: forward returns that buffer, backward hands each state its
    slice of the gradient (views, no kernels)."""

    @staticmethod
    def forward(ctx, slots, *states):
        ctx.n = len(states)
        return slots.base.permute(0, 1, 2, 5, 3, 4)  # logical [n,S,B,C,H,W]

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[k] for k in range(ctx.n))


def stack_states(states, slots=None):
    """torch.stack(states) of logical [S,B,C,H,W] states, keeping the NHWC memory layout (stacking the NHWC views is a
    plain copy; a direct torch.stack would transpose twice per pass) -- and no copy at all when the cells wrote their
    states into the slots of one buffer."""
    if slots is not None and slots.complete() and len(states) == slots.n:
        return _StackInPlace.apply(slots, *states)
    return torch.stack([s.permute(0, 1, 3, 4, 2) for s in states]).permute(0, 1, 2, 5, 3, 4)


class _CellStep(torch.autograd.Function):
    """(input, previous state, residual) -> (output spikes [+ residual], new state).
    state: logical [S,B,C,H,W] (S = 2, or 3 with the adaptation trace), NHWC in memory."""

    @staticmethod
    def forward(ctx, cell, x, state, residual, slots, wff, wrec, p0, p1, p2, p3):
        ctx.set_materialize_grads(False)
        kind = KIND_ID[cell.kind]
        ns = 2 if kind == 0 else 3
        xn = to_nhwc(x)
        B, H, W, Cin = xn.shape
        C, k, s = cell.hidden_size, cell.kernel_size, cell.stride
        Ho, Wo = _out_dim(H, k, s), _out_dim(W, k, s)
        dev = xn.device
        sp = None
        if state is not None:
            sp = state.permute(0, 1, 3, 4, 2)
            if not sp.is_contiguous():
                sp = sp.contiguous()
        rn = to_nhwc(residual) if residual is not None else None
        Cw = wff.shape[1]
        if not 0 <= Cin - Cw < 4:
            raise _lib.EvflowError(f"input has {Cin} channels, the cell expects {Cw}")
        # (Cin - Cw trailing channels = zero padding that keeps 16-byte alignment; the packed weight is zero there)
        cur = _new((B, Ho, Wo, C), dev)
        xtag = spike_tag(x)
        x_exact = xtag[1] if xtag is not None else None  # (first exact channel: 0, or 2 behind a decoder's flow channels)
        rec_exact = None if getattr(cell, "gnorm", False) else 0  # (the previous state's z slice: spikes unless group norm rescaled them)
        parts = None
        if kind == 0 and CONV_B3 and LIF_PARTS:
            # the conv kernels leave their K-split partial sums in scratch, the neuron update adds them itself
            pa = conv_fwd_parts(xn, _wcache(cell, "ff").get(wff, 0, 0, Cin), cur, Cin, C, k, s, x_exact, 0)
            pb = (None, 0, 0)
            if cell.recurrent and sp is not None:
                cur2 = _new((B, Ho, Wo, C), dev)
                pb = conv_fwd_parts(sp[1], _wcache(cell, "rec").get(wrec, 0), cur2, C, C, k, 1, rec_exact, 1)
            parts = (pa, pb)
        else:
            conv_fwd(xn, _wcache(cell, "ff").get(wff, 0, 0, Cin), None, cur, Cin, C, k, s, exact_from=x_exact)
            if cell.recurrent and sp is not None:
                conv_fwd(sp[1], _wcache(cell, "rec").get(wrec, 0), None, cur, C, C, k, 1, accumulate=1, exact_from=rec_exact)
        P = None
        if kind in (1, 3):
            ws = _new((B * H * W,), dev)
            P = _new((B, Ho, Wo), dev)
            _lib.call("evf_pretrace_fwd", _lib.ptr(xn), xn.stride(2), B, H, W, Cin, k, s, _lib.ptr(ws), _lib.ptr(P))
        if slots is not None:
            new = slots.take((ns, B, Ho, Wo, C), dev, state)
        else:
            new = _routed(state, (ns, B, Ho, Wo, C))
            if new is None:
                new = _new((ns, B, Ho, Wo, C), dev)
        # without a residual the cell's output IS its spike tensor, the z entry of the new state: no second copy of it is written
        # (one tensor pass of seven less per forward cell; EVF_OUT_ALIAS=0: a tensor of its own as before)
        alias_out = OUT_ALIAS and rn is None
        out = new[1] if alias_out else _new((B, Ho, Wo, C), dev)
        out_ptr = None if alias_out else _lib.ptr(out)
        prm = [p.detach().reshape(-1).contiguous() if p is not None else None for p in (p0, p1, p2, p3)]
        if parts is not None:
            (ta, na, sa), (tb, nb, sb) = parts
            _lib.call("evf_lif_fwd_parts", _lib.ptr(ta), na, sa, _lib.ptr(tb), nb, sb, _lib.ptr(sp[0]) if sp is not None else None,
                      _lib.ptr(sp[1]) if sp is not None else None, _lib.ptr(rn), _lib.ptr(prm[0]), _lib.ptr(prm[1]), B * Ho * Wo, C,
                      1 if cell.hard_reset else 0, _lib.ptr(new[0]), _lib.ptr(new[1]), out_ptr)
        else:
            _lib.call("evf_neuron_fwd", kind, _lib.ptr(cur), _lib.ptr(sp[0]) if sp is not None else None,
                  _lib.ptr(sp[1]) if sp is not None else None, _lib.ptr(sp[2]) if (sp is not None and ns == 3) else None,
                  _lib.ptr(P), _lib.ptr(rn), _lib.ptr(prm[0]), _lib.ptr(prm[1]), _lib.ptr(prm[2]), _lib.ptr(prm[3]),
                  B * Ho * Wo, C, 1 if cell.hard_reset else 0, _lib.ptr(new[0]), _lib.ptr(new[1]),
                  _lib.ptr(new[2]) if ns == 3 else None, out_ptr)
        ctx.cell, ctx.kind, ctx.ns = cell, kind, ns
        tag = spike_tag(x)
        ctx.exact_from = tag[1] if tag is not None else None  # provenance of the input (conv_wgrad)
        ctx.geom = (B, H, W, Cin, Ho, Wo, C, k, s)
        ctx.saved = (xn, sp, new, P, prm, wff, wrec)
        ctx.has_res = residual is not None
        # the recurrent conv reads the z slice of the previous state: spikes by the cell's definition (a caller who hands in
        # anything else gets NaN weight gradients from the exact path, never rounded ones) -- unless group norm rescaled them
        ctx.rec_spikes = state is not None and not getattr(cell, "gnorm", False)
        # the output TWICE (two views of the same memory): a layer output with two consumers (encoder -> next encoder + skip,
        # decoder -> prediction + next decoder, residual block input) hands its second consumer the twin (hip_ops.twin), and the
        # two upstream gradients arrive HERE as two arguments -- the neuron kernel adds them as it loads them (g_z_out + g_z_out2)
        # instead of autograd launching an add kernel per fork (11 per step of the spiking EV-FlowNet).  An unused twin costs nothing.
        return from_nhwc(out), new.permute(0, 1, 4, 2, 3), from_nhwc(out)

    @staticmethod
    def backward(ctx, g_out, g_state, g_out2=None):
        cell, kind, ns = ctx.cell, ctx.kind, ctx.ns
        B, H, W, Cin, Ho, Wo, C, k, s = ctx.geom
        xn, sp, new, P, prm, wff, wrec = ctx.saved
        dev = xn.device
        need = ctx.needs_input_grad  # (cell, x, state, residual, slots, wff, wrec, p0..p3)
        need = need[:4] + need[5:]  # (indices below: cell, x, state, residual, wff, wrec, p0..p3)
        if g_out is None and g_out2 is not None:
            g_out, g_out2 = g_out2, None
        if g_out is None and g_state is None:
            return (None,) * 11
        gon = to_nhwc(g_out) if g_out is not None else None
        gon2 = to_nhwc(g_out2) if g_out2 is not None else None
        gs = None
        if g_state is not None:
            gs = g_state.permute(0, 1, 3, 4, 2)
            if not gs.is_contiguous():
                gs = gs.contiguous()
        if gon2 is not None and (gs is not None or (ctx.has_res and need[3])):
            # the kernel's second gradient slot is taken by the state's z entry (a window of several passes), or the sum itself
            # is needed (it is the residual's gradient): one add kernel, as autograd would have launched
            ysum = _new(tuple(gon.shape), dev)
            _lib.call("evf_act_fwd", 0, _lib.ptr(gon), _lib.ptr(gon2), ysum.numel(), _lib.ptr(ysum))
            gon, gon2 = ysum, None
            g_out = from_nhwc(ysum)
        g_cur = _new((B, Ho, Wo, C), dev)
        # the kernel writes dL/dv_prev and dL/d(trace)_prev; dL/dz_prev only for ALIF -- otherwise that slice is
        # the recurrent conv's input gradient (written below) or zero
        # (no previous state, or one that takes no gradient: the kernel skips those stores)
        want_prev = sp is not None and need[2]
        g_prev = _new((ns, B, Ho, Wo, C), dev) if want_prev else None
        rec_dgrad = cell.recurrent and sp is not None and need[2]
        if kind != 2 and not rec_dgrad and sp is not None and need[2]:  # (only a returned state gradient needs the zeros)
            _lib.zero_(g_prev[1])
        g_P = _new((B, Ho, Wo), dev) if kind in (1, 3) else None
        params = cell_params(cell)
        d_prm = [bound_grad(params[i]) if (prm[i] is not None and need[6 + i]) else None for i in range(4)]
        g_prm = [d_prm[i].view(-1) if d_prm[i] is not None else
                 (_lib.zeros(C, dtype=torch.float32, device=dev) if (prm[i] is not None and need[6 + i]) else None)
                 for i in range(4)]
        _lib.call("evf_neuron_bwd", kind, _lib.ptr(gs[0]) if gs is not None else None, _lib.ptr(gon),
                  _lib.ptr(gs[1]) if gs is not None else _lib.ptr(gon2), _lib.ptr(gs[2]) if (gs is not None and ns == 3) else None,
                  _lib.ptr(new[0]), _lib.ptr(new[2]) if ns == 3 else None, _lib.ptr(sp[0]) if sp is not None else None,
                  _lib.ptr(sp[1]) if sp is not None else None, _lib.ptr(sp[2]) if (sp is not None and ns == 3) else None,
                  _lib.ptr(P), _lib.ptr(prm[0]), _lib.ptr(prm[1]), _lib.ptr(prm[2]), _lib.ptr(prm[3]), B * Ho * Wo, C,
                  1 if cell.hard_reset else 0, SURROGATE_ID[cell.activation], act_width(cell), _lib.ptr(g_cur),
                  _lib.ptr(g_prev[0]) if want_prev else None, _lib.ptr(g_prev[1]) if want_prev else None,
                  _lib.ptr(g_prev[2]) if (want_prev and ns == 3) else None, _lib.ptr(g_P),
                  _lib.ptr(g_prm[0]), _lib.ptr(g_prm[1]), _lib.ptr(g_prm[2]), _lib.ptr(g_prm[3]), _lib.ptr(_neuron_ws(dev)))
        g_wff = g_wrec = g_x = None
        if need[4]:
            d = bound_grad(wff)
            head = int(getattr(cell, "analog_input_channels", 0))  # (a decoder's flow-prediction channels, models/unet.py)
            if d is not None:
                conv_wgrad(xn, g_cur, d, None, Cin, C, k, s, cin_total=wff.shape[1], accumulate=1, spikes=True, analog_head=head,
                           exact_from=ctx.exact_from)
            else:
                g_wff = _new(tuple(wff.shape), dev)
                conv_wgrad(xn, g_cur, g_wff, None, Cin, C, k, s, cin_total=wff.shape[1], spikes=True, analog_head=head,
                           exact_from=ctx.exact_from)
        if cell.recurrent and need[5]:
            d = bound_grad(wrec)
            # (the recurrent input is the cell's own previous SPIKES unless group norm rescaled them: exact by construction)
            rec_exact = 0 if ctx.rec_spikes else None
            if sp is not None and d is not None:
                conv_wgrad(sp[1], g_cur, d, None, C, C, k, 1, accumulate=1, spikes=True, exact_from=rec_exact)
            elif sp is not None:
                g_wrec = _new(tuple(wrec.shape), dev)
                conv_wgrad(sp[1], g_cur, g_wrec, None, C, C, k, 1, spikes=True, exact_from=rec_exact)
            elif d is None:
                g_wrec = _lib.zeros(tuple(wrec.shape), dtype=torch.float32, device=dev)
        if need[1]:
            g_xn = _new((B, H, W, Cin), dev)
            conv_dgrad(g_cur, _wcache(cell, "ffT").get(wff, 1, 0, Cin), g_xn, Cin, C, k, s)
            if g_P is not None:
                _lib.call("evf_pretrace_bwd", _lib.ptr(xn), xn.stride(2), _lib.ptr(g_P), B, H, W, Cin, k, s, _lib.ptr(g_xn),
                          Cin, 1)
            g_x = from_nhwc(g_xn)
        g_st = None
        if sp is not None and need[2]:
            if cell.recurrent:  # the recurrent conv reads the previous spikes NOT detached (:530)
                conv_dgrad(g_cur, _wcache(cell, "recT").get(wrec, 1), g_prev[1], C, C, k, 1, accumulate=1 if kind == 2 else 0)
            g_st = g_prev.permute(0, 1, 4, 2, 3)
        g_res = g_out if (ctx.has_res and need[3]) else None
        shp = lambda i: g_prm[i].view(C, 1, 1) if (g_prm[i] is not None and d_prm[i] is None) else None  # noqa: E731
        return None, g_x, g_st, g_res, None, g_wff, g_wrec, shp(0), shp(1), shp(2), shp(3)


class _WeightNorm(torch.autograd.Function):
    """w = v * g / ||v|| per output channel (nn.utils.weight_norm, dim = 0), evf_weight_norm_fwd / _bwd."""

    @staticmethod
    def forward(ctx, v, g):
        vc, gc = v.detach().contiguous(), g.detach().reshape(-1).contiguous()
        Cout, n = vc.shape[0], vc[0].numel()
        w = _new(tuple(vc.shape), vc.device)
        nrm = _new((Cout,), vc.device)
        _lib.call("evf_weight_norm_fwd", _lib.ptr(vc), _lib.ptr(gc), Cout, n, _lib.ptr(w), _lib.ptr(nrm))
        ctx.saved = (vc, gc, nrm)
        ctx.gshape = tuple(g.shape)
        return w

    @staticmethod
    def backward(ctx, gw):
        vc, gc, nrm = ctx.saved
        Cout, n = vc.shape[0], vc[0].numel()
        gwc = gw.contiguous()
        gv, gg = _new(tuple(vc.shape), vc.device), _new((Cout,), vc.device)
        _lib.call("evf_weight_norm_bwd", _lib.ptr(gwc), _lib.ptr(vc), _lib.ptr(gc), _lib.ptr(nrm), Cout, n, _lib.ptr(gv), _lib.ptr(gg))
        return gv, gg.reshape(ctx.gshape)


def weight_norm(conv):
    """The effective weight of an nn.utils.weight_norm-wrapped conv (parameters weight_g, weight_v under the reference's
    names); its forward pre-hook never runs here because the conv module itself is never called."""
    v, g = conv.weight_v, conv.weight_g
    w = _WeightNorm.apply(v, g)
    w._evf_src = (v.data_ptr(), v._version, g.data_ptr(), g._version)
    return w


def conv_weight(conv):
    return weight_norm(conv) if hasattr(conv, "weight_g") else conv.weight


def cell_forward(cell, input_, prev_state, residual=0, slots=None):
    """Reference signature: cell(input_, prev_state, residual=0) -> (out, state).  slots: optional StateSlots of the
    enclosing block (the new state is written into the block's stacked-state buffer)."""
    res = residual if torch.is_tensor(residual) else None
    if not torch.is_tensor(residual) and residual != 0:
        raise _lib.EvflowError("residual must be a tensor or 0")
    p = cell_params(cell)
    if getattr(cell, "gnorm", False):  # norm="group" LIF cells: spiking_submodules.py:98-99, :518-529
        input_ = group_norm1(input_, cell.norm_ff if cell.recurrent else cell.norm)
        if cell.recurrent:
            if prev_state is None:  # (the reference normalises the zero state as well: z becomes the layer's bias)
                k, s = cell.kernel_size, cell.stride
                shp = (input_.shape[0], cell.hidden_size, _out_dim(input_.shape[2], k, s), _out_dim(input_.shape[3], k, s))
                prev_state = _lib.zeros((2,) + shp, dtype=torch.float32, device=input_.device)
            # the NORMALISED previous spikes feed the recurrent conv and the (detached) reset alike (:528-529, :539-546)
            prev_state = torch.stack([prev_state[0], group_norm1(prev_state[1], cell.norm_rec)])
    # (norm="weight": the normalised weights are new tensors every call; their packs are keyed by the (v, g) pair: _pack_tag)
    wrec = conv_weight(cell.rec) if cell.recurrent else None
    out, new, out2 = _CellStep.apply(cell, input_, prev_state, res, slots, conv_weight(cell.ff), wrec, *p)
    # provenance: out = spikes (+ residual); the state's z slice is spikes
    rtag = spike_tag(res)
    for o in (out, out2):
        if res is None:
            set_spike_tag(o, 1.0)
        elif rtag is not None and rtag[1] == 0:
            set_spike_tag(o, 1.0 + rtag[0])
            # spikes + a residual that is a bilinear blend (multiples of 1/16) is no longer integer-valued: a second up-sampling of
            # such a sum would make multiples of 1/256, which the exact-input kernels answer with NaN (the promise is guarded)
            o._evf_spike_int = getattr(res, "_evf_spike_int", True)
    if FORK_TWIN:
        out._evf_twin = out2
    return out, new


# OPT-IN (EVF_FORK_TWIN=1): measured on the spiking EV-FlowNet step (BASELINE configs[3], A/B twice on one box) 6.79 ms with the
# twins against 6.775 ms without -- the neuron kernel's state-gradient variant (its second z-gradient slot) costs what the nine
# saved add launches (~9 us each) bring.  Off: every consumer takes the output itself and autograd adds the gradients.
FORK_TWIN = os.environ.get("EVF_FORK_TWIN", "0") == "1"
OUT_ALIAS = os.environ.get("EVF_OUT_ALIAS", "1") != "0"  # a cell without residual returns the z slice of its new state as its output


def twin(x):
    """The second view of a cell output for its SECOND consumer (see _CellStep.forward): same memory, same values, its own slot
    in the producing cell's backward.  Handed out once; a third consumer (or a tensor that is no cell output) gets `x` itself."""
    t = getattr(x, "_evf_twin", None)
    if t is None:
        return x
    x._evf_twin = None
    return t


# ---------------------------------------------------------------------------
# conv + bias + pointwise activation (ConvLayer / ConvLayer_)
# ---------------------------------------------------------------------------
HEAD1X1 = os.environ.get("EVF_HEAD1X1", "1") != "0"


def _head1x1_ok(Cin, Cout, k, stride, residual, weight):
    q = Cin // 4
    return (HEAD1X1 and CONV_B3 and k == 1 and stride == 1 and residual is None and 1 <= Cout <= 4 and Cin % 4 == 0 and Cin == weight.shape[1]
            and 1 <= q <= 64 and (q & (q - 1)) == 0)


class _ConvAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, weight, bias, residual, stride, act):
        xn = to_nhwc(x)
        B, H, W, Cin = xn.shape
        Cout, _, k, _ = weight.shape
        Ho, Wo = _out_dim(H, k, stride), _out_dim(W, k, stride)
        ctx.head1 = _head1x1_ok(Cin, Cout, k, stride, residual, weight)
        if ctx.head1:  # a 1x1 layer with a handful of outputs (the flow predictions): one streaming kernel each way
            y = _new((B, H, W, Cout), xn.device)
            wc = weight.detach().reshape(Cout, Cin).contiguous()
            bc = bias.detach().contiguous() if bias is not None else None
            _lib.call("evf_head1x1_fwd", _lib.ptr_strided(xn), xn.stride(2), _lib.ptr(wc), _lib.ptr(bc), act, B * H * W, Cin, Cout,
                      _lib.ptr(y), Cout)
            ctx.owner, ctx.act, ctx.stride = owner, act, stride
            ctx.saved = (xn, y, weight)
            ctx.bias = bias
            ctx.has_bias, ctx.has_res = bias is not None, False
            return from_nhwc(y)
        if not 0 <= Cin - weight.shape[1] < 4:
            raise _lib.EvflowError(f"input has {Cin} channels, the layer expects {weight.shape[1]}")
        # (Cin - Cw trailing channels = zero padding that keeps 16-byte alignment; the packed weight is zero there)
        y = _new((B, Ho, Wo, Cout), xn.device)
        bc = bias.detach().contiguous() if bias is not None else None
        conv_fwd(xn, _wcache(owner, "w").get(weight, 0, 0, Cin), bc, y, Cin, Cout, k, stride)
        rn = to_nhwc(residual) if residual is not None else None
        if act != 0 or rn is not None:
            _lib.call("evf_act_fwd", act, _lib.ptr(y), _lib.ptr(rn), y.numel(), _lib.ptr(y))
        ctx.owner, ctx.act, ctx.stride = owner, act, stride
        ctx.saved = (xn, y, weight)
        ctx.bias = bias
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g_y):
        xn, y, weight = ctx.saved
        B, H, W, Cin = xn.shape
        Cout, _, k, _ = weight.shape
        need = ctx.needs_input_grad  # (owner, x, weight, bias, residual, stride, act)
        if ctx.head1:
            _lib.require_gpu(g_y, "1x1 head backward")
            # the upstream gradient as it comes: NCHW planes (the loss's flow-map gradient) or NHWC rows -- no layout pass
            if g_y.is_contiguous():
                gsrc, hw = g_y, H * W
            else:
                gsrc, hw = to_nhwc(g_y), 0
            d_w = bound_grad(weight) if need[2] else None
            d_b = bound_grad(ctx.bias) if (ctx.has_bias and need[3]) else None
            direct = d_w is not None and (not ctx.has_bias or d_b is not None)
            g_w = d_w if direct else _new(tuple(weight.shape), xn.device)
            g_b = (d_b if direct else _new((Cout,), xn.device)) if ctx.has_bias else None
            g_xn = _new((B, H, W, Cin), xn.device) if need[1] else None
            ws = _scratch(_lib.load().evf_head1x1_ws(Cin, Cout), xn.device)
            wc = weight.detach().reshape(Cout, Cin).contiguous()
            _lib.call("evf_head1x1_bwd", _lib.ptr_strided(xn), xn.stride(2), _lib.ptr(y), Cout, _lib.ptr(gsrc), hw, _lib.ptr(wc), ctx.act,
                      B * H * W, Cin, Cout, _lib.ptr(g_xn), Cin, _lib.ptr(g_w), _lib.ptr(g_b), 1 if direct else 0, _lib.ptr(ws))
            return (None, from_nhwc(g_xn) if g_xn is not None else None, None if direct else g_w,
                    None if (direct or not ctx.has_bias) else g_b, None, None, None)
        g = to_nhwc(g_y)
        if ctx.act != 0:
            gp = _new(tuple(g.shape), g.device)
            _lib.call("evf_act_bwd", ctx.act, _lib.ptr(y), _lib.ptr(g), g.numel(), _lib.ptr(gp))
            g = gp
        g_w = g_b = g_x = None
        if need[2] or (ctx.has_bias and need[3]):
            d_w, d_b = bound_grad(weight), bound_grad(ctx.bias) if ctx.has_bias else None
            if d_w is not None and (not ctx.has_bias or d_b is not None):
                conv_wgrad(xn, g, d_w, d_b, Cin, Cout, k, ctx.stride, cin_total=weight.shape[1], accumulate=1)
            else:
                g_w = _new(tuple(weight.shape), g.device)
                g_b = _new((Cout,), g.device) if ctx.has_bias else None
                conv_wgrad(xn, g, g_w, g_b, Cin, Cout, k, ctx.stride, cin_total=weight.shape[1])
        if need[1]:
            g_xn = _new((B, H, W, Cin), g.device)
            conv_dgrad(g, _wcache(ctx.owner, "wT").get(weight, 1, 0, Cin), g_xn, Cin, Cout, k, ctx.stride)
            g_x = from_nhwc(g_xn)
        g_res = from_nhwc(g) if (ctx.has_res and need[4]) else None
        return None, g_x, g_w, g_b, g_res, None, None


def conv_act(owner, x, weight, bias, stride=1, activation=None, residual=None):
    if activation not in ACT_ID:
        raise AttributeError(activation)
    return _ConvAct.apply(owner, x, weight, bias, residual, stride, ACT_ID[activation])


# ---------------------------------------------------------------------------
# stand-alone activation (+ residual), batch / instance norm, transposed conv: the ANN layers with norm = "BN" / "IN" and
# the transposed-conv decoders (reference models/submodules.py:46-56, 86-137, 169-180, 273-301)
# ---------------------------------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, act):
        xn = to_nhwc(x)
        rn = to_nhwc(residual) if residual is not None else None
        y = _new(tuple(xn.shape), xn.device)
        _lib.call("evf_act_fwd", act, _lib.ptr(xn), _lib.ptr(rn), xn.numel(), _lib.ptr(y))
        ctx.act, ctx.has_res = act, residual is not None
        ctx.saved = y
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g_y):
        g = to_nhwc(g_y)
        if ctx.act != 0:
            gp = _new(tuple(g.shape), g.device)
            _lib.call("evf_act_bwd", ctx.act, _lib.ptr(ctx.saved), _lib.ptr(g), g.numel(), _lib.ptr(gp))
            g = gp
        gx = from_nhwc(g)
        return gx, (gx if ctx.has_res else None), None


def activation(x, act, residual=None):
    """act(x [+ residual]) with act in tanh / sigmoid / relu / None."""
    if act not in ACT_ID:
        raise NotImplementedError(f"activation {act!r} has no HIP kernel (tanh/sigmoid/relu/None)")
    if act is None and residual is None:
        return x
    return _Act.apply(x, residual if torch.is_tensor(residual) else None, ACT_ID[act])


class _Norm2d(torch.autograd.Function):
    """nn.BatchNorm2d / nn.InstanceNorm2d(track_running_stats=True) forward + backward.  `groups` = 1 (batch statistics over
    B*H*W) or B (per-instance statistics over H*W).  use_input_stats False = eval mode: the running statistics normalise
    (both layer types: torch/nn/modules/instancenorm.py uses them whenever track_running_stats is set)."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer, instance, use_input_stats):
        xn = to_nhwc(x)
        B, H, W, C = xn.shape
        G, npg = (B, H * W) if instance else (1, B * H * W)
        dev = xn.device
        eps = float(layer.eps)
        if use_input_stats:
            s1 = _new((G, C), dev)
            _lib.call("evf_chan_reduce", _lib.ptr(xn), C, None, 0, None, None, 0, G, npg, C, _lib.ptr(s1))
            mean = s1 / npg
            s2 = _new((G, C), dev)
            _lib.call("evf_chan_reduce", _lib.ptr(xn), C, None, 0, _lib.ptr(mean), None, 1, G, npg, C, _lib.ptr(s2))
            var = s2 / npg  # biased: what normalises
            if layer.track_running_stats and layer.running_mean is not None:
                with torch.no_grad():
                    m = layer.momentum if layer.momentum is not None else 1.0 / float(layer.num_batches_tracked + 1)  # cumulative average
                    unb = s2 / max(npg - 1, 1)  # unbiased: what the running variance tracks
                    layer.running_mean.mul_(1 - m).add_(m * mean.mean(0))
                    layer.running_var.mul_(1 - m).add_(m * unb.mean(0))
                    if layer.num_batches_tracked is not None and not instance:
                        layer.num_batches_tracked += 1  # (nn.InstanceNorm2d never counts: F.instance_norm has no counter)
        else:
            mean = layer.running_mean.detach().float().reshape(1, C).expand(G, C).contiguous()
            var = layer.running_var.detach().float().reshape(1, C).expand(G, C).contiguous()
        rstd = torch.rsqrt(var + eps)
        w = weight.detach().float().reshape(1, C) if weight is not None else torch.ones((1, C), device=dev)
        b = bias.detach().float().reshape(1, C) if bias is not None else _lib.zeros((1, C), device=dev)
        scale = (w * rstd).contiguous()
        shift = (b - mean * w * rstd).contiguous()
        y = _new((B, H, W, C), dev)
        _lib.call("evf_chan_affine", None, 0, _lib.ptr(xn), C, None, _lib.ptr(scale), _lib.ptr(shift), G, npg, C, _lib.ptr(y), C)
        ctx.saved = (xn, mean.contiguous(), rstd.contiguous(), w)
        ctx.geo = (G, npg, C, use_input_stats, weight is not None, bias is not None)
        ctx.wshape = None if weight is None else tuple(weight.shape)
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g_y):
        xn, mean, rstd, w = ctx.saved
        G, npg, C, stats, has_w, has_b = ctx.geo
        g = to_nhwc(g_y)
        dev = g.device
        s1 = _new((G, C), dev)  # sum g
        s2 = _new((G, C), dev)  # sum g * xhat
        _lib.call("evf_chan_reduce", _lib.ptr(g), C, None, 0, None, None, 0, G, npg, C, _lib.ptr(s1))
        _lib.call("evf_chan_reduce", _lib.ptr(xn), C, _lib.ptr(g), C, _lib.ptr(mean), _lib.ptr(rstd), 2, G, npg, C, _lib.ptr(s2))
        A = (w * rstd).contiguous()
        if stats:  # the statistics depend on x
            Bc = (-(rstd * rstd) * w * s2 / npg).contiguous()
            Cc = (-Bc * mean - rstd * w * s1 / npg).contiguous()
        else:
            Bc = _lib.zeros(A.shape, dtype=A.dtype, device=A.device)
            Cc = _lib.zeros(A.shape, dtype=A.dtype, device=A.device)
        gx = _new(tuple(xn.shape), dev)
        _lib.call("evf_chan_affine", _lib.ptr(g), C, _lib.ptr(xn), C, _lib.ptr(A), _lib.ptr(Bc), _lib.ptr(Cc), G, npg, C, _lib.ptr(gx), C)
        g_w = s2.sum(0).reshape(ctx.wshape) if has_w else None
        g_b = s1.sum(0).reshape(ctx.wshape) if has_b else None
        return from_nhwc(gx), g_w, g_b, None, None, None


class _GroupNorm1(torch.autograd.Function):
    """nn.GroupNorm(1, C): statistics over (C, H, W) of every sample, per-channel affine.  The per-(sample, channel) sums come
    from evf_chan_reduce, their combination over the channels is [B, C] host-side arithmetic, the element passes are
    evf_chan_affine -- the scheme of _Norm2d with the group's sums in place of the channel's."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xn = to_nhwc(x)
        B, H, W, C = xn.shape
        npg, dev = H * W, xn.device
        N = float(C * npg)
        s1 = _new((B, C), dev)
        _lib.call("evf_chan_reduce", _lib.ptr(xn), C, None, 0, None, None, 0, B, npg, C, _lib.ptr(s1))
        mean = (s1.sum(1, keepdim=True) / N).expand(B, C).contiguous()
        s2 = _new((B, C), dev)
        _lib.call("evf_chan_reduce", _lib.ptr(xn), C, None, 0, _lib.ptr(mean), None, 1, B, npg, C, _lib.ptr(s2))
        rstd = torch.rsqrt(s2.sum(1, keepdim=True) / N + eps).expand(B, C).contiguous()
        w = weight.detach().float().reshape(1, C)
        b = bias.detach().float().reshape(1, C)
        scale = (w * rstd).contiguous()
        shift = (b - mean * w * rstd).contiguous()
        y = _new((B, H, W, C), dev)
        _lib.call("evf_chan_affine", None, 0, _lib.ptr(xn), C, None, _lib.ptr(scale), _lib.ptr(shift), B, npg, C, _lib.ptr(y), C)
        ctx.saved = (xn, mean, rstd, w)
        ctx.wshape = tuple(weight.shape)
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g_y):
        xn, mean, rstd, w = ctx.saved
        B, H, W, C = xn.shape
        npg, dev = H * W, xn.device
        N = float(C * npg)
        g = to_nhwc(g_y)
        s1 = _new((B, C), dev)  # sum g
        s2 = _new((B, C), dev)  # sum g * xhat
        _lib.call("evf_chan_reduce", _lib.ptr(g), C, None, 0, None, None, 0, B, npg, C, _lib.ptr(s1))
        _lib.call("evf_chan_reduce", _lib.ptr(xn), C, _lib.ptr(g), C, _lib.ptr(mean), _lib.ptr(rstd), 2, B, npg, C, _lib.ptr(s2))
        S1 = (w * s1).sum(1, keepdim=True)
        S2 = (w * s2).sum(1, keepdim=True)
        A = (w * rstd).contiguous()
        Bc = (-(rstd * rstd) * S2 / N).contiguous()
        Cc = (-Bc * mean - rstd * S1 / N).contiguous()
        gx = _new(tuple(xn.shape), dev)
        _lib.call("evf_chan_affine", _lib.ptr(g), C, _lib.ptr(xn), C, _lib.ptr(A), _lib.ptr(Bc), _lib.ptr(Cc), B, npg, C, _lib.ptr(gx), C)
        return from_nhwc(gx), s2.sum(0).reshape(ctx.wshape), s1.sum(0).reshape(ctx.wshape), None


def group_norm1(x, layer):
    """Apply an nn.GroupNorm(1, C) module (parameter holder under the reference's names) on the GPU."""
    if layer.num_groups != 1:
        raise _lib.EvflowError("only nn.GroupNorm(1, C) (the reference's cells) is implemented")
    return _GroupNorm1.apply(x, layer.weight, layer.bias, float(layer.eps))


def norm2d(x, layer):
    """Apply an nn.BatchNorm2d / nn.InstanceNorm2d module (parameter / buffer holder under the reference's names) on the GPU."""
    instance = isinstance(layer, torch.nn.InstanceNorm2d)
    use_input_stats = layer.training or not layer.track_running_stats
    return _Norm2d.apply(x, layer.weight, layer.bias, layer, instance, use_input_stats)


class _ConvTranspose(torch.autograd.Function):
    """nn.ConvTranspose2d(k, stride 2, padding k/2, output_padding 1) (reference models/submodules.py:104-112): exactly the
    input gradient of the stride-2 convolution with the same weight tensor [Cin][Cout][k][k] -- forward = evf_conv2d_dgrad,
    gradient w.r.t. the input = evf_conv2d_fwd (stride 2), w.r.t. the weight = evf_conv2d_wgrad with the roles of input and
    output gradient swapped."""

    @staticmethod
    def forward(ctx, owner, x, weight, bias):
        xn = to_nhwc(x)
        B, H, W, Ci = xn.shape
        Ci_w, Co, k, _ = weight.shape
        if Ci != Ci_w:
            raise _lib.EvflowError(f"input has {Ci} channels, the transposed layer expects {Ci_w}")
        y = _new((B, 2 * H, 2 * W, Co), xn.device)
        # as a convolution: Cout_conv = Ci (its output = our input), Cin_conv = Co (its input = our output)
        conv_dgrad(xn, _wcache(owner, "wT").get(weight, 1, 0, Co), y, Co, Ci, k, 2)
        if bias is not None:
            y += bias.detach().reshape(1, 1, 1, Co)
        ctx.owner = owner
        ctx.saved = (xn, weight)
        ctx.has_bias = bias is not None
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g_y):
        xn, weight = ctx.saved
        B, H, W, Ci = xn.shape
        _, Co, k, _ = weight.shape
        need = ctx.needs_input_grad  # (owner, x, weight, bias)
        g = to_nhwc(g_y)
        g_x = g_w = g_b = None
        if need[1]:
            gxn = _new((B, H, W, Ci), g.device)
            conv_fwd(g, _wcache(ctx.owner, "w").get(weight, 0, 0, Co), None, gxn, Co, Ci, k, 2)
            g_x = from_nhwc(gxn)
        if need[2]:
            g_w = _new(tuple(weight.shape), g.device)
            conv_wgrad(g, xn, g_w, None, Co, Ci, k, 2, cin_total=Co)
        if ctx.has_bias and need[3]:
            g_b = g.sum((0, 1, 2))
        return None, g_x, g_w, g_b


def conv_transpose(owner, x, weight, bias):
    return _ConvTranspose.apply(owner, x, weight, bias)


# ---------------------------------------------------------------------------
# leaky state mix of ConvLeaky / ConvLeakyRecurrent (submodules.py:545-554, :488-499)
# ---------------------------------------------------------------------------
class _LeakyMix(torch.autograd.Function):
    """(cur, prev, residual, leak) -> act(mix) and mix = prev * sigmoid(leak) + (1 - sigmoid(leak)) * (cur + residual);
    with act None the one tensor `mix` is returned."""

    @staticmethod
    def forward(ctx, cur, prev, residual, leak, act):
        ctx.set_materialize_grads(False)
        cn = to_nhwc(cur)
        B, H, W, C = cn.shape
        pn = to_nhwc(prev) if prev is not None else None
        rn = to_nhwc(residual) if residual is not None else None
        lk = leak.detach().reshape(-1).contiguous()
        mix = _new((B, H, W, C), cn.device)
        out = _new((B, H, W, C), cn.device) if act != 0 else None
        _lib.call("evf_leaky_fwd", _lib.ptr(cn), _lib.ptr(pn), _lib.ptr(rn), _lib.ptr(lk), act, B * H * W, C, _lib.ptr(mix),
                  _lib.ptr(out))
        ctx.act, ctx.leak = act, leak
        ctx.saved = (mix, pn, lk)
        ctx.has = (prev is not None, residual is not None)
        if act == 0:
            return from_nhwc(mix)
        return from_nhwc(out), from_nhwc(mix)

    @staticmethod
    def backward(ctx, *grads):
        mix, pn, lk = ctx.saved
        g_out, g_mix = (None, grads[0]) if ctx.act == 0 else grads
        need = ctx.needs_input_grad  # (cur, prev, residual, leak, act)
        if g_out is None and g_mix is None:
            return None, None, None, None, None
        B, H, W, C = mix.shape
        gon = to_nhwc(g_out) if g_out is not None else None
        gmn = to_nhwc(g_mix) if g_mix is not None else None
        g_cur = _new((B, H, W, C), mix.device)
        g_prev = _new((B, H, W, C), mix.device) if (ctx.has[0] and need[1]) else None
        g_leak = d_leak = None
        if need[3]:
            d_leak = bound_grad(ctx.leak)
            g_leak = d_leak.view(-1) if d_leak is not None else _lib.zeros(C, dtype=torch.float32, device=mix.device)
        _lib.call("evf_leaky_bwd", _lib.ptr(gon), _lib.ptr(gmn), _lib.ptr(mix), _lib.ptr(pn), _lib.ptr(lk), ctx.act, B * H * W, C,
                  _lib.ptr(g_cur), _lib.ptr(g_prev), _lib.ptr(g_leak))
        gc = from_nhwc(g_cur)
        return (gc if need[0] else None, from_nhwc(g_prev) if g_prev is not None else None,
                gc if (ctx.has[1] and need[2]) else None,
                g_leak.view(ctx.leak.shape) if (g_leak is not None and d_leak is None) else None, None)


def leaky_mix(cur, prev, residual, leak, activation):
    """-> (act(mix), mix); the same tensor twice when activation is None."""
    if activation not in ACT_ID:
        raise NotImplementedError(f"leaky cell activation {activation!r} has no HIP kernel (tanh/sigmoid/relu/None)")
    res = residual if torch.is_tensor(residual) else None
    if not torch.is_tensor(residual) and residual != 0:
        raise _lib.EvflowError("residual must be a tensor or 0")
    r = _LeakyMix.apply(cur, prev, res, leak, ACT_ID[activation])
    return (r, r) if ACT_ID[activation] == 0 else r


# ---------------------------------------------------------------------------
# ConvGRU (submodules.py:400-418): the cat([x, h]) convolutions are evaluated
# as conv(x, W[:, :Cx]) + conv(h, W[:, Cx:]) -- no concatenated copy
# ---------------------------------------------------------------------------
class _ConvGRU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, h, wr, br, wu, bu, wo, bo):
        xn = to_nhwc(x)
        B, H, W, Cx = xn.shape
        Ch, k = wr.shape[0], wr.shape[2]
        hn = to_nhwc(h) if h is not None else None
        dev = xn.device
        n = B * H * W * Ch

        def gate(name, w, b, hsrc):
            y = _new((B, H, W, Ch), dev)
            conv_fwd(xn, _wcache(owner, name + "x").get(w, 0, 0, Cx), b.detach().contiguous(), y, Cx, Ch, k, 1)
            if hsrc is not None:
                conv_fwd(hsrc, _wcache(owner, name + "h").get(w, 0, Cx, Ch), None, y, Ch, Ch, k, 1, accumulate=1)
            return y

        cu, cr = gate("u", wu, bu, hn), gate("r", wr, br, hn)
        u, r, hr = _new((B, H, W, Ch), dev), _new((B, H, W, Ch), dev), _new((B, H, W, Ch), dev)
        _lib.call("evf_gru_gates_fwd", _lib.ptr(cu), _lib.ptr(cr), _lib.ptr(hn), n, _lib.ptr(u), _lib.ptr(r), _lib.ptr(hr))
        co = gate("o", wo, bo, hr if hn is not None else None)
        o, new = _new((B, H, W, Ch), dev), _new((B, H, W, Ch), dev)
        _lib.call("evf_gru_out_fwd", _lib.ptr(co), _lib.ptr(hn), _lib.ptr(u), n, _lib.ptr(o), _lib.ptr(new))
        ctx.owner = owner
        ctx.biases = (br, bu, bo)
        ctx.saved = (xn, hn, u, r, hr, o, wr, wu, wo)
        return from_nhwc(new)

    @staticmethod
    def backward(ctx, g_new):
        owner = ctx.owner
        xn, hn, u, r, hr, o, wr, wu, wo = ctx.saved
        B, H, W, Cx = xn.shape
        Ch, k = wr.shape[0], wr.shape[2]
        dev = xn.device
        n = B * H * W * Ch
        need = ctx.needs_input_grad  # (owner, x, h, wr, br, wu, bu, wo, bo)
        g = to_nhwc(g_new)
        g_co, g_cu, g_h = _new((B, H, W, Ch), dev), _new((B, H, W, Ch), dev), _new((B, H, W, Ch), dev)
        _lib.call("evf_gru_out_bwd", _lib.ptr(g), _lib.ptr(hn), _lib.ptr(u), _lib.ptr(o), n, _lib.ptr(g_co), _lib.ptr(g_cu),
                  _lib.ptr(g_h))
        g_cr = _new((B, H, W, Ch), dev)
        if hn is not None:
            g_hr = _new((B, H, W, Ch), dev)
            conv_dgrad(g_co, _wcache(owner, "ohT").get(wo, 1, Cx, Ch), g_hr, Ch, Ch, k, 1)
            _lib.call("evf_gru_gates_bwd", _lib.ptr(g_hr), _lib.ptr(hn), _lib.ptr(r), n, _lib.ptr(g_cr), _lib.ptr(g_h))
        else:
            _lib.zero_(g_cr)

        def wgrads(w, b, gate_g, hsrc):
            d_w, d_b = bound_grad(w), bound_grad(b)
            direct = d_w is not None and d_b is not None
            g_w = d_w if direct else _lib.zeros(tuple(w.shape), dtype=torch.float32, device=dev)
            g_b = d_b if direct else _lib.zeros((Ch,), dtype=torch.float32, device=dev)
            conv_wgrad(xn, gate_g, g_w, g_b, Cx, Ch, k, 1, cin_total=Cx + Ch, cin_off=0, accumulate=1)
            if hsrc is not None:
                conv_wgrad(hsrc, gate_g, g_w, None, Ch, Ch, k, 1, cin_total=Cx + Ch, cin_off=Cx, accumulate=1)
            return (None, None) if direct else (g_w, g_b)

        g_wo, g_bo = wgrads(wo, ctx.biases[2], g_co, hr if hn is not None else None)
        g_wu, g_bu = wgrads(wu, ctx.biases[1], g_cu, hn)
        g_wr, g_br = wgrads(wr, ctx.biases[0], g_cr, hn)
        g_x = g_hh = None
        if need[1]:
            g_xn = _new((B, H, W, Cx), dev)
            conv_dgrad(g_co, _wcache(owner, "oxT").get(wo, 1, 0, Cx), g_xn, Cx, Ch, k, 1)
            conv_dgrad(g_cu, _wcache(owner, "uxT").get(wu, 1, 0, Cx), g_xn, Cx, Ch, k, 1, accumulate=1)
            conv_dgrad(g_cr, _wcache(owner, "rxT").get(wr, 1, 0, Cx), g_xn, Cx, Ch, k, 1, accumulate=1)
            g_x = from_nhwc(g_xn)
        if hn is not None and need[2]:
            conv_dgrad(g_cu, _wcache(owner, "uhT").get(wu, 1, Cx, Ch), g_h, Ch, Ch, k, 1, accumulate=1)
            conv_dgrad(g_cr, _wcache(owner, "rhT").get(wr, 1, Cx, Ch), g_h, Ch, Ch, k, 1, accumulate=1)
            g_hh = from_nhwc(g_h)
        return None, g_x, g_hh, g_wr, g_br, g_wu, g_bu, g_wo, g_bo


def conv_gru(owner, x, h):
    return _ConvGRU.apply(owner, x, h, owner.reset_gate.weight, owner.reset_gate.bias, owner.update_gate.weight,
                          owner.update_gate.bias, owner.out_gate.weight, owner.out_gate.bias)


# ---------------------------------------------------------------------------
# ConvLSTM (submodules.py:335-374): Gates(cat([x, hidden])) = conv(x, W[:, :Cx]) + conv(hidden, W[:, Cx:])
# ---------------------------------------------------------------------------
class _ConvLSTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, x, hidden, cell, w, b):
        ctx.set_materialize_grads(False)
        xn = to_nhwc(x)
        B, H, W, Cx = xn.shape
        Ch, k = w.shape[0] // 4, w.shape[2]
        hn = to_nhwc(hidden) if hidden is not None else None
        cn = to_nhwc(cell) if cell is not None else None
        dev = xn.device
        gates = _new((B, H, W, 4 * Ch), dev)
        conv_fwd(xn, _wcache(owner, "x").get(w, 0, 0, Cx), b.detach().contiguous(), gates, Cx, 4 * Ch, k, 1)
        if hn is not None:
            conv_fwd(hn, _wcache(owner, "h").get(w, 0, Cx, Ch), None, gates, Ch, 4 * Ch, k, 1, accumulate=1)
        new_c, new_h = _new((B, H, W, Ch), dev), _new((B, H, W, Ch), dev)
        _lib.call("evf_lstm_fwd", _lib.ptr(gates), _lib.ptr(cn), B * H * W, Ch, _lib.ptr(new_c), _lib.ptr(new_h))
        ctx.owner, ctx.bias = owner, b
        ctx.saved = (xn, hn, cn, gates, new_c, w)
        return from_nhwc(new_h), from_nhwc(new_c)

    @staticmethod
    def backward(ctx, g_h, g_c):
        owner = ctx.owner
        xn, hn, cn, gates, new_c, w = ctx.saved
        if g_h is None and g_c is None:
            return (None,) * 6
        B, H, W, Cx = xn.shape
        Ch, k = w.shape[0] // 4, w.shape[2]
        dev = xn.device
        need = ctx.needs_input_grad  # (owner, x, hidden, cell, w, b)
        ghn = to_nhwc(g_h) if g_h is not None else None
        gcn = to_nhwc(g_c) if g_c is not None else None
        g_gates = _new((B, H, W, 4 * Ch), dev)
        g_pc = _new((B, H, W, Ch), dev) if (cn is not None and need[3]) else None
        _lib.call("evf_lstm_bwd", _lib.ptr(ghn), _lib.ptr(gcn), _lib.ptr(gates), _lib.ptr(new_c), _lib.ptr(cn), B * H * W, Ch,
                  _lib.ptr(g_gates), _lib.ptr(g_pc))
        g_w = g_b = None
        if need[4] or need[5]:
            d_w, d_b = bound_grad(w), bound_grad(ctx.bias)
            direct = d_w is not None and d_b is not None
            g_w = d_w if direct else _lib.zeros(tuple(w.shape), dtype=torch.float32, device=dev)
            g_b = d_b if direct else _lib.zeros((4 * Ch,), dtype=torch.float32, device=dev)
            conv_wgrad(xn, g_gates, g_w, g_b, Cx, 4 * Ch, k, 1, cin_total=Cx + Ch, cin_off=0, accumulate=1)
            if hn is not None:
                conv_wgrad(hn, g_gates, g_w, None, Ch, 4 * Ch, k, 1, cin_total=Cx + Ch, cin_off=Cx, accumulate=1)
            if direct:
                g_w = g_b = None
        g_x = g_hid = None
        if need[1]:
            g_xn = _new((B, H, W, Cx), dev)
            conv_dgrad(g_gates, _wcache(owner, "xT").get(w, 1, 0, Cx), g_xn, Cx, 4 * Ch, k, 1)
            g_x = from_nhwc(g_xn)
        if hn is not None and need[2]:
            g_hn = _new((B, H, W, Ch), dev)
            conv_dgrad(g_gates, _wcache(owner, "hT").get(w, 1, Cx, Ch), g_hn, Ch, 4 * Ch, k, 1)
            g_hid = from_nhwc(g_hn)
        return None, g_x, g_hid, from_nhwc(g_pc) if g_pc is not None else None, g_w, g_b


def conv_lstm(owner, x, hidden, cell):
    """-> (hidden', cell')"""
    return _ConvLSTM.apply(owner, x, hidden, cell, owner.Gates.weight, owner.Gates.bias)


# ---------------------------------------------------------------------------
# x1 + x2 (skip_sum, model_util.py:22-27)
# ---------------------------------------------------------------------------
class _Concat(torch.autograd.Function):
    """torch.cat(parts, 1) of logical [B,C,H,W] tensors (+ `pad` trailing zero channels) into one NHWC activation: one
    kernel, every float written once (torch's cat of channels_last views takes its generic strided path, and an NCHW
    part makes the whole result NCHW -- a layout conversion of the largest activations of the network)."""

    @staticmethod
    def forward(ctx, pad, *parts):
        ps = [to_nhwc(p) for p in parts]
        B, H, W = ps[0].shape[0], ps[0].shape[1], ps[0].shape[2]
        Cs = [p.shape[3] for p in ps] + ([pad] if pad else [])
        out = _new((B, H, W, sum(Cs)), ps[0].device)
        n = len(Cs)
        src = (ctypes.c_void_p * n)(*([_lib.ptr_strided(p) for p in ps] + ([None] if pad else [])))
        _lib.call("evf_concat_channels", src, (ctypes.c_int * n)(*Cs), (ctypes.c_int * n)(*([p.stride(2) for p in ps] + ([0] if pad else []))),
                  n, B * H * W, _lib.ptr(out), out.stride(2))
        ctx.Cs = [p.shape[3] for p in ps]
        return from_nhwc(out)

    @staticmethod
    def backward(ctx, g):
        gn = to_nhwc(g)
        outs, off = [], 0
        for c in ctx.Cs:
            outs.append(from_nhwc(gn[..., off:off + c]))
            off += c
        return (None,) + tuple(outs)


def concat_channels(parts, pad=0):
    out = _Concat.apply(int(pad), *parts)
    # provenance: exact from the end of the last untagged part on (the zero padding is exact)
    off, exact_from, bound = 0, 0, 0.0
    for p in parts:
        tag = spike_tag(p)
        if tag is None:
            exact_from = off + p.shape[1]
        else:
            bound = max(bound, tag[0])
            if tag[1] > 0:
                exact_from = off + tag[1]
        off += p.shape[1]
    if exact_from < off:
        set_spike_tag(out, max(bound, 1.0), exact_from)
        out._evf_spike_int = all(getattr(p, "_evf_spike_int", True) for p in parts if spike_tag(p) is not None)
    return out


class _ConcatUp2(torch.autograd.Function):
    """upsample2x_bilinear(concat_channels(parts, pad)) in ONE kernel (evf_concat_up2_fwd): the low-resolution concatenation is
    never written.  Backward: the transposed up-sampling of the whole gradient, then each part's channel slice (views)."""

    @staticmethod
    def forward(ctx, pad, *parts):
        ps = [to_nhwc(p) for p in parts]
        B, H, W = ps[0].shape[0], ps[0].shape[1], ps[0].shape[2]
        Cs = [p.shape[3] for p in ps] + ([pad] if pad else [])
        out = _new((B, 2 * H, 2 * W, sum(Cs)), ps[0].device)
        n = len(Cs)
        src = (ctypes.c_void_p * n)(*([_lib.ptr_strided(p) for p in ps] + ([None] if pad else [])))
        _lib.call("evf_concat_up2_fwd", src, (ctypes.c_int * n)(*Cs), (ctypes.c_int * n)(*([p.stride(2) for p in ps] + ([0] if pad else []))),
                  n, B, H, W, _lib.ptr(out), out.stride(2))
        ctx.Cs = [p.shape[3] for p in ps]
        ctx.geom = (B, H, W, sum(Cs))
        return from_nhwc(out)

    @staticmethod
    def backward(ctx, g_y):
        B, H, W, Ct = ctx.geom
        g = to_nhwc(g_y)
        gx = _new((B, H, W, Ct), g.device)
        _lib.call("evf_upsample2x_bwd", _lib.ptr(g), B, H, W, Ct, _lib.ptr(gx))
        outs, off = [], 0
        for c in ctx.Cs:
            outs.append(from_nhwc(gx[..., off:off + c]))
            off += c
        return (None,) + tuple(outs)


CONCAT_UP2 = os.environ.get("EVF_CONCAT_UP2", "1") != "0"


def concat_up2(parts, pad=0):
    """upsample2x_bilinear(concat_channels(parts, pad)); one kernel when every part has an even channel count and pixel stride."""
    ok = (CONCAT_UP2 and pad % 2 == 0 and all(p.shape[1] % 2 == 0 and p.shape[2:] == parts[0].shape[2:] for p in parts)
          and (sum(p.shape[1] for p in parts) + pad) % 4 == 0)
    if ok:  # the kernel reads channel PAIRS (8-byte loads) through the parts' NHWC views: channel-slice views with an odd pixel
        #     stride or a base that is only 4-byte aligned take the two-kernel form instead of failing with EINVAL
        views = [to_nhwc(p.detach()) for p in parts]
        ok = all(v.stride(2) % 2 == 0 and v.data_ptr() % 8 == 0 for v in views)
    if not ok:
        return upsample2x_bilinear(concat_channels(parts, pad))
    out = _ConcatUp2.apply(int(pad), *parts)
    cat_like = _concat_tag(parts)
    if cat_like is not None and all(getattr(p, "_evf_spike_int", True) for p in parts if spike_tag(p) is not None):
        set_spike_tag(out, cat_like[0], cat_like[1])
        out._evf_spike_int = False
    return out


def _concat_tag(parts):
    """Provenance of a channel concatenation: (bound, exact_from) or None."""
    off, exact_from, bound = 0, 0, 0.0
    for p in parts:
        tag = spike_tag(p)
        if tag is None:
            exact_from = off + p.shape[1]
        else:
            bound = max(bound, tag[0])
            if tag[1] > 0:
                exact_from = off + tag[1]
        off += p.shape[1]
    return (max(bound, 1.0), exact_from) if exact_from < off else None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        an, bn = to_nhwc(a), to_nhwc(b)
        y = _new(tuple(an.shape), an.device)
        _lib.call("evf_act_fwd", 0, _lib.ptr(an), _lib.ptr(bn), y.numel(), _lib.ptr(y))
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    if a.shape != b.shape:
        raise _lib.EvflowError(f"add: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
    return _Add.apply(a, b)


# ---------------------------------------------------------------------------
# up-sampling
# ---------------------------------------------------------------------------
class _Up2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xn = to_nhwc(x)
        B, H, W, Cc = xn.shape
        y = _new((B, 2 * H, 2 * W, Cc), xn.device)
        _lib.call("evf_upsample2x_fwd", _lib.ptr(xn), B, H, W, Cc, _lib.ptr(y))
        ctx.shape = (B, H, W, Cc)
        return from_nhwc(y)

    @staticmethod
    def backward(ctx, g_y):
        B, H, W, Cc = ctx.shape
        g = to_nhwc(g_y)
        gx = _new((B, H, W, Cc), g.device)
        _lib.call("evf_upsample2x_bwd", _lib.ptr(g), B, H, W, Cc, _lib.ptr(gx))
        return from_nhwc(gx)


def upsample2x_bilinear(x):
    """F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)."""
    y = _Up2.apply(x)
    tag = spike_tag(x)
    if tag is not None:  # blends with weights 1, 3, 3, 9 / 16 of multiples of 1 / 16?  No: only of INTEGERS (first-level blends)
        if getattr(x, "_evf_spike_int", True):
            set_spike_tag(y, tag[0], tag[1])
            y._evf_spike_int = False  # (a second blend would need 1 / 256 steps: not tagged again)
    return y


class _UpNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f):
        _lib.require_gpu(x, "nearest up-sampling")
        xc = x.contiguous()
        B, Cc, h, w = xc.shape
        y = _new((B, Cc, h * f, w * f), xc.device)
        _lib.call("evf_upsample_nearest_fwd", _lib.ptr(xc), B * Cc, h, w, f, _lib.ptr(y))
        ctx.geom = (B, Cc, h, w, f)
        return y

    @staticmethod
    def backward(ctx, g_y):
        B, Cc, h, w, f = ctx.geom
        g = g_y.contiguous()
        gx = _new((B, Cc, h, w), g.device)
        _lib.call("evf_upsample_nearest_bwd", _lib.ptr(g), B * Cc, h, w, f, _lib.ptr(gx))
        return gx, None


def upsample_nearest(x, factor):
    """F.interpolate(x, scale_factor=factor) for an integer factor (flow pyramids, model.py:529-539)."""
    f = int(round(factor))
    if abs(f - factor) > 1e-9 or f < 1:
        raise _lib.EvflowError(f"nearest up-sampling: integer factors only, got {factor}")
    if f == 1:
        return x
    return _UpNearest.apply(x, f)
